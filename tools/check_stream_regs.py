#!/usr/bin/env python3
"""Static check of k_tile_stream's hand-issued loads in the device assembly.

The streaming loop issues global_load_dword (and the dirty-bit s_load_dwordx4) from inline asm, which the compiler does not
track.  Between such a load and the explicit s_waitcnt that covers it no instruction may touch the destination register --
neither read it (the data is not there yet) nor write it (the late load would overwrite the new value).  This walks the
CONTROL-FLOW GRAPH of every k_tile_stream instantiation: basic blocks from the labels and branches, a depth-first search
over (block, in-flight vector loads in issue order, in-flight scalar destinations, known loop-exit flags), every state
visited once.  vmcnt(N) inside the asm retires all but the N youngest loads (in-order return); a compiler-inserted
s_waitcnt with vmcnt(0) / lgkmcnt(0) retires everything of its kind.  The only piece of value tracking: a 64-bit scalar pair
set from an immediate (0 / -1) and tested through `s_and_b64 vcc, exec, pair` decides the branch behind it -- that is how the
compiler lowers `if (turn(r)) break;`, and following the other arm would re-enter the loop with the ring of register sets
one step out of phase.

usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -Iinclude -o /tmp/rspt.s rspt_amd/csrc/rspt_hip.hip
       python tools/check_stream_regs.py /tmp/rspt.s [--path]
"""
import re, sys
sys.setrecursionlimit(100000)

def regs_of(tok, pre):
    out = []
    for m in re.finditer(r"\b%s(\d+)\b|\b%s\[(\d+):(\d+)\]" % (pre, pre), tok):
        if m.group(1) is not None: out.append(int(m.group(1)))
        else: out.extend(range(int(m.group(2)), int(m.group(3)) + 1))
    return out

def parse(lines):
    """-> blocks: list of dict(label, ins=[(lineno, text, in_asm)], succ=[labels or None for fallthrough])"""
    blocks = []; cur = dict(label=None, ins=[]); in_asm = False
    def close():
        nonlocal cur
        blocks.append(cur); cur = dict(label=None, ins=[])
    for ln, raw in lines:
        t = raw.strip()
        if t.startswith(";;#ASMSTART"): in_asm = True; continue
        if t.startswith(";;#ASMEND"): in_asm = False; continue
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            if cur["ins"] or cur["label"] is not None: close()
            cur["label"] = m.group(1); continue
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"): continue
        code = t.split(";")[0].strip()
        if not code: continue
        cur["ins"].append((ln, code, in_asm))
        if re.match(r"s_(c?branch|endpgm|setpc)", code) and not in_asm:
            close()
    if cur["ins"] or cur["label"] is not None: close()
    return blocks

def check_cfg(lines, name):
    blocks = parse(lines)
    idx = {b["label"]: i for i, b in enumerate(blocks) if b["label"]}
    bad = {}; seen = {}; nload = nwait = 0; first = None
    stack = [((0, (), frozenset(), frozenset(), None), None)]
    while stack:
        key, par = stack.pop()
        bi, pending, spending, consts, vcc = key
        if bi >= len(blocks): continue
        if key in seen: continue
        seen[key] = par
        if len(seen) > 400000:
            print(name, "state space too large"); return 1
        pend = list(pending); spend = set(spending); cst = dict(consts)
        succ = None
        for ln, code, in_asm in blocks[bi]["ins"]:
            if in_asm:
                m = re.match(r"global_load_(?:dword|sshort|sbyte) v(\d+),", code)
                if m: pend.append(int(m.group(1))); continue
                m = re.match(r"s_load_dwordx4 s\[(\d+):(\d+)\],", code)
                if m: spend |= set(range(int(m.group(1)), int(m.group(2)) + 1)); continue
                m = re.match(r"s_waitcnt lgkmcnt\(0\)", code)
                if m: spend = set(); continue
                m = re.match(r"s_waitcnt vmcnt\((\d+)\)", code)
                if m:
                    n = int(m.group(1)); pend = pend[len(pend) - n:] if n and len(pend) > n else ([] if n == 0 else pend); continue
                continue
            m = re.match(r"s_waitcnt (.*)", code)
            if m:
                if "vmcnt(0)" in m.group(1): pend = []
                if "lgkmcnt(0)" in m.group(1): spend = set()
                continue
            mb = re.match(r"s_branch (\S+)", code)
            if mb: succ = [mb.group(1)]; break
            mc = re.match(r"s_cbranch_(\w+) (\S+)", code)
            if mc:
                # a loop-exit flag set from an immediate decides the branch behind it (`if (turn(r)) break;`): following
                # the other arm would re-enter the loop with the ring one set out of step
                if mc.group(1) == "vccnz" and vcc is not None: succ = [mc.group(2)] if vcc else [None]
                elif mc.group(1) == "vccz" and vcc is not None: succ = [None] if vcc else [mc.group(2)]
                else: succ = [mc.group(2), None]
                break
            if code.startswith("s_endpgm"): succ = []; break
            ops = code.split(None, 1)[1] if " " in code else ""
            # minimal constant tracking: s_mov_b64 of 0 / -1 into a pair, and vcc = exec & / &~ such a pair
            mm = re.match(r"s_mov_b64 s\[(\d+):(\d+)\], (-1|0)$", code)
            ma = re.match(r"s_(and|andn2)_b64 vcc, exec, s\[(\d+):(\d+)\]$", code)
            dst = ops.split(",")[0].strip() if ops else ""
            if mm:
                cst[(int(mm.group(1)), int(mm.group(2)))] = mm.group(3) == "-1"
            elif ma and (int(ma.group(2)), int(ma.group(3))) in cst:
                v = cst[(int(ma.group(2)), int(ma.group(3)))]
                vcc = v if ma.group(1) == "and" else (not v)
            else:
                wr = set(regs_of(dst, "s"))
                for k in [k for k in cst if wr & set(range(k[0], k[1] + 1))]: del cst[k]
                if dst.startswith("vcc") or code.startswith("v_cmp") and "vcc" in dst or code.split()[0].endswith("_e32") and code.startswith("v_cmp"): vcc = None
            if spend:
                hit = set(regs_of(ops, "s")) & spend
                if hit:
                    bad[ln] = "SGPR %s: %s" % (sorted(hit), code)
                    if first is None: first = key
            hit = set(regs_of(ops, "v")) & set(pend)
            if hit: bad[ln] = "%s: %s" % (sorted(hit), code)
        st = (tuple(pend), frozenset(spend), frozenset(cst.items()), vcc)
        if succ is None: succ = [None]
        for s in succ:
            if s is None: stack.append(((bi + 1,) + st, key))
            elif s in idx: stack.append(((idx[s],) + st, key))
    if first is not None and "--path" in sys.argv:
        p = []; k = first
        while k is not None: p.append((k[0], blocks[k[0]]["label"], len(k[2]))); k = seen[k]
        print("path to the first SGPR flag:", p[::-1][-40:])
    for ln in sorted(bad)[:12]: print("%s: line %d touches in-flight %s" % (name, ln, bad[ln]))
    print("%s: %d states, %d suspicious accesses" % (name, len(seen), len(bad)))
    return len(bad)

def main(path):
    text = open(path).read().split("\n")
    i = 0; tot = 0
    while i < len(text):
        m = re.match(r"^(_ZN4rspt13k_tile_stream\w+):", text[i])
        if m:
            j = i
            while j < len(text) and "s_endpgm" not in text[j]: j += 1
            tot += check_cfg([(k + 1, text[k]) for k in range(i + 1, j + 1)], m.group(1)[:40])
            i = j
        i += 1
    print("total", tot)
    return 1 if tot else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))

