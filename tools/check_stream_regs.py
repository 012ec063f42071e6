#!/usr/bin/env python3
"""Static check of k_tile_planes' hand-issued loads in the device assembly.

The streaming loop issues global_load_dword from inline asm, which the compiler does not track.  Between such a load and
the explicit s_waitcnt that covers it no instruction may touch the destination register.  This walks the assembly of
every k_tile_planes instantiation linearly (a conservative approximation: the loop body is straight-line between the
loads and their waits) and reports any access to a register whose hand-issued load may still be in flight.

usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -Iinclude -o /tmp/rspt.s rspt_amd/csrc/rspt_hip.hip
       python tools/check_stream_regs.py /tmp/rspt.s
"""
import re
import sys


def regs_of(tok):
    out = []
    for m in re.finditer(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]", tok):
        if m.group(1) is not None:
            out.append(int(m.group(1)))
        else:
            out.extend(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def sregs_of(tok):
    out = []
    for m in re.finditer(r"\bs(\d+)\b|\bs\[(\d+):(\d+)\]", tok):
        if m.group(1) is not None:
            out.append(int(m.group(1)))
        else:
            out.extend(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def check(lines, name):
    pending = []  # destination registers of hand-issued loads, oldest first
    spending = set()  # destination SGPRs of hand-issued scalar loads (they return out of order: only lgkmcnt(0) clears them)
    in_asm = False
    bad = 0
    nload = nwait = 0
    for ln, raw in lines:
        t = raw.strip()
        if t.startswith(";;#ASMSTART") or t.startswith("; %bb") and False:
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        code = t.split(";")[0].strip()
        if in_asm:
            m = re.match(r"global_load_(?:dword|sshort) v(\d+),", code)
            if m:
                pending.append(int(m.group(1)))
                nload += 1
                continue
            m = re.match(r"s_load_dwordx4 s\[(\d+):(\d+)\],", code)
            if m:
                spending |= set(range(int(m.group(1)), int(m.group(2)) + 1))
                nload += 1
                continue
            if re.match(r"s_waitcnt lgkmcnt\(0\)", code):
                spending = set()
                nwait += 1
                continue
            m = re.match(r"s_waitcnt vmcnt\((\d+)\)", code)
            if m:
                n = int(m.group(1))
                pending = pending[len(pending) - n:] if n and len(pending) > n else ([] if n == 0 else pending)
                nwait += 1
                continue
            continue
        m = re.match(r"s_waitcnt (.*)", code)
        if m:
            if "vmcnt(0)" in m.group(1):
                pending = []
            if "lgkmcnt(0)" in m.group(1):
                spending = set()
            continue
        if spending:
            shit = set(sregs_of(code.split(None, 1)[1] if " " in code else "")) & spending
            if shit:
                bad += 1
                print("%s: line %d touches in-flight SGPR %s: %s" % (name, ln, sorted(shit), code))
        if code.startswith("s_") and "v" not in code.split(None, 1)[-1]:
            continue
        touched = set(regs_of(code.split(None, 1)[1] if " " in code else ""))
        hit = touched & set(pending)
        if hit:
            bad += 1
            print("%s: line %d touches in-flight %s: %s" % (name, ln, sorted(hit), code))
    print("%s: %d hand-issued loads, %d explicit waits, %d suspicious accesses" % (name, nload, nwait, bad))
    return bad


def main(path):
    text = open(path).read().split("\n")
    bad = 0
    i = 0
    while i < len(text):
        m = re.match(r"^(_ZN4rspt13k_tile_stream\w+):", text[i])
        if m:
            j = i
            while j < len(text) and "s_endpgm" not in text[j]:
                j += 1
            bad += check([(k + 1, text[k]) for k in range(i, j)], m.group(1)[:40])
            i = j
        i += 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
