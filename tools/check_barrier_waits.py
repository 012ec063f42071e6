#!/usr/bin/env python3
"""Static check over the device assembly: no wave reaches an s_barrier with an LDS store of its own still in flight.

__syncthreads() has to publish the LDS stores issued in front of it: the compiler emits `s_waitcnt lgkmcnt(0)` before the
`s_barrier`.  hipcc (ROCm 7.2) was seen to leave that wait out where a thread-0-only block of LDS stores reached the barrier
over a loop back-edge (the barrier being the first instruction of the loop header): waves on the other SIMD pair then read the
old values now and then (profiles/r03_notes.md, decoder).  This walks every kernel's control-flow graph with one bit of state --
"an LDS store was issued since the last lgkmcnt(0) wait" -- and reports the barriers that can be reached with it set.

usage: check_barrier_waits.py <file.s>      prints one line per kernel: "<name>: <barriers> barriers, <bad> unpublished"
"""
import re
import sys

LDS_STORE = re.compile(r"^\s*ds_(write|add|sub|rsub|inc|dec|min|max|and|or|xor|mskor|wrxchg|cmpst|wrap|append|consume|swizzle_wr)")
LABEL = re.compile(r"^(\.LBB\S+|_Z\S+):")
BRANCH = re.compile(r"^\s*(s_branch|s_cbranch_\w+)\s+(\.LBB\S+)")


def kernels(lines):
    name, body = None, []
    for ln in lines:
        m = re.match(r"^(_Z\S+):\s*; @", ln)
        if m:
            name, body = m.group(1), []
            continue
        if name is not None:
            if ln.startswith(".Lfunc_end"):
                yield name, body
                name = None
            else:
                body.append(ln)


def check(body):
    # basic blocks: (label, [instructions]); successors by label / fallthrough
    blocks, cur = [], ["<entry>", []]
    for ln in body:
        t = ln.split(";")[0].rstrip()
        if not t.strip():
            continue
        m = LABEL.match(t)
        if m:
            blocks.append(cur)
            cur = [m.group(1), []]
            continue
        if t.startswith("\t.") or t.startswith(".") or t.strip().startswith("."):
            continue  # directives
        cur[1].append(t.strip())
        if BRANCH.match(t):  # proper basic blocks: a branch ends one
            blocks.append(cur)
            cur = ["%s+%d" % (cur[0].split("+")[0], len(blocks)), []]
    blocks.append(cur)
    index = {b[0]: i for i, b in enumerate(blocks)}
    succ = []
    for i, (lab, ins) in enumerate(blocks):
        s = set()
        fall = True
        for t in ins:
            m = BRANCH.match(t)
            if m and m.group(2) in index:
                s.add(index[m.group(2)])
                if m.group(1) == "s_branch":
                    fall = False
            if t.startswith("s_endpgm") or t.startswith("s_setpc"):
                fall = False
        if fall and i + 1 < len(blocks):
            s.add(i + 1)
        succ.append(s)
    state_in = [False] * len(blocks)
    why = [None] * len(blocks)  # the predecessor that brought the pending store in (for the witness path)
    work = list(range(len(blocks)))
    bad = set()
    nbar = sum(1 for _, ins in blocks for t in ins if t.startswith("s_barrier"))
    while work:
        i = work.pop()
        pend = state_in[i]
        for k, t in enumerate(blocks[i][1]):
            if t.startswith("s_waitcnt") and "lgkmcnt(0)" in t:
                pend = False
            elif LDS_STORE.match(t):
                pend = True
            elif t.startswith("s_barrier") and pend:
                bad.add((blocks[i][0], k))
        for j in succ[i]:
            if pend and not state_in[j]:
                state_in[j] = True
                why[j] = i
                work.append(j)
    wit = []
    for lab, k in sorted(bad):
        i, path = index.get(lab, None), []
        while i is not None and len(path) < 12:
            path.append(blocks[i][0])
            i = why[i]
        wit.append((lab, k, list(reversed(path))))
    return nbar, wit


def main(path):
    total_bad = 0
    for name, body in kernels(open(path).read().split("\n")):
        nbar, bad = check(body)
        if nbar:
            print("%s: %d barriers, %d unpublished%s" % (name, nbar, len(bad), "".join("  [%s #%d via %s]" % (b[0], b[1], " > ".join(b[2])) for b in bad[:4])))
        total_bad += len(bad)
    return 1 if total_bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
