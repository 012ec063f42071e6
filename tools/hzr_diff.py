#!/usr/bin/env python3
"""Diagnostic: compress a KAT / packer case on the GPU and with the oracle, find the first hzr block whose bytes differ,
decode both token streams and print where they part.

    python tools/hzr_diff.py kat <name>          raw hzr known-answer input through a 1ch x 8-bit hzr packer
    python tools/hzr_diff.py case <name>         a packer case of tests/cases.py
"""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

EXTRA = {257: (2, 3), 258: (4, 7), 259: (8, 23), 260: (14, 279)}


class Bits:
    def __init__(self, data):
        self.d, self.p = data, 0

    def get(self, n):
        v = 0
        for i in range(n):
            byte = self.d[(self.p + i) >> 3] if ((self.p + i) >> 3) < len(self.d) else 0
            v |= ((byte >> ((self.p + i) & 7)) & 1) << i
        self.p += n
        return v


def read_tree(b):
    if b.get(1):
        return b.get(9)
    a = read_tree(b)
    c = read_tree(b)
    return (a, c)


def tokens(payload, in_size):
    b = Bits(payload)
    tree = read_tree(b)
    out, produced = [], 0
    while produced < in_size and b.p < 8 * len(payload):
        n, start = tree, b.p
        while isinstance(n, tuple):
            n = n[b.get(1)]
        if n == 0:
            z = 1
        elif n == 256:
            z = 2
        elif n in EXTRA:
            z = EXTRA[n][1] + b.get(EXTRA[n][0])
        else:
            z = 0
        out.append((start, n, z, produced))
        produced += z if z else 1
    return tree, out


def blocks(stream, hdr_len=0):
    from streamtools import parse_stream

    return parse_stream(stream, hdr_len)


def main():
    import cases
    from oracle.oracle import Oracle
    from rspt_amd import api

    orc = Oracle()
    kind, name = sys.argv[1], sys.argv[2]
    if kind == "kat":
        data = cases.hzr_kat_inputs()[name]
        c = dict(kind="hzr", bps=1, nch=1, ns=data.size, nb=4, data=data)
    else:
        c = {x["name"]: x for x in cases.packer_cases()}[name]
    pk = api.SignalPacker(c["kind"], c["bps"], c["nch"], c["ns"], c["nb"])
    got = pk.compress(c["data"], dst_max_len=pk.max_compressed_size)
    want = orc.packer(c["kind"], c["bps"], c["nch"], c["ns"], c["nb"]).compress(c["data"])
    print("sizes got %d want %d equal %s" % (len(got), len(want), got == want))
    hdr = 3 * c["nch"] if c["kind"] in ("dct", "hadamard") else 0
    pg, pw = blocks(got, hdr), blocks(want, hdr)
    N = c["nch"] * c["ns"]
    for k, (a, b) in enumerate(zip(pg["planes"], pw["planes"])):
        for j, (ba, bb) in enumerate(zip(a["blocks"], b["blocks"])):
            ga = got[ba[3] : ba[3] + 7 + ba[1]]
            wa = want[bb[3] : bb[3] + 7 + bb[1]]
            if ga == wa:
                continue
            in_size = min(65536, N - j * 65536)
            print("plane %d block %d differs: got (mode %d, payload %d) want (mode %d, payload %d), in_size %d" % (k, j, ba[0], ba[1], bb[0], bb[1], in_size))
            if ba[0] == 1 and bb[0] == 1:
                tg, tokg = tokens(ga[7:], in_size)
                tw, tokw = tokens(wa[7:], in_size)
                print("  trees equal:", tg == tw, " tokens got %d want %d" % (len(tokg), len(tokw)))
                for i, (x, y) in enumerate(zip(tokg, tokw)):
                    if x != y:
                        print("  first differing token #%d: got (bit %d sym %d z %d at byte %d) want (bit %d sym %d z %d at byte %d)" % ((i,) + x + y))
                        print("   wave %d row %d lane %d byte %d" % (y[3] >> 12, (y[3] >> 10) & 3, (y[3] >> 4) & 63, y[3] & 15))
                        for q in range(max(0, i - 3), min(len(tokw), i + 4)):
                            print("     #%d got %s want %s" % (q, tokg[q] if q < len(tokg) else None, tokw[q]))
                        break
            return
    print("no differing block")


if __name__ == "__main__":
    main()
