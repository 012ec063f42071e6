"""How fast does a Huffman decoder that starts at a wrong bit fall into step?  (CPU simulation on one dense hzr block.)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from oracle.oracle import Oracle
from rspt_amd import synth
from streamtools import parse_stream
o = Oracle()
x = synth.synth_native(64, 65536, block_index=0).numpy().reshape(-1)
s = o.packer("xdelta_hzr", 4, 64, 65536, 3).compress(x)
ps = parse_stream(s)
mode, plen, crc, off = ps["planes"][0]["blocks"][5]
pay = np.frombuffer(s[off + 7: off + 7 + plen], dtype=np.uint8)
bits = np.unpackbits(pay, bitorder="little")
# tree
pos = 0; codes = {}
def rec(code, depth):
    global pos
    if bits[pos] == 1:
        sym = int(sum(int(bits[pos + 1 + i]) << i for i in range(9))); pos += 10
        codes[(code, depth)] = sym
    else:
        pos += 1
        rec(code, depth + 1)
        rec(code | (1 << depth), depth + 1)
rec(0, 0)
code0 = pos
tab = {}
for (c, d), sym in codes.items(): tab[(d, c)] = sym
maxd = max(d for d, _ in tab)
EB = {257: 2, 258: 4, 259: 8, 260: 14}
def decode_from(p, stop):
    """boundaries visited from p until >= stop"""
    vis = []
    n = len(bits)
    while p < stop and p < n:
        vis.append(p)
        c = 0; d = 0; sym = None
        while d < maxd and p + d < n:
            c |= int(bits[p + d]) << d; d += 1
            if (d, c) in tab: sym = tab[(d, c)]; break
        if sym is None: return vis, None
        p += d + EB.get(sym, 0)
    return vis, p
true, _ = decode_from(code0, len(bits))
trueset = set(true)
import random
random.seed(1)
dist = []
for t in range(300):
    p = random.randrange(code0 + 100, len(bits) - 3000)
    vis, e = decode_from(p, p + 2500)
    k = next((i for i, q in enumerate(vis) if q in trueset), None)
    dist.append((vis[k] - p) if k is not None else 99999)
dist = np.array(dist)
print("payload bits", len(bits) - code0, "tokens", len(true), "avg bits/token %.2f" % ((len(bits) - code0) / len(true)))
print("sync distance (bits): median %d  p90 %d  p99 %d  max %d  unsynced %d" % (np.median(dist), np.percentile(dist, 90), np.percentile(dist, 99), dist.max(), (dist == 99999).sum()))
