"""Reproducer for an intermittent wrong block CRC found by tools/soak.py (seed 1731: hzr int32 5ch x 116158, a batch of 4):
the same batch compressed over and over, every stream compared with the oracle's; prints which hzr blocks differ and how often."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import soak
from soak import api
from streamtools import parse_stream

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1731
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
keep = []
soak.one_case(seed, keep)
c = keep[0]
print("case:", c["kind"], c["bps"], c["nch"], c["ns"], c["nb0"], len(c["feed"]), "blocks")
pk = api.SignalPacker(c["kind"], c["bps"], c["nch"], c["ns"], c["nb0"])
d_src = torch.from_numpy(np.stack(c["feed"])).cuda()
fails = {}
pattern = ""
for rep in range(reps):
    d_dst, d_sizes = pk.compress_batch(d_src)
    torch.cuda.synchronize()
    sizes = d_sizes.cpu().numpy()
    out = d_dst.cpu().numpy()
    for i, w in enumerate(c["want"]):
        g = out[i, : sizes[i]].tobytes()
        if g != w:
            first = next(q for q in range(min(len(g), len(w))) if g[q] != w[q])
            p = parse_stream(w)
            where = None
            for k, pl in enumerate(p["planes"]):
                for j, (mode, plen, crc, off) in enumerate(pl["blocks"]):
                    if off <= first < off + 7 + plen:
                        where = (i, k, j, mode, plen, first - off)
            fails[where] = fails.get(where, 0) + 1
    pattern += "x" if any(out[i, : sizes[i]].tobytes() != w for i, w in enumerate(c["want"])) else "."
print("reps", reps, "failures (block, plane, hzr block, mode, payload, offset in block): count")
for k, v in sorted(fails.items(), key=lambda kv: -kv[1]):
    print("  ", k, v)
print(pattern[:120])
N = c["nch"] * c["ns"]
nblk = (N + 65535) // 65536
for (b, k, j, mode, plen, offs) in [x for x in fails if x]:
    hb = (b * 4 + k) * nblk + j
    meta = pk.debug_read(4, 4 * nblk * 4 * 16 * len(c["feed"])).view(np.uint32).reshape(-1, 4)
    nz = pk.debug_read(8, 4 * nblk * 4 * 4 * len(c["feed"])).view(np.uint32)
    print("hb", hb, "meta (mode, payload, tree_bits, fill/ntok)", meta[hb].tolist(), "segmask %04x" % nz[hb], "in_size", min(65536, N - j * 65536))
    for jj in range(nblk):
        h2 = (b * 4 + k) * nblk + jj
        print("   plane", k, "hzr block", jj, meta[h2].tolist(), "%04x" % nz[h2])
pk.close()
