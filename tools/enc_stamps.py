"""Diagnostic (RSPT_DIAG build only): per-phase cycle counts of k_encode from thread 0's s_memtime stamps.
   RSPT_HIP_LIB=rspt_amd/librspt_hip_diag.so python tools/enc_stamps.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rspt_amd import api, synth
pk = api.new_xdelta_hzr(4, 64, 65536, 3)
d = [synth.synth_batch_native(64, 64, 65536, first_block=64 * s, device="cuda") for s in range(2)]
for i in range(3):
    pk.compress_batch(d[i & 1])
torch.cuda.synchronize()
st = pk.debug_read(7, 6000 * 16 * 8).view(np.uint64).reshape(6000, 16).astype(np.int64)
L = st[:, 6] & 0xFFFFFFFF
ok = (st[:, 0] > 0) & (st[:, 5] > st[:, 0])
names = ["load+chain (rows path)", "emit", "barrier", "crc", "copyout"]
for label, sel in (("dense (L >= 16384)", ok & (L >= 16384)), ("light (L < 16384)", ok & (L < 16384))):
    s = st[sel]
    if not len(s):
        continue
    print("%s: %d blocks, total median %d p90 %d cycles" % (label, len(s), np.median(s[:, 5] - s[:, 0]), np.percentile(s[:, 5] - s[:, 0], 90)))
    if label.startswith("dense"):
        for k, nm in enumerate(names):
            dd = s[:, k + 1] - s[:, k]
            print("   %-24s median %7d  p90 %7d" % (nm, np.median(dd), np.percentile(dd, 90)))
    else:
        for a, b, nm in ((0, 2, "lists: zero+emit"), (2, 3, "barrier"), (3, 4, "crc"), (4, 5, "copyout")):
            dd = s[:, b] - s[:, a]
            print("   %-24s median %7d  p90 %7d" % (nm, np.median(dd), np.percentile(dd, 90)))
o = np.argsort(st[ok][:, 0]); s = st[ok][o]
print("span %d cycles for %d stamped blocks" % (s[:, 5].max() - s[:, 0].min(), len(s)))
