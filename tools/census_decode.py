"""Concurrency census of k_dec_block: how many workgroups are in flight (s_memrealtime per workgroup, RSPT_ABLATE set)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RSPT_ABLATE"] = str(1 << 20)
import numpy as np, torch
from rspt_amd import api, synth
B, nch, ns = 64, 64, 65536
dev = torch.device("cuda", 0)
d_src = synth.synth_batch_native(B, nch, ns, device=dev)
pk = api.new_xdelta_hzr(4, nch, ns, 3)
stride = (pk.max_compressed_size + 255) // 256 * 256
dst = torch.empty((B, stride), dtype=torch.uint8, device=dev)
sz = torch.empty(B, dtype=torch.int64, device=dev)
pk.compress_batch(d_src, dst, sz, stride)
out = torch.empty_like(d_src); used = torch.empty(B, dtype=torch.int64, device=dev)
for _ in range(2): pk.decompress_batch(dst, B, stride, out, used)
torch.cuda.synchronize()
raw = pk.debug_read(7, (512 * 16 * 8 + 2 * 16384) * 8).view(np.uint64)[65536:].astype(np.int64).reshape(16384, 2)
hb = np.arange(16384)
plane = (hb // 64) % 4
ok = raw[:, 1] > 0
t0 = raw[ok, 0].min()
for name, sel in (("plane0", plane == 0), ("plane1", plane == 1), ("plane2", plane == 2)):
    s = raw[sel & ok] - t0
    dur = (s[:, 1] - s[:, 0]) / 100.0
    print(name, "n", len(s), "dur us median %.1f mean %.1f max %.1f" % (np.median(dur), dur.mean(), dur.max()),
          "first start %.1f last end %.1f us" % (s[:, 0].min() / 100, s[:, 1].max() / 100),
          " time-avg concurrent %.0f" % (dur.sum() / ((s[:, 1].max() - s[:, 0].min()) / 100.0)))
allw = raw[ok] - t0
T = allw[:, 1].max()
row = []
for i in range(20):
    a, b = T * i / 20, T * (i + 1) / 20
    row.append(int(np.clip(np.minimum(allw[:, 1], b) - np.maximum(allw[:, 0], a), 0, None).sum() / (b - a)))
print("workgroups in flight over time (20 buckets of %.0f us):" % (T / 100 / 20), row)
