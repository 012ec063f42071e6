import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
os.environ["RSPT_ABLATE"] = "256"
from rspt_amd import api, synth
pk = api.new_xdelta_hzr(4, 64, 65536, 3)
d = synth.synth_batch_native(64, 64, 65536, device="cuda")
for _ in range(3):
    pk.compress_batch(d)
torch.cuda.synchronize()
raw = pk.debug_read(7, (512 * 16 * 8 + 2 * 16384) * 8).view(np.uint64)[65536:].astype(np.int64).reshape(16384, 2)
hb = np.arange(16384)
plane = (hb // 64) % 4
t0 = raw[:, 0].min()
for name, sel in (("dense(plane0)", plane == 0), ("sparse(plane1,2)", (plane == 1) | (plane == 2))):
    s = raw[sel]
    ok = s[:, 1] > 0
    s = s[ok] - t0
    dur = (s[:, 1] - s[:, 0]) / 100.0  # us
    print(name, "n", len(s), "dur us median %.1f mean %.1f max %.1f" % (np.median(dur), dur.mean(), dur.max()), "first start %.1f last end %.1f us" % (s[:, 0].min() / 100, s[:, 1].max() / 100))
    ev = np.concatenate([np.stack([s[:, 0], np.ones(len(s))], 1), np.stack([s[:, 1], -np.ones(len(s))], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    conc = np.cumsum(ev[:, 1])
    print("   max concurrent", int(conc.max()), " time-avg concurrent %.0f" % ((dur.sum()) / ((s[:, 1].max() - s[:, 0].min()) / 100.0)))
allw = raw[(plane < 3) & (raw[:, 1] > 0)] - t0
T = allw[:, 1].max()
print("concurrency over time (20 buckets), all non-trivial WGs:")
row = []
for i in range(20):
    a, b = T * i / 20, T * (i + 1) / 20
    ov = np.clip(np.minimum(allw[:, 1], b) - np.maximum(allw[:, 0], a), 0, None).sum() / (b - a)
    row.append(int(ov))
print(row)
# start times of dense WGs in dispatch order: is dispatch in order / stalled?
d0 = raw[plane == 0][:, 0] - t0
print("dense WG start times (us) every 256th:", (d0[::256] / 100).astype(int))
