import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
WIN = int(sys.argv[1]) if len(sys.argv) > 1 else 16
os.environ["RSPT_ABLATE"] = str(128 + (WIN << 16))
from rspt_amd import api, synth
B = 64
pk = api.new_xdelta_hzr(4, 64, 65536, 3)
d = synth.synth_batch_native(B, 64, 65536, device="cuda")
for _ in range(3):
    out = pk.compress_batch(d)
torch.cuda.synchronize()
st = pk.debug_read(7, 512 * 16 * 8 * 8).view(np.uint64).reshape(512, 16, 8).astype(np.int64)
names = ["tables+zero+load+chain", "pass1(bits)", "scan", "emit", "barrier", "crc", "copyout"]
for label, sel in (("plane0 blocks (dense)", [i for i in range(512) if (i // 64) % 4 == 0]), ("plane1 blocks (sparse)", [i for i in range(512) if (i // 64) % 4 == 1])):
    s = st[sel]
    ok = s[:, :, 7] > 0
    print(label, "blocks", len(sel), "waves with stamps", ok.sum())
    dur = np.diff(s, axis=2)
    for i, n in enumerate(names):
        v = dur[:, :, i][ok]
        print("   %-24s median %8.0f  mean %8.0f  max %8.0f cycles" % (n, np.median(v), v.mean(), v.max()))
    tot = (s[:, :, 7] - s[:, :, 0])[ok]
    print("   total per wave median %.0f cycles; block span median %.0f" % (np.median(tot), np.median(s[:, :, 7].max(axis=1) - s[:, :, 0].min(axis=1))))
for blk in (64, 0):
    s = st[blk]
    print("block", blk, "per-wave section cycles (rows = waves):")
    print(np.diff(s, axis=1) if False else np.diff(s, axis=1).astype(int))
