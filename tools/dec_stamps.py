import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RSPT_ABLATE"] = "128"
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
st = pk.debug_read(7, 512 * 8 * 8).view(np.uint64).reshape(512, 8).astype(np.int64)
names = ["frame+stage+zero", "tree + tables", "-", "pre-pass + counting rounds", "writing pass", "totals", "end"]
for label, sel in (("plane0 (dense)", range(0, 64)), ("plane1", range(64, 128))):
    s = st[list(sel)]
    d = np.diff(s, axis=1)
    print(label)
    for i, n in enumerate(names[:6]):
        print("   %-28s median %9.0f mean %9.0f max %9.0f" % (n, np.median(d[:, i]), d[:, i].mean(), d[:, i].max()))
    print("   total median %.0f mean %.0f cycles" % (np.median(s[:, 6] - s[:, 0]), (s[:, 6] - s[:, 0]).mean()))
    print("   pre-pass + rounds, sorted:", np.sort(d[:, 3])[:: max(1, len(d) // 16)])
