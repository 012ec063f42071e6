import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rspt_amd import api, synth
pk = api.new_xdelta_hzr(4, 64, 65536, 3)
d = synth.synth_batch_native(64, 64, 65536, device="cuda")
pk.compress_batch(d); torch.cuda.synchronize()
nz = pk.debug_read(8, 64 * 4 * 64 * 4).view(np.uint32).reshape(64, 4, 64)
meta = pk.debug_read(4, 64 * 4 * 64 * 16).view(np.uint32).reshape(64, 4, 64, 4)
pc = np.vectorize(lambda x: bin(int(x)).count("1"))(nz)
for k in range(3):
    print("plane", k, "segment bits set: mean %.2f  hist" % pc[:, k].mean(), np.bincount(pc[:, k].ravel(), minlength=17))
    m = meta[:, k]
    print("   modes", np.bincount(m[..., 0].ravel(), minlength=4), "payload mean %.0f" % m[..., 1].mean(), "ntok(median, max)", np.median(m[..., 3]), m[..., 3].max(), "small-eligible", int(((m[..., 0] == 1) & (m[..., 1] <= 3072) & (m[..., 3] <= 512)).sum()))
