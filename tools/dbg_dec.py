"""decompress a full-size batch and say where it differs from the input (debug aid)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rspt_amd import api, synth
B, nch, ns = int(os.environ.get("DBG_B", "64")), 64, 65536
dev = torch.device("cuda", 0)
d_src = synth.synth_batch_native(B, nch, ns, device=dev)
pk = api.new_xdelta_hzr(4, nch, ns, 3)
stride = (pk.max_compressed_size + 255) // 256 * 256
dst = torch.empty((B, stride), dtype=torch.uint8, device=dev)
sz = torch.empty(B, dtype=torch.int64, device=dev)
pk.compress_batch(d_src, dst, sz, stride)
for rep in range(3):
    out = torch.zeros_like(d_src); used = torch.empty(B, dtype=torch.int64, device=dev)
    pk.decompress_batch(dst, B, stride, out, used)
    torch.cuda.synchronize()
    a = out.view(B, -1).view(torch.int32).cpu().numpy().reshape(B, ns, nch)
    r = d_src.view(B, -1).view(torch.int32).cpu().numpy().reshape(B, ns, nch)
    bad = np.argwhere((a != r).any(axis=(1, 2))).ravel()
    print("rep", rep, "used ok", bool((used.cpu() == sz.cpu()).all()), "bad streams", bad[:20], len(bad))
    for b in bad[:3]:
        d = a[b] != r[b]
        chans = np.argwhere(d.any(axis=0)).ravel()
        print("  stream", b, "channels", chans[:10], len(chans), "first sample", int(np.argmax(d.any(axis=1))), "diff", (a[b][d][:4] - r[b][d][:4]))
