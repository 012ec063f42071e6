"""hzr packer (no transform): where inside which hzr block does a decoded batch differ?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rspt_amd import api, synth
B, nch, ns = 64, 64, 65536
dev = torch.device("cuda", 0)
d_src = synth.synth_batch_native(B, nch, ns, device=dev)
# make it compressible for the plain hzr packer: delta along time on the host side is not needed -- keep the low 3 bytes noisy, top byte constant
pk = api.new_hzr(4, nch, ns)
stride = (pk.max_compressed_size + 255) // 256 * 256
dst = torch.empty((B, stride), dtype=torch.uint8, device=dev)
sz = torch.empty(B, dtype=torch.int64, device=dev)
pk.compress_batch(d_src, dst, sz, stride)
for rep in range(3):
    out = torch.zeros_like(d_src); used = torch.empty(B, dtype=torch.int64, device=dev)
    pk.decompress_batch(dst, B, stride, out, used)
    torch.cuda.synchronize()
    a = out.view(B, -1).cpu().numpy().reshape(B, ns, nch, 4)
    r = d_src.view(B, -1).cpu().numpy().reshape(B, ns, nch, 4)
    bad = np.argwhere((a != r).any(axis=(1, 2, 3))).ravel()
    print("rep", rep, "bad streams", bad[:20], len(bad))
    for b in bad[:4]:
        d = a[b] != r[b]          # [ns][nch][plane]
        for k in range(4):
            ch = np.argwhere(d[:, :, k].any(axis=0)).ravel()
            for c in ch[:3]:
                pos = np.argwhere(d[:, c, k]).ravel()
                print("   stream %d plane %d block(channel) %d: %d bytes differ, first %d last %d; got %s want %s" % (b, k, c, len(pos), pos[0], pos[-1], a[b][pos[:6], c, k], r[b][pos[:6], c, k]))
