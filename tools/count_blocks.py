"""how many hzr blocks of a batch go to which encoder (k_encode heavy / light, k_encode_small, Fill / skipped)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rspt_amd import api, synth
def count(kind, B, nch, ns, bps=4):
    dev = torch.device("cuda", 0)
    d = synth.synth_batch_native(B, nch, ns, bps=bps, device=dev)
    pk = api.SignalPacker(kind, bps, nch, ns, 3)
    pk.compress_batch(d); torch.cuda.synchronize()
    N = nch * ns; nblk = (N + 65535) // 65536; nhb = B * 4 * nblk
    meta = pk.debug_read(4, nhb * 16).view(np.uint32).reshape(nhb, 4)
    nz = pk.debug_read(8, nhb * 4).view(np.uint32)
    mode, plen, fill = meta[:, 0], meta[:, 1], meta[:, 3]
    pc = np.array([bin(int(x)).count("1") for x in nz])
    huff = mode == 1
    small = huff & (plen <= 3072) & (fill <= 512) & (pc <= 2)
    big = (huff | (mode == 0)) & ~small if False else ((mode == 1) | (mode == 3)) & ~small
    print("%-10s B=%d %dx%d: modes %s  small %d  big %d (heavy %d, light %d)" % (kind, B, nch, ns, np.bincount(mode, minlength=5).tolist(), small.sum(), big.sum(), (big & (plen >= 16384)).sum(), (big & (plen < 16384)).sum()))
    pk.close()
count("xdelta_hzr", 64, 64, 65536)
count("xdelta_hzr", 1024, 12, 8192)
count("hadamard", 16, 64, 65536)
count("dct", 16, 64, 65536)
count("xdelta_hzr", 64, 64, 65536, 3)
count("hzr", 64, 64, 65536)
