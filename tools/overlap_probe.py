"""Does the memory-heavy front end of one half batch overlap with the VALU-bound encoders of the other?
Two packers, two streams, half the blocks each, against one packer with all blocks."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rspt_amd import api, synth

B, nch, ns = 64, 64, 65536
dev = torch.device("cuda", 0)
d_src = synth.synth_batch_native(B, nch, ns, device=dev)

def run_single(steps=20):
    pk = api.new_xdelta_hzr(4, nch, ns, 3)
    pk.reserve(B)
    stride = (pk.max_compressed_size + 255) // 256 * 256
    dst = torch.empty((B, stride), dtype=torch.uint8, device=dev)
    sz = torch.empty(B, dtype=torch.int64, device=dev)
    for _ in range(3):
        pk.compress_batch(d_src, dst, sz, stride)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        pk.compress_batch(d_src, dst, sz, stride)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

def run_split(parts, steps=20):
    n = B // parts
    pks = [api.new_xdelta_hzr(4, nch, ns, 3) for _ in range(parts)]
    streams = [torch.cuda.Stream(dev) for _ in range(parts)]
    stride = (pks[0].max_compressed_size + 255) // 256 * 256
    dst = [torch.empty((n, stride), dtype=torch.uint8, device=dev) for _ in range(parts)]
    sz = [torch.empty(n, dtype=torch.int64, device=dev) for _ in range(parts)]
    for p in pks:
        p.reserve(n)
    def step():
        for i in range(parts):
            with torch.cuda.stream(streams[i]):
                pks[i].compress_batch(d_src[i * n:(i + 1) * n], dst[i], sz[i], stride)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

for rep in range(2):
    print("single %.4f ms   split2 %.4f ms   split4 %.4f ms" % (run_single(), run_split(2), run_split(4)), flush=True)
