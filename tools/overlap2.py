"""Two FULL batches in flight on two handles / two streams, with the persistent grids sized to coexist (diagnostic build:
RSPT_HIP_LIB=rspt_amd/librspt_hip_diag.so; the grid knobs are read from the environment when a packer is created).

    python tools/overlap2.py [k1_grid:hist_grid:enc_grid[:tile]] ...

Prints per configuration the time per batch of ONE handle alone (serial) and of two handles on two streams (dual)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rspt_amd import api, synth

B, nch, ns = int(os.environ.get("OV_BLOCKS", "64")), 64, 65536
dev = torch.device("cuda", 0)
srcs = [synth.synth_batch_native(B, nch, ns, first_block=s * B, device=dev) for s in range(2)]


def setenv(k1, hg, eg, tile):
    for k, v in (("RSPT_K1_GRID", k1), ("RSPT_HIST_GRID", hg), ("RSPT_ENC_GRID", eg), ("RSPT_TILE", tile)):
        if v:
            os.environ[k] = str(v)
        else:
            os.environ.pop(k, None)


def run(nhandles, steps=24):
    pks = [api.new_xdelta_hzr(4, nch, ns, 3) for _ in range(nhandles)]
    streams = [torch.cuda.Stream(dev) for _ in range(nhandles)]
    stride = (pks[0].max_compressed_size + 255) // 256 * 256
    dst = [torch.empty((B, stride), dtype=torch.uint8, device=dev) for _ in range(2)]
    sz = [torch.empty(B, dtype=torch.int64, device=dev) for _ in range(2)]
    for p in pks:
        p.reserve(B)

    def step(i):
        h = i % nhandles
        with torch.cuda.stream(streams[h]):
            pks[h].compress_batch(srcs[i & 1], dst[i & 1], sz[i & 1], stride)

    for i in range(4):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    tot = int(sz[0].sum().item()) + int(sz[1].sum().item())
    for p in pks:
        p.close()
    return ms, tot


cfgs = sys.argv[1:] or ["0:0:0"]
for c in cfgs:
    f = [int(x) for x in c.split(":")] + [0, 0, 0, 0]
    setenv(f[0], f[1], f[2], f[3])
    a, ta = run(1)
    b, tb = run(2)
    print("k1=%-4d hist=%-4d enc=%-4d tile=%-4d  serial %.4f ms   dual %.4f ms   (bytes %d %d)" % (f[0], f[1], f[2], f[3], a, b, ta, tb), flush=True)
