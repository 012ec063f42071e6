"""Batched decompress with several batches in flight (one handle + its own stream per slot); ms per batch.

    GPU_MAX_HW_QUEUES=16 python tools/overlap_dec.py [blocks ...]     (OV_NCH / OV_NS: the block shape, default 64 x 65536)"""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rspt_amd import api, synth

nch, ns = int(os.environ.get("OV_NCH", "64")), int(os.environ.get("OV_NS", "65536"))
dev = torch.device("cuda", 0)


def run(B, nslots, steps=24):
    pks = [api.new_xdelta_hzr(4, nch, ns, 3) for _ in range(nslots)]
    streams = [torch.cuda.ExternalStream(p.stream_ptr, device=dev) for p in pks]
    stride = (pks[0].max_compressed_size + 255) // 256 * 256
    src = [synth.synth_batch_native(B, nch, ns, first_block=s * B, device=dev) for s in range(2)]
    comp = []
    for s in range(2):
        d, z = pks[0].compress_batch(src[s], None, None, stride)
        comp.append((d, z))
    torch.cuda.synchronize()
    out = [torch.empty((B, pks[0].block_bytes), dtype=torch.uint8, device=dev) for _ in range(nslots)]
    used = [torch.empty(B, dtype=torch.int64, device=dev) for _ in range(nslots)]
    for p in pks:
        p.reserve(B)

    def step(i):
        h = i % nslots
        with torch.cuda.stream(streams[h]):
            pks[h].decompress_batch(comp[i & 1][0], B, stride, out[h], used[h])

    for i in range(2 * nslots):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    ok = all(bool(torch.equal(out[h], src[(steps - nslots + ((h - steps) % nslots)) & 1])) for h in range(nslots)) if False else None
    for p in pks:
        p.close()
    return ms


for B in [int(x) for x in sys.argv[1:]] or [64, 16]:
    print("blocks %-4d " % B + "   ".join("%d in flight %.4f" % (n, run(B, n)) for n in (1, 2, 3, 4)), flush=True)
