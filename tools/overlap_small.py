"""Small shards (BASELINE configs[4] at N ranks: 1024 / N blocks of 12ch x 8192) with SEVERAL steps in flight: one handle and one
stream per slot, successive steps round-robin over the slots, no event between the slots (every handle has its own workspace).

    python tools/overlap_small.py [blocks ...]          (default 128 256 512 1024; slots 1 2 3 4)

Prints ms per step (compress + container pack) for each number of slots."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rspt_amd import api, synth

nch, ns = int(os.environ.get("OV_NCH", "12")), int(os.environ.get("OV_NS", "8192"))
dev = torch.device("cuda", 0)


def run(B, nslots, steps=48, pack=True, graph=False):
    pks = [api.new_xdelta_hzr(4, nch, ns, 3) for _ in range(nslots)]
    streams = [torch.cuda.Stream(dev) for _ in range(nslots)]
    stride = (pks[0].max_compressed_size + 255) // 256 * 256
    nbuf = 2 * nslots
    srcs = [synth.synth_batch_native(B, nch, ns, first_block=s * B, device=dev) for s in range(nbuf)]
    dst = [torch.empty((B, stride), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    sz = [torch.empty(B, dtype=torch.int64, device=dev) for _ in range(nbuf)]
    cont = [torch.empty(pks[0].pack_bound(B), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    tot = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(nbuf)]
    for p in pks:
        p.reserve(B)

    def step(i):
        h = i % nslots
        s = i % nbuf
        with torch.cuda.stream(streams[h]):
            pks[h].compress_batch(srcs[s], dst[s], sz[s], stride)
            if pack:
                pks[h].pack_batch(dst[s], sz[s], cont[s], tot[s])

    for i in range(2 * nbuf):
        step(i)
    torch.cuda.synchronize()
    if graph:
        # one graph per slot: the slot's next TWO steps (a handle alternates between two workspace sets), replayed
        assert nbuf == 2 * nslots or nslots == 1
        graphs = []
        for h in range(nslots):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=streams[h]):
                for r in range(2):
                    s = (h + r * nslots) % nbuf
                    pks[h].compress_batch(srcs[s], dst[s], sz[s], stride)
                    if pack:
                        pks[h].pack_batch(dst[s], sz[s], cont[s], tot[s])
            graphs.append(g)
        torch.cuda.synchronize()

        def replay(i):
            with torch.cuda.stream(streams[i % nslots]):
                graphs[i % nslots].replay()

        for i in range(2 * nslots):
            replay(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps // 2):
            replay(i)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / (steps // 2 * 2) * 1e3
        enq = (t1 - t0) / (steps // 2 * 2) * 1e3
    else:
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        enq = (t1 - t0) / steps * 1e3
    nbytes = int(sz[0].sum().item())
    for p in pks:
        p.close()
    return ms, enq, nbytes


for B in [int(x) for x in sys.argv[1:]] or [128, 256, 512, 1024]:
    out = []
    for nslots in (1, 2, 3, 4, 6):
        ms, enq, nb = run(B, nslots)
        out.append("%d slots %.4f (host %.4f)" % (nslots, ms, enq))
    if os.environ.get("OV_GRAPH", "1") == "1":
        for nslots in (1, 2, 3, 4, 6):
            ms, enq, nb = run(B, nslots, graph=True)
            out.append("graph x%d %.4f (host %.4f)" % (nslots, ms, enq))
    print("blocks %-5d  " % B + "   ".join(out) + "   (bytes %d)" % nb, flush=True)
