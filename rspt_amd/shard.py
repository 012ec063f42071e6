"""Multi-GPU host logic: block sharding and the gather of compressed streams.

Independent blocks shard across ranks with no data-path collective; the only
exchange is the gather of the packed streams to one rank: sizes by
all_gather, payload by send/recv (a gatherv).  Pure torch.distributed, so the same
code runs over RCCL (backend "nccl", GPU tensors) and over gloo (CPU tensors, used
by the CPU tests).  The container layout is the one rspt_hip_pack_batch_dev writes
(include/rspt_hip.h); `pack_container` is its host-side equivalent for tests and
for callers that already hold the streams on the host.
"""
import struct

import numpy as np

PACK_MAGIC = 0x4B43415054505352  # "RSPTPACK"
PACK_HEAD = 32


def shard_range(nblocks_total, rank, world):
    """contiguous block range of `rank`: [first, first+count)  (blocks [g*B/G, (g+1)*B/G))"""
    first = rank * nblocks_total // world
    last = (rank + 1) * nblocks_total // world
    return first, last - first


LEN_MASK = (1 << 56) - 1  # index length word: length | nb << 56 | invalid << 63


def pack_container(streams, nb):
    """nb: one int for all streams, or one per stream (nb escalates inside a batch)"""
    n = len(streams)
    nbs = [nb] * n if isinstance(nb, int) else list(nb)
    offs, pos = [], 0
    for s in streams:
        offs.append(pos)
        pos += (len(s) + 15) & ~15
    out = bytearray(PACK_HEAD + 16 * n + pos)
    struct.pack_into("<QQQQ", out, 0, PACK_MAGIC, n, pos, nbs[-1] if n else 0)
    for i, s in enumerate(streams):
        struct.pack_into("<QQ", out, PACK_HEAD + 16 * i, offs[i], len(s) | (nbs[i] << 56))
        base = PACK_HEAD + 16 * n + offs[i]
        out[base : base + len(s)] = s
    return bytes(out)


def unpack_container(buf, per_stream_nb=False):
    """-> (list of streams, nb of the last stream); per_stream_nb=True: (streams, [nb per stream]).
    A stream flagged invalid (it did not fit its slot at compress time) comes back as None."""
    buf = bytes(buf)
    magic, n, payload, nbw = struct.unpack_from("<QQQQ", buf, 0)
    if magic != PACK_MAGIC:
        raise ValueError("not an RSPTPACK container")
    base = PACK_HEAD + 16 * n
    if len(buf) < base + payload:
        raise ValueError("truncated container")
    out, nbs = [], []
    for i in range(n):
        off, lw = struct.unpack_from("<QQ", buf, PACK_HEAD + 16 * i)
        ln = lw & LEN_MASK
        if off + ln > payload:
            raise ValueError("container index entry %d points outside the payload" % i)
        out.append(None if lw >> 63 else buf[base + off : base + off + ln])
        nbs.append((lw >> 56) & 0xF)
    return (out, nbs) if per_stream_nb else (out, nbw & 0xFFFFFFFF)


def gather_containers(packed, total, dst=0, group=None, recv_bufs=None):
    """Gather one container per rank to `dst`.

    packed : 1-D uint8 tensor (CPU for gloo, CUDA for nccl), the container in [0, total)
    total  : 1-element int64 tensor on the same device
    Returns on dst: list of (tensor, nbytes) per rank in rank order; elsewhere None.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    totals = [torch.zeros(1, dtype=torch.int64, device=packed.device) for _ in range(world)]
    dist.all_gather(totals, total.view(1), group=group)
    sizes = [int(t.item()) for t in totals]
    if rank != dst:
        dist.send(packed[: sizes[rank]], dst=dst, group=group)
        return None
    out = []
    reqs = []
    for r in range(world):
        if r == dst:
            out.append((packed, sizes[r]))
            continue
        buf = recv_bufs[r] if recv_bufs is not None else torch.empty(sizes[r], dtype=torch.uint8, device=packed.device)
        reqs.append(dist.irecv(buf[: sizes[r]], src=r, group=group))
        out.append((buf, sizes[r]))
    for q in reqs:
        q.wait()
    return out
