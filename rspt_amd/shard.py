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


class LaggedGather:
    """The gather of one container per rank and step, WITHOUT a host round trip inside the step.

    A gatherv needs the byte counts on the host of both ends before the payload transfers can be posted.  Reading them in
    the step they are produced in (gather_containers: `.item()`) stalls the host once per step behind the step's kernels.
    Here the sizes of step i travel by an all-gather on the device (ncclAllGather of the sizes) and a copy into
    page-locked host memory, both on a side stream; the host looks at them one step later -- when they have long arrived --
    and posts the payload of step i then, as one group of send / recv (ncclGroupStart ... ncclGroupEnd), again on the side
    stream -- BEFORE that stream is made to wait for the kernels of step i+1, so the payload of a step overlaps the kernels of
    the next one.  flush() posts the last step's payload.  `done_event` (cuda) is the event behind the payload group that
    `step` / `flush` last returned: a consumer of the receive buffers waits on it (the next payload overwrites them).

        lag = LaggedGather(dst=0, recv_bufs=..., device=dev)        # recv_bufs[r]: bound-sized buffer per source rank (dst only)
        for i in steps:
            ...kernels that write packed[i & 1], totals[i & 1] on the current stream...
            done = lag.step(packed[i & 1], totals[i & 1])           # -> what arrived for step i-1 (dst) / None
            # before packed[i & 1] is written again (two steps on): lag.wait_slot_free(i & 1)
        last = lag.flush()

    Works over gloo with CPU tensors too (no streams: everything is immediate), which is how the CPU tests run it.
    """

    def __init__(self, dst=0, group=None, recv_bufs=None, device=None, timing=False):
        import torch
        import torch.distributed as dist

        self.dist, self.torch = dist, torch
        self.group, self.dst = group, dst
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.recv_bufs = recv_bufs
        self.cuda = device is not None and torch.device(device).type == "cuda"
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        self.side = torch.cuda.Stream(self.device) if self.cuda else None
        self.sizes_dev = [torch.zeros(self.world, dtype=torch.int64, device=self.device) for _ in range(2)]
        self.sizes_host = [torch.zeros(self.world, dtype=torch.int64).pin_memory() if self.cuda else torch.zeros(self.world, dtype=torch.int64)
                           for _ in range(2)]
        self.ev_sizes = [None, None]
        self.slot_free = [None, None]
        self.pending = None  # (slot, packed tensor) whose payload has not been posted yet
        self.k = 0
        self.timing = timing  # keep (start, end) event pairs of the payload groups for mean_payload_ms() (benchmarking only)
        self.payload_ms = []
        self.done_event = None
        self.gathered_bytes = 0

    def _post_payload(self, slot, packed):
        """sizes of `slot` are on the host by now (their event is a step old): post the group of transfers"""
        dist, torch = self.dist, self.torch
        if self.ev_sizes[slot] is not None:
            self.ev_sizes[slot].synchronize()  # (recorded a whole step ago: does not wait in the steady state)
        sizes = [int(x) for x in self.sizes_host[slot].tolist()]
        ops, out = [], None

        def build():
            nonlocal out
            if self.rank != self.dst:
                ops.append(dist.P2POp(dist.isend, packed[: sizes[self.rank]], self.dst, self.group))
                return
            out = []
            for r in range(self.world):
                if r == self.dst:
                    out.append((packed, sizes[r]))
                    continue
                # (temporary receive buffers are allocated under the stream that uses them)
                buf = self.recv_bufs[r] if self.recv_bufs is not None else torch.empty(sizes[r], dtype=torch.uint8, device=packed.device)
                ops.append(dist.P2POp(dist.irecv, buf[: sizes[r]], r, self.group))
                out.append((buf, sizes[r]))
            self.gathered_bytes = sum(n for _, n in out)

        if self.cuda:
            with torch.cuda.stream(self.side):
                build()
                e0 = torch.cuda.Event(enable_timing=True) if self.timing else None
                e1 = torch.cuda.Event(enable_timing=self.timing)
                if e0 is not None:
                    e0.record(self.side)
                reqs = dist.batch_isend_irecv(ops) if ops else []
                for q in reqs:
                    q.wait()  # (stream-ordered for nccl: no host wait)
                e1.record(self.side)
                if self.timing:
                    self.payload_ms.append((e0, e1))
                    del self.payload_ms[:-256]
                self.slot_free[slot] = e1
                self.done_event = e1
        else:
            build()
            for q in (dist.batch_isend_irecv(ops) if ops else []):
                q.wait()
        return out

    def step(self, packed, total):
        """Call behind the kernels that produced `packed` / `total` (enqueued on the current stream)."""
        dist, torch = self.dist, self.torch
        slot = self.k & 1
        done = None
        if self.pending is not None:  # the previous step's payload: posted before the side stream waits for this step's kernels
            done = self._post_payload(*self.pending)
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self.side.wait_event(ev)
            with torch.cuda.stream(self.side):
                dist.all_gather_into_tensor(self.sizes_dev[slot], total.view(1), group=self.group)
                self.sizes_host[slot].copy_(self.sizes_dev[slot], non_blocking=True)
                self.ev_sizes[slot] = torch.cuda.Event()
                self.ev_sizes[slot].record(self.side)
        else:
            parts = [torch.zeros(1, dtype=torch.int64) for _ in range(self.world)]
            dist.all_gather(parts, total.view(1), group=self.group)
            self.sizes_host[slot].copy_(torch.cat(parts))
        self.pending = (slot, packed)
        self.k += 1
        return done

    def flush(self):
        done = None
        if self.pending is not None:
            done = self._post_payload(*self.pending)
            self.pending = None
        return done

    def wait_slot_free(self, slot):
        """make the current stream wait until the payload that last used this slot's container has left"""
        if self.cuda and self.slot_free[slot] is not None:
            self.torch.cuda.current_stream(self.device).wait_event(self.slot_free[slot])

    def mean_payload_ms(self):
        if not (self.cuda and self.payload_ms):
            return None
        self.torch.cuda.synchronize(self.device)
        return sum(a.elapsed_time(b) for a, b in self.payload_ms) / len(self.payload_ms)
