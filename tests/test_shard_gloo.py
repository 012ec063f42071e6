"""The N>1 host path on CPU: world_size 2 over gloo.  Each rank takes its contiguous
block range, produces streams (with the oracle as the stand-in compressor -- test
infrastructure), packs them into a container and the containers are gathered to
rank 0 exactly as bench.py does over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from rspt_amd import shard  # noqa: E402

NCH, NS, BPS, NBLOCKS = 3, 700, 4, 7


def _blocks():
    import cases

    return [cases._rand_native(NCH, NS, BPS, 800 + i, [50, 1 << 14, 60][i % 3], walk=bool(i & 1)) for i in range(NBLOCKS)]


def _worker(rank, world, port, q):
    import torch.distributed as dist

    from oracle.oracle import Oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, count = shard.shard_range(NBLOCKS, rank, world)
    orc = Oracle()
    pk = orc.packer("xdelta_hzr", BPS, NCH, NS, 1)
    blocks = _blocks()[first : first + count]
    streams, nbs = [], []
    for b in blocks:
        streams.append(pk.compress(b))
        nbs.append(orc.packer_nb(pk))  # the planes of THIS stream (nb escalates inside a shard)
    cont = shard.pack_container(streams, nbs)
    packed = torch.frombuffer(bytearray(cont), dtype=torch.uint8)
    total = torch.tensor([len(cont)], dtype=torch.int64)
    got = shard.gather_containers(packed, total, dst=0)
    # the per-step exchange of bench.py: every rank learns the size of every stream (an equal count per rank there)
    mine = torch.tensor([len(x) for x in streams[:3]], dtype=torch.int64)
    everyone = torch.zeros((world, 3), dtype=torch.int64)
    dist.all_gather(list(everyone.unbind(0)), mine)
    assert everyone[rank].tolist() == mine.tolist() and (everyone > 0).all()
    if rank == 0:
        q.put([bytes(t[:n].numpy().tobytes()) for t, n in got])
    dist.barrier()
    dist.destroy_process_group()


def test_shard_ranges_cover_everything():
    for n in (1, 7, 64, 1024, 1000):
        for w in (1, 2, 3, 8):
            spans = [shard.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            for (a, ca), (b2, _) in zip(spans, spans[1:]):
                assert a + ca == b2


def test_container_roundtrip():
    streams = [b"", b"x", bytes(range(40)), b"\xff" * 16]
    c = shard.pack_container(streams, 3)
    out, nb = shard.unpack_container(c)
    assert out == streams and nb == 3
    with pytest.raises(ValueError):
        shard.unpack_container(b"\0" * 64)


def test_gather_world2_gloo():
    from oracle.oracle import Oracle

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    conts = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # rank order == block order; each rank's packer escalates on its own (one instance per rank)
    orc = Oracle()
    blocks = _blocks()
    pos = 0
    for r, c in enumerate(conts):
        streams, nbs = shard.unpack_container(c, per_stream_nb=True)
        first, count = shard.shard_range(NBLOCKS, r, 2)
        assert first == pos and len(streams) == count
        pk = orc.packer("xdelta_hzr", BPS, NCH, NS, 1)
        want, want_nb = [], []
        for b in blocks[first : first + count]:
            want.append(pk.compress(b))
            want_nb.append(orc.packer_nb(pk))
        assert streams == want and nbs == want_nb and shard.unpack_container(c)[1] == orc.packer_nb(pk)
        # EVERY stream decodes back with a decoder configured from its own index entry (each rank escalates on its own,
        # and inside its shard)
        for s_, nb_, b in zip(streams, nbs, blocks[first : first + count]):
            dec = orc.packer("xdelta_hzr", BPS, NCH, NS, nb_)
            assert dec.decompress(s_)[0] == b.tobytes()
        pos += count
    assert pos == NBLOCKS


def _lag_worker(rank, world, port, q):
    """three steps of the lagged gather (bench.py --workload c5 does the same over RCCL): the payload of step i is
    posted at step i + 1, the last one by flush(); every step's containers differ in size and content"""
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lag = shard.LaggedGather(dst=0)
    got = []
    for i in range(3):
        streams = [bytes([rank * 16 + i]) * (5 + 7 * i + 3 * rank + j) for j in range(2 + rank)]
        cont = shard.pack_container(streams, 2)
        packed = torch.frombuffer(bytearray(cont) + bytearray(64), dtype=torch.uint8)  # (a buffer longer than the container)
        done = lag.step(packed, torch.tensor([len(cont)], dtype=torch.int64))
        assert (done is None) == (i == 0 or rank != 0)
        if done is not None:
            got.append([bytes(t[:n].numpy().tobytes()) for t, n in done])
    done = lag.flush()
    if rank == 0:
        got.append([bytes(t[:n].numpy().tobytes()) for t, n in done])
        q.put(got)
    assert lag.flush() is None
    dist.barrier()
    dist.destroy_process_group()


def test_lagged_gather_world2_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_lag_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert len(got) == 3
    for i, per_rank in enumerate(got):
        assert len(per_rank) == 2
        for r, c in enumerate(per_rank):
            streams, nb = shard.unpack_container(c)
            assert nb == 2 and streams == [bytes([r * 16 + i]) * (5 + 7 * i + 3 * r + j) for j in range(2 + r)]
