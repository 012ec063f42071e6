"""Randomised GPU parity for the hzr kernels: byte streams of every density go through a 1-channel 8-bit `hzr` packer
(plane 0 of its stream IS hzr_encode(data), signal_packer_base.cpp:69-82) and must equal the oracle's encoding bit for
bit; the GPU decoder must give the bytes back.  The densities steer the encoder through all of its row paths: dense
quads, quads with long run tokens, register slots, the token queue of light blocks, the one-wave small-block encoder,
Fill and PlainCopy blocks, runs across the 16662 cap and across 64 KiB block edges."""
import struct

import numpy as np
import pytest

from streamtools import describe_mismatch, parse_stream

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from rspt_amd import api as a

    assert a.lib().rspt_hip_device_count() > 0, "no gfx950 device visible"
    return a


def _gen(seed, n, kind):
    r = np.random.default_rng(seed)
    if kind == "dense":  # almost no zeros, wide alphabet
        x = r.integers(1, 256, n, dtype=np.int64)
        z = r.random(n) < 0.03
        x[z] = 0
    elif kind == "peaky":  # a few very frequent symbols + rare ones (long codes), some zeros
        x = r.choice(np.arange(256), size=n, p=_geom(256, 0.35))
    elif kind == "medium":  # ~8 % tokens: literal followed by a short run
        x = np.zeros(n, dtype=np.int64)
        pos = np.cumsum(r.integers(2, 24, n // 8))
        pos = pos[pos < n]
        x[pos] = r.integers(1, 6, pos.size)
    elif kind == "sparse":  # long runs, some beyond the 16662 cap
        x = np.zeros(n, dtype=np.int64)
        pos = np.cumsum(r.integers(1, 40000, max(2, n // 9000)))
        pos = pos[pos < n]
        x[pos] = r.integers(1, 256, pos.size)
    elif kind == "bursty":  # dense stretches and empty stretches alternate (rows of both kinds inside one block)
        x = np.zeros(n, dtype=np.int64)
        p = 0
        while p < n:
            ln = int(r.integers(50, 6000))
            if r.random() < 0.5:
                x[p : p + ln] = r.integers(0, 256, min(ln, n - p))
            p += ln
    elif kind == "noise":  # incompressible: PlainCopy
        x = r.integers(0, 256, n, dtype=np.int64)
    elif kind == "const":
        x = np.full(n, int(r.integers(0, 256)), dtype=np.int64)
    elif kind == "twos":  # runs of exactly one and two zeros everywhere (symbols 0 and 256)
        x = r.integers(1, 40, n, dtype=np.int64)
        idx = r.integers(0, n - 3, n // 6)
        x[idx] = 0
        idx2 = r.integers(0, n - 3, n // 9)
        x[idx2] = 0
        x[idx2 + 1] = 0
    elif kind == "deep":  # Fibonacci counts: codes of up to ~21 bits in dense rows, the rarest symbols side by side (pairs of codes > 32 bits)
        fib = [1, 1]
        while sum(fib) + fib[-1] + fib[-2] <= n:
            fib.append(fib[-1] + fib[-2])
        vals = np.concatenate([np.full(c, 10 + i, dtype=np.int64) for i, c in enumerate(fib)])
        rest = n - vals.size
        x = np.concatenate([vals[:8], r.permutation(vals[8:]), np.full(rest, 10 + len(fib) - 1, dtype=np.int64)])  # rare ones first, in order
    elif kind == "deeprun":  # "deep" with one run of 300 zeros: the long-run symbol (14 extra bits) is the rarest, so its code is the
        # deepest -- code + extra bits reach past 32 stream bits (the decoder's 64-bit window branch)
        if n > 1000:
            y = _gen(seed, n - 300, "deep").astype(np.int64)  # (inserted, not overwritten: the Fibonacci counts stay what they are)
            x = np.concatenate([y[: n // 2], np.zeros(300, dtype=np.int64), y[n // 2 :]])
        else:
            x = _gen(seed, n, "deep").astype(np.int64)
    elif kind == "wide":  # five heavy symbols over 250 equally rare ones: ~100 ten-bit prefixes lead to longer codes (the decoder's
        # second-level table has 32 slots: the rest walk the tree) -- and the payload is long enough for the parallel tree recovery
        x = r.choice(np.arange(5, 255), size=n)
        u = r.random(n)
        for v, q in ((4, 0.92), (3, 0.84), (2, 0.72), (1, 0.52), (0, 0.28)):
            x[u < q] = v
    else:
        raise ValueError(kind)
    return x.astype(np.uint8)


def _geom(k, q):
    p = q ** np.arange(k)
    return p / p.sum()


KINDS = ["dense", "peaky", "medium", "sparse", "bursty", "noise", "const", "twos", "deep", "wide", "deeprun"]
SIZES = [65536, 65536 * 3 + 1234, 4097, 200000, 16, 70000]
CASES = [(k, SIZES[(i + j) % len(SIZES)], 100 * i + j) for i, k in enumerate(KINDS) for j in range(3)]


@pytest.mark.parametrize("kind,n,seed", CASES)
def test_random_stream_bit_exact_and_back(api, orc, kind, n, seed):
    data = _gen(seed, n, kind)
    pk = api.new_hzr(1, 1, n)
    got = pk.compress(data, dst_max_len=pk.max_compressed_size)
    p = parse_stream(got)
    o0 = p["planes"][0]["offset"]
    chunk0 = got[o0 + 4 : o0 + 4 + p["planes"][0]["len"]]
    want = orc.hzr_encode(data)
    assert chunk0 == want, "%s n=%d seed=%d: %s" % (
        kind, n, seed, describe_mismatch(b"\0" + struct.pack("<I", len(chunk0)) + chunk0, b"\0" + struct.pack("<I", len(want)) + want))
    ok, m = orc.hzr_verify(chunk0)
    assert ok and m == n
    dec, used = pk.decompress(got)
    assert used == len(got) and dec == data.tobytes()
    pk.close()


def test_random_batch_matches_single_calls(api, orc):
    """many different blocks in one launch (the work queues, the segment offsets and the side stream see a real mix)"""
    import torch

    n = 65536 * 2 + 777
    kinds = [KINDS[i % len(KINDS)] for i in range(24)]
    blocks = [_gen(1000 + i, n, k) for i, k in enumerate(kinds)]
    pk = api.new_hzr(1, 1, n)
    d_src = torch.from_numpy(np.stack(blocks)).cuda()
    stride = (pk.max_compressed_size + 255) // 256 * 256
    d_dst = torch.zeros((len(blocks), stride), dtype=torch.uint8, device="cuda")
    d_sizes = torch.zeros(len(blocks), dtype=torch.int64, device="cuda")
    pk.compress_batch(d_src, d_dst, d_sizes, stride)
    torch.cuda.synchronize()
    po = orc.packer("hzr", 1, 1, n)
    for i, data in enumerate(blocks):
        got = d_dst[i, : int(d_sizes[i])].cpu().numpy().tobytes()
        want = po.compress(data)
        assert got == want, "block %d (%s): %s" % (i, kinds[i], describe_mismatch(got, want))
    out, used = pk.decompress_batch(d_dst, len(blocks), stride)
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), torch.from_numpy(np.stack(blocks))) and torch.equal(used, d_sizes)
    pk.close()


def test_blocks_with_one_populated_segment_every_launch(api, orc):
    """hzr blocks with a single non-zero 4 KiB segment that holds more tokens than the one-wave encoder takes: k_tree takes their
    histogram itself and k_encode counts each wave's tokens once more for the segment offsets (own_bits).  In that pass the
    sparse-row queue of wave 0 used to overlap the last bins of wave 15's histogram in LDS -- wave 15 counts the run that reaches
    the block end right there -- and about every second launch produced a damaged payload (found by tools/soak.py, round 4).
    A race: so the same batch many times over."""
    import torch

    r = np.random.default_rng(77)
    blocks = []
    for seg in (0, 0, 3, 15, 0, 7):
        x = np.zeros(65536 * 2 + 4096, dtype=np.uint8)
        for blk in range(3):
            n = 4096 if blk < 2 else 2048
            lo = blk * 65536 + (seg * 4096 if blk < 2 else 0)
            pos = lo + np.sort(r.choice(n, size=n // 6, replace=False))
            x[pos] = r.integers(1, 256, pos.size)
        blocks.append(x)
    n = blocks[0].size
    pk = api.new_hzr(1, 1, n)
    po = orc.packer("hzr", 1, 1, n)
    want = [po.compress(b) for b in blocks]
    d_src = torch.from_numpy(np.stack(blocks)).cuda()
    stride = (pk.max_compressed_size + 255) // 256 * 256
    for rep in range(40):
        d_dst = torch.zeros((len(blocks), stride), dtype=torch.uint8, device="cuda")
        d_sizes = torch.zeros(len(blocks), dtype=torch.int64, device="cuda")
        pk.compress_batch(d_src, d_dst, d_sizes, stride)
        torch.cuda.synchronize()
        for i, w in enumerate(want):
            got = d_dst[i, : int(d_sizes[i])].cpu().numpy().tobytes()
            assert got == w, "launch %d block %d: %s" % (rep, i, describe_mismatch(got, w))
    pk.close()
