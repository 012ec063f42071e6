"""GPU parity tests proper: the HIP path, called through the C ABI, against the
oracle on the same inputs and against the committed golden fixtures.  Bit-exact:
this is integer / byte work."""
import struct
import zlib

import numpy as np
import pytest

import cases
from streamtools import describe_mismatch, parse_stream

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from rspt_amd import api as a

    assert a.lib().rspt_hip_device_count() > 0, "no gfx950 device visible: the HIP path cannot run (no CPU fallback)"
    return a


LOSSLESS = [c["name"] for c in cases.packer_cases() if c["kind"] in ("xdelta_hzr", "hzr")]


@pytest.mark.parametrize("name", LOSSLESS)
def test_lossless_stream_bit_exact(api, orc, golden, packer_cases, name):
    c, g = packer_cases[name], golden["packers"][name]
    pk = api.SignalPacker(c["kind"], c["bps"], c["nch"], c["ns"], c["nb"])
    got = pk.compress(c["data"])
    po = orc.packer(c["kind"], c["bps"], c["nch"], c["ns"], c["nb"])
    want = po.compress(c["data"])
    assert got == want, describe_mismatch(got, want)
    assert len(got) == g["size"] and orc.fnv1a(got) == g["fnv1a"] and zlib.crc32(got) == g["crc32"]
    assert pk.nb == g["final_nb"] == orc.packer_nb(po)
    dec, used, _ = po.decompress(got)  # the oracle decodes what the GPU wrote
    assert dec == c["data"].tobytes() and used == len(got)
    pk.close()


@pytest.mark.parametrize("name", sorted(cases.hzr_kat_inputs().keys()))
def test_hzr_kat_through_hzr_packer(api, orc, golden, name):
    """raw hzr_encode known answers: plane 0 of a 1-channel 8-bit `hzr` packer
    IS hzr_encode(data) (signal_packer_base.cpp:69-82)."""
    data = cases.hzr_kat_inputs()[name]
    pk = api.new_hzr(1, 1, data.size)
    got = pk.compress(data, dst_max_len=pk.max_compressed_size)
    p = parse_stream(got)
    assert len(p["planes"]) == 4 and p["end"] == len(got)
    o0 = p["planes"][0]["offset"]
    chunk0 = got[o0 + 4 : o0 + 4 + p["planes"][0]["len"]]
    want = orc.hzr_encode(data)
    assert chunk0 == want, describe_mismatch(b"\0" + struct.pack("<I", len(chunk0)) + chunk0, b"\0" + struct.pack("<I", len(want)) + want)
    g = golden["hzr"][name]
    assert len(chunk0) == g["size"] and orc.fnv1a(chunk0) == g["fnv1a"]
    ok, n = orc.hzr_verify(chunk0)  # every block CRC checks out
    assert ok and n == data.size
    pk.close()


def test_cxx_factories_drive_the_same_path(api, orc, packer_cases):
    """include/signal_packer.h: i_signal_packer::new_xdelta_hzr(...)->compress()."""
    c = packer_cases["readme_sine_xdelta_nb3"]
    pk = api.CxxSignalPacker("xdelta_hzr", 4, 1, 8192, 3)
    s = pk.compress(c["data"])
    assert len(s) == 2028 and orc.fnv1a(s) == 0xF98CEFBD  # SURVEY 6 / README example, current code
    pk.close()


def test_batch_is_sequential_compress_calls(api, orc):
    """rspt_hip_compress_batch_dev == nblocks successive compress() calls on one
    instance, including the persistent nb escalation in block order."""
    import torch

    nch, ns, bps = 3, 1500, 4
    amps = [10, 10, 1 << 12, 10, 1 << 22, 10, 1 << 30, 10]
    blocks = [cases._rand_native(nch, ns, bps, 300 + i, a, walk=bool(i & 1)) for i, a in enumerate(amps)]
    po = orc.packer("xdelta_hzr", bps, nch, ns, 1)
    want = [po.compress(b) for b in blocks]
    pk = api.new_xdelta_hzr(bps, nch, ns, 1)
    d_src = torch.from_numpy(np.stack(blocks)).cuda()
    d_dst, d_sizes = pk.compress_batch(d_src)
    torch.cuda.synchronize()
    sizes = d_sizes.cpu().numpy()
    out = d_dst.cpu().numpy()
    for i in range(len(blocks)):
        got = out[i, : sizes[i]].tobytes()
        assert got == want[i], "block %d: %s" % (i, describe_mismatch(got, want[i]))
    assert pk.nb == orc.packer_nb(po) == 4
    # a second batch starts from the carried nb
    d_dst2, d_sizes2 = pk.compress_batch(d_src[:2].contiguous())
    torch.cuda.synchronize()
    want2 = [po.compress(b) for b in blocks[:2]]
    for i in range(2):
        assert d_dst2[i, : int(d_sizes2[i])].cpu().numpy().tobytes() == want2[i]
    pk.close()


def test_dst_too_small_is_reported(api, packer_cases):
    c = packer_cases["ecg12x8192_xdelta"]
    pk = api.new_xdelta_hzr(c["bps"], c["nch"], c["ns"], c["nb"])
    with pytest.raises(api.RsptHipError) as e:
        pk.compress(c["data"], dst_max_len=1000)
    assert e.value.status == -5
    pk.close()


def test_full_size_roundtrip_properties(api, orc):
    """BASELINE config C3 (64 x 65536 int32), batched: every stream decodes (with
    the oracle's decoder) to its input, every hzr block CRC verifies, blocks are
    independent of their batch position."""
    import torch

    from rspt_amd import synth

    nch, ns, B = 64, 65536, 3
    pk = api.new_xdelta_hzr(4, nch, ns, 3)
    d_src = synth.synth_batch_native(B, nch, ns, first_block=7, device="cuda")
    d_dst, d_sizes = pk.compress_batch(d_src)
    torch.cuda.synchronize()
    sizes = d_sizes.cpu().numpy()
    po = orc.packer("xdelta_hzr", 4, nch, ns, 3)
    for b in range(B):
        s = d_dst[b, : int(sizes[b])].cpu().numpy().tobytes()
        p = parse_stream(s)
        assert len(p["planes"]) == 3 and p["end"] == len(s)
        for pl in p["planes"]:
            chunk = s[pl["offset"] + 4 : pl["offset"] + 4 + pl["len"]]
            ok, n = orc.hzr_verify(chunk)
            assert ok and n == nch * ns
        dec, used, _ = po.decompress(s)
        assert used == len(s) and dec == d_src[b].cpu().numpy().tobytes()
    # position independence: block 1 alone gives the same stream
    d1, s1 = pk.compress_batch(d_src[1:2].contiguous())
    torch.cuda.synchronize()
    assert d1[0, : int(s1[0])].cpu().numpy().tobytes() == d_dst[1, : int(sizes[1])].cpu().numpy().tobytes()
    pk.close()


def test_pack_batch_container(api, orc):
    """rspt_hip_pack_batch_dev: the device-side container == the streams, in block order,
    readable by the host-side parser the multi-GPU gather uses (rspt_amd/shard.py)."""
    import torch

    from rspt_amd import shard

    nch, ns, bps = 4, 2500, 4
    blocks = [cases._rand_native(nch, ns, bps, 600 + i, [40, 1 << 13, 90, 7][i % 4], walk=bool(i & 1)) for i in range(9)]
    pk = api.new_xdelta_hzr(bps, nch, ns, 1)
    d_src = torch.from_numpy(np.stack(blocks)).cuda()
    d_dst, d_sizes = pk.compress_batch(d_src)
    d_packed, d_total = pk.pack_batch(d_dst, d_sizes)
    torch.cuda.synchronize()
    total = int(d_total.item())
    assert total <= pk.pack_bound(len(blocks))
    cont = d_packed[:total].cpu().numpy().tobytes()
    streams, nbs = shard.unpack_container(cont, per_stream_nb=True)
    po = orc.packer("xdelta_hzr", bps, nch, ns, 1)
    want, want_nb = [], []
    for b in blocks:
        want.append(po.compress(b))
        want_nb.append(orc.packer_nb(po))  # nr_bytes_to_compress_ after this call = the planes of this stream
    assert streams == want
    assert nbs == want_nb and len(set(nbs)) > 1, nbs  # nb escalates INSIDE this batch: the index records it per stream
    assert shard.unpack_container(cont)[1] == orc.packer_nb(po) == pk.nb
    # host-side packer produces the same bytes
    assert shard.pack_container(want, want_nb) == cont
    # and the mixed-nb container decodes in ONE call on another instance, whatever that one's nb state is
    other = api.new_xdelta_hzr(bps, nch, ns, 4)
    out, used = other.decompress_packed(d_packed, nbytes=total)
    torch.cuda.synchronize()
    assert torch.equal(out, d_src) and used.tolist() == [len(x) for x in want]
    assert other.nb == 4
    other.close()
    pk.close()


def test_pack_batch_skips_streams_that_did_not_fit(api, orc):
    """A block whose stream exceeds dst_stride is flagged in d_sizes (bit 63) and nothing is written for it: the container
    must carry an empty, flagged entry -- not `len` bytes copied out of its neighbour's slot."""
    import torch

    from rspt_amd import shard

    nch, ns, bps = 2, 3000, 4
    quiet = [cases._rand_native(nch, ns, bps, 900 + i, 5, walk=True) for i in range(3)]
    loud = cases._rand_native(nch, ns, bps, 950, 1 << 22, walk=False)
    blocks = [quiet[0], loud, quiet[1], quiet[2], loud]  # (the last one too: its copy would run past the allocation)
    po = orc.packer("xdelta_hzr", bps, nch, ns, 3)
    want = [po.compress(b) for b in blocks]
    stride = (max(len(want[0]), len(want[2]), len(want[3])) + 64 + 15) // 16 * 16
    assert len(want[1]) > stride
    pk = api.new_xdelta_hzr(bps, nch, ns, 3)
    d_src = torch.from_numpy(np.stack(blocks)).cuda()
    d_dst = torch.zeros((len(blocks), stride), dtype=torch.uint8, device="cuda")
    d_sizes = torch.empty(len(blocks), dtype=torch.int64, device="cuda")
    pk.compress_batch(d_src, d_dst, d_sizes, stride)
    d_packed, d_total = pk.pack_batch(d_dst, d_sizes)
    torch.cuda.synchronize()
    sizes = d_sizes.cpu().numpy().view(np.uint64)
    assert [int(x >> 63) for x in sizes] == [0, 1, 0, 0, 1]
    total = int(d_total.item())
    cont = d_packed[:total].cpu().numpy().tobytes()
    streams, _ = shard.unpack_container(cont)
    assert streams == [want[0], None, want[2], want[3], None]
    assert int(np.frombuffer(cont[24:32], dtype=np.uint64)[0]) >> 32 == 2  # flagged streams counted in the header
    # decode: the good streams come back, the flagged ones are reported (bit 63), nothing is read out of bounds
    out, used = pk.decompress_packed(d_packed, nbytes=total)
    torch.cuda.synchronize()
    u = used.cpu().numpy().view(np.uint64)
    assert [int(x >> 63) for x in u] == [0, 1, 0, 0, 1]
    for i in (0, 2, 3):
        assert out[i].cpu().numpy().tobytes() == blocks[i].tobytes() and int(u[i]) == len(want[i])
    pk.close()


def test_decompress_packed_rejects_a_damaged_container(api, orc):
    """truncated payload / index entry pointing outside / wrong block count: flagged on the device, never an out-of-bounds read"""
    import torch

    nch, ns, bps, B = 3, 2000, 4, 4
    blocks = [cases._rand_native(nch, ns, bps, 970 + i, 200, walk=True) for i in range(B)]
    pk = api.new_xdelta_hzr(bps, nch, ns, 3)
    d_src = torch.from_numpy(np.stack(blocks)).cuda()
    d_dst, d_sizes = pk.compress_batch(d_src)
    d_packed, d_total = pk.pack_batch(d_dst, d_sizes)
    torch.cuda.synchronize()
    total = int(d_total.item())
    good = d_packed[:total].clone()
    out, used = pk.decompress_packed(good)
    torch.cuda.synchronize()
    assert torch.equal(out, d_src)
    # (a) the container is said to be shorter than its header claims
    _, used = pk.decompress_packed(good, nbytes=total - 16)
    torch.cuda.synchronize()
    assert all(int(x) >> 63 for x in used.cpu().numpy().view(np.uint64))
    # (b) one index entry points past the payload: that stream is flagged, the others decode
    bad = good.clone()
    idx = bad[32 : 32 + 16 * B].view(torch.int64)
    idx[2 * 2] = total  # offset of stream 2
    out, used = pk.decompress_packed(bad)
    torch.cuda.synchronize()
    u = used.cpu().numpy().view(np.uint64)
    assert [int(x >> 63) for x in u] == [0, 0, 1, 0]
    assert torch.equal(out[0], d_src[0]) and torch.equal(out[3], d_src[3])
    # (c) a payload word near 2^64: `32 + 16 n + payload` would wrap to a small number and pass a naive bound check
    bad = good.clone()
    bad[16:24].view(torch.int64)[0] = -16  # payload bytes = 2^64 - 16
    bad[32 : 32 + 16 * B].view(torch.int64)[2 * 1] = 1 << 40  # ... and then any offset would be "inside the payload"
    _, used = pk.decompress_packed(bad)
    torch.cuda.synchronize()
    assert all(int(x) >> 63 for x in used.cpu().numpy().view(np.uint64))
    # (d) a block count that does not fit the container (16 * nblocks wraps in 64 bits): refused on the host, before anything is sized from it
    bad = good.clone()
    bad[8:16].view(torch.int64)[0] = 1 << 60
    with pytest.raises(api.RsptHipError):
        pk.decompress_packed(bad)
    pk.close()


def _plane_mix(nch, ns, seed, kind):
    """int32 blocks whose xdelta planes are dense / sparse / empty in chosen places."""
    r = np.random.default_rng(seed)
    if kind == "dense":  # all three low planes busy in every channel
        x = r.integers(-(1 << 22), 1 << 22, (ns, nch))
    elif kind == "quiet":  # plane 0 only, tiny steps
        x = np.cumsum(r.integers(-3, 4, (ns, nch)), axis=0)
    elif kind == "spikes":  # quiet + a handful of big spikes: planes 1 and 2 hold a few isolated bytes (sparse hzr blocks)
        x = np.cumsum(r.integers(-3, 4, (ns, nch)), axis=0)
        for _ in range(12):
            x[r.integers(0, ns), r.integers(0, nch)] += int(r.integers(1 << 17, 1 << 21))
    elif kind == "half":  # half of the channels dense, half silent: dense and all-zero hzr blocks in one plane
        x = np.zeros((ns, nch), dtype=np.int64)
        x[:, ::2] = r.integers(-(1 << 20), 1 << 20, (ns, (nch + 1) // 2))
    elif kind == "const":  # constant non-zero samples: xdelta output is the constant -128 pattern after the first samples
        x = np.full((ns, nch), 12345)
    else:
        x = np.zeros((ns, nch), dtype=np.int64)
    return np.ascontiguousarray(x.astype("<i4")).view(np.uint8).reshape(-1)


@pytest.mark.parametrize("ns", [4096, 65536 + 4096])
def test_plane_workspace_carries_nothing_between_calls(api, orc, ns):
    """The streaming front end leaves all-zero 128-byte lines of a clean plane unwritten and the encoders wipe the sparse
    blocks they read (rspt_hip_packer::plane_dirty).  Whatever a call leaves in the plane workspace must never show in
    a later stream: one packer, batches of changing size and content, every stream against the oracle -- including a
    call whose output does not fit, a decompress in between (it decodes into the same workspace) and a single-block
    compress()."""
    import torch

    nch = 8
    po = orc.packer("xdelta_hzr", 4, nch, ns, 3)
    pk = api.new_xdelta_hzr(4, nch, ns, 3)
    script = [
        ["dense", "quiet", "spikes", "half", "zero", "const"],
        ["quiet", "quiet", "dense"],
        ["spikes", "zero", "half", "dense", "quiet", "spikes", "const", "zero"],
        ["zero"],
        ["half", "spikes"],
        ["quiet", "spikes", "quiet", "spikes", "zero", "dense"],
    ]
    seed = 0
    last = None
    for step, kinds in enumerate(script):
        blocks = []
        for kd in kinds:
            seed += 1
            blocks.append(_plane_mix(nch, ns, seed, kd))
        d_src = torch.from_numpy(np.stack(blocks)).cuda()
        if step == 2:  # first: the same batch into a destination that is too small for it (nothing is encoded, nothing wiped)
            small = torch.empty((len(blocks), 64), dtype=torch.uint8, device="cuda")
            nb_before = pk.nb
            _, d_sz = pk.compress_batch(d_src, small, None, 64)
            torch.cuda.synchronize()
            assert (d_sz.cpu().numpy().view(np.uint64) >> np.uint64(63)).all()
            pk.set_nb(nb_before)
        if step == 4 and last is not None:  # a decode writes the workspace, too
            dec, used = pk.decompress(last[1])
            assert dec == last[0].tobytes() and used == len(last[1])
            pk.set_nb(orc.packer_nb(po))
        d_dst, d_sizes = pk.compress_batch(d_src)
        torch.cuda.synchronize()
        sizes = d_sizes.cpu().numpy()
        out = d_dst.cpu().numpy()
        for i, blk in enumerate(blocks):
            want = po.compress(blk)
            got = out[i, : sizes[i]].tobytes()
            assert got == want, "call %d block %d (%s): %s" % (step, i, kinds[i], describe_mismatch(got, want))
            last = (blk, got)
        assert pk.nb == orc.packer_nb(po)
    one = _plane_mix(nch, ns, 999, "spikes")
    assert pk.compress(one) == po.compress(one)
    pk.close()


def test_front_end_shape_fuzz(api, orc):
    """Random int32 / int24 / int16 geometries through the streaming front end (k_tile_stream; the int24 loads are
    unaligned dwords, the batch's very last sample is read one byte early): channel counts on both sides of the wave
    width, ns with and without whole 16-sample groups, tiles with fewer items than threads, several tiles per block,
    batches that escalate nb in the middle (fix-up pass) -- every stream against the oracle, batch after batch on the
    same handle (the plane workspace persists)."""
    import torch

    r = np.random.default_rng(20251)
    shapes = [(1, 16), (1, 17), (2, 31), (3, 4097), (7, 1000), (12, 8192), (12, 3419), (31, 260), (64, 513), (65, 400), (100, 129), (130, 48),
              (5, 70001), (64, 2048), (20, 16384), (2, 1000003), (1, 1 << 21)]
    for kind in ("xdelta_hzr", "hzr"):
        for si, (nch, ns) in enumerate(shapes):
            bps = (4, 3, 2)[si % 3] if kind == "xdelta_hzr" else (3, 4, 2)[si % 3]  # int32 / int24 / int16 all stream
            nb0 = int(r.integers(1, 4)) if kind == "xdelta_hzr" else int(r.integers(1, 5))
            po = orc.packer(kind, bps, nch, ns, nb0)
            pk = api.SignalPacker(kind, bps, nch, ns, nb0)
            for call in range(2):
                B = int(r.integers(1, 5))
                amps = [int(r.choice([3, 60, 1 << 10, 1 << 14, 1 << 21, 1 << 29])) for _ in range(B)]
                blocks = [cases._rand_native(nch, ns, bps, int(r.integers(1 << 30)), a, walk=bool(r.integers(2))) for a in amps]
                d_src = torch.from_numpy(np.stack(blocks)).cuda()
                d_dst, d_sizes = pk.compress_batch(d_src)
                torch.cuda.synchronize()
                sizes = d_sizes.cpu().numpy()
                out = d_dst.cpu().numpy()
                for i, blk in enumerate(blocks):
                    want = po.compress(blk)
                    got = out[i, : sizes[i]].tobytes()
                    assert got == want, "%s int%d %dx%d call %d block %d amp %d: %s" % (kind, 8 * bps, nch, ns, call, i, amps[i], describe_mismatch(got, want))
                if kind == "xdelta_hzr":
                    assert pk.nb == orc.packer_nb(po)
            pk.close()


@pytest.mark.parametrize("bps", [3, 2])
def test_narrow_samples_full_size_batch(api, orc, bps):
    """int24 and int16 at the BASELINE shape (64 x 65536), batched: streams equal the oracle's, block by block."""
    import torch

    from rspt_amd import synth

    nch, ns, B = 64, 65536, 2
    pk = api.new_xdelta_hzr(bps, nch, ns, 3)
    d_src = synth.synth_batch_native(B, nch, ns, first_block=3, bps=bps, device="cuda")
    d_dst, d_sizes = pk.compress_batch(d_src)
    torch.cuda.synchronize()
    po = orc.packer("xdelta_hzr", bps, nch, ns, 3)
    for b in range(B):
        want = po.compress(d_src[b].cpu().numpy())
        got = d_dst[b, : int(d_sizes[b])].cpu().numpy().tobytes()
        assert got == want, describe_mismatch(got, want)
    assert pk.nb == orc.packer_nb(po)
    pk.close()


@pytest.mark.parametrize("nch", [1, 3])
def test_int24_batch_that_ends_with_its_allocation(api, orc, nch):
    """int24 samples are read as unaligned 4-byte words: nothing may be read past the last byte of the batch.  A batch of
    more than 10 MiB whose size is a multiple of 2 MiB ends exactly where its device allocation ends (an access past it
    is a GPU memory fault, not a silent over-read); one channel makes the discarded halo of flat index 0 the block's last
    sample."""
    import torch

    ns = (1 << 21) if nch == 1 else (1 << 20)  # 6 MiB resp. 9 MiB per block: two blocks are a multiple of 2 MiB
    pk = api.new_xdelta_hzr(3, nch, ns, 2)
    blocks = [cases._rand_native(nch, ns, 3, 4100 + i, a, walk=True) for i, a in enumerate([40, 1 << 13])]
    d_src = torch.from_numpy(np.stack(blocks)).cuda()
    assert d_src.numel() % (2 << 20) == 0 and d_src.numel() > (10 << 20)
    d_dst, d_sizes = pk.compress_batch(d_src)
    torch.cuda.synchronize()
    po = orc.packer("xdelta_hzr", 3, nch, ns, 2)
    for i, blk in enumerate(blocks):
        want = po.compress(blk)
        got = d_dst[i, : int(d_sizes[i])].cpu().numpy().tobytes()
        assert got == want, describe_mismatch(got, want)
    pk.close()


def test_full_size_dense_then_quiet_on_one_handle(api, orc):
    """BASELINE shape (64 x 65536 x int32): a batch of wide random samples (nb escalates to 4, every plane dense: all hzr
    blocks flagged dirty), then quiet synthetic blocks on the same handle (the upper planes turn sparse: the front end
    must overwrite what the dense batch left), then the dense ones again -- every stream against the oracle."""
    import torch

    from rspt_amd import synth

    nch, ns = 64, 65536
    pk = api.new_xdelta_hzr(4, nch, ns, 3)
    po = orc.packer("xdelta_hzr", 4, nch, ns, 3)
    dense = [cases._rand_native(nch, ns, 4, 900 + i, 1 << 29) for i in range(2)]
    quiet = [synth.synth_native(nch, ns, block_index=40 + i).numpy().reshape(-1) for i in range(2)]
    for blocks in (dense, quiet, dense[:1] + quiet[:1]):
        d_src = torch.from_numpy(np.stack(blocks)).cuda()
        d_dst, d_sizes = pk.compress_batch(d_src)
        torch.cuda.synchronize()
        for i, blk in enumerate(blocks):
            want = po.compress(blk)
            got = d_dst[i, : int(d_sizes[i])].cpu().numpy().tobytes()
            assert got == want, describe_mismatch(got, want)
        assert pk.nb == orc.packer_nb(po) == 4
    pk.close()


@pytest.mark.parametrize("nblocks,first", [(128, 896), (1024, 0)])
def test_baseline_config5_scale(api, orc, nblocks, first):
    """BASELINE configs[4]: 1024 independent 12ch x 8192 blocks -- the whole job in ONE launch, and the last rank's shard of
    an 8-GPU run (128 blocks from block 896).  Two hzr blocks per plane (65536 + 32768 bytes) that straddle channels
    (SURVEY D4); thousands of hzr blocks through the work queues and the stream-offset scan.  Every stream against the oracle,
    then the whole batch back through the GPU decoder."""
    import torch

    from rspt_amd import synth

    nch, ns = 12, 8192
    d_src = synth.synth_batch_native(nblocks, nch, ns, first_block=first, device="cuda")
    pk = api.new_xdelta_hzr(4, nch, ns, 3)
    stride = (pk.max_compressed_size + 255) // 256 * 256
    d_dst = torch.empty((nblocks, stride), dtype=torch.uint8, device="cuda")
    d_sizes = torch.empty(nblocks, dtype=torch.int64, device="cuda")
    pk.compress_batch(d_src, d_dst, d_sizes, stride)
    torch.cuda.synchronize()
    sizes = d_sizes.cpu().numpy()
    assert (sizes > 0).all() and (sizes <= stride).all()
    host_src = d_src.cpu().numpy()
    host_dst = d_dst.cpu().numpy()
    po = orc.packer("xdelta_hzr", 4, nch, ns, 3)
    bad = []
    for b in range(nblocks):
        want = po.compress(host_src[b])
        if host_dst[b, : sizes[b]].tobytes() != want:
            bad.append(b)
    assert not bad, "streams differ from the oracle at blocks %s" % bad[:10]
    assert pk.nb == orc.packer_nb(po)
    out, used = pk.decompress_batch(d_dst, nblocks, stride)
    torch.cuda.synchronize()
    assert torch.equal(out, d_src) and torch.equal(used, d_sizes)
    pk.close()


@pytest.mark.parametrize("kind,bps,nch,ns", [("xdelta_hzr", 4, 1100, 70), ("xdelta_hzr", 2, 3000, 33), ("xdelta_hzr", 3, 1302, 80), ("hzr", 1, 1236, 105),
                                              ("hzr", 4, 1024, 64), ("hadamard", 4, 1500, 64), ("dct", 2, 2100, 48), ("xdelta_hzr", 4, 8000, 9)])
def test_blocks_wider_than_a_front_end_tile(api, orc, kind, bps, nch, ns):
    """More channels than a 16-sample tile of the front-end kernels holds in LDS (about a thousand and up; the reference takes
    any count): the wide-block front end -- a 64 x 64 transpose to the planar block (k_wide_planar), then the flat stage over it
    -- gives the oracle's streams, escalation inside the batch included, and the decoder gives the blocks back."""
    import torch

    po = orc.packer(kind, bps, nch, ns, 1)
    pk = api.SignalPacker(kind, bps, nch, ns, 1)
    lim = 1 << (8 * bps - 1)
    for call in range(2):
        amps = [3, min(lim - 1, 1 << 14), min(lim - 1, 1 << 29)] if call == 0 else [min(lim - 1, 1 << 29), 60]
        blocks = [cases._rand_native(nch, ns, bps, 8800 + 10 * call + i, a, walk=bool(i & 1)) for i, a in enumerate(amps)]
        want = [po.compress(b) for b in blocks]
        ref_dec = None
        d_src = torch.from_numpy(np.stack(blocks)).cuda()
        d_dst, d_sizes = pk.compress_batch(d_src)
        torch.cuda.synchronize()
        for i, w in enumerate(want):
            got = d_dst[i, : int(d_sizes[i])].cpu().numpy().tobytes()
            assert got == w, "call %d block %d: %s" % (call, i, describe_mismatch(got, w, 3 * nch if kind in ("dct", "hadamard") else 0))
        if kind == "xdelta_hzr":
            assert pk.nb == orc.packer_nb(po)
        # the last stream through the host-pointer calls, and back
        one = pk.compress(blocks[-1])
        assert one == po.compress(blocks[-1])
        dec, used = pk.decompress(one)
        assert used == len(one) and dec == po.decompress(one)[0]
        if kind == "xdelta_hzr":
            assert dec == blocks[-1].tobytes()
    pk.close()
