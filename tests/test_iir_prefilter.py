"""The IIR pre-filter stage: the step in front of the packers in the reference's own pipeline
(lib_rspt_test/rspt_test.cpp:116-136, lib_rspt/lib_filter/iir_filter.cpp:46-116).

CPU: the restatement (oracle/rspt_oracle.c: orc_iir_prefilter_native) against the golden fixtures generated from the
real reference, and against the reference itself where oracle/_ref is present.
GPU (-m gpu): rspt_hip_iir_prefilter_batch_dev against the same fixtures -- bit-exact: the filtered block and the
xdelta_hzr stream of the filtered block, which is what the harness compresses next."""
import zlib

import numpy as np
import pytest

import cases


@pytest.fixture(scope="module")
def iir_cases():
    return {c["name"]: c for c in cases.iir_cases()}


NAMES = [c["name"] for c in cases.iir_cases()]


@pytest.mark.parametrize("name", NAMES)
def test_restatement_matches_golden(orc, golden, iir_cases, name):
    c, g = iir_cases[name], golden["iir"][name]
    assert zlib.crc32(c["data"].tobytes()) == g["in_crc32"]
    filt = orc.iir_prefilter(c["data"], c["bps"], c["nch"], c["ns"], c["n"], c["d"], c["init"])
    assert zlib.crc32(filt) == g["filtered_crc32"] and orc.fnv1a(filt) == g["filtered_fnv1a"]
    s = orc.packer("xdelta_hzr", c["bps"], c["nch"], c["ns"], 3).compress(np.frombuffer(filt, dtype=np.uint8))
    assert len(s) == g["xdelta_size"] and orc.fnv1a(s) == g["xdelta_fnv1a"]


@pytest.mark.parametrize("name", NAMES)
def test_restatement_matches_reference(orc, ref, iir_cases, name):
    c = iir_cases[name]
    assert orc.iir_prefilter(c["data"], c["bps"], c["nch"], c["ns"], c["n"], c["d"], c["init"]) == ref.iir_prefilter(
        c["data"], c["bps"], c["nch"], c["ns"], c["n"], c["d"], c["init"]
    )


def test_shared_filter_state_matters(orc, iir_cases):
    """the harness shares ONE filter object between the channels: on the 24-bit recording the carried state moves
    thousands of samples -- which is why the bit-exact GPU mode is the serial one"""
    c = iir_cases["ds3x20000_i24_bandpass"]
    a = orc.iir_prefilter(c["data"], c["bps"], c["nch"], c["ns"], c["n"], c["d"], c["init"], shared_state=True)
    b = orc.iir_prefilter(c["data"], c["bps"], c["nch"], c["ns"], c["n"], c["d"], c["init"], shared_state=False)
    pa = orc.native_to_i32(np.frombuffer(a, dtype=np.uint8), c["ns"], c["nch"], c["bps"]).astype(np.int64)
    pb = orc.native_to_i32(np.frombuffer(b, dtype=np.uint8), c["ns"], c["nch"], c["bps"]).astype(np.int64)
    assert (pa[0] == pb[0]).all()  # the first channel starts from a fresh filter either way
    assert (pa[1:] != pb[1:]).sum() > 1000


@pytest.fixture(scope="module")
def api():
    from rspt_amd import api as a

    assert a.lib().rspt_hip_device_count() > 0, "no gfx950 device visible"
    return a


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_prefilter_bit_exact(api, orc, golden, iir_cases, name):
    import torch

    c, g = iir_cases[name], golden["iir"][name]
    pk = api.new_xdelta_hzr(c["bps"], c["nch"], c["ns"], 3)
    # three copies of the block in one batch: every block gets its own filter object, as separate test_data() calls do
    d_buf = torch.from_numpy(np.stack([c["data"]] * 3)).cuda()
    pk.iir_prefilter_batch(d_buf, c["n"], c["d"], c["init"])
    torch.cuda.synchronize()
    for b in range(3):
        filt = d_buf[b].cpu().numpy().tobytes()
        assert zlib.crc32(filt) == g["filtered_crc32"], "block %d differs from the reference's filtered block" % b
    # ... and what the harness does next: xdelta_hzr on the filtered block, on the device
    d_dst, d_sizes = pk.compress_batch(d_buf)
    torch.cuda.synchronize()
    s = d_dst[0, : int(d_sizes[0])].cpu().numpy().tobytes()
    assert len(s) == g["xdelta_size"] and orc.fnv1a(s) == g["xdelta_fnv1a"]
    pk.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_prefilter_per_channel_mode(api, orc, iir_cases, name):
    """a fresh filter per channel (one thread per channel): the same arithmetic, so bit-exact with the restatement run that way"""
    import torch

    c = iir_cases[name]
    pk = api.new_xdelta_hzr(c["bps"], c["nch"], c["ns"], 3)
    d_buf = torch.from_numpy(np.stack([c["data"]] * 2)).cuda()
    pk.iir_prefilter_batch(d_buf, c["n"], c["d"], c["init"], per_channel=True)
    torch.cuda.synchronize()
    want = orc.iir_prefilter(c["data"], c["bps"], c["nch"], c["ns"], c["n"], c["d"], c["init"], shared_state=False)
    for b in range(2):
        assert d_buf[b].cpu().numpy().tobytes() == want
    pk.close()


@pytest.mark.gpu
def test_gpu_prefilter_rejects_bad_orders(api):
    import torch

    pk = api.new_xdelta_hzr(4, 2, 64, 3)
    d_buf = torch.zeros(4 * 2 * 64, dtype=torch.uint8, device="cuda")
    for k in (1, 6):
        with pytest.raises(api.RsptHipError):
            pk.iir_prefilter_batch(d_buf, [1.0] * k, [1.0] * k)
    pk.close()


# ---- the full-size block: the size at which fused multiply-adds once flipped output counts (no small fixture ever did) ----
BIG_MODES = [("shared", 2000), ("per_channel", 2000), ("shared", 0), ("per_channel", 0)]


@pytest.mark.parametrize("mode,init", BIG_MODES)
def test_restatement_matches_reference_at_full_size(orc, golden, mode, init):
    c = cases.iir_big_case()
    g = golden["iir_big"][c["name"]]
    data = cases.iir_big_data(c)
    assert zlib.crc32(data.tobytes()) == g["in_crc32"]
    filt = orc.iir_prefilter(data, c["bps"], c["nch"], c["ns"], c["n"], c["d"], init, shared_state=mode == "shared")
    want = g["modes"]["%s_init%d" % (mode, init)]
    assert zlib.crc32(filt) == want["crc32"] and orc.fnv1a(filt) == want["fnv1a"]


@pytest.mark.gpu
@pytest.mark.parametrize("mode,init", BIG_MODES)
def test_gpu_prefilter_full_size_block(api, golden, mode, init):
    """64 ch x 65536 int32 through rspt_hip_iir_prefilter_batch_dev, both modes, the six-wave kernel (history
    initialisation) and the one-thread-per-channel kernel (none): CRCs of the real reference's filtered block"""
    import torch

    c = cases.iir_big_case()
    want = golden["iir_big"][c["name"]]["modes"]["%s_init%d" % (mode, init)]
    data = cases.iir_big_data(c)
    pk = api.new_xdelta_hzr(c["bps"], c["nch"], c["ns"], 3)
    d_buf = torch.from_numpy(np.stack([data, data])).cuda()
    pk.iir_prefilter_batch(d_buf, c["n"], c["d"], init, per_channel=mode == "per_channel")
    torch.cuda.synchronize()
    for b in range(2):
        assert zlib.crc32(d_buf[b].cpu().numpy().tobytes()) == want["crc32"], "block %d" % b
    pk.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["shared", "per_channel"])
def test_gpu_prefilter_at_full_scale(api, orc, mode):
    """int32 samples near full scale: the band-pass overshoots past 2^31 in places, where `(int)double` is what the reference's
    x86-64 build makes of it (0x80000000), not the GPU's saturating conversion (rspt_test.cpp:130; found by tools/soak.py)"""
    import torch

    nch, ns, bps = 5, 5000, 4
    data = cases._rand_native(nch, ns, bps, 424242, (1 << 31) - 1, walk=False)
    pk = api.new_xdelta_hzr(bps, nch, ns, 3)
    for coef in (cases.IIR_BANDPASS, cases.IIR_HIGHPASS):
        d_buf = torch.from_numpy(np.stack([data, data])).cuda()
        pk.iir_prefilter_batch(d_buf, coef[0], coef[1], 2000, per_channel=mode == "per_channel")
        torch.cuda.synchronize()
        want = orc.iir_prefilter(data, bps, nch, ns, coef[0], coef[1], 2000, shared_state=mode == "shared")
        assert d_buf[0].cpu().numpy().tobytes() == want and d_buf[1].cpu().numpy().tobytes() == want
    pk.close()
