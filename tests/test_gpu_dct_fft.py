"""GPU parity for the dct packer beyond the dense cosine table (ns > 8192, BASELINE config 4).

The reference cannot run there (n x n float table), so the checker is the fp64 restatement in
oracle/oracle.py (dct_big_*), itself compared with the real reference at ns <= 8192 in
tests/test_oracle_dct_big.py.  Gate: |PRDN_gpu - PRDN_oracle| <= 0.05 percentage points and
|CR_gpu / CR_oracle - 1| <= 1 %.  With RSPT_HIP_DCT_FORCE_FFT the same FFT kernels run at small ns, where the stream
of the bit-exact table path (== the reference's) is available for comparison."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PRDN_TOL = 0.05
CR_TOL = 0.01


@pytest.fixture(scope="module")
def api():
    from rspt_amd import api as a

    assert a.lib().rspt_hip_device_count() > 0, "no gfx950 device visible"
    return a


def _block(nch, ns, idx, bps=4, ecg=True):
    from rspt_amd import synth

    return synth.synth_native(nch, ns, block_index=idx, bps=bps, ecg=ecg).numpy().reshape(-1)


def _coeff_mismatch(orc, a, b, bps, nch, ns):
    import ctypes as C

    def coeffs(stream):
        s = np.frombuffer(stream + b"\0" * 16, dtype=np.uint8).copy()
        co = np.zeros((nch, ns), dtype=np.int32)
        me = np.zeros(nch, dtype=np.int32)
        used = C.c_size_t(0)
        pk = orc.packer("dct", bps, nch, ns)
        f = orc.lib.orc_packer_decompress_coeffs
        f.restype = C.c_int
        rc = f(C.c_void_p(pk._h), s.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(used), co.ctypes.data_as(C.POINTER(C.c_int32)),
               me.ctypes.data_as(C.POINTER(C.c_int32)))
        pk.close()
        assert rc == 0
        return co, me

    ca, ma = coeffs(a)
    cb, mb = coeffs(b)
    assert (ma == mb).all()
    d = np.abs(ca.astype(np.int64) - cb.astype(np.int64))
    return float((d != 0).mean()), int(d.max())


@pytest.mark.parametrize("nch,ns", [(3, 16), (2, 64), (3, 256), (2, 512), (5, 1024), (2, 2048), (4, 4096), (2, 8192)])
def test_forced_fft_path_vs_reference_arithmetic(api, orc, nch, ns, monkeypatch):
    data = _block(nch, ns, 3)
    want = orc.packer("dct", 4, nch, ns).compress(data)  # the reference's arithmetic (pinned in test_oracle_vs_ref)
    pk = api.SignalPacker(api.KIND_DCT | api.DCT_FORCE_FFT, 4, nch, ns, 2)  # the FFT kernels where the table path would run
    got = pk.compress(data)
    frac, dmax = _coeff_mismatch(orc, got, want, 4, nch, ns)
    assert dmax <= 1 and frac <= 2e-3, (frac, dmax)  # only truncation-boundary flips
    assert abs(len(got) / len(want) - 1) <= CR_TOL
    po = orc.packer("dct", 4, nch, ns)
    dec_w, _, _ = po.decompress(want)
    dec_g_ref, used, _ = po.decompress(got)
    assert used == len(got)
    p_w = orc.prdn(data, dec_w, ns, nch, 4)
    assert abs(orc.prdn(data, dec_g_ref, ns, nch, 4) - p_w) <= PRDN_TOL
    # the FFT inverse on the reference's stream
    dec_g, used = pk.decompress(want)
    assert used == len(want)
    assert abs(orc.prdn(data, dec_g, ns, nch, 4) - p_w) <= PRDN_TOL
    a = np.frombuffer(dec_g, dtype="<i4").astype(np.int64)
    b = np.frombuffer(dec_w, dtype="<i4").astype(np.int64)
    assert np.abs(a - b).max() <= 1
    pk.close()


@pytest.mark.parametrize("bps,nch,ns,nblocks", [(4, 3, 16384, 2), (3, 2, 32768, 1), (4, 64, 65536, 1), (4, 2, 262144, 1), (4, 1, 1048576, 1), (4, 1, 4194304, 1)])
def test_dct_large_ns(api, orc, bps, nch, ns, nblocks):
    import torch

    pk = api.new_dct(bps, nch, ns)
    blocks = [_block(nch, ns, 11 + i, bps=bps) for i in range(nblocks)]
    d_src = torch.from_numpy(np.stack(blocks)).cuda()
    stride = (pk.max_compressed_size + 255) // 256 * 256
    d_dst = torch.zeros((nblocks, stride), dtype=torch.uint8, device="cuda")
    d_sizes = torch.zeros(nblocks, dtype=torch.int64, device="cuda")
    pk.compress_batch(d_src, d_dst, d_sizes, stride)
    torch.cuda.synchronize()
    for i, data in enumerate(blocks):
        got = d_dst[i, : int(d_sizes[i])].cpu().numpy().tobytes()
        want, _ = orc.dct_big_compress(data, bps, nch, ns)
        assert abs(len(got) / len(want) - 1) <= CR_TOL
        frac, dmax = _coeff_mismatch(orc, got, want, bps, nch, ns)
        assert dmax <= 1 and frac <= 2e-3, (frac, dmax)
        dec_w, _ = orc.dct_big_decompress(want, bps, nch, ns)
        dec_o, used = orc.dct_big_decompress(got, bps, nch, ns)
        assert used == len(got)
        p_w = orc.prdn(data, dec_w, ns, nch, bps)
        assert abs(orc.prdn(data, dec_o, ns, nch, bps) - p_w) <= PRDN_TOL
        dec_g, used = pk.decompress(got)
        assert used == len(got)
        assert abs(orc.prdn(data, dec_g, ns, nch, bps) - p_w) <= PRDN_TOL
        assert p_w < 20.0  # sanity: it is a usable reconstruction
    pk.close()


def test_dct_16384_against_the_reference_stream(api, orc, golden):
    """The FFT path held to the REAL reference (fixtures generated from oracle/_ref at ns = 16384 and at ns = 32768, the
    reference's own limit -- one octave below BASELINE config 4): CR / PRDN gate of SURVEY 8d, coefficients equal except
    truncation-boundary flips of 1."""
    import cases

    for c in cases.dct_big_cases():
        g = golden["dct_big"][c["name"]]
        want = bytes.fromhex(g["stream"])
        bps, nch, ns = c["bps"], c["nch"], c["ns"]
        pk = api.new_dct(bps, nch, ns)
        got = pk.compress(c["data"])
        assert abs(len(got) / len(want) - 1) <= CR_TOL
        frac, dmax = _coeff_mismatch(orc, got, want, bps, nch, ns)
        assert dmax <= 1 and frac <= 2e-3, (frac, dmax)
        # the GPU inverse on the reference's stream, and on its own
        for s in (want, got):
            dec, used = pk.decompress(s)
            assert used == len(s)
            assert abs(orc.prdn(c["data"], dec, ns, nch, bps) - g["prdn"]) <= PRDN_TOL
        pk.close()


def test_dct_dense_table_beyond_8192(api, orc, golden):
    """ns = 10000 (not a power of two): the reference's dense table on the GPU -- the REAL reference's stream byte for byte,
    and its decode."""
    import zlib

    import cases

    for c in cases.dct_dense_big_cases():
        g = golden["dct_dense_big"][c["name"]]
        want = bytes.fromhex(g["stream"])
        pk = api.new_dct(c["bps"], c["nch"], c["ns"])
        got = pk.compress(c["data"])
        assert got == want
        dec, used = pk.decompress(want)
        assert used == len(want) and zlib.crc32(dec) == g["decoded_crc32"]
        pk.close()


def test_dct_unsupported_sizes(api):
    with pytest.raises(Exception):
        api.new_dct(4, 1, 32768 + 8)  # beyond the reference's own reach (int table index) and not a power of two
    with pytest.raises(Exception):
        api.new_dct(4, 1, 1 << 23)  # the FFT path stops at 2^22
