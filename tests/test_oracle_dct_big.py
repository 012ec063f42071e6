"""The fp64 restatement used as the checker for dct at ns > 8192 (oracle.py: dct_big_*) against the real
reference where the reference can run (ns <= 8192): SURVEY 8(d)'s gate, and in practice identical streams."""
import zlib

import numpy as np
import pytest

from rspt_amd import synth

PRDN_TOL = 0.05
CR_TOL = 0.01


@pytest.mark.parametrize("bps,nch,ns,ecg", [(4, 3, 256, False), (4, 2, 1000, True), (3, 4, 2048, True), (4, 2, 4096, True)])
def test_fp64_restatement_vs_reference(orc, ref, bps, nch, ns, ecg):
    data = synth.synth_native(nch, ns, block_index=5, bps=bps, ecg=ecg).numpy().reshape(-1)
    pr = ref.packer("dct", bps, nch, ns)
    s_ref = pr.compress(data)
    d_ref, _, _ = pr.decompress(s_ref)
    s_big, _ = orc.dct_big_compress(data, bps, nch, ns)
    d_big, used = orc.dct_big_decompress(s_big, bps, nch, ns)
    assert used == len(s_big)
    assert abs(len(s_big) / len(s_ref) - 1) <= CR_TOL
    p_ref = orc.prdn(data, d_ref, ns, nch, bps)
    assert abs(orc.prdn(data, d_big, ns, nch, bps) - p_ref) <= PRDN_TOL
    # cross decode: fp64 inverse of the reference's stream
    d_x, _ = orc.dct_big_decompress(s_ref, bps, nch, ns)
    assert abs(orc.prdn(data, d_x, ns, nch, bps) - p_ref) <= PRDN_TOL
    a = np.frombuffer(d_x, dtype=np.uint8)
    b = np.frombuffer(d_ref, dtype=np.uint8)
    assert a.size == b.size


def test_big_handle_refuses_table_paths(orc):
    pk = orc.packer("dct", 4, 1, 16384)
    with pytest.raises(RuntimeError):
        pk.compress(np.zeros(4 * 16384, dtype=np.uint8))
    pk.close()


def test_fp64_restatement_equals_reference_at_16384(orc, golden):
    """ns = 16384: beyond the GPU build's dense-table limit, still within the reference's reach (1 GiB table).  The fixture
    holds the REAL reference's stream (tests/golden/make_golden.py); the fp64 restatement reproduces it byte for byte."""
    import cases

    for c in cases.dct_big_cases():
        g = golden["dct_big"][c["name"]]
        want = bytes.fromhex(g["stream"])
        got, _ = orc.dct_big_compress(c["data"], c["bps"], c["nch"], c["ns"])
        assert got == want
        dec, used = orc.dct_big_decompress(want, c["bps"], c["nch"], c["ns"])
        assert used == len(want)
        assert abs(orc.prdn(c["data"], dec, c["ns"], c["nch"], c["bps"]) - g["prdn"]) <= PRDN_TOL


def test_table_restatement_equals_reference_at_10000(orc, golden):
    """ns = 10000: not a power of two and beyond 8192 -- the GPU build takes the reference's dense table there too (any ns up
    to the reach of the reference's int table index, 32768).  The C restatement reproduces the REAL reference's stream."""
    import cases

    for c in cases.dct_dense_big_cases():
        g = golden["dct_dense_big"][c["name"]]
        want = bytes.fromhex(g["stream"])
        po = orc.packer("dct", c["bps"], c["nch"], c["ns"])
        got = po.compress(c["data"])
        assert got == want
        dec, used, _ = po.decompress(want)
        assert used == len(want) and zlib.crc32(dec) == g["decoded_crc32"]
