"""Pin the CPU restatement (oracle/rspt_oracle.c) against the committed golden
fixtures, which were produced by the compiled reference (tests/golden/make_golden.py).
Runs everywhere (no GPU, no /root/reference)."""
import struct
import zlib

import numpy as np
import pytest

import cases

SLOW = {"ecg12x4096_dct"}  # ~2 s each of O(n^2) cosine sums: still run, listed for the record


def test_crc32c_check_value(orc):
    assert orc.crc32c(b"123456789") == 0xE3069283  # standard CRC-32C check (hzr_crc32c.c:77-97)
    assert orc.crc32c(b"") == 0
    assert orc.crc32c(b"\x00") == 0x527D5351  # the Fill(0) block CRC seen in every all-zero block


def test_survey_kats(orc):
    """Known answers quoted in SURVEY.md 8(c), independent of golden.json."""
    enc = lambda b: orc.hzr_encode(np.frombuffer(b, dtype=np.uint8)).hex()
    assert enc(bytes(100)) == "64000000" + "0000" + "51537d52" + "02" + "00"
    assert enc(b"A" * 10) == "0a000000" + "0000" + "eecd6de1" + "02" + "41"
    assert enc(b"abracadabra abracadabra") == "170000000e00ccc813820186e1184124a39c627ca27dca27da07"


@pytest.mark.parametrize("name", sorted(cases.hzr_kat_inputs().keys()))
def test_hzr_kat(orc, golden, name):
    data = cases.hzr_kat_inputs()[name]
    g = golden["hzr"][name]
    assert zlib.crc32(data.tobytes()) == g["in_crc32"], "test input drifted"
    s = orc.hzr_encode(data)
    assert len(s) == g["size"]
    assert orc.fnv1a(s) == g["fnv1a"] and zlib.crc32(s) == g["crc32"]
    if "stream" in g:
        assert s.hex() == g["stream"]
    dec, used = orc.hzr_decode(s, data.size)
    assert dec == data.tobytes() and used == len(s)
    ok, n = orc.hzr_verify(s)
    assert ok and n == data.size
    assert len(s) <= orc.hzr_max_compressed_size(data.size)


def _names():
    return [c["name"] for c in cases.packer_cases()]


@pytest.mark.parametrize("name", _names())
def test_packer_golden(orc, golden, packer_cases, name):
    c, g = packer_cases[name], golden["packers"][name]
    assert zlib.crc32(c["data"].tobytes()) == g["in_crc32"], "test input drifted"
    pk = orc.packer(c["kind"], c["bps"], c["nch"], c["ns"], c["nb"])
    s = pk.compress(c["data"])
    assert len(s) == g["size"]
    assert orc.fnv1a(s) == g["fnv1a"] and zlib.crc32(s) == g["crc32"]
    if "stream" in g:
        assert s.hex() == g["stream"]
    assert orc.packer_nb(pk) == g["final_nb"]
    assert len(s) <= orc.packer_max_compressed_size(pk)
    dec, used, rc = pk.decompress(s)
    assert rc == 0 and used == len(s)
    assert zlib.crc32(dec) == g["decoded_crc32"]
    if g["lossless"]:
        assert dec == c["data"].tobytes()
    if g.get("prdn") is not None:
        assert abs(orc.prdn(c["data"], dec, c["ns"], c["nch"], c["bps"]) - g["prdn"]) < 1e-9
    pk.close()


def test_fast_verify_gives_same_stream_and_nb(orc, golden, packer_cases):
    """The derived escalation criterion (SURVEY 8 note a-3) == round-trip check."""
    for name, c in packer_cases.items():
        if c["kind"] != "xdelta_hzr" or c["nch"] * c["ns"] > 500000:
            continue
        pk = orc.packer(c["kind"], c["bps"], c["nch"], c["ns"], c["nb"])
        orc.packer_set_fast_verify(pk, True)
        s = pk.compress(c["data"])
        assert orc.fnv1a(s) == golden["packers"][name]["fnv1a"], name
        assert orc.packer_nb(pk) == golden["packers"][name]["final_nb"], name
        pk.close()


def test_nb_state_persists_across_calls(orc):
    """nr_bytes_to_compress_ mutates on escalation and stays (xdelta_hzr.cpp:63-69)."""
    big = cases._rand_native(2, 500, 4, 91, 1 << 20)
    small = cases._rand_native(2, 500, 4, 92, 10)
    pk = orc.packer("xdelta_hzr", 4, 2, 500, 1)
    s_small_before = pk.compress(small)
    nb_small = orc.xdelta_needed_nb(orc.xdelta_forward(orc.native_to_i32(small, 500, 2, 4)), 4, 1)
    assert orc.packer_nb(pk) == nb_small
    pk.compress(big)
    nb = orc.packer_nb(pk)
    v = orc.xdelta_forward(orc.native_to_i32(big, 500, 2, 4))
    assert nb == orc.xdelta_needed_nb(v, 4, 1) and nb > nb_small
    s_small_after = pk.compress(small)
    assert orc.packer_nb(pk) == nb and len(s_small_after) > len(s_small_before)
    # stream structure: method byte + nb chunks
    pos, k = 1, 0
    while pos < len(s_small_after):
        pos += 4 + struct.unpack_from("<I", s_small_after, pos)[0]
        k += 1
    assert k == nb and pos == len(s_small_after)


@pytest.mark.parametrize("name", [c["name"] for c in cases.hadamard_big_cases()])
def test_hadamard_beyond_65536_points_golden(orc, golden, name):
    """ns = 2^k > 65536: the restatement against the real reference's stream and decoded block (fwht.c:4-28 takes any 2^k)"""
    c = {x["name"]: x for x in cases.hadamard_big_cases()}[name]
    g = golden["hadamard_big"][name]
    assert zlib.crc32(c["data"].tobytes()) == g["in_crc32"], "test input drifted"
    pk = orc.packer(c["kind"], c["bps"], c["nch"], c["ns"], c["nb"])
    s = pk.compress(c["data"])
    assert len(s) == g["size"] and orc.fnv1a(s) == g["fnv1a"] and zlib.crc32(s) == g["crc32"]
    dec, used, rc = pk.decompress(s)
    assert rc == 0 and used == len(s) and zlib.crc32(dec) == g["decoded_crc32"]
    pk.close()


@pytest.mark.parametrize("bps,nch,ns,amp", [(4, 8, 2048, 1 << 29), (3, 4, 2048, 1 << 21), (4, 1, 3000, 1 << 29), (4, 3, 257, (1 << 31) - 1)])
def test_dct_beyond_its_range_restatement_equals_the_reference(orc, ref, bps, nch, ns, amp):
    """Out of the dct packer's range the decoded doubles overflow int32 and `(int)x` (signal_packer_dct.cpp:98) is whatever the
    build makes of it: the reference compiled here (oracle/_ref, x86-64: 0x80000000) is the authority, the restatement follows
    it -- and the GPU path follows the restatement (tests/test_gpu_lossy_and_decode.py, same shapes)."""
    for seed in range(3):
        x = cases._rand_native(nch, ns, bps, 9100 + seed, amp, walk=bool(seed & 1))
        po, pr = orc.packer("dct", bps, nch, ns), ref.packer("dct", bps, nch, ns)
        s = po.compress(x)
        assert s == pr.compress(x)
        assert po.decompress(s)[0] == pr.decompress(s)[0]


def test_iir_at_full_scale_restatement_equals_the_reference(orc, ref):
    """the band-pass overshoots past 2^31 on full-scale int32 input: the truncated double is the reference build's (rspt_test.cpp:130)"""
    nch, ns, bps = 5, 5000, 4
    data = cases._rand_native(nch, ns, bps, 424242, (1 << 31) - 1, walk=False)
    for coef in (cases.IIR_BANDPASS, cases.IIR_HIGHPASS):
        a = orc.iir_prefilter(data, bps, nch, ns, coef[0], coef[1], 2000)
        assert a == ref.iir_prefilter(data, bps, nch, ns, coef[0], coef[1], 2000)
        v = orc.native_to_i32(np.frombuffer(a, dtype=np.uint8), ns, nch, bps)
        x = orc.native_to_i32(data, ns, nch, bps)
    # (the case is only worth its name if the overflow happens)
    assert (v == -(1 << 31)).any() or (np.abs(v.astype(np.int64)) > np.abs(x.astype(np.int64)).max()).any()
