"""Host-logic tests: the parallel formulations the HIP kernels use
(tools/kernel_model.py) against the oracle, on the CPU."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import kernel_model as km  # noqa: E402

import cases


def _blocks():
    k = cases.hzr_kat_inputs()
    out = [k[n][:65536] for n in ("mixed40", "abracadabra", "runs_small", "uniform4_x4096", "fib20", "all_symbols_twice")]
    out.append(k["runs_cap"][:65536])
    out.append(k["runs_cap"][16000:16000 + 65536])
    out.append(k["run_to_block_end"][:65536])
    out.append(k["sparse_ones"][:65536])
    out.append(k["mostly_zero_geo"][:20000])
    z = np.zeros(65536, dtype=np.uint8)
    z[0] = 1
    out.append(z)  # 65535 zeros after one byte: three capped runs + remainder
    z2 = np.zeros(50001, dtype=np.uint8)
    z2[-1] = 9
    out.append(z2)
    r = np.random.RandomState(5)
    for n in (1, 15, 16, 17, 31, 33, 1000, 4099):
        out.append(np.where(r.rand(n) < 0.7, 0, r.randint(1, 5, n)).astype(np.uint8))
    return out


@pytest.mark.parametrize("i", range(len(_blocks())))
def test_granule_tokenizer_equals_serial(i):
    b = _blocks()[i]
    assert km.granule_tokens(b) == km.reference_tokens(b)


def test_histogram_equals_oracle(orc):
    for b in _blocks():
        hist = np.zeros(261, dtype=np.int64)
        for s, _, _ in km.granule_tokens(b):
            hist[s] += 1
        oh, mode, plen = orc.hzr_block_stats(b)
        assert (hist == oh).all()


def test_block_payload_equals_oracle(orc):
    n = 0
    for b in _blocks():
        oh, mode, plen = orc.hzr_block_stats(b)
        if mode != 1:
            continue
        s = orc.hzr_encode(b)
        assert km.encode_block_model(b) == s[11:], len(b)
        n += 1
    assert n >= 8


def test_gf_shift_identities():
    one = 0x80000000
    assert km.gf_mul(one, 0x12345678) == 0x12345678
    a, b = b"hello world, ", b"parallel crc!"
    assert km.raw_crc_bytes(a + b) == km.gf_mul(km.raw_crc_bytes(a), km.x_pow_bytes(len(b))) ^ km.raw_crc_bytes(b)


def test_parallel_crc_equals_oracle(orc):
    r = np.random.RandomState(11)
    for n in (1, 2, 3, 4, 5, 12, 15, 16, 17, 28, 29, 1000, 1023, 1024, 1025, 5000):
        m = r.randint(0, 256, n).astype(np.uint8).tobytes()
        assert km.crc_parallel(m, lanes=4, waves=2) == orc.crc32c(m), n
    m = r.randint(0, 256, 3000).astype(np.uint8).tobytes()
    assert km.crc_parallel(m, lanes=64, waves=16) == orc.crc32c(m)


def test_strided_crc_equals_oracle(orc):
    """the big encoder's CRC: word-strided lanes over the unpadded image (hzr_kernels.hip: encode_block)"""
    r = np.random.RandomState(12)
    for n in (1, 2, 3, 4, 5, 12, 15, 16, 17, 28, 29, 63, 64, 65, 255, 256, 257, 1000, 1023, 1024, 1025, 5000):
        m = r.randint(0, 256, n).astype(np.uint8).tobytes()
        assert km.crc_strided(m, nthr=8) == orc.crc32c(m), n
        assert km.crc_strided(m, nthr=64) == orc.crc32c(m), n
    m = r.randint(0, 256, 9000).astype(np.uint8).tobytes()
    assert km.crc_strided(m, nthr=1024) == orc.crc32c(m)
