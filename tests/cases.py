"""Deterministic inputs shared by the golden-fixture generator and the tests.

Everything here is integer arithmetic (or the repo's shipped data files), so the
same bytes come out on every box.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from rspt_amd import synth  # noqa: E402


def hash_bytes(n, seed, lo=0, hi=256):
    """n pseudo-random bytes in [lo,hi) from a counter hash (stable everywhere)."""
    i = np.arange(n, dtype=np.uint64) + np.uint64((seed * 0x9E3779B9) & 0xFFFFFFFF)
    x = i & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    return (lo + (x % np.uint64(hi - lo))).astype(np.uint8)


def hash_i32(n, seed, amplitude):
    """n pseudo-random int32 in [-amplitude, amplitude)."""
    i = np.arange(n, dtype=np.uint64) + np.uint64((seed * 0x85EBCA6B) & 0xFFFFFFFF)
    x = i & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    return (x.astype(np.int64) % (2 * amplitude) - amplitude).astype(np.int32)


def xorshift32_bytes(n, seed=1):
    out = np.zeros(n, dtype=np.uint8)
    x = seed
    for i in range(n):
        x ^= (x << 13) & 0xFFFFFFFF
        x ^= x >> 17
        x ^= (x << 5) & 0xFFFFFFFF
        out[i] = x & 0xFF
    return out


def _runs(lengths, sep=7):
    """zero runs of the given lengths separated by a non-zero byte."""
    parts = []
    for k, z in enumerate(lengths):
        parts.append(np.zeros(z, dtype=np.uint8))
        parts.append(np.array([1 + (sep + k) % 255], dtype=np.uint8))
    return np.concatenate(parts)


def _fib_counts(k):
    """symbols with Fibonacci counts -> maximally deep Huffman tree."""
    a, b, out = 1, 1, []
    for s in range(k):
        out.append(np.full(a, 1 + s, dtype=np.uint8))
        a, b = b, a + b
    return np.concatenate(out)


def hzr_kat_inputs():
    """name -> bytes; raw hzr_encode known-answer inputs (SURVEY 8c list + extras)."""
    mixed40 = np.zeros(40, dtype=np.uint8)
    mixed40[0], mixed40[3], mixed40[7], mixed40[15], mixed40[17], mixed40[39] = 1, 2, 3, 4, 1, 5
    z140k = np.zeros(140000, dtype=np.uint8)
    z140k[70000] = 7
    c = {
        "zeros100": np.zeros(100, dtype=np.uint8),
        "A10": np.full(10, 0x41, dtype=np.uint8),
        "abracadabra": np.frombuffer(b"abracadabra abracadabra", dtype=np.uint8),
        "mixed40": mixed40,
        "xorshift300": xorshift32_bytes(300, 1),
        "zeros140k_one7": z140k,
        "one_byte": np.array([9], dtype=np.uint8),
        "one_zero": np.array([0], dtype=np.uint8),
        "two_distinct": np.array([5, 6], dtype=np.uint8),
        "zero_then_one": np.array([0, 1], dtype=np.uint8),
        "uniform4_x4096": hash_bytes(4096, 11, 1, 5),
        "uniform16_x70000": hash_bytes(70000, 12, 0, 16),
        "uniform256_x65536": hash_bytes(65536, 13),
        "uniform200_x65537": hash_bytes(65537, 14, 0, 200),
        "all_symbols_once": np.arange(256, dtype=np.uint8),
        "all_symbols_twice": np.concatenate([np.arange(256, dtype=np.uint8)] * 2),
        "fib20": _fib_counts(20),
        "fib22": _fib_counts(22),
        "runs_small": _runs([1, 2, 3, 6, 7, 22, 23, 278, 279, 1000]),
        "runs_cap": _runs([16661, 16662, 16663, 33324, 33325]),
        "run_cross_block": np.concatenate([hash_bytes(65530, 15, 1, 256), np.zeros(12, dtype=np.uint8), hash_bytes(100, 16, 1, 256)]),
        "run_to_block_end": np.concatenate([hash_bytes(60000, 17, 1, 9), np.zeros(5536 + 300, dtype=np.uint8), hash_bytes(50, 18, 1, 9)]),
        "sparse_ones": (hash_bytes(200000, 19, 0, 64) == 0).astype(np.uint8),
        "mostly_zero_geo": np.where(hash_bytes(131072, 20, 0, 8) == 0, hash_bytes(131072, 21, 1, 4), 0).astype(np.uint8),
        "three_blocks_mixed": np.concatenate([hash_bytes(65536, 22, 0, 3), np.full(65536, 0xAA, dtype=np.uint8), hash_bytes(777, 23)]),
    }
    return {k: np.ascontiguousarray(v) for k, v in c.items()}


def _sine_native(ns, bps):
    if bps == 4:
        return synth.sine_readme(ns, 1000.0, np.int32).view(np.uint8)
    if bps == 2:
        return synth.sine_readme(ns, 1000.0, np.int16).view(np.uint8)
    return synth.sine_readme(ns, 100.0, np.int8).view(np.uint8)


def _rand_native(nch, ns, bps, seed, amplitude, walk=False):
    x = hash_i32(nch * ns, seed, amplitude).astype(np.int64)
    if walk:
        x = np.cumsum(x.reshape(ns, nch), axis=0).reshape(-1)
    lim = 1 << (8 * bps - 1)
    x = ((x + lim) % (2 * lim) - lim).astype(np.int32)
    b = x.view(np.uint8).reshape(-1, 4)[:, :bps]
    return np.ascontiguousarray(b).reshape(-1)


def packer_cases():
    """list of dicts: name, kind, bps, nch, ns, nb, data (np.uint8), store ('full'|'hash')."""
    ecg = np.frombuffer(synth.ecg_12ch_i32(), dtype=np.uint8)
    ds = np.frombuffer(synth.data_stream_3ch_i24(), dtype=np.uint8)
    C = []

    def add(name, kind, bps, nch, ns, nb, data, store="hash"):
        data = np.ascontiguousarray(data[: bps * nch * ns])
        assert data.size == bps * nch * ns, (name, data.size)
        C.append(dict(name=name, kind=kind, bps=bps, nch=nch, ns=ns, nb=nb, data=data, store=store))

    # README / rspt_test.cpp test_5, test_2..4
    add("readme_sine_xdelta_nb3", "xdelta_hzr", 4, 1, 8192, 3, _sine_native(8192, 4), "full")
    add("readme_sine_xdelta_nb1", "xdelta_hzr", 4, 1, 8192, 1, _sine_native(8192, 4), "full")
    add("readme_sine_hzr", "hzr", 4, 1, 8192, 4, _sine_native(8192, 4))
    add("readme_sine_hadamard", "hadamard", 4, 1, 8192, 3, _sine_native(8192, 4), "full")
    add("sine16384_i32_xdelta", "xdelta_hzr", 4, 1, 16384, 3, _sine_native(16384, 4))
    add("sine16384_i32_hadamard", "hadamard", 4, 1, 16384, 3, _sine_native(16384, 4), "full")
    add("sine4096_i32_dct", "dct", 4, 1, 4096, 2, _sine_native(4096, 4), "full")
    add("sine16384_i16_xdelta", "xdelta_hzr", 2, 1, 16384, 3, _sine_native(16384, 2))
    add("sine16384_i16_hadamard", "hadamard", 2, 1, 16384, 3, _sine_native(16384, 2))
    add("sine4096_i16_dct", "dct", 2, 1, 4096, 2, _sine_native(4096, 2))
    add("sine16384_i8_xdelta", "xdelta_hzr", 1, 1, 16384, 3, _sine_native(16384, 1))
    add("sine16384_i8_hadamard", "hadamard", 1, 1, 16384, 3, _sine_native(16384, 1))
    # repo ECG file (rspt_test.cpp test_7) -- BASELINE config C2
    for nb in (1, 2, 3, 4):
        add("ecg12x34199_xdelta_nb%d" % nb, "xdelta_hzr", 4, 12, 34199, nb, ecg)
    add("ecg12x34199_hzr", "hzr", 4, 12, 34199, 4, ecg)
    add("ecg12x16384_hadamard", "hadamard", 4, 12, 16384, 3, ecg)
    add("ecg12x4096_dct", "dct", 4, 12, 4096, 2, ecg)
    add("ecg12x8192_xdelta", "xdelta_hzr", 4, 12, 8192, 3, ecg)
    # repo 24-bit stream (rspt_test.cpp test_1)
    add("ds3x20000_i24_xdelta", "xdelta_hzr", 3, 3, 20000, 3, ds)
    add("ds3x20000_i24_hzr", "hzr", 3, 3, 20000, 4, ds)
    add("ds3x16384_i24_hadamard", "hadamard", 3, 3, 16384, 3, ds)
    add("ds3x4096_i24_dct", "dct", 3, 3, 4096, 2, ds)
    # synthetic integer generator
    import torch  # noqa: F401

    s64 = synth.synth_native(64, 65536, 0).numpy()
    add("synth64x65536_xdelta", "xdelta_hzr", 4, 64, 65536, 3, s64)
    add("synth64x65536_hzr", "hzr", 4, 64, 65536, 4, s64)
    add("synth64x65536_hadamard", "hadamard", 4, 64, 65536, 3, s64)
    for b in (0, 1, 1023):
        add("synth12x8192_b%d_xdelta" % b, "xdelta_hzr", 4, 12, 8192, 3, synth.synth_native(12, 8192, b).numpy())
    add("synth12x8192_ecg_xdelta", "xdelta_hzr", 4, 12, 8192, 3, synth.synth_native(12, 8192, 5, ecg=True).numpy())
    add("synth8x1024_dct", "dct", 4, 8, 1024, 2, synth.synth_native(8, 1024, 2).numpy())
    # escalation 1->2->3->4, all sample widths, ragged shapes, random walks
    add("esc_i32_big", "xdelta_hzr", 4, 3, 1000, 1, _rand_native(3, 1000, 4, 31, 1 << 30))
    add("esc_i32_mid", "xdelta_hzr", 4, 5, 777, 1, _rand_native(5, 777, 4, 32, 1 << 20))
    add("esc_i32_small", "xdelta_hzr", 4, 2, 300, 1, _rand_native(2, 300, 4, 33, 1 << 12))
    add("esc_i24_walk", "xdelta_hzr", 3, 4, 5000, 1, _rand_native(4, 5000, 3, 34, 3000, walk=True))
    add("esc_i16_big", "xdelta_hzr", 2, 7, 333, 1, _rand_native(7, 333, 2, 35, 1 << 15))
    add("esc_i16_nb3", "xdelta_hzr", 2, 2, 4096, 3, _rand_native(2, 4096, 2, 36, 1 << 15))
    add("i8_nb1", "xdelta_hzr", 1, 9, 1001, 1, _rand_native(9, 1001, 1, 37, 100))
    add("i8_nb4", "xdelta_hzr", 1, 2, 70000, 4, _rand_native(2, 70000, 1, 38, 128))
    add("tiny_1x1", "xdelta_hzr", 4, 1, 1, 3, _rand_native(1, 1, 4, 39, 1000), "full")
    add("tiny_3x1", "xdelta_hzr", 4, 3, 1, 3, _rand_native(3, 1, 4, 40, 1000), "full")
    add("tiny_2x2", "xdelta_hzr", 4, 2, 2, 2, _rand_native(2, 2, 4, 41, 1000), "full")
    add("tiny_1x3_hzr", "hzr", 2, 1, 3, 4, _rand_native(1, 3, 2, 42, 1000), "full")
    add("ragged_5x13107_hzr", "hzr", 4, 5, 13107, 4, _rand_native(5, 13107, 4, 43, 50, walk=True))
    add("ragged_3x43691_xdelta", "xdelta_hzr", 4, 3, 43691, 2, _rand_native(3, 43691, 4, 44, 20, walk=True))
    add("const_4x5000_xdelta", "xdelta_hzr", 4, 4, 5000, 3, np.zeros(4 * 4 * 5000, dtype=np.uint8))
    add("const_4x4096_hadamard", "hadamard", 4, 4, 4096, 3, np.zeros(4 * 4 * 4096, dtype=np.uint8))
    add("had_negmean_2x64", "hadamard", 4, 2, 64, 3, _rand_native(2, 64, 4, 45, 1000) , "full")
    add("had_i16_8x2048", "hadamard", 2, 8, 2048, 3, _rand_native(8, 2048, 2, 46, 200, walk=True))
    # negative-sum mean with ns not a power of two (average_32 quirk, utils.cpp:30-40)
    neg = (-np.abs(hash_i32(3 * 100, 47, 500)) - 3).astype(np.int32).view(np.uint8)
    add("dct_negmean_3x100", "dct", 4, 3, 100, 2, neg, "full")
    add("dct_rand_2x37", "dct", 4, 2, 37, 2, _rand_native(2, 37, 4, 48, 3000), "full")
    add("dct_i16_4x128", "dct", 2, 4, 128, 2, _rand_native(4, 128, 2, 49, 3000), "full")
    return C


def dct_big_cases():
    """dct beyond the dense-table limit of the GPU build (ns > 8192) where the REAL reference can still run (n x n float
    table: 1 GiB at 16384; ~20 s per case here).  The reference's full stream is the fixture: the FFT path is held to it."""
    return [dict(name="synth2x16384_dct", kind="dct", bps=4, nch=2, ns=16384, nb=2,
                 data=np.ascontiguousarray(synth.synth_native(2, 16384, block_index=11, ecg=True).numpy().reshape(-1))),
            # the reference's own limit: at ns = 32768 its table index (2x+1)*i still fits an int (signal_packer_dct.cpp:60-74) and
            # the table is 4 GiB -- one octave below BASELINE config 4 (ns = 65536), where only the restatement can check
            dict(name="synth1x32768_dct", kind="dct", bps=4, nch=1, ns=32768, nb=2,
                 data=np.ascontiguousarray(synth.synth_native(1, 32768, block_index=13, ecg=True).numpy().reshape(-1)))]


def hadamard_big_cases():
    """hadamard beyond one workgroup's reach (ns = 2^k > 65536): the reference's transform takes any power of two (fwht.c:4-28;
    its stack arrays need `ulimit -s unlimited`), the GPU build goes through two passes over the planar row.  Amplitudes are
    kept small enough that |x - mean| * n stays inside int32 (the reference's own precondition for defined behaviour)."""
    def wave(nch, ns, amp, seed, bps):
        rng = np.random.default_rng(seed)
        t = np.arange(ns, dtype=np.float64)[:, None]
        x = amp * np.sin(t / (97.0 + 13.0 * np.arange(nch)[None, :])) + rng.integers(-8, 8, size=(ns, nch))
        x = np.round(x).astype(np.int32) + (np.arange(nch, dtype=np.int32)[None, :] * 37 - 50)  # a mean per channel, some negative
        if bps == 4:
            return np.ascontiguousarray(x).view(np.uint8).reshape(-1)
        if bps == 2:
            return np.ascontiguousarray(x.astype(np.int16)).view(np.uint8).reshape(-1)
        b = np.ascontiguousarray(x).view(np.uint8).reshape(ns * nch, 4)[:, :bps]
        return np.ascontiguousarray(b).reshape(-1)

    return [dict(name="wave2x131072_hadamard", kind="hadamard", bps=4, nch=2, ns=131072, nb=3, data=wave(2, 131072, 3000.0, 71, 4)),
            dict(name="wave3x262144_i24_hadamard", kind="hadamard", bps=3, nch=3, ns=262144, nb=3, data=wave(3, 262144, 900.0, 72, 3)),
            dict(name="wave1x4194304_i16_hadamard", kind="hadamard", bps=2, nch=1, ns=4194304, nb=3, data=wave(1, 4194304, 150.0, 73, 2))]


def dct_dense_big_cases():
    """dct at ns > 8192 that is NOT a power of two: the reference's own n x n table on the GPU as well (bit-exact), up to the
    reach of the reference's int table index (ns <= 32768).  The reference's full stream is the fixture."""
    return [dict(name="synth2x10000_dct", kind="dct", bps=4, nch=2, ns=10000, nb=2,
                 data=np.ascontiguousarray(synth.synth_native(2, 10000, block_index=12, ecg=True).numpy().reshape(-1)))]


# the reference's band-pass (0.4-200 Hz Butterworth @ 2000 Sps, lib_rspt_test/rspt_test.cpp:123-125) and two shorter filters
# of lib_rspt/filter.h:104-112, as (n = feedback, d = feed-forward)
IIR_BANDPASS = ([1.00000000000, -3.14332095199, 3.70064088865, -1.97083923944, 0.41351972908],
                [0.06722876941, 0.00000000000, -0.13445753881, 0.00000000000, 0.06722876941])
IIR_LOWPASS = ([1.00000000000, -1.56101807580, 0.64135153806], [0.02008336556, 0.04016673113, 0.02008336556])
IIR_HIGHPASS = ([1.00000000000, -1.99955571171, 0.99955581039], [0.99977788053, -1.99955576105, 0.99977788053])


def iir_cases():
    """name, bps, nch, ns, (n, d), init_nr_samples, data: the pre-filter step of rspt_test.cpp:116-136 on the repo's two
    recordings (the harness's own use) and on synthetic blocks of every sample width / filter order."""
    ecg = np.frombuffer(synth.ecg_12ch_i32(), dtype=np.uint8)
    ds = np.frombuffer(synth.data_stream_3ch_i24(), dtype=np.uint8)
    C = []

    def add(name, bps, nch, ns, coef, init, data):
        data = np.ascontiguousarray(data[: bps * nch * ns])
        C.append(dict(name=name, bps=bps, nch=nch, ns=ns, n=coef[0], d=coef[1], init=init, data=data))

    add("ecg12x34199_bandpass", 4, 12, 34199, IIR_BANDPASS, 2000, ecg)
    add("ds3x20000_i24_bandpass", 3, 3, 20000, IIR_BANDPASS, 2000, ds)
    add("synth5x3000_i16_lowpass", 2, 5, 3000, IIR_LOWPASS, 500, synth.synth_native(5, 3000, 3, bps=2, ecg=True).numpy())
    add("synth4x2500_i32_highpass", 4, 4, 2500, IIR_HIGHPASS, 100, synth.synth_native(4, 2500, 4, ecg=True).numpy())
    add("rand7x1001_i8_order1", 1, 7, 1001, ([1.0, -0.9], [0.05, 0.05]), 50, _rand_native(7, 1001, 1, 51, 100))
    add("rand3x777_i24_order3", 3, 3, 777, ([1.0, -1.2, 0.5, -0.05], [0.1, 0.2, 0.2, 0.1]), 0, _rand_native(3, 777, 3, 52, 1 << 20))
    return C


def iir_big_case():
    """The pre-filter at the size at which round 2's kernel was wrong: one full 64 ch x 65536 int32 block (ECG-like synthetic,
    SURVEY 8d), the harness's band-pass and history length (rspt_test.cpp:123-129).  Fused multiply-adds moved an output by one
    count about once in two million samples -- none of the small fixtures above ever showed it."""
    return dict(name="synth64x65536_i32_bandpass", bps=4, nch=64, ns=65536, n=IIR_BANDPASS[0], d=IIR_BANDPASS[1], init=2000, block=5)


def iir_big_data(c):
    return synth.synth_native(c["nch"], c["ns"], c["block"], bps=c["bps"], ecg=True).numpy()
