"""Pin the restatement against the compiled reference itself (oracle/_ref) on
many seeded random inputs.  Skipped where oracle/_ref is absent."""
import numpy as np
import pytest

import cases


def _rand_block(seed):
    """hzr-oriented byte soup: alphabet size, zero density and run structure vary."""
    r = np.random.RandomState(seed)
    n = int(r.choice([1, 2, 3, 17, 255, 256, 1000, 4096, 65535, 65536, 65537, 70000, 140000]))
    kind = seed % 6
    if kind == 0:
        a = r.randint(0, 256, n)
    elif kind == 1:
        a = r.randint(0, r.randint(2, 9), n)  # tie-heavy small alphabets
    elif kind == 2:
        a = np.minimum(r.geometric(0.3, n), 255)  # deep trees
    elif kind == 3:
        a = np.where(r.rand(n) < r.choice([0.5, 0.9, 0.99, 0.999]), 0, r.randint(1, 256, n))
    elif kind == 4:
        a = np.zeros(n, dtype=np.int64)
        for _ in range(r.randint(1, 12)):
            a[r.randint(0, n)] = r.randint(1, 4)
    else:
        a = np.full(n, r.randint(0, 256))
        if n > 4 and r.rand() < 0.5:
            a[r.randint(0, n)] ^= 1
    return a.astype(np.uint8)


@pytest.mark.parametrize("chunk", range(8))
def test_hzr_encode_matches_reference(orc, ref, chunk):
    for seed in range(chunk * 40, chunk * 40 + 40):
        d = _rand_block(seed)
        so, sr = orc.hzr_encode(d), ref.hzr_encode(d)
        assert so == sr, "seed %d n %d" % (seed, d.size)
        assert orc.hzr_decode(sr, d.size)[0] == d.tobytes()


def test_hzr_payload_boundary(orc, ref):
    """Blocks whose Huffman payload lands at / next to in_size (PlainCopy edge,
    hzr_encode.c:377-382,463-469)."""
    hits = set()
    for n in (16, 24, 32, 48, 64, 96, 128):
        for seed in range(300):
            r = np.random.RandomState(10000 + seed)
            d = r.randint(0, r.randint(2, 40), n).astype(np.uint8)
            so, sr = orc.hzr_encode(d), ref.hzr_encode(d)
            assert so == sr
            hist, mode, plen = orc.hzr_block_stats(d)
            assert mode == sr[10] and (plen == len(sr) - 11)
            if abs(plen - n) <= 1 and mode != 2:
                hits.add((mode, int(plen) - n))
    assert (1, 0) in hits and (0, 0) in hits  # exactly-fits Huffman and copy both seen


@pytest.mark.parametrize("seed", range(60))
def test_packers_match_reference(orc, ref, seed):
    r = np.random.RandomState(seed)
    kind = ["xdelta_hzr", "hzr", "hadamard", "dct"][seed % 4]
    bps = int(r.randint(1, 5))
    nch = int(r.randint(1, 6))
    if kind == "hadamard":
        ns = 1 << int(r.randint(1, 11))
    elif kind == "dct":
        ns = int(r.randint(2, 130))
    else:
        ns = int(r.randint(1, 3000))
    nb = int(r.randint(1, 5))
    amp = int(r.choice([3, 100, 3000, 1 << 15, 1 << 22, 1 << 30]))
    data = cases._rand_native(nch, ns, bps, 500 + seed, amp, walk=bool(r.randint(0, 2)))
    po, pr = orc.packer(kind, bps, nch, ns, nb), ref.packer(kind, bps, nch, ns, nb)
    for rep in range(2):  # second call exercises the persisted nb
        so, sr = po.compress(data), pr.compress(data)
        assert so == sr, (kind, bps, nch, ns, nb, amp)
        do, uo, _ = po.decompress(sr)
        dr, ur, _ = pr.decompress(sr)
        assert do == dr and uo == ur == len(sr)
    po.close()
    pr.close()


def test_needed_nb_matches_reference_escalation(orc, ref):
    """criterion (SURVEY 8 note a-3) == the reference's round-trip escalation."""
    import struct

    for seed in range(150):
        r = np.random.RandomState(7000 + seed)
        bps, nch, ns = int(r.randint(1, 5)), int(r.randint(1, 5)), int(r.randint(1, 200))
        nb0 = int(r.randint(1, 5))
        amp = int(r.choice([2, 60, 130, 3000, 40000, 1 << 23, 1 << 30]))
        data = cases._rand_native(nch, ns, bps, 900 + seed, amp, walk=bool(seed & 1))
        pr = ref.packer("xdelta_hzr", bps, nch, ns, nb0)
        s = pr.compress(data)
        pos, k = 1, 0
        while pos < len(s):
            pos += 4 + struct.unpack_from("<I", s, pos)[0]
            k += 1
        v = orc.xdelta_forward(orc.native_to_i32(data, ns, nch, bps))
        assert orc.xdelta_needed_nb(v, bps, nb0) == k, (bps, nch, ns, nb0, amp)
        pr.close()
