"""Executable model of the HIP kernels' algorithms (host logic, pure Python).

Each function restates, step for step, what one kernel in
rspt_amd/csrc/hzr_kernels.hip does per lane / per wave, so the parallel
formulations can be checked on the CPU against the oracle before (and
independently of) a GPU run:

  granule tokenizer   16-byte granules + zeros-before / zeros-after scans
  tree links          merge loop -> per-node up-links -> per-leaf walk
  parallel CRC-32C    right-aligned 16-byte granules, GF(2) shift constants

tests/test_kernel_model.py drives this against oracle/.
"""
import numpy as np

CAP = 16662
RUN_BASE = (2, 3, 7, 23, 279)
RUN_EXTRA = (0, 2, 4, 8, 14)
POLY = 0x82F63B78  # reflected CRC-32C


# --------------------------------------------------------------------------
# granule tokenizer (k_hist / k_encode)
# --------------------------------------------------------------------------
def run_symbol(z):
    """zero-run length -> (symbol, extra value, extra bits)"""
    if z == 1:
        return 0, 0, 0
    if z == 2:
        return 256, 0, 0
    c = 1 if z <= 6 else 2 if z <= 22 else 3 if z <= 278 else 4
    return 256 + c, z - RUN_BASE[c], RUN_EXTRA[c]


def granule_scan(block):
    """per 16-byte granule: (nv, zmask, zb, za).  zb = zeros immediately before
    the granule, za = zeros immediately after it (both cut at the block ends)."""
    n = len(block)
    G = (n + 15) // 16
    nv = [min(16, n - 16 * g) for g in range(G)]
    zm, lead, trail, allz = [], [], [], []
    for g in range(G):
        b = block[16 * g : 16 * g + nv[g]]
        m = 0
        for i, x in enumerate(b):
            if x == 0:
                m |= 1 << i
        zm.append(m)
        full = (1 << nv[g]) - 1
        a = m == full and nv[g] == 16  # a partial granule ends the block: never "all zero" for chaining
        ld = 0
        while ld < nv[g] and (m >> ld) & 1:
            ld += 1
        tr = 0
        while tr < nv[g] and (m >> (nv[g] - 1 - tr)) & 1:
            tr += 1
        lead.append(ld)
        trail.append(tr if nv[g] == 16 else 0)
        allz.append(a)
    zb = [0] * G
    for g in range(1, G):
        zb[g] = (16 + zb[g - 1]) if allz[g - 1] else trail[g - 1]
    za = [0] * G
    for g in range(G - 2, -1, -1):
        # a partial last granule contributes its leading zeros (its "lead" counts valid bytes only)
        za[g] = (16 + za[g + 1]) if allz[g + 1] else lead[g + 1]
    return nv, zm, zb, za


def granule_tokens(block):
    """tokens (sym, extra, extrabits) in stream order, produced granule by granule
    exactly as a lane does it: only local bytes + (zb, za)."""
    nv, zm, zb, za = granule_scan(block)
    out = []
    for g in range(len(nv)):
        dist = zb[g]  # distance of byte i from the start of its zero run (valid while in a run)
        for i in range(nv[g]):
            x = block[16 * g + i]
            if x != 0:
                out.append((int(x), 0, 0))
                dist = 0
                continue
            if dist % CAP == 0:  # a token starts here
                ahead = 0
                while i + ahead < nv[g] and (zm[g] >> (i + ahead)) & 1:
                    ahead += 1
                rem = ahead + (za[g] if i + ahead == nv[g] else 0)
                out.append(run_symbol(min(CAP, rem)))
            dist += 1
    return out


def reference_tokens(block):
    """the serial tokenizer (hzr_encode.c:133-173) for comparison"""
    out, i, n = [], 0, len(block)
    while i < n:
        if block[i] != 0:
            out.append((int(block[i]), 0, 0))
            i += 1
            continue
        z = 1
        while z < CAP and i + z < n and block[i + z] == 0:
            z += 1
        out.append(run_symbol(z))
        i += z
    return out


# --------------------------------------------------------------------------
# tree links (k_tree)
# --------------------------------------------------------------------------
def tree_links(hist):
    """returns (codes, lens, desc_bits list of (offset, 10-bit value), tree_bits)"""
    syms = [s for s in range(261) if hist[s]]
    S = len(syms)
    assert S >= 2
    BIG = 0xFFFFFFFF
    key = [BIG] * (2 * S)
    for i, s in enumerate(syms):
        key[i] = (int(hist[s]) << 10) | (1023 - i)
    sbits = [10] * S + [0] * S
    up = [0] * (2 * S)  # parent | isB<<10 | add<<11
    for it in range(S - 1):
        m1 = min(key)
        key[key.index(m1)] = BIG
        m2 = min(key)
        key[key.index(m2)] = BIG
        i1, i2 = 1023 - (m1 & 1023), 1023 - (m2 & 1023)
        n = S + it
        key[n] = (((m1 >> 10) + (m2 >> 10)) << 10) | (1023 - n)
        sbits[n] = 1 + sbits[i1] + sbits[i2]
        up[i1] = n | (0 << 10) | (1 << 11)
        up[i2] = n | (1 << 10) | ((1 + sbits[i1]) << 11)
    root = 2 * S - 2
    codes, lens, desc = {}, {}, []
    for i, s in enumerate(syms):
        cur, code, ln, off = i, 0, 0, 0
        while cur != root:
            u = up[cur]
            code = (code << 1) | ((u >> 10) & 1)
            off += u >> 11
            ln += 1
            cur = u & 1023
        codes[s], lens[s] = code, ln
        desc.append((off, 1 | (s << 1)))
    return codes, lens, desc, sbits[root]


def encode_block_model(block):
    """whole Huffman block payload from the parallel formulations (mode 1 only)"""
    toks = granule_tokens(block)
    hist = np.zeros(261, dtype=np.int64)
    for s, _, _ in toks:
        hist[s] += 1
    codes, lens, desc, tbits = tree_links(hist)
    total = tbits + sum(lens[s] + eb for s, _, eb in toks)
    buf = bytearray((total + 7) // 8)

    def put(pos, val, width):
        for i in range(width):
            if (val >> i) & 1:
                buf[(pos + i) >> 3] |= 1 << ((pos + i) & 7)

    for off, v in desc:
        put(off, v, 10)
    pos = tbits
    for s, ex, eb in toks:
        put(pos, codes[s] | (ex << lens[s]), lens[s] + eb)
        pos += lens[s] + eb
    return bytes(buf)


# --------------------------------------------------------------------------
# parallel CRC-32C (k_encode tail)
# --------------------------------------------------------------------------
def gf_mul(a, b):
    """product of two reflected 32-bit polynomials mod P (bit 31 = x^0)."""
    r = 0
    for i in range(32):
        if (a >> (31 - i)) & 1:  # coefficient of x^i in a
            r ^= b
        b = (b >> 1) ^ (POLY if b & 1 else 0)  # b *= x
    return r


def x_pow_bytes(nbytes):
    """x^(8*nbytes) mod P in reflected form"""
    r = 0x80000000  # 1
    base = 0x00800000  # x^8
    e = nbytes
    while e:
        if e & 1:
            r = gf_mul(r, base)
        base = gf_mul(base, base)
        e >>= 1
    return r


def raw_crc_bytes(bs):
    """CRC register after feeding bs from state 0 (no init / final xor)"""
    c = 0
    for x in bs:
        c ^= x
        for _ in range(8):
            c = (c >> 1) ^ (POLY if c & 1 else 0)
    return c


def init_prefix():
    """4 bytes X with raw_crc(X) == 0xFFFFFFFF: feeding X from state 0 leaves the
    register at the standard init value, so crc32c(M) == ~raw_crc(X || M)."""
    # raw over 4 bytes from state 0 is linear and bijective: solve by basis images
    basis = [raw_crc_bytes(((1 << b).to_bytes(4, "little"))) for b in range(32)]
    # Gaussian elimination over GF(2) for target 0xFFFFFFFF
    rows = [(basis[b], 1 << b) for b in range(32)]
    target, sol = 0xFFFFFFFF, 0
    piv = {}
    for bit in range(32):
        for idx, (v, tag) in enumerate(rows):
            if (v >> bit) & 1 and idx not in piv.values():
                piv[bit] = idx
                for j, (v2, tag2) in enumerate(rows):
                    if j != idx and (v2 >> bit) & 1:
                        rows[j] = (v2 ^ v, tag2 ^ tag)
                break
    for bit in range(32):
        if (target >> bit) & 1:
            v, tag = rows[piv[bit]]
            sol ^= tag
    x = sol.to_bytes(4, "little")
    assert raw_crc_bytes(x) == 0xFFFFFFFF
    return x


def crc_parallel(msg, lanes=64, waves=16):
    """CRC-32C of msg the way k_encode does it: virtual message V = X || msg,
    cut into 16-byte granules counted from the END; lane partial CRCs are
    shifted to the wave-slot end (per-lane constant), xor-reduced per wave,
    Horner-combined over workgroup rows, then shifted per wave."""
    X = init_prefix()
    V = X + bytes(msg)
    Lv = len(V)
    G = (Lv + 15) // 16
    per_row = lanes * waves
    rows = (G + per_row - 1) // per_row
    lane_shift = [x_pow_bytes(16 * (lanes - 1 - l)) for l in range(lanes)]
    wave_shift = [x_pow_bytes(16 * lanes * (waves - 1 - w)) for w in range(waves)]
    row_shift = x_pow_bytes(16 * per_row)
    acc = [0] * waves
    for rowE in range(rows - 1, -1, -1):  # front-most row first
        for w in range(waves):
            red = 0
            for l in range(lanes):
                tid = w * lanes + l
                ge = rowE * per_row + (per_row - 1 - tid)  # granule index from the end
                hi = Lv - 16 * ge  # exclusive end
                lo = hi - 16
                if hi <= 0:
                    continue
                gran = bytes(V[p] if p >= 0 else 0 for p in range(lo, hi))
                red ^= gf_mul(raw_crc_bytes(gran), lane_shift[l])
            acc[w] = gf_mul(acc[w], row_shift) ^ red
    total = 0
    for w in range(waves):
        total ^= gf_mul(acc[w], wave_shift[w])
    return total ^ 0xFFFFFFFF


def crc_strided(msg, nthr=1024):
    """CRC-32C of msg the way the big encoder does it on its unpadded LDS image: V = X || msg as 4-byte virtual words
    counted from the END; thread tid owns the words tid, tid + nthr, ... (consecutive threads read consecutive
    words: no bank conflicts), Horner over its words with x^(8*4*nthr) per step, then x^(8*4*(tid+1)) to the end."""
    X = init_prefix()
    V = X + bytes(msg)
    Lv = len(V)
    nvw = (Lv + 3) // 4

    def vword(r):
        lo = Lv - 4 * (r + 1)
        return int.from_bytes(bytes(V[p] if p >= 0 else 0 for p in range(lo, lo + 4)), "little")

    K = (nvw + nthr - 1) // nthr
    G = x_pow_bytes(4 * nthr)
    total = 0
    for tid in range(min(nthr, nvw)):
        c = 0
        for k in range(K - 1, 0, -1):
            r = tid + nthr * k
            if r < nvw:
                c ^= vword(r)
            c = gf_mul(c, G)
        c ^= vword(tid)
        total ^= gf_mul(c, x_pow_bytes(4 * (tid + 1)))
    return total ^ 0xFFFFFFFF


def crc_constants(lanes=64, waves=16):
    return dict(
        prefix=int.from_bytes(init_prefix(), "little"),
        lane_shift=[x_pow_bytes(16 * (lanes - 1 - l)) for l in range(lanes)],
        wave_shift=[x_pow_bytes(16 * lanes * (waves - 1 - w)) for w in range(waves)],
        row_shift=x_pow_bytes(16 * lanes * waves),
    )


# --------------------------------------------------------------------------
# v2 tokenizer: mask formulation used by k_hist / k_encode (no per-byte state)
# --------------------------------------------------------------------------
def _ctz(x):
    return (x & -x).bit_length() - 1


def granule_masks(zm, nv, zb):
    """(lits, starts): literal positions and positions where a zero-run token starts."""
    valid = (1 << nv) - 1
    zm &= valid
    lits = ~zm & valid
    starts = zm & ~(zm << 1) & ~1
    lead = _ctz((~zm & 0xFFFFFFFF) | (1 << nv))
    if lead > 0:
        q = (zb >= CAP) + (zb >= 2 * CAP) + (zb >= 3 * CAP)
        r = zb - q * CAP
        i_cap = CAP - r if r else 0
        if i_cap < lead:
            starts |= 1 << i_cap
    return lits, starts


def granule_tokens_v2(block):
    nv, zm, zb, za = granule_scan(block)
    out = []
    for g in range(len(nv)):
        lits, starts = granule_masks(zm[g], nv[g], zb[g])
        # v3 refinement: run tokens of length 1 / 2 carry no extra bits and are looked up like
        # literals (symbol 0 = byte value 0; symbol 256 for two zeros)
        ze = zm[g] | ((1 << nv[g]) if za[g] >= 1 else 0) | ((2 << nv[g]) if za[g] >= 2 else 0)
        len1 = starts & ~(ze >> 1)
        two = starts & (ze >> 1) & ~(ze >> 2)
        single, runs = lits | len1 | two, starts & ~(len1 | two)
        assert single & runs == 0
        for i in range(nv[g]):
            if (single >> i) & 1:
                out.append((256 if (two >> i) & 1 else int(block[16 * g + i]), 0, 0))
            elif (runs >> i) & 1:
                ahead = _ctz(((~(zm[g] >> i)) & 0xFFFFFFFF) | (1 << (nv[g] - i)))
                rem = ahead + (za[g] if i + ahead == nv[g] else 0)
                out.append(run_symbol(min(CAP, rem)))
    return out
