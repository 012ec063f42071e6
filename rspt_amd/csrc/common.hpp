// common.hpp -- shared definitions for the gfx950 signal_packer kernels.
//
// Data model (names follow the reference's domain):
//   block        one compress() call: nch x ns samples of bps bytes, interleaved
//                sample-major (lib_signalpacker/utils.cpp:123-191)
//   plane k      byte k of every transformed int32, flat channel-major order,
//                N = nch*ns bytes (signal_packer_base.cpp:40-68)
//   hzr block    <= 65536 consecutive bytes of one plane (hzr_encode.c:528-539):
//                the unit one workgroup encodes
//   granule      16 consecutive bytes of an hzr block: the unit one lane owns
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Diagnostics (timing probes that skip work, tuning knobs read from the environment, in-kernel time stamps) exist only in
// builds with -DRSPT_DIAG (`python -m rspt_amd.build --diag` -> rspt_amd/librspt_hip_diag.so, loaded only via RSPT_HIP_LIB).  In the
// product library the probe word is the constant 0: every probe branch folds away and no environment variable changes a result.
#ifdef RSPT_DIAG
#define RSPT_DIAG_ONLY(x) (x)
#else
#define RSPT_DIAG_ONLY(x) 0u
#endif

namespace rspt {

constexpr uint32_t kHzrBlock = 65536;  // hzr_internal.h:109
constexpr int kNumSym = 261;           // hzr_internal.h:114
constexpr uint32_t kRunCap = 16662;    // hzr_internal.h:121
constexpr int kSymStride = 264;        // per-block stride of hist[] / cw[] rows (u32)
constexpr int kTdescWords = 92;        // >= ceil((261*10+260)/32)=90, padded
constexpr int kWave = 64;
constexpr int kEncThreads = 1024;      // one hzr block per 1024-thread workgroup
constexpr int kEncWaves = kEncThreads / kWave;
constexpr int kMaxPlanes = 4;

enum : uint32_t { kModeCopy = 0, kModeHuff = 1, kModeFill = 2, kModeSkip = 3 };

// Shape of one packer, passed to kernels by value.
struct Geom {
    uint32_t bps, nch, ns;
    uint32_t N;             // nch*ns  (< 2^31, as in the reference's int indices)
    uint32_t nblk;          // hzr blocks per plane = ceil(N/65536)
    uint32_t hdr_len;       // 0, or 3*nch for dct/hadamard (means header)
    uint32_t method;        // stream method byte (signal_packer_base.cpp:83)
    uint32_t kind;          // rspt_hip_kind
    uint32_t be;            // samples arrive with their bytes reversed (rspt_hip_set_byte_order): the front ends swap on the way in
    uint64_t plane_stride;  // bytes between planes in the workspace (N rounded up to 256)
    uint64_t block_bytes;   // bps*nch*ns
};

// CRC-32C constants for the parallel checksum (tools/kernel_model.py:crc_parallel).
struct CrcConsts {
    uint32_t table[4][256];  // slice-by-4 LUTs, reflected poly 0x82F63B78; table[0] = hzr_crc32c.c:32
    // multiplication by a fixed power of x as four byte-indexed lookups (gf_mul is linear in its first operand):
    //   shift[i][j][b] = gf_mul(b << 8j, K_i),  K_i = x^(8*64*(63-i)) for i < 64 (lane chunk -> end of the wave's 4 KiB slot),
    //   K_{64+w} = x^(8*4096*(15-w)) (wave slot -> end of the 64 KiB window), K_80 = x^(8*65536) (the one chunk in front of it)
    uint32_t shift[81][4][256];
    // shift4[l] = multiplication by x^(8*4*(l+1)): a lane's word-strided partial CRC to the end of its 64-word group
    // (the big encoder's CRC: lane tid owns the virtual words tid, tid+1024, ... counted from the END of the image)
    uint32_t shift4[64][4][256];
    uint32_t prefix;  // 4 bytes X (LE) with raw_crc(X) = 0xFFFFFFFF
    uint32_t pad[3];
};

__device__ __forceinline__ uint32_t gf_shift(const CrcConsts* cc, uint32_t i, uint32_t a) {
    const uint32_t(*t)[256] = cc->shift[i];
    return t[0][a & 0xFFu] ^ t[1][(a >> 8) & 0xFFu] ^ t[2][(a >> 16) & 0xFFu] ^ t[3][a >> 24];
}

__device__ __forceinline__ uint32_t gf_shift4(const CrcConsts* cc, uint32_t i, uint32_t a) {
    const uint32_t(*t)[256] = cc->shift4[i];
    return t[0][a & 0xFFu] ^ t[1][(a >> 8) & 0xFFu] ^ t[2][(a >> 16) & 0xFFu] ^ t[3][a >> 24];
}

// Per-hzr-block record written by k_tree, read by k_layout and k_encode.
struct BlockMeta {
    uint32_t mode;         // kMode*
    uint32_t payload_len;  // bytes after the 7-byte block header
    uint32_t tree_bits;    // length of the tree description (mode 1)
    uint32_t fill;         // fill byte (mode 2)
};

__device__ __forceinline__ uint32_t hb_index(const Geom& g, uint32_t b, uint32_t k, uint32_t j) {
    return (b * kMaxPlanes + k) * g.nblk + j;
}

// ---------------------------------------------------------------------------
// wave-level helpers (wave = 64 lanes)
// ---------------------------------------------------------------------------
// C's `(int)x` of a double AS THE REFERENCE'S BUILD COMPUTES IT: out of range is undefined in C, and the x86-64 conversion the
// reference (and the oracle) compile to, cvttsd2si, returns 0x80000000 for everything it cannot represent -- too large, too small
// or NaN.  The GPU's own conversion saturates (+overflow -> 0x7FFFFFFF, NaN -> 0): one count off in exactly those samples (found by
// tools/soak.py on dct blocks whose coefficients overflow their two planes; signal_packer_dct.cpp:85,98, rspt_test.cpp:130).
// (The test is on the high word: x >= 2^31, +inf or a NaN without sign bit <=> hi >= 0x41E00000 as a signed integer -- two full-rate
// integer instructions beside the conversion, which matters in the IIR's recurrence wave.  Below -2^31 both conversions give
// 0x80000000; a NaN with the sign bit set cannot arise from finite samples.)
__device__ __forceinline__ int32_t trunc_i32_c(double x) { return __double2hiint(x) >= 0x41E00000 ? (int32_t)0x80000000u : (int32_t)x; }

__device__ __forceinline__ uint32_t lane_id() { return __lane_id(); }

// Cross-lane data movement goes through DPP (a VALU operand modifier, a few cycles) rather than
// __shfl* (ds_bpermute_b32: an LDS-crossbar round trip per step).  Controls (GFX9 encoding):
//   row_shr:n 0x110|n   row_shl:n 0x100|n   row_bcast:15 0x142   row_bcast:31 0x143
// A row is 16 lanes, a bank 4 lanes.  Lanes whose source is outside the row, or that the masks
// disable, keep `identity`.
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF>
__device__ __forceinline__ uint32_t dpp(uint32_t identity, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, CTRL, ROW_MASK, BANK_MASK, false);
}

// inclusive scan over the 64 lanes; op(far, near): `far` aggregates lower lanes.  op must be associative
// with identity `id`.
template <class Op>
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v, uint32_t id, Op op) {
    uint32_t r = op(dpp<0x111>(id, v), v);
    r = op(dpp<0x112>(id, v), r);
    r = op(dpp<0x113>(id, v), r);
    r = op(dpp<0x114, 0xF, 0xE>(id, r), r);  // lanes 4..15 of each row: + [i-7, i-4]
    r = op(dpp<0x118, 0xF, 0xC>(id, r), r);  // lanes 8..15: + [i-15, i-8]
    r = op(dpp<0x142, 0xA, 0xF>(id, r), r);  // rows 1, 3: + total of the row before
    r = op(dpp<0x143, 0xC, 0xF>(id, r), r);  // rows 2, 3: + total of rows 0..1
    return r;
}

// scans over one row of 16 lanes (rows are independent): prefix (lower lanes = far) and suffix (higher lanes = far)
template <class Op>
__device__ __forceinline__ uint32_t row_scan_prefix(uint32_t v, uint32_t id, Op op) {
    uint32_t r = op(dpp<0x111>(id, v), v);
    r = op(dpp<0x112>(id, v), r);
    r = op(dpp<0x113>(id, v), r);
    r = op(dpp<0x114, 0xF, 0xE>(id, r), r);
    r = op(dpp<0x118, 0xF, 0xC>(id, r), r);
    return r;
}
template <class Op>
__device__ __forceinline__ uint32_t row_scan_suffix(uint32_t v, uint32_t id, Op op) {
    uint32_t r = op(dpp<0x101>(id, v), v);
    r = op(dpp<0x102>(id, v), r);
    r = op(dpp<0x103>(id, v), r);
    r = op(dpp<0x104, 0xF, 0x7>(id, r), r);  // lanes 0..11: + [i+4, i+7]
    r = op(dpp<0x108, 0xF, 0x3>(id, r), r);  // lanes 0..7:  + [i+8, i+15]
    return r;
}

__device__ __forceinline__ uint32_t read_lane(uint32_t v, uint32_t lane /* wave-uniform */) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane);
}

// reductions: the result is wave-uniform (lives in an SGPR)
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    return read_lane(wave_scan_incl(v, 0xFFFFFFFFu, [](uint32_t a, uint32_t b) { return a < b ? a : b; }), 63);
}
__device__ __forceinline__ uint32_t wave_xor_u32(uint32_t v) {
    return read_lane(wave_scan_incl(v, 0u, [](uint32_t a, uint32_t b) { return a ^ b; }), 63);
}
__device__ __forceinline__ uint32_t wave_or_u32(uint32_t v) {
    return read_lane(wave_scan_incl(v, 0u, [](uint32_t a, uint32_t b) { return a | b; }), 63);
}
__device__ __forceinline__ uint32_t wave_add_u32(uint32_t v) {
    return read_lane(wave_scan_incl(v, 0u, [](uint32_t a, uint32_t b) { return a + b; }), 63);
}

// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t v) {
    return wave_scan_incl(v, 0u, [](uint32_t a, uint32_t b) { return a + b; });
}

// ---------------------------------------------------------------------------
// zero-run chaining: a scan element is (all<<31 | count).  "all" = everything
// covered so far is zero; count = zeros adjacent to the open side.
// ---------------------------------------------------------------------------
constexpr uint32_t kZAll = 0x80000000u;
constexpr uint32_t kZIdentity = kZAll;  // (all=1, count=0)

// `far` is the side further from the open end, `near` the side next to it.
__device__ __forceinline__ uint32_t zcomb(uint32_t far, uint32_t near) {
    return (near & kZAll) ? ((far & kZAll) | ((far & ~kZAll) + (near & ~kZAll))) : near;
}

// inclusive forward scan: result at lane l covers lanes 0..l, open side = after lane l
__device__ __forceinline__ uint32_t wave_zscan_fwd(uint32_t v) {
    return wave_scan_incl(v, kZIdentity, [](uint32_t far, uint32_t near) { return zcomb(far, near); });
}

// ---------------------------------------------------------------------------
// granule analysis
// ---------------------------------------------------------------------------
// 0x80 in every byte of w that is zero (exact, no borrow artefacts)
__device__ __forceinline__ uint32_t zero_byte_flags(uint32_t w) {
    uint32_t t = (w & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
    return ~(t | w | 0x7F7F7F7Fu);
}

// 4-bit mask: bit i set iff byte i of w is zero
__device__ __forceinline__ uint32_t zero_nibble(uint32_t w) {
    return (((zero_byte_flags(w) >> 7) * 0x01020408u) >> 24) & 0xFu;
}

struct Granule {
    uint32_t w[4];  // 16 bytes, little-endian
    uint32_t nv;    // valid bytes (0..16)
    uint32_t zm;    // bit i set iff byte i is valid and zero
};

__device__ __forceinline__ void granule_finish(Granule& g) {
    uint32_t m = zero_nibble(g.w[0]) | (zero_nibble(g.w[1]) << 4) | (zero_nibble(g.w[2]) << 8) | (zero_nibble(g.w[3]) << 12);
    g.zm = m & ((1u << g.nv) - 1u);
}

__device__ __forceinline__ uint32_t granule_byte(const Granule& g, uint32_t i) { return (g.w[i >> 2] >> ((i & 3) * 8)) & 0xFFu; }

// leading zeros (capped at nv), as a chain element for the BACKWARD scan
__device__ __forceinline__ uint32_t granule_lead_elem(const Granule& g) {
    uint32_t lead = (uint32_t)__builtin_ctz(~g.zm | (1u << g.nv));
    return (g.nv == 16 && g.zm == 0xFFFFu) ? (kZAll | 16u) : lead;
}

// trailing zeros, as a chain element for the FORWARD scan
__device__ __forceinline__ uint32_t granule_trail_elem(const Granule& g) {
    if (g.nv != 16) return 0;  // a partial granule ends the block
    if (g.zm == 0xFFFFu) return kZAll | 16u;
    uint32_t nz = (~g.zm) & 0xFFFFu;
    return (uint32_t)__builtin_clz(nz) - 16u;
}

// zero-run length -> symbol / extra bits (hzr_internal.h:117-121, hzr_encode.c:422-447)
__device__ __forceinline__ uint32_t run_symbol(uint32_t z) {
    return z == 1 ? 0u : z == 2 ? 256u : z <= 6 ? 257u : z <= 22 ? 258u : z <= 278 ? 259u : 260u;
}
__device__ __forceinline__ uint32_t run_extra_bits(uint32_t sym) {
    return sym == 257 ? 2u : sym == 258 ? 4u : sym == 259 ? 8u : sym == 260 ? 14u : 0u;
}
__device__ __forceinline__ uint32_t run_extra_value(uint32_t sym, uint32_t z) {
    return sym == 257 ? z - 3 : sym == 258 ? z - 7 : sym == 259 ? z - 23 : sym == 260 ? z - 279 : 0u;
}

// Token positions of a granule as bit masks (tools/kernel_model.py:granule_masks):
//   lits    valid non-zero bytes: one literal token each
//   starts  zero bytes at which a zero-run token starts: a zero whose distance
//           from its run start is a multiple of 16662 (hzr_encode.c:149,417)
//   single  tokens without extra bits that are looked up directly: literals (index = byte
//           value), zero runs of length 1 (symbol 0 = byte value 0) and of length 2 (symbol 256)
//   two     the subset of `single` that are runs of exactly 2 zeros (lookup index 256)
//   runs    the remaining run tokens (length >= 3), rare in dense planes
struct GranuleMasks {
    uint32_t lits, starts, single, two, runs;
};

__device__ __forceinline__ GranuleMasks granule_masks(uint32_t zm, uint32_t nv, uint32_t zb, uint32_t za) {
    GranuleMasks m;
    const uint32_t valid = (1u << nv) - 1u;
    m.lits = ~zm & valid;
    m.starts = zm & ~(zm << 1) & ~1u;  // interior runs: the byte before is non-zero
    const uint32_t lead = (uint32_t)__builtin_ctz(~zm | (1u << nv));
    if (lead) {  // the leading zeros continue a run of zb zeros (zb = 0: a run starts at byte 0)
        const uint32_t q = (zb >= kRunCap) + (zb >= 2 * kRunCap) + (zb >= 3 * kRunCap);
        const uint32_t r = zb - q * kRunCap;
        const uint32_t icap = r ? kRunCap - r : 0u;
        if (icap < lead) m.starts |= 1u << icap;
    }
    // zero map extended by the first two bytes behind the granule (bits nv, nv+1 <- za >= 1, za >= 2)
    const uint32_t ze = zm | ((za >= 1 ? 1u : 0u) << nv) | ((za >= 2 ? 2u : 0u) << nv);
    const uint32_t len1 = m.starts & ~(ze >> 1);              // next byte is not zero
    m.two = m.starts & (ze >> 1) & ~(ze >> 2);                // exactly two zeros
    m.single = m.lits | len1 | m.two;
    m.runs = m.starts & ~(len1 | m.two);
    return m;
}

// length of the run token that starts at byte i of the granule
__device__ __forceinline__ uint32_t run_token_length(uint32_t zm, uint32_t nv, uint32_t za, uint32_t i) {
    const uint32_t ahead = (uint32_t)__builtin_ctz(~(zm >> i) | (1u << (nv - i)));
    const uint32_t rem = ahead + ((i + ahead == nv) ? za : 0u);
    return rem < kRunCap ? rem : kRunCap;
}

// ---------------------------------------------------------------------------
// GF(2) arithmetic for CRC-32C in the reflected domain (bit 31 = x^0)
// ---------------------------------------------------------------------------
constexpr uint32_t kCrcPoly = 0x82F63B78u;

__host__ __device__ inline uint32_t gf_mul(uint32_t a, uint32_t b) {
    uint32_t r = 0;
#pragma unroll 4
    for (int i = 0; i < 32; ++i) {
        r ^= b & (0u - ((a >> (31 - i)) & 1u));
        b = (b >> 1) ^ (kCrcPoly & (0u - (b & 1u)));
    }
    return r;
}

}  // namespace rspt
