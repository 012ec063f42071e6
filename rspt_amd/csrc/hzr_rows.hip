// hzr_rows.hip -- the workgroup-per-block half of the hzr encoder: tokenizer + histogram (k_hist) and code emission
// (k_encode) for the blocks that are not "small" (hzr_kernels.hip: k_tree / k_encode_small take those).
//
//   k_hist     persistent 1024-thread workgroups, one hzr block at a time (four 16-byte granules per lane): zero-run
//              tokenizer (hzr_encode.c:133-173), per-wave 261-bin histograms in LDS -> the block's histogram and the
//              per-4-KiB-segment histograms (k_tree turns those into the stream bit at which each wave's tokens start)
//   k_encode   the same block again: codes into an LDS image of the payload (:410-457), CRC-32C (hzr_crc32c.c:77-84),
//              block header (:475-481), coalesced copy-out to the offset k_layout computed
//
// Token attribution (any rule that keeps stream order is the reference's greedy walk, hzr_encode.c:410-457): the tokens
// of a zero run belong to the run's LAST byte.  A run of length R is floor(R / 16662) capped tokens and one token for the
// remainder; only backward context (zeros before a granule) and ONE byte of lookahead (is the next byte zero?) are needed,
// so the chain over the workgroup runs in one direction and nothing about it has to travel from k_hist to k_encode.
//
// Two row shapes (a row = 1 KiB, one granule per lane), chosen per row by the wave:
//   dense   the zero flags of byte i of all 64 lanes live in one 64-bit scalar mask per byte position, so run ends, runs of
//           one and two zeros and dead zero bytes are scalar logic; every byte position is one table lookup per lane.
//           Runs of three and more zeros (rare here) are handled per lane afterwards.
//   sparse  per-lane masks; a lane lists its entries (literal with the run in front of it, or a run that ends the
//           granule) into a per-wave queue, and the wave then takes the queue 64 entries at a time.
#include "common.hpp"

namespace rspt {

constexpr uint32_t kQueueEntries = 512;  // entries per wave in the sparse-row queue (8 x 64)
static_assert(kTokQueueBase + kEncWaves * kQueueEntries <= kStageWords, "emit-phase queues sit in the image tail");
// k_encode's own token count of a block (own_bits): sixteen per-wave histograms at the start of the image, the queues behind them
constexpr uint32_t kOwnHistQueueBase = kEncWaves * kSymStride;
static_assert(kOwnHistQueueBase >= kTokQueueBase && kOwnHistQueueBase + kEncWaves * kQueueEntries <= kStageWords, "own-count queues: behind the histograms, inside the image");

// trailing (highest-address) zero bytes of a granule that is not all zero
__device__ __forceinline__ uint32_t trail_zero_bytes(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
    return w3 ? ((uint32_t)__clz((int)w3) >> 3) : w2 ? 4u + ((uint32_t)__clz((int)w2) >> 3) : w1 ? 8u + ((uint32_t)__clz((int)w1) >> 3)
                                                                                             : 12u + ((uint32_t)__clz((int)w0) >> 3);
}
__device__ __forceinline__ uint32_t zero_mask16(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
    return zero_nibble(w0) | (zero_nibble(w1) << 4) | (zero_nibble(w2) << 8) | (zero_nibble(w3) << 12);
}
__device__ __forceinline__ bool lane_bit(unsigned long long m) { return __builtin_amdgcn_inverse_ballot_w64(m); }

// zeros immediately before every lane's granule: the nearest lower lane whose granule is not all zero closes the run
__device__ __forceinline__ uint32_t zeros_before(bool allz, uint32_t trail, uint32_t zb0) {
    const uint32_t l = lane_id();
    const unsigned long long nonall = ~__ballot(allz);
    const unsigned long long below = nonall & ((1ull << l) - 1ull);
    const uint32_t p = below ? 63u - (uint32_t)__builtin_clzll(below) : 0u;
    const uint32_t tp = (uint32_t)__shfl((int)trail, (int)p, 64);
    return below ? (16u * (l - 1u - p) + tp) : (16u * l + zb0);
}

// length of the zero run whose last byte is byte i of the granule (lits = its non-zero bytes, zb = zeros before it)
__device__ __forceinline__ uint32_t run_ending_at(uint32_t i, uint32_t lits, uint32_t zb) {
    const uint32_t m = lits & ((1u << i) - 1u);
    return m ? i - (31u - (uint32_t)__clz((int)m)) : i + 1u + zb;
}

// full kRunCap tokens in a run of R <= 65536 zeros.  Almost never any: the three compares sit behind one wave-wide test.
__device__ __forceinline__ uint32_t run_caps(uint32_t R) {
    uint32_t q = 0;
    if (__builtin_amdgcn_ballot_w64(R >= kRunCap)) q = (R >= kRunCap) + (R >= 2 * kRunCap) + (R >= 3 * kRunCap);
    return q;
}

__device__ __forceinline__ void hist_run(uint32_t* h, const uint32_t* runcls, uint32_t R) {
    const uint32_t q = run_caps(R);
    const uint32_t rem = R - q * kRunCap;
    if (q) atomicAdd(&h[260], q);
    if (rem) atomicAdd(&h[runcls[min(rem, kRunClsEntries - 1u)] & 0xFFFu], 1u);
}

// stream bits of the tokens of a zero run
__device__ __forceinline__ uint32_t run_bits_tab(const uint2* tab, const uint32_t* runcls, uint32_t R) {
    const uint32_t q = run_caps(R);
    const uint32_t rem = R - q * kRunCap;
    uint32_t bits = q ? q * (tab[260].y + 14u) : 0u;
    if (rem) {
        const uint32_t e = runcls[min(rem, kRunClsEntries - 1u)];
        bits += tab[e & 0xFFFu].y + ((e >> 12) & 0xFu);
    }
    return bits;
}

// the remainder token of a run (0 < rem < 16662) as a bit string
__device__ __forceinline__ void run_token(const uint2* tab, const uint32_t* runcls, uint32_t rem, uint64_t& v, uint32_t& len) {
    const uint32_t e = runcls[min(rem, kRunClsEntries - 1u)];
    const uint2 cw = tab[e & 0xFFFu];
    v = (uint64_t)cw.x | ((uint64_t)(rem - (e >> 16)) << cw.y);
    len = cw.y + ((e >> 12) & 0xFu);
}

// OR the tokens of a zero run into the image at bit `pos`; returns the bits written
__device__ __forceinline__ uint32_t emit_run(uint32_t* stage, const uint2* tab, const uint32_t* runcls, uint32_t pos, uint32_t R) {
    const uint32_t q = run_caps(R);
    const uint32_t rem = R - q * kRunCap;
    uint32_t done = 0;
    if (q) {
        const uint2 c = tab[260];
        const uint64_t v = (uint64_t)c.x | ((uint64_t)(kRunCap - 279u) << c.y);
        for (uint32_t i = 0; i < q; ++i) {
            or_token(stage, pos + done, (uint32_t)v, (uint32_t)(v >> 32), c.y + 14u);
            done += c.y + 14u;
        }
    }
    if (rem) {
        uint64_t v;
        uint32_t len;
        run_token(tab, runcls, rem, v, len);
        or_token(stage, pos + done, (uint32_t)v, (uint32_t)(v >> 32), len);
        done += len;
    }
    return done;
}

// what a row needs to know about its surroundings (wave-uniform)
struct RowCtx {
    uint32_t zb0;    // zeros immediately before the row's first byte (cut at the block start)
    uint32_t last;   // 1: the block ends with this row's last valid byte
    uint32_t valid;  // bytes of this row inside the block (0..1024)
    uint32_t rb;     // position of the row's first byte in the block
};

// Attribution inside a block, rule B: the tokens of a zero run belong to the byte BEHIND the run (the literal that ends
// it); a run that reaches the block end belongs to the block's last byte.  Inside a dense row the run's own last byte
// carries them (the same thing, seen from the scalar masks); what crosses a row boundary is the count `zb0` alone.

// ===========================================================================
// dense rows: scalar zero masks per byte position
// ===========================================================================
// Scalar and vector pipes issue one wave instruction per cycle and CU each, so the mask logic is kept to six scalar
// operations per byte position (next to ~5 vector ones): with ZZ(i) = z(i) & z(i+1) ("a zero followed by a zero")
//   A(i)    = ZZ(i-1) & ~z(i+1)   a run of two or more zeros ends here
//   Len2(i) = A(i) & ~z(i-2)      ... of exactly two: lookup index 256
//   Long(i) = A(i) &  z(i-2)      ... of three or more: handled per lane afterwards
//   Dead(i) = ZZ(i) | Long(i)     zero bytes that end no token (a zero that ends a run of one is a lookup of index 0)
// z(16) of lane 63 is "zero" unless the block ends there: a run that ends with the row is the next row's business.
struct DenseMasks {
    unsigned long long Z[16];
    unsigned long long z15s, z14s, z0n;  // z(-1), z(-2) and z(16) of every lane: the neighbours' bytes, the row's surroundings at the ends
    __device__ __forceinline__ unsigned long long p1(int i) const { return i >= 1 ? Z[i - 1] : z15s; }
    __device__ __forceinline__ unsigned long long p2(int i) const { return i >= 2 ? Z[i - 2] : i == 1 ? z15s : z14s; }
    __device__ __forceinline__ unsigned long long nx(int i) const { return i <= 14 ? Z[i + 1] : z0n; }
};
// "byte i of every lane is zero" as sixteen scalar masks.  The byte select rides on the compare's operand (SDWA): one vector
// instruction per mask where `v_and` + `v_cmp` took two.  (The trailing s_nop covers the wait states this target wants
// between a vector write of a scalar register and a vector read of it as a mask; the compiler does not look inside.)
__device__ __forceinline__ void dense_masks(const uint32_t (&w)[4], const RowCtx& rc, DenseMasks& M) {
    const uint32_t zero = 0;
#define RSPT_ZB(o, wi, b) "v_cmp_eq_u32_sdwa %" #o ", %" #wi ", %20 src0_sel:BYTE_" #b " src1_sel:DWORD\n\t"
    asm volatile(RSPT_ZB(0, 16, 0) RSPT_ZB(1, 16, 1) RSPT_ZB(2, 16, 2) RSPT_ZB(3, 16, 3) RSPT_ZB(4, 17, 0) RSPT_ZB(5, 17, 1) RSPT_ZB(6, 17, 2)
                     RSPT_ZB(7, 17, 3) RSPT_ZB(8, 18, 0) RSPT_ZB(9, 18, 1) RSPT_ZB(10, 18, 2) RSPT_ZB(11, 18, 3) RSPT_ZB(12, 19, 0)
                         RSPT_ZB(13, 19, 1) RSPT_ZB(14, 19, 2) RSPT_ZB(15, 19, 3) "s_nop 1"
                 : "=s"(M.Z[0]), "=s"(M.Z[1]), "=s"(M.Z[2]), "=s"(M.Z[3]), "=s"(M.Z[4]), "=s"(M.Z[5]), "=s"(M.Z[6]), "=s"(M.Z[7]), "=s"(M.Z[8]),
                   "=s"(M.Z[9]), "=s"(M.Z[10]), "=s"(M.Z[11]), "=s"(M.Z[12]), "=s"(M.Z[13]), "=s"(M.Z[14]), "=s"(M.Z[15])
                 : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(zero));
#undef RSPT_ZB
    M.z15s = (M.Z[15] << 1) | (rc.zb0 >= 1 ? 1ull : 0ull);
    M.z14s = (M.Z[14] << 1) | (rc.zb0 >= 2 ? 1ull : 0ull);
    M.z0n = (M.Z[0] >> 1) | (rc.last ? 0ull : (1ull << 63));
}
// the run in front of the row, if the row's first byte ends it (lane 0 emits it in front of its first token)
__device__ __forceinline__ uint32_t lead_run(const RowCtx& rc, const DenseMasks& M) { return (M.Z[0] & 1ull) ? 0u : rc.zb0; }

// per-lane view of the same row, for the lanes that hold a run of three or more zeros (or a slow quad)
struct LaneView {
    uint32_t zm, lits, zb;
    uint32_t len1, len2, lng;  // run ends by class (bit i = byte i)
};
__device__ __forceinline__ LaneView lane_view(const uint32_t (&w)[4], const RowCtx& rc, const DenseMasks& M) {
    LaneView v;
    v.zm = zero_mask16(w[0], w[1], w[2], w[3]);
    v.lits = ~v.zm & 0xFFFFu;
    const bool allz = v.zm == 0xFFFFu;
    const uint32_t trail = allz ? 0u : trail_zero_bytes(w[0], w[1], w[2], w[3]);
    v.zb = zeros_before(allz, trail, rc.zb0);
    const uint32_t b1 = lane_bit(M.p1(0)) ? 1u : 0u, b2 = lane_bit(M.p2(0)) ? 1u : 0u, bn = lane_bit(M.nx(15)) ? 1u : 0u;
    const uint32_t end = v.zm & ~((v.zm >> 1) | (bn << 15));
    const uint32_t zme = (v.zm << 2) | (b1 << 1) | b2;  // bit i+2 = z(i), bit 1 = z(-1), bit 0 = z(-2)
    const uint32_t pz1 = zme >> 1, pz2 = zme;           // bit i = z(i-1), z(i-2)
    v.len1 = end & ~pz1;
    v.len2 = end & pz1 & ~pz2;
    v.lng = end & pz1 & pz2;
    return v;
}

__device__ __forceinline__ void hist_row_dense(const uint32_t (&w)[4], const RowCtx& rc, uint32_t* h, const uint32_t* runcls) {
    DenseMasks M;
    dense_masks(w, rc, M);
    unsigned long long longany = 0;
    unsigned long long zzprev = M.z15s & M.Z[0];  // ZZ(-1)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const unsigned long long Nx = M.nx(i);
        const unsigned long long ZZ = M.Z[i] & Nx;
        const unsigned long long A = zzprev & ~Nx;
        const unsigned long long P2 = M.p2(i);
        const unsigned long long Len2 = A & ~P2, Long = A & P2;
        const unsigned long long Dead = ZZ | Long;
        uint32_t idx = (w[i >> 2] >> ((i & 3) * 8)) & 0xFFu;
        idx = lane_bit(Dead) ? 261u : idx;  // (bins 261..263 are never read)
        idx = lane_bit(Len2) ? 256u : idx;
        atomicAdd(&h[idx], 1u);
        longany |= Long;
        zzprev = ZZ;
    }
    const uint32_t lead = lead_run(rc, M);
    if (lead && lane_id() == 0) hist_run(h, runcls, lead);
    if (longany) {
        const LaneView v = lane_view(w, rc, M);
        uint32_t t = v.lng;
        while (t) {
            const uint32_t i = (uint32_t)__builtin_ctz(t);
            t &= t - 1;
            hist_run(h, runcls, run_ending_at(i, v.lits, v.zb));
        }
    }
}

// byte B of w, times 8: the byte select is part of the shift's operand (SDWA)
__device__ __forceinline__ uint32_t byte_times8(uint32_t w, int B /* constant after unrolling */) {
    uint32_t r;
    const uint32_t three = 3;
    if (B == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(three), "v"(w));
    else if (B == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(three), "v"(w));
    else if (B == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(three), "v"(w));
    else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(three), "v"(w));
    return r;
}

// One dense row from lookup to image.  `base` = stream bit at which this row's tokens start.
__device__ __forceinline__ void emit_row_dense(const uint32_t (&w)[4], const RowCtx& rc, const uint2* tab, const uint32_t* runcls, uint32_t* stage,
                                               uint32_t& base) {
    DenseMasks M;
    dense_masks(w, rc, M);
    unsigned long long longany = 0;
    unsigned long long zzprev = M.z15s & M.Z[0];  // ZZ(-1)
    uint32_t qlo[4], qhi[4], qlen[4];
    uint32_t slow = 0;  // bit q: quad q is emitted token by token
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
        uint2 c[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = 4 * qd + e;
            const unsigned long long Nx = M.nx(i);
            const unsigned long long ZZ = M.Z[i] & Nx;
            const unsigned long long A = zzprev & ~Nx;
            const unsigned long long P2 = M.p2(i);
            const unsigned long long Len2 = A & ~P2, Long = A & P2;
            const unsigned long long Dead = ZZ | Long;
            uint32_t off = byte_times8(w[i >> 2], i & 3);  // table entries are 8 bytes
            off = lane_bit(Dead) ? 261u * 8u : off;                    // {0, 0}
            off = lane_bit(Len2) ? 256u * 8u : off;
            c[e] = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint8_t*>(tab) + off);
            longany |= Long;
            zzprev = ZZ;
        }
        __builtin_amdgcn_sched_barrier(0);  // four lookups in flight, one wait
        const uint32_t s1 = c[0].y + c[1].y, s2 = c[2].y + c[3].y;
        const uint32_t v01 = c[0].x | (c[1].x << (c[0].y & 31u));
        const uint32_t v23 = c[2].x | (c[3].x << (c[2].y & 31u));
        const uint64_t V = (uint64_t)v01 | ((uint64_t)v23 << (s1 & 63u));
        qlo[qd] = (uint32_t)V;
        qhi[qd] = (uint32_t)(V >> 32);
        qlen[qd] = s1 + s2;
        if ((s1 > 32u) | (s2 > 32u)) slow |= 1u << qd;  // a pair of codes longer than 32 bits (deep trees only)
        // pin the quad's string here: left to itself the scheduler sums the lengths first (they feed the scan) and parks all
        // sixteen table entries in scratch until the strings are joined after it
        asm volatile("" : "+v"(qlo[qd]), "+v"(qhi[qd]), "+v"(qlen[qd]), "+v"(slow));
        __builtin_amdgcn_sched_barrier(0);
    }
    LaneView v{};
    const bool anylong = longany != 0;
    if (anylong) {  // (wave-uniform) runs of three and more zeros: their bits join the quad's length, the quad goes the slow way
        v = lane_view(w, rc, M);
        uint32_t t = v.lng;
        while (t) {
            const uint32_t i = (uint32_t)__builtin_ctz(t);
            t &= t - 1;
            const uint32_t bits = run_bits_tab(tab, runcls, run_ending_at(i, v.lits, v.zb));
            const uint32_t qd = i >> 2;
            qlen[0] += qd == 0 ? bits : 0u;
            qlen[1] += qd == 1 ? bits : 0u;
            qlen[2] += qd == 2 ? bits : 0u;
            qlen[3] += qd == 3 ? bits : 0u;
            slow |= 1u << qd;
        }
    }
    const uint32_t lead = lane_id() == 0 ? lead_run(rc, M) : 0u;  // the run in front of the row: lane 0, ahead of its first token
    if (lead_run(rc, M)) {
        if (lead) {
            qlen[0] += run_bits_tab(tab, runcls, lead);
            slow |= 1u;
        }
    }
    const uint32_t tot = qlen[0] + qlen[1] + qlen[2] + qlen[3];
    const uint32_t inc = wave_scan_add(tot);
    uint32_t pq[4];
    pq[0] = base + inc - tot;
    pq[1] = pq[0] + qlen[0];
    pq[2] = pq[1] + qlen[1];
    pq[3] = pq[2] + qlen[2];
    base += read_lane(inc, 63);
#pragma unroll
    for (int qd = 0; qd < 4; ++qd)
        if (qlen[qd] && !((slow >> qd) & 1u)) or_bits64(stage, pq[qd], qlo[qd], qhi[qd], qlen[qd]);
    if (__ballot(slow != 0)) {
        if (!anylong) v = lane_view(w, rc, M);
        const uint32_t qmask = ((slow & 1u) ? 0x000Fu : 0u) | ((slow & 2u) ? 0x00F0u : 0u) | ((slow & 4u) ? 0x0F00u : 0u) | ((slow & 8u) ? 0xF000u : 0u);
        uint32_t ts = (v.lits | v.len1 | v.len2 | v.lng) & qmask;
        uint32_t prevq = 4, pp = 0;
        while (ts) {
            const uint32_t i = (uint32_t)__builtin_ctz(ts);
            ts &= ts - 1;
            const uint32_t qd = i >> 2;
            if (qd != prevq) {
                pp = qd == 0 ? pq[0] : qd == 1 ? pq[1] : qd == 2 ? pq[2] : pq[3];
                if (qd == 0 && lead) pp += emit_run(stage, tab, runcls, pp, lead);  // (its quad 0 starts with the literal that ends the run)
            }
            prevq = qd;
            if ((v.lng >> i) & 1u) {
                pp += emit_run(stage, tab, runcls, pp, run_ending_at(i, v.lits, v.zb));
            } else {
                const uint32_t idx = ((v.len2 >> i) & 1u) ? 256u : granule_byte_dyn(w[0], w[1], w[2], w[3], i);
                const uint2 cw = tab[idx];
                or_token(stage, pp, cw.x, 0u, cw.y);
                pp += cw.y;
            }
        }
    }
}

// ===========================================================================
// sparse rows: the row's literals, in order, as (position, value) entries in a per-wave queue; the wave then takes the
// queue 64 entries at a time -- the run in front of a literal is the gap to the entry before it, so nothing has to be
// chained from lane to lane.  A row that ends the block appends a virtual entry at in_size (the run that reaches the end).
// ===========================================================================
constexpr uint32_t kNoLit = 0x100u;

// the lane's non-zero bytes (bit i = byte i) -- bytes behind the block end were masked to zero at the load
__device__ __forceinline__ uint32_t lane_lits(const uint32_t (&w)[4]) { return ~zero_mask16(w[0], w[1], w[2], w[3]) & 0xFFFFu; }

// fills the queue; returns the number of entries (0: nothing ends in this row)
__device__ __forceinline__ uint32_t sparse_queue(const uint32_t (&w)[4], const RowCtx& rc, uint32_t in_size, uint32_t lits, uint32_t* queue,
                                                 bool& queued) {
    const uint32_t l = lane_id();
    const uint32_t n = (uint32_t)__popc(lits);
    const uint32_t incl = wave_scan_add(n);
    const uint32_t T = read_lane(incl, 63) + (rc.last ? 1u : 0u);
    queued = queue != nullptr && T <= kQueueEntries;
    if (T == 0 || !queued) return T;
    uint32_t k = incl - n, t = lits;
    const uint32_t pos0 = rc.rb + 16u * l;
    while (t) {
        const uint32_t i = (uint32_t)__builtin_ctz(t);
        t &= t - 1;
        queue[k++] = ((pos0 + i) << 9) | granule_byte_dyn(w[0], w[1], w[2], w[3], i);
    }
    if (rc.last && l == 0) queue[T - 1u] = (in_size << 9) | kNoLit;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return T;
}

// entry k of the queue and the zeros in front of it
__device__ __forceinline__ void queue_entry(const uint32_t* queue, const RowCtx& rc, uint32_t k, uint32_t T, uint32_t& R, uint32_t& lit) {
    R = 0;
    lit = kNoLit;
    if (k < T) {
        const uint32_t e = queue[k];
        const uint32_t before = k ? (queue[k - 1u] >> 9) + 1u : rc.rb - rc.zb0;  // position behind the literal before this one
        lit = e & 0x1FFu;
        R = (e >> 9) - before;
    }
}

// without a queue: f(R, lit) for the lane's own entries, in stream order (the gap to the literal of a lower lane comes from a scan)
template <class F>
__device__ __forceinline__ void lane_entries(const uint32_t (&w)[4], const RowCtx& rc, uint32_t in_size, uint32_t lits, F f) {
    const uint32_t l = lane_id();
    const uint32_t pos0 = rc.rb + 16u * l;
    const uint32_t mine = lits ? pos0 + (32u - (uint32_t)__clz((int)lits)) : 0u;  // position behind this lane's last literal
    const uint32_t incl = wave_scan_incl(mine, 0u, [](uint32_t a, uint32_t b) { return a > b ? a : b; });
    uint32_t before = dpp<0x138>(0u, incl);  // wave_shr:1: behind the last literal of the lanes below
    before = max(l ? before : 0u, rc.rb - rc.zb0);
    uint32_t t = lits;
    while (t) {
        const uint32_t i = (uint32_t)__builtin_ctz(t);
        t &= t - 1;
        f(pos0 + i - before, granule_byte_dyn(w[0], w[1], w[2], w[3], i));
        before = pos0 + i + 1u;
    }
    if (rc.last && l == 63u) f(in_size - max(read_lane(incl, 63), rc.rb - rc.zb0), kNoLit);  // the run that reaches the block end
}

// The entries of a wave's sparse rows, kept for k_encode: a block whose sixteen segments all fit their list is encoded from
// the lists alone -- no second pass over its 64 KiB of mostly zeros (entry lists).
constexpr uint32_t kListCap = 512;             // entries per 4 KiB segment
constexpr uint32_t kListNone = 0xFFFFFFFFu;    // segment info: no list (a dense row, or too many entries)
struct ListSink {
    uint32_t* dst;  // this wave's list
    uint32_t n;     // entries so far, or kListNone
};

__device__ __forceinline__ void hist_row_sparse(const uint32_t (&w)[4], const RowCtx& rc, uint32_t in_size, uint32_t* h, const uint32_t* runcls,
                                                uint32_t* queue, ListSink& sink) {
    if (rc.valid == 0u) return;
    if (!rc.last && !__ballot((w[0] | w[1] | w[2] | w[3]) != 0u)) return;  // nothing but zeros, and the block goes on: no token ends here
    const uint32_t lits = lane_lits(w);
    bool queued;
    const uint32_t T = sparse_queue(w, rc, in_size, lits, queue, queued);
    if (T == 0) return;
    if (queued && sink.n != kListNone && sink.n + T <= kListCap) {
        for (uint32_t c = lane_id(); c < T; c += 64) sink.dst[sink.n + c] = queue[c];
        sink.n += T;
    } else {
        sink.n = kListNone;
    }
    if (queued) {
        const uint32_t l = lane_id();
        for (uint32_t c = 0; c < T; c += 64) {
            uint32_t R, lit;
            queue_entry(queue, rc, c + l, T, R, lit);
            if (R) hist_run(h, runcls, R);
            if (lit < 256u) atomicAdd(&h[lit], 1u);
        }
        __builtin_amdgcn_wave_barrier();  // the queue is reused by the next row
    } else {
        lane_entries(w, rc, in_size, lits, [&](uint32_t R, uint32_t lit) {
            if (R) hist_run(h, runcls, R);
            if (lit < 256u) atomicAdd(&h[lit], 1u);
        });
    }
}

// one entry per lane (R zeros, then the literal unless lit == kNoLit; R = 0 and no literal: nothing), in lane order
__device__ __forceinline__ void emit_entries(uint32_t R, uint32_t lit, const uint2* tab, const uint32_t* runcls, uint32_t* stage, uint32_t& base) {
    const uint32_t q = run_caps(R);
    const uint32_t rem = R - q * kRunCap;
    uint64_t rv = 0;
    uint32_t rl = 0;
    if (rem) run_token(tab, runcls, rem, rv, rl);
    const uint2 lc = tab[lit < 256u ? lit : 261u];
    const uint64_t V = rv | ((uint64_t)lc.x << rl);  // <= 38 + 24 bits
    const uint32_t vlen = rl + lc.y;
    const uint32_t caplen = q ? q * (tab[260].y + 14u) : 0u;
    const uint32_t len = caplen + vlen;
    const uint32_t inc = wave_scan_add(len);
    uint32_t pos = base + inc - len;
    base += read_lane(inc, 63);
    if (q) pos += emit_run(stage, tab, runcls, pos, q * kRunCap);
    if (vlen) or_bits64(stage, pos, (uint32_t)V, (uint32_t)(V >> 32), vlen);
}

__device__ __forceinline__ void emit_row_sparse(const uint32_t (&w)[4], const RowCtx& rc, uint32_t in_size, const uint2* tab, const uint32_t* runcls,
                                                uint32_t* stage, uint32_t& base, uint32_t* queue) {
    if (rc.valid == 0u) return;
    if (!rc.last && !__ballot((w[0] | w[1] | w[2] | w[3]) != 0u)) return;
    const uint32_t lits = lane_lits(w);
    bool queued;
    const uint32_t T = sparse_queue(w, rc, in_size, lits, queue, queued);
    if (T == 0) return;
    if (queued) {
        const uint32_t l = lane_id();
        for (uint32_t c = 0; c < T; c += 64) {
            uint32_t R, lit;
            queue_entry(queue, rc, c + l, T, R, lit);
            emit_entries(R, lit, tab, runcls, stage, base);
        }
        __builtin_amdgcn_wave_barrier();
    } else {
        // no queue (a heavy block's image leaves no room) or a row with too many entries: every lane walks its own, twice
        uint32_t bits = 0;
        lane_entries(w, rc, in_size, lits, [&](uint32_t R, uint32_t lit) {
            if (R) bits += run_bits_tab(tab, runcls, R);
            if (lit < 256u) bits += tab[lit].y;
        });
        const uint32_t inc = wave_scan_add(bits);
        uint32_t pos = base + inc - bits;
        base += read_lane(inc, 63);
        lane_entries(w, rc, in_size, lits, [&](uint32_t R, uint32_t lit) {
            if (R) pos += emit_run(stage, tab, runcls, pos, R);
            if (lit < 256u) {
                const uint2 cw = tab[lit];
                or_token(stage, pos, cw.x, 0u, cw.y);
                pos += cw.y;
            }
        });
    }
}

// dense or sparse?  (wave-uniform)  Dense needs a whole row; it pays when few granules are all zero.
__device__ __forceinline__ bool row_is_dense(const uint32_t (&w)[4], const RowCtx& rc) {
    if (rc.valid != 1024u) return false;
    return __popcll(__ballot((w[0] | w[1] | w[2] | w[3]) == 0u)) < 16;
}

// ===========================================================================
// the block in registers: four granules per lane, and what every row has to know about its surroundings
// ===========================================================================
// The loads alone: issued early (for the NEXT block while the current one is in its last phases), consumed by chain_block_rows.
__device__ __forceinline__ void issue_block_loads(const uint8_t* __restrict__ in, uint32_t in_size, uint32_t segmask, uint32_t (&W)[4][4]) {
    const uint32_t tid = thread_id(), l = tid & 63u, w = tid >> 6;
    const bool seg_nz = (segmask >> w) & 1u;  // a wave whose 4 KiB segment is all zero (front-end non-zero map) does not read HBM at all
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t pos = w * 4096u + r * 1024u + l * 16u;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (seg_nz && pos < in_size) v = *reinterpret_cast<const uint4*>(in + pos);  // plane rows are padded: a partial granule may over-read
        W[r][0] = v.x;
        W[r][1] = v.y;
        W[r][2] = v.z;
        W[r][3] = v.w;
    }
}

// scr: kEncWaves words of LDS.  Two barriers inside (they also publish whatever the caller wrote to LDS before).
__device__ __forceinline__ void chain_block_rows(uint32_t in_size, uint32_t (&W)[4][4], RowCtx (&rc)[4], uint32_t* scr) {
    const uint32_t tid = thread_id(), l = tid & 63u;
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));  // (wave-uniform: keeps the row contexts in scalar registers)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t pos = w * 4096u + r * 1024u + l * 16u;
        if (pos + 16u > in_size && pos < in_size) {  // the block's last, partial granule: bytes behind it read as zero
            const uint32_t nv = in_size - pos;
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) {
                const uint32_t lo = q * 4u;
                W[r][q] = nv >= lo + 4u ? W[r][q] : nv <= lo ? 0u : (W[r][q] & ((1u << ((nv - lo) * 8u)) - 1u));
            }
        }
    }
    // position behind the last literal of every row (0: the row holds none); the same for the wave, chained over the waves
    uint32_t behind[4];
    uint32_t wave_behind = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const unsigned long long nzg = __ballot((W[r][0] | W[r][1] | W[r][2] | W[r][3]) != 0u);
        behind[r] = 0;
        if (nzg) {
            const uint32_t ph = 63u - (uint32_t)__builtin_clzll(nzg);
            const uint32_t tr = read_lane(trail_zero_bytes(W[r][0], W[r][1], W[r][2], W[r][3]), ph);
            behind[r] = w * 4096u + r * 1024u + 16u * ph + 16u - tr;
        }
        wave_behind = max(wave_behind, behind[r]);
    }
    if (l == 0) scr[w] = wave_behind;
    __syncthreads();
    uint32_t sf = l < (uint32_t)kEncWaves ? scr[l] : 0u;
    sf = row_scan_prefix(sf, 0u, [](uint32_t a, uint32_t b) { return a > b ? a : b; });
    uint32_t c = w ? read_lane(sf, w - 1u) : 0u;  // behind the last literal before this wave
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t rb = w * 4096u + r * 1024u;
        rc[r].rb = rb;
        rc[r].valid = rb >= in_size ? 0u : min(1024u, in_size - rb);
        rc[r].last = (rc[r].valid > 0u && rb + rc[r].valid == in_size) ? 1u : 0u;
        rc[r].zb0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(rb - min(c, rb)));
        c = max(c, behind[r]);
    }
    __syncthreads();  // scr may be reused by the caller
}

__device__ __forceinline__ void load_block_rows(const uint8_t* __restrict__ in, uint32_t in_size, uint32_t segmask, uint32_t (&W)[4][4],
                                                RowCtx (&rc)[4], uint32_t* scr) {
    issue_block_loads(in, in_size, segmask, W);
    chain_block_rows(in_size, W, rc, scr);
}

// rotate the rows through W[0] / rc[0]: the row bodies exist once in the instruction stream; four turns put everything back
__device__ __forceinline__ void rotate_rows(uint32_t (&W)[4][4], RowCtx (&rc)[4]) {
    const uint32_t t0 = W[0][0], t1 = W[0][1], t2 = W[0][2], t3 = W[0][3];
    const RowCtx tc = rc[0];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        W[q][0] = W[q + 1][0];
        W[q][1] = W[q + 1][1];
        W[q][2] = W[q + 1][2];
        W[q][3] = W[q + 1][3];
        rc[q] = rc[q + 1];
    }
    W[3][0] = t0;
    W[3][1] = t1;
    W[3][2] = t2;
    W[3][3] = t3;
    rc[3] = tc;
}

// A wave whose four rows are all sparse (the segments of the "medium" blocks: every 4 KiB non-zero, a few literals per row)
// takes them as ONE queue: two packed scans instead of four, one pass over the entries instead of one per row, and the
// queue is the segment's entry list as it stands.  Between sparse rows nothing special happens at a row boundary: the zeros
// in front of a row's first literal are the gap to the entry before it, whichever row that one came from; only entry 0
// looks outside the segment (rc[0].zb0).  False: too many entries for the queue -- the caller goes row by row.
__device__ __forceinline__ bool hist_segment_sparse(const uint32_t (&W)[4][4], const RowCtx (&rc)[4], uint32_t in_size, uint32_t* h,
                                                    const uint32_t* runcls, uint32_t* queue, ListSink& sink) {
    const uint32_t l = lane_id();
    uint32_t lits[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) lits[r] = lane_lits(W[r]);  // (bytes behind the block end were masked to zero at the load)
    const uint32_t n01 = (uint32_t)__popc(lits[0]) | ((uint32_t)__popc(lits[1]) << 16);
    const uint32_t n23 = (uint32_t)__popc(lits[2]) | ((uint32_t)__popc(lits[3]) << 16);
    const uint32_t inc01 = wave_scan_add(n01), inc23 = wave_scan_add(n23);
    const uint32_t tot01 = read_lane(inc01, 63), tot23 = read_lane(inc23, 63);
    const uint32_t last = rc[0].last | rc[1].last | rc[2].last | rc[3].last;  // the block ends in this segment: the run that reaches its end
    const uint32_t S1 = tot01 & 0xFFFFu, S2 = S1 + (tot01 >> 16), S3 = S2 + (tot23 & 0xFFFFu);
    const uint32_t T = S3 + (tot23 >> 16) + (last ? 1u : 0u);
    if (T > kQueueEntries) return false;
    if (T == 0) return true;  // nothing but zeros, and the block goes on
    const uint32_t ex01 = inc01 - n01, ex23 = inc23 - n23;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        uint32_t k = (r == 0 ? 0u : r == 1 ? S1 : r == 2 ? S2 : S3) + (((r < 2 ? ex01 : ex23) >> ((r & 1) * 16)) & 0xFFFFu);
        uint32_t t = lits[r];
        const uint32_t pos0 = rc[r].rb + 16u * l;
        while (t) {
            const uint32_t i = (uint32_t)__builtin_ctz(t);
            t &= t - 1;
            queue[k++] = ((pos0 + i) << 9) | granule_byte_dyn(W[r][0], W[r][1], W[r][2], W[r][3], i);
        }
    }
    if (last && l == 0) queue[T - 1u] = (in_size << 9) | kNoLit;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (sink.n != kListNone && sink.n + T <= kListCap) {
        for (uint32_t c = l; c < T; c += 64) sink.dst[sink.n + c] = queue[c];
        sink.n += T;
    } else {
        sink.n = kListNone;
    }
    const uint32_t before0 = rc[0].rb - rc[0].zb0;
    for (uint32_t c = 0; c < T; c += 64) {
        const uint32_t k = c + l;
        if (k < T) {
            const uint32_t e = queue[k];
            const uint32_t before = k ? (queue[k - 1u] >> 9) + 1u : before0;  // position behind the literal before this one
            const uint32_t R = (e >> 9) - before, lit = e & 0x1FFu;
            if (R) hist_run(h, runcls, R);
            if (lit < 256u) atomicAdd(&h[lit], 1u);
        }
    }
    __builtin_amdgcn_wave_barrier();  // the queue is reused by the next block
    return true;
}

__device__ __forceinline__ void hist_rows(uint32_t (&W)[4][4], RowCtx (&rc)[4], uint32_t in_size, uint32_t* myhist, const uint32_t* runcls,
                                          uint32_t* queue, ListSink& sink) {
    if (queue != nullptr && !(row_is_dense(W[0], rc[0]) | row_is_dense(W[1], rc[1]) | row_is_dense(W[2], rc[2]) | row_is_dense(W[3], rc[3]))) {
        if (hist_segment_sparse(W, rc, in_size, myhist, runcls, queue, sink)) return;
    }
#pragma unroll 1
    for (int r = 0; r < 4; ++r) {
        if (row_is_dense(W[0], rc[0])) {
            hist_row_dense(W[0], rc[0], myhist, runcls);
            sink.n = kListNone;
        } else {
            hist_row_sparse(W[0], rc[0], in_size, myhist, runcls, queue, sink);
        }
        rotate_rows(W, rc);
    }
}

// ===========================================================================
// k_hist
// ===========================================================================
// Every wave counts the tokens whose LAST byte lies in its own 4 KiB segment into its own LDS histogram.  Two results:
//   hist     [hb][264] u32     the block's histogram (sum of the 16), input of k_tree
//   seghist  [hb][16][264] u16 the per-segment histograms: with the code lengths they give k_tree the stream bit
//                              at which each segment's tokens start, so k_encode needs no bit-count pass and no
//                              cross-wave prefix of its own (a 4 KiB segment holds <= 4096 + 4 tokens: u16 is enough)
struct HistLds {
    uint32_t hist[kEncWaves][kSymStride];
    uint32_t queue[kEncWaves][kQueueEntries];
    uint32_t runcls[kRunClsEntries];
    uint32_t scr[2 * kEncWaves];
    uint32_t slot;
};
__shared__ HistLds g_h;

// The blocks k_hist takes (a plane in use, more than kSmallSegments non-zero segments), in any order: work_ctr[2] = count.
__global__ __launch_bounds__(256) void k_histlist(const uint32_t* __restrict__ nzflag, const uint32_t* __restrict__ nbuse, Geom g, uint32_t nhb_total,
                                                 uint32_t* __restrict__ list, uint32_t* __restrict__ count, uint32_t psel_arg) {
    const uint32_t psel = RSPT_DIAG_ONLY(psel_arg);  // timing probes (diagnostic builds only): leave out plane 0 (bit 0) / planes >= 1 (bit 1)
    const uint32_t v = blockIdx.x * 256u + threadIdx.x, l = lane_id();
    bool take = false;
    uint32_t hb = 0;
    if (v < nhb_total) {
        // plane fastest: consecutive list entries alternate between the dense plane and the light ones (a persistent
        // workgroup that takes them in turn keeps a CU's two workgroups in different phases more often)
        const uint32_t k = v % kMaxPlanes, j = (v / kMaxPlanes) % g.nblk, b = v / (kMaxPlanes * g.nblk);
        hb = hb_index(g, b, k, j);
        take = k < nbuse[b] && (uint32_t)__popc(nzflag[hb]) > kSmallSegments;
        if (psel && ((k == 0 && (psel & 1u)) || (k >= 1 && (psel & 2u)))) take = false;
    }
    const unsigned long long m = __ballot(take);
    if (!m) return;
    uint32_t base = 0;
    if (l == (uint32_t)__builtin_ctzll(m)) base = atomicAdd(count, (uint32_t)__popcll(m));
    base = read_lane(base, (uint32_t)__builtin_ctzll(m));
    if (take) list[base + (uint32_t)__popcll(m & ((1ull << l) - 1ull))] = hb;
}

__global__ __launch_bounds__(kEncThreads, 8) void k_hist(const uint8_t* __restrict__ planes, Geom g, const uint32_t* __restrict__ nzflag,
                                                     uint32_t* __restrict__ hist, uint32_t* __restrict__ seghist, uint32_t* __restrict__ counter,
                                                     const uint32_t* __restrict__ list, const uint32_t* __restrict__ count,
                                                     uint32_t* __restrict__ lists, uint2* __restrict__ listinfo) {
    HistLds& d = g_h;
    for (uint32_t i = threadIdx.x; i < (uint32_t)kEncWaves * kSymStride; i += kEncThreads) (&d.hist[0][0])[i] = 0;
    if (threadIdx.x < kRunClsEntries) d.runcls[threadIdx.x] = run_class_entry(threadIdx.x);
    const uint32_t n = *count;
    // Persistent, software-pipelined: the first block is static, the rest come from a counter whose fetch-add is issued
    // one block ahead; the NEXT block's granules are loaded as soon as the current block's rows are done, so that their
    // latency falls into the reduction (and the wait for the slowest wave) instead of the start of the next pass.
    auto block_of = [&](uint32_t hb, const uint8_t*& in, uint32_t& in_size) {
        const uint32_t j = hb % g.nblk, k = (hb / g.nblk) % kMaxPlanes, b = hb / (g.nblk * kMaxPlanes);
        in_size = min(kHzrBlock, g.N - j * kHzrBlock);
        in = planes + ((size_t)b * kMaxPlanes + k) * g.plane_stride + (size_t)j * kHzrBlock;
    };
    uint32_t cur = blockIdx.x;
    uint32_t W[4][4];
    uint32_t hb = 0, in_size = 0;
    if (cur < n) {
        hb = list[cur];
        const uint8_t* in;
        block_of(hb, in, in_size);
        issue_block_loads(in, in_size, nzflag[hb], W);
    }
    while (cur < n) {
        const uint32_t tid = thread_id();
        if (tid == 0) d.slot = gridDim.x + atomicAdd(counter, 1u);  // (read behind the chain's barriers)
        RowCtx rc[4];
        chain_block_rows(in_size, W, rc, d.scr);  // (its barriers also order the zeroing below against this block's adds)
        const uint32_t nxt = d.slot;
        uint32_t hbn = 0, segn = 0;
        if (nxt < n) {
            hbn = list[nxt];
            segn = nzflag[hbn];
        }
        {
            const uint32_t wv = tid >> 6;
            ListSink sink{lists + ((size_t)hb * kEncWaves + wv) * kListCap, 0u};
            const uint32_t before = rc[0].rb - rc[0].zb0;  // position behind the last literal in front of this wave's segment
#ifdef RSPT_PROBE_NOROWS  // timing probe (never in the product): everything but the row bodies
            asm volatile("" ::"v"(W[0][0] ^ W[1][1] ^ W[2][2] ^ W[3][3]), "s"(rc[0].zb0 + rc[1].zb0 + rc[2].zb0 + rc[3].zb0));
#else
            hist_rows(W, rc, in_size, d.hist[wv], d.runcls, d.queue[wv], sink);
#endif
            if ((tid & 63u) == 0) listinfo[(size_t)hb * kEncWaves + wv] = make_uint2(sink.n, before);
        }
        uint32_t in_size_n = 0;
        if (nxt < n) {  // the rows are done: their registers take the next block
            const uint8_t* in_n;
            block_of(hbn, in_n, in_size_n);
            issue_block_loads(in_n, in_size_n, segn, W);
        }
        __syncthreads();
#ifndef RSPT_PROBE_NOREDUCE  // timing probe (never in the product): no block / segment histograms leave the workgroup
        if (tid < (uint32_t)kSymStride) {
            uint32_t t = 0;
#pragma unroll
            for (int wv = 0; wv < kEncWaves; ++wv) t += d.hist[wv][tid];
            hist[(size_t)hb * kSymStride + tid] = t;
        }
        uint32_t* sh = seghist + (size_t)hb * (kSegHistStride / 2);
        for (uint32_t q = tid; q < kSegHistStride / 2; q += kEncThreads) {
            const uint32_t lo = (&d.hist[0][0])[2 * q], hi = (&d.hist[0][0])[2 * q + 1];
            sh[q] = lo | (hi << 16);
        }
        __syncthreads();  // everyone has read the histograms
        for (uint32_t i = tid; i < (uint32_t)kEncWaves * kSymStride; i += kEncThreads) (&d.hist[0][0])[i] = 0;
#endif
        cur = nxt;
        hb = hbn;
        in_size = in_size_n;
    }
}

// ===========================================================================
// k_encode
// ===========================================================================
// RSPT_DIAG builds (never the product library): thread 0 stores s_memtime at the phase seams of the first 6000 list entries
#ifdef RSPT_DIAG
#define ENC_STAMP(i)                                                                                              \
    do {                                                                                                          \
        if (stamps && threadIdx.x == 0 && list_pos < 6000u) stamps[list_pos * 16u + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define ENC_STAMP(i) \
    do {             \
    } while (0)
#endif

constexpr uint32_t kYieldSmallBlocks = 2048;  // k_encode leaves an eighth of its slots to k_encode_small from this many small blocks on
struct EncRowsLds {
    uint2 tab[kSymStride];  // {code, length} per lookup index; 261..263 = {0, 0} (a byte that ends no token)
    uint32_t crc[4][256];   // multiplication by x^(8*4096) as four byte-indexed lookups (CrcConsts::shift[78])
    uint32_t runcls[kRunClsEntries];
    uint32_t scr[2 * kEncWaves];
    uint32_t wsum[kEncWaves];
    uint32_t crc_out;
    uint32_t slot;
    uint32_t zero_word;          // stays 0: what the CRC reads in front of the image (word -1)
    uint32_t stage[kStagePhys];  // X || payload (+ read slack); light blocks queue sparse-row entries in its tail
};
static_assert(sizeof(EncRowsLds) <= 80 * 1024, "two workgroups per CU");
__shared__ EncRowsLds g_e;

__device__ __forceinline__ void encode_block_rows(uint32_t hb, uint8_t* __restrict__ planes, const Geom& g, const uint32_t* __restrict__ nzflag,
                                                  const BlockMeta* __restrict__ meta, const uint32_t* __restrict__ cw,
                                                  const uint32_t* __restrict__ tdesc, const uint64_t* __restrict__ out_off,
                                                  const CrcConsts* __restrict__ cc, uint8_t* __restrict__ dst, uint64_t dst_stride,
                                                  const uint32_t* __restrict__ segbase, const uint32_t* __restrict__ lists,
                                                  const uint2* __restrict__ listinfo, unsigned long long* __restrict__ stamps, uint32_t list_pos) {
    EncRowsLds& d = g_e;
    ENC_STAMP(0);
    const uint32_t j = hb % g.nblk, k = (hb / g.nblk) % kMaxPlanes, b = hb / (g.nblk * kMaxPlanes);
    // three independent loads in one round trip (their addresses depend on the block index only)
    const BlockMeta m = meta[hb];
    const uint64_t off = out_off[hb];
    const uint32_t segmask = nzflag[hb];
    if (off == ~0ull) return;  // stream does not fit dst_stride (flagged in sizes[b])
    const uint32_t tid = thread_id(), l = tid & 63u;
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    uint8_t* o = dst + (size_t)b * dst_stride + off;
    const uint32_t in_size = min(kHzrBlock, g.N - j * kHzrBlock);
    uint8_t* in = planes + ((size_t)b * kMaxPlanes + k) * g.plane_stride + (size_t)j * kHzrBlock;
    const uint32_t L = m.payload_len;
    const uint32_t zwords = ((L + 4u) >> 2) + 24u;  // the part of the image the payload (and the CRC's read slack) touches

    uint32_t W[4][4];
    RowCtx rc[4];
    if (m.mode == kModeHuff) {
        const uint32_t first_base = segbase[(size_t)hb * kEncWaves];  // 0xFFFFFFFF: no offsets from k_tree (its own-histogram blocks)
        uint32_t base = segbase[(size_t)hb * kEncWaves + w];           // stream bit at which this wave's tokens start
        if (tid < (uint32_t)kSymStride) {
            const uint32_t c = tid < (uint32_t)kNumSym ? cw[(size_t)hb * kSymStride + tid] : 0u;
            d.tab[tid] = make_uint2(c & 0x00FFFFFFu, c >> 24);
        }
        const bool own_bits = first_base == 0xFFFFFFFFu;  // (block-uniform)
        // Did k_hist leave the entry lists of all sixteen segments?  Then the block is encoded from them alone.
        const uint2 li = listinfo[(size_t)hb * kEncWaves + (l & 15u)];
        const bool from_lists = !own_bits && !__ballot(li.x == kListNone);
        if (from_lists) {
            for (uint32_t i = tid; i < zwords; i += kEncThreads) d.stage[i] = 0;
            __syncthreads();  // the table and the zeroed image are in place
            const uint32_t twords = (m.tree_bits + 31u) >> 5;
            if (tid < twords) atomicOr(&d.stage[1u + tid], tdesc[(size_t)hb * kTdescWords + tid]);
            const uint32_t n = read_lane(li.x, w), before0 = read_lane(li.y, w);
            const uint32_t* list = lists + ((size_t)hb * kEncWaves + w) * kListCap;
            const bool wipe = block_is_wiped(m);
            for (uint32_t c = 0; c < n; c += 64) {
                const uint32_t kq = c + l;
                uint32_t R = 0, lit = kNoLit;
                if (kq < n) {
                    const uint32_t e = list[kq];
                    const uint32_t before = kq ? (list[kq - 1u] >> 9) + 1u : before0;
                    lit = e & 0x1FFu;
                    R = (e >> 9) - before;
                    // clean-block invariant: a light block leaves zeros behind (the granule of every literal)
                    if (wipe && lit < 256u) *reinterpret_cast<uint4*>(in + ((e >> 9) & ~15u)) = make_uint4(0, 0, 0, 0);
                }
                emit_entries(R, lit, d.tab, d.runcls, d.stage, base);
            }
        } else {
        if (!own_bits)
            for (uint32_t i = tid; i < zwords; i += kEncThreads) d.stage[i] = 0;
        load_block_rows(in, in_size, segmask, W, rc, d.scr);  // barriers inside publish the table and the zeroed image
        ENC_STAMP(1);
        if (own_bits) {
            // the block's histogram was taken by k_tree in one piece (few non-zero segments, but too big for
            // k_encode_small): count this wave's tokens here -- histogram in the (still unused) image, times the code lengths
            uint32_t* myhist = d.stage + w * kSymStride;
            for (uint32_t i = l; i < (uint32_t)kSymStride; i += 64) myhist[i] = 0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            ListSink none{nullptr, kListNone};
            // (the sparse-row queues of this pass start BEHIND the sixteen histograms: from kTokQueueBase = 4200 on, where the emit
            //  phase keeps them, wave 0's first 24 queue entries were wave 15's bins 240..263 -- the block-end run that wave 15 counts
            //  landed in entries wave 0 was reading back, and its own first entries in those bins: a wrong segment offset now and then,
            //  i.e. a damaged payload in about every second launch of such a block; tools/soak.py seed 1731)
            hist_rows(W, rc, in_size, myhist, d.runcls, d.stage + kOwnHistQueueBase + w * kQueueEntries, none);
            __builtin_amdgcn_wave_barrier();
            uint32_t bits = 0;
            for (uint32_t s = l; s < (uint32_t)kNumSym; s += 64) bits += myhist[s] * (d.tab[s].y + run_extra_bits(s));  // (unused symbols: count 0)
            bits = wave_add_u32(bits);
            if (l == 0) d.wsum[w] = bits;
            __syncthreads();
            uint32_t ws = l < (uint32_t)kEncWaves ? d.wsum[l] : 0u;
            ws = row_scan_prefix(ws, 0u, [](uint32_t a, uint32_t c) { return a + c; });
            base = 32u + m.tree_bits + (w ? read_lane(ws, w - 1u) : 0u);  // the payload starts at image byte 4
            for (uint32_t i = tid; i < zwords; i += kEncThreads) d.stage[i] = 0;
            __syncthreads();
        }
        if (block_is_wiped(m)) {
            // clean-block invariant: a light block leaves zeros behind.  Each lane is the only reader of its granules.
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (W[r][0] | W[r][1] | W[r][2] | W[r][3]) *reinterpret_cast<uint4*>(in + w * 4096u + r * 1024u + l * 16u) = make_uint4(0, 0, 0, 0);
        }
        const uint32_t twords = (m.tree_bits + 31u) >> 5;  // tree description (hzr_encode.c:177-219), from logical word 1
        if (tid < twords) atomicOr(&d.stage[1u + tid], tdesc[(size_t)hb * kTdescWords + tid]);
        uint32_t* queue = L < kLightPayload ? d.stage + kTokQueueBase + w * kQueueEntries : nullptr;
#pragma unroll 1
        for (int r = 0; r < 4; ++r) {
#if defined(ENC_PROBE) && ENC_PROBE == 2  // no emit at all
            if (true) {
                asm volatile("" ::"v"(W[0][0] ^ W[0][1] ^ W[0][2] ^ W[0][3]));
            } else
#endif
            if (row_is_dense(W[0], rc[0]))
                emit_row_dense(W[0], rc[0], d.tab, d.runcls, d.stage, base);
            else
                emit_row_sparse(W[0], rc[0], in_size, d.tab, d.runcls, d.stage, base, queue);
            rotate_rows(W, rc);
        }
        }  // (rows path)
    } else {
        // PlainCopy (hzr_encode.c:307-339): the payload is the raw block; words past it stay defined (zero)
        for (uint32_t i = tid; i < zwords; i += kEncThreads) d.stage[i] = 0;
        load_block_rows(in, in_size, segmask, W, rc, d.scr);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t pos = w * 4096u + r * 1024u + l * 16u;
            if (pos < in_size) {
#pragma unroll
                for (uint32_t q = 0; q < 4; ++q) d.stage[1u + (pos >> 2) + q] = W[r][q];  // (bytes past in_size were masked to zero at the load)
            }
        }
    }
    if (tid == 0) atomicOr(&d.stage[0], cc->prefix);  // X (word 0 was zeroed above)
    ENC_STAMP(2);
    __syncthreads();
    ENC_STAMP(3);

    // ---- CRC-32C: V = X || payload (Lv = L + 4 bytes from image byte 0) as 4-byte virtual words counted from its END;
    //      lane tid owns the words tid, tid + 1024, ... (consecutive lanes read consecutive LDS words), Horner over its
    //      words with x^(8*4096) per step, then x^(8*4*(tid+1)) to the end of V; crc = ~raw(V) -------------------------
    {
        const int32_t Lv = (int32_t)L + 4;
        const uint32_t nvw = (uint32_t)(Lv + 3) >> 2;
        const uint32_t K = (nvw + kEncThreads - 1) / kEncThreads;  // steps of the fullest lane (block-uniform)
        // Virtual word r = bytes [Lv - 4(r+1), Lv - 4r) of V = image words a and a + 1 joined at byte (Lv & 3), a =
        // (Lv >> 2) - 1 - r; bytes in front of V are zero (a = -1 is the zero word in front of the image).  The byte offset
        // is the same for every word of the block, and only a lane's TOP word may be missing: one test, before the loop.
        const uint32_t sh = (uint32_t)Lv & 3u;
        const uint32_t* img = &d.zero_word + 1;  // img[-1] = 0, img[i] = stage[i]
        auto vword = [&](int32_t a) -> uint32_t { return __builtin_amdgcn_alignbyte(img[a + 1], img[a], sh); };
        auto times_x4096 = [&](uint32_t c) -> uint32_t {  // * x^(8*4096)
            return d.crc[0][c & 0xFFu] ^ d.crc[1][(c >> 8) & 0xFFu] ^ d.crc[2][(c >> 16) & 0xFFu] ^ d.crc[3][c >> 24];
        };
        uint32_t c = 0;
#if defined(ENC_PROBE) && ENC_PROBE == 1  // timing probes (never in the product): no CRC
        if (false) {
#else
        if (tid < nvw) {
#endif
            int32_t a = (Lv >> 2) - 1 - (int32_t)(tid + kEncThreads * (K - 1u));
            if (K > 1u) {
                c = a >= -1 ? vword(max(a, -1)) : 0u;  // (the top word: there for the low lanes only)
                c = times_x4096(c);
                for (uint32_t kk = K - 2u; kk >= 1u; --kk) {
                    a += (int32_t)kEncThreads;
                    c ^= vword(a);
                    c = times_x4096(c);
                }
                a += (int32_t)kEncThreads;
            }
            c ^= vword(a);
        }
        if (__ballot(c != 0)) {  // waves without data skip the shifts
            const uint32_t red = wave_xor_u32(gf_shift4(cc, l, c));    // every lane to the end of the wave's 64 words
            if (l == 0) d.wsum[w] = gf_shift(cc, 63u - 4u * w, red);  // x^(8*256*w): the wave's group to the end of V
        } else if (l == 0) {
            d.wsum[w] = 0;
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t t = 0;
            for (int i = 0; i < kEncWaves; ++i) t ^= d.wsum[i];
            d.crc_out = ~t;
        }
        __syncthreads();
        ENC_STAMP(4);
    }

    // ---- block header + payload to the stream (hzr_encode.c:475-481) --------
    if (tid == 0) {
        const uint32_t crc = d.crc_out;
        o[0] = (uint8_t)(L - 1);
        o[1] = (uint8_t)((L - 1) >> 8);
        o[2] = (uint8_t)crc;
        o[3] = (uint8_t)(crc >> 8);
        o[4] = (uint8_t)(crc >> 16);
        o[5] = (uint8_t)(crc >> 24);
        o[6] = (uint8_t)m.mode;
    }
    uint8_t* po = o + 7;
    const uint32_t head = min(L, (uint32_t)((4u - (uint32_t)(reinterpret_cast<uintptr_t>(po) & 3u)) & 3u));
    const uint32_t nd = (L - head) >> 2;
    const uint32_t tail = L - head - 4 * nd;
    if (tid < head) po[tid] = (uint8_t)stage_byte(d.stage, 4 + tid);
    if (tid < tail) po[head + 4 * nd + tid] = (uint8_t)stage_byte(d.stage, 4 + head + 4 * nd + tid);
    uint32_t* pw = reinterpret_cast<uint32_t*>(po + head);
    // payload bytes [head+4i, head+4i+4) = image bytes from 4+head+4i: off the LDS word grid by (head & 3)
#if !(defined(ENC_PROBE) && ENC_PROBE == 3)  // no copy-out
    for (uint32_t i = tid; i < nd; i += kEncThreads) pw[i] = __builtin_amdgcn_alignbyte(d.stage[i + 2], d.stage[i + 1], head);
#endif
    ENC_STAMP(5);
#ifdef RSPT_DIAG
    if (stamps && threadIdx.x == 0 && list_pos < 6000u) stamps[list_pos * 16u + 6u] = ((unsigned long long)m.mode << 32) | L;
#endif
}

__global__ __launch_bounds__(kEncThreads, 8) void k_encode(uint8_t* __restrict__ planes, Geom g, const uint32_t* __restrict__ nzflag,
                                                          const BlockMeta* __restrict__ meta, const uint32_t* __restrict__ cw,
                                                          const uint32_t* __restrict__ tdesc, const uint64_t* __restrict__ out_off,
                                                          const CrcConsts* __restrict__ cc, uint8_t* __restrict__ dst, uint64_t dst_stride,
                                                          WorkQueues* __restrict__ wq, const uint32_t* __restrict__ big_list,
                                                          const uint32_t* __restrict__ segbase, const uint32_t* __restrict__ lists,
                                                          const uint2* __restrict__ listinfo, unsigned long long* __restrict__ stamps, uint32_t yield) {
    (&g_e.crc[0][0])[threadIdx.x] = (&cc->shift[78][0][0])[threadIdx.x];  // multiplication by x^(8*4096): 4 x 256 entries, once per workgroup
    if (threadIdx.x < kRunClsEntries) g_e.runcls[threadIdx.x] = run_class_entry(threadIdx.x);
    if (threadIdx.x == 0) g_e.zero_word = 0;
    const uint32_t n_big = wq->n_big;
    // The small blocks are encoded beside this kernel (k_encode_small, side stream) -- but two of these workgroups hold every
    // wave slot of a CU, so the small ones only get in as these retire, and ran on for ~20 us behind the last big block.  When
    // there are enough small blocks for that to matter an eighth of the grid steps aside from the start: the small-block kernel
    // then ends well before this one (64-block batch: encode 0.333 + 0.021 behind it -> 0.328 + 0.010; with few small blocks
    // the full grid is the faster one).  Every workgroup takes the same decision from the same counter.  `yield` = 0: the small-block
    // kernel of this batch ran earlier, beside the next batch's front end (two batches in flight).
    const uint32_t act = yield && wq->n_small >= kYieldSmallBlocks && gridDim.x >= 16u ? gridDim.x - gridDim.x / 8u : gridDim.x;
    if (blockIdx.x >= act) return;
    // persistent: one big block per workgroup pass; the first one is static (index = workgroup id), the
    // rest come from a counter (one shared word sustains only ~88 fetch-adds per microsecond)
    for (uint32_t pass = 0;; ++pass) {
        __syncthreads();  // everyone is done with the previous block (and with the slot)
        if (threadIdx.x == 0) g_e.slot = pass == 0 ? blockIdx.x : act + atomicAdd(&wq->next_big, 1u);
        __syncthreads();
        const uint32_t i = g_e.slot;
        if (i >= n_big) break;
        encode_block_rows(big_list[i], planes, g, nzflag, meta, cw, tdesc, out_off, cc, dst, dst_stride, segbase, lists, listinfo, stamps, i);
    }
}

}  // namespace rspt
