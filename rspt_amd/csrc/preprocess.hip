// preprocess.hip -- front end of every packer: interleaved native samples ->
// channel-major byte planes (and the xdelta pre-transform).
//
// Restates, fused into one pass over HBM:
//   convert_native_to_i32   lib_signalpacker/utils.cpp:123-191 (LE branches)
//   delta_encode            utils.cpp:193-202   } over the FLAT nch*ns array:
//   offset_32(-128)         utils.cpp:215-219   } sample 0 of channel c follows
//   xor_encode_32           utils.cpp:221-230   } the last sample of channel c-1
//   byte-plane split        lib_signalpacker/signal_packer_base.cpp:40-68
//   nb escalation test      signal_packer_xdelta_hzr.cpp:59-69, as the exact
//                           "fits in nb bytes" reduction
//
// Layout.  The input is sample-major ([ns][nch][bps]): with lane <-> channel every
// wave load is a contiguous row segment, so the transform reads straight from HBM
// and keeps 16 samples of one channel in registers; only the plane bytes pass
// through LDS, so that every plane row leaves as 16-byte coalesced stores
// (T contiguous bytes per channel per plane).
#include "common.hpp"

namespace rspt {

// q / d for q < 2^32 / d by one multiply: M = ceil(2^32 / d)  (d >= 2; d == 1 handled by the caller)
__device__ __forceinline__ uint32_t magic_of(uint32_t d) { return (uint32_t)(((1ull << 32) + d - 1) / d); }
__device__ __forceinline__ uint32_t fast_div(uint32_t q, uint32_t d, uint32_t M) { return d == 1 ? q : __umulhi(q, M); }

// `be`: the sample's bytes are stored most significant first (convert_native_to_i32 with reverse_byte_order = true,
// lib_signalpacker/utils.cpp:127-137,145-154,162-170)
template <int BPS>
__device__ __forceinline__ int32_t sample_from_bytes(const uint8_t* p, bool aligned, bool be = false) {
    if (be) {
        if (BPS == 4) {
            if (aligned) {
                const uint32_t u = *reinterpret_cast<const uint32_t*>(p);
                return (int32_t)__builtin_amdgcn_perm(u, u, 0x00010203u);
            }
            return (int32_t)((uint32_t)p[3] | ((uint32_t)p[2] << 8) | ((uint32_t)p[1] << 16) | ((uint32_t)p[0] << 24));
        } else if (BPS == 3) {
            const uint32_t u = (uint32_t)p[2] | ((uint32_t)p[1] << 8) | ((uint32_t)p[0] << 16);
            return (int32_t)(u << 8) >> 8;
        } else if (BPS == 2) {
            const uint32_t u = (uint32_t)p[1] | ((uint32_t)p[0] << 8);
            return (int32_t)(u << 16) >> 16;
        }
        return (int32_t)(int8_t)p[0];
    }
    if (BPS == 4) {
        if (aligned) return *reinterpret_cast<const int32_t*>(p);
        uint32_t u = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
        return (int32_t)u;
    } else if (BPS == 3) {
        uint32_t u = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
        return (int32_t)(u << 8) >> 8;
    } else if (BPS == 2) {
        uint32_t u = (uint32_t)p[0] | ((uint32_t)p[1] << 8);
        return (int32_t)(u << 16) >> 16;
    } else {
        return (int32_t)(int8_t)p[0];
    }
}

// p[flat] of block `blk`, straight from HBM; flat < 0 -> 0 (the transforms start from 0)
template <int BPS>
__device__ __forceinline__ int32_t sample_global(const uint8_t* blk, const Geom& g, int64_t flat) {
    if (flat < 0) return 0;
    uint32_t f = (uint32_t)flat;
    uint32_t c = f / g.ns, s = f - c * g.ns;
    return sample_from_bytes<BPS>(blk + ((size_t)s * g.nch + c) * BPS, false, g.be != 0);
}

__device__ __forceinline__ uint32_t need_from_mask(uint32_t m) { return m < 0x80u ? 1u : m < 0x8000u ? 2u : m < 0x800000u ? 3u : 4u; }

// nbuse[b] = max(nb carried in, need(0..b)); the last value is carried to the next call (the reference's persistent
// escalation, signal_packer_xdelta_hzr.cpp:63-69).  One workgroup of any size; `wmax` = 16 words of LDS.
// needmask is read past the vector L1 (agent scope): its bits were set by other workgroups of the running kernel.
__device__ __forceinline__ void nb_scan_body(const uint32_t* __restrict__ needmask, uint32_t nblocks, uint32_t* __restrict__ nb_state,
                                             uint32_t* __restrict__ nbuse, int use_mask, uint32_t* wmax) {
    const uint32_t tid = threadIdx.x, nthr = blockDim.x;
    const uint32_t per = (nblocks + nthr - 1) / nthr;
    const uint32_t lo = min(nblocks, tid * per), hi = min(nblocks, lo + per);
    const uint32_t carry_in = *nb_state;
    auto need = [&](uint32_t b) -> uint32_t {
        return use_mask ? need_from_mask(__hip_atomic_load(&needmask[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : 0u;
    };
    uint32_t m = 0;
    for (uint32_t b = lo; b < hi; ++b) m = max(m, need(b));
    const uint32_t l = lane_id(), w = tid >> 6;
    const uint32_t v = wave_scan_incl(m, 0u, [](uint32_t a, uint32_t c) { return a > c ? a : c; });  // inclusive max-scan over the wave
    if (l == 63) wmax[w] = v;
    __syncthreads();
    uint32_t pre = carry_in;
    for (uint32_t i = 0; i < w; ++i) pre = max(pre, wmax[i]);
    const uint32_t excl = dpp<0x138>(0u, v);  // wave_shr:1
    uint32_t run = max(pre, l ? excl : 0u);
    for (uint32_t b = lo; b < hi; ++b) {
        run = max(run, need(b));
        nbuse[b] = run;
    }
    __syncthreads();
    if (tid == nthr - 1) *nb_state = max(pre, v);
}

// an item's 18 samples (16 + the two before them, for the delta and the xor) in registers
struct ItemRegs {
    uint32_t pv[16];
    uint32_t p1, p2;
};

// item q (channel q % nch, samples [s0 + 16 (q / nch), +16)) of the tile [s0, s0 + Tn) of block `blk`
template <int BPS, bool XDELTA>
__device__ __forceinline__ void load_item(const uint8_t* blk, const Geom& g, uint32_t m_nch, uint32_t s0, uint32_t Tn, bool aligned4,
                                          uint32_t q, ItemRegs& R) {
    const size_t rstride = (size_t)g.nch * BPS;
    const uint32_t grp = fast_div(q, g.nch, m_nch);
    const uint32_t c = q - grp * g.nch;
    const uint32_t t0 = grp << 4;
    const uint32_t cnt = min(16u, Tn - t0);
    const uint8_t* col = blk + ((size_t)(s0 + t0) * g.nch + c) * BPS;  // sample (s0+t0, c); next sample: + nch*BPS
    const bool be = g.be != 0;
    if (cnt == 16) {  // the common case carries no per-element branches: 16 loads in flight
#pragma unroll
        for (uint32_t e = 0; e < 16; ++e) R.pv[e] = (uint32_t)sample_from_bytes<BPS>(col + e * rstride, aligned4, be);
    } else {
#pragma unroll
        for (uint32_t e = 0; e < 16; ++e) R.pv[e] = e < cnt ? (uint32_t)sample_from_bytes<BPS>(col + e * rstride, aligned4, be) : 0u;
    }
    R.p1 = R.p2 = 0;
    if (XDELTA) {
        if (s0 + t0 >= 2) {  // same channel, two samples back
            R.p1 = (uint32_t)sample_from_bytes<BPS>(col - rstride, aligned4, be);
            R.p2 = (uint32_t)sample_from_bytes<BPS>(col - 2 * rstride, aligned4, be);
        } else {  // channel start: the flat array continues from the end of channel c-1
            const int64_t flat = (int64_t)c * g.ns + s0 + t0;
            R.p1 = (uint32_t)sample_global<BPS>(blk, g, flat - 1);
            R.p2 = (uint32_t)sample_global<BPS>(blk, g, flat - 2);
        }
    }
}

// what an item's transform needs to know about its tile
struct TileCtx {
    Geom g;
    uint32_t m_nch, s0, Tn, b, kfirst, kcount, RS, ablate;
    bool fixup;
    uint8_t* out;      // LDS rows [plane][channel][RS]
    uint32_t* s_nz;    // LDS dedupe masks [rel][plane][channel]
    uint32_t* nzflag;  // HBM non-zero map
    uint8_t* gplanes;  // (timing probe, diagnostic builds: plane bytes straight from the registers to HBM)
    uint32_t dirty_shift;
};

// item q, its 18 samples in R: transform, plane split into the LDS rows, non-zero map, escalation magnitude
template <int BPS, bool XDELTA, bool DIRECT = false>
__device__ __forceinline__ void transform_item(const ItemRegs& R, uint32_t q, const TileCtx& tc, uint32_t& mag, uint32_t& nz_seg,
                                               uint32_t& nz_done) {
    const Geom& g = tc.g;
    const uint32_t m_nch = tc.m_nch, s0 = tc.s0, Tn = tc.Tn, b = tc.b, kfirst = tc.kfirst, kcount = tc.kcount, RS = tc.RS, ablate = tc.ablate;
    const bool fixup = tc.fixup;
    uint8_t* out = tc.out;
    uint32_t* s_nz = tc.s_nz;
    uint32_t* nzflag = tc.nzflag;
    const uint32_t grp = fast_div(q, g.nch, m_nch);
    const uint32_t c = q - grp * g.nch;
    const uint32_t t0 = grp << 4;
    const uint32_t cnt = min(16u, Tn - t0);
    uint32_t p1 = R.p1, oprev = 0;  // p[i-1], o[i-1]
    if (XDELTA) {
        const bool flat0 = (c | s0 | t0) == 0;  // first element of the flat array: delta_encode and xor_encode_32 start from 0
        p1 = flat0 ? 0u : R.p1;
        oprev = flat0 ? 0u : (R.p1 - R.p2 - 128u);
    }
    const uint32_t* pv = R.pv;
    uint32_t vv[16];
    if (cnt == 16) {  // (all but the last group of a ragged tile)
#pragma unroll
        for (uint32_t e = 0; e < 16; ++e) {
            const uint32_t p = pv[e];
            if (XDELTA) {
                const uint32_t o = p - p1 - 128u;
                vv[e] = o ^ oprev;
                oprev = o;
                p1 = p;
                // sign-extend from the sample width, fold to a magnitude (escalation test)
                const int32_t x = BPS < 4 ? ((int32_t)(vv[e] << (32 - 8 * BPS)) >> (32 - 8 * BPS)) : (int32_t)vv[e];
                mag |= (uint32_t)(x ^ (x >> 31));
            } else {
                vv[e] = p;
            }
        }
    } else {
        // elements past cnt are computed on zeros and never stored or flagged: their plane bytes are masked off
#pragma unroll
        for (uint32_t e = 0; e < 16; ++e) {
            const uint32_t p = pv[e];
            uint32_t v;
            if (XDELTA) {
                const uint32_t o = p - p1 - 128u;
                v = o ^ oprev;
                oprev = o;
                p1 = p;
                const int32_t x = BPS < 4 ? ((int32_t)(v << (32 - 8 * BPS)) >> (32 - 8 * BPS)) : (int32_t)v;
                mag |= e < cnt ? (uint32_t)(x ^ (x >> 31)) : 0u;
            } else {
                v = p;
            }
            vv[e] = e < cnt ? v : 0u;
        }
    }
    // byte-plane split of four samples at a time: a 4 x 4 byte transpose in 8 v_perm_b32
    // (selector bytes 0-3 pick from the second operand, 4-7 from the first)
    uint32_t pw[4][4];
#pragma unroll
    for (uint32_t g4 = 0; g4 < 4; ++g4) {
        const uint32_t a0 = vv[4 * g4], a1 = vv[4 * g4 + 1], a2 = vv[4 * g4 + 2], a3 = vv[4 * g4 + 3];
        const uint32_t lo01 = __builtin_amdgcn_perm(a1, a0, 0x05010400u), hi01 = __builtin_amdgcn_perm(a1, a0, 0x07030602u);
        const uint32_t lo23 = __builtin_amdgcn_perm(a3, a2, 0x05010400u), hi23 = __builtin_amdgcn_perm(a3, a2, 0x07030602u);
        pw[0][g4] = __builtin_amdgcn_perm(lo23, lo01, 0x05040100u);
        pw[1][g4] = __builtin_amdgcn_perm(lo23, lo01, 0x07060302u);
        pw[2][g4] = __builtin_amdgcn_perm(hi23, hi01, 0x05040100u);
        pw[3][g4] = __builtin_amdgcn_perm(hi23, hi01, 0x07060302u);
    }
    uint32_t nzm = 0;  // bit k: plane k of this item holds a non-zero byte
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        if (k >= kfirst && k < kfirst + kcount) {
            if (DIRECT && ((ablate & (1u << 26)) || ((ablate & (1u << 27)) && k >= 1))) {  // timing probes: no LDS rows, 16-byte stores from the registers, all-zero units of clean blocks skipped (bit 27: the planes above 0 only)
                const bool nz = (pw[k][0] | pw[k][1] | pw[k][2] | pw[k][3]) != 0;
                const uint32_t bucket = ((c * g.ns + s0 + t0) >> 16) >> tc.dirty_shift;
                const bool dirty = (s_nz[8 * g.nch + 1 + (k - kfirst) * 4 + (bucket >> 5)] >> (bucket & 31u)) & 1u;
                if (nz || dirty)
                    *reinterpret_cast<uint4*>(tc.gplanes + ((size_t)b * kMaxPlanes + k) * g.plane_stride + (size_t)c * g.ns + s0 + t0) =
                        make_uint4(pw[k][0], pw[k][1], pw[k][2], pw[k][3]);
            } else {
                *reinterpret_cast<uint4*>(out + (size_t)((k - kfirst) * g.nch + c) * RS + t0) = make_uint4(pw[k][0], pw[k][1], pw[k][2], pw[k][3]);
            }
        }
        nzm |= ((pw[k][0] | pw[k][1] | pw[k][2] | pw[k][3]) != 0 ? 1u : 0u) << k;
    }
    // Non-zero map.  A thread mostly walks along one channel, and a tile row spans at most two 4 KiB segments: the
    // bits this thread has already forwarded for the segment it is in live in a register, so the common item costs
    // a compare and no LDS round trip.
    if (!fixup && !(ablate & 65536u)) {
        const uint32_t f0 = c * g.ns + s0 + t0, f1 = f0 + cnt - 1;  // flat range of this item
        const uint32_t seg = f0 >> 12;
        if (seg != nz_seg) {
            nz_seg = seg;
            nz_done = 0;
        }
        const bool two = (f1 >> 12) != seg;  // (an item straddling a segment edge marks both sides: conservative)
        const uint32_t need = two ? nzm : (nzm & ~nz_done);
        nz_done |= nzm;
        if (need) {
            const uint32_t jb = (c * g.ns + s0) >> 16;
            const uint32_t ja = f0 >> 16, jz = f1 >> 16;
            const uint32_t ba = 1u << (seg & 15u), bz = 1u << ((f1 >> 12) & 15u);
            uint32_t* za = &s_nz[((ja - jb) * 4) * g.nch + c];
            uint32_t* zz = &s_nz[((jz - jb) * 4) * g.nch + c];
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) {
                if (!((need >> k) & 1u)) continue;
                // the first setter in the workgroup forwards the bit to HBM (fire-and-forget)
                if (!(atomicOr(&za[k * g.nch], ba) & ba)) atomicOr(&nzflag[hb_index(g, b, k, ja)], ba);
                if (two && !(atomicOr(&zz[k * g.nch], bz) & bz)) atomicOr(&nzflag[hb_index(g, b, k, jz)], bz);
            }
        }
    }
}

// One workgroup = one tile of T samples x all channels of one block.
//   XDELTA  true : v = (p[i]-p[i-1]-128) ^ (p[i-1]-p[i-2]-128), flat order; accumulates needmask
//           false: v = p (hzr packer)
// Work item = (channel c, 16 consecutive samples); consecutive lanes take consecutive
// channels, so each of an item's 18 loads is, across the wave, one contiguous row
// segment of the sample-major input (256 B for 64 int32 channels) -- no input staging.
// The four plane words of an item go to LDS rows [plane][channel][T+16] and leave as
// 16-byte coalesced stores, T contiguous bytes per (plane, channel).
// Planes [kfirst, kfirst+kcount) are written.  The main pass (nbuse == nullptr) writes the
// planes the host knows are needed (kfirst = 0, kcount = nb as last seen) and computes the
// escalation magnitude and the non-zero map for all four planes; if nb escalates in this very
// call (signal_packer_xdelta_hzr.cpp:63-69), a second launch with nbuse != nullptr adds the
// missing planes for exactly the blocks that need them and exits at once for all others.
template <int BPS, bool XDELTA>
__global__ __launch_bounds__(1024) void k_tile_planes(const uint8_t* __restrict__ src, Geom g, uint32_t T, uint32_t kfirst, uint32_t kcount,
                                                     uint8_t* __restrict__ planes, uint32_t* __restrict__ needmask,
                                                     uint32_t* __restrict__ nzflag, const uint32_t* __restrict__ nbuse, uint32_t ablate,
                                                     uint32_t nblocks, uint32_t* __restrict__ ticket, uint32_t* __restrict__ nb_state,
                                                     uint32_t* __restrict__ nbuse_out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t tid = threadIdx.x, nthr = blockDim.x;
    const bool fixup = nbuse != nullptr;
    if (fixup && *nb_state <= kfirst) return;  // (no block of the batch needs more planes than the main pass wrote)
    const uint32_t tiles_per_block = (g.ns + T - 1) / T;
    // persistent: the grid is a few workgroups per CU, each walks tiles with a fixed stride
    const uint32_t total = tiles_per_block * nblocks;
    const uint32_t m_nch = magic_of(g.nch);
    // (the fix-up pass touches only the blocks whose nb grew past kfirst)
    auto skip_untouched = [&](uint32_t wk) {
        while (wk < total && fixup && nbuse[wk / tiles_per_block] <= kfirst) wk += gridDim.x;
        return wk;
    };
    auto tile_s0 = [&](uint32_t wk) { return (wk - (wk / tiles_per_block) * tiles_per_block) * T; };
    uint32_t work = skip_untouched(blockIdx.x);
    // The first item of a tile is loaded before the previous tile's rows are stored: vmcnt retires in issue order, so a
    // load issued behind 18 KiB of stores would wait for their write acknowledgements before the tile could start.
    ItemRegs cur, nxt, nx2;
    if (work < total) {
        const uint32_t s0f = tile_s0(work), Tnf = min(T, g.ns - s0f);
        const uint8_t* blkf = src + (size_t)(work / tiles_per_block) * g.block_bytes;
        if (tid < g.nch * ((Tnf + 15) >> 4))
            load_item<BPS, XDELTA>(blkf, g, m_nch, s0f, Tnf, (BPS == 4) && ((reinterpret_cast<uintptr_t>(blkf) & 3u) == 0), tid, cur);
    }
    while (work < total) {
    const uint32_t b = work / tiles_per_block;
    const uint32_t s0 = tile_s0(work);
    __syncthreads();  // the previous tile's rows have left LDS
    const uint32_t Tn = min(T, g.ns - s0);
    const uint8_t* blk = src + (size_t)b * g.block_bytes;
    const bool aligned4 = (BPS == 4) && ((reinterpret_cast<uintptr_t>(blk) & 3u) == 0);

    uint8_t* out = lds;
    const uint32_t RS = T + 16;  // out row stride (bytes): rows stay 16-aligned, banks rotate per row
    // Non-zero map: bit s of nzflag[hzr block] = "4 KiB segment s of the block holds a non-zero byte".
    // The first setter of a bit in the workgroup forwards it to HBM.  A tile row (<= 2048
    // samples) touches at most two hzr blocks: dedupe masks are kept as [rel][plane][channel].
    uint32_t* s_nz = reinterpret_cast<uint32_t*>(out + (size_t)kcount * g.nch * RS);
    for (uint32_t i = tid; i < 8 * g.nch; i += nthr) s_nz[i] = 0;
    __syncthreads();

    const TileCtx tc{g, m_nch, s0, Tn, b, kfirst, kcount, RS, RSPT_DIAG_ONLY(ablate), fixup, out, s_nz, nzflag};
    // ---- per (channel, 16-sample group): load, transform, plane split ---------
    const uint32_t ngrp = (Tn + 15) >> 4;
    const uint32_t nitems = g.nch * ngrp;
    uint32_t mag = 0;
    // the next item's loads are issued before the current item is transformed: the HBM round trip hides behind
    // ~250 VALU instructions instead of stalling the wave (8 waves per CU cannot hide it by themselves)
    uint32_t nz_seg = 0xFFFFFFFFu, nz_done = 0;  // segment (flat index >> 12) this thread is in, planes already flagged for it
    uint32_t q = tid;
    bool have = q < nitems;  // (item `tid` is in `cur` already)
    if (q + nthr < nitems) load_item<BPS, XDELTA>(blk, g, m_nch, s0, Tn, aligned4, q + nthr, nxt);
    while (have) {
        const uint32_t qn = q + nthr;
        const bool have_next = qn < nitems;
        // two items ahead: ~36 row-segment loads (9 KiB) in flight per wave while this item is transformed
        if (qn + nthr < nitems) load_item<BPS, XDELTA>(blk, g, m_nch, s0, Tn, aligned4, qn + nthr, nx2);
        transform_item<BPS, XDELTA>(cur, q, tc, mag, nz_seg, nz_done);
        cur = nxt;
        nxt = nx2;
        q = qn;
        have = have_next;
    }
    const uint32_t work_next = skip_untouched(work + gridDim.x);
    if (work_next < total) {
        const uint32_t s0n = tile_s0(work_next), Tnn = min(T, g.ns - s0n);
        const uint8_t* blkn = src + (size_t)(work_next / tiles_per_block) * g.block_bytes;
        if (tid < g.nch * ((Tnn + 15) >> 4))
            load_item<BPS, XDELTA>(blkn, g, m_nch, s0n, Tnn, (BPS == 4) && ((reinterpret_cast<uintptr_t>(blkn) & 3u) == 0), tid, cur);
    }
    if (XDELTA && !fixup) {
        // only the three thresholds matter (need_from_mask): fold to the top bit of each byte range, and
        // skip the same-address atomic when the block's mask already has it (750 tiles x waves per block)
        mag = wave_or_u32(mag);
        const uint32_t f = (mag >= 0x80u ? 0x80u : 0u) | (mag >= 0x8000u ? 0x8000u : 0u) | (mag >= 0x800000u ? 0x800000u : 0u);
        if (lane_id() == 0 && f && !(RSPT_DIAG_ONLY(ablate) & 131072u) && (__builtin_nontemporal_load(&needmask[b]) & f) != f) atomicOr(&needmask[b], f);
    }
    __syncthreads();

    // ---- plane rows -> HBM ------------------------------------------------------
    const uint32_t upr = ngrp;  // 16-byte units per row
    const uint32_t nunits = kcount * g.nch * upr;
    const uint32_t m_upr = magic_of(upr);
    const bool fast = ((g.ns & 15) == 0) && ((s0 & 15) == 0);
    for (uint32_t u = tid; u < nunits; u += nthr) {
        const uint32_t row = fast_div(u, upr, m_upr);  // (k-kfirst)*nch + c
        const uint32_t colu = u - row * upr;
        const uint32_t kr = fast_div(row, g.nch, m_nch);
        const uint32_t c = row - kr * g.nch;
        const uint32_t k = kfirst + kr;
        const uint32_t nbytes = min(16u, Tn - colu * 16);
        const uint8_t* sp = out + (size_t)row * RS + colu * 16;
        uint8_t* dp = planes + ((size_t)b * kMaxPlanes + k) * g.plane_stride + (size_t)c * g.ns + s0 + colu * 16;
        if (RSPT_DIAG_ONLY(ablate) & 16384u) continue;  // timing probe: no stores (diagnostic builds only)
        if (fast && nbytes == 16) {
            *reinterpret_cast<uint4*>(dp) = *reinterpret_cast<const uint4*>(sp);
        } else if (nbytes == 16) {  // a row that starts at any byte alignment (ns not a multiple of 16): unaligned 16-byte store
            const uint4 v = *reinterpret_cast<const uint4*>(sp);
            __builtin_memcpy(dp, &v, 16);
        } else {
            for (uint32_t i = 0; i < nbytes; ++i) dp[i] = sp[i];
        }
    }
    work = work_next;
    }  // tile loop
    // main pass: the last workgroup to get here runs the escalation scan over the blocks (saves a launch and its gap)
    if (ticket) {
        __shared__ uint32_t s_last, s_wmax[16];
        __syncthreads();
        if (tid == 0) {
            __threadfence();  // this workgroup's needmask atomics are out
            s_last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
        }
        __syncthreads();
        if (s_last) {
            __threadfence();
            nb_scan_body(needmask, nblocks, nb_state, nbuse_out, XDELTA ? 1 : 0, s_wmax);
        }
    }
}

// ---- streaming form of k_tile_planes (aligned int32 samples, ns a multiple of 16, 256 threads) ---------------------------
// Same tiles, same LDS rows, same stores; what differs is how the input gets here.  k_tile_planes' loads go through the
// compiler's wait-count bookkeeping, which drains the whole queue (vmcnt(0)) before every item across that kernel's
// control flow: one item's loads in flight per wave, 2.9 TB/s.  Here a thread's items form one stream across its
// workgroup's tiles, held in a ring of three register sets.  A set is refilled (item three ahead, of this tile or the
// next) straight after its item has been transformed, by loads issued from inline asm, and the consumer waits with an
// explicit count: loads return in issue order, exactly two sets (2 x 18 loads) are issued behind any set, and whatever
// the compiler adds in between (plane stores, flag atomics) is younger still -- "at most 36 outstanding" therefore
// means the set has landed, with two items' worth of loads (9 KiB per wave) still in flight, also across the store
// phase.  Every load and every wait is unconditional and sits in one place per set: the ring registers meet no
// control-flow join, so the compiler has no reason to copy a register whose load is in flight
// (tools/check_stream_regs.py checks the generated code for exactly that).  A stream that has run out of tiles keeps
// loading item 0 of block 0; lanes past a tile's item count load its last item and skip the transform.
// One sample per load for every width: int32 -> dword; int24 -> an UNALIGNED dword at the sample's first byte (the hardware
// takes it; the byte behind the sample comes along and is shifted out after the wait), int16 -> a sign-extending short, int8 -> a
// sign-extending byte (utils.cpp:186-189).
template <int BPS>
__device__ __forceinline__ uint32_t stream_load(const uint8_t* base, uint32_t off) {
    uint32_t v;
    if (BPS == 1)
        asm volatile("global_load_sbyte %0, %1, %2" : "=v"(v) : "v"(off), "s"(base));
    else if (BPS == 2)
        asm volatile("global_load_sshort %0, %1, %2" : "=v"(v) : "v"(off), "s"(base));
    else
        asm volatile("global_load_dword %0, %1, %2" : "=v"(v) : "v"(off), "s"(base));
    return v;
}
// RAGGED (ns not a multiple of 16): the last group of a block's last tile is short; its missing rows re-read the last
// valid one (the transform masks them), so that no load leaves the block.  `tn` = samples in the tile.
// int24: `lim4` = (bytes from `blk` to the end of the batch) - 4: the dword of the batch's very last sample would end one
// byte past the buffer, so it starts one byte early instead (stream_fix24 knows).
template <int BPS, bool XDELTA, bool RAGGED>
__device__ __forceinline__ void load_item_stream(const uint8_t* blk, const Geom& g, uint32_t m_nch, uint32_t s0, uint32_t tn, uint32_t lim4, uint32_t q,
                                                 ItemRegs& R) {
    const uint32_t rstride = g.nch * (uint32_t)BPS;
    const uint32_t grp = fast_div(q, g.nch, m_nch);
    const uint32_t c = q - grp * g.nch;
    const uint32_t t = s0 + (grp << 4);
    const uint32_t off = (t * g.nch + c) * (uint32_t)BPS;  // (block_bytes < 4 GiB: the host checks)
    const uint32_t last = RAGGED ? (min(16u, tn - (grp << 4)) - 1u) * rstride : 0u;
#pragma unroll
    for (uint32_t e = 0; e < 16; ++e) {
        uint32_t o = off + (RAGGED ? min(e * rstride, last) : e * rstride);
        if (BPS == 3) o = min(o, lim4);
        R.pv[e] = stream_load<BPS>(blk, o);
    }
    if (XDELTA) {
        // channel start: the flat array continues from the end of channel c-1 (flat index 0: patched by the consumer)
        const uint32_t cm = c ? c - 1 : 0u;
        uint32_t o1 = t ? off - rstride : ((g.ns - 1) * g.nch + cm) * (uint32_t)BPS;
        uint32_t o2 = t ? off - 2 * rstride : ((g.ns - 2) * g.nch + cm) * (uint32_t)BPS;
        if (BPS == 3) {
            // with one channel the (discarded) halo of flat index 0 is the block's last sample: its dword, too, must not
            // leave the batch.  A clamped halo is only ever the discarded one, so stream_fix24 need not know.
            o1 = min(o1, lim4);
            o2 = min(o2, lim4);
        }
        R.p1 = stream_load<BPS>(blk, o1);
        R.p2 = stream_load<BPS>(blk, o2);
    } else {
        R.p1 = R.p2 = 0;
    }
}
// int24: the loaded dwords -> sign-extended samples.  `tail_tile`: the tile that holds the batch's last sample.
// Big-endian samples (g.be): the loaded dword holds the sample's bytes most significant first; one v_perm_b32 puts them into
// the top three bytes in little-endian order (from the low three of the dword, or from its top three where the dword was
// loaded one byte early) and the same arithmetic shift sign-extends.
template <bool RAGGED>
__device__ __forceinline__ void stream_fix24(ItemRegs& R, const Geom& g, uint32_t m_nch, uint32_t s0, uint32_t tn, uint32_t lim4, bool tail_tile, uint32_t q) {
    if (g.be) {
        const uint32_t rstride = g.nch * 3u;
        const uint32_t grp = fast_div(q, g.nch, m_nch);
        const uint32_t c = q - grp * g.nch;
        const uint32_t off = ((s0 + (grp << 4)) * g.nch + c) * 3u;
        const uint32_t last = RAGGED ? (min(16u, tn - (grp << 4)) - 1u) * rstride : 0u;
#pragma unroll
        for (uint32_t e = 0; e < 16; ++e) {
            const uint32_t o = off + (RAGGED ? min(e * rstride, last) : e * rstride);
            const uint32_t early = tail_tile && o > lim4 ? 1u : 0u;  // (loaded one byte early: the sample is the top three bytes)
            const uint32_t t = early ? __builtin_amdgcn_perm(R.pv[e], R.pv[e], 0x01020303u) : __builtin_amdgcn_perm(R.pv[e], R.pv[e], 0x00010202u);
            R.pv[e] = (uint32_t)((int32_t)t >> 8);
        }
        R.p1 = (uint32_t)((int32_t)__builtin_amdgcn_perm(R.p1, R.p1, 0x00010202u) >> 8);
        R.p2 = (uint32_t)((int32_t)__builtin_amdgcn_perm(R.p2, R.p2, 0x00010202u) >> 8);
        return;
    }
    if (!tail_tile) {
#pragma unroll
        for (uint32_t e = 0; e < 16; ++e) R.pv[e] = (uint32_t)((int32_t)(R.pv[e] << 8) >> 8);
    } else {
        const uint32_t rstride = g.nch * 3u;
        const uint32_t grp = fast_div(q, g.nch, m_nch);
        const uint32_t c = q - grp * g.nch;
        const uint32_t off = ((s0 + (grp << 4)) * g.nch + c) * 3u;
        const uint32_t last = RAGGED ? (min(16u, tn - (grp << 4)) - 1u) * rstride : 0u;
#pragma unroll
        for (uint32_t e = 0; e < 16; ++e) {
            const uint32_t o = off + (RAGGED ? min(e * rstride, last) : e * rstride);
            // (loaded one byte early: the sample is the top three bytes)
            R.pv[e] = o > lim4 ? (uint32_t)((int32_t)R.pv[e] >> 8) : (uint32_t)((int32_t)(R.pv[e] << 8) >> 8);
        }
    }
    R.p1 = (uint32_t)((int32_t)(R.p1 << 8) >> 8);
    R.p2 = (uint32_t)((int32_t)(R.p2 << 8) >> 8);
}
// int32 / int16 big-endian samples: byte reversal of the landed registers (int16: the load sign-extended the wrong byte)
template <int BPS>
__device__ __forceinline__ void stream_swap(ItemRegs& R) {
    if (BPS == 1) return;  // (one byte has no order)
    auto sw = [](uint32_t v) -> uint32_t {
        if (BPS == 4) return __builtin_amdgcn_perm(v, v, 0x00010203u);
        return (uint32_t)((int32_t)(__builtin_amdgcn_perm(v, v, 0x00010001u) << 16) >> 16);  // bytes 0 and 1 swapped, then sign-extended from 16 bits
    };
#pragma unroll
    for (uint32_t e = 0; e < 16; ++e) R.pv[e] = sw(R.pv[e]);
    R.p1 = sw(R.p1);
    R.p2 = sw(R.p2);
}
// the set's registers pass through the wait, so nothing that reads them can be scheduled above it
template <int N>
__device__ __forceinline__ void item_wait(ItemRegs& R) {
    asm volatile("s_waitcnt vmcnt(%18)"
                 : "+v"(R.pv[0]), "+v"(R.pv[1]), "+v"(R.pv[2]), "+v"(R.pv[3]), "+v"(R.pv[4]), "+v"(R.pv[5]), "+v"(R.pv[6]), "+v"(R.pv[7]),
                   "+v"(R.pv[8]), "+v"(R.pv[9]), "+v"(R.pv[10]), "+v"(R.pv[11]), "+v"(R.pv[12]), "+v"(R.pv[13]), "+v"(R.pv[14]),
                   "+v"(R.pv[15]), "+v"(R.p1), "+v"(R.p2)
                 : "n"(N));
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int BPS, bool XDELTA, bool RAGGED>
__global__ __launch_bounds__(256) void k_tile_stream(const uint8_t* __restrict__ src, Geom g, uint32_t T, uint32_t kfirst, uint32_t kcount,
                                                    uint8_t* __restrict__ planes, uint32_t* __restrict__ needmask,
                                                    uint32_t* __restrict__ nzflag, const uint32_t* __restrict__ nbuse, uint32_t ablate,
                                                    uint32_t nblocks, uint32_t* __restrict__ ticket, uint32_t* __restrict__ nb_state,
                                                    uint32_t* __restrict__ nbuse_out, const uint32_t* __restrict__ plane_dirty, uint32_t dirty_shift) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    constexpr uint32_t nthr = 256;
    constexpr int kAhead = 2 * (XDELTA ? 18 : 16);  // hand-issued loads of the two younger sets
    const uint32_t tid = threadIdx.x;
    const bool fixup = nbuse != nullptr;
    // the fix-up pass is launched blind (the host does not know whether nb escalated in this call): the escalation scan of the main
    // pass has left the batch's final nb in nb_state -- nothing beyond the planes already written is needed: leave at once
    // (instead of every workgroup walking its tiles' nbuse entries: ~4 us per call for nothing)
    if (fixup && *nb_state <= kfirst) return;
    const uint32_t tiles_per_block = (g.ns + T - 1) / T;
    const uint32_t total = tiles_per_block * nblocks;
    const uint32_t m_nch = magic_of(g.nch);
    auto skip_untouched = [&](uint32_t wk) {  // (the fix-up pass touches only the blocks whose nb grew past kfirst)
        while (wk < total && fixup && nbuse[wk / tiles_per_block] <= kfirst) wk += gridDim.x;
        return wk;
    };
    auto tile_s0 = [&](uint32_t wk) { return (wk - (wk / tiles_per_block) * tiles_per_block) * T; };
    auto tile_items = [&](uint32_t s0_) { return g.nch * ((min(T, g.ns - s0_) + 15u) >> 4); };

    // ---- load side of the stream: tile lw, ordinal lj of its l_ipt ----
    uint32_t lw = skip_untouched(blockIdx.x), lj = 0, l_s0 = 0, l_tn = 16, l_nitems = 1, l_ipt = 0xFFFFFFFFu, l_lim4 = 0;
    auto lim4_of = [&](uint32_t bb) {  // int24: last byte offset (from block bb) at which a dword load stays inside the batch
        const unsigned long long rest = (unsigned long long)(nblocks - bb) * g.block_bytes - 4ull;
        return (uint32_t)(rest > 0xFFFFFFFFull ? 0xFFFFFFFFull : rest);
    };
    const uint8_t* l_blk = src;
    auto l_open = [&]() {
        if (lw < total) {
            l_s0 = tile_s0(lw);
            l_tn = min(T, g.ns - l_s0);
            l_nitems = tile_items(l_s0);
            l_ipt = (l_nitems + nthr - 1) / nthr;
            l_blk = src + (size_t)(lw / tiles_per_block) * g.block_bytes;
            l_lim4 = lim4_of(lw / tiles_per_block);
        } else {  // out of tiles: keep the ring turning on item 0 of block 0
            l_s0 = 0;
            l_tn = 16;
            l_nitems = 1;
            l_ipt = 0xFFFFFFFFu;
            l_blk = src;
            l_lim4 = lim4_of(0);
        }
    };
    auto fetch = [&](ItemRegs& R) __attribute__((always_inline)) {
        load_item_stream<BPS, XDELTA, RAGGED>(l_blk, g, m_nch, l_s0, l_tn, l_lim4, min(tid + lj * nthr, l_nitems - 1), R);
        if (++lj == l_ipt) {
            lw = skip_untouched(lw + gridDim.x);
            lj = 0;
            l_open();
        }
    };

    // ---- transform side: tile tw, ordinal tj of its t_ipt ----
    uint32_t tw = lw, tj = 0, t_ipt = 0, t_nitems = 0;
    // dirty bits of the NEXT tile to open, [plane - kfirst]: scalar loads issued by hand one tile ahead (they do not queue
    // behind the ring's vector loads, and their latency hides behind the store phase)
    u32x4 dq0 = {0, 0, 0, 0}, dq1 = dq0, dq2 = dq0, dq3 = dq0;
    auto dirty_fetch = [&](uint32_t wk) {
        const uint32_t bb = wk < total ? wk / tiles_per_block : 0u;
        auto at = [&](uint32_t kr) {
            const uint64_t a = reinterpret_cast<uint64_t>(plane_dirty + ((size_t)bb * kMaxPlanes + min(kfirst + kr, kMaxPlanes - 1u)) * 4);
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
            return reinterpret_cast<const uint32_t*>(((uint64_t)hi << 32) | lo);
        };
        asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=&s"(dq0) : "s"(at(0)));
        asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=&s"(dq1) : "s"(at(1)));
        asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=&s"(dq2) : "s"(at(2)));
        asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=&s"(dq3) : "s"(at(3)));
    };
    uint32_t mag = 0, nz_seg = 0xFFFFFFFFu, nz_done = 0;
    const uint32_t RS = T + 16;  // out row stride (bytes): rows stay 16-aligned, banks rotate per row
    uint8_t* out = lds;
    uint32_t* s_nz = reinterpret_cast<uint32_t*>(out + (size_t)kcount * g.nch * RS);
    TileCtx tc{g, m_nch, 0, 0, 0, kfirst, kcount, RS, RSPT_DIAG_ONLY(ablate), fixup, out, s_nz, nzflag, planes, dirty_shift};
    auto t_open = [&]() {
        tc.b = tw / tiles_per_block;
        tc.s0 = tile_s0(tw);
        tc.Tn = min(T, g.ns - tc.s0);
        t_nitems = tile_items(tc.s0);
        t_ipt = (t_nitems + nthr - 1) / nthr;
        tj = 0;
        mag = 0;
        nz_seg = 0xFFFFFFFFu;
        nz_done = 0;
        __syncthreads();  // the previous tile's rows have left LDS
        for (uint32_t i = tid; i < 8 * g.nch + 1; i += nthr) s_nz[i] = 0;  // (+1: the tile's escalation bits)
        // which hzr blocks of this block's planes hold stale data: 128 bits per plane, fetched by dirty_fetch() a tile ago
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(dq0), "+s"(dq1), "+s"(dq2), "+s"(dq3));
        if (tid == 0) {
            uint32_t* sd = s_nz + 8 * g.nch + 1;
            const u32x4 dq[4] = {dq0, dq1, dq2, dq3};
#pragma unroll
            for (uint32_t i = 0; i < 16; ++i) sd[i] = dq[i >> 2][i & 3];
        }
        __syncthreads();
    };
    auto t_close = [&]() {
        // Escalation magnitude: only the three thresholds matter (need_from_mask).  The workgroup folds them in LDS and
        // sends one fire-and-forget atomic per tile -- a read of needmask[b] to skip redundant ones would be a
        // compiler-tracked load queued behind the ring's prefetch, i.e. a full drain of the stream once per tile.
        if (XDELTA && !fixup) {
            mag = wave_or_u32(mag);
            const uint32_t f = (mag >= 0x80u ? 0x80u : 0u) | (mag >= 0x8000u ? 0x8000u : 0u) | (mag >= 0x800000u ? 0x800000u : 0u);
            if (lane_id() == 0 && f) atomicOr(&s_nz[8 * g.nch], f);
        }
        __syncthreads();
        if (XDELTA && !fixup && tid == 0) {
            const uint32_t f = s_nz[8 * g.nch];
            if (f && !(RSPT_DIAG_ONLY(ablate) & 131072u)) atomicOr(&needmask[tc.b], f);
        }
        // plane rows -> HBM: 16-byte units, T contiguous bytes per (plane, channel)
        const uint32_t upr = (tc.Tn + 15u) >> 4;
        const uint32_t nunits = kcount * g.nch * upr;
        const bool whole_lines = (upr & 7u) == 0 && (tc.Tn & 15u) == 0 && !(RSPT_DIAG_ONLY(ablate) & (1u << 21));  // rows are whole 128-byte lines (else: store everything)
        const uint32_t m_upr = magic_of(upr);
        for (uint32_t u = tid; u < nunits; u += nthr) {
            const uint32_t row = fast_div(u, upr, m_upr);  // (k-kfirst)*nch + c
            const uint32_t colu = u - row * upr;
            const uint32_t kr = fast_div(row, g.nch, m_nch);
            const uint32_t c = row - kr * g.nch;
            const uint8_t* sp = out + (size_t)row * RS + colu * 16;
            uint8_t* dp = planes + ((size_t)tc.b * kMaxPlanes + kfirst + kr) * g.plane_stride + (size_t)c * g.ns + tc.s0 + colu * 16;
            if (RSPT_DIAG_ONLY(ablate) & (16384u | (1u << 26))) continue;  // timing probes: no stores / stores straight from the registers (diagnostic builds only)
            if ((RSPT_DIAG_ONLY(ablate) & (1u << 27)) && kr >= 1) continue;
            const uint4 v = *reinterpret_cast<const uint4*>(sp);
            // a clean hzr block (zeros everywhere, see rspt_hip_packer::plane_dirty) only takes the 128-byte lines that hold
            // a non-zero byte: a line is eight consecutive units = eight aligned lanes (T is a multiple of 128)
            const unsigned long long bal = __ballot((v.x | v.y | v.z | v.w) != 0);
            const bool line_nz = ((bal >> (lane_id() & ~7u)) & 0xFFull) != 0;
            if (!line_nz && whole_lines) {
                const uint32_t bucket = ((c * g.ns + tc.s0 + colu * 16) >> 16) >> dirty_shift;
                if (!((s_nz[8 * g.nch + 1 + kr * 4 + (bucket >> 5)] >> (bucket & 31u)) & 1u)) continue;
            }
            if (!RAGGED) {
                *reinterpret_cast<uint4*>(dp) = v;
            } else {
                // rows start at c * ns: any byte alignment (the hardware takes unaligned 16-byte stores); the row's last unit may be short
                const uint32_t nbytes = min(16u, tc.Tn - colu * 16);
                if (nbytes == 16) {
                    __builtin_memcpy(dp, &v, 16);
                } else {
                    for (uint32_t i = 0; i < nbytes; ++i) dp[i] = sp[i];
                }
            }
        }
    };
    // one turn of the ring for set R; true when the stream's last tile has been stored
    auto turn = [&](ItemRegs& R) __attribute__((always_inline)) -> bool {
        item_wait<kAhead>(R);
        // (Nothing of a dirty_fetch is in flight here -- t_open waited for it when the tile was opened.  Said once more, where
        //  it costs nothing, so that it holds on every path of the control-flow graph and not only on the feasible ones:
        //  tools/check_stream_regs.py.  In front of the tile switch below the same wait cost 7 us per launch.)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(dq0), "+s"(dq1), "+s"(dq2), "+s"(dq3)::"memory");
        const uint32_t q = tid + tj * nthr;
        if (q < t_nitems) {
            if (BPS == 3) stream_fix24<RAGGED>(R, g, m_nch, tc.s0, tc.Tn, lim4_of(tc.b), tc.b + 1 == nblocks && tc.s0 + tc.Tn == g.ns, q);
            if (BPS != 3 && g.be) stream_swap<BPS>(R);  // (wave-uniform; the registers have landed: nothing of this set is in flight here)
#ifdef RSPT_DIAG
            transform_item<BPS, XDELTA, (BPS == 4 && XDELTA && !RAGGED)>(R, q, tc, mag, nz_seg, nz_done);
#else
            transform_item<BPS, XDELTA>(R, q, tc, mag, nz_seg, nz_done);
#endif
        }
        fetch(R);  // (ahead of this tile's stores: a load queued behind them would wait for their acknowledgements)
        if (++tj == t_ipt) {
            const uint32_t tw_next = skip_untouched(tw + gridDim.x);
            dirty_fetch(tw_next);  // (consumed by t_open, after this tile's store phase)
            t_close();
            tw = tw_next;
            if (tw >= total) return true;
            t_open();
        }
        return false;
    };

    if (tw < total) {
        l_open();
        ItemRegs ra, rb, rc;
        fetch(ra);
        fetch(rb);
        fetch(rc);
        dirty_fetch(tw);
        t_open();
        for (;;) {
            if (turn(ra)) break;
            if (turn(rb)) break;
            if (turn(rc)) break;
        }
        // The ring's last refills are still in flight: let them land.  All three register sets pass through the wait, so the
        // allocator cannot hand any of their registers to something else between the loop exit and this point (a value
        // parked there would be overwritten by the late load: the compiler does not know these loads exist).
        item_wait<0>(ra);
        item_wait<0>(rb);
        item_wait<0>(rc);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(dq0), "+s"(dq1), "+s"(dq2), "+s"(dq3));  // (the dirty bits fetched for a tile that never opens)
        asm volatile("" ::: "memory");
    }
    // main pass: the last workgroup to get here runs the escalation scan over the blocks (saves a launch and its gap)
    if (ticket) {
        __shared__ uint32_t s_last, s_wmax[16];
        __syncthreads();
        if (tid == 0) {
            __threadfence();  // this workgroup's needmask atomics are out
            s_last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
        }
        __syncthreads();
        if (s_last) {
            __threadfence();
            nb_scan_body(needmask, nblocks, nb_state, nbuse_out, XDELTA ? 1 : 0, s_wmax);
        }
    }
}

// planar int32 [nch][ns] (the output of a transform kernel) -> planes, with the
// optional flat xdelta stage (dct: signal_packer_dct.cpp:117-119; hadamard: none).
// Element i only needs p[i-1], p[i-2]: no transposition, one thread per 16 elements.
// needmask (optional; the wide-block front end of the xdelta packer): the magnitudes of the values, folded to the three
// thresholds that decide the number of planes (need_from_mask)
template <bool XDELTA>
__global__ __launch_bounds__(256) void k_planar_planes(const int32_t* __restrict__ planar, Geom g, uint32_t nplanes,
                                                       uint8_t* __restrict__ planes, uint32_t* __restrict__ nzflag, uint32_t* __restrict__ needmask) {
    const uint32_t b = blockIdx.y;
    // (sign-extended from the sample width first: what lies above it never reaches the decoded samples; transform_item does the same)
    const uint32_t sx = 32u - 8u * g.bps;
    auto magnitude = [&](uint32_t v) -> uint32_t {
        const int32_t x = (int32_t)(v << sx) >> sx;
        return (uint32_t)(x ^ (x >> 31));
    };
    auto publish_mag = [&](uint32_t mag) {
        mag = wave_or_u32(mag);
        const uint32_t f = (mag >= 0x80u ? 0x80u : 0u) | (mag >= 0x8000u ? 0x8000u : 0u) | (mag >= 0x800000u ? 0x800000u : 0u);
        if (f && lane_id() == (uint32_t)__builtin_ctzll(__ballot(1))) atomicOr(&needmask[b], f);
    };
    {
        // Fully coalesced form.  A wave takes 1024 consecutive elements (one KiB of every plane); in each of four rounds
        // lane l loads elements [256 q + 4 l, +4) as 16 bytes and stores one dword per plane.  XDELTA: element i needs
        // p[i-1], p[i-2] -- the previous lane's last two values (DPP shift), for lane 0 the previous round's lane 63 or the
        // two elements in front of the wave.
        const uint32_t wbase = (blockIdx.x * 256 + (threadIdx.x & ~63u)) * 16;  // first element of this wave
        const int32_t* pw_ = planar + (size_t)b * g.N;
        if (wbase + 1024 <= g.N && (reinterpret_cast<uintptr_t>(pw_ + wbase) & 15u) == 0) {
            const uint32_t l = threadIdx.x & 63u;
            uint4 v[4];
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) v[q] = reinterpret_cast<const uint4*>(pw_ + wbase)[q * 64 + l];
            uint32_t front1 = 0, front2 = 0;  // p[wbase-1], p[wbase-2] (the flat array starts from zeros)
            if (XDELTA && wbase) {
                front1 = (uint32_t)pw_[wbase - 1];
                front2 = (uint32_t)pw_[wbase - 2];
            }
            uint32_t nzk[4] = {0, 0, 0, 0};
            uint32_t mag = 0;
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) {
                uint32_t a0 = v[q].x, a1 = v[q].y, a2 = v[q].z, a3 = v[q].w;
                if (XDELTA) {
                    uint32_t pm1 = dpp<0x138>(0u, a3), pm2 = dpp<0x138>(0u, a2);  // wave_shr:1
                    if (l == 0) {
                        pm1 = q ? read_lane(v[q ? q - 1 : 0].w, 63) : front1;
                        pm2 = q ? read_lane(v[q ? q - 1 : 0].z, 63) : front2;
                    }
                    const bool first = (wbase | q | l) == 0;  // flat index 0: delta_encode and xor_encode_32 start from 0
                    const uint32_t om1 = first ? 0u : pm1 - pm2 - 128u;
                    const uint32_t o0 = a0 - (first ? 0u : pm1) - 128u, o1 = a1 - a0 - 128u, o2 = a2 - a1 - 128u, o3 = a3 - a2 - 128u;
                    a0 = o0 ^ om1;
                    a1 = o1 ^ o0;
                    a2 = o2 ^ o1;
                    a3 = o3 ^ o2;
                }
                if (needmask)
                    mag |= magnitude(a0) | magnitude(a1) | magnitude(a2) | magnitude(a3);
                const uint32_t lo01 = __builtin_amdgcn_perm(a1, a0, 0x05010400u), hi01 = __builtin_amdgcn_perm(a1, a0, 0x07030602u);
                const uint32_t lo23 = __builtin_amdgcn_perm(a3, a2, 0x05010400u), hi23 = __builtin_amdgcn_perm(a3, a2, 0x07030602u);
                const uint32_t pl[4] = {__builtin_amdgcn_perm(lo23, lo01, 0x05040100u), __builtin_amdgcn_perm(lo23, lo01, 0x07060302u),
                                        __builtin_amdgcn_perm(hi23, hi01, 0x05040100u), __builtin_amdgcn_perm(hi23, hi01, 0x07060302u)};
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k) {
                    if (k < nplanes) {
                        reinterpret_cast<uint32_t*>(planes + ((size_t)b * kMaxPlanes + k) * g.plane_stride + wbase)[q * 64 + l] = pl[k];
                        nzk[k] |= pl[k];
                    }
                }
            }
            for (uint32_t k = 0; k < nplanes; ++k) {
                if (__ballot(nzk[k] != 0) && l == 0) atomicOr(&nzflag[hb_index(g, b, k, wbase >> 16)], 1u << ((wbase >> 12) & 15u));
            }
            if (needmask) publish_mag(mag);
            return;
        }
    }
    const uint32_t i0 = (blockIdx.x * 256 + threadIdx.x) * 16;
    // (a wave's 1024 elements never straddle a 64 KiB hzr block.)  Lanes past the block's end stay in the wave with nothing to do
    // -- the magnitude reduction at the end is a cross-lane one, and lanes that have left would hand it stale registers
    if (!__ballot(i0 < g.N)) return;
    const int32_t* p = planar + (size_t)b * g.N;
    const uint32_t cnt = i0 < g.N ? min(16u, g.N - i0) : 0u;
    uint32_t p1 = 0, oprev = 0;
    if (XDELTA && i0 >= 1 && cnt) {
        p1 = (uint32_t)p[i0 - 1];
        uint32_t p2 = i0 >= 2 ? (uint32_t)p[i0 - 2] : 0u;
        oprev = p1 - p2 - 128u;
    }
    uint32_t pw[4][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    uint32_t xs[16];
    uint32_t mag = 0;
    if (cnt == 16 && (reinterpret_cast<uintptr_t>(p + i0) & 15u) == 0) {  // four 16-byte loads instead of sixteen dword loads at a 64-byte lane stride
        const uint4* p4 = reinterpret_cast<const uint4*>(p + i0);
#pragma unroll
        for (uint32_t q4 = 0; q4 < 4; ++q4) {
            const uint4 v4 = p4[q4];
            xs[4 * q4] = v4.x;
            xs[4 * q4 + 1] = v4.y;
            xs[4 * q4 + 2] = v4.z;
            xs[4 * q4 + 3] = v4.w;
        }
    } else {
#pragma unroll
        for (uint32_t e = 0; e < 16; ++e) xs[e] = e < cnt ? (uint32_t)p[i0 + e] : 0u;
    }
#pragma unroll
    for (uint32_t e = 0; e < 16; ++e) {
        if (e < cnt) {
            uint32_t x = xs[e];
            uint32_t v = x;
            if (XDELTA) {
                uint32_t o = x - p1 - 128u;
                v = o ^ oprev;
                oprev = o;
                p1 = x;
            }
            mag |= magnitude(v);
            const uint32_t sh = (e & 3) * 8;
            pw[0][e >> 2] |= (v & 0xFFu) << sh;
            pw[1][e >> 2] |= ((v >> 8) & 0xFFu) << sh;
            pw[2][e >> 2] |= ((v >> 16) & 0xFFu) << sh;
            pw[3][e >> 2] |= (v >> 24) << sh;
        }
    }
    for (uint32_t k = 0; k < nplanes; ++k) {
        const bool nz = (pw[k][0] | pw[k][1] | pw[k][2] | pw[k][3]) != 0;
        const unsigned long long any = __ballot(nz);
        if (any && lane_id() == (uint32_t)__builtin_ctzll(__ballot(1))) atomicOr(&nzflag[hb_index(g, b, k, i0 >> 16)], 1u << ((i0 >> 12) & 15u));
        uint8_t* dp = planes + ((size_t)b * kMaxPlanes + k) * g.plane_stride + i0;
        if (cnt == 16) {
            *reinterpret_cast<uint4*>(dp) = make_uint4(pw[k][0], pw[k][1], pw[k][2], pw[k][3]);
        } else {
            for (uint32_t i = 0; i < cnt; ++i) dp[i] = (uint8_t)(pw[k][i >> 2] >> ((i & 3) * 8));
        }
    }
    if (needmask) publish_mag(mag);
}

// interleaved native -> planar int32 [nch][ns] (front end of the transform packers)
template <int BPS>
__global__ __launch_bounds__(256) void k_tile_planar(const uint8_t* __restrict__ src, Geom g, uint32_t T, int32_t* __restrict__ planar) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t tid = threadIdx.x;
    const uint32_t b = blockIdx.y;
    const uint32_t s0 = blockIdx.x * T;
    const uint32_t Tn = min(T, g.ns - s0);
    const uint8_t* blk = src + (size_t)b * g.block_bytes;
    const uint32_t rowb = g.nch * BPS;
    const size_t gs = (size_t)s0 * rowb;
    const uintptr_t abs_s = reinterpret_cast<uintptr_t>(blk) + gs;
    const uint32_t lo = (uint32_t)(abs_s & 15);
    const uint8_t* abase = reinterpret_cast<const uint8_t*>(abs_s - lo);
    const uint32_t span = lo + Tn * rowb;
    for (uint32_t o = tid * 16; o < span; o += 256 * 16) *reinterpret_cast<uint4*>(lds + o) = *reinterpret_cast<const uint4*>(abase + o);
    __syncthreads();
    const bool aligned4 = (BPS == 4) && ((lo & 3) == 0);
    const uint8_t* tile = lds + lo;
    // thread <-> (channel row, sample): consecutive lanes write consecutive samples of one channel
    const uint32_t total = g.nch * Tn;
    for (uint32_t q = tid; q < total; q += 256) {
        const uint32_t c = q / Tn, t = q - c * Tn;
        planar[(size_t)b * g.N + (size_t)c * g.ns + s0 + t] = sample_from_bytes<BPS>(tile + ((size_t)t * g.nch + c) * BPS, aligned4, g.be != 0);
    }
}

// The same for blocks too WIDE for that (a 16-sample tile of all channels no longer fits the LDS: more than ~1000 channels): a
// plain 64 x 64 transpose per workgroup, any sample width and byte order.  The front end of every packer for such shapes -- the
// xdelta / hzr packers go on with k_planar_planes over the planar block (rspt_hip.hip: wide); a fallback, not a fast path.
template <int BPS>
__global__ __launch_bounds__(256) void k_wide_planar(const uint8_t* __restrict__ src, Geom g, int32_t* __restrict__ planar) {
    __shared__ int32_t tile[64][65];
    const uint32_t tid = threadIdx.x, b = blockIdx.z;
    const uint32_t s0 = blockIdx.x * 64u, c0 = blockIdx.y * 64u;
    const uint8_t* blk = src + (size_t)b * g.block_bytes;
    for (uint32_t q = tid; q < 4096u; q += 256u) {
        const uint32_t t = q >> 6, c = q & 63u;  // consecutive lanes: consecutive channels of one sample row
        if (s0 + t < g.ns && c0 + c < g.nch) tile[c][t] = sample_from_bytes<BPS>(blk + ((size_t)(s0 + t) * g.nch + c0 + c) * BPS, false, g.be != 0);
    }
    __syncthreads();
    for (uint32_t q = tid; q < 4096u; q += 256u) {
        const uint32_t c = q >> 6, t = q & 63u;  // consecutive lanes: consecutive samples of one channel
        if (s0 + t < g.ns && c0 + c < g.nch) planar[(size_t)b * g.N + (size_t)(c0 + c) * g.ns + s0 + t] = tile[c][t];
    }
}
template __global__ void k_wide_planar<1>(const uint8_t*, Geom, int32_t*);
template __global__ void k_wide_planar<2>(const uint8_t*, Geom, int32_t*);
template __global__ void k_wide_planar<3>(const uint8_t*, Geom, int32_t*);
template __global__ void k_wide_planar<4>(const uint8_t*, Geom, int32_t*);

// The common shape of the same conversion -- int32 samples, nch % 4 == 0, ns % 4 == 0, 16-byte aligned input -- with 16-byte
// global accesses on both sides and no division per element (the mirror image of decode.hip: k_planar_native_i32x4).
// row_sum (optional): [block][channel] int64 sums of the samples, accumulated tile by tile (the dct's channel means without
// a second pass over the block)
__global__ __launch_bounds__(256) void k_tile_planar_i32x4(const uint8_t* __restrict__ src, Geom g, uint32_t T4, int32_t* __restrict__ planar,
                                                          long long* __restrict__ row_sum) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    int32_t* tile = reinterpret_cast<int32_t*>(lds);  // [nch][T4+1]
    const uint32_t tid = threadIdx.x, b = blockIdx.y;
    const uint32_t s0 = blockIdx.x * T4;
    const uint32_t Tn = min(T4, g.ns - s0);  // a multiple of 4
    const uint32_t RS = T4 + 1;
    {
        const uint32_t cpr = g.nch >> 2;  // 16-byte pieces per sample row
        const uint32_t step_t = 256u / cpr, step_c = 256u - step_t * cpr;
        uint32_t t = tid / cpr, c4 = tid - t * cpr;
        const int4* in = reinterpret_cast<const int4*>(src + (size_t)b * g.block_bytes + (size_t)s0 * g.nch * 4u);
        for (uint32_t u = tid; u < Tn * cpr; u += 256) {
            int4 v = in[u];
            if (g.be) {  // big-endian samples: the four bytes of every int32 reversed on the way in
                v.x = (int)__builtin_amdgcn_perm((uint32_t)v.x, (uint32_t)v.x, 0x00010203u);
                v.y = (int)__builtin_amdgcn_perm((uint32_t)v.y, (uint32_t)v.y, 0x00010203u);
                v.z = (int)__builtin_amdgcn_perm((uint32_t)v.z, (uint32_t)v.z, 0x00010203u);
                v.w = (int)__builtin_amdgcn_perm((uint32_t)v.w, (uint32_t)v.w, 0x00010203u);
            }
            int32_t* r = tile + (4u * c4) * RS + t;
            r[0] = v.x;
            r[RS] = v.y;
            r[2 * RS] = v.z;
            r[3 * RS] = v.w;
            c4 += step_c;
            const uint32_t carry = c4 >= cpr ? 1u : 0u;
            c4 -= carry ? cpr : 0u;
            t += step_t + carry;
        }
    }
    __syncthreads();
    if (row_sum) {  // four threads per channel row, one 64-bit atomic per (tile, channel)
        for (uint32_t c = tid >> 2; c < g.nch; c += 64) {
            long long sm = 0;
            for (uint32_t t = tid & 3u; t < Tn; t += 4) sm += tile[c * RS + t];
            sm += __shfl_xor(sm, 1);
            sm += __shfl_xor(sm, 2);
            if ((tid & 3u) == 0) atomicAdd(reinterpret_cast<unsigned long long*>(row_sum + (size_t)b * g.nch + c), (unsigned long long)sm);
        }
    }
    {
        const uint32_t qpr = Tn >> 2;  // 16-byte pieces per channel row
        const uint32_t step_c = 256u / qpr, step_t = 256u - step_c * qpr;
        uint32_t c = tid / qpr, t4 = tid - c * qpr;
        for (uint32_t u = tid; u < g.nch * qpr; u += 256) {
            const int32_t* r = tile + c * RS + 4u * t4;
            *reinterpret_cast<int4*>(planar + (size_t)b * g.N + (size_t)c * g.ns + s0 + 4u * t4) = make_int4(r[0], r[1], r[2], r[3]);
            t4 += step_t;
            const uint32_t carry = t4 >= qpr ? 1u : 0u;
            t4 -= carry ? qpr : 0u;
            c += step_c + carry;
        }
    }
}

// ---------------------------------------------------------------------------
// nb bookkeeping (signal_packer_xdelta_hzr.cpp:63-69): nb used by block b =
// max(nb carried in, need(0..b)); the last value is carried to the next call.
// ---------------------------------------------------------------------------
// single workgroup; nblocks arbitrary (the xdelta / hzr front ends run the same scan in their last workgroup)
__global__ __launch_bounds__(1024) void k_nb_scan(const uint32_t* __restrict__ needmask, uint32_t nblocks, uint32_t* __restrict__ nb_state,
                                                  uint32_t* __restrict__ nbuse, int use_mask) {
    __shared__ uint32_t wmax[16];
    nb_scan_body(needmask, nblocks, nb_state, nbuse, use_mask, wmax);
}

// explicit instantiations used by rspt_hip.cpp
#define INST_TILE(BPS)                                                                                                   \
    template __global__ void k_tile_planes<BPS, true>(const uint8_t*, Geom, uint32_t, uint32_t, uint32_t, uint8_t*, uint32_t*, uint32_t*, const uint32_t*, uint32_t, uint32_t, uint32_t*, uint32_t*, uint32_t*);    \
    template __global__ void k_tile_planes<BPS, false>(const uint8_t*, Geom, uint32_t, uint32_t, uint32_t, uint8_t*, uint32_t*, uint32_t*, const uint32_t*, uint32_t, uint32_t, uint32_t*, uint32_t*, uint32_t*);   \
    template __global__ void k_tile_planar<BPS>(const uint8_t*, Geom, uint32_t, int32_t*);
INST_TILE(1)
INST_TILE(2)
INST_TILE(3)
INST_TILE(4)
template __global__ void k_tile_stream<1, true, false>(const uint8_t*, Geom, uint32_t, uint32_t, uint32_t, uint8_t*, uint32_t*, uint32_t*, const uint32_t*, uint32_t, uint32_t, uint32_t*, uint32_t*, uint32_t*, const uint32_t*, uint32_t);
template __global__ void k_tile_stream<1, true, true>(const uint8_t*, Geom, uint32_t, uint32_t, uint32_t, uint8_t*, uint32_t*, uint32_t*, const uint32_t*, uint32_t, uint32_t, uint32_t*, uint32_t*, uint32_t*, const uint32_t*, uint32_t);
template __global__ void k_tile_stream<1, false, false>(const uint8_t*, Geom, uint32_t, uint32_t, uint32_t, uint8_t*, uint32_t*, uint32_t*, const uint32_t*, uint32_t, uint32_t, uint32_t*, uint32_t*, uint32_t*, const uint32_t*, uint32_t);
template __global__ void k_tile_stream<1, false, true>(const uint8_t*, Geom, uint32_t, uint32_t, uint32_t, uint8_t*, uint32_t*, uint32_t*, const uint32_t*, uint32_t, uint32_t, uint32_t*, uint32_t*, uint32_t*, const uint32_t*, uint32_t);
template __global__ void k_tile_stream<2, true, false>(const uint8_t*, Geom, uint32_t, uint32_t, uint32_t, uint8_t*, uint32_t*, uint32_t*, const uint32_t*, uint32_t, uint32_t, uint32_t*, uint32_t*, uint32_t*, const uint32_t*, uint32_t);
template __global__ void k_tile_stream<3, true, false>(const uint8_t*, Geom, uint32_t, uint32_t, uint32_t, uint8_t*, uint32_t*, uint32_t*, const uint32_t*, uint32_t, uint32_t, uint32_t*, uint32_t*, uint32_t*, const uint32_t*, uint32_t);
template __global__ void k_tile_stream<4, true, false>(const uint8_t*, Geom, uint32_t, uint32_t, uint32_t, uint8_t*, uint32_t*, uint32_t*, const uint32_t*, uint32_t, uint32_t, uint32_t*, uint32_t*, uint32_t*, const uint32_t*, uint32_t);
template __global__ void k_tile_stream<2, true, true>(const uint8_t*, Geom, uint32_t, uint32_t, uint32_t, uint8_t*, uint32_t*, uint32_t*, const uint32_t*, uint32_t, uint32_t, uint32_t*, uint32_t*, uint32_t*, const uint32_t*, uint32_t);
template __global__ void k_tile_stream<3, true, true>(const uint8_t*, Geom, uint32_t, uint32_t, uint32_t, uint8_t*, uint32_t*, uint32_t*, const uint32_t*, uint32_t, uint32_t, uint32_t*, uint32_t*, uint32_t*, const uint32_t*, uint32_t);
template __global__ void k_tile_stream<4, true, true>(const uint8_t*, Geom, uint32_t, uint32_t, uint32_t, uint8_t*, uint32_t*, uint32_t*, const uint32_t*, uint32_t, uint32_t, uint32_t*, uint32_t*, uint32_t*, const uint32_t*, uint32_t);
template __global__ void k_tile_stream<2, false, false>(const uint8_t*, Geom, uint32_t, uint32_t, uint32_t, uint8_t*, uint32_t*, uint32_t*, const uint32_t*, uint32_t, uint32_t, uint32_t*, uint32_t*, uint32_t*, const uint32_t*, uint32_t);
template __global__ void k_tile_stream<3, false, false>(const uint8_t*, Geom, uint32_t, uint32_t, uint32_t, uint8_t*, uint32_t*, uint32_t*, const uint32_t*, uint32_t, uint32_t, uint32_t*, uint32_t*, uint32_t*, const uint32_t*, uint32_t);
template __global__ void k_tile_stream<4, false, false>(const uint8_t*, Geom, uint32_t, uint32_t, uint32_t, uint8_t*, uint32_t*, uint32_t*, const uint32_t*, uint32_t, uint32_t, uint32_t*, uint32_t*, uint32_t*, const uint32_t*, uint32_t);
template __global__ void k_tile_stream<2, false, true>(const uint8_t*, Geom, uint32_t, uint32_t, uint32_t, uint8_t*, uint32_t*, uint32_t*, const uint32_t*, uint32_t, uint32_t, uint32_t*, uint32_t*, uint32_t*, const uint32_t*, uint32_t);
template __global__ void k_tile_stream<3, false, true>(const uint8_t*, Geom, uint32_t, uint32_t, uint32_t, uint8_t*, uint32_t*, uint32_t*, const uint32_t*, uint32_t, uint32_t, uint32_t*, uint32_t*, uint32_t*, const uint32_t*, uint32_t);
template __global__ void k_tile_stream<4, false, true>(const uint8_t*, Geom, uint32_t, uint32_t, uint32_t, uint8_t*, uint32_t*, uint32_t*, const uint32_t*, uint32_t, uint32_t, uint32_t*, uint32_t*, uint32_t*, const uint32_t*, uint32_t);
template __global__ void k_planar_planes<true>(const int32_t*, Geom, uint32_t, uint8_t*, uint32_t*, uint32_t*);
template __global__ void k_planar_planes<false>(const int32_t*, Geom, uint32_t, uint8_t*, uint32_t*, uint32_t*);

}  // namespace rspt
