// hzr_kernels.hip -- the hzr block codec (RLE of zero runs + per-block Huffman,
// LSB-first bit stream, CRC-32C) as gfx950 kernels.  Bit-exact with
// lib_hzr/hzr_encode.c; the parallel formulations are modelled and checked on
// the CPU in tools/kernel_model.py + tests/test_kernel_model.py.
//
//   k_hist    persistent 1024-thread workgroups, one hzr block at a time: zero-run tokenizer
//             (hzr_encode.c:133-173) on 16-byte granules; per-wave 261-bin histograms -> block histogram,
//             per-segment histograms and the zero-run context of every granule (for k_tree / k_encode)
//   k_tree    one wave per hzr block: Fill test (:285-305), Huffman tree with
//             the reference's tie-break (:222-283), codes + pre-order tree
//             description (:177-219), exact payload size -> block mode (:377-469),
//             stream bit at which each 4 KiB segment's tokens start
//   k_layout  one workgroup per block: sizes -> stream offsets, stream framing
//             (signal_packer_base.cpp:69-95, hzr_encode.c:521-522), work queues
//   k_encode  persistent 1024-thread workgroups, one big hzr block at a time: one pass from
//             lookup to an LDS image of the payload (:410-457), parallel CRC-32C
//             (hzr_crc32c.c:77-84), block header (:475-481), coalesced copy-out
//   k_encode_small   one wave per small hzr block (few non-zero segments, tokens, payload bytes)
//   k_pack_*  the streams of a batch as one container (rspt_hip_pack_batch_dev)
#include "common.hpp"

namespace rspt {

// ===========================================================================
// shared: load the 4 granules a lane owns and chain the zero runs across the
// workgroup.  Wave w owns bytes [4096w, 4096w+4096) of the hzr block as 4 rows
// of 1 KiB; lane l of row r owns the granule at 4096w + 1024r + 16l, so every
// global load is a fully coalesced 1 KiB wave access.
// ===========================================================================
// threadIdx.x behind an empty asm: inside the persistent loops this keeps everything derived
// from the thread id (lane masks, offsets) loop-VARIANT, so hipcc recomputes it per block instead
// of hoisting it out of the loop and holding it in VGPRs across the register-tight emit pass
// (k_hist: 20 -> 0 bytes of scratch, k_encode: 188 -> ~100).
__device__ __forceinline__ uint32_t thread_id() {
    uint32_t t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}

// byte i (dynamic) of a granule held in four registers
__device__ __forceinline__ uint32_t granule_byte_dyn(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t i) {
    const uint32_t lo = (i & 8u) ? w2 : w0, hi = (i & 8u) ? w3 : w1;
    return __builtin_amdgcn_perm(hi, lo, 0x0C0C0C00u | (i & 7u));  // byte (i & 7) of {hi,lo}, zero-extended
}

// "Small" hzr blocks -- few non-zero 4 KiB segments -- are walked by ONE wave, row by row, instead of
// by a 1024-thread workgroup: their histogram is taken inside k_tree (k_hist skips them) and, if they
// also have few tokens and a small payload, they are encoded by k_encode_small.
// Clean-block invariant (rspt_hip_packer::plane_dirty): a Huffman block with at most this many tokens has its non-zero
// granules wiped by the encoder that read it; k_layout flags every other block that holds a non-zero byte as dirty.
// (Every small block qualifies: kSmallTokens <= kWipeTokens.)
constexpr uint32_t kWipeTokens = 2048;
__device__ __forceinline__ bool block_is_wiped(const BlockMeta& m) { return m.mode == kModeHuff && m.fill <= kWipeTokens; }
constexpr uint32_t kSmallSegments = 2;    // non-zero 4 KiB segments (each costs one dependent HBM round trip)
constexpr uint32_t kSmallTokens = 512;    // tokens
constexpr uint32_t kSmallPayload = 3072;  // bytes; a wave's LDS slot holds X + payload + read slack

struct WorkQueues {
    uint32_t n_big, n_small;        // filled by k_layout
    uint32_t next_big, next_small;  // consumed by k_encode / k_encode_small
};

// Work distribution for k_hist / k_encode.  The hardware places workgroup i on XCD i % 8 and,
// inside the XCD, walks the CUs round-robin; a grid in which heavy (plane 0) and light
// workgroups alternate with a period that divides 256 parks the heavy ones on a fraction of
// the CUs (measured: 117 of 512 slots busy, profiles/r01_notes.md).  So both kernels are
// persistent: 2 workgroups per CU pull hzr-block indices from a counter until none are left.
struct WorkItem {
    uint32_t b, k, j;
};
__device__ __forceinline__ bool next_work(uint32_t* counter, uint32_t total, const Geom& g, uint32_t* s_slot, WorkItem& wi, uint32_t pass) {
    __syncthreads();  // everyone is done with the previous block (and with *s_slot)
    if (threadIdx.x == 0) *s_slot = pass == 0 ? blockIdx.x : gridDim.x + atomicAdd(counter, 1u);  // first item static
    __syncthreads();
    const uint32_t v = *s_slot;
    if (v >= total) return false;
    wi.k = v % kMaxPlanes;  // plane fastest
    wi.j = (v / kMaxPlanes) % g.nblk;
    wi.b = v / (kMaxPlanes * g.nblk);
    return true;
}

constexpr uint32_t kSegHistStride = kEncWaves * kSymStride;  // u16 elements per hzr block (hzr_rows.hip: k_hist)

// ===========================================================================
// k_tree: one wave per hzr block, 4 waves per workgroup
// ===========================================================================
constexpr int kTreeWaves = 4;
constexpr uint32_t kKeyMax = 0xFFFFFFFFu;

// (5084 bytes per tree: eight 4-wave workgroups = 32 trees in flight per CU.  The kernel is a latency chain per wave -- one dense
//  tree alone takes 0.045 ms -- so trees in flight are what it lives on: at 6.4 KB per tree a CU held 24.)
struct TreeLds {
    uint32_t key[kSymStride];     // leaf keys: count<<10 | (1023 - index); index order = creation order
    uint32_t lcnt[kNumSym];       // leaves in the subtree of each node, two u16 counts per word (LDS per wave decides how many
                                  // trees a CU builds at once, and the merge loop is a chain of dependent reductions)
    uint32_t up[2 * kNumSym];     // parent | isB<<10 | sibling<<11 (sibling = child_a, kept for child_b only).  Until the merges start
                                  // its first 264 words hold the block's token histogram (lhist(): dead once the counts are in registers)
    uint16_t leafsym[kSymStride];
    uint32_t tdesc[kTdescWords];
    __device__ __forceinline__ uint32_t* lhist() { return up; }
};
static_assert(sizeof(TreeLds) * 4 * 8 <= 160 * 1024, "eight k_tree workgroups per CU");

// add the tokens of a zero run of length R to a histogram (same split as run_bits / run_emit)
__device__ __forceinline__ void run_count(uint32_t* h, uint32_t R) {
    const uint32_t q = (R >= kRunCap) + (R >= 2 * kRunCap) + (R >= 3 * kRunCap);
    const uint32_t rem = R - q * kRunCap;
    if (q) atomicAdd(&h[260], q);
    if (rem) atomicAdd(&h[run_symbol(rem)], 1u);
}

// token histogram of a small block by one wave (zero runs are counted where they END; see encode_small_block)
__device__ __forceinline__ void small_block_hist(const uint8_t* __restrict__ in, uint32_t in_size, uint32_t segmask, uint32_t* h) {
    const uint32_t l = lane_id();
    const unsigned long long lt = (1ull << l) - 1ull;
    const uint32_t nseg = (in_size + 4095u) >> 12;
    uint32_t pend = 0;
    for (uint32_t seg = 0; seg < nseg; ++seg) {
        const uint32_t seg_base = seg << 12;
        if (!((segmask >> seg) & 1u)) {
            pend += min(4096u, in_size - seg_base);
            continue;
        }
        uint4 rows[4];
#pragma unroll
        for (uint32_t r = 0; r < 4; ++r) {  // the segment's four rows in one round trip
            const uint32_t pos = seg_base + (r << 10) + 16u * l;
            rows[r] = make_uint4(0, 0, 0, 0);
            if (pos < in_size) rows[r] = *reinterpret_cast<const uint4*>(in + pos);
        }
#pragma unroll
        for (uint32_t r = 0; r < 4; ++r) {
            const uint32_t base = seg_base + (r << 10);
            if (base < in_size) {
                const uint32_t row_valid = min(1024u, in_size - base);
                Granule gr;
                const uint32_t pos = base + 16u * l;
                gr.nv = pos < in_size ? min(16u, in_size - pos) : 0u;
                gr.w[0] = rows[r].x;
                gr.w[1] = rows[r].y;
                gr.w[2] = rows[r].z;
                gr.w[3] = rows[r].w;
                granule_finish(gr);
                const uint32_t lits = ~gr.zm & ((1u << gr.nv) - 1u);
                const unsigned long long nzb = __ballot(lits != 0);
                if (!nzb) {
                    pend += row_valid;
                } else {
                    const uint32_t last_nz = lits ? 31u - (uint32_t)__builtin_clz(lits) : 0u;
                    const uint32_t trail = lits ? (15u - last_nz) : 16u;
                    const unsigned long long below = nzb & lt;
                    const uint32_t p = below ? 63u - (uint32_t)__builtin_clzll(below) : 0u;
                    const uint32_t tp = (uint32_t)__shfl((int)trail, (int)p, 64);
                    const uint32_t zb = below ? (16u * (l - 1u - p) + tp) : (16u * l + pend);
                    uint32_t t = lits, prev_end = 0;
                    bool first = true;
                    while (t) {
                        const uint32_t i = (uint32_t)__builtin_ctz(t);
                        t &= t - 1;
                        const uint32_t R = first ? (zb + i) : (i - prev_end);
                        if (R) run_count(h, R);
                        atomicAdd(&h[granule_byte_dyn(gr.w[0], gr.w[1], gr.w[2], gr.w[3], i)], 1u);
                        prev_end = i + 1;
                        first = false;
                    }
                    const uint32_t pl = 63u - (uint32_t)__builtin_clzll(nzb);
                    const uint32_t lastlit = read_lane(last_nz, pl);  // pl from a ballot: wave-uniform
                    pend = row_valid - (16u * pl + lastlit + 1u);
                }
            }
        }
    }
    if (pend && l == 0) run_count(h, pend);
}

// The merge loop with the live keys kept SORTED (descending) across lanes and registers: element e lives in lane e % 64 of
// register e / 64, the n live ones are elements 0 .. n-1, so the two smallest keys -- the reference's `<=` scan picks exactly
// them (hzr_encode.c:251-260: count ascending, node index descending = key ascending) -- are simply elements n-1 and n-2: two
// v_readlane instead of two wave-wide reductions.  Popping them is n -= 2 (nothing moves); the parent's key goes to its place
// among the rest by ONE lane shift of the elements below it:
//     new[e] = old[e] > nk ? old[e] : (old[e-1] > nk ? nk : old[e-1])            (old[-1] = +inf)
// (what lies at or beyond n is stale and never read again: n only shrinks).  k_tree is bound by vector issue, not by latency
// (profiles/r03_notes.md: 0.9 vector instructions per cycle and CU, nearly all of them the reductions' half-rate DPP steps):
// per merge this is 2 + 5 per live register instructions against the 54 .. 76 of round 2's form (the live keys unordered in
// ceil(S/64) registers, two DPP wave-min extractions per merge).  The sort in front (rank = number of greater keys, keys are
// unique) costs ~7 instructions per leaf.
template <int NR>
__device__ __forceinline__ void sorted_merge_phase(TreeLds& t, uint32_t (&kreg)[5], uint32_t& n, uint32_t& node) {
    const uint32_t l = lane_id();
    while (n > 64u * (NR - 1) && n >= 2u) {
        const uint32_t e1 = n - 1u, e2 = n - 2u;
        uint32_t m1 = 0, m2 = 0;
#pragma unroll
        for (int r = 0; r < NR; ++r) {  // (wave-uniform selects: the two smallest sit in the last one or two live registers)
            if ((e1 >> 6) == (uint32_t)r) m1 = read_lane(kreg[r], e1 & 63u);
            if ((e2 >> 6) == (uint32_t)r) m2 = read_lane(kreg[r], e2 & 63u);
        }
        const uint32_t nk = (((m1 >> 10) + (m2 >> 10)) << 10) | (1023u - node);
        // insert nk among elements 0 .. n-3, from the top register down (a register's lane 0 looks at the OLD lane 63 below it)
#pragma unroll
        for (int r = NR - 1; r >= 0; --r) {
            const uint32_t below = r ? read_lane(kreg[r ? r - 1 : 0], 63u) : kKeyMax;
            const uint32_t prev = dpp<0x138>(below, kreg[r]);  // wave_shr:1 -- lane l sees lane l-1, lane 0 keeps `below`
            kreg[r] = kreg[r] > nk ? kreg[r] : (prev > nk ? nk : prev);
        }
        const uint32_t i1 = 1023u - (m1 & 1023u), i2 = 1023u - (m2 & 1023u);
        if (l == 0) {
            t.up[i1] = node;                             // child_a: code bit 0, described right after the branch bit
            t.up[i2] = node | (1u << 10) | (i1 << 11);  // child_b: code bit 1, described after child_a's subtree
        }
        ++node;
        --n;
    }
}

__device__ __forceinline__ void sorted_merge(TreeLds& t, uint32_t S) {
    const uint32_t l = lane_id();
    const uint32_t nreg = (S + 63u) >> 6;
    // ---- sort: every key's rank = the number of keys greater than it (unique keys: a permutation) ----
    uint32_t mine[5], rank[5];
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        mine[r] = (uint32_t)(r * 64) + l < S ? t.key[r * 64 + l] : kKeyMax;  // (lanes past the last leaf take no part)
        rank[r] = 0;
    }
    auto count_greater = [&](uint32_t nr) {
        for (uint32_t j = 0; j < S; ++j) {
            const uint32_t kj = t.key[j];  // (one address for all lanes: a broadcast read)
#pragma unroll
            for (int r = 0; r < 5; ++r)
                if ((uint32_t)r < nr) rank[r] += kj > mine[r] ? 1u : 0u;
        }
    };
    if (nreg <= 1) count_greater(1);
    else if (nreg <= 2) count_greater(2);
    else if (nreg <= 3) count_greater(3);
    else count_greater(5);
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();  // (every lane has read all the keys: they are sorted in place)
#pragma unroll
    for (int r = 0; r < 5; ++r)
        if ((uint32_t)r < nreg && (uint32_t)(r * 64) + l < S) t.key[rank[r]] = mine[r];
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
    uint32_t kreg[5];
#pragma unroll
    for (int r = 0; r < 5; ++r) kreg[r] = ((uint32_t)(r * 64) + l < S) ? t.key[r * 64 + l] : 0u;
    // ---- merges, with as many registers as still hold live keys ----
    uint32_t n = S, node = S;
    if (nreg > 4) sorted_merge_phase<5>(t, kreg, n, node);
    if (nreg > 3) sorted_merge_phase<4>(t, kreg, n, node);
    if (nreg > 2) sorted_merge_phase<3>(t, kreg, n, node);
    if (nreg > 1) sorted_merge_phase<2>(t, kreg, n, node);
    sorted_merge_phase<1>(t, kreg, n, node);
}

struct TreeOut {  // wave-uniform
    uint32_t mode, payload_len, tree_bits, ntok, fill;
};

// One wave: token histogram h[0..260] (LDS) of an hzr block of in_size bytes -> Fill test (hzr_encode.c:285-305),
// Huffman tree with the reference's tie-break (:222-283), code words via put_cw(sym, code, len), the pre-order tree
// description in t.tdesc (:177-219), stream bits per token of every used symbol in t.key[sym], and the block's mode
// and exact payload size (:377-469).
template <class PutCw>
__device__ __forceinline__ TreeOut build_tree(TreeLds& t, const uint32_t* h, uint32_t in_size, PutCw put_cw) {
    const uint32_t l = lane_id();
    // ---- leaves in ascending symbol order (hzr_encode.c:226-234) ----------
    uint32_t cnt[5], idx[5];
    uint32_t base = 0, nonzero_syms = 0, zero_kind = 0, fillval = 0;
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        const uint32_t s = r * 64 + l;
        cnt[r] = s < (uint32_t)kNumSym ? h[s] : 0u;
        const unsigned long long bal = __ballot(cnt[r] != 0);
        idx[r] = base + (uint32_t)__popcll(bal & ((1ull << l) - 1ull));
        base += (uint32_t)__popcll(bal);
        // OnlySingleCode (hzr_encode.c:285-305): all zero-type symbols count as one code
        const bool is_zero_kind = (s == 0) || (s >= 256);
        const unsigned long long nzb = __ballot(cnt[r] != 0 && !is_zero_kind);
        nonzero_syms += (uint32_t)__popcll(nzb);
        zero_kind |= __ballot(cnt[r] != 0 && is_zero_kind) ? 1u : 0u;
        if (nzb) fillval = max(fillval, (uint32_t)(r * 64 + (63 - __builtin_clzll(nzb))));
    }
    const uint32_t S = base;
    if (zero_kind + nonzero_syms == 1) return TreeOut{kModeFill, 1u, 0u, 0u, zero_kind ? 0u : fillval};  // EncodeFill (:341-367)

    // ---- node arrays -------------------------------------------------------
    for (uint32_t i = l; i < (uint32_t)kSymStride; i += 64) t.key[i] = kKeyMax;
    for (uint32_t i = l; i < (uint32_t)kNumSym; i += 64) t.lcnt[i] = (2u * i < S ? 1u : 0u) | (2u * i + 1u < S ? 1u << 16 : 0u);
    for (uint32_t i = l; i < (uint32_t)kTdescWords; i += 64) t.tdesc[i] = 0;
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        if (cnt[r]) {
            t.key[idx[r]] = (cnt[r] << 10) | (1023u - idx[r]);
            t.leafsym[idx[r]] = (uint16_t)(r * 64 + l);
        }
    }
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();

    const uint32_t nnodes = 2 * S - 1;
#if defined(TREE_PROBE) && TREE_PROBE == 2  // timing probes (never in the product)
    return TreeOut{kModeFill, 1u, 0u, 0u, 0u};
#endif
    sorted_merge(t, S);
#if defined(TREE_PROBE) && TREE_PROBE == 1
    return TreeOut{kModeFill, 1u, 0u, 0u, 0u};
#endif
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();

    // ---- leaves under every internal node: each leaf credits its ancestors ----
    const uint32_t root = nnodes - 1;
    for (uint32_t i = l; i < S; i += 64) {
        uint32_t cur = i, guard = 0;
        while (cur != root && guard++ < 64) {  // depth <= 22 for <= 65536 tokens; the bound only guards against a corrupted link
            cur = t.up[cur] & 1023u;
            atomicAdd(&t.lcnt[cur >> 1], 1u << ((cur & 1u) * 16u));
        }
    }
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();

    // ---- per-leaf walk to the root: code, length, description offset -------
    // a subtree with L leaves is described in 10 L + (L - 1) bits (leaf = '1' + 9-bit symbol, branch = '0')
    const uint32_t tree_bits = 11u * S - 1u;
    uint32_t bits_sum = 0;
    for (uint32_t i = l; i < S; i += 64) {
        uint32_t cur = i, code = 0, len = 0, off = 0;
        while (cur != root && len < 64) {
            const uint32_t u = t.up[cur];
            const uint32_t is_b = (u >> 10) & 1u;
            code = (code << 1) | is_b;
            off += is_b ? 11u * ((t.lcnt[u >> 12] >> (((u >> 11) & 1u) * 16u)) & 0xFFFFu) : 1u;  // '0' of the branch (+ all of child_a's subtree: 1 + 11 L - 1)
            ++len;
            cur = u & 1023u;
        }
        const uint32_t sym = t.leafsym[i];
        put_cw(sym, code, len);
        t.key[sym] = len + run_extra_bits(sym);  // stream bits per token of this symbol (the key slots are free now)
        // description: '1' then the 9-bit symbol, LSB first (hzr_encode.c:184-191)
        const uint32_t v = 1u | (sym << 1);
        const uint32_t wi = off >> 5, sh = off & 31u;
        atomicOr(&t.tdesc[wi], v << sh);
        if (sh > 22) atomicOr(&t.tdesc[wi + 1], v >> (32 - sh));
    }
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
    // payload bits of the codes: every symbol's count (still in this lane's registers: the histogram's LDS words were taken over
    // by the tree's links) times its stream bits per token
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        const uint32_t s2 = r * 64 + l;
        if (cnt[r]) bits_sum += cnt[r] * t.key[s2 < (uint32_t)kNumSym ? s2 : 0u];
    }
    bits_sum = wave_add_u32(bits_sum);
    const uint32_t ntok = wave_add_u32(cnt[0] + cnt[1] + cnt[2] + cnt[3] + cnt[4]);
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
    const uint32_t total_bits = tree_bits + bits_sum;
    const uint32_t nbytes = (total_bits + 7) >> 3;
    // Huffman iff the payload fits in in_size bytes and is < 65536 (hzr_encode.c:377-382,463-469)
    const bool huff = nbytes <= in_size && nbytes < kHzrBlock;
    if (huff) return TreeOut{kModeHuff, nbytes, tree_bits, ntok, 0u};  // (ntok travels in BlockMeta::fill in this mode)
    return TreeOut{kModeCopy, in_size, 0u, 0u, 0u};
}

// k_tree: one wave per hzr block, 4 waves per workgroup.  Small blocks (<= kSmallSegments non-zero 4 KiB segments) take their
// own histogram here; for the others k_hist left the block histogram and the per-segment histograms, which -- times the
// code lengths -- give the stream bit at which each 4 KiB segment's tokens start (k_encode then needs no bit-count pass).
__global__ __launch_bounds__(kTreeWaves * 64) void k_tree(const uint32_t* __restrict__ hist, const uint8_t* __restrict__ planes, Geom g,
                                                         const uint32_t* __restrict__ nbuse, const uint32_t* __restrict__ nzflag,
                                                         uint32_t nhb_total, uint32_t* __restrict__ cw, uint32_t* __restrict__ tdesc,
                                                         BlockMeta* __restrict__ meta, const uint32_t* __restrict__ seghist,
                                                         uint32_t* __restrict__ segbase, uint32_t* __restrict__ zero_next, uint32_t zero_words,
                                                         uint32_t psel_arg) {
    const uint32_t psel = RSPT_DIAG_ONLY(psel_arg);  // timing probes (diagnostic builds only): no tree for plane 0 (bit 4) / planes >= 1 (bit 5)
    __shared__ TreeLds s_t[kTreeWaves];
    const uint32_t l = lane_id();
    const uint32_t wv = threadIdx.x >> 6;
    // the other copy of the per-call zero region, for the next call (rspt_hip.hip: zbuf)
    for (uint32_t i = blockIdx.x * (kTreeWaves * 64u) + threadIdx.x; i < zero_words; i += gridDim.x * (kTreeWaves * 64u)) zero_next[i] = 0;
    // wave -> hzr block, plane-major: all the blocks of plane 0 (the dense, expensive ones) are dispatched first and
    // next to each other, so they spread over every CU; in (b, k, j) order they recur with a period that the
    // dispatcher's round-robin maps onto a quarter of the CUs (profiles/r01_notes.md: placement resonance)
    const uint32_t v = blockIdx.x * kTreeWaves + wv;
    if (v >= nhb_total) return;
    const uint32_t per_plane = nhb_total / kMaxPlanes;  // = blocks * nblk
    const uint32_t k = v / per_plane, rest = v - k * per_plane;
    const uint32_t b = rest / g.nblk, j = rest - b * g.nblk;
    const uint32_t hb = hb_index(g, b, k, j);
    TreeLds& t = s_t[wv];
    if (k >= nbuse[b]) {
        if (l == 0) meta[hb] = BlockMeta{kModeSkip, 0, 0, 0};
        return;
    }
    const uint32_t segmask = (psel && ((k == 0 && (psel & 16u)) || (k >= 1 && (psel & 32u)))) ? 0u : nzflag[hb];
    if (!segmask) {  // all-zero block (flagged by the front end): EncodeFill with value 0
        if (l == 0) meta[hb] = BlockMeta{kModeFill, 1u, 0u, 0u};
        return;
    }
    const uint32_t in_size = min(kHzrBlock, g.N - j * kHzrBlock);
    uint32_t* h = t.lhist();
    const bool own_hist = (uint32_t)__popc(segmask) <= kSmallSegments;
    if (l == 0) segbase[(size_t)hb * kEncWaves] = 0xFFFFFFFFu;  // "no segment offsets" unless set below
    if (own_hist) {  // small block: this wave takes the histogram itself
        for (uint32_t i = l; i < (uint32_t)kSymStride; i += 64) h[i] = 0;
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        small_block_hist(planes + ((size_t)b * kMaxPlanes + k) * g.plane_stride + (size_t)j * kHzrBlock, in_size, segmask, h);
    } else {
        for (uint32_t i = l; i < (uint32_t)kSymStride; i += 64) h[i] = hist[(size_t)hb * kSymStride + i];
    }
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
    uint32_t* cwo = cw + (size_t)hb * kSymStride;
    const TreeOut r = build_tree(t, h, in_size, [&](uint32_t sym, uint32_t code, uint32_t len) { cwo[sym] = code | (len << 24); });
    if (l == 0) meta[hb] = BlockMeta{r.mode, r.payload_len, r.tree_bits, r.mode == kModeHuff ? r.ntok : r.fill};
    if (r.mode != kModeHuff) return;
    uint32_t* tdo = tdesc + (size_t)hb * kTdescWords;
    for (uint32_t i = l; i < (uint32_t)kTdescWords; i += 64) tdo[i] = t.tdesc[i];
    if (own_hist) return;
    // (bins 261..263 of k_hist's histograms count the zero bytes that end no token: they cost nothing)
    if (l < 3u) t.key[(uint32_t)kNumSym + l] = 0u;
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();

    // ---- stream bit at which the tokens of each 4 KiB segment start (k_hist's per-segment histograms x code lengths) ----
    // lane l: segment l / 4, every fourth of its 132 u16 pairs from pair l % 4 on (the four lanes of a segment read 16
    // consecutive bytes: a wave load touches 16 lines, not 64)
    {
        const uint32_t seg = l >> 2, part = l & 3u;
        const uint32_t* sh = seghist + (size_t)hb * (kSegHistStride / 2) + seg * (kSymStride / 2) + part;
        uint32_t pr[33];
#pragma unroll
        for (int i = 0; i < 33; ++i) pr[i] = sh[4 * i];
        uint32_t acc = 0;
#pragma unroll
        for (int i = 0; i < 33; ++i) {
            const uint32_t s0 = 2u * (4u * (uint32_t)i + part);
            // unused symbols have count 0: whatever their cost slots hold is multiplied away
            acc += (pr[i] & 0xFFFFu) * t.key[s0];
            acc += (pr[i] >> 16) * t.key[s0 + 1u];
        }
        acc += dpp<0xB1>(0u, acc);  // quad_perm [1,0,3,2]
        acc += dpp<0x4E>(0u, acc);  // quad_perm [2,3,0,1]: every lane of the quad holds the segment's bits
        // exclusive prefix over the 16 segments (one value per quad): scan with the quad's bits counted once
        const uint32_t mine = part == 0 ? acc : 0u;
        const uint32_t incl = wave_scan_add(mine);
        if (part == 0) segbase[(size_t)hb * kEncWaves + seg] = 32u + r.tree_bits + incl - mine;  // the payload starts at image byte 4
    }
}

// ===========================================================================
// k_layout: one workgroup per block.  Stream grammar:
//   [method][means header][ per plane k<nb: u32 len_k, u32 N, hzr blocks... ]
// ===========================================================================
__device__ __forceinline__ void store_le32(uint8_t* p, uint32_t v) {
    p[0] = (uint8_t)v;
    p[1] = (uint8_t)(v >> 8);
    p[2] = (uint8_t)(v >> 16);
    p[3] = (uint8_t)(v >> 24);
}

// It also sorts the hzr blocks into the work queues of k_encode: Fill blocks (8 bytes) are written
// right here; Huffman blocks with few tokens and a small payload go to the `small` queue (one WAVE encodes one such
// block), everything else to the `big` queue (one 1024-thread workgroup per block).


__global__ __launch_bounds__(256) void k_layout(Geom g, const uint32_t* __restrict__ nbuse, const BlockMeta* __restrict__ meta,
                                               const uint8_t* __restrict__ means_hdr, uint8_t* __restrict__ dst, uint64_t dst_stride,
                                               uint64_t* __restrict__ out_off, uint64_t* __restrict__ sizes, const CrcConsts* __restrict__ cc,
                                               const uint32_t* __restrict__ nzflag, WorkQueues* __restrict__ wq,
                                               uint32_t* __restrict__ big_list, uint32_t* __restrict__ small_list,
                                               uint32_t* __restrict__ plane_dirty, uint32_t dirty_shift, uint32_t psel_arg) {
    const uint32_t psel = RSPT_DIAG_ONLY(psel_arg);  // timing probes (diagnostic builds only): k_encode gets no plane 0 (bit 2) / planes >= 1 (bit 3)
    __shared__ uint64_t s_part[256];
    __shared__ uint64_t s_plane_end[kMaxPlanes + 1];
    __shared__ uint32_t s_dirty[kMaxPlanes * 4];  // 128 bits per plane: hzr blocks (j >> dirty_shift) that keep their data
    if (threadIdx.x < kMaxPlanes * 4) s_dirty[threadIdx.x] = 0;
    const uint32_t b = blockIdx.x, tid = threadIdx.x;
    const uint32_t nb = nbuse[b];
    const uint32_t n = nb * g.nblk;  // hzr blocks of this block, (k,j) order
    const uint32_t per = (n + 255) / 256;
    const uint32_t lo = min(n, tid * per), hi = min(n, lo + per);
    const uint32_t hb0 = hb_index(g, b, 0, 0);
    auto enc_size = [&](uint32_t q) -> uint64_t {  // q = k*nblk + j
        return 7ull + meta[hb0 + q].payload_len;
    };
    uint64_t sum = 0;
    __syncthreads();  // (s_dirty is zeroed)
    for (uint32_t q = lo; q < hi; ++q) {
        sum += enc_size(q);
        // non-zero data that no encoder wipes (dense Huffman blocks, PlainCopy, constant non-zero blocks) stays behind
        if (nzflag[hb0 + q] && !block_is_wiped(meta[hb0 + q])) {
            const uint32_t k = q / g.nblk, bucket = (q - k * g.nblk) >> dirty_shift;
            atomicOr(&s_dirty[k * 4 + (bucket >> 5)], 1u << (bucket & 31u));
        }
    }
    // exclusive prefix of the 256 partial sums: a scan inside each wave, the four wave totals through LDS (the single-thread
    // loop over 256 LDS words this replaces was most of the kernel's 13 us: ~50 dependent LDS round trips per microsecond)
    uint64_t run;
    {
        const uint32_t l = tid & 63u, wv = tid >> 6;
        uint64_t inc = sum;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
            const uint64_t up = (uint64_t)__shfl_up((long long)inc, dd, 64);
            if (l >= (uint32_t)dd) inc += up;
        }
        if (l == 63u) s_part[wv] = inc;
        __syncthreads();
        uint64_t pre = 0;
        for (uint32_t q2 = 0; q2 < wv; ++q2) pre += s_part[q2];
        run = pre + inc - sum;
    }
    // absolute offset of hzr block q: 1 + hdr + 8*(k+1) + sum of the encoded blocks before it
    const uint64_t head = 1ull + g.hdr_len;
    for (uint32_t q = lo; q < hi; ++q) {
        const uint32_t k = q / g.nblk;
        out_off[hb0 + q] = head + 8ull * (k + 1) + run;
        run += enc_size(q);
        if ((q + 1) % g.nblk == 0) s_plane_end[k + 1] = run;  // encoded bytes up to the end of plane k
    }
    if (tid == 0) s_plane_end[0] = 0;
    __syncthreads();
    const uint64_t total = head + 8ull * nb + s_plane_end[nb];
    const bool fits = total <= dst_stride;
    if (tid == 0) sizes[b] = fits ? total : (total | (1ull << 63));
    // the planes written in this call: dirty when a dense block stays behind, or when nothing will be encoded (and wiped) at all
    if (tid < nb * 4) plane_dirty[(size_t)b * kMaxPlanes * 4 + tid] = fits ? s_dirty[tid] : 0xFFFFFFFFu;
    if (!fits) {  // tell k_encode to leave this block alone
        for (uint32_t q = lo; q < hi; ++q) out_off[hb0 + q] = ~0ull;
        return;
    }
    uint8_t* o = dst + (size_t)b * dst_stride;
    for (uint32_t q = lo; q < hi; ++q) {
        const BlockMeta m = meta[hb0 + q];
        if (m.mode == kModeFill) {  // EncodeFill (hzr_encode.c:341-367): [00 00][crc32c(value)][02][value]
            uint8_t* f = o + out_off[hb0 + q];
            uint32_t c = 0xFFFFFFFFu ^ m.fill;
            c = ~((c >> 8) ^ cc->table[0][c & 0xFFu]);
            f[0] = 0;
            f[1] = 0;
            f[2] = (uint8_t)c;
            f[3] = (uint8_t)(c >> 8);
            f[4] = (uint8_t)(c >> 16);
            f[5] = (uint8_t)(c >> 24);
            f[6] = (uint8_t)kModeFill;
            f[7] = (uint8_t)m.fill;
        } else if (m.mode == kModeHuff && m.payload_len <= kSmallPayload && m.fill <= kSmallTokens &&
                   __popc(nzflag[hb0 + q]) <= (int)kSmallSegments) {
            small_list[atomicAdd(&wq->n_small, 1u)] = hb0 + q;
        } else if (m.mode == kModeHuff || m.mode == kModeCopy) {
            if (psel && ((q / g.nblk == 0 && (psel & 4u)) || (q / g.nblk >= 1 && (psel & 8u)))) continue;
            big_list[atomicAdd(&wq->n_big, 1u)] = hb0 + q;
        }
    }
    if (tid == 0) o[0] = (uint8_t)g.method;  // signal_packer_base.cpp:83
    for (uint32_t i = tid; i < g.hdr_len; i += 256) o[1 + i] = means_hdr[(size_t)b * g.hdr_len + i];  // :86-91
    if (tid < nb) {
        const uint32_t k = tid;
        const uint64_t pstart = head + 8ull * k + s_plane_end[k];
        const uint64_t plen = 4ull + (s_plane_end[k + 1] - s_plane_end[k]);  // hzr stream = master header + blocks
        store_le32(o + pstart, (uint32_t)plen);                               // base.cpp:78
        store_le32(o + pstart + 4, g.N);                                      // hzr_encode.c:521-522
    }
}

// ===========================================================================
// k_encode
// ===========================================================================
// The payload image lives in LDS as plain consecutive words.  Logical word 0 holds the CRC prefix X, the payload starts at
// byte 4: the image IS the virtual CRC input V = X || payload.  Both access patterns are bank-conflict free without
// padding: consecutive words by consecutive lanes (emit, copy-out), and the CRC reads word-strided -- lane tid owns
// the virtual words tid, tid + 1024, ... counted from the END of V (tools/kernel_model.py:crc_strided).
constexpr uint32_t kStageWords = kHzrBlock / 4 + 32;  // X + payload + read slack
constexpr uint32_t kStagePhys = kStageWords + 2;

constexpr uint32_t kRunClsEntries = 280;  // lengths 0..278 and ">= 279" (hzr_internal.h:117-121)
__device__ __forceinline__ uint32_t run_class_entry(uint32_t z) {
    const uint32_t sym = run_symbol(z);
    const uint32_t base = sym == 257 ? 3u : sym == 258 ? 7u : sym == 259 ? 23u : sym == 260 ? 279u : z;  // (runs of 1 and 2: no extra bits, value 0)
    return sym | (run_extra_bits(sym) << 12) | (base << 16);
}

// OR a string of `len` <= 64 bits (hi:lo) into the image at bit `pos`.  Two words always; the third only where some lane's
// string reaches into it -- with the usual four codes of ~5.5 bits per string that is no lane of the wave, and the third
// atomic with its shifts and address is one instruction in eight of the dense emit (one wave-wide test instead)
__device__ __forceinline__ void or_bits64(uint32_t* stage, uint32_t pos, uint32_t lo, uint32_t hi, uint32_t len) {
    const uint32_t word = pos >> 5, sh = pos & 31u;
    const uint64_t sv = (((uint64_t)hi << 32) | lo) << sh;
    atomicOr(&stage[word], (uint32_t)sv);
    atomicOr(&stage[(word + 1)], (uint32_t)(sv >> 32));
    if (__builtin_amdgcn_ballot_w64(sh + len > 64u)) atomicOr(&stage[(word + 2)], (hi >> 1) >> (31u - sh));
}

// OR one token (<= 38 bits) into the image at bit `pos`
__device__ __forceinline__ void or_token(uint32_t* stage, uint32_t pos, uint32_t lo, uint32_t hi, uint32_t len) {
    const uint32_t word = pos >> 5, sh = pos & 31u;
    const uint64_t sv = (((uint64_t)hi << 32) | lo) << sh;
    atomicOr(&stage[word], (uint32_t)sv);
    atomicOr(&stage[(word + 1)], (uint32_t)(sv >> 32));
    if (sh + len > 64) atomicOr(&stage[(word + 2)], (hi >> 1) >> (31u - sh));
}

constexpr uint32_t kLightPayload = 16384;    // bytes: below it the image words from kTokQueueBase on are free (sparse-row queues)
constexpr uint32_t kTokQueueBase = 4200;     // stage word (> (16384 + 4) / 4 + 24)

// byte q of the image (q = 0..3: X, q >= 4: payload byte q-4)
__device__ __forceinline__ uint32_t stage_byte(const uint32_t* stage, uint32_t q) { return (stage[(q >> 2)] >> ((q & 3u) * 8)) & 0xFFu; }

// ---------------------------------------------------------------------------------------------
// Small blocks: one WAVE encodes one hzr block, 16 blocks per workgroup pass, no workgroup barrier.
// The wave streams over the block's rows in order, so it needs only forward information: every
// zero run is emitted when it ENDS, right before the literal that ends it (or at the block end),
// as floor(R/16662) capped tokens plus a remainder token -- the same token sequence as the
// reference's greedy walk (hzr_encode.c:410-457).  Zero 4 KiB segments are skipped without a read.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kSlotWords = 1090;  // per-wave LDS slot of the small-block encoder: [cw 264][image]
constexpr uint32_t kSlotImage = kSlotWords - kSymStride;  // words of X || payload (+ slack)
static_assert(kSlotImage * 4 >= kSmallPayload + 4 + 72, "small-block slot too small");

struct LinSink {  // bit sink of the one-wave encoder: a 32-bit partial word, whole words stored to the wave's image
    uint32_t* img;
    uint32_t lo, n, word;
    __device__ __forceinline__ void start(uint32_t* s, uint32_t bitpos) {
        img = s;
        lo = 0;
        n = bitpos & 31u;
        word = bitpos >> 5;
    }
    __device__ __forceinline__ void put(uint32_t v, uint32_t len) {
        lo |= v << n;
        const uint32_t tot = n + len;
        if (tot >= 32) {
            atomicOr(&img[word], lo);
            ++word;
            lo = (v >> 1) >> (31u - n);
            n = tot - 32;
        } else {
            n = tot;
        }
    }
    __device__ __forceinline__ void flush() {
        if (n) atomicOr(&img[word], lo);
    }
};

// stream bits of the tokens of a zero run of length R (0 < R): capped tokens, then the remainder
__device__ __forceinline__ uint32_t run_bits(const uint32_t* cwt, uint32_t R) {
    const uint32_t q = (R >= kRunCap) + (R >= 2 * kRunCap) + (R >= 3 * kRunCap);
    const uint32_t rem = R - q * kRunCap;
    uint32_t bits = q * ((cwt[260] >> 24) + 14u);
    if (rem) {
        const uint32_t sym = run_symbol(rem);
        bits += (cwt[sym] >> 24) + run_extra_bits(sym);
    }
    return bits;
}

template <typename Sink>
__device__ __forceinline__ void run_emit(Sink& sink, const uint32_t* cwt, uint32_t R) {
    const uint32_t q = (R >= kRunCap) + (R >= 2 * kRunCap) + (R >= 3 * kRunCap);
    const uint32_t rem = R - q * kRunCap;
    for (uint32_t i = 0; i < q; ++i) {
        const uint32_t c = cwt[260];
        sink.put(c & 0x00FFFFFFu, c >> 24);
        sink.put(kRunCap - 279u, 14);
    }
    if (rem) {
        const uint32_t sym = run_symbol(rem);
        const uint32_t c = cwt[sym];
        sink.put(c & 0x00FFFFFFu, c >> 24);
        const uint32_t eb = run_extra_bits(sym);
        if (eb) sink.put(run_extra_value(sym, rem), eb);
    }
}

__device__ __forceinline__ uint32_t crc_chunk64_lin(const uint32_t* img, const uint32_t (*tab)[256], int32_t lo) {
    const int32_t a = lo >> 2;
    const uint32_t sh = (uint32_t)lo & 3u;
    uint32_t c = 0;
    uint32_t prev = a >= 0 ? img[a] : 0u;
#pragma unroll
    for (int32_t q = 0; q < 16; ++q) {
        const int32_t ix = a + q + 1;
        const uint32_t next = ix >= 0 ? img[ix] : 0u;
        c ^= __builtin_amdgcn_alignbyte(next, prev, sh);
        prev = next;
        c = tab[3][c & 0xFFu] ^ tab[2][(c >> 8) & 0xFFu] ^ tab[1][(c >> 16) & 0xFFu] ^ tab[0][c >> 24];
    }
    return c;
}

__device__ __forceinline__ void encode_small_block(uint32_t* cwt, uint32_t* img, const uint32_t (*crc_tab)[256], uint32_t hb,
                                                   uint8_t* __restrict__ planes, const Geom& g,
                                                   const uint32_t* __restrict__ nzflag, const BlockMeta* __restrict__ meta,
                                                   const uint32_t* __restrict__ cw, const uint32_t* __restrict__ tdesc,
                                                   const uint64_t* __restrict__ out_off, const CrcConsts* __restrict__ cc,
                                                   uint8_t* __restrict__ dst, uint64_t dst_stride, uint32_t ablate_arg) {
    const uint32_t ablate = RSPT_DIAG_ONLY(ablate_arg);  // timing probes: diagnostic builds only
    const uint32_t l = lane_id();
    if (ablate & 512u) return;
    const uint32_t j = hb % g.nblk, k = (hb / g.nblk) % kMaxPlanes, b = hb / (g.nblk * kMaxPlanes);
    const BlockMeta m = meta[hb];
    const uint64_t off = out_off[hb];
    const uint32_t segmask = nzflag[hb];
    if (off == ~0ull) return;
    const uint32_t in_size = min(kHzrBlock, g.N - j * kHzrBlock);
    uint8_t* in = planes + ((size_t)b * kMaxPlanes + k) * g.plane_stride + (size_t)j * kHzrBlock;
    const uint32_t L = m.payload_len;
    for (uint32_t i = l; i < (uint32_t)kSymStride; i += 64) cwt[i] = cw[(size_t)hb * kSymStride + i];
    for (uint32_t i = l; i < kSlotImage; i += 64) img[i] = 0;
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    if (l == 0) img[0] = cc->prefix;
    const uint32_t twords = (m.tree_bits + 31) >> 5;
    for (uint32_t i = l; i < twords; i += 64) atomicOr(&img[1 + i], tdesc[(size_t)hb * kTdescWords + i]);

    uint32_t pend = 0;                       // zeros pending in front of the current position (wave-uniform)
    uint32_t bitpos = 32u + m.tree_bits;     // next stream bit (wave-uniform)
    const unsigned long long lt = (1ull << l) - 1ull;
    const uint32_t nseg = (in_size + 4095u) >> 12;
    // One wave walks 64 rows in order, so HBM latency is hidden by reading a whole 4 KiB segment
    // (4 rows) at once, one segment ahead of the one being processed.
    uint4 cur[4], nxt[4];
    auto issue = [&](uint32_t seg, uint4* buf) {
#pragma unroll
        for (uint32_t r = 0; r < 4; ++r) {
            const uint32_t pos = (seg << 12) + (r << 10) + 16u * l;
            buf[r] = make_uint4(0, 0, 0, 0);
            if (seg < nseg && ((segmask >> seg) & 1u) && pos < in_size) buf[r] = *reinterpret_cast<const uint4*>(in + pos);
        }
    };
    issue(0, cur);
    for (uint32_t seg = 0; seg < nseg && !(ablate & 1024u); ++seg) {
        issue(seg + 1, nxt);
        const uint32_t seg_base = seg << 12;
        if (!((segmask >> seg) & 1u)) {
            pend += min(4096u, in_size - seg_base);
        } else {
#pragma unroll
            for (uint32_t r = 0; r < 4; ++r) {
                const uint32_t base = seg_base + (r << 10);
                if (base < in_size) {
                    const uint32_t row_valid = min(1024u, in_size - base);
                    Granule gr;
                    const uint32_t pos = base + 16u * l;
                    gr.nv = pos < in_size ? min(16u, in_size - pos) : 0u;
                    gr.w[0] = cur[r].x;
                    gr.w[1] = cur[r].y;
                    gr.w[2] = cur[r].z;
                    gr.w[3] = cur[r].w;
                    granule_finish(gr);
                    const uint32_t lits = ~gr.zm & ((1u << gr.nv) - 1u);
                    // clean-block invariant: every small block is wiped (block_is_wiped); this lane alone read the granule
                    if (lits) *reinterpret_cast<uint4*>(in + pos) = make_uint4(0, 0, 0, 0);
                    const unsigned long long nzb = __ballot(lits != 0);
                    if (!nzb) {
                        pend += row_valid;
                    } else {
                        // zeros in front of this granule: the nearest lower lane with a literal closes the run
                        const uint32_t last_nz = lits ? 31u - (uint32_t)__builtin_clz(lits) : 0u;  // index of the granule's last literal
                        const uint32_t trail = lits ? (15u - last_nz) : 16u;                         // zeros behind it inside the (full) granule
                        const unsigned long long below = nzb & lt;
                        const uint32_t p = below ? 63u - (uint32_t)__builtin_clzll(below) : 0u;
                        const uint32_t tp = (uint32_t)__shfl((int)trail, (int)p, 64);
                        const uint32_t zb = below ? (16u * (l - 1u - p) + tp) : (16u * l + pend);
                        // pass A: bits of this lane's tokens
                        uint32_t nbits = 0;
                        {
                            uint32_t t = lits, prev_end = 0;  // prev_end = index after the previous literal
                            bool first = true;
                            while (t) {
                                const uint32_t i = (uint32_t)__builtin_ctz(t);
                                t &= t - 1;
                                const uint32_t R = first ? (zb + i) : (i - prev_end);
                                if (R) nbits += run_bits(cwt, R);
                                nbits += cwt[granule_byte_dyn(gr.w[0], gr.w[1], gr.w[2], gr.w[3], i)] >> 24;
                                prev_end = i + 1;
                                first = false;
                            }
                        }
                        const uint32_t inc = wave_scan_add(nbits);
                        const uint32_t mypos = bitpos + inc - nbits;
                        bitpos += read_lane(inc, 63);
                        // pass B: emit
                        if (nbits) {
                            LinSink sink;
                            sink.start(img, mypos);
                            uint32_t t = lits, prev_end = 0;
                            bool first = true;
                            while (t) {
                                const uint32_t i = (uint32_t)__builtin_ctz(t);
                                t &= t - 1;
                                const uint32_t R = first ? (zb + i) : (i - prev_end);
                                if (R) run_emit(sink, cwt, R);
                                const uint32_t c = cwt[granule_byte_dyn(gr.w[0], gr.w[1], gr.w[2], gr.w[3], i)];
                                sink.put(c & 0x00FFFFFFu, c >> 24);
                                prev_end = i + 1;
                                first = false;
                            }
                            sink.flush();
                        }
                        // zeros behind the row's last literal stay pending
                        const uint32_t pl = 63u - (uint32_t)__builtin_clzll(nzb);
                        const uint32_t lastlit = read_lane(last_nz, pl);  // pl from a ballot: wave-uniform
                        pend = row_valid - (16u * pl + lastlit + 1u);
                    }
                }
            }
        }
#pragma unroll
        for (uint32_t r = 0; r < 4; ++r) cur[r] = nxt[r];
    }
    if (pend && l == 0) {  // the run that reaches the block end
        LinSink sink;
        sink.start(img, bitpos);
        run_emit(sink, cwt, pend);
        sink.flush();
    }
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();

    // CRC-32C of X || payload: lane l owns the 64-byte chunk (63 - l) counted from the end
    const int32_t Lv = (int32_t)L + 4;
    const int32_t hi = Lv - 64 * (int32_t)(63u - l);
    uint32_t c = 0;
    if (hi > 0 && !(ablate & 2048u)) c = crc_chunk64_lin(img, crc_tab, hi - 64);
    const uint32_t crc = (ablate & 2048u) ? 0u : ~wave_xor_u32(gf_shift(cc, l, c));

    uint8_t* o = dst + (size_t)b * dst_stride + off;
    if (l == 0) {
        o[0] = (uint8_t)(L - 1);
        o[1] = (uint8_t)((L - 1) >> 8);
        o[2] = (uint8_t)crc;
        o[3] = (uint8_t)(crc >> 8);
        o[4] = (uint8_t)(crc >> 16);
        o[5] = (uint8_t)(crc >> 24);
        o[6] = (uint8_t)kModeHuff;
    }
    const uint8_t* img8 = reinterpret_cast<const uint8_t*>(img) + 4;
    for (uint32_t i = l; i < L; i += 64) o[7 + i] = img8[i];
    __builtin_amdgcn_wave_barrier();  // the slot is reused by this wave's next block
}

// small blocks: 4 waves per workgroup, each wave pulls blocks on its own (no workgroup barrier after the table load)
constexpr int kSmallWaves = 4;
__global__ __launch_bounds__(kSmallWaves * 64) void k_encode_small(uint8_t* __restrict__ planes, Geom g, const uint32_t* __restrict__ nzflag,
                                                                  const BlockMeta* __restrict__ meta, const uint32_t* __restrict__ cw,
                                                                  const uint32_t* __restrict__ tdesc, const uint64_t* __restrict__ out_off,
                                                                  const CrcConsts* __restrict__ cc, uint8_t* __restrict__ dst, uint64_t dst_stride,
                                                                  WorkQueues* __restrict__ wq, const uint32_t* __restrict__ small_list, uint32_t ablate,
                                                                  uint32_t* __restrict__ report) {
    __shared__ uint32_t s_crc[4][256];
    __shared__ uint32_t s_slot[kSmallWaves][kSlotWords];
    // the host reads this (a word of its own memory) before the NEXT batches: a batch without small blocks needs no side stream
    if (blockIdx.x == 0 && threadIdx.x == 0) *report = wq->n_small;
    for (uint32_t i = threadIdx.x; i < 1024; i += kSmallWaves * 64) (&s_crc[0][0])[i] = (&cc->table[0][0])[i];
    __syncthreads();
    const uint32_t n_small = wq->n_small;
    const uint32_t wv = threadIdx.x >> 6;
    const uint32_t nwaves = gridDim.x * kSmallWaves;
    for (uint32_t pass = 0;; ++pass) {
        uint32_t i = blockIdx.x * kSmallWaves + wv;  // first block: static
        if (pass) {
            if (lane_id() == 0) i = nwaves + atomicAdd(&wq->next_small, 1u);
            i = (uint32_t)__builtin_amdgcn_readfirstlane((int)i);
        }
        if (i >= n_small) break;
        const uint32_t hb = small_list[i];
        encode_small_block(s_slot[wv], s_slot[wv] + kSymStride, s_crc, hb, planes, g, nzflag, meta, cw, tdesc, out_off, cc, dst, dst_stride, ablate);
    }
}

// ===========================================================================
// container packing (rspt_hip_pack_batch_dev)
// ===========================================================================
constexpr uint64_t kPackMagic = 0x4B43415054505352ull;  // "RSPTPACK"
constexpr uint32_t kPackHead = 32;                      // magic, nblocks, payload bytes, nb | bad << 32
constexpr uint64_t kPackLenMask = (1ull << 56) - 1ull;  // index length word: length | nb << 56 | invalid << 63

// The index is self-describing per stream: nb escalates inside a batch and every rank escalates on its own, so the
// plane count of stream i (nbuse[i], the reference's nr_bytes_to_compress_ at that call) travels in its length word.
// A stream that did not fit dst_stride (bit 63 of sizes[i]: nothing was written) becomes an empty, flagged entry.
__global__ __launch_bounds__(1024) void k_pack_index(const uint64_t* __restrict__ sizes, uint32_t nblocks, const uint32_t* __restrict__ nb_state,
                                                    const uint32_t* __restrict__ nbuse, uint8_t* __restrict__ packed,
                                                    uint64_t* __restrict__ total) {
    __shared__ uint64_t s_w[16];
    __shared__ uint64_t s_carry;
    __shared__ uint32_t s_bad;
    uint64_t* head = reinterpret_cast<uint64_t*>(packed);
    uint64_t* index = head + 4;
    const uint32_t tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    if (tid == 0) {
        s_carry = 0;
        s_bad = 0;
    }
    __syncthreads();
    for (uint32_t base = 0; base < nblocks; base += 1024) {
        const uint32_t i = base + tid;
        const uint64_t sz = i < nblocks ? sizes[i] : 0ull;
        const bool bad = (sz >> 63) != 0;
        const uint64_t len = bad ? 0ull : sz;
        if (bad) atomicAdd(&s_bad, 1u);
        const uint64_t v = (len + 15ull) & ~15ull;
        uint64_t inc = v;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
            const uint64_t o = (uint64_t)__shfl_up((long long)inc, dd, 64);
            if (l >= (uint32_t)dd) inc += o;
        }
        if (l == 63) s_w[w] = inc;
        __syncthreads();
        uint64_t pre = s_carry, tot = s_carry;
        for (uint32_t q = 0; q < 16; ++q) {
            if (q < w) pre += s_w[q];
            tot += s_w[q];
        }
        if (i < nblocks) {
            index[2 * i] = pre + inc - v;
            index[2 * i + 1] = len | ((uint64_t)(nbuse[i] & 0xFu) << 56) | (bad ? (1ull << 63) : 0ull);
        }
        __syncthreads();
        if (tid == 0) s_carry = tot;
        __syncthreads();
    }
    if (tid == 0) {
        head[0] = kPackMagic;
        head[1] = nblocks;
        head[2] = s_carry;
        head[3] = (uint64_t)*nb_state | ((uint64_t)s_bad << 32);
        *total = kPackHead + 16ull * nblocks + s_carry;
    }
}

// grid (chunks, nblocks): 16-byte units of stream b, strided over the chunk workgroups
__global__ __launch_bounds__(256) void k_pack_copy(const uint8_t* __restrict__ dst, uint64_t dst_stride, uint32_t nblocks,
                                                  uint8_t* __restrict__ packed) {
    const uint32_t b = blockIdx.y;
    const uint64_t* index = reinterpret_cast<const uint64_t*>(packed) + 4;
    const uint64_t off = index[2 * b], len = index[2 * b + 1] & kPackLenMask;  // (0 for a flagged stream: nothing to copy)
    const uint8_t* s = dst + (size_t)b * dst_stride;
    uint8_t* o = packed + kPackHead + 16ull * nblocks + off;
    const uint64_t units = (len + 15) >> 4;
    for (uint64_t u = (uint64_t)blockIdx.x * 256 + threadIdx.x; u < units; u += (uint64_t)gridDim.x * 256) {
        uint4 v = *reinterpret_cast<const uint4*>(s + u * 16);
        if (u * 16 + 16 > len) {  // zero the padding so that containers are reproducible
            const uint32_t keep = (uint32_t)(len - u * 16);
            uint32_t x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) {
                const uint32_t lo = q * 4;
                x[q] = keep >= lo + 4 ? x[q] : keep <= lo ? 0u : (x[q] & ((1u << ((keep - lo) * 8)) - 1u));
            }
            v = make_uint4(x[0], x[1], x[2], x[3]);
        }
        *reinterpret_cast<uint4*>(o + u * 16) = v;
    }
}

}  // namespace rspt
