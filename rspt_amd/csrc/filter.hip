// filter.hip -- the IIR pre-filter stage in front of the packers.
//
// Restates i_filter::new_iir / init_history_values / filter / filter_opt of lib_rspt/lib_filter/iir_filter.cpp:46-116 as the
// reference's test harness drives them (lib_rspt_test/rspt_test.cpp:116-136): per channel, 4 * nr_samples copies of the
// channel's first sample through filter(), then filter_opt() on every sample, the double result truncated to int32 and
// written back in the native sample width.  Double arithmetic in the reference's order of operations, every product and sum
// rounded on its own (plain operators under `#pragma clang fp contract(off)`, see below: no fused multiply-add, as in the reference's x86-64 build), so the filtered
// block is bit-identical with the reference's.
//
// Two modes, because the harness shares ONE filter object between the channels and its state runs on from channel to channel
// (the history initialisation damps the old state by ~e^-10, it does not erase it: on the 24-bit test recording the carried
// state moves a third of the samples, by up to 2569 counts):
//   shared       bit-exact with the harness; the channels of a block are a serial chain, so one thread takes one block
//   per channel  a fresh filter per channel (what a caller with one i_filter per channel gets): one thread per channel,
//                lane <-> channel so that every wave access is a contiguous row segment of the interleaved block
#include "common.hpp"

// NO contraction in this file.  hipcc's default is -ffp-contract=fast, and HIP's __dmul_rn / __dadd_rn are plain operators (not
// the contraction barriers their CUDA namesakes are): left alone, the compiler fuses the recurrence's products and sums into
// v_fma_f64 (83 of them in the round-2 kernel).  One fused rounding is ~1e-16 relative -- but this band-pass has poles at
// 0.9994 and coefficients that cancel (3.14 y1 - 3.70 y2 + 1.97 y3 - 0.41 y4), which amplifies it to ~1e-7 absolute, enough to
// move the truncated output by one count about once in 2 million samples (found on the 64 x 65536 bench batch; the small
// fixtures never hit it).  With contraction off every product and sum is rounded on its own, as in the reference's x86-64 build.
#pragma clang fp contract(off)

namespace rspt {

struct IirCoef {
    double n[5], d[5];  // feedback (n[0] unused) and feed-forward coefficients
    uint32_t nc;        // 2..5
    int32_t init_steps;  // 4 * nr_samples of init_history_values (iir_filter.cpp:106-110)
};

// sample access: one load / store per sample where the block's base allows it (aligned = the block base is a multiple of 4 for
// int32, of 2 for int16; rows are then aligned too), bytes otherwise and for int24 / int8
template <int BPS>
__device__ __forceinline__ int32_t iir_load(const uint8_t* p, bool aligned) {
    if (BPS == 4 && aligned) return *reinterpret_cast<const int32_t*>(p);
    if (BPS == 2 && aligned) return (int32_t) * reinterpret_cast<const int16_t*>(p);
    if (BPS == 4) return (int32_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24));
    if (BPS == 3) return (int32_t)(((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16)) << 8) >> 8;
    if (BPS == 2) return (int32_t)(int16_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8));
    return (int32_t)(int8_t)p[0];
}
template <int BPS>
__device__ __forceinline__ void iir_store(uint8_t* p, int32_t v, bool aligned) {
    if (BPS == 4 && aligned) {
        *reinterpret_cast<int32_t*>(p) = v;
        return;
    }
    if (BPS == 2 && aligned) {
        *reinterpret_cast<int16_t*>(p) = (int16_t)v;
        return;
    }
#pragma unroll
    for (int k = 0; k < BPS; ++k) p[k] = (uint8_t)((uint32_t)v >> (8 * k));
}

// The filter state: x[i] = input i samples ago, y[i] = output i samples ago (x_ring_ / y_ring_ of iir_filter.cpp:46-62).
template <int NC>
struct IirState {
    double x[NC], y[NC];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int i = 0; i < NC; ++i) x[i] = y[i] = 0.0;
    }
    __device__ __forceinline__ void shift(double in) {  // iir_filter.cpp:66-71
#pragma unroll
        for (int i = NC - 1; i > 0; --i) {
            x[i] = x[i - 1];
            y[i] = y[i - 1];
        }
        x[0] = in;
    }
    // i_filter::filter (iir_filter.cpp:64-77): the terms join the sum one by one, feed-forward and feedback interleaved
    __device__ __forceinline__ double step(const IirCoef& c, double in) {
        shift(in);
        double acc = (c.d[0] * x[0]);
#pragma unroll
        for (int i = 1; i < NC; ++i) {
            acc = (acc + (c.d[i] * x[i]));
            acc = (acc - (c.n[i] * y[i]));
        }
        y[0] = acc;
        return acc;
    }
    // the same step once the whole x ring holds one value (init_history_values feeds a constant): the feed-forward products
    // d[i] * x[i] are the same numbers every time -- P[i], computed once -- and only the feedback products are new
    __device__ __forceinline__ void step_const(const IirCoef& c, const double (&P)[NC]) {
#pragma unroll
        for (int i = NC - 1; i > 0; --i) y[i] = y[i - 1];
        double acc = P[0];
#pragma unroll
        for (int i = 1; i < NC; ++i) {
            acc = (acc + P[i]);
            acc = (acc - (c.n[i] * y[i]));
        }
        y[0] = acc;
    }
};

// One channel: history initialisation with its first sample, then every sample in place.
// filter_opt (iir_filter.cpp:79-104 with :23-41) is ONE expression evaluated left to right: all feed-forward terms first,
//     ff = (((d0 x0 + d1 x1) + d2 x2) + d3 x3) + d4 x4            -- no output in it: computed for a whole chunk ahead of time
//     y  = (((ff - n1 y1) - n2 y2) - n3 y3) - n4 y4                -- the serial part: one product and NC-1 subtractions per sample
// Every product and sum is rounded on its own, in the reference's order.  Samples are handled in chunks of CH: the next
// chunk's loads are in flight while this one is filtered, the feed-forward sums of the chunk are independent work the
// scheduler places into the latency of the dependent chain, and the results leave as one store per sample.
template <int BPS, int NC>
__device__ __forceinline__ void iir_channel(uint8_t* p, size_t stride, uint32_t ns, const IirCoef& c, IirState<NC>& f, bool aligned) {
    const double x0 = (double)iir_load<BPS>(p, aligned);
    {
        int32_t i = 0;
        for (; i < c.init_steps && i < NC - 1; ++i) f.step(c, x0);  // (the x ring still holds older inputs)
        if (i < c.init_steps) {
            f.step(c, x0);  // this one fills the ring's last place
            ++i;
            double P[NC];
#pragma unroll
            for (int k = 0; k < NC; ++k) P[k] = (c.d[k] * x0);
            for (; i < c.init_steps; ++i) f.step_const(c, P);
        }
    }
    constexpr uint32_t CH = 16;
    int32_t cur[CH], nxt[CH];
    const uint32_t nfull = ns / CH;
    if (nfull) {
#pragma unroll
        for (uint32_t e = 0; e < CH; ++e) cur[e] = iir_load<BPS>(p + (size_t)e * stride, aligned);
    }
    for (uint32_t k = 0; k < nfull; ++k) {
        uint8_t* q = p + (size_t)k * CH * stride;
        if (k + 1 < nfull) {
#pragma unroll
            for (uint32_t e = 0; e < CH; ++e) nxt[e] = iir_load<BPS>(q + (size_t)(CH + e) * stride, aligned);
        }
        // feed-forward sums of the chunk (xs[e + NC - 1] = sample e of the chunk, the NC - 1 values in front come from the state)
        double xs[CH + NC - 1], ff[CH];
#pragma unroll
        for (int i = 0; i < NC - 1; ++i) xs[i] = f.x[NC - 2 - i];
#pragma unroll
        for (uint32_t e = 0; e < CH; ++e) xs[NC - 1 + e] = (double)cur[e];
#pragma unroll
        for (uint32_t e = 0; e < CH; ++e) {
            double a = (c.d[0] * xs[NC - 1 + e]);
#pragma unroll
            for (int i = 1; i < NC; ++i) a = (a + (c.d[i] * xs[NC - 1 + e - i]));
            ff[e] = a;
        }
        // the recurrence, then one store per sample
        int32_t out[CH];
#pragma unroll
        for (uint32_t e = 0; e < CH; ++e) {
            double a = ff[e];
#pragma unroll
            for (int i = 1; i < NC; ++i) a = (a - (c.n[i] * f.y[i - 1]));  // (y[i-1] now = y[i] of the step being taken)
#pragma unroll
            for (int i = NC - 1; i > 0; --i) f.y[i] = f.y[i - 1];
            f.y[0] = a;
            out[e] = trunc_i32_c(a);  // C truncation (rspt_test.cpp:130)
        }
#pragma unroll
        for (int i = 0; i < NC; ++i) f.x[i] = xs[CH + NC - 2 - i];  // the last NC inputs, newest first
#pragma unroll
        for (uint32_t e = 0; e < CH; ++e) iir_store<BPS>(q + (size_t)e * stride, out[e], aligned);
#pragma unroll
        for (uint32_t e = 0; e < CH; ++e) cur[e] = nxt[e];
    }
    for (uint32_t s = nfull * CH; s < ns; ++s) {  // the tail, sample by sample
        uint8_t* q = p + (size_t)s * stride;
        f.shift((double)iir_load<BPS>(q, aligned));
        double a = (c.d[0] * f.x[0]);
#pragma unroll
        for (int i = 1; i < NC; ++i) a = (a + (c.d[i] * f.x[i]));
#pragma unroll
        for (int i = 1; i < NC; ++i) a = (a - (c.n[i] * f.y[i]));
        f.y[0] = a;
        iir_store<BPS>(q, trunc_i32_c(a), aligned);
    }
}

template <int BPS, int NC, bool SHARED>
__global__ __launch_bounds__(64) void k_iir(uint8_t* __restrict__ buf, uint32_t nch, uint32_t ns, uint64_t block_bytes, IirCoef c, uint32_t nblocks) {
    const uint32_t t = blockIdx.x * 64u + threadIdx.x;
    IirState<NC> f;
    f.clear();
    const size_t stride = (size_t)nch * BPS;
    // (wave-uniform: every block base and every row start is aligned when the first one is and the sizes are multiples)
    const bool aligned = (BPS == 4 || BPS == 2) && (reinterpret_cast<uintptr_t>(buf) % BPS) == 0 && (block_bytes % BPS) == 0;
    if (SHARED) {  // one filter object for all channels of the block, as in the harness
        if (t >= nblocks) return;
        for (uint32_t ch = 0; ch < nch; ++ch) iir_channel<BPS, NC>(buf + (size_t)t * block_bytes + (size_t)ch * BPS, stride, ns, c, f, aligned);
    } else {
        const uint32_t b = t / nch, ch = t - b * nch;
        if (b >= nblocks) return;
        iir_channel<BPS, NC>(buf + (size_t)b * block_bytes + (size_t)ch * BPS, stride, ns, c, f, aligned);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The pipelined form.  A lone wave issues one instruction every ~2 ns whatever it is (tools/issue_rate.hip), so the time of a
// channel is its sample count times the instructions the wave that HOLDS THE FILTER STATE has to issue per sample.  k_iir above
// issues everything from that wave (load, conversion, 2 NC - 1 feed-forward operations, 2 (NC - 1) feedback operations,
// conversion, store: ~22 instructions, 6.6 ms for the 64-block batch).  Here a workgroup of six waves splits the work, lane <->
// channel (or lane <-> block in shared mode) in all of them:
//   wave 0      the recurrence alone: per sample one LDS read of the feed-forward sum, NC - 1 products, NC - 1 subtractions, the
//               truncation and one LDS write -- and the history initialisation at the start of every channel
//   waves 1-4   load the samples (two chunks ahead), convert them and form the feed-forward sums of a quarter chunk each
//               (every product and sum rounded on its own, in the reference's left-to-right order) into LDS
//   wave 5      stores the filtered samples of the chunk before
// One workgroup barrier per chunk of 64 samples; the chunk being produced, the one in the recurrence and the one being stored
// live in double-buffered LDS tiles [sample][lane] (conflict-free).  Needs ns >= 64 and init_steps >= NC - 1 (else k_iir).
constexpr uint32_t kIirChunk = 64, kIirProd = 4, kIirPart = kIirChunk / kIirProd;  // four producer waves, 16 samples of a chunk each
constexpr uint32_t kIirThreads = 64 * (2 + kIirProd);
struct IirPipeLds {
    double ff[2][kIirChunk][64];
    int32_t out[2][kIirChunk][64];
    double xlast[5][64];  // the channel's last inputs, newest first (shared mode: the x ring the next channel's initialisation starts from)
};
// ~98.5 KiB of static LDS: one workgroup per CU, and only on a part with more than 64 KiB per workgroup (gfx950: 160 KiB)
static_assert(sizeof(IirPipeLds) <= 160 * 1024, "k_iir_pipe: the tiles must fit one CU's LDS");

template <int BPS, int NC, bool SHARED, bool ALIGNED>
__global__ __launch_bounds__(kIirThreads) void k_iir_pipe(uint8_t* __restrict__ buf, uint32_t nch, uint32_t ns, uint64_t block_bytes, IirCoef c, uint32_t nblocks,
                                                 uint32_t lanes_per_wg) {
    static_assert(NC >= 2 && NC <= 5, "IirPipeLds::xlast holds five inputs per channel");
    __shared__ IirPipeLds L;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t role = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // lanes_per_wg < 64 spreads the lanes over more workgroups: in shared mode a lane is a BLOCK, its accesses are 16 MiB apart
    // from its neighbours', and 64 of them per CU would ask one CU's memory path for more than it can give
    const uint32_t unit = blockIdx.x * lanes_per_wg + lane;
    uint32_t b, ch0, nser;
    if (SHARED) {
        b = unit;
        ch0 = 0;
        nser = nch;
    } else {
        b = unit / nch;
        ch0 = unit - b * nch;
        nser = 1;
    }
    // six waves on four SIMDs: the wave with the recurrence shares its SIMD with a producer -- it goes first whenever it can issue
    if (role == 0u) __builtin_amdgcn_s_setprio(3);
    const bool valid = lane < lanes_per_wg && b < nblocks;  // (lanes past the batch read block 0 along with the others and store nothing)
    const size_t stride = (size_t)nch * BPS;
    constexpr bool aligned = ALIGNED;  // (the host has looked at the base address: one load / store instruction per sample)
    uint8_t* base = buf + (size_t)(valid ? b : 0u) * block_bytes + (size_t)(valid ? ch0 : 0u) * BPS;
    const uint32_t nchunks = (ns + kIirChunk - 1) / kIirChunk;
    IirState<NC> f;
    f.clear();
    constexpr int H = NC - 1;            // inputs in front of a sample that its feed-forward sum needs
    constexpr uint32_t SET = kIirPart + H;  // a producer's samples per chunk: its part and the H in front of it
    constexpr uint32_t kWriter = 1 + kIirProd;
    for (uint32_t sc = 0; sc < nser; ++sc) {
        uint8_t* p = base + (size_t)sc * BPS;
        const double x0 = (double)iir_load<BPS>(p, aligned);
        int32_t cur[SET], nxt[SET], nx2[SET];
        // element j of a producer's set in chunk t is sample t * 64 + (role - 1) * 16 - H + j; in front of the channel: the history (x0, see below)
        auto load_set = [&](int32_t (&v)[SET], uint32_t t) {
            const int32_t s0 = (int32_t)(t * kIirChunk + (role - 1u) * kIirPart) - H;
            if (s0 >= 0 && s0 + (int32_t)SET <= (int32_t)ns) {  // (wave-uniform; all but a channel's first and last sets)
                const uint8_t* q = p + (size_t)s0 * stride;
#pragma unroll
                for (uint32_t j = 0; j < SET; ++j) v[j] = iir_load<BPS>(q + (size_t)j * stride, aligned);
            } else {
#pragma unroll
                for (uint32_t j = 0; j < SET; ++j) {
                    int32_t si = s0 + (int32_t)j;
                    si = si < 0 ? 0 : si >= (int32_t)ns ? (int32_t)ns - 1 : si;  // (clamped: what lies outside is never used as such)
                    v[j] = iir_load<BPS>(p + (size_t)si * stride, aligned);
                }
            }
        };
        const bool producer = role >= 1u && role <= kIirProd;
        if (producer) {
            load_set(cur, 0);
            if (nchunks > 1) load_set(nxt, 1);
        }
        for (uint32_t t = 0; t < nchunks + 2; ++t) {
            if (role == 0u) {
                if (t == 0) {
                    // init_history_values (iir_filter.cpp:106-110): 4 * nr_samples calls of filter() on the channel's first sample
                    int32_t i = 0;
                    for (; i < c.init_steps && i < NC; ++i) f.step(c, x0);  // (until the x ring holds nothing but x0)
                    if (i < c.init_steps) {
                        double P[NC];
#pragma unroll
                        for (int k = 0; k < NC; ++k) P[k] = c.d[k] * x0;
                        // (unrolled by the ring's length: the shifts of y become register names instead of moves)
#pragma unroll 4
                        for (; i < c.init_steps; ++i) f.step_const(c, P);
                    }
                } else if (t <= nchunks) {
                    const uint32_t k = t - 1, bi = k & 1u;
                    const uint32_t cnt = min(kIirChunk, ns - k * kIirChunk);
                    auto rec = [&](double a) -> double {
#if defined(IIR_PROBE) && IIR_PROBE == 1  // timing probe (never in the product): no recurrence
                        return a;
#endif
#pragma unroll
                        for (int i = 1; i < NC; ++i) a = a - c.n[i] * f.y[i - 1];  // (y[i-1] now = y[i] of the step being taken)
#pragma unroll
                        for (int i = NC - 1; i > 0; --i) f.y[i] = f.y[i - 1];
                        f.y[0] = a;
                        return a;
                    };
                    if (cnt == kIirChunk) {
                        // sixteen samples at a time: their feed-forward sums are read from LDS together (one wait), the results
                        // written together -- per sample the wave issues the recurrence and little else.  The truncation is the
                        // GPU's own (saturating) conversion here; C's as the reference's build does it (trunc_i32_c, common.hpp) differs
                        // from it only where the double is out of range, i.e. where this conversion returns INT_MAX: the largest
                        // result of the chunk is tracked (half an instruction per sample instead of two), and a chunk that
                        // reaches INT_MAX -- full-scale input through a filter that overshoots -- is done once more, exactly.
                        double ysave[NC];
#pragma unroll
                        for (int i = 0; i < NC; ++i) ysave[i] = f.y[i];
                        int32_t mx = (int32_t)0x80000000u;
#pragma unroll 1
                        for (uint32_t e0 = 0; e0 < kIirChunk; e0 += 16) {
                            double a[16];
                            int32_t o[16];
#pragma unroll
                            for (uint32_t e = 0; e < 16; ++e) a[e] = L.ff[bi][e0 + e][lane];
#pragma unroll
                            for (uint32_t e = 0; e < 16; ++e) o[e] = (int32_t)rec(a[e]);
#pragma unroll
                            for (uint32_t e = 0; e < 16; e += 2) mx = max(mx, max(o[e], o[e + 1]));  // (v_max3_i32)
#pragma unroll
                            for (uint32_t e = 0; e < 16; ++e) L.out[bi][e0 + e][lane] = o[e];
                        }
                        if (__builtin_expect(__builtin_amdgcn_ballot_w64(mx == 0x7FFFFFFF) != 0ull, 0)) {
#pragma unroll
                            for (int i = 0; i < NC; ++i) f.y[i] = ysave[i];
#pragma unroll 1
                            for (uint32_t e = 0; e < kIirChunk; ++e) L.out[bi][e][lane] = trunc_i32_c(rec(L.ff[bi][e][lane]));
                        }
                    } else {
                        for (uint32_t e = 0; e < cnt; ++e) L.out[bi][e][lane] = trunc_i32_c(rec(L.ff[bi][e][lane]));  // C truncation (rspt_test.cpp:130)
                    }
                    if (SHARED && t == nchunks) {  // the x ring the next channel's initialisation starts from (its first NC - 1 calls see it)
#pragma unroll
                        for (int i = 0; i < NC; ++i) f.x[i] = L.xlast[i][lane];
                    }
                }
            } else if (role == kWriter) {
                if (t >= 2) {
                    const uint32_t k = t - 2, bi = k & 1u;
                    const uint32_t cnt = min(kIirChunk, ns - k * kIirChunk);
                    uint8_t* q = p + (size_t)k * kIirChunk * stride;
                    if (cnt == kIirChunk) {
#pragma unroll 1
                        for (uint32_t e0 = 0; e0 < kIirChunk; e0 += 16) {
                            int32_t v[16];
#pragma unroll
                            for (uint32_t e = 0; e < 16; ++e) v[e] = L.out[bi][e0 + e][lane];
#if defined(IIR_PROBE) && IIR_PROBE == 3  // timing probe (never in the product): one store in sixteen
                            if (valid) iir_store<BPS>(q + (size_t)e0 * stride, v[0] ^ v[5] ^ v[15], aligned);
#else
                            if (valid) {
#pragma unroll
                                for (uint32_t e = 0; e < 16; ++e) iir_store<BPS>(q + (size_t)(e0 + e) * stride, v[e], aligned);
                            }
#endif
                        }
                    } else {
                        for (uint32_t e = 0; e < cnt; ++e) {
                            const int32_t v = L.out[bi][e][lane];
                            if (valid) iir_store<BPS>(q + (size_t)e * stride, v, aligned);
                        }
                    }
                }
            } else if (t < nchunks) {
                if (t + 2 < nchunks) load_set(nx2, t + 2);  // (two chunks ahead: the loads have two ticks to arrive)
                // the inputs as doubles; in front of the channel's first sample the x ring holds x0 (init_steps >= NC - 1: the host checks)
                const int32_t s0 = (int32_t)(t * kIirChunk + (role - 1u) * kIirPart) - H;
                double xs[SET];
#pragma unroll
                for (uint32_t j = 0; j < SET; ++j) xs[j] = (double)cur[j];
                if (s0 < 0) {  // (wave-uniform: the first producer's first set only)
#pragma unroll
                    for (uint32_t j = 0; j < SET; ++j) xs[j] = (s0 + (int32_t)j < 0) ? x0 : xs[j];
                }
#pragma unroll
                for (uint32_t e = 0; e < kIirPart; ++e) {
                    double a = c.d[0] * xs[H + e];
#if !(defined(IIR_PROBE) && IIR_PROBE == 2)  // timing probe (never in the product): no feed-forward sums
#pragma unroll
                    for (int i = 1; i < NC; ++i) a = a + c.d[i] * xs[H + e - i];
#endif
                    L.ff[t & 1u][(role - 1u) * kIirPart + e][lane] = a;
                }
                if (SHARED && t + 1 == nchunks) {  // whoever holds the channel's last sample hands its last inputs on
                    const int32_t last = (int32_t)ns - 1 - (s0 + H);  // index of sample ns-1 in this wave's part
                    if (last >= 0 && last < (int32_t)kIirPart) {
#pragma unroll
                        for (int i = 0; i < NC; ++i) {
                            double v = 0.0;
#pragma unroll
                            for (uint32_t j = 0; j < SET; ++j)
                                if ((int32_t)j == last + H - i) v = xs[j];
                            L.xlast[i][lane] = v;
                        }
                    }
                }
#pragma unroll
                for (uint32_t j = 0; j < SET; ++j) {
                    cur[j] = nxt[j];
                    nxt[j] = nx2[j];
                }
            }
            __syncthreads();
        }
    }
}

}  // namespace rspt

#pragma clang fp contract(fast)
