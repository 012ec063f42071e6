// filter.hip -- the IIR pre-filter stage in front of the packers.
//
// Restates i_filter::new_iir / init_history_values / filter / filter_opt of lib_rspt/lib_filter/iir_filter.cpp:46-116 as the
// reference's test harness drives them (lib_rspt_test/rspt_test.cpp:116-136): per channel, 4 * nr_samples copies of the
// channel's first sample through filter(), then filter_opt() on every sample, the double result truncated to int32 and
// written back in the native sample width.  Double arithmetic in the reference's order of operations, every product and sum
// rounded on its own (__dmul_rn / __dadd_rn: no fused multiply-add, as in the reference's x86-64 build), so the filtered
// block is bit-identical with the reference's.
//
// Two modes, because the harness shares ONE filter object between the channels and its state runs on from channel to channel
// (the history initialisation damps the old state by ~e^-10, it does not erase it: on the 24-bit test recording the carried
// state moves a third of the samples, by up to 2569 counts):
//   shared       bit-exact with the harness; the channels of a block are a serial chain, so one thread takes one block
//   per channel  a fresh filter per channel (what a caller with one i_filter per channel gets): one thread per channel,
//                lane <-> channel so that every wave access is a contiguous row segment of the interleaved block
#include "common.hpp"

namespace rspt {

struct IirCoef {
    double n[5], d[5];  // feedback (n[0] unused) and feed-forward coefficients
    uint32_t nc;        // 2..5
    int32_t init_steps;  // 4 * nr_samples of init_history_values (iir_filter.cpp:106-110)
};

template <int BPS>
__device__ __forceinline__ int32_t iir_load(const uint8_t* p) {
    if (BPS == 4) return (int32_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24));
    if (BPS == 3) return (int32_t)(((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16)) << 8) >> 8;
    if (BPS == 2) return (int32_t)(int16_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8));
    return (int32_t)(int8_t)p[0];
}
template <int BPS>
__device__ __forceinline__ void iir_store(uint8_t* p, int32_t v) {
#pragma unroll
    for (int k = 0; k < BPS; ++k) p[k] = (uint8_t)((uint32_t)v >> (8 * k));
}

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
        double acc = __dmul_rn(c.d[0], x[0]);
#pragma unroll
        for (int i = 1; i < NC; ++i) {
            acc = __dadd_rn(acc, __dmul_rn(c.d[i], x[i]));
            acc = __dsub_rn(acc, __dmul_rn(c.n[i], y[i]));
        }
        y[0] = acc;
        return acc;
    }
    // i_filter::filter_opt (iir_filter.cpp:79-104 with :23-41): one expression, left to right -- all feed-forward terms, then the feedback
    __device__ __forceinline__ double step_opt(const IirCoef& c, double in) {
        shift(in);
        double acc = __dmul_rn(c.d[0], x[0]);
#pragma unroll
        for (int i = 1; i < NC; ++i) acc = __dadd_rn(acc, __dmul_rn(c.d[i], x[i]));
#pragma unroll
        for (int i = 1; i < NC; ++i) acc = __dsub_rn(acc, __dmul_rn(c.n[i], y[i]));
        y[0] = acc;
        return acc;
    }
};

// one channel: history initialisation with its first sample, then every sample in place
template <int BPS, int NC>
__device__ __forceinline__ void iir_channel(uint8_t* p, size_t stride, uint32_t ns, const IirCoef& c, IirState<NC>& f) {
    const double x0 = (double)iir_load<BPS>(p);
    for (int32_t i = 0; i < c.init_steps; ++i) f.step(c, x0);
    constexpr uint32_t CH = 8;  // samples loaded ahead of the (serial) recurrence
    uint32_t s = 0;
    for (; s + CH <= ns; s += CH) {
        int32_t v[CH];
#pragma unroll
        for (uint32_t e = 0; e < CH; ++e) v[e] = iir_load<BPS>(p + (size_t)(s + e) * stride);
#pragma unroll
        for (uint32_t e = 0; e < CH; ++e) iir_store<BPS>(p + (size_t)(s + e) * stride, (int32_t)f.step_opt(c, (double)v[e]));  // C truncation (rspt_test.cpp:130)
    }
    for (; s < ns; ++s) iir_store<BPS>(p + (size_t)s * stride, (int32_t)f.step_opt(c, (double)iir_load<BPS>(p + (size_t)s * stride)));
}

template <int BPS, int NC, bool SHARED>
__global__ __launch_bounds__(64) void k_iir(uint8_t* __restrict__ buf, uint32_t nch, uint32_t ns, uint64_t block_bytes, IirCoef c, uint32_t nblocks) {
    const uint32_t t = blockIdx.x * 64u + threadIdx.x;
    IirState<NC> f;
    f.clear();
    const size_t stride = (size_t)nch * BPS;
    if (SHARED) {  // one filter object for all channels of the block, as in the harness
        if (t >= nblocks) return;
        for (uint32_t ch = 0; ch < nch; ++ch) iir_channel<BPS, NC>(buf + (size_t)t * block_bytes + (size_t)ch * BPS, stride, ns, c, f);
    } else {
        const uint32_t b = t / nch, ch = t - b * nch;
        if (b >= nblocks) return;
        iir_channel<BPS, NC>(buf + (size_t)b * block_bytes + (size_t)ch * BPS, stride, ns, c, f);
    }
}

}  // namespace rspt
