// transforms.hip -- per-channel transforms of the lossy packers.
//
//   k_fwht     mean removal + natural-order Walsh-Hadamard transform + truncating
//              divide by n (signal_packer_hadamard.cpp:57-72, lib_fwht/fwht.c:4-34,
//              utils.cpp:30-40).  All-integer, bit-exact with the reference.
//   k_dct      mean removal + dense DCT-II with the reference's float32 cosine
//              table, float products, sequential double accumulation and C
//              truncation (signal_packer_dct.cpp:60-87,102-116).  The table is
//              built on the host with libm exactly as the reference constructor
//              does, so the result is bit-exact where the reference can run.
//   k_idct     signal_packer_dct.cpp:89-100 (decompress side)
#include "common.hpp"

namespace rspt {

__device__ __forceinline__ long long wave_add_i64(long long v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// average_32 (utils.cpp:30-40): int64 sum, then `/= size_t` = UNSIGNED 64-bit division
__device__ __forceinline__ int32_t mean_from_sum(long long sum, uint32_t n) {
    return (int32_t)(long long)((unsigned long long)sum / (unsigned long long)n);
}

__device__ __forceinline__ void store_mean_hdr(uint8_t* means, const Geom& g, uint32_t b, uint32_t c, int32_t m) {
    uint8_t* h = means + (size_t)b * g.hdr_len + 3 * c;  // low 24 bits, LE (hadamard.cpp:73-79, dct.cpp:120-126)
    h[0] = (uint8_t)m;
    h[1] = (uint8_t)((uint32_t)m >> 8);
    h[2] = (uint8_t)((uint32_t)m >> 16);
}

__device__ __forceinline__ int32_t load_mean_hdr(const uint8_t* means, const Geom& g, uint32_t b, uint32_t c) {
    const uint8_t* h = means + (size_t)b * g.hdr_len + 3 * c;
    uint32_t u = (uint32_t)h[0] | ((uint32_t)h[1] << 8) | ((uint32_t)h[2] << 16);
    return (int32_t)(u << 8) >> 8;  // sign-extend 24 bits (hadamard.cpp:98-99)
}

// block-wide int64 sum of row[0..n) (1024 threads)
__device__ __forceinline__ long long block_sum_row(const int32_t* row, uint32_t n, long long* s_red) {
    long long s = 0;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) s += row[i];
    s = wave_add_i64(s);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = s;
    __syncthreads();
    long long t = 0;
    for (uint32_t i = 0; i < (blockDim.x >> 6); ++i) t += s_red[i];
    __syncthreads();
    return t;
}

// One workgroup per (channel, block).  n = 2^k <= 65536.  The transform runs in
// LDS on at most 32768 points; for n = 65536 the top butterfly stage is folded
// into the load and the two halves go through LDS one after the other (stage
// order is free: the stages commute exactly in wrap-around arithmetic).
//   FORWARD  true : y = WHT(x - mean) / n, mean header written      (compress)
//            false: x = WHT(y) + mean (header read), no normalisation
//                   (decompress: fwht_normalize2 with ratio 1 is a no-op, fwht.c:36-40)
template <bool FORWARD>
__global__ __launch_bounds__(1024) void k_fwht(int32_t* __restrict__ planar, Geom g, uint8_t* __restrict__ means) {
    extern __shared__ __attribute__((aligned(16))) int32_t sh[];
    __shared__ long long s_red[16];
    const uint32_t tid = threadIdx.x;
    const uint32_t c = blockIdx.x, b = blockIdx.y;
    const uint32_t n = g.ns;
    int32_t* row = planar + (size_t)b * g.N + (size_t)c * n;
    const uint32_t k = 31u - (uint32_t)__builtin_clz(n);

    int32_t mean;
    if (FORWARD) {
        mean = mean_from_sum(block_sum_row(row, n, s_red), n);
        if (tid == 0) store_mean_hdr(means, g, b, c, mean);
    } else {
        mean = load_mean_hdr(means, g, b, c);
    }

    const uint32_t hn = n > 32768u ? 32768u : n;
    const uint32_t halves = n / hn;  // 1 or 2
    int32_t keep[32];                // results of half 0 while half 1's inputs are still being read
    for (uint32_t h = 0; h < halves; ++h) {
        for (uint32_t i = tid; i < hn; i += 1024) {
            uint32_t a = (uint32_t)row[i];
            if (halves == 2) {
                const uint32_t bb = (uint32_t)row[i + hn];
                a = h == 0 ? a + bb : a - bb;
            }
            sh[i] = (int32_t)a;
        }
        __syncthreads();
        if (h == 1) {
#pragma unroll
            for (int jj = 0; jj < 32; ++jj) row[tid + 1024u * jj] = keep[jj];
        }
        for (uint32_t w = hn >> 1; w >= 1; w >>= 1) {
            for (uint32_t q = tid; q < (hn >> 1); q += 1024) {
                const uint32_t lo = ((q & ~(w - 1)) << 1) | (q & (w - 1));
                const uint32_t x = (uint32_t)sh[lo], y = (uint32_t)sh[lo + w];
                sh[lo] = (int32_t)(x + y);  // (lo,hi) -> (lo+hi, lo-hi), fwht.c:19-22
                sh[lo + w] = (int32_t)(x - y);
            }
            __syncthreads();
        }
        auto finish = [&](uint32_t idx, int32_t y) -> int32_t {
            if (FORWARD) {
                // WHT(x - m) = WHT(x) - m*n*delta_0 (exact mod 2^32)
                if (idx == 0) y = (int32_t)((uint32_t)y - (uint32_t)mean * n);
                // fwht_normalize (fwht.c:30-34): int /= (n/1.0) = truncation toward zero
                return (y + ((y >> 31) & (int32_t)(n - 1))) >> k;
            }
            return (int32_t)((uint32_t)y + (uint32_t)mean);
        };
        if (halves == 2 && h == 0) {
#pragma unroll
            for (int jj = 0; jj < 32; ++jj) keep[jj] = finish(tid + 1024u * jj, sh[tid + 1024u * jj]);
        } else {
            for (uint32_t i = tid; i < hn; i += 1024) row[h * hn + i] = finish(h * hn + i, sh[i]);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// dense DCT-II / inverse, reference arithmetic.  thread <-> output index,
// CH channels per workgroup share every table load.
//   tab   [n][n] float: forward uses COS[x][i] (row x contiguous in i);
//         the inverse is given the transposed table so that its reads coalesce too.
// ---------------------------------------------------------------------------
constexpr int kDctCh = 4;
constexpr uint32_t kDctChunk = 1024;

template <bool FORWARD>
__global__ __launch_bounds__(256) void k_dct(const int32_t* __restrict__ in, Geom g, uint8_t* __restrict__ means,
                                            const float* __restrict__ tab, double scale0, double scale1, float cs0,
                                            int32_t* __restrict__ out) {
    __shared__ float s_src[kDctCh][kDctChunk];
    __shared__ long long s_red[4];
    __shared__ int32_t s_mean[kDctCh];
    const uint32_t tid = threadIdx.x;
    const uint32_t n = g.ns;
    const uint32_t i = blockIdx.x * 256 + tid;
    const uint32_t c0 = blockIdx.y * kDctCh;
    const uint32_t b = blockIdx.z;
    const uint32_t nc = min((uint32_t)kDctCh, g.nch - c0);

    if (FORWARD) {
        for (uint32_t cc = 0; cc < nc; ++cc) {
            long long t = block_sum_row(in + (size_t)b * g.N + (size_t)(c0 + cc) * n, n, s_red);
            if (tid == 0) {
                const int32_t m = mean_from_sum(t, n);
                s_mean[cc] = m;
                if (blockIdx.x == 0) store_mean_hdr(means, g, b, c0 + cc, m);
            }
        }
    } else if (tid < nc) {
        s_mean[tid] = load_mean_hdr(means, g, b, c0 + tid);
    }
    __syncthreads();

    double sum[kDctCh] = {0, 0, 0, 0};
    for (uint32_t x0 = 0; x0 < n; x0 += kDctChunk) {
        const uint32_t cn = min(kDctChunk, n - x0);
        for (uint32_t cc = 0; cc < nc; ++cc)
            for (uint32_t x = tid; x < cn; x += 256) {
                int32_t v = in[(size_t)b * g.N + (size_t)(c0 + cc) * n + x0 + x];
                float f;
                if (FORWARD) {
                    v = (int32_t)((uint32_t)v - (uint32_t)s_mean[cc]);  // offset_32(-mean), dct.cpp:108-109
                    f = (float)v;                                        // `int * float`: the int converts to float
                } else {
                    f = (float)v;
                    if (x0 + x == 0) f = __fmul_rn(cs0, f);  // Cs[x]*dct[x] in float, Cs[0]=(float)(1/sqrt 2) (dct.cpp:95)
                }
                s_src[cc][x] = f;
            }
        __syncthreads();
        if (i < n) {
            for (uint32_t x = 0; x < cn; ++x) {
                const float cv = tab[(size_t)(x0 + x) * n + i];
#pragma unroll
                for (int cc = 0; cc < kDctCh; ++cc) {
                    // float product, then sequential accumulation in double, no contraction
                    sum[cc] = __dadd_rn(sum[cc], (double)__fmul_rn(s_src[cc][x], cv));
                }
            }
        }
        __syncthreads();
    }
    if (i < n) {
        for (uint32_t cc = 0; cc < nc; ++cc) {
            double s;
            if (FORWARD)
                s = __dmul_rn(sum[cc], i == 0 ? scale0 : scale1);  // sum *= Cs[i]*sqrt(2/n)/128 (dct.cpp:84)
            else
                s = __dmul_rn(sum[cc], scale1);                    // sum *= sqrt(2/n)*128 (dct.cpp:97)
            int32_t r = (int32_t)s;                                  // C truncation (dct.cpp:85,98)
            if (!FORWARD) r = (int32_t)((uint32_t)r + (uint32_t)s_mean[cc]);
            out[(size_t)b * g.N + (size_t)(c0 + cc) * n + i] = r;
        }
    }
}

template __global__ void k_fwht<true>(int32_t*, Geom, uint8_t*);
template __global__ void k_fwht<false>(int32_t*, Geom, uint8_t*);
template __global__ void k_dct<true>(const int32_t*, Geom, uint8_t*, const float*, double, double, float, int32_t*);
template __global__ void k_dct<false>(const int32_t*, Geom, uint8_t*, const float*, double, double, float, int32_t*);

}  // namespace rspt
