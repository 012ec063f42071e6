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
        // (lo,hi) -> (lo+hi, lo-hi) per stage (fwht.c:19-22); the stages commute exactly in wrap-around arithmetic, so
        // three of them are taken at a time on eight values held in registers: a third of the LDS traffic and barriers
        uint32_t w = hn >> 1;
        while (w >= 4) {  // stages w, w/2, w/4 together: elements base + {0..7} * (w/4)
            const uint32_t st = w >> 2;
            for (uint32_t q = tid; q < (hn >> 3); q += 1024) {
                const uint32_t base = ((q & ~(st - 1)) << 3) | (q & (st - 1));
                uint32_t v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = (uint32_t)sh[base + i * st];
#pragma unroll
                for (int d = 4; d >= 1; d >>= 1) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        if (!(i & d)) {
                            const uint32_t x = v[i], y = v[i + d];
                            v[i] = x + y;
                            v[i + d] = x - y;
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) sh[base + i * st] = (int32_t)v[i];
            }
            __syncthreads();
            w >>= 3;
        }
        for (; w >= 1; w >>= 1) {  // the one or two stages left
            for (uint32_t q = tid; q < (hn >> 1); q += 1024) {
                const uint32_t lo = ((q & ~(w - 1)) << 1) | (q & (w - 1));
                const uint32_t x = (uint32_t)sh[lo], y = (uint32_t)sh[lo + w];
                sh[lo] = (int32_t)(x + y);
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

// ---- rows longer than 65536 points (fwht.c:4-28 transforms any n = 2^k): two passes over the planar row ------------------------
// WHT_n = WHT_{n / 32768} (x) WHT_32768 -- the stages act on one index bit each and commute exactly in wrap-around arithmetic.
//   k_fwht_seg    the low 15 index bits: every contiguous 32768-point piece of the row through LDS, in place, nothing else
//   k_fwht_cross  the high bits: thread <-> column j of the [n / 32768][32768] view, its m <= 64 values (stride `stride` points)
//                 in registers, butterflies, and -- on the last pass -- what k_fwht does at the end: forward WHT(x - mean) =
//                 WHT(x) - mean * n * delta_0 and the truncating division by n (fwht.c:30-34), inverse + mean.
//                 n = 2^22 takes two cross passes (64 x 2).
__global__ __launch_bounds__(1024) void k_fwht_seg(int32_t* __restrict__ planar, Geom g) {
    extern __shared__ __attribute__((aligned(16))) int32_t sh[];
    const uint32_t tid = threadIdx.x;
    constexpr uint32_t hn = 32768u;
    int32_t* seg = planar + (size_t)blockIdx.z * g.N + (size_t)blockIdx.y * g.ns + (size_t)blockIdx.x * hn;
    for (uint32_t i = tid; i < hn / 4; i += 1024) reinterpret_cast<int4*>(sh)[i] = reinterpret_cast<const int4*>(seg)[i];
    __syncthreads();
    uint32_t w = hn >> 1;
    while (w >= 4) {  // three stages at a time on eight values in registers (as in k_fwht)
        const uint32_t st = w >> 2;
        for (uint32_t q = tid; q < (hn >> 3); q += 1024) {
            const uint32_t base = ((q & ~(st - 1)) << 3) | (q & (st - 1));
            uint32_t v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = (uint32_t)sh[base + i * st];
#pragma unroll
            for (int d = 4; d >= 1; d >>= 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (!(i & d)) {
                        const uint32_t x = v[i], y = v[i + d];
                        v[i] = x + y;
                        v[i + d] = x - y;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) sh[base + i * st] = (int32_t)v[i];
        }
        __syncthreads();
        w >>= 3;
    }
    // (15 stages = five rounds of three: none left over)
    for (uint32_t i = tid; i < hn / 4; i += 1024) reinterpret_cast<int4*>(seg)[i] = reinterpret_cast<const int4*>(sh)[i];
}

template <bool FORWARD, int M>
__device__ __forceinline__ void fwht_cross_body(int32_t* __restrict__ row, uint32_t col, uint32_t stride, uint32_t group_span, bool last, uint32_t n,
                                                int32_t mean) {
    // the M values of this column: row[base + i * stride], base = (col / stride) * group_span + col % stride
    const uint32_t base = (col / stride) * group_span + (col % stride);
    uint32_t v[M];
#pragma unroll
    for (int i = 0; i < M; ++i) v[i] = (uint32_t)row[base + (size_t)i * stride];
#pragma unroll
    for (int d = M / 2; d >= 1; d >>= 1) {
#pragma unroll
        for (int i = 0; i < M; ++i) {
            if (!(i & d)) {
                const uint32_t x = v[i], y = v[i + d];
                v[i] = x + y;
                v[i + d] = x - y;
            }
        }
    }
    const uint32_t k = 31u - (uint32_t)__builtin_clz(n);
#pragma unroll
    for (int i = 0; i < M; ++i) {
        int32_t y = (int32_t)v[i];
        if (last) {
            if (FORWARD) {
                if (base + (size_t)i * stride == 0) y = (int32_t)((uint32_t)y - (uint32_t)mean * n);  // WHT(x - m) = WHT(x) - m*n*delta_0 (mod 2^32)
                y = (y + ((y >> 31) & (int32_t)(n - 1))) >> k;                                          // truncation toward zero (fwht.c:30-34)
            } else {
                y = (int32_t)((uint32_t)y + (uint32_t)mean);
            }
        }
        row[base + (size_t)i * stride] = y;
    }
}

// grid (n / m / 256, nch, blocks); m = values per column of this pass, `stride` = their distance in points
// the mean: forward all 32 bits of it (mean_i32, k_row_means: the subtraction uses them all, hadamard.cpp:62-66), inverse the 24
// bits the header kept (hadamard.cpp:98-99)
template <bool FORWARD>
__global__ __launch_bounds__(256) void k_fwht_cross(int32_t* __restrict__ planar, Geom g, const uint8_t* __restrict__ means,
                                                   const int32_t* __restrict__ mean_i32, uint32_t m, uint32_t stride, uint32_t last) {
    const uint32_t c = blockIdx.y, b = blockIdx.z, n = g.ns;
    int32_t* row = planar + (size_t)b * g.N + (size_t)c * n;
    const uint32_t col = blockIdx.x * 256u + threadIdx.x;  // < n / m
    const int32_t mean = !last ? 0 : FORWARD ? mean_i32[(size_t)b * g.nch + c] : load_mean_hdr(means, g, b, c);
    const uint32_t span = stride * m;
    switch (m) {
        case 2: fwht_cross_body<FORWARD, 2>(row, col, stride, span, last != 0, n, mean); break;
        case 4: fwht_cross_body<FORWARD, 4>(row, col, stride, span, last != 0, n, mean); break;
        case 8: fwht_cross_body<FORWARD, 8>(row, col, stride, span, last != 0, n, mean); break;
        case 16: fwht_cross_body<FORWARD, 16>(row, col, stride, span, last != 0, n, mean); break;
        case 32: fwht_cross_body<FORWARD, 32>(row, col, stride, span, last != 0, n, mean); break;
        default: fwht_cross_body<FORWARD, 64>(row, col, stride, span, last != 0, n, mean); break;
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

// ---- 65536-point rows: the whole row lives in registers --------------------------------------------------------------
// The stages (lo,hi) -> (lo+hi, lo-hi) act on one index bit each and commute exactly in wrap-around arithmetic, so the
// 16 of them are taken as three groups of index bits, each group inside the registers of a thread (64 values = 6 bits):
//   layout 1   regs = bits {15..12, 1, 0}   thread = bits 11..2     (16-byte loads, 1 KiB per wave instruction)
//   layout 2   regs = bits 7..2             thread = bits {15..12, 11..8, 1..0}
//   layout 3   regs = bits {11..8, 1, 0}    thread = bits {15..12, 7..2}   (16-byte stores, 1 KiB per wave instruction)
// with two transposes through a 128 KiB LDS image, one half of the row (bit 15) at a time.  The row is read once and
// written once (k_fwht's general form reads it three times: the sum, then both halves for each half).  Element i' of a
// half sits at word i' ^ (((i' >> 8) & 7) << 2): 16-byte accesses by consecutive lanes stay conflict-free (the flipped
// bits are constant within an instruction), and the dword accesses of layout 2 -- lanes 4 words apart in blocks 256
// words apart -- spread over all 32 banks.
__device__ __forceinline__ uint32_t fwht_swz(uint32_t a) { return a ^ (((a >> 8) & 7u) << 2); }

template <int NBITS_LO, int NBITS>
__device__ __forceinline__ void fwht_regs(uint32_t (&v)[64]) {  // butterflies over register-index bits [NBITS_LO, NBITS)
#pragma unroll
    for (int d = 1 << NBITS_LO; d < (1 << NBITS); d <<= 1) {
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            if (!(i & d)) {
                const uint32_t x = v[i], y = v[i + d];
                v[i] = x + y;
                v[i + d] = x - y;
            }
        }
    }
}

// PLANES (compress): the normalised coefficients leave as byte planes + non-zero map straight away (what k_planar_planes
// would do in another pass over HBM); a wave's 4096 outputs are exactly one 4 KiB segment of every plane.
template <bool FORWARD, bool PLANES>
__global__ __launch_bounds__(1024) void k_fwht64k(int32_t* __restrict__ planar, Geom g, uint8_t* __restrict__ means, uint8_t* __restrict__ planes,
                                                 uint32_t* __restrict__ nzflag, uint32_t nplanes) {
    extern __shared__ __attribute__((aligned(16))) int32_t sh_i[];
    uint32_t* sh = reinterpret_cast<uint32_t*>(sh_i);
    __shared__ long long s_red[16];
    const uint32_t tid = threadIdx.x;
    const uint32_t c = blockIdx.x, b = blockIdx.y;
    constexpr uint32_t n = 65536u, k = 16u;
    int32_t* row = planar + (size_t)b * g.N + (size_t)c * n;

    uint32_t v[64];  // layout 1: v[4 q + e] = row[4096 q + 4 tid + e]
#pragma unroll
    for (uint32_t q = 0; q < 16; ++q) {
        const uint4 x = reinterpret_cast<const uint4*>(row)[q * 1024 + tid];
        v[4 * q] = x.x;
        v[4 * q + 1] = x.y;
        v[4 * q + 2] = x.z;
        v[4 * q + 3] = x.w;
    }
    int32_t mean;
    if (FORWARD) {
        long long sum = 0;
#pragma unroll
        for (int i = 0; i < 64; ++i) sum += (int32_t)v[i];
        sum = wave_add_i64(sum);
        if ((tid & 63u) == 0) s_red[tid >> 6] = sum;
        __syncthreads();
        long long t = 0;
        for (uint32_t i = 0; i < 16; ++i) t += s_red[i];
        mean = mean_from_sum(t, n);
        if (tid == 0) store_mean_hdr(means, g, b, c, mean);
    } else {
        mean = load_mean_hdr(means, g, b, c);
    }
    fwht_regs<0, 6>(v);  // bits 0, 1, 12..15
    // (pinned: left alone the scheduler sinks the last stage's differences -- needed in the second round only -- behind the first
    //  round's LDS reads and keeps both of their inputs instead: 64 + 64 live registers and two dozen of them in scratch memory)
#pragma unroll
    for (int i = 32; i < 64; ++i) asm volatile("" : "+v"(v[i]));

    // transpose 1: layout 1 -> layout 2
    uint32_t w[64];
    const uint32_t half = tid >> 9;  // bit 15 of the elements this thread holds in layouts 2 and 3
    const uint32_t base2 = (((tid >> 6) & 7u) << 12) | (((tid >> 2) & 15u) << 8) | (tid & 3u);
    const uint32_t swz2 = (base2 >> 8) & 7u;  // what fwht_swz flips in bits 2..4 of this thread's layout-2 addresses
#pragma unroll
    for (uint32_t h = 0; h < 2; ++h) {
#pragma unroll
        for (uint32_t qq = 0; qq < 8; ++qq) {
            const uint32_t q = h * 8 + qq;
            *reinterpret_cast<uint4*>(&sh[fwht_swz((qq << 12) | (tid << 2))]) = make_uint4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
        }
        __syncthreads();
        if (half == h) {
            // fwht_swz(base2 | r << 2) = base2 + ((r & 7 ^ c) << 2) + (r >> 3 << 5) with c = bits 8..10 of base2: eight addresses and
            // immediate offsets (sixty-four computed addresses are sixty-four registers the three arrays do not leave)
#pragma unroll
            for (uint32_t n = 0; n < 8; ++n) {
                const uint32_t* pn = sh + (base2 | ((n ^ swz2) << 2));
#pragma unroll
                for (uint32_t m = 0; m < 8; ++m) w[8 * m + n] = pn[m << 5];
            }
        }
        __syncthreads();
    }
    fwht_regs<0, 6>(w);  // bits 2..7

    // transpose 2: layout 2 -> layout 3 (each half: written and read by the same 512 threads).  Read back INTO w: with an
    // array of its own the second round's stores still need w while the first round's loads are live -- 128 registers for the
    // compiler, which cannot know that no thread takes part in both rounds, and 22 of them went to scratch memory (104
    // scratch instructions per thread next to 64 loads and stores of data).
    uint32_t(&u)[64] = w;
    const uint32_t base3 = (((tid >> 6) & 7u) << 12) | ((tid & 63u) << 2);
#pragma unroll
    for (uint32_t h = 0; h < 2; ++h) {
        if (half == h) {
#pragma unroll
            for (uint32_t n = 0; n < 8; ++n) {
                uint32_t* pn = sh + (base2 | ((n ^ swz2) << 2));
#pragma unroll
                for (uint32_t m = 0; m < 8; ++m) pn[m << 5] = w[8 * m + n];
            }
        }
        __syncthreads();
        if (half == h) {
#pragma unroll
            for (uint32_t pp = 0; pp < 16; ++pp) {
                const uint4 x = *reinterpret_cast<const uint4*>(&sh[fwht_swz(base3 | (pp << 8))]);
                u[4 * pp] = x.x;
                u[4 * pp + 1] = x.y;
                u[4 * pp + 2] = x.z;
                u[4 * pp + 3] = x.w;
            }
        }
        __syncthreads();
    }
    fwht_regs<2, 6>(u);  // bits 8..11 (register-index bits 2..5; bits 0, 1 of the index are done)
#pragma unroll
    for (int i = 0; i < 64; ++i) asm volatile("" : "+v"(u[i]));  // (the output phase starts from 64 finished values, not in the middle of the butterflies)

    const uint32_t i0 = ((tid >> 6) << 12) | ((tid & 63u) << 2);  // bits 15..12 and 7..2
    uint32_t nzk[4] = {0, 0, 0, 0};
#pragma unroll
    for (uint32_t pp = 0; pp < 16; ++pp) {
        const uint32_t idx = i0 | (pp << 8);
        uint32_t o[4];
#pragma unroll
        for (uint32_t e = 0; e < 4; ++e) {
            int32_t y = (int32_t)u[4 * pp + e];
            if (FORWARD) {
                // WHT(x - m) = WHT(x) - m*n*delta_0 (exact mod 2^32); fwht_normalize (fwht.c:30-34): truncation toward zero
                if (idx + e == 0) y = (int32_t)((uint32_t)y - (uint32_t)mean * n);
                y = (y + ((y >> 31) & (int32_t)(n - 1))) >> k;
            } else {
                y = (int32_t)((uint32_t)y + (uint32_t)mean);
            }
            o[e] = (uint32_t)y;
        }
        if (!PLANES) {
            *reinterpret_cast<uint4*>(row + idx) = make_uint4(o[0], o[1], o[2], o[3]);
        } else {
            // 4 x 4 byte transpose (preprocess.hip: transform_item): one dword per plane
            const uint32_t lo01 = __builtin_amdgcn_perm(o[1], o[0], 0x05010400u), hi01 = __builtin_amdgcn_perm(o[1], o[0], 0x07030602u);
            const uint32_t lo23 = __builtin_amdgcn_perm(o[3], o[2], 0x05010400u), hi23 = __builtin_amdgcn_perm(o[3], o[2], 0x07030602u);
            const uint32_t pl[4] = {__builtin_amdgcn_perm(lo23, lo01, 0x05040100u), __builtin_amdgcn_perm(lo23, lo01, 0x07060302u),
                                    __builtin_amdgcn_perm(hi23, hi01, 0x05040100u), __builtin_amdgcn_perm(hi23, hi01, 0x07060302u)};
#pragma unroll
            for (uint32_t kk = 0; kk < 4; ++kk) {
                if (kk < nplanes) {
                    *reinterpret_cast<uint32_t*>(planes + ((size_t)b * kMaxPlanes + kk) * g.plane_stride + (size_t)c * n + idx) = pl[kk];
                    nzk[kk] |= pl[kk];
                }
            }
            asm volatile("" ::: "memory");  // (one quad of outputs at a time: interleaved, their temporaries spill)
        }
    }
    if (PLANES) {
        const uint32_t flat = c * n + i0;  // this wave's outputs: [flat & ~4095, +4096)
        for (uint32_t kk = 0; kk < nplanes; ++kk)
            if (__ballot(nzk[kk] != 0) && (tid & 63u) == 0) atomicOr(&nzflag[hb_index(g, b, kk, flat >> 16)], 1u << ((flat >> 12) & 15u));
    }
}

// n x n float matrix -> its transpose (the inverse DCT reads the cosine table the other way round), 32 x 32 tiles
__global__ __launch_bounds__(256) void k_transpose_f32(const float* __restrict__ a, float* __restrict__ t, uint32_t n) {
    __shared__ float s[32][33];
    const uint32_t tx = threadIdx.x & 31u, ty = threadIdx.x >> 5;  // 32 x 8
    const uint32_t x0 = blockIdx.x * 32u, y0 = blockIdx.y * 32u;
    for (uint32_t r = ty; r < 32u; r += 8u)
        if (y0 + r < n && x0 + tx < n) s[r][tx] = a[(size_t)(y0 + r) * n + x0 + tx];
    __syncthreads();
    for (uint32_t r = ty; r < 32u; r += 8u)
        if (x0 + r < n && y0 + tx < n) t[(size_t)(x0 + r) * n + y0 + tx] = s[tx][r];
}

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
            int32_t r = trunc_i32_c(s);                              // C truncation (dct.cpp:85,98)
            if (!FORWARD) r = (int32_t)((uint32_t)r + (uint32_t)s_mean[cc]);
            out[(size_t)b * g.N + (size_t)(c0 + cc) * n + i] = r;
        }
    }
}

// ---------------------------------------------------------------------------
// DCT for n = 2^k beyond the dense table (n > 8192; BASELINE config 4 is n = 65536,
// where the reference itself cannot run: 4n^2-byte table).  Same
// definition as signal_packer_dct.cpp:76-100 -- X[i] = trunc(sum_x s[x]*cos(pi(2x+1)i/2n)
// * Cs[i]*sqrt(2/n)/128) -- evaluated in fp64 through an n-point complex FFT
// (Makhoul's even/odd permutation), so the parity gate is SURVEY 8(d)'s PRDN / CR
// tolerance, not bit-exactness.
//
// Four-step FFT, n = n1*n2, input index j = j1*n2 + j2, output k = k1 + n1*k2:
//   k_dctfft_cols  for W adjacent columns j2: length-n1 FFT over j1 in LDS, times
//                  w_n^(j2*k1), to scratch[k1][j2]
//   k_dctfft_rows  for R adjacent rows k1: length-n2 FFT over j2 in LDS, then the
//                  DCT post-rotation (forward) or the inverse permutation.
// tw[t] = (cos, sin)(2*pi*t/n), post[k] = (cos, sin)(pi*k/(2n)), both from host libm.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_row_means(const int32_t* __restrict__ planar, Geom g, uint8_t* __restrict__ means,
                                                   int32_t* __restrict__ mean_i32) {
    __shared__ long long s_red[16];
    const uint32_t c = blockIdx.x, b = blockIdx.y;
    const long long t = block_sum_row(planar + (size_t)b * g.N + (size_t)c * g.ns, g.ns, s_red);
    if (threadIdx.x == 0) {
        const int32_t m = mean_from_sum(t, g.ns);
        store_mean_hdr(means, g, b, c, m);
        mean_i32[(size_t)b * g.nch + c] = m;  // the subtraction uses all 32 bits, the header keeps 24 (dct.cpp:108-109,120-126)
    }
}

// the same from channel sums that the de-interleave pass (k_tile_planar_i32x4) has already taken
__global__ __launch_bounds__(256) void k_means_from_sums(const long long* __restrict__ row_sum, Geom g, uint32_t nblocks, uint8_t* __restrict__ means,
                                                        int32_t* __restrict__ mean_i32) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nblocks * g.nch) return;
    const uint32_t b = i / g.nch, c = i - b * g.nch;
    const int32_t m = mean_from_sum(row_sum[i], g.ns);
    store_mean_hdr(means, g, b, c, m);
    mean_i32[i] = m;
}

__device__ __forceinline__ uint32_t bitrev(uint32_t v, uint32_t bits) { return __brev(v) >> (32u - bits); }

// radix-2 DIT over `len = 1<<logn` points of W = 1<<lw interleaved sequences: point p of sequence q at sh[p*W + q].
// stw[m] = (cos, sin)(2 pi m / len), m < len/2: the stage twiddles, staged in LDS by the caller (the butterflies would
// otherwise wait on a global load each).
__device__ __forceinline__ void lds_fft(double2* sh, const double2* stw, uint32_t logn, uint32_t lw, bool inverse) {
    const uint32_t W = 1u << lw;
    auto mul = [&](const double2& x, const double2& w) -> double2 {  // x * (w.x + i ws), ws = -/+ w.y
        const double ws = inverse ? w.y : -w.y;
        return make_double2(x.x * w.x - x.y * ws, x.x * ws + x.y * w.x);
    };
    uint32_t s = 1;
    // two stages at a time on four points in registers (half the LDS passes and barriers): stage s pairs (p0,p1), (p2,p3)
    // with twiddle index m; stage s+1 pairs (p0,p2) with m and (p1,p3) with m + half
    const uint32_t nq = (1u << (logn - 2)) << lw;
    for (; s + 1 <= logn; s += 2) {
        const uint32_t half = 1u << (s - 1);
        for (uint32_t q = threadIdx.x; q < nq; q += blockDim.x) {
            const uint32_t col = q & (W - 1), gq = q >> lw;
            const uint32_t m = gq & (half - 1);
            const uint32_t i0 = ((gq >> (s - 1)) << (s + 1)) + m;
            const double2 w1 = stw[m << (logn - s)], w2 = stw[m << (logn - s - 1)], w3 = stw[(m + half) << (logn - s - 1)];
            double2 p0 = sh[i0 * W + col], p1 = sh[(i0 + half) * W + col], p2 = sh[(i0 + 2 * half) * W + col], p3 = sh[(i0 + 3 * half) * W + col];
            const double2 t1 = mul(p1, w1), t3 = mul(p3, w1);
            const double2 a0 = make_double2(p0.x + t1.x, p0.y + t1.y), a1 = make_double2(p0.x - t1.x, p0.y - t1.y);
            const double2 a2 = make_double2(p2.x + t3.x, p2.y + t3.y), a3 = make_double2(p2.x - t3.x, p2.y - t3.y);
            const double2 u2 = mul(a2, w2), u3 = mul(a3, w3);
            sh[i0 * W + col] = make_double2(a0.x + u2.x, a0.y + u2.y);
            sh[(i0 + 2 * half) * W + col] = make_double2(a0.x - u2.x, a0.y - u2.y);
            sh[(i0 + half) * W + col] = make_double2(a1.x + u3.x, a1.y + u3.y);
            sh[(i0 + 3 * half) * W + col] = make_double2(a1.x - u3.x, a1.y - u3.y);
        }
        __syncthreads();
    }
    const uint32_t nbf = (1u << (logn - 1)) << lw;
    for (; s <= logn; ++s) {  // (odd logn: the last stage alone)
        const uint32_t half = 1u << (s - 1);
        for (uint32_t q = threadIdx.x; q < nbf; q += blockDim.x) {
            const uint32_t col = q & (W - 1), bf = q >> lw;
            const uint32_t m = bf & (half - 1);
            const uint32_t i0 = ((bf >> (s - 1)) << s) + m;
            const double2 a = sh[i0 * W + col];
            const double2 t = mul(sh[(i0 + half) * W + col], stw[m << (logn - s)]);
            sh[i0 * W + col] = make_double2(a.x + t.x, a.y + t.y);
            sh[(i0 + half) * W + col] = make_double2(a.x - t.x, a.y - t.y);
        }
        __syncthreads();
    }
}

// stage twiddle table of an FFT of length 1 << logn out of the n-point table (n = 1 << nlog)
__device__ __forceinline__ void load_stage_twiddles(double2* stw, const double2* __restrict__ tw, uint32_t logn, uint32_t nlog) {
    for (uint32_t m = threadIdx.x; m < (1u << (logn - 1)); m += blockDim.x) stw[m] = tw[(size_t)m << (nlog - logn)];
}

constexpr uint32_t kFftLdsLog = 12;  // 4096 complex fp64 points (64 KiB) per workgroup

template <bool FORWARD>
__global__ __launch_bounds__(1024) void k_dctfft_cols(const int32_t* __restrict__ in, Geom g, const int32_t* __restrict__ mean_i32,
                                                    const double2* __restrict__ tw, const double2* __restrict__ post,
                                                    double2* __restrict__ scratch, uint32_t l1, uint32_t l2, uint32_t b0, float cs0) {
    extern __shared__ __attribute__((aligned(16))) double2 shf[];
    const uint32_t n = g.ns, nlog = l1 + l2;
    const uint32_t lw = min(kFftLdsLog - l1, l2), W = 1u << lw;
    const uint32_t j2_0 = blockIdx.x << lw, c = blockIdx.y, bl = blockIdx.z, b = b0 + bl;
    const int32_t* row = in + (size_t)b * g.N + (size_t)c * n;
    const uint32_t total = 1u << (l1 + lw);
    double2* stw = shf + total;  // (the launcher sizes the dynamic LDS for both)
    load_stage_twiddles(stw, tw, l1, nlog);
    const int32_t mean = FORWARD ? mean_i32[(size_t)b * g.nch + c] : 0;
    for (uint32_t idx = threadIdx.x; idx < total; idx += blockDim.x) {
        const uint32_t col = idx & (W - 1), j1 = idx >> lw;
        const uint32_t j = (j1 << l2) + j2_0 + col;
        double2 v;
        if (FORWARD) {
            // v[j] = s[2j] (j < n/2), v[n-1-j] = s[2j+1]; s = (float)(src - mean) as in dct.cpp:80,108-109
            const uint32_t x = j < (n >> 1) ? 2u * j : 2u * (n - 1u - j) + 1u;
            const int32_t iv = (int32_t)((uint32_t)row[x] - (uint32_t)mean);
            v = make_double2((double)(float)iv, 0.0);
        } else {
            // y = A^T Y  <=>  DCT-II(y) = (n, n/2, n/2, ...) * Y;  V[k]/n = e^{+i pi k/2n} (Y'[k] - i Y'[n-k])
            double yk, ynk;
            if (j == 0) {
                yk = (double)__fmul_rn(cs0, (float)row[0]);  // Cs[0]*dct[0] in float (dct.cpp:95)
                ynk = 0.0;
            } else {
                yk = 0.5 * (double)(float)row[j];
                ynk = 0.5 * (double)(float)row[n - j];
            }
            const double2 pw = post[j];
            v = make_double2(pw.x * yk + pw.y * ynk, pw.y * yk - pw.x * ynk);
        }
        shf[bitrev(j1, l1) * W + col] = v;
    }
    __syncthreads();
    lds_fft(shf, stw, l1, lw, !FORWARD);
    double2* dst = scratch + ((size_t)bl * g.nch + c) * n;
    for (uint32_t idx = threadIdx.x; idx < total; idx += blockDim.x) {
        const uint32_t col = idx & (W - 1), k1 = idx >> lw;
        const uint32_t j2 = j2_0 + col;
        const double2 w = tw[j2 * k1];  // j2*k1 < n2*n1 = n
        const double ws = FORWARD ? -w.y : w.y;
        const double2 z = shf[k1 * W + col];
        dst[((size_t)k1 << l2) + j2] = make_double2(z.x * w.x - z.y * ws, z.x * ws + z.y * w.x);
    }
}

template <bool FORWARD>
__global__ __launch_bounds__(1024) void k_dctfft_rows(const double2* __restrict__ scratch, Geom g, const uint8_t* __restrict__ means,
                                                    const double2* __restrict__ tw, const double2* __restrict__ post,
                                                    int32_t* __restrict__ out, uint32_t l1, uint32_t l2, uint32_t b0, double scale0,
                                                    double scale1) {
    extern __shared__ __attribute__((aligned(16))) double2 shf[];
    const uint32_t n = g.ns, nlog = l1 + l2, n2 = 1u << l2;
    const uint32_t lr = min(kFftLdsLog - l2, l1), R = 1u << lr;
    const uint32_t k1_0 = blockIdx.x << lr, c = blockIdx.y, bl = blockIdx.z, b = b0 + bl;
    const uint32_t total = 1u << (l2 + lr);
    double2* stw = shf + total;
    load_stage_twiddles(stw, tw, l2, nlog);
    const double2* src = scratch + ((size_t)bl * g.nch + c) * n + ((size_t)k1_0 << l2);
    for (uint32_t idx = threadIdx.x; idx < total; idx += blockDim.x) {
        const uint32_t r = idx >> l2, j2 = idx & (n2 - 1);
        shf[bitrev(j2, l2) * R + r] = src[idx];
    }
    __syncthreads();
    lds_fft(shf, stw, l2, lr, !FORWARD);
    int32_t* orow = out + (size_t)b * g.N + (size_t)c * n;
    const int32_t mean = FORWARD ? 0 : load_mean_hdr(means, g, b, c);
    for (uint32_t idx = threadIdx.x; idx < total; idx += blockDim.x) {
        const uint32_t r = idx & (R - 1), k2 = idx >> lr;
        const uint32_t k = k1_0 + r + (k2 << l1);
        const double2 z = shf[k2 * R + r];
        if (FORWARD) {
            const double2 pw = post[k];
            const double cval = pw.x * z.x + pw.y * z.y;            // Re(e^{-i pi k/2n} V[k])
            const double sres = cval * (k == 0 ? scale0 : scale1);  // Cs[k]*sqrt(2/n)/128 (dct.cpp:84)
            orow[k] = trunc_i32_c(sres);                              // C truncation (dct.cpp:85)
        } else {
            const uint32_t i = k < (n >> 1) ? 2u * k : 2u * (n - 1u - k) + 1u;
            const double sres = z.x * scale1;  // sqrt(2/n)*128 (dct.cpp:97)
            orow[i] = (int32_t)((uint32_t)trunc_i32_c(sres) + (uint32_t)mean);
        }
    }
}

// ---- forward DCT through a REAL-input FFT: half the points, half the scratch round trip ----------------------------------
// v (Makhoul's permutation of the n real samples) is packed as z[m] = v[2m] + i v[2m+1], M = n/2 complex points;
// Z = FFT_M(z) by the same four-step scheme (M = m1*m2, m = j1*m2 + j2, k = k1 + m1*k2), and
//   E[k] = (Z[k] + conj Z[M-k]) / 2,  O[k] = (Z[k] - conj Z[M-k]) / 2i,  w = e^{-2 pi i k / n}
//   V[k] = E + w O,  V[k+M] = E - w O,  V[M-k] = conj V[k+M],  V[n-k] = conj V[k]           (0 < k < M, k != M/2)
//   V[0] = Re Z0 + Im Z0,  V[M] = Re Z0 - Im Z0;  k = M/2 pairs with itself
// so a (k, M-k) pair of Z gives four DCT coefficients C[x] = Re(e^{-i pi x / 2n} V[x]).  Z[M-k] sits in row m1-k1 (column
// m2-1-k2) of the second FFT stage: a workgroup of k_dctr_rows therefore takes R rows p..p+R-1 together with their
// mirrors m1-p-R+1..m1-p; rows 0 and m1/2 pair within themselves and go to one extra workgroup.  m2 = 64 where it can
// be, so that R = 32 and every run of output coefficients is a whole 128-byte line.
__device__ __forceinline__ double dctr_sample(const int32_t* __restrict__ row, uint32_t n, int32_t mean, uint32_t j) {
    // v[j] = s[2j] (j < n/2), v[n-1-j] = s[2j+1]; s = (float)(src - mean) as in dct.cpp:80,108-109
    const uint32_t x = j < (n >> 1) ? 2u * j : 2u * (n - 1u - j) + 1u;
    return (double)(float)(int32_t)((uint32_t)row[x] - (uint32_t)mean);
}

__global__ __launch_bounds__(1024) void k_dctr_cols(const int32_t* __restrict__ in, Geom g, const int32_t* __restrict__ mean_i32,
                                                   const double2* __restrict__ tw, double2* __restrict__ scratch, uint32_t la, uint32_t lb,
                                                   uint32_t b0) {
    extern __shared__ __attribute__((aligned(16))) double2 shf[];
    const uint32_t n = g.ns, nlog = la + lb + 1, M = n >> 1;
    const uint32_t lw = min(kFftLdsLog - la, lb), W = 1u << lw;
    const uint32_t j2_0 = blockIdx.x << lw, c = blockIdx.y, bl = blockIdx.z, b = b0 + bl;
    const int32_t* row = in + (size_t)b * g.N + (size_t)c * n;
    const uint32_t total = 1u << (la + lw);
    double2* stw = shf + total;
    load_stage_twiddles(stw, tw, la, nlog);
    const int32_t mean = mean_i32[(size_t)b * g.nch + c];
    for (uint32_t idx = threadIdx.x; idx < total; idx += blockDim.x) {
        const uint32_t col = idx & (W - 1), j1 = idx >> lw;
        const uint32_t m = (j1 << lb) + j2_0 + col;
        shf[bitrev(j1, la) * W + col] = make_double2(dctr_sample(row, n, mean, 2u * m), dctr_sample(row, n, mean, 2u * m + 1u));
    }
    __syncthreads();
    lds_fft(shf, stw, la, lw, false);
    double2* dst = scratch + ((size_t)bl * g.nch + c) * M;
    for (uint32_t idx = threadIdx.x; idx < total; idx += blockDim.x) {
        const uint32_t col = idx & (W - 1), k1 = idx >> lw;
        const uint32_t j2 = j2_0 + col;
        const double2 w = tw[2u * j2 * k1];  // e^{2 pi i j2 k1 / M}; j2*k1 < M
        const double2 z = shf[k1 * W + col];
        dst[((size_t)k1 << lb) + j2] = make_double2(z.x * w.x + z.y * w.y, z.y * w.x - z.x * w.y);  // z * conj(w)
    }
}

__global__ __launch_bounds__(1024) void k_dctr_rows(const double2* __restrict__ scratch, Geom g, const double2* __restrict__ tw,
                                                   const double2* __restrict__ post, int32_t* __restrict__ out, uint32_t la, uint32_t lb,
                                                   uint32_t b0, double scale0, double scale1) {
    extern __shared__ __attribute__((aligned(16))) double2 shf[];
    const uint32_t n = g.ns, nlog = la + lb + 1, M = n >> 1, m1 = 1u << la, m2 = 1u << lb;
    const uint32_t lr = kFftLdsLog - lb, RR = 1u << lr, R = RR >> 1;  // RR sequences in LDS: R rows and their R mirrors
    const uint32_t npairs = (m1 >> 1) - 1;                            // rows 1 .. m1/2-1 pair with m1-1 .. m1/2+1
    const uint32_t ngroups = (npairs + R - 1) / R;
    const uint32_t grp = blockIdx.x, c = blockIdx.y, bl = blockIdx.z, b = b0 + bl;
    const bool special = grp == ngroups;  // rows 0 and m1/2
    const uint32_t p0 = 1u + grp * R;
    const uint32_t total = RR << lb;
    double2* stw = shf + total;
    load_stage_twiddles(stw, tw, lb, nlog);
    const double2* src = scratch + ((size_t)bl * g.nch + c) * M;
    // sequence q of the workgroup: q < R: row p0 + q; q >= R: row m1 - (p0 + q - R)   (special: q = 0: row 0, q = 1: row m1/2)
    auto row_of = [&](uint32_t q) -> uint32_t {
        if (special) return q == 0 ? 0u : q == 1 ? (m1 >> 1) : 0xFFFFFFFFu;
        const uint32_t p = p0 + (q & (R - 1));
        if (p > npairs) return 0xFFFFFFFFu;
        return q < R ? p : m1 - p;
    };
    for (uint32_t idx = threadIdx.x; idx < total; idx += blockDim.x) {
        const uint32_t q = idx >> lb, j2 = idx & (m2 - 1);
        const uint32_t r = row_of(q);
        shf[bitrev(j2, lb) * RR + q] = r != 0xFFFFFFFFu ? src[((size_t)r << lb) + j2] : make_double2(0.0, 0.0);
    }
    __syncthreads();
    lds_fft(shf, stw, lb, lr, false);
    int32_t* orow = out + (size_t)b * g.N + (size_t)c * n;
    // C[x] = trunc(Re(e^{-i pi x / 2n} V) * Cs[x]*sqrt(2/n)/128)   (dct.cpp:84-85)
    auto emit = [&](uint32_t x, double vr, double vi) {
        const double2 pw = post[x];
        orow[x] = trunc_i32_c((pw.x * vr + pw.y * vi) * (x == 0 ? scale0 : scale1));
    };
    auto pair_out = [&](uint32_t k, const double2& za, const double2& zb) {  // za = Z[k], zb = Z[M-k]; 0 < k < M, k != M/2
        const double er = 0.5 * (za.x + zb.x), ei = 0.5 * (za.y - zb.y);     // E
        const double orr = 0.5 * (za.y + zb.y), oi = -0.5 * (za.x - zb.x);   // O = (za - conj zb) / 2i
        const double2 w = tw[k];                                             // e^{+2 pi i k/n}: multiply by its conjugate
        const double tr = orr * w.x + oi * w.y, ti = oi * w.x - orr * w.y;   // w' O
        emit(k, er + tr, ei + ti);              // V[k]
        emit(k + M, er - tr, ei - ti);          // V[k+M]
        emit(M - k, er - tr, -(ei - ti));       // V[M-k] = conj V[k+M]
        emit(n - k, er + tr, -(ei + ti));       // V[n-k] = conj V[k]
    };
    if (!special) {
        for (uint32_t idx = threadIdx.x; idx < (R << lb); idx += blockDim.x) {
            const uint32_t r = idx & (R - 1), k2 = idx >> (lr - 1);
            const uint32_t p = p0 + r;
            if (p > npairs) continue;
            pair_out(p + (k2 << la), shf[k2 * RR + r], shf[(m2 - 1 - k2) * RR + R + r]);
        }
    } else {
        // row 0: k = m1 k2 pairs with column m2 - k2; k2 = 0 and k2 = m2/2 pair with themselves
        for (uint32_t k2 = threadIdx.x; k2 <= (m2 >> 1); k2 += blockDim.x) {
            const double2 za = shf[k2 * RR];
            if (k2 == 0) {
                emit(0, za.x + za.y, 0.0);  // V[0] = Re Z0 + Im Z0
                emit(M, za.x - za.y, 0.0);  // V[M] = Re Z0 - Im Z0
            } else if (k2 == (m2 >> 1)) {
                // k = M/2: E = Re Z, O = Im Z, w = e^{-i pi/2}: V[M/2] = conj Z, V[3M/2] = Z
                emit(M >> 1, za.x, -za.y);
                emit(M + (M >> 1), za.x, za.y);
            } else {
                pair_out(k2 << la, za, shf[(m2 - k2) * RR]);
            }
        }
        // row m1/2: k = m1/2 + m1 k2 pairs with column m2 - 1 - k2 of the same row (never with itself)
        if (m1 >= 2) {
            for (uint32_t k2 = threadIdx.x; k2 < (m2 >> 1); k2 += blockDim.x)
                pair_out((m1 >> 1) + (k2 << la), shf[k2 * RR + 1], shf[(m2 - 1 - k2) * RR + 1]);
        }
    }
}

// ---- inverse DCT through a real-OUTPUT FFT ---------------------------------------------------------------------------------
// With V'[k] = e^{+i pi k/2n} (Y'[k] - i Y'[n-k]) (Hermitian: the n-point inverse transform v is real),
//   Z'[k] = (V'[k] + V'[k+M]) + i e^{+2 pi i k/n} (V'[k] - V'[k+M]),  k < M = n/2
// and the M-point inverse transform of Z' is z[m] = v[2m] + i v[2m+1]: half the points and half the scratch again.  The
// pairing is on the input side here (every Z' needs four coefficients), the rows come out unpaired.
__global__ __launch_bounds__(1024) void k_idctr_cols(const int32_t* __restrict__ in, Geom g, const double2* __restrict__ tw,
                                                    const double2* __restrict__ post, double2* __restrict__ scratch, uint32_t la, uint32_t lb,
                                                    uint32_t b0, float cs0) {
    extern __shared__ __attribute__((aligned(16))) double2 shf[];
    const uint32_t n = g.ns, nlog = la + lb + 1, M = n >> 1;
    const uint32_t lw = min(kFftLdsLog - la, lb), W = 1u << lw;
    const uint32_t j2_0 = blockIdx.x << lw, c = blockIdx.y, bl = blockIdx.z, b = b0 + bl;
    const int32_t* row = in + (size_t)b * g.N + (size_t)c * n;
    const uint32_t total = 1u << (la + lw);
    double2* stw = shf + total;
    load_stage_twiddles(stw, tw, la, nlog);
    auto vprime = [&](uint32_t x) -> double2 {  // V'[x], x < n  (dct.cpp:95: Cs[0]*dct[0] in float)
        double yk, ynk;
        if (x == 0) {
            yk = (double)__fmul_rn(cs0, (float)row[0]);
            ynk = 0.0;
        } else {
            yk = 0.5 * (double)(float)row[x];
            ynk = 0.5 * (double)(float)row[n - x];
        }
        const double2 pw = post[x];
        return make_double2(pw.x * yk + pw.y * ynk, pw.y * yk - pw.x * ynk);
    };
    for (uint32_t idx = threadIdx.x; idx < total; idx += blockDim.x) {
        const uint32_t col = idx & (W - 1), j1 = idx >> lw;
        const uint32_t k = (j1 << lb) + j2_0 + col;
        const double2 a = vprime(k), bq = vprime(k + M);
        const double2 w = tw[k];
        const double dx = a.x - bq.x, dy = a.y - bq.y;
        // (a + b) + i w (a - b)
        shf[bitrev(j1, la) * W + col] = make_double2(a.x + bq.x - (w.x * dy + w.y * dx), a.y + bq.y + (w.x * dx - w.y * dy));
    }
    __syncthreads();
    lds_fft(shf, stw, la, lw, true);
    double2* dst = scratch + ((size_t)bl * g.nch + c) * M;
    for (uint32_t idx = threadIdx.x; idx < total; idx += blockDim.x) {
        const uint32_t col = idx & (W - 1), k1 = idx >> lw;
        const uint32_t j2 = j2_0 + col;
        const double2 w = tw[2u * j2 * k1];  // e^{+2 pi i j2 k1 / M}
        const double2 z = shf[k1 * W + col];
        dst[((size_t)k1 << lb) + j2] = make_double2(z.x * w.x - z.y * w.y, z.x * w.y + z.y * w.x);
    }
}

__global__ __launch_bounds__(1024) void k_idctr_rows(const double2* __restrict__ scratch, Geom g, const uint8_t* __restrict__ means,
                                                    const double2* __restrict__ tw, int32_t* __restrict__ out, uint32_t la, uint32_t lb,
                                                    uint32_t b0, double scale1) {
    extern __shared__ __attribute__((aligned(16))) double2 shf[];
    const uint32_t n = g.ns, nlog = la + lb + 1, M = n >> 1, m2 = 1u << lb;
    const uint32_t lr = min(kFftLdsLog - lb, la), R = 1u << lr;
    const uint32_t k1_0 = blockIdx.x << lr, c = blockIdx.y, bl = blockIdx.z, b = b0 + bl;
    const uint32_t total = R << lb;
    double2* stw = shf + total;
    load_stage_twiddles(stw, tw, lb, nlog);
    const double2* src = scratch + ((size_t)bl * g.nch + c) * M + ((size_t)k1_0 << lb);
    for (uint32_t idx = threadIdx.x; idx < total; idx += blockDim.x) {
        const uint32_t r = idx >> lb, j2 = idx & (m2 - 1);
        shf[bitrev(j2, lb) * R + r] = src[idx];
    }
    __syncthreads();
    lds_fft(shf, stw, lb, lr, true);
    int32_t* orow = out + (size_t)b * g.N + (size_t)c * n;
    const int32_t mean = load_mean_hdr(means, g, b, c);
    auto put = [&](uint32_t j, double v) {  // v[j] -> its sample: s[2j] = v[j] (j < n/2), s[2j+1] = v[n-1-j]
        const uint32_t i = j < (n >> 1) ? 2u * j : 2u * (n - 1u - j) + 1u;
        orow[i] = (int32_t)((uint32_t)trunc_i32_c(v * scale1) + (uint32_t)mean);  // sqrt(2/n)*128, C truncation (dct.cpp:97-98)
    };
    for (uint32_t idx = threadIdx.x; idx < total; idx += blockDim.x) {
        const uint32_t r = idx & (R - 1), k2 = idx >> lr;
        const uint32_t m = k1_0 + r + (k2 << la);
        const double2 z = shf[k2 * R + r];
        put(2u * m, z.x);
        put(2u * m + 1u, z.y);
    }
}

template __global__ void k_dctfft_cols<true>(const int32_t*, Geom, const int32_t*, const double2*, const double2*, double2*, uint32_t, uint32_t,
                                             uint32_t, float);
template __global__ void k_dctfft_cols<false>(const int32_t*, Geom, const int32_t*, const double2*, const double2*, double2*, uint32_t, uint32_t,
                                              uint32_t, float);
template __global__ void k_dctfft_rows<true>(const double2*, Geom, const uint8_t*, const double2*, const double2*, int32_t*, uint32_t, uint32_t,
                                             uint32_t, double, double);
template __global__ void k_dctfft_rows<false>(const double2*, Geom, const uint8_t*, const double2*, const double2*, int32_t*, uint32_t, uint32_t,
                                              uint32_t, double, double);

template __global__ void k_fwht<true>(int32_t*, Geom, uint8_t*);
template __global__ void k_fwht_cross<true>(int32_t*, Geom, const uint8_t*, const int32_t*, uint32_t, uint32_t, uint32_t);
template __global__ void k_fwht_cross<false>(int32_t*, Geom, const uint8_t*, const int32_t*, uint32_t, uint32_t, uint32_t);
template __global__ void k_fwht<false>(int32_t*, Geom, uint8_t*);
template __global__ void k_fwht64k<true, true>(int32_t*, Geom, uint8_t*, uint8_t*, uint32_t*, uint32_t);
template __global__ void k_fwht64k<false, false>(int32_t*, Geom, uint8_t*, uint8_t*, uint32_t*, uint32_t);
template __global__ void k_dct<true>(const int32_t*, Geom, uint8_t*, const float*, double, double, float, int32_t*);
template __global__ void k_dct<false>(const int32_t*, Geom, uint8_t*, const float*, double, double, float, int32_t*);

}  // namespace rspt
