// rspt_hip.cpp -- C ABI (include/rspt_hip.h) over the gfx950 kernels.
//
// One handle = one reference packer instance: it owns the device workspace
// (what enc_/serialized_ are in signal_packer_base.h:20-21), one HIP stream and
// the persistent nr_bytes_to_compress_ state (signal_packer_xdelta_hzr.cpp:39,66),
// which lives in device memory so that batches chain without a host round trip.
// There is no CPU path: every entry point fails loudly if the device is missing.
#include "../../include/rspt_hip.h"

#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <link.h>
#include <climits>
#include <string>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include "common.hpp"

// unity build: the kernels live in the same translation unit
#include "preprocess.hip"
#include "hzr_kernels.hip"
#include "hzr_rows.hip"
#include "transforms.hip"
#include "decode.hip"
#include "filter.hip"

using namespace rspt;

namespace {

// GF(2) helpers on the host (tools/kernel_model.py has the same math)
uint32_t x_pow_bytes(uint64_t nbytes) {
    uint32_t r = 0x80000000u, base = 0x00800000u;
    while (nbytes) {
        if (nbytes & 1) r = gf_mul(r, base);
        base = gf_mul(base, base);
        nbytes >>= 1;
    }
    return r;
}

uint32_t raw_crc4(uint32_t le) {
    uint32_t c = le;
    for (int i = 0; i < 32; ++i) c = (c >> 1) ^ (kCrcPoly & (0u - (c & 1u)));
    return c;
}

void make_crc_consts(CrcConsts& cc) {
    for (uint32_t b = 0; b < 256; ++b) {
        uint32_t r = b;
        for (int k = 0; k < 8; ++k) r = (r >> 1) ^ (kCrcPoly & (0u - (r & 1u)));
        cc.table[0][b] = r;
    }
    for (int t = 1; t < 4; ++t)  // slice-by-4: table[t][b] = state after byte b followed by t zero bytes
        for (uint32_t b = 0; b < 256; ++b) cc.table[t][b] = (cc.table[t - 1][b] >> 8) ^ cc.table[0][cc.table[t - 1][b] & 0xFFu];
    for (int i = 0; i < 81; ++i) {
        const uint32_t K = i < 64 ? x_pow_bytes(64ull * (63 - i)) : i < 80 ? x_pow_bytes(4096ull * (15 - (i - 64))) : x_pow_bytes(65536);
        for (int jx = 0; jx < 4; ++jx)
            for (uint32_t b = 0; b < 256; ++b) cc.shift[i][jx][b] = gf_mul(b << (8 * jx), K);
    }
    for (int l = 0; l < 64; ++l) {
        const uint32_t K = x_pow_bytes(4ull * (l + 1));
        for (int jx = 0; jx < 4; ++jx)
            for (uint32_t b = 0; b < 256; ++b) cc.shift4[l][jx][b] = gf_mul(b << (8 * jx), K);
    }
    // X with raw_crc(X) = 0xFFFFFFFF: the 4-byte raw CRC map is linear and invertible
    uint32_t img[32];
    for (int b = 0; b < 32; ++b) img[b] = raw_crc4(1u << b);
    uint32_t rows_v[32], rows_t[32];
    for (int b = 0; b < 32; ++b) {
        rows_v[b] = img[b];
        rows_t[b] = 1u << b;
    }
    int piv[32];
    bool used[32] = {false};
    for (int bit = 0; bit < 32; ++bit) {
        piv[bit] = -1;
        for (int i = 0; i < 32; ++i)
            if (!used[i] && ((rows_v[i] >> bit) & 1u)) {
                piv[bit] = i;
                used[i] = true;
                for (int jx = 0; jx < 32; ++jx)
                    if (jx != i && ((rows_v[jx] >> bit) & 1u)) {
                        rows_v[jx] ^= rows_v[i];
                        rows_t[jx] ^= rows_t[i];
                    }
                break;
            }
    }
    uint32_t x = 0;
    for (int bit = 0; bit < 32; ++bit) x ^= rows_t[piv[bit]];  // target has every bit set
    cc.prefix = x;
    cc.pad[0] = cc.pad[1] = cc.pad[2] = 0;
}

enum Stage { ST_PRE = 0, ST_NB, ST_HIST, ST_TREE, ST_LAYOUT, ST_ENCODE, ST_ENCODE_SMALL, ST_COUNT };
const char* kStageNames[ST_COUNT] = {"preprocess", "nb_scan", "hzr_hist", "hzr_tree", "layout", "hzr_encode", "hzr_encode_small"};

}  // namespace

struct Feed;
struct rspt_hip_packer {
    Geom g{};
    int device = 0;
    hipStream_t stream = nullptr;
    int last_hip_error = 0;
    unsigned nb_ctor = 0;
    unsigned nb_host = 0;  // last value of the device nb_state the host has seen (a lower bound: nb only grows)

    // workspace
    size_t cap_blocks = 0;
    uint8_t* planes = nullptr;     // [cap][4][plane_stride]
    int32_t* planar = nullptr;     // [cap][N] (transform packers, decode)
    uint32_t* needmask = nullptr;  // [cap]
    uint32_t* nbuse = nullptr;     // [cap]
    uint32_t* dec_nb = nullptr;    // [cap] decode: planes of each stream (container index entry, else nb_state)
    uint32_t* work_ctr = nullptr;  // [16] work counter of the persistent k_hist at 0, the WorkQueues of k_encode from 4 (zeroed per call)
    uint32_t* big_list = nullptr;  // [cap*4*nblk] hzr blocks for the workgroup-per-block encoder (filled by k_layout)
    uint32_t* small_list = nullptr;  // [cap*4*nblk] hzr blocks for the wave-per-block encoder
    int num_cu = 256;
    uint32_t* nzflag = nullptr;    // [cap*4*nblk] set by the front end when an hzr block holds a non-zero byte (= zbuf[set of the last call])
    // The per-call zero region [nzflag | needmask | work counters | row sums] exists twice: while a call works in one copy its
    // k_tree zeroes the other for the next call (one store per thread) -- the memset in front of every call was a 9 us launch.
    uint32_t* zbuf[2] = {nullptr, nullptr};
    size_t zcap_words = 0;
    bool zero_ready[2] = {false, false};  // the copy is known to be all zero
    int zset = 0;                          // the copy the next call works in
    // Clean-block invariant (k_tile_stream's skipped stores): between calls, hzr block j of plane k of block slot b holds
    // zeros everywhere unless its bit in plane_dirty is set (128 bits per plane, bit = j >> dirty_shift).  The streaming
    // front end writes only the 128-byte lines that hold a non-zero byte into a clean block; k_layout sets the bits of
    // the blocks in which data stays behind, and the encoders wipe the non-zero granules of all others right after
    // reading them (light blocks only: block_is_wiped).
    uint32_t* plane_dirty = nullptr;  // [cap*4][4]
    uint32_t dirty_shift = 0;
    bool planes_unknown = false;      // something else (decompress, a diagnostic run) wrote the planes: flag them all
    uint32_t* nb_state = nullptr;  // [4] persistent: [0] = nb; [2] = work counter of the decoder's persistent grid (zeroed by k_dec_frame)
    uint32_t* hist = nullptr;      // [cap*4*nblk][264]
    uint32_t* seghist = nullptr;   // [cap*4*nblk][16][264] u16: tokens ending in each 4 KiB segment (k_hist -> k_tree)
    uint32_t* segbase = nullptr;   // [cap*4*nblk][16] stream bit at which each segment's tokens start (k_tree -> k_encode)
    uint32_t* lists = nullptr;     // [cap*4*nblk][16][kListCap] (position << 9 | value) entries of the sparse segments (k_hist -> k_encode)
    uint2* listinfo = nullptr;     // [cap*4*nblk][16] {entries or kListNone, position behind the last literal before the segment}
    uint32_t* cw = nullptr;        // [cap*4*nblk][264] code | length << 24 per symbol
    uint32_t* tdesc = nullptr;     // [..][92]
    BlockMeta* meta = nullptr;     // [..]
    uint64_t* out_off = nullptr;   // [..]
    uint8_t* means = nullptr;      // [cap][hdr_len]
    int32_t* planar2 = nullptr;    // [cap][N] second int32 buffer (dct output / idct output)
    uint32_t* txor = nullptr;      // [cap][ntile] decode scans
    uint32_t* tsum = nullptr;      // [cap][ntile]
    uint32_t* rowrec = nullptr;    // [cap][N / 256][kRowRec] row tile records of the int32 decode path (k_inv_rows)
    uint64_t* blk_off = nullptr;   // [cap*4*nblk] decode: hzr block offsets inside each stream
    CrcConsts* crc = nullptr;
    // dct (signal_packer_dct.cpp:60-74): COS[x][i] and its transpose, built on the host like the reference ctor
    float* cos_tab = nullptr;
    float* cos_tab_t = nullptr;
    double dct_scale0 = 0, dct_scale1 = 0, idct_scale = 0;
    float dct_cs0 = 0;
    // dct beyond the dense table: fp64 FFT path (transforms.hip: k_dctfft_*)
    bool dct_fft = false;
    uint32_t fft_l1 = 0, fft_l2 = 0;   // n = 2^(l1+l2)
    bool dct_real = false;             // forward transform through the real-input FFT (n >= 256)
    uint32_t fftr_la = 0, fftr_lb = 0; // n/2 = 2^(la+lb)
    double2* fft_tw = nullptr;         // [n] (cos, sin)(2 pi t / n)
    double2* fft_post = nullptr;       // [n] (cos, sin)(pi k / 2n)
    double2* fft_scratch = nullptr;    // [fft_bpp][nch][n]
    size_t fft_bpp = 0;                // blocks per pass (bounds the scratch to ~1 GiB)
    int32_t* mean_i32 = nullptr;       // [cap][nch]
    long long* row_sum = nullptr;      // [blocks][nch] channel sums taken by the de-interleave pass (dct at large ns); lives in the per-call zero region
    bool have_row_sum = false;         // this call's front end filled row_sum
    uint32_t ntile = 0;
    uint32_t Tn_native = 0;  // tile of k_planar_native

    // host API staging
    uint8_t* h_src = nullptr;  // device
    uint8_t* h_dst = nullptr;  // device
    uint64_t* h_size = nullptr;
    size_t h_dst_cap = 0;
    // rspt_hip_compress_many: two slots of a chunk of blocks each
    uint8_t* m_src[2] = {nullptr, nullptr};   // device
    uint8_t* m_dst[2] = {nullptr, nullptr};   // device
    uint64_t* m_sizes[2] = {nullptr, nullptr};  // device
    uint64_t* m_hsizes = nullptr;             // page-locked host, 2 x chunk
    uint64_t* m_idx[2] = {nullptr, nullptr};  // device: [4 + 2 x chunk] a container header + index over a slot's streams (decompress_many with src_len)
    uint64_t* m_hidx = nullptr;               // page-locked host, 2 x (4 + 2 x chunk)
    struct Feed* feed = nullptr;              // rspt_hip_feed_*: a ring of block groups in flight
    uint64_t* gat_totals = nullptr;           // device [gat_world]: container lengths of all ranks (rspt_hip_gather_containers)
    int gat_world = 0;
    // rspt_hip_gather_post_*: two slots of sizes (device + page-locked host), events, the gather stream
    uint64_t* lag_dtotals[2] = {nullptr, nullptr};
    uint64_t* lag_htotals[2] = {nullptr, nullptr};
    hipEvent_t lag_ev_in[2] = {}, lag_ev_sizes[2] = {}, lag_ev_payload[2] = {};
    bool lag_posted[2] = {false, false};
    hipStream_t lag_stream = nullptr;
    int lag_world = 0;
    size_t m_chunk = 0, m_stride = 0;
    hipStream_t m_up = nullptr, m_down = nullptr;
    hipEvent_t m_ev_up[2] = {}, m_ev_comp[2] = {}, m_ev_down[2] = {};

    // tile geometry for the front end
    uint32_t T = 0, in_lds = 0;  // k_tile_planar: tile staged in LDS
    uint32_t Tp[5] = {0, 0, 0, 0, 0};  // k_tile_planes: tile length when kcount planes are staged: rows [kcount*nch][Tp+16] + nz flags

    unsigned long long* stamps = nullptr;  // diagnostic s_memtime stamps: [512 hzr blocks][16 waves][8]
    bool wide = false;          // more channels than a 16-sample tile of the front-end kernels holds in LDS: k_wide_planar + k_planar_planes
    uint32_t k1_threads = 256;  // workgroup size of k_tile_planes (RSPT_K1_THREADS)
    uint32_t k1_grid = 0;       // workgroups of k_tile_planes; 0 = by LDS footprint (RSPT_K1_GRID, tuning knob)
    uint32_t hist_grid = 0;     // workgroups of the persistent k_hist / k_encode; 0 = two per CU (a CU's wave slots: one batch at a time)
    uint32_t enc_grid = 0;
    uint32_t ablate = 0;  // RSPT_ABLATE (diagnostic builds only; the product kernels ignore it): timing probes
    uint32_t psel = 0;    // RSPT_PLANESEL (diagnostic builds only): which planes the hzr kernels take; bit 8 / 9: stop behind k_hist / k_tree
    int verify = 0;       // decompress checks the block CRCs (rspt_hip_set_verify)
    int big_endian = 0;   // samples arrive / leave with their bytes reversed (rspt_hip_set_byte_order)

    // the small-block encoder runs beside the big one (it fills the CUs the persistent grid frees in its tail)
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    uint32_t* h_nsmall = nullptr;  // page-locked, device-visible: the small-block count of a recent batch (written by k_encode_small)

    // profiling
    bool profiling = false;
    hipEvent_t ev[ST_COUNT + 1] = {};
    bool ev_valid = false;
};

#define HIPCHK(p, call)                         \
    do {                                        \
        hipError_t e_ = (call);                 \
        if (e_ != hipSuccess) {                 \
            (p)->last_hip_error = (int)e_;      \
            return RSPT_HIP_ERR_LAUNCH;         \
        }                                       \
    } while (0)

static void stamp(rspt_hip_packer* p, int i, hipStream_t st) {
    if (p->profiling) hipEventRecord(p->ev[i], st);
}

template <int BPS, bool XD>
static void launch_planes(rspt_hip_packer* p, const uint8_t* d_src, size_t nblocks, uint32_t kfirst, uint32_t kcount, const uint32_t* nbuse,
                          hipStream_t st) {
    const Geom& g = p->g;
    const uint32_t T = p->Tp[kcount];
    const bool stream_ok = (reinterpret_cast<uintptr_t>(d_src) & 3u) == 0 && g.ns >= 16 && g.block_bytes < (1ull << 32) && !(p->ablate & (1u << 22));
    const uint32_t lds = kcount * g.nch * (T + 16u) + 32u * g.nch + 96u;
    const uint32_t ntiles = (uint32_t)((g.ns + T - 1) / T * nblocks);
    const uint32_t per_cu = lds <= 40 * 1024 ? 4u : lds <= 80 * 1024 ? 2u : 1u;
    uint32_t want = per_cu * (uint32_t)p->num_cu;
    if (p->k1_grid) want = p->k1_grid;
    dim3 grid(want < ntiles ? want : ntiles);
    // every sample width streams (k_tile_stream, one load per sample: dword / unaligned dword / short / byte)
    const bool stream = stream_ok;
    if (stream) {
        constexpr int SB = BPS;
        auto go = [&](auto kern) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, d_src, g, T, kfirst, kcount, p->planes, p->needmask, p->nzflag, nbuse, p->ablate,
                               (uint32_t)nblocks, nbuse ? nullptr : p->work_ctr + 1, p->nb_state, p->nbuse, p->plane_dirty, p->dirty_shift);
        };
        if (g.ns & 15u)
            go(&k_tile_stream<SB, XD, true>);  // (a short last group per channel, plane rows at any byte alignment)
        else
            go(&k_tile_stream<SB, XD, false>);
        return;
    }
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tile_planes<BPS, XD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k_tile_planes<BPS, XD>), grid, dim3(p->k1_threads), lds, st, d_src, g, T, kfirst, kcount, p->planes, p->needmask, p->nzflag, nbuse, p->ablate, (uint32_t)nblocks,
                       nbuse ? nullptr : p->work_ctr + 1, p->nb_state, p->nbuse);
}

// main front-end pass; returns the number of planes it wrote (xdelta: nb as last seen by the host)
template <int BPS>
static uint32_t launch_front(rspt_hip_packer* p, const uint8_t* d_src, size_t nblocks, hipStream_t st) {
    const Geom& g = p->g;
    if (p->wide) {
        // wide blocks (> ~1000 channels): transpose to the planar block, then -- for the two hzr packers -- the flat stage over it.
        // All four planes are written (an escalation inside the batch needs no second pass), nbuse[] says how many the encoders take.
        hipLaunchKernelGGL(k_wide_planar<BPS>, dim3((g.ns + 63) / 64, (g.nch + 63) / 64, (unsigned)nblocks), dim3(256), 0, st, d_src, g, p->planar);
        if (g.kind == RSPT_HIP_KIND_XDELTA_HZR || g.kind == RSPT_HIP_KIND_HZR) {
            const bool xd = g.kind == RSPT_HIP_KIND_XDELTA_HZR;
            const dim3 pg((g.N + 4095) / 4096, (unsigned)nblocks);
            if (xd)
                hipLaunchKernelGGL((k_planar_planes<true>), pg, dim3(256), 0, st, p->planar, g, 4u, p->planes, p->nzflag, p->needmask);
            else
                hipLaunchKernelGGL((k_planar_planes<false>), pg, dim3(256), 0, st, p->planar, g, 4u, p->planes, p->nzflag, (uint32_t*)nullptr);
            hipLaunchKernelGGL(k_nb_scan, dim3(1), dim3(1024), 0, st, p->needmask, (uint32_t)nblocks, p->nb_state, p->nbuse, xd ? 1 : 0);
        }
        return 4;
    }
    if (g.kind == RSPT_HIP_KIND_XDELTA_HZR) {
        const uint32_t np = p->nb_host;
        launch_planes<BPS, true>(p, d_src, nblocks, 0, np, nullptr, st);
        return np;
    }
    if (g.kind == RSPT_HIP_KIND_HZR) {
        launch_planes<BPS, false>(p, d_src, nblocks, 0, 4, nullptr, st);
        return 4;
    }
    if (BPS == 4 && (g.nch & 3) == 0 && (g.ns & 3) == 0 && g.nch <= 1024 && (reinterpret_cast<uintptr_t>(d_src) & 15) == 0) {
        uint32_t T4 = (uint32_t)((32768ull / (4ull * g.nch) - 1) & ~3ull);  // T4 samples x nch channels in at most 32 KiB of LDS
        T4 = T4 > 1024 ? 1024 : T4 < 4 ? 4 : T4;
        if (T4 > g.ns) T4 = g.ns;
        hipLaunchKernelGGL(k_tile_planar_i32x4, dim3((g.ns + T4 - 1) / T4, (unsigned)nblocks), dim3(256), g.nch * (T4 + 1) * 4, st, d_src, g, T4,
                           p->planar, p->row_sum);
        p->have_row_sum = p->row_sum != nullptr;
        return 4;
    }
    dim3 grid((g.ns + p->T - 1) / p->T, (unsigned)nblocks);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tile_planar<BPS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->in_lds);
    hipLaunchKernelGGL((k_tile_planar<BPS>), grid, dim3(256), p->in_lds, st, d_src, g, p->T, p->planar);
    return 4;
}

// escalation fix-up: planes [np, 4) for the blocks whose nb grew past np in this call
// dct / idct of every channel of B blocks through the fp64 FFT path, `fft_bpp` blocks per pass (scratch bound)
template <bool FORWARD>
static void launch_dct_fft(rspt_hip_packer* p, uint32_t B, const int32_t* in, int32_t* out, hipStream_t st) {
    const Geom& g = p->g;
    const uint32_t l1 = p->fft_l1, l2 = p->fft_l2;
    const uint32_t lw = std::min(kFftLdsLog - l1, l2), lr = std::min(kFftLdsLog - l2, l1);
    const uint32_t lds_c = ((uint32_t)sizeof(double2) << (l1 + lw)) + ((uint32_t)sizeof(double2) << (l1 - 1)),
                   lds_r = ((uint32_t)sizeof(double2) << (l2 + lr)) + ((uint32_t)sizeof(double2) << (l2 - 1));  // points + stage twiddles
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dctfft_cols<FORWARD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dctfft_rows<FORWARD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_r);
    const uint32_t fthr = 1024;  // 4096 points per workgroup: 256 threads (4 waves) left the LDS passes latency-bound (34 -> 46 GS/s)
    if (FORWARD && p->dct_real) {
        // real-input form (k_dctr_*): M = n/2 complex points, half the scratch round trip
        const uint32_t la = p->fftr_la, lb = p->fftr_lb;
        const uint32_t lwr = std::min(kFftLdsLog - la, lb);
        const uint32_t ldsc = ((uint32_t)sizeof(double2) << (la + lwr)) + ((uint32_t)sizeof(double2) << (la - 1));
        const uint32_t ldsr = ((uint32_t)sizeof(double2) << kFftLdsLog) + ((uint32_t)sizeof(double2) << (lb - 1));
        const uint32_t R = 1u << (kFftLdsLog - lb - 1), npairs = (1u << (la - 1)) - 1, ngroups = (npairs + R - 1) / R;
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dctr_cols), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsc);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dctr_rows), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsr);
        for (uint32_t b0 = 0; b0 < B; b0 += (uint32_t)p->fft_bpp) {
            const uint32_t nbk = std::min<uint32_t>((uint32_t)p->fft_bpp, B - b0);
            hipLaunchKernelGGL(k_dctr_cols, dim3(1u << (lb - lwr), g.nch, nbk), dim3(fthr), ldsc, st, in, g, p->mean_i32, p->fft_tw, p->fft_scratch, la, lb, b0);
            hipLaunchKernelGGL(k_dctr_rows, dim3(ngroups + 1, g.nch, nbk), dim3(fthr), ldsr, st, p->fft_scratch, g, p->fft_tw, p->fft_post, out, la, lb, b0,
                               p->dct_scale0, p->dct_scale1);
        }
        return;
    }
    if (!FORWARD && p->dct_real) {
        // real-output form (k_idctr_*)
        const uint32_t la = p->fftr_la, lb = p->fftr_lb;
        const uint32_t lwr = std::min(kFftLdsLog - la, lb), lrr = std::min(kFftLdsLog - lb, la);
        const uint32_t ldsc = ((uint32_t)sizeof(double2) << (la + lwr)) + ((uint32_t)sizeof(double2) << (la - 1));
        const uint32_t ldsr = ((uint32_t)sizeof(double2) << (lb + lrr)) + ((uint32_t)sizeof(double2) << (lb - 1));
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_idctr_cols), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsc);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_idctr_rows), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsr);
        for (uint32_t b0 = 0; b0 < B; b0 += (uint32_t)p->fft_bpp) {
            const uint32_t nbk = std::min<uint32_t>((uint32_t)p->fft_bpp, B - b0);
            hipLaunchKernelGGL(k_idctr_cols, dim3(1u << (lb - lwr), g.nch, nbk), dim3(fthr), ldsc, st, in, g, p->fft_tw, p->fft_post, p->fft_scratch, la, lb, b0,
                               p->dct_cs0);
            hipLaunchKernelGGL(k_idctr_rows, dim3(1u << (la - lrr), g.nch, nbk), dim3(fthr), ldsr, st, p->fft_scratch, g, p->means, p->fft_tw, out, la, lb, b0,
                               p->idct_scale);
        }
        return;
    }
    for (uint32_t b0 = 0; b0 < B; b0 += (uint32_t)p->fft_bpp) {
        const uint32_t nbk = std::min<uint32_t>((uint32_t)p->fft_bpp, B - b0);
        hipLaunchKernelGGL((k_dctfft_cols<FORWARD>), dim3(1u << (l2 - lw), g.nch, nbk), dim3(fthr), lds_c, st, in, g, p->mean_i32, p->fft_tw,
                           p->fft_post, p->fft_scratch, l1, l2, b0, p->dct_cs0);
        hipLaunchKernelGGL((k_dctfft_rows<FORWARD>), dim3(1u << (l1 - lr), g.nch, nbk), dim3(fthr), lds_r, st, p->fft_scratch, g, p->means,
                           p->fft_tw, p->fft_post, out, l1, l2, b0, FORWARD ? p->dct_scale0 : 0.0, FORWARD ? p->dct_scale1 : p->idct_scale);
    }
}

// WHT of rows longer than 65536 points, in place in `planar` (transforms.hip: k_fwht_seg / k_fwht_cross)
template <bool FORWARD>
static void launch_fwht_big(rspt_hip_packer* p, uint32_t B, hipStream_t st) {
    const Geom& g = p->g;
    const uint32_t segs = g.ns >> 15;  // 32768-point pieces per row: 4 .. 128
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fwht_seg), hipFuncAttributeMaxDynamicSharedMemorySize, 32768 * 4);
    hipLaunchKernelGGL(k_fwht_seg, dim3(segs, g.nch, B), dim3(1024), 32768 * 4, st, p->planar, g);
    const uint32_t m1 = segs > 64u ? 64u : segs, m2 = segs / m1;
    hipLaunchKernelGGL((k_fwht_cross<FORWARD>), dim3(g.ns / m1 / 256u, g.nch, B), dim3(256), 0, st, p->planar, g, p->means, p->mean_i32, m1, 32768u,
                       m2 == 1u ? 1u : 0u);
    if (m2 > 1u)
        hipLaunchKernelGGL((k_fwht_cross<FORWARD>), dim3(g.ns / m2 / 256u, g.nch, B), dim3(256), 0, st, p->planar, g, p->means, p->mean_i32, m2,
                           32768u * m1, 1u);
}

template <int BPS>
static void launch_fixup(rspt_hip_packer* p, const uint8_t* d_src, size_t nblocks, uint32_t np, hipStream_t st) {
    launch_planes<BPS, true>(p, d_src, nblocks, np, 4 - np, p->nbuse, st);
}


template <int BPS, int NC>
static void launch_iir(rspt_hip_packer* p, uint8_t* buf, uint32_t B, const IirCoef& c, int per_channel, hipStream_t st) {
    const Geom& g = p->g;
    if (g.ns >= kIirChunk && c.init_steps >= NC - 1) {  // the pipelined form: recurrence, feed-forward sums and stores on waves of their own
        const bool al = (BPS == 4 || BPS == 2) && (reinterpret_cast<uintptr_t>(buf) % BPS) == 0;  // (block_bytes is a multiple of BPS)
        if (per_channel) {
            const uint32_t units = B * g.nch;
            auto go = [&](auto kern) { hipLaunchKernelGGL(kern, dim3((units + 63) / 64), dim3(kIirThreads), 0, st, buf, g.nch, g.ns, (uint64_t)g.block_bytes, c, B, 64u); };
            if (al) go(&k_iir_pipe<BPS, NC, false, (BPS == 4 || BPS == 2)>);
            else go(&k_iir_pipe<BPS, NC, false, false>);
        } else {
            // shared mode: lane <-> block; few lanes per workgroup so that the blocks' scattered accesses spread over the CUs
            uint32_t lpw = (B + (uint32_t)p->num_cu - 1) / (uint32_t)p->num_cu;
            lpw = lpw < 1 ? 1 : lpw > 64 ? 64 : lpw;
            auto go = [&](auto kern) { hipLaunchKernelGGL(kern, dim3((B + lpw - 1) / lpw), dim3(kIirThreads), 0, st, buf, g.nch, g.ns, (uint64_t)g.block_bytes, c, B, lpw); };
            if (al) go(&k_iir_pipe<BPS, NC, true, (BPS == 4 || BPS == 2)>);
            else go(&k_iir_pipe<BPS, NC, true, false>);
        }
        return;
    }
    if (per_channel) {
        const uint32_t threads = B * g.nch;
        hipLaunchKernelGGL((k_iir<BPS, NC, false>), dim3((threads + 63) / 64), dim3(64), 0, st, buf, g.nch, g.ns, (uint64_t)g.block_bytes, c, B);
    } else {
        hipLaunchKernelGGL((k_iir<BPS, NC, true>), dim3((B + 63) / 64), dim3(64), 0, st, buf, g.nch, g.ns, (uint64_t)g.block_bytes, c, B);
    }
}
template <int BPS>
static void launch_iir_nc(rspt_hip_packer* p, uint8_t* buf, uint32_t B, const IirCoef& c, int per_channel, hipStream_t st) {
    switch (c.nc) {
        case 2: launch_iir<BPS, 2>(p, buf, B, c, per_channel, st); break;
        case 3: launch_iir<BPS, 3>(p, buf, B, c, per_channel, st); break;
        case 4: launch_iir<BPS, 4>(p, buf, B, c, per_channel, st); break;
        default: launch_iir<BPS, 5>(p, buf, B, c, per_channel, st); break;
    }
}

template <bool XDELTA, int CG>
static void launch_inv_native(rspt_hip_packer* p, uint32_t B, uint32_t nrow, void* d_dst, hipStream_t st) {
    const Geom& g = p->g;
    constexpr uint32_t S = (1024u / CG) * 16u, lds = CG * (S + 1u) * 4u;  // > 64 KiB: the limit is raised per kernel
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_inv_native<XDELTA, CG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const dim3 ng((g.ns + S - 1) / S, (g.nch + CG - 1) / CG, B);
    hipLaunchKernelGGL((k_inv_native<XDELTA, CG>), ng, dim3(1024), lds, st, p->planes, g, p->dec_nb, nrow, p->txor, p->tsum, (uint8_t*)d_dst);
}

extern "C" {

const char* rspt_hip_status_string(int s) {
    switch (s) {
        case RSPT_HIP_OK: return "ok";
        case RSPT_HIP_ERR_ARG: return "invalid argument";
        case RSPT_HIP_ERR_NO_DEVICE: return "no usable gfx950 device (there is no CPU path)";
        case RSPT_HIP_ERR_ALLOC: return "allocation failed";
        case RSPT_HIP_ERR_LAUNCH: return "HIP call or kernel launch failed";
        case RSPT_HIP_ERR_DST_TOO_SMALL: return "destination too small for the compressed stream";
        case RSPT_HIP_ERR_CORRUPT: return "malformed stream";
        case RSPT_HIP_ERR_UNSUPPORTED: return "shape not supported by the kernels";
        case RSPT_HIP_ERR_BUSY: return "every slot of the feed is in flight";
        default: return "unknown status";
    }
}

int rspt_hip_last_hip_error(const rspt_hip_packer* p) { return p ? p->last_hip_error : 0; }

int rspt_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static void free_workspace(rspt_hip_packer* p) {
    hipFree(p->planes);
    hipFree(p->planar);
    hipFree(p->nbuse);
    hipFree(p->dec_nb);
    p->dec_nb = nullptr;
    hipFree(p->zbuf[0]);
    hipFree(p->zbuf[1]);
    p->zbuf[0] = p->zbuf[1] = nullptr;
    hipFree(p->plane_dirty);
    p->plane_dirty = nullptr;
    hipFree(p->big_list);
    hipFree(p->small_list);
    p->nzflag = p->big_list = p->small_list = nullptr;
    hipFree(p->hist);
    hipFree(p->seghist);
    hipFree(p->segbase);
    hipFree(p->lists);
    hipFree(p->listinfo);
    p->seghist = p->segbase = p->lists = nullptr;
    p->listinfo = nullptr;
    hipFree(p->cw);
    hipFree(p->tdesc);
    hipFree(p->meta);
    hipFree(p->out_off);
    hipFree(p->means);
    hipFree(p->planar2);
    hipFree(p->txor);
    hipFree(p->tsum);
    hipFree(p->rowrec);
    hipFree(p->blk_off);
    hipFree(p->fft_scratch);
    hipFree(p->mean_i32);
    p->fft_scratch = nullptr;
    p->mean_i32 = nullptr;
    p->planar2 = nullptr;
    p->txor = p->tsum = p->rowrec = nullptr;
    p->blk_off = nullptr;
    p->planes = nullptr;
    p->planar = nullptr;
    p->needmask = p->nbuse = p->hist = p->cw = p->tdesc = nullptr;
    p->meta = nullptr;
    p->out_off = nullptr;
    p->means = nullptr;
    p->cap_blocks = 0;
}

// everything ensure_many() creates; leaves the fields null so that a later call can start over
static void free_many(rspt_hip_packer* p) {
    for (int i = 0; i < 2; ++i) {
        hipFree(p->m_src[i]);
        hipFree(p->m_dst[i]);
        hipFree(p->m_sizes[i]);
        hipFree(p->m_idx[i]);
        p->m_src[i] = p->m_dst[i] = nullptr;
        p->m_sizes[i] = p->m_idx[i] = nullptr;
        if (p->m_ev_up[i]) hipEventDestroy(p->m_ev_up[i]);
        if (p->m_ev_comp[i]) hipEventDestroy(p->m_ev_comp[i]);
        if (p->m_ev_down[i]) hipEventDestroy(p->m_ev_down[i]);
        p->m_ev_up[i] = p->m_ev_comp[i] = p->m_ev_down[i] = nullptr;
    }
    if (p->m_hsizes) hipHostFree(p->m_hsizes);
    if (p->m_hidx) hipHostFree(p->m_hidx);
    p->m_hsizes = p->m_hidx = nullptr;
    if (p->m_up) hipStreamDestroy(p->m_up);
    if (p->m_down) hipStreamDestroy(p->m_down);
    p->m_up = p->m_down = nullptr;
    p->m_chunk = p->m_stride = 0;
}

int rspt_hip_packer_create(rspt_hip_packer** out, int kind_and_flags, size_t bps, size_t nch, size_t ns, size_t nb, int device) {
    if (!out) return RSPT_HIP_ERR_ARG;
    *out = nullptr;
    const bool force_fft = (kind_and_flags & RSPT_HIP_DCT_FORCE_FFT) != 0;  // (test hook, see rspt_hip.h)
    const int kind = kind_and_flags & ~RSPT_HIP_DCT_FORCE_FFT;
    if (kind < 0 || kind > 3 || bps < 1 || bps > 4 || nch == 0 || ns == 0) return RSPT_HIP_ERR_ARG;
    if ((unsigned long long)nch * ns >= (1ull << 31)) return RSPT_HIP_ERR_ARG;  // the reference indexes with int
    if (kind == RSPT_HIP_KIND_XDELTA_HZR && (nb < 1 || nb > 4)) return RSPT_HIP_ERR_ARG;
    if (kind == RSPT_HIP_KIND_HADAMARD && (ns & (ns - 1))) return RSPT_HIP_ERR_ARG;  // fwht.c needs n = 2^k
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return RSPT_HIP_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return RSPT_HIP_ERR_NO_DEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return RSPT_HIP_ERR_NO_DEVICE;  // code objects are gfx950 only
    if (hipSetDevice(device) != hipSuccess) return RSPT_HIP_ERR_NO_DEVICE;

    rspt_hip_packer* p = new (std::nothrow) rspt_hip_packer();
    if (!p) return RSPT_HIP_ERR_ALLOC;
    p->device = device;
    Geom& g = p->g;
    g.bps = (uint32_t)bps;
    g.nch = (uint32_t)nch;
    g.ns = (uint32_t)ns;
    g.N = (uint32_t)(nch * ns);
    g.nblk = (g.N + kHzrBlock - 1) / kHzrBlock;
    g.kind = (uint32_t)kind;
    g.hdr_len = (kind == RSPT_HIP_KIND_DCT || kind == RSPT_HIP_KIND_HADAMARD) ? 3u * g.nch : 0u;
    g.method = kind == RSPT_HIP_KIND_DCT ? 1u : kind == RSPT_HIP_KIND_HADAMARD ? 2u : 0u;
    g.plane_stride = ((uint64_t)g.N + 255ull) & ~255ull;
    g.block_bytes = (uint64_t)bps * nch * ns;
    p->nb_host = p->nb_ctor = kind == RSPT_HIP_KIND_HZR ? 4u : kind == RSPT_HIP_KIND_DCT ? 2u : kind == RSPT_HIP_KIND_HADAMARD ? 3u : (unsigned)nb;

    // tile geometry
    {
        const uint64_t rowb = (uint64_t)g.nch * g.bps;
        const uint32_t ns16 = (g.ns + 15u) & ~15u;
        // k_tile_planar stages the contiguous input tile (T*nch*bps + 32 bytes) in LDS
        uint64_t t = (64 * 1024 - 32) / rowb;
        t &= ~15ull;
        if (t < 16) {  // a 16-sample tile of all channels does not fit: the wide-block front end (k_wide_planar), any packer
            p->wide = true;
            t = 16;
        }
        p->T = (uint32_t)(t > 4096 ? 4096 : t);
        if (p->T > ns16) p->T = ns16;
        p->in_lds = (uint32_t)(((uint64_t)p->T * rowb + 16 + 15) & ~15ull);
        // k_tile_planes keeps only the plane rows in LDS: kcount*nch*(Tp+16) + 32*nch bytes, <= 79 KiB
        // (two workgroups per CU).  Long row segments matter more than occupancy here: 256-byte plane
        // rows beat 64-byte ones by 2.5x (profiles/r01_tile_sweep.txt); segments that are whole 128-byte
        // lines beat ragged ones of about the same length (384 vs 352: -4 %).
        for (uint32_t kc = 1; kc <= 4; ++kc) {
            auto fit = [&](uint64_t budget) -> uint32_t {
                const uint64_t fixed = (16ull * kc + 32ull) * g.nch + 96;  // row padding, non-zero dedupe masks, escalation word + dirty bits
                if (budget <= fixed) return 0;
                uint64_t tt = (budget - fixed) / ((uint64_t)kc * g.nch);
                tt &= tt >= 256 ? ~127ull : ~15ull;
                return (uint32_t)(tt > 2048 ? 2048 : tt);
            };
            uint32_t Tp = fit(79 * 1024);
            if (Tp < 16) Tp = fit(150 * 1024);
#ifdef RSPT_DIAG
            if (const char* e = getenv("RSPT_TILE")) Tp = (uint32_t)atoi(e) & ~15u;  // tuning knob (diagnostic builds only)
#endif
            if (Tp < 16) {  // the plane rows of a 16-sample tile do not fit either way (about a thousand channels and up)
                p->wide = true;
                Tp = 16;
            }
            if (Tp > ns16) Tp = ns16;
            p->Tp[kc] = Tp;
        }
    }
#ifdef RSPT_DIAG  // timing probes and tuning knobs: never in the product library
    if (const char* e = getenv("RSPT_ABLATE")) p->ablate = (uint32_t)atoi(e);
    if (const char* e = getenv("RSPT_PLANESEL")) p->psel = (uint32_t)atoi(e);
    if (const char* e = getenv("RSPT_K1_THREADS")) p->k1_threads = (uint32_t)atoi(e) / 64 * 64;
    if (const char* e = getenv("RSPT_K1_GRID")) p->k1_grid = (uint32_t)atoi(e);
    if (const char* e = getenv("RSPT_HIST_GRID")) p->hist_grid = (uint32_t)atoi(e);
    if (const char* e = getenv("RSPT_ENC_GRID")) p->enc_grid = (uint32_t)atoi(e);
#endif
    if (p->k1_threads < 64 || p->k1_threads > 256) p->k1_threads = 256;  // (k_tile_planes is compiled for <= 256)
    p->ntile = (g.N + kInvTile - 1) / kInvTile;
    {
        // k_planar_native tile: nch rows of (T+1) int32 within 64 KiB
        uint32_t T = (uint32_t)(65536ull / (4ull * g.nch));
        T = T > 1 ? T - 1 : 0;
        if (T > 1024) T = 1024;
        if (T < 1) {
            delete p;
            return RSPT_HIP_ERR_UNSUPPORTED;
        }
        p->Tn_native = T;
    }
    if (kind == RSPT_HIP_KIND_HADAMARD && ns > (1u << 22)) {  // (up to 65536 one workgroup per channel; beyond, two passes: launch_fwht_big)
        delete p;
        return RSPT_HIP_ERR_UNSUPPORTED;
    }
    if (kind == RSPT_HIP_KIND_DCT) {
        // The reference's dense n x n table (bit-exact) for every n it can run itself -- its table index `(2x+1)*i` is an
        // int (signal_packer_dct.cpp:60-74): n <= 32768 -- except n = 2^k > 8192, which take the fp64 FFT path (PRDN / CR
        // tolerance) as do the sizes beyond the reference's reach, n = 2^k <= 2^22.  The table is n^2 floats twice over
        // (8.6 GB at 32768).  RSPT_HIP_DCT_FORCE_FFT forces the FFT path for small n = 2^k (cross-check against the table).
        const bool pow2 = (ns & (ns - 1)) == 0;
        p->dct_fft = (ns > 8192 && pow2) || (force_fft && pow2 && ns >= 16);
        if (p->dct_fft ? ns > (1u << 22) : ns > 32768) {
            delete p;
            return RSPT_HIP_ERR_UNSUPPORTED;
        }
        if (p->dct_fft) {
            uint32_t k = 0;
            while ((1ull << k) < ns) ++k;
            p->fft_l1 = (k + 1) / 2;
            p->fft_l2 = k / 2;
            bool real_form = true;
#ifdef RSPT_DIAG
            if (const char* noreal = getenv("RSPT_DCT_REAL")) real_form = atoi(noreal) != 0;  // complex-input form, for comparison
#endif
            if (k >= 8 && real_form) {  // n/2 = m1*m2 with m2 = 64 where it can be (whole-line output runs)
                const uint32_t lM = k - 1;
                p->fftr_lb = lM > 18 ? lM - 12 : 6;
                p->fftr_la = lM - p->fftr_lb;
                p->dct_real = true;
            }
        }
    }
    hipError_t e = hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete p;
        return RSPT_HIP_ERR_LAUNCH;
    }
    p->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (hipMalloc(&p->stamps, (512 * 16 * 8 + 2 * 16384) * sizeof(unsigned long long)) != hipSuccess) {
        rspt_hip_packer_destroy(p);
        return RSPT_HIP_ERR_ALLOC;
    }
    {
        // ~87 KB of constants, computed once per process (a function-local static: packers are created from several host
        // threads at once -- one per device, tests/cxx/shard_devices.cpp -- and the initialisation of such a static is thread-safe)
        static const CrcConsts& cc = *[] {
            CrcConsts* c = new CrcConsts;
            make_crc_consts(*c);
            return c;
        }();
        if (hipMalloc(&p->crc, sizeof(CrcConsts)) != hipSuccess || hipMalloc(&p->nb_state, 4 * sizeof(uint32_t)) != hipSuccess) {
            rspt_hip_packer_destroy(p);
            return RSPT_HIP_ERR_ALLOC;
        }
        uint32_t nb0 = p->nb_ctor;
        if (hipMemcpy(p->crc, &cc, sizeof(cc), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(p->nb_state, &nb0, sizeof(nb0), hipMemcpyHostToDevice) != hipSuccess) {
            rspt_hip_packer_destroy(p);
            return RSPT_HIP_ERR_LAUNCH;
        }
    }
    for (int i = 0; i <= ST_COUNT; ++i) hipEventCreate(&p->ev[i]);
    // (highest priority: the latency-bound small blocks go first and are done long before the big encoder, so that the
    // join at the end of the call finds its event complete)
    int prio_lo = 0, prio_hi = 0;
    hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    if (hipStreamCreateWithPriority(&p->side, hipStreamNonBlocking, prio_hi) != hipSuccess ||
        hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming) != hipSuccess) {
        rspt_hip_packer_destroy(p);
        return RSPT_HIP_ERR_LAUNCH;
    }
    if (kind == RSPT_HIP_KIND_DCT) {
        const double ratio1 = sqrt(2.0 / (double)(int)ns);
        const float cs0 = (float)(1 / sqrt(2));
        p->dct_cs0 = cs0;
        p->dct_scale0 = cs0 * ratio1 / 128.0;   // Cs[0]*ratio1/quality (dct.cpp:84)
        p->dct_scale1 = 1.0f * ratio1 / 128.0;  // Cs[i>0] = 1
        p->idct_scale = ratio1 * 128.0;          // dct.cpp:97
    }
    if (kind == RSPT_HIP_KIND_DCT && p->dct_fft) {
        const size_t n = ns;
        std::vector<double2> tw(n), post(n);
        const double PI = 3.14159265358979323846;
        for (size_t t = 0; t < n; ++t) {
            const double a = 2.0 * PI * (double)t / (double)n, b = PI * (double)t / (2.0 * (double)n);
            tw[t] = make_double2(cos(a), sin(a));
            post[t] = make_double2(cos(b), sin(b));
        }
        if (hipMalloc(&p->fft_tw, n * sizeof(double2)) != hipSuccess || hipMalloc(&p->fft_post, n * sizeof(double2)) != hipSuccess) {
            rspt_hip_packer_destroy(p);
            return RSPT_HIP_ERR_ALLOC;
        }
        if (hipMemcpy(p->fft_tw, tw.data(), n * sizeof(double2), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(p->fft_post, post.data(), n * sizeof(double2), hipMemcpyHostToDevice) != hipSuccess) {
            rspt_hip_packer_destroy(p);
            return RSPT_HIP_ERR_LAUNCH;
        }
    } else if (kind == RSPT_HIP_KIND_DCT) {
        // init_cos_table (signal_packer_dct.cpp:60-74), host libm, same expression and types.  Built in slabs of rows by
        // all host threads (10^9 cosines at n = 32768) and uploaded slab by slab; the transposed copy is made on the device.
        const size_t n = ns;
        if (hipMalloc(&p->cos_tab, n * n * sizeof(float)) != hipSuccess || hipMalloc(&p->cos_tab_t, n * n * sizeof(float)) != hipSuccess) {
            rspt_hip_packer_destroy(p);
            return RSPT_HIP_ERR_ALLOC;
        }
        const double PI = 3.14159265358979323846;
        const double pi_n_2 = PI / ((double)(int)n * 2.0);
        const size_t slab = std::max<size_t>(1, std::min<size_t>(n, (64u << 20) / (n * sizeof(float))));
        std::vector<float> tab(slab * n);
        unsigned nthr = std::thread::hardware_concurrency();
        nthr = nthr < 1 ? 1 : nthr > 32 ? 32 : nthr;
        for (size_t x0 = 0; x0 < n; x0 += slab) {
            const size_t nr = std::min(slab, n - x0);
            auto fill = [&](size_t r0, size_t r1) {
                for (size_t x = x0 + r0; x < x0 + r1; ++x)
                    for (size_t i = 0; i < n; ++i) {
                        const int arg = ((int)x << 1) * (int)i + (int)i;
                        tab[(x - x0) * n + i] = (float)cos(arg * pi_n_2);
                    }
            };
            if (nr * n < (1u << 20) || nthr == 1) {
                fill(0, nr);
            } else {
                std::vector<std::thread> th;
                for (unsigned t = 0; t < nthr; ++t) th.emplace_back(fill, nr * t / nthr, nr * (t + 1) / nthr);
                for (auto& t : th) t.join();
            }
            if (hipMemcpy(p->cos_tab + x0 * n, tab.data(), nr * n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
                rspt_hip_packer_destroy(p);
                return RSPT_HIP_ERR_LAUNCH;
            }
        }
        hipLaunchKernelGGL(k_transpose_f32, dim3((unsigned)((n + 31) / 32), (unsigned)((n + 31) / 32)), dim3(256), 0, 0, p->cos_tab, p->cos_tab_t,
                           (uint32_t)n);
        if (hipGetLastError() != hipSuccess) {
            rspt_hip_packer_destroy(p);
            return RSPT_HIP_ERR_LAUNCH;
        }
    }
    if (hipDeviceSynchronize() != hipSuccess) {  // setup copies ran on the null stream; the handle's stream does not wait for it
        rspt_hip_packer_destroy(p);
        return RSPT_HIP_ERR_LAUNCH;
    }
    *out = p;
    return RSPT_HIP_OK;
}

void rspt_hip_packer_destroy(rspt_hip_packer* p) {
    if (!p) return;
    hipSetDevice(p->device);
    if (p->stream) hipStreamSynchronize(p->stream);
    free_workspace(p);
    hipFree(p->stamps);
    hipFree(p->crc);
    hipFree(p->nb_state);
    hipFree(p->cos_tab);
    hipFree(p->cos_tab_t);
    hipFree(p->fft_tw);
    hipFree(p->fft_post);
    hipFree(p->h_src);
    hipFree(p->h_dst);
    hipFree(p->h_size);
    if (p->feed) rspt_hip_feed_end(p);
    free_many(p);
    hipFree(p->gat_totals);
    if (p->lag_stream) hipStreamSynchronize(p->lag_stream);
    for (int i = 0; i < 2; ++i) {
        hipFree(p->lag_dtotals[i]);
        if (p->lag_htotals[i]) hipHostFree(p->lag_htotals[i]);
        if (p->lag_ev_in[i]) hipEventDestroy(p->lag_ev_in[i]);
        if (p->lag_ev_sizes[i]) hipEventDestroy(p->lag_ev_sizes[i]);
        if (p->lag_ev_payload[i]) hipEventDestroy(p->lag_ev_payload[i]);
    }
    if (p->lag_stream) hipStreamDestroy(p->lag_stream);
    for (int i = 0; i <= ST_COUNT; ++i)
        if (p->ev[i]) hipEventDestroy(p->ev[i]);
    if (p->side) {
        hipStreamSynchronize(p->side);
        hipStreamDestroy(p->side);
    }
    if (p->h_nsmall) hipHostFree(p->h_nsmall);
    if (p->ev_fork) hipEventDestroy(p->ev_fork);
    if (p->ev_join) hipEventDestroy(p->ev_join);
    if (p->stream) hipStreamDestroy(p->stream);
    delete p;
}

size_t rspt_hip_block_bytes(const rspt_hip_packer* p) { return p ? (size_t)p->g.block_bytes : 0; }

size_t rspt_hip_max_compressed_size(const rspt_hip_packer* p) {
    if (!p) return 0;
    const unsigned nbmax = p->g.kind == RSPT_HIP_KIND_XDELTA_HZR ? 4u : p->nb_ctor;
    const size_t hzr_max = 4 + (size_t)p->g.N + 7ull * p->g.nblk;  // hzr_encode.c:489-497
    return 1 + p->g.hdr_len + (size_t)nbmax * (4 + hzr_max);
}

int rspt_hip_reserve(rspt_hip_packer* p, size_t max_blocks) {
    if (!p || max_blocks == 0) return RSPT_HIP_ERR_ARG;
    if (max_blocks <= p->cap_blocks) return RSPT_HIP_OK;
    if (max_blocks > 65535) return RSPT_HIP_ERR_ARG;  // grid.y / grid.z limit; shard larger batches
    HIPCHK(p, hipSetDevice(p->device));
    HIPCHK(p, hipStreamSynchronize(p->stream));
    free_workspace(p);
    const Geom& g = p->g;
    const size_t nhb = max_blocks * kMaxPlanes * g.nblk;
    bool ok = true;
    ok &= hipMalloc(&p->planes, max_blocks * kMaxPlanes * g.plane_stride + 4096) == hipSuccess;
    ok &= hipMalloc(&p->nbuse, max_blocks * sizeof(uint32_t)) == hipSuccess;
    ok &= hipMalloc(&p->dec_nb, max_blocks * sizeof(uint32_t)) == hipSuccess;
    ok &= hipMalloc(&p->plane_dirty, max_blocks * kMaxPlanes * 4 * sizeof(uint32_t)) == hipSuccess;
    // one region zeroed per call by a single memset: [nzflag: B*4*nblk][needmask: B][work counters: 16]; the last two are
    // placed per call right behind the part of nzflag in use
    p->zcap_words = nhb + max_blocks + 32 + 2 * max_blocks * (size_t)g.nch + 2;
    for (int i = 0; i < 2; ++i) {
        ok &= hipMalloc(&p->zbuf[i], p->zcap_words * sizeof(uint32_t)) == hipSuccess;
        p->zero_ready[i] = false;
    }
    p->zset = 0;
    p->nzflag = p->zbuf[0];
    ok &= hipMalloc(&p->big_list, nhb * sizeof(uint32_t)) == hipSuccess;
    ok &= hipMalloc(&p->small_list, nhb * sizeof(uint32_t)) == hipSuccess;
    ok &= hipMalloc(&p->hist, nhb * kSymStride * sizeof(uint32_t)) == hipSuccess;
    ok &= hipMalloc(&p->cw, nhb * kSymStride * sizeof(uint32_t)) == hipSuccess;
    ok &= hipMalloc(&p->seghist, nhb * (size_t)kSegHistStride * sizeof(uint16_t)) == hipSuccess;
    ok &= hipMalloc(&p->segbase, nhb * (size_t)kEncWaves * sizeof(uint32_t)) == hipSuccess;
    ok &= hipMalloc(&p->lists, nhb * (size_t)kEncWaves * kListCap * sizeof(uint32_t)) == hipSuccess;
    ok &= hipMalloc(&p->listinfo, nhb * (size_t)kEncWaves * sizeof(uint2)) == hipSuccess;
    ok &= hipMalloc(&p->tdesc, nhb * kTdescWords * sizeof(uint32_t)) == hipSuccess;
    ok &= hipMalloc(&p->meta, nhb * sizeof(BlockMeta)) == hipSuccess;
    ok &= hipMalloc(&p->out_off, nhb * sizeof(uint64_t)) == hipSuccess;
    ok &= hipMalloc(&p->means, max_blocks * (size_t)(g.hdr_len ? g.hdr_len : 4)) == hipSuccess;
    // planar int32 scratch: transform packers on compress, every packer on decompress
    ok &= hipMalloc(&p->planar, max_blocks * (size_t)g.N * sizeof(int32_t) + 4096) == hipSuccess;
    const size_t nscan = std::max<size_t>(p->ntile, g.N / kRowTile + 1);  // tiles of 4096, or row tiles of 256 (k_inv_native)
    ok &= hipMalloc(&p->txor, max_blocks * nscan * sizeof(uint32_t)) == hipSuccess;
    ok &= hipMalloc(&p->tsum, max_blocks * nscan * sizeof(uint32_t)) == hipSuccess;
    if (g.ns % kRowTile == 0 && g.bps == 4) ok &= hipMalloc(&p->rowrec, max_blocks * (g.N / kRowTile) * (size_t)kRowRec * sizeof(uint32_t)) == hipSuccess;
    ok &= hipMalloc(&p->blk_off, nhb * sizeof(uint64_t)) == hipSuccess;
    if (g.kind == RSPT_HIP_KIND_DCT) ok &= hipMalloc(&p->planar2, max_blocks * (size_t)g.N * sizeof(int32_t) + 4096) == hipSuccess;
    if (g.kind == RSPT_HIP_KIND_HADAMARD && g.ns > 65536u) ok &= hipMalloc(&p->mean_i32, max_blocks * (size_t)g.nch * sizeof(int32_t)) == hipSuccess;
    if (g.kind == RSPT_HIP_KIND_DCT && p->dct_fft) {
        const size_t per_block = (size_t)g.N * sizeof(double2);
        p->fft_bpp = std::max<size_t>(1, std::min<size_t>(max_blocks, ((size_t)1 << 30) / per_block));
        ok &= hipMalloc(&p->fft_scratch, p->fft_bpp * per_block) == hipSuccess;
        ok &= hipMalloc(&p->mean_i32, max_blocks * (size_t)g.nch * sizeof(int32_t)) == hipSuccess;
    }
    if (!ok) {
        free_workspace(p);
        return RSPT_HIP_ERR_ALLOC;
    }
    // the clean-plane invariant starts from zeroed planes
    HIPCHK(p, hipMemset(p->planes, 0, max_blocks * kMaxPlanes * g.plane_stride + 4096));
    HIPCHK(p, hipMemset(p->plane_dirty, 0, max_blocks * kMaxPlanes * 4 * sizeof(uint32_t)));
    p->dirty_shift = 0;
    while (((g.nblk - 1) >> p->dirty_shift) >= 128u) ++p->dirty_shift;
    HIPCHK(p, hipDeviceSynchronize());  // (the calls that follow may come on any stream)
    p->planes_unknown = false;
    p->cap_blocks = max_blocks;
    return RSPT_HIP_OK;
}

// ---- the compress sequence, phase by phase ----------------------------------------------------------------------------------------
// (Round 4 measured whether the phases of TWO batches can overlap -- two whole batches racing on two streams, and an ordered
// schedule with the latency-bound middle of batch i on a second stream beside the front end of batch i+1: neither beats one
// batch at a time on one stream; profiles/r04_notes.md, profiles/r04_pipeline_experiment.patch.)
// front:  zero region, front-end kernel(s), escalation scan / fix-up, list of k_hist's blocks     (HBM-bound)
// hist:   k_hist                                                                                  (vector-issue bound)
// tree:   k_tree + k_layout                                                                       (latency chains, chip mostly idle)
// small:  k_encode_small                                                                          (latency-bound, one wave per block)
// encode: k_encode                                                                                (vector-issue bound)
static int phase_front(rspt_hip_packer* p, const uint8_t* src, size_t nblocks, hipStream_t st) {
    const Geom& g = p->g;
    const uint32_t B = (uint32_t)nblocks;
    stamp(p, ST_PRE, st);
    const bool xd = g.kind == RSPT_HIP_KIND_XDELTA_HZR;
    {
        const size_t nhb_call = nblocks * kMaxPlanes * g.nblk;
        p->nzflag = p->zbuf[p->zset];
        p->needmask = p->nzflag + nhb_call;
        p->work_ctr = p->needmask + ((nblocks + 3) & ~(size_t)3);
        size_t zwords = (size_t)((p->work_ctr + 16) - p->nzflag);
        p->row_sum = nullptr;
        p->have_row_sum = false;
        if (g.kind == RSPT_HIP_KIND_DCT && p->dct_fft) {  // channel sums of the de-interleave pass, 8-byte aligned behind the counters
            zwords = (zwords + 1) & ~(size_t)1;
            p->row_sum = reinterpret_cast<long long*>(p->nzflag + zwords);
            zwords += 2 * nblocks * (size_t)g.nch;
        }
        if (!p->zero_ready[p->zset]) HIPCHK(p, hipMemsetAsync(p->nzflag, 0, zwords * sizeof(uint32_t), st));  // (first call, or after a failed one)
        p->zero_ready[0] = p->zero_ready[1] = false;  // this copy is in use now; the other one becomes ready once k_tree is launched
    }
    if (p->planes_unknown || (p->ablate & ~(3u << 26)) || p->psel) {  // (diagnostic runs skip kernels and stores: never trust the planes they leave; probes 26 / 27 store everything)
        HIPCHK(p, hipMemsetAsync(p->plane_dirty, 0xFF, p->cap_blocks * kMaxPlanes * 4 * sizeof(uint32_t), st));
        p->planes_unknown = false;
    }
    uint32_t np = 4;
    switch (g.bps) {
        case 1: np = launch_front<1>(p, src, nblocks, st); break;
        case 2: np = launch_front<2>(p, src, nblocks, st); break;
        case 3: np = launch_front<3>(p, src, nblocks, st); break;
        default: np = launch_front<4>(p, src, nblocks, st); break;
    }
    if (g.kind == RSPT_HIP_KIND_HADAMARD) {
        // per channel: mean removal, WHT, truncating /n (signal_packer_hadamard.cpp:57-72)
        const uint32_t fw_lds = (g.ns > 32768u ? 32768u : g.ns) * 4u;
        if (g.ns > 65536u) {  // two passes over the planar row (any 2^k the reference's own transform takes, fwht.c:4-28)
            hipLaunchKernelGGL(k_row_means, dim3(g.nch, B), dim3(1024), 0, st, p->planar, g, p->means, p->mean_i32);
            launch_fwht_big<true>(p, B, st);
            hipLaunchKernelGGL((k_planar_planes<false>), dim3((g.N + 4095) / 4096, B), dim3(256), 0, st, p->planar, g, 3u, p->planes, p->nzflag, (uint32_t*)nullptr);
        } else if (g.ns == 65536u) {  // the whole row in registers: read once, and the byte planes written straight from them
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fwht64k<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fw_lds);
            hipLaunchKernelGGL((k_fwht64k<true, true>), dim3(g.nch, B), dim3(1024), fw_lds, st, p->planar, g, p->means, p->planes, p->nzflag, 3u);
        } else {
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fwht<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fw_lds);
            hipLaunchKernelGGL((k_fwht<true>), dim3(g.nch, B), dim3(1024), fw_lds, st, p->planar, g, p->means);
            hipLaunchKernelGGL((k_planar_planes<false>), dim3((g.N + 4095) / 4096, B), dim3(256), 0, st, p->planar, g, 3u, p->planes, p->nzflag, (uint32_t*)nullptr);
        }
    } else if (g.kind == RSPT_HIP_KIND_DCT) {
        if (p->dct_fft) {
            if (p->have_row_sum)  // the de-interleave pass summed the channels on its way: no second pass over the planar block
                hipLaunchKernelGGL(k_means_from_sums, dim3((B * g.nch + 255) / 256), dim3(256), 0, st, p->row_sum, g, B, p->means, p->mean_i32);
            else
                hipLaunchKernelGGL(k_row_means, dim3(g.nch, B), dim3(1024), 0, st, p->planar, g, p->means, p->mean_i32);
            launch_dct_fft<true>(p, B, p->planar, p->planar2, st);
        } else {
            hipLaunchKernelGGL((k_dct<true>), dim3((g.ns + 255) / 256, (g.nch + kDctCh - 1) / kDctCh, B), dim3(256), 0, st, p->planar, g, p->means,
                               p->cos_tab, p->dct_scale0, p->dct_scale1, p->dct_cs0, p->planar2);
        }
        hipLaunchKernelGGL((k_planar_planes<true>), dim3((g.N + 4095) / 4096, B), dim3(256), 0, st, p->planar2, g, 2u, p->planes, p->nzflag, (uint32_t*)nullptr);
    }
    HIPCHK(p, hipGetLastError());

    stamp(p, ST_NB, st);
    if (g.kind != RSPT_HIP_KIND_XDELTA_HZR && g.kind != RSPT_HIP_KIND_HZR)  // (k_tile_planes runs the scan in its last workgroup)
        hipLaunchKernelGGL(k_nb_scan, dim3(1), dim3(1024), 0, st, p->needmask, B, p->nb_state, p->nbuse, 0);
    if (xd && np < 4) {  // nb may have escalated in this call: add the planes the main pass did not write
        switch (g.bps) {
            case 1: launch_fixup<1>(p, src, nblocks, np, st); break;
            case 2: launch_fixup<2>(p, src, nblocks, np, st); break;
            case 3: launch_fixup<3>(p, src, nblocks, np, st); break;
            default: launch_fixup<4>(p, src, nblocks, np, st); break;
        }
    }
    stamp(p, ST_HIST, st);
    // (the list of k_hist's blocks sits in big_list until k_layout refills that array for k_encode; its count in work_ctr[2])
    const uint32_t nhb = B * kMaxPlanes * g.nblk;
    hipLaunchKernelGGL(k_histlist, dim3((nhb + 255) / 256), dim3(256), 0, st, p->nzflag, p->nbuse, g, nhb, p->big_list, p->work_ctr + 2, p->psel);
    HIPCHK(p, hipGetLastError());
    return RSPT_HIP_OK;
}

static uint32_t persistent_grid(const rspt_hip_packer* p, uint32_t nhb, uint32_t knob) {
    const uint32_t persist = (uint32_t)(2 * p->num_cu) < nhb ? (uint32_t)(2 * p->num_cu) : nhb;  // 2 x 1024 threads fill a CU
    return knob && knob < persist ? knob : persist;
}

static int phase_hist(rspt_hip_packer* p, uint32_t B, hipStream_t st) {
    const Geom& g = p->g;
    const uint32_t nhb = B * kMaxPlanes * g.nblk;
    hipLaunchKernelGGL(k_hist, dim3(persistent_grid(p, nhb, p->hist_grid)), dim3(kEncThreads), 0, st, p->planes, g, p->nzflag, p->hist, p->seghist, p->work_ctr,
                       p->big_list, p->work_ctr + 2, p->lists, p->listinfo);
    HIPCHK(p, hipGetLastError());  // (a failing launch is reported at its own stage)
    return RSPT_HIP_OK;
}

static int phase_tree(rspt_hip_packer* p, uint32_t B, hipStream_t st) {
    const Geom& g = p->g;
    const uint32_t nhb = B * kMaxPlanes * g.nblk;
    hipLaunchKernelGGL(k_tree, dim3((nhb + 3) / 4), dim3(256), 0, st, p->hist, p->planes, g, p->nbuse, p->nzflag, nhb, p->cw, p->tdesc, p->meta, p->seghist, p->segbase,
                       p->zbuf[p->zset ^ 1], (uint32_t)p->zcap_words, p->psel);
    HIPCHK(p, hipGetLastError());
    return RSPT_HIP_OK;
}

static int phase_layout(rspt_hip_packer* p, uint32_t B, void* d_dst, size_t dst_stride, uint64_t* d_sizes, hipStream_t st) {
    const Geom& g = p->g;
    WorkQueues* wq = reinterpret_cast<WorkQueues*>(p->work_ctr + 4);
    hipLaunchKernelGGL(k_layout, dim3(B), dim3(256), 0, st, g, p->nbuse, p->meta, p->means, (uint8_t*)d_dst, (uint64_t)dst_stride, p->out_off,
                       d_sizes, p->crc, p->nzflag, wq, p->big_list, p->small_list, p->plane_dirty, p->dirty_shift, p->psel);
    HIPCHK(p, hipGetLastError());
    return RSPT_HIP_OK;
}

static int phase_small(rspt_hip_packer* p, uint32_t B, void* d_dst, size_t dst_stride, hipStream_t ss) {
    const Geom& g = p->g;
    const uint32_t nhb = B * kMaxPlanes * g.nblk;
    WorkQueues* wq = reinterpret_cast<WorkQueues*>(p->work_ctr + 4);
    const uint32_t want = (nhb + kSmallWaves - 1) / kSmallWaves;
    const uint32_t sgrid = (uint32_t)(6 * p->num_cu) < want ? (uint32_t)(6 * p->num_cu) : want;  // ~22 KiB of LDS per workgroup
    hipLaunchKernelGGL(k_encode_small, dim3(sgrid), dim3(kSmallWaves * 64), 0, ss, p->planes, g, p->nzflag, p->meta, p->cw, p->tdesc, p->out_off,
                       p->crc, (uint8_t*)d_dst, (uint64_t)dst_stride, wq, p->small_list, p->ablate, p->h_nsmall);
    HIPCHK(p, hipGetLastError());
    return RSPT_HIP_OK;
}

// `yield`: an eighth of the grid steps aside for k_encode_small running beside it on the side stream
static int phase_encode(rspt_hip_packer* p, uint32_t B, void* d_dst, size_t dst_stride, hipStream_t st, uint32_t yield) {
    const Geom& g = p->g;
    const uint32_t nhb = B * kMaxPlanes * g.nblk;
    WorkQueues* wq = reinterpret_cast<WorkQueues*>(p->work_ctr + 4);
    hipLaunchKernelGGL(k_encode, dim3(persistent_grid(p, nhb, p->enc_grid)), dim3(kEncThreads), 0, st, p->planes, g, p->nzflag, p->meta, p->cw, p->tdesc, p->out_off,
                       p->crc, (uint8_t*)d_dst, (uint64_t)dst_stride, wq, p->big_list, p->segbase, p->lists, p->listinfo, p->stamps, yield);
    HIPCHK(p, hipGetLastError());
    return RSPT_HIP_OK;
}

static int ensure_nsmall(rspt_hip_packer* p) {
    if (!p->h_nsmall) {
        if (hipHostMalloc((void**)&p->h_nsmall, sizeof(uint32_t), hipHostMallocMapped) != hipSuccess) return RSPT_HIP_ERR_ALLOC;
        *p->h_nsmall = 0xFFFFFFFFu;  // (unknown yet)
    }
    return RSPT_HIP_OK;
}

// one batch, start to end on one stream (the small-block encoder beside the big one on the handle's side stream)
static int compress_batch_serial(rspt_hip_packer* p, const void* d_src, size_t nblocks, void* d_dst, size_t dst_stride, uint64_t* d_sizes, hipStream_t st) {
    int rc = rspt_hip_reserve(p, nblocks);
    if (rc) return rc;
    HIPCHK(p, hipSetDevice(p->device));
    const uint32_t B = (uint32_t)nblocks;
    if ((rc = phase_front(p, (const uint8_t*)d_src, nblocks, st)) != 0) return rc;
    if ((rc = phase_hist(p, B, st)) != 0) return rc;

    stamp(p, ST_TREE, st);
    if (p->psel & 256u) {  // (diagnostic builds only: time the front end and k_hist alone)
        for (int i = ST_TREE + 1; i <= ST_COUNT; ++i) stamp(p, i, st);
        if (p->profiling) p->ev_valid = true;
        return RSPT_HIP_OK;
    }
    if ((rc = phase_tree(p, B, st)) != 0) return rc;
    const int zset_next = p->zset ^ 1;

    stamp(p, ST_LAYOUT, st);
    if (p->psel & 512u) {  // (diagnostic builds only: stop behind k_tree)
        for (int i = ST_LAYOUT + 1; i <= ST_COUNT; ++i) stamp(p, i, st);
        if (p->profiling) p->ev_valid = true;
        p->zero_ready[zset_next] = true;
        p->zset = zset_next;
        return RSPT_HIP_OK;
    }
    if ((rc = phase_layout(p, B, d_dst, dst_stride, d_sizes, st)) != 0) return rc;

    stamp(p, ST_ENCODE, st);
    if ((rc = ensure_nsmall(p)) != 0) return rc;
    // Both encoders depend on k_layout only.  The small-block one goes to the side stream (the big one yields it room) -- unless the
    // recent batches of this handle held no small blocks at all: the fork and join of a second stream cost ~10 us, an empty
    // kernel in line 2.  The guess only decides where the kernel runs.
    const bool side = *(volatile uint32_t*)p->h_nsmall != 0u;
    hipStream_t ss = side ? p->side : st;
    if (side) {
        HIPCHK(p, hipEventRecord(p->ev_fork, st));
        HIPCHK(p, hipStreamWaitEvent(p->side, p->ev_fork, 0));
    }
    if ((rc = phase_small(p, B, d_dst, dst_stride, ss)) != 0) return rc;
    if (side) HIPCHK(p, hipEventRecord(p->ev_join, p->side));
    if ((rc = phase_encode(p, B, d_dst, dst_stride, st, 1u)) != 0) return rc;
    stamp(p, ST_ENCODE_SMALL, st);
    if (side) HIPCHK(p, hipStreamWaitEvent(st, p->ev_join, 0));
    stamp(p, ST_COUNT, st);
    if (p->profiling) p->ev_valid = true;
    HIPCHK(p, hipGetLastError());
    p->zero_ready[zset_next] = true;  // every launch went out: the other copy is zero when the next call starts
    p->zset = zset_next;
    return RSPT_HIP_OK;
}

int rspt_hip_compress_batch_dev(rspt_hip_packer* p, const void* d_src, size_t nblocks, void* d_dst, size_t dst_stride, uint64_t* d_sizes,
                                void* stream) {
    if (!p || !d_src || !d_dst || !d_sizes || nblocks == 0) return RSPT_HIP_ERR_ARG;
    if (reinterpret_cast<uintptr_t>(d_src) & 15) return RSPT_HIP_ERR_ARG;  // tile loads are 16-byte aligned chunks
    if (p->feed) return RSPT_HIP_ERR_ARG;  // (the feed owns the handle's workspace until rspt_hip_feed_end)
    return compress_batch_serial(p, d_src, nblocks, d_dst, dst_stride, d_sizes, (hipStream_t)stream);
}

size_t rspt_hip_pack_bound(const rspt_hip_packer* p, size_t nblocks) {
    if (!p) return 0;
    return 32 + 16 * nblocks + nblocks * ((rspt_hip_max_compressed_size(p) + 15) & ~(size_t)15);
}

int rspt_hip_pack_batch_dev(rspt_hip_packer* p, const void* d_dst, size_t dst_stride, const uint64_t* d_sizes, size_t nblocks, void* d_packed,
                            uint64_t* d_total, void* stream) {
    if (!p || !d_dst || !d_sizes || !d_packed || !d_total || nblocks == 0 || nblocks > 65535) return RSPT_HIP_ERR_ARG;
    if (nblocks > p->cap_blocks) return RSPT_HIP_ERR_ARG;  // the per-stream nb comes from the handle's last compress call of >= nblocks blocks
    if ((reinterpret_cast<uintptr_t>(d_dst) & 15) || (dst_stride & 15) || (reinterpret_cast<uintptr_t>(d_packed) & 15)) return RSPT_HIP_ERR_ARG;
    HIPCHK(p, hipSetDevice(p->device));
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_pack_index, dim3(1), dim3(1024), 0, st, d_sizes, (uint32_t)nblocks, p->nb_state, p->nbuse, (uint8_t*)d_packed, d_total);
    hipLaunchKernelGGL(k_pack_copy, dim3(32, (unsigned)nblocks), dim3(256), 0, st, (const uint8_t*)d_dst, (uint64_t)dst_stride, (uint32_t)nblocks,
                       (uint8_t*)d_packed);
    HIPCHK(p, hipGetLastError());
    return RSPT_HIP_OK;
}

void* rspt_hip_stream(rspt_hip_packer* p) { return p ? (void*)p->stream : nullptr; }

int rspt_hip_synchronize(rspt_hip_packer* p) {
    if (!p) return RSPT_HIP_ERR_ARG;
    HIPCHK(p, hipSetDevice(p->device));
    HIPCHK(p, hipStreamSynchronize(p->stream));
    return RSPT_HIP_OK;
}

unsigned rspt_hip_current_nb(rspt_hip_packer* p) {
    if (!p) return 0;
    hipSetDevice(p->device);
    hipDeviceSynchronize();
    uint32_t nb = 0;
    if (hipMemcpy(&nb, p->nb_state, sizeof(nb), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    if (nb >= 1 && nb <= 4) p->nb_host = nb;
    return nb;
}

int rspt_hip_set_nb(rspt_hip_packer* p, unsigned nb) {
    if (!p || nb < 1 || nb > 4) return RSPT_HIP_ERR_ARG;
    if (p->g.kind != RSPT_HIP_KIND_XDELTA_HZR) return RSPT_HIP_ERR_ARG;
    HIPCHK(p, hipSetDevice(p->device));
    HIPCHK(p, hipDeviceSynchronize());
    uint32_t v = nb;
    HIPCHK(p, hipMemcpy(p->nb_state, &v, sizeof(v), hipMemcpyHostToDevice));
    p->nb_host = nb;
    return RSPT_HIP_OK;
}

int rspt_hip_set_byte_order(rspt_hip_packer* p, int big_endian) {
    if (!p) return RSPT_HIP_ERR_ARG;
    p->big_endian = big_endian ? 1 : 0;
    p->g.be = p->big_endian && p->g.bps > 1 ? 1u : 0u;  // compress: every front end reverses the bytes of a sample as it reads it (no extra pass)
    return RSPT_HIP_OK;
}

void* rspt_hip_host_alloc(size_t bytes) {
    void* q = nullptr;
    if (bytes == 0 || hipHostMalloc(&q, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    return q;
}

void rspt_hip_host_free(void* q) {
    if (q) hipHostFree(q);
}

int rspt_hip_set_verify(rspt_hip_packer* p, int on) {
    if (!p) return RSPT_HIP_ERR_ARG;
    p->verify = on ? 1 : 0;
    return RSPT_HIP_OK;
}

static int ensure_host_staging(rspt_hip_packer* p) {
    const size_t need_dst = rspt_hip_max_compressed_size(p) + 64;
    if (!p->h_src) {
        if (hipMalloc(&p->h_src, p->g.block_bytes + 64) != hipSuccess) return RSPT_HIP_ERR_ALLOC;
        hipMemset(p->h_src, 0, p->g.block_bytes + 64);
        hipDeviceSynchronize();  // the memset runs on the null stream; our copies use a non-blocking stream
    }
    if (!p->h_size && hipMalloc(&p->h_size, sizeof(uint64_t)) != hipSuccess) return RSPT_HIP_ERR_ALLOC;
    if (p->h_dst_cap < need_dst) {
        hipFree(p->h_dst);
        p->h_dst = nullptr;
        if (hipMalloc(&p->h_dst, need_dst) != hipSuccess) return RSPT_HIP_ERR_ALLOC;
        p->h_dst_cap = need_dst;
    }
    return RSPT_HIP_OK;
}

// Page-locked host memory (rspt_hip_host_alloc, hipHostMalloc, hipHostRegister) is visible to the device: returns its device
// address, or nullptr for pageable memory (which has to be staged).
static void* device_view_of_host(const void* host_ptr) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, host_ptr) != hipSuccess) {
        (void)hipGetLastError();  // (pageable memory: not an error of ours)
        return nullptr;
    }
    if (a.type != hipMemoryTypeHost || !a.devicePointer) return nullptr;
    return a.devicePointer;
}

int rspt_hip_compress(rspt_hip_packer* p, const void* src_host, void* dst_host, size_t dst_max_len, size_t* dst_len) {
    if (!p || !src_host || !dst_host || !dst_len) return RSPT_HIP_ERR_ARG;
    if (p->feed) return RSPT_HIP_ERR_ARG;
    HIPCHK(p, hipSetDevice(p->device));
    int rc = ensure_host_staging(p);
    if (rc) return rc;
    // With page-locked buffers neither copy is a phase of its own: the front end reads the samples across the link as it
    // transforms them, the encoders write the stream straight into the caller's buffer -- one synchronisation at the end.
    // Pageable buffers go through the device staging copies as before.
    uint8_t* d_dst = dst_max_len >= 64 ? (uint8_t*)device_view_of_host(dst_host) : nullptr;
    // A page-locked, 16-byte aligned source is read in place by the front end (the upload IS the front end, at ~44 GB/s of 4-byte
    // loads across the link: 0.513 ms per 16 MiB block against 0.551 with a copy phase; measured and dropped: the upload cut
    // into four sample ranges on the copy stream with a front-end launch behind each range's event -- 0.627 ms, every
    // cross-stream dependency costs 25-60 us on this runtime).
    const uint8_t* d_src = (const uint8_t*)device_view_of_host(src_host);
    if (d_src && (reinterpret_cast<uintptr_t>(d_src) & 15)) d_src = nullptr;  // (tile loads are 16-byte aligned chunks)
    if (!d_src) {
        HIPCHK(p, hipMemcpyAsync(p->h_src, src_host, p->g.block_bytes, hipMemcpyHostToDevice, p->stream));
        d_src = p->h_src;
    }
    rc = compress_batch_serial(p, d_src, 1, d_dst ? d_dst : p->h_dst, d_dst ? dst_max_len : p->h_dst_cap, p->h_size, p->stream);
    if (rc) return rc;
    uint64_t sz = 0;
    uint32_t nb_now = 0;
    HIPCHK(p, hipMemcpyAsync(&sz, p->h_size, sizeof(sz), hipMemcpyDeviceToHost, p->stream));
    HIPCHK(p, hipMemcpyAsync(&nb_now, p->nb_state, sizeof(nb_now), hipMemcpyDeviceToHost, p->stream));
    HIPCHK(p, hipStreamSynchronize(p->stream));
    if (nb_now >= 1 && nb_now <= 4) p->nb_host = nb_now;  // the next call writes exactly the planes it needs
    if (sz >> 63) {  // did not fit the space the kernels were given (nothing written): the size it needs is in the low bits
        *dst_len = (size_t)(sz & ~(1ull << 63));
        return RSPT_HIP_ERR_DST_TOO_SMALL;
    }
    if (sz > dst_max_len) {
        *dst_len = (size_t)sz;
        return RSPT_HIP_ERR_DST_TOO_SMALL;
    }
    if (!d_dst) {
        HIPCHK(p, hipMemcpyAsync(dst_host, p->h_dst, (size_t)sz, hipMemcpyDeviceToHost, p->stream));
        HIPCHK(p, hipStreamSynchronize(p->stream));
    }
    *dst_len = (size_t)sz;
    return RSPT_HIP_OK;
}

static int ensure_many(rspt_hip_packer* p) {
    if (p->m_chunk) return RSPT_HIP_OK;
    // ~64 MiB of samples per chunk: long enough copies for the DMA engines, short enough that the pipeline fills quickly
    size_t chunk = (64ull << 20) / p->g.block_bytes;
    chunk = chunk < 1 ? 1 : chunk > 64 ? 64 : chunk;
    const size_t stride = (rspt_hip_max_compressed_size(p) + 255) & ~(size_t)255;
    bool ok = true;
    for (int i = 0; i < 2; ++i) {
        ok &= hipMalloc(&p->m_src[i], chunk * p->g.block_bytes + 64) == hipSuccess;
        ok &= hipMalloc(&p->m_dst[i], chunk * stride) == hipSuccess;
        ok &= hipMalloc(&p->m_sizes[i], chunk * sizeof(uint64_t)) == hipSuccess;
        ok &= hipEventCreateWithFlags(&p->m_ev_up[i], hipEventDisableTiming) == hipSuccess;
        ok &= hipEventCreateWithFlags(&p->m_ev_comp[i], hipEventDisableTiming) == hipSuccess;
        ok &= hipEventCreateWithFlags(&p->m_ev_down[i], hipEventDisableTiming) == hipSuccess;
    }
    for (int i = 0; i < 2; ++i) ok &= hipMalloc(&p->m_idx[i], (4 + 2 * chunk) * sizeof(uint64_t)) == hipSuccess;
    ok &= hipHostMalloc((void**)&p->m_hsizes, 2 * chunk * sizeof(uint64_t), hipHostMallocDefault) == hipSuccess;
    ok &= hipHostMalloc((void**)&p->m_hidx, 2 * (4 + 2 * chunk) * sizeof(uint64_t), hipHostMallocDefault) == hipSuccess;
    if (!p->m_up) ok &= hipStreamCreateWithFlags(&p->m_up, hipStreamNonBlocking) == hipSuccess;  // (rspt_hip_feed_begin may have made them)
    if (!p->m_down) ok &= hipStreamCreateWithFlags(&p->m_down, hipStreamNonBlocking) == hipSuccess;
    if (!ok) {  // nothing half-made stays behind: a retry starts from null fields instead of allocating over live pointers
        free_many(p);
        return RSPT_HIP_ERR_ALLOC;
    }
    p->m_chunk = chunk;
    p->m_stride = stride;
    return rspt_hip_reserve(p, chunk);
}

static int compress_many_pipeline(rspt_hip_packer* p, const void* src_host, size_t nblocks, void* dst_host, size_t dst_stride, size_t* dst_len);

int rspt_hip_compress_many(rspt_hip_packer* p, const void* src_host, size_t nblocks, void* dst_host, size_t dst_stride, size_t* dst_len) {
    if (!p || !src_host || !dst_host || !dst_len || nblocks == 0) return RSPT_HIP_ERR_ARG;
    if (p->feed) return RSPT_HIP_ERR_ARG;  // (the feed owns the workspace and the copy streams until rspt_hip_feed_end)
    HIPCHK(p, hipSetDevice(p->device));
    int rc = ensure_many(p);
    if (rc) return rc;
    rc = compress_many_pipeline(p, src_host, nblocks, dst_host, dst_stride, dst_len);
    if (rc != RSPT_HIP_OK && rc != RSPT_HIP_ERR_DST_TOO_SMALL) {
        // a failure in the middle: nothing may still be copying from or into the caller's buffers when we return
        hipStreamSynchronize(p->m_up);
        hipStreamSynchronize(p->stream);
        hipStreamSynchronize(p->m_down);
    }
    return rc;
}

static int compress_many_pipeline(rspt_hip_packer* p, const void* src_host, size_t nblocks, void* dst_host, size_t dst_stride, size_t* dst_len) {
    int rc = RSPT_HIP_OK;
    const size_t C = p->m_chunk, bb = p->g.block_bytes;
    const size_t nchunk = (nblocks + C - 1) / C;
    const uint8_t* src = (const uint8_t*)src_host;
    uint8_t* dst = (uint8_t*)dst_host;
    bool too_small = false;
    // the streams of chunk k leave for the host (exact lengths: its sizes have to be here first)
    auto download = [&](size_t k) -> int {
        const int slot = (int)(k & 1);
        const size_t first = k * C, cnt = nblocks - first < C ? nblocks - first : C;
        HIPCHK(p, hipEventSynchronize(p->m_ev_comp[slot]));
        const uint64_t* hs = p->m_hsizes + (size_t)slot * C;
        for (size_t i = 0; i < cnt; ++i) {
            const uint64_t sz = hs[i];
            if ((sz >> 63) || sz > dst_stride) {  // flagged by the device (did not fit the staging stride), or too long for the caller's
                dst_len[first + i] = (sz >> 63) ? 0 : (size_t)sz;
                too_small = true;
                continue;
            }
            dst_len[first + i] = (size_t)sz;
            HIPCHK(p, hipMemcpyAsync(dst + (first + i) * dst_stride, p->m_dst[slot] + i * p->m_stride, (size_t)sz, hipMemcpyDeviceToHost, p->m_down));
        }
        HIPCHK(p, hipEventRecord(p->m_ev_down[slot], p->m_down));
        return RSPT_HIP_OK;
    };
    for (size_t k = 0; k < nchunk; ++k) {
        const int slot = (int)(k & 1);
        const size_t first = k * C, cnt = nblocks - first < C ? nblocks - first : C;
        if (k >= 2) {
            HIPCHK(p, hipStreamWaitEvent(p->m_up, p->m_ev_comp[slot], 0));     // chunk k-2 has been read out of this slot
            HIPCHK(p, hipStreamWaitEvent(p->stream, p->m_ev_down[slot], 0));  // ... and its streams have left it
        }
        HIPCHK(p, hipMemcpyAsync(p->m_src[slot], src + first * bb, cnt * bb, hipMemcpyHostToDevice, p->m_up));
        HIPCHK(p, hipEventRecord(p->m_ev_up[slot], p->m_up));
        HIPCHK(p, hipStreamWaitEvent(p->stream, p->m_ev_up[slot], 0));
        rc = compress_batch_serial(p, p->m_src[slot], cnt, p->m_dst[slot], p->m_stride, p->m_sizes[slot], p->stream);
        if (rc) return rc;
        HIPCHK(p, hipMemcpyAsync(p->m_hsizes + (size_t)slot * C, p->m_sizes[slot], cnt * sizeof(uint64_t), hipMemcpyDeviceToHost, p->stream));
        HIPCHK(p, hipEventRecord(p->m_ev_comp[slot], p->stream));
        if (k >= 1) {
            rc = download(k - 1);
            if (rc) return rc;
        }
    }
    rc = download(nchunk - 1);
    if (rc) return rc;
    uint32_t nb_now = 0;
    HIPCHK(p, hipMemcpyAsync(&nb_now, p->nb_state, sizeof(nb_now), hipMemcpyDeviceToHost, p->stream));
    HIPCHK(p, hipStreamSynchronize(p->stream));
    HIPCHK(p, hipStreamSynchronize(p->m_down));
    if (nb_now >= 1 && nb_now <= 4) p->nb_host = nb_now;
    return too_small ? RSPT_HIP_ERR_DST_TOO_SMALL : RSPT_HIP_OK;
}

// ---- rspt_hip_feed_*: blocks that arrive over time ---------------------------------------------------------------------------
struct FeedSlot {
    enum State { FREE, FILLING, COMPRESSING, DOWNLOADING, DONE } state = FREE;
    uint8_t* d_src = nullptr;
    uint8_t* d_dst = nullptr;
    uint64_t* d_sizes = nullptr;
    uint64_t* h_sizes = nullptr;  // page-locked: [G] stream lengths + [1] nb_state behind this group
    std::vector<void*> dst_host;
    std::vector<size_t> dst_cap;
    size_t count = 0, delivered = 0, first_seq = 0;
    int error = 0;  // the group's launch failed: every block of it is reported with this status
    hipEvent_t ev_up = nullptr, ev_comp = nullptr, ev_down = nullptr;
};
struct Feed {
    size_t G = 0, stride = 0;
    std::vector<FeedSlot> slots;
    size_t head = 0, tail = 0;  // ring positions: oldest slot not yet FREE; the slot being filled / filled next
    size_t next_seq = 0;
};

static void feed_free(rspt_hip_packer* p) {
    Feed* f = p->feed;
    if (!f) return;
    for (auto& s : f->slots) {
        hipFree(s.d_src);
        hipFree(s.d_dst);
        hipFree(s.d_sizes);
        if (s.h_sizes) hipHostFree(s.h_sizes);
        if (s.ev_up) hipEventDestroy(s.ev_up);
        if (s.ev_comp) hipEventDestroy(s.ev_comp);
        if (s.ev_down) hipEventDestroy(s.ev_down);
    }
    delete f;
    p->feed = nullptr;
}

int rspt_hip_feed_begin(rspt_hip_packer* p, size_t blocks_per_launch, size_t slots) {
    if (!p || blocks_per_launch == 0 || blocks_per_launch > 4096 || slots < 2 || slots > 64 || p->feed) return RSPT_HIP_ERR_ARG;
    HIPCHK(p, hipSetDevice(p->device));
    if (!p->m_up && hipStreamCreateWithFlags(&p->m_up, hipStreamNonBlocking) != hipSuccess) return RSPT_HIP_ERR_ALLOC;
    if (!p->m_down && hipStreamCreateWithFlags(&p->m_down, hipStreamNonBlocking) != hipSuccess) return RSPT_HIP_ERR_ALLOC;
    int rc = rspt_hip_reserve(p, blocks_per_launch);
    if (rc) return rc;
    Feed* f = new (std::nothrow) Feed();
    if (!f) return RSPT_HIP_ERR_ALLOC;
    p->feed = f;
    f->G = blocks_per_launch;
    f->stride = (rspt_hip_max_compressed_size(p) + 255) & ~(size_t)255;
    f->slots.resize(slots);
    bool ok = true;
    for (auto& s : f->slots) {
        ok &= hipMalloc(&s.d_src, f->G * p->g.block_bytes + 64) == hipSuccess;
        ok &= hipMalloc(&s.d_dst, f->G * f->stride) == hipSuccess;
        ok &= hipMalloc(&s.d_sizes, f->G * sizeof(uint64_t)) == hipSuccess;
        ok &= hipHostMalloc((void**)&s.h_sizes, (f->G + 1) * sizeof(uint64_t), hipHostMallocDefault) == hipSuccess;
        ok &= hipEventCreateWithFlags(&s.ev_up, hipEventDisableTiming) == hipSuccess;
        ok &= hipEventCreateWithFlags(&s.ev_comp, hipEventDisableTiming) == hipSuccess;
        ok &= hipEventCreateWithFlags(&s.ev_down, hipEventDisableTiming) == hipSuccess;
        s.dst_host.resize(f->G);
        s.dst_cap.resize(f->G);
    }
    if (!ok) {
        feed_free(p);
        return RSPT_HIP_ERR_ALLOC;
    }
    return RSPT_HIP_OK;
}

static int feed_launch(rspt_hip_packer* p, FeedSlot& s) {
    Feed* f = p->feed;
    HIPCHK(p, hipEventRecord(s.ev_up, p->m_up));
    HIPCHK(p, hipStreamWaitEvent(p->stream, s.ev_up, 0));
    const int rc = compress_batch_serial(p, s.d_src, s.count, s.d_dst, f->stride, s.d_sizes, p->stream);
    if (rc) return rc;
    HIPCHK(p, hipMemcpyAsync(s.h_sizes, s.d_sizes, s.count * sizeof(uint64_t), hipMemcpyDeviceToHost, p->stream));
    HIPCHK(p, hipMemcpyAsync(s.h_sizes + f->G, p->nb_state, sizeof(uint32_t), hipMemcpyDeviceToHost, p->stream));
    HIPCHK(p, hipEventRecord(s.ev_comp, p->stream));
    s.state = FeedSlot::COMPRESSING;
    return RSPT_HIP_OK;
}

int rspt_hip_feed_submit(rspt_hip_packer* p) {
    if (!p || !p->feed) return RSPT_HIP_ERR_ARG;
    HIPCHK(p, hipSetDevice(p->device));
    Feed* f = p->feed;
    FeedSlot& s = f->slots[f->tail];
    if (s.state != FeedSlot::FILLING || s.count == 0) return RSPT_HIP_OK;
    const int rc = feed_launch(p, s);
    if (rc) {  // nothing of this group will arrive: its blocks are reported by rspt_hip_feed_poll with the failure as their status
        s.error = rc;
        s.state = FeedSlot::DONE;
    }
    f->tail = (f->tail + 1) % f->slots.size();
    return rc;
}

int rspt_hip_feed_push(rspt_hip_packer* p, const void* src_host, void* dst_host, size_t dst_cap) {
    if (!p || !p->feed || !src_host || !dst_host) return RSPT_HIP_ERR_ARG;
    HIPCHK(p, hipSetDevice(p->device));
    Feed* f = p->feed;
    FeedSlot& s = f->slots[f->tail];
    if (s.state != FeedSlot::FREE && s.state != FeedSlot::FILLING) return RSPT_HIP_ERR_BUSY;  // the ring is full: poll first
    if (s.state == FeedSlot::FREE) {
        s.state = FeedSlot::FILLING;
        s.count = s.delivered = 0;
        s.error = 0;
        s.first_seq = f->next_seq;
    }
    const size_t i = s.count;
    HIPCHK(p, hipMemcpyAsync(s.d_src + i * p->g.block_bytes, src_host, p->g.block_bytes, hipMemcpyHostToDevice, p->m_up));
    s.dst_host[i] = dst_host;
    s.dst_cap[i] = dst_cap;
    ++s.count;
    ++f->next_seq;
    if (s.count == f->G) return rspt_hip_feed_submit(p);
    return RSPT_HIP_OK;
}

// move every slot as far as it can go without waiting (wait = true: wait for each step instead)
static int feed_advance(rspt_hip_packer* p, bool wait) {
    Feed* f = p->feed;
    const size_t n = f->slots.size();
    for (size_t k = 0; k < n; ++k) {
        FeedSlot& s = f->slots[(f->head + k) % n];
        if (s.state == FeedSlot::COMPRESSING) {
            if (wait) HIPCHK(p, hipEventSynchronize(s.ev_comp));
            const hipError_t q = hipEventQuery(s.ev_comp);
            if (q == hipErrorNotReady) break;  // (the slots behind it are not further along: one compute stream)
            if (q != hipSuccess) {
                p->last_hip_error = (int)q;
                return RSPT_HIP_ERR_LAUNCH;
            }
            const uint32_t nb_now = (uint32_t)s.h_sizes[f->G];
            if (nb_now >= 1 && nb_now <= 4 && nb_now > p->nb_host) p->nb_host = nb_now;  // the next launch writes exactly the planes it needs
            HIPCHK(p, hipStreamWaitEvent(p->m_down, s.ev_comp, 0));
            for (size_t i = 0; i < s.count; ++i) {
                const uint64_t sz = s.h_sizes[i];
                if (!(sz >> 63) && sz <= s.dst_cap[i])
                    HIPCHK(p, hipMemcpyAsync(s.dst_host[i], s.d_dst + i * f->stride, (size_t)sz, hipMemcpyDeviceToHost, p->m_down));
            }
            HIPCHK(p, hipEventRecord(s.ev_down, p->m_down));
            s.state = FeedSlot::DOWNLOADING;
        }
        if (s.state == FeedSlot::DOWNLOADING) {
            if (wait) HIPCHK(p, hipEventSynchronize(s.ev_down));
            const hipError_t q = hipEventQuery(s.ev_down);
            if (q == hipErrorNotReady) continue;  // (a later slot's compress may still be ready for its downloads)
            if (q != hipSuccess) {
                p->last_hip_error = (int)q;
                return RSPT_HIP_ERR_LAUNCH;
            }
            s.state = FeedSlot::DONE;
        }
    }
    return RSPT_HIP_OK;
}

int rspt_hip_feed_poll(rspt_hip_packer* p, size_t* seq, size_t* dst_len, int* status) {
    if (!p || !p->feed || !seq || !dst_len || !status) return RSPT_HIP_ERR_ARG;
    HIPCHK(p, hipSetDevice(p->device));
    Feed* f = p->feed;
    const int rc = feed_advance(p, false);
    if (rc) return rc;
    FeedSlot& s = f->slots[f->head];
    if (s.state != FeedSlot::DONE) return 0;
    const size_t i = s.delivered;
    const uint64_t sz = s.error ? 0 : s.h_sizes[i];
    *seq = s.first_seq + i;
    if (s.error) {
        *dst_len = 0;
        *status = s.error;
    } else if ((sz >> 63) || sz > s.dst_cap[i]) {
        *dst_len = (sz >> 63) ? 0 : (size_t)sz;
        *status = RSPT_HIP_ERR_DST_TOO_SMALL;
    } else {
        *dst_len = (size_t)sz;
        *status = RSPT_HIP_OK;
    }
    if (++s.delivered == s.count) {
        s.state = FeedSlot::FREE;
        f->head = (f->head + 1) % f->slots.size();
    }
    return 1;
}

int rspt_hip_feed_flush(rspt_hip_packer* p) {
    if (!p || !p->feed) return RSPT_HIP_ERR_ARG;
    int rc = rspt_hip_feed_submit(p);
    if (rc) return rc;
    return feed_advance(p, true);
}

int rspt_hip_feed_end(rspt_hip_packer* p) {
    if (!p || !p->feed) return RSPT_HIP_ERR_ARG;
    hipSetDevice(p->device);
    rspt_hip_feed_submit(p);
    hipStreamSynchronize(p->m_up);
    hipStreamSynchronize(p->stream);
    hipStreamSynchronize(p->m_down);  // nothing is copying from or into the caller's buffers any more
    feed_free(p);
    return RSPT_HIP_OK;
}

static int decompress_dev(rspt_hip_packer* p, const void* d_src, size_t src_stride, const uint64_t* pidx, size_t packed_len, size_t nblocks,
                          void* d_dst, uint64_t* d_consumed, void* stream);

int rspt_hip_decompress_batch_dev(rspt_hip_packer* p, const void* d_src, size_t src_stride, size_t nblocks, void* d_dst, uint64_t* d_consumed,
                                  void* stream) {
    return decompress_dev(p, d_src, src_stride, nullptr, 0, nblocks, d_dst, d_consumed, stream);
}

int rspt_hip_decompress_packed_dev(rspt_hip_packer* p, const void* d_packed, size_t packed_len, size_t nblocks, void* d_dst, uint64_t* d_consumed,
                                   void* stream) {
    if (!d_packed || (reinterpret_cast<uintptr_t>(d_packed) & 15)) return RSPT_HIP_ERR_ARG;
    // header + index must be there before the device reads them (compared without a product that could wrap for a huge nblocks)
    if (nblocks == 0 || nblocks > 65535) return RSPT_HIP_ERR_ARG;
    if (packed_len < 32 || nblocks > (packed_len - 32) / 16) return RSPT_HIP_ERR_CORRUPT;
    const uint8_t* base = (const uint8_t*)d_packed;
    // header 32 bytes, index 16 bytes per stream, then the payload the offsets are relative to
    return decompress_dev(p, base + 32 + 16 * nblocks, 0, reinterpret_cast<const uint64_t*>(base + 32), packed_len, nblocks, d_dst, d_consumed,
                          stream);
}

static int decompress_many_pipeline(rspt_hip_packer* p, const void* src_host, size_t src_stride, const size_t* src_len, size_t nblocks, void* dst_host,
                                    size_t* consumed) {
    const size_t C = p->m_chunk, bb = p->g.block_bytes;
    const size_t nchunk = (nblocks + C - 1) / C;
    const uint8_t* src = (const uint8_t*)src_host;
    uint8_t* dst = (uint8_t*)dst_host;
    bool corrupt = false;
    // the slots are used the other way round: streams go up into m_dst, blocks come back out of m_src
    auto finish = [&](size_t k) -> int {
        const int slot = (int)(k & 1);
        const size_t first = k * C, cnt = nblocks - first < C ? nblocks - first : C;
        HIPCHK(p, hipEventSynchronize(p->m_ev_comp[slot]));
        const uint64_t* hs = p->m_hsizes + (size_t)slot * C;
        for (size_t i = 0; i < cnt; ++i) {
            const bool bad = (hs[i] >> 63) != 0;
            consumed[first + i] = bad ? 0 : (size_t)hs[i];
            corrupt |= bad;
        }
        return RSPT_HIP_OK;
    };
    for (size_t k = 0; k < nchunk; ++k) {
        const int slot = (int)(k & 1);
        const size_t first = k * C, cnt = nblocks - first < C ? nblocks - first : C;
        if (k >= 2) {
            HIPCHK(p, hipStreamWaitEvent(p->m_up, p->m_ev_comp[slot], 0));     // chunk k-2 has been decoded out of this slot
            HIPCHK(p, hipStreamWaitEvent(p->stream, p->m_ev_down[slot], 0));  // ... and its blocks have left it
        }
        uint64_t* hidx = nullptr;
        if (src_len) {
            // Only src_len[i] bytes of every stream go up, into a slot that still holds an earlier chunk's bytes behind them: the
            // decoder is therefore bounded by each stream's OWN length -- an index over the slot in the container's form
            // (offset, length; nb 0 = the handle's state), checked on the device like any container -- and not by the slot stride.
            hidx = p->m_hidx + (size_t)slot * (4 + 2 * C);
            if (k >= 2) HIPCHK(p, hipEventSynchronize(p->m_ev_up[slot]));  // (the upload of chunk k-2 has read this staging index)
            hidx[0] = 0x4B43415054505352ull;
            hidx[1] = cnt;
            hidx[2] = (uint64_t)cnt * p->m_stride;
            hidx[3] = 0;
            for (size_t i = 0; i < cnt; ++i) {
                const size_t nbytes = src_len[first + i] < src_stride ? src_len[first + i] : src_stride;
                hidx[4 + 2 * i] = (uint64_t)i * p->m_stride;
                hidx[4 + 2 * i + 1] = nbytes;
                if (nbytes) HIPCHK(p, hipMemcpyAsync(p->m_dst[slot] + i * p->m_stride, src + (first + i) * src_stride, nbytes, hipMemcpyHostToDevice, p->m_up));
            }
            HIPCHK(p, hipMemcpyAsync(p->m_idx[slot], hidx, (4 + 2 * cnt) * sizeof(uint64_t), hipMemcpyHostToDevice, p->m_up));
        } else {
            HIPCHK(p, hipMemcpy2DAsync(p->m_dst[slot], p->m_stride, src + first * src_stride, src_stride, src_stride, cnt, hipMemcpyHostToDevice, p->m_up));
        }
        HIPCHK(p, hipEventRecord(p->m_ev_up[slot], p->m_up));
        HIPCHK(p, hipStreamWaitEvent(p->stream, p->m_ev_up[slot], 0));
        const int rc = hidx ? decompress_dev(p, p->m_dst[slot], 0, p->m_idx[slot] + 4, 32 + 16 * cnt + cnt * p->m_stride, cnt, p->m_src[slot],
                                             p->m_sizes[slot], (void*)p->stream)
                            : rspt_hip_decompress_batch_dev(p, p->m_dst[slot], p->m_stride, cnt, p->m_src[slot], p->m_sizes[slot], (void*)p->stream);
        if (rc) return rc;
        HIPCHK(p, hipMemcpyAsync(p->m_hsizes + (size_t)slot * C, p->m_sizes[slot], cnt * sizeof(uint64_t), hipMemcpyDeviceToHost, p->stream));
        HIPCHK(p, hipEventRecord(p->m_ev_comp[slot], p->stream));
        // the blocks leave as soon as they are decoded: their size is known beforehand
        HIPCHK(p, hipStreamWaitEvent(p->m_down, p->m_ev_comp[slot], 0));
        HIPCHK(p, hipMemcpyAsync(dst + first * bb, p->m_src[slot], cnt * bb, hipMemcpyDeviceToHost, p->m_down));
        HIPCHK(p, hipEventRecord(p->m_ev_down[slot], p->m_down));
        if (k >= 1) {
            const int rf = finish(k - 1);
            if (rf) return rf;
        }
    }
    const int rf = finish(nchunk - 1);
    if (rf) return rf;
    HIPCHK(p, hipStreamSynchronize(p->m_down));
    return corrupt ? RSPT_HIP_ERR_CORRUPT : RSPT_HIP_OK;
}

int rspt_hip_decompress_many(rspt_hip_packer* p, const void* src_host, size_t src_stride, const size_t* src_len, size_t nblocks, void* dst_host,
                             size_t* consumed) {
    if (!p || !src_host || !dst_host || !consumed || nblocks == 0 || src_stride == 0) return RSPT_HIP_ERR_ARG;
    if (p->feed) return RSPT_HIP_ERR_ARG;  // (the feed owns the workspace and the copy streams until rspt_hip_feed_end)
    HIPCHK(p, hipSetDevice(p->device));
    int rc = ensure_many(p);
    if (rc) return rc;
    if (src_stride > p->m_stride) return RSPT_HIP_ERR_ARG;
    rc = decompress_many_pipeline(p, src_host, src_stride, src_len, nblocks, dst_host, consumed);
    if (rc != RSPT_HIP_OK && rc != RSPT_HIP_ERR_CORRUPT) {
        hipStreamSynchronize(p->m_up);
        hipStreamSynchronize(p->stream);
        hipStreamSynchronize(p->m_down);
    }
    return rc;
}

static int decompress_dev(rspt_hip_packer* p, const void* d_src, size_t src_stride, const uint64_t* pidx, size_t packed_len, size_t nblocks,
                          void* d_dst, uint64_t* d_consumed, void* stream) {
    if (!p || !d_src || !d_dst || !d_consumed || nblocks == 0) return RSPT_HIP_ERR_ARG;
    if (p->feed) return RSPT_HIP_ERR_ARG;  // (the feed owns the plane workspace until rspt_hip_feed_end)
    int rc = rspt_hip_reserve(p, nblocks);
    if (rc) return rc;
    HIPCHK(p, hipSetDevice(p->device));
    p->planes_unknown = true;  // the decoded planes land in the compressor's plane workspace
    hipStream_t st = (hipStream_t)stream;
    {
        const Geom& g = p->g;
        const uint32_t B = (uint32_t)nblocks;
        const uint8_t* src = (const uint8_t*)d_src;
        HIPCHK(p, hipMemsetAsync(d_consumed, 0, nblocks * sizeof(uint64_t), st));
        hipLaunchKernelGGL(k_dec_frame, dim3((B * kMaxPlanes + 63) / 64), dim3(64), 0, st, src, (uint64_t)src_stride, B, g, p->nb_state, p->blk_off,
                           d_consumed, p->means, pidx, p->nb_state + 2, (uint64_t)packed_len, p->dec_nb);
        {
            // persistent: block costs differ 10x (dense plane 0 against light planes) and the dispatcher places workgroup i
            // on XCD i % 8 in order, so a plain grid ran its second half at a quarter of the slots (tools/census_decode.py)
            // (k_dec_block takes the blocks plane-fastest: dense and light ones in turns)
            const uint32_t total = g.nblk * B * kMaxPlanes;
            const uint32_t want = 2u * (uint32_t)p->num_cu;  // two 1024-thread workgroups (76 KiB of LDS each) per CU
            hipLaunchKernelGGL(k_dec_block, dim3(want < total ? want : total), dim3(kDecThreads), 0, st, src, (uint64_t)src_stride, g, p->dec_nb, p->blk_off,
                               p->planes, d_consumed, p->ablate ? p->stamps : nullptr, p->verify ? p->crc : nullptr, pidx, p->nb_state + 2, total);
        }
        const bool xd = g.kind == RSPT_HIP_KIND_XDELTA_HZR || g.kind == RSPT_HIP_KIND_DCT;
        const dim3 tg(p->ntile, B);
        // int32 samples of the two hzr packers: the last inverse pass writes the interleaved block itself (k_inv_native)
        const bool direct = (g.kind == RSPT_HIP_KIND_XDELTA_HZR || g.kind == RSPT_HIP_KIND_HZR) && g.bps == 4 && (g.nch & 3) == 0 &&
                            g.ns % kRowTile == 0 && (reinterpret_cast<uintptr_t>(d_dst) & 15) == 0;
        if (direct) {
            const uint32_t nrow = g.N / kRowTile;
            const dim3 rg((g.N / 16 + 255) / 256, B);
            if (xd) {
                hipLaunchKernelGGL(k_inv_rows, rg, dim3(256), 0, st, p->planes, g, p->dec_nb, nrow, p->rowrec);
                hipLaunchKernelGGL(k_inv_scan_rows, dim3(B), dim3(1024), 0, st, p->rowrec, nrow, p->txor, p->tsum);
            }
            if (g.nch <= 16) {
                if (xd)
                    launch_inv_native<true, 16>(p, B, nrow, d_dst, st);
                else
                    launch_inv_native<false, 16>(p, B, nrow, d_dst, st);
            } else {
                if (xd)
                    launch_inv_native<true, 64>(p, B, nrow, d_dst, st);
                else
                    launch_inv_native<false, 64>(p, B, nrow, d_dst, st);
            }
            // (big-endian samples: k_inv_native reverses each sample as it writes it -- g.be)
            HIPCHK(p, hipGetLastError());
            return RSPT_HIP_OK;
        }
        if (xd) {
            hipLaunchKernelGGL((k_inv_tile<0, true>), tg, dim3(256), 0, st, p->planes, g, p->dec_nb, p->ntile, p->txor, p->tsum, p->planar);
            hipLaunchKernelGGL((k_inv_scan_tiles<true>), dim3(B), dim3(1024), 0, st, p->txor, p->ntile);
            hipLaunchKernelGGL((k_inv_tile<1, true>), tg, dim3(256), 0, st, p->planes, g, p->dec_nb, p->ntile, p->txor, p->tsum, p->planar);
            hipLaunchKernelGGL((k_inv_scan_tiles<false>), dim3(B), dim3(1024), 0, st, p->tsum, p->ntile);
            hipLaunchKernelGGL((k_inv_tile<2, true>), tg, dim3(256), 0, st, p->planes, g, p->dec_nb, p->ntile, p->txor, p->tsum, p->planar);
        } else {
            hipLaunchKernelGGL((k_inv_tile<2, false>), tg, dim3(256), 0, st, p->planes, g, p->dec_nb, p->ntile, p->txor, p->tsum, p->planar);
        }
        const int32_t* final_planar = p->planar;
        if (g.kind == RSPT_HIP_KIND_HADAMARD) {
            const uint32_t fw_lds = (g.ns > 32768u ? 32768u : g.ns) * 4u;
            if (g.ns > 65536u) {
                launch_fwht_big<false>(p, B, st);
            } else if (g.ns == 65536u) {
                hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fwht64k<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fw_lds);
                hipLaunchKernelGGL((k_fwht64k<false, false>), dim3(g.nch, B), dim3(1024), fw_lds, st, p->planar, g, p->means, (uint8_t*)nullptr,
                                   (uint32_t*)nullptr, 0u);
            } else {
                hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fwht<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fw_lds);
                hipLaunchKernelGGL((k_fwht<false>), dim3(g.nch, B), dim3(1024), fw_lds, st, p->planar, g, p->means);
            }
        } else if (g.kind == RSPT_HIP_KIND_DCT) {
            if (p->dct_fft)
                launch_dct_fft<false>(p, B, p->planar, p->planar2, st);
            else
                hipLaunchKernelGGL((k_dct<false>), dim3((g.ns + 255) / 256, (g.nch + kDctCh - 1) / kDctCh, B), dim3(256), 0, st, p->planar, g,
                                   p->means, p->cos_tab_t, 0.0, p->idct_scale, p->dct_cs0, p->planar2);
            final_planar = p->planar2;
        }
        const uint32_t T = min(p->Tn_native, g.ns);
        const uint32_t lds = g.nch * (T + 1) * 4;
        const dim3 ng((g.ns + T - 1) / T, B);
        if (g.bps == 4 && (g.nch & 3) == 0 && (g.ns & 3) == 0 && g.nch <= 1024 && (reinterpret_cast<uintptr_t>(d_dst) & 15) == 0) {
            // T4 samples x nch channels in at most 32 KiB of LDS (four workgroups per CU), T4 a multiple of 4
            uint32_t T4 = (uint32_t)((32768ull / (4ull * g.nch) - 1) & ~3ull);
            T4 = T4 > 1024 ? 1024 : T4 < 4 ? 4 : T4;
            if (T4 > g.ns) T4 = g.ns;
            hipLaunchKernelGGL(k_planar_native_i32x4, dim3((g.ns + T4 - 1) / T4, B), dim3(256), g.nch * (T4 + 1) * 4, st, final_planar, g, T4,
                               (uint8_t*)d_dst);
        } else
        switch (g.bps) {
            case 1: hipLaunchKernelGGL((k_planar_native<1>), ng, dim3(256), lds, st, final_planar, g, T, (uint8_t*)d_dst); break;
            case 2: hipLaunchKernelGGL((k_planar_native<2>), ng, dim3(256), lds, st, final_planar, g, T, (uint8_t*)d_dst); break;
            case 3: hipLaunchKernelGGL((k_planar_native<3>), ng, dim3(256), lds, st, final_planar, g, T, (uint8_t*)d_dst); break;
            default: hipLaunchKernelGGL((k_planar_native<4>), ng, dim3(256), lds, st, final_planar, g, T, (uint8_t*)d_dst); break;
        }
        // (big-endian samples: the kernels above reverse each sample as they write it -- g.be)
    }
    HIPCHK(p, hipGetLastError());
    return RSPT_HIP_OK;
}

// src_cap: bytes readable at src_host (SIZE_MAX: the reference's contract -- the stream says how long it is)
static int decompress_host(rspt_hip_packer* p, const void* src_host, size_t src_cap, size_t* src_len, void* dst_host) {
    if (!p || !src_host || !src_len || !dst_host) return RSPT_HIP_ERR_ARG;
    HIPCHK(p, hipSetDevice(p->device));
    int rc = ensure_host_staging(p);
    if (rc) return rc;
    // The stream length is not an input (signal_packer.h:50-57): walk the chunk
    // lengths on the host to find it -- framing only, no decoding.
    const uint8_t* s = (const uint8_t*)src_host;
    const unsigned nb = rspt_hip_current_nb(p);
    size_t pos = 1 + p->g.hdr_len;
    for (unsigned k = 0; k < nb; ++k) {
        if (pos > src_cap || src_cap - pos < 4) return RSPT_HIP_ERR_CORRUPT;  // (never a read past what the caller vouched for)
        uint32_t len;
        memcpy(&len, s + pos, 4);
        pos += 4 + (size_t)len;
        if (pos > p->h_dst_cap || pos > src_cap) return RSPT_HIP_ERR_CORRUPT;
    }
    HIPCHK(p, hipMemcpyAsync(p->h_dst, src_host, pos, hipMemcpyHostToDevice, p->stream));
    // a page-locked destination takes the samples straight from the inverse's last kernel (the download is that kernel's
    // stores, across the link): one synchronisation, no copy phase of its own
    uint8_t* d_out = (uint8_t*)device_view_of_host(dst_host);
    if (d_out && (reinterpret_cast<uintptr_t>(d_out) & 15)) d_out = nullptr;
    rc = rspt_hip_decompress_batch_dev(p, p->h_dst, p->h_dst_cap, 1, d_out ? d_out : p->h_src, p->h_size, (void*)p->stream);
    if (rc) return rc;
    uint64_t used = 0;
    HIPCHK(p, hipMemcpyAsync(&used, p->h_size, sizeof(used), hipMemcpyDeviceToHost, p->stream));
    HIPCHK(p, hipStreamSynchronize(p->stream));
    if (used >> 63) return RSPT_HIP_ERR_CORRUPT;
    if (!d_out) {
        HIPCHK(p, hipMemcpyAsync(dst_host, p->h_src, p->g.block_bytes, hipMemcpyDeviceToHost, p->stream));
        HIPCHK(p, hipStreamSynchronize(p->stream));
    }
    *src_len = (size_t)used;
    return RSPT_HIP_OK;
}

int rspt_hip_decompress(rspt_hip_packer* p, const void* src_host, size_t* src_len, void* dst_host) {
    return decompress_host(p, src_host, (size_t)-1, src_len, dst_host);
}

int rspt_hip_decompress_bounded(rspt_hip_packer* p, const void* src_host, size_t src_cap, size_t* src_len, void* dst_host) {
    return decompress_host(p, src_host, src_cap, src_len, dst_host);
}

int rspt_hip_iir_prefilter_batch_dev(rspt_hip_packer* p, void* d_buf, size_t nblocks, const double* n, const double* d, size_t nr_coefficients,
                                     int init_nr_samples, int per_channel, void* stream) {
    if (!p || !d_buf || !n || !d || nblocks == 0 || nblocks > 0x7FFFFFFFu / (p->g.nch ? p->g.nch : 1)) return RSPT_HIP_ERR_ARG;
    if (nr_coefficients < 2 || nr_coefficients > 5 || init_nr_samples < 0 || init_nr_samples > (1 << 28)) return RSPT_HIP_ERR_ARG;  // filter_opt covers 2..5 (iir_filter.cpp:87-103)
    HIPCHK(p, hipSetDevice(p->device));
    IirCoef c{};
    for (size_t i = 0; i < nr_coefficients; ++i) {
        c.n[i] = n[i];
        c.d[i] = d[i];
    }
    c.nc = (uint32_t)nr_coefficients;
    c.init_steps = 4 * init_nr_samples;
    hipStream_t st = (hipStream_t)stream;
    switch (p->g.bps) {
        case 1: launch_iir_nc<1>(p, (uint8_t*)d_buf, (uint32_t)nblocks, c, per_channel, st); break;
        case 2: launch_iir_nc<2>(p, (uint8_t*)d_buf, (uint32_t)nblocks, c, per_channel, st); break;
        case 3: launch_iir_nc<3>(p, (uint8_t*)d_buf, (uint32_t)nblocks, c, per_channel, st); break;
        default: launch_iir_nc<4>(p, (uint8_t*)d_buf, (uint32_t)nblocks, c, per_channel, st); break;
    }
    HIPCHK(p, hipGetLastError());
    return RSPT_HIP_OK;
}

// ---- multi-GPU gather over RCCL.  RCCL is bound at run time: a process that never gathers (the C++ drop-in on one
// GPU, the tests on the CPU box) does not load it.  A communicator must never cross library instances -- an ncclComm_t made by one
// copy of RCCL is garbage to another (PyTorch wheels bundle their own librccl.so next to /opt/rocm's) -- so the binding goes to the
// copy the process has ALREADY mapped (that is where the caller's ncclComm_t came from); only a process without any gets
// librccl.so.1 from the loader's path; a process with two different copies mapped is refused unless RSPT_RCCL_LIB names the one.
namespace {
struct Rccl {
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    bool ok = false;
};
constexpr int kNcclUint8 = 1, kNcclUint64 = 5;  // ncclDataType_t (rccl.h)
int collect_rccl(struct dl_phdr_info* info, size_t, void* data) {
    auto* v = static_cast<std::vector<std::string>*>(data);
    if (info->dlpi_name && strstr(info->dlpi_name, "librccl.so")) {
        char real[PATH_MAX];
        const std::string path = realpath(info->dlpi_name, real) ? real : info->dlpi_name;
        bool seen = false;
        for (const auto& q : *v) seen = seen || q == path;
        if (!seen) v->push_back(path);
    }
    return 0;
}
}  // namespace
static const Rccl& rccl() {
    static Rccl r = [] {
        Rccl q;
        void* h = nullptr;
        if (const char* want = getenv("RSPT_RCCL_LIB")) {
            h = dlopen(want, RTLD_NOW | RTLD_LOCAL);
        } else {
            std::vector<std::string> mapped;
            dl_iterate_phdr(collect_rccl, &mapped);
            if (mapped.size() > 1) return q;  // two copies in one process: which one made the caller's communicator is not ours to guess
            if (mapped.size() == 1) {
                h = dlopen(mapped[0].c_str(), RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL);  // the instance already in the process
            } else {
                h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
                if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
            }
        }
        if (!h) return q;
        q.AllGather = reinterpret_cast<decltype(q.AllGather)>(dlsym(h, "ncclAllGather"));
        q.Send = reinterpret_cast<decltype(q.Send)>(dlsym(h, "ncclSend"));
        q.Recv = reinterpret_cast<decltype(q.Recv)>(dlsym(h, "ncclRecv"));
        q.GroupStart = reinterpret_cast<decltype(q.GroupStart)>(dlsym(h, "ncclGroupStart"));
        q.GroupEnd = reinterpret_cast<decltype(q.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
        q.ok = q.AllGather && q.Send && q.Recv && q.GroupStart && q.GroupEnd;
        return q;
    }();
    return r;
}

int rspt_hip_gather_sizes(rspt_hip_packer* p, void* comm, int world, const uint64_t* d_total, uint64_t* d_totals, uint64_t* h_totals, void* stream) {
    if (!p || !comm || world < 1 || !d_total || !d_totals) return RSPT_HIP_ERR_ARG;
    const Rccl& R = rccl();
    if (!R.ok) return RSPT_HIP_ERR_UNSUPPORTED;
    HIPCHK(p, hipSetDevice(p->device));
    hipStream_t st = (hipStream_t)stream;
    if (R.AllGather(d_total, d_totals, 1, kNcclUint64, comm, st) != 0) return RSPT_HIP_ERR_LAUNCH;
    if (h_totals) HIPCHK(p, hipMemcpyAsync(h_totals, d_totals, (size_t)world * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    return RSPT_HIP_OK;
}

int rspt_hip_gather_payload(rspt_hip_packer* p, void* comm, int rank, int world, int root, const void* d_packed, const uint64_t* h_totals,
                            void* d_recv, size_t recv_stride, void* stream) {
    if (!p || !comm || world < 1 || rank < 0 || rank >= world || root < 0 || root >= world || !d_packed || !h_totals) return RSPT_HIP_ERR_ARG;
    if (rank == root && !d_recv) return RSPT_HIP_ERR_ARG;
    if (recv_stride & 15) return RSPT_HIP_ERR_ARG;  // (every rank's container must land 16-byte aligned: rspt_hip_decompress_packed_dev)
    const Rccl& R = rccl();
    if (!R.ok) return RSPT_HIP_ERR_UNSUPPORTED;
    HIPCHK(p, hipSetDevice(p->device));
    hipStream_t st = (hipStream_t)stream;
    for (int r = 0; r < world; ++r)
        if (h_totals[r] > recv_stride) return RSPT_HIP_ERR_DST_TOO_SMALL;  // (every rank sees the same sizes and the same stride: nobody posts anything)
    // one group: the root's receives and the peers' sends are matched pairwise, straight over each peer's own link to the root
    if (R.GroupStart() != 0) return RSPT_HIP_ERR_LAUNCH;
    int rc = 0;
    if (rank == root) {
        for (int r = 0; r < world && !rc; ++r)
            if (r != root && h_totals[r]) rc = R.Recv((uint8_t*)d_recv + (size_t)r * recv_stride, (size_t)h_totals[r], kNcclUint8, r, comm, st);
    } else if (h_totals[rank]) {
        rc = R.Send(d_packed, (size_t)h_totals[rank], kNcclUint8, root, comm, st);
    }
    if (R.GroupEnd() != 0 || rc) return RSPT_HIP_ERR_LAUNCH;
    if (rank == root && h_totals[root])
        HIPCHK(p, hipMemcpyAsync((uint8_t*)d_recv + (size_t)root * recv_stride, d_packed, (size_t)h_totals[root], hipMemcpyDeviceToDevice, st));
    return RSPT_HIP_OK;
}

int rspt_hip_gather_containers(rspt_hip_packer* p, void* comm, int rank, int world, int root, const void* d_packed, const uint64_t* d_total,
                               void* d_recv, size_t recv_stride, uint64_t* h_totals, void* stream) {
    if (!p || !h_totals || world < 1) return RSPT_HIP_ERR_ARG;
    HIPCHK(p, hipSetDevice(p->device));
    if (p->gat_world < world) {  // (a few words, kept with the handle)
        hipFree(p->gat_totals);
        p->gat_totals = nullptr;
        p->gat_world = 0;
        if (hipMalloc(&p->gat_totals, (size_t)world * sizeof(uint64_t)) != hipSuccess) return RSPT_HIP_ERR_ALLOC;
        p->gat_world = world;
    }
    uint64_t* d_all = p->gat_totals;
    int rc = rspt_hip_gather_sizes(p, comm, world, d_total, d_all, h_totals, stream);
    if (rc == RSPT_HIP_OK && hipStreamSynchronize((hipStream_t)stream) != hipSuccess) rc = RSPT_HIP_ERR_LAUNCH;  // the sizes are on the host now
    if (rc == RSPT_HIP_OK) rc = rspt_hip_gather_payload(p, comm, rank, world, root, d_packed, h_totals, d_recv, recv_stride, stream);
    return rc;
}

// The same gather without a host synchronisation in the step (what rspt_amd/shard.py LaggedGather does over torch.distributed):
// the sizes of step i travel by a device all-gather and a copy into page-locked memory of the handle, on the handle's own gather
// stream behind an event on `stream`; the host reads them when it posts the payload -- one step later, when they have long
// arrived -- again on the gather stream, so that the payload of step i overlaps the kernels of step i + 1.
static int gather_lag_ensure(rspt_hip_packer* p, int world) {
    if (p->lag_world >= world && p->lag_stream) return RSPT_HIP_OK;
    for (int i = 0; i < 2; ++i) {
        hipFree(p->lag_dtotals[i]);
        if (p->lag_htotals[i]) hipHostFree(p->lag_htotals[i]);
        p->lag_dtotals[i] = p->lag_htotals[i] = nullptr;
    }
    p->lag_world = 0;
    bool ok = true;
    for (int i = 0; i < 2; ++i) {
        ok &= hipMalloc(&p->lag_dtotals[i], (size_t)world * sizeof(uint64_t)) == hipSuccess;
        ok &= hipHostMalloc((void**)&p->lag_htotals[i], (size_t)world * sizeof(uint64_t), hipHostMallocDefault) == hipSuccess;
        if (!p->lag_ev_in[i]) ok &= hipEventCreateWithFlags(&p->lag_ev_in[i], hipEventDisableTiming) == hipSuccess;
        if (!p->lag_ev_sizes[i]) ok &= hipEventCreateWithFlags(&p->lag_ev_sizes[i], hipEventDisableTiming) == hipSuccess;
        if (!p->lag_ev_payload[i]) ok &= hipEventCreateWithFlags(&p->lag_ev_payload[i], hipEventDisableTiming) == hipSuccess;
    }
    if (!p->lag_stream) ok &= hipStreamCreateWithFlags(&p->lag_stream, hipStreamNonBlocking) == hipSuccess;
    if (!ok) return RSPT_HIP_ERR_ALLOC;
    p->lag_world = world;
    return RSPT_HIP_OK;
}

int rspt_hip_gather_post_sizes(rspt_hip_packer* p, void* comm, int world, const uint64_t* d_total, int slot, void* stream) {
    if (!p || !comm || world < 1 || !d_total || slot < 0 || slot > 1) return RSPT_HIP_ERR_ARG;
    HIPCHK(p, hipSetDevice(p->device));
    int rc = gather_lag_ensure(p, world);
    if (rc) return rc;
    HIPCHK(p, hipEventRecord(p->lag_ev_in[slot], (hipStream_t)stream));  // d_total (and the container) are written on `stream`
    HIPCHK(p, hipStreamWaitEvent(p->lag_stream, p->lag_ev_in[slot], 0));
    rc = rspt_hip_gather_sizes(p, comm, world, d_total, p->lag_dtotals[slot], p->lag_htotals[slot], (void*)p->lag_stream);
    if (rc) return rc;
    HIPCHK(p, hipEventRecord(p->lag_ev_sizes[slot], p->lag_stream));
    p->lag_posted[slot] = true;
    return RSPT_HIP_OK;
}

int rspt_hip_gather_post_payload(rspt_hip_packer* p, void* comm, int rank, int world, int root, const void* d_packed, int slot, void* d_recv,
                                 size_t recv_stride, uint64_t* h_totals) {
    if (!p || slot < 0 || slot > 1 || !p->lag_posted[slot] || world > p->lag_world) return RSPT_HIP_ERR_ARG;
    HIPCHK(p, hipSetDevice(p->device));
    HIPCHK(p, hipEventSynchronize(p->lag_ev_sizes[slot]));  // (a step old in the steady state: does not wait)
    p->lag_posted[slot] = false;
    if (h_totals) memcpy(h_totals, p->lag_htotals[slot], (size_t)world * sizeof(uint64_t));
    const int rc = rspt_hip_gather_payload(p, comm, rank, world, root, d_packed, p->lag_htotals[slot], d_recv, recv_stride, (void*)p->lag_stream);
    if (rc) return rc;
    HIPCHK(p, hipEventRecord(p->lag_ev_payload[slot], p->lag_stream));
    return RSPT_HIP_OK;
}

int rspt_hip_gather_wait(rspt_hip_packer* p, int slot, void* stream) {
    if (!p || slot < 0 || slot > 1) return RSPT_HIP_ERR_ARG;
    if (!p->lag_ev_payload[slot]) return RSPT_HIP_OK;  // (nothing was ever posted)
    HIPCHK(p, hipSetDevice(p->device));
    HIPCHK(p, hipStreamWaitEvent((hipStream_t)stream, p->lag_ev_payload[slot], 0));
    return RSPT_HIP_OK;
}

long long rspt_hip_debug_read(rspt_hip_packer* p, int which, void* host_buf, size_t cap) {
    if (!p || !host_buf || p->cap_blocks == 0) return RSPT_HIP_ERR_ARG;
    if (hipSetDevice(p->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return RSPT_HIP_ERR_LAUNCH;
    const Geom& g = p->g;
    const size_t nhb = p->cap_blocks * kMaxPlanes * g.nblk;
    const void* src = nullptr;
    size_t n = 0;
    switch (which) {
        case 0: src = p->planes; n = p->cap_blocks * kMaxPlanes * g.plane_stride; break;
        case 1: src = p->planar; n = p->cap_blocks * (size_t)g.N * 4; break;
        case 2: src = p->planar2; n = p->planar2 ? p->cap_blocks * (size_t)g.N * 4 : 0; break;
        case 3: src = p->hist; n = nhb * kSymStride * 4; break;
        case 4: src = p->meta; n = nhb * sizeof(BlockMeta); break;
        case 5: src = p->nbuse; n = p->cap_blocks * 4; break;
        case 6: src = p->means; n = p->cap_blocks * (size_t)g.hdr_len; break;
        case 8: src = p->nzflag; n = nhb * 4; break;
        case 7: src = p->stamps; n = (512 * 16 * 8 + 2 * 16384) * sizeof(unsigned long long); break;
        default: return RSPT_HIP_ERR_ARG;
    }
    if (!src || n == 0) return 0;
    if (n > cap) n = cap;
    if (hipMemcpy(host_buf, src, n, hipMemcpyDeviceToHost) != hipSuccess) return RSPT_HIP_ERR_LAUNCH;
    return (long long)n;
}

int rspt_hip_set_profiling(rspt_hip_packer* p, int on) {
    if (!p) return RSPT_HIP_ERR_ARG;
    p->profiling = on != 0;
    p->ev_valid = false;
    return RSPT_HIP_OK;
}

int rspt_hip_stage_count(const rspt_hip_packer*) { return ST_COUNT; }
const char* rspt_hip_stage_name(const rspt_hip_packer*, int i) { return (i >= 0 && i < ST_COUNT) ? kStageNames[i] : ""; }

int rspt_hip_stage_times(rspt_hip_packer* p, float* ms, int n) {
    if (!p || !ms || !p->ev_valid) return RSPT_HIP_ERR_ARG;
    HIPCHK(p, hipSetDevice(p->device));
    HIPCHK(p, hipEventSynchronize(p->ev[ST_COUNT]));
    for (int i = 0; i < n && i < ST_COUNT; ++i) {
        float t = 0;
        HIPCHK(p, hipEventElapsedTime(&t, p->ev[i], p->ev[i + 1]));
        ms[i] = t;
    }
    return RSPT_HIP_OK;
}

}  // extern "C"
