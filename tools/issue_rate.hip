// issue_rate.hip -- what do the issue pipes of a gfx950 CU sustain for the INTEGER instructions the hzr kernels are made of?
// (MI355X_MICROARCH.md gives v_fma_f32 wave64 = 2 cycles per SIMD; the round-2 floor table assumed one vector and one
// scalar wave-instruction per cycle and CU.  This settles it by measurement.)
//
// Every test: 512 workgroups (2 per CU) of THREADS threads, each wave runs REPS x 32 copies of one instruction pattern on
// independent registers; wave 0 stamps s_memtime around the loop.  Reported: wave-instructions per cycle and CU
// = 32 * REPS * (waves per CU) / cycles.  Build: hipcc --offload-arch=gfx950 -O2 tools/issue_rate.hip -o tools/issue_rate
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define R4(x) x x x x
#define R8(x) R4(x) R4(x)
#define R32(x) R8(x) R8(x) R8(x) R8(x)

constexpr int REPS = 2000;

#define DEF_TEST(NAME, BODY, ...)                                                                              \
    __global__ __launch_bounds__(1024) void NAME(unsigned long long* out, unsigned* sink) {                   \
        __shared__ unsigned lds[4096];                                                                         \
        for (unsigned i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i * 2654435761u;                    \
        __syncthreads();                                                                                       \
        unsigned a0 = threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 11u, a5 = a0 ^ 0x55u, a6 = a0 | 1u, a7 = ~a0; \
        unsigned la = ((threadIdx.x * 4u) & 0x3FFCu), lb = ((threadIdx.x * 2654435761u) >> 18) & 0x3FFCu;     \
        unsigned ls = 64u;  /* all lanes the same address */                                                   \
        unsigned long long p0 = a0, p1 = a1, p2 = a2, p3 = a3;                                                 \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                  \
        for (int r = 0; r < REPS; ++r) {                                                                       \
            asm volatile(BODY : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), \
                           [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3)                         \
                         : [la] "v"(la), [lb] "v"(lb), [ls] "v"(ls)                                            \
                         : __VA_ARGS__);                                                                       \
        }                                                                                                      \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                     \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                  \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;                      \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (unsigned)(p0 + p1 + p2 + p3) == 0x12345u) sink[0] = lds[a0 & 4095];                      \
    }

// ---- vector ALU, independent chains of 8 registers
DEF_TEST(t_v_and, R4("v_and_b32 %0, 0x7f7f7f7f, %0\n v_and_b32 %1, 0x7f7f7f7f, %1\n v_and_b32 %2, 0x7f7f7f7f, %2\n v_and_b32 %3, 0x7f7f7f7f, %3\n"
                     "v_and_b32 %4, 0x7f7f7f7f, %4\n v_and_b32 %5, 0x7f7f7f7f, %5\n v_and_b32 %6, 0x7f7f7f7f, %6\n v_and_b32 %7, 0x7f7f7f7f, %7\n"), "memory")
DEF_TEST(t_v_xor_r, R4("v_xor_b32 %0, %1, %0\n v_xor_b32 %1, %2, %1\n v_xor_b32 %2, %3, %2\n v_xor_b32 %3, %4, %3\n"
                       "v_xor_b32 %4, %5, %4\n v_xor_b32 %5, %6, %5\n v_xor_b32 %6, %7, %6\n v_xor_b32 %7, %0, %7\n"), "memory")
DEF_TEST(t_v_lshl, R4("v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3\n"
                      "v_lshlrev_b32 %4, 1, %4\n v_lshlrev_b32 %5, 1, %5\n v_lshlrev_b32 %6, 1, %6\n v_lshlrev_b32 %7, 1, %7\n"), "memory")
DEF_TEST(t_v_bfe, R4("v_bfe_u32 %0, %0, 3, 9\n v_bfe_u32 %1, %1, 3, 9\n v_bfe_u32 %2, %2, 3, 9\n v_bfe_u32 %3, %3, 3, 9\n"
                     "v_bfe_u32 %4, %4, 3, 9\n v_bfe_u32 %5, %5, 3, 9\n v_bfe_u32 %6, %6, 3, 9\n v_bfe_u32 %7, %7, 3, 9\n"), "memory")
DEF_TEST(t_v_perm, R4("v_perm_b32 %0, %0, %1, %2\n v_perm_b32 %1, %1, %2, %3\n v_perm_b32 %2, %2, %3, %4\n v_perm_b32 %3, %3, %4, %5\n"
                      "v_perm_b32 %4, %4, %5, %6\n v_perm_b32 %5, %5, %6, %7\n v_perm_b32 %6, %6, %7, %0\n v_perm_b32 %7, %7, %0, %1\n"), "memory")
DEF_TEST(t_v_alignbit, R4("v_alignbit_b32 %0, %0, %1, 5\n v_alignbit_b32 %1, %1, %2, 5\n v_alignbit_b32 %2, %2, %3, 5\n v_alignbit_b32 %3, %3, %4, 5\n"
                          "v_alignbit_b32 %4, %4, %5, 5\n v_alignbit_b32 %5, %5, %6, 5\n v_alignbit_b32 %6, %6, %7, 5\n v_alignbit_b32 %7, %7, %0, 5\n"), "memory")
DEF_TEST(t_v_lshl_or, R4("v_lshl_or_b32 %0, %0, 3, %1\n v_lshl_or_b32 %1, %1, 3, %2\n v_lshl_or_b32 %2, %2, 3, %3\n v_lshl_or_b32 %3, %3, 3, %4\n"
                         "v_lshl_or_b32 %4, %4, 3, %5\n v_lshl_or_b32 %5, %5, 3, %6\n v_lshl_or_b32 %6, %6, 3, %7\n v_lshl_or_b32 %7, %7, 3, %0\n"), "memory")
DEF_TEST(t_v_add3, R4("v_add3_u32 %0, %0, %1, %2\n v_add3_u32 %1, %1, %2, %3\n v_add3_u32 %2, %2, %3, %4\n v_add3_u32 %3, %3, %4, %5\n"
                      "v_add3_u32 %4, %4, %5, %6\n v_add3_u32 %5, %5, %6, %7\n v_add3_u32 %6, %6, %7, %0\n v_add3_u32 %7, %7, %0, %1\n"), "memory")
DEF_TEST(t_v_mul_lo, R4("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %1, %1, %2\n v_mul_lo_u32 %2, %2, %3\n v_mul_lo_u32 %3, %3, %4\n"
                        "v_mul_lo_u32 %4, %4, %5\n v_mul_lo_u32 %5, %5, %6\n v_mul_lo_u32 %6, %6, %7\n v_mul_lo_u32 %7, %7, %0\n"), "memory")
DEF_TEST(t_v_mad_u24, R4("v_mad_u32_u24 %0, %0, %1, %2\n v_mad_u32_u24 %1, %1, %2, %3\n v_mad_u32_u24 %2, %2, %3, %4\n v_mad_u32_u24 %3, %3, %4, %5\n"
                         "v_mad_u32_u24 %4, %4, %5, %6\n v_mad_u32_u24 %5, %5, %6, %7\n v_mad_u32_u24 %6, %6, %7, %0\n v_mad_u32_u24 %7, %7, %0, %1\n"), "memory")
DEF_TEST(t_v_lshl64, R4("v_lshlrev_b64 %[p0], 3, %[p0]\n v_lshlrev_b64 %[p1], 3, %[p1]\n v_lshlrev_b64 %[p2], 3, %[p2]\n v_lshlrev_b64 %[p3], 3, %[p3]\n"
                        "v_lshlrev_b64 %[p0], 5, %[p0]\n v_lshlrev_b64 %[p1], 5, %[p1]\n v_lshlrev_b64 %[p2], 5, %[p2]\n v_lshlrev_b64 %[p3], 5, %[p3]\n"), "memory")
DEF_TEST(t_v_bcnt, R4("v_bcnt_u32_b32 %0, %0, %1\n v_bcnt_u32_b32 %1, %1, %2\n v_bcnt_u32_b32 %2, %2, %3\n v_bcnt_u32_b32 %3, %3, %4\n"
                      "v_bcnt_u32_b32 %4, %4, %5\n v_bcnt_u32_b32 %5, %5, %6\n v_bcnt_u32_b32 %6, %6, %7\n v_bcnt_u32_b32 %7, %7, %0\n"), "memory")
DEF_TEST(t_v_ffbl, R4("v_ffbl_b32 %0, %0\n v_ffbl_b32 %1, %1\n v_ffbl_b32 %2, %2\n v_ffbl_b32 %3, %3\n"
                      "v_ffbl_b32 %4, %4\n v_ffbl_b32 %5, %5\n v_ffbl_b32 %6, %6\n v_ffbl_b32 %7, %7\n"), "memory")
DEF_TEST(t_v_sdwa, R4("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
                      "v_lshlrev_b32_sdwa %1, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n"
                      "v_lshlrev_b32_sdwa %2, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n"
                      "v_lshlrev_b32_sdwa %3, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n"
                      "v_lshlrev_b32_sdwa %4, %5, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
                      "v_lshlrev_b32_sdwa %5, %6, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n"
                      "v_lshlrev_b32_sdwa %6, %7, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n"
                      "v_lshlrev_b32_sdwa %7, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n"), "memory")
DEF_TEST(t_v_dpp, R4("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                     "v_mov_b32_dpp %2, %3 row_shr:2 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 row_shr:3 row_mask:0xf bank_mask:0xf\n"
                     "v_mov_b32_dpp %4, %5 row_shr:4 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 row_shr:8 row_mask:0xf bank_mask:0xf\n"
                     "v_mov_b32_dpp %6, %7 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_mov_b32_dpp %7, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n"), "memory")
DEF_TEST(t_v_min_dpp, R4("v_min_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_min_u32_dpp %1, %2, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "v_min_u32_dpp %2, %3, %2 row_shr:2 row_mask:0xf bank_mask:0xf\n v_min_u32_dpp %3, %4, %3 row_shr:3 row_mask:0xf bank_mask:0xf\n"
                         "v_min_u32_dpp %4, %5, %4 row_shr:4 row_mask:0xf bank_mask:0xf\n v_min_u32_dpp %5, %6, %5 row_shr:8 row_mask:0xf bank_mask:0xf\n"
                         "v_min_u32_dpp %6, %7, %6 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_min_u32_dpp %7, %0, %7 row_bcast:31 row_mask:0xc bank_mask:0xf\n"), "memory")
// ---- compares into scalar registers / select from them
DEF_TEST(t_v_cmp_vcc, R32("v_cmp_eq_u32 vcc, %0, %1\n"), "vcc", "memory")
DEF_TEST(t_v_cmp_sgpr, R8("v_cmp_eq_u32 s[40:41], %0, %1\n v_cmp_eq_u32 s[42:43], %1, %2\n v_cmp_eq_u32 s[44:45], %2, %3\n v_cmp_eq_u32 s[46:47], %3, %4\n"),
         "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "memory")
DEF_TEST(t_v_cmp_sdwa, R8("v_cmp_eq_u32_sdwa s[40:41], %0, %1 src0_sel:BYTE_0 src1_sel:DWORD\n v_cmp_eq_u32_sdwa s[42:43], %1, %2 src0_sel:BYTE_1 src1_sel:DWORD\n"
                          "v_cmp_eq_u32_sdwa s[44:45], %2, %3 src0_sel:BYTE_2 src1_sel:DWORD\n v_cmp_eq_u32_sdwa s[46:47], %3, %4 src0_sel:BYTE_3 src1_sel:DWORD\n"),
         "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "memory")
DEF_TEST(t_v_cndmask, R4("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                         "v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc\n"), "memory")
DEF_TEST(t_v_readlane, R8("v_readlane_b32 s40, %0, 5\n v_readlane_b32 s41, %1, 7\n v_readlane_b32 s42, %2, 9\n v_readlane_b32 s43, %3, 63\n"),
         "s40", "s41", "s42", "s43", "memory")
// ---- scalar ALU
DEF_TEST(t_s_and64, R4("s_and_b64 s[40:41], s[40:41], s[42:43]\n s_and_b64 s[42:43], s[42:43], s[44:45]\n s_and_b64 s[44:45], s[44:45], s[46:47]\n"
                       "s_and_b64 s[46:47], s[46:47], s[48:49]\n s_andn2_b64 s[48:49], s[48:49], s[50:51]\n s_or_b64 s[50:51], s[50:51], s[52:53]\n"
                       "s_lshl_b64 s[52:53], s[52:53], 1\n s_xor_b64 s[54:55], s[54:55], s[40:41]\n"),
         "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "scc", "memory")
DEF_TEST(t_s_add32, R4("s_add_u32 s40, s40, s41\n s_add_u32 s41, s41, s42\n s_add_u32 s42, s42, s43\n s_add_u32 s43, s43, s44\n"
                       "s_and_b32 s44, s44, s45\n s_lshl_b32 s45, s45, 1\n s_bfe_u32 s46, s46, 0x90003\n s_mul_i32 s47, s47, s40\n"),
         "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "scc", "memory")
// ---- mixes: does the scalar pipe issue beside the vector one?
DEF_TEST(t_mix_1v1s, R4("v_and_b32 %0, 0x7f7f7f7f, %0\n s_and_b64 s[40:41], s[40:41], s[42:43]\n v_and_b32 %1, 0x7f7f7f7f, %1\n s_and_b64 s[42:43], s[42:43], s[44:45]\n"
                        "v_and_b32 %2, 0x7f7f7f7f, %2\n s_and_b64 s[44:45], s[44:45], s[46:47]\n v_and_b32 %3, 0x7f7f7f7f, %3\n s_and_b64 s[46:47], s[46:47], s[40:41]\n"),
         "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "scc", "memory")
DEF_TEST(t_mix_3v1s, R4("v_and_b32 %0, 0x7f7f7f7f, %0\n v_and_b32 %4, 0x7f7f7f7f, %4\n v_and_b32 %1, 0x7f7f7f7f, %1\n s_and_b64 s[42:43], s[42:43], s[44:45]\n"
                        "v_and_b32 %2, 0x7f7f7f7f, %2\n v_and_b32 %5, 0x7f7f7f7f, %5\n v_and_b32 %3, 0x7f7f7f7f, %3\n s_and_b64 s[46:47], s[46:47], s[40:41]\n"),
         "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "scc", "memory")
// ---- LDS
DEF_TEST(t_ds_add_lin, R32("ds_add_u32 %[la], %0\n"), "memory")        // conflict-free: lane -> consecutive words
DEF_TEST(t_ds_add_rand, R32("ds_add_u32 %[lb], %0\n"), "memory")       // per-lane pseudo-random word
DEF_TEST(t_ds_add_same, R32("ds_add_u32 %[ls], %0\n"), "memory")      // all 64 lanes one word
DEF_TEST(t_ds_or_lin, R32("ds_or_b32 %[la], %0\n"), "memory")
DEF_TEST(t_ds_read_lin, R4("ds_read_b32 %0, %[la]\n ds_read_b32 %1, %[la] offset:256\n ds_read_b32 %2, %[la] offset:512\n ds_read_b32 %3, %[la] offset:768\n"
                           "ds_read_b32 %4, %[la] offset:1024\n ds_read_b32 %5, %[la] offset:1280\n ds_read_b32 %6, %[la] offset:1536\n ds_read_b32 %7, %[la] offset:1792\n"), "memory")
DEF_TEST(t_ds_read_rand, R4("ds_read_b32 %0, %[lb]\n ds_read_b32 %1, %[lb] offset:4\n ds_read_b32 %2, %[lb] offset:8\n ds_read_b32 %3, %[lb] offset:12\n"
                            "ds_read_b32 %4, %[lb] offset:16\n ds_read_b32 %5, %[lb] offset:20\n ds_read_b32 %6, %[lb] offset:24\n ds_read_b32 %7, %[lb] offset:28\n"), "memory")
DEF_TEST(t_ds_read64_rand, R8("ds_read_b64 %[p0], %[lb]\n ds_read_b64 %[p1], %[lb] offset:8\n ds_read_b64 %[p2], %[lb] offset:16\n ds_read_b64 %[p3], %[lb] offset:24\n"), "memory")

// ---- round 2 of questions: why is v_cndmask slow?  scalar operands, carries, which VOP2 ops run at the double rate
DEF_TEST(t_v_cndmask_sg, R4("v_cndmask_b32 %0, %0, %1, s[40:41]\n v_cndmask_b32 %1, %1, %2, s[40:41]\n v_cndmask_b32 %2, %2, %3, s[42:43]\n v_cndmask_b32 %3, %3, %4, s[42:43]\n"
                            "v_cndmask_b32 %4, %4, %5, s[40:41]\n v_cndmask_b32 %5, %5, %6, s[40:41]\n v_cndmask_b32 %6, %6, %7, s[42:43]\n v_cndmask_b32 %7, %7, %0, s[42:43]\n"), "memory")
DEF_TEST(t_v_cndmask_c, R4("v_cndmask_b32 %0, 5, %1, vcc\n v_cndmask_b32 %1, 7, %2, vcc\n v_cndmask_b32 %2, 9, %3, vcc\n v_cndmask_b32 %3, 11, %4, vcc\n"
                           "v_cndmask_b32 %4, 13, %5, vcc\n v_cndmask_b32 %5, 15, %6, vcc\n v_cndmask_b32 %6, 17, %7, vcc\n v_cndmask_b32 %7, 19, %0, vcc\n"), "memory")
DEF_TEST(t_v_cmp_cnd, R4("v_cmp_eq_u32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_eq_u32 vcc, %4, %5\n v_cndmask_b32 %6, %6, %7, vcc\n"
                         "v_cmp_eq_u32 vcc, %1, %2\n v_cndmask_b32 %3, %3, %4, vcc\n v_cmp_eq_u32 vcc, %5, %6\n v_cndmask_b32 %7, %7, %0, vcc\n"), "vcc", "memory")
DEF_TEST(t_v_and_sgpr, R4("v_and_b32 %0, s40, %0\n v_and_b32 %1, s41, %1\n v_and_b32 %2, s42, %2\n v_and_b32 %3, s43, %3\n"
                          "v_and_b32 %4, s40, %4\n v_and_b32 %5, s41, %5\n v_and_b32 %6, s42, %6\n v_and_b32 %7, s43, %7\n"), "memory")
DEF_TEST(t_v_lshl_sgpr, R4("v_lshlrev_b32 %0, s40, %0\n v_lshlrev_b32 %1, s41, %1\n v_lshlrev_b32 %2, s42, %2\n v_lshlrev_b32 %3, s43, %3\n"
                           "v_lshlrev_b32 %4, s40, %4\n v_lshlrev_b32 %5, s41, %5\n v_lshlrev_b32 %6, s42, %6\n v_lshlrev_b32 %7, s43, %7\n"), "memory")
DEF_TEST(t_v_add_co, R4("v_add_co_u32 %0, vcc, %0, %1\n v_add_co_u32 %1, vcc, %1, %2\n v_add_co_u32 %2, vcc, %2, %3\n v_add_co_u32 %3, vcc, %3, %4\n"
                        "v_add_co_u32 %4, vcc, %4, %5\n v_add_co_u32 %5, vcc, %5, %6\n v_add_co_u32 %6, vcc, %6, %7\n v_add_co_u32 %7, vcc, %7, %0\n"), "vcc", "memory")
DEF_TEST(t_v_addc, R4("v_addc_co_u32 %0, vcc, %0, %1, vcc\n v_addc_co_u32 %1, vcc, %1, %2, vcc\n v_addc_co_u32 %2, vcc, %2, %3, vcc\n v_addc_co_u32 %3, vcc, %3, %4, vcc\n"
                      "v_addc_co_u32 %4, vcc, %4, %5, vcc\n v_addc_co_u32 %5, vcc, %5, %6, vcc\n v_addc_co_u32 %6, vcc, %6, %7, vcc\n v_addc_co_u32 %7, vcc, %7, %0, vcc\n"), "vcc", "memory")
DEF_TEST(t_v_add, R4("v_add_u32 %0, %1, %0\n v_add_u32 %1, %2, %1\n v_add_u32 %2, %3, %2\n v_add_u32 %3, %4, %3\n"
                     "v_add_u32 %4, %5, %4\n v_add_u32 %5, %6, %5\n v_add_u32 %6, %7, %6\n v_add_u32 %7, %0, %7\n"), "memory")
DEF_TEST(t_v_or, R4("v_or_b32 %0, %1, %0\n v_or_b32 %1, %2, %1\n v_or_b32 %2, %3, %2\n v_or_b32 %3, %4, %3\n"
                    "v_or_b32 %4, %5, %4\n v_or_b32 %5, %6, %5\n v_or_b32 %6, %7, %6\n v_or_b32 %7, %0, %7\n"), "memory")
DEF_TEST(t_v_min, R4("v_min_u32 %0, %1, %0\n v_min_u32 %1, %2, %1\n v_min_u32 %2, %3, %2\n v_min_u32 %3, %4, %3\n"
                     "v_min_u32 %4, %5, %4\n v_min_u32 %5, %6, %5\n v_min_u32 %6, %7, %6\n v_min_u32 %7, %0, %7\n"), "memory")
DEF_TEST(t_v_mov, R4("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n"
                     "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n"), "memory")
DEF_TEST(t_v_and_or, R4("v_and_or_b32 %0, %0, %1, %2\n v_and_or_b32 %1, %1, %2, %3\n v_and_or_b32 %2, %2, %3, %4\n v_and_or_b32 %3, %3, %4, %5\n"
                        "v_and_or_b32 %4, %4, %5, %6\n v_and_or_b32 %5, %5, %6, %7\n v_and_or_b32 %6, %6, %7, %0\n v_and_or_b32 %7, %7, %0, %1\n"), "memory")
DEF_TEST(t_v_xor_dep, R32("v_xor_b32 %0, %1, %0\n"), "memory")   // one dependent chain per wave
DEF_TEST(t_v_perm_dep, R32("v_perm_b32 %0, %0, %1, %2\n"), "memory")
DEF_TEST(t_v_mbcnt, R4("v_mbcnt_lo_u32_b32 %0, %1, %0\n v_mbcnt_hi_u32_b32 %1, %2, %1\n v_mbcnt_lo_u32_b32 %2, %3, %2\n v_mbcnt_hi_u32_b32 %3, %4, %3\n"
                       "v_mbcnt_lo_u32_b32 %4, %5, %4\n v_mbcnt_hi_u32_b32 %5, %6, %5\n v_mbcnt_lo_u32_b32 %6, %7, %6\n v_mbcnt_hi_u32_b32 %7, %0, %7\n"), "memory")
DEF_TEST(t_v_rfl, R8("v_readfirstlane_b32 s40, %0\n v_readfirstlane_b32 s41, %1\n v_readfirstlane_b32 s42, %2\n v_readfirstlane_b32 s43, %3\n"),
         "s40", "s41", "s42", "s43", "memory")
DEF_TEST(t_s_bit64, R4("s_bcnt1_i32_b64 s40, s[42:43]\n s_ff1_i32_b64 s41, s[44:45]\n s_flbit_i32_b64 s46, s[42:43]\n s_lshr_b64 s[44:45], s[44:45], 1\n"
                       "s_not_b64 s[48:49], s[48:49]\n s_cselect_b64 s[50:51], s[48:49], s[44:45]\n s_cmp_lg_u64 s[48:49], 0\n s_bfm_b64 s[52:53], s40, s41\n"),
         "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s48", "s49", "s50", "s51", "s52", "s53", "scc", "memory")
DEF_TEST(t_s_nop, R32("s_nop 0\n"), "memory")
// LDS atomics with hot bins: lanes are spread over 64 / M distinct words (M lanes per word)
#define DEF_HOT(NAME, M)                                                                                       \
    __global__ __launch_bounds__(1024) void NAME(unsigned long long* out, unsigned* sink) {                   \
        __shared__ unsigned lds[4096];                                                                         \
        for (unsigned i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 0;                                  \
        __syncthreads();                                                                                       \
        unsigned one = 1u, addr = (((threadIdx.x & 63u) / M) * 37u + (threadIdx.x >> 6) * 264u) * 4u;          \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                  \
        for (int r = 0; r < REPS; ++r) asm volatile(R32("ds_add_u32 %0, %1\n") : : "v"(addr), "v"(one) : "memory"); \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                     \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                  \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;                      \
        if (lds[threadIdx.x] == 0x12345u) sink[0] = 1;                                                         \
    }
DEF_HOT(t_ds_add_hot1, 1)
DEF_HOT(t_ds_add_hot2, 2)
DEF_HOT(t_ds_add_hot4, 4)
DEF_HOT(t_ds_add_hot8, 8)
DEF_HOT(t_ds_add_hot16, 16)

// ---- fp64: the IIR pre-filter's recurrence is a chain of dependent v_mul_f64 / v_add_f64 (filter.hip): latency and issue rate
DEF_TEST(t_f64_add_dep, R32("v_add_f64 %[p0], %[p0], %[p1]\n"), "memory")
DEF_TEST(t_f64_mul_dep, R32("v_mul_f64 %[p0], %[p0], %[p1]\n"), "memory")
DEF_TEST(t_f64_muladd_dep, R8("v_mul_f64 %[p2], %[p0], %[p1]\n v_add_f64 %[p0], %[p3], -%[p2]\n v_add_f64 %[p0], %[p0], -%[p1]\n v_add_f64 %[p0], %[p0], -%[p3]\n"), "memory")
DEF_TEST(t_f64_add, R8("v_add_f64 %[p0], %[p0], %[p1]\n v_add_f64 %[p1], %[p1], %[p2]\n v_add_f64 %[p2], %[p2], %[p3]\n v_add_f64 %[p3], %[p3], %[p0]\n"), "memory")
DEF_TEST(t_f64_mul, R8("v_mul_f64 %[p0], %[p0], %[p1]\n v_mul_f64 %[p1], %[p1], %[p2]\n v_mul_f64 %[p2], %[p2], %[p3]\n v_mul_f64 %[p3], %[p3], %[p0]\n"), "memory")
DEF_TEST(t_f32_fma, R4("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %4\n v_fma_f32 %3, %3, %4, %5\n"
                       "v_fma_f32 %4, %4, %5, %6\n v_fma_f32 %5, %5, %6, %7\n v_fma_f32 %6, %6, %7, %0\n v_fma_f32 %7, %7, %0, %1\n"), "memory")

typedef void (*kern_t)(unsigned long long*, unsigned*);
static void run(const char* name, kern_t k, int threads, int per_rep) {
    const int nwg = 512;
    unsigned long long* out;
    unsigned* sink;
    hipMalloc(&out, nwg * 16 * 8);
    hipMalloc(&sink, 64);
    hipMemset(out, 0, nwg * 16 * 8);
    hipLaunchKernelGGL(k, dim3(nwg), dim3(threads), 0, 0, out, sink);
    hipDeviceSynchronize();
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(nwg), dim3(threads), 0, 0, out, sink);
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h(nwg * 16);
    hipMemcpy(h.data(), out, nwg * 16 * 8, hipMemcpyDeviceToHost);
    std::vector<unsigned long long> v;
    for (auto x : h)
        if (x) v.push_back(x);
    std::sort(v.begin(), v.end());
    const double cyc = (double)v[v.size() / 2];
    const int waves_cu = 2 * threads / 64;
    const double ipc = (double)per_rep * REPS * waves_cu / cyc;
    // wall-clock view (s_memtime ticks turned out to run at about half the shader clock on this part): instructions per ns and CU,
    // and the time one wave needs per instruction (the latency of a dependent chain when the wave is alone on its SIMD)
    const double per_cu_ns = (double)per_rep * REPS * waves_cu / (ms * 1e6), ns_per_wave_instr = ms * 1e6 / ((double)per_rep * REPS);
    printf("%-18s threads %4d  waves/CU %2d  ticks %9.0f -> %.3f instr/tick/CU | wall %.3f ms -> %.3f instr/ns/CU, %.2f ns per instr of one wave\n", name,
           threads, waves_cu, cyc, ipc, ms, per_cu_ns, ns_per_wave_instr);
    hipFree(out);
    hipFree(sink);
}

#define RUN(k) run(#k, k, 1024, 32); run(#k, k, 256, 32);
int main() {
    RUN(t_v_and) RUN(t_v_xor_r) RUN(t_v_lshl) RUN(t_v_bfe) RUN(t_v_perm) RUN(t_v_alignbit) RUN(t_v_lshl_or) RUN(t_v_add3) RUN(t_v_mul_lo)
    RUN(t_v_mad_u24) RUN(t_v_lshl64) RUN(t_v_bcnt) RUN(t_v_ffbl) RUN(t_v_sdwa) RUN(t_v_dpp) RUN(t_v_min_dpp)
    RUN(t_v_cmp_vcc) RUN(t_v_cmp_sgpr) RUN(t_v_cmp_sdwa) RUN(t_v_cndmask) RUN(t_v_readlane)
    RUN(t_s_and64) RUN(t_s_add32) RUN(t_mix_1v1s) RUN(t_mix_3v1s)
    RUN(t_f32_fma) RUN(t_f64_add) RUN(t_f64_mul) RUN(t_f64_add_dep) RUN(t_f64_mul_dep) RUN(t_f64_muladd_dep)
    run("t_f64_add_dep", t_f64_add_dep, 64, 32); run("t_f64_muladd_dep", t_f64_muladd_dep, 64, 32); run("t_v_xor_dep", t_v_xor_dep, 64, 32); run("t_v_perm_dep", t_v_perm_dep, 64, 32);
    RUN(t_v_cndmask_sg) RUN(t_v_cndmask_c) RUN(t_v_cmp_cnd) RUN(t_v_and_sgpr) RUN(t_v_lshl_sgpr) RUN(t_v_add_co) RUN(t_v_addc) RUN(t_v_add) RUN(t_v_or) RUN(t_v_min)
    RUN(t_v_mov) RUN(t_v_and_or) RUN(t_v_xor_dep) RUN(t_v_perm_dep) RUN(t_v_mbcnt) RUN(t_v_rfl) RUN(t_s_bit64) RUN(t_s_nop)
    RUN(t_ds_add_hot1) RUN(t_ds_add_hot2) RUN(t_ds_add_hot4) RUN(t_ds_add_hot8) RUN(t_ds_add_hot16)
    RUN(t_ds_add_lin) RUN(t_ds_add_rand) RUN(t_ds_add_same) RUN(t_ds_or_lin) RUN(t_ds_read_lin) RUN(t_ds_read_rand) RUN(t_ds_read64_rand)
    return 0;
}
