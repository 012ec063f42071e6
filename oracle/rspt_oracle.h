/*
 * rspt_oracle.h -- CPU restatement of the rspt signal_packer hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call it, and only as the checker.  The product path lives in
 * rspt_amd/csrc (HIP) and never includes this header.
 *
 * Every function cites the reference file:line (paths under
 * /root/reference/lib_rspt/) whose behaviour it restates.  The restatement is
 * pinned against the compiled reference (oracle/_ref, built by the Makefile
 * from the reference's own sources) and against tests/golden/ fixtures that
 * were generated from that compiled reference -- see tests/test_oracle_*.py.
 */
#ifndef RSPT_ORACLE_H_
#define RSPT_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Factory selector (NOT the stream's method byte, which is 0,0,1,2). */
enum {
    ORC_KIND_HZR = 0,        /* signal_packer_hzr.cpp:34-68        */
    ORC_KIND_XDELTA_HZR = 1, /* signal_packer_xdelta_hzr.cpp:34-88 */
    ORC_KIND_DCT = 2,        /* signal_packer_dct.cpp:36-156       */
    ORC_KIND_HADAMARD = 3    /* signal_packer_hadamard.cpp:35-107  */
};

/* ---- lib_hzr ------------------------------------------------------------ */

/* CRC-32C, init ~0, reflected poly 0x82F63B78, final ~ (hzr_crc32c.c:77-97). */
uint32_t orc_crc32c(const void* data, size_t len);

/* hzr_encode.c:489-497 */
size_t orc_hzr_max_compressed_size(size_t n);

/* hzr_encode.c:499-544.  Returns 1 on success, 0 on failure (out_cap must be
 * >= orc_hzr_max_compressed_size(n)). */
int orc_hzr_encode(const uint8_t* in, size_t n, uint8_t* out, size_t out_cap, size_t* out_len);

/* hzr_decode.c:626-674.  *consumed = bytes of `in` used.  Returns 1 / 0. */
int orc_hzr_decode(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_cap, size_t* consumed);

/* hzr_decode.c:569-624: walk the blocks and check every CRC.  Returns 1 / 0. */
int orc_hzr_verify(const uint8_t* in, size_t in_len, size_t* decoded_size);

/* Per-block facts used by the GPU parity tests: for one block of <= 65536
 * bytes report the 261-bin token histogram (hzr_encode.c:133-173), the mode
 * chosen (0 copy / 1 huffman / 2 fill) and the payload size in bytes. */
void orc_hzr_block_stats(const uint8_t* in, size_t n, uint32_t hist[261], int* mode, size_t* payload_len);

/* ---- lib_signalpacker/utils.cpp ------------------------------------------ */

/* utils.cpp:123-191 (LE branches) / :51-121.  planar is [nch][ns] int32. */
void orc_native_to_i32(int32_t* planar, const uint8_t* native, size_t ns, size_t nch, size_t bps);
void orc_i32_to_native(uint8_t* native, const int32_t* planar, size_t ns, size_t nch, size_t bps);

/* utils.cpp:193-202, :215-219, :221-230 fused over the flat array, in place. */
void orc_xdelta_forward(int32_t* a, size_t n);
/* utils.cpp:232-236, :215-219, :204-213 fused, in place. */
void orc_xdelta_inverse(int32_t* a, size_t n);

/* utils.cpp:30-40: int64 sum divided by a size_t (unsigned 64-bit division). */
int32_t orc_average_32(const int32_t* a, size_t len);

/* Smallest nb in [nb_min,4] for which the reference's round-trip self check
 * (signal_packer_xdelta_hzr.cpp:59-69) passes on transformed values v[0..n):
 * derived criterion of SURVEY.md section 8 note a-3. */
unsigned orc_xdelta_needed_nb(const int32_t* v, size_t n, size_t bps, unsigned nb_min);

/* lib_fwht/fwht.c:4-28 (natural-order WHT, int32 wrap) */
void orc_fwht(int32_t* a, size_t n);

/* ---- packers ------------------------------------------------------------- */

typedef struct orc_packer orc_packer;

/* nb is used by ORC_KIND_XDELTA_HZR only (ctor arg nr_bytes_to_encode). */
orc_packer* orc_packer_new(int kind, size_t bps, size_t nch, size_t ns, size_t nb);
void orc_packer_free(orc_packer* p);

/* i_signal_packer::compress (signal_packer.h:44).  Returns 0 on success. */
int orc_packer_compress(orc_packer* p, const uint8_t* src, uint8_t* dst, size_t dst_max_len, size_t* dst_len);
/* i_signal_packer::decompress (signal_packer.h:57).  *src_len is an output. */
int orc_packer_decompress(orc_packer* p, const uint8_t* src, size_t* src_len, uint8_t* dst);

/* dct / hadamard framing around transform coefficients the caller evaluated itself
 * (signal_packer_dct.cpp:117-127,130-139; signal_packer_hadamard.cpp:73-80,83-92).  Used for dct at
 * ns > 8192, where the reference's n*n float table cannot be built and the transform is
 * restated in fp64 (oracle.py: dct_big_*).  coeffs = [nch][ns], means = [nch]. */
int orc_packer_compress_coeffs(orc_packer* p, const int32_t* coeffs, const int32_t* means, uint8_t* dst, size_t dst_max_len,
                               size_t* dst_len);
int orc_packer_decompress_coeffs(orc_packer* p, const uint8_t* src, size_t* src_len, int32_t* coeffs, int32_t* means);

/* current nr_bytes_to_compress_ (mutates on escalation, xdelta_hzr.cpp:66). */
unsigned orc_packer_nb(const orc_packer* p);
/* 0: escalate by round-trip + memcmp like the reference (default);
 * 1: escalate by the derived criterion (no decode).  Same streams. */
void orc_packer_set_fast_verify(orc_packer* p, int on);

/* Worst-case stream size for this packer at its current nb. */
size_t orc_packer_max_compressed_size(const orc_packer* p);

/* Intermediate views for kernel-level parity tests: after a compress() call,
 * the transformed planar int32 matrix [nch*ns] the planes were cut from. */
const int32_t* orc_packer_last_enc(const orc_packer* p);

/* PRDN[%] exactly as lib_rspt_test/rspt_test.cpp:98-111. */
double orc_prdn(const uint8_t* orig_native, const uint8_t* dec_native, size_t ns, size_t nch, size_t bps);

/* 32-bit FNV-1a, the hash SURVEY.md section 6 quotes for golden streams. */
uint32_t orc_fnv1a(const void* data, size_t len);

/* IIR pre-filter of an interleaved native block, in place: the reference's pipeline step in front of the packers
 * (lib_rspt_test/rspt_test.cpp:116-136 with lib_rspt/lib_filter/iir_filter.cpp:46-116).  nc = 2..5 coefficients,
 * n = feedback (n[0] unused), d = feed-forward.  shared_state != 0: one filter object for all channels, as there. */
int orc_iir_prefilter_native(uint8_t* native, size_t bps, size_t nch, size_t ns, const double* n, const double* d, size_t nc, int init_nr_samples,
                             int shared_state);

#ifdef __cplusplus
}
#endif
#endif /* RSPT_ORACLE_H_ */
