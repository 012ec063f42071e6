/*
 * ref_shim.cpp -- C-linkage handle around the REAL reference, for oracle/_ref.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is ours; it contains no reference code.
 * It is compiled (by oracle/Makefile, only when /root/reference is present)
 * together with the reference's own sources where they lie, into
 * oracle/_ref/librspt_ref.so.  It exposes i_signal_packer
 * (lib_rspt/signal_packer.h:29-73) and the vendored hzr C API
 * (lib_rspt/lib_hzr/libhzr.h:46-88) through plain C entry points so that
 * Python (ctypes) can pin the restatement in rspt_oracle.c against them and
 * generate the golden fixtures under tests/golden/.
 */
#include <cstddef>
#include <cstdint>
#include <iostream>
#include <vector>

#include "signal_packer.h" /* -I$(REF)/lib_rspt ; the header has no includes of its own */

extern "C" {
#include "lib_hzr/libhzr.h"
}
using namespace std;       /* filter.h names vector<double> unqualified */
#include "filter.h"        /* i_filter (lib_rspt/filter.h:24-89) */
#include "lib_signalpacker/utils.h" /* convert_native_to_i32 / convert_i32_to_native */

namespace {
struct ref_handle {
    int kind;
    i_signal_packer* p;
};
}  // namespace

extern "C" {

void* ref_packer_new(int kind, size_t bps, size_t nch, size_t ns, size_t nb) {
    i_signal_packer* p = nullptr;
    switch (kind) {
        case 0: p = i_signal_packer::new_hzr(bps, nch, ns); break;
        case 1: p = i_signal_packer::new_xdelta_hzr(bps, nch, ns, nb); break;
        case 2: p = i_signal_packer::new_dct(bps, nch, ns); break;
        case 3: p = i_signal_packer::new_hadamard(bps, nch, ns); break;
        default: return nullptr;
    }
    return new ref_handle{kind, p};
}

void ref_packer_free(void* h) {
    ref_handle* r = static_cast<ref_handle*>(h);
    if (!r) return;
    switch (r->kind) {
        case 0: i_signal_packer::delete_hzr(r->p); break;
        case 1: i_signal_packer::delete_xdelta_hzr(r->p); break;
        case 2: i_signal_packer::delete_dct(r->p); break;
        case 3: i_signal_packer::delete_hadamard(r->p); break;
    }
    delete r;
}

int ref_packer_compress(void* h, const uint8_t* src, uint8_t* dst, size_t dst_max_len, size_t* dst_len) {
    size_t len = 0;
    static_cast<ref_handle*>(h)->p->compress(src, dst, dst_max_len, len);
    *dst_len = len;
    return 0;
}

int ref_packer_decompress(void* h, const uint8_t* src, size_t* src_len, uint8_t* dst) {
    size_t len = 0;
    int rc = static_cast<ref_handle*>(h)->p->decompress(src, len, dst);
    *src_len = len;
    return rc;
}

size_t ref_hzr_max_compressed_size(size_t n) { return hzr_max_compressed_size(n); }

int ref_hzr_encode(const uint8_t* in, size_t n, uint8_t* out, size_t out_cap, size_t* out_len) {
    return hzr_encode(in, n, out, out_cap, out_len) == HZR_OK;
}

int ref_hzr_decode(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_cap) {
    return hzr_decode(in, in_len, out, out_cap) == HZR_OK;
}

int ref_hzr_verify(const uint8_t* in, size_t in_len, size_t* decoded) {
    return hzr_verify(in, in_len, decoded) == HZR_OK;
}


/* The reference's pre-filter step, driven exactly as its test harness does (lib_rspt_test/rspt_test.cpp:116-136):
 * one i_filter for all channels, init_history_values(first sample, init_nr_samples), filter_opt per sample. */
int ref_iir_prefilter_native(uint8_t* native, size_t bps, size_t nch, size_t ns, const double* n, const double* d, size_t nc, int init_nr_samples) {
    std::vector<int32_t> flat(nch * ns);
    std::vector<int32_t*> rows(nch);
    for (size_t c = 0; c < nch; ++c) rows[c] = flat.data() + c * ns;
    convert_native_to_i32(rows.data(), native, (int)ns, (int)nch, (int)bps, false);
    i_filter* f = i_filter::new_iir(n, d, nc);
    for (size_t j = 0; j < nch; ++j) {
        f->init_history_values(rows[j][0], init_nr_samples);
        for (size_t i = 0; i < ns; ++i) rows[j][i] = f->filter_opt(rows[j][i]);
    }
    i_filter::delete_iir(f);
    convert_i32_to_native(native, rows.data(), (int)ns, (int)nch, (int)bps, false);
    return 0;
}

/* convert_native_to_i32 with the byte order the caller picks (utils.cpp:123-191): fixtures for big-endian ingest */
void ref_native_to_i32(int32_t* planar, const uint8_t* native, size_t ns, size_t nch, size_t bps, int reverse_byte_order) {
    std::vector<int32_t*> rows(nch);
    for (size_t c = 0; c < nch; ++c) rows[c] = planar + c * ns;
    convert_native_to_i32(rows.data(), native, (int)ns, (int)nch, (int)bps, reverse_byte_order != 0);
}
void ref_i32_to_native(uint8_t* native, const int32_t* planar, size_t ns, size_t nch, size_t bps, int reverse_byte_order) {
    std::vector<int32_t*> rows(nch);
    for (size_t c = 0; c < nch; ++c) rows[c] = const_cast<int32_t*>(planar) + c * ns;
    convert_i32_to_native(native, rows.data(), (int)ns, (int)nch, (int)bps, reverse_byte_order != 0);
}

}  /* extern "C" */
