// signal_packer_hip.cpp -- the i_signal_packer factories of include/signal_packer.h
// (reference: lib_rspt/signal_packer.h:59-72), each backed by one rspt_hip_packer
// handle.  This is the host-language side of the drop-in boundary: plain C++,
// no HIP types; everything below the virtuals goes through the C ABI.
//
// Behaviour kept from the reference packers:
//   * compress()/decompress() never throw; decompress() returns 0
//     (signal_packer_xdelta_hzr.cpp:74-85 and siblings);
//   * an nb escalation prints "Compression needs one more byte to encode."
//     once per added byte (signal_packer_xdelta_hzr.cpp:63-69);
//   * objects are destroyed only through the matching static delete_*.
// Difference: a failure of the GPU path is reported on std::cerr and leaves
// dst_len = 0 -- there is no CPU fallback to hide it.
#include <cstddef>
#include <cstdlib>
#include <iostream>

#include "../../include/rspt_hip.h"
#include "../../include/signal_packer.h"

namespace {

// Which GPU the factories place a packer on (the reference's constructors have no such argument; SURVEY.md section 5:
// "env var only for device selection"): the calling thread's rspt_cxx_set_device() value if it set one, else the
// RSPT_HIP_DEVICE environment variable, else device 0.  One packer per host thread, one thread per GPU: a shard per device.
thread_local int t_device = -1;
int factory_device() {
    if (t_device >= 0) return t_device;
    if (const char* e = std::getenv("RSPT_HIP_DEVICE")) return std::atoi(e);
    return 0;
}

class signal_packer_hip : public i_signal_packer {
public:
    signal_packer_hip(int kind, size_t bps, size_t nch, size_t ns, size_t nb) {
        int rc = rspt_hip_packer_create(&h_, kind, bps, nch, ns, nb, factory_device());
        if (rc != RSPT_HIP_OK) {
            std::cerr << "ERROR: rspt_hip_packer_create: " << rspt_hip_status_string(rc) << std::endl;
            h_ = nullptr;
        } else {
            nb_seen_ = rspt_hip_current_nb(h_);
        }
    }
    ~signal_packer_hip() { rspt_hip_packer_destroy(h_); }

    void compress(const unsigned char* src, unsigned char* dst, size_t dst_max_len, size_t& dst_len) override {
        dst_len = 0;
        if (!h_) return;
        size_t len = 0;
        int rc = rspt_hip_compress(h_, src, dst, dst_max_len, &len);
        if (rc != RSPT_HIP_OK) {
            std::cerr << "ERROR: rspt_hip_compress: " << rspt_hip_status_string(rc) << std::endl;
            return;
        }
        dst_len = len;
        const unsigned nb = rspt_hip_current_nb(h_);
        for (; nb_seen_ < nb; ++nb_seen_) std::cout << "Compression needs one more byte to encode." << std::endl;
    }

    int decompress(const unsigned char* src, size_t& src_len, unsigned char* dst) override {
        src_len = 0;
        if (!h_) return 0;
        size_t used = 0;
        int rc = rspt_hip_decompress(h_, src, &used, dst);
        if (rc != RSPT_HIP_OK)
            std::cerr << "ERROR: rspt_hip_decompress: " << rspt_hip_status_string(rc) << std::endl;
        else
            src_len = used;
        return 0;
    }

    rspt_hip_packer* handle() const { return h_; }

private:
    rspt_hip_packer* h_ = nullptr;
    unsigned nb_seen_ = 0;
};

}  // namespace

i_signal_packer* i_signal_packer::new_xdelta_hzr(size_t bytes_per_channel, size_t nr_of_channels, size_t nr_of_samples_in_each_channel, size_t nr_bytes_to_encode) {
    return new signal_packer_hip(RSPT_HIP_KIND_XDELTA_HZR, bytes_per_channel, nr_of_channels, nr_of_samples_in_each_channel, nr_bytes_to_encode);
}
void i_signal_packer::delete_xdelta_hzr(i_signal_packer* instance) { delete static_cast<signal_packer_hip*>(instance); }

i_signal_packer* i_signal_packer::new_hzr(size_t bytes_per_channel, size_t nr_of_channels, size_t nr_of_samples_in_each_channel) {
    return new signal_packer_hip(RSPT_HIP_KIND_HZR, bytes_per_channel, nr_of_channels, nr_of_samples_in_each_channel, 4);
}
void i_signal_packer::delete_hzr(i_signal_packer* instance) { delete static_cast<signal_packer_hip*>(instance); }

i_signal_packer* i_signal_packer::new_dct(size_t bytes_per_channel, size_t nr_of_channels, size_t nr_of_samples_in_each_channel) {
    return new signal_packer_hip(RSPT_HIP_KIND_DCT, bytes_per_channel, nr_of_channels, nr_of_samples_in_each_channel, 2);
}
void i_signal_packer::delete_dct(i_signal_packer* instance) { delete static_cast<signal_packer_hip*>(instance); }

i_signal_packer* i_signal_packer::new_hadamard(size_t bytes_per_channel, size_t nr_of_channels, size_t nr_of_samples_in_each_channel) {
    return new signal_packer_hip(RSPT_HIP_KIND_HADAMARD, bytes_per_channel, nr_of_channels, nr_of_samples_in_each_channel, 3);
}
void i_signal_packer::delete_hadamard(i_signal_packer* instance) { delete static_cast<signal_packer_hip*>(instance); }

// C shim so that non-C++ hosts (the Python tests) can drive the C++ factories
// themselves, not just the C ABI underneath them.
extern "C" {
// device for the packers this THREAD creates from now on (-1: back to RSPT_HIP_DEVICE / 0); returns the previous setting
int rspt_cxx_set_device(int device) {
    const int prev = t_device;
    t_device = device;
    return prev;
}
void* rspt_cxx_new(int kind, size_t bps, size_t nch, size_t ns, size_t nb) {
    switch (kind) {
        case RSPT_HIP_KIND_HZR: return i_signal_packer::new_hzr(bps, nch, ns);
        case RSPT_HIP_KIND_XDELTA_HZR: return i_signal_packer::new_xdelta_hzr(bps, nch, ns, nb);
        case RSPT_HIP_KIND_DCT: return i_signal_packer::new_dct(bps, nch, ns);
        case RSPT_HIP_KIND_HADAMARD: return i_signal_packer::new_hadamard(bps, nch, ns);
    }
    return nullptr;
}
void rspt_cxx_delete(int kind, void* p) {
    i_signal_packer* q = static_cast<i_signal_packer*>(p);
    switch (kind) {
        case RSPT_HIP_KIND_HZR: i_signal_packer::delete_hzr(q); break;
        case RSPT_HIP_KIND_XDELTA_HZR: i_signal_packer::delete_xdelta_hzr(q); break;
        case RSPT_HIP_KIND_DCT: i_signal_packer::delete_dct(q); break;
        case RSPT_HIP_KIND_HADAMARD: i_signal_packer::delete_hadamard(q); break;
    }
}
void rspt_cxx_compress(void* p, const unsigned char* src, unsigned char* dst, size_t dst_max_len, size_t* dst_len) {
    size_t len = 0;
    static_cast<i_signal_packer*>(p)->compress(src, dst, dst_max_len, len);
    *dst_len = len;
}
int rspt_cxx_decompress(void* p, const unsigned char* src, size_t* src_len, unsigned char* dst) {
    size_t len = 0;
    int rc = static_cast<i_signal_packer*>(p)->decompress(src, len, dst);
    *src_len = len;
    return rc;
}
}
