/*
 * signal_packer.h -- the i_signal_packer C++ surface of rspt, served by the
 * MI355X HIP path.
 *
 * Mirrors lib_rspt/signal_packer.h:29-73 of the reference: the same class
 * name, the same two virtuals in the same order (vtable slot 0 = compress,
 * slot 1 = decompress, no virtual destructor), and the same static
 * new_ and delete_ factory pairs with the same argument meaning, so code written
 * against the reference header (README.md:63-75, rspt_test.cpp:71,78,247,253)
 * compiles and links against librspt_hip.so unchanged.  Differences, all
 * additive: this header has an include guard and pulls in <cstddef>; the
 * objects behind the factories own a GPU workspace (rspt_hip.h) instead of
 * host tensors.  new_lala/delete_lala are declared by the reference but
 * defined nowhere (signal_packer.h:71-72); they are declared here for source
 * compatibility and likewise left undefined.
 */
#ifndef RSPT_AMD_SIGNAL_PACKER_H_
#define RSPT_AMD_SIGNAL_PACKER_H_

#include <cstddef>

class i_signal_packer
{
public:
    /* Compress one block of bytes_per_channel * nr_of_channels * nr_of_samples
     * bytes (interleaved, sample-major, little-endian) from `src` into `dst`.
     * `dst_len` receives the stream length.  (signal_packer.h:44) */
    virtual void compress(const unsigned char* src, unsigned char* dst, size_t dst_max_len, size_t& dst_len) = 0;

    /* Decompress one stream.  `src_len` is an OUTPUT: the number of stream
     * bytes consumed.  Returns 0.  (signal_packer.h:57) */
    virtual int decompress(const unsigned char* src, size_t& src_len, unsigned char* dst) = 0;

    static i_signal_packer* new_xdelta_hzr(size_t bytes_per_channel, size_t nr_of_channels, size_t nr_of_samples_in_each_channel, size_t nr_bytes_to_encode);
    static void delete_xdelta_hzr(i_signal_packer* instance);

    static i_signal_packer* new_hzr(size_t bytes_per_channel, size_t nr_of_channels, size_t nr_of_samples_in_each_channel);
    static void delete_hzr(i_signal_packer* instance);

    static i_signal_packer* new_dct(size_t bytes_per_channel, size_t nr_of_channels, size_t nr_of_samples_in_each_channel);
    static void delete_dct(i_signal_packer* instance);

    static i_signal_packer* new_hadamard(size_t bytes_per_channel, size_t nr_of_channels, size_t nr_of_samples_in_each_channel);
    static void delete_hadamard(i_signal_packer* instance);

    static i_signal_packer* new_lala(size_t bytes_per_channel, size_t nr_of_channels, size_t nr_of_samples_in_each_channel);
    static void delete_lala(i_signal_packer* instance);
};

/* GPU placement of the packers the calling thread creates from now on (additive; the reference's factories have no such
 * argument): -1 = the RSPT_HIP_DEVICE environment variable, else device 0.  Returns the previous setting.  One packer per
 * host thread and one thread per GPU is how a C++ caller shards independent blocks over the devices of a node. */
extern "C" int rspt_cxx_set_device(int device);

#endif /* RSPT_AMD_SIGNAL_PACKER_H_ */
