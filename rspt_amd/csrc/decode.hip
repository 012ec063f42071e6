// decode.hip -- decompress path (filled in below in this round).
#include "common.hpp"

namespace rspt {

inline void launch_decode(const Geom& g, const uint8_t* d_src, size_t src_stride, size_t nblocks, uint8_t* planes, int32_t* planar,
                          uint32_t* nb_state, uint8_t* d_dst, uint64_t* d_consumed, uint8_t* means, double* dscratch, hipStream_t st) {}

}  // namespace rspt
