// transforms.hip -- per-channel transforms of the lossy packers (placeholder bodies
// are filled in below in this round).
#include "common.hpp"

namespace rspt {

__global__ __launch_bounds__(1024) void k_fwht(int32_t* planar, Geom g, uint8_t* means) {}
__global__ __launch_bounds__(1024) void k_dct(int32_t* planar, Geom g, uint8_t* means, double* scratch) {}

}  // namespace rspt
