// One instance of the 512-wide chunk-stream kernel (wide_ring.h); its own translation unit because it takes minutes to compile.
#include "wide_ring.h"

namespace rb {

int launch_cesr_ring_normal(const float* x, long M, const f4* W, float us, float* Y, unsigned* rw, int grid, hipStream_t s) {
  hipLaunchKernelGGL((k_wide_ring<CesrNet<64, 464>, 0>), dim3(grid), dim3(256), 0, s, x, (const float*)nullptr, M, 1, W, us, 3, Y, rw);
  return check_launch("k_wide_ring<normal_net>");
}

}  // namespace rb
