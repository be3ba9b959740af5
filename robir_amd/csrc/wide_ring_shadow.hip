// One instance of the 512-wide chunk-stream kernel (wide_ring.h); its own translation unit because it takes minutes to compile.
#include "wide_ring.h"

namespace rb {

int launch_cesr_ring_shadow(const float* x, long M, int n_label, const f4* W, float us, float* Y, unsigned* rw, int grid, hipStream_t s) {
  hipLaunchKernelGGL((k_wide_ring<CesrNet<192, 336>, 1>), dim3(grid), dim3(256), 0, s, x, (const float*)nullptr, M, n_label, W, us, 2, Y, rw);
  return check_launch("k_wide_ring<shadow_net>");
}

}  // namespace rb
