// One instance of the 512-wide chunk-stream kernel (wide_ring.h); its own translation unit because it takes minutes to compile.
#include "wide_ring.h"

namespace rb {

int launch_wide_ring_encoder_rows(const float* X, long M, const f4* W, float us, float* Y, unsigned* rw, int grid, hipStream_t s) {
  hipLaunchKernelGGL((k_wide_ring<WideNet<true>, 3>), dim3(grid), dim3(256), 0, s, X, (const float*)nullptr, M, 1, W, us, 32, Y, rw);
  return check_launch("k_wide_ring<encoder, rows>");
}

}  // namespace rb
