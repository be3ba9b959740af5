// Micro-benchmark behind the tile-shape decision of the visibility kernel: one wave per SIMD (512-register budget forced
// by launch bounds), a stream of f16 MFMAs on two alternating accumulators with F independent VALU fillers after each.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_fill mfma_fill.hip && ./mfma_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int SHAPE, int F, int CH>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters, float s) {
  h8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = (_Float16)(threadIdx.x * 0.001f + i);
    b[i] = (_Float16)(threadIdx.x * 0.002f - i);
  }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
  f4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
  f16v d0 = {}, d1 = {};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < (CH == 3 ? 18 : 16); ++u) {
      if constexpr (SHAPE == 16) {
        if (((u & 1) && CH == 2) || (CH == 3 && (u % 6) >= 3)) c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
        else c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
      } else {
        if ((u & 1) && CH == 2) d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d1, 0, 0, 0);
        else d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d0, 0, 0, 0);
      }
#pragma unroll
      for (int f = 0; f < F; ++f) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[f & 7]) : "v"(s));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float r = 0;
  for (int i = 0; i < 8; ++i) r += v[i];
  for (int i = 0; i < 4; ++i) r += c0[i] + c1[i];
  for (int i = 0; i < 16; ++i) r += d0[i] + d1[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int SHAPE, int F, int CH = 2>
void run(float* d) {
  const int iters = 4000, blocks = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<SHAPE, F, CH>), dim3(blocks), dim3(256), 0, 0, d, 10, 1.0001f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<SHAPE, F, CH>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double mf = (double)iters * (CH == 3 ? 18 : 16) * blocks * 4;
  const double flop = mf * (SHAPE == 16 ? 16.0 * 16 * 32 * 2 : 32.0 * 32 * 16 * 2);
  printf("%dx%d  chains %d  fillers/MFMA %d : %.3f ms  %.1f TFLOP/s  (%.2f of 2500)\n", SHAPE, SHAPE, CH, F, ms, flop / ms / 1e9, flop / ms / 1e9 / 2500);
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 256 * 4);
  run<16, 0>(d); run<16, 1>(d); run<16, 2>(d); run<16, 3>(d); run<16, 4>(d);
  run<16, 1, 1>(d); run<16, 2, 1>(d); run<16, 3, 1>(d); run<16, 4, 1>(d); run<16, 1, 3>(d); run<16, 2, 3>(d); run<16, 3, 3>(d);
  run<32, 0, 1>(d); run<32, 2, 1>(d); run<32, 3, 1>(d); run<32, 4, 1>(d); run<32, 5, 1>(d); run<16, 2, 1>(d);
  run<32, 0>(d); run<32, 2>(d); run<32, 3>(d); run<32, 4>(d); run<32, 5>(d); run<32, 6>(d); run<32, 8>(d);
  return 0;
}
