// Issue cost of vector instructions for ONE wave per SIMD (the occupancy of the 512-register ring kernels): cycles per
// instruction of dependent and independent fp32 chains, of transcendentals, and of vector work placed behind MFMAs.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_issue.hip -o tools/ubench/valu_issue.bin && tools/ubench/valu_issue.bin
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// TWO waves per SIMD (512 threads; waves w and w + 4 share a SIMD): waves 0..3 issue 16 MFMAs per iteration, waves 4..7 64
// independent multiplies -- does one wave's vector work issue while the other's MFMAs run?
__global__ __launch_bounds__(512, 1) void k_pair(float* out, long long* cyc, int iters, int what) {
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = 1.0f + 0.001f * (threadIdx.x + i);
  f4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  h8 w, x;
#pragma unroll
  for (int i = 0; i < 8; ++i) { w[i] = (_Float16)(0.01f * i); x[i] = (_Float16)(0.02f * i); }
  const float c = 1.0001f;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool do_mfma = wave < 4 ? (what & 1) : (what & 4), do_valu = wave < 4 ? (what & 2) : (what & 8);
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (do_mfma) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if ((i / 3) & 1) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc1) : "v"(w), "v"(x));
        else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc0) : "v"(w), "v"(x));
      }
    }
    if (do_valu) {
#pragma unroll
      for (int i = 0; i < 64; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i & 15]) : "v"(c));
    }
  }
  long long t1 = __builtin_readcyclecounter();
  __syncthreads();
  long long t2 = __builtin_readcyclecounter();
  float s = acc0[0] + acc1[0];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i];
  out[(blockIdx.x * 512 + threadIdx.x) & 65535] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t0; }
}

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* cyc, int iters) {
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = 1.0f + 0.001f * (threadIdx.x + i);
  f4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  h8 w, x;
#pragma unroll
  for (int i = 0; i < 8; ++i) { w[i] = (_Float16)(0.01f * i); x[i] = (_Float16)(0.02f * i); }
  const float c = 1.0001f;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0) {            // 64 dependent multiplies (one chain)
#pragma unroll
      for (int i = 0; i < 64; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[0]) : "v"(c));
    } else if constexpr (MODE == 1) {     // 64 multiplies, 16 independent chains
#pragma unroll
      for (int i = 0; i < 64; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i & 15]) : "v"(c));
    } else if constexpr (MODE == 2) {     // 64 multiplies, 2 chains
#pragma unroll
      for (int i = 0; i < 64; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i & 1]) : "v"(c));
    } else if constexpr (MODE == 3) {     // 64 dependent exp2
#pragma unroll
      for (int i = 0; i < 64; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[0]));
    } else if constexpr (MODE == 4) {     // 64 exp2, 16 chains
#pragma unroll
      for (int i = 0; i < 64; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i & 15]));
    } else if constexpr (MODE == 5) {     // 16 MFMAs back to back, two accumulators alternating
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (i & 1) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc1) : "v"(w), "v"(x));
        else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc0) : "v"(w), "v"(x));
      }
    } else if constexpr (MODE >= 6 && MODE <= 12) {   // 16 MFMAs (chains of 3 like the ring kernels), F = MODE - 6 independent multiplies behind each
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if ((i / 3) & 1) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc1) : "v"(w), "v"(x));
        else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc0) : "v"(w), "v"(x));
#pragma unroll
        for (int f = 0; f < MODE - 6; ++f) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[(i * 7 + f) & 15]) : "v"(c));
      }
    } else if constexpr (MODE == 13) {    // 16 MFMAs in a block, then 64 independent multiplies in a block (same work as F = 4)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if ((i / 3) & 1) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc1) : "v"(w), "v"(x));
        else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc0) : "v"(w), "v"(x));
      }
#pragma unroll
      for (int i = 0; i < 64; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i & 15]) : "v"(c));
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = acc0[0] + acc1[0];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
void run(const char* label, int n_inst, float* out, long long* cyc) {
  const int iters = 20000;
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, out, cyc, 100);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
  long long h = 0;
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-86s %7.2f cycles per iteration, %5.2f per instruction\n", label, (double)h / iters, (double)h / iters / n_inst);
}

void run_pair(const char* label, int what, float* out, long long* cyc) {
  const int iters = 20000;
  hipLaunchKernelGGL(k_pair, dim3(256), dim3(512), 0, 0, out, cyc, 100, what);
  hipLaunchKernelGGL(k_pair, dim3(256), dim3(512), 0, 0, out, cyc, iters, what);
  long long h[2] = {0, 0};
  hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
  printf("%-86s %7.2f cycles per iteration (wave 0), %7.2f until all eight waves are done\n", label, (double)h[0] / iters, (double)h[1] / iters);
}

int main() {
  float* out;
  long long* cyc;
  hipMalloc(&out, 256 * 256 * 4);
  hipMalloc(&cyc, 16);
  run<0>("64 v_mul_f32, one dependent chain", 64, out, cyc);
  run<2>("64 v_mul_f32, two chains", 64, out, cyc);
  run<1>("64 v_mul_f32, sixteen chains", 64, out, cyc);
  run<3>("64 v_exp_f32, one dependent chain", 64, out, cyc);
  run<4>("64 v_exp_f32, sixteen chains", 64, out, cyc);
  run<5>("16 v_mfma_f32_16x16x32_f16, two accumulators alternating", 16, out, cyc);
  run<6>("16 MFMA in chains of three, nothing else", 16, out, cyc);
  run<7>("16 MFMA in chains of three + 1 independent v_mul behind each", 16, out, cyc);
  run<8>("16 MFMA in chains of three + 2 behind each", 16, out, cyc);
  run<9>("16 MFMA in chains of three + 3 behind each", 16, out, cyc);
  run<10>("16 MFMA in chains of three + 4 behind each", 16, out, cyc);
  run<11>("16 MFMA in chains of three + 5 behind each", 16, out, cyc);
  run<12>("16 MFMA in chains of three + 6 behind each", 16, out, cyc);
  run<13>("16 MFMA in a block, then 64 independent v_mul in a block", 16, out, cyc);
  run_pair("two waves per SIMD: 16 MFMA in waves 0-3, waves 4-7 idle", 1, out, cyc);
  run_pair("two waves per SIMD: waves 0-3 idle, 64 v_mul in waves 4-7", 8, out, cyc);
  run_pair("two waves per SIMD: 16 MFMA in waves 0-3 WHILE 64 v_mul in waves 4-7", 1 | 8, out, cyc);
  run_pair("two waves per SIMD: 16 MFMA + 64 v_mul in every wave", 1 | 2 | 4 | 8, out, cyc);
  run_pair("two waves per SIMD: 16 MFMA in every wave", 1 | 4, out, cyc);
  return 0;
}
