// Micro-benchmark: cycles per v_mfma_f32_16x16x32_f16 for one wave per SIMD under different orders of the accumulator chains, bare and
// with vector-instruction fillers between the runs.  Build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_order tools/microbench/mfma_order.hip && /tmp/mfma_order
// Patterns (six accumulators = two tiles x three weight classes, a window of four k-blocks, as x6t_engine.h):
//   0  runs of four on ONE accumulator, twelve runs per part (the engine's order)
//   1  two accumulators alternating inside a run of eight (A k0, B k0, A k1, B k1, ...): a chain's links one MFMA apart
//   2  six accumulators round robin (links five MFMAs apart)
//   3  runs of two on one accumulator
// FILL = vector instructions (independent v_fma_f32 on private registers) behind every fourth MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int PATTERN, int FILL>
__global__ __launch_bounds__(256, 1) void k_order(float* out, long long* cycles, int iters) {
  h8 w[4], x[6][4];
  for (int k = 0; k < 4; ++k) {
    for (int e = 0; e < 8; ++e) w[k][e] = (_Float16)(0.001f * (threadIdx.x + k + e));
    for (int a = 0; a < 6; ++a)
      for (int e = 0; e < 8; ++e) x[a][k][e] = (_Float16)(0.002f * (threadIdx.x + a + k + e));
  }
  f4 acc[6];
  for (int a = 0; a < 6; ++a) acc[a] = f4{0.f, 0.f, 0.f, 0.f};
  float v[4] = {1.f, 2.f, 3.f, 4.f};
  const float c = out[0];
  auto fill = [&]() {
#pragma unroll
    for (int i = 0; i < FILL; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i & 3]) : "v"(c));
  };
  // One asm statement per run keeps the program order exactly as written and the hazard recogniser out of the run (the scheduler
  // reorders builtin MFMAs and renames accumulators; between asm statements it inserts s_nop 0).  The only dependencies are
  // SrcC = vDst chains of one opcode, which need no software wait states.
#define M_ "v_mfma_f32_16x16x32_f16 "
  // four k-blocks on one accumulator
#define RUN4(A) asm volatile(M_ "%0, %1, %5, %0\n\t" M_ "%0, %2, %6, %0\n\t" M_ "%0, %3, %7, %0\n\t" M_ "%0, %4, %8, %0" \
                             : "+v"(acc[A]) : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(x[A][0]), "v"(x[A][1]), "v"(x[A][2]), "v"(x[A][3]))
  // two k-blocks on one accumulator
#define RUN2(A, K) asm volatile(M_ "%0, %1, %3, %0\n\t" M_ "%0, %2, %4, %0" : "+v"(acc[A]) : "v"(w[K]), "v"(w[K + 1]), "v"(x[A][K]), "v"(x[A][K + 1]))
  // two accumulators alternating over two k-blocks: A k, B k, A k+1, B k+1
#define ALT2(A, B, K) asm volatile(M_ "%0, %2, %4, %0\n\t" M_ "%1, %2, %6, %1\n\t" M_ "%0, %3, %5, %0\n\t" M_ "%1, %3, %7, %1" \
                                   : "+v"(acc[A]), "+v"(acc[B]) : "v"(w[K]), "v"(w[K + 1]), "v"(x[A][K]), "v"(x[A][K + 1]), "v"(x[B][K]), "v"(x[B][K + 1]))
  // three accumulators, one k-block
#define RR3(A, K) asm volatile(M_ "%0, %3, %4, %0\n\t" M_ "%1, %3, %5, %1\n\t" M_ "%2, %3, %6, %2" \
                               : "+v"(acc[A]), "+v"(acc[A + 1]), "+v"(acc[A + 2]) : "v"(w[K]), "v"(x[A][K]), "v"(x[A + 1][K]), "v"(x[A + 2][K]))
#define ONE(A, K) asm volatile(M_ "%0, %1, %2, %0" : "+v"(acc[A]) : "v"(w[K]), "v"(x[A][K]))
  auto fill4 = [&]() {
#pragma unroll
    for (int i = 0; i < FILL / 4; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i & 3]) : "v"(c));
  };
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {      // 2 x 24 = 48 MFMAs per iteration, FILL fillers behind every fourth
      if constexpr (PATTERN == 0) {
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          RUN4(a);
          fill();
        }
      } else if constexpr (PATTERN == 1) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          ALT2(p, p + 3, 0);
          fill();
          ALT2(p, p + 3, 2);
          fill();
        }
      } else if constexpr (PATTERN == 2) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          RR3(0, k);
          fill();
          RR3(3, k);
          if (k & 1) fill();
        }
      } else if constexpr (PATTERN == 3) {
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          RUN2(a, 0);
          RUN2(a, 2);
          fill();
        }
      } else if constexpr (PATTERN == 4) {      // six accumulators round robin, FILL / 4 fillers behind EVERY MFMA
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int a = 0; a < 6; ++a) {
            ONE(a, k);
            fill4();
          }
      } else if constexpr (PATTERN == 5) {      // runs of four on one accumulator, FILL / 4 fillers behind EVERY MFMA (inside the chain)
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            ONE(a, k);
            fill4();
          }
      } else {                                  // two accumulators alternating (A k, B k, ...), FILL / 4 fillers behind EVERY MFMA
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            ONE(p, k);
            fill4();
            ONE(p + 3, k);
            fill4();
          }
      }
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  const long long t1 = __builtin_readcyclecounter();
  float s = v[0] + v[1] + v[2] + v[3];
  for (int a = 0; a < 6; ++a) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
  out[1 + blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int P, int F>
static void run(const char* name) {
  const int blocks = 256, iters = 2000;
  float* out;
  long long* cyc;
  hipMalloc(&out, sizeof(float) * (1 + blocks * 256));
  hipMalloc(&cyc, sizeof(long long) * blocks);
  hipMemset(out, 0, sizeof(float) * (1 + blocks * 256));
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL((k_order<P, F>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
  hipEventRecord(a);
  hipLaunchKernelGGL((k_order<P, F>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, a, b);
  std::vector<long long> h(blocks);
  hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
  double mean = 0;
  for (auto c : h) mean += (double)c;
  mean /= blocks;
  const double n = 48.0 * iters;
  // readcyclecounter ticks at the constant 100 MHz reference: report time per MFMA from the event time instead, and the tick ratio
  printf("%-44s fill %d: %.2f ns per MFMA (kernel %.3f ms; %.1f ref ticks per 1000 MFMA)\n", name, F, ms * 1e6 / n, ms, mean / n * 1000.0);
  hipFree(out);
  hipFree(cyc);
}

int main() {
  run<0, 0>("0 runs of four on one accumulator");
  run<1, 0>("1 two accumulators alternating");
  run<2, 0>("2 six accumulators round robin");
  run<3, 0>("3 runs of two");
  printf("-- fillers behind every fourth MFMA (clusters of FILL)\n");
  run<0, 4>("0 runs of four");
  run<0, 8>("0 runs of four");
  run<0, 12>("0 runs of four");
  run<2, 4>("2 round robin, clusters");
  run<2, 8>("2 round robin, clusters");
  run<2, 12>("2 round robin, clusters");
  printf("-- FILL / 4 fillers behind EVERY MFMA\n");
  run<4, 4>("4 six accumulators round robin");
  run<4, 8>("4 six accumulators round robin");
  run<4, 12>("4 six accumulators round robin");
  run<6, 4>("6 two accumulators alternating");
  run<6, 8>("6 two accumulators alternating");
  run<6, 12>("6 two accumulators alternating");
  run<5, 4>("5 inside the chain of one accumulator");
  run<5, 8>("5 inside the chain of one accumulator");
  run<5, 12>("5 inside the chain of one accumulator");
  return 0;
}
