// Energy per product of the two f16 MFMA shapes under the visibility kernel's filler mix:
// does v_mfma_f32_32x32x16_f16 (half the operand-register reads per MAC of 16x16x32) let the package hold a higher
// clock -- or draw less -- at the same FLOP rate?  One wave per SIMD (launch bounds 256,1), 256 workgroups, ~3 s per
// variant; package power and shader clock are sampled with rocm-smi while the kernel runs.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_power.bin mfma_power.hip && ./mfma_power.bin
// Filler mix per 49152 MACs (= 6 MFMAs 16x16x32 = 3 MFMAs 32x32x16): 8 independent VALU ops + 2 ds_read_b128 -- the
// measured 1.3 VALU + 0.37 LDS per 16x16x32 MFMA of k_dvis_v2 (profiles/r02_dvis_pmc.md).  `bare` = no fillers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int SHAPE, bool FILL, int NOPER>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters, float s) {
  __shared__ u4 lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = u4{(unsigned)i, 1u, 2u, 3u};
  __syncthreads();
  // NOPER distinct operand tuples (the real kernel reads a different register tuple for every MFMA)
  h8 a[NOPER], b[NOPER];
  for (int j = 0; j < NOPER; ++j)
    for (int i = 0; i < 8; ++i) {
      a[j][i] = (_Float16)(threadIdx.x * 0.001f + i + j);
      b[j][i] = (_Float16)(threadIdx.x * 0.002f - i - j);
    }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
  f4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
  f16v d0 = {};
  u4 r0 = {}, r1 = {};
  const u4* lp = lds + (threadIdx.x & 63);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {        // 4 x 49152 MACs per wave
      if constexpr (SHAPE == 16) {
#pragma unroll
        for (int u = 0; u < 6; ++u) {
          if (u < 3) c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(blk * 6 + u) % NOPER], b[(blk * 6 + u + 1) % NOPER], c0, 0, 0, 0);
          else c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(blk * 6 + u) % NOPER], b[(blk * 6 + u + 1) % NOPER], c1, 0, 0, 0);
          if constexpr (FILL) {
            asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[u & 7]) : "v"(s));
            if (u == 1 || u == 4) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[6 + (u & 1)]) : "v"(s));
            if (u == 0) r0 = lp[(it & 7) * 64];
            if (u == 3) r1 = lp[(it & 7) * 64 + 512];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(blk * 3 + u) % NOPER], b[(blk * 3 + u + 1) % NOPER], d0, 0, 0, 0);
          if constexpr (FILL) {
            asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[2 * u]) : "v"(s));
            asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[2 * u + 1]) : "v"(s));
            if (u < 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[6 + u]) : "v"(s));
            if (u == 0) r0 = lp[(it & 7) * 64];
            if (u == 1) r1 = lp[(it & 7) * 64 + 512];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  float r = 0;
  for (int i = 0; i < 8; ++i) r += v[i];
  for (int i = 0; i < 4; ++i) r += c0[i] + c1[i];
  for (int i = 0; i < 16; ++i) r += d0[i];
  r += (float)(r0[0] + r1[1]);
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

static bool sample(double* watts, double* mhz) {
  FILE* f = popen("rocm-smi --showpower --showclocks 2>/dev/null", "r");
  if (!f) return false;
  char line[512];
  bool gw = false, gm = false;
  while (fgets(line, sizeof line, f)) {
    const char* p;
    if ((p = strstr(line, "Package Power (W):")) && !strstr(line, "Max")) { *watts = atof(p + 18); gw = true; }
    if ((p = strstr(line, "sclk clock level")) && (p = strchr(p, '('))) { *mhz = atof(p + 1); gm = true; }
  }
  pclose(f);
  return gw && gm;
}

template <int SHAPE, bool FILL, int NOPER>
void run(float* d, const char* name) {
  const int blocks = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<SHAPE, FILL, NOPER>), dim3(blocks), dim3(256), 0, 0, d, 2000, 1.0001f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<SHAPE, FILL, NOPER>), dim3(blocks), dim3(256), 0, 0, d, 2000, 1.0001f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const int iters = (int)(2000 * 3500.0 / ms);       // ~3.5 s
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<SHAPE, FILL, NOPER>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f);
  hipEventRecord(e1);
  std::vector<double> w, m;
  while (hipEventQuery(e1) == hipErrorNotReady) {
    double ww, mm;
    if (sample(&ww, &mm)) { w.push_back(ww); m.push_back(mm); }
  }
  hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1);
  // drop the first and the last sample (ramp)
  if (w.size() > 4) { w.erase(w.begin()); w.pop_back(); m.erase(m.begin()); m.pop_back(); }
  std::sort(w.begin(), w.end());
  std::sort(m.begin(), m.end());
  const double flop = (double)iters * 4 * 49152.0 * 2 * blocks * 4;
  const double tf = flop / ms / 1e9, pw = w.empty() ? 0 : w[w.size() / 2], mh = m.empty() ? 0 : m[m.size() / 2];
  printf("%-28s %8.1f ms  %7.1f TFLOP/s (%.2f of 2500)  power median %6.0f W (n=%zu)  sclk %5.0f MHz  %.3f pJ/FLOP\n", name, ms, tf,
         tf / 2500, pw, w.size(), mh, pw / tf);
  fflush(stdout);
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 256 * 4);
  run<16, false, 1>(d, "16x16x32 bare, 1 operand");
  run<32, false, 1>(d, "32x32x16 bare, 1 operand");
  run<16, false, 8>(d, "16x16x32 bare, 8 operands");
  run<32, false, 8>(d, "32x32x16 bare, 8 operands");
  run<16, true, 8>(d, "16x16x32 + dvis filler mix");
  run<32, true, 8>(d, "32x32x16 + dvis filler mix");
  run<16, true, 8>(d, "16x16x32 + dvis filler mix");
  run<32, true, 8>(d, "32x32x16 + dvis filler mix");
  return 0;
}
