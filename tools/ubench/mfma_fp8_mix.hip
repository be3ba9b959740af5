// what the exact-operand product mix would gain from running its low-weight class on the fp8 MFMA.
// An fp32 multiply-add is six f16 products today (h.xh | h.xm, m.xh | h.xl, m.xm, l.xh).  tools/emulate_fp8_c2.py shows that the last class
// can take bf8 operands without a measurable change of the error against float64.  This times the candidate mixes per 128 values of K and
// one 16-row tile (one wave per SIMD, 256 workgroups, ~3 s per variant, package power and clock sampled with rocm-smi like mfma_power.hip):
//   V0  6 x v_mfma_f32_16x16x32_f16 per k-block of 32                                  (today: 24 per 128 K)
//   V1  4 x f16 per k-block + 2 x v_mfma_f32_16x16x128_f8f6f4 (bf8) per 128 K          (register- and stream-neutral variant)
//   V2  3 x f16 per k-block + 3 x 16x16x128 bf8 per 128 K                              (the whole low-weight class)
//   V3 / V4  the same with v_mfma_f32_16x16x32_bf8_bf8 (the K = 32 form: the f16 fragment layout at half the bytes)
//   V5 / V6 / V7  bare streams of 16x16x128 bf8 / 16x16x32 bf8 / 16x16x32 f16
// FILL: seven independent v_fma + half a ds_read_b128 per k-block (the ~1.2 vector instructions per MFMA of k_dvis_x6t).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_fp8_mix.bin tools/ubench/mfma_fp8_mix.hip && ./mfma_fp8_mix.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i8v __attribute__((ext_vector_type(8)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int V, bool FILL>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters, float s) {
  __shared__ u4 lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = u4{(unsigned)i, 1u, 2u, 3u};
  __syncthreads();
  constexpr int NOP = 8;
  h8 a[NOP], b[NOP];
  i8v a8[2], b8[2];
  long a2[4], b2[4];
  for (int j = 0; j < NOP; ++j)
    for (int i = 0; i < 8; ++i) {
      a[j][i] = (_Float16)(threadIdx.x * 0.001f + i + j);
      b[j][i] = (_Float16)(threadIdx.x * 0.002f - i - j);
    }
  for (int j = 0; j < 2; ++j)
    for (int i = 0; i < 8; ++i) {
      a8[j][i] = 0x3c3c3c3c + (int)threadIdx.x * 0x01010101 * (i + j + 1);
      b8[j][i] = 0x38383838 + (int)threadIdx.x * 0x00010001 * (i + 2 * j + 1);
    }
  for (int j = 0; j < 4; ++j) {
    a2[j] = 0x3c3c3c3c3c3c3c3cL + (long)threadIdx.x * (j + 1);
    b2[j] = 0x3838383838383838L + (long)threadIdx.x * (j + 3);
  }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
  f4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0}, c2 = {0, 0, 0, 0};
  u4 r0 = {};
  const u4* lp = lds + (threadIdx.x & 63);
  constexpr int NF16 = V == 0 ? 6 : (V == 1 || V == 3 ? 4 : (V == 2 || V == 4 ? 3 : (V == 7 ? 6 : 0)));       // f16 MFMAs per k-block
  constexpr int N128 = V == 1 ? 2 : (V == 2 ? 3 : (V == 5 ? 6 : 0));                                           // 16x16x128 bf8 per 128 K
  constexpr int N32 = V == 3 ? 2 : (V == 4 ? 3 : (V == 6 ? 6 : 0));                                            // 16x16x32 bf8 per k-block
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int grp = 0; grp < 2; ++grp) {          // two groups of 128 K per iteration
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
        for (int u = 0; u < NF16; ++u) {
          const int o = (grp * 24 + kb * 6 + u) % NOP;
          if (u < 1) c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[o], b[(o + 1) % NOP], c0, 0, 0, 0);
          else if (u < 3) c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[o], b[(o + 1) % NOP], c1, 0, 0, 0);
          else c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[o], b[(o + 1) % NOP], c2, 0, 0, 0);
          if constexpr (FILL) {
            asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[u & 7]) : "v"(s));
            if (u == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[6 + (kb & 1)]) : "v"(s));
            if (u == 1 && (kb & 1)) r0 = lp[(it & 7) * 64];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int u = 0; u < N32; ++u) {
          c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf8_bf8(a2[(kb + u) & 3], b2[(kb + u + 1) & 3], c2, 0, 0, 0);
          if constexpr (FILL) {
            asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[(NF16 + u) & 7]) : "v"(s));
            if (NF16 == 0 && u == 1 && (kb & 1)) r0 = lp[(it & 7) * 64];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (FILL && NF16 + N32 < 6) {      // the k-block's remaining vector work (the same instructions in every mixed variant)
#pragma unroll
          for (int u = NF16 + N32; u < 6; ++u) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[u & 7]) : "v"(s));
        }
      }
#pragma unroll
      for (int u = 0; u < N128; ++u) {
        c2 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8[u & 1], b8[(u + 1) & 1], c2, 1, 1, 0, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float r = 0;
  for (int i = 0; i < 8; ++i) r += v[i];
  for (int i = 0; i < 4; ++i) r += c0[i] + c1[i] + c2[i];
  r += (float)r0[0];
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

template <int V, bool FILL>
void run(float* d, const char* name) {
  const int blocks = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<V, FILL>), dim3(blocks), dim3(256), 0, 0, d, 2000, 1.0001f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<V, FILL>), dim3(blocks), dim3(256), 0, 0, d, 2000, 1.0001f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const int iters = (int)(2000 * 3000.0 / ms);       // ~3 s
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<V, FILL>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f);
  hipEventRecord(e1);
  std::vector<double> w, m;
  while (hipEventQuery(e1) == hipErrorNotReady) {
    double ww, mm;
    if (sample(&ww, &mm)) { w.push_back(ww); m.push_back(mm); }
  }
  hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1);
  if (w.size() > 4) { w.erase(w.begin()); w.pop_back(); m.erase(m.begin()); m.pop_back(); }
  std::sort(w.begin(), w.end());
  std::sort(m.begin(), m.end());
  // one "exact multiply-add" = one element of a 16 x 16 x 128 group's K: per iteration 2 groups x 16 x 16 x 128 of them per wave
  const double emac = (double)iters * 2 * 16 * 16 * 128 * blocks * 4;
  const double ns_per_group = ms * 1e6 / ((double)iters * 2);
  const double pw = w.empty() ? 0 : w[w.size() / 2], mh = m.empty() ? 0 : m[m.size() / 2];
  printf("%-58s %7.1f ns per 128 K  %6.1f T exact multiply-adds/s  power median %5.0f W (n=%zu)  sclk %5.0f MHz\n", name, ns_per_group,
         emac / ms / 1e9, pw, w.size(), mh);
  fflush(stdout);
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 256 * 4);
  run<7, false>(d, "bare 16x16x32 f16 (24 per 128 K)");
  run<5, false>(d, "bare 16x16x128 bf8 (6 per 128 K = the same MACs)");
  run<6, false>(d, "bare 16x16x32 bf8 (24 per 128 K = the same MACs)");
  for (int rep = 0; rep < 2; ++rep) {
    run<0, true>(d, "V0 today: 6 f16 per k-block + fillers");
    run<1, true>(d, "V1 4 f16 per k-block + 2 bf8 16x16x128 per 128 K + fillers");
    run<2, true>(d, "V2 3 f16 per k-block + 3 bf8 16x16x128 per 128 K + fillers");
    run<3, true>(d, "V3 4 f16 + 2 bf8 16x16x32 per k-block + fillers");
    run<4, true>(d, "V4 3 f16 + 3 bf8 16x16x32 per k-block + fillers");
  }
  return 0;
}
