// Fused MLP kernels of the per-ray hot path + weight packing + feature construction (C-ABI entry points).
// See mlp_engine.h for the execution scheme.  All pointers are device pointers; nothing allocates.
#include "../../include/robir_hip.h"
#include "common.h"
#include "mlp_engine.h"
#include <type_traits>

namespace rb {

static thread_local char g_err[512] = "";
char* err_buf() { return g_err; }
int device_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
    cus = prop.multiProcessorCount;
  }
  return cus;
}
int persistent_grid(long rounds, int n_workgroups) {
  if (n_workgroups <= 0) n_workgroups = device_cus();
  if (n_workgroups <= 0) return -1;
  return (int)(rounds < n_workgroups ? rounds : n_workgroups);
}

// Range sentinel words (common.h): pinned host memory mapped into every device's address space, allocated once.
static unsigned* g_range_host = nullptr;
static unsigned* g_range_dev = nullptr;
static void range_init() {
  void* h = nullptr;
  if (hipHostMalloc(&h, RB_RANGE_WORDS * sizeof(unsigned), hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) {
    (void)hipGetLastError();
    return;
  }
  memset(h, 0, RB_RANGE_WORDS * sizeof(unsigned));
  void* d = nullptr;
  if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipHostFree(h);
    return;
  }
  g_range_host = static_cast<unsigned*>(h);
  g_range_dev = static_cast<unsigned*>(d);
}
unsigned* range_flags() {
  static const bool once = (range_init(), true);
  (void)once;
  return g_range_dev;
}
unsigned* range_flags_host() {
  (void)range_flags();
  return g_range_host;
}
int fail(const char* what, const char* detail) {
  snprintf(g_err, sizeof(g_err), "%s: %s", what, detail);
  return 1;
}
unsigned* range_flags_host();
int check_launch(const char* kernel) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "%s: launch failed: %s", kernel, hipGetErrorString(e));
    return 2;
  }
  return 0;
}

// =====================================================================================================
// Weight packing.  A packed layer = N/16 chunks; chunk jb = [bias(16jb .. 16jb+15)] ++
// [kb = 0..K/16-1][lane = 0..63][r = 0..3] -> W[16jb + (lane&15)][16kb + 4(lane>>4) + r].
// k_perm (optional, length k_pad): packed input column k reads source column k_perm[k]; -1 = structural zero.
// =====================================================================================================
__global__ void k_pack_layer(const float* __restrict__ W, const float* __restrict__ b, int n_out, int k_in, int n_pad,
                             int k_pad, const int* __restrict__ k_perm, float w_scale, float* __restrict__ out) {
  const long chunk = 16 + (long)k_pad * 16;
  const long total = (long)(n_pad / 16) * chunk;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int jb = (int)(i / chunk);
    const long o = i - (long)jb * chunk;
    float v = 0.f;
    if (o < 16) {
      const int j = jb * 16 + (int)o;
      if (b != nullptr && j < n_out) v = b[j];
    } else {
      const long q = o - 16;
      const int r = (int)(q & 3), lane = (int)((q >> 2) & 63), kb = (int)(q >> 8);
      const int j = jb * 16 + (lane & 15);
      int k = kb * 16 + 4 * (lane >> 4) + r;
      if (k_perm != nullptr) k = k_perm[k];
      if (j < n_out && k >= 0 && k < k_in) v = W[(long)j * k_in + k] * w_scale;
    }
    out[i] = v;
  }
}

// f16x3 packing (mlp_engine.h, H3Ring): chunk jb = [bias*2^s (16 floats)] ++ [kb][hi|lo][lane][8 halves],
// half slot (lane, j) = W[16jb + (lane&15)][32kb + (j<4 ? 4g+j : 16+4g+j-4)] * 2^s, g = lane>>4.
__global__ void k_pack_layer_h3(const float* __restrict__ W, const float* __restrict__ b, int n_out, int k_in,
                                int n_pad, int k_pad, const int* __restrict__ perm, float scale, float* __restrict__ out) {
  const long chunk = 16 + (long)k_pad * 16;        // in floats (one float = two halves)
  const long total = (long)(n_pad / 16) * chunk;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int jb = (int)(i / chunk);
    const long o = i - (long)jb * chunk;
    if (o < 16) {
      const int j = jb * 16 + (int)o;
      out[i] = (b != nullptr && j < n_out) ? b[j] * scale : 0.f;
      continue;
    }
    // float index q inside the half area: [kb][hilo][lane][4 floats = 8 halves]
    const long q = o - 16;
    const int pair = (int)(q & 3), lane = (int)((q >> 2) & 63), hilo = (int)((q >> 8) & 1), kb = (int)(q >> 9);
    const int row = jb * 16 + (lane & 15), g = lane >> 4;
    _Float16 hv[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int j = pair * 2 + e;
      const int k = 32 * kb + (j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4));
      float v = 0.f;
      const int kin = perm ? perm[k] : k;          // packed slot k <- input column kin (-1: zero padding)
      if (row < n_out && kin >= 0 && kin < k_in) v = W[(long)row * k_in + kin] * scale;
      const _Float16 h = (_Float16)v;
      hv[e] = hilo == 0 ? h : (_Float16)(v - (float)h);
    }
    union {
      _Float16 h[2];
      float f;
    } u;
    u.h[0] = hv[0];
    u.h[1] = hv[1];
    out[i] = u.f;
  }
}

// f16x6 packing (csrc/vis_diffuse_x6.hip): every weight as three halves, w 2^s = h + m 2^-11 + l 2^-22 exactly
// (h, m round-to-nearest, the residuals are exact in fp32).  chunk jb = [bias*2^s (16 floats)] ++ [kb][piece h|m|l][lane][8 halves],
// half slots mapped to input columns like the f16x3 layout above.
__global__ void k_pack_layer_x6(const float* __restrict__ W, const float* __restrict__ b, int n_out, int k_in,
                                int n_pad, int k_pad, const int* __restrict__ perm, float scale, float* __restrict__ out) {
  const long chunk = 16 + (long)k_pad * 24;        // in floats (one float = two halves)
  const long total = (long)(n_pad / 16) * chunk;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int jb = (int)(i / chunk);
    const long o = i - (long)jb * chunk;
    if (o < 16) {
      const int j = jb * 16 + (int)o;
      out[i] = (b != nullptr && j < n_out) ? b[j] * scale : 0.f;
      continue;
    }
    const long q = o - 16;                         // [kb][piece][lane][4 floats = 8 halves]
    const int pair = (int)(q & 3), lane = (int)((q >> 2) & 63), kp = (int)(q >> 8), kb = kp / 3, piece = kp - 3 * kb;
    const int row = jb * 16 + (lane & 15), g = lane >> 4;
    _Float16 hv[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int j = pair * 2 + e;
      const int k = 32 * kb + (j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4));
      float v = 0.f;
      const int kin = perm ? perm[k] : k;
      if (row < n_out && kin >= 0 && kin < k_in) v = W[(long)row * k_in + kin] * scale;
      const _Float16 h = (_Float16)v;
      const float r1 = (v - (float)h) * 2048.0f;   // exact
      const _Float16 m = (_Float16)r1;
      const float r2 = (r1 - (float)m) * 2048.0f;  // exact
      hv[e] = piece == 0 ? h : (piece == 1 ? m : (_Float16)r2);
    }
    union {
      _Float16 h[2];
      float f;
    } u;
    u.h[0] = hv[0];
    u.h[1] = hv[1];
    out[i] = u.f;
  }
}

// =====================================================================================================
// Feature construction (positional encodings).  Accurate sinf/cosf: arguments reach 2^9*|x|.
// PE layout (model/embedder.py:17-38): [x(3) | sin(2^0 x)(3) | cos(2^0 x)(3) | sin(2^1 x)(3) | ...].
// =====================================================================================================
template <int L>
__device__ __forceinline__ void write_pe(const float x[3], float* __restrict__ dst) {
  dst[0] = x[0];
  dst[1] = x[1];
  dst[2] = x[2];
#pragma unroll
  for (int k = 0; k < L; ++k) {
    const float f = (float)(1 << k);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float a = x[c] * f;
      float sn, cs;
      sincosf(a, &sn, &cs);                  // one argument reduction for both (same values as sinf / cosf)
      dst[3 + 6 * k + c] = sn;
      dst[3 + 6 * k + 3 + c] = cs;
    }
  }
}
// d(PE)/dx_c : only the components of coordinate c are non-zero.
template <int L>
__device__ __forceinline__ void write_pe_tangent(const float x[3], int c, float* __restrict__ dst) {
#pragma unroll
  for (int i = 0; i < 3 + 6 * L; ++i) dst[i] = 0.f;
  dst[c] = 1.f;
#pragma unroll
  for (int k = 0; k < L; ++k) {
    const float f = (float)(1 << k);
    const float a = x[c] * f;
    float sn, cs;
    sincosf(a, &sn, &cs);
    dst[3 + 6 * k + c] = f * cs;
    dst[3 + 6 * k + 3 + c] = -f * sn;
  }
}

// Sixteen threads per row, one float4 of the row each: the stores of a wave are 4 x 256 contiguous bytes (a thread per row wrote
// 64 floats 256 B apart from its neighbours': 0.29 ms per 2^20 rows, neither compute- nor bandwidth-bound).  Every feature is the
// same sincosf of the same argument as before: identical rows.
__device__ __forceinline__ float pe10_feature(const float a[3], int f, int tangent_of /* -1: value row */, float last) {
  if (f == 63) return tangent_of < 0 ? last : 0.f;
  if (f < 3) return tangent_of < 0 ? a[f] : (f == tangent_of ? 1.f : 0.f);
  const int k = (f - 3) / 6, r = (f - 3) - 6 * k, c = r >= 3 ? r - 3 : r;
  if (tangent_of >= 0 && c != tangent_of) return 0.f;
  const float fr = (float)(1 << k);
  float sn, cs;
  sincosf((c == 0 ? a[0] : (c == 1 ? a[1] : a[2])) * fr, &sn, &cs);
  if (tangent_of < 0) return r >= 3 ? cs : sn;
  return r >= 3 ? -fr * sn : fr * cs;
}
// X[M,128] = [PE10(p) | PE10(d) | 0 0]   (VisNetwork input, implicit_differentiable_renderer.py:250-256); 32 threads per row
__global__ void k_feat_vis(const float* __restrict__ p, const float* __restrict__ d, long M, int rep,
                           float* __restrict__ X) {
  for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < M * 32; t += (long)gridDim.x * blockDim.x) {
    const long i = t >> 5;
    const int q = (int)(t & 31);
    const long ip = i / rep;  // each point is paired with `rep` consecutive directions
    const float a[3] = {p[3 * ip], p[3 * ip + 1], p[3 * ip + 2]};
    const float b[3] = {d[3 * i], d[3 * i + 1], d[3 * i + 2]};
    f4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int f = 4 * q + e;
      v[e] = f < 63 ? pe10_feature(a, f, -1, 0.f) : (f < 126 ? pe10_feature(b, f - 63, -1, 0.f) : 0.f);
    }
    reinterpret_cast<f4*>(X + i * 128)[q] = v;
  }
}

// X[M,64] = [PE10(x*scale) | extra]  (extra = 0, or hdr_shift for the indirect-illumination net)
// jvp != 0: X[4M,64], row 4m = PE, rows 4m+1..3 = dPE/dx, dPE/dy, dPE/dz  (forward-mode SDF gradient)
__global__ void k_feat_pe10(const float* __restrict__ x, long M, float scale, const float* __restrict__ extra, int jvp,
                            float* __restrict__ X) {
  const long rows = jvp ? 4 * M : M;
  for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < rows * 16; t += (long)gridDim.x * blockDim.x) {
    const long row = t >> 4;
    const int q = (int)(t & 15);
    const long i = jvp ? row >> 2 : row;
    const int tangent_of = jvp ? (int)(row & 3) - 1 : -1;
    const float a[3] = {x[3 * i] * scale, x[3 * i + 1] * scale, x[3 * i + 2] * scale};
    const float last = (!jvp && extra) ? extra[i] : 0.f;
    f4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = pe10_feature(a, 4 * q + e, tangent_of, last);
    reinterpret_cast<f4*>(X + row * 64)[q] = v;
  }
}

// Integrated PE, isotropic covariance var*I, full-covariance code path of the reference
// (model/neus_model.py:14-57): [exp(-v_k/2) sin(2^k x_c) (k-major, 30) | exp(-v_k/2) sin(2^k x_c + pi/2) (30)],
// arguments wrapped mod 100*pi when |arg| >= 100*pi.  X[M,64], columns 60..63 zero.
__device__ __forceinline__ float py_mod(float a, float m) {  // torch.remainder: sign of the divisor
  float r = fmodf(a, m);
  if (r != 0.f && ((r < 0.f) != (m < 0.f))) r += m;
  return r;
}
__global__ void k_feat_ipe(const float* __restrict__ x, long M, float var, const float* __restrict__ noise,
                           float noise_scale, float* __restrict__ X) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= M) return;
  const float big = (float)(100.0 * 3.14159265358979323846);
  const float half_pi = (float)(0.5 * 3.14159265358979323846);
  float* row = X + i * 64;
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    const float f = (float)(1 << k);
    const float damp = expf(-0.5f * (var * (f * f)));
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float y = x[3 * i + c] * f;
      float y2 = y + half_pi;
      float a1 = fabsf(y) < big ? y : py_mod(y, big);
      float a2 = fabsf(y2) < big ? y2 : py_mod(y2, big);
      float e1 = damp * sinf(a1), e2 = damp * sinf(a2);
      if (noise) {
        e1 += noise[i * 60 + 3 * k + c] * noise_scale;
        e2 += noise[i * 60 + 30 + 3 * k + c] * noise_scale;
      }
      row[3 * k + c] = e1;
      row[30 + 3 * k + c] = e2;
    }
  }
  row[60] = row[61] = row[62] = row[63] = 0.f;
}

// X[M,304] = [feat*feat_scale (256) | x*x_scale (3) | PE4(view) (27) | normal (3) | 0 x15]
// (RenderingNetwork input, model/neus_model.py:535-545; the packed first layer is column-permuted to match)
__global__ void k_feat_color(const float* __restrict__ x, float x_scale, const float* __restrict__ view,
                             const float* __restrict__ normal, const float* __restrict__ feat, long feat_stride,
                             float feat_scale, long M, float* __restrict__ X) {
  const int t = threadIdx.x;  // 256 threads: one feature each; rows by grid stride (M * 256 may exceed 2^32 work-items)
  for (long i = blockIdx.x; i < M; i += gridDim.x) {
    float* row = X + i * 304;
    row[t] = feat[i * feat_stride + t] * feat_scale;
    if (t == 0) {
      float v[3] = {view[3 * i], view[3 * i + 1], view[3 * i + 2]};
      row[256] = x[3 * i] * x_scale;
      row[257] = x[3 * i + 1] * x_scale;
      row[258] = x[3 * i + 2] * x_scale;
      write_pe<4>(v, row + 259);
      row[286] = normal[3 * i];
      row[287] = normal[3 * i + 1];
      row[288] = normal[3 * i + 2];
#pragma unroll
      for (int k = 289; k < 304; ++k) row[k] = 0.f;
    }
  }
}

// T[M,48] = columns 256..303 of the row above: [x*x_scale (3) | PE4(view) (27) | normal (3) | 0 x15] (rb_color_mlp_h3_two)
__global__ void k_feat_color_tail(const float* __restrict__ x, float x_scale, const float* __restrict__ view,
                                  const float* __restrict__ normal, long M, float* __restrict__ T) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= M) return;
  float row[48];
  float v[3] = {view[3 * i], view[3 * i + 1], view[3 * i + 2]};
  row[0] = x[3 * i] * x_scale;
  row[1] = x[3 * i + 1] * x_scale;
  row[2] = x[3 * i + 2] * x_scale;
  write_pe<4>(v, row + 3);
  row[30] = normal[3 * i];
  row[31] = normal[3 * i + 1];
  row[32] = normal[3 * i + 2];
#pragma unroll
  for (int k = 33; k < 48; ++k) row[k] = 0.f;
  f4* dst = reinterpret_cast<f4*>(T + i * 48);
#pragma unroll
  for (int k = 0; k < 12; ++k) dst[k] = f4{row[4 * k], row[4 * k + 1], row[4 * k + 2], row[4 * k + 3]};
}

// =====================================================================================================
// Kernels.  All: 256 threads (4 waves), 1 wave per SIMD, rows_per_block = 64*NT.
// =====================================================================================================

// ---- visibility MLP: X[M,128] -> logits[M,2]   126(128) -> 256 x4 ReLU -> 2(16)
// FUSED: X = points p, Xd = directions d (rep directions per point): [PE10(p) | PE10(d)] encoded in the kernel (load_features_vis).
template <bool FUSED>
__global__ __launch_bounds__(256, 1) void k_vis_mlp(const float* __restrict__ X, long M, const f4* __restrict__ Wp,
                                                     float* __restrict__ Y, const float* __restrict__ Xd, int rep) {
  __shared__ f4 lds[2 * chunk_f4(256)];
  __shared__ float pe_scratch[FUSED ? 4 * 2 * 16 * 128 : 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  WStream<256> ws;
  ws.init(lds, tid);
  const f4* wl0 = Wp;
  const f4* wl1 = wl0 + layer_f4<128, 256>();
  const f4* wl4 = wl1 + 3 * layer_f4<256, 256>();
  const long row0 = ((long)blockIdx.x * 4 + wave) * 32 + (lane & 15);
  float h[2][64], z[2][64];
  {
    float in0[2][32];
    if constexpr (FUSED) {
      load_features_vis(X, Xd, rep, row0, M, lane, pe_scratch + (wave * 2 + 0) * 2048, in0[0]);
      load_features_vis(X, Xd, rep, row0 + 16, M, lane, pe_scratch + (wave * 2 + 1) * 2048, in0[1]);
    } else {
      load_features<128>(X, row0, M, lane, in0[0]);
      load_features<128>(X, row0 + 16, M, lane, in0[1]);
    }
    ws.prime<chunk_f4(128)>(wl0);
    dense_layer<128, 256, 2, 256>(ws, wl0, wl1, in0, z, lane, true);
  }
  activate<256, 2, ACT_RELU>(z, h);
#pragma unroll 1
  for (int l = 0; l < 3; ++l) {
    const f4* wl = wl1 + l * layer_f4<256, 256>();
    dense_layer<256, 256, 2, 256>(ws, wl, wl + layer_f4<256, 256>(), h, z, lane, true);
    activate<256, 2, ACT_RELU>(z, h);
  }
  float o[2][4];
  dense_layer<256, 16, 2, 0>(ws, wl4, nullptr, h, o, lane, true);
  if ((lane >> 4) == 0) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      long row = row0 + 16 * t;
      if (row < M) {
        Y[row * 2] = o[t][0];
        Y[row * 2 + 1] = o[t][1];
      }
    }
  }
}

// ---- single linear layer X[M,64] -> Y[M,256] (no activation); used to split the visibility net's first layer
//      into a per-point and a per-direction half for the fused diffuse-visibility kernel.
// FUSED: X = points / directions [M,3], encoded in the kernel.
template <bool FUSED>
__global__ __launch_bounds__(256, 1) void k_linear_64_256(const float* __restrict__ X, long M,
                                                           const f4* __restrict__ Wp, float* __restrict__ Y) {
  __shared__ f4 lds[2 * chunk_f4(64)];
  __shared__ float pe_scratch[FUSED ? 4 * 2 * 16 * 64 : 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  WStream<64> ws;
  ws.init(lds, tid);
  const long row0 = ((long)blockIdx.x * 4 + wave) * 32 + (lane & 15);
  float in0[2][16], z[2][64];
  if constexpr (FUSED) {
    load_features_pe10x(X, nullptr, row0, M, lane, pe_scratch + (wave * 2 + 0) * 1024, in0[0]);
    load_features_pe10x(X, nullptr, row0 + 16, M, lane, pe_scratch + (wave * 2 + 1) * 1024, in0[1]);
  } else {
    load_features<64>(X, row0, M, lane, in0[0]);
    load_features<64>(X, row0 + 16, M, lane, in0[1]);
  }
  ws.prime<chunk_f4(64)>(Wp);
  dense_layer<64, 256, 2, 0>(ws, Wp, nullptr, in0, z, lane, true);
  const int g = lane >> 4;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    long row = row0 + 16 * t;
    if (row < M) {
      f4* dst = reinterpret_cast<f4*>(Y + row * 256) + g;
#pragma unroll
      for (int jb = 0; jb < 16; ++jb) dst[jb * 4] = f4{z[t][jb * 4], z[t][jb * 4 + 1], z[t][jb * 4 + 2], z[t][jb * 4 + 3]};
    }
  }
}

// ---- NeuS SDF network (model/neus_model.py:385-417): PE10 (64) -> 256,256,256,193(208) -> skip cat /sqrt2 (272)
//      -> 256 x4 -> 257(272) | 1(16);  Softplus(beta=100, threshold 20).
// MODE 0: sdf only -> out0[M]          MODE 1: sdf+feat -> out0[M,257]
// MODE 2: forward-mode jvp, sdf only: X[4M,64] (value row + 3 tangent rows per point); out0[M], grad[M,3]
// MODE 3: forward-mode jvp, full:     X[4M,64]; out0[M,257], grad[M,3]
// FUSED: the rows are not read but encoded in the kernel from the points (mlp_engine.h: load_features_pe10).
template <int MODE, bool PRECISE = false, bool FUSED = false>
__global__ __launch_bounds__(256, 1) void k_sdf_mlp(const float* __restrict__ X, long MR, const f4* __restrict__ Wp,
                                                     float out_scale, float grad_scale, float* __restrict__ out0,
                                                     float* __restrict__ grad, float in_scale) {
  // MODE 5 (value rows, all 257 outputs): also stores sigmoid(100 z) of every hidden pre-activation for the reverse-mode gradient
  // pass k_sdf_back_f32 (below) through `grad` -- [tile][layer 8][k-block 16][lane 64] float4, in the register order of this engine.
  constexpr bool JVP = MODE == 2 || MODE == 3;
  constexpr bool FULL = (MODE == 1 || MODE == 3 || MODE == 5);
  constexpr bool STORE = MODE == 5;
  constexpr int NL = FULL ? 272 : 16;
  __shared__ f4 lds[2 * chunk_f4(272)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  WStream<272> ws;
  ws.init(lds, tid);
  constexpr long LF = layer_f4<256, 256>();
  const f4* w0 = Wp;
  const f4* w1 = w0 + layer_f4<64, 256>();
  const f4* w3 = w1 + 2 * LF;
  const f4* w4 = w3 + layer_f4<256, 208>();
  const f4* w5 = w4 + layer_f4<272, 256>();
  const f4* w8 = w5 + 3 * LF;
  const long row0 = ((long)blockIdx.x * 4 + wave) * 32 + (lane & 15);  // row of X (a jvp column when JVP)
  const bool bias_on = JVP ? ((lane & 3) == 0) : true;
  const float inv_sqrt2 = 0.70710678118654752440f;
  float x0[2][16], ha[2][64], z[2][64];
  if constexpr (FUSED) {                               // X = the points xyz[M,3]
    __shared__ float pe_scratch[4 * 2 * 16 * 64];      // 32 KB: one encoded tile per wave and tile
    load_features_pe10<JVP>(X, in_scale, row0, MR, lane, pe_scratch + (wave * 2 + 0) * 1024, x0[0]);
    load_features_pe10<JVP>(X, in_scale, row0 + 16, MR, lane, pe_scratch + (wave * 2 + 1) * 1024, x0[1]);
  } else {
    load_features<64>(X, row0, MR, lane, x0[0]);
    load_features<64>(X, row0 + 16, MR, lane, x0[1]);
  }
  // activation of one hidden layer (softplus_into), in MODE 5 with the sigmoid store
  f4* sig_tile = nullptr;
  if constexpr (STORE) sig_tile = reinterpret_cast<f4*>(grad) + (((long)blockIdx.x * 4 + wave) * 2) * (8L * 16 * 64) + lane;
  auto act = [&](auto nreg_tag, auto hreg_tag, const auto& zz, auto& hh, float scale, int layer) {
    constexpr int NREG = decltype(nreg_tag)::value, HREG = decltype(hreg_tag)::value;
    if constexpr (STORE) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int kb = 0; kb < NREG / 4; ++kb) {
          f4 sg;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float s1;
            hh[t][kb * 4 + r] = softplus100<PRECISE>(zz[t][kb * 4 + r], &s1) * scale;
            sg[r] = s1;
          }
          sig_tile[((long)t * 8 + layer) * (16 * 64) + kb * 64] = sg;
        }
    } else {
      softplus_into<NREG, HREG, JVP, PRECISE>(zz, hh, lane, scale);
    }
  };
  using I52 = std::integral_constant<int, 52>;
  using I64 = std::integral_constant<int, 64>;
  using I68 = std::integral_constant<int, 68>;
  ws.prime<chunk_f4(64)>(w0);
  dense_layer<64, 256, 2, 256>(ws, w0, w1, x0, z, lane, bias_on);
  act(I64{}, I64{}, z, ha, 1.0f, 0);
#pragma unroll 1
  for (int l = 0; l < 2; ++l) {
    dense_layer<256, 256, 2, 256>(ws, w1 + l * LF, w1 + (l + 1) * LF, ha, z, lane, bias_on);
    act(I64{}, I64{}, z, ha, 1.0f, 1 + l);
  }
  {
    float hs[2][68];
    {
      float z3[2][52];
      dense_layer<256, 208, 2, 272>(ws, w3, w4, ha, z3, lane, bias_on);
      act(I52{}, I68{}, z3, hs, inv_sqrt2, 3);   // neurons 193..207 are padding: zero weights downstream
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) hs[t][52 + i] = x0[t][i] * inv_sqrt2;
    dense_layer<272, 256, 2, 256>(ws, w4, w5, hs, z, lane, bias_on);
  }
  act(I64{}, I64{}, z, ha, 1.0f, 4);
#pragma unroll 1
  for (int l = 0; l < 3; ++l) {
    dense_layer<256, 256, 2, 256>(ws, w5 + l * LF, w5 + (l + 1) * LF, ha, z, lane, bias_on);
    act(I64{}, I64{}, z, ha, 1.0f, 5 + l);
  }
  float zo[2][NL / 4];
  dense_layer<256, NL, 2, 0>(ws, w8, nullptr, ha, zo, lane, bias_on);

  const int g = lane >> 4;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const long row = row0 + 16 * t;
    if (row >= MR) continue;
    if constexpr (!JVP) {
      if constexpr (FULL) {
#pragma unroll
        for (int jb = 0; jb < NL / 16; ++jb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int j = jb * 16 + 4 * g + r;
            if (j < 257) out0[row * 257 + j] = zo[t][jb * 4 + r] * out_scale;
          }
      } else {
        if (g == 0) out0[row] = zo[t][0] * out_scale;
      }
    } else {
      const long m = row >> 2;
      const int c = (int)(row & 3);
      if (c == 0) {
        if constexpr (FULL) {
#pragma unroll
          for (int jb = 0; jb < NL / 16; ++jb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int j = jb * 16 + 4 * g + r;
              if (j < 257) out0[m * 257 + j] = zo[t][jb * 4 + r] * out_scale;
            }
        } else {
          if (g == 0) out0[m] = zo[t][0] * out_scale;
        }
      } else if (g == 0) {
        grad[m * 3 + (c - 1)] = zo[t][0] * grad_scale;
      }
    }
  }
}

// ---- d sdf / d (encoded input) by reverse mode on the f32-input MFMA (the exact policy's counterpart of sdf_back.hip): the sigmoid
// tiles of k_sdf_mlp<5> in, two 64-wide gradient rows per point out (layer 0's and the skip connection's share: k_pe_grad_points
// contracts them with the encoding's Jacobian).  One pass over the transposed layers instead of the three tangent rows per point of
// the forward-mode kernels (modes 2 / 3): 2 x the value pass's MACs instead of 4 x.
// Wt (packing.pack_sdf_back): W7^T, W6^T, W5^T, [W4^T: 193 -> 208 rows | 63 -> 64 skip rows] (N = 272), W3^T (K = 208), W2^T, W1^T,
// W0^T (N = 64), no biases; w8row = row 0 of layer 8 (d sdf / d h7).
__global__ __launch_bounds__(256, 1) void k_sdf_back_f32(const f4* __restrict__ sig, long M, const f4* __restrict__ Wt,
                                                          const float* __restrict__ w8row, float* __restrict__ gfeat) {
  __shared__ f4 lds[2 * chunk_f4(272)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  WStream<272> ws;
  ws.init(lds, tid);
  constexpr long LF = layer_f4<256, 256>();
  const f4* b7 = Wt;
  const f4* b4 = b7 + 3 * LF;
  const f4* b3 = b4 + layer_f4<256, 272>();
  const f4* b2 = b3 + layer_f4<208, 256>();
  const f4* b0 = b2 + 2 * LF;
  const long row0 = ((long)blockIdx.x * 4 + wave) * 32 + (lane & 15);
  const f4* sig_tile = sig + (((long)blockIdx.x * 4 + wave) * 2) * (8L * 16 * 64) + lane;
  const float inv_sqrt2 = 0.70710678118654752440f;
  float dh[2][64], dz[2][64];
  // dz = dh (.) sigmoid(100 z) of `layer`, NB k-blocks
  auto gate = [&](auto nb_tag, const auto& hh, auto& zz, int layer) {
    constexpr int NB = decltype(nb_tag)::value;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) {
        const f4 sg = sig_tile[((long)t * 8 + layer) * (16 * 64) + kb * 64];
#pragma unroll
        for (int r = 0; r < 4; ++r) zz[t][kb * 4 + r] = hh[t][kb * 4 + r] * sg[r];
      }
  };
  using I13 = std::integral_constant<int, 13>;
  using I16 = std::integral_constant<int, 16>;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) {
      const f4 v = *reinterpret_cast<const f4*>(w8row + kb * 16 + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) dh[t][kb * 4 + r] = v[r];
    }
  ws.prime<chunk_f4(256)>(b7);
#pragma unroll 1
  for (int l = 0; l < 3; ++l) {                 // layers 7, 6, 5
    gate(I16{}, dh, dz, 7 - l);
    dense_layer<256, 256, 2, 256>(ws, b7 + l * LF, l < 2 ? b7 + (l + 1) * LF : b4, dz, dh, lane, false);
  }
  gate(I16{}, dh, dz, 4);
  float skip[2][16];
  {
    float dhs[2][68];
    dense_layer<256, 272, 2, 208>(ws, b4, b3, dz, dhs, lane, false);
    float dz3[2][52];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int i = 0; i < 52; ++i) dh[t][i] = dhs[t][i] * inv_sqrt2;
#pragma unroll
      for (int i = 0; i < 16; ++i) skip[t][i] = dhs[t][52 + i] * inv_sqrt2;
    }
    gate(I13{}, dh, dz3, 3);
    dense_layer<208, 256, 2, 256>(ws, b3, b2, dz3, dh, lane, false);
  }
  gate(I16{}, dh, dz, 2);
  dense_layer<256, 256, 2, 256>(ws, b2, b2 + LF, dz, dh, lane, false);
  gate(I16{}, dh, dz, 1);
  dense_layer<256, 256, 2, 256>(ws, b2 + LF, b0, dz, dh, lane, false);
  gate(I16{}, dh, dz, 0);
  float dx[2][16];
  dense_layer<256, 64, 2, 0>(ws, b0, nullptr, dz, dx, lane, false);
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const long row = row0 + 16 * t;
    if (row >= M) continue;
    f4* dst = reinterpret_cast<f4*>(gfeat + row * 128) + g;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      dst[kb * 4] = f4{dx[t][kb * 4], dx[t][kb * 4 + 1], dx[t][kb * 4 + 2], dx[t][kb * 4 + 3]};
      dst[16 + kb * 4] = f4{skip[t][kb * 4], skip[t][kb * 4 + 1], skip[t][kb * 4 + 2], skip[t][kb * 4 + 3]};
    }
  }
}

// host-side launchers for sdf_back.hip (rb_sdf_value_grad_f32_points)
int launch_sdf_f32_store(const float* xyz, long M, float in_scale, const float* Wp, float out_scale, float* out0, float* sig,
                         hipStream_t s) {
  hipLaunchKernelGGL((k_sdf_mlp<5, false, true>), grid1d(M, 128), dim3(256), 0, s, xyz, M, (const f4*)Wp, out_scale, 1.0f, out0, sig, in_scale);
  return check_launch("k_sdf_mlp<5>");
}
int launch_sdf_back_f32(const float* sig, long M, const float* Wt, const float* w8row, float* gfeat, hipStream_t s) {
  hipLaunchKernelGGL(k_sdf_back_f32, grid1d(M, 128), dim3(256), 0, s, (const f4*)sig, M, (const f4*)Wt, w8row, gfeat);
  return check_launch("k_sdf_back_f32");
}

// ---- NeuS colour network (model/neus_model.py:535-560): 289(304) -> 256 x4 ReLU -> 3(16) -> sigmoid
// FUSED: no assembled rows: the 256 feature columns are read in place from the SDF net's output rows and lane (n, g) ENCODES its
// twelve tail columns [x * x_scale | PE4(view) | normal | 0 x15] itself (embedview_fn, model/neus_model.py:535-545; the sincosf
// calls of write_pe<4>: bit-identical operands).
__device__ __forceinline__ float color_tail_feature(const float* __restrict__ px, float x_scale, const float* __restrict__ pv,
                                                    const float* __restrict__ pn, long i, int f) {
  if (f < 3) return px[3 * i + f] * x_scale;
  if (f < 6) return pv[3 * i + f - 3];
  if (f >= 30) return f < 33 ? pn[3 * i + f - 30] : 0.f;
  const int k = (f - 6) / 6, r = (f - 6) - 6 * k, c = r >= 3 ? r - 3 : r;
  float sn, cs;
  sincosf(pv[3 * i + c] * (float)(1 << k), &sn, &cs);
  return r >= 3 ? cs : sn;
}
template <bool FUSED>
__global__ __launch_bounds__(256, 1) void k_color_mlp(const float* __restrict__ X, long M, const f4* __restrict__ Wp,
                                                       float* __restrict__ rgb, const float* __restrict__ feat, long feat_stride,
                                                       float feat_scale, const float* __restrict__ px, float x_scale,
                                                       const float* __restrict__ pv, const float* __restrict__ pn) {
  __shared__ f4 lds[2 * chunk_f4(304)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  WStream<304> ws;
  ws.init(lds, tid);
  constexpr long LF = layer_f4<256, 256>();
  const f4* w0 = Wp;
  const f4* w1 = w0 + layer_f4<304, 256>();
  const f4* w4 = w1 + 3 * LF;
  const long row0 = ((long)blockIdx.x * 4 + wave) * 32 + (lane & 15);
  float h[2][64], z[2][64];
  {
    float in0[2][76];
    if constexpr (FUSED) {
      const int g = lane >> 4;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const long row = row0 + 16 * t;
        const bool ok = row < M;
        const long rr = ok ? row : 0;
        const float* pf = feat + rr * feat_stride + g * 4;
#pragma unroll
        for (int kb = 0; kb < 16; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) in0[t][kb * 4 + r] = ok ? pf[kb * 16 + r] * feat_scale : 0.f;
#pragma unroll 1
        for (int kb = 0; kb < 3; ++kb) {
          float v[4];
#pragma unroll 1
          for (int r = 0; r < 4; ++r) v[r] = color_tail_feature(px, x_scale, pv, pn, rr, 16 * kb + 4 * g + r);
#pragma unroll
          for (int r = 0; r < 4; ++r) in0[t][64 + kb * 4 + r] = ok ? v[r] : 0.f;
        }
      }
    } else {
      load_features<304>(X, row0, M, lane, in0[0]);
      load_features<304>(X, row0 + 16, M, lane, in0[1]);
    }
    ws.prime<chunk_f4(304)>(w0);
    dense_layer<304, 256, 2, 256>(ws, w0, w1, in0, z, lane, true);
  }
  activate<256, 2, ACT_RELU>(z, h);
#pragma unroll 1
  for (int l = 0; l < 3; ++l) {
    dense_layer<256, 256, 2, 256>(ws, w1 + l * LF, w1 + (l + 1) * LF, h, z, lane, true);
    activate<256, 2, ACT_RELU>(z, h);
  }
  float o[2][4];
  dense_layer<256, 16, 2, 0>(ws, w4, nullptr, h, o, lane, true);
  if ((lane >> 4) == 0) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const long row = row0 + 16 * t;
      if (row < M) {
#pragma unroll
        for (int c = 0; c < 3; ++c) rgb[row * 3 + c] = 1.0f / (1.0f + expf(-o[t][c]));
      }
    }
  }
}

// ---- 512-wide nets, one 16-sample tile per wave.
// ENC = false: IndirctIllumNetwork.lobe_layer  64 -> 512 x4 ReLU -> 144            (raw outputs [M,144])
// ENC = true : SparseAE encoder                64 -> 512 x4 LeakyReLU(0.2) -> 32   (raw latent  [M,32])
// FUSED: X = points [M,3], extra [M] or NULL = column 63 (hdr_shift of the indirect-illumination net): [PE10(x) | extra] encoded here.
template <bool ENC, bool FUSED = false>
__global__ __launch_bounds__(256, 1) void k_wide_mlp(const float* __restrict__ X, long M, const f4* __restrict__ Wp,
                                                      float* __restrict__ Y, const float* __restrict__ extra = nullptr) {
  constexpr int NO = ENC ? 32 : 144;
  constexpr int ACT = ENC ? ACT_LEAKY02 : ACT_RELU;
  __shared__ f4 lds[2 * chunk_f4(512)];
  __shared__ float pe_scratch[FUSED ? 4 * 16 * 64 : 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  WStream<512> ws;
  ws.init(lds, tid);
  constexpr long LF = layer_f4<512, 512>();
  const f4* w0 = Wp;
  const f4* w1 = w0 + layer_f4<64, 512>();
  const f4* w4 = w1 + 3 * LF;
  const long row = ((long)blockIdx.x * 4 + wave) * 16 + (lane & 15);
  float h[1][128], z[1][128];
  {
    float in0[1][16];
    if constexpr (FUSED) {
      load_features_pe10x(X, extra, row, M, lane, pe_scratch + wave * 1024, in0[0]);
    } else {
      load_features<64>(X, row, M, lane, in0[0]);
    }
    ws.prime<chunk_f4(64)>(w0);
    dense_layer<64, 512, 1, 512>(ws, w0, w1, in0, z, lane, true);
  }
  activate<512, 1, ACT>(z, h);
#pragma unroll 1
  for (int l = 0; l < 3; ++l) {
    dense_layer<512, 512, 1, 512>(ws, w1 + l * LF, w1 + (l + 1) * LF, h, z, lane, true);
    activate<512, 1, ACT>(z, h);
  }
  float o[1][NO / 4];
  dense_layer<512, NO, 1, 0>(ws, w4, nullptr, h, o, lane, true);
  if (row < M) {
    const int g = lane >> 4;
#pragma unroll
    for (int jb = 0; jb < NO / 16; ++jb) {
      f4* dst = reinterpret_cast<f4*>(Y + row * NO + jb * 16) + g;
      *dst = f4{o[0][jb * 4], o[0][jb * 4 + 1], o[0][jb * 4 + 2], o[0][jb * 4 + 3]};
    }
  }
}

// ---- SparseAE decoder (sg_envmap_material.py:61-68): 32 -> 128 -> 128 LeakyReLU(0.2) -> n_out(16)
__global__ __launch_bounds__(256, 1) void k_ae_decode(const float* __restrict__ L, long M, const f4* __restrict__ Wp,
                                                       int n_out, int sigmoid_out, float* __restrict__ Y) {
  __shared__ f4 lds[2 * chunk_f4(128)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  WStream<128> ws;
  ws.init(lds, tid);
  const f4* w0 = Wp;
  const f4* w1 = w0 + layer_f4<32, 128>();
  const f4* w2 = w1 + layer_f4<128, 128>();
  const long row0 = ((long)blockIdx.x * 4 + wave) * 32 + (lane & 15);
  float in0[2][8], h[2][32], z[2][32];
  load_features<32>(L, row0, M, lane, in0[0]);
  load_features<32>(L, row0 + 16, M, lane, in0[1]);
  ws.prime<chunk_f4(32)>(w0);
  dense_layer<32, 128, 2, 128>(ws, w0, w1, in0, z, lane, true);
  activate<128, 2, ACT_LEAKY02>(z, h);
  dense_layer<128, 128, 2, 128>(ws, w1, w2, h, z, lane, true);
  activate<128, 2, ACT_LEAKY02>(z, h);
  float o[2][4];
  dense_layer<128, 16, 2, 0>(ws, w2, nullptr, h, o, lane, true);
  const int g = lane >> 4;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const long row = row0 + 16 * t;
    if (row < M) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 4 * g + r;
        if (j < n_out) {
          float v = o[t][r];
          Y[row * n_out + j] = sigmoid_out ? 1.0f / (1.0f + expf(-v)) : v;
        }
      }
    }
  }
}

// ---- 512-wide SDFNetwork-style nets of the CESR stage (training/train_cesr.py:106-110):
//   shadow_net = SDFNetwork(63+128 -> 2, 512 x 8, skip [4], multires 0)   input [PE10(x) | one-hot light-lobe label]
//   normal_net = SDFNetwork(63 -> 3,     512 x 8, skip [4], multires 0)   input PE10(x)
// (model/neus_model.py:312-417 with multires = 0: no internal encoding).  Softplus(beta=100), skip concat /sqrt(2).
// K0P = padded input width (192 / 64), N3P = padded width of lin3 (336 / 464), K4 = N3P + K0P = 528 for both.
// ONEHOT: rows are (point, label) pairs, label = row % n_label; the kernel reads PE features of point row / n_label
// from Xp[n,64] and synthesises the one-hot part in registers (never materialised: it would be 98 KB per point).
// FUSED: X = the points [.,3]: the 63 encoded columns (model/embedder.py:17-38) are computed in the kernel (mlp_engine.h).
template <int K0P, int N3P, bool ONEHOT, bool FUSED = false>
__global__ __launch_bounds__(256, 1) void k_softplus512(const float* __restrict__ X, long M, int n_label,
                                                         const f4* __restrict__ Wp, int n_out, float* __restrict__ Y) {
  constexpr int K4 = N3P + K0P;
  static_assert(K4 == 528, "both CESR nets give a 528-wide skip layer");
  __shared__ f4 lds[2 * chunk_f4(528)];
  __shared__ float pe_scratch[FUSED ? 4 * 16 * 64 : 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  WStream<528> ws;
  ws.init(lds, tid);
  constexpr long LF = layer_f4<512, 512>();
  const f4* w0 = Wp;
  const f4* w1 = w0 + layer_f4<K0P, 512>();
  const f4* w3 = w1 + 2 * LF;
  const f4* w4 = w3 + layer_f4<512, N3P>();
  const f4* w5 = w4 + layer_f4<K4, 512>();
  const f4* w8 = w5 + 3 * LF;
  const long row = ((long)blockIdx.x * 4 + wave) * 16 + (lane & 15);
  const float inv_sqrt2 = 0.70710678118654752440f;
  float x0[1][K0P / 4], h[1][128], z[1][128];
  if constexpr (ONEHOT) {
    const bool ok = row < M;
    const long pt = ok ? row / n_label : 0;
    const int label = ok ? (int)(row % n_label) : -1;
    float enc[16];
    if constexpr (FUSED) load_features_pe10x(X, nullptr, row, M, lane, pe_scratch + wave * 1024, enc, n_label);
    const f4* p = reinterpret_cast<const f4*>(X + (FUSED ? 0 : pt * 64)) + g;
#pragma unroll
    for (int kb = 0; kb < K0P / 16; ++kb) {
      f4 v = f4{0.f, 0.f, 0.f, 0.f};
      if constexpr (FUSED) {
        if (kb < 4) v = f4{enc[kb * 4], enc[kb * 4 + 1], enc[kb * 4 + 2], enc[kb * 4 + 3]};
      } else {
        if (kb < 4 && ok) v = p[kb * 4];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = kb * 16 + 4 * g + r;
        float e = v[r];
        if (k == 63) e = 0.f;                       // column 63 of Xp is padding; the one-hot block starts here
        if (k >= 63 && k - 63 == label) e = 1.f;
        x0[0][kb * 4 + r] = e;
      }
    }
  } else if constexpr (FUSED && K0P == 64) {
    load_features_pe10x(X, nullptr, row, M, lane, pe_scratch + wave * 1024, x0[0]);
  } else {
    load_features<K0P>(X, row, M, lane, x0[0]);
  }
  ws.prime<chunk_f4(K0P)>(w0);
  dense_layer<K0P, 512, 1, 512>(ws, w0, w1, x0, z, lane, true);
  activate<512, 1, ACT_SOFTPLUS100>(z, h);
#pragma unroll 1
  for (int l = 0; l < 2; ++l) {
    dense_layer<512, 512, 1, 512>(ws, w1 + l * LF, w1 + (l + 1) * LF, h, z, lane, true);
    activate<512, 1, ACT_SOFTPLUS100>(z, h);
  }
  {
    float hs[1][K4 / 4];
    {
      float z3[1][N3P / 4];
      dense_layer<512, N3P, 1, K4>(ws, w3, w4, h, z3, lane, true);
#pragma unroll
      for (int i = 0; i < N3P / 4; ++i) hs[0][i] = act_fn<ACT_SOFTPLUS100>(z3[0][i]) * inv_sqrt2;
    }
#pragma unroll
    for (int i = 0; i < K0P / 4; ++i) hs[0][N3P / 4 + i] = x0[0][i] * inv_sqrt2;
    dense_layer<K4, 512, 1, 512>(ws, w4, w5, hs, z, lane, true);
  }
  activate<512, 1, ACT_SOFTPLUS100>(z, h);
#pragma unroll 1
  for (int l = 0; l < 3; ++l) {
    dense_layer<512, 512, 1, 512>(ws, w5 + l * LF, w5 + (l + 1) * LF, h, z, lane, true);
    activate<512, 1, ACT_SOFTPLUS100>(z, h);
  }
  float o[1][4];
  dense_layer<512, 16, 1, 0>(ws, w8, nullptr, h, o, lane, true);
  if (row < M && g == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (r < n_out) Y[row * n_out + r] = o[0][r];
  }
}

// ---- small element-wise pieces of the auto-encoders / indirect-illumination head
// latent = act(raw * (1 - var));  act: 0 sigmoid, 1 softplus(beta=1, threshold 20)   (sg_envmap_material.py:74-99)
// writes lat[M,32]; if lat2 != null also lat2 = lat + noise*noise_scale (smooth_on_latent branch)
__global__ void k_ae_latent(const float* __restrict__ raw, long M, const float* __restrict__ var, int act,
                            const float* __restrict__ noise, float noise_scale, float* __restrict__ lat,
                            float* __restrict__ lat2) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= M * 32) return;
  float v = raw[i];
  if (var) v = v * (1.0f - var[i & 31]);
  float a;
  if (act == 0) {
    a = 1.0f / (1.0f + expf(-v));
  } else if (act == 1) {
    a = v > 20.0f ? v : log1pf(expf(v));
  } else {
    a = v;                      // SparseAE.encode: the pre-activation latent
  }
  lat[i] = a;
  if (lat2) lat2[i] = a + noise[i] * noise_scale;
}

// y = a + s*b over n floats
__global__ void k_axpy(const float* __restrict__ a, const float* __restrict__ b, float s, long n, float* __restrict__ y) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i < n) y[i] = a[i] + s * b[i];
}

// IndirctIllumNetwork head (implicit_differentiable_renderer.py:206-218): raw[M,24,6] -> sgs[M,24,7]
__global__ void k_illum_decode(const float* __restrict__ raw, long n_lobes, float* __restrict__ sgs) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= n_lobes) return;
  const float* r = raw + i * 6;
  const float two_pi = (float)(2.0 * 3.14159265358979323846), pi = (float)3.14159265358979323846;
  float a = 1.0f / (1.0f + expf(-r[0])), b = 1.0f / (1.0f + expf(-r[1]));
  float theta = a * 2.0f * pi, phi = b * pi;
  (void)two_pi;
  float* o = sgs + i * 7;
  o[0] = cosf(theta) * sinf(phi);
  o[1] = sinf(theta) * sinf(phi);
  o[2] = cosf(phi);
  o[3] = (1.0f / (1.0f + expf(-r[2]))) * 30.0f + 0.1f;
  o[4] = fmaxf(r[3], 0.f);
  o[5] = fmaxf(r[4], 0.f);
  o[6] = fmaxf(r[5], 0.f);
}

}  // namespace rb

// =====================================================================================================
// C-ABI
// =====================================================================================================
using namespace rb;

extern "C" {

int rb_abi_version(void) { return RB_ABI_VERSION; }

int rb_range_check(int synchronize, rb_stream_t stream, unsigned* mask_out) {
  RB_REQUIRE(mask_out, "null pointer");
  *mask_out = 0u;
  unsigned* w = rb::range_flags_host();
  RB_REQUIRE(w, "range sentinel: pinned host memory could not be allocated / mapped");
  if (synchronize && hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return rb::fail(__func__, hipGetErrorString(hipGetLastError()));
  unsigned m = 0u;
  for (int i = 0; i < RB_RANGE_WORDS; ++i) {
    volatile unsigned* p = w + i;
    if (*p) {
      m |= 1u << i;
      *p = 0u;
    }
  }
  *mask_out = m;
  return 0;
}
const char* rb_last_error(void) { return rb::err_buf(); }

long rb_packed_layer_floats(int n_pad, int k_pad) { return (long)(n_pad / 16) * (16 + (long)k_pad * 16); }

int rb_pack_layer(const float* W, const float* b, int n_out, int k_in, int n_pad, int k_pad, const int* k_perm,
                  float w_scale, float* out, rb_stream_t stream) {
  RB_REQUIRE(W && out, "null pointer");
  RB_REQUIRE(n_pad % 16 == 0 && k_pad % 16 == 0 && n_pad >= n_out && (k_perm || k_pad >= k_in), "bad padding");
  long total = rb_packed_layer_floats(n_pad, k_pad);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_pack_layer, dim3(blocks), dim3(256), 0, (hipStream_t)stream, W, b, n_out, k_in, n_pad, k_pad,
                     k_perm, w_scale, out);
  return check_launch("k_pack_layer");
}

#ifdef RB_LEGACY
int rb_pack_layer_h3(const float* W, const float* b, int n_out, int k_in, int n_pad, int k_pad, const int* perm,
                     int scale_log2, float* out, rb_stream_t stream) {
  RB_REQUIRE(W && out, "null pointer");
  RB_REQUIRE(n_pad % 16 == 0 && k_pad % 32 == 0 && n_pad >= n_out && (perm || k_pad >= k_in), "bad padding");
  RB_REQUIRE(scale_log2 >= -14 && scale_log2 <= 14, "scale_log2 out of range");
  long total = rb_packed_layer_floats(n_pad, k_pad);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_pack_layer_h3, dim3(blocks), dim3(256), 0, (hipStream_t)stream, W, b, n_out, k_in, n_pad, k_pad,
                     perm, ldexpf(1.0f, scale_log2), out);
  return check_launch("k_pack_layer_h3");
}
#endif  // RB_LEGACY

long rb_packed_layer_x6_floats(int n_pad, int k_pad) { return (long)(n_pad / 16) * (16 + (long)k_pad * 24); }

int rb_pack_layer_x6(const float* W, const float* b, int n_out, int k_in, int n_pad, int k_pad, const int* perm,
                     int scale_log2, float* out, rb_stream_t stream) {
  RB_REQUIRE(W && out, "null pointer");
  RB_REQUIRE(n_pad % 16 == 0 && k_pad % 32 == 0 && n_pad >= n_out && (perm || k_pad >= k_in), "bad padding");
  RB_REQUIRE(scale_log2 >= -14 && scale_log2 <= 14, "scale_log2 out of range");
  long total = rb_packed_layer_x6_floats(n_pad, k_pad);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_pack_layer_x6, dim3(blocks), dim3(256), 0, (hipStream_t)stream, W, b, n_out, k_in, n_pad, k_pad,
                     perm, ldexpf(1.0f, scale_log2), out);
  return check_launch("k_pack_layer_x6");
}

#ifdef RB_LEGACY
int rb_feat_vis(const float* p, const float* d, long M, int rep, float* X, rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(p && d && X, "null pointer");
  RB_REQUIRE(rep >= 1, "rep must be >= 1");
  const long vblocks = (M * 32 + 255) / 256;
  hipLaunchKernelGGL(k_feat_vis, dim3((unsigned)(vblocks < RB_MAX_BLOCKS ? vblocks : RB_MAX_BLOCKS)), dim3(256), 0, (hipStream_t)stream, p, d, M,
                     rep, X);
  return check_launch("k_feat_vis");
}
#endif  // RB_LEGACY

int rb_feat_pe10(const float* x, long M, float scale, const float* extra, int jvp, float* X, rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(x && X, "null pointer");
  RB_REQUIRE(!(jvp && extra), "extra column not supported with jvp rows");
  const long blocks = ((jvp ? 4 * M : M) * 16 + 255) / 256;
  hipLaunchKernelGGL(k_feat_pe10, dim3((unsigned)(blocks < RB_MAX_BLOCKS ? blocks : RB_MAX_BLOCKS)), dim3(256), 0, (hipStream_t)stream, x, M,
                     scale, extra, jvp, X);
  return check_launch("k_feat_pe10");
}

int rb_feat_ipe(const float* x, long M, float var, const float* noise, float noise_scale, float* X,
                rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(x && X, "null pointer");
  hipLaunchKernelGGL(k_feat_ipe, grid1d(M, 128), dim3(128), 0, (hipStream_t)stream, x, M, var, noise, noise_scale, X);
  return check_launch("k_feat_ipe");
}

#ifdef RB_LEGACY
int rb_feat_color(const float* x, float x_scale, const float* view, const float* normal, const float* feat,
                  long feat_stride, float feat_scale, long M, float* X, rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(x && view && normal && feat && X, "null pointer");
  hipLaunchKernelGGL(k_feat_color, dim3((unsigned)(M < RB_MAX_BLOCKS ? M : RB_MAX_BLOCKS)), dim3(256), 0, (hipStream_t)stream, x, x_scale, view, normal, feat,
                     feat_stride, feat_scale, M, X);
  return check_launch("k_feat_color");
}

int rb_feat_color_tail(const float* x, float x_scale, const float* view, const float* normal, long M, float* T, rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(x && view && normal && T, "null pointer");
  hipLaunchKernelGGL(k_feat_color_tail, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, x_scale, view, normal, M, T);
  return check_launch("k_feat_color_tail");
}

int rb_vis_mlp(const float* X, long M, const float* Wp, float* logits, rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(X && Wp && logits, "null pointer");
  hipLaunchKernelGGL(k_vis_mlp<false>, grid1d(M, 128), dim3(256), 0, (hipStream_t)stream, X, M, (const f4*)Wp, logits, nullptr, 1);
  return check_launch("k_vis_mlp");
}
#endif  // RB_LEGACY

int rb_vis_mlp_points(const float* p, const float* d, long M, int rep, const float* Wp, float* logits, rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(p && d && Wp && logits, "null pointer");
  RB_REQUIRE(rep >= 1, "rep must be >= 1");
  hipLaunchKernelGGL(k_vis_mlp<true>, grid1d(M, 128), dim3(256), 0, (hipStream_t)stream, p, M, (const f4*)Wp, logits, d, rep);
  return check_launch("k_vis_mlp<points>");
}

#ifdef RB_LEGACY
int rb_linear_64_256(const float* X, long M, const float* Wp, float* Y, rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(X && Wp && Y, "null pointer");
  hipLaunchKernelGGL(k_linear_64_256<false>, grid1d(M, 128), dim3(256), 0, (hipStream_t)stream, X, M, (const f4*)Wp, Y);
  return check_launch("k_linear_64_256");
}
#endif  // RB_LEGACY

int rb_linear_pe10_256(const float* x, long M, const float* Wp, float* Y, rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(x && Wp && Y, "null pointer");
  hipLaunchKernelGGL(k_linear_64_256<true>, grid1d(M, 128), dim3(256), 0, (hipStream_t)stream, x, M, (const f4*)Wp, Y);
  return check_launch("k_linear_64_256<points>");
}

#ifdef RB_LEGACY
int rb_sdf_mlp(const float* X, long M, const float* Wp, int mode, float out_scale, float grad_scale, float* out0,
               float* grad, rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(X && Wp && out0, "null pointer");
  RB_REQUIRE((mode >= 0 && mode <= 3) || mode == 4 || mode == 6, "mode must be 0..3, or 4 / 6 (= 0 / 2 with library-grade activations)");
  RB_REQUIRE((mode & 3) < 2 || grad, "jvp modes need a gradient output");
  const long MR = (mode & 3) >= 2 ? 4 * M : M;
  dim3 grid = grid1d(MR, 128), block(256);
  hipStream_t s = (hipStream_t)stream;
  const f4* W = (const f4*)Wp;
  switch (mode) {
    case 0: hipLaunchKernelGGL(k_sdf_mlp<0>, grid, block, 0, s, X, MR, W, out_scale, grad_scale, out0, grad, 1.0f); break;
    case 1: hipLaunchKernelGGL(k_sdf_mlp<1>, grid, block, 0, s, X, MR, W, out_scale, grad_scale, out0, grad, 1.0f); break;
    case 2: hipLaunchKernelGGL(k_sdf_mlp<2>, grid, block, 0, s, X, MR, W, out_scale, grad_scale, out0, grad, 1.0f); break;
    case 3: hipLaunchKernelGGL(k_sdf_mlp<3>, grid, block, 0, s, X, MR, W, out_scale, grad_scale, out0, grad, 1.0f); break;
    case 4: hipLaunchKernelGGL((k_sdf_mlp<0, true>), grid, block, 0, s, X, MR, W, out_scale, grad_scale, out0, grad, 1.0f); break;
    default: hipLaunchKernelGGL((k_sdf_mlp<2, true>), grid, block, 0, s, X, MR, W, out_scale, grad_scale, out0, grad, 1.0f); break;
  }
  return check_launch("k_sdf_mlp");
}
#endif  // RB_LEGACY

int rb_sdf_mlp_points(const float* x, long M, float in_scale, const float* Wp, int mode, float out_scale, float grad_scale,
                      float* out0, float* grad, rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(x && Wp && out0, "null pointer");
  RB_REQUIRE((mode >= 0 && mode <= 3) || mode == 4 || mode == 6, "mode must be 0..3, or 4 / 6 (= 0 / 2 with library-grade activations)");
  RB_REQUIRE((mode & 3) < 2 || grad, "jvp modes need a gradient output");
  const long MR = (mode & 3) >= 2 ? 4 * M : M;
  dim3 grid = grid1d(MR, 128), block(256);
  hipStream_t s = (hipStream_t)stream;
  const f4* W = (const f4*)Wp;
  switch (mode) {
    case 0: hipLaunchKernelGGL((k_sdf_mlp<0, false, true>), grid, block, 0, s, x, MR, W, out_scale, grad_scale, out0, grad, in_scale); break;
    case 1: hipLaunchKernelGGL((k_sdf_mlp<1, false, true>), grid, block, 0, s, x, MR, W, out_scale, grad_scale, out0, grad, in_scale); break;
    case 2: hipLaunchKernelGGL((k_sdf_mlp<2, false, true>), grid, block, 0, s, x, MR, W, out_scale, grad_scale, out0, grad, in_scale); break;
    case 3: hipLaunchKernelGGL((k_sdf_mlp<3, false, true>), grid, block, 0, s, x, MR, W, out_scale, grad_scale, out0, grad, in_scale); break;
    case 4: hipLaunchKernelGGL((k_sdf_mlp<0, true, true>), grid, block, 0, s, x, MR, W, out_scale, grad_scale, out0, grad, in_scale); break;
    default: hipLaunchKernelGGL((k_sdf_mlp<2, true, true>), grid, block, 0, s, x, MR, W, out_scale, grad_scale, out0, grad, in_scale); break;
  }
  return check_launch("k_sdf_mlp<points>");
}

#ifdef RB_LEGACY
int rb_color_mlp(const float* X, long M, const float* Wp, float* rgb, rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(X && Wp && rgb, "null pointer");
  hipLaunchKernelGGL(k_color_mlp<false>, grid1d(M, 128), dim3(256), 0, (hipStream_t)stream, X, M, (const f4*)Wp, rgb, nullptr, 0L, 1.0f,
                     nullptr, 1.0f, nullptr, nullptr);
  return check_launch("k_color_mlp");
}
#endif  // RB_LEGACY

int rb_color_mlp_points(const float* feat, long feat_stride, float feat_scale, const float* x, float x_scale, const float* view,
                        const float* normal, long M, const float* Wp, float* rgb, rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(feat && x && view && normal && Wp && rgb, "null pointer");
  hipLaunchKernelGGL(k_color_mlp<true>, grid1d(M, 128), dim3(256), 0, (hipStream_t)stream, nullptr, M, (const f4*)Wp, rgb, feat, feat_stride,
                     feat_scale, x, x_scale, view, normal);
  return check_launch("k_color_mlp<points>");
}

#ifdef RB_LEGACY
int rb_illum_mlp(const float* X, long M, const float* Wp, float* raw, rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(X && Wp && raw, "null pointer");
  hipLaunchKernelGGL(k_wide_mlp<false>, grid1d(M, 64), dim3(256), 0, (hipStream_t)stream, X, M, (const f4*)Wp, raw);
  return check_launch("k_wide_mlp<illum>");
}
#endif  // RB_LEGACY

int rb_wide_mlp_points(const float* x, const float* extra, long M, const float* Wp, int encoder, float* Y, rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(x && Wp && Y, "null pointer");
  if (encoder) {
    hipLaunchKernelGGL((k_wide_mlp<true, true>), grid1d(M, 64), dim3(256), 0, (hipStream_t)stream, x, M, (const f4*)Wp, Y, extra);
  } else {
    hipLaunchKernelGGL((k_wide_mlp<false, true>), grid1d(M, 64), dim3(256), 0, (hipStream_t)stream, x, M, (const f4*)Wp, Y, extra);
  }
  return check_launch("k_wide_mlp<points>");
}

int rb_cesr_net_points(const float* x, long M, int kind, int n_label, const float* Wp, float* Y, rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(x && Wp && Y, "null pointer");
  hipStream_t s = (hipStream_t)stream;
  dim3 grid = grid1d(M, 64), block(256);
  const f4* W = (const f4*)Wp;
  switch (kind) {
    case 0: hipLaunchKernelGGL((k_softplus512<64, 464, false, true>), grid, block, 0, s, x, M, 1, W, 3, Y); break;
    case 2:
      RB_REQUIRE(n_label >= 1 && n_label <= 128, "n_label must be 1..128");
      hipLaunchKernelGGL((k_softplus512<192, 336, true, true>), grid, block, 0, s, x, M, n_label, W, 2, Y);
      break;
    default: return rb::fail("rb_cesr_net_points", "kind: 0 normal_net on PE10(x), 2 shadow_net on (point, one-hot label) rows");
  }
  return check_launch("k_softplus512<points>");
}

int rb_cesr_net(const float* X, long M, int kind, int n_label, const float* Wp, float* Y, rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(X && Wp && Y, "null pointer");
  hipStream_t s = (hipStream_t)stream;
  dim3 grid = grid1d(M, 64), block(256);
  const f4* W = (const f4*)Wp;
  switch (kind) {
    case 0: hipLaunchKernelGGL((k_softplus512<64, 464, false>), grid, block, 0, s, X, M, 1, W, 3, Y); break;
    case 1: hipLaunchKernelGGL((k_softplus512<192, 336, false>), grid, block, 0, s, X, M, 1, W, 2, Y); break;
    case 2:
      RB_REQUIRE(n_label >= 1 && n_label <= 128, "n_label must be 1..128");
      hipLaunchKernelGGL((k_softplus512<192, 336, true>), grid, block, 0, s, X, M, n_label, W, 2, Y);
      break;
    default: return rb::fail("rb_cesr_net", "kind: 0 normal_net, 1 shadow_net (dense rows), 2 shadow_net (point x one-hot label)");
  }
  return check_launch("k_softplus512");
}

int rb_ae_encode(const float* X, long M, const float* Wp, float* raw_latent, rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(X && Wp && raw_latent, "null pointer");
  hipLaunchKernelGGL(k_wide_mlp<true>, grid1d(M, 64), dim3(256), 0, (hipStream_t)stream, X, M, (const f4*)Wp,
                     raw_latent);
  return check_launch("k_wide_mlp<enc>");
}

int rb_ae_latent(const float* raw, long M, const float* var, int act, const float* noise, float noise_scale, float* lat,
                 float* lat2, rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(raw && lat, "null pointer");
  RB_REQUIRE(!lat2 || noise, "lat2 needs noise");
  RB_REQUIRE(act >= 0 && act <= 2, "act: 0 sigmoid, 1 softplus, 2 none");
  hipLaunchKernelGGL(k_ae_latent, grid1d(M * 32, 256), dim3(256), 0, (hipStream_t)stream, raw, M, var, act, noise,
                     noise_scale, lat, lat2);
  return check_launch("k_ae_latent");
}

int rb_ae_decode(const float* lat, long M, const float* Wp, int n_out, int sigmoid_out, float* Y, rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(lat && Wp && Y, "null pointer");
  RB_REQUIRE(n_out >= 1 && n_out <= 16, "n_out must be 1..16");
  hipLaunchKernelGGL(k_ae_decode, grid1d(M, 128), dim3(256), 0, (hipStream_t)stream, lat, M, (const f4*)Wp, n_out,
                     sigmoid_out, Y);
  return check_launch("k_ae_decode");
}

int rb_axpy(const float* a, const float* b, float s, long n, float* y, rb_stream_t stream) {
  if (n <= 0) return 0;
  RB_REQUIRE(a && b && y, "null pointer");
  hipLaunchKernelGGL(k_axpy, grid1d(n, 256), dim3(256), 0, (hipStream_t)stream, a, b, s, n, y);
  return check_launch("k_axpy");
}

int rb_illum_decode(const float* raw, long M, float* sgs, rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(raw && sgs, "null pointer");
  hipLaunchKernelGGL(k_illum_decode, grid1d(M * 24, 256), dim3(256), 0, (hipStream_t)stream, raw, M * 24, sgs);
  return check_launch("k_illum_decode");
}

}  // extern "C"
