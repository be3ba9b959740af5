// Light-SG ("diffuse") visibility: get_diffuse_visibility, model/sg_render.py:111-195 -- 90 % of the reference's
// PBR forward time.  Per surface point: L*nsamp (128*32 = 4096) cone-sampled directions shared by all points of a
// chunk, cull n.d <= 1e-6, visibility MLP on the survivors, softmax[...,1], SG-weighted mean per lobe.
//
// MI355X design: the [n,4096,*] tensors of the reference are never materialised.
//   * the first MLP layer is split:  relu(W0p.PE(p) + b0  +  W0d.PE(d))  -> a per-point row A[n,256] and a
//     per-direction table B[C*4096,256] (both produced by rb_linear_64_256), so layer 0 costs two loads + add;
//   * one workgroup per point: cull + wave-ballot compaction of the surviving directions into LDS, then rounds of
//     128 survivors (4 waves x 2 tiles x 16) through the three 256x256 hidden layers on the fp32 MFMA engine
//     (mlp_engine.h; activations stay in registers), VALU 256->2 head, softmax, result scattered to an LDS
//     visibility table; finally the SG-weighted mean per lobe (deterministic order).
#include "../../include/robir_hip.h"
#include "common.h"
#include "mlp_engine.h"

namespace rb {

#define RB_TINY 1e-6f

__device__ __forceinline__ void unit_eps(float v[3]) {  // norm_axis (sg_render.py:107-108)
  float n = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]) + RB_TINY;
  v[0] /= n;
  v[1] /= n;
  v[2] /= n;
}

// Directions in a cone around `axis` (sg_render.py:123-146 / 213-240): z-cross tangent frame, theta = 2*pi*u1,
// phi = u2 * phi_range.
__device__ __forceinline__ void cone_dir(const float axis[3], float u_theta, float u_phi, float phi_range, float d[3]) {
  float U[3] = {0.f * axis[2] - 1.f * axis[1], 1.f * axis[0] - 0.f * axis[2], 0.f * axis[1] - 0.f * axis[0]};
  unit_eps(U);
  float V[3] = {axis[1] * U[2] - axis[2] * U[1], axis[2] * U[0] - axis[0] * U[2], axis[0] * U[1] - axis[1] * U[0]};
  unit_eps(V);
  const float pi = (float)3.14159265358979323846;
  float th = u_theta * 2.f * pi;
  float ph = u_phi * phi_range;
  float ct = cosf(th), st = sinf(th), cp = cosf(ph), sp = sinf(ph);
#pragma unroll
  for (int c = 0; c < 3; ++c) d[c] = U[c] * ct * sp + V[c] * st * sp + axis[c] * cp;
}

// One thread per (chunk, lobe): sample directions, SG weights and their per-lobe sum.
// lgt[L,7] raw light SGs (first row's light is used for every point: sg_render.py:388-390).
// direct = 0: lgt are RAW light SGs as render_with_sg receives them (it normalises the lobe and takes |lambda| before the
// call, sg_render.py:364-366, and get_diffuse_visibility normalises again, :126);  direct = 1: lgt[:, :3] / lgt[:, 3] are
// the lgtSGLobes / lgtSGLambdas arguments of a direct get_diffuse_visibility call (one normalisation, lambda as given).
__global__ void k_dvis_dirs(const float* __restrict__ lgt, int L, int nsamp, int C, int direct,
                            const float* __restrict__ u_theta,
                            const float* __restrict__ u_phi, float thr, float* __restrict__ dirs,
                            float* __restrict__ wdir, float* __restrict__ wsum) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= C * L) return;
  const int l = i % L;
  // batch-global minimum sharpness over the lobes (sg_render.py:131-133)
  float mn = 3.0e38f;
  for (int k = 0; k < L; ++k) mn = fminf(mn, fmaxf(direct ? lgt[k * 7 + 3] : fabsf(lgt[k * 7 + 3]), 1e-4f));
  const float rng = fminf(mn, thr);
  float a[3] = {lgt[l * 7], lgt[l * 7 + 1], lgt[l * 7 + 2]};
  if (!direct) unit_eps(a);  // render_with_sg normalisation (sg_render.py:364)
  unit_eps(a);               // norm_axis inside get_diffuse_visibility (sg_render.py:126)
  const float lam = direct ? lgt[l * 7 + 3] : fabsf(lgt[l * 7 + 3]);
  const float sharp = fmaxf(lam, 1e-4f);
  const float phi_range = acosf((-0.95f * rng) / sharp + 1.f);
  float s = 0.f;
  for (int k = 0; k < nsamp; ++k) {
    const long j = (long)i * nsamp + k;
    float d[3];
    cone_dir(a, u_theta[j], u_phi[j], phi_range, d);
    dirs[3 * j] = d[0];
    dirs[3 * j + 1] = d[1];
    dirs[3 * j + 2] = d[2];
    const float w = expf(lam * ((d[0] * a[0] + d[1] * a[1] + d[2] * a[2]) - 1.f));
    wdir[j] = w;
    s += w;
  }
  wsum[i] = s + RB_TINY;
}

constexpr int DV_MAX_DIRS = 4096;

template <bool H3, int CH, int NT, bool DMA = false>
__global__ __launch_bounds__(256, NT == 1 ? 2 : 1) void k_dvis_fused(
    const float* __restrict__ normals, const int* __restrict__ cid, long n, const float* __restrict__ A,
    const float* __restrict__ Bd, const float* __restrict__ dirs, const float* __restrict__ wdir,
    const float* __restrict__ wsum, const f4* __restrict__ Whid, const float* __restrict__ wlast,
    const float* __restrict__ blast, int L, int nsamp, int argmax_vis, float w_unscale, float* __restrict__ vis_out,
    unsigned long long* __restrict__ eval_count, unsigned* __restrict__ range_word) {
  unsigned sat = 0u;   // range sentinel of the split-precision path (operands are ReLU outputs: >= 0)
  __shared__ f4 lds_w[(H3 ? 3 : 2) * chunk_f4(256)];
  __shared__ float vis_tab[DV_MAX_DIRS];
  __shared__ unsigned short idx_list[DV_MAX_DIRS];
  __shared__ f4 a_row[64];
  __shared__ f4 w_last[128];  // [2][256]
  __shared__ int s_count;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  const long p = blockIdx.x;
  const int LS = L * nsamp;
  const long dbase = (cid ? (long)cid[p] : 0L) * LS;
  if (tid == 0) s_count = 0;
  if (tid < 64) a_row[tid] = reinterpret_cast<const f4*>(A + p * 256)[tid];
  if (tid < 128) w_last[tid] = reinterpret_cast<const f4*>(wlast)[tid];
  for (int j = tid; j < LS; j += 256) vis_tab[j] = 0.f;
  __syncthreads();
  // ---- cull + compaction (order inside the list is irrelevant: results are scattered by direction index)
  const float nx = normals[3 * p], ny = normals[3 * p + 1], nz = normals[3 * p + 2];
  for (int j0 = 0; j0 < LS; j0 += 256) {
    const int j = j0 + tid;
    bool front = false;
    if (j < LS) {
      const float* d = dirs + 3 * (dbase + j);
      const float c = nx * d[0] + ny * d[1] + nz * d[2];  // sum(n*d): separate mul/add (-ffp-contract=off)
      front = c > RB_TINY;
    }
    const unsigned long long m = __ballot(front);
    int base = 0;
    if (lane == 0 && m) base = atomicAdd(&s_count, __popcll(m));
    base = __shfl(base, 0);
    if (front) idx_list[base + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)j;
  }
  __syncthreads();
  const int S = s_count;
  if (tid == 0 && eval_count) atomicAdd(eval_count, (unsigned long long)S);

  WStream<256> ws;
  H3Ring<NT, 48, DMA> ring;
  constexpr long LF = (long)16 * chunk_f4(256);
  const float b0 = blast[0], b1 = blast[1];
  constexpr int RS = 64 * NT;   // samples per workgroup round
  const int rounds = (S + RS - 1) / RS;
  if constexpr (H3) {
    if (rounds > 0) ring.start(lds_w, Whid, tid);
  } else {
    ws.init(lds_w, tid);
    if (rounds > 0) ws.prime<chunk_f4(256)>(Whid);
  }
  for (int rd = 0; rd < rounds; ++rd) {
    float z[NT][64];    // pre-activation of the last hidden layer (fp32 path: every layer)
    int jj[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int si = rd * RS + wave * (16 * NT) + t * 16 + (lane & 15);
      jj[t] = si < S ? (int)idx_list[si] : -1;
    }
    if constexpr (!H3) {
      float h[NT][64];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int j = jj[t] < 0 ? 0 : jj[t];
        const f4* brow = reinterpret_cast<const f4*>(Bd + (dbase + j) * 256) + g;
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) {
          const f4 bv = brow[kb * 4];
          const f4 av = a_row[kb * 4 + g];
#pragma unroll
          for (int r = 0; r < 4; ++r) h[t][kb * 4 + r] = fmaxf(av[r] + bv[r], 0.f);
        }
      }
#pragma unroll 1
      for (int l = 0; l < 3; ++l) {
        const f4* wl = Whid + l * LF;
        const f4* wn = (l < 2) ? wl + LF : Whid;  // wrap: the next round starts again at hidden layer 0
        dense_layer<256, 256, NT, 256>(ws, wl, wn, h, z, lane, true);
        if (l < 2) activate<256, NT, ACT_RELU>(z, h);
      }
    } else {
      // split-precision hidden stack: activations travel as packed hi/lo halves, accumulators are fp32
      unsigned xh[NT][8][4], xl[NT][8][4];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int j = jj[t] < 0 ? 0 : jj[t];
        const f4* brow = reinterpret_cast<const f4*>(Bd + (dbase + j) * 256) + g;
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) {
          const f4 bv = brow[kb * 4];
          const f4 av = a_row[kb * 4 + g];
#pragma unroll
          for (int q = 0; q < 2; ++q)
            split_pair(fmaxf(av[2 * q] + bv[2 * q], 0.f), fmaxf(av[2 * q + 1] + bv[2 * q + 1], 0.f),
                       xh[t][kb / 2][(kb & 1) * 2 + q], xl[t][kb / 2][(kb & 1) * 2 + q]);
        }
      }
#pragma unroll 1
      for (int l = 0; l < 3; ++l) {
        if (l > 0) relu_split<256, NT>(z, w_unscale, xh, xl);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int kb = 0; kb < 8; ++kb)
#pragma unroll
            for (int q = 0; q < 4; ++q) sat = sat_acc_nonneg(sat, xh[t][kb][q]);
#pragma unroll
        for (int jb = 0; jb < 16; ++jb) {
          f4 res[NT];
          ring.template chunk<CH>(xh, xl, res);
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) z[t][jb * 4 + r] = res[t][r];
        }
      }
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 64; ++i) z[t][i] = z[t][i] * w_unscale;
    }
    // 256 -> 2 head on the VALU: each lane owns 64 of the 256 activations of its sample
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float l0 = 0.f, l1 = 0.f;
#pragma unroll
      for (int kb = 0; kb < 16; ++kb) {
        const f4 wa = w_last[kb * 4 + g];
        const f4 wb = w_last[64 + kb * 4 + g];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float hv = fmaxf(z[t][kb * 4 + r], 0.f);
          l0 += hv * wa[r];
          l1 += hv * wb[r];
        }
      }
      l0 += __shfl_xor(l0, 16);
      l0 += __shfl_xor(l0, 32);
      l1 += __shfl_xor(l1, 16);
      l1 += __shfl_xor(l1, 32);
      l0 += b0;
      l1 += b1;
      if (g == 0 && jj[t] >= 0) {
        float v;
        if (argmax_vis) {
          v = l1 > l0 ? 1.f : 0.f;
        } else {
          const float mx = fmaxf(l0, l1);
          const float e0 = expf(l0 - mx), e1 = expf(l1 - mx);
          v = e1 / (e0 + e1);
        }
        vis_tab[jj[t]] = v;
      }
    }
  }
  if constexpr (H3) range_report<true>(sat, range_word);
  // drain the weight ring (the LDS-DMA variant still has chunks in flight that target this workgroup's LDS)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid < L) {
    const float* w = wdir + dbase + (long)tid * nsamp;
    float acc = 0.f;
    for (int k = 0; k < nsamp; ++k) acc += vis_tab[tid * nsamp + k] * w[k];
    vis_out[p * L + tid] = acc / wsum[(cid ? cid[p] : 0) * L + tid];
  }
}

}  // namespace rb

using namespace rb;

extern "C" {

int rb_dvis_dirs(const float* lgt, int L, int nsamp, int C, int direct, const float* u_theta, const float* u_phi, float thr,
                 float* dirs, float* wdir, float* wsum, rb_stream_t stream) {
  RB_REQUIRE(lgt && u_theta && u_phi && dirs && wdir && wsum, "null pointer");
  RB_REQUIRE(L > 0 && nsamp > 0 && C > 0, "bad sizes");
  hipLaunchKernelGGL(k_dvis_dirs, grid1d((long)C * L, 64), dim3(64), 0, (hipStream_t)stream, lgt, L, nsamp, C, direct, u_theta,
                     u_phi, thr, dirs, wdir, wsum);
  return check_launch("k_dvis_dirs");
}

int rb_dvis_fused(const float* normals, const int* chunk_id, long n, const float* A, const float* Bd, const float* dirs,
                  const float* wdir, const float* wsum, const float* Whid, const float* wlast, const float* blast, int L,
                  int nsamp, int argmax_vis, int precision, int scale_log2, float* vis_out,
                  unsigned long long* eval_count, rb_stream_t stream) {
  if (n <= 0) return 0;
  RB_REQUIRE(precision == 0 || precision == 5, "precision: 0 = fp32 MFMA; 5 = f16x3 split (first-generation kernel)");
  RB_REQUIRE(normals && A && Bd && dirs && wdir && wsum && Whid && wlast && blast && vis_out, "null pointer");
  RB_REQUIRE(n <= RB_MAX_BLOCKS, "too many points for one launch (one workgroup each)");
  RB_REQUIRE(L > 0 && L <= 256 && nsamp > 0 && (long)L * nsamp <= DV_MAX_DIRS, "need L <= 256 and L*nsamp <= 4096");
  if (precision == 0) {
    hipLaunchKernelGGL((k_dvis_fused<false, 1, 2>), dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, normals, chunk_id, n, A,
                       Bd, dirs, wdir, wsum, (const f4*)Whid, wlast, blast, L, nsamp, argmax_vis, 1.0f, vis_out, eval_count,
                       (unsigned*)nullptr);
  } else {
#ifndef RB_LEGACY
    return rb::fail(__func__, "precision 5 (first-generation split-precision kernel) is built into librobir_hip_legacy.so only (make legacy)");
#else
#define RB_LAUNCH_H3(CH, NT, ...)                                                                                     \
  hipLaunchKernelGGL((k_dvis_fused<true, CH, NT, ##__VA_ARGS__>), dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, normals, chunk_id, \
                     n, A, Bd, dirs, wdir, wsum, (const f4*)Whid, wlast, blast, L, nsamp, argmax_vis,                  \
                     ldexpf(1.0f, -scale_log2), vis_out, eval_count,                                                   \
                     range_flags() ? range_flags() + RB_RANGE_DVIS : nullptr)
    RB_LAUNCH_H3(2, 1, true);   // one tile per wave, two workgroups per CU, weights staged by LDS-DMA (global_load_lds)
#undef RB_LAUNCH_H3
#endif
  }
  return check_launch("k_dvis_fused");
}

}  // extern "C"
