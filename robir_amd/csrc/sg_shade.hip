// Spherical-Gaussian shading: render_with_sg / lambda_trick / hemisphere_int (model/sg_render.py:62-104,343-565)
// and the BRDF-lobe ("specular") visibility sampling get_specular_visibility (model/sg_render.py:198-301).
// One wavefront per surface point, lanes over light lobes (128 direct / 24 indirect), wave-shuffle lobe sum.
// Plain fp32 with accurate expf/sqrtf/exp2f; compiled with -ffp-contract=off so products and sums round like the
// reference's separate tensor ops.
#include "../../include/robir_hip.h"
#include "common.h"

namespace rb {

#define RB_TINY 1e-6f
#define RB_PI_F ((float)3.14159265358979323846)
#define MU_COS 32.7080f
#define LAMBDA_COS 0.0315f
#define ALPHA_COS 31.7003f

struct V3 {
  float x, y, z;
};
__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float norm3(V3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
__device__ __forceinline__ V3 unit_eps3(V3 a) {  // norm_axis (sg_render.py:107-108)
  float n = norm3(a) + RB_TINY;
  return v3(a.x / n, a.y / n, a.z / n);
}
__device__ __forceinline__ V3 cross3(V3 a, V3 b) {
  return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

// hemisphere_int (sg_render.py:62-81); both branches evaluated and blended like the reference
__device__ __forceinline__ float hemi_int(float lam, float cb) {
  lam = lam + RB_TINY;
  const float il = 1.f / lam;
  const float t = sqrtf(lam) * (1.6988f + 10.8438f * il) / (1.f + 6.2201f * il + 10.2415f * il * il);
  const float ea = expf(-t);
  const float mask = cb >= 0.f ? 1.f : 0.f;
  const float eb = expf(-t * fmaxf(cb, 0.f));
  const float s1 = (1.f - ea * eb) / (1.f - ea + eb - ea * eb);
  const float b = expf(t * fminf(cb, 0.f));
  const float s2 = (b - ea) / ((1.f - ea) * (b + 1.f));
  const float s = mask * s1 + (1.f - mask) * s2;
  const float two_pi = 2.f * RB_PI_F;
  const float a_b = two_pi / lam * (expf(-lam) - expf(-2.f * lam));
  const float a_u = two_pi / lam * (1.f - expf(-lam));
  return a_b * (1.f - s) + a_u * s;
}

// lambda_trick (sg_render.py:84-104): SG1 (lam1 << lam2) x SG2; mu handled by the caller (factor returned)
__device__ __forceinline__ void sg_product(V3 lobe1, float lam1, V3 lobe2, float lam2, V3& lobe3, float& lam3,
                                           float& mu_factor) {
  const float ratio = lam1 / lam2;
  lobe1 = unit_eps3(lobe1);
  lobe2 = unit_eps3(lobe2);
  const float d = dot3(lobe1, lobe2);
  float tmp = sqrtf(ratio * ratio + 1.f + 2.f * ratio * d);
  tmp = fminf(tmp, ratio + 1.f);
  lam3 = lam2 * tmp;
  const float a = ratio / tmp, b = 1.f / tmp;
  lobe3 = v3(a * lobe1.x + b * lobe2.x, a * lobe1.y + b * lobe2.y, a * lobe1.z + b * lobe2.z);
  mu_factor = expf(lam2 * (tmp - ratio - 1.f));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Point-only part of the specular BRDF SG (sg_render.py:414-458): warped lobe/lambda and the 3-channel amplitude.
struct SpecLobe {
  V3 w_lobe;
  float w_lam;
  float w_mu[3];
};
__device__ __forceinline__ SpecLobe spec_lobe(V3 n, V3 v, float rough, float f0, const float* albedo,
                                              const float* metallic) {
  SpecLobe s;
  const float r4 = 2.f / (rough * rough * rough * rough);
  const float b_mu = r4 / RB_PI_F;
  const float vdl = fmaxf(dot3(n, v), 0.f);
  V3 wl = v3(2.f * vdl * n.x - v.x, 2.f * vdl * n.y - v.y, 2.f * vdl * n.z - v.z);
  const float wn = norm3(wl) + RB_TINY;
  wl = v3(wl.x / wn, wl.y / wn, wl.z / wn);
  s.w_lobe = wl;
  s.w_lam = r4 / (4.f * vdl + RB_TINY);
  V3 h = v3(wl.x + v.x, wl.y + v.y, wl.z + v.z);
  const float hn = norm3(h) + RB_TINY;
  h = v3(h.x / hn, h.y / hn, h.z / hn);
  const float vdh = fmaxf(dot3(v, h), 0.f);
  const float fw = exp2f(-(5.55473f * vdh + 6.8316f) * vdh);
  const float d1 = fmaxf(dot3(wl, n), 0.f);
  const float d2 = fmaxf(dot3(v, n), 0.f);
  const float k = (rough + 1.f) * (rough + 1.f) / 8.f;
  const float G = (d1 / (d1 * (1.f - k) + k + RB_TINY)) * (d2 / (d2 * (1.f - k) + k + RB_TINY));
  const float den = 4.f * d1 * d2 + RB_TINY;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float sc = f0;
    if (metallic) sc = (1.f - metallic[0]) * f0 + albedo[c] * metallic[0];
    const float Fr = sc + (1.f - sc) * fw;
    s.w_mu[c] = b_mu * (Fr * G / den);
  }
  return s;
}

// ---------------------------------------------------------------------------------------------------------
// Specular visibility sampling.
// ---------------------------------------------------------------------------------------------------------
__global__ void k_fill_u32(unsigned* p, long n, unsigned v) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// per point: clipped warped-BRDF sharpness, and its per-chunk minimum (batch-global min of sg_render.py:222)
__global__ void k_spec_sharp(const float* __restrict__ normal, const float* __restrict__ view,
                             const float* __restrict__ rough, const int* __restrict__ cid, long n,
                             float* __restrict__ sharp, unsigned* __restrict__ chunk_min) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= n) return;
  V3 nn = v3(normal[3 * i], normal[3 * i + 1], normal[3 * i + 2]);
  V3 vv = v3(view[3 * i], view[3 * i + 1], view[3 * i + 2]);
  const float r = rough[i];
  const float r4 = 2.f / (r * r * r * r);
  const float vdl = fmaxf(dot3(nn, vv), 0.f);
  const float w_lam = r4 / (4.f * vdl + RB_TINY);
  const float s = fminf(fmaxf(w_lam, 0.1f), 50.f);
  sharp[i] = s;
  atomicMin(chunk_min + (cid ? cid[i] : 0), __float_as_uint(s));  // s >= 0.1 > 0: uint order == float order
}

// per (point, sample): direction in the reflection cone, front-facing flag and SG weight
__global__ void k_spec_dirs(const float* __restrict__ normal, const float* __restrict__ view,
                            const float* __restrict__ sharp, const int* __restrict__ cid,
                            const unsigned* __restrict__ chunk_min, const float* __restrict__ u_theta,
                            const float* __restrict__ u_phi, long n, int nsamp, float* __restrict__ dirs,
                            float* __restrict__ wts, unsigned char* __restrict__ front) {
  long j = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (j >= n * nsamp) return;
  const long i = j / nsamp;
  V3 nn = v3(normal[3 * i], normal[3 * i + 1], normal[3 * i + 2]);
  V3 vv = v3(view[3 * i], view[3 * i + 1], view[3 * i + 2]);
  const float ndv = fmaxf(dot3(nn, vv), 0.f);
  V3 refl = v3(-vv.x + 2.f * ndv * nn.x, -vv.y + 2.f * ndv * nn.y, -vv.z + 2.f * ndv * nn.z);
  // warped BRDF lobe = normalised reflection of the view about the normal (sg_render.py:426-427)
  V3 wl = v3(2.f * ndv * nn.x - vv.x, 2.f * ndv * nn.y - vv.y, 2.f * ndv * nn.z - vv.z);
  const float wn = norm3(wl) + RB_TINY;
  wl = v3(wl.x / wn, wl.y / wn, wl.z / wn);
  V3 U = unit_eps3(cross3(v3(0.f, 0.f, 1.f), refl));
  V3 V = unit_eps3(cross3(refl, U));
  const float s = sharp[i];
  const float rng = fminf(__uint_as_float(chunk_min[cid ? cid[i] : 0]), 1.f);
  const float phi_range = acosf((-0.95f * rng) / s + 1.f);
  const float th = u_theta[j] * 2.f * RB_PI_F;
  const float ph = u_phi[j] * phi_range;
  const float ct = cosf(th), st = sinf(th), cp = cosf(ph), sp = sinf(ph);
  V3 d = v3(U.x * ct * sp + V.x * st * sp + refl.x * cp, U.y * ct * sp + V.y * st * sp + refl.y * cp,
            U.z * ct * sp + V.z * st * sp + refl.z * cp);
  dirs[3 * j] = d.x;
  dirs[3 * j + 1] = d.y;
  dirs[3 * j + 2] = d.z;
  front[j] = dot3(nn, d) > RB_TINY ? 1 : 0;
  wts[j] = expf(s * (dot3(d, wl) - 1.f));
}

// get_specular_visibility with the CALLER'S lobes (the reference's own signature, model/sg_render.py:198-223: light_dirs = the passed
// lgtSGLobes [n,3] used as they are, sharpness = clip(passed lgtSGLambdas [n], 0.1, 50), batch-global minimum): two kernels like the
// roughness form above -- the first records the clipped sharpness and its per-chunk minimum, the second samples the cone.
__global__ void k_spec_sharp_lobes(const float* __restrict__ lambdas, const int* __restrict__ cid, long n, float* __restrict__ sharp,
                                   unsigned* __restrict__ chunk_min) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float s = fminf(fmaxf(lambdas[i], 0.1f), 50.f);      // NaN -> 0.1 under fmaxf/fminf; torch.clip keeps NaN (a NaN lambda is a caller error)
  sharp[i] = s;
  atomicMin(chunk_min + (cid ? cid[i] : 0), __float_as_uint(s));
}

__global__ void k_spec_dirs_lobes(const float* __restrict__ normal, const float* __restrict__ view, const float* __restrict__ lobes,
                                  const float* __restrict__ sharp, const int* __restrict__ cid,
                                  const unsigned* __restrict__ chunk_min, const float* __restrict__ u_theta,
                                  const float* __restrict__ u_phi, long n, int nsamp, float* __restrict__ dirs,
                                  float* __restrict__ wts, unsigned char* __restrict__ front) {
  long j = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (j >= n * nsamp) return;
  const long i = j / nsamp;
  V3 nn = v3(normal[3 * i], normal[3 * i + 1], normal[3 * i + 2]);
  V3 vv = v3(view[3 * i], view[3 * i + 1], view[3 * i + 2]);
  V3 wl = v3(lobes[3 * i], lobes[3 * i + 1], lobes[3 * i + 2]);
  const float ndv = fmaxf(dot3(nn, vv), 0.f);
  V3 refl = v3(-vv.x + 2.f * ndv * nn.x, -vv.y + 2.f * ndv * nn.y, -vv.z + 2.f * ndv * nn.z);
  V3 U = unit_eps3(cross3(v3(0.f, 0.f, 1.f), refl));
  V3 V = unit_eps3(cross3(refl, U));
  const float s = sharp[i];
  const float rng = fminf(__uint_as_float(chunk_min[cid ? cid[i] : 0]), 1.f);
  const float phi_range = acosf((-0.95f * rng) / s + 1.f);
  const float th = u_theta[j] * 2.f * RB_PI_F;
  const float ph = u_phi[j] * phi_range;
  const float ct = cosf(th), st = sinf(th), cp = cosf(ph), sp = sinf(ph);
  V3 d = v3(U.x * ct * sp + V.x * st * sp + refl.x * cp, U.y * ct * sp + V.y * st * sp + refl.y * cp,
            U.z * ct * sp + V.z * st * sp + refl.z * cp);
  dirs[3 * j] = d.x;
  dirs[3 * j + 1] = d.y;
  dirs[3 * j + 2] = d.z;
  front[j] = dot3(nn, d) > RB_TINY ? 1 : 0;
  wts[j] = expf(s * (dot3(d, wl) - 1.f));
}

// per point: weighted mean of the sampled visibilities (sg_render.py:269-294)
__global__ void k_spec_reduce(const float* __restrict__ logits, const unsigned char* __restrict__ front,
                              const float* __restrict__ wts, long n, int nsamp, int inv, int argmax_vis, int testing,
                              float* __restrict__ bvis) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= n) return;
  float wsum = 0.f;
  for (int k = 0; k < nsamp; ++k) wsum += wts[i * nsamp + k];
  const bool fix_inf = testing && isinf(wsum);
  float num = 0.f, den = 0.f;
  for (int k = 0; k < nsamp; ++k) {
    const long j = i * nsamp + k;
    float w = wts[j];
    if (fix_inf) w = isinf(w) ? 1.f : 0.f;
    float v = 0.f;
    if (front[j]) {
      const float l0 = logits[2 * j], l1 = logits[2 * j + 1];
      if (argmax_vis) {
        // argmax -> first maximum; argmin -> first minimum
        v = inv ? (l1 < l0 ? 1.f : 0.f) : (l1 > l0 ? 1.f : 0.f);
      } else {
        const float mx = fmaxf(l0, l1);
        const float e0 = expf(l0 - mx), e1 = expf(l1 - mx);
        v = (inv ? e0 : e1) / (e0 + e1);
      }
    }
    num += v * w;
    den += w;
  }
  bvis[i] = num / (den + RB_TINY);
}

// ---------------------------------------------------------------------------------------------------------
// SG shading: one wave per point.
// lgt: [M,7] shared or [n,M,7] per point; light_vis: [n,M] or null (comp_vis=False);
// indir_integral: [n,3] or null (replaces the diffuse term, sg_render.py:532-536).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sg_shade(const float* __restrict__ normal, const float* __restrict__ view,
                                                   const float* __restrict__ lgt, int per_point_lgt, int M,
                                                   const float* __restrict__ f0_dev,
                                                   const float* __restrict__ rough, const float* __restrict__ albedo,
                                                   const float* __restrict__ metallic,
                                                   const float* __restrict__ light_vis, const float* __restrict__ bvis,
                                                   const float* __restrict__ indir_integral, int lin_diff, long n,
                                                   float* __restrict__ out_rgb, float* __restrict__ out_spec,
                                                   float* __restrict__ out_diff, float* __restrict__ out_shadow) {
  const int lane = threadIdx.x & 63;
  const long p = blockIdx.x * 4L + (threadIdx.x >> 6);
  if (p >= n) return;
  const V3 nn = v3(normal[3 * p], normal[3 * p + 1], normal[3 * p + 2]);
  const V3 vv = v3(view[3 * p], view[3 * p + 1], view[3 * p + 2]);
  const float alb[3] = {albedo[3 * p], albedo[3 * p + 1], albedo[3 * p + 2]};
  const float f0 = f0_dev[0];      // scalar Fresnel F0 read on the device: no host copy of the parameter
  const SpecLobe sl = spec_lobe(nn, vv, rough[p], f0, alb, metallic ? metallic + p : nullptr);
  const float bv = bvis[p];
  const float* L = lgt + (per_point_lgt ? p * (long)M * 7 : 0L);
  float spec[3] = {0.f, 0.f, 0.f}, diff[3] = {0.f, 0.f, 0.f}, sh_num[3] = {0.f, 0.f, 0.f}, sh_den[3] = {0.f, 0.f, 0.f};
  for (int k = lane; k < M; k += 64) {
    const float* s = L + k * 7;
    V3 ll = v3(s[0], s[1], s[2]);
    const float ln = norm3(ll) + RB_TINY;
    ll = v3(ll.x / ln, ll.y / ln, ll.z / ln);
    const float l_lam = fabsf(s[3]);
    const float mu0[3] = {fabsf(s[4]), fabsf(s[5]), fabsf(s[6])};
    const float lv = light_vis ? light_vis[p * M + k] : 1.f;
    // ---- specular: light SG x warped BRDF SG x clamped cosine SG
    V3 f_lobe, p_lobe;
    float f_lam, f_fac, p_lam, p_fac;
    sg_product(ll, l_lam, sl.w_lobe, sl.w_lam, f_lobe, f_lam, f_fac);
    sg_product(nn, LAMBDA_COS, f_lobe, f_lam, p_lobe, p_lam, p_fac);
    const float h_p = hemi_int(p_lam, dot3(p_lobe, nn));
    const float h_f = hemi_int(f_lam, dot3(f_lobe, nn));
    // ---- diffuse: light SG x clamped cosine SG
    V3 q_lobe;
    float q_lam, q_fac;
    sg_product(nn, LAMBDA_COS, ll, l_lam, q_lobe, q_lam, q_fac);
    const float h_q = hemi_int(q_lam, dot3(q_lobe, nn));
    const float h_l = hemi_int(l_lam, dot3(ll, nn));
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float f_mu = (mu0[c] * bv) * sl.w_mu[c] * f_fac;
      const float p_mu = MU_COS * f_mu * p_fac;
      spec[c] += p_mu * h_p - f_mu * ALPHA_COS * h_f;
      float dmu = light_vis ? mu0[c] * lv : mu0[c];
      if (!lin_diff) dmu = dmu * (alb[c] / RB_PI_F);
      const float q_mu = MU_COS * dmu * q_fac;
      diff[c] += q_mu * h_q - dmu * ALPHA_COS * h_l;
      sh_num[c] += lv * mu0[c];
      sh_den[c] += mu0[c];
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    spec[c] = fmaxf(wave_sum(spec[c]), 0.f);
    diff[c] = fmaxf(wave_sum(diff[c]), 0.f);
    sh_num[c] = wave_sum(sh_num[c]);
    sh_den[c] = wave_sum(sh_den[c]);
  }
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float d = diff[c];
      if (indir_integral) d = lin_diff ? indir_integral[3 * p + c] : indir_integral[3 * p + c] * (alb[c] / RB_PI_F);
      out_spec[3 * p + c] = spec[c];
      out_diff[3 * p + c] = d;
      out_rgb[3 * p + c] = spec[c] + d;
      if (out_shadow) out_shadow[3 * p + c] = light_vis ? sh_num[c] / fmaxf(sh_den[c], 1e-4f) : 0.f;
    }
  }
}

// render_envmap_sg (sg_render.py:26-42): rgb[i] = sum_k |mu_k| exp(|lambda_k| (d_i . lobe_k/|lobe_k| - 1)); no eps in the norm
__global__ void k_envmap_sg(const float* __restrict__ lgt, int M, const float* __restrict__ dirs, long n,
                            float* __restrict__ rgb) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const V3 d = v3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]);
  float acc[3] = {0.f, 0.f, 0.f};
  for (int k = 0; k < M; ++k) {
    const float* s = lgt + k * 7;
    V3 l = v3(s[0], s[1], s[2]);
    const float ln = norm3(l);
    l = v3(l.x / ln, l.y / ln, l.z / ln);
    const float e = expf(fabsf(s[3]) * (dot3(d, l) - 1.f));
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[c] += fabsf(s[4 + c]) * e;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) rgb[3 * i + c] = acc[c];
}

// x / (|x| + eps) (mode 0) or x / max(|x|, eps) (mode 1), rows of 3
__global__ void k_normalize3(const float* __restrict__ x, long n, float eps, int mode, float* __restrict__ y) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= n) return;
  V3 a = v3(x[3 * i], x[3 * i + 1], x[3 * i + 2]);
  float nn = norm3(a);
  nn = mode == 0 ? nn + eps : fmaxf(nn, eps);
  y[3 * i] = a.x / nn;
  y[3 * i + 1] = a.y / nn;
  y[3 * i + 2] = a.z / nn;
}

}  // namespace rb

using namespace rb;

// render_envmap (model/sg_render.py:45-59): bilinear lat-long lookup, torch.grid_sample(align_corners=True, zero padding)
// semantics: phi = acos(d.z) - 1e-6, theta = atan2(d.y, d.x), x = -theta/pi, y = 2 phi/pi - 1 in [-1, 1].
__global__ void k_envmap_lookup(const float* __restrict__ env, int H, int W, const float* __restrict__ dirs, long n,
                                float* __restrict__ rgb) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float pi = (float)3.14159265358979323846;
  const float phi = acosf(dirs[3 * i + 2]) - 1e-6f;
  const float theta = atan2f(dirs[3 * i + 1], dirs[3 * i]);
  const float gx = -theta / pi, gy = (phi / pi) * 2.f - 1.f;
  const float ix = ((gx + 1.f) / 2.f) * (float)(W - 1), iy = ((gy + 1.f) / 2.f) * (float)(H - 1);
  const float fx = floorf(ix), fy = floorf(iy);
  const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
  const float wnw = ((fx + 1.f) - ix) * ((fy + 1.f) - iy), wne = (ix - fx) * ((fy + 1.f) - iy);
  const float wsw = ((fx + 1.f) - ix) * (iy - fy), wse = (ix - fx) * (iy - fy);
  auto tex = [&](int y, int x, int c) { return (x >= 0 && x < W && y >= 0 && y < H) ? env[((long)y * W + x) * 3 + c] : 0.f; };
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float a = 0.f;
    a += tex(y0, x0, c) * wnw;
    a += tex(y0, x1, c) * wne;
    a += tex(y1, x0, c) * wsw;
    a += tex(y1, x1, c) * wse;
    rgb[3 * i + c] = a;
  }
}

extern "C" {

int rb_envmap_lookup(const float* env, int H, int W, const float* dirs, long n, float* rgb, rb_stream_t stream) {
  if (n <= 0) return 0;
  RB_REQUIRE(env && dirs && rgb && H >= 1 && W >= 1, "null pointer / empty map");
  hipLaunchKernelGGL(k_envmap_lookup, grid1d(n, 256), dim3(256), 0, (hipStream_t)stream, env, H, W, dirs, n, rgb);
  return check_launch("k_envmap_lookup");
}

int rb_envmap_sg(const float* lgt, int M, const float* dirs, long n, float* rgb, rb_stream_t stream) {
  if (n <= 0) return 0;
  RB_REQUIRE(lgt && dirs && rgb && M >= 1, "null pointer");
  hipLaunchKernelGGL(k_envmap_sg, grid1d(n, 256), dim3(256), 0, (hipStream_t)stream, lgt, M, dirs, n, rgb);
  return check_launch("k_envmap_sg");
}

int rb_normalize3(const float* x, long n, float eps, int mode, float* y, rb_stream_t stream) {
  if (n <= 0) return 0;
  RB_REQUIRE(x && y, "null pointer");
  hipLaunchKernelGGL(k_normalize3, grid1d(n, 256), dim3(256), 0, (hipStream_t)stream, x, n, eps, mode, y);
  return check_launch("k_normalize3");
}

int rb_spec_vis_sample(const float* normal, const float* view, const float* rough, const int* chunk_id, long n,
                       int n_chunks, int nsamp, const float* u_theta, const float* u_phi, float* sharp,
                       unsigned* chunk_min, float* dirs, float* wts, unsigned char* front, rb_stream_t stream) {
  if (n <= 0) return 0;
  RB_REQUIRE(normal && view && rough && u_theta && u_phi && sharp && chunk_min && dirs && wts && front, "null pointer");
  RB_REQUIRE(n_chunks >= 1 && nsamp >= 1, "bad sizes");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_fill_u32, grid1d(n_chunks, 256), dim3(256), 0, s, chunk_min, (long)n_chunks, 0x7f800000u);
  hipLaunchKernelGGL(k_spec_sharp, grid1d(n, 256), dim3(256), 0, s, normal, view, rough, chunk_id, n, sharp, chunk_min);
  hipLaunchKernelGGL(k_spec_dirs, grid1d(n * nsamp, 256), dim3(256), 0, s, normal, view, sharp, chunk_id, chunk_min,
                     u_theta, u_phi, n, nsamp, dirs, wts, front);
  return check_launch("k_spec_dirs");
}

int rb_spec_vis_sample_lobes(const float* normal, const float* view, const float* lobes, const float* lambdas, const int* chunk_id,
                             long n, int n_chunks, int nsamp, const float* u_theta, const float* u_phi, float* sharp,
                             unsigned* chunk_min, float* dirs, float* wts, unsigned char* front, rb_stream_t stream) {
  if (n <= 0) return 0;
  RB_REQUIRE(normal && view && lobes && lambdas && u_theta && u_phi && sharp && chunk_min && dirs && wts && front, "null pointer");
  RB_REQUIRE(n_chunks >= 1 && nsamp >= 1, "bad sizes");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_fill_u32, grid1d(n_chunks, 256), dim3(256), 0, s, chunk_min, (long)n_chunks, 0x7f800000u);
  hipLaunchKernelGGL(k_spec_sharp_lobes, grid1d(n, 256), dim3(256), 0, s, lambdas, chunk_id, n, sharp, chunk_min);
  hipLaunchKernelGGL(k_spec_dirs_lobes, grid1d(n * nsamp, 256), dim3(256), 0, s, normal, view, lobes, sharp, chunk_id, chunk_min,
                     u_theta, u_phi, n, nsamp, dirs, wts, front);
  return check_launch("k_spec_dirs_lobes");
}

int rb_spec_vis_reduce(const float* logits, const unsigned char* front, const float* wts, long n, int nsamp, int inv,
                       int argmax_vis, int testing, float* bvis, rb_stream_t stream) {
  if (n <= 0) return 0;
  RB_REQUIRE(logits && front && wts && bvis, "null pointer");
  hipLaunchKernelGGL(k_spec_reduce, grid1d(n, 256), dim3(256), 0, (hipStream_t)stream, logits, front, wts, n, nsamp,
                     inv, argmax_vis, testing, bvis);
  return check_launch("k_spec_reduce");
}

int rb_sg_shade(const float* normal, const float* view, const float* lgt, int per_point_lgt, int M, const float* f0,
                const float* rough, const float* albedo, const float* metallic, const float* light_vis,
                const float* bvis, const float* indir_integral, int lin_diff, long n, float* out_rgb, float* out_spec,
                float* out_diff, float* out_shadow, rb_stream_t stream) {
  if (n <= 0) return 0;
  RB_REQUIRE(normal && view && lgt && f0 && rough && albedo && bvis && out_rgb && out_spec && out_diff, "null pointer");
  RB_REQUIRE(M >= 1, "need at least one lobe");
  hipLaunchKernelGGL(k_sg_shade, grid1d(n, 4), dim3(256), 0, (hipStream_t)stream, normal, view, lgt, per_point_lgt, M, f0,
                     rough, albedo, metallic, light_vis, bvis, indir_integral, lin_diff, n, out_rgb, out_spec, out_diff,
                     out_shadow);
  return check_launch("k_sg_shade");
}

}  // extern "C"
