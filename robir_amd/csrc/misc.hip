// Ray generation and tone mapping (element-wise edges of the per-ray path).
// utils/rend_util.py:51-97 (get_camera_params / lift, 4x4 pose branch); model/color_correction.py:31-60,116-137.
#include "../../include/robir_hip.h"
#include "common.h"

namespace rb {

// camera rays (get_camera_params + lift, utils/rend_util.py:51-97), pose / intrinsics read from device memory (a per-chunk forward() must not wait for a device-to-host copy)
__global__ void k_camera_rays_dev(const float* __restrict__ pose, const float* __restrict__ K, const float* __restrict__ uv, long N,
                                  float* __restrict__ dirs) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= N) return;
  const float fx = K[0], sk = K[1], cx = K[2], fy = K[4], cy = K[5];
  const float x = uv[2 * i], y = uv[2 * i + 1], z = 1.0f;
  const float xl = (x - cx + cy * sk / fy - sk * y / fy) / fx * z;
  const float yl = (y - cy) / fy * z;
  const float pc[4] = {xl, -yl, -z, 1.0f};
  float w[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    float acc = pose[4 * r] * pc[0];
    acc = acc + pose[4 * r + 1] * pc[1];
    acc = acc + pose[4 * r + 2] * pc[2];
    acc = acc + pose[4 * r + 3] * pc[3];
    w[r] = acc - pose[4 * r + 3];
  }
  const float n = fmaxf(sqrtf(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]), 1e-12f);
  dirs[3 * i] = w[0] / n;
  dirs[3 * i + 1] = w[1] / n;
  dirs[3 * i + 2] = w[2] / n;
}

// points = origin (+ per-ray or shared) + t * dir      (implicit_differentiable_renderer.py:324)
__global__ void k_points_along(const float* __restrict__ origins, int per_ray_origin, long batch,
                               const float* __restrict__ dirs, const float* __restrict__ t, long N,
                               float* __restrict__ pts) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= N) return;
  const float* o = origins + 3 * (per_ray_origin ? i : i / batch);
#pragma unroll
  for (int c = 0; c < 3; ++c) pts[3 * i + c] = o[c] + t[i] * dirs[3 * i + c];
}

__device__ __forceinline__ float aces(float x) { return x * (2.51f * x + 0.03f) / (x * (2.43f * x + 0.59f) + 0.14f); }
__device__ __forceinline__ float aces_inv(float x) {
  const float q = 0.59f * x - 0.03f;
  return (q + sqrtf(q * q + 4.f * (2.51f - 2.43f * x) * 0.14f * x)) / (2.f * (2.51f - 2.43f * x));
}
// ACESToneMapping.hdr2ldr / ldr2hdr (color_correction.py:31-73,116-134).  op 0: hdr2ldr, 1: ldr2hdr, 2: ldr2hdr(x^2.2)
// (trace_radiance); hdr_mode 0: aces(x) / t^0.2 | aces^-1(x t^0.2) (every shipped conf); 1: energy-warped ACES;
// 2: log-space curve; anything else: identity.  shift: [n] per row (rows of 3 channels) or a single value
// (shift_stride 0), clamped to [1e-4, 1].
__global__ void k_tonemap(const float* __restrict__ x, long n, const float* __restrict__ shift, int shift_stride,
                          int op, int hdr_mode, float* __restrict__ y) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= 3 * n) return;
  float t = shift[(i / 3) * shift_stride];
  t = fminf(fmaxf(t, 1e-4f), 1.f);
  float v = x[i];
  if (op == 2) v = powf(v, 2.2f);
  const bool inverse = op != 0;
  float r;
  if (hdr_mode == 0) {
    const float tp = powf(t, 0.2f);
    r = inverse ? aces_inv(v * tp) : aces(v) / tp;
  } else if (hdr_mode == 1) {
    r = inverse ? 0.73f * aces_inv(v * t) / aces_inv(0.73f * t) : aces(aces_inv(0.73f * t) / 0.73f * v) / t;
  } else if (hdr_mode == 2) {
    if (inverse) {
      const float u = v / (1.f - t * v);
      r = u * 0.5f / (0.5f + t);
    } else {
      const float u = v * (0.5f + t) / 0.5f;
      r = u / (1.f + t * u);
    }
  } else {
    r = v;
  }
  y[i] = r;
}

// ---- secondary-ray pieces (implicit_differentiable_renderer.py:583-641, neus_model.py:828-871)
// uniform sphere directions from two uniform draws; back-face flag against the (unnormalised) normal;
// secondary origin = x + 0.005 * n/|n|
__global__ void k_sphere_dirs(const float* __restrict__ u1, const float* __restrict__ u2,
                              const float* __restrict__ normals, const float* __restrict__ points, long n, int nsamp,
                              float* __restrict__ dirs, unsigned char* __restrict__ back, float* __restrict__ cosw,
                              float* __restrict__ origins) {
  long j = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (j >= n * nsamp) return;
  const long i = j / nsamp;
  const float pi = (float)3.14159265358979323846;
  const float u = u1[j] * 2.f - 1.f;
  const float t = u2[j] * pi * 2.f;
  const float s = powf(1.f - u * u, 0.5f);
  const float d[3] = {s * cosf(t), s * sinf(t), u};
  float nn[3] = {normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]};
  const float ln = fmaxf(sqrtf(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]), 1e-4f);
  nn[0] /= ln;
  nn[1] /= ln;
  nn[2] /= ln;
  const float c = nn[0] * d[0] + nn[1] * d[1] + nn[2] * d[2];
#pragma unroll
  for (int k = 0; k < 3; ++k) dirs[3 * j + k] = d[k];
  back[j] = c < 0.f ? 1 : 0;
  cosw[j] = fmaxf(c, 0.f);
  if (j % nsamp == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) origins[3 * i + k] = points[3 * i + k] + nn[k] * 0.005f;
  }
}

// NeuS sample points of borrow_color: x = 2p + dir*t_k with dir = -view/|view| (neus_model.py:856-865)
__global__ void k_borrow_points(const float* __restrict__ points, const float* __restrict__ view,
                                const float* __restrict__ tk, long m, int ns, float* __restrict__ x,
                                float* __restrict__ dirs) {
  long j = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (j >= m * ns) return;
  const long i = j / ns;
  const int k = (int)(j % ns);
  const float v[3] = {view[3 * i], view[3 * i + 1], view[3 * i + 2]};
  const float ln = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float d = -v[c] / ln;
    x[3 * j + c] = points[3 * i + c] * 2.f + d * tk[k];
    dirs[3 * j + c] = d;
  }
}

// NeuS alpha compositing of ns samples per ray (neus_model.py:828-854 / sdf_render.py:209-240):
// alpha from consecutive-sample SDFs, clipped to [lo,hi], optional per-sample mask, weights = alpha * cumprod.
__global__ void k_neus_composite(const float* __restrict__ sdf, const float* __restrict__ color,
                                 const float* __restrict__ mask, long m, int ns, float inv_s, float lo, float hi,
                                 float eps, float* __restrict__ rgb, float* __restrict__ weights) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= m) return;
  float T = 1.f, acc[3] = {0.f, 0.f, 0.f};
  for (int k = 0; k < ns; ++k) {
    const float prv = sdf[i * ns + k];
    const float nxt = sdf[i * ns + (k + 1 < ns ? k + 1 : ns - 1)];
    const float sp = (k + 1 < ns) ? prv : sdf[i * ns + ns - 1];
    const float c0 = 1.f / (1.f + expf(-(sp * inv_s)));
    const float c1 = 1.f / (1.f + expf(-(nxt * inv_s)));
    float a = ((c0 - c1) + 1e-5f) / (c0 + 1e-5f);
    a = fminf(fmaxf(a, lo), hi);
    if (mask) a = a * mask[i * ns + k];
    const float w = a * T;
    if (weights) weights[i * ns + k] = w;
    if (color) {
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[c] += color[(i * ns + k) * 3 + c] * w;
    }
    T = T * (1.f - a + eps);
  }
  if (rgb) {
#pragma unroll
    for (int c = 0; c < 3; ++c) rgb[3 * i + c] = acc[c];
  }
}

// cosine-weighted hemisphere mean of the traced radiance (implicit_differentiable_renderer.py:639-641)
__global__ void k_trace_integrate(const float* __restrict__ rad, const float* __restrict__ cosw,
                                  const unsigned char* __restrict__ back, long n, int nsamp, float* __restrict__ out) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= n) return;
  float acc[3] = {0.f, 0.f, 0.f};
  int cnt = 0;
  for (int k = 0; k < nsamp; ++k) {
    const long j = i * nsamp + k;
    cnt += back[j] ? 0 : 1;
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[c] += rad[3 * j + c] * cosw[j];
  }
  const float den = fmaxf((float)cnt, 1e-4f);
#pragma unroll
  for (int c = 0; c < 3; ++c) out[3 * i + c] = acc[c] / den;
}

// EnvmapMaterialNetwork head (sg_envmap_material.py:205-211): brdf / brdf_r [n,5] (sigmoid outputs of the spec AE) ->
// albedo [n,3], roughness [n,1] = b3*0.9+0.09, metallic [n,1] = b4*0.99+0.01, and the random_xi_* twins
// (random_xi_metallic is NOT rescaled in the reference).
__global__ void k_material_decode(const float* __restrict__ brdf, const float* __restrict__ brdf_r, long n,
                                  float* __restrict__ albedo, float* __restrict__ rough, float* __restrict__ metal,
                                  float* __restrict__ albedo_r, float* __restrict__ rough_r, float* __restrict__ metal_r) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= n) return;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    albedo[3 * i + c] = brdf[5 * i + c];
    albedo_r[3 * i + c] = brdf_r[5 * i + c];
  }
  rough[i] = brdf[5 * i + 3] * 0.9f + 0.09f;
  metal[i] = brdf[5 * i + 4] * 0.99f + 0.01f;
  rough_r[i] = brdf_r[5 * i + 3] * 0.9f + 0.09f;
  metal_r[i] = brdf_r[5 * i + 4];
}

// y = (take_abs ? |x| : x) * s   (IndirctIllumNetwork: env_int = abs(...), implicit_differentiable_renderer.py:220; hooks: * 2 pi)
__global__ void k_abs_scale(const float* __restrict__ x, long n, float s, int take_abs, float* __restrict__ y) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i < n) y[i] = (take_abs ? fabsf(x[i]) : x[i]) * s;
}

// softmax over pairs of logits, component `which` (torch.softmax(x, -1)[..., which])
__global__ void k_softmax2(const float* __restrict__ logits, long n, int which, float* __restrict__ p) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float l0 = logits[2 * i], l1 = logits[2 * i + 1];
  const float mx = fmaxf(l0, l1);
  const float e0 = expf(l0 - mx), e1 = expf(l1 - mx);
  p[i] = (which ? e1 : e0) / (e0 + e1);
}

// CESR recombination (training/train_cesr.py:523-524): rgb = diffuse * albedo / pi + specular, rows of 3
__global__ void k_lin_diff_combine(const float* __restrict__ diffuse, const float* __restrict__ albedo,
                                   const float* __restrict__ spec, long n3, float* __restrict__ rgb) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i < n3) rgb[i] = diffuse[i] * albedo[i] / (float)3.14159265358979323846 + spec[i];
}

}  // namespace rb

using namespace rb;

// ---- the per-pixel outputs of a forward(): K row sets [n, 1 or 3] of the hit pixels scattered into K consecutive [N, 1 or 3] blocks
// of one flat buffer (pre-filled with the defaults) in ONE launch, instead of one index_put per output
struct ScatterArgs {
  const float* src[32];
  long dst_off[32];          // first float of block k in the flat buffer
  unsigned char src_w[32], dst_w[32];
  unsigned char col_k[96], col_c[96];      // destination column (all blocks side by side) -> block, column in block
  int total_cols;
};
__global__ void k_scatter_rows(ScatterArgs a, const long* __restrict__ idx, long n, float* __restrict__ flat) {
  const long t = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (t >= n * a.total_cols) return;
  const long j = t / a.total_cols;
  const int col = (int)(t - j * a.total_cols), k = a.col_k[col], c = a.col_c[col];
  const int sw = a.src_w[k];
  flat[a.dst_off[k] + idx[j] * a.dst_w[k] + c] = a.src[k][j * sw + (sw == 1 ? 0 : c)];
}

extern "C" {

int rb_scatter_rows(const float* const* src, const int* src_width, const int* dst_width, int K, const long* idx, long n, long N,
                    float* flat, rb_stream_t stream) {
  if (n <= 0 || K <= 0) return 0;
  RB_REQUIRE(src && src_width && dst_width && idx && flat, "null pointer");
  RB_REQUIRE(K <= 32, "at most 32 row sets");
  ScatterArgs a;
  long off = 0;
  int cols = 0;
  for (int k = 0; k < K; ++k) {
    RB_REQUIRE(src[k] && (src_width[k] == dst_width[k] || src_width[k] == 1) && dst_width[k] >= 1 && dst_width[k] <= 4, "row widths: source = destination or 1 (broadcast), destination 1..4");
    a.src[k] = src[k];
    a.src_w[k] = (unsigned char)src_width[k];
    a.dst_w[k] = (unsigned char)dst_width[k];
    a.dst_off[k] = off;
    off += N * dst_width[k];
    for (int c = 0; c < dst_width[k]; ++c) {
      RB_REQUIRE(cols < 96, "at most 96 destination columns");
      a.col_k[cols] = (unsigned char)k;
      a.col_c[cols] = (unsigned char)c;
      ++cols;
    }
  }
  a.total_cols = cols;
  hipLaunchKernelGGL(k_scatter_rows, grid1d(n * cols, 256), dim3(256), 0, (hipStream_t)stream, a, idx, n, flat);
  return check_launch("k_scatter_rows");
}

int rb_camera_rays_dev(const float* pose_dev, const float* K_dev, const float* uv, long N, float* dirs, rb_stream_t stream) {
  if (N <= 0) return 0;
  RB_REQUIRE(pose_dev && K_dev && uv && dirs, "null pointer");
  hipLaunchKernelGGL(k_camera_rays_dev, grid1d(N, 256), dim3(256), 0, (hipStream_t)stream, pose_dev, K_dev, uv, N, dirs);
  return check_launch("k_camera_rays_dev");
}

int rb_points_along(const float* origins, int per_ray_origin, long batch, const float* dirs, const float* t, long N,
                    float* pts, rb_stream_t stream) {
  if (N <= 0) return 0;
  RB_REQUIRE(origins && dirs && t && pts, "null pointer");
  RB_REQUIRE(per_ray_origin || batch >= 1, "batch must be >= 1");
  hipLaunchKernelGGL(k_points_along, grid1d(N, 256), dim3(256), 0, (hipStream_t)stream, origins, per_ray_origin, batch,
                     dirs, t, N, pts);
  return check_launch("k_points_along");
}

int rb_sphere_dirs(const float* u1, const float* u2, const float* normals, const float* points, long n, int nsamp,
                   float* dirs, unsigned char* back, float* cosw, float* origins, rb_stream_t stream) {
  if (n <= 0) return 0;
  RB_REQUIRE(u1 && u2 && normals && points && dirs && back && cosw && origins, "null pointer");
  hipLaunchKernelGGL(k_sphere_dirs, grid1d(n * nsamp, 256), dim3(256), 0, (hipStream_t)stream, u1, u2, normals, points, n,
                     nsamp, dirs, back, cosw, origins);
  return check_launch("k_sphere_dirs");
}

int rb_borrow_points(const float* points, const float* view, const float* tk, long m, int ns, float* x, float* dirs,
                     rb_stream_t stream) {
  if (m <= 0) return 0;
  RB_REQUIRE(points && view && tk && x && dirs, "null pointer");
  hipLaunchKernelGGL(k_borrow_points, grid1d(m * ns, 256), dim3(256), 0, (hipStream_t)stream, points, view, tk, m, ns, x,
                     dirs);
  return check_launch("k_borrow_points");
}

int rb_neus_composite(const float* sdf, const float* color, const float* mask, long m, int ns, float inv_s, float lo,
                      float hi, float eps, float* rgb, float* weights, rb_stream_t stream) {
  if (m <= 0) return 0;
  RB_REQUIRE(sdf && (rgb == nullptr || color != nullptr), "null pointer");
  hipLaunchKernelGGL(k_neus_composite, grid1d(m, 128), dim3(128), 0, (hipStream_t)stream, sdf, color, mask, m, ns, inv_s,
                     lo, hi, eps, rgb, weights);
  return check_launch("k_neus_composite");
}

int rb_trace_integrate(const float* rad, const float* cosw, const unsigned char* back, long n, int nsamp, float* out,
                       rb_stream_t stream) {
  if (n <= 0) return 0;
  RB_REQUIRE(rad && cosw && back && out, "null pointer");
  hipLaunchKernelGGL(k_trace_integrate, grid1d(n, 128), dim3(128), 0, (hipStream_t)stream, rad, cosw, back, n, nsamp, out);
  return check_launch("k_trace_integrate");
}

int rb_material_decode(const float* brdf, const float* brdf_r, long n, float* albedo, float* rough, float* metal,
                       float* albedo_r, float* rough_r, float* metal_r, rb_stream_t stream) {
  if (n <= 0) return 0;
  RB_REQUIRE(brdf && brdf_r && albedo && rough && metal && albedo_r && rough_r && metal_r, "null pointer");
  hipLaunchKernelGGL(k_material_decode, grid1d(n, 256), dim3(256), 0, (hipStream_t)stream, brdf, brdf_r, n, albedo, rough,
                     metal, albedo_r, rough_r, metal_r);
  return check_launch("k_material_decode");
}

int rb_abs_scale(const float* x, long n, float s, int take_abs, float* y, rb_stream_t stream) {
  if (n <= 0) return 0;
  RB_REQUIRE(x && y, "null pointer");
  hipLaunchKernelGGL(k_abs_scale, grid1d(n, 256), dim3(256), 0, (hipStream_t)stream, x, n, s, take_abs, y);
  return check_launch("k_abs_scale");
}

int rb_softmax2(const float* logits, long n, int which, float* p, rb_stream_t stream) {
  if (n <= 0) return 0;
  RB_REQUIRE(logits && p && (which == 0 || which == 1), "bad arguments");
  hipLaunchKernelGGL(k_softmax2, grid1d(n, 256), dim3(256), 0, (hipStream_t)stream, logits, n, which, p);
  return check_launch("k_softmax2");
}

int rb_lin_diff_combine(const float* diffuse, const float* albedo, const float* spec, long n, float* rgb,
                        rb_stream_t stream) {
  if (n <= 0) return 0;
  RB_REQUIRE(diffuse && albedo && spec && rgb, "null pointer");
  hipLaunchKernelGGL(k_lin_diff_combine, grid1d(3 * n, 256), dim3(256), 0, (hipStream_t)stream, diffuse, albedo, spec,
                     3 * n, rgb);
  return check_launch("k_lin_diff_combine");
}

int rb_tonemap(const float* x, long n, const float* shift, int shift_stride, int mode, float* y, rb_stream_t stream) {
  if (n <= 0) return 0;
  RB_REQUIRE(x && shift && y, "null pointer");
  RB_REQUIRE((mode & 15) <= 2 && mode >= 0, "mode = op (0 hdr2ldr, 1 ldr2hdr, 2 ldr2hdr(x^2.2)) + 16 * hdr_mode code (0, 1, 2, 3 = identity)");
  hipLaunchKernelGGL(k_tonemap, grid1d(3 * n, 256), dim3(256), 0, (hipStream_t)stream, x, n, shift, shift_stride, mode & 15,
                     mode >> 4, y);
  return check_launch("k_tonemap");
}

}  // extern "C"
