// Small element-wise / per-ray kernels behind the public helper functions of the overlaid reference modules that the fused
// forward never needs by themselves (round 6: the Python operator surface of SURVEY.md section 8b exports EVERY public name):
//   model/neus_model.py:136-184   PE / Embedder.embed for any (input_dims, num_freq)          rb_pe_encode
//   model/neus_model.py:14-24     expected_sin (integrated positional encoding)               rb_expected_sin
//   model/color_correction.py:31-73  aces_fn ... ln_space_inv as free functions (no clamp)    rb_tonemap_curve
//   model/sdf_render.py:37-67     sample_pdf with arbitrary (unsorted, per-ray) u             rb_sample_pdf
//   model/sdf_render.py:186-225   render_core's dists / cdf / inside_sphere entries           rb_neus_core_aux
//   model/implicit_differentiable_renderer.py:548-564  IDRNetwork.sample_dirs                 rb_sample_dirs
//   model/octree_tracing.py:70-76 OctreeVisModel.intersect_sphere                             rb_intersect_sphere
// fp32, reference operation order (-ffp-contract=off), one thread per output row; none of these is on the metric's path.
#include "../../include/robir_hip.h"
#include "common.h"

namespace rb {

// out[i, :] = [x (d, if include_input) | for k < n_freq: sin(x * freq[k]) (d), cos(x * freq[k]) (d)]
__global__ void k_pe_encode(const float* __restrict__ x, long n, int d, const float* __restrict__ freq, int n_freq,
                            int include_input, float* __restrict__ out) {
  const int width = (include_input ? d : 0) + 2 * d * n_freq;
  const long t = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (t >= n * width) return;
  const long i = t / width;
  int c = (int)(t % width);
  const float* xi = x + i * d;
  if (include_input) {
    if (c < d) {
      out[t] = xi[c];
      return;
    }
    c -= d;
  }
  const int k = c / (2 * d), r = c % (2 * d);
  const float a = xi[r % d] * freq[k];
  out[t] = r < d ? sinf(a) : cosf(a);
}

__device__ __forceinline__ float py_mod_(float a, float m) {      // torch.remainder: sign of the divisor
  float r = fmodf(a, m);
  if (r != 0.f && ((r < 0.f) != (m < 0.f))) r += m;
  return r;
}
__device__ __forceinline__ float safe_arg(float v) {              // safe_trig_helper: wrap |v| >= 100 pi
  const float big = (float)(100.0 * 3.14159265358979323846);
  return fabsf(v) < big ? v : py_mod_(v, big);
}
// y = exp(-0.5 var) sin(x);  y_var = relu(0.5 (1 - exp(-2 var) cos(2x)) - y^2)
__global__ void k_expected_sin(const float* __restrict__ x, const float* __restrict__ var, long n, float* __restrict__ y,
                               float* __restrict__ yvar) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = var[i], a = x[i];
  const float yy = expf(-0.5f * v) * sinf(safe_arg(a));
  y[i] = yy;
  if (yvar) {
    const float r = 0.5f * (1.f - expf(-2.f * v) * cosf(safe_arg(2.f * a))) - yy * yy;
    yvar[i] = fmaxf(r, 0.f);
  }
}

__device__ __forceinline__ float aces_f(float x) { return x * (2.51f * x + 0.03f) / (x * (2.43f * x + 0.59f) + 0.14f); }
__device__ __forceinline__ float aces_i(float x) {
  const float b = 0.59f * x - 0.03f;
  return (b + sqrtf(b * b + 4.f * (2.51f - 2.43f * x) * 0.14f * x)) / (2.f * (2.51f - 2.43f * x));
}
// curve: 0 aces_fn(x) 1 aces_inv(x) 2 warp_aces_fn 3 warp_aces_inv 4 scale_aces_fn 5 scale_aces_inv 6 identity_fn
//        7 ln_space_fn 8 ln_space_inv; t = shift[(i / width) * stride] as given (the free functions do not clamp)
__global__ void k_tonemap_curve(const float* __restrict__ x, long n, int width, const float* __restrict__ shift, int stride,
                                int curve, float* __restrict__ y) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = x[i];
  const float t = shift ? shift[(i / width) * stride] : 1.f;
  float r;
  switch (curve) {
    case 0: r = aces_f(v); break;
    case 1: r = aces_i(v); break;
    case 2: r = aces_f(aces_i(0.73f * t) / 0.73f * v) / t; break;
    case 3: r = 0.73f * aces_i(v * t) / aces_i(0.73f * t); break;
    case 4: r = aces_f(v) / powf(t, 0.2f); break;
    case 5: r = aces_i(v * powf(t, 0.2f)); break;
    case 7: {
      const float u = v * (0.5f + t) / 0.5f;
      r = u / (1.f + t * u);
      break;
    }
    case 8: {
      const float u = v / (1.f - t * v);
      r = u * 0.5f / (0.5f + t);
      break;
    }
    default: r = v;
  }
  y[i] = r;
}

// sample_pdf: bins [R,n], weights [R,n-1] (the +1e-5 is applied here), u [R,n_s] (u_stride = n_s) or [n_s] shared (0).
// cdf [R,n] scratch/out: cdf[0] = 0, cdf[i+1] = cdf[i] + (w[i] + 1e-5)/sum.  One thread per ray.
__global__ void k_sample_pdf(const float* __restrict__ bins, const float* __restrict__ weights, long R, int n,
                             const float* __restrict__ u, long u_stride, int n_s, float* __restrict__ cdf,
                             float* __restrict__ samples) {
  const long r = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (r >= R) return;
  const float* b = bins + r * n;
  const float* w = weights + r * (n - 1);
  float* c = cdf + r * n;
  float sum = 0.f;
  for (int i = 0; i + 1 < n; ++i) sum += w[i] + 1e-5f;
  float acc = 0.f;
  c[0] = 0.f;
  for (int i = 0; i + 1 < n; ++i) {
    acc = acc + (w[i] + 1e-5f) / sum;
    c[i + 1] = acc;
  }
  for (int k = 0; k < n_s; ++k) {
    const float uk = u[r * u_stride + k];
    int lo = 0, hi = n;                      // searchsorted(right=True): first index with cdf[idx] > u
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (c[mid] > uk) hi = mid; else lo = mid + 1;
    }
    const int below = lo - 1 > 0 ? lo - 1 : 0;
    const int above = lo < n - 1 ? lo : n - 1;
    float den = c[above] - c[below];
    if (den < 1e-5f) den = 1.f;
    const float t = (uk - c[below]) / den;
    samples[r * n_s + k] = b[below] + t * (b[above] - b[below]);
  }
}

// render_core's per-sample extras: dists (last = sample_dist), cdf = sigmoid(sdf_k * inv_s) (`c`, sdf_render.py:214-218),
// inside = |p| < radius (sdf: column 0 of an [M, sdf_stride] matrix)
__global__ void k_neus_core_aux(const float* __restrict__ sdf, long sdf_stride, const float* __restrict__ pts,
                                const float* __restrict__ z, long R, int n, float inv_s, float radius, float sample_dist,
                                float* __restrict__ dists, float* __restrict__ cdf, float* __restrict__ inside) {
  const long j = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (j >= R * n) return;
  const int k = (int)(j % n);
  if (dists) dists[j] = k + 1 < n ? z[j + 1] - z[j] : sample_dist;
  if (cdf) cdf[j] = 1.f / (1.f + expf(-(sdf[j * sdf_stride] * inv_s)));
  if (inside) {
    const float px = pts[3 * j], py = pts[3 * j + 1], pz = pts[3 * j + 2];
    inside[j] = sqrtf(px * px + py * py + pz * pz) < radius ? 1.f : 0.f;
  }
}

__device__ __forceinline__ void norm_axis3(float* v) {            // x / (|x| + 1e-6)
  const float l = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]) + 1e-6f;
  v[0] /= l;
  v[1] /= l;
  v[2] /= l;
}
// IDRNetwork.sample_dirs: tangent frame from z_axis = (1,0,0): U = norm(z x n), V = norm(n x U);
// dir = U cos(theta) sin(phi) + V sin(theta) sin(phi) + n cos(phi)
__global__ void k_sample_dirs(const float* __restrict__ normals, const float* __restrict__ theta,
                              const float* __restrict__ phi, long n, float* __restrict__ dirs) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= n) return;
  float nn[3] = {normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]};
  norm_axis3(nn);
  float U[3] = {0.f * nn[2] - 0.f * nn[1], 0.f * nn[0] - 1.f * nn[2], 1.f * nn[1] - 0.f * nn[0]};
  norm_axis3(U);
  float V[3] = {nn[1] * U[2] - nn[2] * U[1], nn[2] * U[0] - nn[0] * U[2], nn[0] * U[1] - nn[1] * U[0]};
  norm_axis3(V);
  const float ct = cosf(theta[i]), st = sinf(theta[i]), cp = cosf(phi[i]), sp = sinf(phi[i]);
#pragma unroll
  for (int c = 0; c < 3; ++c) dirs[3 * i + c] = (U[c] * ct * sp + V[c] * st * sp) + nn[c] * cp;
}

// OctreeVisModel.intersect_sphere: d = d / max(|d|, 1e-4); closest = (-o.d) d + o; out = closest + d sqrt(r^2 - |closest|^2)
__global__ void k_intersect_sphere(const float* __restrict__ o, const float* __restrict__ d, long n, float radius,
                                   float* __restrict__ out) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= n) return;
  float dd[3] = {d[3 * i], d[3 * i + 1], d[3 * i + 2]};
  const float oo[3] = {o[3 * i], o[3 * i + 1], o[3 * i + 2]};
  const float l = fmaxf(sqrtf(dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2]), 1e-4f);
  dd[0] /= l;
  dd[1] /= l;
  dd[2] /= l;
  const float s = (-oo[0] * dd[0] + -oo[1] * dd[1]) + -oo[2] * dd[2];
  float cl[3], c2 = 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) cl[c] = s * dd[c] + oo[c];
  c2 = (cl[0] * cl[0] + cl[1] * cl[1]) + cl[2] * cl[2];
  const float t = sqrtf(radius * radius - c2);                     // NaN outside the sphere, like the reference
#pragma unroll
  for (int c = 0; c < 3; ++c) out[3 * i + c] = cl[c] + dd[c] * t;
}

}  // namespace rb

using namespace rb;

extern "C" {

int rb_pe_encode(const float* x, long n, int d, const float* freq, int n_freq, int include_input, float* out,
                 rb_stream_t stream) {
  if (n <= 0) return 0;
  RB_REQUIRE(x && out && (freq || n_freq == 0), "null pointer");
  RB_REQUIRE(d >= 1 && n_freq >= 0, "need d >= 1, n_freq >= 0");
  const long width = (include_input ? d : 0) + 2L * d * n_freq;
  if (width == 0) return 0;
  hipLaunchKernelGGL(k_pe_encode, grid1d(n * width, 256), dim3(256), 0, (hipStream_t)stream, x, n, d, freq, n_freq,
                     include_input, out);
  return check_launch("k_pe_encode");
}

int rb_expected_sin(const float* x, const float* var, long n, float* y, float* yvar, rb_stream_t stream) {
  if (n <= 0) return 0;
  RB_REQUIRE(x && var && y, "null pointer");
  hipLaunchKernelGGL(k_expected_sin, grid1d(n, 256), dim3(256), 0, (hipStream_t)stream, x, var, n, y, yvar);
  return check_launch("k_expected_sin");
}

int rb_tonemap_curve(const float* x, long n, int width, const float* shift, int shift_stride, int curve, float* y,
                     rb_stream_t stream) {
  if (n <= 0) return 0;
  RB_REQUIRE(x && y, "null pointer");
  RB_REQUIRE(curve >= 0 && curve <= 8 && width >= 1, "curve 0..8, width >= 1");
  RB_REQUIRE(shift || curve <= 1 || curve == 6, "this curve needs a shift");
  hipLaunchKernelGGL(k_tonemap_curve, grid1d(n, 256), dim3(256), 0, (hipStream_t)stream, x, n, width, shift, shift_stride,
                     curve, y);
  return check_launch("k_tonemap_curve");
}

int rb_sample_pdf(const float* bins, const float* weights, long R, int n, const float* u, long u_stride, int n_s,
                  float* cdf, float* samples, rb_stream_t stream) {
  if (R <= 0 || n_s <= 0) return 0;
  RB_REQUIRE(bins && weights && u && cdf && samples, "null pointer");
  RB_REQUIRE(n >= 2, "need n >= 2 bins");
  RB_REQUIRE(u_stride == 0 || u_stride == n_s, "u is [n_s] (stride 0) or [R, n_s]");
  hipLaunchKernelGGL(k_sample_pdf, grid1d(R, 128), dim3(128), 0, (hipStream_t)stream, bins, weights, R, n, u, u_stride,
                     n_s, cdf, samples);
  return check_launch("k_sample_pdf");
}

int rb_neus_core_aux(const float* sdf, long sdf_stride, const float* pts, const float* z, long R, int n, float inv_s,
                     float radius, float sample_dist, float* dists, float* cdf, float* inside, rb_stream_t stream) {
  if (R <= 0) return 0;
  RB_REQUIRE((!dists || z) && (!cdf || sdf) && (!inside || pts), "null pointer");
  hipLaunchKernelGGL(k_neus_core_aux, grid1d(R * n, 256), dim3(256), 0, (hipStream_t)stream, sdf, sdf_stride, pts, z, R, n,
                     inv_s, radius, sample_dist, dists, cdf, inside);
  return check_launch("k_neus_core_aux");
}

int rb_sample_dirs(const float* normals, const float* theta, const float* phi, long n, float* dirs, rb_stream_t stream) {
  if (n <= 0) return 0;
  RB_REQUIRE(normals && theta && phi && dirs, "null pointer");
  hipLaunchKernelGGL(k_sample_dirs, grid1d(n, 256), dim3(256), 0, (hipStream_t)stream, normals, theta, phi, n, dirs);
  return check_launch("k_sample_dirs");
}

int rb_intersect_sphere(const float* origins, const float* dirs, long n, float radius, float* out, rb_stream_t stream) {
  if (n <= 0) return 0;
  RB_REQUIRE(origins && dirs && out, "null pointer");
  hipLaunchKernelGGL(k_intersect_sphere, grid1d(n, 256), dim3(256), 0, (hipStream_t)stream, origins, dirs, n, radius, out);
  return check_launch("k_intersect_sphere");
}

}  // extern "C"
