// NeuS ray-march with hierarchical sampling: the non-MLP parts of render_neus / up_sample / sample_pdf / cat_z_vals /
// render_core (model/sdf_render.py:37-132,175-374) and of NormalTrainRunner.get_neus_surface
// (training/train_normal.py:239-286).  The SDF / gradient / colour evaluations between these steps are the MFMA
// kernels of mlp_kernels.hip; everything here is one thread per ray over <= 128 samples, fp32, reference rounding
// (-ffp-contract=off).
#include "../../include/robir_hip.h"
#include "common.h"

namespace rb {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// pts[r,i,:] = o[r] + d[r] * z[r,i]     (optionally also the per-sample copy of d)
__global__ void k_ray_points(const float* __restrict__ o, const float* __restrict__ d, const float* __restrict__ z,
                             long R, int n, float* __restrict__ pts, float* __restrict__ dirs) {
  const long j = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (j >= R * n) return;
  const long r = j / n;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    pts[3 * j + c] = o[3 * r + c] + d[3 * r + c] * z[j];
    if (dirs) dirs[3 * j + c] = d[3 * r + c];
  }
}

// z[r,i] = near[r] + (far[r] - near[r]) * lin[i]           (sdf_render.py:279-283)
__global__ void k_coarse_z(const float* __restrict__ near, const float* __restrict__ far, const float* __restrict__ lin,
                           long R, int n, float* __restrict__ z) {
  const long j = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (j >= R * n) return;
  const long r = j / n;
  z[j] = near[r] + (far[r] - near[r]) * lin[j % n];
}

// perturb > 0 (sdf_render.py:293-295): ONE uniform draw per ray shifts all of its coarse samples,
// z[r,i] += (u[r] - 0.5) * 2.0 / n  -- the reference's operation order ((t * 2.0) / n), -ffp-contract=off
__global__ void k_jitter_z(const float* __restrict__ u, long R, int n, float* __restrict__ z) {
  const long j = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (j >= R * n) return;
  const float t = u[j / n] - 0.5f;
  z[j] = z[j] + (t * 2.0f) / (float)n;
}

// up_sample + sample_pdf(det=True) (sdf_render.py:70-114, 37-67): n_new importance samples per ray.
// wtmp[R, n] is scratch for the interval weights.
__global__ void k_upsample(const float* __restrict__ o, const float* __restrict__ d, const float* __restrict__ z,
                           const float* __restrict__ sdf, long R, int n, int n_new, float inv_s, float radius,
                           const float* __restrict__ u, float* __restrict__ wtmp, float* __restrict__ z_new) {
  const long r = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (r >= R) return;
  const float* zz = z + r * n;
  const float* ss = sdf + r * n;
  float* w = wtmp + r * n;
  const float ox = o[3 * r], oy = o[3 * r + 1], oz = o[3 * r + 2];
  const float dx = d[3 * r], dy = d[3 * r + 1], dz = d[3 * r + 2];
  auto rad = [&](float t) {
    const float px = ox + dx * t, py = oy + dy * t, pz = oz + dz * t;
    return sqrtf(px * px + py * py + pz * pz);
  };
  float prev_cos = 0.f, T = 1.f, wsum = 0.f;
  float r0 = rad(zz[0]);
  for (int i = 0; i + 1 < n; ++i) {
    const float r1 = rad(zz[i + 1]);
    const float inside = (r0 < radius || r1 < radius) ? 1.f : 0.f;
    const float s0 = ss[i], s1 = ss[i + 1], z0 = zz[i], z1 = zz[i + 1];
    const float mid = (s0 + s1) * 0.5f;
    const float cosv = (s1 - s0) / (z1 - z0 + 1e-5f);
    float c = fminf(prev_cos, cosv);
    prev_cos = cosv;
    c = fminf(fmaxf(c, -1e3f), 0.f) * inside;
    const float dist = z1 - z0;
    const float c0 = sigmoidf_((mid - c * dist * 0.5f) * inv_s);
    const float c1 = sigmoidf_((mid + c * dist * 0.5f) * inv_s);
    const float alpha = (c0 - c1 + 1e-5f) / (c0 + 1e-5f);
    const float wi = alpha * T + 1e-5f;                 // weights + 1e-5 (sample_pdf)
    T = T * (1.f - alpha + 1e-7f);
    w[i] = wi;
    wsum += wi;
    r0 = r1;
  }
  // inverse CDF: cdf[0] = 0, cdf[i+1] = cdf[i] + w[i]/wsum; u ascending -> single sweep
  int k = 0;
  float cdf_lo = 0.f;          // cdf[idx-1]
  int idx = 1;                 // searchsorted(right=True): first idx with cdf[idx] > u
  float cdf_hi = w[0] / wsum;  // cdf[1]
  float acc = cdf_hi;
  while (k < n_new) {
    const float uk = u[k];
    while (idx < n && !(cdf_hi > uk)) {
      cdf_lo = cdf_hi;
      ++idx;
      if (idx < n) {
        acc = acc + w[idx - 1] / wsum;
        cdf_hi = acc;
      }
    }
    // idx in [1, n]; below = idx-1, above = min(idx, n-1)
    const int below = idx - 1;
    const int above = idx < n ? idx : n - 1;
    const float c_below = cdf_lo;
    const float c_above = idx < n ? cdf_hi : cdf_lo;
    float den = c_above - c_below;
    if (den < 1e-5f) den = 1.f;
    const float t = (uk - c_below) / den;
    z_new[r * n_new + k] = zz[below] + t * (zz[above] - zz[below]);
    ++k;
  }
}

// cat_z_vals (sdf_render.py:117-132): merge two ascending lists; sdf follows z (sdf_new may be NULL on the last step)
__global__ void k_merge_z(const float* __restrict__ z_old, const float* __restrict__ sdf_old, int n,
                          const float* __restrict__ z_new, const float* __restrict__ sdf_new, int m, long R,
                          float* __restrict__ z_out, float* __restrict__ sdf_out) {
  const long r = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (r >= R) return;
  const float* a = z_old + r * n;
  const float* b = z_new + r * m;
  int i = 0, j = 0;
  for (int k = 0; k < n + m; ++k) {
    const bool take_a = j >= m || (i < n && a[i] <= b[j]);
    if (take_a) {
      z_out[r * (n + m) + k] = a[i];
      if (sdf_out) sdf_out[r * (n + m) + k] = sdf_old[r * n + i];
      ++i;
    } else {
      z_out[r * (n + m) + k] = b[j];
      if (sdf_out) sdf_out[r * (n + m) + k] = sdf_new ? sdf_new[r * m + j] : 0.f;
      ++j;
    }
  }
}

// render_core set-up (sdf_render.py:186-196): zmid = z + dz/2 with the last dz = sample_dist
__global__ void k_mid_z(const float* __restrict__ z, long R, int n, float sample_dist, float* __restrict__ zmid) {
  const long j = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (j >= R * n) return;
  const int i = (int)(j % n);
  const float dz = i + 1 < n ? z[j + 1] - z[j] : sample_dist;
  zmid[j] = z[j] + dz * 0.5f;
}

// Stage-2 weights of render_core from the mid-point SDFs alone (sdf_render.py:203-237: alpha from consecutive SDFs, inside-sphere
// mask, transmittance product) -- the SAME expressions k_neus_finish evaluates, so `weights` and `keep` (w != 0) are bit-identical
// to what the full pass produces.  Lets the caller skip gradient + colour for samples whose weight is exactly zero: behind the
// first few opaque samples the transmittance underflows to 0 (trained sharpness: inv_s in the hundreds), and 0 * colour adds
// nothing to any output.  keep[j] = 1 if weights[j] != 0; count += number kept.
__global__ void k_neus_weights(const float* __restrict__ sdf, const float* __restrict__ pts, long R, int n, float inv_s,
                               float radius, float* __restrict__ weights, unsigned char* __restrict__ keep,
                               unsigned long long* __restrict__ count) {
  const long r = blockIdx.x * (long)blockDim.x + threadIdx.x;
  int kept = 0;
  if (r < R) {
    float T = 1.f;
    for (int k = 0; k < n; ++k) {
      const long j = r * n + k;
      const float s0 = sdf[j];
      const float s1 = sdf[k + 1 < n ? j + 1 : j];
      const float c0 = sigmoidf_(s0 * inv_s), c1 = sigmoidf_(s1 * inv_s);
      float a = ((c0 - c1) + 1e-5f) / (c0 + 1e-5f);
      a = fminf(fmaxf(a, 0.f), 1.f);
      const float px = pts[3 * j], py = pts[3 * j + 1], pz = pts[3 * j + 2];
      const float pn = sqrtf(px * px + py * py + pz * pz);
      a = a * (pn < radius ? 1.f : 0.f);
      const float w = a * T;
      T = T * (1.f - a + 1e-7f);
      weights[j] = w;
      const bool k1 = w != 0.f;        // NaN weights are kept (they must reach the outputs like in the reference)
      keep[j] = k1 ? 1 : 0;
      kept += k1 ? 1 : 0;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) kept += __shfl_xor(kept, o);
  if ((threadIdx.x & 63) == 0 && kept) atomicAdd(count, (unsigned long long)kept);
}

// render_core compositing + render_neus epilogue (sdf_render.py:203-260, 354-374).
// sdf: column 0 of an [M, sdf_stride] matrix.  gerr[2] += (sum relax*(|g|-1)^2, sum relax).
__global__ void k_neus_finish(const float* __restrict__ sdf, long sdf_stride, const float* __restrict__ color,
                              const float* __restrict__ grad, const float* __restrict__ pts,
                              const float* __restrict__ zmid, const float* __restrict__ near,
                              const float* __restrict__ far, long R, int n, float inv_s, float radius, int white,
                              const float* __restrict__ z, const float* __restrict__ rays_d, float sample_dist,
                              float cos_anneal, float* __restrict__ rgb, float* __restrict__ dist, float* __restrict__ acc_out,
                              float* __restrict__ normal, float* __restrict__ weights, float* __restrict__ gerr) {
  const long r = blockIdx.x * (long)blockDim.x + threadIdx.x;
  float e_num = 0.f, e_den = 0.f;
  if (r < R) {
    float T = 1.f, acc = 0.f, col[3] = {0.f, 0.f, 0.f}, nr[3] = {0.f, 0.f, 0.f}, dsum = 0.f;
    for (int k = 0; k < n; ++k) {
      const long j = r * n + k;
      float s0 = sdf[j * sdf_stride];
      float s1 = sdf[(k + 1 < n ? j + 1 : j) * sdf_stride];
      if (z) {
        // stage-1 render_core (neus/volume_render/sdf_render.py:172-190): the SDF of this sample extrapolated half a
        // section back / forth with the annealed cosine between ray and gradient
        const float dz = k + 1 < n ? z[j + 1] - z[j] : sample_dist;
        const float tc = (rays_d[3 * r] * grad[3 * j] + rays_d[3 * r + 1] * grad[3 * j + 1]) + rays_d[3 * r + 2] * grad[3 * j + 2];
        const float ic = -(fmaxf(-tc * 0.5f + 0.5f, 0.f) * (1.0f - cos_anneal) + fmaxf(-tc, 0.f) * cos_anneal);
        const float h = ic * dz * 0.5f;
        s1 = s0 + h;
        s0 = s0 - h;
      }
      const float c0 = sigmoidf_(s0 * inv_s), c1 = sigmoidf_(s1 * inv_s);
      float a = ((c0 - c1) + 1e-5f) / (c0 + 1e-5f);
      a = fminf(fmaxf(a, 0.f), 1.f);
      const float px = pts[3 * j], py = pts[3 * j + 1], pz = pts[3 * j + 2];
      const float pn = sqrtf(px * px + py * py + pz * pz);
      a = a * (pn < radius ? 1.f : 0.f);
      const float w = a * T;
      T = T * (1.f - a + 1e-7f);
      weights[j] = w;
      acc += w;
      const float gx = grad[3 * j], gy = grad[3 * j + 1], gz = grad[3 * j + 2];
#pragma unroll
      for (int c = 0; c < 3; ++c) col[c] += color[3 * j + c] * w;
      nr[0] += w * gx;
      nr[1] += w * gy;
      nr[2] += w * gz;
      dsum += w * zmid[j];
      const float relax = pn < radius * 1.2f ? 1.f : 0.f;
      const float gn = sqrtf(gx * gx + gy * gy + gz * gz) - 1.f;
      e_num += relax * (gn * gn);
      e_den += relax;
    }
    const float nn = sqrtf(nr[0] * nr[0] + nr[1] * nr[1] + nr[2] * nr[2]) + 0.0001f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      rgb[3 * r + c] = white ? col[c] + 1.f * (1.f - acc) : col[c];
      normal[3 * r + c] = acc > 0.8f ? 1.f : nr[c] / nn;          // sic: normal[acc > 0.8] = 1.0 (sdf_render.py:361)
    }
    float dd = dsum / acc;
    if (dd != dd) dd = __int_as_float(0x7f800000);                // nan_to_num(nan=inf)
    dd = fminf(fmaxf(dd, near[r]), far[r]);
    dist[r] = dd;
    acc_out[r] = acc;
  }
  // block reduction of the eikonal terms, one atomic pair per wave
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    e_num += __shfl_xor(e_num, o);
    e_den += __shfl_xor(e_den, o);
  }
  if ((threadIdx.x & 63) == 0 && gerr) {
    atomicAdd(gerr, e_num);
    atomicAdd(gerr + 1, e_den);
  }
}

// get_neus_surface (train_normal.py:239-286): xs[m*ns,3] = p - t_k * dir; then (after SDF+grad on xs)
// alpha clipped to [0.01,0.99], residual weight to the input point / normal.
__global__ void k_surface_points(const float* __restrict__ p, const float* __restrict__ dir, const float* __restrict__ tk,
                                 long m, int ns, float* __restrict__ xs) {
  const long j = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (j >= m * ns) return;
  const long i = j / ns;
  const float t = tk[j % ns];
#pragma unroll
  for (int c = 0; c < 3; ++c) xs[3 * j + c] = p[3 * i + c] - t * dir[3 * i + c];
}

__global__ void k_surface_finish(const float* __restrict__ sdf, const float* __restrict__ grad,
                                 const float* __restrict__ xs, const float* __restrict__ p,
                                 const float* __restrict__ pred_n, long m, int ns, float s, float* __restrict__ x_out,
                                 float* __restrict__ n_out, float* __restrict__ gerr) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  float e_num = 0.f, e_den = 0.f;
  if (i < m) {
    float T = 1.f, wsum = 0.f, ax[3] = {0.f, 0.f, 0.f}, an[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < ns; ++k) {
      const long j = i * ns + k;
      const float s0 = sdf[j], s1 = sdf[k + 1 < ns ? j + 1 : j];
      const float c0 = sigmoidf_(s0 * s), c1 = sigmoidf_(s1 * s);
      float a = ((c0 - c1) + 1e-5f) / (c0 + 1e-5f);
      a = fminf(fmaxf(a, 0.01f), 0.99f);
      const float w = a * T;
      T = T * (1.f - a + 1e-10f);
      wsum += w;
      float gn = 0.f, pn = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        ax[c] += xs[3 * j + c] * w;
        an[c] += grad[3 * j + c] * w;
        gn += grad[3 * j + c] * grad[3 * j + c];
        pn += xs[3 * j + c] * xs[3 * j + c];
      }
      const float relax = sqrtf(pn) < 1.2f ? 1.f : 0.f;
      const float ge = sqrtf(gn) - 1.f;
      e_num += relax * (ge * ge);
      e_den += relax;
    }
    const float res = 1.f - wsum;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      x_out[3 * i + c] = ax[c] + res * p[3 * i + c];
      n_out[3 * i + c] = an[c] + res * pred_n[3 * i + c];
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    e_num += __shfl_xor(e_num, o);
    e_den += __shfl_xor(e_den, o);
  }
  if ((threadIdx.x & 63) == 0 && gerr) {
    atomicAdd(gerr, e_num);
    atomicAdd(gerr + 1, e_den);
  }
}

}  // namespace rb

using namespace rb;

extern "C" {

int rb_ray_points(const float* o, const float* d, const float* z, long R, int n, float* pts, float* dirs,
                  rb_stream_t stream) {
  if (R <= 0) return 0;
  RB_REQUIRE(o && d && z && pts, "null pointer");
  hipLaunchKernelGGL(k_ray_points, grid1d(R * n, 256), dim3(256), 0, (hipStream_t)stream, o, d, z, R, n, pts, dirs);
  return check_launch("k_ray_points");
}

int rb_neus_coarse_z(const float* near, const float* far, const float* lin, long R, int n, float* z, rb_stream_t stream) {
  if (R <= 0) return 0;
  RB_REQUIRE(near && far && lin && z, "null pointer");
  hipLaunchKernelGGL(k_coarse_z, grid1d(R * n, 256), dim3(256), 0, (hipStream_t)stream, near, far, lin, R, n, z);
  return check_launch("k_coarse_z");
}

int rb_neus_jitter_z(const float* u, long R, int n, float* z, rb_stream_t stream) {
  if (R <= 0) return 0;
  RB_REQUIRE(u && z, "null pointer");
  RB_REQUIRE(n >= 1, "need n >= 1 samples");
  hipLaunchKernelGGL(k_jitter_z, grid1d(R * n, 256), dim3(256), 0, (hipStream_t)stream, u, R, n, z);
  return check_launch("k_jitter_z");
}

int rb_neus_upsample(const float* o, const float* d, const float* z, const float* sdf, long R, int n, int n_new,
                     float inv_s, float radius, const float* u, float* wtmp, float* z_new, rb_stream_t stream) {
  if (R <= 0) return 0;
  RB_REQUIRE(o && d && z && sdf && u && wtmp && z_new, "null pointer");
  RB_REQUIRE(n >= 2 && n_new >= 1, "need n >= 2 samples");
  hipLaunchKernelGGL(k_upsample, grid1d(R, 128), dim3(128), 0, (hipStream_t)stream, o, d, z, sdf, R, n, n_new, inv_s,
                     radius, u, wtmp, z_new);
  return check_launch("k_upsample");
}

int rb_neus_merge(const float* z_old, const float* sdf_old, int n, const float* z_new, const float* sdf_new, int m,
                  long R, float* z_out, float* sdf_out, rb_stream_t stream) {
  if (R <= 0) return 0;
  RB_REQUIRE(z_old && z_new && z_out, "null pointer");
  RB_REQUIRE(!sdf_out || sdf_old, "sdf_out needs sdf_old");
  hipLaunchKernelGGL(k_merge_z, grid1d(R, 128), dim3(128), 0, (hipStream_t)stream, z_old, sdf_old, n, z_new, sdf_new, m,
                     R, z_out, sdf_out);
  return check_launch("k_merge_z");
}

int rb_neus_mid_z(const float* z, long R, int n, float sample_dist, float* zmid, rb_stream_t stream) {
  if (R <= 0) return 0;
  RB_REQUIRE(z && zmid, "null pointer");
  hipLaunchKernelGGL(k_mid_z, grid1d(R * n, 256), dim3(256), 0, (hipStream_t)stream, z, R, n, sample_dist, zmid);
  return check_launch("k_mid_z");
}

int rb_neus_weights(const float* sdf, const float* pts, long R, int n, float inv_s, float radius, float* weights,
                    unsigned char* keep, unsigned long long* count, rb_stream_t stream) {
  if (R <= 0) return 0;
  RB_REQUIRE(sdf && pts && weights && keep && count, "null pointer");
  hipLaunchKernelGGL(k_neus_weights, grid1d(R, 128), dim3(128), 0, (hipStream_t)stream, sdf, pts, R, n, inv_s, radius, weights,
                     keep, count);
  return check_launch("k_neus_weights");
}

int rb_neus_finish(const float* sdf, long sdf_stride, const float* color, const float* grad, const float* pts,
                   const float* zmid, const float* near, const float* far, long R, int n, float inv_s, float radius,
                   int white, const float* z, const float* rays_d, float sample_dist, float cos_anneal, float* rgb,
                   float* dist, float* acc, float* normal, float* weights, float* gerr, rb_stream_t stream) {
  if (R <= 0) return 0;
  RB_REQUIRE(sdf && color && grad && pts && zmid && near && far && rgb && dist && acc && normal && weights,
             "null pointer");
  RB_REQUIRE(!z || rays_d, "the stage-1 alpha needs the ray directions");
  hipLaunchKernelGGL(k_neus_finish, grid1d(R, 128), dim3(128), 0, (hipStream_t)stream, sdf, sdf_stride, color, grad, pts,
                     zmid, near, far, R, n, inv_s, radius, white, z, rays_d, sample_dist, cos_anneal, rgb, dist, acc,
                     normal, weights, gerr);
  return check_launch("k_neus_finish");
}

int rb_surface_points(const float* p, const float* dir, const float* tk, long m, int ns, float* xs, rb_stream_t stream) {
  if (m <= 0) return 0;
  RB_REQUIRE(p && dir && tk && xs, "null pointer");
  hipLaunchKernelGGL(k_surface_points, grid1d(m * ns, 256), dim3(256), 0, (hipStream_t)stream, p, dir, tk, m, ns, xs);
  return check_launch("k_surface_points");
}

int rb_surface_finish(const float* sdf, const float* grad, const float* xs, const float* p, const float* pred_n, long m,
                      int ns, float s, float* x_out, float* n_out, float* gerr, rb_stream_t stream) {
  if (m <= 0) return 0;
  RB_REQUIRE(sdf && grad && xs && p && pred_n && x_out && n_out, "null pointer");
  hipLaunchKernelGGL(k_surface_finish, grid1d(m, 128), dim3(128), 0, (hipStream_t)stream, sdf, grad, xs, p, pred_n, m, ns,
                     s, x_out, n_out, gerr);
  return check_launch("k_surface_finish");
}

}  // extern "C"
