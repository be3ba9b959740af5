// Ray generation and tone mapping (element-wise edges of the per-ray path).
// utils/rend_util.py:51-97 (get_camera_params / lift, 4x4 pose branch); model/color_correction.py:31-60,116-137.
#include "../../include/robir_hip.h"
#include "common.h"

namespace rb {

struct Cam {
  float p[12];            // rows 0..2 of the camera-to-world matrix
  float fx, fy, cx, cy, sk;
};

__global__ void k_camera_rays(Cam c, const float* __restrict__ uv, long N, float* __restrict__ dirs) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= N) return;
  const float x = uv[2 * i], y = uv[2 * i + 1], z = 1.0f;
  const float xl = (x - c.cx + c.cy * c.sk / c.fy - c.sk * y / c.fy) / c.fx * z;
  const float yl = (y - c.cy) / c.fy * z;
  const float pc[4] = {xl, -yl, -z, 1.0f};
  float w[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    float acc = c.p[4 * r] * pc[0];
    acc = acc + c.p[4 * r + 1] * pc[1];
    acc = acc + c.p[4 * r + 2] * pc[2];
    acc = acc + c.p[4 * r + 3] * pc[3];
    w[r] = acc - c.p[4 * r + 3];        // world - cam_loc
  }
  const float n = fmaxf(sqrtf(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]), 1e-12f);   // F.normalize
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
// mode 0: hdr2ldr = aces(x)/t^0.2 ; mode 1: ldr2hdr = aces^-1(x * t^0.2) ; mode 2: ldr2hdr(x^2.2) (trace_radiance)
// shift: [n] per row (rows of 3 channels) or a single value (shift_stride 0); clamped to [1e-4, 1]
__global__ void k_tonemap(const float* __restrict__ x, long n, const float* __restrict__ shift, int shift_stride,
                          int mode, float* __restrict__ y) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= 3 * n) return;
  float t = shift[(i / 3) * shift_stride];
  t = fminf(fmaxf(t, 1e-4f), 1.f);
  const float tp = powf(t, 0.2f);
  float v = x[i];
  if (mode == 0) {
    y[i] = aces(v) / tp;
  } else {
    if (mode == 2) v = powf(v, 2.2f);
    y[i] = aces_inv(v * tp);
  }
}

}  // namespace rb

using namespace rb;

extern "C" {

int rb_camera_rays(const float* pose_host, const float* K_host, const float* uv, long N, float* dirs,
                   rb_stream_t stream) {
  if (N <= 0) return 0;
  RB_REQUIRE(pose_host && K_host && uv && dirs, "null pointer");
  Cam c;
  for (int i = 0; i < 12; ++i) c.p[i] = pose_host[i];
  c.fx = K_host[0];
  c.sk = K_host[1];
  c.cx = K_host[2];
  c.fy = K_host[4];
  c.cy = K_host[5];
  hipLaunchKernelGGL(k_camera_rays, grid1d(N, 256), dim3(256), 0, (hipStream_t)stream, c, uv, N, dirs);
  return check_launch("k_camera_rays");
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

int rb_tonemap(const float* x, long n, const float* shift, int shift_stride, int mode, float* y, rb_stream_t stream) {
  if (n <= 0) return 0;
  RB_REQUIRE(x && shift && y, "null pointer");
  RB_REQUIRE(mode >= 0 && mode <= 2, "mode must be 0..2");
  hipLaunchKernelGGL(k_tonemap, grid1d(3 * n, 256), dim3(256), 0, (hipStream_t)stream, x, n, shift, shift_stride, mode, y);
  return check_launch("k_tonemap");
}

}  // extern "C"
