// IDR sphere tracer (the use_octree=False ray tracer): per-ray state updates of RayTracing.forward / sphere_tracing /
// ray_sampler / secant in eval mode (model/ray_tracing.py:26-297; utils/rend_util.py:141-163).  The SDF evaluations
// between these steps are the MFMA kernel rb_sdf_mlp (mode 0) on the 2N start/end points; rays are independent, so
// masked-out rays are simply carried along (their SDF value is ignored), which removes every data-dependent host
// branch of the reference except the global "nobody is unfinished" stop, kept as a device-side sticky flag.
#include "../../include/robir_hip.h"
#include "common.h"

namespace rb {

struct RtState {          // all arrays of N rays
  float* acc_s;           // distance of the start point (front side)
  float* acc_e;           // distance of the end point (back side)
  float* cur_s;           // sdf used for the current step
  float* cur_e;
  float* nxt_s;           // sdf at the current points
  float* nxt_e;
  unsigned char* un_s;    // unfinished masks
  unsigned char* un_e;
  unsigned char* bad_s;   // "crossed the surface" flags of the line search
  unsigned char* bad_e;
  float* pts;             // [2N,3]: start points then end points
  int* ctrl;              // [0] = number unfinished at the last loop top, [1] = sticky stop flag
};

// get_sphere_intersection + initialisation of sphere_tracing (ray_tracing.py:105-126)
__global__ void k_rt_init(const float* __restrict__ cam, int cs, const float* __restrict__ dirs, long N, float r2,
                          RtState s) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= N) return;
  const float cx = cam[cs * i], cy = cam[cs * i + 1], cz = cam[cs * i + 2];
  const float dx = dirs[3 * i], dy = dirs[3 * i + 1], dz = dirs[3 * i + 2];
  const float dot = dx * cx + dy * cy + dz * cz;
  const float cn = sqrtf(cx * cx + cy * cy + cz * cz);
  const float under = dot * dot - (cn * cn - r2);
  const bool hit = under > 0.f;
  float t0 = 0.f, t1 = 0.f;
  if (hit) {
    const float root = sqrtf(under);
    t0 = -root - dot;
    t1 = root - dot;
  }
  t0 = fmaxf(t0, 0.01f);   // clamp_min applies to every entry, also to the zeros of non-intersecting rays
  t1 = fmaxf(t1, 0.01f);
  s.acc_s[i] = hit ? t0 : 0.f;
  s.acc_e[i] = hit ? t1 : 0.f;
  s.un_s[i] = hit;
  s.un_e[i] = hit;
  const float cc[3] = {cx, cy, cz}, dd[3] = {dx, dy, dz};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    s.pts[3 * i + c] = hit ? cc[c] + t0 * dd[c] : 0.f;
    s.pts[3 * (N + i) + c] = hit ? cc[c] + t1 * dd[c] : 0.f;
  }
  s.nxt_s[i] = 0.f;
  s.nxt_e[i] = 0.f;
  s.bad_s[i] = 0;
  s.bad_e[i] = 0;
}

// next_sdf[mask] = sdf(points[mask]) for the rows selected by `which`: 0 = unfinished masks, 1 = line-search flags;
// afterwards bad = next < 0 (ray_tracing.py:131-135, 169-176, 195-200)
__global__ void k_rt_take_sdf(const float* __restrict__ sdf2, long N, int which, RtState s) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= N) return;
  if (which == 0) {
    s.nxt_s[i] = s.un_s[i] ? sdf2[i] : 0.f;
    s.nxt_e[i] = s.un_e[i] ? sdf2[N + i] : 0.f;
  } else {
    if (s.bad_s[i]) s.nxt_s[i] = sdf2[i];
    if (s.bad_e[i]) s.nxt_e[i] = sdf2[N + i];
  }
  s.bad_s[i] = s.nxt_s[i] < 0.f;
  s.bad_e[i] = s.nxt_e[i] < 0.f;
}

// loop top (ray_tracing.py:138-149): current step sizes, thresholding, mask update, count of unfinished rays
__global__ void k_rt_top(long N, float thr, RtState s) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  bool any = false;
  if (i < N && !s.ctrl[1]) {
    float cs = s.un_s[i] ? s.nxt_s[i] : 0.f;
    if (cs <= thr) cs = 0.f;
    float ce = s.un_e[i] ? s.nxt_e[i] : 0.f;
    if (ce <= thr) ce = 0.f;
    s.cur_s[i] = cs;
    s.cur_e[i] = ce;
    const bool us = s.un_s[i] && cs > thr, ue = s.un_e[i] && ce > thr;
    s.un_s[i] = us;
    s.un_e[i] = ue;
    any = us || ue;
  }
  const unsigned long long m = __ballot(any);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(&s.ctrl[0], __popcll(m));
}

// step (ray_tracing.py:156-166) unless the loop has stopped (nobody unfinished at this or an earlier top)
__global__ void k_rt_step(const float* __restrict__ cam, int cs, const float* __restrict__ dirs, long N, RtState s) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= N) return;
  if (s.ctrl[1] || s.ctrl[0] == 0) return;
  const float as = s.acc_s[i] + s.cur_s[i], ae = s.acc_e[i] - s.cur_e[i];
  s.acc_s[i] = as;
  s.acc_e[i] = ae;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    s.pts[3 * i + c] = cam[cs * i + c] + as * dirs[3 * i + c];
    s.pts[3 * (N + i) + c] = cam[cs * i + c] + ae * dirs[3 * i + c];
  }
}
// bookkeeping between top and step: latch the stop flag, reset the counter (one thread)
__global__ void k_rt_latch(RtState s, int phase) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    if (phase == 0) {
      if (s.ctrl[0] == 0) s.ctrl[1] = 1;
    } else {
      s.ctrl[0] = 0;
    }
  }
}

// one line-search back-step k (ray_tracing.py:178-192)
__global__ void k_rt_backstep(const float* __restrict__ cam, int cs, const float* __restrict__ dirs, long N, float factor,
                              RtState s) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= N || s.ctrl[1]) return;
  if (s.bad_s[i]) {
    const float a = s.acc_s[i] - factor * s.cur_s[i];
    s.acc_s[i] = a;
#pragma unroll
    for (int c = 0; c < 3; ++c) s.pts[3 * i + c] = cam[cs * i + c] + a * dirs[3 * i + c];
  }
  if (s.bad_e[i]) {
    const float a = s.acc_e[i] + factor * s.cur_e[i];
    s.acc_e[i] = a;
#pragma unroll
    for (int c = 0; c < 3; ++c) s.pts[3 * (N + i) + c] = cam[cs * i + c] + a * dirs[3 * i + c];
  }
}

// end of an iteration (ray_tracing.py:203-204)
__global__ void k_rt_close(long N, RtState s) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= N || s.ctrl[1]) return;
  const bool ok = s.acc_s[i] < s.acc_e[i];
  s.un_s[i] = s.un_s[i] && ok;
  s.un_e[i] = s.un_e[i] && ok;
}

// ---- sampler on the m unfinished rays (ray_tracing.py:208-274, eval mode)
// sample points: z[j,k] = lo + lin[k]*(hi-lo); P = cam + z*d
__global__ void k_rt_samples(const float* __restrict__ cam, int cs, const float* __restrict__ dirs, const float* __restrict__ lo,
                             const float* __restrict__ hi, const float* __restrict__ lin, long m, int n_steps,
                             float* __restrict__ z, float* __restrict__ P) {
  long j = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (j >= m * n_steps) return;
  const long r = j / n_steps;
  const float zz = lo[r] + lin[j % n_steps] * (hi[r] - lo[r]);
  z[j] = zz;
#pragma unroll
  for (int c = 0; c < 3; ++c) P[3 * j + c] = cam[cs * r + c] + zz * dirs[3 * r + c];
}
// per ray: first negative sample (argmin of sign(sdf)*(n-k)), minimal-SDF fallback, secant bracket
__global__ void k_rt_pick(const float* __restrict__ sdf, const float* __restrict__ z, const float* __restrict__ P,
                          const unsigned char* __restrict__ obj, long m, int n, float* __restrict__ out_pts,
                          float* __restrict__ out_dist, unsigned char* __restrict__ out_hit, float* __restrict__ zlo,
                          float* __restrict__ zhi, float* __restrict__ slo, float* __restrict__ shi) {
  long r = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (r >= m) return;
  const float* s = sdf + r * n;
  int first = 0, jmin = 0;
  float best = 3.0e38f, smin = 3.0e38f;
  for (int k = 0; k < n; ++k) {
    const float v = s[k];
    const float sg = v > 0.f ? 1.f : (v < 0.f ? -1.f : (v == 0.f ? 0.f : v));   // torch.sign (NaN stays NaN)
    const float w = sg * (float)(n - k);
    if (w < best) {
      best = w;
      first = k;
    }
    if (v < smin) {
      smin = v;
      jmin = k;
    }
  }
  const bool neg = s[first] < 0.f;
  int pick = first;
  if (!(obj[r] && neg)) pick = jmin;
#pragma unroll
  for (int c = 0; c < 3; ++c) out_pts[3 * r + c] = P[3 * (r * n + pick) + c];
  out_dist[r] = z[r * n + pick];
  out_hit[r] = neg;
  const int prev = first > 0 ? first - 1 : n - 1;        // index -1 wraps to the last sample in the reference
  zhi[r] = z[r * n + first];
  shi[r] = s[first];
  zlo[r] = z[r * n + prev];
  slo[r] = s[prev];
}
// secant (ray_tracing.py:276-297): phase 0 = initial prediction, phase 1 = bracket update with sdf_mid + new prediction.
// Only rays with out_hit (first sample negative) are refined; zp / points of the others are left untouched.
__global__ void k_rt_secant(const float* __restrict__ cam, int cs, const float* __restrict__ dirs,
                            const unsigned char* __restrict__ on,
                            const float* __restrict__ smid, long m, int phase, float* __restrict__ zlo,
                            float* __restrict__ zhi, float* __restrict__ slo, float* __restrict__ shi,
                            float* __restrict__ zp, float* __restrict__ pmid) {
  long r = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (r >= m || !on[r]) return;
  if (phase == 1) {
    const float sm = smid[r];
    if (sm > 0.f) {
      zlo[r] = zp[r];
      slo[r] = sm;
    }
    if (sm < 0.f) {
      zhi[r] = zp[r];
      shi[r] = sm;
    }
  }
  float z = -slo[r] * (zhi[r] - zlo[r]) / (shi[r] - slo[r] + 1e-8f) + zlo[r];
  z = fminf(fmaxf(z, 0.f), 2e1f);
  zp[r] = z;
#pragma unroll
  for (int c = 0; c < 3; ++c) pmid[3 * r + c] = cam[cs * r + c] + z * dirs[3 * r + c];
}

}  // namespace rb

using namespace rb;

extern "C" {

static RtState make_state(float* fl, unsigned char* by, float* pts, int* ctrl, long N) {
  RtState s;
  s.acc_s = fl;
  s.acc_e = fl + N;
  s.cur_s = fl + 2 * N;
  s.cur_e = fl + 3 * N;
  s.nxt_s = fl + 4 * N;
  s.nxt_e = fl + 5 * N;
  s.un_s = by;
  s.un_e = by + N;
  s.bad_s = by + 2 * N;
  s.bad_e = by + 3 * N;
  s.pts = pts;
  s.ctrl = ctrl;
  return s;
}

// op: 0 init (param = r^2) | 1 take sdf (unfinished rows) | 2 take sdf (line-search rows) | 3 loop top (+latch) | 4 step |
//     5 back-step (factor) | 6 close iteration
int rb_raytrace_step(int op, const float* cam, int cam_stride, const float* dirs, long N, float param, const float* sdf2, float* state_f,
                     unsigned char* state_b, float* pts, int* ctrl, rb_stream_t stream) {
  if (N <= 0) return 0;
  RB_REQUIRE(cam && dirs && state_f && state_b && pts && ctrl, "null pointer");
  RB_REQUIRE(cam_stride == 0 || cam_stride == 3, "cam_stride must be 0 (one camera) or 3 (one origin per ray)");
  hipStream_t st = (hipStream_t)stream;
  RtState s = make_state(state_f, state_b, pts, ctrl, N);
  dim3 g = grid1d(N, 256), b(256);
  switch (op) {
    case 0: hipLaunchKernelGGL(k_rt_init, g, b, 0, st, cam, cam_stride, dirs, N, param, s); break;
    case 1:
    case 2:
      RB_REQUIRE(sdf2, "sdf values needed");
      hipLaunchKernelGGL(k_rt_take_sdf, g, b, 0, st, sdf2, N, op - 1, s);
      break;
    case 3:
      hipLaunchKernelGGL(k_rt_latch, dim3(1), dim3(64), 0, st, s, 1);
      hipLaunchKernelGGL(k_rt_top, g, b, 0, st, N, param, s);
      hipLaunchKernelGGL(k_rt_latch, dim3(1), dim3(64), 0, st, s, 0);
      break;
    case 4: hipLaunchKernelGGL(k_rt_step, g, b, 0, st, cam, cam_stride, dirs, N, s); break;
    case 5: hipLaunchKernelGGL(k_rt_backstep, g, b, 0, st, cam, cam_stride, dirs, N, param, s); break;
    case 6: hipLaunchKernelGGL(k_rt_close, g, b, 0, st, N, s); break;
    default: return rb::fail("rb_raytrace_step", "op must be 0..6");
  }
  return check_launch("k_rt_*");
}

int rb_raytrace_samples(const float* cam, int cam_stride, const float* dirs, const float* lo, const float* hi, const float* lin, long m,
                        int n_steps, float* z, float* P, rb_stream_t stream) {
  if (m <= 0) return 0;
  RB_REQUIRE(cam && dirs && lo && hi && lin && z && P, "null pointer");
  hipLaunchKernelGGL(k_rt_samples, grid1d(m * n_steps, 256), dim3(256), 0, (hipStream_t)stream, cam, cam_stride, dirs, lo, hi,
                     lin, m,
                     n_steps, z, P);
  return check_launch("k_rt_samples");
}

int rb_raytrace_pick(const float* sdf, const float* z, const float* P, const unsigned char* obj, long m, int n,
                     float* out_pts, float* out_dist, unsigned char* out_hit, float* bracket, rb_stream_t stream) {
  if (m <= 0) return 0;
  RB_REQUIRE(sdf && z && P && obj && out_pts && out_dist && out_hit && bracket, "null pointer");
  hipLaunchKernelGGL(k_rt_pick, grid1d(m, 128), dim3(128), 0, (hipStream_t)stream, sdf, z, P, obj, m, n, out_pts, out_dist,
                     out_hit, bracket, bracket + m, bracket + 2 * m, bracket + 3 * m);
  return check_launch("k_rt_pick");
}

int rb_raytrace_secant(const float* cam, int cam_stride, const float* dirs, const unsigned char* on, const float* smid, long m, int phase,
                       float* bracket, float* zp, float* pmid, rb_stream_t stream) {
  if (m <= 0) return 0;
  RB_REQUIRE(cam && dirs && on && bracket && zp && pmid && (phase == 0 || smid), "null pointer");
  hipLaunchKernelGGL(k_rt_secant, grid1d(m, 128), dim3(128), 0, (hipStream_t)stream, cam, cam_stride, dirs, on, smid,
                     m, phase,
                     bracket, bracket + m, bracket + 2 * m, bracket + 3 * m, zp, pmid);
  return check_launch("k_rt_secant");
}

}  // extern "C"
