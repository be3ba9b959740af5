// Octree over the SDF zero set and the lock-step sphere tracer.
// Reference: utils/octree.py:19-57 (box helpers), :124-199 (build), :217-265 (query), :377-438 (OctreeSDF, cast),
// :459-471 (fast_volume_render), :493-585 (multi_step_cast), :588-592 (first_nonzero / torch_scatter.scatter_min);
// model/octree_tracing.py:31-60.
//
// Device layout (44 B/node instead of the reference's 125 B):
//   node[B][2] float4:  {min.x, min.y, min.z, bits(first_child:int32, -1 = leaf)}, {size.x, size.y, size.z, sdf_val}
//   nrm[B][3]  float :  unit SDF gradient at the box centre
// The 8 children of a split node are contiguous (first_child + 4*ox + 2*oy + oz), the base grid is row-major
// (ix*ny + iy)*nz + iz, "hit" cells are sdf_val <= 1e-4 (== relu(sdf_val) <= 1e-4).
//
// The reference advances all rays of a batch in lock step and derives the fine-march sample count from the number
// of still-active rays (octree.py:545-549), so results depend on the batch: a batch here is one workgroup when it
// has <= 1024 rays (one launch renders many 1024-pixel chunks, the per-iteration count is a workgroup reduction),
// otherwise one launch per iteration with device-side counters.  Geometry arithmetic keeps the reference's rounding
// (-ffp-contract=off, IEEE 1/d, NaN-propagating min/max like torch.minimum/maximum).
#include "../../include/robir_hip.h"
#include "common.h"

namespace rb {

typedef float f4 __attribute__((ext_vector_type(4)));

struct Root {
  float mn[3], sz[3];
  int res[3];
};

struct Oct {
  const f4* node;
  const float* nrm;
  long B;
  Root root;
};

__device__ __forceinline__ float tmin(float a, float b) { return (a != a || b != b) ? __int_as_float(0x7fc00000) : fminf(a, b); }
__device__ __forceinline__ float tmax(float a, float b) { return (a != a || b != b) ? __int_as_float(0x7fc00000) : fmaxf(a, b); }

// strictly inside the root box (inside_box(exactly=True), octree.py:19-29)
__device__ __forceinline__ bool in_root(const Root& r, const float x[3]) {
  bool ok = true;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float rel = (x[c] - r.mn[c]) / r.sz[c];
    ok = ok && (rel < 1.f) && (rel > 0.f);
  }
  return ok;
}

// first-level cell of a point strictly inside the root
__device__ __forceinline__ int base_cell(const Oct& T, const float x[3]) {
  int ci[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) ci[c] = (int)floorf(((x[c] - T.root.mn[c]) / T.root.sz[c]) * (float)T.root.res[c]);
  return (ci[0] * T.root.res[1] + ci[1]) * T.root.res[2] + ci[2];
}
// child of a split node that contains x: truncation toward zero, then clip (octree.py:32-38)
__device__ __forceinline__ int child_of(const f4& a, const f4& b, int fc, const float x[3]) {
  int o[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int v = (int)(((x[c] - a[c]) / b[c]) * 2.f);
    o[c] = v < 0 ? 0 : (v > 1 ? 1 : v);
  }
  return fc + 4 * o[0] + 2 * o[1] + o[2];
}
// Octree.query for a point known to be strictly inside the root (octree.py:231-262); also returns the leaf's cached SDF.
// Both halves of a node are fetched together: one memory latency per level.
__device__ __forceinline__ int descend(const Oct& T, const float x[3], float& leaf_sdf) {
  int ptr = base_cell(T, x);
  while (true) {
    const f4 a = T.node[2 * (long)ptr], b = T.node[2 * (long)ptr + 1];
    const int fc = __float_as_int(a[3]);
    if (fc < 0) {
      leaf_sdf = b[3];
      break;
    }
    ptr = child_of(a, b, fc, x);
  }
  return ptr;
}
__device__ __forceinline__ int descend(const Oct& T, const float x[3]) {
  float sv;
  return descend(T, x, sv);
}

__device__ __forceinline__ int locate(const Oct& T, const float x[3]) { return in_root(T.root, x) ? descend(T, x) : -1; }

// intersect_box (octree.py:41-57): returns far; near/valid through references
__device__ __forceinline__ float slab(const float mn[3], const float sz[3], const float o[3], const float d[3],
                                      float& near_out) {
  float near = 0.f, far = 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float inv = 1.0f / d[c];
    const float ta = (mn[c] - o[c]) * inv;
    const float tb = (sz[c] + mn[c] - o[c]) * inv;
    const float t1 = tmin(ta, tb), t2 = tmax(ta, tb);
    near = c == 0 ? t1 : tmax(near, t1);
    far = c == 0 ? t2 : tmin(far, t2);
  }
  near_out = near;
  return far;
}

// torch.linspace(0, 1, m+1)[i] (symmetric evaluation of the CPU/CUDA kernels)
__device__ __forceinline__ float lin01(int i, int m) {
  const float s = 1.0f / (float)m;
  const int steps = m + 1;
  return i < steps / 2 ? 0.f + s * (float)i : 1.f - s * (float)(steps - i - 1);
}

struct RayState {
  float t;
  int leaf;
  bool active;
};

// ray set-up (octree.py:504-519)
__device__ __forceinline__ RayState cast_init(const Oct& T, const float o[3], const float d[3]) {
  RayState s;
  float near;
  const float far = slab(T.root.mn, T.root.sz, o, d, near);
  const bool ok = (near <= far) && (far >= 0.f);
  near = tmax(near, 0.f);
  s.t = ok ? near + 1e-3f : -1.f;
  s.leaf = -1;
  if (ok) {
    const float pos[3] = {o[0] + s.t * d[0], o[1] + s.t * d[1], o[2] + s.t * d[2]};
    s.leaf = locate(T, pos);
  }
  s.active = s.leaf >= 0;
  return s;
}

// one lock-step iteration for one active ray (octree.py:528-573)
__device__ __forceinline__ void cast_step(const Oct& T, const float o[3], const float d[3], RayState& s, int m,
                                          double step) {
  float pos[3] = {o[0] + s.t * d[0], o[1] + s.t * d[1], o[2] + s.t * d[2]};
  const f4 a = T.node[2 * (long)s.leaf], b = T.node[2 * (long)s.leaf + 1];
  const float mn[3] = {a[0], a[1], a[2]}, sz[3] = {b[0], b[1], b[2]};
  float near;
  float far = slab(mn, sz, pos, d, near);
  if (far < (float)((double)m * step)) {   // python: far < multi_samp * step_size (double product, cast to fp32)
    // fine march on the cached cell SDF: sample i at t_(i+1), stop one step before the first cell with sdf <= step.
    // FB samples descend the tree together (branch-free, so their node reads are in flight at the same time): the walk
    // is a chain of dependent L2 reads per sample, and a lock-step batch of 1024 rays has no other work to hide it.
    const float stepf = (float)step;
    constexpr int FB = 4;
    int j = m;
    for (int i0 = 0; i0 < m && j == m; i0 += FB) {
      float q[FB][3], sv[FB];
      int ptr[FB];
      bool done[FB];
#pragma unroll
      for (int k = 0; k < FB; ++k) {
        const float tm = lin01(i0 + k + 1, m) * (float)m * stepf + stepf;
#pragma unroll
        for (int c = 0; c < 3; ++c) q[k][c] = pos[c] + d[c] * tm;
        const bool inside = (i0 + k < m) && in_root(T.root, q[k]);
        ptr[k] = inside ? base_cell(T, q[k]) : (int)(T.B - 1);   // outside: sdf_val[-1] (octree.py:465-466)
        done[k] = !inside;
        sv[k] = 0.f;
      }
      bool any = true;
      while (any) {
        f4 a[FB], b[FB];
#pragma unroll
        for (int k = 0; k < FB; ++k) {
          a[k] = T.node[2 * (long)ptr[k]];
          b[k] = T.node[2 * (long)ptr[k] + 1];
        }
        any = false;
#pragma unroll
        for (int k = 0; k < FB; ++k) {
          const int fc = __float_as_int(a[k][3]);
          const bool leaf = done[k] || fc < 0;
          sv[k] = b[k][3];
          const int nxt = child_of(a[k], b[k], fc, q[k]);
          ptr[k] = leaf ? ptr[k] : nxt;
          done[k] = leaf;
          any = any || !leaf;
        }
      }
#pragma unroll
      for (int k = FB - 1; k >= 0; --k)
        if (i0 + k < m && sv[k] <= stepf) j = i0 + k;
    }
    far = lin01(j, m) * (float)m * stepf + stepf;
  }
  s.t = s.t + (far + 1e-3f);
  pos[0] = o[0] + s.t * d[0];
  pos[1] = o[1] + s.t * d[1];
  pos[2] = o[2] + s.t * d[2];
  if (!in_root(T.root, pos)) {
    s.leaf = -1;
    s.active = false;
  } else {
    float sv;
    s.leaf = descend(T, pos, sv);
    s.active = !(sv <= 1e-4f);
  }
}

// ---- pieces of cast_step for the workgroup-cooperative form of k_cast_batched
// first half: exit distance of the current cell; `need` = the fine march applies (octree.py:540-546)
__device__ __forceinline__ float step_begin(const Oct& T, const float o[3], const float d[3], const RayState& s, int m,
                                            double step, float pos[3], bool& need) {
  pos[0] = o[0] + s.t * d[0];
  pos[1] = o[1] + s.t * d[1];
  pos[2] = o[2] + s.t * d[2];
  const f4 a = T.node[2 * (long)s.leaf], b = T.node[2 * (long)s.leaf + 1];
  const float mn[3] = {a[0], a[1], a[2]}, sz[3] = {b[0], b[1], b[2]};
  float near;
  const float far = slab(mn, sz, pos, d, near);
  need = far < (float)((double)m * step);
  return far;
}
// FB fine-march samples (ray slot, sample index), any rays: cached SDF of the cell each one falls in
template <int FB>
__device__ __forceinline__ void march_samples(const Oct& T, const float (*req)[6], const int (&slot)[FB], const int (&idx)[FB],
                                              const bool (&valid)[FB], int m, float stepf, float (&sv)[FB]) {
  float q[FB][3];
  int ptr[FB];
  bool done[FB];
#pragma unroll
  for (int k = 0; k < FB; ++k) {
    const float tm = lin01(idx[k] + 1, m) * (float)m * stepf + stepf;
    const float* r = req[slot[k]];
#pragma unroll
    for (int c = 0; c < 3; ++c) q[k][c] = r[c] + r[3 + c] * tm;
    const bool inside = valid[k] && in_root(T.root, q[k]);
    ptr[k] = inside ? base_cell(T, q[k]) : (int)(T.B - 1);   // outside: sdf_val[-1] (octree.py:465-466)
    done[k] = !inside;
    sv[k] = 0.f;
  }
  bool any = true;
  while (any) {
    f4 a[FB], b[FB];
#pragma unroll
    for (int k = 0; k < FB; ++k) {
      a[k] = T.node[2 * (long)ptr[k]];
      b[k] = T.node[2 * (long)ptr[k] + 1];
    }
    any = false;
#pragma unroll
    for (int k = 0; k < FB; ++k) {
      const int fc = __float_as_int(a[k][3]);
      const bool leaf = done[k] || fc < 0;
      sv[k] = b[k][3];
      const int nxt = child_of(a[k], b[k], fc, q[k]);
      ptr[k] = leaf ? ptr[k] : nxt;
      done[k] = leaf;
      any = any || !leaf;
    }
  }
}
// second half: advance by `far`, next cell (octree.py:560-573)
__device__ __forceinline__ void step_end(const Oct& T, const float o[3], const float d[3], RayState& s, float far) {
  s.t = s.t + (far + 1e-3f);
  const float pos[3] = {o[0] + s.t * d[0], o[1] + s.t * d[1], o[2] + s.t * d[2]};
  if (!in_root(T.root, pos)) {
    s.leaf = -1;
    s.active = false;
  } else {
    float sv;
    s.leaf = descend(T, pos, sv);
    s.active = !(sv <= 1e-4f);
  }
}

// plane projection onto the hit cell's tangent plane (octree.py:421-438) and outputs of OctreeTracing.forward
__device__ __forceinline__ void cast_finish(const Oct& T, const float o_cast[3], const float o_orig[3], const float d[3],
                                            const RayState& s, float clamp_dt, float* x_out, unsigned char* hit_out,
                                            float* t_out) {
  float t = s.t;
  const bool hit = s.leaf >= 0;
  if (hit) {
    const f4 a = T.node[2 * (long)s.leaf], b = T.node[2 * (long)s.leaf + 1];
    const float* n = T.nrm + 3 * (long)s.leaf;
    const float sv = b[3];
    float dist = 0.f, speed = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float ctr = a[c] + b[c] * 0.5f;
      const float q = ctr - n[c] * sv;
      const float pos = o_cast[c] + s.t * d[c];
      const float pd = (q - pos) * n[c];
      const float ps = d[c] * n[c];
      dist = c == 0 ? pd : dist + pd;
      speed = c == 0 ? ps : speed + ps;
    }
    if (speed == 0.f) speed = 1e-4f;
    float dt = dist / speed;
    if (dt == dt) dt = fminf(fmaxf(dt, -clamp_dt), clamp_dt);
    t = t + dt;
  }
  x_out[0] = t * d[0] + o_orig[0];
  x_out[1] = t * d[1] + o_orig[1];
  x_out[2] = t * d[2] + o_orig[2];
  *hit_out = hit ? 1 : 0;
  *t_out = t;
}

__device__ __forceinline__ int multi_samp(long R, int n_act) {
  long a = 10 * R;
  a = a < 1 ? 1 : (a > 2000000 ? 2000000 : a);
  long m = a / n_act;
  return (int)(m < 1 ? 1 : (m > 100 ? 100 : m));
}

// ---------------------------------------------------------------------------------------------------------
// Batched cast: one workgroup (1024 threads) per lock-step batch of <= 1024 rays.
// origins: [nb,3] (per_ray_origin = 0: one camera per batch) or [nb*batch... ,3] per ray (per_ray_origin = 1).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_cast_batched(Oct T, const float* __restrict__ origins, int per_ray_origin,
                                                        const float* __restrict__ dirs, long R_total, int batch,
                                                        int max_iter, double step, float clamp_dt,
                                                        float* __restrict__ x_out, unsigned char* __restrict__ hit_out,
                                                        float* __restrict__ t_out, int* __restrict__ sched,
                                                        int sched_cap) {
  __shared__ int s_cnt[2];
  __shared__ int s_nreq;
  __shared__ float req[1024][6];   // fine-march requests of this iteration: sample origin and direction
  __shared__ int jmin[1024];       // first sample of a request whose cell has sdf <= step (m = none)
  const int tid = threadIdx.x;
  const long base = (long)blockIdx.x * batch;
  const long rem = R_total - base;
  const int R = (int)(rem < batch ? rem : batch);
  const long ray = base + tid;
  const bool mine = tid < R;
  float o[3] = {0.f, 0.f, 0.f}, oc[3] = {0.f, 0.f, 0.f}, d[3] = {1.f, 0.f, 0.f};
  RayState s;
  s.t = -1.f;
  s.leaf = -1;
  s.active = false;
  if (mine) {
    const float* op = origins + 3 * (per_ray_origin ? ray : (long)blockIdx.x);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      o[c] = op[c];
      d[c] = dirs[3 * ray + c];
      oc[c] = max_iter > 0 ? o[c] + d[c] * 0.005f : o[c];
    }
    s = cast_init(T, oc, d);
  }
  if (tid == 0) {
    s_cnt[0] = 0;
    s_cnt[1] = 0;
    s_nreq = 0;
  }
  __syncthreads();
  {
    const unsigned long long m = __ballot(s.active);
    if ((tid & 63) == 0 && m) atomicAdd(&s_cnt[0], __popcll(m));
  }
  __syncthreads();
  const float stepf = (float)step;
  int it = 0;
  while (true) {
    const int n_act = s_cnt[it & 1];
    if (n_act == 0) break;
    if (max_iter > 0 && it > max_iter) break;
    const int m = multi_samp(R, n_act);
    if (tid == 0) {
      s_cnt[(it + 1) & 1] = 0;
      if (sched && it < sched_cap) {
        sched[((long)blockIdx.x * sched_cap + it) * 2] = n_act;
        sched[((long)blockIdx.x * sched_cap + it) * 2 + 1] = m;
      }
    }
    // -- every active ray: exit distance of its cell; the rays that need the fine march queue a request
    float far = 0.f;
    bool need = false;
    int my_slot = -1;
    if (s.active) {
      float pos[3];
      far = step_begin(T, oc, d, s, m, step, pos, need);
      if (need) {
        my_slot = atomicAdd(&s_nreq, 1);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          req[my_slot][c] = pos[c];
          req[my_slot][3 + c] = d[c];
        }
        jmin[my_slot] = m;
      }
    }
    __syncthreads();
    // -- the n_req x m samples of the iteration (the reference's flat [n_act * multi_samp] query, octree.py:546-558;
    // at most ~10 R of them by construction of multi_samp) spread over all 1024 threads, four tree walks in flight each
    {
      constexpr int FB = 4;
      const int total = s_nreq * m;
      for (int w0 = tid; w0 < total; w0 += 1024 * FB) {
        int slot[FB], idx[FB];
        bool valid[FB];
        float sv[FB];
#pragma unroll
        for (int k = 0; k < FB; ++k) {
          const int w = w0 + 1024 * k;
          valid[k] = w < total;
          const int wc = valid[k] ? w : 0;
          slot[k] = wc / m;
          idx[k] = wc - slot[k] * m;
        }
        march_samples<FB>(T, req, slot, idx, valid, m, stepf, sv);
#pragma unroll
        for (int k = 0; k < FB; ++k)
          if (valid[k] && sv[k] <= stepf) atomicMin(&jmin[slot[k]], idx[k]);
      }
    }
    __syncthreads();
    if (tid == 0) s_nreq = 0;
    if (s.active) {
      if (need) far = lin01(jmin[my_slot], m) * (float)m * stepf + stepf;
      step_end(T, oc, d, s, far);
    }
    const unsigned long long bm = __ballot(s.active);
    if ((tid & 63) == 0 && bm) atomicAdd(&s_cnt[(it + 1) & 1], __popcll(bm));
    __syncthreads();
    ++it;
  }
  if (mine) cast_finish(T, oc, o, d, s, clamp_dt, x_out + 3 * ray, hit_out + ray, t_out + ray);
}

// ---------------------------------------------------------------------------------------------------------
// One lock-step batch of arbitrary size: init / one launch per iteration / finish, counters on the device.
// state: t[R] float, leaf[R] int32, active[R] uint8;  counters[it] = number of rays active at the start of iteration it.
// ---------------------------------------------------------------------------------------------------------
__global__ void k_cast_init(Oct T, const float* __restrict__ origins, const float* __restrict__ dirs, long R,
                            int max_iter, float* __restrict__ t, int* __restrict__ leaf,
                            unsigned char* __restrict__ active, int* __restrict__ counters) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  bool act = false;
  if (i < R) {
    float o[3], d[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      d[c] = dirs[3 * i + c];
      o[c] = origins[3 * i + c];
      if (max_iter > 0) o[c] = o[c] + d[c] * 0.005f;
    }
    RayState s = cast_init(T, o, d);
    t[i] = s.t;
    leaf[i] = s.leaf;
    active[i] = s.active;
    act = s.active;
  }
  const unsigned long long m = __ballot(act);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(&counters[0], __popcll(m));
}

__global__ void k_cast_iter(Oct T, const float* __restrict__ origins, const float* __restrict__ dirs, long R,
                            int max_iter, double step, int it, float* __restrict__ t, int* __restrict__ leaf,
                            unsigned char* __restrict__ active, int* __restrict__ counters) {
  const int n_act = counters[it];
  if (n_act == 0) return;
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  bool act = false;
  if (i < R && active[i]) {
    float o[3], d[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      d[c] = dirs[3 * i + c];
      o[c] = origins[3 * i + c];
      if (max_iter > 0) o[c] = o[c] + d[c] * 0.005f;
    }
    RayState s;
    s.t = t[i];
    s.leaf = leaf[i];
    s.active = true;
    cast_step(T, o, d, s, multi_samp(R, n_act), step);
    t[i] = s.t;
    leaf[i] = s.leaf;
    active[i] = s.active;
    act = s.active;
  }
  const unsigned long long m = __ballot(act);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(&counters[it + 1], __popcll(m));
}

__global__ void k_cast_finish(Oct T, const float* __restrict__ origins, const float* __restrict__ dirs, long R,
                              int max_iter, float clamp_dt, const float* __restrict__ t, const int* __restrict__ leaf,
                              float* __restrict__ x_out, unsigned char* __restrict__ hit_out, float* __restrict__ t_out) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= R) return;
  float o[3], oc[3], d[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    d[c] = dirs[3 * i + c];
    o[c] = origins[3 * i + c];
    oc[c] = max_iter > 0 ? o[c] + d[c] * 0.005f : o[c];
  }
  RayState s;
  s.t = t[i];
  s.leaf = leaf[i];
  s.active = false;
  cast_finish(T, oc, o, d, s, clamp_dt, x_out + 3 * i, hit_out + i, t_out + i);
}

// ---------------------------------------------------------------------------------------------------------
// Build (octree.py:124-181, 377-409).
// ---------------------------------------------------------------------------------------------------------
// base grid: cell (ix,iy,iz) -> min = (i/n)*size + root_min, size = ((i+1)/n)*size + root_min - min
__global__ void k_oct_base(Root r, f4* __restrict__ node, float* __restrict__ centre) {
  const long n = (long)r.res[0] * r.res[1] * r.res[2];
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int iz = (int)(i % r.res[2]), iy = (int)((i / r.res[2]) % r.res[1]), ix = (int)(i / ((long)r.res[2] * r.res[1]));
  const int id[3] = {ix, iy, iz};
  float mn[3], sz[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float lo = ((float)id[c] / (float)r.res[c]) * r.sz[c] + r.mn[c];
    const float hi = (((float)id[c] + 1.0f) / (float)r.res[c]) * r.sz[c] + r.mn[c];
    mn[c] = lo;
    sz[c] = hi - lo;
    centre[3 * i + c] = lo + sz[c] * 0.5f;
  }
  node[2 * i] = f4{mn[0], mn[1], mn[2], __int_as_float(-1)};
  node[2 * i + 1] = f4{sz[0], sz[1], sz[2], 0.f};
}

// split flag of a level: |sdf(centre)| < |size| * thr   (octree.py:381-385)
__global__ void k_oct_mark(const f4* __restrict__ node, long first, long count, const float* __restrict__ sdf, float thr,
                           int* __restrict__ flag) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= count) return;
  const f4 b = node[2 * (first + i) + 1];
  const float nrm = sqrtf(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
  flag[i] = fabsf(sdf[i]) < nrm * thr ? 1 : 0;
}

// exclusive scan of int flags, 3 passes (block sums -> scan of sums -> apply); n up to 2^31
__global__ void k_scan_block(const int* __restrict__ in, long n, int* __restrict__ out, int* __restrict__ sums) {
  __shared__ int s[1024];
  const long i = blockIdx.x * 1024L + threadIdx.x;
  const int v = i < n ? in[i] : 0;
  s[threadIdx.x] = v;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    int a = threadIdx.x >= o ? s[threadIdx.x - o] : 0;
    __syncthreads();
    s[threadIdx.x] += a;
    __syncthreads();
  }
  if (i < n) out[i] = s[threadIdx.x] - v;
  if (threadIdx.x == 1023) sums[blockIdx.x] = s[1023];
}
__global__ void k_scan_sums(int* __restrict__ sums, int nb, int* __restrict__ total) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    int acc = 0;
    for (int i = 0; i < nb; ++i) {
      const int v = sums[i];
      sums[i] = acc;
      acc += v;
    }
    *total = acc;
  }
}
__global__ void k_scan_apply(int* __restrict__ out, long n, const int* __restrict__ sums) {
  const long i = blockIdx.x * 1024L + threadIdx.x;
  if (i < n) out[i] += sums[blockIdx.x];
}

// children of the split nodes of a level (divide(), octree.py:60-72): child block r of parent starts at new_first + 8 r
__global__ void k_oct_split(f4* __restrict__ node, float* __restrict__ centre, long first, long count,
                            const int* __restrict__ flag, const int* __restrict__ rank, long new_first) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= count || !flag[i]) return;
  const long p = first + i;
  const long c0 = new_first + 8L * rank[i];
  f4 a = node[2 * p];
  const f4 b = node[2 * p + 1];
  a[3] = __int_as_float((int)c0);
  node[2 * p] = a;
  for (int o = 0; o < 8; ++o) {
    const int of[3] = {(o >> 2) & 1, (o >> 1) & 1, o & 1};
    float mn[3], sz[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      mn[c] = a[c] + ((float)of[c] * b[c]) / 2.f;
      sz[c] = b[c] / 2.f;
      centre[3 * (c0 + o) + c] = mn[c] + sz[c] * 0.5f;
    }
    node[2 * (c0 + o)] = f4{mn[0], mn[1], mn[2], __int_as_float(-1)};
    node[2 * (c0 + o) + 1] = f4{sz[0], sz[1], sz[2], 0.f};
  }
}

// cached cell values: sdf at the centre and the unit gradient (octree.py:390-401)
__global__ void k_oct_store(f4* __restrict__ node, float* __restrict__ nrm, long first, long count,
                            const float* __restrict__ sdf, const float* __restrict__ grad) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= count) return;
  f4 b = node[2 * (first + i) + 1];
  b[3] = sdf[i];
  node[2 * (first + i) + 1] = b;
  const float gx = grad[3 * i], gy = grad[3 * i + 1], gz = grad[3 * i + 2];
  const float n = fmaxf(sqrtf(gx * gx + gy * gy + gz * gz), 1e-4f);
  nrm[3 * (first + i)] = gx / n;
  nrm[3 * (first + i) + 1] = gy / n;
  nrm[3 * (first + i) + 2] = gz / n;
}

static inline Oct make_oct(const float* node, const float* nrm, long B, const float* root_min, const float* root_size,
                           const int* res) {
  Oct T;
  T.node = (const f4*)node;
  T.nrm = nrm;
  T.B = B;
  for (int c = 0; c < 3; ++c) {
    T.root.mn[c] = root_min[c];
    T.root.sz[c] = root_size[c];
    T.root.res[c] = res[c];
  }
  return T;
}

}  // namespace rb

using namespace rb;

extern "C" {

int rb_octree_base_grid(const float* root_min, const float* root_size, const int* res, float* node, float* centre,
                        rb_stream_t stream) {
  RB_REQUIRE(root_min && root_size && res && node && centre, "null pointer");
  Oct T = make_oct(node, nullptr, 0, root_min, root_size, res);
  const long n = (long)res[0] * res[1] * res[2];
  hipLaunchKernelGGL(k_oct_base, grid1d(n, 256), dim3(256), 0, (hipStream_t)stream, T.root, (f4*)node, centre);
  return check_launch("k_oct_base");
}

int rb_octree_mark_split(const float* node, long first, long count, const float* sdf, float thr, int* flag, int* rank,
                         int* scan_tmp, int* total, rb_stream_t stream) {
  if (count <= 0) return 0;
  RB_REQUIRE(node && sdf && flag && rank && scan_tmp && total, "null pointer");
  hipStream_t s = (hipStream_t)stream;
  const int nb = (int)((count + 1023) / 1024);
  hipLaunchKernelGGL(k_oct_mark, grid1d(count, 256), dim3(256), 0, s, (const f4*)node, first, count, sdf, thr, flag);
  hipLaunchKernelGGL(k_scan_block, dim3(nb), dim3(1024), 0, s, flag, count, rank, scan_tmp);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(64), 0, s, scan_tmp, nb, total);
  hipLaunchKernelGGL(k_scan_apply, dim3(nb), dim3(1024), 0, s, rank, count, scan_tmp);
  return check_launch("k_oct_mark/scan");
}

int rb_octree_subdivide(float* node, float* centre, long first, long count, const int* flag, const int* rank,
                        long new_first, rb_stream_t stream) {
  if (count <= 0) return 0;
  RB_REQUIRE(node && centre && flag && rank, "null pointer");
  hipLaunchKernelGGL(k_oct_split, grid1d(count, 256), dim3(256), 0, (hipStream_t)stream, (f4*)node, centre, first, count,
                     flag, rank, new_first);
  return check_launch("k_oct_split");
}

int rb_octree_store_cells(float* node, float* nrm, long first, long count, const float* sdf, const float* grad,
                          rb_stream_t stream) {
  if (count <= 0) return 0;
  RB_REQUIRE(node && nrm && sdf && grad, "null pointer");
  hipLaunchKernelGGL(k_oct_store, grid1d(count, 256), dim3(256), 0, (hipStream_t)stream, (f4*)node, nrm, first, count,
                     sdf, grad);
  return check_launch("k_oct_store");
}

int rb_octree_cast_batched(const float* node, const float* nrm, long B, const float* root_min, const float* root_size,
                           const int* res, const float* origins, int per_ray_origin, const float* dirs, long R_total,
                           int batch, int max_iter, double step, float clamp_dt, float* x_out, unsigned char* hit_out,
                           float* t_out, int* sched, int sched_cap, rb_stream_t stream) {
  if (R_total <= 0) return 0;
  RB_REQUIRE(node && nrm && origins && dirs && x_out && hit_out && t_out, "null pointer");
  RB_REQUIRE(batch >= 1 && batch <= 1024, "a lock-step batch of the batched kernel holds 1..1024 rays");
  Oct T = make_oct(node, nrm, B, root_min, root_size, res);
  const long nb = (R_total + batch - 1) / batch;
  RB_REQUIRE(nb <= RB_MAX_BLOCKS, "too many lock-step batches for one launch");
  hipLaunchKernelGGL(k_cast_batched, dim3((unsigned)nb), dim3(1024), 0, (hipStream_t)stream, T, origins, per_ray_origin,
                     dirs, R_total, batch, max_iter, step, clamp_dt, x_out, hit_out, t_out, sched, sched_cap);
  return check_launch("k_cast_batched");
}

int rb_octree_cast_init(const float* node, const float* nrm, long B, const float* root_min, const float* root_size,
                        const int* res, const float* origins, const float* dirs, long R, int max_iter, float* t,
                        int* leaf, unsigned char* active, int* counters, rb_stream_t stream) {
  if (R <= 0) return 0;
  RB_REQUIRE(node && origins && dirs && t && leaf && active && counters, "null pointer");
  Oct T = make_oct(node, nrm, B, root_min, root_size, res);
  hipLaunchKernelGGL(k_cast_init, grid1d(R, 256), dim3(256), 0, (hipStream_t)stream, T, origins, dirs, R, max_iter, t,
                     leaf, active, counters);
  return check_launch("k_cast_init");
}

int rb_octree_cast_iter(const float* node, const float* nrm, long B, const float* root_min, const float* root_size,
                        const int* res, const float* origins, const float* dirs, long R, int max_iter, double step,
                        int it_first, int it_count, float* t, int* leaf, unsigned char* active, int* counters,
                        rb_stream_t stream) {
  if (R <= 0) return 0;
  RB_REQUIRE(node && origins && dirs && t && leaf && active && counters, "null pointer");
  Oct T = make_oct(node, nrm, B, root_min, root_size, res);
  for (int it = it_first; it < it_first + it_count; ++it)
    hipLaunchKernelGGL(k_cast_iter, grid1d(R, 256), dim3(256), 0, (hipStream_t)stream, T, origins, dirs, R, max_iter, step,
                       it, t, leaf, active, counters);
  return check_launch("k_cast_iter");
}

int rb_octree_cast_finish(const float* node, const float* nrm, long B, const float* root_min, const float* root_size,
                          const int* res, const float* origins, const float* dirs, long R, int max_iter, float clamp_dt,
                          const float* t, const int* leaf, float* x_out, unsigned char* hit_out, float* t_out,
                          rb_stream_t stream) {
  if (R <= 0) return 0;
  RB_REQUIRE(node && nrm && origins && dirs && t && leaf && x_out && hit_out && t_out, "null pointer");
  Oct T = make_oct(node, nrm, B, root_min, root_size, res);
  hipLaunchKernelGGL(k_cast_finish, grid1d(R, 256), dim3(256), 0, (hipStream_t)stream, T, origins, dirs, R, max_iter,
                     clamp_dt, t, leaf, x_out, hit_out, t_out);
  return check_launch("k_cast_finish");
}

}  // extern "C"
