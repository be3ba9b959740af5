// Device-side octree walk and lock-step cast steps shared by octree.hip (OctreeTracing / OctreeSDF.cast) and octree_vis.hip
// (OctreeVisModel as the light-visibility model).  See octree.hip for the layout and the reference lines.
#pragma once
#include <hip/hip_runtime.h>

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
// nfetch (optional): running count of 32-byte node records read -- the tracer's algorithmic memory traffic
__device__ __forceinline__ int descend(const Oct& T, const float x[3], float& leaf_sdf, int* nfetch = nullptr) {
  int ptr = base_cell(T, x);
  while (true) {
    const f4 a = T.node[2 * (long)ptr], b = T.node[2 * (long)ptr + 1];
    if (nfetch) *nfetch += 1;
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
                                          double step, int* nfetch = nullptr) {
  float pos[3] = {o[0] + s.t * d[0], o[1] + s.t * d[1], o[2] + s.t * d[2]};
  const f4 a = T.node[2 * (long)s.leaf], b = T.node[2 * (long)s.leaf + 1];
  if (nfetch) *nfetch += 1;
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
        if (nfetch) *nfetch += FB;
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
    s.leaf = descend(T, pos, sv, nfetch);
    s.active = !(sv <= 1e-4f);
  }
}

// cast_step with the fine march spread over the LPR consecutive lanes that share the ray (all of them call this with the same ray and
// state; sub = lane % LPR): lane `sub` walks samples sub, sub + LPR, ... (a round of LPR samples at a time, stopping after the first
// round that contains a cell with sdf <= step), the first such sample is a min over the group.  Same t, leaf, active as cast_step:
// the sample positions and the comparison are the same expressions, only evaluated side by side.
template <int LPR>
__device__ __forceinline__ void cast_step_group(const Oct& T, const float o[3], const float d[3], RayState& s, int m, double step,
                                                int sub) {
  float pos[3] = {o[0] + s.t * d[0], o[1] + s.t * d[1], o[2] + s.t * d[2]};
  const f4 a = T.node[2 * (long)s.leaf], b = T.node[2 * (long)s.leaf + 1];
  const float mn[3] = {a[0], a[1], a[2]}, sz[3] = {b[0], b[1], b[2]};
  float near;
  float far = slab(mn, sz, pos, d, near);
  if (far < (float)((double)m * step)) {
    const float stepf = (float)step;
    int j = m;
    for (int i0 = 0; i0 < m && j == m; i0 += LPR) {     // uniform over the group: j is the group's min after every round
      const int i = i0 + sub;
      const float tm = lin01(i + 1, m) * (float)m * stepf + stepf;
      float q[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) q[c] = pos[c] + d[c] * tm;
      const bool inside = (i < m) && in_root(T.root, q);
      int ptr = inside ? base_cell(T, q) : (int)(T.B - 1);   // outside: sdf_val[-1] (octree.py:465-466)
      float sv = 0.f;
      bool done = !inside;
      while (true) {
        const f4 na = T.node[2 * (long)ptr], nb = T.node[2 * (long)ptr + 1];
        const int fc = __float_as_int(na[3]);
        sv = nb[3];
        if (done || fc < 0) break;
        ptr = child_of(na, nb, fc, q);
      }
      int mine = (i < m && sv <= stepf) ? i : m;
#pragma unroll
      for (int off = LPR / 2; off > 0; off >>= 1) {
        const int other = __shfl_xor(mine, off);
        mine = other < mine ? other : mine;
      }
      j = mine;
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
    s.leaf = descend(T, pos, sv, nullptr);
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
