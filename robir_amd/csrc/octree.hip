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
#include <atomic>

#include "octree_dev.h"
#include <cstdlib>

namespace rb {

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
// The same lock-step batch in ONE launch (round 3): a persistent grid (every workgroup resident: the host caps it at one per compute
// unit, 1024 threads each) walks its rays through init / all iterations / finish; the per-iteration count of active rays -- the only thing the rays of a
// batch share (multi_samp, octree.py:545-549) -- is the device-side counter of the per-iteration launches, completed by a grid-wide
// arrival counter (release / acquire at agent scope).  33 launches of ~33 us each per trace_radiance call (16 % of BASELINE config 5,
// profiles/r03_config5_kernel_stats.md) become one.  A thread owns rays i = tid + k * threads for the whole cast; their state lives
// in t / leaf / active as before (any batch size), results are those of the per-iteration kernels bit for bit.
// counters: [max_it + 2] active counts (output), arrive: 2 x 512 64-bit words (two per workgroup, by epoch parity; zeroed).
// ---------------------------------------------------------------------------------------------------------
// Grid barrier + sum without same-address atomics (those serialise at the L2 at ~0.4 us each: 118 workgroups = 40 us per iteration):
// workgroup g publishes ONE 64-bit word {epoch, its active count} in its own slot; thread k of every workgroup polls slot k until it
// carries the epoch, and the counts are summed in the workgroup.  The slots are DOUBLE-BUFFERED by epoch parity (words[2][512], zeroed
// before the launch; epochs start at 1): with one slot per workgroup a fast workgroup A could pass barrier e, publish e+1 into its slot
// and so hide epoch e from a slow poller C that had not read A's slot yet -- C would never see e, never publish e+1, and A would wait for
// C for ever.  With two slots A can overwrite the slot of epoch e only with e+2, i.e. after passing barrier e+1, which needs C's e+1,
// which C publishes only after it has read every slot at epoch e.
constexpr int CC_MAX_GROUPS = 512;
__device__ __forceinline__ int grid_sum_and_wait(unsigned long long* words, int epoch, int my_active, int n_groups, int* lds_sum) {
  words += (epoch & 1) * CC_MAX_GROUPS;
  __threadfence();          // every thread: its own stores (ray state) are visible device-wide before the workgroup publishes
  if (threadIdx.x == 0) *lds_sum = 0;
  __syncthreads();
  if (threadIdx.x == 0)
    __hip_atomic_store(&words[blockIdx.x], ((unsigned long long)(unsigned)epoch << 32) | (unsigned)my_active, __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_AGENT);
  int c = 0;
  if ((int)threadIdx.x < n_groups) {
    unsigned long long v;
    while ((int)((v = __hip_atomic_load(&words[threadIdx.x], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != epoch)
      __builtin_amdgcn_s_sleep(1);
    c = (int)(v & 0xffffffffull);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(lds_sum, c);
  __syncthreads();
  __threadfence();          // ... and it sees the others' (a ray changes hands between the init / iteration / finish phases)
  return *lds_sum;
}

// CC_LPR lanes per ray: the fine march (10..100 samples, each a chain of dependent node reads) runs CC_LPR samples at a time
template <int CC_LPR>
__global__ __launch_bounds__(1024) void k_cast_coop(Oct T, const float* __restrict__ origins, const float* __restrict__ dirs, long R,
                                                  int max_iter, double step, int it_limit, float clamp_dt, float* __restrict__ t,
                                                  int* __restrict__ leaf, unsigned char* __restrict__ active,
                                                  int* __restrict__ counters, unsigned long long* __restrict__ arrive, float* __restrict__ x_out,
                                                  unsigned char* __restrict__ hit_out, float* __restrict__ t_out) {
  const long nthreads = (long)gridDim.x * blockDim.x, tid = blockIdx.x * (long)blockDim.x + threadIdx.x;
  const long ngrp = nthreads / CC_LPR, grp = tid / CC_LPR;          // ray groups of CC_LPR lanes
  const int sub = (int)(tid % CC_LPR);
  const int ngroups = (int)gridDim.x;
  __shared__ int wg_count, wg_total;
  auto load_ray = [&](long i, float (&o)[3], float (&d)[3]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      d[c] = dirs[3 * i + c];
      o[c] = origins[3 * i + c];
      if (max_iter > 0) o[c] = o[c] + d[c] * 0.005f;
    }
  };
  // active rays of this workgroup -> the grid-wide count (also recorded in counters[slot] by workgroup 0, for the caller)
  auto count_and_sync = [&](int n_mine, int slot) {
    if (threadIdx.x == 0) wg_count = 0;
    __syncthreads();
    int total = n_mine;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) total += __shfl_down(total, off);
    if ((threadIdx.x & 63) == 0 && total) atomicAdd(&wg_count, total);
    __syncthreads();
    const int mine_wg = wg_count;
    const int n = grid_sum_and_wait(arrive, slot + 1, mine_wg, ngroups, &wg_total);
    if (blockIdx.x == 0 && threadIdx.x == 0) counters[slot] = n;
    return n;
  };
  int n_act = 0;
  {   // init (k_cast_init): one ray per thread
    int mine = 0;
    for (long i = tid; i < R; i += nthreads) {
      float o[3], d[3];
      load_ray(i, o, d);
      RayState s = cast_init(T, o, d);
      t[i] = s.t;
      leaf[i] = s.leaf;
      active[i] = s.active;
      mine += s.active ? 1 : 0;
    }
    n_act = count_and_sync(mine, 0);
  }
  for (int it = 0; it < it_limit; ++it) {      // k_cast_iter: CC_LPR lanes per ray (ray i is always walked by group i mod ngrp)
    if (n_act == 0) break;
    const int ms = multi_samp(R, n_act);
    int mine = 0;
    for (long i = grp; i < R; i += ngrp) {
      if (!__hip_atomic_load(&active[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) continue;
      float o[3], d[3];
      load_ray(i, o, d);
      RayState s;
      s.t = __hip_atomic_load(&t[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s.leaf = __hip_atomic_load(&leaf[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s.active = true;
      cast_step_group<CC_LPR>(T, o, d, s, ms, step, sub);
      if (sub == 0) {
        __hip_atomic_store(&t[i], s.t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&leaf[i], s.leaf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&active[i], (unsigned char)(s.active ? 1 : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        mine += s.active ? 1 : 0;
      }
    }
    n_act = count_and_sync(mine, it + 1);
  }
  for (long i = tid; i < R; i += nthreads) {   // k_cast_finish: one ray per thread
    float o[3], oc[3], d[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      d[c] = dirs[3 * i + c];
      o[c] = origins[3 * i + c];
      oc[c] = max_iter > 0 ? o[c] + d[c] * 0.005f : o[c];
    }
    RayState s;
    s.t = __hip_atomic_load(&t[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s.leaf = __hip_atomic_load(&leaf[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s.active = false;
    cast_finish(T, oc, o, d, s, clamp_dt, x_out + 3 * i, hit_out + i, t_out + i);
  }
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

int rb_octree_cast_coop(const float* node, const float* nrm, long B, const float* root_min, const float* root_size,
                        const int* res, const float* origins, const float* dirs, long R, int max_iter, double step, int max_total,
                        float clamp_dt, float* t, int* leaf, unsigned char* active, int* counters, int* arrive, float* x_out,
                        unsigned char* hit_out, float* t_out, rb_stream_t stream) {
  if (R <= 0) return 0;
  RB_REQUIRE(node && nrm && origins && dirs && t && leaf && active && counters && arrive && x_out && hit_out && t_out, "null pointer");
  RB_REQUIRE(max_total >= 1, "max_total >= 1");
  // Co-residency of the whole grid is a REQUIREMENT (the grid barrier spins).  What guarantees it: (a) the grid never exceeds ONE workgroup
  // per compute unit of THIS device (cached per device id; the callers' 16 384 rays x 4 lanes are 64 workgroups of 1024 threads) and the
  // launch is a cooperative one, which the runtime validates against the kernel's STATIC occupancy (hipErrorCooperativeLaunchTooLarge:
  // status 2 = "not launched, take the per-iteration launches", ops.octree_cast_general does); (b) under concurrent work on other streams /
  // by a second rank on the device, residency rests on the runtime's cooperative queue, which dispatches a cooperative grid as a unit --
  // the HIP API does not test DYNAMIC residency itself, and this repository has no stress test of that case (a hang would cost a GPU box);
  // ROBIR_CAST_ONE_LAUNCH=0 takes the per-iteration launches wherever that guarantee is in doubt.
  static int lpr = 0;
  if (!lpr) {
    const char* e = getenv("ROBIR_CAST_LPR");
    lpr = e ? atoi(e) : 4;       // measured (tools/ab_cast.py, 7500 secondary rays): 1.13 / 0.82 / 2.0 ms at 1 / 4 / 16 lanes per ray
    if (lpr != 1 && lpr != 4 && lpr != 16) lpr = 4;
  }
  const void* kern = lpr == 1 ? (const void*)k_cast_coop<1> : lpr == 4 ? (const void*)k_cast_coop<4> : (const void*)k_cast_coop<16>;
  constexpr int MAX_DEV = 64;
  static std::atomic<int> resident[MAX_DEV];  // workgroups of this kernel the grid may have on the device; 0 = not queried yet (any thread may fill it: same value)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return rb::fail(__func__, "device query failed");
  if (!resident[dev].load(std::memory_order_acquire)) {
    hipDeviceProp_t prop;
    int per_cu = 0;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 1024, 0) != hipSuccess || per_cu < 1 || !prop.cooperativeLaunch) {
      (void)hipGetLastError();
      rb::fail(__func__, "cooperative launch not available on this device");
      return 2;
    }
    resident[dev].store(prop.multiProcessorCount, std::memory_order_release);      // one workgroup per compute unit, whatever per_cu allows
  }
  Oct T = make_oct(node, nrm, B, root_min, root_size, res);
  const long want = (R * lpr + 1023) / 1024;
  const int res_dev = resident[dev].load(std::memory_order_acquire);
  const long cap = res_dev < CC_MAX_GROUPS ? res_dev : CC_MAX_GROUPS;
  const unsigned grid = (unsigned)(want < cap ? want : cap);
  int it_limit = max_iter > 0 ? max_iter + 1 : max_total;
  unsigned long long* arrive64 = (unsigned long long*)arrive;
  void* args[] = {&T, &origins, &dirs, &R, &max_iter, &step, &it_limit, &clamp_dt, &t, &leaf, &active, &counters, &arrive64, &x_out,
                  &hit_out, &t_out};
  const hipError_t e = hipLaunchCooperativeKernel(kern, dim3(grid), dim3(1024), args, 0, (hipStream_t)stream);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    rb::fail(__func__, hipGetErrorString(e));
    return 2;
  }
  return check_launch("k_cast_coop");
}

}  // extern "C"
