// Traced visibility: OctreeVisModel (model/octree_tracing.py:63-85) as the VisModel of get_diffuse_visibility
// (model/sg_render.py:111-195) -- the mode the reference's runners switch on with `trace_vis`
// (training/train_pbr.py:409-410, train_cesr.py:584-585): every surviving (point, direction) pair casts a secondary ray
// through the cached-SDF octree instead of evaluating the visibility MLP; the "logits" are the float pair [is_hit, ~is_hit].
//
// What the reference does per 1024-pixel chunk: cull n.d <= 1e-6, take the survivors in (point, direction) order, hand them to
// the VisModel in batches of 2 000 000, and each batch is ONE lock-step cast (utils/octree.py:493-585): its step size depends
// on the batch size (0.01 beyond 100 000 rays, else 0.005) and its fine-march sample count on the number of rays of the batch
// still active at each of the <= 33 iterations.  So the unit that shares state is a GROUP = (chunk, 2 M-pair batch).
//
// Here, for any number of chunks in one call and without ever materialising origins / directions of the pairs:
//   k_ovis_count   per point: number of front-facing directions
//   k_ovis_scan    per chunk: rank of each point's first pair inside the chunk (pair order = reference order)
//   k_ovis_layout  chunk offsets into the global pair list, group table (start, size), totals
//   k_ovis_fill    per point: ORDERED compaction -> pair (point, direction index), ray set-up, first active counts
//   k_ovis_iter    x (max_iter + 1): one lock-step iteration of every group (grid-stride; per-group device counters)
//   k_ovis_reduce  per point: softmax([hit, !hit])[1] (or argmax) per pair, SG-weighted mean per lobe in sample order
// plus k_cast_grouped_* : the same grouped lock-step cast for rays given explicitly (BRDF-lobe visibility of several chunks).
// Nothing is read back to the host: grids are fixed, sizes live in device memory.
#include "../../include/robir_hip.h"
#include "common.h"
#include "octree_dev.h"

namespace rb {

#define RB_TINY 1e-6f
constexpr int OV_MAX_DIRS = 4096;
constexpr int OV_ITERS = 34;        // counters per group: iterations 0 .. max_iter + 1 (max_iter <= 32)

struct OvisLayout {                 // device scalars written by k_ovis_layout / accumulated by k_ovis_iter
  long total_pairs;
  int total_groups;
  int pad;
  unsigned long long node_fetches;  // 32-byte octree records read by the lock-step iterations (algorithmic gather traffic)
  unsigned long long ray_steps;     // (ray, iteration) pairs advanced
};

// Statistics of the iteration kernels: one (records read, ray steps) slot per workgroup behind the OvisLayout struct, updated by
// plain read-modify-write (a slot belongs to one blockIdx, launches of a call are ordered on their stream); the host adds them up.
// Device-wide atomics on two addresses cost ~0.6 us each, serialised: 16 k of them per launch were a 10 ms floor per iteration.
constexpr int OV_STAT_SLOTS = 4096;
__device__ __forceinline__ void ovis_stats_add(OvisLayout* stats, int nfetch, int nstep) {
  __shared__ int s_f[4], s_s[4];
  for (int o = 32; o > 0; o >>= 1) {
    nfetch += __shfl_xor(nfetch, o);
    nstep += __shfl_xor(nstep, o);
  }
  if ((threadIdx.x & 63) == 0) {
    s_f[threadIdx.x >> 6] = nfetch;
    s_s[threadIdx.x >> 6] = nstep;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long* slot = reinterpret_cast<unsigned long long*>(stats + 1) + 2 * (blockIdx.x % OV_STAT_SLOTS);
    slot[0] += (unsigned long long)(s_f[0] + s_f[1] + s_f[2] + s_f[3]);
    slot[1] += (unsigned long long)(s_s[0] + s_s[1] + s_s[2] + s_s[3]);
  }
}

// points of chunk c are the contiguous range [lower_bound(cid, c), lower_bound(cid, c + 1))  (cid ascending)
__device__ __forceinline__ long lower_bound_i32(const int* __restrict__ a, long n, int v) {
  long lo = 0, hi = n;
  while (lo < hi) {
    const long mid = (lo + hi) >> 1;
    if (a[mid] < v) lo = mid + 1; else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(256) void k_ovis_count(const float* __restrict__ normals, const int* __restrict__ cid, long n,
                                                     const float* __restrict__ dirs, int LS, int* __restrict__ pcount) {
  __shared__ int s_cnt[4];
  const int tid = threadIdx.x, lane = tid & 63;
  const long p = blockIdx.x;
  const long dbase = (long)(cid ? cid[p] : 0) * LS;
  const float nx = normals[3 * p], ny = normals[3 * p + 1], nz = normals[3 * p + 2];
  int c = 0;
  for (int j = tid; j < LS; j += 256) {
    const float* d = dirs + 3 * (dbase + j);
    const float dot = nx * d[0] + ny * d[1] + nz * d[2];   // sum(n*d): separate mul/add (-ffp-contract=off)
    c += dot > RB_TINY ? 1 : 0;
  }
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if (lane == 0) s_cnt[tid >> 6] = c;
  __syncthreads();
  if (tid == 0) pcount[p] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// one workgroup per chunk (strips of 1024 points with a running carry: a chunk is normally <= 1024 pixels, but a direct call
// may hand over any number of points as one chunk)
__global__ __launch_bounds__(1024) void k_ovis_scan(const int* __restrict__ cid, long n, int n_chunks,
                                                     const int* __restrict__ pcount, int* __restrict__ prank,
                                                     long* __restrict__ cstart, long* __restrict__ ctotal) {
  __shared__ int s_wave[16];
  __shared__ long s_carry;
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long p0 = cid ? lower_bound_i32(cid, n, c) : 0, p1 = cid ? lower_bound_i32(cid, n, c + 1) : n;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (long base = p0; base < p1; base += 1024) {
    const long p = base + tid;
    const int v = p < p1 ? pcount[p] : 0;
    int incl = v;
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(incl, o);
      if (lane >= o) incl += u;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int off = 0;
    for (int w = 0; w < wave; ++w) off += s_wave[w];
    const long carry = s_carry;
    if (p < p1) prank[p] = (int)(carry + off + incl - v);     // < 2^31 for chunks below 524 288 points x 4096 directions
    __syncthreads();
    if (tid == 1023) s_carry = carry + off + incl;
    __syncthreads();
  }
  if (tid == 0) {
    ctotal[c] = s_carry;
    cstart[c] = p0;
    if (c == n_chunks - 1) cstart[n_chunks] = p1;
  }
}

// single workgroup: offsets of the chunks in the global pair list, group table
__global__ __launch_bounds__(256) void k_ovis_layout(int n_chunks, const long* __restrict__ ctotal, long batch,
                                                      long* __restrict__ coff, int* __restrict__ goff,
                                                      long* __restrict__ gstart, long* __restrict__ gsize,
                                                      OvisLayout* __restrict__ lay, int* __restrict__ counters, int max_groups,
                                                      unsigned long long* __restrict__ eval_count) {
  if (threadIdx.x == 0) {          // C <= a few thousand: a serial scan is microseconds
    long po = 0;
    int go = 0;
    for (int c = 0; c < n_chunks; ++c) {
      coff[c] = po;
      goff[c] = go;
      const long t = ctotal[c];
      const int ng = (int)((t + batch - 1) / batch);
      for (int k = 0; k < ng && go + k < max_groups; ++k) {
        gstart[go + k] = po + (long)k * batch;
        gsize[go + k] = (k + 1 < ng) ? batch : t - (long)k * batch;
      }
      po += t;
      go += ng;
    }
    lay->total_pairs = po;
    lay->total_groups = go < max_groups ? go : max_groups;
    lay->node_fetches = 0ull;      // (the iteration kernels count in the per-workgroup slots behind the struct)
    lay->ray_steps = 0ull;
    if (eval_count) atomicAdd(eval_count, (unsigned long long)po);     // statistics: secondary rays traced
  }
  for (long i = threadIdx.x; i < (long)max_groups * OV_ITERS; i += 256) counters[i] = 0;
  unsigned long long* slots = reinterpret_cast<unsigned long long*>(lay + 1);
  for (int i = threadIdx.x; i < 2 * OV_STAT_SLOTS; i += 256) slots[i] = 0ull;
}

struct PairRays {                   // rays of the light-visibility pairs: origin = surface point, direction from the table
  const float* points;
  const float* dirs;
  const int* cid;
  const int* pair_p;
  const unsigned short* pair_j;
  int LS;
  __device__ __forceinline__ void fetch(long i, float o[3], float d[3]) const {
    const int p = pair_p[i];
    const long row = (long)(cid ? cid[p] : 0) * LS + pair_j[i];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      o[c] = points[3 * (long)p + c];
      d[c] = dirs[3 * row + c];
    }
  }
};
struct ExplicitRays {
  const float* origins;
  const float* dirs;
  __device__ __forceinline__ void fetch(long i, float o[3], float d[3]) const {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      o[c] = origins[3 * i + c];
      d[c] = dirs[3 * i + c];
    }
  }
};

// wave-aggregated counter update: lanes of a wave almost always belong to the same group
__device__ __forceinline__ void count_active(bool act, int g, int* __restrict__ counters, int it) {
  const int g0 = __shfl(g, __ffsll((long long)__ballot(true)) - 1);
  const unsigned long long same = __ballot(act && g == g0);
  if (act && g != g0) atomicAdd(&counters[(long)g * OV_ITERS + it], 1);
  if (same && (threadIdx.x & 63) == (__ffsll((long long)same) - 1)) atomicAdd(&counters[(long)g0 * OV_ITERS + it], __popcll(same));
}

// per point: ordered compaction of its front-facing directions + ray set-up (octree.py:504-519; secondary rays start 0.005
// along the direction)
__global__ __launch_bounds__(256) void k_ovis_fill(Oct T, const float* __restrict__ points, const float* __restrict__ normals,
                                                    const int* __restrict__ cid, long n, const float* __restrict__ dirs, int LS,
                                                    const int* __restrict__ prank, const long* __restrict__ coff,
                                                    const int* __restrict__ goff, long batch, int max_groups,
                                                    int* __restrict__ pair_p, unsigned short* __restrict__ pair_j,
                                                    float* __restrict__ t_st, int* __restrict__ leaf_st,
                                                    unsigned char* __restrict__ act_st, int* __restrict__ grp,
                                                    long2* __restrict__ point_span, int* __restrict__ counters) {
  __shared__ int s_w[4];
  __shared__ int s_base;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long p = blockIdx.x;
  const int c = cid ? cid[p] : 0;
  const long dbase = (long)c * LS;
  const long first = coff[c] + prank[p];
  const float nx = normals[3 * p], ny = normals[3 * p + 1], nz = normals[3 * p + 2];
  const float ox = points[3 * p], oy = points[3 * p + 1], oz = points[3 * p + 2];
  if (tid == 0) s_base = 0;
  __syncthreads();
  // a point's pairs are consecutive in its chunk's order: they fall into at most two consecutive lock-step groups; the active
  // counts go to the groups' counters once per point (one device-wide atomic per step and wave serialised on the counter)
  int gfirst = goff[c] + (int)((long)prank[p] / batch);
  if (gfirst >= max_groups) gfirst = max_groups - 1;
  int a0 = 0, a1 = 0;
  for (int j0 = 0; j0 < LS; j0 += 256) {
    const int j = j0 + tid;
    bool front = false;
    float d[3] = {0.f, 0.f, 0.f};
    if (j < LS) {
      const float* dp = dirs + 3 * (dbase + j);
      d[0] = dp[0];
      d[1] = dp[1];
      d[2] = dp[2];
      front = (nx * d[0] + ny * d[1] + nz * d[2]) > RB_TINY;
    }
    const unsigned long long m = __ballot(front);
    if (lane == 0) s_w[wave] = __popcll(m);
    __syncthreads();
    int off = s_base;
    for (int w = 0; w < wave; ++w) off += s_w[w];
    bool act = false;
    int g = 0;
    if (front) {
      const long rank_in_chunk = (long)prank[p] + off + __popcll(m & ((1ull << lane) - 1ull));
      const long i = coff[c] + rank_in_chunk;
      g = goff[c] + (int)(rank_in_chunk / batch);
      if (g >= max_groups) g = max_groups - 1;
      pair_p[i] = (int)p;
      pair_j[i] = (unsigned short)j;
      grp[i] = g;
      const float o[3] = {ox + d[0] * 0.005f, oy + d[1] * 0.005f, oz + d[2] * 0.005f};
      const RayState s = cast_init(T, o, d);
      t_st[i] = s.t;
      leaf_st[i] = s.leaf;
      act_st[i] = s.active;
      act = s.active;
    }
    if (act) {
      if (g == gfirst) ++a0; else ++a1;
    }
    __syncthreads();
    if (tid == 0) s_base += s_w[0] + s_w[1] + s_w[2] + s_w[3];
    __syncthreads();
  }
  for (int o = 32; o > 0; o >>= 1) {
    a0 += __shfl_xor(a0, o);
    a1 += __shfl_xor(a1, o);
  }
  __shared__ int s_a[4][2];
  if (lane == 0) {
    s_a[wave][0] = a0;
    s_a[wave][1] = a1;
  }
  __syncthreads();
  if (tid == 0) {
    const int t0 = s_a[0][0] + s_a[1][0] + s_a[2][0] + s_a[3][0], t1 = s_a[0][1] + s_a[1][1] + s_a[2][1] + s_a[3][1];
    if (t0) atomicAdd(&counters[(long)gfirst * OV_ITERS], t0);
    if (t1) atomicAdd(&counters[(long)(gfirst + 1 < max_groups ? gfirst + 1 : max_groups - 1) * OV_ITERS], t1);
    point_span[p] = make_long2(first, (long)s_base);
  }
}

// one lock-step iteration `it` of every group (octree.py:528-573): rays [0, total) in a grid-stride loop
template <class Rays>
__global__ __launch_bounds__(256) void k_ovis_iter(Oct T, Rays rays, const long* __restrict__ total_ptr, long total_fixed,
                                                    const long* __restrict__ gsize, const int* __restrict__ grp, int it,
                                                    float* __restrict__ t_st, int* __restrict__ leaf_st,
                                                    unsigned char* __restrict__ act_st, int* __restrict__ counters,
                                                    OvisLayout* __restrict__ stats) {
  const long total = total_ptr ? *total_ptr : total_fixed;
  const long stride = (long)gridDim.x * blockDim.x;
  const long rounds = (total + stride - 1) / stride;
  int nfetch = 0, nstep = 0;
  for (long r = 0; r < rounds; ++r) {
    const long i = r * stride + blockIdx.x * (long)blockDim.x + threadIdx.x;
    bool act = false;
    int g = 0;
    if (i < total && act_st[i]) {
      g = grp[i];
      const int n_act = counters[(long)g * OV_ITERS + it];
      const long R = gsize[g];
      float o[3], d[3];
      rays.fetch(i, o, d);
#pragma unroll
      for (int c = 0; c < 3; ++c) o[c] = o[c] + d[c] * 0.005f;
      RayState s;
      s.t = t_st[i];
      s.leaf = leaf_st[i];
      s.active = true;
      cast_step(T, o, d, s, multi_samp(R, n_act), R > 100000 ? 0.01 : 0.005, &nfetch);
      nstep += 1;
      t_st[i] = s.t;
      leaf_st[i] = s.leaf;
      act_st[i] = s.active;
      act = s.active;
    }
    count_active(act, g, counters, it + 1);
  }
  if (stats) ovis_stats_add(stats, nfetch, nstep);
}

// ---- stable compaction of the rays still active (round 3).  k_ovis_iter above walks ALL pairs at every one of the 33 iterations and
// half of what it reads belongs to rays that have finished; their lanes idle inside the waves of the others.  Here every iteration
// walks a dense, ORDER-PRESERVING list of the active pairs (neighbours in the list are neighbours in (point, direction) order: their
// octree walks stay coherent -- an unordered append by atomics was measured slower than no compaction at all): the
// step kernel writes one flag per list entry, three small kernels turn (list, flags) into the next list: per-block counts, a
// single-workgroup scan of the counts, an ordered scatter.  Sizes live in device memory (n_alive[2], ping-pong); grids are fixed.
constexpr int OV_CB = 2048;     // list entries per compaction block (256 threads x 8)

template <class Rays>
__global__ __launch_bounds__(256) void k_ovis_iter_list(Oct T, Rays rays, const long* __restrict__ n_alive,
                                                         const int* __restrict__ list_in, const long* __restrict__ gsize,
                                                         const int* __restrict__ grp, int it, float* __restrict__ t_st,
                                                         int* __restrict__ leaf_st, unsigned char* __restrict__ flag_out,
                                                         int* __restrict__ counters, OvisLayout* __restrict__ stats) {
  // Every workgroup walks ONE contiguous stretch of the list (waves still read 64 consecutive entries per step), so a wave meets
  // one or two lock-step groups per launch and adds its active count to a group's counter once per group, not once per step:
  // the counters are device-wide atomics on one address per (group, iteration), ~0.6 us each and serialised -- with a grid-stride
  // walk (a new group every other step) they put a floor of ~6 ms under every iteration launch whatever its size, which made a
  // view traced in chunk groups 3x slower than in one piece.  (A dynamic hand-out of the list through one device-wide cursor was
  // measured 4x slower still, for the same reason.)
  const long total = *n_alive;
  const long per_block = ((total + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
  const long k_begin = blockIdx.x * per_block, k_end = k_begin + per_block < total ? k_begin + per_block : total;
  int nfetch = 0, nstep = 0;
  int pend_g = -1, pend_n = 0;            // wave-uniform: active rays of group pend_g not yet added to its counter
  for (long kb = k_begin; kb < k_end; kb += 256) {
    const long k = kb + threadIdx.x;
    bool act = false;
    int g = 0;
    if (k < k_end) {
      const long i = list_in[k];
      g = grp[i];
      const int n_act = counters[(long)g * OV_ITERS + it];
      const long R = gsize[g];
      float o[3], d[3];
      rays.fetch(i, o, d);
#pragma unroll
      for (int c = 0; c < 3; ++c) o[c] = o[c] + d[c] * 0.005f;
      RayState s;
      s.t = t_st[i];
      s.leaf = leaf_st[i];
      s.active = true;
      cast_step(T, o, d, s, multi_samp(R, n_act), R > 100000 ? 0.01 : 0.005, &nfetch);
      nstep += 1;
      t_st[i] = s.t;
      leaf_st[i] = s.leaf;
      flag_out[k] = s.active;
      act = s.active;
    }
    const unsigned long long am = __ballot(act);
    if (am) {
      const int g0 = __shfl(g, __ffsll((long long)am) - 1);
      const unsigned long long same = __ballot(act && g == g0);
      if (act && g != g0) atomicAdd(&counters[(long)g * OV_ITERS + it + 1], 1);       // a wave across a group boundary: rare
      if (g0 != pend_g) {
        if (pend_n && (threadIdx.x & 63) == 0) atomicAdd(&counters[(long)pend_g * OV_ITERS + it + 1], pend_n);
        pend_g = g0;
        pend_n = 0;
      }
      pend_n += __popcll(same);
    }
  }
  if (pend_n && (threadIdx.x & 63) == 0) atomicAdd(&counters[(long)pend_g * OV_ITERS + it + 1], pend_n);
  if (stats) ovis_stats_add(stats, nfetch, nstep);
}

// active entries per block of OV_CB list entries
__global__ __launch_bounds__(256) void k_cmp_count(const long* __restrict__ n_ptr, const unsigned char* __restrict__ flags,
                                                    int* __restrict__ blk_cnt) {
  __shared__ int s_w[4];
  const long n = *n_ptr;
  const long nblk = (n + OV_CB - 1) / OV_CB;
  const int tid = threadIdx.x, lane = tid & 63;
  for (long b = blockIdx.x; b < nblk; b += gridDim.x) {
    const long k0 = b * OV_CB + (long)tid * 8;
    int c = 0;
    if (k0 + 8 <= n) {
      const unsigned long long v = *reinterpret_cast<const unsigned long long*>(flags + k0);     // eight 0/1 bytes
      c = __popcll(v & 0x0101010101010101ull);
    } else {
      for (int e = 0; e < 8; ++e) c += (k0 + e < n && flags[k0 + e]) ? 1 : 0;
    }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if (lane == 0) s_w[tid >> 6] = c;
    __syncthreads();
    if (tid == 0) blk_cnt[b] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    __syncthreads();
  }
}

// exclusive scan of the block counts (one workgroup; strips of 1024 with a running carry); total -> *n_out
__global__ __launch_bounds__(1024) void k_cmp_scan(const long* __restrict__ n_ptr, const int* __restrict__ blk_cnt,
                                                    long* __restrict__ blk_off, long* __restrict__ n_out) {
  __shared__ long s_wave[16];
  __shared__ long s_carry;
  const long nblk = (*n_ptr + OV_CB - 1) / OV_CB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (long base = 0; base < nblk; base += 1024) {
    const long b = base + tid;
    const long v = b < nblk ? blk_cnt[b] : 0;
    long incl = v;
    for (int o = 1; o < 64; o <<= 1) {
      const long u = __shfl_up(incl, o);
      if (lane >= o) incl += u;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    long off = 0;
    for (int w = 0; w < wave; ++w) off += s_wave[w];
    const long carry = s_carry;
    if (b < nblk) blk_off[b] = carry + off + incl - v;
    __syncthreads();
    if (tid == 1023) s_carry = carry + off + incl;
    __syncthreads();
  }
  if (tid == 0) *n_out = s_carry;
}

// ordered scatter: list_out[rank of k among the active entries] = list_in[k]  (list_in == nullptr: the identity list)
__global__ __launch_bounds__(256) void k_cmp_scatter(const long* __restrict__ n_ptr, const unsigned char* __restrict__ flags,
                                                      const int* __restrict__ list_in, const long* __restrict__ blk_off,
                                                      int* __restrict__ list_out) {
  __shared__ int s_w[4];
  const long n = *n_ptr;
  const long nblk = (n + OV_CB - 1) / OV_CB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (long b = blockIdx.x; b < nblk; b += gridDim.x) {
    const long k0 = b * OV_CB + (long)tid * 8;
    unsigned m = 0;                                   // bit e: entry k0 + e is active
    for (int e = 0; e < 8; ++e) m |= (k0 + e < n && flags[k0 + e]) ? (1u << e) : 0u;
    const int c = __popc(m);
    int incl = c;
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(incl, o);
      if (lane >= o) incl += u;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    int off = 0;
    for (int w = 0; w < wave; ++w) off += s_w[w];
    long dst = blk_off[b] + off + incl - c;
    for (int e = 0; e < 8; ++e)
      if (m >> e & 1u) list_out[dst++] = list_in ? list_in[k0 + e] : (int)(k0 + e);
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_ovis_reduce(const int* __restrict__ cid, long n, const float* __restrict__ wdir,
                                                      const float* __restrict__ wsum, const unsigned short* __restrict__ pair_j,
                                                      const int* __restrict__ leaf_st, const long2* __restrict__ point_span,
                                                      int L, int nsamp, int argmax_vis, float* __restrict__ vis_out) {
  __shared__ float vis_tab[OV_MAX_DIRS];
  const int tid = threadIdx.x;
  const long p = blockIdx.x;
  const int LS = L * nsamp;
  const int c = cid ? cid[p] : 0;
  for (int j = tid; j < LS; j += 256) vis_tab[j] = 0.f;
  __syncthreads();
  // logits [is_hit, ~is_hit] as floats (octree_tracing.py:85): softmax(.)[1] (sg_render.py:173) or argmax (:171)
  const float e1 = expf(-1.0f);
  const float v_hit = argmax_vis ? 0.f : e1 / (1.0f + e1);      // logits [1, 0]: exp(0 - 1) / (exp(0) + exp(-1))
  const float v_free = argmax_vis ? 1.f : 1.0f / (e1 + 1.0f);   // logits [0, 1]
  const long2 sp = point_span[p];
  for (long i = tid; i < sp.y; i += 256) vis_tab[pair_j[sp.x + i]] = leaf_st[sp.x + i] >= 0 ? v_hit : v_free;
  __syncthreads();
  if (tid < L) {
    const float* w = wdir + (long)c * LS + (long)tid * nsamp;
    float acc = 0.f;
    for (int k = 0; k < nsamp; ++k) acc += vis_tab[tid * nsamp + k] * w[k];
    vis_out[p * L + tid] = acc / wsum[(long)c * L + tid];
  }
}

// ---- explicit rays in groups (contiguous ranges gstart[g] .. gstart[g+1]): set-up and result
__global__ __launch_bounds__(256) void k_cast_grouped_init(Oct T, ExplicitRays rays, long R, const long* __restrict__ gstart,
                                                            int G, long* __restrict__ gsize, int* __restrict__ grp,
                                                            float* __restrict__ t_st, int* __restrict__ leaf_st,
                                                            unsigned char* __restrict__ act_st, int* __restrict__ counters) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i < G) gsize[i] = gstart[i + 1] - gstart[i];
  bool act = false;
  int g = 0;
  if (i < R) {
    int lo = 0, hi = G;                      // last g with gstart[g] <= i
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (gstart[mid] <= i) lo = mid; else hi = mid;
    }
    g = lo;
    grp[i] = g;
    float o[3], d[3];
    rays.fetch(i, o, d);
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = o[c] + d[c] * 0.005f;
    const RayState s = cast_init(T, o, d);
    t_st[i] = s.t;
    leaf_st[i] = s.leaf;
    act_st[i] = s.active;
    act = s.active;
  }
  count_active(act, g, counters, 0);
}

__global__ __launch_bounds__(256) void k_cast_grouped_finish(Oct T, ExplicitRays rays, long R, float clamp_dt,
                                                              const float* __restrict__ t_st, const int* __restrict__ leaf_st,
                                                              float* __restrict__ x_out, unsigned char* __restrict__ hit_out,
                                                              float* __restrict__ t_out) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= R) return;
  float o[3], oc[3], d[3];
  rays.fetch(i, o, d);
#pragma unroll
  for (int c = 0; c < 3; ++c) oc[c] = o[c] + d[c] * 0.005f;
  RayState s;
  s.t = t_st[i];
  s.leaf = leaf_st[i];
  s.active = false;
  cast_finish(T, oc, o, d, s, clamp_dt, x_out + 3 * i, hit_out + i, t_out + i);
}

}  // namespace rb

using namespace rb;

static int ovis_grid() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 2048;
    cus = prop.multiProcessorCount;
  }
  return cus * 8;
}

// the plain walk (every iteration visits every pair slot): rb_dvis_octree without the compaction scratch
static int dvis_octree_plain(const float* node, const float* nrm, long B, const float* root_min, const float* root_size, const int* res,
                   const float* points, const float* normals, const int* chunk_id, long n, int n_chunks, const float* dirs,
                   const float* wdir, const float* wsum, int L, int nsamp, int argmax_vis, long batch_pairs, int max_iter,
                   int* pcount, int* prank, long* chunk_tab, long* group_tab, int max_groups, int* counters, int* pair_p,
                   unsigned short* pair_j, float* t_st, int* leaf_st, unsigned char* act_st, int* grp, long* point_span,
                   long* layout, float* vis_out, unsigned long long* eval_count, rb_stream_t stream) {
  if (n <= 0) return 0;
  RB_REQUIRE(node && nrm && points && normals && dirs && wdir && wsum && vis_out, "null pointer");
  RB_REQUIRE(pcount && prank && chunk_tab && group_tab && counters && pair_p && pair_j && t_st && leaf_st && act_st && grp &&
                 point_span && layout,
             "null scratch pointer");
  RB_REQUIRE(n <= RB_MAX_BLOCKS, "too many points for one launch");
  RB_REQUIRE(L > 0 && L <= 256 && nsamp > 0 && (long)L * nsamp <= OV_MAX_DIRS, "need L <= 256 and L*nsamp <= 4096");
  RB_REQUIRE(n_chunks >= 1 && (chunk_id || n_chunks == 1), "chunk_id is required for more than one chunk");
  RB_REQUIRE(max_iter > 0 && max_iter + 2 <= OV_ITERS, "secondary cast: 0 < max_iter <= 32");
  RB_REQUIRE(batch_pairs > 0 && max_groups >= 1, "bad batch size / group capacity");
  hipStream_t s = (hipStream_t)stream;
  const int LS = L * nsamp;
  Oct T = make_oct(node, nrm, B, root_min, root_size, res);
  // chunk_tab: cstart[C+1] | ctotal[C] | coff[C] | goff[C] (as int, in a long slot each);  group_tab: gstart[G] | gsize[G]
  long* cstart = chunk_tab;
  long* ctotal = chunk_tab + (n_chunks + 1);
  long* coff = ctotal + n_chunks;
  int* goff = reinterpret_cast<int*>(coff + n_chunks);
  long* gstart = group_tab;
  long* gsize = group_tab + max_groups;
  OvisLayout* lay = reinterpret_cast<OvisLayout*>(layout);
  hipLaunchKernelGGL(k_ovis_count, dim3((unsigned)n), dim3(256), 0, s, normals, chunk_id, n, dirs, LS, pcount);
  if (int rc = check_launch("k_ovis_count")) return rc;
  hipLaunchKernelGGL(k_ovis_scan, dim3((unsigned)n_chunks), dim3(1024), 0, s, chunk_id, n, n_chunks, pcount, prank, cstart, ctotal);
  if (int rc = check_launch("k_ovis_scan")) return rc;
  hipLaunchKernelGGL(k_ovis_layout, dim3(1), dim3(256), 0, s, n_chunks, ctotal, batch_pairs, coff, goff, gstart, gsize, lay,
                     counters, max_groups, eval_count);
  if (int rc = check_launch("k_ovis_layout")) return rc;
  hipLaunchKernelGGL(k_ovis_fill, dim3((unsigned)n), dim3(256), 0, s, T, points, normals, chunk_id, n, dirs, LS, prank, coff,
                     goff, batch_pairs, max_groups, pair_p, pair_j, t_st, leaf_st, act_st, grp,
                     reinterpret_cast<long2*>(point_span), counters);
  if (int rc = check_launch("k_ovis_fill")) return rc;
  PairRays rays{points, dirs, chunk_id, pair_p, pair_j, LS};
  const int grid = ovis_grid();
  for (int it = 0; it <= max_iter; ++it)      // the reference leaves its loop when it > max_iter: max_iter + 1 iterations
    hipLaunchKernelGGL(k_ovis_iter<PairRays>, dim3((unsigned)grid), dim3(256), 0, s, T, rays, &lay->total_pairs, 0L, gsize, grp, it,
                       t_st, leaf_st, act_st, counters, lay);
  if (int rc = check_launch("k_ovis_iter")) return rc;
  hipLaunchKernelGGL(k_ovis_reduce, dim3((unsigned)n), dim3(256), 0, s, chunk_id, n, wdir, wsum, pair_j, leaf_st,
                     reinterpret_cast<const long2*>(point_span), L, nsamp, argmax_vis, vis_out);
  if (int rc = check_launch("k_ovis_reduce")) return rc;
  return 0;
}

extern "C" {

/* With the compaction scratch (alive_a .. n_alive all non-NULL) the active rays are compacted between the lock-step iterations (same
 * results bit for bit: a ray's state never depends on where it sits in a launch): alive_a, alive_b int32[cap], flags uint8[cap] (cap = the
 * size of pair_p), blk_cnt int32[cap / 2048 + 2], blk_off int64[cap / 2048 + 2], n_alive int64[40] (list sizes + per-iteration cursors).
 * All six NULL: the plain walk. */
int rb_dvis_octree(const float* node, const float* nrm, long B, const float* root_min, const float* root_size, const int* res,
                           const float* points, const float* normals, const int* chunk_id, long n, int n_chunks, const float* dirs,
                           const float* wdir, const float* wsum, int L, int nsamp, int argmax_vis, long batch_pairs, int max_iter,
                           int* pcount, int* prank, long* chunk_tab, long* group_tab, int max_groups, int* counters, int* pair_p,
                           unsigned short* pair_j, float* t_st, int* leaf_st, unsigned char* act_st, int* grp, long* point_span,
                           long* layout, int* alive_a, int* alive_b, unsigned char* flags, int* blk_cnt, long* blk_off, long* n_alive,
                           float* vis_out, unsigned long long* eval_count, rb_stream_t stream) {
  if (n <= 0) return 0;
  if (!alive_a && !alive_b && !flags && !blk_cnt && !blk_off && !n_alive)
    return dvis_octree_plain(node, nrm, B, root_min, root_size, res, points, normals, chunk_id, n, n_chunks, dirs, wdir, wsum, L, nsamp,
                             argmax_vis, batch_pairs, max_iter, pcount, prank, chunk_tab, group_tab, max_groups, counters, pair_p, pair_j,
                             t_st, leaf_st, act_st, grp, point_span, layout, vis_out, eval_count, stream);
  RB_REQUIRE(node && nrm && points && normals && dirs && wdir && wsum && vis_out, "null pointer");
  RB_REQUIRE(pcount && prank && chunk_tab && group_tab && counters && pair_p && pair_j && t_st && leaf_st && act_st && grp &&
                 point_span && layout && alive_a && alive_b && flags && blk_cnt && blk_off && n_alive,
             "null scratch pointer");
  RB_REQUIRE(n <= RB_MAX_BLOCKS, "too many points for one launch");
  RB_REQUIRE(L > 0 && L <= 256 && nsamp > 0 && (long)L * nsamp <= OV_MAX_DIRS, "need L <= 256 and L*nsamp <= 4096");
  RB_REQUIRE(n_chunks >= 1 && (chunk_id || n_chunks == 1), "chunk_id is required for more than one chunk");
  RB_REQUIRE(max_iter > 0 && max_iter + 2 <= OV_ITERS, "secondary cast: 0 < max_iter <= 32");
  RB_REQUIRE(batch_pairs > 0 && max_groups >= 1, "bad batch size / group capacity");
  RB_REQUIRE((long)n * L * nsamp < 2147483647L, "pair indices are 32-bit: call in groups of chunks");
  hipStream_t s = (hipStream_t)stream;
  const int LS = L * nsamp;
  Oct T = make_oct(node, nrm, B, root_min, root_size, res);
  long* cstart = chunk_tab;
  long* ctotal = chunk_tab + (n_chunks + 1);
  long* coff = ctotal + n_chunks;
  int* goff = reinterpret_cast<int*>(coff + n_chunks);
  long* gstart = group_tab;
  long* gsize = group_tab + max_groups;
  OvisLayout* lay = reinterpret_cast<OvisLayout*>(layout);
  hipLaunchKernelGGL(k_ovis_count, dim3((unsigned)n), dim3(256), 0, s, normals, chunk_id, n, dirs, LS, pcount);
  if (int rc = check_launch("k_ovis_count")) return rc;
  hipLaunchKernelGGL(k_ovis_scan, dim3((unsigned)n_chunks), dim3(1024), 0, s, chunk_id, n, n_chunks, pcount, prank, cstart, ctotal);
  if (int rc = check_launch("k_ovis_scan")) return rc;
  hipLaunchKernelGGL(k_ovis_layout, dim3(1), dim3(256), 0, s, n_chunks, ctotal, batch_pairs, coff, goff, gstart, gsize, lay,
                     counters, max_groups, eval_count);
  if (int rc = check_launch("k_ovis_layout")) return rc;
  hipLaunchKernelGGL(k_ovis_fill, dim3((unsigned)n), dim3(256), 0, s, T, points, normals, chunk_id, n, dirs, LS, prank, coff,
                     goff, batch_pairs, max_groups, pair_p, pair_j, t_st, leaf_st, act_st, grp,
                     reinterpret_cast<long2*>(point_span), counters);
  if (int rc = check_launch("k_ovis_fill")) return rc;
  PairRays rays{points, dirs, chunk_id, pair_p, pair_j, LS};
  const int grid = ovis_grid();
  // first list: the pairs the ray set-up left active (identity list, flags = act_st)
  hipLaunchKernelGGL(k_cmp_count, dim3((unsigned)grid), dim3(256), 0, s, &lay->total_pairs, act_st, blk_cnt);
  hipLaunchKernelGGL(k_cmp_scan, dim3(1), dim3(1024), 0, s, &lay->total_pairs, blk_cnt, blk_off, n_alive);
  hipLaunchKernelGGL(k_cmp_scatter, dim3((unsigned)grid), dim3(256), 0, s, &lay->total_pairs, act_st, (const int*)nullptr, blk_off, alive_a);
  if (int rc = check_launch("k_cmp_*")) return rc;
  int* lin = alive_a;
  int* lout = alive_b;
  for (int it = 0; it <= max_iter; ++it) {      // the reference leaves its loop when it > max_iter: max_iter + 1 iterations
    long* n_in = n_alive + (it & 1);
    long* n_out = n_alive + ((it + 1) & 1);
    hipLaunchKernelGGL(k_ovis_iter_list<PairRays>, dim3((unsigned)grid), dim3(256), 0, s, T, rays, n_in, lin, gsize, grp, it, t_st,
                       leaf_st, flags, counters, lay);
    if (it < max_iter) {
      hipLaunchKernelGGL(k_cmp_count, dim3((unsigned)grid), dim3(256), 0, s, n_in, flags, blk_cnt);
      hipLaunchKernelGGL(k_cmp_scan, dim3(1), dim3(1024), 0, s, n_in, blk_cnt, blk_off, n_out);
      hipLaunchKernelGGL(k_cmp_scatter, dim3((unsigned)grid), dim3(256), 0, s, n_in, flags, lin, blk_off, lout);
      int* t = lin;
      lin = lout;
      lout = t;
    }
  }
  if (int rc = check_launch("k_ovis_iter_list")) return rc;
  hipLaunchKernelGGL(k_ovis_reduce, dim3((unsigned)n), dim3(256), 0, s, chunk_id, n, wdir, wsum, pair_j, leaf_st,
                     reinterpret_cast<const long2*>(point_span), L, nsamp, argmax_vis, vis_out);
  if (int rc = check_launch("k_ovis_reduce")) return rc;
  return 0;
}

int rb_octree_cast_grouped(const float* node, const float* nrm, long B, const float* root_min, const float* root_size,
                           const int* res, const float* origins, const float* dirs, long R, const long* group_start, int G,
                           int max_iter, float clamp_dt, long* gsize, int* grp, float* t_st, int* leaf_st,
                           unsigned char* act_st, int* counters, float* x_out, unsigned char* hit_out, float* t_out,
                           rb_stream_t stream) {
  if (R <= 0) return 0;
  RB_REQUIRE(node && nrm && origins && dirs && group_start && gsize && grp && t_st && leaf_st && act_st && counters && x_out &&
                 hit_out && t_out,
             "null pointer");
  RB_REQUIRE(G >= 1 && max_iter > 0 && max_iter + 2 <= OV_ITERS, "grouped cast: secondary mode, 0 < max_iter <= 32");
  hipStream_t s = (hipStream_t)stream;
  Oct T = make_oct(node, nrm, B, root_min, root_size, res);
  if (hipMemsetAsync(counters, 0, (size_t)G * OV_ITERS * sizeof(int), s) != hipSuccess) return rb::fail(__func__, "memset failed");
  ExplicitRays rays{origins, dirs};
  const long work = R > G ? R : G;
  hipLaunchKernelGGL(k_cast_grouped_init, grid1d(work, 256), dim3(256), 0, s, T, rays, R, group_start, G, gsize, grp, t_st, leaf_st,
                     act_st, counters);
  if (int rc = check_launch("k_cast_grouped_init")) return rc;
  const long blocks = (R + 255) / 256;
  const int grid = (int)(blocks < ovis_grid() ? blocks : ovis_grid());
  for (int it = 0; it <= max_iter; ++it)
    hipLaunchKernelGGL(k_ovis_iter<ExplicitRays>, dim3((unsigned)grid), dim3(256), 0, s, T, rays, (const long*)nullptr, R, gsize,
                       grp, it, t_st, leaf_st, act_st, counters, (OvisLayout*)nullptr);
  if (int rc = check_launch("k_ovis_iter")) return rc;
  hipLaunchKernelGGL(k_cast_grouped_finish, grid1d(R, 256), dim3(256), 0, s, T, rays, R, clamp_dt, t_st, leaf_st, x_out, hit_out,
                     t_out);
  return check_launch("k_cast_grouped_finish");
}

}  // extern "C"
