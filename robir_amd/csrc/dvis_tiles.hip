// The two passes around the persistent tile-list ("stream") form of the light-visibility kernels -- k_dvis_x6t<stream>
// (vis_diffuse_x6t.hip), k_dvis_f16t (vis_diffuse_f16t.hip) and the legacy k_dvis3_stream (vis_diffuse_v3.hip, where they came from):
//   k_dvis3_cull    one workgroup per point: cull n.d <= 1e-6 (model/sg_render.py:155), compact the surviving direction indices into a
//                   GLOBAL list of 16-sample tiles (wave ballot + prefix count, one atomic per point for its tile range);
//   k_dvis3_reduce  one workgroup per point: scatter its pair values by direction index, SG-weighted mean per lobe in the fixed sample
//                   order (sg_render.py:177-190).
#include "../../include/robir_hip.h"
#include "common.h"

namespace rb {

#define RB_TINY 1e-6f
constexpr int V3_MAX_DIRS = 4096;

struct V3Tile {
  int point;       // -1: no such tile
  int dir_base;    // first row of the point's chunk in dirs / Bd (chunk id * L * nsamp)
};

// ---------------------------------------------------------------------------------------------------------------------
// cull + compaction: one workgroup per point
// counters[0] = tiles allocated so far (atomic), counters[1] = surviving pairs (statistics)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_dvis3_cull(const float* __restrict__ normals, const int* __restrict__ cid, long n,
                                                     const float* __restrict__ dirs, int LS, unsigned short* __restrict__ pair_j,
                                                     V3Tile* __restrict__ tile_info, int2* __restrict__ point_info,
                                                     unsigned long long* __restrict__ counters,
                                                     unsigned long long* __restrict__ eval_count) {
  __shared__ unsigned short idx_list[V3_MAX_DIRS];
  __shared__ int s_count, s_tile0;
  const int tid = threadIdx.x, lane = tid & 63;
  const long p = blockIdx.x;
  const int dbase = (cid ? cid[p] : 0) * LS;
  if (tid == 0) s_count = 0;
  __syncthreads();
  const float nx = normals[3 * p], ny = normals[3 * p + 1], nz = normals[3 * p + 2];
  for (int j0 = 0; j0 < LS; j0 += 256) {
    const int j = j0 + tid;
    bool front = false;
    if (j < LS) {
      const float* d = dirs + 3 * ((long)dbase + j);
      const float c = nx * d[0] + ny * d[1] + nz * d[2];  // sum(n*d): separate mul/add (-ffp-contract=off)
      front = c > RB_TINY;
    }
    const unsigned long long m = __ballot(front);
    int base = 0;
    if (lane == 0 && m) base = atomicAdd(&s_count, __popcll(m));
    base = __shfl(base, 0);
    if (front) idx_list[base + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)j;
  }
  __syncthreads();
  const int S = s_count;
  const int nt = (S + 15) >> 4;
  if (tid == 0) {
    s_tile0 = nt ? (int)atomicAdd(&counters[0], (unsigned long long)nt) : 0;
    atomicAdd(&counters[1], (unsigned long long)S);
    if (eval_count) atomicAdd(eval_count, (unsigned long long)S);
  }
  __syncthreads();
  const int t0 = s_tile0;
  if (tid == 0) point_info[p] = make_int2(t0, S);
  for (int i = tid; i < nt * 16; i += 256) pair_j[(long)t0 * 16 + i] = i < S ? idx_list[i] : (unsigned short)0xFFFF;
  for (int i = tid; i < nt; i += 256) tile_info[t0 + i] = V3Tile{(int)p, dbase};
}

// ---------------------------------------------------------------------------------------------------------------------
// per point: scatter the pair values by direction index, SG-weighted mean per lobe in the fixed sample order
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_dvis3_reduce(const int* __restrict__ cid, long n, const float* __restrict__ wdir,
                                                       const float* __restrict__ wsum, const unsigned short* __restrict__ pair_j,
                                                       const float* __restrict__ pair_vis, const int2* __restrict__ point_info,
                                                       int L, int nsamp, float* __restrict__ vis_out) {
  __shared__ float vis_tab[V3_MAX_DIRS];
  const int tid = threadIdx.x;
  const long p = blockIdx.x;
  const int LS = L * nsamp;
  const int c = cid ? cid[p] : 0;
  for (int j = tid; j < LS; j += 256) vis_tab[j] = 0.f;
  __syncthreads();
  const int2 pi = point_info[p];
  const long base = (long)pi.x * 16;
  for (int i = tid; i < pi.y; i += 256) vis_tab[pair_j[base + i]] = pair_vis[base + i];
  __syncthreads();
  if (tid < L) {
    const float* w = wdir + (long)c * LS + (long)tid * nsamp;
    float acc = 0.f;
    for (int k = 0; k < nsamp; ++k) acc += vis_tab[tid * nsamp + k] * w[k];
    vis_out[p * L + tid] = acc / wsum[(long)c * L + tid];
  }
}

}  // namespace rb
