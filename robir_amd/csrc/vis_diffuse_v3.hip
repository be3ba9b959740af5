// Light-SG visibility, third-generation kernel family (get_diffuse_visibility, model/sg_render.py:111-195).
//
// Same arithmetic as k_dvis_v2 (vis_diffuse_v2.hip) -- every (point, direction) pair goes through the same split-precision
// instruction sequence, so results are bit-identical -- but the unit of work is no longer "one workgroup = one point":
//
//   k_dvis3_cull    one workgroup per point: cull n.d <= 1e-6 (sg_render.py:155), compact the surviving direction indices
//                   into a GLOBAL list of 16-sample TILES (a point's tiles are contiguous; its last tile is padded with
//                   0xFFFF), tile -> (point, table base) records, per-point (first tile, count);
//   k_dvis3_stream  PERSISTENT grid, one workgroup per CU: round r = tiles 8r .. 8r+7 (4 waves x 2 tiles) of the global
//                   list, workgroup b takes rounds b, b+G, b+2G, ...  A round's tiles may belong to different points: each
//                   tile has its own layer-0 point row (fetched by LDS-DMA into a wave-private LDS slot one round ahead).
//                   The weight ring runs continuously across rounds.  Per-pair visibilities go to a global array;
//   k_dvis3_reduce  one workgroup per point: scatter its pair values by direction index, SG-weighted mean per lobe in the
//                   fixed sample order (identical summation order to the fused kernels).
//
// Why: (1) tile padding drops from "last round of every point" (S ~ 2053 of 4096 directions -> half the points carried a
// nearly empty 17th round, ~3 %) to "last tile of every point" (0.4 %); (2) the work is balanced over the CUs whatever the
// number of points -- a single 1024-pixel chunk (~660 points) filled 2.6 waves of one-point workgroups before; (3) no
// host-visible sizes anywhere: the grid is fixed, the tile count is read from device memory, so the caller never
// synchronises.
#include "../../include/robir_hip.h"
#include "common.h"
#include "mlp_engine.h"
#include <cstdlib>

namespace rb {

#define RB_TINY 1e-6f
constexpr int V3_MAX_DIRS = 4096;
constexpr int V3_CF4 = chunk_f4(256);   // float4s per packed chunk in global memory (bias + weights)
constexpr int V3_WF4 = 1024;            // weight part of a chunk (16 KB)
constexpr int V3_SLOTS = 4, V3_DIST = 3;

struct V3Tile {
  int point;       // -1: no such tile
  int dir_base;    // first row of the point's chunk in dirs / Bd (chunk id * L * nsamp)
};

// k_dvis3_cull / k_dvis3_reduce: the passes around the persistent tile-list kernel, shared with the exact-operand and f16 forms of the
// default library -- defined in dvis_tiles.hip (round 5: this file is legacy-only)
__global__ void k_dvis3_cull(const float* __restrict__ normals, const int* __restrict__ cid, long n, const float* __restrict__ dirs, int LS,
                             unsigned short* __restrict__ pair_j, V3Tile* __restrict__ tile_info, int2* __restrict__ point_info,
                             unsigned long long* __restrict__ counters, unsigned long long* __restrict__ eval_count);
__global__ void k_dvis3_reduce(const int* __restrict__ cid, long n, const float* __restrict__ wdir, const float* __restrict__ wsum,
                               const unsigned short* __restrict__ pair_j, const float* __restrict__ pair_vis,
                               const int2* __restrict__ point_info, int L, int nsamp, float* __restrict__ vis_out);

// global -> LDS copy of 16 B per lane: wave-uniform LDS base in M0, uniform global base + per-lane byte offset
__device__ __forceinline__ void v3_dma16(const f4* gbase_uniform, unsigned lane_byte_off, unsigned lds_byte_uniform) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_byte_uniform), "v"(lane_byte_off),
               "s"(gbase_uniform)
               : "memory");
}

// hi/lo split of two fp32 values in 3 VALU ops (see vis_diffuse_v2.hip)
__device__ __forceinline__ void v3_split_pair(float v0, float v1, unsigned& hi, unsigned& lo) {
  const h2 h = __builtin_amdgcn_cvt_pkrtz(v0, v1);
  const unsigned hu = __builtin_bit_cast(unsigned, h);
  unsigned l;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(hu), "v"(v0));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(hu), "v"(v1));
  hi = hu;
  lo = l;
}

struct V3Acc {
  f4 a[2];
};

// SENT: accumulate the activation-range sentinel (mlp_engine.h) -- a template switch so that its cost can be A/B-measured
template <bool SENT>
__global__ __launch_bounds__(256, 1) void k_dvis3_stream(
    const float* __restrict__ A, const float* __restrict__ Bd, const f4* __restrict__ W49,
    const unsigned short* __restrict__ pair_j, const V3Tile* __restrict__ tile_info,
    const unsigned long long* __restrict__ counters, int argmax_vis, float w_unscale, float* __restrict__ pair_vis,
    unsigned* __restrict__ range_word) {
  __shared__ f4 ring[V3_SLOTS * V3_WF4];   // 64 KB
  __shared__ f4 headw[V3_WF4];             // 16 KB: chunk 48 (256 -> 2 head, rows 2..15 zero)
  __shared__ f4 bias_tab[49 * 4];
  __shared__ f4 a_rows[2 * 8 * 64];        // 16 KB: [round parity][tile of the round][256 floats]
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = gridDim.x, b = blockIdx.x;
  const long total_tiles = (long)counters[0];
  const long total_rounds = (total_tiles + 7) >> 3;
  for (int i = tid; i < 49 * 4; i += 256) bias_tab[i] = W49[(long)(i >> 2) * V3_CF4 + (i & 3)];
  for (int i = tid; i < V3_WF4; i += 256) headw[i] = W49[48L * V3_CF4 + 4 + i];
  __syncthreads();
  if (b >= total_rounds) return;           // workgroup-uniform

  // ---- weight ring state
  const unsigned ring_b = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)ring);
  const unsigned arow_b = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)a_rows);
  const unsigned lane_off = (unsigned)tid * 16u;                  // byte offset of this lane inside a 4 KB DMA row
  const unsigned wave_lds = ring_b + (unsigned)wave * 1024u;      // + slot * 16384 + i * 4096
  auto dma_chunk = [&](const f4* chunk_weights_uniform, int slot) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      v3_dma16(chunk_weights_uniform + i * 256, lane_off, wave_lds + (unsigned)slot * 16384u + (unsigned)i * 4096u);
  };
  const u4* ring_u = reinterpret_cast<const u4*>(ring) + lane;
  u4 wreg[16];
  f4 bias;

  // ---- per-round tile state (wave-uniform records; per-lane direction index)
  int tp[2], tb[2];        // point / table base of this wave's two tiles in the CURRENT round
  int tpn[2], tbn[2];      // ... in the NEXT round of this workgroup (r + G)
  int jraw[2];             // direction index of this lane's sample (lane & 15) in the NEXT round, 0xFFFF = padding
  int jraw2[2];            // ... in the round after next
  auto tile_rec = [&](long round, int t, int& pt, int& base) {
    const long T = round * 8 + wave * 2 + t;
    V3Tile rec{-1, 0};
    if (round < total_rounds && T < total_tiles) rec = tile_info[T];   // wave-uniform address: scalar load
    pt = __builtin_amdgcn_readfirstlane(rec.point);
    base = __builtin_amdgcn_readfirstlane(rec.dir_base);
  };
  auto load_idx = [&](long round, int t) -> int {
    const long T = round * 8 + wave * 2 + t;
    return (round < total_rounds && T < total_tiles) ? (int)pair_j[T * 16 + (lane & 15)] : 0xFFFF;
  };
  // A rows of a round's two tiles -> this wave's private slots of a_rows[parity] by LDS-DMA (1 KB per tile)
  auto dma_arows = [&](const int (&pt)[2], int parity) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const long prow = pt[t] < 0 ? 0L : (long)pt[t];
      v3_dma16(reinterpret_cast<const f4*>(A + prow * 256), (unsigned)lane * 16u,
               arow_b + (unsigned)(parity * 8 + wave * 2 + t) * 1024u);
    }
  };

  unsigned sat = 0u;                   // range sentinel: running max of the hi halves (all >= 0 here: ReLU outputs)
  // 128-bit tuples (one MFMA B operand each), declared as vectors so that a k-block's four registers stay contiguous
  u4 xh[2][8], xl[2][8];   // B operands of the current layer (packed hi / lo halves)
  u4 yh[2][8], yl[2][8];   // ... of the next layer, filled chunk by chunk

  auto mfma_kb = [&](int kb, V3Acc& acc, const u4 (&wsrc)[16]) {
    const h8 wh = __builtin_bit_cast(h8, wsrc[kb * 2]);
    const h8 wlo = __builtin_bit_cast(h8, wsrc[kb * 2 + 1]);
    h8 a[2], bb[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      a[t] = __builtin_bit_cast(h8, xh[t][kb]);
      bb[t] = __builtin_bit_cast(h8, xl[t][kb]);
    }
    // product order hi*lo, hi*hi, lo*hi into ONE accumulator per tile, the three MFMAs of a tile back to back (see v2)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      acc.a[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, bb[t], acc.a[t], 0, 0, 0);
      acc.a[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, a[t], acc.a[t], 0, 0, 0);
      acc.a[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo, a[t], acc.a[t], 0, 0, 0);
      if (t == 0) asm volatile("" : "+a"(acc.a[0]), "+a"(acc.a[1]));
    }
  };
  auto epilogue_piece = [&](const V3Acc& acc, int jb, int piece) {
    const int t = piece >> 1, q = piece & 1;
    const float v0 = fmaxf(acc.a[t][2 * q] * w_unscale, 0.f), v1 = fmaxf(acc.a[t][2 * q + 1] * w_unscale, 0.f);
    unsigned hi, lo;
    v3_split_pair(v0, v1, hi, lo);
    yh[t][jb >> 1][(jb & 1) * 2 + q] = hi;
    yl[t][jb >> 1][(jb & 1) * 2 + q] = lo;
    if constexpr (SENT) sat = sat_acc_nonneg(sat, hi);
  };
  auto epilogue = [&](const V3Acc& acc, int jb) {
#pragma unroll
    for (int pc = 0; pc < 4; ++pc) epilogue_piece(acc, jb, pc);
  };

  // rows of the per-direction table of the NEXT round, fetched a layer ahead into `raw`
  f4 raw[2][16];
  auto fetch_rows = [&]() {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int j = (jraw[t] == 0xFFFF) ? 0 : jraw[t];
      const f4* brow = reinterpret_cast<const f4*>(Bd + ((long)tbn[t] + j) * 256) + g;
#pragma unroll
      for (int kb = 0; kb < 16; ++kb) raw[t][kb] = brow[kb * 4];
    }
  };

  // ---- prologue: ring start, first round's records / indices / rows / A rows
  long rd = b;
  tile_rec(rd, 0, tpn[0], tbn[0]);
  tile_rec(rd, 1, tpn[1], tbn[1]);
  jraw[0] = load_idx(rd, 0);
  jraw[1] = load_idx(rd, 1);
  jraw2[0] = load_idx(rd + G, 0);
  jraw2[1] = load_idx(rd + G, 1);
  dma_chunk(W49 + 0L * V3_CF4 + 4, 0);
  dma_chunk(W49 + 1L * V3_CF4 + 4, 1);
  dma_arows(tpn, 0);
  fetch_rows();
  dma_chunk(W49 + 2L * V3_CF4 + 4, 2);   // stays in flight: the first chunk waits for everything older only
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int i = 0; i < 16; ++i) wreg[i] = ring_u[i * 64];
  bias = bias_tab[g];

  int parity = 0;
  for (; rd < total_rounds; rd += G) {
    // ---- this round's records; the next round's (rd + G) are looked up now and used by the fetches of layer 1
    int jj[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      tp[t] = tpn[t];
      tb[t] = tbn[t];
      jj[t] = (tp[t] < 0 || jraw[t] == 0xFFFF) ? -1 : jraw[t];
      jraw[t] = jraw2[t];
    }
    tile_rec(rd + G, 0, tpn[0], tbn[0]);
    tile_rec(rd + G, 1, tpn[1], tbn[1]);
    // ---- layer 0: relu(A[point of the tile] + Bd[dir]) straight into the operand registers.  The A rows were copied into
    // this wave's LDS slots by its own LDS-DMA during the previous round (prologue for the first): older than every row load
    // of `raw`, whose wait the compiler places below
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    const f4* arow = a_rows + (parity * 8 + wave * 2) * 64;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int kb = 0; kb < 16; ++kb) {
        const f4 bv = raw[t][kb];
        const f4 av = arow[t * 64 + kb * 4 + g];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          unsigned hi, lo;
          v3_split_pair(fmaxf(av[2 * q] + bv[2 * q], 0.f), fmaxf(av[2 * q + 1] + bv[2 * q + 1], 0.f), hi, lo);
          xh[t][kb / 2][(kb & 1) * 2 + q] = hi;
          xl[t][kb / 2][(kb & 1) * 2 + q] = lo;
          if constexpr (SENT) sat = sat_acc_nonneg(sat, hi);
        }
      }
    }
#pragma unroll 1
    for (int l = 0; l < 3; ++l) {
      const f4* Wl = W49 + (long)l * 16 * V3_CF4 + 4;                          // this layer's chunk 0 weights
      const f4* Wn = W49 + (long)(l == 2 ? 0 : l + 1) * 16 * V3_CF4 + 4;        // next layer's (next round wraps to 0)
      V3Acc prev;
      // next round's direction indices (two rounds ahead), A rows and table rows: issued in any case (after the last round
      // they re-fetch valid dummy rows), the counted waits below assume them
      if (l == 1) {
        jraw2[0] = load_idx(rd + 2L * G, 0);
        jraw2[1] = load_idx(rd + 2L * G, 1);
        dma_arows(tpn, parity ^ 1);
        fetch_rows();
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int jb = 0; jb < 16; ++jb) {
        V3Acc acc;
        acc.a[0] = bias;
        acc.a[1] = bias;
        // chunk jb+1 has landed in its slot once at most the copy of chunk jb+2 (4 instructions) is still in flight;
        // at the top of layer 1 the 2 index loads, 2 A-row copies and 32 row loads issued there are younger than the
        // copies the first two chunks wait for
        if (jb < 2 && l == 1) {
          asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int nx3 = jb + V3_DIST;
        const f4* dsrc = nx3 < 16 ? Wl + (long)nx3 * V3_CF4 : Wn + (long)(nx3 - 16) * V3_CF4;
        const unsigned ddst = wave_lds + (unsigned)(nx3 & 3) * 16384u;
        const int ns = (jb + 1) & 3;
        const f4 nbias = bias_tab[(l * 16 + jb + 1) * 4 + g];     // index 48 = head chunk after the last layer
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
          mfma_kb(kb, acc, wreg);
          wreg[2 * kb] = ring_u[ns * V3_WF4 + (2 * kb) * 64];
          wreg[2 * kb + 1] = ring_u[ns * V3_WF4 + (2 * kb + 1) * 64];
          if (jb > 0 && (kb & 1)) epilogue_piece(prev, jb - 1, kb >> 1);
          if (!(kb & 1)) v3_dma16(dsrc + (kb >> 1) * 256, lane_off, ddst + (unsigned)(kb >> 1) * 4096u);
          __builtin_amdgcn_sched_barrier(0);
        }
        prev = acc;
        bias = nbias;
      }
      epilogue(prev, 15);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int kb = 0; kb < 8; ++kb)
        {
          xh[t][kb] = yh[t][kb];
          xl[t][kb] = yl[t][kb];
        }
    }
    // ---- head: chunk 48 from its resident LDS copy; `bias` holds its bias, wreg the next round's chunk 0 fragments
    {
      const u4* hw = reinterpret_cast<const u4*>(headw) + lane;
      u4 hreg[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) hreg[i] = hw[i * 64];
      V3Acc acc;
      acc.a[0] = bias;
      acc.a[1] = bias;
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) mfma_kb(kb, acc, hreg);
      bias = bias_tab[g];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const f4 r = acc.a[t];
        const float l0 = r[0] * w_unscale, l1 = r[1] * w_unscale;
        if (g == 0 && jj[t] >= 0) {
          float v;
          if (argmax_vis) {
            v = l1 > l0 ? 1.f : 0.f;
          } else {
            const float mx = fmaxf(l0, l1);
            const float e0 = expf(l0 - mx), e1 = expf(l1 - mx);
            v = e1 / (e0 + e1);
          }
          pair_vis[(rd * 8 + wave * 2 + t) * 16 + (lane & 15)] = v;
        }
      }
    }
    parity ^= 1;
  }
  if constexpr (SENT) range_report<true>(sat, range_word);
  // drain the ring (copies still target this workgroup's LDS)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

}  // namespace rb

using namespace rb;

extern "C" int rb_dvis_stream(const float* normals, const int* chunk_id, long n, const float* A, const float* Bd,
                              const float* dirs, const float* wdir, const float* wsum, const float* W49, int L, int nsamp,
                              int argmax_vis, int scale_log2, unsigned short* pair_j, float* pair_vis, int* tile_info,
                              int* point_info, unsigned long long* counters, int n_workgroups, float* vis_out,
                              unsigned long long* eval_count, rb_stream_t stream) {
  if (n <= 0) return 0;
  RB_REQUIRE(normals && A && Bd && dirs && wdir && wsum && W49 && vis_out, "null pointer");
  RB_REQUIRE(pair_j && pair_vis && tile_info && point_info && counters, "null scratch pointer");
  RB_REQUIRE(n <= RB_MAX_BLOCKS, "too many points for one launch (one workgroup each in the cull / reduce passes)");
  RB_REQUIRE(L > 0 && L <= 256 && nsamp > 0 && (long)L * nsamp <= V3_MAX_DIRS && (L * nsamp) % 16 == 0,
             "need L <= 256, L*nsamp <= 4096 and a multiple of 16");
  RB_REQUIRE((long)n * (L * nsamp / 16) < (1L << 31), "tile index would overflow 31 bits");
  hipStream_t s = (hipStream_t)stream;
  if (n_workgroups <= 0) {
    static int cus = 0;
    if (!cus) {
      int dev = 0;
      hipDeviceProp_t prop;
      if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return rb::fail(__func__, "device query failed");
      cus = prop.multiProcessorCount;
    }
    n_workgroups = cus;
  }
  if (hipMemsetAsync(counters, 0, 2 * sizeof(unsigned long long), s) != hipSuccess) return rb::fail(__func__, "memset failed");
  hipLaunchKernelGGL(k_dvis3_cull, dim3((unsigned)n), dim3(256), 0, s, normals, chunk_id, n, dirs, L * nsamp, pair_j,
                     reinterpret_cast<V3Tile*>(tile_info), reinterpret_cast<int2*>(point_info), counters, eval_count);
  if (int rc = check_launch("k_dvis3_cull")) return rc;
  static const char* const nosent = getenv("RB_V3_NO_SENTINEL");   // A/B switch for measuring the sentinel's cost
  const float us = ldexpf(1.0f, -scale_log2);
  unsigned* rw = range_flags() ? range_flags() + RB_RANGE_DVIS : nullptr;
  if (nosent && nosent[0] == '1') {
    hipLaunchKernelGGL(k_dvis3_stream<false>, dim3((unsigned)n_workgroups), dim3(256), 0, s, A, Bd, (const f4*)W49, pair_j,
                       reinterpret_cast<const V3Tile*>(tile_info), counters, argmax_vis, us, pair_vis, rw);
  } else {
    hipLaunchKernelGGL(k_dvis3_stream<true>, dim3((unsigned)n_workgroups), dim3(256), 0, s, A, Bd, (const f4*)W49, pair_j,
                       reinterpret_cast<const V3Tile*>(tile_info), counters, argmax_vis, us, pair_vis, rw);
  }
  if (int rc = check_launch("k_dvis3_stream")) return rc;
  hipLaunchKernelGGL(k_dvis3_reduce, dim3((unsigned)n), dim3(256), 0, s, chunk_id, n, wdir, wsum, pair_j, pair_vis,
                     reinterpret_cast<const int2*>(point_info), L, nsamp, vis_out);
  return check_launch("k_dvis3_reduce");
}
