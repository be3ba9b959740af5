// Light-SG visibility (get_diffuse_visibility, model/sg_render.py:111-195): the second-generation kernel (vis_diffuse_v2.hip) as
// EIGHT waves of ONE 16-sample tile per workgroup.
//
// tools/ubench/valu_issue.hip and MI355X_MICROARCH.md: a wave that issues an LDS-DMA slice is held for ~100 cycles, and with one
// wave per SIMD nothing else issues meanwhile -- k_dvis_v2 pays four slices per wave and chunk, 8 of its 28.6 cycles per MFMA.
// Here two waves share a SIMD (half the operand registers each), a wave requests two slices per chunk, and one wave's MFMAs issue
// while the other waits for its slice to be accepted; weight fragments are read from the ring just before use instead of a chunk
// ahead (two k-blocks in registers instead of sixteen).  Same arithmetic in the same order on the same 128 samples per round:
// vis_out is bit-identical to k_dvis_v2's.
//
// Per point (one workgroup): cull + compaction of the front-facing directions, then rounds of 128 samples through
// layer 0 (relu(A[p] + Bd[dir]), factored first layer) -> three 256x256 ReLU layers as a stream of 48 chunks (16 neurons x 256)
// through a four-slot LDS ring filled by LDS-DMA three chunks ahead -> the 256 -> 2 head from its resident LDS copy -> softmax.
#include "../../include/robir_hip.h"
#include "common.h"
#include "mlp_engine.h"
#include <cstdlib>

namespace rb {

#define RB_TINY 1e-6f
constexpr int V4_MAX_DIRS = 4096;
constexpr int V4_CF4 = chunk_f4(256);   // float4s per packed chunk in global memory (bias + weights)
constexpr int V4_WF4 = 1024;            // weight part of a chunk (16 KB)

__device__ __forceinline__ void v4_dma16(const f4* gbase_uniform, unsigned lane_byte_off, unsigned lds_byte_uniform) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_byte_uniform), "v"(lane_byte_off),
               "s"(gbase_uniform)
               : "memory");
}
__device__ __forceinline__ void v4_split_pair(float v0, float v1, unsigned& hi, unsigned& lo) {
  const h2 h = __builtin_amdgcn_cvt_pkrtz(v0, v1);
  const unsigned hu = __builtin_bit_cast(unsigned, h);
  unsigned l;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(hu), "v"(v0));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(hu), "v"(v1));
  hi = hu;
  lo = l;
}

__global__ __launch_bounds__(512, 1) void k_dvis_v4(
    const float* __restrict__ normals, const int* __restrict__ cid, long n, const float* __restrict__ A,
    const float* __restrict__ Bd, const float* __restrict__ dirs, const float* __restrict__ wdir,
    const float* __restrict__ wsum, const f4* __restrict__ W49, int L, int nsamp, int argmax_vis, float w_unscale,
    float* __restrict__ vis_out, unsigned long long* __restrict__ eval_count, unsigned* __restrict__ range_word) {
  __shared__ f4 ring[4 * V4_WF4];          // 64 KB
  __shared__ f4 headw[V4_WF4];             // 16 KB: chunk 48 (256 -> 2 head, rows 2..15 zero)
  __shared__ f4 bias_tab[49 * 4];
  __shared__ float vis_tab[V4_MAX_DIRS];
  __shared__ unsigned short idx_list[V4_MAX_DIRS];
  __shared__ f4 a_row[64];
  __shared__ int s_count;
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // 0 .. 7: samples wave * 16 .. + 15 of a round
  const long p = blockIdx.x;
  const int LS = L * nsamp;
  const long dbase = (cid ? (long)cid[p] : 0L) * LS;
  if (tid == 0) s_count = 0;
  if (tid < 64) a_row[tid] = reinterpret_cast<const f4*>(A + p * 256)[tid];
  for (int i = tid; i < 49 * 4; i += 512) bias_tab[i] = W49[(long)(i >> 2) * V4_CF4 + (i & 3)];
  for (int i = tid; i < V4_WF4; i += 512) headw[i] = W49[48L * V4_CF4 + 4 + i];
  for (int j = tid; j < LS; j += 512) vis_tab[j] = 0.f;
  __syncthreads();
  // ---- cull + compaction (order inside the list is irrelevant: results are scattered by direction index)
  const float nx = normals[3 * p], ny = normals[3 * p + 1], nz = normals[3 * p + 2];
  for (int j0 = 0; j0 < LS; j0 += 512) {
    const int j = j0 + tid;
    bool front = false;
    if (j < LS) {
      const float* d = dirs + 3 * (dbase + j);
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
  if (tid == 0 && eval_count) atomicAdd(eval_count, (unsigned long long)S);
  const int rounds = (S + 127) / 128;

  // ---- weight ring: wave v copies the 1 KB slices v and v + 8 of every 16 KB chunk
  const unsigned ring_b = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)ring);
  const unsigned lane16 = (unsigned)lane * 16u;
  auto dma_chunk_slice = [&](const f4* chunk_weights_uniform, int slot, int d) {
    v4_dma16(chunk_weights_uniform + wave * 64 + d * 512, lane16, ring_b + (unsigned)slot * 16384u + (unsigned)wave * 1024u + (unsigned)d * 8192u);
  };
  if (rounds > 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      dma_chunk_slice(W49 + (long)c * V4_CF4 + 4, c, 0);
      dma_chunk_slice(W49 + (long)c * V4_CF4 + 4, c, 1);
    }
  }
  unsigned sat = 0u;                   // range sentinel: running max of the hi halves (all >= 0 here: ReLU outputs)
  u4 xh[8], xl[8];                     // B operands of the current layer (packed hi / lo halves), one tile
  u4 yh[8], yl[8];                     // ... of the next layer, filled chunk by chunk

  // three products of one k-block into the tile's accumulator (hi*lo, hi*hi, lo*hi: the order of H3Ring::chunk)
  auto mfma_kb = [&](int kb, f4& acc, const u4& wa, const u4& wb) {
    const h8 wh = __builtin_bit_cast(h8, wa), wlo = __builtin_bit_cast(h8, wb);
    const h8 a = __builtin_bit_cast(h8, xh[kb]), b = __builtin_bit_cast(h8, xl[kb]);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, b, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, a, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo, a, acc, 0, 0, 0);
  };
  // relu(z * unscale) of output block jb -> packed operands of the next layer: k-block jb/2, registers 2*(jb&1)+q
  auto epilogue_piece = [&](const f4& acc, int jb, int q) {
    const float v0 = fmaxf(acc[2 * q] * w_unscale, 0.f), v1 = fmaxf(acc[2 * q + 1] * w_unscale, 0.f);
    unsigned hi, lo;
    v4_split_pair(v0, v1, hi, lo);
    yh[jb >> 1][(jb & 1) * 2 + q] = hi;
    yl[jb >> 1][(jb & 1) * 2 + q] = lo;
    sat = sat_acc_nonneg(sat, hi);
  };

  // Layer-0 inputs (rows of the per-direction table) are fetched one round ahead into `raw`: issued while layer 1 of the
  // previous round runs, so their L2/MALL latency is covered by a whole layer of MFMAs.
  f4 raw[16];
  int jj = -1, jjn = -1;
  auto fetch_rows = [&](int rd_next) {
    const int si = rd_next * 128 + wave * 16 + (lane & 15);
    jjn = si < S ? (int)idx_list[si] : -1;
    const int j = jjn < 0 ? 0 : jjn;
    const f4* brow = reinterpret_cast<const f4*>(Bd + (dbase + j) * 256) + g;
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) raw[kb] = brow[kb * 4];
  };
  if (rounds > 0) fetch_rows(0);
  for (int rd = 0; rd < rounds; ++rd) {
    // ---- layer 0: relu(A[p] + Bd[dir]) straight into the operand registers
    jj = jjn;
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) {
      const f4 bv = raw[kb];
      const f4 av = a_row[kb * 4 + g];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        unsigned hi, lo;
        v4_split_pair(fmaxf(av[2 * q] + bv[2 * q], 0.f), fmaxf(av[2 * q + 1] + bv[2 * q + 1], 0.f), hi, lo);
        xh[kb / 2][(kb & 1) * 2 + q] = hi;
        xl[kb / 2][(kb & 1) * 2 + q] = lo;
        sat = sat_acc_nonneg(sat, hi);
      }
    }
#pragma unroll 1
    for (int l = 0; l < 3; ++l) {
      const f4* Wl = W49 + (long)l * 16 * V4_CF4 + 4;                          // this layer's chunk 0 weights
      const f4* Wn = W49 + (long)(l == 2 ? 0 : l + 1) * 16 * V4_CF4 + 4;        // next layer's (next round wraps to 0)
      f4 accs[2];
      // next round's rows (clamped to this round's last samples after the final round: the loads must be issued in any
      // case, the counted waits below assume them)
      if (l == 1) {
        fetch_rows(rd + 1 < rounds ? rd + 1 : rd);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int jb = 0; jb < 16; ++jb) {
        f4& acc = accs[jb & 1];
        acc = bias_tab[(l * 16 + jb) * 4 + g];
        // chunk jb must have landed in its slot: this wave's slices of chunks jb+1 and jb+2 (two each) may still be in flight;
        // the 16 row loads issued at the top of layer 1 are younger than the slices its first three chunks wait for.  Past the
        // barrier every wave has also finished with chunk jb-1, whose slot the copy of chunk jb+3 reuses
        if (jb < 3 && l == 1) {
          asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int nx3 = jb + 3;
        const f4* dsrc = nx3 < 16 ? Wl + (long)nx3 * V4_CF4 : Wn + (long)(nx3 - 16) * V4_CF4;
        const u4* frag = reinterpret_cast<const u4*>(ring) + (jb & 3) * V4_WF4 + lane;
        u4 wa = frag[0], wb = frag[64];
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
          const u4 ca = wa, cb = wb;
          if (kb + 1 < 8) {                                   // next k-block's fragments
            wa = frag[(2 * kb + 2) * 64];
            wb = frag[(2 * kb + 3) * 64];
          }
          mfma_kb(kb, acc, ca, cb);
          if (jb > 0 && kb == 2) epilogue_piece(accs[(jb - 1) & 1], jb - 1, 0);
          if (jb > 0 && kb == 5) epilogue_piece(accs[(jb - 1) & 1], jb - 1, 1);
          if (kb == 1) dma_chunk_slice(dsrc, nx3 & 3, 0);
          if (kb == 4) dma_chunk_slice(dsrc, nx3 & 3, 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      epilogue_piece(accs[1], 15, 0);
      epilogue_piece(accs[1], 15, 1);
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) {
        xh[kb] = yh[kb];
        xl[kb] = yl[kb];
      }
    }
    // ---- head: chunk 48 from its resident LDS copy
    {
      const u4* hw = reinterpret_cast<const u4*>(headw) + lane;
      f4 acc = bias_tab[48 * 4 + g];
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) mfma_kb(kb, acc, hw[(2 * kb) * 64], hw[(2 * kb + 1) * 64]);
      const float l0 = acc[0] * w_unscale, l1 = acc[1] * w_unscale;
      if (g == 0 && jj >= 0) {
        float v;
        if (argmax_vis) {
          v = l1 > l0 ? 1.f : 0.f;
        } else {
          const float mx = fmaxf(l0, l1);
          const float e0 = expf(l0 - mx), e1 = expf(l1 - mx);
          v = e1 / (e0 + e1);
        }
        vis_tab[jj] = v;
      }
    }
  }
  range_report<true>(sat, range_word);
  // drain the ring (copies still target this workgroup's LDS)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid < L) {
    const float* w = wdir + dbase + (long)tid * nsamp;
    float acc = 0.f;
    for (int k = 0; k < nsamp; ++k) acc += vis_tab[tid * nsamp + k] * w[k];
    vis_out[p * L + tid] = acc / wsum[(cid ? cid[p] : 0) * L + tid];
  }
}

}  // namespace rb

using namespace rb;

extern "C" int rb_dvis_fused_v4(const float* normals, const int* chunk_id, long n, const float* A, const float* Bd,
                                const float* dirs, const float* wdir, const float* wsum, const float* W49, int L, int nsamp,
                                int argmax_vis, int scale_log2, float* vis_out, unsigned long long* eval_count,
                                rb_stream_t stream) {
  if (n <= 0) return 0;
  RB_REQUIRE(normals && A && Bd && dirs && wdir && wsum && W49 && vis_out, "null pointer");
  RB_REQUIRE(n <= RB_MAX_BLOCKS, "too many points for one launch (one workgroup each)");
  RB_REQUIRE(L > 0 && L <= 256 && nsamp > 0 && (long)L * nsamp <= V4_MAX_DIRS, "need L <= 256 and L*nsamp <= 4096");
  hipLaunchKernelGGL(k_dvis_v4, dim3((unsigned)n), dim3(512), 0, (hipStream_t)stream, normals, chunk_id, n, A, Bd, dirs, wdir, wsum,
                     (const f4*)W49, L, nsamp, argmax_vis, ldexpf(1.0f, -scale_log2), vis_out, eval_count,
                     range_flags() ? range_flags() + RB_RANGE_DVIS : nullptr);
  return check_launch("k_dvis_v4");
}
