// Light-SG visibility, second-generation kernel (get_diffuse_visibility, model/sg_render.py:111-195).
//
// Same arithmetic as k_dvis_fused<H3> (vis_diffuse.hip): split-precision f16x3 hidden stack with fp32 accumulation,
// first layer factored into a per-point row and a per-direction table.  What changed is the shape of the workgroup:
// the ablation timings of the first kernel (tools/prof_dvis.py variants) showed that the weight stream through LDS
// -- 16 KB staged per 16 output neurons for every 64 samples, and one 1 KB A-fragment read per MFMA -- costs more
// than the MFMAs themselves.  Here a wave owns TWO 16-sample tiles (every A fragment feeds two tiles: half the LDS
// reads and half the staging per MFMA), one workgroup of four waves owns the CU (512 registers per lane), and
// everything that is not an MFMA is slotted between the MFMAs of the same wave instead of relying on a second
// workgroup to fill the gaps:
//   * weights: 4-slot LDS ring filled by LDS-DMA three chunks ahead, issued from inline asm (the compiler's waitcnt
//     pass would otherwise drain the ring in front of every ds_read), counted vmcnt + one s_barrier per chunk;
//   * the relu + hi/lo split of a chunk's result runs under the next chunk's MFMAs and writes straight into the next
//     layer's operand registers (no fp32 copy of the layer output);
//   * the 256->2 head is a 49th, LDS-resident chunk on the matrix pipe.
#include "../../include/robir_hip.h"
#include "common.h"
#include "mlp_engine.h"
#include <cstdlib>

namespace rb {

#define RB_TINY 1e-6f
constexpr int V2_MAX_DIRS = 4096;
constexpr int V2_CF4 = chunk_f4(256);   // float4s per packed chunk in global memory (bias + weights)
constexpr int V2_WF4 = 1024;            // weight part of a chunk (16 KB)
constexpr int V2_SLOTS = 4, V2_DIST = 3;

// global -> LDS copy of 16 B per lane: wave-uniform LDS base in M0, uniform global base + per-lane byte offset
__device__ __forceinline__ void v2_dma16(const f4* gbase_uniform, unsigned lane_byte_off, unsigned lds_byte_uniform) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_byte_uniform), "v"(lane_byte_off),
               "s"(gbase_uniform)
               : "memory");
}

__device__ unsigned long long rb_v2_dbg[8];
#define V2_T(i)                                                   \
  if constexpr (TIMED) {                                          \
    const long long now_ = clock64();                             \
    tacc[i] += now_ - tlast;                                      \
    tlast = now_;                                                 \
  }

// hi/lo split of two fp32 values in 3 VALU ops: packed round-toward-zero hi halves, then lo = f16(v - float(hi)) with the
// mixed-precision fma (f16 source read straight from the packed register, result written to one half of `lo`)
__device__ __forceinline__ void v2_split_pair(float v0, float v1, unsigned& hi, unsigned& lo) {
  const h2 h = __builtin_amdgcn_cvt_pkrtz(v0, v1);
  const unsigned hu = __builtin_bit_cast(unsigned, h);
  unsigned l;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(hu), "v"(v0));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(hu), "v"(v1));
  hi = hu;
  lo = l;
}

constexpr int V2_CH = 1;   // accumulator chains per tile: 1 = all three products into one accumulator
struct V2Acc {
  f4 a[2][V2_CH];
};

template <bool TIMED>
__global__ __launch_bounds__(256, 1) void k_dvis_v2(
    const float* __restrict__ normals, const int* __restrict__ cid, long n, const float* __restrict__ A,
    const float* __restrict__ Bd, const float* __restrict__ dirs, const float* __restrict__ wdir,
    const float* __restrict__ wsum, const f4* __restrict__ W49, int L, int nsamp, int argmax_vis, float w_unscale,
    float* __restrict__ vis_out, unsigned long long* __restrict__ eval_count, unsigned* __restrict__ range_word) {
  __shared__ f4 ring[V2_SLOTS * V2_WF4];   // 64 KB
  __shared__ f4 headw[V2_WF4];             // 16 KB: chunk 48 (256 -> 2 head, rows 2..15 zero)
  __shared__ f4 bias_tab[49 * 4];
  __shared__ float vis_tab[V2_MAX_DIRS];
  __shared__ unsigned short idx_list[V2_MAX_DIRS];
  __shared__ f4 a_row[64];
  __shared__ int s_count;
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tlast = TIMED ? clock64() : 0;
  const long p = blockIdx.x;
  const int LS = L * nsamp;
  const long dbase = (cid ? (long)cid[p] : 0L) * LS;
  if (tid == 0) s_count = 0;
  if (tid < 64) a_row[tid] = reinterpret_cast<const f4*>(A + p * 256)[tid];
  for (int i = tid; i < 49 * 4; i += 256) bias_tab[i] = W49[(long)(i >> 2) * V2_CF4 + (i & 3)];
  for (int i = tid; i < V2_WF4; i += 256) headw[i] = W49[48L * V2_CF4 + 4 + i];
  for (int j = tid; j < LS; j += 256) vis_tab[j] = 0.f;
  __syncthreads();
  // ---- cull + compaction (order inside the list is irrelevant: results are scattered by direction index)
  const float nx = normals[3 * p], ny = normals[3 * p + 1], nz = normals[3 * p + 2];
  for (int j0 = 0; j0 < LS; j0 += 256) {
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
  V2_T(0)

  // ---- weight ring state
  const unsigned ring_b = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)ring);
  const unsigned lane_off = (unsigned)tid * 16u;                  // byte offset of this lane inside a 4 KB DMA row
  const unsigned wave_lds = ring_b + (unsigned)wave * 1024u;      // + slot * 16384 + i * 4096
  auto dma_chunk = [&](const f4* chunk_weights_uniform, int slot) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      v2_dma16(chunk_weights_uniform + i * 256, lane_off, wave_lds + (unsigned)slot * 16384u + (unsigned)i * 4096u);
  };
  const u4* ring_u = reinterpret_cast<const u4*>(ring) + lane;
  u4 wreg[16];
  f4 bias;
  if (rounds > 0) {
    dma_chunk(W49 + 0L * V2_CF4 + 4, 0);
    dma_chunk(W49 + 1L * V2_CF4 + 4, 1);
    dma_chunk(W49 + 2L * V2_CF4 + 4, 2);   // stays in flight: the first chunk waits for chunk 1 only
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < 16; ++i) wreg[i] = ring_u[i * 64];
    bias = bias_tab[g];
  }

  V2_T(1)
  unsigned sat = 0u;                   // range sentinel: running max of the hi halves (all >= 0 here: ReLU outputs)
  // 128-bit tuples (one MFMA B operand each), declared as vectors so that a k-block's four registers stay contiguous
  u4 xh[2][8], xl[2][8];   // B operands of the current layer (packed hi / lo halves)
  u4 yh[2][8], yl[2][8];   // ... of the next layer, filled chunk by chunk

  auto mfma_kb = [&](int kb, V2Acc& acc, const u4 (&wsrc)[16]) {
    const h8 wh = __builtin_bit_cast(h8, wsrc[kb * 2]);
    const h8 wlo = __builtin_bit_cast(h8, wsrc[kb * 2 + 1]);
    h8 a[2], b[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      a[t] = __builtin_bit_cast(h8, xh[t][kb]);
      b[t] = __builtin_bit_cast(h8, xl[t][kb]);
    }
    // same product order as H3Ring::chunk (hi*lo, hi*hi, lo*hi; corrections share an accumulator).  The three MFMAs of
    // a tile are issued back to back: consecutive MFMAs on ONE accumulator hide up to two filler instructions each for
    // free, MFMAs that alternate between two accumulators do not (tools/ubench/mfma_fill.hip: 0.74 vs 0.45 of the f16
    // peak at two fillers per MFMA)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      acc.a[t][V2_CH - 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, b[t], acc.a[t][V2_CH - 1], 0, 0, 0);
      acc.a[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, a[t], acc.a[t][0], 0, 0, 0);
      acc.a[t][V2_CH - 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo, a[t], acc.a[t][V2_CH - 1], 0, 0, 0);
      // (no instruction: orders the two tiles' MFMA groups, which the scheduler would otherwise interleave)
      if (t == 0) asm volatile("" : "+a"(acc.a[0][0]), "+a"(acc.a[1][0]));
    }
  };
  // relu(z * unscale) of output block jb -> packed operands of the next layer: k-block jb/2, registers 2*(jb&1)+{0,1}.
  // piece = (tile, register pair): four pieces per chunk, spread over the next chunk's k-blocks
  auto epilogue_piece = [&](const V2Acc& acc, int jb, int piece) {
    const int t = piece >> 1, q = piece & 1;
    float r0 = acc.a[t][0][2 * q], r1 = acc.a[t][0][2 * q + 1];
    if constexpr (V2_CH == 2) {
      r0 += acc.a[t][1][2 * q];
      r1 += acc.a[t][1][2 * q + 1];
    }
    const float v0 = fmaxf(r0 * w_unscale, 0.f), v1 = fmaxf(r1 * w_unscale, 0.f);
    unsigned hi, lo;
    v2_split_pair(v0, v1, hi, lo);
    yh[t][jb >> 1][(jb & 1) * 2 + q] = hi;
    yl[t][jb >> 1][(jb & 1) * 2 + q] = lo;
    sat = sat_acc_nonneg(sat, hi);
  };
  auto epilogue = [&](const V2Acc& acc, int jb) {
#pragma unroll
    for (int pc = 0; pc < 4; ++pc) epilogue_piece(acc, jb, pc);
  };

  // Layer-0 inputs (rows of the per-direction table) are fetched one round ahead into `raw`: issued while layer 1 of the
  // previous round runs, so their L2/MALL latency is covered by a whole layer of MFMAs.
  f4 raw[2][16];
  int jj[2], jjn[2];
  auto fetch_rows = [&](int rd_next) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int si = rd_next * 128 + wave * 32 + t * 16 + (lane & 15);
      jjn[t] = si < S ? (int)idx_list[si] : -1;
      const int j = jjn[t] < 0 ? 0 : jjn[t];
      const f4* brow = reinterpret_cast<const f4*>(Bd + (dbase + j) * 256) + g;
#pragma unroll
      for (int kb = 0; kb < 16; ++kb) raw[t][kb] = brow[kb * 4];
    }
  };
  if (rounds > 0) fetch_rows(0);
  for (int rd = 0; rd < rounds; ++rd) {
    // ---- layer 0: relu(A[p] + Bd[dir]) straight into the operand registers
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      jj[t] = jjn[t];
#pragma unroll
      for (int kb = 0; kb < 16; ++kb) {
        const f4 bv = raw[t][kb];
        const f4 av = a_row[kb * 4 + g];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          unsigned hi, lo;
          v2_split_pair(fmaxf(av[2 * q] + bv[2 * q], 0.f), fmaxf(av[2 * q + 1] + bv[2 * q + 1], 0.f), hi, lo);
          xh[t][kb / 2][(kb & 1) * 2 + q] = hi;
          xl[t][kb / 2][(kb & 1) * 2 + q] = lo;
          sat = sat_acc_nonneg(sat, hi);
        }
      }
    }
    V2_T(2)
#pragma unroll 1
    for (int l = 0; l < 3; ++l) {
      const f4* Wl = W49 + (long)l * 16 * V2_CF4 + 4;                          // this layer's chunk 0 weights
      const f4* Wn = W49 + (long)(l == 2 ? 0 : l + 1) * 16 * V2_CF4 + 4;        // next layer's (next round wraps to 0)
      V2Acc prev;
      // next round's rows (clamped to this round's last samples after the final round: the loads must be issued in any
      // case, the counted waits below assume them)
      if (l == 1) {
        fetch_rows(rd + 1 < rounds ? rd + 1 : rd);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int jb = 0; jb < 16; ++jb) {
        V2Acc acc;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          acc.a[t][0] = bias;
          if constexpr (V2_CH == 2) acc.a[t][1] = f4{0.f, 0.f, 0.f, 0.f};
        }
        // chunk jb+1 has landed in its slot once at most the copy of chunk jb+2 (4 instructions) is still in flight;
        // past the barrier every wave has also finished with chunk jb-1, whose slot the copy of chunk jb+3 reuses
        // (the 32 row loads issued at the top of layer 1 are younger than the copies the first two chunks wait for)
        if (jb < 2 && l == 1) {
          asm volatile("s_waitcnt vmcnt(36)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int nx3 = jb + V2_DIST;
        const f4* dsrc = nx3 < 16 ? Wl + (long)nx3 * V2_CF4 : Wn + (long)(nx3 - 16) * V2_CF4;
        const unsigned ddst = wave_lds + (unsigned)(nx3 & 3) * 16384u;
        // rolling fragment registers: the pair of k-block kb is refilled with chunk jb+1's as soon as this chunk's
        // MFMAs of that k-block have issued; the previous chunk's relu/split and the four 1 KB pieces of the copy of
        // chunk jb+3 go into the gaps (one piece per two k-blocks: back-to-back pieces stall the issue port)
        const int ns = (jb + 1) & 3;
        const f4 nbias = bias_tab[(l * 16 + jb + 1) * 4 + g];     // index 48 = head chunk after the last layer
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
          mfma_kb(kb, acc, wreg);
          wreg[2 * kb] = ring_u[ns * V2_WF4 + (2 * kb) * 64];
          wreg[2 * kb + 1] = ring_u[ns * V2_WF4 + (2 * kb + 1) * 64];
          if (jb > 0 && (kb & 1)) epilogue_piece(prev, jb - 1, kb >> 1);
          if (!(kb & 1)) v2_dma16(dsrc + (kb >> 1) * 256, lane_off, ddst + (unsigned)(kb >> 1) * 4096u);
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
    V2_T(3)
    // ---- head: chunk 48 from its resident LDS copy; `bias` holds its bias (fetched by the last chunk of layer 2) and
    // wreg already holds the fragments of the next round's chunk 0
    {
      const u4* hw = reinterpret_cast<const u4*>(headw) + lane;
      u4 hreg[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) hreg[i] = hw[i * 64];
      V2Acc acc;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        acc.a[t][0] = bias;
        if constexpr (V2_CH == 2) acc.a[t][1] = f4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) mfma_kb(kb, acc, hreg);
      bias = bias_tab[g];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f4 r = acc.a[t][0];
        if constexpr (V2_CH == 2) r = r + acc.a[t][1];
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
          vis_tab[jj[t]] = v;
        }
      }
    }
    V2_T(4)
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
  V2_T(5)
  if constexpr (TIMED) {
    if (tid == 0)
      for (int i = 0; i < 6; ++i) atomicAdd(&rb_v2_dbg[i], (unsigned long long)tacc[i]);
  }
}

}  // namespace rb

using namespace rb;

extern "C" int rb_dvis_fused_v2(const float* normals, const int* chunk_id, long n, const float* A, const float* Bd,
                                const float* dirs, const float* wdir, const float* wsum, const float* W49, int L, int nsamp,
                                int argmax_vis, int scale_log2, float* vis_out, unsigned long long* eval_count,
                                rb_stream_t stream) {
  if (n <= 0) return 0;
  RB_REQUIRE(normals && A && Bd && dirs && wdir && wsum && W49 && vis_out, "null pointer");
  RB_REQUIRE(n <= RB_MAX_BLOCKS, "too many points for one launch (one workgroup each)");
  RB_REQUIRE(L > 0 && L <= 256 && nsamp > 0 && (long)L * nsamp <= V2_MAX_DIRS, "need L <= 256 and L*nsamp <= 4096");
  // RB_V2_TIMED=1: per-phase shader-clock totals (rb_dvis_v2_debug), a profiling aid -- costs ~10 % in the kernel
  static const char* const tm = getenv("RB_V2_TIMED");   // read once
#define RB_V2(T)                                                                                                          \
  hipLaunchKernelGGL(k_dvis_v2<T>, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, normals, chunk_id, n, A, Bd, dirs, \
                     wdir, wsum, (const f4*)W49, L, nsamp, argmax_vis, ldexpf(1.0f, -scale_log2), vis_out, eval_count,       \
                     range_flags() ? range_flags() + RB_RANGE_DVIS : nullptr)
  if (tm && tm[0] == '1') {
    RB_V2(true);
  } else {
    RB_V2(false);
  }
#undef RB_V2
  return check_launch("k_dvis_v2");
}

// debug: per-phase shader-clock totals of wave 0 (RB_V2_TIMED=1): prologue, ring start, gather, layers, head, final
extern "C" int rb_dvis_v2_debug(unsigned long long* out8) {
  RB_REQUIRE(out8, "null pointer");
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(rb_v2_dbg), sizeof(z)) != hipSuccess) return 1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(rb_v2_dbg), z, sizeof(z)) != hipSuccess) return 1;
  return 0;
}
