// Light-SG visibility with EXACT fp32 operands on the f16 matrix pipe ("f16x6", get_diffuse_visibility,
// model/sg_render.py:111-195; VisNetwork, model/implicit_differentiable_renderer.py:241-258).
//
// k_dvis_v2 (vis_diffuse_v2.hip) carries every fp32 operand as an (hi, lo) half pair = 22 significant bits and keeps three
// of the four partial products: narrower than the reference's fp32.  This kernel carries every operand as THREE halves
//     v = h + m * 2^-11 + l * 2^-22        h = rtz16(v), m = rtz16((v - h) 2^11), l = f16(((v - h) 2^11 - m) 2^11)
// which is exact for every fp32 v inside the f16 exponent range (11 + 11 + 2 bits of a 24-bit significand; the residuals are
// formed exactly by the mixed-precision fma), and keeps the six partial products of weight >= 2^-22:
//     class 0 (2^0)   wh.xh
//     class 1 (2^-11) wh.xm + wm.xh
//     class 2 (2^-22) wm.xm + wh.xl + wl.xh
// (dropped: wm.xl, wl.xm at 2^-33 and wl.xl at 2^-44 of the product -- below an fp32 product's own last bit by 2^-9).
// Every product is exact in the fp32 accumulator; each class has its own accumulator (the correction terms are summed
// among themselves first, so their rounding errors are scaled down by 2^-11 / 2^-22 when the classes are combined):
// the result is not narrower than an fp32 fma chain.  Cost: 6 f16 MFMAs per fp32 multiply-add (bound 2500 / 6 TFLOP/s),
// against v_mfma_f32_16x16x4_f32's 157 TFLOP/s.
//
// Shape: k_dvis_v2's machine with ONE 16-sample tile per wave (three operand pieces of the current and of the next layer
// are 192 registers per tile): one workgroup of four waves per CU, 4-slot LDS ring of 24 KB chunks (16 neurons x K = 256 x
// 3 pieces) filled by LDS-DMA three chunks ahead, rolling fragment registers, the previous chunk's relu + three-way split
// between the MFMAs of the current one, the 256 -> 2 head as a 49th LDS-resident chunk.  A chunk's 48 MFMAs are issued as
// six runs of eight on one accumulator each (wl.xh, wm.xm, wh.xl | wm.xh, wh.xm | wh.xh): a weight piece's registers are
// refilled with the next chunk's as soon as its last run has issued.
#include "../../include/robir_hip.h"
#include "common.h"
#include "mlp_engine.h"

namespace rb {

#define RB_TINY 1e-6f
constexpr int X6_MAX_DIRS = 4096;
constexpr int X6_WF4 = 1536;             // weight part of a packed chunk: [kb 8][piece 3][lane 64] x 16 B = 24 KB
constexpr int X6_CF4 = 4 + X6_WF4;       // packed chunk in global memory: 16 bias floats + weights
constexpr int X6_SLOTS = 4, X6_DIST = 3;
constexpr int X6_PIECES = 6;             // 4 KB rows (1 KB per wave) of one chunk copy
#ifndef X6_VOFF
#define X6_VOFF 1                        // 1: the six rows of a chunk copy addressed by per-row VGPR offsets (no scalar adds)
#endif
#ifndef X6_UNSCALE
#define X6_UNSCALE 0                     // 0: weights packed with scale 2^0, no un-scaling multiply in the epilogue
// (A/B on an MI355X, 32 chunks: both off 121 ms; per-row offsets alone 120.4; no un-scaling alone 120.1; the VGPR form of the MFMAs
// alone (-amdgpu-mfma-vgpr-form, Makefile) 119.3; all three together 114.7 -- the three remove 30 of a chunk's 127 non-MFMA
// instructions.  Unlike the (hi, lo) pairs of the f16x3 kernels the m and l pieces are re-scaled by 2^11 / 2^22, so the weights need
// no power-of-two lift to keep them out of the f16 subnormal range: scale_log2 = 0.)
#endif

__device__ __forceinline__ void x6_dma16(const f4* gbase_uniform, unsigned lane_byte_off, unsigned lds_byte_uniform) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_byte_uniform), "v"(lane_byte_off),
               "s"(gbase_uniform)
               : "memory");
}

// Exact three-way split of two fp32 values (see the header): 11 VALU ops per pair.  k = -2048.0f in a scalar register.
__device__ __forceinline__ void x6_split_pair(float v0, float v1, float negk, unsigned& h, unsigned& m, unsigned& l) {
  const unsigned hu = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v0, v1));
  const float s0 = v0 * 2048.0f, s1 = v1 * 2048.0f;
  float d0, d1;   // (v - h) * 2^11, exact: the residual of an 11-bit truncation of a 24-bit significand has <= 13 bits
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d0) : "v"(hu), "s"(negk), "v"(s0));
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d1) : "v"(hu), "s"(negk), "v"(s1));
  const unsigned mu = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(d0, d1));
  const float e0 = d0 * 2048.0f, e1 = d1 * 2048.0f;
  unsigned lu;    // ((v - h) 2^11 - m) 2^11: <= 2 significant bits, exact in f16
  asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(lu) : "v"(mu), "s"(negk), "v"(e0));
  asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lu) : "v"(mu), "s"(negk), "v"(e1));
  h = hu;
  m = mu;
  l = lu;
}

struct X6Acc {
  f4 c0, c1, c2;   // classes 2^0, 2^-11, 2^-22
};

__global__ __launch_bounds__(256, 1) void k_dvis_x6(
    const float* __restrict__ normals, const int* __restrict__ cid, long n, const float* __restrict__ A,
    const float* __restrict__ Bd, const float* __restrict__ dirs, const float* __restrict__ wdir,
    const float* __restrict__ wsum, const f4* __restrict__ W49, int L, int nsamp, int argmax_vis, float w_unscale,
    float* __restrict__ vis_out, unsigned long long* __restrict__ eval_count, unsigned* __restrict__ range_word) {
  __shared__ f4 ring[X6_SLOTS * X6_WF4];   // 96 KB
  __shared__ f4 headw[X6_WF4];             // 24 KB: chunk 48 (256 -> 2 head, rows 2..15 zero)
  __shared__ f4 bias_tab[49 * 4];
  __shared__ float vis_tab[X6_MAX_DIRS];
  __shared__ unsigned short idx_list[X6_MAX_DIRS];
  __shared__ f4 a_row[64];
  __shared__ int s_count;
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long p = blockIdx.x;
  const int LS = L * nsamp;
  const long dbase = (cid ? (long)cid[p] : 0L) * LS;
  const float negk = -2048.0f;
  constexpr float C11 = 1.0f / 2048.0f;
  if (tid == 0) s_count = 0;
  if (tid < 64) a_row[tid] = reinterpret_cast<const f4*>(A + p * 256)[tid];
  for (int i = tid; i < 49 * 4; i += 256) bias_tab[i] = W49[(long)(i >> 2) * X6_CF4 + (i & 3)];
  for (int i = tid; i < X6_WF4; i += 256) headw[i] = W49[48L * X6_CF4 + 4 + i];
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
    const unsigned long long mk = __ballot(front);
    int base = 0;
    if (lane == 0 && mk) base = atomicAdd(&s_count, __popcll(mk));
    base = __shfl(base, 0);
    if (front) idx_list[base + __popcll(mk & ((1ull << lane) - 1ull))] = (unsigned short)j;
  }
  __syncthreads();
  const int S = s_count;
  if (tid == 0 && eval_count) atomicAdd(eval_count, (unsigned long long)S);
  const int rounds = (S + 63) / 64;

  // ---- weight ring state
  const unsigned ring_b = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)ring);
  const unsigned lane_off = (unsigned)tid * 16u;                  // byte offset of this lane inside a 4 KB DMA row
  const unsigned wave_lds = ring_b + (unsigned)wave * 1024u;      // + slot * 24576 + i * 4096
  auto dma_chunk = [&](const f4* chunk_weights_uniform, int slot) {
#pragma unroll
    for (int i = 0; i < X6_PIECES; ++i)
      x6_dma16(chunk_weights_uniform + i * 256, lane_off, wave_lds + (unsigned)slot * 24576u + (unsigned)i * 4096u);
  };
#if X6_VOFF
  unsigned row_off[X6_PIECES];
#pragma unroll
  for (int i = 0; i < X6_PIECES; ++i) {
    row_off[i] = lane_off + (unsigned)i * 4096u;
    asm volatile("" : "+v"(row_off[i]));       // keep six registers: re-deriving them per copy is what this variant avoids
  }
#endif
  const u4* ring_u = reinterpret_cast<const u4*>(ring) + lane;
  // fragment (kb, piece) of a slot: ring_u[slot * X6_WF4 + (kb * 3 + piece) * 64]
  u4 wh[8], wm[8], wl[8];
  f4 bias;
  if (rounds > 0) {
    dma_chunk(W49 + 0L * X6_CF4 + 4, 0);
    dma_chunk(W49 + 1L * X6_CF4 + 4, 1);
    dma_chunk(W49 + 2L * X6_CF4 + 4, 2);   // stays in flight: the first chunk waits for chunk 1 only
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
      wh[kb] = ring_u[(kb * 3 + 0) * 64];
      wm[kb] = ring_u[(kb * 3 + 1) * 64];
      wl[kb] = ring_u[(kb * 3 + 2) * 64];
    }
    bias = bias_tab[g];
  }

  unsigned sat = 0u;                   // range sentinel: running max of the h pieces (all >= 0 here: ReLU outputs)
  u4 xh[8], xm[8], xl[8];              // B operands of the current layer (one 128-bit tuple per k-block and piece)
  u4 yh[8], ym[8], yl[8];              // ... of the next layer, filled chunk by chunk

#define X6_MFMA(ACC, WREG, XREG) \
  ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, WREG), __builtin_bit_cast(h8, XREG), ACC, 0, 0, 0)

  // relu(z * unscale) of output block jb -> operands of the next layer: k-block jb/2, registers 2*(jb&1)+{0,1}; q = register
  // pair.  Three stages per pair so that the vector work spreads over the MFMA groups of the next chunk.
  float ev0[2], ev1[2], ed0[2], ed1[2];
  unsigned eh[2];
  auto ep_stage1 = [&](const X6Acc& acc, int q) {
    const float r0 = __builtin_fmaf(__builtin_fmaf(acc.c2[2 * q], C11, acc.c1[2 * q]), C11, acc.c0[2 * q]);
    const float r1 = __builtin_fmaf(__builtin_fmaf(acc.c2[2 * q + 1], C11, acc.c1[2 * q + 1]), C11, acc.c0[2 * q + 1]);
#if X6_UNSCALE
    ev0[q] = fmaxf(r0 * w_unscale, 0.f);
    ev1[q] = fmaxf(r1 * w_unscale, 0.f);
#else
    ev0[q] = fmaxf(r0, 0.f);
    ev1[q] = fmaxf(r1, 0.f);
#endif
  };
  auto ep_stage2 = [&](int q) {
    const unsigned hu = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(ev0[q], ev1[q]));
    const float s0 = ev0[q] * 2048.0f, s1 = ev1[q] * 2048.0f;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(ed0[q]) : "v"(hu), "s"(negk), "v"(s0));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(ed1[q]) : "v"(hu), "s"(negk), "v"(s1));
    eh[q] = hu;
    sat = sat_acc_nonneg(sat, hu);
  };
  auto ep_stage3 = [&](int jb, int q) {
    const unsigned mu = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(ed0[q], ed1[q]));
    const float e0 = ed0[q] * 2048.0f, e1 = ed1[q] * 2048.0f;
    unsigned lu;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(lu) : "v"(mu), "s"(negk), "v"(e0));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lu) : "v"(mu), "s"(negk), "v"(e1));
    yh[jb >> 1][(jb & 1) * 2 + q] = eh[q];
    ym[jb >> 1][(jb & 1) * 2 + q] = mu;
    yl[jb >> 1][(jb & 1) * 2 + q] = lu;
  };

  // Layer-0 inputs (rows of the per-direction table) are fetched one round ahead into `raw`
  f4 raw[16];
  int jj, jjn;
  auto fetch_rows = [&](int rd_next) {
    const int si = rd_next * 64 + wave * 16 + (lane & 15);
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
        unsigned h, m, l;
        x6_split_pair(fmaxf(av[2 * q] + bv[2 * q], 0.f), fmaxf(av[2 * q + 1] + bv[2 * q + 1], 0.f), negk, h, m, l);
        xh[kb / 2][(kb & 1) * 2 + q] = h;
        xm[kb / 2][(kb & 1) * 2 + q] = m;
        xl[kb / 2][(kb & 1) * 2 + q] = l;
        sat = sat_acc_nonneg(sat, h);
      }
    }
#pragma unroll 1
    for (int l = 0; l < 3; ++l) {
      const f4* Wl = W49 + (long)l * 16 * X6_CF4 + 4;                          // this layer's chunk 0 weights
      const f4* Wn = W49 + (long)(l == 2 ? 0 : l + 1) * 16 * X6_CF4 + 4;        // next layer's (next round wraps to 0)
      X6Acc prev;
      // next round's rows (clamped to this round's samples after the final round: the loads must be issued in any case,
      // the counted waits below assume them)
      if (l == 1) {
        fetch_rows(rd + 1 < rounds ? rd + 1 : rd);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int jb = 0; jb < 16; ++jb) {
        X6Acc acc;
        acc.c0 = bias;
        acc.c1 = f4{0.f, 0.f, 0.f, 0.f};
        acc.c2 = f4{0.f, 0.f, 0.f, 0.f};
        // chunk jb+1 has landed in its slot once at most the copy of chunk jb+2 (6 instructions) is still in flight; past
        // the barrier every wave has also finished with chunk jb-1, whose slot the copy of chunk jb+3 reuses (the 16 row
        // loads issued at the top of layer 1 are younger than the copies the first two chunks wait for)
        if (jb < 2 && l == 1) {
          asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        }
#ifndef X6_ABL_NOBAR                  // timing ablation (wrong results): no per-chunk barrier
        __builtin_amdgcn_s_barrier();
#endif
        asm volatile("" ::: "memory");
        const int nx3 = jb + X6_DIST;
        const f4* dsrc = nx3 < 16 ? Wl + (long)nx3 * X6_CF4 : Wn + (long)(nx3 - 16) * X6_CF4;
        const unsigned ddst = wave_lds + (unsigned)(nx3 & 3) * 24576u;
        const u4* nring = ring_u + ((jb + 1) & 3) * X6_WF4;
        const f4 nbias = bias_tab[(l * 16 + jb + 1) * 4 + g];     // index 48 = head chunk after the last layer
        // Twelve groups of four MFMAs (six runs of eight on one accumulator each); the fillers of a group are named with it:
        // copy rows of chunk jb+3, fragment reads of chunk jb+1 (a piece's registers are free once its last run has issued),
        // the three stages of the previous chunk's two register pairs.
#define X6_RUN(ACC, WP, XP, K0) \
  _Pragma("unroll") for (int kb = (K0); kb < (K0) + 4; ++kb) X6_MFMA(ACC, WP[kb], XP[kb])
#if defined(X6_ABL_NODMA)            // timing ablation (wrong results): no weight copies
#define X6_COPY(I) (void)0
#elif X6_VOFF
#define X6_COPY(I) x6_dma16(dsrc, row_off[I], ddst + (unsigned)(I) * 4096u)
#else
#define X6_COPY(I) x6_dma16(dsrc + (I) * 256, lane_off, ddst + (unsigned)(I) * 4096u)
#endif
#define X6_FENCE __builtin_amdgcn_sched_barrier(0)
#ifdef X6_ABL_NOLDS                   // timing ablation (wrong results): no fragment reads
#define X6_FRAG(DST, KB, PIECE) asm volatile("" : "+v"(DST))
#else
#define X6_FRAG(DST, KB, PIECE) DST = nring[((KB) * 3 + (PIECE)) * 64]
#endif
        X6_RUN(acc.c2, wl, xh, 0);      // run A: wl.xh -> class 2
        X6_COPY(0);
        X6_FENCE;
        X6_RUN(acc.c2, wl, xh, 4);
        X6_COPY(1);
        if (jb > 0) ep_stage1(prev, 0);
        X6_FENCE;
        X6_RUN(acc.c2, wm, xm, 0);      // run B: wm.xm -> class 2
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) X6_FRAG(wl[kb], kb, 2);
        X6_FENCE;
        X6_RUN(acc.c2, wm, xm, 4);
        if (jb > 0) ep_stage2(0);
        X6_FRAG(wl[4], 4, 2);
        X6_FRAG(wl[5], 5, 2);
        X6_FENCE;
        X6_RUN(acc.c2, wh, xl, 0);      // run C: wh.xl -> class 2
        X6_COPY(2);
        if (jb > 0) ep_stage3(jb - 1, 0);
        X6_FRAG(wl[6], 6, 2);
        X6_FRAG(wl[7], 7, 2);
        X6_FENCE;
        X6_RUN(acc.c2, wh, xl, 4);
        X6_COPY(3);
        if (jb > 0) ep_stage1(prev, 1);
        X6_FENCE;
        X6_RUN(acc.c1, wm, xh, 0);      // run D: wm.xh -> class 1
        X6_COPY(4);
        if (jb > 0) ep_stage2(1);
        X6_FENCE;
        X6_RUN(acc.c1, wm, xh, 4);
        X6_COPY(5);
        if (jb > 0) ep_stage3(jb - 1, 1);
        X6_FENCE;
        X6_RUN(acc.c1, wh, xm, 0);      // run E: wh.xm -> class 1
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) X6_FRAG(wm[kb], kb, 1);
        X6_FENCE;
        X6_RUN(acc.c1, wh, xm, 4);
#pragma unroll
        for (int kb = 4; kb < 8; ++kb) X6_FRAG(wm[kb], kb, 1);
        X6_FENCE;
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {   // run F: wh.xh -> class 0, each fragment refilled right behind its MFMA
          X6_MFMA(acc.c0, wh[kb], xh[kb]);
          X6_FRAG(wh[kb], kb, 0);
          if (kb == 3) X6_FENCE;
        }
        X6_FENCE;
#undef X6_RUN
#undef X6_COPY
#undef X6_FENCE
#undef X6_FRAG
        prev = acc;
        bias = nbias;
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        ep_stage1(prev, q);
        ep_stage2(q);
        ep_stage3(15, q);
      }
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) {
        xh[kb] = yh[kb];
        xm[kb] = ym[kb];
        xl[kb] = yl[kb];
      }
    }
    // ---- head: chunk 48 from its resident LDS copy; `bias` holds its bias (fetched by the last chunk of layer 2) and the
    // fragment registers already hold the next round's chunk 0
    {
      const u4* hw = reinterpret_cast<const u4*>(headw) + lane;
      X6Acc acc;
      acc.c0 = bias;
      acc.c1 = f4{0.f, 0.f, 0.f, 0.f};
      acc.c2 = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) {
        const u4 fh = hw[(kb * 3 + 0) * 64], fm = hw[(kb * 3 + 1) * 64], fl = hw[(kb * 3 + 2) * 64];
        X6_MFMA(acc.c2, fl, xh[kb]);
        X6_MFMA(acc.c2, fm, xm[kb]);
        X6_MFMA(acc.c2, fh, xl[kb]);
        X6_MFMA(acc.c1, fm, xh[kb]);
        X6_MFMA(acc.c1, fh, xm[kb]);
        X6_MFMA(acc.c0, fh, xh[kb]);
      }
      bias = bias_tab[g];
      const float l0 = __builtin_fmaf(__builtin_fmaf(acc.c2[0], C11, acc.c1[0]), C11, acc.c0[0]) * w_unscale;
      const float l1 = __builtin_fmaf(__builtin_fmaf(acc.c2[1], C11, acc.c1[1]), C11, acc.c0[1]) * w_unscale;
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
#undef X6_MFMA
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

extern "C" int rb_dvis_fused_x6(const float* normals, const int* chunk_id, long n, const float* A, const float* Bd,
                                const float* dirs, const float* wdir, const float* wsum, const float* W49, int L, int nsamp,
                                int argmax_vis, int scale_log2, float* vis_out, unsigned long long* eval_count,
                                rb_stream_t stream) {
  if (n <= 0) return 0;
  RB_REQUIRE(normals && A && Bd && dirs && wdir && wsum && W49 && vis_out, "null pointer");
  RB_REQUIRE(n <= RB_MAX_BLOCKS, "too many points for one launch (one workgroup each)");
  RB_REQUIRE(L > 0 && L <= 256 && nsamp > 0 && (long)L * nsamp <= X6_MAX_DIRS, "need L <= 256 and L*nsamp <= 4096");
#if !X6_UNSCALE
  RB_REQUIRE(scale_log2 == 0, "this build of k_dvis_x6 takes weights packed with scale_log2 = 0");
#endif
  hipLaunchKernelGGL(k_dvis_x6, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, normals, chunk_id, n, A, Bd, dirs, wdir,
                     wsum, (const f4*)W49, L, nsamp, argmax_vis, ldexpf(1.0f, -scale_log2), vis_out, eval_count,
                     range_flags() ? range_flags() + RB_RANGE_DVIS : nullptr);
  return check_launch("k_dvis_x6");
}
