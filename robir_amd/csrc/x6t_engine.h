// The two-tile form of the exact-operand ("f16x6") chunk-stream machine, round 4 (first built in vis_diffuse_x6t.hip).
//
// Round 3's stand-alone exact-operand kernels give a wave ONE 16-row tile: every weight fragment read from the LDS feeds one MFMA per
// product and a workgroup re-copies the whole net for 64 rows -- per chunk and CU 4 x 3 K / 32 KB of fragment reads and as many bytes of
// LDS-DMA writes next to 4 x 6 K / 32 MFMAs, a third of the time (profiles/r03_sdf_x6_ablation.md).  Here a wave holds the three-piece
// operands of TWO tiles for the current and the next layer (2 x 2 x 12 K / 32 registers), so a fragment feeds two MFMAs per product and a
// pass of the weights serves 128 rows: half the LDS traffic per MFMA.  The registers come from the fragments: not a chunk's worth but a
// rolling WINDOW of WK k-blocks (WK x 3 pieces x 4 registers), refilled piece by piece right behind the last run that reads it.
//
// A chunk (16 output neurons x K) = K / 32 / WK parts; a part = twelve runs of WK MFMAs on one accumulator each:
//     h.xh(A) h.xm(A) h.xl(A) h.xh(B) h.xm(B) h.xl(B)* | m.xh(A) m.xm(A) m.xh(B) m.xm(B)* | l.xh(A) l.xh(B)*        (* = the piece's refill)
// (A, B = the two tiles; classes: c0 += h.xh; c1 += h.xm, m.xh; c2 += h.xl, m.xm, l.xh: vis_diffuse_x6.hip has the arithmetic).  Parts
// alternate the direction in which they walk the window, so a part starts with the fragment requested LAST: one s_waitcnt covers the
// window (fragments return in order).  Every layer of the nets built on this has an even number of parts, so a layer starts upwards.
// Between the runs go the "fillers": the LDS-DMA copies of the chunk three ahead, the staged epilogue of the previous chunk.
#pragma once
#include "common.h"
#include "mlp_engine.h"
#include "x6_ring.h"

namespace rb {

typedef const __attribute__((address_space(3))) u4* xt_lds_u4p;

template <int KB>
struct XtOps {
  u4 h[2][KB], m[2][KB], l[2][KB];      // B operands of a layer, two tiles (one 128-bit tuple per k-block and piece)
};
struct XtWin {
  u4 h[4], m[4], l[4];                  // the fragment window: up to four k-blocks x three pieces
};

__host__ __device__ constexpr int xt_wk(int K) { return K % 128 == 0 ? 4 : (K % 96 == 0 ? 3 : 2); }     // 256 -> 4, 288 -> 3, 64 / 320 -> 2
__host__ __device__ constexpr int xt_parts(int K) { return K / 32 / xt_wk(K); }

// one 1 KB piece of a chunk copy: (chunk base + this wave's first piece) in SGPRs + (lane 16 | lane 16 + 4096) + an immediate 0..3 KB, which
// advances the global AND the LDS address (M0 = slot + this wave's first piece [+ 4096])
template <int OFF>
__device__ __forceinline__ void xt_dma16_imm(const f4* gbase_uniform, unsigned lane_byte_off, unsigned lds_byte_uniform) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%3" ::"s"(lds_byte_uniform), "v"(lane_byte_off),
               "s"(gbase_uniform), "n"(OFF)
               : "memory");
}
// lane 16 (+ extra) from no live register at all: a per-lane byte offset kept in a register is long-lived and rarely used -- the first
// thing the register allocator spills in these 512-register kernels, and every reload of it (a scratch load) drains the whole copy
// queue (s_waitcnt vmcnt(0)); a value re-derived from a register the fragment reads use (ring address + lane 16) got that register's
// live range split and spilled the same way.  Three vector instructions per copy instead.
template <int EXTRA>
__device__ __forceinline__ unsigned xt_lane16() {
  unsigned v;
  if constexpr (EXTRA == 0)
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_lshlrev_b32 %0, 4, %0" : "=v"(v));
  else
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_lshl_add_u32 %0, %0, 4, %1" : "=v"(v) : "s"(EXTRA));
  return v;
}
// piece i (0 .. sx_nsw(K) - 1) of this wave's span of a chunk copy; src / dst already point at the span's first piece
__device__ __forceinline__ void xt_copy_piece(int i, const f4* src_span, unsigned dst_span) {
  switch (i) {
    case 0: xt_dma16_imm<0>(src_span, xt_lane16<0>(), dst_span); break;
    case 1: xt_dma16_imm<1024>(src_span, xt_lane16<0>(), dst_span); break;
    case 2: xt_dma16_imm<2048>(src_span, xt_lane16<0>(), dst_span); break;
    case 3: xt_dma16_imm<3072>(src_span, xt_lane16<0>(), dst_span); break;
    case 4: xt_dma16_imm<0>(src_span, xt_lane16<4096>(), dst_span + 4096u); break;
    case 5: xt_dma16_imm<1024>(src_span, xt_lane16<4096>(), dst_span + 4096u); break;
    case 6: xt_dma16_imm<2048>(src_span, xt_lane16<4096>(), dst_span + 4096u); break;
    default: xt_dma16_imm<3072>(src_span, xt_lane16<4096>(), dst_span + 4096u); break;
  }
}
// The same pieces issued IN ORDER i = 0, 1, 2, ... by one wave within a chunk (the x6t kernels place them in consecutive filler
// positions): six issue slots per piece (three for the lane offset, M0, the hazard nop, the copy) were 12 % of a K = 256 chunk's slots
// (-DSXT_NODMA: 4.50 -> 3.88 ms per 2^20 points).  Piece 0 derives the lane offset and sets M0, piece 4 advances both by 4 KB, every
// other piece is ONE instruction: the offset register `lv` and M0 are carried from piece to piece.  Nothing else between the pieces of
// a chunk may write M0: no other instruction of these kernels does (checked in the ISA; sdf_back_x6t's gate copies, which set M0,
// are issued before piece 0).
template <int OFF>
__device__ __forceinline__ void xt_dma16_keep(const f4* gbase_uniform, unsigned lane_byte_off) {
  asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" ::"v"(lane_byte_off), "s"(gbase_uniform), "n"(OFF) : "memory");
}
// have_lv: `lv` already holds lane 16 (a copy issued earlier in the chunk derived it: sdf_back_x6t's gate copies)
__device__ __forceinline__ void xt_copy_piece_seq(int i, const f4* src_span, unsigned dst_span, unsigned& lv, bool have_lv = false) {
  switch (i) {
    case 0:
      if (!have_lv) lv = xt_lane16<0>();
      xt_dma16_imm<0>(src_span, lv, dst_span);
      break;
    case 1: xt_dma16_keep<1024>(src_span, lv); break;
    case 2: xt_dma16_keep<2048>(src_span, lv); break;
    case 3: xt_dma16_keep<3072>(src_span, lv); break;
    case 4:
      lv += 4096u;
      xt_dma16_imm<0>(src_span, lv, dst_span + 4096u);
      break;
    case 5: xt_dma16_keep<1024>(src_span, lv); break;
    case 6: xt_dma16_keep<2048>(src_span, lv); break;
    default: xt_dma16_keep<3072>(src_span, lv); break;
  }
}
// first piece of wave w's span: min(w NSW, NS - NSW) (the last wave's span is shifted back into the chunk: a few pieces are copied twice)
__device__ __forceinline__ int xt_span_first(int K_, int wave) {
  const int ns = 3 * K_ / 32, nsw = (ns + 3) / 4;
  return wave * nsw < ns - nsw ? wave * nsw : ns - nsw;
}

#define XT_MFMA_(ACC, W, X) ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, W), __builtin_bit_cast(h8, X), ACC, 0, 0, 0)

// The MFMAs of one chunk for two tiles.  PART0 = parts of this layer before this chunk (its parity = the walking direction of the first
// part).  filler(pos): called behind run pos = 0 .. 12 NPART - 1 of the chunk (a sched_barrier follows); refill(piece, part): requests
// the next part's fragments of `piece` into the window (called behind the last run of the part that reads the piece).
template <int K, int KBX, class Filler, class Refill>
__device__ __forceinline__ void xt_chunk(int part0, SxAcc (&acc)[2], XtWin& w, const XtOps<KBX>& x, Filler&& filler, Refill&& refill) {
  constexpr int WK = xt_wk(K), NPART = xt_parts(K);
  static_assert(K / 32 <= KBX, "");
#pragma unroll
  for (int part = 0; part < NPART; ++part) {
    const bool down = ((part0 + part) & 1) != 0;
    const int kb0 = part * WK;
#define XT_RUN_(ACC, WP, XP)                           \
  _Pragma("unroll") for (int k_ = 0; k_ < WK; ++k_) { \
    const int k = down ? WK - 1 - k_ : k_;             \
    XT_MFMA_(ACC, WP[k], XP[kb0 + k]);                 \
  }
  // Experiment switch (default off): interleave up to XT_MIX instructions of a position's fillers with the MFMAs of the run in front
  // of them through sched_group_barrier.  A wave that owns its SIMD issues in order, and tools/microbench/mfma_order.hip measures 17.9
  // cycles per MFMA with two independent vector instructions behind EVERY MFMA against 23.8 with eight behind every fourth -- but in
  // these kernels the interleaved stream draws more hazard wait states (fillers that write registers an MFMA in flight still reads)
  // and finer lgkmcnt waits than it hides: colour net unchanged for every mask and count, SDF distance 4.35 -> 4.55 ms.
#ifndef XT_MIX
#define XT_MIX 0
#endif
#ifndef XT_MIX_MASK
#define XT_MIX_MASK 0x002
#endif
#if XT_MIX > 0
#define XT_MIX_                                                      \
  _Pragma("unroll") for (int k_ = 0; k_ < WK; ++k_) {                \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);               \
    __builtin_amdgcn_sched_group_barrier(XT_MIX_MASK, XT_MIX, 0);    \
  }
#else
#define XT_MIX_
#endif
#define XT_END_(POS)            \
  filler(part * 12 + (POS));    \
  XT_MIX_                       \
  __builtin_amdgcn_sched_barrier(0)
    XT_RUN_(acc[0].c0, w.h, x.h[0]);
    XT_END_(0);
    XT_RUN_(acc[0].c1, w.h, x.m[0]);
    XT_END_(1);
    XT_RUN_(acc[0].c2, w.h, x.l[0]);
    XT_END_(2);
    XT_RUN_(acc[1].c0, w.h, x.h[1]);
    XT_END_(3);
    XT_RUN_(acc[1].c1, w.h, x.m[1]);
    XT_END_(4);
    XT_RUN_(acc[1].c2, w.h, x.l[1]);
    refill(0, part);
    XT_END_(5);
    XT_RUN_(acc[0].c1, w.m, x.h[0]);
    XT_END_(6);
    XT_RUN_(acc[0].c2, w.m, x.m[0]);
    XT_END_(7);
    XT_RUN_(acc[1].c1, w.m, x.h[1]);
    XT_END_(8);
    XT_RUN_(acc[1].c2, w.m, x.m[1]);
    refill(1, part);
    XT_END_(9);
    XT_RUN_(acc[0].c2, w.l, x.h[0]);
    XT_END_(10);
    XT_RUN_(acc[1].c2, w.l, x.h[1]);
    refill(2, part);
    XT_END_(11);
#undef XT_RUN_
#undef XT_END_
#undef XT_MIX_
  }
}

// The K-MAJOR form of a chunk (round 6, VERDICT r5 task 2): the same MFMAs, but k-block by k-block -- the twelve products of one k-block
// (six per tile) go to twelve DIFFERENT accumulator/piece combinations, so no two consecutive MFMAs extend the same chain -- with a filler
// slot behind EVERY MFMA instead of a cluster behind every run of WK.  tools/microbench/mfma_order.hip: two independent vector
// instructions behind every MFMA cost 17.9 cycles per MFMA, the same instructions in clusters of eight behind every fourth 23.8 -- only the
// last MFMA of a run shadows its cluster.  filler(slot): slot = 0 .. 12 K/32 - 1, at most ~2 single-issue instructions each (the caller
// cuts its epilogue into such steps); a sched_barrier pins every [MFMA, fillers] group.  refill(piece, part, k): the window entry k of
// `piece` is free (its last product of this part has been issued): request the fragment of the next part / the next chunk into it.
// Summation order: a class's products are added k-block by k-block (c2: h.xl, m.xm, l.xh of k-block 0, then of k-block 1, ...) instead of
// product by product over a part -- another fp32 order than xt_chunk's (last-bit differences, like the one-tile kernels').
template <int K, int KBX, class Filler, class Refill>
__device__ __forceinline__ void xt_chunk_km(SxAcc (&acc)[2], XtWin& w, const XtOps<KBX>& x, Filler&& filler, Refill&& refill) {
  constexpr int WK = xt_wk(K), NPART = xt_parts(K);
  static_assert(K / 32 <= KBX, "");
#pragma unroll
  for (int part = 0; part < NPART; ++part) {
#pragma unroll
    for (int k = 0; k < WK; ++k) {
      const int kb = part * WK + k, s0 = (part * WK + k) * 12;
#define XT_KM_(I, ACC, WP, XP)            \
  XT_MFMA_(ACC, WP[k], XP[kb]);           \
  filler(s0 + (I));                       \
  __builtin_amdgcn_sched_barrier(0)
      XT_KM_(0, acc[0].c0, w.h, x.h[0]);
      XT_KM_(1, acc[1].c0, w.h, x.h[1]);
      XT_KM_(2, acc[0].c1, w.h, x.m[0]);
      XT_KM_(3, acc[1].c1, w.h, x.m[1]);
      XT_KM_(4, acc[0].c2, w.h, x.l[0]);
      XT_MFMA_(acc[1].c2, w.h[k], x.l[1][kb]);
      refill(0, part, k);
      filler(s0 + 5);
      __builtin_amdgcn_sched_barrier(0);
      XT_KM_(6, acc[0].c1, w.m, x.h[0]);
      XT_KM_(7, acc[1].c1, w.m, x.h[1]);
      XT_KM_(8, acc[0].c2, w.m, x.m[0]);
      XT_MFMA_(acc[1].c2, w.m[k], x.m[1][kb]);
      refill(1, part, k);
      filler(s0 + 9);
      __builtin_amdgcn_sched_barrier(0);
      XT_KM_(10, acc[0].c2, w.l, x.h[0]);
      XT_MFMA_(acc[1].c2, w.l[k], x.h[1][kb]);
      refill(2, part, k);
      filler(s0 + 11);
      __builtin_amdgcn_sched_barrier(0);
#undef XT_KM_
    }
  }
}
// one window entry (k-block `kb` of the chunk in `slot_lane_addr`) of one piece
__device__ __forceinline__ void xt_request_one(u4& dst, unsigned slot_lane_addr, int kb, int piece) {
  dst = ((xt_lds_u4p)slot_lane_addr)[(kb * 3 + piece) * 64];
}

// positions of a chunk that carry fillers: all but the three refill runs of a part (5, 9, 11) -> 9 per part; index of position pos among
// them, or -1
__host__ __device__ constexpr int xt_free_index(int pos) {
  const int part = pos / 12, r = pos % 12;
  if (r == 5 || r == 9 || r == 11) return -1;
  return part * 9 + (r < 5 ? r : (r < 9 ? r - 1 : r - 2));
}
// item i of n spread evenly (and in order: items depend on earlier ones) over nfree positions -> the free index that carries it
__host__ __device__ constexpr int xt_item_slot(int i, int n, int nfree) { return (i * nfree) / n; }

// fragment window of (slot base address, first k-block, count) -> registers of one piece, requested in the order k = first .. last
// (down: last .. first)
__device__ __forceinline__ void xt_request(u4 (&dst)[4], unsigned slot_lane_addr, int kb_first, int count, int piece, bool down) {
  const xt_lds_u4p base = (xt_lds_u4p)slot_lane_addr;
#pragma unroll
  for (int k_ = 0; k_ < 4; ++k_)
    if (k_ < count) {
      const int k = down ? count - 1 - k_ : k_;
      dst[k] = base[((kb_first + k) * 3 + piece) * 64];
    }
}

}  // namespace rb
