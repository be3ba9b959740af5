// Shared pieces of the exact-operand ("f16x6") chunk-stream kernels (sdf_x6.hip, color_x6.hip): the LDS-DMA copy of a packed chunk
// (rb_pack_layer_x6: 16 bias floats, then [k-block][piece h | m | l][lane] float4), counted waits, the exact three-way split of an fp32
// value and the three accumulators by weight class.  See vis_diffuse_x6.hip for the arithmetic.
#pragma once
#include "common.h"
#include "mlp_engine.h"

namespace rb {

__host__ __device__ constexpr long sx_cf4(int K) { return 4 + 6L * K; }                    // float4s of a packed chunk (bias first)
// copies of a chunk by one wave: the bias head (one 256-byte instruction), then its span of the chunk's NS = 3 K / 32 fragment slices
// of 1 KB: NSW = ceil(NS / 4) consecutive slices from min(v NSW, NS - NSW) (the last wave's span is shifted back into the chunk: a few
// slices are copied twice), in blocks of <= 4 instructions that share one M0 write.
__host__ __device__ constexpr int sx_ns(int K) { return 3 * K / 32; }
__host__ __device__ constexpr int sx_nsw(int K) { return (sx_ns(K) + 3) / 4; }
__host__ __device__ constexpr int sx_np(int K) { return 1 + sx_nsw(K); }                   // instructions per wave and chunk
__host__ __device__ constexpr int sx_units(int K) { return 1 + (sx_nsw(K) + 3) / 4; }

__device__ __forceinline__ void sx_dma4(const f4* gbase_uniform, unsigned lane_byte_off, unsigned lds_byte_uniform) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(lds_byte_uniform), "v"(lane_byte_off), "s"(gbase_uniform) : "memory");
}
template <int NPC>
__device__ __forceinline__ void sx_dma_block(const f4* g, unsigned v, unsigned l) {
  static_assert(NPC >= 1 && NPC <= 4, "");
  if constexpr (NPC == 1) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(l), "v"(v), "s"(g) : "memory");
  else if constexpr (NPC == 2) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024" ::"s"(l), "v"(v), "s"(g) : "memory");
  else if constexpr (NPC == 3) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048" ::"s"(l), "v"(v), "s"(g) : "memory");
  else asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072" ::"s"(l), "v"(v), "s"(g) : "memory");
}
template <int K>
__device__ __forceinline__ void sx_copy_unit(int u, const f4* src_chunk, unsigned lane4, unsigned lane16, unsigned bias_dst,
                                             unsigned slot_dst, int wave) {
  constexpr int NS = sx_ns(K), NSW = sx_nsw(K);
  if (u == 0) {
    sx_dma4(src_chunk, lane4, bias_dst);
  } else {
    const int first = wave * NSW < NS - NSW ? wave * NSW : NS - NSW;
    const unsigned sb = (unsigned)(first + 4 * (u - 1)) * 1024u;
    const f4* src = src_chunk + 4 + sb / 16;
    if (NSW - 4 * (u - 1) >= 4) sx_dma_block<4>(src, lane16, slot_dst + sb);
    else if (NSW - 4 * (u - 1) == 3) sx_dma_block<3>(src, lane16, slot_dst + sb);
    else if (NSW - 4 * (u - 1) == 2) sx_dma_block<2>(src, lane16, slot_dst + sb);
    else sx_dma_block<1>(src, lane16, slot_dst + sb);
  }
}
template <int N>
__device__ __forceinline__ void sx_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
#ifndef SX_PK_SPLIT
#define SX_PK_SPLIT 0      // 1: the two scalings as v_pk_mul_f32: bit-identical, no gain (profiles/r03_sdf_x6_ablation.md)
#endif
// exact three-way split of two fp32 values (vis_diffuse_x6.hip): v = h + m 2^-11 + l 2^-22
__device__ __forceinline__ void sx_split_pair(float v0, float v1, float negk, unsigned& h, unsigned& m, unsigned& l) {
  const unsigned hu = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v0, v1));
#if SX_PK_SPLIT
  typedef float sx_f2 __attribute__((ext_vector_type(2)));
  const sx_f2 sv = sx_f2{v0, v1} * 2048.0f;             // v_pk_mul_f32: the same rounding as two v_mul_f32
  const float s0 = sv[0], s1 = sv[1];
#else
  const float s0 = v0 * 2048.0f, s1 = v1 * 2048.0f;
#endif
  float d0, d1;
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d0) : "v"(hu), "s"(negk), "v"(s0));
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d1) : "v"(hu), "s"(negk), "v"(s1));
  const unsigned mu = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(d0, d1));
#if SX_PK_SPLIT
  const sx_f2 ev = sx_f2{d0, d1} * 2048.0f;
  const float e0 = ev[0], e1 = ev[1];
#else
  const float e0 = d0 * 2048.0f, e1 = d1 * 2048.0f;
#endif
  unsigned lu;
  asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(lu) : "v"(mu), "s"(negk), "v"(e0));
  asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lu) : "v"(mu), "s"(negk), "v"(e1));
  h = hu;
  m = mu;
  l = lu;
}
struct SxAcc {
  f4 c0, c1, c2;   // classes 2^0, 2^-11, 2^-22
};


}  // namespace rb
