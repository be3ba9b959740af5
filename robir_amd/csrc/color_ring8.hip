// NeuS colour network (RenderingNetwork.forward, model/neus_model.py:535-560) on the eight-wave chunk-stream machine of
// sdf_ring8.hip -- split precision, round 3.
//
// k_color_mlp_h3 (mlp_kernels_h3.hip) runs the first-generation engine: weights global -> VGPR -> LDS, one __syncthreads per chunk,
// one wave per SIMD, a workgroup per 128 rows that re-streams the whole 1.1 MB net (36 % matrix-pipe occupancy, 30 % of its wave
// cycles parked).  Here: persistent workgroups of eight waves (two per SIMD, one 16-row tile each), the five layers as ONE cyclic
// stream of 65 chunks (16 output neurons x K; K = 320 for the first layer, 256 after) through a 4-slot LDS ring filled by LDS-DMA
// three chunks ahead under counted waits, weight fragments read from the ring just before use, the relu + hi/lo split of chunk j
// between the MFMAs of chunk j+1.  Inputs are not rows either: the 256 feature columns are read where the SDF network wrote them
// and [x | PE4(view) | normal] is encoded in the kernel (as k_color_mlp_h3<2>).  Same products in the same order, same bias lift,
// same epilogue arithmetic as the first-generation kernel: bit-identical rgb.
//
//   layer    0    1    2    3    4          K   320  256  256  256  256
//   chunks  16   16   16   16    1      first     0   16   32   48   64          65 chunks = 1 (mod 4): the slot table rotates by one
//   slices per wave and chunk (1 KB each; wave v copies slices v, v + 8 (, v + 16)): 3 for K = 320 (24 KB requested, 20 KB used), 2 after
#include "../../include/robir_hip.h"
#include "common.h"
#include "mlp_engine.h"
#include <type_traits>

namespace rb {

constexpr int C8_SLOT_B = 24 * 1024;
constexpr int C8_NCHUNK = 65;
constexpr long C8_CF320 = chunk_f4(320), C8_CF256 = chunk_f4(256);
__host__ __device__ constexpr long c8_coff(int c) {       // float4 offset of chunk c in the packed blob (rb_pack_layer_h3 x 5)
  return c < 16 ? (long)c * C8_CF320 : 16 * C8_CF320 + (long)(c - 16) * C8_CF256;
}
typedef float f4u8 __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ void c8_dma16(const f4* gbase_uniform, unsigned lane_byte_off, unsigned lds_byte_uniform) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_byte_uniform), "v"(lane_byte_off),
               "s"(gbase_uniform)
               : "memory");
}
template <int N>
__device__ __forceinline__ void c8_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__global__ __launch_bounds__(512, 1) void k_color_ring8(const float* __restrict__ feat, long feat_stride, float feat_scale,
                                                         const float* __restrict__ pxyz, float x_scale,
                                                         const float* __restrict__ pview, const float* __restrict__ pnormal, long M,
                                                         const f4* __restrict__ Wp, float us, float* __restrict__ rgb,
                                                         unsigned* __restrict__ range_word) {
  constexpr float AS = 16.0f;
  __shared__ f4 ring[4 * C8_SLOT_B / 16];              // 96 KB
  __shared__ f4 bias_tab[C8_NCHUNK * 4];
  __shared__ float tail_lds[8 * 16 * 48];              // 24 KB: the encoded tail of the round's rows
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long nrounds = (M + 127) >> 7;
  if (tid < C8_NCHUNK) {
    const f4* src = Wp + c8_coff(tid);
#pragma unroll
    for (int q = 0; q < 4; ++q) bias_tab[tid * 4 + q] = src[q];
  }
  __syncthreads();
  if ((long)blockIdx.x >= nrounds) return;

  const float zs = us * (1.0f / AS);
  const unsigned ring_b = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)ring);
  const unsigned lane16 = (unsigned)lane * 16u;
  unsigned slot_b[4] = {0u, (unsigned)C8_SLOT_B, 2u * C8_SLOT_B, 3u * C8_SLOT_B};
  unsigned sat = 0u;
  u4 xh[10], xl[10];                   // operands of the current layer (K <= 320: ten k-blocks of 32), one tile
  u4 yh[8], yl[8];                     // ... of the next layer (K = 256)
  f4u8 fraw[16];                       // the NEXT round's 256 feature columns of this lane's row (prefetched)
  float pv[3], px[3], pn[3];           // ... and its view direction / point / normal
  bool pok = false;
  long rrow = 0;

  auto fetch_inputs = [&](long round) {
    const long row = round * 128 + wave * 16 + (lane & 15);
    const bool ok = round < nrounds && row < M;
    const long rr = ok ? row : 0;
    const float* pf = feat + rr * feat_stride + g * 4;
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) fraw[kb] = *reinterpret_cast<const f4u8*>(pf + kb * 16);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      pv[c] = pview[3 * rr + c];
      px[c] = pxyz[3 * rr + c];
      pn[c] = pnormal[3 * rr + c];
    }
    pok = ok;
  };
  // this round's rows -> operands of layer 0: features in place (x feat_scale), tail encoded by the four lanes of a row
  auto load_layer0 = [&]() {
    float in0[1][76];
#pragma unroll
    for (int kb = 0; kb < 16; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) in0[0][kb * 4 + r] = pok ? fraw[kb][r] * feat_scale : 0.f;
    float* trow = tail_lds + (wave * 16 + (lane & 15)) * 48;
    if (g == 0) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        trow[c] = px[c] * x_scale;
        trow[3 + c] = pv[c];
        trow[30 + c] = pn[c];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 5; ++i) trow[33 + (g - 1) * 5 + i] = 0.f;
    }
#pragma unroll 1
    for (int j = g; j < 12; j += 4) {         // write_pe<4>: frequency k = j / 3, axis c = j % 3
      const int k = j / 3, c = j - 3 * k;
      float sn, cs;
      sincosf((c == 0 ? pv[0] : (c == 1 ? pv[1] : pv[2])) * (float)(1 << k), &sn, &cs);
      trow[6 + 6 * k + c] = sn;
      trow[6 + 6 * k + 3 + c] = cs;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // same wave, in-order LDS: all four lane groups have written
    const f4* pt = reinterpret_cast<const f4*>(trow) + g;
#pragma unroll
    for (int kb = 0; kb < 3; ++kb) {
      const f4 v = pok ? pt[kb * 4] : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r) in0[0][64 + kb * 4 + r] = v[r];
    }
    unsigned ih[1][10][4], il[1][10][4];
    split_operands<320, 76, 1>(in0, ih, il, AS);
#pragma unroll
    for (int kb = 0; kb < 10; ++kb) {
      xh[kb] = u4{ih[0][kb][0], ih[0][kb][1], ih[0][kb][2], ih[0][kb][3]};
      xl[kb] = u4{il[0][kb][0], il[0][kb][1], il[0][kb][2], il[0][kb][3]};
#pragma unroll
      for (int q = 0; q < 4; ++q) sat = sat_acc(sat, ih[0][kb][q]);
    }
  };
  // relu(z * zs) * AS of output block jb -> operands of the next layer: k-block jb/2, registers 2*(jb&1)+q (act_split<.., ACT_RELU>)
  auto hidden_piece = [&](const f4& acc, int jb, int q) {
    unsigned hi, lo;
    split_pair_mix(fmaxf(acc[2 * q] * zs, 0.f) * AS, fmaxf(acc[2 * q + 1] * zs, 0.f) * AS, hi, lo);
    yh[jb >> 1][(jb & 1) * 2 + q] = hi;
    yl[jb >> 1][(jb & 1) * 2 + q] = lo;
    sat = sat_acc(sat, hi);
  };

  // ---- one layer (sdf_ring8.hip's run_layer).  Compile time: K, NCH, LAST (output layer), NF = slices per wave of the three chunks
  // that follow the layer in the stream, as decimal digits.  Run time: src_of(j) = packed chunk j counted from the layer's first
  // (j runs three past its last), cb = stream index of its first chunk.
  auto run_layer = [&](auto K_tag, auto NCH_tag, auto LAST_tag, auto NF_tag, auto src_of, int cb) {
    constexpr int K = decltype(K_tag)::value, KB = K / 32, NCH = decltype(NCH_tag)::value;
    constexpr bool LAST = decltype(LAST_tag)::value != 0;
    constexpr int NFS = decltype(NF_tag)::value, NP = K == 320 ? 3 : 2;
    constexpr int NF0 = NFS / 100, NF1 = (NFS / 10) % 10, NF2 = NFS % 10;
    f4 accs[2];
#pragma unroll
    for (int jb = 0; jb < NCH; ++jb) {
      f4& acc = accs[jb & 1];
      acc = bias_tab[(cb + jb) * 4 + g] * AS;
      {   // chunk jb must have landed: this wave's slices of chunks jb+1 and jb+2 may still be in flight
        const int n1 = jb + 1 < NCH ? NP : (jb + 1 == NCH ? NF0 : NF1);
        const int n2 = jb + 2 < NCH ? NP : (jb + 2 == NCH ? NF0 : (jb + 2 == NCH + 1 ? NF1 : NF2));
        const int allowed = n1 + n2;
        if (allowed <= 4) c8_wait<4>(); else if (allowed == 5) c8_wait<5>(); else c8_wait<6>();
      }
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const u4* frag = reinterpret_cast<const u4*>(reinterpret_cast<const char*>(ring) + slot_b[(cb + jb) & 3]) + lane;
      const int n3 = jb + 3 < NCH ? NP : (jb + 3 == NCH ? NF0 : (jb + 3 == NCH + 1 ? NF1 : NF2));
      const f4* src3 = src_of(jb + 3) + 4 + wave * 64;
      const unsigned dst3 = ring_b + slot_b[(cb + jb + 3) & 3] + (unsigned)wave * 1024u;
      u4 wfa[3], wfb[3];
#ifdef C8_ACC2
      f4 acc2;
#endif
      wfa[0] = frag[0];
      wfb[0] = frag[64];
      wfa[1] = frag[2 * 64];
      wfb[1] = frag[3 * 64];
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const h8 wh = __builtin_bit_cast(h8, wfa[kb % 3]), wlo = __builtin_bit_cast(h8, wfb[kb % 3]);
        if (kb + 2 < KB) {
          wfa[(kb + 2) % 3] = frag[(2 * kb + 4) * 64];
          wfb[(kb + 2) % 3] = frag[(2 * kb + 5) * 64];
        }
        const h8 a = __builtin_bit_cast(h8, xh[kb]), b = __builtin_bit_cast(h8, xl[kb]);
#ifdef C8_ACC2
        f4& ak = (kb & 1) ? acc2 : acc;
        if (kb == 1) {
          acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, b, f4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        } else {
          ak = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, b, ak, 0, 0, 0);
        }
        ak = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, a, ak, 0, 0, 0);
        ak = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo, a, ak, 0, 0, 0);
        if (kb == KB - 1) acc = acc + acc2;
#else
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, b, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, a, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo, a, acc, 0, 0, 0);
#endif
        if (jb > 0 && !LAST) {
          if (kb == 1) hidden_piece(accs[(jb - 1) & 1], jb - 1, 0);
          if (kb == 4) hidden_piece(accs[(jb - 1) & 1], jb - 1, 1);
        }
#pragma unroll
        for (int d = 0; d < 3; ++d)
          if (d < n3 && 2 * d + 1 == kb) c8_dma16(src3 + d * 512, lane16, dst3 + (unsigned)d * 8192u);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    const f4& last = accs[(NCH - 1) & 1];
    if constexpr (LAST) {
      if (g == 0 && rrow < M) {
#pragma unroll
        for (int c = 0; c < 3; ++c) rgb[rrow * 3 + c] = 1.0f / (1.0f + expf(-(last[c] * zs)));
      }
    } else {
      hidden_piece(last, NCH - 1, 0);
      hidden_piece(last, NCH - 1, 1);
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  auto y_to_x = [&]() {
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
      xh[kb] = yh[kb];
      xl[kb] = yl[kb];
    }
  };

  // ---- prologue: chunks 0, 1, 2 of the stream (K = 320: three slices per wave); first round's inputs
  long round = blockIdx.x;
  fetch_inputs(round);
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int d = 0; d < 3; ++d)
      c8_dma16(Wp + c8_coff(c) + 4 + wave * 64 + d * 512, lane16, ring_b + slot_b[c] + (unsigned)wave * 1024u + (unsigned)d * 8192u);
  c8_wait<0>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  for (; round < nrounds; round += gridDim.x) {
    rrow = round * 128 + wave * 16 + (lane & 15);
    load_layer0();
    {   // layer 0 (K = 320), followed by layer 1
      const f4* w0 = Wp;
      const f4* w1 = Wp + 16 * C8_CF320;
      asm volatile("" : "+s"(w0), "+s"(w1));
      run_layer(std::integral_constant<int, 320>{}, std::integral_constant<int, 16>{}, I0{}, std::integral_constant<int, 222>{},
                [&](int j) { return j < 16 ? w0 + (long)j * C8_CF320 : w1 + (long)(j - 16) * C8_CF256; }, 0);
    }
    y_to_x();
    fetch_inputs(round + gridDim.x);               // next round's rows: consumed at the top of the next round
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int l = 1; l <= 3; ++l) {                  // hidden layers: one copy of the code
      const f4* wl = Wp + 16 * C8_CF320 + (long)(l - 1) * 16 * C8_CF256;
      const f4* wstart = Wp;                        // after layer 3: the output chunk, then chunks 0, 1 of the next round (K = 320)
      asm volatile("" : "+s"(wl), "+s"(wstart));
      if (l < 3) {
        run_layer(std::integral_constant<int, 256>{}, std::integral_constant<int, 16>{}, I0{}, std::integral_constant<int, 222>{},
                  [&](int j) { return wl + (long)j * C8_CF256; }, 16 * l);
      } else {
        run_layer(std::integral_constant<int, 256>{}, std::integral_constant<int, 16>{}, I0{}, std::integral_constant<int, 233>{},
                  [&](int j) { return j <= 16 ? wl + (long)j * C8_CF256 : wstart + (long)(j - 17) * C8_CF320; }, 48);
      }
      y_to_x();
    }
    {   // output layer (one chunk), followed by chunks 0, 1, 2 of the next round
      const f4* w4 = Wp + 16 * C8_CF320 + 48 * C8_CF256;
      const f4* wstart = Wp;
      asm volatile("" : "+s"(w4), "+s"(wstart));
      run_layer(std::integral_constant<int, 256>{}, std::integral_constant<int, 1>{}, I1{}, std::integral_constant<int, 333>{},
                [&](int j) { return j < 1 ? w4 : wstart + (long)(j - 1) * C8_CF320; }, 64);
    }
    {   // 65 chunks = 1 (mod 4): the stream continues one slot on
      const unsigned a = slot_b[0];
      slot_b[0] = slot_b[1];
      slot_b[1] = slot_b[2];
      slot_b[2] = slot_b[3];
      slot_b[3] = a;
    }
  }
  range_report(sat, range_word);
  c8_wait<0>();
  __syncthreads();
}

}  // namespace rb

using namespace rb;

extern "C" int rb_color_ring_points(const float* feat, long feat_stride, float feat_scale, const float* x, float x_scale,
                                    const float* view, const float* normal, long M, const float* Wp, int scale_log2, float* rgb,
                                    int n_workgroups, rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(feat && x && view && normal && Wp && rgb, "null pointer");
  const int pg = persistent_grid((M + 127) / 128, n_workgroups);
  if (pg <= 0) return rb::fail(__func__, "device query failed");
  const unsigned grid = (unsigned)pg;
  hipLaunchKernelGGL(k_color_ring8, dim3(grid), dim3(512), 0, (hipStream_t)stream, feat, feat_stride, feat_scale, x, x_scale, view, normal,
                     M, (const f4*)Wp, ldexpf(1.0f, -scale_log2), rgb, range_flags() ? range_flags() + RB_RANGE_COLOR : nullptr);
  return check_launch("k_color_ring8");
}
