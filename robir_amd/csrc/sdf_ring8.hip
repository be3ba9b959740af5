// NeuS SDF network, value rows, split precision -- the chunk-stream machine of sdf_ring.hip with TWO waves per SIMD.
//
// tools/ubench/valu_issue.hip: for the single wave a SIMD holds in sdf_ring.hip, vector work does not hide behind MFMAs (one
// v_mul behind each 16x16x32 MFMA costs 1 cycle, every further one 4: matrix time and vector time ADD), and an LDS-DMA row
// costs its issuing wave 60-185 cycles.  k_sdf_ring therefore spends 1800 cycles per chunk on 768 cycles of MFMAs.  Here a
// workgroup is eight waves of ONE 16-row tile each (the same 128 rows per round): half the operand registers per wave (<= 256,
// so two waves share a SIMD and one's softplus / split / copies issue while the other's MFMAs run), two LDS-DMA slices per wave
// and chunk instead of four, weight fragments read from the ring just before use (two k-blocks in registers instead of a whole
// chunk).  Arithmetic per element, packed weights, results and the layout of the sigmoid blob are those of k_sdf_ring.
//
// Stream (as sdf_ring.hip):   layer 0  1  2  3  4  5  6  7  8      K 64 256 256 256 288 256 256 256 256
//                             chunks 16 16 16 13 16 16 16 16 17|1  first 0 16 32 48 61 77 93 109 125
// Ring: four slots of 24 KB.  A chunk of K inputs is K/16 KB + bias; wave v copies the 1 KB slices v, v + 8 (, v + 16) of it
// -- 1 / 2 / 3 slices per wave for K = 64 / 256 / 288, i.e. 8 / 16 / 24 KB per chunk: more than the chunk where K is not 256
// (the surplus is the following chunk's bytes, never used), so that every wave requests the same number of rows and the
// counted wait is the same immediate for all of them.
#include "../../include/robir_hip.h"
#include "common.h"
#include "mlp_engine.h"
#include <cstdlib>
#include <type_traits>

// Timing ablations (tools/ab_sdf.py; results are WRONG with any of them): -DS8_ABL_NOBAR no per-chunk barrier, -DS8_ABL_NODMA no
// weight copies, -DS8_ABL_NOLDS one fragment read per chunk instead of sixteen, -DS8_ABL_NOVAL no softplus / split / output work.
namespace rb {

constexpr int S8_SLOT_B = 24 * 1024;
__host__ __device__ constexpr int s8_nch(int l, int last) { return l == 3 ? 13 : (l == 8 ? last : 16); }
__host__ __device__ constexpr int s8_cbase(int l) { return l < 4 ? 16 * l : 61 + 16 * (l - 4); }
__host__ __device__ constexpr int s8_K(int l) { return l == 0 ? 64 : (l == 4 ? 288 : 256); }
__host__ __device__ constexpr int s8_layer_of(int c) { return c < 48 ? (c >> 4) : (c < 61 ? 3 : (c < 125 ? 4 + ((c - 61) >> 4) : 8)); }
__host__ __device__ constexpr long s8_loff(int l) {       // float4 offset of layer l in the packed blob (sdf_ring.hip)
  return l == 0 ? 0L : (l <= 3 ? 4160L + 16448L * (l - 1) : (l == 4 ? 50420L : 68916L + 16448L * (l - 5)));
}
__host__ __device__ constexpr long s8_coff(int c) {
  const int l = s8_layer_of(c);
  return s8_loff(l) + (long)(c - s8_cbase(l)) * chunk_f4(s8_K(l));
}
__host__ __device__ constexpr int s8_np(int K) { return K == 64 ? 1 : (K == 256 ? 2 : 3); }     // slices per wave and chunk

__device__ __forceinline__ void s8_dma16(const f4* gbase_uniform, unsigned lane_byte_off, unsigned lds_byte_uniform) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_byte_uniform), "v"(lane_byte_off),
               "s"(gbase_uniform)
               : "memory");
}
__device__ __forceinline__ void s8_store16(const f4* base_uniform, unsigned lane_byte_off, f4 v) {
  // (a store of more than 64 bits reads its data registers late: wait states before a VALU may overwrite them)
  asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(lane_byte_off), "v"(v), "s"(base_uniform) : "memory");
}
template <int N>
__device__ __forceinline__ void s8_wait() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// MODE: 0 = signed distance only, 1 = all 257 outputs, 5 = all outputs + the sigmoid blob of the reverse-mode gradient
// FUSED: the positional encoding is computed HERE from the points xyz[M,3] (x in_scale) instead of being read as rows X[M,64]
// (model/embedder.py:17-38 + model/neus_model.py:385-417 in one kernel): once per round the four lanes that share a point
// evaluate its 30 (frequency, axis) sine / cosine pairs between them -- the same sincosf of the same argument as k_feat_pe10
// (mlp_kernels.hip), so the features and everything downstream are bit-identical to the row form --, exchange them through a
// 4 KB LDS scratch per wave and pick up their sixteen B-operand features.  8 x 16 x 30 sincosf per round of 142 chunks: 1 % of
// the round; the 256 B per point of feature rows and the encoding kernel disappear.
template <int MODE, bool FUSED>
__global__ __launch_bounds__(512, 1) void k_sdf_ring8(const float* __restrict__ X, const float* __restrict__ xyz, float in_scale,
                                                       long M, const f4* __restrict__ Wp, float us,
                                                       float out_scale, float* __restrict__ out0,
                                                       unsigned* __restrict__ range_word, f4* __restrict__ sig) {
  constexpr bool FULL = (MODE & 1) != 0;
  constexpr bool STORE = (MODE & 4) != 0;
  static_assert(MODE == 0 || MODE == 1 || MODE == 5, "value rows only");
  constexpr int LAST = FULL ? 17 : 1;
  constexpr int NCHUNK = 125 + LAST;                  // 142 / 126: = 2 (mod 4)
  constexpr float AS = 64.0f;
  __shared__ f4 ring[4 * S8_SLOT_B / 16];             // 96 KB
  __shared__ f4 bias_tab[NCHUNK * 4];
  __shared__ float feat_lds[FUSED ? 8 * 16 * 64 : 4];   // FUSED: one 64-float encoding row per point of the round (32 KB)
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // 0 .. 7: rows wave * 16 .. + 15 of the round
  const long nrounds = (M + 127) >> 7;
  if (tid < NCHUNK) {
    const f4* src = Wp + s8_coff(tid);
#pragma unroll
    for (int q = 0; q < 4; ++q) bias_tab[tid * 4 + q] = src[q];
  }
  __syncthreads();
  if ((long)blockIdx.x >= nrounds) return;

  const float zs = us * (1.0f / AS);
  const float inv_sqrt2 = 0.70710678118654752440f;
  const float os = out_scale * us * (1.0f / AS);
  const unsigned ring_b = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)ring);
  const unsigned lane16 = (unsigned)lane * 16u;
  // the sigmoid blob keeps the layout of k_sdf_ring (four waves x two tiles): wave v is tile v & 1 of that kernel's wave v >> 1
  const unsigned sig_lane_off = (unsigned)((wave >> 1) * 64 + lane) * 16u;
  unsigned slot_b[4] = {0u, (unsigned)S8_SLOT_B, 2u * S8_SLOT_B, 3u * S8_SLOT_B};
  unsigned sat = 0u;
  u4 xh[9], xl[9];                     // operands of the current layer (K <= 288: nine k-blocks of 32), one tile
  u4 yh[9], yl[9];                     // ... of the next layer
  u4 shh[2], shl[2];                   // the 64 input features / sqrt(2), lifted: skip operands of layer 4
  f4 fraw[4];                          // input features of the NEXT round (prefetched)
  long rrow = 0;

  float pxyz[3] = {0.f, 0.f, 0.f};     // FUSED: the NEXT round's point of this lane (prefetched like the rows)
  bool pok = false;
  auto fetch_features = [&](long round) {
    const long row = round * 128 + wave * 16 + (lane & 15);
    const bool ok = round < nrounds && row < M;
    if constexpr (FUSED) {
      const float* p = xyz + (ok ? row : 0) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) pxyz[c] = p[c];
      pok = ok;
    } else {
      const f4* p = reinterpret_cast<const f4*>(X + (ok ? row : 0) * 64) + g;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) fraw[kb] = ok ? p[kb * 4] : f4{0.f, 0.f, 0.f, 0.f};
    }
  };
  // FUSED: encoding of this round's 16 points of the wave -> fraw (lane (n, g): features 16 kb + 4 g + r of point n).
  // Lane group g evaluates the (frequency k, axis c) pairs j = g, g + 4, ... < 30 (j = 3 k + c) of its point.
  auto encode_round = [&]() {
    float* frow = feat_lds + (wave * 16 + (lane & 15)) * 64;
    const float a[3] = {pxyz[0] * in_scale, pxyz[1] * in_scale, pxyz[2] * in_scale};
    if (g == 0) {
      frow[0] = a[0];
      frow[1] = a[1];
      frow[2] = a[2];
      frow[63] = 0.f;
    }
#pragma unroll 1
    for (int j = g; j < 30; j += 4) {
      const int k = j / 3, c = j - 3 * k;
      const float fr = (float)(1 << k);
      float sn, cs;
      sincosf((c == 0 ? a[0] : (c == 1 ? a[1] : a[2])) * fr, &sn, &cs);
      frow[3 + 6 * k + c] = sn;
      frow[3 + 6 * k + 3 + c] = cs;
    }
    // (same wave, in-order LDS: the reads below follow the writes of all four lane groups)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const f4* fr4 = reinterpret_cast<const f4*>(frow) + g;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) fraw[kb] = pok ? fr4[kb * 4] : f4{0.f, 0.f, 0.f, 0.f};
  };

  // ---- value rows, staged over three k-block gaps (sdf_ring.hip): per piece q = register pair (2q, 2q+1) of the tile
  f4 sstage = {0.f, 0.f, 0.f, 0.f};
  const f4* sig_round = sig;
  float pz[2][2], pt[2][2], pe[2][2], pr[2][2], pl[2][2];
  auto val_stage1 = [&](const f4& acc, int q) {                // softplus100_fast (mlp_engine.h), cut at its transcendentals
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float z = acc[2 * q + e] * zs;
      pz[q][e] = z;
      pt[q][e] = z * SP_T_PER_Z;
      pe[q][e] = __builtin_amdgcn_exp2f(pt[q][e]);
    }
  };
  auto val_stage2 = [&](int q) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float u = 1.0f + pe[q][e];
      if constexpr (STORE) pr[q][e] = __builtin_amdgcn_rcpf(u);
      pl[q][e] = __builtin_amdgcn_logf(u);
    }
  };
  auto val_stage3 = [&](int jb, int q, float sa, int cb) {
    float v[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const bool lin = pt[q][e] > SP_T_LINEAR;
      const float sp = pl[q][e] * SP_LN2_OVER_100;
      v[e] = lin ? pz[q][e] : sp;
      if constexpr (STORE) sstage[2 * q + e] = lin ? 1.0f : pe[q][e] * pr[q][e];
    }
    unsigned hi, lo;
    split_pair_mix(v[0] * sa, v[1] * sa, hi, lo);
    sat = sat_acc(sat, hi);
    yh[jb >> 1][(jb & 1) * 2 + q] = hi;
    yl[jb >> 1][(jb & 1) * 2 + q] = lo;
    if constexpr (STORE) {
      if (q == 1) {
        const f4* base = sig_round;
        asm volatile("" : "+s"(base));
        s8_store16(base + ((cb + jb) * 2 + (wave & 1)) * 256, sig_lane_off, sstage);
      }
    }
  };
  auto output_piece = [&](const f4& acc, int jb, int q) {
    if (rrow >= M) return;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int r = 2 * q + e;
      const int j = jb * 16 + 4 * g + r;
      if constexpr (FULL) {
        if (j < 257) out0[rrow * 257 + j] = acc[r] * os;
      } else {
        if (g == 0 && r == 0) out0[rrow] = acc[r] * os;
      }
    }
  };

  // ---- one layer.  Compile time: K, NCH, EPI (0 softplus, 1 = layer 3: x 1/sqrt 2, 2 = output stores), NF = slices per wave of
  // the three chunks that follow the layer in the stream, as decimal digits.  Run time: src_of(j) = packed chunk j counted from
  // the layer's first (j runs three past its last), cb = stream index of its first chunk, sl[k] = ring slot of chunk cb + k.
  auto run_layer = [&](auto K_tag, auto NCH_tag, auto EPI_tag, auto NF_tag, auto src_of, int cb, const unsigned (&sl)[4]) {
    constexpr int K = decltype(K_tag)::value, KB = K / 32, NCH = decltype(NCH_tag)::value, EPI = decltype(EPI_tag)::value;
    constexpr int NFS = decltype(NF_tag)::value, NP = s8_np(K);
    constexpr int NF0 = NFS / 100, NF1 = (NFS / 10) % 10, NF2 = NFS % 10;
    const float sa = (EPI == 1 ? inv_sqrt2 : 1.0f) * AS;
    f4 accs[2];
#pragma unroll
    for (int jb = 0; jb < NCH; ++jb) {
      f4& acc = accs[jb & 1];
      acc = bias_tab[(cb + jb) * 4 + g] * AS;
      // chunk jb must have landed: this wave's slices of chunks jb+1 and jb+2 may still be in flight (stores are not credited)
      {
        const int n1 = jb + 1 < NCH ? NP : (jb + 1 == NCH ? NF0 : NF1);
        const int n2 = jb + 2 < NCH ? NP : (jb + 2 == NCH ? NF0 : (jb + 2 == NCH + 1 ? NF1 : NF2));
        const int allowed = n1 + n2;
#ifndef S8_REGSTAGE
        if (allowed <= 2) s8_wait<2>(); else if (allowed == 3) s8_wait<3>(); else if (allowed == 4) s8_wait<4>();
        else if (allowed == 5) s8_wait<5>(); else s8_wait<6>();
#else
        (void)allowed;
#endif
      }
#ifndef S8_ABL_NOBAR
      __builtin_amdgcn_s_barrier();
#endif
      asm volatile("" ::: "memory");
      const u4* frag = reinterpret_cast<const u4*>(reinterpret_cast<const char*>(ring) + sl[jb & 3]) + lane;
      const int n3 = jb + 3 < NCH ? NP : (jb + 3 == NCH ? NF0 : (jb + 3 == NCH + 1 ? NF1 : NF2));
      const f4* src3 = src_of(jb + 3) + 4 + wave * 64;                       // this wave's first 1 KB slice of chunk jb+3
      const unsigned dst3 = ring_b + sl[(jb + 3) & 3] + (unsigned)wave * 1024u;
#ifdef S8_REGSTAGE
      // register staging instead of LDS-DMA: this wave's slices of chunk jb+3 are loaded at the top of chunk jb and stored into
      // the ring slot of chunk jb-1 (free since this chunk's barrier) at its end; in-order LDS puts the stores ahead of the
      // fragment reads of the next two chunks, whose waits therefore cover them before the barrier of chunk jb+3
      f4 stg[3];
#pragma unroll
      for (int d = 0; d < 3; ++d)
        if (d < n3) stg[d] = src3[d * 512 + lane];
#endif
      // weight fragments: three k-blocks in registers, read two k-blocks (six MFMAs, ~100 cycles: the LDS latency) ahead
      u4 wfa[3], wfb[3];
      wfa[0] = frag[0];
      wfb[0] = frag[64];
      if (KB > 1) {
        wfa[1] = frag[2 * 64];
        wfb[1] = frag[3 * 64];
      }
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const h8 wh = __builtin_bit_cast(h8, wfa[kb % 3]), wlo = __builtin_bit_cast(h8, wfb[kb % 3]);
#ifndef S8_ABL_NOLDS
        if (kb + 2 < KB) {
          wfa[(kb + 2) % 3] = frag[(2 * kb + 4) * 64];
          wfb[(kb + 2) % 3] = frag[(2 * kb + 5) * 64];
        }
#else
        if (kb + 2 < KB) {
          wfa[(kb + 2) % 3] = wfa[kb % 3];
          wfb[(kb + 2) % 3] = wfb[kb % 3];
        }
#endif
        const h8 a = __builtin_bit_cast(h8, xh[kb]), b = __builtin_bit_cast(h8, xl[kb]);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, b, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, a, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo, a, acc, 0, 0, 0);
#ifndef S8_ABL_NOVAL
        if (jb > 0) {
          const f4& prev = accs[(jb - 1) & 1];
          if constexpr (EPI == 2) {
            if (kb == 0) output_piece(prev, jb - 1, 0);
            if (kb == (KB > 4 ? 4 : KB - 1)) output_piece(prev, jb - 1, 1);
          } else if constexpr (KB >= 8) {
            if (kb == 2) val_stage3(jb - 1, 0, sa, cb);
            if (kb == 5) val_stage3(jb - 1, 1, sa, cb);
            if (kb == 1) val_stage2(0);
            if (kb == 4) val_stage2(1);
            if (kb == 0) val_stage1(prev, 0);
            if (kb == 3) val_stage1(prev, 1);
          } else {
            if (kb == 0) { val_stage1(prev, 0); val_stage2(0); val_stage3(jb - 1, 0, sa, cb); }
            if (kb == KB - 1) { val_stage1(prev, 1); val_stage2(1); val_stage3(jb - 1, 1, sa, cb); }
          }
        }
#else
        if (jb > 0 && kb == 0) { yh[(jb - 1) >> 1][0] ^= __builtin_bit_cast(unsigned, accs[(jb - 1) & 1][0]); }
#endif
#ifndef S8_ABL_NODMA
#ifdef S8_REGSTAGE
        if (kb == KB - 1) {
          f4* wdst = reinterpret_cast<f4*>(reinterpret_cast<char*>(ring) + sl[(jb + 3) & 3]) + wave * 64 + lane;
#pragma unroll
          for (int d = 0; d < 3; ++d)
            if (d < n3) wdst[d * 512] = stg[d];
        }
#else
#pragma unroll
        for (int d = 0; d < 3; ++d)
          if (d < n3 && (2 * d + 1 < KB ? 2 * d + 1 : KB - 1) == kb) s8_dma16(src3 + d * 512, lane16, dst3 + (unsigned)d * 8192u);
#endif
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // the last chunk's results: no MFMA follows that would cover the accumulator latency, the compiler inserts the wait states
    {
      const f4& prev = accs[(NCH - 1) & 1];
      if constexpr (EPI == 2) {
        output_piece(prev, NCH - 1, 0);
        output_piece(prev, NCH - 1, 1);
      } else {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          val_stage1(prev, q);
          val_stage2(q);
          val_stage3(NCH - 1, q, sa, cb);
        }
      }
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  auto y_to_x = [&](int nkb) {
#pragma unroll
    for (int kb = 0; kb < 9; ++kb)
      if (kb < nkb) {
        xh[kb] = yh[kb];
        xl[kb] = yl[kb];
      }
  };
  // slices per wave with which the stream's first two chunks (K = 64) are requested: by the output layer's own look-ahead (one
  // each) when it has 17 chunks, by layer 7's shared code (two each, like every 256-wide chunk) when it has one
  constexpr int NP_HEAD = FULL ? 1 : 2;

  // ---- prologue: chunks 0, 1, 2 of the stream; first round's features
  long round = blockIdx.x;
  fetch_features(round);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int n = c < 2 ? NP_HEAD : 1;
#pragma unroll
    for (int d = 0; d < 2; ++d)
      if (d < n) {
#ifdef S8_REGSTAGE
        reinterpret_cast<f4*>(reinterpret_cast<char*>(ring) + slot_b[c])[wave * 64 + d * 512 + lane] = (Wp + s8_coff(c) + 4 + wave * 64 + d * 512)[lane];
#else
        s8_dma16(Wp + s8_coff(c) + 4 + wave * 64 + d * 512, lane16, ring_b + slot_b[c] + (unsigned)wave * 1024u + (unsigned)d * 8192u);
#endif
      }
  }
#ifdef S8_REGSTAGE
  __syncthreads();
#else
  s8_wait<0>();
#endif
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  constexpr long CF256 = chunk_f4(256);
  for (; round < nrounds; round += gridDim.x) {
    rrow = round * 128 + wave * 16 + (lane & 15);
    if constexpr (STORE) sig_round = sig + round * (125L * 2 * 256);
    // ---- input features -> operands of layer 0 and the skip operands of layer 4
    if constexpr (FUSED) encode_round();
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const f4 v = fraw[kb];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        unsigned hi, lo;
        split_pair_mix(v[2 * q] * AS, v[2 * q + 1] * AS, hi, lo);
        xh[kb / 2][(kb & 1) * 2 + q] = hi;
        xl[kb / 2][(kb & 1) * 2 + q] = lo;
        sat = sat_acc(sat, hi);
        split_pair_mix(v[2 * q] * inv_sqrt2 * AS, v[2 * q + 1] * inv_sqrt2 * AS, hi, lo);
        shh[kb / 2][(kb & 1) * 2 + q] = hi;
        shl[kb / 2][(kb & 1) * 2 + q] = lo;
      }
    }
    {   // layer 0 (K = 64): followed by layer 1
      const unsigned s4[4] = {slot_b[0], slot_b[1], slot_b[2], slot_b[3]};
      const f4* w0 = Wp;
      const f4* w1 = Wp + s8_loff(1);
      asm volatile("" : "+s"(w0), "+s"(w1));
      run_layer(std::integral_constant<int, 64>{}, std::integral_constant<int, 16>{}, I0{}, std::integral_constant<int, 222>{},
                [&](int j) { return j < 16 ? w0 + (long)j * chunk_f4(64) : w1 + (long)(j - 16) * CF256; }, 0, s4);
    }
    y_to_x(8);
#pragma unroll 1
    for (int seg = 0; seg < 2; ++seg) {
      const int nrep = seg == 0 ? 2 : 3;
#pragma unroll 1
      for (int rep = 0; rep < nrep; ++rep) {       // layers 1, 2 (seg 0) and 5, 6, 7 (seg 1): one copy of the code
        const int cb = seg == 0 ? 16 + 16 * rep : 77 + 16 * rep;
        const f4* wl = Wp + (seg == 0 ? s8_loff(1) : s8_loff(5)) + (long)rep * 16 * CF256;
        // what follows: 256-wide chunks, contiguous -- except after layer 7 of a distance-only pass: one output chunk, then the
        // stream restarts (chunks 0, 1 of the blob, requested with two slices per wave like everything this code requests)
        const bool wraps = !FULL && seg == 1 && rep == 2;
        const f4* wrap = Wp - 17L * CF256;           // chunk j >= 17 of that layer is chunk j - 17 of the blob ... in K = 64 steps
        asm volatile("" : "+s"(wl), "+s"(wrap));
        const int rot = cb & 3;
        const unsigned s4[4] = {rot ? slot_b[1] : slot_b[0], rot ? slot_b[2] : slot_b[1], rot ? slot_b[3] : slot_b[2],
                                rot ? slot_b[0] : slot_b[3]};
        run_layer(std::integral_constant<int, 256>{}, std::integral_constant<int, 16>{}, I0{}, std::integral_constant<int, 222>{},
                  [&](int j) { return wraps && j >= 17 ? Wp + (long)(j - 17) * chunk_f4(64) : wl + (long)j * CF256; }, cb, s4);
        (void)wrap;
        y_to_x(8);
      }
      if (seg == 0) {
        {   // layer 3 (chunks 48..60): followed by layer 4 (K = 288: three slices per wave)
          const unsigned s4[4] = {slot_b[0], slot_b[1], slot_b[2], slot_b[3]};
          const f4* w3 = Wp + s8_loff(3);
          const f4* w4 = Wp + s8_loff(4);
          asm volatile("" : "+s"(w3), "+s"(w4));
          run_layer(std::integral_constant<int, 256>{}, std::integral_constant<int, 13>{}, I1{}, std::integral_constant<int, 333>{},
                    [&](int j) { return j < 13 ? w3 + (long)j * CF256 : w4 + (long)(j - 13) * chunk_f4(288); }, 48, s4);
        }
        // layer 4 input = [softplus(layer 3) (208 slots) | features (64 slots) | 16 zero slots] / sqrt 2
        yh[6][2] = shh[0][0]; yl[6][2] = shl[0][0];
        yh[6][3] = shh[0][1]; yl[6][3] = shl[0][1];
        yh[7][0] = shh[0][2]; yl[7][0] = shl[0][2];
        yh[7][1] = shh[0][3]; yl[7][1] = shl[0][3];
        yh[7][2] = shh[1][0]; yl[7][2] = shl[1][0];
        yh[7][3] = shh[1][1]; yl[7][3] = shl[1][1];
        yh[8][0] = shh[1][2]; yl[8][0] = shl[1][2];
        yh[8][1] = shh[1][3]; yl[8][1] = shl[1][3];
        yh[8][2] = 0u; yl[8][2] = 0u;
        yh[8][3] = 0u; yl[8][3] = 0u;
        y_to_x(9);
        {   // layer 4 (chunks 61..76, first slot = slot_b[1]): followed by layer 5
          const unsigned s4[4] = {slot_b[1], slot_b[2], slot_b[3], slot_b[0]};
          const f4* w4 = Wp + s8_loff(4);
          const f4* w5 = Wp + s8_loff(5);
          asm volatile("" : "+s"(w4), "+s"(w5));
          run_layer(std::integral_constant<int, 288>{}, std::integral_constant<int, 16>{}, I0{}, std::integral_constant<int, 222>{},
                    [&](int j) { return j < 16 ? w4 + (long)j * chunk_f4(288) : w5 + (long)(j - 16) * CF256; }, 61, s4);
        }
        y_to_x(8);
        fetch_features(round + gridDim.x);
        __builtin_amdgcn_sched_barrier(0);
      } else {
        // output layer (chunks 125.., first slot = slot_b[1]): followed by the start of the stream (K = 64: one slice per wave;
        // in the distance-only modes chunks 0, 1 were already requested by layer 7)
        const unsigned s4[4] = {slot_b[1], slot_b[2], slot_b[3], slot_b[0]};
        const f4* w8 = Wp + s8_loff(8);
        const f4* w0 = Wp;
        asm volatile("" : "+s"(w8), "+s"(w0));
        run_layer(std::integral_constant<int, 256>{}, std::integral_constant<int, LAST>{}, I2{},
                  std::integral_constant<int, FULL ? 111 : 221>{},
                  [&](int j) { return j < LAST ? w8 + (long)j * CF256 : w0 + (long)(j - LAST) * chunk_f4(64); }, 125, s4);
      }
    }
    {   // the stream continues at slot (NCHUNK & 3) = 2
      const unsigned a = slot_b[0], b = slot_b[1];
      slot_b[0] = slot_b[2];
      slot_b[1] = slot_b[3];
      slot_b[2] = a;
      slot_b[3] = b;
    }
  }
  range_report(sat, range_word);
  s8_wait<0>();
  __syncthreads();
}

// X != nullptr: feature rows X[M,64]; X == nullptr: points xyz[M,3] x in_scale, encoded in the kernel
int launch_sdf_ring8(int mode, const float* X, const float* xyz, float in_scale, long M, const f4* W, float us, float out_scale,
                     float* out0, f4* sig, unsigned grid, hipStream_t s) {
  unsigned* rw = range_flags() ? range_flags() + RB_RANGE_SDF : nullptr;
#define RB_S8(MODE, FUSED) \
  hipLaunchKernelGGL((k_sdf_ring8<MODE, FUSED>), dim3(grid), dim3(512), 0, s, X, xyz, in_scale, M, W, us, out_scale, out0, rw, sig)
  if (X) {
    switch (mode) {
      case 0: RB_S8(0, false); break;
      case 1: RB_S8(1, false); break;
      default: RB_S8(5, false); break;
    }
  } else {
    switch (mode) {
      case 0: RB_S8(0, true); break;
      case 1: RB_S8(1, true); break;
      default: RB_S8(5, true); break;
    }
  }
#undef RB_S8
  return check_launch("k_sdf_ring8");
}

}  // namespace rb
