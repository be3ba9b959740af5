// NeuS SDF network (model/neus_model.py:385-438) with EXACT fp32 operands on the f16 matrix pipe ("f16x6") -- the value pass of the
// default precision policy, round 3.
//
// Under the exact policy the SDF net ran on v_mfma_f32_16x16x4_f32 (k_sdf_mlp, mlp_kernels.hip): 0.69 of a 157 TFLOP/s peak, 49 % of
// BASELINE config 2 at the reference's precision.  This kernel carries every operand as three halves and keeps the six partial products
// of weight >= 2^-22 in three fp32 accumulators by weight class -- the arithmetic of k_dvis_x6 (vis_diffuse_x6.hip: exact operands,
// not narrower than an fp32 fma chain; bound 2500 / 6 = 417 TFLOP/s) -- on the chunk-stream machine of wide_ring.h: persistent
// workgroups of four waves (one 16-row tile each, one wave per SIMD), the nine layers as ONE cyclic stream of 142 / 126 chunks (16
// output neurons x K x 3 pieces: 6 / 24 / 27 KB) through a 4-slot LDS ring filled by LDS-DMA under counted waits, one s_barrier per
// chunk in its middle, fragment reads running across the chunk boundary, the softplus + three-way split of chunk j between the MFMAs
// of chunk j+1, the positional encoding computed in the kernel.
// MODE 0: signed distance only -> out0[M].   MODE 1: all 257 outputs -> out0[M,257].   MODE 5: MODE 1 + sigmoid(100 z) of every hidden
// pre-activation -> sig [tile = row / 16][layer 8][chunk 16][lane 64] float4, the layout k_sdf_back_f32 (mlp_kernels.hip) reads.
// Weights: packing.pack_sdf_x6 (rb_pack_layer_x6, scale 2^0; K padded to 64, 256 x3, 288 = [208 | 64 | 16 zero], 256 x4).
#include "../../include/robir_hip.h"
#include "common.h"
#include "mlp_engine.h"
#include "x6_ring.h"
#include "sdf_x6_layout.h"
#include <cstdlib>
#include <type_traits>

#ifndef SX_PK
#define SX_PK 0      // 1: the softplus stage on value pairs (v_pk_* f32): bit-identical, measured 1 % slower (profiles/r03_sdf_x6_ablation.md)
#endif

namespace rb {

template <int MODE>
__global__ __launch_bounds__(256, 1) void k_sdf_x6(const float* __restrict__ xyz, float in_scale, long M, const f4* __restrict__ Wp,
                                                    float out_scale, float* __restrict__ out0, f4* __restrict__ sig,
                                                    unsigned* __restrict__ range_word) {
  constexpr bool FULL = MODE != 0, STORE = MODE == 5;
  constexpr int LAST = FULL ? 17 : 1;
  __shared__ f4 ring[4 * SX_SLOT_B / 16];              // 110 KB
  __shared__ f4 bias_ring[4 * 16];
  __shared__ float pe_scratch[4 * 16 * 64];            // 16 KB
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long nrounds = (M + 63) >> 6;
  if ((long)blockIdx.x >= nrounds) return;

  const float negk = -2048.0f, inv_sqrt2 = 0.70710678118654752440f;
  constexpr float C11 = 1.0f / 2048.0f;
  const unsigned ring_b = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)ring);
  const unsigned bias_b = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)bias_ring);
  const unsigned lane16 = (unsigned)lane * 16u, lane4 = (unsigned)lane * 4u;
  unsigned slot_b[4] = {0u, (unsigned)SX_SLOT_B, 2u * SX_SLOT_B, 3u * SX_SLOT_B};
  unsigned bslot_b[4] = {0u, 256u, 512u, 768u};
  unsigned sat = 0u;
  u4 xh[9], xm[9], xl[9];              // operands of the current layer (K <= 288): three pieces, one tile
  u4 yh[8], ym[8], yl[8];              // ... of the next layer
  u4 skh[3], skm[3], skl[3];           // skip layer, k-blocks 6 (second half), 7, 8: the net's inputs / sqrt 2, once per round
  long rrow = 0;

  // range sentinel: `sat` in the domain of sat_acc_nonneg (hidden activations are >= 0: one instruction per pair); the signed inputs
  // of a round go through sat_acc into `sat_in`, folded into `sat` behind them
  unsigned sat_in = 0u;
  auto put_pair = [&](float v0, float v1, u4& dh, u4& dm, u4& dl, int q, auto nonneg) {
    unsigned h, m, l;
    sx_split_pair(v0, v1, negk, h, m, l);
    dh[q] = h;
    dm[q] = m;
    dl[q] = l;
    // softplus outputs are >= 0 but NaN passes through them (NaN points of axis-parallel rays): sat_acc_pos, not the one-instruction
    // raw-pattern form of the ReLU kernels, which a positive NaN would trip; signed inputs: sat_acc.  One domain (limit 0x7ffe).
    if constexpr (decltype(nonneg)::value) sat = sat_acc_pos(sat, h);
    else sat = sat_acc(sat, h);
  };
  auto fold_sat_in = [&]() {
    (void)sat_in;          // round 5: inputs and activations share sat_acc's domain, nothing to fold
  };
  auto load_layer0 = [&]() {
    float x0[16];
    load_features_pe10<false>(xyz, in_scale, rrow, M, lane, pe_scratch + wave * 1024, x0);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = (2 * kb + (q >> 1)) * 4 + (q & 1) * 2;
        put_pair(x0[i], x0[i + 1], xh[kb], xm[kb], xl[kb], q, std::false_type{});
      }
    // skip layer operands [softplus(h3) (13 blocks of 16) | x0 (4 blocks) | 0] / sqrt 2: blocks 13..17 = k-block 6 second half .. 8
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      skh[i] = u4{0u, 0u, 0u, 0u};
      skm[i] = u4{0u, 0u, 0u, 0u};
      skl[i] = u4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int b = 13; b < 17; ++b)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int q = (b & 1) * 2 + p, i = (b - 13) * 4 + p * 2;
        put_pair(x0[i] * inv_sqrt2, x0[i + 1] * inv_sqrt2, skh[(b >> 1) - 6], skm[(b >> 1) - 6], skl[(b >> 1) - 6], q, std::false_type{});
      }
    fold_sat_in();
  };

  auto run_layer = [&](auto LI_tag, int cb, int lrt) {
    constexpr int LI = decltype(LI_tag)::value;
    constexpr int K = sx_K(LI), KB = K / 32, NCH = sx_nch(LI, LAST), NP = sx_np(K), CB = sx_cbase(LI, LAST);
    constexpr bool OUT = LI == 8, SKIPOUT = LI == 3;
    constexpr int BS = KB >= 8 ? 2 : 1, DB = 1, D = BS * DB, NB = BS * (DB + 1);     // six MFMAs per k-block: two k-blocks ahead cover the LDS latency
    constexpr int HB = KB / 2, NSTEP = NCH * KB;
    static_assert(D + BS - 1 <= KB - HB, "reads of the next chunk start after the barrier");
    SxAcc accs[2];
    f4 bnext = f4{0.f, 0.f, 0.f, 0.f};
    u4 wfh[NB], wfm[NB], wfl[NB];
    const f4* wl = Wp + sx_coff(cb, LAST);
    const f4* wnext[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) wnext[i] = Wp + sx_coff(cb + NCH + i, LAST);
    asm volatile("" : "+s"(wl));
    auto frag_of = [&](int c) { return reinterpret_cast<const u4*>(reinterpret_cast<const char*>(ring) + slot_b[c & 3]) + lane; };
    auto bias_of = [&](int c) { return *(reinterpret_cast<const f4*>(reinterpret_cast<const char*>(bias_ring) + bslot_b[c & 3]) + g); };
    auto zero_acc = [&](SxAcc& a, const f4& b) {
      a.c0 = b;
      a.c1 = f4{0.f, 0.f, 0.f, 0.f};
      a.c2 = f4{0.f, 0.f, 0.f, 0.f};
    };
    auto combine = [&](const SxAcc& a, int r) { return __builtin_fmaf(__builtin_fmaf(a.c2[r], C11, a.c1[r]), C11, a.c0[r]); };
    // hidden chunk pj: softplus (+ its sigmoid in MODE 5), three-way split into the next layer's operand registers
    // hidden chunk pj in four stages (they go between the MFMA runs of the next chunk, one per run): A = softplus (+ its sigmoid in
    // MODE 5) of a value pair, B = exact three-way split into the next layer's operand registers
    float ev0[2], ev1[2];
#if SX_PK
    // the same arithmetic on value PAIRS: v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 round like their scalar forms (bit-identical),
    // the transcendentals and the selects stay per value
    auto stage_a = [&](const SxAcc& a, int q, f4& sg) {
      const f2 c0 = q ? f2{a.c0[2], a.c0[3]} : f2{a.c0[0], a.c0[1]};
      const f2 c1 = q ? f2{a.c1[2], a.c1[3]} : f2{a.c1[0], a.c1[1]};
      const f2 c2 = q ? f2{a.c2[2], a.c2[3]} : f2{a.c2[0], a.c2[1]};
      const f2 k11 = f2{C11, C11};
      const f2 z = __builtin_elementwise_fma(__builtin_elementwise_fma(c2, k11, c1), k11, c0);
      // softplus100_stable (mlp_engine.h) on a value pair
      const f2 t = f2{-__builtin_fabsf(z[0]), -__builtin_fabsf(z[1])} * SP_T_PER_Z;
      const f2 e = f2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
      const f2 u = e + 1.0f;
      const f2 lg = f2{__builtin_amdgcn_logf(u[0]), __builtin_amdgcn_logf(u[1])};
      if constexpr (STORE) {
        const f2 r = f2{__builtin_amdgcn_rcpf(u[0]), __builtin_amdgcn_rcpf(u[1])};
        sg[2 * q] = (z[0] > 0.0f ? 1.0f : e[0]) * r[0];
        sg[2 * q + 1] = (z[1] > 0.0f ? 1.0f : e[1]) * r[1];
      }
      f2 v = __builtin_elementwise_fma(lg, f2{SP_LN2_OVER_100, SP_LN2_OVER_100}, f2{__builtin_fmaxf(z[0], 0.0f), __builtin_fmaxf(z[1], 0.0f)});
      if (SKIPOUT) v = v * inv_sqrt2;
      ev0[q] = v[0];
      ev1[q] = v[1];
    };
#else
    auto stage_a = [&](const SxAcc& a, int q, f4& sg) {
      float s0, s1;
      // the overflow-free form (mlp_engine.h: max(z, 0) + a correction <= 0.0069 from the hardware exp2 / log2)
      float v0 = softplus100_stable(combine(a, 2 * q), &s0), v1 = softplus100_stable(combine(a, 2 * q + 1), &s1);
      if (SKIPOUT) {
        v0 *= inv_sqrt2;
        v1 *= inv_sqrt2;
      }
      sg[2 * q] = s0;
      sg[2 * q + 1] = s1;
      ev0[q] = v0;
      ev1[q] = v1;
    };
#endif
    auto stage_b = [&](int pj, int q) { put_pair(ev0[q], ev1[q], yh[pj >> 1], ym[pj >> 1], yl[pj >> 1], (pj & 1) * 2 + q, std::true_type{}); };
    auto store_sig = [&](int pj, const f4& sg) {
      if constexpr (STORE) sig[((rrow >> 4) * 8 + lrt) * (16L * 64) + pj * 64 + lane] = sg;
    };
    auto output_chunk = [&](const SxAcc& a, int pj) {
      if (rrow >= M) return;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = pj * 16 + 4 * g + r;
        if constexpr (FULL) {
          if (j < 257) out0[rrow * 257 + j] = combine(a, r) * out_scale;
        } else {
          if (j == 0) out0[rrow] = combine(a, r) * out_scale;
        }
      }
    };
    zero_acc(accs[0], bias_of(0));
#pragma unroll
#ifdef SXA_NOREAD
    for (int i = 0; i < NB; ++i)
#else
    for (int i = 0; i < D; ++i)
#endif
      if (i < NSTEP) {
        const u4* f = frag_of(i / KB) + (3 * (i % KB)) * 64;
        wfh[i % NB] = f[0];
        wfm[i % NB] = f[64];
        wfl[i % NB] = f[128];
      }
    f4 sgp = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jb = 0; jb < NCH; ++jb) {
      SxAcc& acc = accs[jb & 1];
      if (jb > 0) zero_acc(acc, bnext);
      constexpr int dummy2 = 0;
      (void)dummy2;
      const int K3 = jb + 3 < NCH ? K : sx_K(sx_layer_of(CB + jb + 3, LAST));
      const int nu3 = sx_units(K3);
      const f4* src3 = jb + 3 < NCH ? wl + (long)(jb + 3) * sx_cf4(K) : wnext[jb + 3 - NCH < 3 ? jb + 3 - NCH : 0];
      const int sl3 = (jb + 3) & 3;
      const unsigned dst3 = ring_b + slot_b[sl3], bdst3 = bias_b + bslot_b[sl3];
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const int st = jb * KB + kb;
        if (kb == HB) {   // chunk jb+1 must have landed: this wave's copies of chunk jb+2 may still be in flight
          const int allowed = jb + 2 < NCH ? NP : sx_np(sx_K(sx_layer_of(CB + jb + 2, LAST)));
          if (allowed >= 8) sx_wait<8>();
          else if (allowed >= 7) sx_wait<7>();
          else sx_wait<3>();
#ifndef SXA_NOBAR
          __builtin_amdgcn_s_barrier();
#endif
          asm volatile("" ::: "memory");
          bnext = bias_of(jb + 1);
        }
#ifdef SXA_NOREAD                     // no fragment reads: the registers are redefined behind the compiler's back (no instruction)
        if (st % BS == 0) {
#pragma unroll
          for (int i = BS - 1; i >= 0; --i) {
            const int s2 = st + D + i;
            if (s2 < NSTEP) asm volatile("" : "+v"(wfl[s2 % NB]), "+v"(wfm[s2 % NB]), "+v"(wfh[s2 % NB]));
          }
        }
#else
        if (st % BS == 0) {
#pragma unroll
          for (int i = BS - 1; i >= 0; --i) {
            const int s2 = st + D + i;
            if (s2 < NSTEP) {
              const u4* f = frag_of(s2 / KB) + (3 * (s2 % KB)) * 64;
              wfl[s2 % NB] = f[128];
              wfm[s2 % NB] = f[64];
              wfh[s2 % NB] = f[0];
            }
          }
        }
#endif
        // the six products of BS k-blocks, a run per accumulator: class 2 (wl.xh, wm.xm, wh.xl), class 1 (wm.xh, wh.xm), class 0 (wh.xh)
        if (st % BS == BS - 1 || kb == KB - 1) {
          const int k0 = (st % BS == BS - 1) ? (kb - (BS - 1) > 0 ? kb - (BS - 1) : 0) : kb - (st % BS);
#define SX_MFMA(ACC, W, X) ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, W), __builtin_bit_cast(h8, X), ACC, 0, 0, 0)
#pragma unroll
          for (int k = k0; k <= kb; ++k) SX_MFMA(acc.c2, wfl[(jb * KB + k) % NB], xh[k]);
#pragma unroll
          for (int k = k0; k <= kb; ++k) SX_MFMA(acc.c2, wfm[(jb * KB + k) % NB], xm[k]);
#pragma unroll
          for (int k = k0; k <= kb; ++k) SX_MFMA(acc.c2, wfh[(jb * KB + k) % NB], xl[k]);
#pragma unroll
          for (int k = k0; k <= kb; ++k) SX_MFMA(acc.c1, wfm[(jb * KB + k) % NB], xh[k]);
#pragma unroll
          for (int k = k0; k <= kb; ++k) SX_MFMA(acc.c1, wfh[(jb * KB + k) % NB], xm[k]);
#pragma unroll
          for (int k = k0; k <= kb; ++k) SX_MFMA(acc.c0, wfh[(jb * KB + k) % NB], xh[k]);
#undef SX_MFMA
        }
#ifdef SXA_NOEPI                      // timing ablations (wrong results): -DSXA_NOEPI / _NOREAD / _NODMA / _NOBAR
        if (jb > 0 && kb == KB - 1) {         // the data dependence of the epilogue without its arithmetic
          const SxAcc& pa = accs[(jb - 1) & 1];
          const int pj = jb - 1;
          if (OUT) output_chunk(pa, pj);
          else
            for (int q = 0; q < 2; ++q) {
              const unsigned hv = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(pa.c0[2 * q] + pa.c1[2 * q], pa.c0[2 * q + 1] + pa.c2[2 * q + 1]));
              yh[pj >> 1][(pj & 1) * 2 + q] = hv;
              ym[pj >> 1][(pj & 1) * 2 + q] = hv;
              yl[pj >> 1][(pj & 1) * 2 + q] = hv;
            }
        }
#else
        if (jb > 0 && (st % BS == BS - 1 || kb == KB - 1)) {      // epilogue of chunk jb-1: a stage beside every MFMA run
          constexpr int dummy3 = 0;
          (void)dummy3;
          int nm = 0, mi = 0;                    // MFMA-issuing k-blocks of this chunk; index of this one
          for (int k = 0; k < KB; ++k) {
            const bool issues = ((jb * KB + k) % BS == BS - 1) || k == KB - 1;
            if (issues && k < kb) ++mi;
            if (issues) ++nm;
          }
#pragma unroll
          for (int sgi = 0; sgi < 4; ++sgi)
            if ((sgi * nm) / 4 == mi) {
              const SxAcc& pa = accs[(jb - 1) & 1];
              if (OUT) {
                if (sgi == 0) output_chunk(pa, jb - 1);
              } else if (sgi == 0) stage_a(pa, 0, sgp);
              else if (sgi == 1) stage_b(jb - 1, 0);
              else if (sgi == 2) stage_a(pa, 1, sgp);
              else {
                stage_b(jb - 1, 1);
                store_sig(jb - 1, sgp);
              }
            }
        }
#endif
#ifndef SXA_NODMA
        if (kb >= HB) {
#pragma unroll
          for (int u = 0; u < 3; ++u)
            if (u < nu3 && (u * (KB - HB)) / nu3 == kb - HB) {
              if (K3 == 64) sx_copy_unit<64>(u, src3, lane4, lane16, bdst3, dst3, wave);
              else if (K3 == 256) sx_copy_unit<256>(u, src3, lane4, lane16, bdst3, dst3, wave);
              else sx_copy_unit<288>(u, src3, lane4, lane16, bdst3, dst3, wave);
            }
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    {   // slot 0 = the slot of the next layer's first chunk
      constexpr int R = NCH & 3;
      unsigned a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[i] = slot_b[(i + R) & 3];
        b[i] = bslot_b[(i + R) & 3];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        slot_b[i] = a[i];
        bslot_b[i] = b[i];
      }
    }
    const SxAcc& last = accs[(NCH - 1) & 1];
    if constexpr (OUT) {
      output_chunk(last, NCH - 1);
    } else {
#ifndef SXA_NOEPI
      stage_a(last, 0, sgp);
      stage_b(NCH - 1, 0);
      stage_a(last, 1, sgp);
      stage_b(NCH - 1, 1);
      store_sig(NCH - 1, sgp);
#else
      for (int q = 0; q < 2; ++q) {
        const unsigned hv = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(last.c0[2 * q] + last.c1[2 * q], last.c0[2 * q + 1] + last.c2[2 * q + 1]));
        yh[(NCH - 1) >> 1][((NCH - 1) & 1) * 2 + q] = hv;
        ym[(NCH - 1) >> 1][((NCH - 1) & 1) * 2 + q] = hv;
        yl[(NCH - 1) >> 1][((NCH - 1) & 1) * 2 + q] = hv;
      }
#endif
      if constexpr (SKIPOUT) {
#pragma unroll
        for (int kb = 0; kb < 9; ++kb) {
          if (kb < 6) {
            xh[kb] = yh[kb < 6 ? kb : 0];
            xm[kb] = ym[kb < 6 ? kb : 0];
            xl[kb] = yl[kb < 6 ? kb : 0];
          } else if (kb == 6) {
            xh[kb] = u4{yh[6][0], yh[6][1], skh[0][2], skh[0][3]};
            xm[kb] = u4{ym[6][0], ym[6][1], skm[0][2], skm[0][3]};
            xl[kb] = u4{yl[6][0], yl[6][1], skl[0][2], skl[0][3]};
          } else {
            xh[kb] = skh[kb > 6 ? kb - 6 : 0];
            xm[kb] = skm[kb > 6 ? kb - 6 : 0];
            xl[kb] = skl[kb > 6 ? kb - 6 : 0];
          }
        }
      } else {
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
          xh[kb] = yh[kb];
          xm[kb] = ym[kb];
          xl[kb] = yl[kb];
        }
      }
    }
  };

  // ---- prologue: chunks 0, 1, 2 of the stream (layer 0: K = 64)
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int u = 0; u < 2; ++u)
      sx_copy_unit<64>(u, Wp + sx_coff(c, LAST), lane4, lane16, bias_b + bslot_b[c], ring_b + slot_b[c], wave);
  sx_wait<0>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  for (long round = blockIdx.x; round < nrounds; round += gridDim.x) {
    rrow = round * 64 + wave * 16 + (lane & 15);
    load_layer0();
    // layer 0 | 1, 2 (one instance) | 3 (skip layer's own outputs) | 4 (K = 288) | 5, 6 (the instance of 1, 2) | 7 | 8
#pragma unroll 1
    for (int l = 0; l < 9; ++l) {
      const int cb = l < 4 ? 16 * l : 16 * (l - 1) + 13;
      if (l == 0) run_layer(std::integral_constant<int, 0>{}, 0, 0);
      else if (l == 3) run_layer(std::integral_constant<int, 3>{}, cb, 3);
      else if (l == 4) run_layer(std::integral_constant<int, 4>{}, cb, 4);
      else if (l == 7) run_layer(std::integral_constant<int, 7>{}, cb, 7);
      else if (l == 8) run_layer(std::integral_constant<int, 8>{}, cb, 8);
      else run_layer(std::integral_constant<int, 1>{}, cb, l);
    }
  }
  range_report<false>(sat, range_word);
  sx_wait<0>();
  __syncthreads();
}

}  // namespace rb

using namespace rb;

namespace {
int sx_grid(long M, int n_workgroups) { return persistent_grid((M + 63) / 64, n_workgroups); }
int sx_launch(const float* x, long M, float in_scale, const float* Wp, int mode, float out_scale, float* out0, float* sig, int n_workgroups,
              hipStream_t s) {
  const int grid = sx_grid(M, n_workgroups);
  if (grid <= 0) return rb::fail("rb_sdf_x6_points", "device query failed");
  unsigned* rw = range_flags() ? range_flags() + RB_RANGE_SDF : nullptr;
  const f4* W = (const f4*)Wp;
  switch (mode) {
    case 0: hipLaunchKernelGGL(k_sdf_x6<0>, dim3(grid), dim3(256), 0, s, x, in_scale, M, W, out_scale, out0, (f4*)nullptr, rw); break;
    case 1: hipLaunchKernelGGL(k_sdf_x6<1>, dim3(grid), dim3(256), 0, s, x, in_scale, M, W, out_scale, out0, (f4*)nullptr, rw); break;
    default: hipLaunchKernelGGL(k_sdf_x6<5>, dim3(grid), dim3(256), 0, s, x, in_scale, M, W, out_scale, out0, (f4*)sig, rw); break;
  }
  return check_launch("k_sdf_x6");
}
}  // namespace

extern "C" {

int rb_sdf_x6_points(const float* x, long M, float in_scale, const float* Wp, int mode, float out_scale, float* out0, int two_tile,
                     int n_workgroups, rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(x && Wp && out0, "null pointer");
  RB_REQUIRE(mode == 0 || mode == 1, "mode: 0 signed distance only (blob packed with full = 0), 1 all 257 outputs");
  if (two_tile)      // two 16-row tiles per wave, rounds of 128 rows (sdf_x6t.hip)
    return rb::launch_sdf_x6t(x, M, in_scale, Wp, mode, out_scale, out0, nullptr, n_workgroups, (hipStream_t)stream);
  return sx_launch(x, M, in_scale, Wp, mode, out_scale, out0, nullptr, n_workgroups, (hipStream_t)stream);
}

}  // extern "C"

// value + reverse-mode gradient with the value pass on exact operands: k_sdf_x6<5>, k_sdf_back_f32, k_pe_grad_points (sdf_back.hip)
namespace rb {
int launch_sdf_x6_store(const float* x, long M, float in_scale, const float* Wp, float out_scale, float* out0, float* sig, hipStream_t s) {
  return sx_launch(x, M, in_scale, Wp, 5, out_scale, out0, sig, 0, s);
}
}  // namespace rb
