// Visibility MLP (VisNetwork.forward, model/implicit_differentiable_renderer.py:241-258: [PE10(p) | PE10(d)] -> 256 x4 ReLU -> 2
// logits) with EXACT fp32 operands on the f16 matrix pipe ("f16x6") -- the default precision policy's kernel for the stand-alone
// visibility evaluations (specular visibility, trace_radiance), round 3.  The machine of color_x6.hip / sdf_x6.hip: four waves, one
// 16-row tile each, three operand pieces, six MFMA products per multiply-add in three accumulators by weight class, the net as one cyclic
// stream of 65 chunks (K = 128 for the first layer, 256 after) through a 4-slot LDS ring, both encodings computed in the kernel
// (load_features_vis: row i takes point i / rep).  Replaces k_vis_mlp (f32-input MFMA).  Weights: packing.pack_vis_x6.
#include "../../include/robir_hip.h"
#include "common.h"
#include "mlp_engine.h"
#include "x6_ring.h"
#include <type_traits>

#ifndef VX_FP8
#define VX_FP8 0            // 1: EXPERIMENT: h.xl and l.xh as bf8 products on v_mfma_f32_16x16x128_f8f6f4 (weights: packing.repack_vis_x6_fp8)
#endif

namespace rb {

constexpr int VX_SLOT_B = 24 * 1024 + 512;
__host__ __device__ constexpr int vx_K(int l) { return l == 0 ? 128 : 256; }
__host__ __device__ constexpr int vx_nch(int l) { return l == 4 ? 1 : 16; }
__host__ __device__ constexpr int vx_layer_of(int c) {      // stream position (cyclic: 65 chunks) -> layer
  if (c >= 65) c -= 65;
  return c >> 4;
}
__host__ __device__ constexpr long vx_coff(int c) {
  if (c >= 65) c -= 65;
  return c < 16 ? (long)c * sx_cf4(128) : 16 * sx_cf4(128) + (long)(c - 16) * sx_cf4(256);
}

__global__ __launch_bounds__(256, 1) void k_vis_x6(const float* __restrict__ P, const float* __restrict__ Dr, int rep, long M,
                                                    const f4* __restrict__ Wp, float* __restrict__ logits,
                                                    unsigned* __restrict__ range_word) {
  __shared__ f4 ring[4 * VX_SLOT_B / 16];              // 98 KB
  __shared__ f4 bias_ring[4 * 16];
  __shared__ float pe_scratch[4 * 16 * 128];           // 32 KB: the encoder's exchange rows (mlp_engine.h)
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long nrounds = (M + 63) >> 6;
  if ((long)blockIdx.x >= nrounds) return;

  const float negk = -2048.0f;
  constexpr float C11 = 1.0f / 2048.0f;
  const unsigned ring_b = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)ring);
  const unsigned bias_b = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)bias_ring);
  const unsigned lane16 = (unsigned)lane * 16u, lane4 = (unsigned)lane * 4u;
  unsigned slot_b[4] = {0u, (unsigned)VX_SLOT_B, 2u * VX_SLOT_B, 3u * VX_SLOT_B};
  unsigned bslot_b[4] = {0u, 256u, 512u, 768u};
  unsigned sat = 0u;
  u4 xh[8], xm[8], xl[8];              // operands of the current layer (K <= 256): three pieces, one tile
  u4 yh[8], ym[8], yl[8];              // ... of the next layer
#if VX_FP8
  typedef int vx_i8 __attribute__((ext_vector_type(8)));
  vx_i8 xh8[2], xl8[2], yh8[2], yl8[2];      // bf8 (e5m2) copies of the h and l pieces, 32 K values per lane and group of 128 (xl / yl stay unused)
  unsigned l_keep = 0u;
#endif
  long rrow = 0;

#if VX_FP8
  // pair q (0..3) of k-block kb: the f16 h and m pieces as before; behind the second pair of a 16-value block (q odd) the top bytes of the
  // block's four h and four l halves become dword 2 kb + q / 2 of the group's bf8 operands (truncation to e5m2: one v_perm_b32 each)
  auto put_pair8 = [&](float v0, float v1, u4& dh, u4& dm, vx_i8 (&d8h)[2], vx_i8 (&d8l)[2], int kb, int q) {
    unsigned h, m, l;
    sx_split_pair(v0, v1, negk, h, m, l);
    dh[q] = h;
    dm[q] = m;
    if (q & 1) {
      const int j8 = 2 * kb + (q >> 1);
      d8h[j8 >> 3][j8 & 7] = (int)__builtin_amdgcn_perm(h, dh[q - 1], 0x07050301u);
      d8l[j8 >> 3][j8 & 7] = (int)__builtin_amdgcn_perm(l, l_keep, 0x07050301u);
    } else {
      l_keep = l;
    }
    sat = sat_acc(sat, h);
  };
#endif
  auto put_pair = [&](float v0, float v1, u4& dh, u4& dm, u4& dl, int q) {
    unsigned h, m, l;
    sx_split_pair(v0, v1, negk, h, m, l);
    dh[q] = h;
    dm[q] = m;
    dl[q] = l;
    sat = sat_acc(sat, h);
  };
  // this round's rows -> operands of layer 0: both encodings by the four lane groups of a row
  auto load_layer0 = [&]() {
    float in0[32];
    load_features_vis(P, Dr, rep, rrow, M, lane, pe_scratch + wave * 2048, in0);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = (2 * kb + (q >> 1)) * 4 + (q & 1) * 2;
#if VX_FP8
        put_pair8(in0[i], in0[i + 1], xh[kb], xm[kb], xh8, xl8, kb, q);
#else
        put_pair(in0[i], in0[i + 1], xh[kb], xm[kb], xl[kb], q);
#endif
      }
  };

  auto run_layer = [&](auto LI_tag, int cb) {
    constexpr int LI = decltype(LI_tag)::value;
    constexpr int K = vx_K(LI), KB = K / 32, NCH = vx_nch(LI), NP = sx_np(K), CB = 16 * LI;
    constexpr bool OUT = LI == 4;
    constexpr int BS = KB >= 8 ? 2 : 1, DB = KB >= 8 ? 1 : 2, D = BS * DB, NB = BS * (DB + 1);
    constexpr int HB = KB / 2, NSTEP = NCH * KB;
    static_assert(D + BS - 1 <= KB - HB, "reads of the next chunk start after the barrier");
    SxAcc accs[2];
    f4 bnext = f4{0.f, 0.f, 0.f, 0.f};
    u4 wfh[NB], wfm[NB], wfl[NB];
    const f4* wl = Wp + vx_coff(cb);
    const f4* wnext[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) wnext[i] = Wp + vx_coff(cb + NCH + i);
    asm volatile("" : "+s"(wl));
    auto frag_of = [&](int c) { return reinterpret_cast<const u4*>(reinterpret_cast<const char*>(ring) + slot_b[c & 3]) + lane; };
    auto bias_of = [&](int c) { return *(reinterpret_cast<const f4*>(reinterpret_cast<const char*>(bias_ring) + bslot_b[c & 3]) + g); };
    auto zero_acc = [&](SxAcc& a, const f4& b) {
      a.c0 = b;
      a.c1 = f4{0.f, 0.f, 0.f, 0.f};
      a.c2 = f4{0.f, 0.f, 0.f, 0.f};
    };
    auto combine = [&](const SxAcc& a, int r) { return __builtin_fmaf(__builtin_fmaf(a.c2[r], C11, a.c1[r]), C11, a.c0[r]); };
    auto hidden_pair = [&](const SxAcc& a, int pj, int q) {
#if VX_FP8
      put_pair8(fmaxf(combine(a, 2 * q), 0.f), fmaxf(combine(a, 2 * q + 1), 0.f), yh[pj >> 1], ym[pj >> 1], yh8, yl8, pj >> 1, (pj & 1) * 2 + q);
#else
      put_pair(fmaxf(combine(a, 2 * q), 0.f), fmaxf(combine(a, 2 * q + 1), 0.f), yh[pj >> 1], ym[pj >> 1], yl[pj >> 1], (pj & 1) * 2 + q);
#endif
    };
    zero_acc(accs[0], bias_of(0));
#if VX_FP8
    vx_i8 w8h, w8l;                    // the group's bf8 weight fragments (requested at its first k-block, used behind its last)
    // k-block kb of a chunk: group kb / 4 (12 KB = 768 lane-strided u4), its h plane at 2 (kb % 4), m behind it; h8 / l8 at 512 / 640
    auto frag16 = [&](int c, int kb) { return frag_of(c) + (kb >> 2) * 768 + (2 * (kb & 3)) * 64; };
    auto frag8 = [&](int c, int grp) { return frag_of(c) + grp * 768 + 512; };
#endif
#pragma unroll
    for (int i = 0; i < D; ++i)
      if (i < NSTEP) {
#if VX_FP8
        const u4* f = frag16(i / KB, i % KB);
        wfh[i % NB] = f[0];
        wfm[i % NB] = f[64];
#else
        const u4* f = frag_of(i / KB) + (3 * (i % KB)) * 64;
        wfh[i % NB] = f[0];
        wfm[i % NB] = f[64];
        wfl[i % NB] = f[128];
#endif
      }
#pragma unroll
    for (int jb = 0; jb < NCH; ++jb) {
      SxAcc& acc = accs[jb & 1];
      if (jb > 0) zero_acc(acc, bnext);
      constexpr int dummy2 = 0;
      (void)dummy2;
      const int K3 = jb + 3 < NCH ? K : vx_K(vx_layer_of(CB + jb + 3));
      const int nu3 = sx_units(K3);
      const f4* src3 = jb + 3 < NCH ? wl + (long)(jb + 3) * sx_cf4(K) : wnext[jb + 3 - NCH < 3 ? jb + 3 - NCH : 0];
      const int sl3 = (jb + 3) & 3;
      const unsigned dst3 = ring_b + slot_b[sl3], bdst3 = bias_b + bslot_b[sl3];
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const int st = jb * KB + kb;
        if (kb == HB) {   // chunk jb+1 must have landed: this wave's copies of chunk jb+2 may still be in flight
          const int allowed = jb + 2 < NCH ? NP : sx_np(vx_K(vx_layer_of(CB + jb + 2)));
          if (allowed >= 7) sx_wait<7>();
          else sx_wait<4>();
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          bnext = bias_of(jb + 1);
        }
        if (st % BS == 0) {
#pragma unroll
          for (int i = BS - 1; i >= 0; --i) {
            const int s2 = st + D + i;
            if (s2 < NSTEP) {
#if VX_FP8
              const u4* f = frag16(s2 / KB, s2 % KB);
              wfm[s2 % NB] = f[64];
              wfh[s2 % NB] = f[0];
#else
              const u4* f = frag_of(s2 / KB) + (3 * (s2 % KB)) * 64;
              wfl[s2 % NB] = f[128];
              wfm[s2 % NB] = f[64];
              wfh[s2 % NB] = f[0];
#endif
            }
          }
        }
        if (st % BS == BS - 1 || kb == KB - 1) {
          const int k0 = (st % BS == BS - 1) ? (kb - (BS - 1) > 0 ? kb - (BS - 1) : 0) : kb - (st % BS);
#define VX_MFMA(ACC, W, X) ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, W), __builtin_bit_cast(h8, X), ACC, 0, 0, 0)
#if !VX_FP8
#pragma unroll
          for (int k = k0; k <= kb; ++k) VX_MFMA(acc.c2, wfl[(jb * KB + k) % NB], xh[k]);
#endif
#pragma unroll
          for (int k = k0; k <= kb; ++k) VX_MFMA(acc.c2, wfm[(jb * KB + k) % NB], xm[k]);
#if !VX_FP8
#pragma unroll
          for (int k = k0; k <= kb; ++k) VX_MFMA(acc.c2, wfh[(jb * KB + k) % NB], xl[k]);
#endif
#pragma unroll
          for (int k = k0; k <= kb; ++k) VX_MFMA(acc.c1, wfm[(jb * KB + k) % NB], xh[k]);
#pragma unroll
          for (int k = k0; k <= kb; ++k) VX_MFMA(acc.c1, wfh[(jb * KB + k) % NB], xm[k]);
#pragma unroll
          for (int k = k0; k <= kb; ++k) VX_MFMA(acc.c0, wfh[(jb * KB + k) % NB], xh[k]);
#undef VX_MFMA
        }
#if VX_FP8
        if ((kb & 3) == 0) {                   // the group's two bf8 fragments: four reads, used three k-blocks later
          const u4* f8 = frag8(jb, kb >> 2);
          const u4 a0 = f8[0], a1 = f8[64], b0 = f8[128], b1 = f8[192];
          w8h = vx_i8{(int)a0[0], (int)a0[1], (int)a0[2], (int)a0[3], (int)a1[0], (int)a1[1], (int)a1[2], (int)a1[3]};
          w8l = vx_i8{(int)b0[0], (int)b0[1], (int)b0[2], (int)b0[3], (int)b1[0], (int)b1[1], (int)b1[2], (int)b1[3]};
        }
        if ((kb & 3) == 3) {                   // c2 += l.xh + h.xl over the group's 128 K: two MFMAs at twice the f16 rate
          acc.c2 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w8l, xh8[kb >> 2], acc.c2, 1, 1, 0, 0, 0, 0);
          acc.c2 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w8h, xl8[kb >> 2], acc.c2, 1, 1, 0, 0, 0, 0);
        }
#endif
        if (jb > 0 && !OUT) {                  // relu + three-way split of chunk jb-1
          if (kb == 0) hidden_pair(accs[(jb - 1) & 1], jb - 1, 0);
          if (kb == 3) hidden_pair(accs[(jb - 1) & 1], jb - 1, 1);
        }
        if (kb >= HB) {
#pragma unroll
          for (int u = 0; u < 3; ++u)
            if (u < nu3 && (u * (KB - HB)) / nu3 == kb - HB) {
              if (K3 == 128) sx_copy_unit<128>(u, src3, lane4, lane16, bdst3, dst3, wave);
              else sx_copy_unit<256>(u, src3, lane4, lane16, bdst3, dst3, wave);
            }
        }
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
      if (g == 0 && rrow < M) {
        logits[rrow * 2] = combine(last, 0);
        logits[rrow * 2 + 1] = combine(last, 1);
      }
    } else {
      hidden_pair(last, NCH - 1, 0);
      hidden_pair(last, NCH - 1, 1);
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) {
        xh[kb] = yh[kb];
        xm[kb] = ym[kb];
#if !VX_FP8
        xl[kb] = yl[kb];
#endif
      }
#if VX_FP8
      xh8[0] = yh8[0];
      xh8[1] = yh8[1];
      xl8[0] = yl8[0];
      xl8[1] = yl8[1];
#endif
    }
  };

  // ---- prologue: chunks 0, 1, 2 of the stream (layer 0: K = 128)
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int u = 0; u < 2; ++u)
      sx_copy_unit<128>(u, Wp + vx_coff(c), lane4, lane16, bias_b + bslot_b[c], ring_b + slot_b[c], wave);
  sx_wait<0>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  for (long round = blockIdx.x; round < nrounds; round += gridDim.x) {
    rrow = round * 64 + wave * 16 + (lane & 15);
    load_layer0();
    // layer 0 (K = 128) | 1, 2 (one instance) | 3 (followed by the output chunk and the next round's first chunks) | 4
#pragma unroll 1
    for (int l = 0; l < 5; ++l) {
      if (l == 0) run_layer(std::integral_constant<int, 0>{}, 0);
      else if (l == 3) run_layer(std::integral_constant<int, 3>{}, 48);
      else if (l == 4) run_layer(std::integral_constant<int, 4>{}, 64);
      else run_layer(std::integral_constant<int, 1>{}, 16 * l);
    }
  }
  range_report(sat, range_word);
  sx_wait<0>();
  __syncthreads();
}

}  // namespace rb

using namespace rb;

extern "C" int rb_vis_x6_points(const float* p, const float* d, long M, int rep, const float* Wp, float* logits, int n_workgroups,
                               rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(p && d && Wp && logits, "null pointer");
  RB_REQUIRE(rep >= 1, "rep must be >= 1");
  const int pg = persistent_grid((M + 63) / 64, n_workgroups);
  if (pg <= 0) return rb::fail(__func__, "device query failed");
  const unsigned grid = (unsigned)pg;
  hipLaunchKernelGGL(k_vis_x6, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, d, rep, M, (const f4*)Wp, logits,
                     range_flags() ? range_flags() + RB_RANGE_VIS : nullptr);
  return check_launch("k_vis_x6");
}
