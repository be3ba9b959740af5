// NeuS colour network (RenderingNetwork.forward, model/neus_model.py:535-560) with EXACT fp32 operands on the f16 matrix pipe
// ("f16x6") -- the default precision policy's colour kernel, round 3.  The machine of sdf_x6.hip (four waves, one 16-row tile each,
// three operand pieces, six MFMA products per multiply-add in three accumulators by weight class, the net as one cyclic stream of 65
// chunks -- K = 320 for the first layer, 256 after -- through a 4-slot LDS ring of 30 KB slots) with k_color_ring8's inputs: the 256
// feature columns read where the SDF network wrote them, [x | PE4(view) | normal] encoded in the kernel.  Replaces k_color_mlp
// (f32-input MFMA: 134 ms of BASELINE config 2's 584 at the reference's precision).  Weights: packing.pack_color_x6.
#include "../../include/robir_hip.h"
#include "common.h"
#include "mlp_engine.h"
#include "x6_ring.h"
#include <type_traits>

namespace rb {

constexpr int CX_SLOT_B = 30 * 1024 + 512;
__host__ __device__ constexpr int cx_K(int l) { return l == 0 ? 320 : 256; }
__host__ __device__ constexpr int cx_nch(int l) { return l == 4 ? 1 : 16; }
__host__ __device__ constexpr int cx_layer_of(int c) {      // stream position (cyclic: 65 chunks) -> layer
  if (c >= 65) c -= 65;
  return c >> 4;
}
__host__ __device__ constexpr long cx_coff(int c) {
  if (c >= 65) c -= 65;
  return c < 16 ? (long)c * sx_cf4(320) : 16 * sx_cf4(320) + (long)(c - 16) * sx_cf4(256);
}
typedef float f4u8x __attribute__((ext_vector_type(4), aligned(4)));

__global__ __launch_bounds__(256, 1) void k_color_x6(const float* __restrict__ feat, long feat_stride, float feat_scale,
                                                      const float* __restrict__ pxyz, float x_scale, const float* __restrict__ pview,
                                                      const float* __restrict__ pnormal, long M, const f4* __restrict__ Wp,
                                                      float* __restrict__ rgb, unsigned* __restrict__ range_word) {
  __shared__ f4 ring[4 * CX_SLOT_B / 16];              // 122 KB
  __shared__ f4 bias_ring[4 * 16];
  __shared__ float tail_lds[4 * 16 * 48];              // 12 KB: the encoded tail of the round's rows
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long nrounds = (M + 63) >> 6;
  if ((long)blockIdx.x >= nrounds) return;

  const float negk = -2048.0f;
  constexpr float C11 = 1.0f / 2048.0f;
  const unsigned ring_b = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)ring);
  const unsigned bias_b = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)bias_ring);
  const unsigned lane16 = (unsigned)lane * 16u, lane4 = (unsigned)lane * 4u;
  unsigned slot_b[4] = {0u, (unsigned)CX_SLOT_B, 2u * CX_SLOT_B, 3u * CX_SLOT_B};
  unsigned bslot_b[4] = {0u, 256u, 512u, 768u};
  unsigned sat = 0u;
  u4 xh[10], xm[10], xl[10];           // operands of the current layer (K <= 320): three pieces, one tile
  u4 yh[8], ym[8], yl[8];              // ... of the next layer
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
    if constexpr (decltype(nonneg)::value) sat = sat_acc_nonneg(sat, h);
    else sat_in = sat_acc(sat_in, h);
  };
  auto fold_sat_in = [&]() {
    if ((short)(sat_in & 0xffffu) >= 0x7ffe || (short)(sat_in >> 16) >= 0x7ffe) sat = 0x7c007c00u;
    sat_in = 0u;
  };
  // this round's rows -> operands of layer 0: features in place (x feat_scale), tail encoded by the four lane groups of a row
  auto load_layer0 = [&]() {
    const bool ok = rrow < M;
    const long rr = ok ? rrow : 0;
    float in0[80];
    const float* pf = feat + rr * feat_stride + g * 4;
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) {
      const f4u8x v = *reinterpret_cast<const f4u8x*>(pf + kb * 16);
#pragma unroll
      for (int r = 0; r < 4; ++r) in0[kb * 4 + r] = ok ? v[r] * feat_scale : 0.f;
    }
    float pv[3], px[3], pn[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      pv[c] = pview[3 * rr + c];
      px[c] = pxyz[3 * rr + c];
      pn[c] = pnormal[3 * rr + c];
    }
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
      const f4 v = ok ? pt[kb * 4] : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r) in0[64 + kb * 4 + r] = v[r];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) in0[76 + r] = 0.f;
#pragma unroll
    for (int kb = 0; kb < 10; ++kb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = (2 * kb + (q >> 1)) * 4 + (q & 1) * 2;
        put_pair(in0[i], in0[i + 1], xh[kb], xm[kb], xl[kb], q, std::false_type{});
      }
    fold_sat_in();
  };

  auto run_layer = [&](auto LI_tag, int cb) {
    constexpr int LI = decltype(LI_tag)::value;
    constexpr int K = cx_K(LI), KB = K / 32, NCH = cx_nch(LI), NP = sx_np(K), CB = 16 * LI;
    constexpr bool OUT = LI == 4;
    constexpr int BS = 2, DB = 1, D = BS * DB, NB = BS * (DB + 1);
    constexpr int HB = KB / 2, NSTEP = NCH * KB;
    static_assert(D + BS - 1 <= KB - HB, "reads of the next chunk start after the barrier");
    SxAcc accs[2];
    f4 bnext = f4{0.f, 0.f, 0.f, 0.f};
    u4 wfh[NB], wfm[NB], wfl[NB];
    const f4* wl = Wp + cx_coff(cb);
    const f4* wnext[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) wnext[i] = Wp + cx_coff(cb + NCH + i);
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
      put_pair(fmaxf(combine(a, 2 * q), 0.f), fmaxf(combine(a, 2 * q + 1), 0.f), yh[pj >> 1], ym[pj >> 1], yl[pj >> 1], (pj & 1) * 2 + q, std::true_type{});
    };
    zero_acc(accs[0], bias_of(0));
#pragma unroll
    for (int i = 0; i < D; ++i)
      if (i < NSTEP) {
        const u4* f = frag_of(i / KB) + (3 * (i % KB)) * 64;
        wfh[i % NB] = f[0];
        wfm[i % NB] = f[64];
        wfl[i % NB] = f[128];
      }
#pragma unroll
    for (int jb = 0; jb < NCH; ++jb) {
      SxAcc& acc = accs[jb & 1];
      if (jb > 0) zero_acc(acc, bnext);
      constexpr int dummy2 = 0;
      (void)dummy2;
      const int K3 = jb + 3 < NCH ? K : cx_K(cx_layer_of(CB + jb + 3));
      const int nu3 = sx_units(K3);
      const f4* src3 = jb + 3 < NCH ? wl + (long)(jb + 3) * sx_cf4(K) : wnext[jb + 3 - NCH < 3 ? jb + 3 - NCH : 0];
      const int sl3 = (jb + 3) & 3;
      const unsigned dst3 = ring_b + slot_b[sl3], bdst3 = bias_b + bslot_b[sl3];
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const int st = jb * KB + kb;
        if (kb == HB) {   // chunk jb+1 must have landed: this wave's copies of chunk jb+2 may still be in flight
          const int allowed = jb + 2 < NCH ? NP : sx_np(cx_K(cx_layer_of(CB + jb + 2)));
          if (allowed >= 9) sx_wait<9>();
          else sx_wait<7>();
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          bnext = bias_of(jb + 1);
        }
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
        if (st % BS == BS - 1 || kb == KB - 1) {
          const int k0 = (st % BS == BS - 1) ? (kb - (BS - 1) > 0 ? kb - (BS - 1) : 0) : kb - (st % BS);
#define CX_MFMA(ACC, W, X) ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, W), __builtin_bit_cast(h8, X), ACC, 0, 0, 0)
#pragma unroll
          for (int k = k0; k <= kb; ++k) CX_MFMA(acc.c2, wfl[(jb * KB + k) % NB], xh[k]);
#pragma unroll
          for (int k = k0; k <= kb; ++k) CX_MFMA(acc.c2, wfm[(jb * KB + k) % NB], xm[k]);
#pragma unroll
          for (int k = k0; k <= kb; ++k) CX_MFMA(acc.c2, wfh[(jb * KB + k) % NB], xl[k]);
#pragma unroll
          for (int k = k0; k <= kb; ++k) CX_MFMA(acc.c1, wfm[(jb * KB + k) % NB], xh[k]);
#pragma unroll
          for (int k = k0; k <= kb; ++k) CX_MFMA(acc.c1, wfh[(jb * KB + k) % NB], xm[k]);
#pragma unroll
          for (int k = k0; k <= kb; ++k) CX_MFMA(acc.c0, wfh[(jb * KB + k) % NB], xh[k]);
#undef CX_MFMA
        }
        if (jb > 0 && !OUT) {                  // relu + three-way split of chunk jb-1
          if (kb == 0) hidden_pair(accs[(jb - 1) & 1], jb - 1, 0);
          if (kb == 3) hidden_pair(accs[(jb - 1) & 1], jb - 1, 1);
        }
        if (kb >= HB) {
#pragma unroll
          for (int u = 0; u < 3; ++u)
            if (u < nu3 && (u * (KB - HB)) / nu3 == kb - HB) {
              if (K3 == 320) sx_copy_unit<320>(u, src3, lane4, lane16, bdst3, dst3, wave);
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
#pragma unroll
        for (int c = 0; c < 3; ++c) rgb[rrow * 3 + c] = 1.0f / (1.0f + expf(-combine(last, c)));
      }
    } else {
      hidden_pair(last, NCH - 1, 0);
      hidden_pair(last, NCH - 1, 1);
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) {
        xh[kb] = yh[kb];
        xm[kb] = ym[kb];
        xl[kb] = yl[kb];
      }
    }
  };

  // ---- prologue: chunks 0, 1, 2 of the stream (layer 0: K = 320)
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int u = 0; u < 3; ++u)
      sx_copy_unit<320>(u, Wp + cx_coff(c), lane4, lane16, bias_b + bslot_b[c], ring_b + slot_b[c], wave);
  sx_wait<0>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  for (long round = blockIdx.x; round < nrounds; round += gridDim.x) {
    rrow = round * 64 + wave * 16 + (lane & 15);
    load_layer0();
    // layer 0 (K = 320) | 1, 2 (one instance) | 3 (followed by the output chunk and the next round's first chunks) | 4
#pragma unroll 1
    for (int l = 0; l < 5; ++l) {
      if (l == 0) run_layer(std::integral_constant<int, 0>{}, 0);
      else if (l == 3) run_layer(std::integral_constant<int, 3>{}, 48);
      else if (l == 4) run_layer(std::integral_constant<int, 4>{}, 64);
      else run_layer(std::integral_constant<int, 1>{}, 16 * l);
    }
  }
  range_report<true>(sat, range_word);
  sx_wait<0>();
  __syncthreads();
}

}  // namespace rb

using namespace rb;

extern "C" int rb_color_x6_points(const float* feat, long feat_stride, float feat_scale, const float* x, float x_scale, const float* view,
                                  const float* normal, long M, const float* Wp, float* rgb, int two_tile, int n_workgroups,
                                  rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(feat && x && view && normal && Wp && rgb, "null pointer");
  if (two_tile)      // two 16-row tiles per wave, rounds of 128 rows (color_x6t.hip): the form for launches that fill the grid
    return rb::launch_color_x6t(feat, feat_stride, feat_scale, x, x_scale, view, normal, M, Wp, rgb, n_workgroups, (hipStream_t)stream);
  const int pg = persistent_grid((M + 63) / 64, n_workgroups);
  if (pg <= 0) return rb::fail(__func__, "device query failed");
  const unsigned grid = (unsigned)pg;
  hipLaunchKernelGGL(k_color_x6, dim3(grid), dim3(256), 0, (hipStream_t)stream, feat, feat_stride, feat_scale, x, x_scale, view, normal, M,
                     (const f4*)Wp, rgb, range_flags() ? range_flags() + RB_RANGE_COLOR : nullptr);
  return check_launch("k_color_x6");
}
