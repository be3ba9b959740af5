// d sdf / d (encoded input) by reverse mode with EXACT fp32 operands on the f16 matrix pipe ("f16x6") -- the default precision
// policy's gradient pass, round 3: k_sdf_back_f32 (mlp_kernels.hip, f32-input MFMA) on the machine of sdf_x6.hip.
//
// In: the sigmoid tiles the value pass stored (k_sdf_x6<5> / k_sdf_mlp<5>: [tile = row / 16][layer 8][chunk 16][lane 64] float4).
// Out: two 64-wide gradient rows per point (layer 0's and the skip connection's share; k_pe_grad_points contracts them with the
// encoding's Jacobian).  The net, back to front, as one cyclic stream of 117 chunks (16 output rows x K x 3 pieces):
//   stream layer   0     1     2     3             4            5     6     7
//   matrix         W7^T  W6^T  W5^T  [W4^T]        W3^T         W2^T  W1^T  W0^T
//   K              256   256   256   256           224          256   256   256
//   chunks         16    16    16    13 + 4 skip   16           16    16    4
//   gate (sigmoid) l6    l5    l4    l3 (x 1/sqrt2; skip rows: x 1/sqrt2, out)   l2  l1  l0   -- (out)
// A layer's operands are dz = dh (.) sigmoid(100 z): the gate of a chunk's sixteen rows is one float4 per lane, loaded at the top of the
// layer (sixteen plain loads, credited in the counted waits of the layer's first two chunks) and multiplied in where the forward
// kernels apply the softplus.  Weights: packing.pack_sdf_back_x6.
#include "../../include/robir_hip.h"
#include "common.h"
#include "mlp_engine.h"
#include "x6_ring.h"
#include <cstdlib>
#include <type_traits>

namespace rb {

constexpr int BX_SLOT_B = 24 * 1024 + 512;
constexpr int BX_NCHUNK = 117;
#ifndef BX_NSLOT
#define BX_NSLOT 4                       // ring slots: copies run BX_NSLOT - 1 chunks ahead of the MFMAs (5: measured the same, 13.0-13.3 vs
                                         // 13.0-13.5 ms per 2^20 points for value + gradient: the copies' latency is not what the waves wait for)
#endif
constexpr int BX_AH = BX_NSLOT - 1;
__host__ __device__ constexpr int bx_K(int l) { return l == 4 ? 224 : 256; }
__host__ __device__ constexpr int bx_nch(int l) { return l == 3 ? 17 : (l == 7 ? 4 : 16); }
__host__ __device__ constexpr int bx_cbase(int l) {
  int n = 0;
  for (int i = 0; i < l; ++i) n += bx_nch(i);
  return n;
}
__host__ __device__ constexpr int bx_layer_of(int c) {
  if (c >= BX_NCHUNK) c -= BX_NCHUNK;
  int l = 0, first = 0;
  for (int i = 0; i < 7; ++i) {
    first += bx_nch(i);
    if (c >= first) l = i + 1;
  }
  return l;
}
__host__ __device__ constexpr long bx_coff(int c) {
  if (c >= BX_NCHUNK) c -= BX_NCHUNK;
  long off = 0;
  int first = 0, base = 0, kl = bx_K(0);
  for (int i = 0; i < 7; ++i) {
    first += bx_nch(i);
    if (c >= first) {
      off += (long)bx_nch(i) * sx_cf4(bx_K(i));
      base = first;
      kl = bx_K(i + 1);
    }
  }
  return off + (long)(c - base) * sx_cf4(kl);
}
static_assert(sx_np(224) == 7 && sx_np(256) == 7 && sx_units(224) == 3 && sx_units(256) == 3, "every chunk of this stream is seven copies in three blocks per wave");

__global__ __launch_bounds__(256, 1) void k_sdf_back_x6(const f4* __restrict__ sig, long M, const f4* __restrict__ Wt,
                                                         const float* __restrict__ w8row, float* __restrict__ gfeat,
                                                         unsigned* __restrict__ range_word) {
  __shared__ f4 ring[BX_NSLOT * BX_SLOT_B / 16];       // 98 / 123 KB
  __shared__ f4 bias_ring[BX_NSLOT * 16];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long nrounds = (M + 63) >> 6;
  if ((long)blockIdx.x >= nrounds) return;

  const float negk = -2048.0f, inv_sqrt2 = 0.70710678118654752440f;
  constexpr float C11 = 1.0f / 2048.0f;
  const unsigned ring_b = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)ring);
  const unsigned bias_b = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)bias_ring);
  const unsigned lane16 = (unsigned)lane * 16u, lane4 = (unsigned)lane * 4u;
  unsigned slot_b[BX_NSLOT], bslot_b[BX_NSLOT];
#pragma unroll
  for (int i = 0; i < BX_NSLOT; ++i) {
    slot_b[i] = (unsigned)i * BX_SLOT_B;
    bslot_b[i] = (unsigned)i * 256u;
  }
  unsigned sat = 0u;
  u4 xh[8], xm[8], xl[8];              // operands of the current layer: three pieces, one tile
  u4 yh[8], ym[8], yl[8];              // ... of the next layer
  long rrow = 0;
  const f4* sig_tile = sig;            // this wave's tile of the round: + (layer * 16 + chunk) * 64

  auto put_pair = [&](float v0, float v1, u4& dh, u4& dm, u4& dl, int q) {
    unsigned h, m, l;
    sx_split_pair(v0, v1, negk, h, m, l);
    dh[q] = h;
    dm[q] = m;
    dl[q] = l;
    sat = sat_acc(sat, h);
  };
  // operands of stream layer 0: d sdf / d h7 (row 0 of layer 8) gated by layer 7's sigmoid
  auto load_layer0 = [&]() {
    float p[64];
#pragma unroll
    for (int blk = 0; blk < 16; ++blk) {
      const f4 w = *reinterpret_cast<const f4*>(w8row + blk * 16 + 4 * g);
      const f4 s = sig_tile[(7 * 16 + blk) * 64];
#pragma unroll
      for (int r = 0; r < 4; ++r) p[blk * 4 + r] = w[r] * s[r];
    }
#pragma unroll
    for (int kb = 0; kb < 8; ++kb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = (2 * kb + (q >> 1)) * 4 + (q & 1) * 2;
        put_pair(p[i], p[i + 1], xh[kb], xm[kb], xl[kb], q);
      }
  };

  // GATE: sigmoid layer that gates this stream layer's outputs (-1: none, stream layer 7)
  auto run_layer = [&](auto LI_tag, int cb, int gate) {
    constexpr int LI = decltype(LI_tag)::value;
    constexpr int K = bx_K(LI), KB = K / 32, NCH = bx_nch(LI), CB = bx_cbase(LI);
    constexpr bool OUT = LI == 7, SKIPL = LI == 3;
    constexpr int NSIG = OUT ? 0 : (SKIPL ? 13 : 16);
    constexpr int BS = (KB % 2 == 0) ? 2 : 1, DB = BS == 2 ? 1 : 2, D = BS * DB, NB = BS * (DB + 1);
    constexpr int HB = KB / 2, NSTEP = NCH * KB;
    static_assert(D + BS - 1 <= KB - HB, "reads of the next chunk start after the barrier");
    SxAcc accs[2];
    f4 bnext = f4{0.f, 0.f, 0.f, 0.f};
    u4 wfh[NB], wfm[NB], wfl[NB];
    const f4* wl = Wt + bx_coff(cb);
    const f4* wnext[BX_AH];
#pragma unroll
    for (int i = 0; i < BX_AH; ++i) wnext[i] = Wt + bx_coff(cb + NCH + i);
    asm volatile("" : "+s"(wl));
    auto frag_of = [&](int c) { return reinterpret_cast<const u4*>(reinterpret_cast<const char*>(ring) + slot_b[c % BX_NSLOT]) + lane; };
    auto bias_of = [&](int c) { return *(reinterpret_cast<const f4*>(reinterpret_cast<const char*>(bias_ring) + bslot_b[c % BX_NSLOT]) + g); };
    auto zero_acc = [&](SxAcc& a, const f4& b) {
      a.c0 = b;
      a.c1 = f4{0.f, 0.f, 0.f, 0.f};
      a.c2 = f4{0.f, 0.f, 0.f, 0.f};
    };
    auto combine = [&](const SxAcc& a, int r) { return __builtin_fmaf(__builtin_fmaf(a.c2[r], C11, a.c1[r]), C11, a.c0[r]); };
    // the gates of this layer's output chunks (plain loads: in flight over the first chunks)
    f4 sg[NSIG > 0 ? NSIG : 1];
    if constexpr (NSIG > 0) {
#pragma unroll
      for (int i = 0; i < NSIG; ++i) sg[i] = sig_tile[((long)gate * 16 + i) * 64];
    }
    auto hidden_pair = [&](const SxAcc& a, int pj, int q) {
      float v0 = combine(a, 2 * q), v1 = combine(a, 2 * q + 1);
      if (SKIPL) {
        v0 *= inv_sqrt2;
        v1 *= inv_sqrt2;
      }
      const f4 s = sg[pj < NSIG ? pj : 0];
      put_pair(v0 * s[2 * q], v1 * s[2 * q + 1], yh[pj >> 1], ym[pj >> 1], yl[pj >> 1], (pj & 1) * 2 + q);
    };
    auto output_chunk = [&](const SxAcc& a, int col0, float scale) {      // sixteen gradient columns of this lane's row
      if (rrow < M) *(reinterpret_cast<f4*>(gfeat + rrow * 128 + col0) + g) = f4{combine(a, 0) * scale, combine(a, 1) * scale, combine(a, 2) * scale, combine(a, 3) * scale};
    };
    auto epilogue = [&](const SxAcc& a, int pj, int q) {      // q = 0, 1: halves of a hidden chunk; outputs go out at q = 0
      if (OUT) {
        if (q == 0) output_chunk(a, pj * 16, 1.0f);
      } else if (SKIPL && pj >= 13) {
        if (q == 0) output_chunk(a, 64 + (pj - 13) * 16, inv_sqrt2);
      } else {
        hidden_pair(a, pj, q);
      }
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
      const int K3 = jb + BX_AH < NCH ? K : bx_K(bx_layer_of(CB + jb + BX_AH));
      const f4* src3 = jb + BX_AH < NCH ? wl + (long)(jb + BX_AH) * sx_cf4(K) : wnext[jb + BX_AH - NCH < BX_AH ? jb + BX_AH - NCH : 0];
      const int sl3 = (jb + BX_AH) % BX_NSLOT;
      const unsigned dst3 = ring_b + slot_b[sl3], bdst3 = bias_b + bslot_b[sl3];
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const int st = jb * KB + kb;
        if (kb == HB) {   // chunk jb+1 must have landed: the copies of chunk jb+2 (7) -- and, in the layer's first two chunks, the
                          // younger gate loads -- may still be in flight
          // in flight behind chunk jb+1: the copies of chunks jb+2 .. jb+BX_AH-1 (7 each)
          if (jb < BX_AH - 1 && NSIG == 16) sx_wait<7 * (BX_AH - 2) + 16>();
          else if (jb < BX_AH - 1 && NSIG == 13) sx_wait<7 * (BX_AH - 2) + 13>();
          else sx_wait<7 * (BX_AH - 2)>();
#ifndef BX_ABL_NOBAR                  // timing ablation (wrong results)
          __builtin_amdgcn_s_barrier();
#endif
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
#define BX_MFMA(ACC, W, X) ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, W), __builtin_bit_cast(h8, X), ACC, 0, 0, 0)
#pragma unroll
          for (int k = k0; k <= kb; ++k) BX_MFMA(acc.c2, wfl[(jb * KB + k) % NB], xh[k]);
#pragma unroll
          for (int k = k0; k <= kb; ++k) BX_MFMA(acc.c2, wfm[(jb * KB + k) % NB], xm[k]);
#pragma unroll
          for (int k = k0; k <= kb; ++k) BX_MFMA(acc.c2, wfh[(jb * KB + k) % NB], xl[k]);
#pragma unroll
          for (int k = k0; k <= kb; ++k) BX_MFMA(acc.c1, wfm[(jb * KB + k) % NB], xh[k]);
#pragma unroll
          for (int k = k0; k <= kb; ++k) BX_MFMA(acc.c1, wfh[(jb * KB + k) % NB], xm[k]);
#pragma unroll
          for (int k = k0; k <= kb; ++k) BX_MFMA(acc.c0, wfh[(jb * KB + k) % NB], xh[k]);
#undef BX_MFMA
        }
        if (jb > 0) {
          if (kb == 0) epilogue(accs[(jb - 1) & 1], jb - 1, 0);
          if (kb == 3) epilogue(accs[(jb - 1) & 1], jb - 1, 1);
        }
        if (kb >= HB) {
#pragma unroll
          for (int u = 0; u < 3; ++u)
#ifdef BX_ABL_NODMA
            if (false) {
#else
            if ((u * (KB - HB)) / 3 == kb - HB) {
#endif
              if (K3 == 224) sx_copy_unit<224>(u, src3, lane4, lane16, bdst3, dst3, wave);
              else sx_copy_unit<256>(u, src3, lane4, lane16, bdst3, dst3, wave);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    {   // slot 0 = the slot of the next layer's first chunk
      constexpr int R = NCH % BX_NSLOT;
      unsigned a[BX_NSLOT], b[BX_NSLOT];
#pragma unroll
      for (int i = 0; i < BX_NSLOT; ++i) {
        a[i] = slot_b[(i + R) % BX_NSLOT];
        b[i] = bslot_b[(i + R) % BX_NSLOT];
      }
#pragma unroll
      for (int i = 0; i < BX_NSLOT; ++i) {
        slot_b[i] = a[i];
        bslot_b[i] = b[i];
      }
    }
    const SxAcc& last = accs[(NCH - 1) & 1];
    epilogue(last, NCH - 1, 0);
    epilogue(last, NCH - 1, 1);
    if constexpr (!OUT) {
      constexpr int KBN = bx_K(LI + 1) / 32;
#pragma unroll
      for (int kb = 0; kb < KBN; ++kb) {
        xh[kb] = yh[kb];
        xm[kb] = ym[kb];
        xl[kb] = yl[kb];
      }
      if constexpr (SKIPL) {      // W3^T takes 224 = 208 + 16 zero slots: the second half of k-block 6 is padding
#pragma unroll
        for (int q = 2; q < 4; ++q) {
          xh[6][q] = 0u;
          xm[6][q] = 0u;
          xl[6][q] = 0u;
        }
      }
    }
  };

  // ---- prologue: chunks 0, 1, 2 of the stream
#pragma unroll
  for (int c = 0; c < BX_AH; ++c)
#pragma unroll
    for (int u = 0; u < 3; ++u)
      sx_copy_unit<256>(u, Wt + bx_coff(c), lane4, lane16, bias_b + bslot_b[c], ring_b + slot_b[c], wave);
  sx_wait<0>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  for (long round = blockIdx.x; round < nrounds; round += gridDim.x) {
    rrow = round * 64 + wave * 16 + (lane & 15);
    sig_tile = sig + ((round * 4 + wave) * 8) * (16L * 64) + lane;
    load_layer0();
    // stream layers 0, 1, 2, 5, 6 (one instance) | 3 (W4^T: gate + skip rows) | 4 (W3^T, K = 224) | 7 (W0^T: outputs)
#pragma unroll 1
    for (int l = 0; l < 8; ++l) {
      const int cb = l < 4 ? 16 * l : (l == 4 ? 65 : 81 + 16 * (l - 5));
      if (l == 3) run_layer(std::integral_constant<int, 3>{}, cb, 3);
      else if (l == 4) run_layer(std::integral_constant<int, 4>{}, cb, 2);
      else if (l == 7) run_layer(std::integral_constant<int, 7>{}, cb, -1);
      else run_layer(std::integral_constant<int, 0>{}, cb, 6 - l);
    }
  }
  range_report(sat, range_word);
  sx_wait<0>();
  __syncthreads();
}

// host-side launcher for sdf_back.hip (rb_sdf_value_grad_x6_points)
int launch_sdf_back_x6(const float* sig, long M, const float* Wt, const float* w8row, float* gfeat, hipStream_t s) {
  const int pg = persistent_grid((M + 63) / 64, 0);
  if (pg <= 0) return rb::fail(__func__, "device query failed");
  const unsigned grid = (unsigned)pg;
  hipLaunchKernelGGL(k_sdf_back_x6, dim3(grid), dim3(256), 0, s, (const f4*)sig, M, (const f4*)Wt, w8row, gfeat,
                     range_flags() ? range_flags() + RB_RANGE_SDF : nullptr);
  return check_launch("k_sdf_back_x6");
}

}  // namespace rb
