// CESR normal_net / shadow_net (eight softplus(100) layers of 512 with a skip connection into layer 4; inputs PE10(x) or [PE10(x) | one-hot
// label]: training/train_cesr.py:106-110,331-352,492-504) in PLAIN f16 -- the labelled THROUGHPUT mode `BASELINE.json configs[4]` names
// ("CESR stage full pipeline, fp16 MLP weights on MFMA"; ROBIR_PRECISION=f16), round 6.  ONE f16 MFMA product per multiply-add: f16 weights
// (the h pieces of the exact-operand blob: what cesr_x6.hip multiplies first), f16 activations (truncated between the layers:
// v_cvt_pkrtz), fp32 accumulation, fp32 softplus.  NARROWER than the reference's fp32: never a default, never a parity claim
// (tests/test_precision_gpu.py prints its distance from a float64 evaluation).
//
// Shape: persistent workgroups of four waves, T 16-row tiles per wave (rounds of 64 T rows); a chunk = 16 output neurons x K halves = K / 32
// fragments of 1 KB, streamed through a 4-slot LDS ring by LDS-DMA three chunks ahead under counted waits (K / 128 copies per wave and
// chunk: a third of the exact-operand stream's bytes for a sixth of its MFMAs -- the copies' issue cost is what bounds this kernel, so
// T = 3 tiles amortise them over 48 MFMAs); per k-block ONE fragment read feeds T MFMAs; the previous chunk's epilogue (softplus + pack,
// seven single-instruction steps per value) goes two steps behind every MFMA (x6t_engine.h: the k-major finding of round 6).
// All biases are resident in the LDS; the encoded rows of a round too (the skip layer's input part is rebuilt from them).
#include "../../include/robir_hip.h"
#include "common.h"
#include "mlp_engine.h"
#include "x6_ring.h"
#include "x6t_engine.h"
#include <type_traits>

#ifndef FX_CG
#define FX_CG 1             // MFMAs between two LDS-DMA copies of a wave within a chunk (4: measured no gain, 3.47 / 3.40 -> 3.59 / 3.42 ms: profiles/r06_dma_placement.md)
#endif

namespace rb {

constexpr int FX_SLOT_B = 18 * 1024;      // K = 576: eighteen 1 KB fragments
template <int K0P, int N3P>
struct FxNet {
  static_assert(K0P + N3P == 528 && (K0P == 64 || K0P == 192), "normal_net (64, 464) or shadow_net (192, 336)");
  __host__ __device__ static constexpr int K(int l) { return l == 0 ? K0P : (l == 4 ? 576 : 512); }
  __host__ __device__ static constexpr int nch(int l) { return l == 3 ? N3P / 16 : (l == 8 ? 1 : 32); }
  __host__ __device__ static constexpr int total() {
    int n = 0;
    for (int l = 0; l < 9; ++l) n += nch(l);
    return n;
  }
  __host__ __device__ static constexpr int cbase(int l) {
    int n = 0;
    for (int i = 0; i < l; ++i) n += nch(i);
    return n;
  }
  __host__ __device__ static constexpr int layer_of(int c) {      // cyclic: the stream runs into the next round
    while (c >= total()) c -= total();
    int l = 0, first = 0;
    for (int i = 0; i < 8; ++i) {
      first += nch(i);
      if (c >= first) l = i + 1;
    }
    return l;
  }
  __host__ __device__ static constexpr long foff(int c) {         // float4 offset of chunk c's fragments behind the bias table
    while (c >= total()) c -= total();
    long off = 0;
    for (int l = 0; l < 9; ++l)
      for (int j = 0; j < nch(l); ++j) {
        if (cbase(l) + j == c) return off;
        off += (long)(K(l) / 32) * 64;
      }
    return off;
  }
};
__host__ __device__ constexpr int fx_nsw(int K) { return (K / 32 + 3) / 4; }      // copies per wave and chunk

// ONEHOT: rows = (point, label) pairs, row = point * n_label + label (shadow_net); else one row per point (normal_net)
template <int K0P, int N3P, bool ONEHOT, int T>
__global__ __launch_bounds__(256, 1) void k_cesr_f16(const float* __restrict__ X, long M, int n_label, const f4* __restrict__ Wp, int n_out,
                                                      float* __restrict__ Y, unsigned* __restrict__ range_word) {
  using Net = FxNet<K0P, N3P>;
  constexpr int NCHUNK = Net::total();
  __shared__ f4 ring[4 * FX_SLOT_B / 16];              // 72 KB
  __shared__ f4 bias_tab[NCHUNK * 4];                  // 16 KB: the 16 biases of every chunk of the stream
  __shared__ float pe_scratch[4 * T * 16 * 64];        // [wave][tile][row 16][64 encoded inputs]
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr long R = 64 * T;
  const long nrounds = (M + R - 1) / R;
  if ((long)blockIdx.x >= nrounds) return;

  const float inv_sqrt2 = 0.70710678118654752440f;
  const unsigned ring_b = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)ring);
  unsigned ring_lane = ring_b + (unsigned)lane * 16u;
  asm volatile("" : "+v"(ring_lane));
  const f4* frag_base = Wp + NCHUNK * 4;
  unsigned sat = 0u;
  u4 x[T][18];                         // operands of the current layer (K <= 576), T tiles: one 128-bit tuple (8 halves) per k-block
  u4 y[T][16];                         // ... of the next layer
  long rrow[T];
  int label[T];
  long round = 0;

  for (int i = tid; i < NCHUNK * 4; i += 256) bias_tab[i] = Wp[i];

  auto pack2 = [&](float v0, float v1) { return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v0, v1)); };
  // input value k of tile t's row of this lane (k = 16 blk + 4 g + r): the encoder's LDS row, then the one-hot block from column 63 on
  auto x0_block = [&](int t, int blk, float scale, float(&v)[4]) {
    const float* frow = pe_scratch + (wave * T + t) * 1024 + (lane & 15) * 64;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = blk * 16 + 4 * g + r;
      float e = 0.f;
      if (blk < 4) e = frow[blk * 16 + 4 * g + r];
      if (ONEHOT) {
        if (k == 63) e = 0.f;
        if (k >= 63 && k - 63 == label[t]) e = 1.f;
      }
      v[r] = e * scale;
    }
  };
  auto load_layer0 = [&]() {
#pragma unroll
    for (int t = 0; t < T; ++t) {
      // rows beyond M compute on the LAST valid row's inputs (no select on the data path; only the store below is guarded)
      const long rin = rrow[t] < M ? rrow[t] : M - 1;
      float enc[16];
      load_features_pe10x(X, nullptr, rin, M, lane, pe_scratch + (wave * T + t) * 1024, enc, ONEHOT ? (long)n_label : 1L);
      (void)enc;
      label[t] = ONEHOT ? (int)(rin % n_label) : -1;
#pragma unroll
      for (int blk = 0; blk < K0P / 16; ++blk) {
        float v[4];
        x0_block(t, blk, 1.0f, v);
        x[t][blk >> 1][(blk & 1) * 2] = pack2(v[0], v[1]);
        x[t][blk >> 1][(blk & 1) * 2 + 1] = pack2(v[2], v[3]);
      }
    }
  };
  // the skip layer's operands: [softplus(h3) / sqrt 2 (N3P / 16 blocks, in y) | x0 / sqrt 2 (K0P / 16 blocks) | 0]
  auto build_skip_operands = [&]() {
    constexpr int B3 = N3P / 16;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int b = 0; b < 36; ++b) {
        const int kb = b >> 1, q0 = (b & 1) * 2;
        if (b < B3) {
          x[t][kb][q0] = y[t][kb < 16 ? kb : 0][q0];
          x[t][kb][q0 + 1] = y[t][kb < 16 ? kb : 0][q0 + 1];
        } else if (b < B3 + K0P / 16) {
          float v[4];
          x0_block(t, b - B3, inv_sqrt2, v);
          x[t][kb][q0] = pack2(v[0], v[1]);
          x[t][kb][q0 + 1] = pack2(v[2], v[3]);
        } else {
          x[t][kb][q0] = 0u;
          x[t][kb][q0 + 1] = 0u;
        }
      }
  };

  unsigned slot_b[4] = {0u, (unsigned)FX_SLOT_B, 2u * FX_SLOT_B, 3u * FX_SLOT_B};

  // runtime layer index -> its first chunk / the float4 offset of its fragments (a code instance serves several layers of one shape)
  auto cbase_rt = [&](int l) {
    int n = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i)
      if (i < l) n += Net::nch(i);
    return n;
  };
  auto loff_rt = [&](int l) {
    long off = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i)
      if (i < l) off += (long)Net::nch(i) * (Net::K(i) / 32) * 64;
    return off;
  };
  auto frag_of = [&](unsigned addr, int kb) { return ((xt_lds_u4p)addr)[kb * 64]; };

  // LI: the layer whose SHAPE (K, chunks, the K of the chunks up to three behind its end) this instance is compiled for; l: the layer it runs
  auto run_layer = [&](auto LI_tag, int l) {
    constexpr int LI = decltype(LI_tag)::value;
    constexpr int K = Net::K(LI), KB = K / 32, NCH = Net::nch(LI), CB = Net::cbase(LI);
    const int cb = cbase_rt(l);
    const f4* wl[3];                                   // fragments of this layer and of the two layers behind it (cyclic)
#pragma unroll
    for (int i = 0; i < 3; ++i) wl[i] = frag_base + loff_rt((l + i) % 9);
    constexpr bool OUT = LI == 8, SKIPOUT = LI == 3;
    constexpr int D = KB >= 4 ? 4 : 2;                 // fragment registers = prefetch distance in k-blocks
    constexpr int NSLOT = KB * T;                      // MFMAs (= filler slots) per chunk
    constexpr int NV = SKIPOUT ? 7 : 6;                // steps per value: -|z| k, exp2, 1 + e, log2, max, fma [, / sqrt 2]
    constexpr int NT = 4 * NV + 4, NM = T * NT;        // ... + per pair: pack, range sentinel
    // steps behind every MFMA that carries no copy (at most five copies per wave and chunk): two where the chunk is long enough, at most
    // four; what a short chunk (layer 0: K = 64 / 192) cannot carry runs behind its MFMAs
    constexpr int PER_ = (NM + (NSLOT > 5 ? NSLOT - 5 : 1) - 1) / (NSLOT > 5 ? NSLOT - 5 : 1);
    constexpr int PER = OUT ? 0 : (PER_ < 2 ? 2 : (PER_ > 4 ? 4 : PER_));
    constexpr bool LEFTOVER = !OUT && PER * (NSLOT > 5 ? NSLOT - 5 : 0) < NM;
    u4 win[4];
    f4 accs[2][T];
    unsigned lv = 0u;
    auto bias_of = [&](int c) { return bias_tab[(c >= NCHUNK ? c - NCHUNK : c) * 4 + g]; };
    // epilogue of hidden chunk pj (accumulators pa) as single-instruction steps
    float tt[4], uu[4], lg[4], mx[4], vv[4];
    unsigned pk[2];
    auto micro = [&](int sidx, int pj, const f4(&pa)[T]) {
      const int t = sidx / NT, u = sidx % NT;
      const f4& a = pa[t];
#ifdef FX_ABL_NOSOFTPLUS                 // timing ablation (wrong results): the activation is the identity -- pack + sentinel only
      if (u < 4) vv[u] = a[u];
      else if (u < 4 * NV) {
      } else
#else
      // the six steps of the four values as a skewed pipeline: every pair of consecutive steps (= one MFMA slot) holds at most ONE of the two
      // quarter-rate transcendentals, and a step's input was produced two or more steps earlier
      //   T0 T1 | E0 T2 | E1 T3 | U0 E2 | U1 E3 | L0 U2 | L1 U3 | M0 L2 | M1 L3 | V0 M2 | V1 M3 | V2 V3      (T: -|z| k, E: exp2, U: 1 + e, L: log2, M: max, V: fma)
      constexpr int KIND[24] = {0, 0, 1, 0, 1, 0, 2, 1, 2, 1, 3, 2, 3, 2, 4, 3, 4, 3, 5, 4, 5, 4, 5, 5};
      constexpr int RIDX[24] = {0, 1, 0, 2, 1, 3, 0, 2, 1, 3, 0, 2, 1, 3, 0, 2, 1, 3, 0, 2, 1, 3, 2, 3};
      if (u < 24) {
        const int kd = KIND[u], r = RIDX[u];
        if (kd == 0) tt[r] = -__builtin_fabsf(a[r]) * SP_T_PER_Z;
        else if (kd == 1) tt[r] = __builtin_amdgcn_exp2f(tt[r]);
        else if (kd == 2) uu[r] = 1.0f + tt[r];
        else if (kd == 3) lg[r] = __builtin_amdgcn_logf(uu[r]);
        else if (kd == 4) asm("v_max_f32 %0, 0, %1" : "=v"(mx[r]) : "v"(a[r]));      // (fmaxf on an MFMA result costs a canonicalising v_max more)
        else vv[r] = __builtin_fmaf(lg[r], SP_LN2_OVER_100, mx[r]);
      } else if (u < 4 * NV) vv[u - 24] = vv[u - 24] * inv_sqrt2;
      else
#endif
      {
        const int w_ = u - 4 * NV, p = w_ & 1;
        if (w_ < 2) {
          pk[p] = pack2(vv[2 * p], vv[2 * p + 1]);
        } else {
          sat = sat_acc_pos(sat, pk[p]);
          y[t][pj >> 1][(pj & 1) * 2 + p] = pk[p];
        }
      }
    };
    // outputs: lanes g == 0 hold neurons 0..3 of their row (rows beyond M computed on the last valid row's inputs: not stored)
    auto output_chunk = [&](const f4(&pa)[T]) {
#pragma unroll
      for (int t = 0; t < T; ++t)
        if (g == 0 && rrow[t] < M) {
          float* yp = Y + rrow[t] * n_out;
          yp[0] = pa[t][0];
          yp[1] = pa[t][1];
          if (n_out > 2) yp[2] = pa[t][2];
        }
    };
    f4 bias = bias_of(cb);
#pragma unroll
    for (int jb = 0; jb < NCH; ++jb) {
      // chunk jb+1 has landed once at most this wave's copies of chunk jb+2 are in flight; past the barrier every wave has finished
      // with chunk jb-1, whose slot the copies of chunk jb+3 reuse
      constexpr int dummy = 0;
      (void)dummy;
      const int K2 = Net::K(Net::layer_of(CB + jb + 2));
      switch (fx_nsw(K2)) {
        case 1: sx_wait<1>(); break;
        case 2: sx_wait<2>(); break;
        case 4: sx_wait<4>(); break;
        default: sx_wait<5>(); break;
      }
#ifndef FX_ABL_NOBAR                     // timing ablation (races): no per-chunk barrier
      __builtin_amdgcn_s_barrier();
#endif
      asm volatile("" ::: "memory");
      const int L3 = Net::layer_of(CB + jb + 3), K3 = Net::K(L3);
      const int e3 = (CB + jb + 3) % NCHUNK - Net::cbase(L3);                    // chunk jb+3's index inside its layer, which lies (L3 - LI) mod 9 layers behind
      const int NC3 = fx_nsw(K3), P3 = K3 / 32;
      const int f3 = wave * NC3 < P3 - NC3 ? wave * NC3 : P3 - NC3;             // this wave's first piece (the last wave's span is shifted back)
      const f4* b3 = wl[(L3 - LI + 9) % 9];
      asm volatile("" : "+s"(b3));                                                   // formed here: hoisted out of the round loop the chunks' source addresses would not fit the SGPR file
      const f4* src3 = b3 + (long)e3 * P3 * 64 + f3 * 64;
      const unsigned dst3 = ring_b + slot_b[(jb + 3) & 3] + (unsigned)f3 * 1024u;
      f4(&acc)[T] = accs[jb & 1];
#pragma unroll
      for (int t = 0; t < T; ++t) acc[t] = bias;
      if (jb == 0) {      // a layer requests its own first fragments (the window never runs across a layer boundary)
#pragma unroll
        for (int i = 0; i < D; ++i) win[i] = frag_of(ring_lane + slot_b[0], i);
      }
      f4 nbias = bias;
      const int nep = (jb > 0 && !OUT) ? NM : 0;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const int sl = jb * KB + kb;                       // position in the layer's fragment stream: register sl % D
#pragma unroll
        for (int t = 0; t < T; ++t) {
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, win[sl % D]), __builtin_bit_cast(h8, x[t][kb]), acc[t], 0, 0, 0);
          const int s = kb * T + t;
          if (t == 0 && sl >= 1) {      // refill the register of the PREVIOUS k-block one MFMA behind its last reader (a read-after-use hazard otherwise: s_nop)
            const int s2 = sl - 1 + D, j2 = s2 / KB, k2 = s2 % KB;
            if (j2 < NCH) win[(sl - 1) % D] = frag_of(ring_lane + slot_b[j2 & 3], k2);
          }
          if (s == NSLOT - T) nbias = bias_of(cb + jb + 1);
          // copy i of this wave goes behind MFMA FX_CG i
          // (a short chunk -- layer 0's K = 64 / 192 -- packs them closer: all of them must fit its NSLOT positions)
          const int cg_fit = NC3 > 1 ? (NSLOT - 1) / (NC3 - 1) : 1;
          const int cg = cg_fit < 1 ? 1 : (cg_fit < FX_CG ? cg_fit : FX_CG);
          const bool copy_here = s % cg == 0 && s / cg < NC3;
          const int copies_before = (s + cg - 1) / cg < NC3 ? (s + cg - 1) / cg : NC3;
          if (copy_here) {
#ifndef FX_ABL_NODMA                     // timing ablation (wrong results): no LDS-DMA copies after the prologue's
            xt_copy_piece_seq(s / cg, src3, dst3, lv);
#endif
          } else if (nep > 0) {
            const int m0 = PER * (s - copies_before);
#pragma unroll
            for (int i = 0; i < PER; ++i)
              if (m0 + i < nep) micro(m0 + i, jb - 1, accs[(jb - 1) & 1]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if constexpr (LEFTOVER) {
        if (nep > 0) {
          const int done = PER * (NSLOT - NC3 > 0 ? NSLOT - NC3 : 0);
#pragma unroll
          for (int m = 0; m < NM; ++m)
            if (m >= done) micro(m, jb - 1, accs[(jb - 1) & 1]);
        }
      }
      bias = nbias;
    }
    {   // slot 0 = the slot of the next layer's first chunk
      constexpr int Rr = NCH & 3;
      unsigned a[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = slot_b[(i + Rr) & 3];
#pragma unroll
      for (int i = 0; i < 4; ++i) slot_b[i] = a[i];
    }
    // the layer's last chunk: finished here (not beside the next layer's first chunk: ~1 % of a round)
    if constexpr (OUT) {
      output_chunk(accs[(NCH - 1) & 1]);
    } else {
#pragma unroll
      for (int i = 0; i < NM; ++i) micro(i, NCH - 1, accs[(NCH - 1) & 1]);
      if constexpr (SKIPOUT) {
        build_skip_operands();
      } else {
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
          for (int kb = 0; kb < 16; ++kb) x[t][kb] = y[t][kb];
      }
    }
  };

  // ---- prologue: chunks 0, 1, 2 of the stream (layer 0)
  {
    constexpr int P0 = K0P / 32, N0 = fx_nsw(K0P);
    const int f0 = wave * N0 < P0 - N0 ? wave * N0 : P0 - N0;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int i = 0; i < N0; ++i) xt_copy_piece(i, frag_base + Net::foff(c) + f0 * 64, ring_b + slot_b[c] + (unsigned)f0 * 1024u);
  }
  sx_wait<0>();
  __syncthreads();

  for (round = blockIdx.x; round < nrounds; round += gridDim.x) {
#pragma unroll
    for (int t = 0; t < T; ++t) rrow[t] = round * R + (wave * T + t) * 16 + (lane & 15);
    load_layer0();
    // layer 0 | 1, 2 | 3 (skip layer's own outputs) | 4 (K = 576) | 5, 6 | 7 (the output layer and the next round's layer 0 follow) | 8
    run_layer(std::integral_constant<int, 0>{}, 0);
#pragma unroll 1
    for (int l = 1; l < 3; ++l) run_layer(std::integral_constant<int, 1>{}, l);
    run_layer(std::integral_constant<int, 3>{}, 3);
    run_layer(std::integral_constant<int, 4>{}, 4);
#pragma unroll 1
    for (int l = 5; l < 7; ++l) run_layer(std::integral_constant<int, 5>{}, l);
    run_layer(std::integral_constant<int, 7>{}, 7);
    run_layer(std::integral_constant<int, 8>{}, 8);
  }
  range_report<false>(sat, range_word);
  sx_wait<0>();
  __syncthreads();
}

}  // namespace rb

using namespace rb;

extern "C" int rb_cesr_net_f16_points(const float* x, long M, int kind, int n_label, const float* Wp, float* Y, int tiles, int n_workgroups,
                                      rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(x && Wp && Y, "null pointer");
#ifndef FX_TILES
#define FX_TILES 3          // tiles per wave the library is built with (2 or 3: one instance per net keeps the build short)
#endif
  RB_REQUIRE(tiles == FX_TILES, "this build carries one tile count (FX_TILES)");
  const int pg = persistent_grid((M + 64 * tiles - 1) / (64 * tiles), n_workgroups);
  if (pg <= 0) return rb::fail(__func__, "device query failed");
  const unsigned grid = (unsigned)pg;
  unsigned* rw = range_flags() ? range_flags() + RB_RANGE_SOFTPLUS512 : nullptr;
  hipStream_t s = (hipStream_t)stream;
  const f4* W = (const f4*)Wp;
  switch (kind) {
    case 0: hipLaunchKernelGGL((k_cesr_f16<64, 464, false, FX_TILES>), dim3(grid), dim3(256), 0, s, x, M, 1, W, 3, Y, rw); break;
    case 2:
      RB_REQUIRE(n_label >= 1 && n_label <= 128, "n_label must be 1..128");
      hipLaunchKernelGGL((k_cesr_f16<192, 336, true, FX_TILES>), dim3(grid), dim3(256), 0, s, x, M, n_label, W, 2, Y, rw);
      break;
    default: return rb::fail(__func__, "kind: 0 normal_net on PE10(x), 2 shadow_net on (point, one-hot label) rows");
  }
  return check_launch("k_cesr_f16");
}
