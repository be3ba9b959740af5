// NeuS SDF network (model/neus_model.py:385-438) with EXACT fp32 operands on the f16 matrix pipe, TWO 16-row tiles per wave -- the value
// pass of the default precision policy, round 4 (x6t_engine.h has the machine; sdf_x6.hip is round 3's one-tile kernel, the arithmetic
// and the stream layout are its own: three-piece operands, six products, one fp32 accumulator per weight class, the nine layers as ONE
// cyclic stream of 142 / 126 chunks through a 4-slot LDS ring filled by LDS-DMA under counted waits).
//
// Persistent workgroups of four waves, rounds of 128 rows (wave w, tile t: rows 128 r + 64 t + 16 w ..).  Per chunk: s_waitcnt + s_barrier
// at its top, K / 32 / WK parts of twelve MFMA runs (WK = 2 / 4 / 3 k-blocks for K = 64 / 256 / 288), the fragment window refilled piece
// by piece, the copies of the chunk three ahead FIRST (a store of the epilogue is then younger than its chunk's copies and has two
// chunks to retire in the shared in-order vmcnt queue) and the previous chunk's epilogue (softplus + exact three-way split of four value
// pairs: thirteen items) over the runs that carry no refill; a hidden layer's LAST chunk is finished beside the next layer's first
// chunk, written straight into that layer's own operands.  The biases of all chunks are resident in the LDS (9 KB), the encoded rows of
// a round too (32 KB: the skip layer's input part is rebuilt from them, the operand registers are all taken); outputs and sigmoid tiles
// leave through per-round buffer descriptors (one address register, lanes out of range dropped: no branch splits a chunk); the lane
// offset of a copy is re-derived per copy (x6t_engine.h: xt_lane16).
// MODE 0: signed distance only -> out0[M].   MODE 1: all 257 outputs -> out0[M,257].   MODE 5: MODE 1 + sigmoid(100 z) of every hidden
// pre-activation -> sig [tile = row / 16][layer 8][chunk 16][lane 64] float4, the layout k_sdf_back_* read.
// The products of a class are summed part by part, not product by product over the whole chunk as in k_sdf_x6: another fp32 summation
// order, results differ from the one-tile kernel's in the last bits.
#include "../../include/robir_hip.h"
#include "common.h"
#include "mlp_engine.h"
#include "x6_ring.h"
#include "sdf_x6_layout.h"
#include "x6t_engine.h"
#include <type_traits>

#ifndef SXT_FINE
#define SXT_FINE 0          // 1: hidden 256 x 256 layers (the instance of layers 1, 2, 5, 6) in the K-MAJOR form with the epilogue cut into single-instruction steps, two behind every MFMA (x6t_engine.h: xt_chunk_km)
#endif
#ifndef SXT_STORE128
#define SXT_STORE128 0      // 1: one 16-byte sigmoid store per tile and chunk instead of two 8-byte ones: measured equal (11.14 vs 11.09 ms value + gradient), more spills
#endif

namespace rb {

template <int MODE>
__global__ __launch_bounds__(256, 1) void k_sdf_x6t(const float* __restrict__ xyz, float in_scale, long M, const f4* __restrict__ Wp,
                                                     float out_scale, float* __restrict__ out0, f4* __restrict__ sig,
                                                     unsigned* __restrict__ range_word) {
  constexpr bool FULL = MODE != 0, STORE = MODE == 5;
  constexpr int LAST = FULL ? 17 : 1, NCHUNK = sx_nchunk(LAST);
  __shared__ f4 ring[4 * SX_SLOT_B / 16];              // 110 KB
  __shared__ f4 bias_tab[NCHUNK * 4];                  // 9 KB: the 16 biases of every chunk of the stream
  __shared__ float pe_scratch[4 * 2 * 16 * 64];        // 32 KB: [wave][tile][row 16][64 encoded inputs]
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long nrounds = (M + 127) >> 7;
  if ((long)blockIdx.x >= nrounds) return;

  const float negk = -2048.0f, inv_sqrt2 = 0.70710678118654752440f;
  constexpr float C11 = 1.0f / 2048.0f;
  const unsigned ring_b = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)ring);
  unsigned ring_lane = ring_b + (unsigned)lane * 16u;                  // + slot + fragment offset: the fragment reads
  asm volatile("" : "+v"(ring_lane));
  // first 1 KB piece of this wave's span of a chunk copy, per K of the stream
  const int first64 = xt_span_first(64, wave), first256 = xt_span_first(256, wave), first288 = xt_span_first(288, wave);
  auto span_first = [&](int K_) { return K_ == 64 ? first64 : (K_ == 256 ? first256 : first288); };
  __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(out0, 0, 0, 0x00020000);
  __amdgpu_buffer_rsrc_t sig_rsrc = __builtin_amdgcn_make_buffer_rsrc(sig, 0, 0, 0x00020000);
  const int sig_wave = wave * (8 * 16 * 64 * 16);        // tile (4 t + wave) of the round: wave and t parts go into the scalar offset
  unsigned slot_b[4] = {0u, (unsigned)SX_SLOT_B, 2u * SX_SLOT_B, 3u * SX_SLOT_B};
  unsigned sat = 0u;
  XtOps<9> P;                          // operands of the current layer (K <= 288), two tiles
  XtOps<8> Q;                          // ... of the next layer
  XtWin win;
  // epilogue state that crosses a layer boundary: the LAST chunk of a hidden layer is finished beside the first chunk of the next layer
  // (its softplus + split would otherwise run with the matrix pipe idle: ~1400 cycles per layer, 5 % of a round)
  SxAcc prev[2];
  float z[2][4];
  const int rlocal = wave * 16 + (lane & 15);      // row of a tile's lane inside its half-round: row = 128 round + 64 tile + rlocal
  long round = 0;
  auto row_of = [&](int t) { return round * 128 + t * 64 + rlocal; };

  for (int i = tid; i < NCHUNK * 4; i += 256) bias_tab[i] = Wp[sx_coff(i >> 2, LAST) + (i & 3)];

  // range sentinel: `sat` in the domain of sat_acc_nonneg (softplus outputs are >= 0: the raw pattern of the h piece orders like the value,
  // one instruction per pair); the signed inputs of a round (encoded rows) go through sat_acc into `sat_in`, folded into `sat` behind them
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
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float x0[16];
      load_features_pe10<false>(xyz, in_scale, row_of(t), M, lane, pe_scratch + (wave * 2 + t) * 1024, x0);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = (2 * kb + (q >> 1)) * 4 + (q & 1) * 2;
          put_pair(x0[i], x0[i + 1], P.h[t][kb], P.m[t][kb], P.l[t][kb], q, std::false_type{});
        }
    }
    fold_sat_in();
  };
  // skip layer operands [softplus(h3) / sqrt 2 (13 blocks of 16) | x0 / sqrt 2 (4 blocks) | 0]: blocks 13..17 = k-block 6 second half .. 8,
  // rebuilt from the round's encoded rows in the LDS (rows beyond M are zero like their first-layer operands)
  auto load_skip_part = [&]() {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const bool ok = row_of(t) < M;
      const f4* fr4 = reinterpret_cast<const f4*>(pe_scratch + (wave * 2 + t) * 1024 + (lane & 15) * 64) + g;
#pragma unroll
      for (int b = 13; b < 18; ++b) {
        f4 v = f4{0.f, 0.f, 0.f, 0.f};
        if (b < 17) v = fr4[(b - 13) * 4];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const int q = (b & 1) * 2 + p;
          const float v0 = ok ? v[2 * p] : 0.f, v1 = ok ? v[2 * p + 1] : 0.f;
          put_pair(v0 * inv_sqrt2, v1 * inv_sqrt2, P.h[t][b >> 1], P.m[t][b >> 1], P.l[t][b >> 1], q, std::false_type{});
        }
      }
    }
    fold_sat_in();
  };

  auto run_layer = [&](auto LI_tag, int cb, int lrt) {
    constexpr int LI = decltype(LI_tag)::value;
    constexpr int K = sx_K(LI), NCH = sx_nch(LI, LAST), CB = sx_cbase(LI, LAST);
    constexpr int WK = xt_wk(K), NPART = xt_parts(K), NFREE = 9 * NPART;
    constexpr bool OUT = LI == 8, SKIPOUT = LI == 3;
    constexpr int KN = sx_K(LI == 8 ? 0 : LI + 1), WKN = xt_wk(KN);
    static_assert((NCH * NPART) % 2 == 0, "a layer has an even number of parts");
    constexpr int PNCH = sx_nch(LI == 0 ? 8 : LI - 1, LAST);      // chunks of the layer that runs before this one (layer 8 of the previous round)
    // stores of the epilogue of chunk pj (pj < 0: no such chunk) of an output / a hidden layer
    auto ep_stores = [](bool out_layer, int pj) {
      if (pj < 0) return 0;
      if (out_layer) return FULL ? (pj < 16 ? 8 : 2) : (pj == 0 ? 2 : 0);
      return STORE ? (SXT_STORE128 ? 2 : 4) : 0;
    };
    constexpr bool FINE = SXT_FINE != 0 && LI == 1;      // K = 256, hidden, not the skip layer
    SxAcc accs[FINE ? 2 : 1][2];                          // FINE: chunk jb accumulates into set jb & 1 while the steps of chunk jb - 1 read the other
    constexpr bool PREV_SKIPOUT = LI == 4, PEND = LI != 0;      // the layer before this one: its outputs are scaled (layer 3) / it left its last chunk pending (every hidden layer)
    const f4* wl = Wp + sx_coff(cb, LAST);
    const f4* wnext[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) wnext[i] = Wp + sx_coff(cb + NCH + i, LAST);
    asm volatile("" : "+s"(wl));
    auto bias_of = [&](int c) {      // c = stream position (may run past the end once)
      const int cc = c >= NCHUNK ? c - NCHUNK : c;
      return bias_tab[cc * 4 + g];
    };
    auto combine = [&](const SxAcc& a, int r) { return __builtin_fmaf(__builtin_fmaf(a.c2[r], C11, a.c1[r]), C11, a.c0[r]); };
    // epilogue of hidden chunk pj, twelve items: per value pair (tile t, register pair q) A0 / A1 = softplus (+ its sigmoid in MODE 5) of
    // its two values, B = exact three-way split into the next layer's operand registers (+ the tile's sigmoid store behind its second pair)
    float ev[4][2];
    float sg[4][2];
    unsigned lv = 0u;              // lane offset of the running chunk's copies, carried from piece to piece (x6t_engine.h)
    bool ep_own_voff = true;       // the epilogue item that runs has no copy of this chunk in front of it
    int ep_voff_adj = 0;           // 4096 once piece 4 of the chunk's copies has advanced lv
    // Z = the three classes of a tile's four values combined (frees the previous chunk's accumulators early)
    auto item_z = [&](int t, const SxAcc& a) {
#pragma unroll
      for (int r = 0; r < 4; ++r) z[t][r] = combine(a, r);
    };
    // sk: the chunk's layer feeds the skip concatenation (its outputs / sqrt 2); lr: that layer's index (sigmoid tiles); Y: the operand
    // set the split writes (the next layer's -- or, for a chunk finished across the layer boundary, the running layer's own inputs)
    auto item_a = [&](int i, int e, int pj, bool sk, int lr) {
      const int t = i >> 1, q = i & 1;
      float s;
      float v = softplus100_stable(z[t][2 * q + e], &s);    // max(z, 0) + log(1 + exp(-|100 z|)) / 100 (mlp_engine.h)
      if (sk) v *= inv_sqrt2;
      ev[i][e] = v;
      sg[i][e] = s;
      if constexpr (STORE)
#ifdef SXT_ABL_HALF_STORES              // timing ablation (wrong results): only the first pair of a tile's float4 is stored
        if (e == 1 && q == 0) {
#else
        if (e == 1) {
#endif
          // [tile = row / 16][layer 8][chunk 16][lane 64] float4 (a pair = 8 bytes of it), through the round's descriptor
          typedef unsigned u2v __attribute__((ext_vector_type(2)));
          // lane offset: the one this chunk's copies carry (they are issued in front of the epilogue: lane 16, + 4 KB once piece 4
          // has gone), or re-derived where the epilogue runs in front of the copies (x6t_engine.h: no long-lived register)
          const unsigned voff = ep_own_voff ? xt_lane16<0>() : lv;
          int sbase = sig_wave + lr * (16 * 1024) - (ep_own_voff ? 0 : ep_voff_adj);      // formed here: hoisted out of the round loop the scalar offsets of a layer's 64 stores would not fit the SGPR file
          asm volatile("" : "+s"(sbase));
#ifdef SXT_ABL_STORE_FIXED              // timing ablation (wrong results): every sigmoid store of a wave goes to the same kilobyte
          sbase = sig_wave - (t * 4 * 8 * 16 + pj) * 1024 - q * 8 - (ep_own_voff ? 0 : ep_voff_adj);
#endif
#if SXT_STORE128                        // one 16-byte store per tile and chunk behind its second pair: whole lines, half the store instructions
          if (q == 1)
            __builtin_amdgcn_raw_buffer_store_b128(u4{__builtin_bit_cast(unsigned, sg[i - 1][0]), __builtin_bit_cast(unsigned, sg[i - 1][1]),
                                                      __builtin_bit_cast(unsigned, sg[i][0]), __builtin_bit_cast(unsigned, sg[i][1])},
                                                   sig_rsrc, (int)voff, sbase + (t * 4 * 8 * 16 + pj) * 1024, 0);
#else
          __builtin_amdgcn_raw_buffer_store_b64(u2v{__builtin_bit_cast(unsigned, sg[i][0]), __builtin_bit_cast(unsigned, sg[i][1])}, sig_rsrc,
                                                (int)voff, sbase + (t * 4 * 8 * 16 + pj) * 1024 + q * 8, 0);
#endif
        }
    };
    auto item_b = [&](int i, int pj, auto& Y) {
      const int t = i >> 1, q = i & 1;
      put_pair(ev[i][0], ev[i][1], Y.h[t][pj >> 1], Y.m[t][pj >> 1], Y.l[t][pj >> 1], (pj & 1) * 2 + q, std::true_type{});
    };
    // outputs through a buffer descriptor over the round's valid rows: lanes beyond it (rows >= M, columns >= 257) are dropped by the
    // bounds check, no branch splits the chunk
    auto output_tile = [&](int t, int pj) {
      // byte offset of (row, column 4 g) inside the round's rows: ONE register per tile, the chunk / register part of the column goes
      // into the instruction (17 x 4 x 2 loop-invariant offsets would be hoisted out of the round loop and spilled)
      int base = FULL ? ((t * 64 + rlocal) * 257 + 4 * g) * 4 : (t * 64 + rlocal) * 4;
      asm volatile("" : "+v"(base));
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = combine(prev[t], r) * out_scale;
        if constexpr (FULL) {
          if (pj < 16) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), out_rsrc, base, pj * 64 + r * 4, 0);
          else if (r == 0) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), out_rsrc, g == 0 ? base : -1, pj * 64, 0);
        } else {
          if (pj == 0 && r == 0) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), out_rsrc, g == 0 ? base : -1, 0, 0);
        }
      }
    };
    constexpr int NE = OUT ? 2 : 13;
    // item order: the second tile's combine (the first tile's went behind the chunk's own last run: its accumulators were complete), then
    // per pair the two softplus halves with the previous pair's split one pair behind
    auto hidden_item = [&](int s, int pj, bool sk, int lr, auto& Y) {
      switch (s) {
        case 0: item_z(1, prev[1]); break;
        case 1: item_a(0, 0, pj, sk, lr); break;
        case 2: item_a(0, 1, pj, sk, lr); break;
        case 3: item_a(1, 0, pj, sk, lr); break;
        case 4: item_a(1, 1, pj, sk, lr); break;
        case 5: item_b(0, pj, Y); break;
        case 6: item_a(2, 0, pj, sk, lr); break;
        case 7: item_a(2, 1, pj, sk, lr); break;
        case 8: item_b(1, pj, Y); break;
        case 9: item_a(3, 0, pj, sk, lr); break;
        case 10: item_a(3, 1, pj, sk, lr); break;
        case 11: item_b(2, pj, Y); break;
        default: item_b(3, pj, Y); break;
      }
    };
    // epilogue item s of this layer's chunk pj
    auto ep_item = [&](int s, int pj) {
      if constexpr (OUT) {
        if (s < 2) output_tile(s, pj);
      } else {
        hidden_item(s, pj, SKIPOUT, lrt, Q);
      }
    };
    f4 bias = bias_of(cb);
#pragma unroll
    for (int jb = 0; jb < NCH; ++jb) {
      // chunk jb+1 has landed once at most this wave's copies of chunk jb+2 are in flight (stores of the epilogue, younger, only make the
      // wait stricter); past the barrier every wave has finished with chunk jb-1, whose slot the copies of chunk jb+3 reuse
      constexpr int dummy = 0;
      (void)dummy;
      const int K2 = jb + 2 < NCH ? K : sx_K(sx_layer_of(CB + jb + 2, LAST));
      // ... plus the stores of the epilogue items that ran beside them (exact: every store is issued, lanes out of range are dropped by
      // the descriptor): the previous chunk carried the epilogue of the chunk before it; a layer's first chunk follows the previous
      // layer's last chunk AND the tail epilogue of that chunk
      int ns = 0;
      if (jb >= 3) ns = ep_stores(OUT, jb - 2) + ep_stores(OUT, jb - 3);      // copies first: the stores of two chunks may be in flight
      else if (jb == 2) ns = ep_stores(OUT, jb - 2);
      else if (jb == 1) ns = PEND ? ep_stores(false, 0) : 0;          // chunk 0 carried the previous layer's pending last chunk
      else ns = ep_stores(LI == 0, PNCH - 2) + (LI == 0 ? ep_stores(true, PNCH - 1) : 0);      // an output layer finishes its last chunk at once
      switch (sx_nsw(K2) + ns) {
        case 2: sx_wait<2>(); break;
        case 3: sx_wait<3>(); break;
        case 4: sx_wait<4>(); break;
        case 5: sx_wait<5>(); break;
        case 6: sx_wait<6>(); break;
        case 7: sx_wait<7>(); break;
        case 8: sx_wait<8>(); break;
        case 9: sx_wait<9>(); break;
        case 10: sx_wait<10>(); break;
        case 11: sx_wait<11>(); break;
        case 12: sx_wait<12>(); break;
        case 13: sx_wait<13>(); break;
        case 14: sx_wait<14>(); break;
        case 15: sx_wait<15>(); break;
        case 16: sx_wait<16>(); break;
        case 17: sx_wait<17>(); break;
        case 18: sx_wait<18>(); break;
        case 19: sx_wait<19>(); break;
        case 20: sx_wait<20>(); break;
        case 21: sx_wait<21>(); break;
        case 22: sx_wait<22>(); break;
        case 23: sx_wait<23>(); break;
        case 24: sx_wait<24>(); break;
        default: sx_wait<2>(); break;      // any combination not listed: the strictest wait (safe)
      }
#ifndef SXT_NOBAR
      __builtin_amdgcn_s_barrier();
#endif
      asm volatile("" ::: "memory");
      const int K3 = jb + 3 < NCH ? K : sx_K(sx_layer_of(CB + jb + 3, LAST));
      const int NC3 = sx_nsw(K3);
      const int f3 = span_first(K3);
      const f4* src3 = (jb + 3 < NCH ? wl + (long)(jb + 3) * sx_cf4(K) : wnext[jb + 3 - NCH < 3 ? jb + 3 - NCH : 0]) + 4 + f3 * 64;
      const unsigned dst3 = ring_b + slot_b[(jb + 3) & 3] + (unsigned)f3 * 1024u;
      f4 nbias;
      SxAcc(&acc)[2] = accs[FINE ? (jb & 1) : 0];
      acc[0].c0 = bias;
      acc[1].c0 = bias;
      acc[0].c1 = acc[0].c2 = acc[1].c1 = acc[1].c2 = f4{0.f, 0.f, 0.f, 0.f};
      if constexpr (FINE) {
        if (jb >= 1) {
          // ---- K-major chunk: the epilogue of chunk jb - 1 (both tiles' accumulators are still in the other set) as NM single-instruction
          // steps, two behind every MFMA from slot NC3 on (the copies take the first NC3 slots: their stores / the next wait see the same
          // order as in the coarse form).  Per tile: combine (2 x 4), -|z| k, exp2, 1 + e, log2, max(z, 0), fma (4 each) [, sigmoid: select,
          // rcp, mul (4 each) + the tile's two 8-byte stores], then the exact three-way split of its two value pairs, interleaved step by
          // step (a step's input was produced >= 2 steps = one MFMA slot earlier).
          const SxAcc(&pa)[2] = accs[(jb - 1) & 1];
          constexpr int NSIG = STORE ? 14 : 0, NT = 32 + NSIG + 22, NM = 2 * NT;
          float zc[4], tt[4], uu[4], lg[4], mx[4], vv[4], sgm[4];
          unsigned hu[2], mu[2], lu[2];
          float s0[2], s1[2], d0[2], d1[2], e0[2], e1[2];
          const int pj = jb - 1;
          auto micro = [&](int sidx) {
            const int t = sidx / NT, u = sidx % NT;
            const SxAcc& a = pa[t];
            if (u < 4) zc[u] = __builtin_fmaf(a.c2[u], C11, a.c1[u]);
            else if (u < 8) zc[u - 4] = __builtin_fmaf(zc[u - 4], C11, a.c0[u - 4]);
            else if (u < 12) tt[u - 8] = -__builtin_fabsf(zc[u - 8]) * SP_T_PER_Z;
            else if (u < 16) tt[u - 12] = __builtin_amdgcn_exp2f(tt[u - 12]);
            else if (u < 20) uu[u - 16] = 1.0f + tt[u - 16];
            else if (u < 24) lg[u - 20] = __builtin_amdgcn_logf(uu[u - 20]);
            else if (u < 28) mx[u - 24] = __builtin_fmaxf(zc[u - 24], 0.0f);
            else if (u < 32) vv[u - 28] = __builtin_fmaf(lg[u - 28], SP_LN2_OVER_100, mx[u - 28]);
            else if (u < 32 + NSIG) {
              if constexpr (STORE) {
                const int w_ = u - 32;
                if (w_ < 4) sgm[w_] = zc[w_] > 0.0f ? 1.0f : tt[w_];
                else if (w_ < 8) uu[w_ - 4] = __builtin_amdgcn_rcpf(uu[w_ - 4]);
                else if (w_ < 12) sgm[w_ - 8] = sgm[w_ - 8] * uu[w_ - 8];
                else {
                  const int q = w_ - 12;
                  typedef unsigned u2v __attribute__((ext_vector_type(2)));
                  const unsigned voff = lv;                      // the chunk's copies are in front of every step: lane 16 (+ 4 KB once piece 4 has gone)
                  int sbase = sig_wave + lrt * (16 * 1024) - ep_voff_adj;
                  asm volatile("" : "+s"(sbase));
                  __builtin_amdgcn_raw_buffer_store_b64(u2v{__builtin_bit_cast(unsigned, sgm[2 * q]), __builtin_bit_cast(unsigned, sgm[2 * q + 1])}, sig_rsrc,
                                                        (int)voff, sbase + (t * 4 * 8 * 16 + pj) * 1024 + q * 8, 0);
                }
              }
            } else {
              const int v_ = u - 32 - NSIG, q = v_ & 1, st = v_ >> 1;
              const float v0 = vv[2 * q], v1 = vv[2 * q + 1];
              if (st == 0) hu[q] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v0, v1));
              else if (st == 1) s0[q] = v0 * 2048.0f;
              else if (st == 2) s1[q] = v1 * 2048.0f;
              else if (st == 3) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d0[q]) : "v"(hu[q]), "s"(negk), "v"(s0[q]));
              else if (st == 4) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d1[q]) : "v"(hu[q]), "s"(negk), "v"(s1[q]));
              else if (st == 5) mu[q] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(d0[q], d1[q]));
              else if (st == 6) e0[q] = d0[q] * 2048.0f;
              else if (st == 7) e1[q] = d1[q] * 2048.0f;
              else if (st == 8) asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(lu[q]) : "v"(mu[q]), "s"(negk), "v"(e0[q]));
              else if (st == 9) asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lu[q]) : "v"(mu[q]), "s"(negk), "v"(e1[q]));
              else {
                sat = sat_acc_pos(sat, hu[q]);
                const int kq = (pj & 1) * 2 + q;
                Q.h[t][pj >> 1][kq] = hu[q];
                Q.m[t][pj >> 1][kq] = mu[q];
                Q.l[t][pj >> 1][kq] = lu[q];
              }
            }
          };
          ep_voff_adj = NC3 > 4 ? 4096 : 0;
          auto filler_km = [&](int slot) {
            if (slot == 12 * WK * (NPART - 1)) nbias = bias_of(cb + jb + 1);      // before the last part's fragment requests (see the coarse form)
            if (slot < NC3) {
#ifndef SXT_NODMA
              xt_copy_piece_seq(slot, src3, dst3, lv);
#endif
              return;
            }
            const int m0 = 2 * (slot - NC3);
            if (m0 < NM) micro(m0);
            if (m0 + 1 < NM) micro(m0 + 1);
          };
          auto refill_km = [&](int piece, int part, int k) {
            int slot, kb;
            bool ok = true;
            if (part + 1 < NPART) {
              slot = jb & 3, kb = (part + 1) * WK + k;
            } else {
              slot = (jb + 1) & 3, kb = k, ok = k < (jb + 1 < NCH ? WK : WKN);
            }
#ifndef SXT_NOREAD
            if (ok) xt_request_one(piece == 0 ? win.h[k] : (piece == 1 ? win.m[k] : win.l[k]), ring_lane + slot_b[slot], kb, piece);
#endif
          };
          xt_chunk_km<K, 9>(acc, win, P, filler_km, refill_km);
          static_assert((12 * K / 32 - 8) * 2 >= NM, "the steps fit behind the chunk's MFMAs (at most 8 copy slots in front)");
          if (jb == NCH - 1) {      // the layer's last chunk is finished beside the next layer's first chunk: hand it over in the coarse form
            item_z(0, acc[0]);
            prev[1] = acc[1];
          }
          bias = nbias;
          continue;
        }
      }
      const int ne = jb > 0 ? NE : 0, ni = ne + NC3;
      ep_voff_adj = NC3 > 4 ? 4096 : 0;
      auto filler = [&](int pos) {
        // the next chunk's bias, requested BEFORE the last part's fragment requests: behind them, its use at the top of the next chunk
        // would wait for all of them (lgkmcnt(0))
        if (pos == 12 * (NPART - 1)) nbias = bias_of(cb + jb + 1);
        const int a = xt_free_index(pos);
        if (a < 0) return;
        if (jb == 0 && PEND) {
          // the previous layer's last chunk, finished here: its outputs are this layer's operands of the LAST k-blocks, which the
          // second part reads from its first run on -- all thirteen items go into the first part, the copies behind them
#pragma unroll
          for (int i = 0; i < 13; ++i)
            if (xt_item_slot(i, 13, 9) == a) {
              ep_own_voff = true;
              hidden_item(i, PNCH - 1, PREV_SKIPOUT, lrt - 1, P);      // straight into this layer's own operands
            }
#pragma unroll
          for (int i = 0; i < 8; ++i)
            if (i < NC3 && 9 + xt_item_slot(i, NC3, NFREE - 9) == a) {
#ifndef SXT_NODMA
              xt_copy_piece_seq(i, src3, dst3, lv);
#endif
            }
          return;
        }
        // the copies take the chunk's first free positions, the epilogue items the rest: a store of the epilogue is then younger than
        // the copies of its own chunk and has two chunks to retire before a counted wait needs it, not one (stores and copies share
        // the in-order vmcnt queue; the sigmoid tiles go to HBM)
#pragma unroll
        for (int i = 0; i < 22; ++i)
          if (i < ni && (i < ne ? NC3 + xt_item_slot(i, ne > 0 ? ne : 1, NFREE - NC3) : i - ne) == a) {
            if (i < ne) {
              ep_own_voff = NC3 == 0;
              if (jb > 0) ep_item(i, jb - 1);
            } else {
#ifndef SXT_NODMA
              xt_copy_piece_seq(i - ne, src3, dst3, lv);
#endif
            }
          }
      };
      auto refill = [&](int piece, int part) {
        const bool down = ((jb * NPART + part) & 1) != 0;
        int slot, kb_first, count;
        if (part + 1 < NPART) {
          slot = jb & 3, kb_first = (part + 1) * WK, count = WK;
        } else {
          slot = (jb + 1) & 3, kb_first = 0, count = jb + 1 < NCH ? WK : WKN;
        }
#ifndef SXT_NOREAD
        xt_request(piece == 0 ? win.h : (piece == 1 ? win.m : win.l), ring_lane + slot_b[slot], kb_first, count, piece, down);
#endif
      };
      xt_chunk<K, 9>(jb * NPART, acc, win, P, filler, refill);
      if constexpr (FINE) {
        // chunk 0 stays in set 0: the steps beside chunk 1 read it there
      } else {
        if constexpr (OUT) prev[0] = acc[0];
        else item_z(0, acc[0]);
        prev[1] = acc[1];
      }
      bias = nbias;
    }
    {   // slot 0 = the slot of the next layer's first chunk
      constexpr int R = NCH & 3;
      unsigned a[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = slot_b[(i + R) & 3];
#pragma unroll
      for (int i = 0; i < 4; ++i) slot_b[i] = a[i];
    }
    if constexpr (OUT) {      // hidden layers leave their last chunk (z[0], prev[1]) to the next layer's first chunk
#pragma unroll
      for (int s = 0; s < NE; ++s) ep_item(s, NCH - 1);
    }
    if constexpr (!OUT) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
          if (SKIPOUT && kb == 6) {          // second half of k-block 6 and blocks 7, 8: load_skip_part
            P.h[t][kb][0] = Q.h[t][kb][0], P.h[t][kb][1] = Q.h[t][kb][1];
            P.m[t][kb][0] = Q.m[t][kb][0], P.m[t][kb][1] = Q.m[t][kb][1];
            P.l[t][kb][0] = Q.l[t][kb][0], P.l[t][kb][1] = Q.l[t][kb][1];
          } else if (!(SKIPOUT && kb > 6)) {
            P.h[t][kb] = Q.h[t][kb];
            P.m[t][kb] = Q.m[t][kb];
            P.l[t][kb] = Q.l[t][kb];
          }
        }
      if constexpr (SKIPOUT) load_skip_part();
    }
  };

  // ---- prologue: chunks 0, 1, 2 of the stream (layer 0: K = 64, two pieces per wave), the first fragment window
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int i = 0; i < 2; ++i)
      xt_copy_piece(i, Wp + sx_coff(c, LAST) + 4 + first64 * 64, ring_b + slot_b[c] + (unsigned)first64 * 1024u);
  sx_wait<0>();
  __syncthreads();
  xt_request(win.h, ring_lane + slot_b[0], 0, 2, 0, true);
  xt_request(win.m, ring_lane + slot_b[0], 0, 2, 1, true);
  xt_request(win.l, ring_lane + slot_b[0], 0, 2, 2, true);

#ifdef SXT_TIMING   // phase stamps per round (tools/build_variant.sh ... -DSXT_TIMING; profiles/r05_sdf_phases.md)
#define SXT_T(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tacc[i] += t_ - tlast; tlast = t_; }
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
  int trounds = 0;
#else
#define SXT_T(i)
#endif
  for (round = blockIdx.x; round < nrounds; round += gridDim.x) {
#ifdef SXT_TIMING
    ++trounds;
#endif
    {
      const long row0 = round * 128, rows = M - row0 < 128 ? M - row0 : 128;
      out_rsrc = __builtin_amdgcn_make_buffer_rsrc(out0 + row0 * (FULL ? 257 : 1), 0, (int)rows * (FULL ? 257 : 1) * 4, 0x00020000);
      if constexpr (STORE) sig_rsrc = __builtin_amdgcn_make_buffer_rsrc(sig + round * (8L * 8 * 16 * 64), 0, 8 * 8 * 16 * 64 * 16, 0x00020000);
    }
    load_layer0();
    SXT_T(0)
    // layer 0 | 1, 2 (one instance) | 3 (skip layer's own outputs) | 4 (K = 288) | 5, 6 (a second copy of the instance of 1, 2) | 7 | 8:
    // straight-line, so that the skip layer's ninth k-block is live between layers 3 and 4 only (inside one loop over all nine layers it is
    // carried through every iteration: 24 registers this kernel does not have)
    run_layer(std::integral_constant<int, 0>{}, 0, 0);
    SXT_T(1)
#pragma unroll 1
    for (int l = 1; l < 3; ++l) run_layer(std::integral_constant<int, 1>{}, 16 * l, l);
    SXT_T(2)
    run_layer(std::integral_constant<int, 3>{}, 48, 3);
    SXT_T(3)
    run_layer(std::integral_constant<int, 4>{}, 61, 4);
    SXT_T(4)
#pragma unroll 1
    for (int l = 5; l < 7; ++l) run_layer(std::integral_constant<int, 1>{}, 16 * (l - 1) + 13, l);
    SXT_T(5)
    run_layer(std::integral_constant<int, 7>{}, 109, 7);
    SXT_T(6)
    run_layer(std::integral_constant<int, 8>{}, 125, 8);
    SXT_T(7)
  }
#ifdef SXT_TIMING
  if ((blockIdx.x == 3 || blockIdx.x == 200) && tid == 0)
    printf("sdf_x6t<%d> wg %d rounds %d cycles/round: load+encode %llu L0 %llu L1-2 %llu L3 %llu L4 %llu L5-6 %llu L7 %llu L8 %llu\n", MODE, (int)blockIdx.x,
           trounds, tacc[0] / trounds, tacc[1] / trounds, tacc[2] / trounds, tacc[3] / trounds, tacc[4] / trounds, tacc[5] / trounds,
           tacc[6] / trounds, tacc[7] / trounds);
#endif
  range_report<false>(sat, range_word);
  sx_wait<0>();
  __syncthreads();
}

}  // namespace rb

using namespace rb;

namespace rb {
int launch_sdf_x6t(const float* x, long M, float in_scale, const float* Wp, int mode, float out_scale, float* out0, float* sig,
                   int n_workgroups, hipStream_t s) {
  const int grid = persistent_grid((M + 127) / 128, n_workgroups);
  if (grid <= 0) return rb::fail("rb_sdf_x6_points", "device query failed");
  unsigned* rw = range_flags() ? range_flags() + RB_RANGE_SDF : nullptr;
  const f4* W = (const f4*)Wp;
  switch (mode) {
    case 0: hipLaunchKernelGGL(k_sdf_x6t<0>, dim3(grid), dim3(256), 0, s, x, in_scale, M, W, out_scale, out0, (f4*)nullptr, rw); break;
    case 1: hipLaunchKernelGGL(k_sdf_x6t<1>, dim3(grid), dim3(256), 0, s, x, in_scale, M, W, out_scale, out0, (f4*)nullptr, rw); break;
    default: hipLaunchKernelGGL(k_sdf_x6t<5>, dim3(grid), dim3(256), 0, s, x, in_scale, M, W, out_scale, out0, (f4*)sig, rw); break;
  }
  return check_launch("k_sdf_x6t");
}
}  // namespace rb
