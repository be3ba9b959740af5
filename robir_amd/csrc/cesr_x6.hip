// CESR normal_net / shadow_net (model/cesr_net.py: eight softplus(100) layers of 512 with a skip connection into layer 4; inputs PE10(x)
// or [PE10(x) | one-hot label], training/train_cesr.py:331-352) with EXACT fp32 operands on the f16 matrix pipe ("f16x6") -- the default
// precision policy's kernels for the CESR hook, round 3.  The half-chunk stream of wide_x6.hip (an exact-operand chunk of K = 512 is
// 48 KB: the stream moves halves of 24 KB -- 27 KB for the skip layer, padded to K = 576 -- and a chunk's three accumulators run across
// its two halves) with the softplus epilogue of sdf_x6.hip; the skip layer's input part [x0 | 0] / sqrt 2 is rebuilt at that layer from
// the encoder's LDS rows and the label (operands + next operands already take 408 of the 512 registers: nothing else stays resident).
// Replaces k_softplus512 (f32-input MFMA: 41 % of BASELINE config 5 at the default precision).  Weights: packing.pack_softplus512_x6.
#include "../../include/robir_hip.h"
#include "common.h"
#include "mlp_engine.h"
#include "x6_ring.h"
#include "x6t_engine.h"
#include <type_traits>

#ifndef QX_FP8
#define QX_FP8 0            // 1 (needs QX_FINE): EXPERIMENT -- in the layers whose units are K = 256 the products h.xl and l.xh as bf8 MFMAs (vis_diffuse_x6t.hip XT_FP8)
#endif
#ifndef QX_FINE
#define QX_FINE 1           // a filler slot behind EVERY MFMA, the previous chunk's softplus + split as single-instruction steps (0: two clusters per chunk; -1.3 % / -2 %, bit-identical: profiles/r06_dma_placement.md)
#endif
#ifndef QX_SPREAD
#define QX_SPREAD 1         // the LDS-DMA copies of a unit one at a time, three MFMAs apart (0: blocks of 1 / 4 / 2 instructions; profiles/r06_dma_placement.md: -2.4 %)
#endif

static_assert(!QX_FP8 || QX_FINE, "QX_FP8 is written into the QX_FINE form of the k-block");

namespace rb {

constexpr int QX_SLOT_B = 27 * 1024 + 512;
template <int K0P, int N3P>
struct QxNet {
  static_assert(K0P + N3P == 528 && (K0P == 64 || K0P == 192), "normal_net (64, 464) or shadow_net (192, 336)");
  // per layer: K of a UNIT (what one ring slot holds), units per chunk, chunks
  __host__ __device__ static constexpr int kunit(int l) { return l == 0 ? K0P : (l == 4 ? 288 : 256); }
  __host__ __device__ static constexpr int hv(int l) { return l == 0 ? 1 : 2; }
  __host__ __device__ static constexpr int nch(int l) { return l == 3 ? N3P / 16 : (l == 8 ? 1 : 32); }
  __host__ __device__ static constexpr int nunits(int l) { return hv(l) * nch(l); }
  __host__ __device__ static constexpr long chunk_f4s(int l) { return sx_cf4(kunit(l) * hv(l)); }
  __host__ __device__ static constexpr int total() {
    int n = 0;
    for (int l = 0; l < 9; ++l) n += nunits(l);
    return n;
  }
  __host__ __device__ static constexpr int ubase(int l) {
    int n = 0;
    for (int i = 0; i < l; ++i) n += nunits(i);
    return n;
  }
  __host__ __device__ static constexpr int layer_of(int u) {
    if (u >= total()) u -= total();
    int l = 0, first = 0;
    for (int i = 0; i < 8; ++i) {
      first += nunits(i);
      if (u >= first) l = i + 1;
    }
    return l;
  }
  __host__ __device__ static constexpr long uoff(int u) {      // chunk head (+ the half's offset) of stream unit u, in float4
    if (u >= total()) u -= total();
    const int l = layer_of(u), r = u - ubase(l);
    long off = 0;
    for (int i = 0; i < l; ++i) off += (long)nch(i) * chunk_f4s(i);
    return off + (long)(r / hv(l)) * chunk_f4s(l) + (long)(r % hv(l)) * (kunit(l) / 32) * 192;
  }
};
template <int KU>
__device__ __forceinline__ void qx_copy(int un, const f4* src, unsigned lane4, unsigned lane16, unsigned bdst, unsigned dst, int wave) {
  sx_copy_unit<KU>(un, src, lane4, lane16, bdst, dst, wave);
}

// ONEHOT: rows = (point, label) pairs, row = point * n_label + label (shadow_net); else one row per point (normal_net)
template <int K0P, int N3P, bool ONEHOT>
__global__ __launch_bounds__(256, 1) void k_cesr_x6(const float* __restrict__ X, long M, int n_label, const f4* __restrict__ Wp, int n_out,
                                                     float* __restrict__ Y, unsigned* __restrict__ range_word) {
  using Net = QxNet<K0P, N3P>;
  __shared__ f4 ring[4 * QX_SLOT_B / 16];              // 110 KB
  __shared__ f4 bias_ring[4 * 16];
  __shared__ float pe_scratch[4 * 16 * 64];            // 16 KB: the encoded rows of the round (read again at the skip layer)
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long nrounds = (M + 63) >> 6;
  if ((long)blockIdx.x >= nrounds) return;

  const float negk = -2048.0f, inv_sqrt2 = 0.70710678118654752440f;
  constexpr float C11 = 1.0f / 2048.0f;
  const unsigned ring_b = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)ring);
  const unsigned bias_b = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)bias_ring);
  const unsigned lane16 = (unsigned)lane * 16u, lane4 = (unsigned)lane * 4u;
  unsigned slot_b[4] = {0u, (unsigned)QX_SLOT_B, 2u * QX_SLOT_B, 3u * QX_SLOT_B};
  unsigned bslot_b[4] = {0u, 256u, 512u, 768u};
  unsigned sat = 0u;
  u4 xh[18], xm[18], xl[18];           // operands of the current layer (K <= 576): three pieces, one tile
  u4 yh[16], ym[16], yl[16];           // ... of the next layer
#if QX_FP8
  typedef int qx_i8 __attribute__((ext_vector_type(8)));
  qx_i8 xh8[4], xl8[4], yh8[4], yl8[4];      // bf8 copies of the h and l pieces for the layers that run two products on the bf8 MFMA (K = 512: four groups of 128)
#endif
  long rrow = 0;
  int label = -1;

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
  // input value k of this lane's row (k = 16 blk + 4 g + r): the encoder's LDS row, then the one-hot block from column 63 on
  auto x0_block = [&](int blk, float scale, float (&v)[4]) {
    const bool ok = rrow < M;
    const float* frow = pe_scratch + wave * 1024 + (lane & 15) * 64;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = blk * 16 + 4 * g + r;
      float e = 0.f;
      if (blk < 4) e = frow[blk * 16 + 4 * g + r];
      if (ONEHOT) {
        if (k == 63) e = 0.f;
        if (k >= 63 && k - 63 == label) e = 1.f;
      }
      v[r] = ok ? e * scale : 0.f;
    }
  };
  auto load_layer0 = [&]() {
    float enc[16];
    load_features_pe10x(X, nullptr, rrow, M, lane, pe_scratch + wave * 1024, enc, ONEHOT ? (long)n_label : 1L);
    (void)enc;
    label = (ONEHOT && rrow < M) ? (int)(rrow % n_label) : -1;
#pragma unroll
    for (int kb = 0; kb < K0P / 32; ++kb)
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        float v[4];
        x0_block(2 * kb + hb, 1.0f, v);
        put_pair(v[0], v[1], xh[kb], xm[kb], xl[kb], hb * 2, std::false_type{});
        put_pair(v[2], v[3], xh[kb], xm[kb], xl[kb], hb * 2 + 1, std::false_type{});
      }
    fold_sat_in();
  };
  // the skip layer's operands: [softplus(h3) / sqrt 2 (in y) | x0 / sqrt 2 | 0]
  auto build_skip_operands = [&]() {
    constexpr int B3 = N3P / 16;          // 16-blocks of the layer's own part (odd: the next block shares a k-block with the last)
    static_assert((B3 & 1) == 1, "");
#pragma unroll
    for (int kb = 0; kb < 18; ++kb) {
      if (2 * kb + 1 < B3) {
        xh[kb] = yh[kb < 16 ? kb : 0];
        xm[kb] = ym[kb < 16 ? kb : 0];
        xl[kb] = yl[kb < 16 ? kb : 0];
      } else {
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
          const int b = 2 * kb + hb;
          if (b < B3) {
            xh[kb][0] = yh[kb < 16 ? kb : 0][0];
            xh[kb][1] = yh[kb < 16 ? kb : 0][1];
            xm[kb][0] = ym[kb < 16 ? kb : 0][0];
            xm[kb][1] = ym[kb < 16 ? kb : 0][1];
            xl[kb][0] = yl[kb < 16 ? kb : 0][0];
            xl[kb][1] = yl[kb < 16 ? kb : 0][1];
          } else if (b < B3 + K0P / 16) {
            float v[4];
            x0_block(b - B3, inv_sqrt2, v);
            put_pair(v[0], v[1], xh[kb], xm[kb], xl[kb], hb * 2, std::false_type{});
            put_pair(v[2], v[3], xh[kb], xm[kb], xl[kb], hb * 2 + 1, std::false_type{});
          } else {
            xh[kb][hb * 2] = xh[kb][hb * 2 + 1] = 0u;
            xm[kb][hb * 2] = xm[kb][hb * 2 + 1] = 0u;
            xl[kb][hb * 2] = xl[kb][hb * 2 + 1] = 0u;
          }
        }
      }
    }
  };

  auto run_layer = [&](auto LI_tag, int ub) {
    constexpr int LI = decltype(LI_tag)::value;
    constexpr int KU = Net::kunit(LI), KB = KU / 32, HV = Net::hv(LI), NU = Net::nunits(LI), UB = Net::ubase(LI);
    constexpr bool OUT = LI == 8, SKIPOUT = LI == 3;
    constexpr bool FP8 = QX_FP8 && KU == 256;                       // this layer's weights are in the bf8 layout (packing.repack_x6_chunks_fp8)
    constexpr bool NEXT_FP8 = QX_FP8 && LI != 3 && LI != 8;         // ... and so are the next layer's: its operands get bf8 copies instead of f16 l pieces
#ifndef QX_DB
#define QX_DB 1              // fragment sets read ahead per k-block: 2 measured 1 % slower (and 66 instead of 42 spilled registers in the shadow_net instance)
#endif
    constexpr int BS = 1, DB = KB >= 6 ? QX_DB : 1, D = BS * DB, NB = BS * (DB + 1);      // three fragment sets: the register file is full
    constexpr int HB = KB / 2, NSTEP = NU * KB;
    static_assert(D + BS - 1 <= KB - HB, "reads of the next unit start after the barrier");
    SxAcc accs[2];
    f4 bnext = f4{0.f, 0.f, 0.f, 0.f};
    u4 wfh[NB], wfm[NB], wfl[NB];
    const f4* wnext[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) wnext[i] = Wp + Net::uoff(ub + NU + i);
    const f4* wl = Wp + Net::uoff(ub);
    asm volatile("" : "+s"(wl));
    auto frag_of = [&](int u) { return reinterpret_cast<const u4*>(reinterpret_cast<const char*>(ring) + slot_b[u & 3]) + lane; };
    auto bias_of = [&](int u) { return *(reinterpret_cast<const f4*>(reinterpret_cast<const char*>(bias_ring) + bslot_b[u & 3]) + g); };
    auto zero_acc = [&](SxAcc& a, const f4& b) {
      a.c0 = b;
      a.c1 = f4{0.f, 0.f, 0.f, 0.f};
      a.c2 = f4{0.f, 0.f, 0.f, 0.f};
    };
    auto combine = [&](const SxAcc& a, int r) { return __builtin_fmaf(__builtin_fmaf(a.c2[r], C11, a.c1[r]), C11, a.c0[r]); };
    auto hidden_pair = [&](const SxAcc& a, int pj, int q) {
      float v0 = softplus100_stable(combine(a, 2 * q), nullptr), v1 = softplus100_stable(combine(a, 2 * q + 1), nullptr);
      if (SKIPOUT) {
        v0 *= inv_sqrt2;
        v1 *= inv_sqrt2;
      }
      put_pair(v0, v1, yh[pj >> 1], ym[pj >> 1], yl[pj >> 1], (pj & 1) * 2 + q, std::true_type{});
    };
    auto epilogue = [&](const SxAcc& a, int pj, int q) {
      if (OUT) {
        if (q == 0 && g == 0 && rrow < M) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (r < n_out) Y[rrow * n_out + r] = combine(a, r);
        }
      } else {
        hidden_pair(a, pj, q);
      }
    };
#if QX_FINE
    // The epilogue of a hidden chunk (four values per lane) as single-instruction steps in a skewed order: step 4 t + r = stage t - r of
    // value r (stages: two combines, -|z| k, exp2, 1 + e, log2, max, fma [, / sqrt 2]) -- consecutive steps belong to different values, a
    // step's input is four steps old, at most every other step is a quarter-rate transcendental -- then the two pairs' exact three-way
    // splits (sx_split_pair's instructions, the pairs alternating) and the sentinel.  The same operations in the same order per value as
    // hidden_pair(): bit-identical results.
    constexpr int FNS = SKIPOUT ? 9 : 8, FNA = 4 * (FNS + 3), FNP = 2 * 11, FNSTEP = FNA + FNP;
    float f_t[4], f_z[4], f_e[4], f_u[4], f_v[4];
    unsigned f_h[2], f_m[2], f_l[2];
    float f_s[2][2], f_d[2][2];
    auto fine_step = [&](int sidx, int pj, const SxAcc& a) {
      if (sidx < FNA) {
        const int tt = sidx >> 2, r = sidx & 3, st = tt - r;
        if (st < 0 || st >= FNS) return;
        if (st == 0) f_t[r] = __builtin_fmaf(a.c2[r], C11, a.c1[r]);
        else if (st == 1) f_z[r] = __builtin_fmaf(f_t[r], C11, a.c0[r]);
        else if (st == 2) f_e[r] = -__builtin_fabsf(f_z[r]) * SP_T_PER_Z;
        else if (st == 3) f_e[r] = __builtin_amdgcn_exp2f(f_e[r]);
        else if (st == 4) f_u[r] = 1.0f + f_e[r];
        else if (st == 5) f_u[r] = __builtin_amdgcn_logf(f_u[r]);
        else if (st == 6) f_t[r] = __builtin_fmaxf(f_z[r], 0.0f);
        else if (st == 7) f_v[r] = __builtin_fmaf(f_u[r], SP_LN2_OVER_100, f_t[r]);
        else f_v[r] = f_v[r] * inv_sqrt2;
      } else {
        const int w = sidx - FNA, q = w & 1, st = w >> 1;
        const float v0 = f_v[2 * q], v1 = f_v[2 * q + 1];
        if (st == 0) f_h[q] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v0, v1));
        else if (st == 1) f_s[q][0] = v0 * 2048.0f;
        else if (st == 2) f_s[q][1] = v1 * 2048.0f;
        else if (st == 3) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(f_d[q][0]) : "v"(f_h[q]), "s"(negk), "v"(f_s[q][0]));
        else if (st == 4) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(f_d[q][1]) : "v"(f_h[q]), "s"(negk), "v"(f_s[q][1]));
        else if (st == 5) f_m[q] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(f_d[q][0], f_d[q][1]));
        else if (st == 6) f_s[q][0] = f_d[q][0] * 2048.0f;
        else if (st == 7) f_s[q][1] = f_d[q][1] * 2048.0f;
        else if (st == 8) asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(f_l[q]) : "v"(f_m[q]), "s"(negk), "v"(f_s[q][0]));
        else if (st == 9) asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(f_l[q]) : "v"(f_m[q]), "s"(negk), "v"(f_s[q][1]));
        else {
          const int o = (pj & 1) * 2 + q;
          yh[pj >> 1][o] = f_h[q];
          ym[pj >> 1][o] = f_m[q];
#if QX_FP8
          if constexpr (NEXT_FP8) {
            if (q == 1) {      // the top bytes of the block's four h / l halves: dword pj of the next layer's bf8 operands
              yh8[pj >> 3][pj & 7] = (int)__builtin_amdgcn_perm(f_h[1], f_h[0], 0x07050301u);
              yl8[pj >> 3][pj & 7] = (int)__builtin_amdgcn_perm(f_l[1], f_l[0], 0x07050301u);
            }
          } else
#endif
          yl[pj >> 1][o] = f_l[q];
          sat = sat_acc_pos(sat, f_h[q]);
        }
      }
    };
#endif
    zero_acc(accs[0], bias_of(0));
#pragma unroll
    for (int i = 0; i < D; ++i)
      if (i < NSTEP) {
#if QX_FP8
        if constexpr (FP8) {      // a group of 128 K = 768 lane-strided u4: [k-block 0..3][h | m] (512), h8 (128), l8 (128)
          const u4* f = frag_of(i / KB) + ((i % KB) >> 2) * 768 + (2 * ((i % KB) & 3)) * 64;
          wfh[i % NB] = f[0];
          wfm[i % NB] = f[64];
        } else
#endif
        {
          const u4* f = frag_of(i / KB) + (3 * (i % KB)) * 64;
          wfh[i % NB] = f[0];
          wfm[i % NB] = f[64];
          wfl[i % NB] = f[128];
        }
      }
#if QX_FP8
    qx_i8 w8h, w8l;
#endif
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int c = u / HV, hvi = u % HV;
      SxAcc& acc = accs[c & 1];
      if (u > 0 && hvi == 0) zero_acc(acc, bnext);
      constexpr int dummy2 = 0;
      (void)dummy2;
      const int L3 = u + 3 < NU ? LI : Net::layer_of(UB + u + 3);
      const int K3 = Net::kunit(L3);
      const int nu3 = sx_units(K3);
      (void)nu3;
      const long in_layer = (long)((u + 3) / HV) * Net::chunk_f4s(LI) + (long)((u + 3) % HV) * (KU / 32) * 192;
      const f4* src3 = u + 3 < NU ? wl + in_layer : wnext[u + 3 - NU < 3 ? u + 3 - NU : 0];
      const int sl3 = (u + 3) & 3;
      const unsigned dst3 = ring_b + slot_b[sl3], bdst3 = bias_b + bslot_b[sl3];
#if QX_SPREAD
      // The copies of unit u+3 ONE AT A TIME behind every third MFMA of the unit's second half instead of blocks of 1 / 4 / 2
      // (tools/ubench/dma_stagger.hip: back-to-back copies cost the issuing wave ~70 cycles each, copies four MFMAs apart ~26): copy 0 = the
      // bias head, copy c = piece c - 1 of this wave's span (M0 carried from piece to piece as in x6t_engine.h: nothing else writes it).
      const int ncp3 = sx_np(K3), ns3 = sx_ns(K3), nsw3 = sx_nsw(K3);
      const int first3 = wave * nsw3 < ns3 - nsw3 ? wave * nsw3 : ns3 - nsw3;
      const f4* span3 = src3 + 4 + first3 * 64;
      const unsigned dspan3 = dst3 + (unsigned)first3 * 1024u;
      unsigned lv3 = lane16;
      auto spread_site = [&](int site) {
        constexpr int NSITE = 2 * (KB - HB);
#ifdef QX_ABL_NODMA
        return;
#endif
#pragma unroll
        for (int cidx = 0; cidx < 8; ++cidx)
          if (cidx < ncp3 && (cidx * NSITE) / ncp3 == site) {
            if (cidx == 0) sx_dma4(src3, lane4, bdst3);
            else xt_copy_piece_seq(cidx - 1, span3, dspan3, lv3, true);
          }
      };
#endif
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const int st = u * KB + kb;
        if (kb == HB) {   // unit u+1 must have landed: this wave's copies of unit u+2 may still be in flight
          const int L2 = u + 2 < NU ? LI : Net::layer_of(UB + u + 2);
          const int allowed = sx_np(Net::kunit(L2));
          if (allowed >= 8) sx_wait<8>();
          else if (allowed >= 7) sx_wait<7>();
          else if (allowed >= 6) sx_wait<6>();
          else sx_wait<3>();
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          if ((u + 1) % HV == 0) bnext = bias_of(u + 1);
        }
        {
          const int s2 = st + D;
#ifdef QX_ABL_NOREAD                     // timing ablation (wrong results): the fragment registers are never refilled
          if (false) {
#else
          if (s2 < NSTEP) {
#endif
#if QX_FP8
            if constexpr (FP8) {
              const u4* f = frag_of(s2 / KB) + ((s2 % KB) >> 2) * 768 + (2 * ((s2 % KB) & 3)) * 64;
              wfm[s2 % NB] = f[64];
              wfh[s2 % NB] = f[0];
            } else
#endif
            {
              const u4* f = frag_of(s2 / KB) + (3 * (s2 % KB)) * 64;
              wfl[s2 % NB] = f[128];
              wfm[s2 % NB] = f[64];
              wfh[s2 % NB] = f[0];
            }
          }
        }
#if QX_FP8
        if constexpr (FP8) {
          if ((kb & 3) == 0) {      // the group's two bf8 fragments: four reads, used behind the group's last k-block
            const u4* f8 = frag_of(u) + (kb >> 2) * 768 + 512;
            const u4 a0 = f8[0], a1 = f8[64], b0 = f8[128], b1 = f8[192];
            w8h = qx_i8{(int)a0[0], (int)a0[1], (int)a0[2], (int)a0[3], (int)a1[0], (int)a1[1], (int)a1[2], (int)a1[3]};
            w8l = qx_i8{(int)b0[0], (int)b0[1], (int)b0[2], (int)b0[3], (int)b1[0], (int)b1[1], (int)b1[2], (int)b1[3]};
          }
        }
#endif
        {
          const int xk = hvi * KB + kb;
#define QX_MFMA(ACC, W, X) ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, W), __builtin_bit_cast(h8, X), ACC, 0, 0, 0)
#if QX_FINE && QX_FP8
          if constexpr (FP8) {
            // four f16 products per k-block + two bf8 MFMAs per group of four: 4 KB + 2 (KB / 4) slots per unit
            constexpr int NSC8 = HV * (KB * 4 + (KB / 4) * 2);
            constexpr int FPER8_ = (FNSTEP + NSC8 - 1) / NSC8, FPER8 = FPER8_ > 4 ? 4 : FPER8_;
            const bool fine_here8 = !OUT && c > 0;
            const int sbase8 = hvi * (KB * 4 + (KB / 4) * 2) + kb * 4 + (kb >> 2) * 2;
#define QX_SLOT8(J)                                                                                    \
  {                                                                                                    \
    if (fine_here8) {                                                                                  \
      const int sc_ = sbase8 + (J);                                                                    \
      _Pragma("unroll") for (int i_ = 0; i_ < FPER8; ++i_)                                             \
        if (FPER8 * sc_ + i_ < FNSTEP) fine_step(FPER8 * sc_ + i_, c - 1, accs[(c - 1) & 1]);          \
    }                                                                                                  \
    if (QX_SPREAD && kb >= HB && ((J) == 1 || (J) == 3)) spread_site(2 * (kb - HB) + ((J) == 3));      \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
  }
            QX_MFMA(acc.c2, wfm[st % NB], xm[xk]);
            QX_SLOT8(0)
            QX_MFMA(acc.c1, wfm[st % NB], xh[xk]);
            QX_SLOT8(1)
            QX_MFMA(acc.c1, wfh[st % NB], xm[xk]);
            QX_SLOT8(2)
            QX_MFMA(acc.c0, wfh[st % NB], xh[xk]);
            QX_SLOT8(3)
            if ((kb & 3) == 3) {
              acc.c2 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w8l, xh8[xk >> 2], acc.c2, 1, 1, 0, 0, 0, 0);
              QX_SLOT8(4)
              acc.c2 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w8h, xl8[xk >> 2], acc.c2, 1, 1, 0, 0, 0, 0);
              QX_SLOT8(5)
            }
#undef QX_SLOT8
            if (fine_here8 && hvi == HV - 1 && kb == KB - 1) {
#pragma unroll
              for (int m_ = 0; m_ < FNSTEP; ++m_)
                if (m_ >= FPER8 * NSC8) fine_step(m_, c - 1, accs[(c - 1) & 1]);
            }
          } else
#endif
#if QX_FINE
          {
          // a filler slot behind EVERY MFMA: the steps of chunk c-1's epilogue (hidden layers), the copies at their two sites per k-block
          constexpr int NSC = HV * KB * 6;                                        // MFMAs (= slots) per chunk
          constexpr int FPER_ = (FNSTEP + NSC - 1) / NSC, FPER = FPER_ > 4 ? 4 : FPER_;
          const bool fine_here = !OUT && c > 0;
#define QX_SLOT(J)                                                                                   \
  {                                                                                                  \
    if (fine_here) {                                                                                 \
      const int sc_ = (hvi * KB + kb) * 6 + (J);                                                     \
      _Pragma("unroll") for (int i_ = 0; i_ < FPER; ++i_)                                            \
        if (FPER * sc_ + i_ < FNSTEP) fine_step(FPER * sc_ + i_, c - 1, accs[(c - 1) & 1]);          \
    }                                                                                                \
    if (QX_SPREAD && kb >= HB && ((J) == 2 || (J) == 5)) spread_site(2 * (kb - HB) + ((J) == 5));   \
    if ((J) < 5) __builtin_amdgcn_sched_barrier(0);                                                  \
  }
          QX_MFMA(acc.c2, wfl[st % NB], xh[xk]);
          QX_SLOT(0)
          QX_MFMA(acc.c2, wfm[st % NB], xm[xk]);
          QX_SLOT(1)
          QX_MFMA(acc.c2, wfh[st % NB], xl[xk]);
          QX_SLOT(2)
          QX_MFMA(acc.c1, wfm[st % NB], xh[xk]);
          QX_SLOT(3)
          QX_MFMA(acc.c1, wfh[st % NB], xm[xk]);
          QX_SLOT(4)
          QX_MFMA(acc.c0, wfh[st % NB], xh[xk]);
          QX_SLOT(5)
#undef QX_SLOT
          if (fine_here && hvi == HV - 1 && kb == KB - 1) {      // what a short chunk (layer 0) cannot carry behind its MFMAs
#pragma unroll
            for (int m_ = 0; m_ < FNSTEP; ++m_)
              if (m_ >= FPER * NSC) fine_step(m_, c - 1, accs[(c - 1) & 1]);
          }
          }
#else
          QX_MFMA(acc.c2, wfl[st % NB], xh[xk]);
          QX_MFMA(acc.c2, wfm[st % NB], xm[xk]);
          QX_MFMA(acc.c2, wfh[st % NB], xl[xk]);
#if QX_SPREAD
          if (kb >= HB) {
            spread_site(2 * (kb - HB));
            __builtin_amdgcn_sched_barrier(0);
          }
#endif
          QX_MFMA(acc.c1, wfm[st % NB], xh[xk]);
          QX_MFMA(acc.c1, wfh[st % NB], xm[xk]);
          QX_MFMA(acc.c0, wfh[st % NB], xh[xk]);
#endif
#undef QX_MFMA
        }
#ifndef QX_ABL_NOEPI                     // timing ablation (wrong results): no softplus / split between the MFMAs
        if (c > 0 && hvi == 0 && (OUT || !QX_FINE)) {              // softplus + three-way split (or the store) of chunk c-1
          if (kb == 0) epilogue(accs[(c - 1) & 1], c - 1, 0);
          if (kb == (KB >= 6 ? 3 : 1)) epilogue(accs[(c - 1) & 1], c - 1, 1);
        }
#endif
#if QX_SPREAD
        if (kb >= HB && !QX_FINE) spread_site(2 * (kb - HB) + 1);
#else
        if (kb >= HB) {
#pragma unroll
          for (int un = 0; un < 3; ++un)
            if (un < nu3 && (un * (KB - HB)) / nu3 == kb - HB) {
#ifdef QX_ABL_NODMA                      // timing ablation (wrong results): no LDS-DMA copies after the prologue's
              continue;
#endif
              if (K3 == 64) qx_copy<64>(un, src3, lane4, lane16, bdst3, dst3, wave);
              else if (K3 == 192) qx_copy<192>(un, src3, lane4, lane16, bdst3, dst3, wave);
              else if (K3 == 256) qx_copy<256>(un, src3, lane4, lane16, bdst3, dst3, wave);
              else qx_copy<288>(un, src3, lane4, lane16, bdst3, dst3, wave);
            }
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    {   // slot 0 = the slot of the next layer's first unit
      constexpr int R = NU & 3;
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
    constexpr int NCH = NU / HV;
    const SxAcc& last = accs[(NCH - 1) & 1];
#if QX_FINE && QX_FP8
    if constexpr (!OUT) {      // the last chunk through the same steps (they also fill the bf8 operands)
#pragma unroll
      for (int m_ = 0; m_ < FNSTEP; ++m_) fine_step(m_, NCH - 1, last);
    } else
#endif
    {
      epilogue(last, NCH - 1, 0);
      epilogue(last, NCH - 1, 1);
    }
    if constexpr (SKIPOUT) {
      build_skip_operands();
      fold_sat_in();
    } else if constexpr (!OUT) {
#pragma unroll
      for (int kb = 0; kb < 16; ++kb) {
        xh[kb] = yh[kb];
        xm[kb] = ym[kb];
        if constexpr (!NEXT_FP8) xl[kb] = yl[kb];
      }
#if QX_FP8
      if constexpr (NEXT_FP8) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          xh8[gq] = yh8[gq];
          xl8[gq] = yl8[gq];
        }
      }
#endif
    }
  };

  // ---- prologue: units 0, 1, 2 of the stream (layer 0)
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int un = 0; un < 3; ++un)
      if (un < sx_units(K0P)) qx_copy<K0P>(un, Wp + Net::uoff(c), lane4, lane16, bias_b + bslot_b[c], ring_b + slot_b[c], wave);
  sx_wait<0>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  for (long round = blockIdx.x; round < nrounds; round += gridDim.x) {
    rrow = round * 64 + wave * 16 + (lane & 15);
    load_layer0();
    // layer 0 | 1, 2, 5, 6 (one instance) | 3 (skip layer's own outputs) | 4 (K = 576) | 7 | 8
#pragma unroll 1
    for (int l = 0; l < 9; ++l) {
      const int ub = Net::ubase(l);
      if (l == 0) run_layer(std::integral_constant<int, 0>{}, 0);
      else if (l == 3) run_layer(std::integral_constant<int, 3>{}, ub);
      else if (l == 4) run_layer(std::integral_constant<int, 4>{}, ub);
      else if (l == 7) run_layer(std::integral_constant<int, 7>{}, ub);
      else if (l == 8) run_layer(std::integral_constant<int, 8>{}, ub);
      else run_layer(std::integral_constant<int, 1>{}, ub);
    }
  }
  range_report<false>(sat, range_word);
  sx_wait<0>();
  __syncthreads();
}

}  // namespace rb

using namespace rb;

extern "C" int rb_cesr_net_x6_points(const float* x, long M, int kind, int n_label, const float* Wp, float* Y, int n_workgroups,
                                     rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(x && Wp && Y, "null pointer");
  const int pg = persistent_grid((M + 63) / 64, n_workgroups);
  if (pg <= 0) return rb::fail(__func__, "device query failed");
  const unsigned grid = (unsigned)pg;
  unsigned* rw = range_flags() ? range_flags() + RB_RANGE_SOFTPLUS512 : nullptr;
  hipStream_t s = (hipStream_t)stream;
  switch (kind) {
    case 0: hipLaunchKernelGGL((k_cesr_x6<64, 464, false>), dim3(grid), dim3(256), 0, s, x, M, 1, (const f4*)Wp, 3, Y, rw); break;
    case 2:
      RB_REQUIRE(n_label >= 1 && n_label <= 128, "n_label must be 1..128");
      hipLaunchKernelGGL((k_cesr_x6<192, 336, true>), dim3(grid), dim3(256), 0, s, x, M, n_label, (const f4*)Wp, 2, Y, rw);
      break;
    default: return rb::fail(__func__, "kind: 0 normal_net on PE10(x), 2 shadow_net on (point, one-hot label) rows");
  }
  return check_launch("k_cesr_x6");
}
