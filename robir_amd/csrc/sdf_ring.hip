// NeuS SDF network (model/neus_model.py:385-438), split-precision, second generation: the weight-ring structure of the
// light-visibility kernel (vis_diffuse_v2.hip) applied to the nine-layer softplus net with its skip connection.
//
// The first generation (k_sdf_mlp_h3, mlp_kernels_h3.hip) alternates phases -- a layer's MFMAs, then the softplus and the
// hi/lo split of its 256 outputs -- with a register-staged, double-buffered weight stream and one __syncthreads per 16-neuron
// chunk; one wave per SIMD means nothing overlaps, and it ran at 25 % of the split-precision bound (72 % of BASELINE config 2,
// 77 % of config 3: profiles/r02a_config{2,3}_kernel_stats.md).  Here:
//   * the whole net is ONE cyclic stream of 126 / 142 chunks (16 output neurons x K in {64, 256, 288}) through a 4-slot LDS
//     ring filled by LDS-DMA three chunks ahead (counted vmcnt + one s_barrier per chunk), workgroups are persistent over
//     rounds of 128 rows so the ring never drains;
//   * weight fragments roll through registers: a k-block's pair is refilled with the next chunk's right after its last use;
//   * softplus (value rows) / sigmoid x tangent (forward-mode rows) and the hi/lo split of chunk j run in the issue slots
//     between the MFMAs of chunk j+1 and write straight into the next layer's operand registers; the quad broadcast of the
//     value row's pre-activation is a DPP move, not an LDS permute;
//   * the output layer's values are stored from the same slots (no 68-register output array in the full mode).
// Arithmetic per element is that of the first generation (same scales, same product order): results agree to fp32 rounding
// of the softplus evaluation order, the tests hold both to the oracle.
#include "../../include/robir_hip.h"
#include "common.h"
#include "mlp_engine.h"
#include <cstdlib>
#include <type_traits>

namespace rb {

#ifndef SR_FILL
#define SR_FILL 8
#endif
constexpr int SR_SLOT_F4 = 1280;          // 20 KB per ring slot: five 4 KB DMA rows (K = 288 needs 4.5)

// The chunk stream of one pass, in closed form (no loops: these fold to constants once the chunk loops are unrolled).
//   layer      0    1    2    3    4    5    6    7    8
//   K         64  256  256  256  288  256  256  256  256
//   chunks    16   16   16   13   16   16   16   16   17 | 1
//   first      0   16   32   48   61   77   93  109  125
__host__ __device__ constexpr int sr_K(int l) { return l == 0 ? 64 : (l == 4 ? 288 : 256); }
__host__ __device__ constexpr int sr_nch(int l, int last) { return l == 3 ? 13 : (l == 8 ? last : 16); }
__host__ __device__ constexpr int sr_cbase(int l, int last) { return l < 4 ? 16 * l : 61 + 16 * (l - 4); }
__host__ __device__ constexpr int sr_layer_of(int c, int last) {
  return c < 48 ? (c >> 4) : (c < 61 ? 3 : (c < 125 ? 4 + ((c - 61) >> 4) : 8));
}
__host__ __device__ constexpr long sr_loff(int l, int last) {       // float4 offset of layer l in the packed blob
  // chunk_f4: K 64 -> 260, 256 -> 1028, 288 -> 1156
  return l == 0 ? 0L : (l <= 3 ? 4160L + 16448L * (l - 1) : (l == 4 ? 50420L : 68916L + 16448L * (l - 5)));
}
__host__ __device__ constexpr long sr_coff(int c, int last) {       // float4 offset of chunk c (bias first)
  const int l = sr_layer_of(c, last);
  return sr_loff(l, last) + (long)(c - sr_cbase(l, last)) * chunk_f4(sr_K(l));
}
static_assert(sr_loff(1, 17) == 16L * chunk_f4(64) && sr_loff(4, 17) == sr_loff(3, 17) + 13L * chunk_f4(256) &&
                  sr_loff(5, 17) == sr_loff(4, 17) + 16L * chunk_f4(288) && sr_loff(8, 17) == sr_loff(7, 17) + 16L * chunk_f4(256),
              "packed layer offsets");
__host__ __device__ constexpr int sr_pieces(int K) { return K == 64 ? 1 : (K == 256 ? 4 : 5); }   // 4 KB DMA rows per chunk

__device__ __forceinline__ void sr_dma16(const f4* gbase_uniform, unsigned lane_byte_off, unsigned lds_byte_uniform) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_byte_uniform), "v"(lane_byte_off),
               "s"(gbase_uniform)
               : "memory");
}
__device__ __forceinline__ void sr_wait(int allowed) {      // counted wait; `allowed` folds to a constant after unrolling
  if (allowed <= 1) {
    asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  } else if (allowed <= 3) {
    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  } else if (allowed <= 4) {
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  } else if (allowed <= 5) {
    asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  } else if (allowed <= 6) {
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  }
}
// 16 B per lane to uniform base + lane offset (scalar base: no per-store 64-bit vector address, which the compiler would
// otherwise precompute for every store of a layer and spill)
__device__ __forceinline__ void sr_store16(const f4* base_uniform, unsigned lane_byte_off, f4 v) {
  // a store of more than 64 bits reads its data registers late: a VALU write to them needs wait states in between
  // (cdna ISA, manually inserted wait states) -- the compiler adds them for its own stores, not after inline assembly
  asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(lane_byte_off), "v"(v), "s"(base_uniform) : "memory");
}
__device__ __forceinline__ float sr_quad0(float v) {        // value of lane (lane & ~3) of the quad: DPP quad_perm [0,0,0,0]
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x00, 0xf, 0xf, true));
}

struct SrAcc {
  f4 a[2];
};

// MODE: bit 0 = all 257 outputs (else the signed distance only), bit 1 = forward-mode input gradient (rows 4m + {0,1,2,3} =
// value + three tangent columns of point m), as rb_sdf_mlp_h3; bit 2 (with bit 0, without bit 1) = also store sigmoid(100 z) of
// every hidden pre-activation for the reverse-mode gradient pass (sdf_back.hip): `sig` [rounds][125 chunks][2 tiles][256 lanes]
// float4 -- lane-local, in the order the epilogue produces it, 16 B per lane and chunk.
// FUSED (forward-mode rows only; value rows take the eight-wave kernel): X = the points xyz[M,3], encoded at the top of every round
// by load_features_pe10 (mlp_engine.h) -- the value row and the three tangent rows of a point by the sixteen lanes that hold them.
template <int MODE, bool FUSED = false>
__global__ __launch_bounds__(256, 1) void k_sdf_ring(const float* __restrict__ X, long MR, const f4* __restrict__ Wp, float us,
                                                      float out_scale, float grad_scale, float* __restrict__ out0,
                                                      float* __restrict__ grad, unsigned* __restrict__ range_word,
                                                      f4* __restrict__ sig, float in_scale = 1.0f) {
  constexpr bool JVP = (MODE & 2) != 0;
  constexpr bool FULL = (MODE & 1) != 0;
  constexpr bool STORE = (MODE & 4) != 0;
  static_assert(!STORE || (FULL && !JVP), "sigmoid store: value rows, all outputs");
  constexpr int LAST = FULL ? 17 : 1;                 // chunks of the output layer (272 = 257 padded, or the sdf row's chunk)
  constexpr int NCHUNK = sr_cbase(8, LAST) + LAST;    // 142 / 126
  constexpr float AS = 64.0f, TS = 0.25f;             // operand lifts of value rows / tangent rows (powers of two)
  __shared__ f4 ring[4 * SR_SLOT_F4];                 // 80 KB
  __shared__ f4 bias_tab[NCHUNK * 4];
  __shared__ float pe_scratch[FUSED ? 4 * 2 * 16 * 64 : 4];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long nrounds = (MR + 127) >> 7;
  if (tid < NCHUNK) {
    const f4* src = Wp + sr_coff(tid, LAST);
#pragma unroll
    for (int q = 0; q < 4; ++q) bias_tab[tid * 4 + q] = src[q];
  }
  __syncthreads();
  if ((long)blockIdx.x >= nrounds) return;

  const bool is_val = JVP ? ((lane & 3) == 0) : true;
  const float asc = is_val ? AS : TS;          // lift of this lane's operands
  const float bm = is_val ? AS : 0.0f;         // bias multiplier (tangent rows carry no bias)
  const float zs = us / asc;                   // un-scaling of an MFMA result
  const float inv_sqrt2 = 0.70710678118654752440f;
  const float os = out_scale * us * (1.0f / AS), gs = grad_scale * us * (1.0f / TS);

  const unsigned ring_b = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)ring);
  const unsigned lane_off = (unsigned)tid * 16u;
  const unsigned wave_lds = ring_b + (unsigned)wave * 1024u;
  // ring slot of stream position (round * NCHUNK + c) is (c + sbase) & 3 with sbase = (round index of this workgroup *
  // NCHUNK) & 3 in {0, 2} (NCHUNK = 2 mod 4): four byte offsets, rotated by two slots at the end of every round
  static_assert(NCHUNK % 4 == 2, "slot rotation assumes NCHUNK = 2 (mod 4)");
  unsigned slot_b[4] = {0u, SR_SLOT_F4 * 16u, 2u * SR_SLOT_F4 * 16u, 3u * SR_SLOT_F4 * 16u};
  const u4* ring_u = reinterpret_cast<const u4*>(ring) + lane;   // slot 0 (prologue)
  u4 wreg[18];
  unsigned sat = 0u;
  // operand registers are 128-bit tuples (one MFMA B operand each): declared as vectors so that the register allocator keeps a
  // k-block's four registers contiguous -- as scalars it rebuilt every tuple with accumulator-file copies (1.35 per MFMA)
  u4 xh[2][9], xl[2][9];               // operands of the current layer (K <= 288: nine k-blocks of 32)
  u4 yh[2][9], yl[2][9];               // ... of the next layer
  u4 sh[2][2], sl[2][2];               // the 64 input features / sqrt(2), lifted: skip operands of layer 4
  f4 fraw[2][4];                       // input features of the NEXT round (prefetched)
  long rrow[2];                        // this lane's row in each tile of the current round

  auto fetch_features = [&](long round) {
    if constexpr (FUSED) return;        // encoded at the top of the round instead
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const long row = round * 128 + wave * 32 + t * 16 + (lane & 15);
      const bool ok = round < nrounds && row < MR;
      const f4* p = reinterpret_cast<const f4*>(X + (ok ? row : 0) * 64) + g;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) fraw[t][kb] = ok ? p[kb * 4] : f4{0.f, 0.f, 0.f, 0.f};
    }
  };

  auto mfma_kb = [&](int kb, SrAcc& acc) {
    const h8 wh = __builtin_bit_cast(h8, wreg[kb * 2]);
    const h8 wlo = __builtin_bit_cast(h8, wreg[kb * 2 + 1]);
    h8 a[2], b[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      a[t] = __builtin_bit_cast(h8, xh[t][kb]);
      b[t] = __builtin_bit_cast(h8, xl[t][kb]);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      acc.a[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, b[t], acc.a[t], 0, 0, 0);
      acc.a[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, a[t], acc.a[t], 0, 0, 0);
      acc.a[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo, a[t], acc.a[t], 0, 0, 0);
      if (t == 0) asm volatile("" : "+a"(acc.a[0]), "+a"(acc.a[1]));
    }
  };
  // activation of two pre-activations of tile t -> lifted operand pair (softplus on value rows, z' * sigmoid(100 z) on
  // tangent rows with z taken from the quad's value lane), `sa` = output scale x operand lift
  f4 sstage = {0.f, 0.f, 0.f, 0.f};   // STORE: the four sigmoids of a (chunk, tile) on their way to `sig`
  f4* sig_round = sig;                // ... of the current round
  auto act_pair = [&](float r0, float r1, float sa, unsigned& hi, unsigned& lo, int q = 0) {
    const float z0 = r0 * zs, z1 = r1 * zs;
    float v0, v1;
    if constexpr (STORE) {
      float s0, s1;
      v0 = softplus100_fast(z0, &s0);
      v1 = softplus100_fast(z1, &s1);
      sstage[2 * q] = s0;
      sstage[2 * q + 1] = s1;
    } else if constexpr (JVP) {
      float s0, s1;
      const float p0 = softplus100_fast(sr_quad0(z0), &s0), p1 = softplus100_fast(sr_quad0(z1), &s1);
      v0 = is_val ? p0 : z0 * s0;
      v1 = is_val ? p1 : z1 * s1;
    } else {
      v0 = softplus100_fast(z0, nullptr);
      v1 = softplus100_fast(z1, nullptr);
    }
    split_pair_mix(v0 * sa, v1 * sa, hi, lo);
    sat = sat_acc(sat, hi);
  };
  // Forward-mode rows: the four lanes of a quad hold (value, d/dx, d/dy, d/dz) of one point for the SAME four neurons
  // r = 0..3 of the chunk.  Evaluating softplus / sigmoid of the value row's pre-activation in every lane for every r is a
  // four-fold redundancy (and three transcendentals each); instead lane k of the quad evaluates neuron r = k once and the
  // results travel by DPP quad broadcasts: 3 transcendentals per tile instead of 12.  Two stages, placed in different
  // k-block gaps: (a) gather + softplus, (b) redistribute + tangent products + lift + hi/lo split.
  float qz[4], qsp = 0.f, qsig = 0.f;
  auto quad_bcast = [&](float v, int r) {      // value held by lane r of the quad, in all four lanes (r folds to a constant)
    const int iv = __builtin_bit_cast(int, v);
    int o;
    if (r == 0) o = __builtin_amdgcn_update_dpp(0, iv, 0x00, 0xf, 0xf, true);
    else if (r == 1) o = __builtin_amdgcn_update_dpp(0, iv, 0x55, 0xf, 0xf, true);
    else if (r == 2) o = __builtin_amdgcn_update_dpp(0, iv, 0xaa, 0xf, 0xf, true);
    else o = __builtin_amdgcn_update_dpp(0, iv, 0xff, 0xf, 0xf, true);
    return __builtin_bit_cast(float, o);
  };
  const int qk = lane & 3;
  auto jvp_stage_a = [&](const SrAcc& acc, int t) {
    float b[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      qz[r] = acc.a[t][r] * zs;
      b[r] = quad_bcast(qz[r], 0);
    }
    const float zk = qk == 0 ? b[0] : (qk == 1 ? b[1] : (qk == 2 ? b[2] : b[3]));
    qsp = softplus100_fast(zk, &qsig);
  };
  auto jvp_stage_b = [&](int jb, int t, float sa) {
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float sg = quad_bcast(qsig, r), sp = quad_bcast(qsp, r);
      v[r] = (is_val ? sp : qz[r] * sg) * sa;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      unsigned hi, lo;
      split_pair_mix(v[2 * q], v[2 * q + 1], hi, lo);
      yh[t][jb >> 1][(jb & 1) * 2 + q] = hi;
      yl[t][jb >> 1][(jb & 1) * 2 + q] = lo;
      sat = sat_acc(sat, hi);
    }
  };
  // Value rows, staged: softplus is a chain of transcendentals (exp -> log, and rcp for the sigmoid) whose latencies an in-order
  // wave waits out when the chain sits in one gap between MFMAs.  Cut at the transcendentals and spread over three consecutive
  // k-block gaps (six MFMAs between the stages), the operations of softplus100_fast in the same order (bit-identical results)
  // issue without the stalls: stage 1 pre-activation + exponential, stage 2 (reciprocal +) logarithm, stage 3 select, lift, split, store.
  float pz[4][2], pt[4][2], pe[4][2], pr[4][2], pl[4][2];      // per piece (tile, register pair) and element
  auto val_stage1 = [&](const SrAcc& acc, int piece) {         // softplus100_fast, cut at its transcendentals
    const int t = piece >> 1, q = piece & 1;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float z = acc.a[t][2 * q + e] * zs;
      pz[piece][e] = z;
      pt[piece][e] = z * SP_T_PER_Z;
      pe[piece][e] = __builtin_amdgcn_exp2f(pt[piece][e]);
    }
  };
  auto val_stage2 = [&](int piece) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float u = 1.0f + pe[piece][e];
      if constexpr (STORE) pr[piece][e] = __builtin_amdgcn_rcpf(u);
      pl[piece][e] = __builtin_amdgcn_logf(u);
    }
  };
  auto val_stage3 = [&](int jb, int piece, float sa, int cb) {
    const int t = piece >> 1, q = piece & 1;
    float v[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const bool lin = pt[piece][e] > SP_T_LINEAR;
      const float sp = pl[piece][e] * SP_LN2_OVER_100;
      v[e] = lin ? pz[piece][e] : sp;
      if constexpr (STORE) sstage[2 * q + e] = lin ? 1.0f : pe[piece][e] * pr[piece][e];
    }
    unsigned hi, lo;
    split_pair_mix(v[0] * sa, v[1] * sa, hi, lo);
    sat = sat_acc(sat, hi);
    yh[t][jb >> 1][(jb & 1) * 2 + q] = hi;
    yl[t][jb >> 1][(jb & 1) * 2 + q] = lo;
    if constexpr (STORE) {
      if (q == 1) {
        const f4* base = sig_round;
        asm volatile("" : "+s"(base));
        sr_store16(base + ((cb + jb) * 2 + t) * 256, lane_off, sstage);
      }
    }
  };
  // piece (tile, register pair) of hidden chunk jb -> next layer's k-block jb/2, registers 2*(jb&1)+q
  auto hidden_piece = [&](const SrAcc& acc, int jb, int piece, float sa, int cb) {
    const int t = piece >> 1, q = piece & 1;
    if constexpr (JVP) {
      if (q == 0) jvp_stage_a(acc, t); else jvp_stage_b(jb, t, sa);
    } else {
      unsigned hi, lo;
      act_pair(acc.a[t][2 * q], acc.a[t][2 * q + 1], sa, hi, lo, q);
      yh[t][jb >> 1][(jb & 1) * 2 + q] = hi;
      yl[t][jb >> 1][(jb & 1) * 2 + q] = lo;
      if constexpr (STORE) {
        if (q == 1) {
          const f4* base = sig_round;
          asm volatile("" : "+s"(base));            // one scalar add per store, not 32 precomputed addresses per layer
          sr_store16(base + ((cb + jb) * 2 + t) * 256, lane_off, sstage);
        }
      }
    }
  };
  // piece of output chunk jb: stores
  auto output_piece = [&](const SrAcc& acc, int jb, int piece) {
    const int t = piece >> 1, q = piece & 1;
    const long row = rrow[t];
    if (row >= MR) return;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int r = 2 * q + e;
      const float v = acc.a[t][r];
      const int j = jb * 16 + 4 * g + r;
      if constexpr (!JVP) {
        if constexpr (FULL) {
          if (j < 257) out0[row * 257 + j] = v * os;
        } else {
          if (g == 0 && r == 0) out0[row] = v * os;
        }
      } else {
        const long m = row >> 2;
        const int c = (int)(row & 3);
        if (c == 0) {
          if constexpr (FULL) {
            if (j < 257) out0[m * 257 + j] = v * os;
          } else {
            if (g == 0 && r == 0) out0[m] = v * os;
          }
        } else if (g == 0 && r == 0 && jb == 0) {
          grad[m * 3 + (c - 1)] = v * gs;
        }
      }
    }
  };

  // ---- one layer of the chunk stream.  Compile time: K (inputs), NCH (chunks), EPI (0 hidden softplus, 1 = layer 3: scaled
  // by 1/sqrt 2, 2 = output stores), KF / PF (K and DMA rows of the chunks that FOLLOW this layer in the stream: the
  // look-ahead of its last three chunks).  Run time (so that the five 256 -> 256 hidden layers share one copy of the code):
  //   wl      this layer's chunk 0 in the packed blob,
  //   tail    chunks NCH .. NCH+2 of the stream counted from this layer's first chunk (normally wl + (NCH + k) * chunk size --
  //           the blob is contiguous --, the start of the blob when the stream wraps),
  //   cb      stream index of this layer's first chunk (bias table row), sl[k] = LDS byte offset of the slot of chunk cb + k.
  auto run_layer = [&](auto K_tag, auto NCH_tag, auto EPI_tag, auto KF_tag, auto PF_tag, const f4* wl, const f4* tail0,
                       const f4* tail1, const f4* tail2, int cb, const unsigned (&sl)[4]) {
    constexpr int K = decltype(K_tag)::value, KB = K / 32, NCH = decltype(NCH_tag)::value, EPI = decltype(EPI_tag)::value;
    // PF = rows of the copies of the three tail chunks, as decimal digits (444: all four rows)
    constexpr int KF = decltype(KF_tag)::value, PFS = decltype(PF_tag)::value, P = sr_pieces(K);
    constexpr int PF0 = PFS / 100, PF1 = (PFS / 10) % 10, PF2 = PFS % 10;
    constexpr int CF4 = chunk_f4(K);
    const float sa = (EPI == 1 ? inv_sqrt2 : 1.0f) * asc;
    asm volatile("" : "+s"(wl), "+s"(tail0), "+s"(tail1), "+s"(tail2));     // keep the address arithmetic inside the loop
    SrAcc prev;
#pragma unroll
    for (int jb = 0; jb < NCH; ++jb) {
      SrAcc acc;
      const f4 bias = bias_tab[(cb + jb) * 4 + g] * bm;
      acc.a[0] = bias;
      acc.a[1] = bias;
      // chunk jb+1 must have landed; only the copy of chunk jb+2 (issued during chunk jb-1) may still be in flight.  Anything
      // else younger (feature prefetch, output stores) is not credited: waiting for it too is safe, and rare
      // (STORE: the sigmoid stores of the previous iteration's epilogue sit in that window too.  They are not credited: LDS-DMA
      // rows and stores do not retire in order with each other (sdf_back.hip), and a store that retired early must not stand
      // in for a row -- at worst this waits for two rows more than necessary)
      sr_wait(jb + 2 < NCH ? P : (jb + 2 == NCH ? PF0 : PF1));
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const int KBn = (jb + 1 < NCH ? K : KF) / 32;             // next chunk: its fragments roll into wreg
      const u4* ring_n = reinterpret_cast<const u4*>(reinterpret_cast<const char*>(ring) + sl[(jb + 1) & 3]) + lane;
      const int P3 = jb + 3 < NCH ? P : (jb + 3 == NCH ? PF0 : (jb + 3 == NCH + 1 ? PF1 : PF2));   // rows of the copy issued now
      const f4* src3 = jb + 3 < NCH ? wl + (long)(jb + 3) * CF4 : (jb + 3 == NCH ? tail0 : (jb + 3 == NCH + 1 ? tail1 : tail2));
      const unsigned dst3 = wave_lds + sl[(jb + 3) & 3];
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        mfma_kb(kb, acc);
        if (kb < KBn) {
          wreg[2 * kb] = ring_n[(2 * kb) * 64];
          wreg[2 * kb + 1] = ring_n[(2 * kb + 1) * 64];
        }
        if (jb > 0) {
          if constexpr (!JVP && EPI != 2 && KB >= 8) {
            // staged value rows: piece pc starts in gap {0, 1, 3, 5}[pc] and takes three consecutive gaps
#pragma unroll
            for (int pc = 0; pc < 4; ++pc) {
              const int g0 = pc == 0 ? 0 : 2 * pc - 1;
              if (g0 + 2 == kb) val_stage3(jb - 1, pc, sa, cb);
            }
#pragma unroll
            for (int pc = 0; pc < 4; ++pc) {
              const int g0 = pc == 0 ? 0 : 2 * pc - 1;
              if (g0 + 1 == kb) val_stage2(pc);
            }
#pragma unroll
            for (int pc = 0; pc < 4; ++pc) {
              const int g0 = pc == 0 ? 0 : 2 * pc - 1;
              if (g0 == kb) val_stage1(prev, pc);
            }
          } else {
#pragma unroll
            for (int pc = 0; pc < 4; ++pc)
              if ((pc * KB) / 4 == kb) {
                if constexpr (EPI == 2) output_piece(prev, jb - 1, pc); else hidden_piece(prev, jb - 1, pc, sa, cb);
              }
          }
        }
#pragma unroll
        for (int i = 0; i < 5; ++i)
          if (i < P3 && ((i * KB) / P3 + 1 < KB ? (i * KB) / P3 + 1 : KB - 1) == kb)
            sr_dma16(src3 + 4 + i * 256, lane_off, dst3 + (unsigned)i * 4096u);
        // One wave per SIMD issues in order: six MFMAs back to back (two dependent chains of three) block the wave for their
        // whole 96 cycles and the activation work of the gap issues behind them -- matrix and vector time add up (cycle stamps:
        // 1800 cycles per chunk for 768 cycles of MFMAs).  Ask the scheduler for one MFMA followed by a handful of other
        // instructions, six times: the vector work issues in the 12 idle issue cycles behind each MFMA.
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x506, SR_FILL, 0);     // VALU | SALU | DS read | transcendental
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int kb = KB; kb < KBn; ++kb) {                      // next chunk is wider (K 256 -> 288)
        wreg[2 * kb] = ring_n[(2 * kb) * 64];
        wreg[2 * kb + 1] = ring_n[(2 * kb + 1) * 64];
      }
      prev = acc;
    }
#pragma unroll
    for (int pc = 0; pc < 4; ++pc) {
      if constexpr (EPI == 2) output_piece(prev, NCH - 1, pc); else hidden_piece(prev, NCH - 1, pc, sa, cb);
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  auto y_to_x = [&](int nkb) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int kb = 0; kb < 9; ++kb)
        if (kb < nkb) {
          xh[t][kb] = yh[t][kb];
          xl[t][kb] = yl[t][kb];
        }
  };
  // rows of the copies of the first two chunks of the stream: in the distance-only modes the output layer is a single chunk,
  // so the look-ahead of layer 7's last two chunks (shared code: four rows, like every 256-wide chunk) lands on them
  constexpr int P_HEAD = FULL ? 1 : 4;

  // ---- prologue: ring start, first round's features
  long round = blockIdx.x;
  fetch_features(round);
  {
    const unsigned d0 = wave_lds + slot_b[0], d1 = wave_lds + slot_b[1], d2 = wave_lds + slot_b[2];
#pragma unroll
    for (int i = 0; i < P_HEAD; ++i) sr_dma16(Wp + sr_coff(0, LAST) + 4 + i * 256, lane_off, d0 + (unsigned)i * 4096u);
#pragma unroll
    for (int i = 0; i < P_HEAD; ++i) sr_dma16(Wp + sr_coff(1, LAST) + 4 + i * 256, lane_off, d1 + (unsigned)i * 4096u);
    sr_dma16(Wp + sr_coff(2, LAST) + 4, lane_off, d2);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int i = 0; i < 4; ++i) wreg[i] = ring_u[i * 64];        // chunk 0: K = 64, two k-blocks

  constexpr long CF256 = chunk_f4(256);
  for (; round < nrounds; round += gridDim.x) {
#pragma unroll
    for (int t = 0; t < 2; ++t) rrow[t] = round * 128 + wave * 32 + t * 16 + (lane & 15);
    if constexpr (STORE) sig_round = sig + round * (125L * 2 * 256);
    // ---- input features -> operands of layer 0 and the skip operands of layer 4
    if constexpr (FUSED) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        float enc[16];
        load_features_pe10<JVP>(X, in_scale, rrow[t], MR, lane, pe_scratch + (wave * 2 + t) * 1024, enc);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) fraw[t][kb] = f4{enc[kb * 4], enc[kb * 4 + 1], enc[kb * 4 + 2], enc[kb * 4 + 3]};
      }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const f4 v = fraw[t][kb];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          unsigned hi, lo;
          split_pair_mix(v[2 * q] * asc, v[2 * q + 1] * asc, hi, lo);
          xh[t][kb / 2][(kb & 1) * 2 + q] = hi;
          xl[t][kb / 2][(kb & 1) * 2 + q] = lo;
          sat = sat_acc(sat, hi);
          split_pair_mix(v[2 * q] * inv_sqrt2 * asc, v[2 * q + 1] * inv_sqrt2 * asc, hi, lo);
          sh[t][kb / 2][(kb & 1) * 2 + q] = hi;
          sl[t][kb / 2][(kb & 1) * 2 + q] = lo;
        }
      }
    // layer 0 (K = 64; its first two chunks were copied with P_HEAD rows): followed by layer 1
    {
      const unsigned s4[4] = {slot_b[0], slot_b[1], slot_b[2], slot_b[3]};
      const f4* w1 = Wp + sr_loff(1, LAST);
      run_layer(std::integral_constant<int, 64>{}, std::integral_constant<int, 16>{}, I0{}, std::integral_constant<int, 256>{},
                std::integral_constant<int, 444>{}, Wp, w1, w1 + CF256, w1 + 2 * CF256, 0, s4);
    }
    y_to_x(8);
#pragma unroll 1
    for (int seg = 0; seg < 2; ++seg) {
      // the 256 -> 256 hidden layers: 1, 2 (seg 0) and 5, 6, 7 (seg 1) -- one copy of the code
      const int nrep = seg == 0 ? 2 : 3;
#pragma unroll 1
      for (int rep = 0; rep < nrep; ++rep) {
        const int cb = seg == 0 ? 16 + 16 * rep : 77 + 16 * rep;
        const f4* wl = Wp + (seg == 0 ? sr_loff(1, LAST) : sr_loff(5, LAST)) + (long)rep * 16 * CF256;
        const bool wraps = !FULL && seg == 1 && rep == 2;     // layer 7 of a distance-only pass: output chunk, then the stream restarts
        const f4* t0 = wl + 16 * CF256;
        const f4* t1 = wraps ? Wp + sr_coff(0, LAST) : wl + 17 * CF256;
        const f4* t2 = wraps ? Wp + sr_coff(1, LAST) : wl + 18 * CF256;
        const int rot = cb & 3;                                 // 0 for layers 1, 2; 1 for layers 5, 6, 7
        const unsigned s4[4] = {rot ? slot_b[1] : slot_b[0], rot ? slot_b[2] : slot_b[1], rot ? slot_b[3] : slot_b[2],
                                rot ? slot_b[0] : slot_b[3]};
        run_layer(std::integral_constant<int, 256>{}, std::integral_constant<int, 16>{}, I0{}, std::integral_constant<int, 256>{},
                  std::integral_constant<int, 444>{}, wl, t0, t1, t2, cb, s4);
        y_to_x(8);
      }
      if (seg == 0) {
        {   // layer 3 (chunks 48..60, slot of chunk 48 = slot_b[0]): followed by layer 4 (K = 288, five rows)
          const unsigned s4[4] = {slot_b[0], slot_b[1], slot_b[2], slot_b[3]};
          const f4* w3 = Wp + sr_loff(3, LAST);
          const f4* w4 = Wp + sr_loff(4, LAST);
          run_layer(std::integral_constant<int, 256>{}, std::integral_constant<int, 13>{}, I1{}, std::integral_constant<int, 288>{},
                    std::integral_constant<int, 555>{}, w3, w4, w4 + chunk_f4(288), w4 + 2 * chunk_f4(288), 48, s4);
        }
        // layer 4 input = [softplus(layer 3) (208 slots) | features (64 slots) | 16 zero slots] / sqrt 2: the 13 chunks of
        // layer 3 filled k-blocks 0..5 and the first half of 6; the features go to slots 208..271
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          yh[t][6][2] = sh[t][0][0]; yl[t][6][2] = sl[t][0][0];
          yh[t][6][3] = sh[t][0][1]; yl[t][6][3] = sl[t][0][1];
          yh[t][7][0] = sh[t][0][2]; yl[t][7][0] = sl[t][0][2];
          yh[t][7][1] = sh[t][0][3]; yl[t][7][1] = sl[t][0][3];
          yh[t][7][2] = sh[t][1][0]; yl[t][7][2] = sl[t][1][0];
          yh[t][7][3] = sh[t][1][1]; yl[t][7][3] = sl[t][1][1];
          yh[t][8][0] = sh[t][1][2]; yl[t][8][0] = sl[t][1][2];
          yh[t][8][1] = sh[t][1][3]; yl[t][8][1] = sl[t][1][3];
          yh[t][8][2] = 0u; yl[t][8][2] = 0u;
          yh[t][8][3] = 0u; yl[t][8][3] = 0u;
        }
        y_to_x(9);
        {   // layer 4 (chunks 61..76, slot of chunk 61 = slot_b[1]): followed by layer 5
          const unsigned s4[4] = {slot_b[1], slot_b[2], slot_b[3], slot_b[0]};
          const f4* w4 = Wp + sr_loff(4, LAST);
          const f4* w5 = Wp + sr_loff(5, LAST);
          run_layer(std::integral_constant<int, 288>{}, std::integral_constant<int, 16>{}, I0{}, std::integral_constant<int, 256>{},
                    std::integral_constant<int, 444>{}, w4, w5, w5 + CF256, w5 + 2 * CF256, 61, s4);
        }
        y_to_x(8);
        fetch_features(round + gridDim.x);      // next round's input rows: consumed at the top of the next round
        __builtin_amdgcn_sched_barrier(0);
      } else {
        // output layer (chunks 125.., slot of chunk 125 = slot_b[1]): followed by the start of the stream
        const unsigned s4[4] = {slot_b[1], slot_b[2], slot_b[3], slot_b[0]};
        const f4* w8 = Wp + sr_loff(8, LAST);
        // its tail is the start of the stream: chunks 0, 1, 2 (K = 64).  In the distance-only modes chunks 0 and 1 were already
        // requested, with four rows each, by layer 7's shared look-ahead; chunk 2 takes its natural single row
        const f4* h0 = Wp;
        run_layer(std::integral_constant<int, 256>{}, std::integral_constant<int, LAST>{}, I2{}, std::integral_constant<int, 64>{},
                  std::integral_constant<int, FULL ? 111 : 441>{}, w8, h0, h0 + chunk_f4(64), h0 + 2 * chunk_f4(64), 125, s4);
      }
    }
    {                                  // the stream continues at slot (NCHUNK & 3) = 2: rotate the slot table by two
      const unsigned a = slot_b[0], b = slot_b[1];
      slot_b[0] = slot_b[2];
      slot_b[1] = slot_b[3];
      slot_b[2] = a;
      slot_b[3] = b;
    }
  }
  range_report(sat, range_word);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

}  // namespace rb

using namespace rb;

namespace rb {
int launch_sdf_ring8(int mode, const float* X, const float* xyz, float in_scale, long M, const f4* W, float us, float out_scale,
                     float* out0, f4* sig, unsigned grid, hipStream_t s);
int g_sdf_ring_waves = 8;     // value rows (modes 0, 1, 5): 8 = k_sdf_ring8 (two waves per SIMD), 4 = k_sdf_ring
}  // namespace rb

extern "C" int rb_sdf_ring_waves(int waves) {
  const int old = rb::g_sdf_ring_waves;
  if (waves == 4 || waves == 8) rb::g_sdf_ring_waves = waves;
  return old;
}

extern "C" int rb_sdf_mlp_ring(const float* X, long M, const float* Wp, int mode, int scale_log2, float out_scale, float grad_scale,
                               float* out0, float* grad, int n_workgroups, rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(X && Wp && out0, "null pointer");
  RB_REQUIRE(mode >= 0 && mode <= 3, "mode: bit 0 = all 257 outputs, bit 1 = input gradient");
  RB_REQUIRE(mode < 2 || grad, "gradient output missing");
  const long MR = mode >= 2 ? 4 * M : M;
  const long rounds = (MR + 127) / 128;
  if (n_workgroups <= 0) {
    static int cus = 0;
    if (!cus) {
      int dev = 0;
      hipDeviceProp_t prop;
      if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return rb::fail(__func__, "device query failed");
      cus = prop.multiProcessorCount;
    }
    n_workgroups = cus;
  }
  const unsigned grid = (unsigned)(rounds < n_workgroups ? rounds : n_workgroups);
  const float us = ldexpf(1.0f, -scale_log2);
  unsigned* rw = range_flags() ? range_flags() + RB_RANGE_SDF : nullptr;
  hipStream_t s = (hipStream_t)stream;
  const f4* W = (const f4*)Wp;
  if (mode < 2 && rb::g_sdf_ring_waves == 8) return rb::launch_sdf_ring8(mode, X, nullptr, 1.0f, M, W, us, out_scale, out0, nullptr, grid, s);
  switch (mode) {
    case 0: hipLaunchKernelGGL(k_sdf_ring<0>, dim3(grid), dim3(256), 0, s, X, MR, W, us, out_scale, grad_scale, out0, grad, rw, nullptr); break;
    case 1: hipLaunchKernelGGL(k_sdf_ring<1>, dim3(grid), dim3(256), 0, s, X, MR, W, us, out_scale, grad_scale, out0, grad, rw, nullptr); break;
    case 2: hipLaunchKernelGGL(k_sdf_ring<2>, dim3(grid), dim3(256), 0, s, X, MR, W, us, out_scale, grad_scale, out0, grad, rw, nullptr); break;
    default: hipLaunchKernelGGL(k_sdf_ring<3>, dim3(grid), dim3(256), 0, s, X, MR, W, us, out_scale, grad_scale, out0, grad, rw, nullptr); break;
  }
  return check_launch("k_sdf_ring");
}

// Forward-mode rows (value + three tangent rows per point) straight from the points: mode 2 = distance + gradient, 3 = all outputs +
// gradient; bit-identical to rb_feat_pe10(jvp) + rb_sdf_mlp_ring.
extern "C" int rb_sdf_points_ring_jvp(const float* x, long M, float in_scale, const float* Wp, int mode, int scale_log2, float out_scale,
                                      float grad_scale, float* out0, float* grad, int n_workgroups, rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(x && Wp && out0 && grad, "null pointer");
  RB_REQUIRE(mode == 2 || mode == 3, "mode: 2 = distance + gradient, 3 = all 257 outputs + gradient");
  const long MR = 4 * M, rounds = (MR + 127) / 128;
  if (n_workgroups <= 0) {
    static int cus = 0;
    if (!cus) {
      int dev = 0;
      hipDeviceProp_t prop;
      if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return rb::fail(__func__, "device query failed");
      cus = prop.multiProcessorCount;
    }
    n_workgroups = cus;
  }
  const unsigned grid = (unsigned)(rounds < n_workgroups ? rounds : n_workgroups);
  const float us = ldexpf(1.0f, -scale_log2);
  unsigned* rw = range_flags() ? range_flags() + RB_RANGE_SDF : nullptr;
  hipStream_t s = (hipStream_t)stream;
  if (mode == 2) {
    hipLaunchKernelGGL((k_sdf_ring<2, true>), dim3(grid), dim3(256), 0, s, x, MR, (const f4*)Wp, us, out_scale, grad_scale, out0, grad, rw,
                       nullptr, in_scale);
  } else {
    hipLaunchKernelGGL((k_sdf_ring<3, true>), dim3(grid), dim3(256), 0, s, x, MR, (const f4*)Wp, us, out_scale, grad_scale, out0, grad, rw,
                       nullptr, in_scale);
  }
  return check_launch("k_sdf_ring<jvp, points>");
}

// Value rows straight from the points (positional encoding fused into k_sdf_ring8): x[M,3], evaluated at x * in_scale.
// mode 0 = signed distance only, 1 = all 257 outputs; results bit-identical to rb_feat_pe10 + rb_sdf_mlp_ring.
extern "C" int rb_sdf_points_ring(const float* x, long M, float in_scale, const float* Wp, int mode, int scale_log2, float out_scale,
                                  float* out0, int n_workgroups, rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(x && Wp && out0, "null pointer");
  RB_REQUIRE(mode == 0 || mode == 1, "mode: 0 = signed distance, 1 = all 257 outputs (gradients: rb_sdf_value_grad_points)");
  const long rounds = (M + 127) / 128;
  if (n_workgroups <= 0) {
    static int cus = 0;
    if (!cus) {
      int dev = 0;
      hipDeviceProp_t prop;
      if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return rb::fail(__func__, "device query failed");
      cus = prop.multiProcessorCount;
    }
    n_workgroups = cus;
  }
  const unsigned grid = (unsigned)(rounds < n_workgroups ? rounds : n_workgroups);
  return rb::launch_sdf_ring8(mode, nullptr, x, in_scale, M, (const f4*)Wp, ldexpf(1.0f, -scale_log2), out_scale, out0, nullptr, grid,
                              (hipStream_t)stream);
}

namespace rb {
// forward half of the reverse-mode gradient (sdf_back.hip): all outputs of M points + the sigmoid blob
int launch_sdf_ring_store(const float* X, const float* xyz, float in_scale, long M, const f4* W, float us, float out_scale,
                          float* out0, f4* sig, unsigned grid, hipStream_t s) {
  if (g_sdf_ring_waves == 8 || !X) return launch_sdf_ring8(5, X, xyz, in_scale, M, W, us, out_scale, out0, sig, grid, s);
  unsigned* rw = range_flags() ? range_flags() + RB_RANGE_SDF : nullptr;
  hipLaunchKernelGGL(k_sdf_ring<5>, dim3(grid), dim3(256), 0, s, X, M, W, us, out_scale, 0.0f, out0, nullptr, rw, sig);
  return check_launch("k_sdf_ring<5>");
}
}  // namespace rb
