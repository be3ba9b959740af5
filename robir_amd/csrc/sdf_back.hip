// Reverse-mode gradient of the NeuS SDF network (model/neus_model.py:385-438, gradient() :440-452): the second half of
// rb_sdf_value_grad.  The forward-mode form (k_sdf_ring<3>) pushes three tangent columns next to every value row through all
// nine layers -- 4x the matrix work of the values alone.  Here the values run once (k_sdf_ring<5>, which also stores
// sigmoid(100 z) of every hidden pre-activation: the derivative of softplus_100), and ONE row vector per point runs back:
//     g7 = W8[0,:],   gz_l = g_l (.) sig_l,   g_{l-1} = W_l^T gz_l        (l = 7 .. 1),      d/d(features) = W_0^T gz_0,
// with the skip connection of layer 4 (input = [act(layer 3) | features] / sqrt 2) splitting W_4^T gz_4 into the part that
// continues (193 slots) and a direct contribution to the feature gradient.  The contraction with the Jacobian of the positional
// encoding is a small element-wise kernel (k_pe_grad below).  2x the matrix work of the values instead of 4x, and the backward
// epilogue is a multiply + split (no transcendentals).
//
// Same machine as sdf_ring.hip: one cyclic stream of 120 chunks (16 output slots x K) per round of 128 points through a 4-slot
// LDS ring filled by LDS-DMA three chunks ahead, weight fragments rolling through registers, operands as split f16 hi/lo pairs.
//   layer      B0   B1   B2   B3   B4   B5   B6   B7
//   matrix    W7^T W6^T W5^T W4^T W3^T W2^T W1^T W0^T
//   K         256  256  256  256  224  256  256  256
//   chunks     16   16   16   20   16   16   16    4      (B3: 13 continuing + 4 feature chunks + 3 empty ones, so that every
//   first       0   16   32   48   68   84  100  116       layer starts on ring slot 0 and sigmoid register set 0)
// The sigmoids come from the blob k_sdf_ring<5> wrote, 16 B per lane and chunk, in exactly the lane-local order the epilogue
// needs them.  They travel like the weights -- LDS-DMA rows into a four-set staging area, requested three iterations ahead --
// but NOT under the counted-vmcnt discipline: LDS-DMA rows retire out of order (measured: with weight rows from L2 and
// sigmoid rows from HBM in one wave's queue, `s_waitcnt vmcnt(8)` passed ~1e-6 of the time while a sigmoid row issued three
// iterations earlier was still in flight).  So the two kinds of rows never share a queue:
//   * waves 0, 1 request all weight rows (eight 1 KB slices per chunk each) and keep the counted wait of sdf_ring.hip -- rows of
//     one latency class, L2-resident weights, as in every ring kernel of this library;
//   * waves 2, 3 request all sigmoid rows (tile 0 / tile 1 of the four waves) and never wait on vmcnt;
//   * arrival of a sigmoid row is detected by its content: a consumer overwrites the 16 B it has read with a negative marker
//     (sigmoids are >= 0) and re-reads while it still finds one.
// The 32 rows of layer-7 sigmoids that open a round are ordinary loads into the operand registers that are idle after B7,
// issued at the end of a round and waited for in full at the top of the next.
#include "../../include/robir_hip.h"
#include "common.h"
#include "mlp_engine.h"
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace rb {

int launch_sdf_ring_store(const float* X, const float* xyz, float in_scale, long M, const f4* W, float us, float out_scale,
                          float* out0, f4* sig, unsigned grid, hipStream_t s);

constexpr int SB_SLOT_F4 = 1024;          // 16 KB per ring slot (K <= 256: four 4 KB DMA rows)
constexpr int SB_NS = 4;                  // ring slots; a chunk is requested SB_NS - 1 iterations before its MFMAs
constexpr int SB_SS = 8;                  // sigmoid staging sets; a chunk's rows are requested SB_SD iterations before its MFMAs
constexpr int SB_SD = 6;
#ifndef SB_WWAIT
#define SB_WWAIT (8 * (SB_D - 2))
#endif
#ifndef SB_FILL
#define SB_FILL 6
#endif
constexpr int SB_D = SB_NS - 1;
constexpr int SB_NCHUNK = 120;
constexpr long SB_SIG_ROUND_F4 = 125L * 2 * 256;     // float4s of the sigmoid blob per round (sdf_ring.hip)
constexpr float SB_GS = 64.0f;            // operand lift of the gradient rows (power of two)

__host__ __device__ constexpr int sb_K(int l) { return l == 4 ? 224 : 256; }
__host__ __device__ constexpr int sb_nch(int l) { return l == 3 ? 20 : (l == 7 ? 4 : 16); }
__host__ __device__ constexpr long sb_loff(int l) {   // float4 offset of layer l in the packed blob
  return l <= 3 ? 16L * chunk_f4(256) * l
                : (l == 4 ? 48L * chunk_f4(256) + 20L * chunk_f4(256)
                          : 68L * chunk_f4(256) + 16L * chunk_f4(224) + 16L * chunk_f4(256) * (l - 5));
}
constexpr long SB_BLOB_F4 = sb_loff(7) + 4L * chunk_f4(256);
// first chunk (stream index of sdf_ring.hip) of the forward layer whose sigmoids multiply the outputs of backward layer l
__host__ __device__ constexpr int sb_sig_first(int l) { return l == 0 ? 93 : (l == 1 ? 77 : (l == 2 ? 61 : (l == 3 ? 48 : (l == 4 ? 32 : (l == 5 ? 16 : 0))))); }

__device__ __forceinline__ void sb_dma16(const f4* gbase_uniform, unsigned lane_byte_off, unsigned lds_byte_uniform) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_byte_uniform), "v"(lane_byte_off),
               "s"(gbase_uniform)
               : "memory");
}
__device__ __forceinline__ void sb_store16(const float* base_uniform, unsigned lane_byte_off, f4 v) {
  // a store of more than 64 bits reads its data registers late: a VALU write to them needs wait states in between
  // (cdna ISA, manually inserted wait states) -- the compiler adds them for its own stores, not after inline assembly
  asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(lane_byte_off), "v"(v), "s"(base_uniform) : "memory");
}
template <int N>
__device__ __forceinline__ void sb_wait() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct SbAcc {
  f4 a[2];
};
#ifdef SB_DEBUG
__device__ unsigned long long g_sb_spins, g_sb_checks;
#endif
#ifdef SB_TRACE
__device__ unsigned long long g_sb_trace[4][4096];     // [wave][event]: cycle stamps of workgroup 0
#endif

// gfeat [rounds * 128][2][64]: d(sdf_raw)/d(feature slot) -- [.][0] the skip connection's share (layer 4), [.][1] layer 0's
__global__ __launch_bounds__(256, 1) void k_sdf_back(long M, const f4* __restrict__ Wb, const float* __restrict__ w8row, float us,
                                                      const f4* __restrict__ sig, float* __restrict__ gfeat,
                                                      unsigned* __restrict__ range_word) {
  __shared__ f4 ring[SB_NS * SB_SLOT_F4];             // 64 KB
  __shared__ f4 sstg[SB_SS][2][256];                  // 64 KB: sigmoids of chunks p .. p+7 (set = stream position & 7), two tiles
  constexpr f4 poison = {-1.f, -1.f, -1.f, -1.f};
  __shared__ f4 w8s[64];
  __shared__ f4 junk[2][64];                          // target of the copies a sigmoid wave issues only to keep the loop body uniform
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long nrounds = (M + 127) >> 7;
  if (tid < 64) w8s[tid] = reinterpret_cast<const f4*>(w8row)[tid];
#pragma unroll
  for (int i = 0; i < 2 * SB_SS; ++i) sstg[i >> 1][i & 1][tid] = poison;
  __syncthreads();
  if ((long)blockIdx.x >= nrounds) return;

  const float inv_sqrt2 = 0.70710678118654752440f;
  const float zf = us * (1.0f / SB_GS);               // un-scaling of a feature-gradient accumulator
  const unsigned ring_b = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)ring);
  const unsigned lane_off = (unsigned)tid * 16u;
  const unsigned lane16 = (unsigned)lane * 16u;
  const bool w_wave = wave < 2;                       // requests weight rows (else: sigmoid rows)
  // weight rows: wave ww copies the 1 KB slices 2ww, 2ww+1 of every 4 KB row
  const unsigned wslice = (unsigned)(wave & 1) * 2048u;            // byte offset of this wave's slices in a row (global and LDS)
  const unsigned stg_b = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)sstg);
  const unsigned stile = (unsigned)(wave & 1);        // sigmoid rows: wave 2 copies tile 0 of all four waves, wave 3 tile 1
  const unsigned junk_b = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)junk) + stile * 1024u;
  const unsigned gf_off = (unsigned)((wave * 32 + (lane & 15)) * 128 + 4 * g) * 4u;     // this lane's row / slot group in gfeat
  // byte offsets of the ring slots of chunks first .. first + SB_NS - 1 of a layer whose first chunk sits on slot r
  auto slots_from = [&](int r, unsigned (&st)[SB_NS]) {
#pragma unroll
    for (int j = 0; j < SB_NS; ++j) {
      const int v = r + j;
      st[j] = (unsigned)(v >= SB_NS ? v - SB_NS : v) * (SB_SLOT_F4 * 16u);
    }
  };
  const u4* ring_u = reinterpret_cast<const u4*>(ring) + lane;
  u4 wreg[16];
  unsigned sat = 0u;
  u4 xh[2][8], xl[2][8];               // operands of the current layer
  u4 yh[2][8], yl[2][8];               // ... of the next layer; during B7 the next round's layer-7 sigmoids land here

  f4 skeep[2];                         // sigmoids of the chunk in the epilogue, per tile
#ifdef SB_TRACE
  int tr_n = 0;
#endif
  auto mfma_kb = [&](int kb, SbAcc& acc) {
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
  // piece (tile, register pair) of a continuing chunk jb: g (.) sigmoid, lifted (the lift of the operands cancels: acc carries it)
  // arrival of the sigmoid rows of the chunk at stream position pos (both tiles of this wave) -> skeep; marks the slots empty.
  // Runs at the top of an iteration, outside the k-block loop, so that the loop body stays one basic block (below)
  auto take_sigmoids = [&](int pos) {
    // (an LDS-space pointer: through a generic one the re-read would be a FLAT load, which counts in vmcnt too and makes the
    // compiler drain the whole copy queue at every check)
    typedef volatile __attribute__((address_space(3))) f4 lds_f4;
    lds_f4* slot0 = (lds_f4*)&sstg[pos & (SB_SS - 1)][0][tid];
    lds_f4* slot1 = (lds_f4*)&sstg[pos & (SB_SS - 1)][1][tid];
    f4 s0 = *slot0, s1 = *slot1;
    // "not arrived" = some component still holds the negative marker.  fminf skips NaN operands, so a row of NaN sigmoids
    // (a NaN / inf input point: the value pass stores sigmoid(NaN)) counts as arrived -- `!(min >= 0)` would spin on it
    // for ever -- and the NaN then flows into that point's gradient like everywhere else in the library.
    while (__builtin_amdgcn_ballot_w64(fminf(fminf(fminf(s0[0], s0[1]), fminf(s0[2], s0[3])),
                                              fminf(fminf(s1[0], s1[1]), fminf(s1[2], s1[3]))) < 0.0f) != 0ull) {
#ifdef SB_DEBUG
      if (lane == 0) atomicAdd(&g_sb_spins, 1ull);
#endif
      __builtin_amdgcn_s_sleep(1);                  // a row has not arrived yet
      s0 = *slot0;
      s1 = *slot1;
    }
#ifdef SB_DEBUG
    if (lane == 0) atomicAdd(&g_sb_checks, 1ull);
#endif
    *slot0 = poison;                                // marks the slots empty for the rows that land here eight chunks on
    *slot1 = poison;
    skeep[0] = s0;
    skeep[1] = s1;
  };
  // piece (tile, register pair) of a continuing chunk jb: g (.) sigmoid, lifted (the lift of the operands cancels: acc carries it)
  auto hidden_piece = [&](const SbAcc& acc, int jb, int piece, float sa) {
    const int t = piece >> 1, q = piece & 1;
    const f4 s = skeep[t];
    unsigned hi, lo;
    split_pair_mix(acc.a[t][2 * q] * s[2 * q] * sa, acc.a[t][2 * q + 1] * s[2 * q + 1] * sa, hi, lo);
    yh[t][jb >> 1][(jb & 1) * 2 + q] = hi;
    yl[t][jb >> 1][(jb & 1) * 2 + q] = lo;
    sat = sat_acc(sat, hi);
  };
  // feature chunk c (slots 16c .. 16c+15) of share `which`: one 16 B store per tile
  auto feature_piece = [&](const SbAcc& acc, int c, int piece, int which, float sc, const float* gf_round) {
    const int t = piece >> 1, q = piece & 1;
    if (q == 1) {
      const float* base = gf_round;
      asm volatile("" : "+s"(base));
      sb_store16(base + t * 16 * 128 + which * 64 + 16 * c, gf_off, acc.a[t] * sc);
    }
  };

  // sigmoid block (two tiles x 4 KB) that the epilogue of the chunk at stream position pos multiplies with; positions past the
  // end belong to the next round; nullptr for chunks that take none (their staging set must stay marked empty)
  auto sig_src = [&](int pos, const f4* sr, const f4* srn) -> const f4* {
    const f4* base = pos >= SB_NCHUNK ? srn : sr;
    const int p = pos >= SB_NCHUNK ? pos - SB_NCHUNK : pos;
    const int idx = p < 16 ? 93 + p : (p < 32 ? 61 + p : (p < 48 ? 29 + p : (p < 61 ? p : (p < 68 ? -1 : (p < 84 ? p - 36 : (p < 100 ? p - 68 : (p < 116 ? p - 100 : -1)))))));
    return idx < 0 ? nullptr : base + (long)idx * 512;
  };
  // ---- one layer of the chunk stream (compile time: K, NCH chunks, EPI: 0 continuing, 1 = B3, 2 = B7; KF = K of the layer
  // that follows).  Run time: src_of(j) = packed chunk j counted from this layer's first (j runs SB_D past its last one, into
  // the next layer or the next round), st = ring slots of its chunks 0 .. SB_NS - 1 (then cyclic),
  // ... first = stream position of its chunk 0, sr / srn = sigmoid blob of this / the next round, gf = this round's gfeat,
  // esig = next round's layer-7 sigmoids (EPI 2 only).
  auto run_layer = [&](auto K_tag, auto NCH_tag, auto EPI_tag, auto KF_tag, auto src_of, const unsigned (&st)[SB_NS],
                       int first, const f4* sr, const f4* srn, const float* gf, const f4* esig) {
    constexpr int K = decltype(K_tag)::value, KB = K / 32, NCH = decltype(NCH_tag)::value, EPI = decltype(EPI_tag)::value;
    constexpr int KF = decltype(KF_tag)::value;
    constexpr int NS = EPI == 0 ? NCH : (EPI == 1 ? 13 : 0);      // chunks of this layer that take sigmoids
    const float sa = (EPI == 1 ? inv_sqrt2 : 1.0f) * us;
    asm volatile("" : "+s"(sr), "+s"(srn));
    SbAcc prev;
    auto epilogue = [&](const SbAcc& acc, int jb, int pc) {
      if constexpr (EPI == 2) {
        feature_piece(acc, jb, pc, 1, zf, gf);
      } else if constexpr (EPI == 1) {
        if (jb < 13) hidden_piece(acc, jb, pc, sa);
        else if (jb < 17) feature_piece(acc, jb - 13, pc, 0, zf * inv_sqrt2, gf);
      } else {
        hidden_piece(acc, jb, pc, sa);
      }
    };
#pragma unroll
    for (int jb = 0; jb < NCH; ++jb) {
      constexpr f4 zero = {0.f, 0.f, 0.f, 0.f};
      SbAcc acc;
      acc.a[0] = zero;
      acc.a[1] = zero;
      const bool empty = EPI == 1 && jb >= 17;                  // B3's padding chunks: ring cadence only
      // chunk jb+1 must have landed: the eight slices each of chunks jb+2 .. jb+SB_D-1 (the previous SB_D-2 iterations) are
      // the only younger requests in a weight wave's queue.  The stores of a feature chunk's epilogue may sit in that window too; they are not credited (a
      // store that retired early must not stand in for a row), which at worst waits for two slices more than necessary.
      // Sigmoid waves have nothing to wait for.  lgkmcnt: the markers written in the previous iteration are in LDS before the
      // barrier lets a sigmoid wave request the rows that replace them
#ifdef SB_TRACE
      if (blockIdx.x == 0 && lane == 0 && tr_n < 4090) g_sb_trace[wave][tr_n++] = __builtin_readcyclecounter();
#endif
      if (w_wave) sb_wait<SB_WWAIT>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef SB_TRACE
      if (blockIdx.x == 0 && lane == 0 && tr_n < 4090) g_sb_trace[wave][tr_n++] = __builtin_readcyclecounter();
#endif
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#ifdef SB_TRACE
      if (blockIdx.x == 0 && lane == 0 && tr_n < 4090) g_sb_trace[wave][tr_n++] = __builtin_readcyclecounter();
#endif
      if (jb > 0 && (EPI == 0 || (EPI == 1 && jb - 1 < 13))) take_sigmoids(first + jb - 1);
      const int KBn = (jb + 1 < NCH ? K : KF) / 32;             // next chunk: its fragments roll into wreg
      const u4* ring_n = reinterpret_cast<const u4*>(reinterpret_cast<const char*>(ring) + st[(jb + 1) % SB_NS]) + lane;
      // The eight copies this wave requests in this iteration, two in each of four k-block gaps, WITHOUT a branch on the wave's
      // role (the loop body must stay one basic block for the scheduler to interleave it): a weight wave takes its two 1 KB
      // slices of the four rows of chunk jb + SB_D, a sigmoid wave the four 1 KB rows (its tile, the four waves) of the chunk
      // SB_SD positions ahead and four copies into a scratch row (all eight where that chunk takes no sigmoids).
      const int spos = first + jb + SB_SD;
      const f4* sb2 = sig_src(spos, sr, srn);
      const bool s_real = sb2 != nullptr;
      const f4* dma_src = w_wave ? src_of(jb + SB_D) + 4 + (wave & 1) * 128 : (s_real ? sb2 : sr) + stile * 256;
      const unsigned dma_dst = w_wave ? ring_b + st[(jb + SB_D) % SB_NS] + wslice
                                      : (s_real ? stg_b + (unsigned)(spos & (SB_SS - 1)) * 8192u + stile * 4096u : junk_b);
      auto dma_slot = [&](int k) {
        // weight wave: row k/2, slice k&1 of its pair; sigmoid wave: consumer wave k (k < 4), scratch otherwise
        const f4* src = dma_src + (w_wave ? (k >> 1) * 256 + (k & 1) * 64 : (k < 4 ? k * 64 : 0));
        const unsigned dst = w_wave ? dma_dst + (unsigned)((k >> 1) * 4096 + (k & 1) * 1024)
                                    : (k < 4 ? dma_dst + (s_real ? (unsigned)(k * 1024) : 0u) : junk_b);
        sb_dma16(src, lane16, dst);
      };
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        if (!empty) mfma_kb(kb, acc);
        if (kb < KBn) {
          wreg[2 * kb] = ring_n[(2 * kb) * 64];
          wreg[2 * kb + 1] = ring_n[(2 * kb + 1) * 64];
        }
        if (jb > 0) {
#pragma unroll
          for (int pc = 0; pc < 4; ++pc)
            if ((pc * KB) / 4 == kb) epilogue(prev, jb - 1, pc);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (((i * KB) / 4 + 1 < KB ? (i * KB) / 4 + 1 : KB - 1) == kb) {
            dma_slot(2 * i);
            dma_slot(2 * i + 1);
          }
        // One wave per SIMD issues in order: six MFMAs back to back (two dependent chains of three) block the wave for their
        // whole 96 cycles, and the epilogue's VALU work then issues behind them -- matrix and vector time add up.  Ask the
        // scheduler for one MFMA followed by a handful of other instructions, six times: the vector work issues in the
        // 12 idle issue cycles behind each MFMA instead.
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x106, SB_FILL, 0);     // VALU | SALU | DS read
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int kb = KB; kb < KBn; ++kb) {                      // next chunk is wider (K 224 -> 256)
        wreg[2 * kb] = ring_n[(2 * kb) * 64];
        wreg[2 * kb + 1] = ring_n[(2 * kb + 1) * 64];
      }
      __builtin_amdgcn_sched_barrier(0);
      prev = acc;
    }
    if constexpr (EPI == 0) take_sigmoids(first + NCH - 1);
#pragma unroll
    for (int pc = 0; pc < 4; ++pc) epilogue(prev, NCH - 1, pc);
    if constexpr (EPI == 2) {
      // next round's layer-7 sigmoids -> y (idle now): chunk 2kb -> yh[.][kb], 2kb+1 -> yl[.][kb].  Ordinary loads: they do not
      // retire in order with the LDS-DMA rows (a counted wait that passed with these in its window let a round start on
      // operands that had not arrived), so they are issued after the last counted wait of the round and the next round opens
      // with a full wait
      asm volatile("" ::: "memory");
#pragma unroll
      for (int kb = 0; kb < 8; ++kb)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          yh[t][kb] = __builtin_bit_cast(u4, esig[((2 * kb) * 2 + t) * 256 + tid]);
          yl[t][kb] = __builtin_bit_cast(u4, esig[((2 * kb + 1) * 2 + t) * 256 + tid]);
        }
      asm volatile("" ::: "memory");
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using K256 = std::integral_constant<int, 256>;
  using K224 = std::integral_constant<int, 224>;
  auto y_to_x = [&]() {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) {
        xh[t][kb] = yh[t][kb];
        xl[t][kb] = yl[t][kb];
      }
  };
  auto sig_of = [&](long r) { return sig + (r < nrounds ? r : (long)blockIdx.x) * SB_SIG_ROUND_F4; };

  // ---- prologue: ring start; the first round's layer-7 sigmoids (-> y) and those of chunks 0, 1 of B0
  long round = blockIdx.x;
  {
    const f4* s0 = sig_of(round);
    if (w_wave) {
#pragma unroll
      for (int c = 0; c < SB_D; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            sb_dma16(Wb + (long)c * chunk_f4(256) + 4 + i * 256 + j * 64, wslice + lane16,
                     ring_b + (unsigned)c * (SB_SLOT_F4 * 16u) + wslice + (unsigned)(i * 4096 + j * 1024));
    }
#pragma unroll
    for (int kb = 0; kb < 8; ++kb)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        yh[t][kb] = __builtin_bit_cast(u4, s0[((109 + 2 * kb) * 2 + t) * 256 + tid]);
        yl[t][kb] = __builtin_bit_cast(u4, s0[((109 + 2 * kb + 1) * 2 + t) * 256 + tid]);
      }
    if (!w_wave) {
#pragma unroll
      for (int c = 0; c < SB_SD; ++c)
#pragma unroll
        for (int w = 0; w < 4; ++w)
          sb_dma16(s0 + ((sb_sig_first(0) + c) * 2) * 256 + stile * 256 + w * 64, lane16,
                   stg_b + (unsigned)(c * 8192 + w * 1024) + stile * 4096u);
    }
  }
  sb_wait<0>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int i = 0; i < 16; ++i) wreg[i] = ring_u[i * 64];        // chunk 0: K = 256, eight k-blocks

  constexpr long CF256 = chunk_f4(256), CF224 = chunk_f4(224);
  for (; round < nrounds; round += gridDim.x) {
    const f4* sr = sig_of(round);
    const f4* srn = sig_of(round + gridDim.x);
    const float* gf = gfeat + round * (128L * 128);
    sb_wait<0>();                  // the layer-7 sigmoids in y (and, with them, the first copy rows of this round)
    // ---- operands of B0: W8[0, k] * sigmoid_7[k], lifted (y holds the sigmoids of layer 7, chunk 2kb | 2kb+1 per k-block)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int kb = 0; kb < 8; ++kb)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const f4 s = __builtin_bit_cast(f4, e == 0 ? yh[t][kb] : yl[t][kb]);
          const f4 w = w8s[(2 * kb + e) * 4 + g];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            unsigned hi, lo;
            split_pair_mix(w[2 * q] * s[2 * q] * SB_GS, w[2 * q + 1] * s[2 * q + 1] * SB_GS, hi, lo);
            xh[t][kb][e * 2 + q] = hi;
            xl[t][kb][e * 2 + q] = lo;
            sat = sat_acc(sat, hi);
          }
        }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int rep = 0; rep < 3; ++rep) {        // B0, B1, B2: one copy of the code (shared with B5, B6 below)
      const f4* wl = Wb + sb_loff(0) + (long)rep * 16 * CF256;
      asm volatile("" : "+s"(wl));
      unsigned st[SB_NS];
      slots_from((16 * rep) % SB_NS, st);
      // chunks 0 .. 67 of the stream (B0 .. B3) are equally sized and contiguous: no tail cases
      run_layer(K256{}, std::integral_constant<int, 16>{}, I0{}, K256{}, [&](int j) { return wl + (long)j * CF256; }, st, 16 * rep, sr, srn, gf, sr);
      y_to_x();
    }
    {   // B3: W4^T -- 13 continuing chunks (x 1/sqrt 2), 4 feature chunks, 3 empty; followed by B4 (K = 224)
      const f4* w3 = Wb + sb_loff(3);
      const f4* w4 = Wb + sb_loff(4);
      asm volatile("" : "+s"(w3), "+s"(w4));
      unsigned st[SB_NS];
      slots_from(48 % SB_NS, st);
      run_layer(K256{}, std::integral_constant<int, 20>{}, I1{}, K224{},
                [&](int j) { return j < 20 ? w3 + (long)j * CF256 : w4 + (long)(j - 20) * CF224; }, st, 48, sr, srn,
                gf, sr);
#pragma unroll
      for (int t = 0; t < 2; ++t) {           // slots 208 .. 223 of B4's input are padding
        yh[t][6][2] = 0u; yl[t][6][2] = 0u;
        yh[t][6][3] = 0u; yl[t][6][3] = 0u;
      }
      y_to_x();
    }
    {   // B4: W3^T (K = 224); followed by B5
      const f4* w4 = Wb + sb_loff(4);
      const f4* w5 = Wb + sb_loff(5);
      asm volatile("" : "+s"(w4), "+s"(w5));
      unsigned st[SB_NS];
      slots_from(68 % SB_NS, st);
      run_layer(K224{}, std::integral_constant<int, 16>{}, I0{}, K256{},
                [&](int j) { return j < 16 ? w4 + (long)j * CF224 : w5 + (long)(j - 16) * CF256; }, st, 68, sr, srn,
                gf, sr);
      y_to_x();
    }
#pragma unroll 1
    for (int rep = 0; rep < 2; ++rep) {        // B5, B6 (chunks 84 .. 115; B7 follows contiguously, then the stream wraps)
      const f4* wl = Wb + sb_loff(5) + (long)rep * 16 * CF256;
      const f4* wrap = Wb - (long)(36 - 16 * rep) * CF256;          // chunk j >= 36 - 16 rep of this layer is chunk j - that of B0
      asm volatile("" : "+s"(wl), "+s"(wrap));
      const int nwrap = 36 - 16 * rep;
      unsigned st[SB_NS];
      slots_from((84 + 16 * rep) % SB_NS, st);
      run_layer(K256{}, std::integral_constant<int, 16>{}, I0{}, K256{},
                [&](int j) { return (j >= nwrap ? wrap : wl) + (long)j * CF256; }, st, 84 + 16 * rep, sr, srn, gf, sr);
      y_to_x();
    }
    {   // B7: W0^T -> feature gradient; what follows it is the next round's stream, and it requests the next round's first
        // sigmoids (B0 chunks 0, 1 through `snext`, layer 7 through `esig`)
      const f4* w7 = Wb + sb_loff(7);
      const f4* w0 = Wb;
      asm volatile("" : "+s"(w7), "+s"(w0));
      unsigned st[SB_NS];
      slots_from(116 % SB_NS, st);
      run_layer(K256{}, std::integral_constant<int, 4>{}, I2{}, K256{},
                [&](int j) { return j < 4 ? w7 + (long)j * CF256 : w0 + (long)(j - 4) * CF256; }, st, 116, sr,
                srn, gf, srn + 109L * 512);
    }
  }
  range_report(sat, range_word);
  sb_wait<0>();
  __syncthreads();
}

// grad[m, c] = grad_scale * sum_f (gfeat[m,0,f] + gfeat[m,1,f]) * dPE_f/dx_c with the PE values of X [M,64]
// (model/embedder.py:17-38: [x | sin(2^k x) | cos(2^k x)], k = 0..9): d sin(2^k x_c) = 2^k cos(2^k x_c), d cos = -2^k sin.
__global__ void k_pe_grad(const float* __restrict__ gfeat, const float* __restrict__ X, long M, float grad_scale,
                          float* __restrict__ grad) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= 3 * M) return;
  const long m = i / 3;
  const int c = (int)(i - 3 * m);
  const float* gf = gfeat + m * 128;
  const float* x = X + m * 64;
  float acc = gf[c] + gf[64 + c];
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    const float f = (float)(1 << k);
    const int is = 3 + 6 * k + c, ic = is + 3;
    acc += f * ((gf[is] + gf[64 + is]) * x[ic] - (gf[ic] + gf[64 + ic]) * x[is]);
  }
  grad[i] = acc * grad_scale;
}

// The same contraction with the encoding recomputed from the points (the fused form keeps no feature rows): the same sincosf of
// the same argument as the features the value pass used.
__global__ void k_pe_grad_points(const float* __restrict__ gfeat, const float* __restrict__ xyz, float in_scale, long M,
                                 float grad_scale, float* __restrict__ grad) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= 3 * M) return;
  const long m = i / 3;
  const int c = (int)(i - 3 * m);
  const float* gf = gfeat + m * 128;
  const float a = xyz[i] * in_scale;
  float acc = gf[c] + gf[64 + c];
#pragma unroll 1
  for (int k = 0; k < 10; ++k) {
    const float f = (float)(1 << k);
    float sn, cs;
    sincosf(a * f, &sn, &cs);
    const int is = 3 + 6 * k + c, ic = is + 3;
    acc += f * ((gf[is] + gf[64 + is]) * cs - (gf[ic] + gf[64 + ic]) * sn);
  }
  grad[i] = acc * grad_scale;
}

}  // namespace rb

using namespace rb;

#ifdef RB_LEGACY
extern "C" long rb_sdf_value_grad_scratch_floats(long M) {
  const long rounds = (M + 127) / 128;
  return rounds * (SB_SIG_ROUND_F4 * 4 + 128L * 128);
}
#endif  // RB_LEGACY

// X != nullptr: feature rows; X == nullptr: points xyz[M,3] x in_scale with the encoding fused into the value pass
static int sdf_value_grad_impl(const float* X, const float* xyz, float in_scale, long M, const float* Wp, const float* Wb,
                               const float* w8row, int scale_log2, float out_scale, float grad_scale, float* out0, float* grad,
                               float* scratch, int n_workgroups, rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE((X || xyz) && Wp && Wb && w8row && out0 && grad && scratch, "null pointer");
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
  const float us = ldexpf(1.0f, -scale_log2);
  hipStream_t s = (hipStream_t)stream;
  f4* sig = reinterpret_cast<f4*>(scratch);
  float* gfeat = scratch + rounds * SB_SIG_ROUND_F4 * 4;
  if (int rc = launch_sdf_ring_store(X, xyz, in_scale, M, (const f4*)Wp, us, out_scale, out0, sig, grid, s)) return rc;
  unsigned* rw = range_flags() ? range_flags() + RB_RANGE_SDF : nullptr;
  hipLaunchKernelGGL(k_sdf_back, dim3(grid), dim3(256), 0, s, M, (const f4*)Wb, w8row, us, sig, gfeat, rw);
  if (int rc = check_launch("k_sdf_back")) return rc;
#ifdef SB_DEBUG
  {
    unsigned long long a = 0, b = 0;
    hipMemcpyFromSymbol(&a, HIP_SYMBOL(g_sb_spins), 8);
    hipMemcpyFromSymbol(&b, HIP_SYMBOL(g_sb_checks), 8);
    fprintf(stderr, "sdf_back: %llu sigmoid-row checks, %llu spins so far\n", b, a);
  }
#endif
#ifdef SB_TRACE
  {
    static unsigned long long h[4][4096];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(g_sb_trace), sizeof(h));
    FILE* f = fopen("gpurun_out/sb_trace.txt", "w");
    if (f) {
      for (int w = 0; w < 4; ++w) {
        for (int i = 0; i < 4090; ++i) fprintf(f, "%llu ", h[w][i]);
        fprintf(f, "\n");
      }
      fclose(f);
    }
  }
#endif
  const long n = 3 * M;
  if (X) {
    hipLaunchKernelGGL(k_pe_grad, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, gfeat, X, M, grad_scale, grad);
  } else {
    hipLaunchKernelGGL(k_pe_grad_points, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, gfeat, xyz, in_scale, M, grad_scale, grad);
  }
  return check_launch("k_pe_grad");
}

// The same op at the reference's precision (exact policy): k_sdf_mlp<5> (f32-input MFMA, sigmoid tiles out), k_sdf_back_f32,
// k_pe_grad_points.  scratch: rb_sdf_value_grad_f32_scratch_floats(M) floats.
extern "C" long rb_sdf_value_grad_f32_scratch_floats(long M) {
  const long tiles = (M + 127) / 128 * 8;
  return tiles * (8L * 16 * 64 * 4) + ((M + 127) / 128 * 128) * 128L;
}
extern "C" int rb_sdf_value_grad_f32_points(const float* x, long M, float in_scale, const float* Wp, const float* Wt, const float* w8row,
                                            float out_scale, float grad_scale, float* out0, float* grad, float* scratch,
                                            rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(x && Wp && Wt && w8row && out0 && grad && scratch, "null pointer");
  hipStream_t s = (hipStream_t)stream;
  const long tiles = (M + 127) / 128 * 8;
  float* sig = scratch;
  float* gfeat = scratch + tiles * (8L * 16 * 64 * 4);
  int rc = launch_sdf_f32_store(x, M, in_scale, Wp, out_scale, out0, sig, s);
  if (rc) return rc;
  rc = launch_sdf_back_f32(sig, M, Wt, w8row, gfeat, s);
  if (rc) return rc;
  const long n = 3 * M;
  hipLaunchKernelGGL(k_pe_grad_points, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, gfeat, x, in_scale, M, grad_scale, grad);
  return check_launch("k_pe_grad_points");
}

// ... with both passes on exact three-piece operands: same scratch.  two_tile = 0: k_sdf_x6<5> (csrc/sdf_x6.hip, Wp = packing.pack_sdf_x6)
// + k_sdf_back_x6 (csrc/sdf_back_x6.hip, Wt = packing.pack_sdf_back_x6), one 16-row tile per wave; two_tile = 1: k_sdf_x6t<5> +
// k_sdf_back_x6t (csrc/sdf_x6t.hip, sdf_back_x6t.hip; Wt = packing.pack_sdf_back_x6(two_tile=True): W3^T's K padded to 256).
extern "C" int rb_sdf_value_grad_x6_points(const float* x, long M, float in_scale, const float* Wp, const float* Wt, const float* w8row,
                                           float out_scale, float grad_scale, float* out0, float* grad, float* scratch, int two_tile,
                                           rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(x && Wp && Wt && w8row && out0 && grad && scratch, "null pointer");
  hipStream_t s = (hipStream_t)stream;
  const long tiles = (M + 127) / 128 * 8;
  float* sig = scratch;
  float* gfeat = scratch + tiles * (8L * 16 * 64 * 4);
  int rc = two_tile ? launch_sdf_x6t(x, M, in_scale, Wp, 5, out_scale, out0, sig, 0, s) : launch_sdf_x6_store(x, M, in_scale, Wp, out_scale, out0, sig, s);
  if (rc) return rc;
  rc = two_tile ? launch_sdf_back_x6t(sig, M, Wt, w8row, gfeat, s) : launch_sdf_back_x6(sig, M, Wt, w8row, gfeat, s);
  if (rc) return rc;
  const long n = 3 * M;
  hipLaunchKernelGGL(k_pe_grad_points, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, gfeat, x, in_scale, M, grad_scale, grad);
  return check_launch("k_pe_grad_points");
}

#ifdef RB_LEGACY
extern "C" int rb_sdf_value_grad(const float* X, long M, const float* Wp, const float* Wb, const float* w8row, int scale_log2,
                                 float out_scale, float grad_scale, float* out0, float* grad, float* scratch, int n_workgroups,
                                 rb_stream_t stream) {
  RB_REQUIRE(X || M <= 0, "null pointer");
  return sdf_value_grad_impl(X, nullptr, 1.0f, M, Wp, Wb, w8row, scale_log2, out_scale, grad_scale, out0, grad, scratch, n_workgroups,
                             stream);
}

extern "C" int rb_sdf_value_grad_points(const float* x, long M, float in_scale, const float* Wp, const float* Wb, const float* w8row,
                                        int scale_log2, float out_scale, float grad_scale, float* out0, float* grad, float* scratch,
                                        int n_workgroups, rb_stream_t stream) {
  RB_REQUIRE(x || M <= 0, "null pointer");
  return sdf_value_grad_impl(nullptr, x, in_scale, M, Wp, Wb, w8row, scale_log2, out_scale, grad_scale, out0, grad, scratch,
                             n_workgroups, stream);
}
#endif  // RB_LEGACY
