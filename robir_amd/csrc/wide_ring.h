// The 512-wide nets on the chunk-stream machine -- split precision, round 3.
//   * CESR normal_net / shadow_net (model/cesr_net.py via implicit_differentiable_renderer.py: eight softplus(100) layers of 512 with
//     a skip connection into layer 4, inputs PE10(x) or [PE10(x) | one-hot label]), kernels k_softplus512_h3 of the first generation;
//   * SparseAE encoder / decoder and IndirctIllumNetwork (four hidden layers of 512, leaky 0.2 / relu), k_wide_mlp_h3.
// The first-generation kernels stream weights global -> VGPR -> LDS with a __syncthreads per chunk and re-stream the whole net
// (3.7 / 1.7 MB) for every 64 rows: 0.23 of the split-precision bound, 30 % of BASELINE config 5
// (profiles/r03_config5_kernel_stats.md).  Here, as in sdf_ring8.hip / color_ring8.hip: persistent workgroups, the net as ONE cyclic
// stream of 16-neuron chunks through a 4-slot LDS ring filled by LDS-DMA three chunks ahead under counted waits, one s_barrier per
// chunk, the activation + hi/lo split of chunk j in the issue slots between the MFMAs of chunk j+1, straight into the next layer's
// operand registers.  A 512-wide layer's operands are 128 registers and so are the next layer's: one wave per SIMD (four waves, one
// 16-row tile each, 512 registers), not two.  The bias of a chunk rides the stream too (a 256-byte LDS-DMA of the chunk's head into
// a 4-slot bias ring): a bias table of the whole net (16 KB) would not fit beside 4 x 34 KB of ring and the encoder's scratch.
// Same products in the same order, same lifts, same epilogue arithmetic as the first generation: bit-identical outputs.
#pragma once
#include "../../include/robir_hip.h"
#include "common.h"
#include "mlp_engine.h"
#include <type_traits>

namespace rb {

constexpr int WR_SLOT_B = 34 * 1024;          // K = 544: 34 KB of fragments
#ifndef WR_FD
#define WR_FD 3
#endif
#ifndef WR_BS
#define WR_BS 2          // k-blocks per batch of fragment reads
#endif
#ifndef WR_DB
#define WR_DB 2          // batches the fragment reads run ahead of the MFMAs
#endif

// ---- net descriptions: K (padded inputs) and chunks (16 output neurons each) per layer
template <int K0P, int N3P>
struct CesrNet {       // model/cesr_net.py (normal_net: K0P 64, N3P 464, 3 outputs; shadow_net: 192, 336, 2 outputs)
  static constexpr int L = 9, SKIP = 3, ACT = ACT_SOFTPLUS100_FAST;
  static constexpr float AS = 64.0f;
  static constexpr int K0 = K0P, N3 = N3P;
  __host__ __device__ static constexpr int K(int l) { return l == 0 ? K0P : (l == 4 ? 544 : 512); }
  __host__ __device__ static constexpr int NCH(int l) { return l == 3 ? N3P / 16 : (l == 8 ? 1 : 32); }
};
template <bool ENC>
struct WideNet {       // SparseAE encoder (32 outputs, leaky 0.2) / decoder + indirect-illumination net (144, relu)
  static constexpr int L = 5, SKIP = -1, ACT = ENC ? ACT_LEAKY02 : ACT_RELU;
  static constexpr float AS = 16.0f;
  static constexpr int NO = ENC ? 32 : 144;
  __host__ __device__ static constexpr int K(int l) { return l == 0 ? 64 : 512; }
  __host__ __device__ static constexpr int NCH(int l) { return l == 4 ? NO / 16 : 32; }
};
template <class Net>
__host__ __device__ constexpr int wr_nchunk() {
  int n = 0;
  for (int l = 0; l < Net::L; ++l) n += Net::NCH(l);
  return n;
}
template <class Net>
__host__ __device__ constexpr int wr_cbase(int l) {
  int n = 0;
  for (int i = 0; i < l; ++i) n += Net::NCH(i);
  return n;
}
// stream position (may run past the end: the stream is cyclic) -> layer
template <class Net>
__host__ __device__ constexpr int wr_layer_of(int c) {
  constexpr int N = wr_nchunk<Net>();
  if (c >= N) c -= N;
  int l = 0, first = 0;
  for (int i = 0; i < Net::L - 1; ++i) {
    first += Net::NCH(i);
    if (c >= first) l = i + 1;
  }
  return l;
}
template <class Net>
__host__ __device__ constexpr long wr_coff(int c) {            // float4 offset of chunk c (bias first) in the packed blob
  constexpr int N = wr_nchunk<Net>();
  if (c >= N) c -= N;
  long off = 0;
  int first = 0, base = 0, kl = Net::K(0);
  for (int i = 0; i < Net::L - 1; ++i) {
    first += Net::NCH(i);
    if (c >= first) {
      off += (long)Net::NCH(i) * chunk_f4(Net::K(i));
      base = first;
      kl = Net::K(i + 1);
    }
  }
  return off + (long)(c - base) * chunk_f4(kl);
}
// 1 KB fragment copies per wave and chunk: K / 64 (K = 544: 8.5 -> 9, the ninth of waves 2, 3 repeats that of waves 0, 1), + 1 for
// the bias
__host__ __device__ constexpr int wr_np(int K) { return K / 64 + 1 + (K == 544 ? 1 : 0); }

__device__ __forceinline__ void wr_dma16(const f4* gbase_uniform, unsigned lane_byte_off, unsigned lds_byte_uniform) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_byte_uniform), "v"(lane_byte_off),
               "s"(gbase_uniform)
               : "memory");
}
__device__ __forceinline__ void wr_dma4(const f4* gbase_uniform, unsigned lane_byte_off, unsigned lds_byte_uniform) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(lds_byte_uniform), "v"(lane_byte_off),
               "s"(gbase_uniform)
               : "memory");
}
template <int N>
__device__ __forceinline__ void wr_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// Copies of one chunk by one wave, in UNITS of one LDS-DMA block each (one M0 write, then 1-4 instructions whose immediate offsets
// advance the global and the LDS address together): unit 0 = the bias head (256 B), then the wave's contiguous span of the chunk's
// 1 KB fragment slices (K/64 per wave: wave v copies slices v K/64 ...) four at a time, and for K = 544 (34 slices) one more: slice
// 32 + (v & 1), twice over.  Instructions per chunk and wave (the counted waits): wr_np(K).
__host__ __device__ constexpr int wr_nsw(int K) { return K / 64; }                       // slices per wave (K = 544: 8, + the extra unit)
__host__ __device__ constexpr int wr_units(int K) { return 1 + (wr_nsw(K) + 3) / 4 + (K == 544 ? 1 : 0); }
template <int NPC>
__device__ __forceinline__ void wr_dma_block(const f4* gbase_uniform, unsigned lane_byte_off, unsigned lds_byte_uniform) {
  static_assert(NPC >= 1 && NPC <= 4, "");
  if constexpr (NPC == 1)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_byte_uniform), "v"(lane_byte_off), "s"(gbase_uniform) : "memory");
  else if constexpr (NPC == 2)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024" ::"s"(lds_byte_uniform), "v"(lane_byte_off), "s"(gbase_uniform) : "memory");
  else if constexpr (NPC == 3)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048" ::"s"(lds_byte_uniform), "v"(lane_byte_off), "s"(gbase_uniform) : "memory");
  else
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072" ::"s"(lds_byte_uniform), "v"(lane_byte_off), "s"(gbase_uniform) : "memory");
}
// per-lane global byte offsets of the units (from the chunk's first byte), set up once per kernel
template <int K>
__device__ __forceinline__ void wr_copy_unit(int u, const f4* src_chunk, unsigned lane4, unsigned lane16, unsigned bias_dst,
                                             unsigned slot_dst, int wave) {
  constexpr int NSW = wr_nsw(K);
  if (u == 0) {
    wr_dma4(src_chunk, lane4, bias_dst);
  } else if (K == 544 && u == 3) {
    const unsigned sb = (32u + (unsigned)(wave & 1)) * 1024u;
    wr_dma_block<1>(src_chunk + 4 + sb / 16, lane16, slot_dst + sb);
  } else {
    const unsigned sb = ((unsigned)wave * NSW + 4u * (u - 1)) * 1024u;      // first slice of this block
    const f4* src = src_chunk + 4 + sb / 16;
    if (NSW - 4 * (u - 1) >= 4) wr_dma_block<4>(src, lane16, slot_dst + sb);
    else if (NSW - 4 * (u - 1) == 3) wr_dma_block<3>(src, lane16, slot_dst + sb);
    else if (NSW - 4 * (u - 1) == 2) wr_dma_block<2>(src, lane16, slot_dst + sb);
    else wr_dma_block<1>(src, lane16, slot_dst + sb);
  }
}

// INPUT: 0 = PE10(x) (CESR normal_net), 1 = [PE10(x) | one-hot label] rows (CESR shadow_net: row = point * n_label + label),
//        2 = [PE10(x) | extra] (SparseAE encoders: extra = 0; indirect illumination: extra = hdr_shift)
//        3 = feature rows X[M,64] as rb_feat_pe10 writes them (the auto-encoders whose embedded vector is perturbed)
template <class Net, int INPUT>
__global__ __launch_bounds__(256, 1) void k_wide_ring(const float* __restrict__ X, const float* __restrict__ extra, long M, int n_label,
                                                       const f4* __restrict__ Wp, float us, int n_out, float* __restrict__ Y,
                                                       unsigned* __restrict__ range_word) {
  constexpr int L = Net::L, NCHUNK = wr_nchunk<Net>(), K0 = Net::K(0);
  constexpr bool HAS_SKIP = Net::SKIP >= 0;
  constexpr float AS = Net::AS;
  __shared__ f4 ring[4 * WR_SLOT_B / 16];              // 136 KB
  __shared__ f4 bias_ring[4 * 16];                     // 4 x 256 B (the first 64 B of each are the bias)
  __shared__ float pe_scratch[4 * 16 * 64];            // 16 KB: the encoder's exchange rows (mlp_engine.h)
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long nrounds = (M + 63) >> 6;
  if ((long)blockIdx.x >= nrounds) return;

  const float zs = us * (1.0f / AS);
  const float inv_sqrt2 = 0.70710678118654752440f;
  const unsigned ring_b = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)ring);
  const unsigned bias_b = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)bias_ring);
  const unsigned lane16 = (unsigned)lane * 16u, lane4 = (unsigned)lane * 4u;
  // ring slot of a chunk: slot_b[j & 3] for chunk j of the CURRENT layer (the table is rotated by the layer's chunk count at its end)
  unsigned slot_b[4] = {0u, (unsigned)WR_SLOT_B, 2u * WR_SLOT_B, 3u * WR_SLOT_B};
  unsigned bslot_b[4] = {0u, 256u, 512u, 768u};
  unsigned sat = 0u;
  u4 xh[17], xl[17];                   // operands of the current layer (K <= 544), one tile
  u4 yh[16], yl[16];                   // ... of the next layer
  constexpr int SK0 = HAS_SKIP ? Net::NCH(HAS_SKIP ? Net::SKIP : 0) / 2 : 16;    // first k-block of the skip layer that holds net inputs
  u4 skh[17 - SK0], skl[17 - SK0];     // skip net: the input part [x0 | 0] / sqrt 2 of the skip layer's operands, once per round
  long rrow = 0;

  // hidden activation of an MFMA result -> lifted operand value.  zs and AS are powers of two: relu(z zs) AS = relu(z us) bit for bit.
  auto hidden_val = [&](float z) {
    if constexpr (Net::ACT == ACT_RELU) return fmaxf(z * us, 0.f);
    else return act_fn<Net::ACT>(z * zs) * AS;
  };
  auto put_pair = [&](float v0, float v1, u4& dh, u4& dl, int q) {
    unsigned hi, lo;
    split_pair_mix(v0, v1, hi, lo);
    dh[q] = hi;
    dl[q] = lo;
    sat = sat_acc(sat, hi);
  };
  // this round's rows -> operands of layer 0 (and the input part of the skip layer's)
  auto load_layer0 = [&]() {
    const long row = rrow;
    float x0[K0 / 4];
    if constexpr (INPUT == 1) {
      const bool ok = row < M;
      const int label = ok ? (int)(row % n_label) : -1;
      float enc[16];
      load_features_pe10x(X, nullptr, row, M, lane, pe_scratch + wave * 1024, enc, n_label);
#pragma unroll
      for (int kb = 0; kb < K0 / 16; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int k = kb * 16 + 4 * g + r;
          float e = kb < 4 ? enc[(kb < 4 ? kb : 0) * 4 + r] : 0.f;
          if (k == 63) e = 0.f;                       // column 63 is padding; the one-hot block starts here
          if (k >= 63 && k - 63 == label) e = 1.f;
          x0[kb * 4 + r] = e;
        }
    } else if constexpr (INPUT == 3) {
      static_assert(K0 == 64, "feature rows are 64 wide");
      load_features<64>(X, row, M, lane, x0);
    } else {
      static_assert(K0 == 64, "encoded inputs are 64 wide");
      load_features_pe10x(X, INPUT == 2 ? extra : nullptr, row, M, lane, pe_scratch + wave * 1024, x0);
    }
#pragma unroll
    for (int kb = 0; kb < K0 / 32; ++kb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = (2 * kb + (q >> 1)) * 4 + (q & 1) * 2;
        put_pair(x0[i] * AS, x0[i + 1] * AS, xh[kb], xl[kb], q);
      }
    if constexpr (HAS_SKIP) {     // skip layer operands [softplus(h3) | x0 | 0] / sqrt 2: the x0 blocks, once per round
      constexpr int B3 = Net::N3 / 16;         // 16-blocks of the skip layer's own part
      static_assert(Net::K(4) == 544 && (B3 + K0 / 16 == 33) && (B3 & 1) == 1, "skip layer: 528 real inputs padded to 544");
      skh[0] = u4{0u, 0u, 0u, 0u};
      skl[0] = u4{0u, 0u, 0u, 0u};
#pragma unroll
      for (int b = B3; b < 34; ++b)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const int q = (b & 1) * 2 + p, i = (b - B3) * 4 + p * 2;
          if (b < 33) put_pair(x0[(b < 33 ? i : 0)] * inv_sqrt2 * AS, x0[(b < 33 ? i : 0) + 1] * inv_sqrt2 * AS, skh[(b >> 1) - SK0], skl[(b >> 1) - SK0], q);
          else put_pair(0.f, 0.f, skh[(b >> 1) - SK0], skl[(b >> 1) - SK0], q);
        }
    }
  };

  // ---- one layer.  Compile time: the layer index LI (K, chunks, epilogue and the three chunks that follow it in the stream all follow
  // from it); layers with the same shape and the same successors share one instance (LI = representative, `cb` = stream index of the
  // actual first chunk, run time).
  auto run_layer = [&](auto LI_tag, int cb) {
    constexpr int LI = decltype(LI_tag)::value;
    constexpr int K = Net::K(LI), KB = K / 32, NCH = Net::NCH(LI), NP = wr_np(K), CB = wr_cbase<Net>(LI);
    constexpr bool LAST = LI == L - 1, SKIPOUT = LI == Net::SKIP;
    // Per chunk jb (stream position cb + jb): 3 KB MFMAs on one accumulator; at k-block KB/2 this wave's copies of chunk jb+1 are
    // waited for (those of jb+2 stay in flight) and ONE s_barrier makes the whole chunk visible -- and certifies that every wave is
    // done with chunk jb-1, whose slot the copies of chunk jb+3 (issued in the k-blocks after the barrier) overwrite.  The fragment
    // reads run D k-blocks ahead of the MFMAs ACROSS the chunk boundary (the barrier is behind them by then), and so does the bias:
    // the matrix pipe does not drain between chunks.  Step s = jb * KB + kb; fragment buffers are indexed s mod (D + 1).
    // Fragment reads go in batches of BS k-blocks, DB batches ahead, the first-used fragment of a batch LAST (LDS returns in order:
    // one lgkmcnt wait per batch instead of one per fragment).
    constexpr int BS = KB >= 8 ? WR_BS : 1, DB = KB >= 8 ? WR_DB : (KB >= 6 ? 3 : 1), D = BS * DB, NB = BS * (DB + 1);
    constexpr int HB = KB / 2, NSTEP = NCH * KB;
    static_assert(D + BS - 1 <= KB - HB, "reads of the next chunk start after the barrier");
    f4 accs[2];
    f4 bnext = f4{0.f, 0.f, 0.f, 0.f};
    u4 wfa[NB], wfb[NB];
    const f4* wl = Wp + wr_coff<Net>(cb);                     // run time for shared instances
    const f4* wnext[3];                                       // the three chunks after the layer (compile-time distance, run-time base)
#pragma unroll
    for (int i = 0; i < 3; ++i) wnext[i] = Wp + wr_coff<Net>(cb + NCH + i);
    asm volatile("" : "+s"(wl));
    auto frag_of = [&](int c) { return reinterpret_cast<const u4*>(reinterpret_cast<const char*>(ring) + slot_b[c & 3]) + lane; };
    auto bias_of = [&](int c) { return *(reinterpret_cast<const f4*>(reinterpret_cast<const char*>(bias_ring) + bslot_b[c & 3]) + g); };
    // the layer's first chunk is visible (prologue, or the barrier inside the previous layer's last chunk)
    accs[0] = bias_of(0) * AS;
#pragma unroll
    for (int i = 0; i < D; ++i)
      if (i < NSTEP) {
        wfa[i % NB] = frag_of(i / KB)[(2 * (i % KB)) * 64];
        wfb[i % NB] = frag_of(i / KB)[(2 * (i % KB) + 1) * 64];
      }
#pragma unroll
    for (int jb = 0; jb < NCH; ++jb) {
      f4& acc = accs[jb & 1];
      if (jb > 0) acc = bnext * AS;
      // copies of chunk jb + 3
      constexpr int dummy2 = 0;
      (void)dummy2;
      const int K3 = jb + 3 < NCH ? K : Net::K(wr_layer_of<Net>(CB + jb + 3));
      const int nu3 = wr_units(K3);
      const f4* src3 = jb + 3 < NCH ? wl + (long)(jb + 3) * chunk_f4(K) : wnext[jb + 3 - NCH < 3 ? jb + 3 - NCH : 0];
      const int sl3 = (jb + 3) & 3;
      const unsigned dst3 = ring_b + slot_b[sl3], bdst3 = bias_b + bslot_b[sl3];
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const int st = jb * KB + kb;
        if (kb == HB) {   // chunk jb+1 must have landed: this wave's copies of chunk jb+2 may still be in flight
          const int allowed = jb + 2 < NCH ? NP : wr_np(Net::K(wr_layer_of<Net>(CB + jb + 2)));
          if (allowed >= 10) wr_wait<10>();
          else if (allowed >= 9) wr_wait<9>();
          else if (allowed >= 4) wr_wait<4>();
          else wr_wait<2>();
#ifndef WR_ABL_NOBAR
          __builtin_amdgcn_s_barrier();
#endif
          asm volatile("" ::: "memory");
          bnext = bias_of(jb + 1);
        }
        const h8 wh = __builtin_bit_cast(h8, wfa[st % NB]), wlo = __builtin_bit_cast(h8, wfb[st % NB]);
        if (st % BS == 0) {
#pragma unroll
          for (int i = BS - 1; i >= 0; --i) {
            const int s2 = st + D + i;
            if (s2 < NSTEP) {
#ifdef WR_ABL_NOLDS
              wfa[s2 % NB] = wfa[st % NB];
              wfb[s2 % NB] = wfb[st % NB];
#else
              wfb[s2 % NB] = frag_of(s2 / KB)[(2 * (s2 % KB) + 1) * 64];
              wfa[s2 % NB] = frag_of(s2 / KB)[(2 * (s2 % KB)) * 64];
#endif
            }
          }
        }
        const h8 a = __builtin_bit_cast(h8, xh[kb]), b = __builtin_bit_cast(h8, xl[kb]);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, b, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, a, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo, a, acc, 0, 0, 0);
#ifdef WR_ABL_NOVAL
        if (jb > 0 && !LAST && kb == 0) {      // timing ablation: keep the data dependence, drop the arithmetic
          const f4& pa = accs[(jb - 1) & 1];
          yh[(jb - 1) >> 1][((jb - 1) & 1) * 2] = __builtin_bit_cast(unsigned, pa[0]) & 0x3c003c00u;
          yl[(jb - 1) >> 1][((jb - 1) & 1) * 2] = __builtin_bit_cast(unsigned, pa[1]) & 0x3c003c00u;
          yh[(jb - 1) >> 1][((jb - 1) & 1) * 2 + 1] = __builtin_bit_cast(unsigned, pa[2]) & 0x3c003c00u;
          yl[(jb - 1) >> 1][((jb - 1) & 1) * 2 + 1] = __builtin_bit_cast(unsigned, pa[3]) & 0x3c003c00u;
        }
        if (false) {
#else
        if (jb > 0 && !LAST) {                 // activation + split of chunk jb-1, a value pair at a time
#endif
          constexpr int dummy3 = 0;
          (void)dummy3;
          if (kb == 0 || kb == (KB >= 8 ? 3 : 1)) {
            const int q = kb == 0 ? 0 : 1, pj = jb - 1;
            const f4& pa = accs[pj & 1];
            float v0, v1;
            if constexpr (SKIPOUT) {
              v0 = act_fn<Net::ACT>(pa[2 * q] * zs) * inv_sqrt2 * AS;
              v1 = act_fn<Net::ACT>(pa[2 * q + 1] * zs) * inv_sqrt2 * AS;
            } else {
              v0 = hidden_val(pa[2 * q]);
              v1 = hidden_val(pa[2 * q + 1]);
            }
            put_pair(v0, v1, yh[pj >> 1], yl[pj >> 1], (pj & 1) * 2 + q);
          }
        }
        if (LAST && jb > 0 && kb == 0) {       // output chunk jb-1 (nets with more than 16 outputs)
          const f4& pa = accs[(jb - 1) & 1];
          if (rrow < M) *(reinterpret_cast<f4*>(Y + rrow * (long)(NCH * 16) + (jb - 1) * 16) + g) = f4{pa[0] * zs, pa[1] * zs, pa[2] * zs, pa[3] * zs};
        }
#ifndef WR_ABL_NODMA
        if (kb >= HB) {
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (u < nu3 && (u * (KB - HB)) / nu3 == kb - HB) {
              if (K3 == 64) wr_copy_unit<64>(u, src3, lane4, lane16, bdst3, dst3, wave);
              else if (K3 == 192) wr_copy_unit<192>(u, src3, lane4, lane16, bdst3, dst3, wave);
              else if (K3 == 512) wr_copy_unit<512>(u, src3, lane4, lane16, bdst3, dst3, wave);
              else wr_copy_unit<544>(u, src3, lane4, lane16, bdst3, dst3, wave);
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
    const f4& last = accs[(NCH - 1) & 1];
    if constexpr (LAST) {
      if constexpr (NCH == 1) {           // up to four outputs (CESR nets): lanes of group 0 hold them
        if (g == 0 && rrow < M) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (r < n_out) Y[rrow * n_out + r] = last[r] * zs;
        }
      } else {
        if (rrow < M) *(reinterpret_cast<f4*>(Y + rrow * (long)(NCH * 16) + (NCH - 1) * 16) + g) = f4{last[0] * zs, last[1] * zs, last[2] * zs, last[3] * zs};
      }
    } else {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        constexpr int pj = NCH - 1;
        float v0, v1;
        if constexpr (SKIPOUT) {
          v0 = act_fn<Net::ACT>(last[2 * q] * zs) * inv_sqrt2 * AS;
          v1 = act_fn<Net::ACT>(last[2 * q + 1] * zs) * inv_sqrt2 * AS;
        } else {
          v0 = hidden_val(last[2 * q]);
          v1 = hidden_val(last[2 * q + 1]);
        }
        put_pair(v0, v1, yh[pj >> 1], yl[pj >> 1], (pj & 1) * 2 + q);
      }
      // the next layer's operands are complete
      if constexpr (SKIPOUT) {
#pragma unroll
        for (int kb = 0; kb < 17; ++kb) {
          if (kb < SK0) {
            xh[kb] = yh[kb < SK0 ? kb : 0];
            xl[kb] = yl[kb < SK0 ? kb : 0];
          } else if (kb == SK0) {
            xh[kb] = u4{yh[SK0 < 16 ? SK0 : 0][0], yh[SK0 < 16 ? SK0 : 0][1], skh[0][2], skh[0][3]};
            xl[kb] = u4{yl[SK0 < 16 ? SK0 : 0][0], yl[SK0 < 16 ? SK0 : 0][1], skl[0][2], skl[0][3]};
          } else {
            xh[kb] = skh[kb > SK0 ? kb - SK0 : 0];
            xl[kb] = skl[kb > SK0 ? kb - SK0 : 0];
          }
        }
      } else {
        constexpr int KBN = Net::K(LI + 1 < L ? LI + 1 : LI) / 32;
#pragma unroll
        for (int kb = 0; kb < KBN; ++kb) {
          xh[kb] = yh[kb];
          xl[kb] = yl[kb];
        }
      }
    }
  };

  // ---- prologue: chunks 0, 1, 2 of the stream
  long round = blockIdx.x;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    constexpr int dummy = 0;
    (void)dummy;
    const int Kc = Net::K(wr_layer_of<Net>(c));
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (u < wr_units(Kc)) {
        if (Kc == 64) wr_copy_unit<64>(u, Wp + wr_coff<Net>(c), lane4, lane16, bias_b + bslot_b[c], ring_b + slot_b[c], wave);
        else wr_copy_unit<192>(u, Wp + wr_coff<Net>(c), lane4, lane16, bias_b + bslot_b[c], ring_b + slot_b[c], wave);
      }
  }
  wr_wait<0>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  static_assert(K0 == 64 || K0 == 192, "first-layer widths");
  static_assert(Net::NCH(0) >= 3, "the prologue assumes three chunks in layer 0");

  for (; round < nrounds; round += gridDim.x) {
    rrow = round * 64 + wave * 16 + (lane & 15);
    load_layer0();
    if constexpr (HAS_SKIP) {
      // CESR nets: layer 0 | 1, 2 (one instance) | 3 (skip-layer outputs) | 4 (K = 544) | 5, 6 (the same instance as 1, 2) | 7 | 8
#pragma unroll 1
      for (int l = 0; l < 9; ++l) {
        const int cb = l < 4 ? 32 * l : 32 * (l - 1) + Net::NCH(3);
        if (l == 0) run_layer(std::integral_constant<int, 0>{}, 0);
        else if (l == 3) run_layer(std::integral_constant<int, 3>{}, cb);
        else if (l == 4) run_layer(std::integral_constant<int, 4>{}, cb);
        else if (l == 7) run_layer(std::integral_constant<int, 7>{}, cb);
        else if (l == 8) run_layer(std::integral_constant<int, 8>{}, cb);
        else run_layer(std::integral_constant<int, 1>{}, cb);
      }
    } else {
      // layer 0 | 1, 2 (one instance) | 3 | 4
#pragma unroll 1
      for (int l = 0; l < 5; ++l) {
        const int cb = 32 * l;
        if (l == 0) run_layer(std::integral_constant<int, 0>{}, 0);
        else if (l == 3) run_layer(std::integral_constant<int, 3>{}, cb);
        else if (l == 4) run_layer(std::integral_constant<int, 4>{}, cb);
        else run_layer(std::integral_constant<int, 1>{}, cb);
      }
    }
  }
  range_report(sat, range_word);
  wr_wait<0>();
  __syncthreads();
}


// one kernel per translation unit (wide_ring_*.hip: each takes minutes to compile): launchers
int launch_cesr_ring_normal(const float* x, long M, const f4* W, float us, float* Y, unsigned* rw, int grid, hipStream_t s);
int launch_cesr_ring_shadow(const float* x, long M, int n_label, const f4* W, float us, float* Y, unsigned* rw, int grid, hipStream_t s);
int launch_wide_ring_encoder(const float* x, const float* extra, long M, const f4* W, float us, float* Y, unsigned* rw, int grid, hipStream_t s);
int launch_wide_ring_decoder(const float* x, const float* extra, long M, const f4* W, float us, float* Y, unsigned* rw, int grid, hipStream_t s);
int launch_wide_ring_encoder_rows(const float* X, long M, const f4* W, float us, float* Y, unsigned* rw, int grid, hipStream_t s);
int launch_wide_ring_decoder_rows(const float* X, long M, const f4* W, float us, float* Y, unsigned* rw, int grid, hipStream_t s);

}  // namespace rb
