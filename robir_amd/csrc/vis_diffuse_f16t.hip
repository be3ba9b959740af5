// Light-SG visibility in PLAIN f16 -- one f16 MFMA product per multiply-add, fp32 accumulation: the labelled THROUGHPUT mode
// `BASELINE.json configs[4]` names ("fp16 MLP weights on MFMA"), round 4.  NARROWER than the reference's fp32 and never the default:
// weights are the round-to-nearest f16 of the fp32 weights (the h pieces of the exact-operand blob, rb_pack_layer_x6), activations are
// truncated to f16 between the layers (v_cvt_pkrtz), sums are fp32.  robir_amd/precision.py selects it with ROBIR_PRECISION=f16 only;
// DESIGN.md has its measured error against the oracle.  (get_diffuse_visibility, model/sg_render.py:111-195; VisNetwork,
// model/implicit_differentiable_renderer.py:241-258.)
//
// Machine: the persistent tile-list form of k_dvis_x6t (vis_diffuse_x6t.hip: k_dvis3_cull / k_dvis3_reduce around it) with FOUR
// 16-sample tiles per wave -- one-piece operands of four tiles and two layers are 256 registers -- rounds of sixteen tiles, the h
// fragments of a chunk (8 KB) through a four-slot LDS ring, a chunk = four runs of eight MFMAs (one accumulator each), the fragments
// refilled behind the last run, relu + truncation of the previous chunk (four instructions per value pair) between the runs.
#include "../../include/robir_hip.h"
#include "common.h"
#include "mlp_engine.h"
#include "x6_ring.h"
#include "x6t_engine.h"

namespace rb {

constexpr int FT_WF4 = 512;              // h fragments of a chunk: [kb 8][lane 64] x 16 B = 8 KB
constexpr int FT_CF4 = 4 + 1536;         // a packed chunk of the exact-operand blob in global memory (bias, then [kb][piece][lane])
constexpr int FT_TILES = 4;

struct FtTile {
  int point;        // -1: no such tile
  int dir_base;
};
struct FtArgs {
  const float *A, *Bd;
  const f4* W49;
  int argmax_vis;
  const unsigned short* pair_j;
  const FtTile* tile_info;
  const unsigned long long* counters;
  float* pair_vis;
};

__global__ __launch_bounds__(256, 1) void k_dvis_f16t(const FtArgs a) {
  __shared__ f4 ring[4 * FT_WF4];           // 32 KB
  __shared__ f4 headw[FT_WF4];              // 8 KB: chunk 48 (256 -> 2 head)
  __shared__ f4 bias_tab[49 * 4];
  __shared__ f4 a_rows[2 * 16 * 64];        // 32 KB: [round parity][tile of the round][256 floats]
  const float* __restrict__ A = a.A;
  const float* __restrict__ Bd = a.Bd;
  const f4* __restrict__ W49 = a.W49;
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = gridDim.x;
  const long total_tiles = (long)a.counters[0];
  const long total_rounds = (total_tiles + 15) >> 4;
  for (int i = tid; i < 49 * 4; i += 256) bias_tab[i] = W49[(long)(i >> 2) * FT_CF4 + (i & 3)];
  for (int i = tid; i < FT_WF4; i += 256) headw[i] = W49[48L * FT_CF4 + 4 + ((i >> 6) * 3) * 64 + (i & 63)];
  __syncthreads();
  if ((long)blockIdx.x >= total_rounds) return;           // workgroup-uniform

  const unsigned ring_b = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)ring);
  const unsigned arow_b = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)a_rows);
  unsigned ring_lane = ring_b + (unsigned)lane * 16u;
  asm volatile("" : "+v"(ring_lane));
  typedef const __attribute__((address_space(3))) u4* lds_u4p;
  // a wave copies the h fragments of k-blocks 2 w, 2 w + 1 of a chunk: 1 KB each, 3 KB apart in the blob, adjacent in the ring
  auto copy_chunk_piece = [&](int i, const f4* chunk_frag0, int slot) {
    xt_dma16_imm<0>(chunk_frag0 + ((2 * wave + i) * 3) * 64, xt_lane16<0>(), ring_b + (unsigned)slot * 8192u + (unsigned)(2 * wave + i) * 1024u);
  };
  u4 wh[8];
  f4 bias;
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int i = 0; i < 2; ++i) copy_chunk_piece(i, W49 + (long)c * FT_CF4 + 4, c);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // chunk 0 landed; chunks 1, 2 stay in flight
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int k = 0; k < 8; ++k) wh[k] = ((lds_u4p)ring_lane)[k * 64];
  bias = bias_tab[g];

  u4 P[FT_TILES][8], Q[FT_TILES][8];   // one-piece B operands of the current / next layer, four tiles
#define FT_MFMA(ACC, WREG, XREG) \
  ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, WREG), __builtin_bit_cast(h8, XREG), ACC, 0, 0, 0)
  f4 prev[FT_TILES];
  auto ep_tile = [&](int t, int pj) {      // relu + truncation of tile t's sixteen outputs of chunk pj -> the next layer's operands
#pragma unroll
    for (int q = 0; q < 2; ++q)
      Q[t][pj >> 1][(pj & 1) * 2 + q] =
          __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(fmaxf(prev[t][2 * q], 0.f), fmaxf(prev[t][2 * q + 1], 0.f)));
  };

  f4 raw[FT_TILES][16];
  int jj[FT_TILES], jjn[FT_TILES];
  int tpn[FT_TILES], tbn[FT_TILES], jn2[FT_TILES];
  auto tile_lookup = [&](long round) {
#pragma unroll
    for (int t = 0; t < FT_TILES; ++t) {
      const long T = round * 16 + wave * FT_TILES + t;
      FtTile rec{-1, 0};
      int j = 0xFFFF;
      if (round < total_rounds && T < total_tiles) {
        rec = a.tile_info[T];
        j = (int)a.pair_j[T * 16 + (lane & 15)];
      }
      tpn[t] = __builtin_amdgcn_readfirstlane(rec.point);
      tbn[t] = __builtin_amdgcn_readfirstlane(rec.dir_base);
      jn2[t] = j;
    }
  };
  auto fetch_rows = [&](int parity_next) {
#pragma unroll
    for (int t = 0; t < FT_TILES; ++t) {
      const long prow = tpn[t] < 0 ? 0L : (long)tpn[t];
      xt_dma16_imm<0>(reinterpret_cast<const f4*>(A + prow * 256), xt_lane16<0>(), arow_b + (unsigned)(parity_next * 16 + wave * FT_TILES + t) * 1024u);
    }
#pragma unroll
    for (int t = 0; t < FT_TILES; ++t) {
      jjn[t] = (tpn[t] < 0 || jn2[t] == 0xFFFF) ? -1 : jn2[t];
      const long row = (long)tbn[t] + (jjn[t] < 0 ? 0 : jjn[t]);
      const f4* brow = reinterpret_cast<const f4*>(Bd + row * 256) + g;
#pragma unroll
      for (int kb = 0; kb < 16; ++kb) raw[t][kb] = brow[kb * 4];
    }
  };
  long rd = blockIdx.x;
  int parity = 0;
  tile_lookup(rd);
  fetch_rows(0);
  for (; rd < total_rounds; rd += G) {
    tile_lookup(rd + G);
    // ---- layer 0: relu(A[point] + Bd[dir]) truncated to f16, straight into the operand registers
#pragma unroll
    for (int t = 0; t < FT_TILES; ++t) {
      jj[t] = jjn[t];
      const f4* arow = a_rows + (parity * 16 + wave * FT_TILES + t) * 64;
#pragma unroll
      for (int kb = 0; kb < 16; ++kb) {
        const f4 bv = raw[t][kb];
        const f4 av = arow[kb * 4 + g];
#pragma unroll
        for (int q = 0; q < 2; ++q)
          P[t][kb / 2][(kb & 1) * 2 + q] = __builtin_bit_cast(
              unsigned, __builtin_amdgcn_cvt_pkrtz(fmaxf(av[2 * q] + bv[2 * q], 0.f), fmaxf(av[2 * q + 1] + bv[2 * q + 1], 0.f)));
      }
    }
#pragma unroll 1
    for (int l = 0; l < 3; ++l) {
      const f4* Wl = W49 + (long)l * 16 * FT_CF4 + 4;
      const f4* Wn = W49 + (long)(l == 2 ? 0 : l + 1) * 16 * FT_CF4 + 4;
#pragma unroll
      for (int jb = 0; jb < 16; ++jb) {
        f4 acc[FT_TILES];
        // chunk jb+1 has landed once at most this wave's two copies of chunk jb+2 are in flight; past the barrier every wave has finished
        // with chunk jb-1, whose slot the copies of chunk jb+3 reuse
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int nx3 = jb + 3;
        const f4* dsrc = nx3 < 16 ? Wl + (long)nx3 * FT_CF4 : Wn + (long)(nx3 - 16) * FT_CF4;
        const lds_u4p nfrag = (lds_u4p)(ring_lane + (unsigned)((jb + 1) & 3) * 8192u);
        const f4 nbias = bias_tab[(l * 16 + jb + 1) * 4 + g];
        const bool down = (jb & 1) != 0;        // alternate the walking direction: a chunk starts with the fragment requested last
#pragma unroll
        for (int t = 0; t < FT_TILES; ++t) {
          acc[t] = bias;
#pragma unroll
          for (int k_ = 0; k_ < 8; ++k_) {
            const int k = down ? 7 - k_ : k_;
            FT_MFMA(acc[t], wh[k], P[t][k]);
            if (t == FT_TILES - 1) wh[k] = nfrag[k * 64];
          }
          if (t < 2) copy_chunk_piece(t, dsrc, nx3 & 3);
          if (jb > 0) ep_tile(t, jb - 1);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int t = 0; t < FT_TILES; ++t) prev[t] = acc[t];
        bias = nbias;
      }
#pragma unroll
      for (int t = 0; t < FT_TILES; ++t) ep_tile(t, 15);
#pragma unroll
      for (int t = 0; t < FT_TILES; ++t)
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) P[t][kb] = Q[t][kb];
    }
    // ---- head: chunk 48 from its resident LDS copy; next round's rows are requested first
    fetch_rows(parity ^ 1);
    {
      const lds_u4p hw = (lds_u4p)((unsigned)(unsigned long)((__attribute__((address_space(3))) char*)headw) + (unsigned)lane * 16u);
      f4 acc[FT_TILES];
#pragma unroll
      for (int t = 0; t < FT_TILES; ++t) acc[t] = bias;
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) {
        const u4 fh = hw[kb * 64];
#pragma unroll
        for (int t = 0; t < FT_TILES; ++t) FT_MFMA(acc[t], fh, P[t][kb]);
      }
      bias = bias_tab[g];
#pragma unroll
      for (int t = 0; t < FT_TILES; ++t) {
        const float l0 = acc[t][0], l1 = acc[t][1];
        if (g == 0 && jj[t] >= 0) {
          float v;
          if (a.argmax_vis) {
            v = l1 > l0 ? 1.f : 0.f;
          } else {
            const float mx = fmaxf(l0, l1);
            const float e0 = expf(l0 - mx), e1 = expf(l1 - mx);
            v = e1 / (e0 + e1);
          }
          a.pair_vis[(rd * 16 + wave * FT_TILES + t) * 16 + (lane & 15)] = v;
        }
      }
    }
    parity ^= 1;
  }
#undef FT_MFMA
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Second generation (round 5).  The first kernel above runs at 0.41 of the f16 pipe WITHOUT reaching the package cap (1212 W of 1400):
// with 32 MFMAs per chunk and wave, what surrounds them decides -- a counted wait + workgroup barrier per chunk, two LDS-DMA copies of
// six issue slots each (the h fragments of a chunk are 3 KB apart in the exact-operand blob: no shared M0), sixteen accumulator moves,
// five vector instructions per value pair of the epilogue, and 128 moves per tile set and layer for the operand hand-over.  Here:
//   * STEPS of two chunks: one counted wait + barrier per 64 MFMAs; an eight-slot ring (64 KB), the copies of step s+2 issued at the top
//     of step s, in front of its wait (their slots belong to step s-2, which every wave has left: it passed barrier s-1);
//   * the weights as a blob of their own (packing.pack_vis_f16_head: 49 x 16 biases, then the h fragments of the 49 chunks, 8 KB each,
//     contiguous): a wave's 2 KB of a chunk are ONE M0 setting and two copies;
//   * the three hidden layers straight-line with the operand sets P / Q changing roles (no hand-over moves), two accumulator sets
//     alternating by chunk (no accumulator moves: a chain starts from the bias registers);
//   * relu AFTER the truncation, on pairs (v_cvt_pkrtz + v_pk_max_f16: max(rtz(x), 0) = rtz(max(x, 0))).
// The same products summed in the same order: bit-identical to the first kernel (tests/test_sg_gpu.py).
constexpr int FT2_BIAS_F4 = 49 * 4;            // the blob's bias head, in float4
#ifndef FT2_NS
#define FT2_NS 8
#endif
#ifndef FT2_ABL
#define FT2_ABL 0                              // timing ablations (wrong results), tools/build_variant.sh: bit 0 no copies / barriers, bit 1 no fragment reads, bit 2 no epilogue of the hidden chunks, 8 no layer-0 conversion
#endif
constexpr int FT2_SLOTS = FT2_NS;              // ring slots (8 KB each): 48 % FT2_SLOTS == 0
constexpr int FT2_DIST = FT2_SLOTS / 2 - 2;    // the copies of step s + FT2_DIST are issued at the top of step s
static_assert(48 % FT2_SLOTS == 0 && FT2_SLOTS % 2 == 0 && FT2_DIST >= 1, "ring");
#ifdef FT2_TIMING
#define FT2_T(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tacc[i] += t_ - tlast; tlast = t_; }
#else
#define FT2_T(i)
#endif
typedef _Float16 ft_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned ft_relu_pack(float a, float b) {
  const ft_h2 h = __builtin_bit_cast(ft_h2, __builtin_amdgcn_cvt_pkrtz(a, b));
  const ft_h2 z = ft_h2{(_Float16)0.0f, (_Float16)0.0f};
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(h, z));
}

__global__ __launch_bounds__(256, 1) void k_dvis_f16t2(const FtArgs a) {
  __shared__ f4 ring[FT2_SLOTS * FT_WF4];   // slot = chunk % FT2_SLOTS
  __shared__ f4 headw[FT_WF4];              // 8 KB: chunk 48 (256 -> 2 head)
  __shared__ f4 bias_tab[49 * 4];
  __shared__ f4 a_rows[2 * 16 * 64];        // 32 KB: [round parity][tile of the round][256 floats]
  const float* __restrict__ A = a.A;
  const float* __restrict__ Bd = a.Bd;
  const f4* __restrict__ Wb = a.W49;        // [49][4] biases | [49][8][64] h fragments
  const f4* __restrict__ Wf = a.W49 + FT2_BIAS_F4;
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = gridDim.x;
  const long total_tiles = (long)a.counters[0];
  const long total_rounds = (total_tiles + 15) >> 4;
  for (int i = tid; i < 49 * 4; i += 256) bias_tab[i] = Wb[i];
  for (int i = tid; i < FT_WF4; i += 256) headw[i] = Wf[48L * FT_WF4 + i];
  __syncthreads();
  if ((long)blockIdx.x >= total_rounds) return;           // workgroup-uniform

  const unsigned ring_b = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)ring);
  const unsigned arow_b = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)a_rows);
  unsigned ring_lane = ring_b + (unsigned)lane * 16u;
  asm volatile("" : "+v"(ring_lane));
  typedef const __attribute__((address_space(3))) u4* lds_u4p;
  // this wave's 2 KB (k-blocks 2 w, 2 w + 1) of chunk c -> slot c & 7: one M0 setting, two copies
  auto copy_chunk = [&](int c, unsigned lv) {
    const f4* src = Wf + (long)c * FT_WF4 + (2 * wave) * 64;
    xt_dma16_imm<0>(src, lv, ring_b + (unsigned)(c % FT2_SLOTS) * 8192u + (unsigned)(2 * wave) * 1024u);
    xt_dma16_keep<1024>(src, lv);
  };
  u4 wh[2][8];                    // the fragments of the current chunk and of the next one, by chunk parity
  f4 bias;
  {
    const unsigned lv = xt_lane16<0>();
#pragma unroll
    for (int c = 0; c < 2 * FT2_DIST; ++c) copy_chunk(c, lv);          // steps 0 .. FT2_DIST - 1
  }
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (FT2_DIST - 1)) : "memory");           // step 0 landed
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int k = 0; k < 8; ++k) wh[0][k] = ((lds_u4p)ring_lane)[k * 64];
#if FT2_ABL & 2
#pragma unroll
  for (int k = 0; k < 8; ++k) wh[1][k] = ((lds_u4p)ring_lane)[k * 64 + 512];
#endif
  bias = bias_tab[g];

  u4 P[FT_TILES][8], Q[FT_TILES][8];
#define FT_MFMA(ACC, WREG, XREG) \
  ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, WREG), __builtin_bit_cast(h8, XREG), ACC, 0, 0, 0)

  f4 raw[FT_TILES][16];
  int jj[FT_TILES], jjn[FT_TILES];
  int tpn[FT_TILES], tbn[FT_TILES], jn2[FT_TILES];
  auto tile_lookup = [&](long round) {
#pragma unroll
    for (int t = 0; t < FT_TILES; ++t) {
      const long T = round * 16 + wave * FT_TILES + t;
      FtTile rec{-1, 0};
      int j = 0xFFFF;
      if (round < total_rounds && T < total_tiles) {
        rec = a.tile_info[T];
        j = (int)a.pair_j[T * 16 + (lane & 15)];
      }
      tpn[t] = __builtin_amdgcn_readfirstlane(rec.point);
      tbn[t] = __builtin_amdgcn_readfirstlane(rec.dir_base);
      jn2[t] = j;
    }
  };
  auto fetch_rows = [&](int parity_next) {
#pragma unroll
    for (int t = 0; t < FT_TILES; ++t) {
      const long prow = tpn[t] < 0 ? 0L : (long)tpn[t];
      xt_dma16_imm<0>(reinterpret_cast<const f4*>(A + prow * 256), xt_lane16<0>(), arow_b + (unsigned)(parity_next * 16 + wave * FT_TILES + t) * 1024u);
    }
#pragma unroll
    for (int t = 0; t < FT_TILES; ++t) {
      jjn[t] = (tpn[t] < 0 || jn2[t] == 0xFFFF) ? -1 : jn2[t];
      const long row = (long)tbn[t] + (jjn[t] < 0 ? 0 : jjn[t]);
      const f4* brow = reinterpret_cast<const f4*>(Bd + row * 256) + g;
#pragma unroll
      for (int kb = 0; kb < 16; ++kb) raw[t][kb] = brow[kb * 4];
    }
  };
  // one hidden layer: operands X -> outputs Y (the next layer's operands); cb = the layer's first chunk (0, 16, 32)
  auto run_layer = [&](const u4 (&X)[FT_TILES][8], u4 (&Y)[FT_TILES][8], int cb) {
    f4 acc[2][FT_TILES];
    auto ep_tile = [&](int t, int pj, const f4 (&pv)[FT_TILES]) {
      Y[t][pj >> 1][(pj & 1) * 2] = ft_relu_pack(pv[t][0], pv[t][1]);
      Y[t][pj >> 1][(pj & 1) * 2 + 1] = ft_relu_pack(pv[t][2], pv[t][3]);
    };
#pragma unroll
    for (int sb = 0; sb < 8; ++sb) {
      {   // the copies of step sb + 2 (cyclic over the 48 chunks), then: step sb + 1 has landed once only these four are in flight
#if !(FT2_ABL & 1)
        const int n4 = (cb + 2 * sb + 2 * FT2_DIST) % 48;
        const unsigned lv = xt_lane16<0>();
        copy_chunk(n4, lv);
        copy_chunk(n4 + 1, lv);
#endif
      }
#if !(FT2_ABL & 1)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (FT2_DIST - 1)) : "memory");
      __builtin_amdgcn_s_barrier();
#endif
      asm volatile("" ::: "memory");
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int jb = 2 * sb + h;
        const lds_u4p nfrag = (lds_u4p)(ring_lane + (unsigned)((cb + jb + 1) % FT2_SLOTS) * 8192u);
        const f4 nbias = bias_tab[(cb + jb + 1) * 4 + g];
        // the next chunk's fragments go to the OTHER register set, two per tile: requested a whole chunk before their first use (step
        // s + 1 has landed by barrier s: the wait above leaves only step s + 2 in flight), no wait on the LDS in front of a chunk
        const u4 (&wc)[8] = wh[jb & 1];
        u4 (&wn)[8] = wh[(jb & 1) ^ 1];
        f4 (&ac)[FT_TILES] = acc[jb & 1];
        const f4 (&pv)[FT_TILES] = acc[(jb & 1) ^ 1];
        // k-block outermost, the four tiles inside: consecutive MFMAs never share an accumulator (a vector instruction between two
        // MFMAs of ONE accumulator chain costs ~43 cycles, MI355X_MICROARCH.md per-instruction constants; between different chains ~6),
        // and each k-step carries its share of the other work: one fragment read of the next chunk, one value pair of the previous
        // chunk's epilogue per accumulator half
#pragma unroll
        for (int k_ = 0; k_ < 8; ++k_) {
          const int k = (jb & 1) ? 7 - k_ : k_;       // the first kernel's summation order (it walks odd chunks downwards): the same bits
#pragma unroll
          for (int t = 0; t < FT_TILES; ++t) {
            if (k_ == 0) ac[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, wc[k]), __builtin_bit_cast(h8, X[t][k]), bias, 0, 0, 0);
            else FT_MFMA(ac[t], wc[k], X[t][k]);
          }
#if !(FT2_ABL & 2)
          wn[k_] = nfrag[k_ * 64];
#endif
#if FT2_ABL & 4
          if (jb > 0) asm volatile("" ::"v"(pv[k_ >> 1]));     // the accumulators stay "used": the MFMAs survive
#endif
          if (jb > 0 && !(FT2_ABL & 4)) {
            const int t = k_ >> 1, pj = jb - 1;
            Y[t][pj >> 1][(pj & 1) * 2 + (k_ & 1)] = ft_relu_pack(pv[t][(k_ & 1) * 2], pv[t][(k_ & 1) * 2 + 1]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        bias = nbias;
      }
    }
#pragma unroll
    for (int t = 0; t < FT_TILES; ++t) ep_tile(t, 15, acc[1]);
  };

  long rd = blockIdx.x;
  int parity = 0;
  tile_lookup(rd);
  fetch_rows(0);
#ifdef FT2_TIMING
  unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
  int trounds = 0;
#endif
  for (; rd < total_rounds; rd += G) {
    tile_lookup(rd + G);
#ifdef FT2_TIMING
    ++trounds;
    FT2_T(5)
#endif
    // ---- layer 0: relu(A[point] + Bd[dir]) truncated to f16, straight into the operand registers
#pragma unroll
    for (int t = 0; t < FT_TILES; ++t) {
      jj[t] = jjn[t];
      const f4* arow = a_rows + (parity * 16 + wave * FT_TILES + t) * 64;
#pragma unroll
      for (int kb = 0; kb < (FT2_ABL == 8 ? 1 : 16); ++kb) {
        const f4 bv = raw[t][kb];
        const f4 av = arow[kb * 4 + g];
        P[t][kb / 2][(kb & 1) * 2] = ft_relu_pack(av[0] + bv[0], av[1] + bv[1]);
        P[t][kb / 2][(kb & 1) * 2 + 1] = ft_relu_pack(av[2] + bv[2], av[3] + bv[3]);
      }
#if FT2_ABL == 8
#pragma unroll
      for (int kb = 1; kb < 16; ++kb) {
        P[t][kb / 2][(kb & 1) * 2] = __builtin_bit_cast(unsigned, raw[t][kb][0]);
        P[t][kb / 2][(kb & 1) * 2 + 1] = __builtin_bit_cast(unsigned, raw[t][kb][2]);
      }
#endif
    }
    FT2_T(0)
    run_layer(P, Q, 0);
    FT2_T(1)
    run_layer(Q, P, 16);
    FT2_T(2)
    run_layer(P, Q, 32);
    FT2_T(3)
    // ---- head: chunk 48 from its resident LDS copy, operands in Q; next round's rows are requested first
    fetch_rows(parity ^ 1);
    {
      const lds_u4p hw = (lds_u4p)((unsigned)(unsigned long)((__attribute__((address_space(3))) char*)headw) + (unsigned)lane * 16u);
      f4 acc[FT_TILES];
#pragma unroll
      for (int t = 0; t < FT_TILES; ++t) acc[t] = bias;
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) {
        const u4 fh = hw[kb * 64];
#pragma unroll
        for (int t = 0; t < FT_TILES; ++t) FT_MFMA(acc[t], fh, Q[t][kb]);
      }
      bias = bias_tab[g];
#pragma unroll
      for (int t = 0; t < FT_TILES; ++t) {
        const float l0 = acc[t][0], l1 = acc[t][1];
        if (g == 0 && jj[t] >= 0) {
          float v;
          if (a.argmax_vis) {
            v = l1 > l0 ? 1.f : 0.f;
          } else {
            const float mx = fmaxf(l0, l1);
            const float e0 = expf(l0 - mx), e1 = expf(l1 - mx);
            v = e1 / (e0 + e1);
          }
          a.pair_vis[(rd * 16 + wave * FT_TILES + t) * 16 + (lane & 15)] = v;
        }
      }
    }
    parity ^= 1;
    FT2_T(4)
  }
#ifdef FT2_TIMING
  if ((blockIdx.x == 0 || blockIdx.x == 100) && tid == 0)
    printf("f16t2 wg %d rounds %d cycles/round: conv %llu L1 %llu L2 %llu L3 %llu head+fetch %llu lookup %llu\n", (int)blockIdx.x, trounds, tacc[0] / trounds,
           tacc[1] / trounds, tacc[2] / trounds, tacc[3] / trounds, tacc[4] / trounds, tacc[5] / trounds);
#endif
#undef FT_MFMA
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

struct V3Tile;
__global__ void k_dvis3_cull(const float* __restrict__ normals, const int* __restrict__ cid, long n, const float* __restrict__ dirs, int LS,
                             unsigned short* __restrict__ pair_j, V3Tile* __restrict__ tile_info, int2* __restrict__ point_info,
                             unsigned long long* __restrict__ counters, unsigned long long* __restrict__ eval_count);
__global__ void k_dvis3_reduce(const int* __restrict__ cid, long n, const float* __restrict__ wdir, const float* __restrict__ wsum,
                               const unsigned short* __restrict__ pair_j, const float* __restrict__ pair_vis,
                               const int2* __restrict__ point_info, int L, int nsamp, float* __restrict__ vis_out);

}  // namespace rb

using namespace rb;

extern "C" int rb_dvis_stream_f16(const float* normals, const int* chunk_id, long n, const float* A, const float* Bd,
                                  const float* dirs, const float* wdir, const float* wsum, const float* W49, int L, int nsamp,
                                  int argmax_vis, int scale_log2, unsigned short* pair_j, float* pair_vis, int* tile_info,
                                  int* point_info, unsigned long long* counters, int n_workgroups, float* vis_out,
                                  unsigned long long* eval_count, rb_stream_t stream) {
  // scale_log2 carries the blob format: 0 = the exact-operand blob of rb_dvis_stream_x6 (first kernel: reads its h pieces in place),
  // 1 = the f16 blob (49 x 16 biases, then the h fragments chunk by chunk: packing.pack_vis_f16_head) -> the second-generation kernel
  if (n <= 0) return 0;
  RB_REQUIRE(normals && A && Bd && dirs && wdir && wsum && W49 && vis_out, "null pointer");
  RB_REQUIRE(pair_j && pair_vis && tile_info && point_info && counters, "null scratch pointer");
  RB_REQUIRE(n <= RB_MAX_BLOCKS, "too many points for one launch (one workgroup each in the cull / reduce passes)");
  RB_REQUIRE(L > 0 && L <= 256 && nsamp > 0 && (long)L * nsamp <= 4096 && (L * nsamp) % 16 == 0,
             "need L <= 256, L*nsamp <= 4096 and a multiple of 16");
  RB_REQUIRE((long)n * (L * nsamp / 16) < (1L << 31), "tile index would overflow 31 bits");
  RB_REQUIRE(scale_log2 == 0 || scale_log2 == 1, "scale_log2: 0 = exact-operand blob (first-generation kernel), 1 = f16 blob (second generation)");
  hipStream_t s = (hipStream_t)stream;
  if (n_workgroups <= 0) n_workgroups = device_cus();
  RB_REQUIRE(n_workgroups > 0, "device query failed");
  if (hipMemsetAsync(counters, 0, 2 * sizeof(unsigned long long), s) != hipSuccess) return rb::fail(__func__, "memset failed");
  hipLaunchKernelGGL(k_dvis3_cull, dim3((unsigned)n), dim3(256), 0, s, normals, chunk_id, n, dirs, L * nsamp, pair_j,
                     reinterpret_cast<V3Tile*>(tile_info), reinterpret_cast<int2*>(point_info), counters, eval_count);
  if (int rc = check_launch("k_dvis3_cull")) return rc;
  FtArgs a{};
  a.A = A, a.Bd = Bd, a.W49 = (const f4*)W49, a.argmax_vis = argmax_vis;
  a.pair_j = pair_j, a.tile_info = reinterpret_cast<const FtTile*>(tile_info), a.counters = counters, a.pair_vis = pair_vis;
  if (scale_log2 == 1) hipLaunchKernelGGL(k_dvis_f16t2, dim3((unsigned)n_workgroups), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(k_dvis_f16t, dim3((unsigned)n_workgroups), dim3(256), 0, s, a);
  if (int rc = check_launch("k_dvis_f16t")) return rc;
  hipLaunchKernelGGL(k_dvis3_reduce, dim3((unsigned)n), dim3(256), 0, s, chunk_id, n, wdir, wsum, pair_j, pair_vis,
                     reinterpret_cast<const int2*>(point_info), L, nsamp, vis_out);
  return check_launch("k_dvis3_reduce");
}
