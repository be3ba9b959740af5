// Light-SG visibility with EXACT fp32 operands on the f16 matrix pipe, TWO 16-sample tiles per wave (round 4;
// get_diffuse_visibility, model/sg_render.py:111-195; VisNetwork, model/implicit_differentiable_renderer.py:241-258).
//
// k_dvis_x6 (vis_diffuse_x6.hip) gives a wave one tile: every weight fragment it reads from the LDS feeds one MFMA per product, and a
// workgroup re-copies the net's 1.15 MB of three-piece weights for every 64 samples -- per chunk and CU 96 KB of fragment reads and
// 24 KB of LDS-DMA writes next to 192 MFMAs, a quarter of the kernel's time (profiles/r03_*).  Here a wave holds the operands of TWO
// tiles (2 x 2 x 96 registers: current and next layer), so a fragment feeds two MFMAs and a pass of the weights serves 128 samples:
// half the LDS traffic per MFMA.  What pays for the registers: the fragments of a chunk are no longer resident (96 registers) but a
// rolling window of HALF a chunk (4 k-blocks x 3 pieces = 48 registers: a piece's registers are refilled with the next half's right
// behind the last MFMA that reads them), and next round's table rows are not held across a round (64 registers per tile) but fetched
// during the head into the registers the next-layer operands have just left.
//
// Arithmetic: that of k_dvis_x6 (three-piece operands, six products, one fp32 accumulator per weight class; see its header).  The
// products of a class are summed half-chunk by half-chunk (h.h | h.m, m.h | h.l, m.m, l.h per four k-blocks), not product by product
// over the whole chunk: another -- equally valid -- fp32 summation order, so the results differ from k_dvis_x6's in the last bits.
#include "../../include/robir_hip.h"
#include "common.h"
#include "mlp_engine.h"
#include "x6_ring.h"

namespace rb {

#define RB_TINY 1e-6f
constexpr int XT_MAX_DIRS = 4096;
constexpr int XT_WF4 = 1536;             // weight part of a packed chunk: [kb 8][piece 3][lane 64] x 16 B = 24 KB
constexpr int XT_CF4 = 4 + XT_WF4;       // packed chunk in global memory: 16 bias floats + weights
constexpr int XT_SLOTS = 4, XT_DIST = 3;
constexpr int XT_PIECES = 6;             // 4 KB rows (1 KB per wave) of one chunk copy
// Round 6 (XT_FP8, default on; DESIGN section 5.6 item 10, profiles/r06_fp8_c2.md): of the six products of a multiply-add the two outer ones of
// the 2^-22 class -- h.xl and l.xh, each with an operand of at most three significant bits -- are formed from bf8 (e5m2) copies of their
// operands on v_mfma_f32_16x16x128_f8f6f4, which does a half chunk's 128 K in one instruction at twice the f16 rate: four f16 + two bf8
// products = five f16-equivalents instead of six.  The other four products (h.xh | h.xm, m.xh | m.xm) stay exact.  Error against a float64
// evaluation: unchanged (the two terms enter at 2^-22 and are right to two or three bits: <= 2^-24 of a product, below what fp32's own
// accumulation rounds away; tests/test_precision_gpu.py).  Register- and stream-neutral: the f16 l pieces of weights and activations are
// used nowhere else and go; bf8 copies of the h and l pieces take their place (weights: packing.repack_x6_chunks_fp8, named by
// scale_log2 = 8; activations: the top bytes of the f16 pieces, one v_perm_b32 per four values).  -DXT_FP8=0: six f16 products (rounds 4-5).
#ifndef XT_FP8
#define XT_FP8 1
#endif
#ifndef XT_PINGPONG
#define XT_PINGPONG 0
#endif

__device__ __forceinline__ void xt_dma16(const f4* gbase_uniform, unsigned lane_byte_off, unsigned lds_byte_uniform) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_byte_uniform), "v"(lane_byte_off),
               "s"(gbase_uniform)
               : "memory");
}
// ... with an immediate offset, which advances the global AND the LDS address (x6_ring.h: sx_dma_block)
template <int OFF>
__device__ __forceinline__ void xt_dma16_imm(const f4* gbase_uniform, unsigned lane_byte_off, unsigned lds_byte_uniform) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%3" ::"s"(lds_byte_uniform), "v"(lane_byte_off),
               "s"(gbase_uniform), "n"(OFF)
               : "memory");
}

template <int OFF>
__device__ __forceinline__ void xt_dma16_m0kept(const f4* gbase_uniform, unsigned lane_byte_off) {
  asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" ::"v"(lane_byte_off), "s"(gbase_uniform), "n"(OFF) : "memory");
}

struct XtTile {     // = V3Tile (vis_diffuse_v3.hip): record of a 16-sample tile of the global list
  int point;        // -1: no such tile
  int dir_base;     // first row of the point's chunk in dirs / Bd (chunk id * L * nsamp)
};
struct XtArgs {
  // both forms
  const float *A, *Bd;
  const f4* W49;
  int argmax_vis;
  unsigned* range_word;
  // one workgroup per point (STREAM = false)
  const float *normals, *dirs, *wdir, *wsum;
  const int* cid;
  int L, nsamp;
  float* vis_out;
  unsigned long long* eval_count;
  // persistent grid over the global tile list (STREAM = true; k_dvis3_cull / k_dvis3_reduce of vis_diffuse_v3.hip around it)
  const unsigned short* pair_j;
  const XtTile* tile_info;
  const unsigned long long* counters;
  float* pair_vis;
};

// STREAM = false: one workgroup per surface point, rounds of 128 of its front-facing directions (cull and per-lobe means in the kernel).
// STREAM = true: a persistent grid walks the global list of 16-sample tiles eight tiles per round, whatever point they belong to:
//   every tile has its own layer-0 point row (LDS-DMA into a wave-private slot during the previous round's head); the weight ring
//   runs continuously across rounds; per-pair visibilities go to a global array.  The same instruction sequence per pair: the two
//   forms are bit-identical.
template <bool STREAM>
__global__ __launch_bounds__(256, 1) void k_dvis_x6t(const XtArgs a) {
  __shared__ f4 ring[XT_SLOTS * XT_WF4];   // 96 KB
  __shared__ f4 headw[XT_WF4];             // 24 KB: chunk 48 (256 -> 2 head, rows 2..15 zero)
  __shared__ f4 bias_tab[49 * 4];
  // per point: vis_tab[4096] float | idx_list[4096] u16 | a_row[64] f4 = 25 KB; stream: a_rows [round parity][tile of the round][64] f4 = 16 KB
  __shared__ f4 aux[(XT_MAX_DIRS * 4 + XT_MAX_DIRS * 2) / 16 + 64];
  __shared__ int s_count;
  float* const vis_tab = reinterpret_cast<float*>(aux);
  unsigned short* const idx_list = reinterpret_cast<unsigned short*>(aux + XT_MAX_DIRS / 4);
  f4* const a_row = aux + (XT_MAX_DIRS * 6) / 16;
  f4* const a_rows = aux;
  const float* __restrict__ A = a.A;
  const float* __restrict__ Bd = a.Bd;
  const f4* __restrict__ W49 = a.W49;
  const int argmax_vis = a.argmax_vis;
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float negk = -2048.0f;
  constexpr float C11 = 1.0f / 2048.0f;
  // per-point form
  const long p = blockIdx.x;
  const int L = a.L, nsamp = a.nsamp, LS = L * nsamp;
  long dbase = 0;
  int S = 0, rounds = 0;
  // stream form
  const int G = gridDim.x;
  long total_tiles = 0, total_rounds = 0;
  for (int i = tid; i < 49 * 4; i += 256) bias_tab[i] = W49[(long)(i >> 2) * XT_CF4 + (i & 3)];
  for (int i = tid; i < XT_WF4; i += 256) headw[i] = W49[48L * XT_CF4 + 4 + i];
  if constexpr (STREAM) {
    total_tiles = (long)a.counters[0];
    total_rounds = (total_tiles + 7) >> 3;
    __syncthreads();
    if ((long)blockIdx.x >= total_rounds) return;           // workgroup-uniform
    rounds = 1;
  } else {
  const float* __restrict__ normals = a.normals;
  const float* __restrict__ dirs = a.dirs;
  const int* __restrict__ cid = a.cid;
  dbase = (cid ? (long)cid[p] : 0L) * LS;
  if (tid == 0) s_count = 0;
  if (tid < 64) a_row[tid] = reinterpret_cast<const f4*>(A + p * 256)[tid];
  for (int j = tid; j < LS; j += 256) vis_tab[j] = 0.f;
  __syncthreads();
  // ---- cull + compaction (order inside the list is irrelevant: results are scattered by direction index)
  const float nx = normals[3 * p], ny = normals[3 * p + 1], nz = normals[3 * p + 2];
  for (int j0 = 0; j0 < LS; j0 += 256) {
    const int j = j0 + tid;
    bool front = false;
    if (j < LS) {
      const float* d = dirs + 3 * (dbase + j);
      const float c = nx * d[0] + ny * d[1] + nz * d[2];  // sum(n*d): separate mul/add (-ffp-contract=off)
      front = c > RB_TINY;
    }
    const unsigned long long mk = __ballot(front);
    int base = 0;
    if (lane == 0 && mk) base = atomicAdd(&s_count, __popcll(mk));
    base = __shfl(base, 0);
    if (front) idx_list[base + __popcll(mk & ((1ull << lane) - 1ull))] = (unsigned short)j;
  }
  __syncthreads();
  S = s_count;
  if (tid == 0 && a.eval_count) atomicAdd(a.eval_count, (unsigned long long)S);
  rounds = (S + 127) / 128;
  }

  // ---- weight ring state
  const unsigned ring_b = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)ring);
  // a wave copies six consecutive 1 KB pieces of a 24 KB chunk: bytes [wave 6 KB, + 6 KB), addressed as (chunk base in SGPRs) + (one
  // of two per-lane offsets, 4 KB apart) + (an immediate 0 / 1 / 2 / 3 KB): no scalar address arithmetic per copy
  unsigned voff_a = (unsigned)wave * 6144u + (unsigned)lane * 16u, voff_b = voff_a + 4096u;
  asm volatile("" : "+v"(voff_a), "+v"(voff_b));
  const unsigned wave_lds = ring_b + (unsigned)wave * 6144u;      // + slot * 24576 (+ 4096 for pieces 4, 5) + the immediate
  // fragment (kb, piece) of a slot: ring_u[slot * XT_WF4 + (kb * 3 + piece) * 64].  Two base registers (slots 0, 1 | 2, 3) keep every
  // read's offset inside the 16-bit immediate of ds_read_b128 (the ring is 96 KB: from one base the compiler keeps an address
  // register per fragment, parked in the accumulator file and fetched back before every read)
  typedef const __attribute__((address_space(3))) u4* lds_u4p;
  unsigned ring_a0 = ring_b + (unsigned)lane * 16u, ring_a2 = ring_a0 + 2u * XT_WF4 * 16u;
  asm volatile("" : "+v"(ring_a0), "+v"(ring_a2));
  const lds_u4p ring_u = (lds_u4p)ring_a0, ring_u2 = (lds_u4p)ring_a2;
  u4 wh[4], wm[4], wl[4];              // rolling window: the fragments of four k-blocks (half a chunk)
#if XT_FP8
  typedef int xt_i8 __attribute__((ext_vector_type(8)));
  xt_i8 w8h, w8l;                      // ... and the half's bf8 fragments of the h and l pieces (wl unused)
  // a half chunk = 12 KB = 768 lane-strided u4: [k-block 0..3][h | m] (512), h8 (two planes: 128), l8 (128)
#define XT_F16OFF(KB, PIECE) (((KB) >> 2) * 768 + (((KB) & 3) * 2 + (PIECE)) * 64)
#define XT_LOAD8(DST, BASE, OFF)                                                                                             \
  {                                                                                                                          \
    const u4 a_ = (BASE)[(OFF)], b_ = (BASE)[(OFF) + 64];                                                                    \
    DST = xt_i8{(int)a_[0], (int)a_[1], (int)a_[2], (int)a_[3], (int)b_[0], (int)b_[1], (int)b_[2], (int)b_[3]};             \
  }
#endif
  f4 bias;
  if (rounds > 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int i = 0; i < XT_PIECES; ++i)
        xt_dma16(W49 + (long)c * XT_CF4 + 4 + i * 64, voff_a, wave_lds + (unsigned)c * 24576u + (unsigned)i * 1024u);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // chunk 0 landed; chunks 1, 2 stay in flight
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#if XT_FP8
      wh[k] = ring_u[XT_F16OFF(k, 0)];
      wm[k] = ring_u[XT_F16OFF(k, 1)];
#else
      wh[k] = ring_u[(k * 3 + 0) * 64];
      wm[k] = ring_u[(k * 3 + 1) * 64];
      wl[k] = ring_u[(k * 3 + 2) * 64];
#endif
    }
#if XT_FP8
    XT_LOAD8(w8h, ring_u, 512)
    XT_LOAD8(w8l, ring_u, 640)
#endif
    bias = bias_tab[g];
  }

  unsigned sat = 0u;                   // range sentinel: running max of the h pieces (all >= 0 here: ReLU outputs)
#if XT_FP8
  struct Ops {
    u4 h[2][8], m[2][8];               // B operands of a layer, two tiles: the f16 h and m pieces (one 128-bit tuple per k-block)
    xt_i8 h8[2][2], l8[2][2];          // ... and bf8 copies of the h and l pieces, 32 K values per lane and half (the top bytes of the f16 pieces)
  };
  unsigned el_keep[2] = {0u, 0u};
#else
  struct Ops {
    u4 h[2][8], m[2][8], l[2][8];      // B operands of a layer, two tiles (one 128-bit tuple per k-block and piece)
  };
#endif
  Ops P, Q;                            // current / next layer's operands, filled chunk by chunk; the layers alternate the roles

#define XT_MFMA(ACC, WREG, XREG) \
  ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, WREG), __builtin_bit_cast(h8, XREG), ACC, 0, 0, 0)

  // relu(z) of output block jb -> operands of the next layer: k-block jb/2, registers 2*(jb&1)+{0,1}; q = register pair.
  // Three stages per (tile, pair), spread over the MFMA groups of the next chunk.
  float ev0[2][2], ev1[2][2], ed0, ed1;
  unsigned eh;
  SxAcc prev[2];
  auto ep_stage1 = [&](int t, int q) {
    const SxAcc& a = prev[t];
    const float r0 = __builtin_fmaf(__builtin_fmaf(a.c2[2 * q], C11, a.c1[2 * q]), C11, a.c0[2 * q]);
    const float r1 = __builtin_fmaf(__builtin_fmaf(a.c2[2 * q + 1], C11, a.c1[2 * q + 1]), C11, a.c0[2 * q + 1]);
    ev0[t][q] = fmaxf(r0, 0.f);
    ev1[t][q] = fmaxf(r1, 0.f);
  };
  auto ep_stage2 = [&](int t, int q) {
    const unsigned hu = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(ev0[t][q], ev1[t][q]));
    const float s0 = ev0[t][q] * 2048.0f, s1 = ev1[t][q] * 2048.0f;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(ed0) : "v"(hu), "s"(negk), "v"(s0));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(ed1) : "v"(hu), "s"(negk), "v"(s1));
    eh = hu;
    sat = sat_acc_nonneg(sat, hu);
  };
  auto ep_stage3 = [&](Ops& Y, int t, int jb, int q) {
    const unsigned mu = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(ed0, ed1));
    const float e0 = ed0 * 2048.0f, e1 = ed1 * 2048.0f;
    unsigned lu;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(lu) : "v"(mu), "s"(negk), "v"(e0));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lu) : "v"(mu), "s"(negk), "v"(e1));
    Y.h[t][jb >> 1][(jb & 1) * 2 + q] = eh;
    Y.m[t][jb >> 1][(jb & 1) * 2 + q] = mu;
#if XT_FP8
    if (q == 0) {
      el_keep[t] = lu;
    } else {      // the block's four h and four l halves -> dword jb of the next layer's bf8 operands (their top bytes: e5m2 by truncation)
      Y.h8[t][jb >> 3][jb & 7] = (int)__builtin_amdgcn_perm(eh, Y.h[t][jb >> 1][(jb & 1) * 2], 0x07050301u);
      Y.l8[t][jb >> 3][jb & 7] = (int)__builtin_amdgcn_perm(lu, el_keep[t], 0x07050301u);
    }
#else
    Y.l[t][jb >> 1][(jb & 1) * 2 + q] = lu;
#endif
  };
  // the twelve stage instances of a chunk's epilogue in issue order
  auto ep_slot = [&](Ops& Y, int s, int pj) {
    switch (s) {
      case 0: ep_stage1(0, 0); break;
      case 1: ep_stage1(0, 1); break;
      case 2: ep_stage1(1, 0); break;
      case 3: ep_stage1(1, 1); break;
      case 4: ep_stage2(0, 0); break;
      case 5: ep_stage3(Y, 0, pj, 0); break;
      case 6: ep_stage2(0, 1); break;
      case 7: ep_stage3(Y, 0, pj, 1); break;
      case 8: ep_stage2(1, 0); break;
      case 9: ep_stage3(Y, 1, pj, 0); break;
      case 10: ep_stage2(1, 1); break;
      default: ep_stage3(Y, 1, pj, 1); break;
    }
  };

  // Layer-0 inputs (rows of the per-direction table): fetched during the head of the previous round into `raw`, which takes the
  // registers the next-layer operands have just left
  f4 raw[2][16];
  int jj[2], jjn[2];
  // stream form: records of this wave's two tiles in the round after the current one (wave-uniform point / table base, per-lane
  // direction index, 0xFFFF = padding), looked up at the top of a round and used by the fetches of its head
  int tpn[2] = {-1, -1}, tbn[2] = {0, 0}, jn2[2] = {0xFFFF, 0xFFFF};
  const unsigned arow_b = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)a_rows);
  // (branch-free: a conditional load makes the compiler wait for the record at the join, i.e. at the top of the round; loaded
  // unconditionally from a clamped index the record is waited for where the head uses it, a round later)
  XtTile recn[2] = {{-1, 0}, {-1, 0}};
  bool okn[2] = {false, false};
  auto tile_lookup = [&](long round) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const long T = round * 8 + wave * 2 + t;
      okn[t] = round < total_rounds && T < total_tiles;
      const long Tc = T < total_tiles ? T : total_tiles - 1;   // total_tiles >= 1 here
      recn[t] = a.tile_info[Tc];                               // wave-uniform address: scalar load
      jn2[t] = (int)a.pair_j[Tc * 16 + (lane & 15)];
    }
  };
  auto tile_resolve = [&]() {      // at the point of use (the head)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      tpn[t] = okn[t] ? __builtin_amdgcn_readfirstlane(recn[t].point) : -1;
      tbn[t] = okn[t] ? __builtin_amdgcn_readfirstlane(recn[t].dir_base) : 0;
      if (!okn[t]) jn2[t] = 0xFFFF;
    }
  };
  auto fetch_rows = [&](long rd_next, int parity_next) {
    if constexpr (STREAM) {
      tile_resolve();
      // the A rows of the next round's two tiles -> this wave's private slots by LDS-DMA (1 KB per tile): OLDER than the row loads
      // below, whose wait the compiler places in front of the layer-0 split
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const long prow = tpn[t] < 0 ? 0L : (long)tpn[t];
        xt_dma16(reinterpret_cast<const f4*>(A + prow * 256), (unsigned)lane * 16u,
                 arow_b + (unsigned)(parity_next * 8 + wave * 2 + t) * 1024u);
      }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      long row;
      if constexpr (STREAM) {
        jjn[t] = (tpn[t] < 0 || jn2[t] == 0xFFFF) ? -1 : jn2[t];
        row = (long)tbn[t] + (jjn[t] < 0 ? 0 : jjn[t]);
      } else {
        const int si = (int)rd_next * 128 + t * 64 + wave * 16 + (lane & 15);
        jjn[t] = si < S ? (int)idx_list[si] : -1;
        row = dbase + (jjn[t] < 0 ? 0 : jjn[t]);
      }
      const f4* brow = reinterpret_cast<const f4*>(Bd + row * 256) + g;
#pragma unroll
      for (int kb = 0; kb < 16; ++kb) raw[t][kb] = brow[kb * 4];
    }
  };
  long rd = STREAM ? (long)blockIdx.x : 0L;
  const long rd_end = STREAM ? total_rounds : (long)rounds, rd_step = STREAM ? (long)G : 1L;
  int parity = 0;
  if constexpr (STREAM) tile_lookup(rd);
  if (rounds > 0) fetch_rows(rd, 0);
#ifdef XT_TIMING   // phase stamps (tools/build_variant.sh ... -DXT_TIMING): profiles/r05_dvis_f16_phases.md
#define XT_T(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tacc[i] += t_ - tlast; tlast = t_; }
  unsigned long long tacc[4] = {0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
  int trounds = 0;
#else
#define XT_T(i)
#endif
  for (; rd < rd_end; rd += rd_step) {
    if constexpr (STREAM) tile_lookup(rd + rd_step);
#ifdef XT_TIMING
    ++trounds;
#endif
    XT_T(3)
    // ---- layer 0: relu(A[point] + Bd[dir]) straight into the operand registers
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      jj[t] = jjn[t];
      const f4* arow = STREAM ? a_rows + (parity * 8 + wave * 2 + t) * 64 : a_row;
#pragma unroll
      for (int kb = 0; kb < 16; ++kb) {
        const f4 bv = raw[t][kb];
        const f4 av = arow[kb * 4 + g];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          unsigned h, m, l;
          sx_split_pair(fmaxf(av[2 * q] + bv[2 * q], 0.f), fmaxf(av[2 * q + 1] + bv[2 * q + 1], 0.f), negk, h, m, l);
          P.h[t][kb / 2][(kb & 1) * 2 + q] = h;
          P.m[t][kb / 2][(kb & 1) * 2 + q] = m;
#if XT_FP8
          if (q == 0) {
            el_keep[t] = l;
          } else {
            P.h8[t][kb >> 3][kb & 7] = (int)__builtin_amdgcn_perm(h, P.h[t][kb / 2][(kb & 1) * 2], 0x07050301u);
            P.l8[t][kb >> 3][kb & 7] = (int)__builtin_amdgcn_perm(l, el_keep[t], 0x07050301u);
          }
#else
          P.l[t][kb / 2][(kb & 1) * 2 + q] = l;
#endif
          sat = sat_acc_nonneg(sat, h);
        }
      }
    }
    XT_T(0)
    auto layer = [&](int l, Ops& X, Ops& Y) {
      const f4* Wl = W49 + (long)l * 16 * XT_CF4 + 4;                          // this layer's chunk 0 weights
      const f4* Wn = W49 + (long)(l == 2 ? 0 : l + 1) * 16 * XT_CF4 + 4;        // next layer's (next round wraps to 0)
#pragma unroll
      for (int jb = 0; jb < 16; ++jb) {
        SxAcc acc[2];
        // chunk jb+1 has landed in its slot once at most the copy of chunk jb+2 (6 instructions) is still in flight; past the barrier
        // every wave has also finished with chunk jb-1 (its second half was read during chunk jb-1's first half), whose slot the copy
        // of chunk jb+3 reuses
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
#ifndef XT_ABL_NOBAR                  // timing ablation (wrong results): no per-chunk barrier
        __builtin_amdgcn_s_barrier();
#endif
        asm volatile("" ::: "memory");
        const int nx3 = jb + XT_DIST;
        const f4* dsrc = nx3 < 16 ? Wl + (long)nx3 * XT_CF4 : Wn + (long)(nx3 - 16) * XT_CF4;
        const unsigned ddst = wave_lds + (unsigned)(nx3 & 3) * 24576u;
        const f4 nbias = bias_tab[(l * 16 + jb + 1) * 4 + g];     // index 48 = head chunk after the last layer
#if defined(XT_ABL_NODMA)            // timing ablation (wrong results): no weight copies
#define XT_COPY(I) do { } while (0)
#else
  // pieces 0..5 are issued in this order in the chunk's second half with no other LDS-DMA between them: M0 is set by pieces 0 and 4
  // and carried (two scalar issue slots less for the other four)
#ifdef XT_M0_EVERY_PIECE             // A/B switch: M0 written by every piece (the round-4 form before this change)
#define XT_COPY(I)                                                                  \
  do {                                                                              \
    if ((I) < 4) xt_dma16_imm<((I) & 3) * 1024>(dsrc, voff_a, ddst);                \
    else xt_dma16_imm<((I) & 3) * 1024>(dsrc, voff_b, ddst + 4096u);                \
  } while (0)
#else
#define XT_COPY(I)                                                                  \
  do {                                                                              \
    if ((I) == 0) xt_dma16_imm<0>(dsrc, voff_a, ddst);                              \
    else if ((I) < 4) xt_dma16_m0kept<((I) & 3) * 1024>(dsrc, voff_a);              \
    else if ((I) == 4) xt_dma16_imm<0>(dsrc, voff_b, ddst + 4096u);                 \
    else xt_dma16_m0kept<((I) & 3) * 1024>(dsrc, voff_b);                           \
  } while (0)
#endif
#endif
#define XT_FENCE __builtin_amdgcn_sched_barrier(0)
#ifdef XT_ABL_NOLDS                   // timing ablation (wrong results): no fragment reads
#define XT_FRAG(DST, KB, PIECE) asm volatile("" : "+v"(DST))
#else
#if XT_FP8
#define XT_FRAG(DST, KB, PIECE) DST = nfrag[XT_F16OFF(KB, PIECE)]
#else
#define XT_FRAG(DST, KB, PIECE) DST = nfrag[((KB) * 3 + (PIECE)) * 64]
#endif
#endif
#define XT_MFMA8(ACC, WREG, XREG) ACC = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(WREG, XREG, ACC, 1, 1, 0, 0, 0, 0)
#define XT_EP(S)                        \
  do {                                  \
    if (jb > 0) ep_slot(Y, (S), jb - 1); \
  } while (0)
        // four MFMAs on one accumulator, the window's k-blocks k = 0..3 <-> k-blocks 4 H + k of the chunk; a chunk's first half walks
        // them upwards, its second half downwards, so that a half starts with the fragment that was requested LAST: its wait
        // covers the other three (fragments return in order), one s_waitcnt per piece and half instead of four
#define XT_RUN(ACC, WP, XP)                     \
  _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_) { \
    const int k = H ? 3 - k_ : k_;               \
    XT_MFMA(ACC, WP[k], XP[4 * H + k]);          \
  }
        // ... each fragment refilled with the next half's right behind the last MFMA that reads it
#define XT_RUN_REFILL(ACC, WP, XP, PIECE)          \
  _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_) { \
    const int k = H ? 3 - k_ : k_;                   \
    XT_MFMA(ACC, WP[k], XP[4 * H + k]);              \
    XT_FRAG(WP[k], 4 * (1 - H) + k, PIECE);          \
  }
#pragma unroll
        for (int H = 0; H < 2; ++H) {
          // the next half's fragments: second half of this chunk's slot | first half of the next chunk's
          const lds_u4p nfrag = (((jb + H) & 2) ? ring_u2 : ring_u) + ((jb + H) & 1) * XT_WF4;
          if (H == 0) {
            acc[0].c0 = bias;
            acc[1].c0 = bias;
            acc[0].c1 = acc[0].c2 = acc[1].c1 = acc[1].c2 = f4{0.f, 0.f, 0.f, 0.f};
          }
          XT_RUN(acc[0].c0, wh, X.h[0]);                       // 1
          if (H == 0) XT_EP(0); else XT_EP(9);
          XT_FENCE;
          XT_RUN(acc[0].c1, wh, X.m[0]);                       // 2
          if (H == 0) XT_EP(1); else XT_EP(10);
          XT_FENCE;
#if XT_FP8
          XT_MFMA8(acc[0].c2, w8h, X.l8[0][H]);                // 3: h.xl of the half's 128 K as ONE bf8 MFMA
          if (H == 0) XT_EP(2); else XT_EP(11);
          XT_FENCE;
          XT_RUN(acc[1].c0, wh, X.h[1]);                       // 4
          if (H == 0) XT_EP(3); else XT_COPY(0);
          XT_FENCE;
#ifndef XT_FP8_EP_AT6
#define XT_FP8_EP_AT6 1     // the filler of position 5 (which also carries the four h refills) behind the bf8 MFMA of position 6: a single 32-cycle MFMA shadows its fillers better than the last of a run of four (-0.8 %, bit-identical); 0: at position 5
#endif
          XT_RUN_REFILL(acc[1].c1, wh, X.m[1], 0);             // 5: the f16 h fragments' last use
          if (!XT_FP8_EP_AT6) { if (H == 0) XT_EP(4); else XT_COPY(1); }
          XT_FENCE;
          XT_MFMA8(acc[1].c2, w8h, X.l8[1][H]);                // 6: the bf8 h fragment's last use
#ifndef XT_ABL_NOLDS
          XT_LOAD8(w8h, nfrag, (1 - H) * 768 + 512)
#endif
          if (XT_FP8_EP_AT6) { if (H == 0) XT_EP(4); else XT_COPY(1); }
          XT_FENCE;
#else
          XT_RUN(acc[0].c2, wh, X.l[0]);                       // 3
          if (H == 0) XT_EP(2); else XT_EP(11);
          XT_FENCE;
          XT_RUN(acc[1].c0, wh, X.h[1]);                       // 4
          if (H == 0) XT_EP(3); else XT_COPY(0);
          XT_FENCE;
          XT_RUN(acc[1].c1, wh, X.m[1]);                       // 5
          if (H == 0) XT_EP(4); else XT_COPY(1);
          XT_FENCE;
          XT_RUN_REFILL(acc[1].c2, wh, X.l[1], 0);             // 6: the h fragments' last use
          XT_FENCE;
#endif
          XT_RUN(acc[0].c1, wm, X.h[0]);                       // 7
          if (H == 0) XT_EP(5); else XT_COPY(2);
          XT_FENCE;
          XT_RUN(acc[0].c2, wm, X.m[0]);                       // 8
          if (H == 0) XT_EP(6); else XT_COPY(3);
          XT_FENCE;
#ifndef XT_FP8_EP_AT12
#define XT_FP8_EP_AT12 0    // 1: the fillers of positions 9 / 11 behind the bf8 MFMAs of positions 11 / 12 (XT_FP8 only)
#endif
          XT_RUN(acc[1].c1, wm, X.h[1]);                       // 9
          if (!(XT_FP8 && XT_FP8_EP_AT12)) { if (H == 0) XT_EP(7); else XT_COPY(4); }
          XT_FENCE;
          XT_RUN_REFILL(acc[1].c2, wm, X.m[1], 1);             // 10: the m fragments' last use
          XT_FENCE;
#if XT_FP8
          XT_MFMA8(acc[0].c2, w8l, X.h8[0][H]);                // 11: l.xh as one bf8 MFMA
          if (XT_FP8_EP_AT12) { if (H == 0) XT_EP(7); else XT_COPY(4); } else { if (H == 0) XT_EP(8); else XT_COPY(5); }
          XT_FENCE;
          XT_MFMA8(acc[1].c2, w8l, X.h8[1][H]);                // 12: the bf8 l fragment's last use
#ifndef XT_ABL_NOLDS
          XT_LOAD8(w8l, nfrag, (1 - H) * 768 + 640)
#endif
          if (XT_FP8_EP_AT12) { if (H == 0) XT_EP(8); else XT_COPY(5); }
          XT_FENCE;
#else
          XT_RUN(acc[0].c2, wl, X.h[0]);                       // 11
          if (H == 0) XT_EP(8); else XT_COPY(5);
          XT_FENCE;
          XT_RUN_REFILL(acc[1].c2, wl, X.h[1], 2);             // 12: the l fragments' last use
          XT_FENCE;
#endif
        }
#undef XT_RUN
#undef XT_RUN_REFILL
#undef XT_COPY
#undef XT_FENCE
#undef XT_FRAG
#undef XT_MFMA8
#undef XT_EP
        prev[0] = acc[0];
        prev[1] = acc[1];
        bias = nbias;
      }
#pragma unroll
      for (int s = 0; s < 12; ++s) ep_slot(Y, s, 15);
    };
#if XT_PINGPONG     // three copies of the layer body, the operand sets alternate: no register moves at a layer's end
    layer(0, P, Q);
    layer(1, Q, P);
    layer(2, P, Q);
    Ops& HX = Q;
#else
#pragma unroll 1
    for (int l = 0; l < 3; ++l) {
      layer(l, P, Q);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
          P.h[t][kb] = Q.h[t][kb];
          P.m[t][kb] = Q.m[t][kb];
#if !XT_FP8
          P.l[t][kb] = Q.l[t][kb];
#endif
        }
#if XT_FP8
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          P.h8[t][hf] = Q.h8[t][hf];
          P.l8[t][hf] = Q.l8[t][hf];
        }
#endif
    }
    Ops& HX = P;
#endif
    // ---- head: chunk 48 from its resident LDS copy; `bias` holds its bias (fetched by the last chunk of layer 2) and the fragment
    // window already holds the first half of the next round's chunk 0.  Next round's rows are requested first: they arrive under
    // the head's MFMAs (clamped to this round's samples after the final round: harmless)
    XT_T(1)
    if constexpr (STREAM) fetch_rows(rd + rd_step, parity ^ 1);
    else fetch_rows(rd + 1 < rd_end ? rd + 1 : rd, 0);
    {
      const u4* hw = reinterpret_cast<const u4*>(headw) + lane;
      SxAcc acc[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        acc[t].c0 = bias;
        acc[t].c1 = f4{0.f, 0.f, 0.f, 0.f};
        acc[t].c2 = f4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
#if XT_FP8
      for (int kb = 0; kb < 8; ++kb) {
        const u4 fh = hw[XT_F16OFF(kb, 0)], fm = hw[XT_F16OFF(kb, 1)];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          XT_MFMA(acc[t].c0, fh, HX.h[t][kb]);
          XT_MFMA(acc[t].c1, fh, HX.m[t][kb]);
          XT_MFMA(acc[t].c1, fm, HX.h[t][kb]);
          XT_MFMA(acc[t].c2, fm, HX.m[t][kb]);
        }
      }
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        xt_i8 f8h, f8l;
        XT_LOAD8(f8h, hw, hf * 768 + 512)
        XT_LOAD8(f8l, hw, hf * 768 + 640)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          acc[t].c2 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(f8h, HX.l8[t][hf], acc[t].c2, 1, 1, 0, 0, 0, 0);
          acc[t].c2 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(f8l, HX.h8[t][hf], acc[t].c2, 1, 1, 0, 0, 0, 0);
        }
      }
#else
      for (int kb = 0; kb < 8; ++kb) {
        const u4 fh = hw[(kb * 3 + 0) * 64], fm = hw[(kb * 3 + 1) * 64], fl = hw[(kb * 3 + 2) * 64];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          XT_MFMA(acc[t].c0, fh, HX.h[t][kb]);
          XT_MFMA(acc[t].c1, fh, HX.m[t][kb]);
          XT_MFMA(acc[t].c2, fh, HX.l[t][kb]);
          XT_MFMA(acc[t].c1, fm, HX.h[t][kb]);
          XT_MFMA(acc[t].c2, fm, HX.m[t][kb]);
          XT_MFMA(acc[t].c2, fl, HX.h[t][kb]);
        }
      }
#endif
      bias = bias_tab[g];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float l0 = __builtin_fmaf(__builtin_fmaf(acc[t].c2[0], C11, acc[t].c1[0]), C11, acc[t].c0[0]);
        const float l1 = __builtin_fmaf(__builtin_fmaf(acc[t].c2[1], C11, acc[t].c1[1]), C11, acc[t].c0[1]);
        if (g == 0 && jj[t] >= 0) {
          float v;
          if (argmax_vis) {
            v = l1 > l0 ? 1.f : 0.f;
          } else {
            const float mx = fmaxf(l0, l1);
            const float e0 = expf(l0 - mx), e1 = expf(l1 - mx);
            v = e1 / (e0 + e1);
          }
          if constexpr (STREAM) a.pair_vis[(rd * 8 + wave * 2 + t) * 16 + (lane & 15)] = v;
          else vis_tab[jj[t]] = v;
        }
      }
    }
    parity ^= 1;
    XT_T(2)
  }
#ifdef XT_TIMING
  if ((blockIdx.x == 7 || blockIdx.x == 4000) && tid == 0 && trounds > 0)
    printf("x6t wg %d rounds %d cycles/round: conv %llu layers %llu head+fetch %llu lookup %llu\n", (int)blockIdx.x, trounds, tacc[0] / trounds,
           tacc[1] / trounds, tacc[2] / trounds, tacc[3] / trounds);
#endif
#undef XT_MFMA
  range_report<true>(sat, a.range_word);
  // drain the ring (copies still target this workgroup's LDS)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if constexpr (!STREAM) {
    if (tid < L) {
      const float* w = a.wdir + dbase + (long)tid * nsamp;
      float acc = 0.f;
      for (int k = 0; k < nsamp; ++k) acc += vis_tab[tid * nsamp + k] * w[k];
      a.vis_out[p * L + tid] = acc / a.wsum[(a.cid ? a.cid[p] : 0) * L + tid];
    }
  }
}

// the cull / per-lobe reduce passes around the stream form: precision-agnostic (dvis_tiles.hip), shared with the legacy split-precision family
struct V3Tile;
__global__ void k_dvis3_cull(const float* __restrict__ normals, const int* __restrict__ cid, long n, const float* __restrict__ dirs, int LS,
                             unsigned short* __restrict__ pair_j, V3Tile* __restrict__ tile_info, int2* __restrict__ point_info,
                             unsigned long long* __restrict__ counters, unsigned long long* __restrict__ eval_count);
__global__ void k_dvis3_reduce(const int* __restrict__ cid, long n, const float* __restrict__ wdir, const float* __restrict__ wsum,
                               const unsigned short* __restrict__ pair_j, const float* __restrict__ pair_vis,
                               const int2* __restrict__ point_info, int L, int nsamp, float* __restrict__ vis_out);

}  // namespace rb

using namespace rb;

extern "C" int rb_dvis_fused_x6t(const float* normals, const int* chunk_id, long n, const float* A, const float* Bd,
                                 const float* dirs, const float* wdir, const float* wsum, const float* W49, int L, int nsamp,
                                 int argmax_vis, int scale_log2, float* vis_out, unsigned long long* eval_count,
                                 rb_stream_t stream) {
  if (n <= 0) return 0;
  RB_REQUIRE(normals && A && Bd && dirs && wdir && wsum && W49 && vis_out, "null pointer");
  RB_REQUIRE(n <= RB_MAX_BLOCKS, "too many points for one launch (one workgroup each)");
  RB_REQUIRE(L > 0 && L <= 256 && nsamp > 0 && (long)L * nsamp <= XT_MAX_DIRS, "need L <= 256 and L*nsamp <= 4096");
#if XT_FP8
  RB_REQUIRE(scale_log2 == 8, "this build of k_dvis_x6t takes the bf8 weight layout (packing.repack_x6_chunks_fp8; scale_log2 = 8 names it)");
#else
  RB_REQUIRE(scale_log2 == 0, "k_dvis_x6t takes weights packed with scale_log2 = 0");
#endif
  XtArgs a{};
  a.A = A, a.Bd = Bd, a.W49 = (const f4*)W49, a.argmax_vis = argmax_vis;
  a.range_word = range_flags() ? range_flags() + RB_RANGE_DVIS : nullptr;
  a.normals = normals, a.dirs = dirs, a.wdir = wdir, a.wsum = wsum, a.cid = chunk_id, a.L = L, a.nsamp = nsamp;
  a.vis_out = vis_out, a.eval_count = eval_count;
  hipLaunchKernelGGL(k_dvis_x6t<false>, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("k_dvis_x6t");
}

extern "C" int rb_dvis_stream_x6(const float* normals, const int* chunk_id, long n, const float* A, const float* Bd,
                                 const float* dirs, const float* wdir, const float* wsum, const float* W49, int L, int nsamp,
                                 int argmax_vis, int scale_log2, unsigned short* pair_j, float* pair_vis, int* tile_info,
                                 int* point_info, unsigned long long* counters, int n_workgroups, float* vis_out,
                                 unsigned long long* eval_count, rb_stream_t stream) {
  if (n <= 0) return 0;
  RB_REQUIRE(normals && A && Bd && dirs && wdir && wsum && W49 && vis_out, "null pointer");
  RB_REQUIRE(pair_j && pair_vis && tile_info && point_info && counters, "null scratch pointer");
  RB_REQUIRE(n <= RB_MAX_BLOCKS, "too many points for one launch (one workgroup each in the cull / reduce passes)");
  RB_REQUIRE(L > 0 && L <= 256 && nsamp > 0 && (long)L * nsamp <= XT_MAX_DIRS && (L * nsamp) % 16 == 0,
             "need L <= 256, L*nsamp <= 4096 and a multiple of 16");
  RB_REQUIRE((long)n * (L * nsamp / 16) < (1L << 31), "tile index would overflow 31 bits");
#if XT_FP8
  RB_REQUIRE(scale_log2 == 8, "this build of k_dvis_x6t takes the bf8 weight layout (packing.repack_x6_chunks_fp8; scale_log2 = 8 names it)");
#else
  RB_REQUIRE(scale_log2 == 0, "k_dvis_x6t takes weights packed with scale_log2 = 0");
#endif
  hipStream_t s = (hipStream_t)stream;
  if (n_workgroups <= 0) n_workgroups = device_cus();
  RB_REQUIRE(n_workgroups > 0, "device query failed");
  if (hipMemsetAsync(counters, 0, 2 * sizeof(unsigned long long), s) != hipSuccess) return rb::fail(__func__, "memset failed");
  hipLaunchKernelGGL(k_dvis3_cull, dim3((unsigned)n), dim3(256), 0, s, normals, chunk_id, n, dirs, L * nsamp, pair_j,
                     reinterpret_cast<V3Tile*>(tile_info), reinterpret_cast<int2*>(point_info), counters, eval_count);
  if (int rc = check_launch("k_dvis3_cull")) return rc;
  XtArgs a{};
  a.A = A, a.Bd = Bd, a.W49 = (const f4*)W49, a.argmax_vis = argmax_vis;
  a.range_word = range_flags() ? range_flags() + RB_RANGE_DVIS : nullptr;
  a.pair_j = pair_j, a.tile_info = reinterpret_cast<const XtTile*>(tile_info), a.counters = counters, a.pair_vis = pair_vis;
  hipLaunchKernelGGL(k_dvis_x6t<true>, dim3((unsigned)n_workgroups), dim3(256), 0, s, a);
  if (int rc = check_launch("k_dvis_x6t<stream>")) return rc;
  hipLaunchKernelGGL(k_dvis3_reduce, dim3((unsigned)n), dim3(256), 0, s, chunk_id, n, wdir, wsum, pair_j, pair_vis,
                     reinterpret_cast<const int2*>(point_info), L, nsamp, vis_out);
  return check_launch("k_dvis3_reduce");
}
