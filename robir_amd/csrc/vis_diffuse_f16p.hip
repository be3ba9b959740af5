// Light-SG visibility in PLAIN f16 (the labelled throughput mode of vis_diffuse_f16t.hip: one f16 MFMA product per multiply-add, f16
// weights, activations truncated to f16 between the layers, fp32 sums -- NARROWER than the reference's fp32, ROBIR_PRECISION=f16 only)
// in the POINT-BLOCK form, round 5.  (get_diffuse_visibility, model/sg_render.py:111-195; VisNetwork,
// model/implicit_differentiable_renderer.py:241-258.)
//
// Why another form.  The tile-list kernels (k_dvis_x6t<stream>, k_dvis_f16t / f16t2) cut ONE point's surviving directions into 16-sample
// tiles: the sixteen samples of a tile are sixteen different rows of the per-direction table Bd (1 KB each), and whatever the operand
// layout, the sixteen lanes of a quarter-wave hold sixteen different samples -- every row load of the layer-0 gather touches sixteen
// cache lines for 64 B each.  With six MFMAs per multiply-add (exact operands) that gather hides; with ONE it does not: s_memtime
// stamps in k_dvis_f16t2 gave, per round of sixteen tiles, 31.5 k cycles in the three hidden layers and 16 k in the head + the issue
// of the next round's 64 row loads per wave (the texture path accepts them at ~250 cycles each), 5.9 k in the layer-0 conversion that
// waits for them, 2.2 k in the tile lookup: 43 % of the kernel outside its 49 chunks (profiles/r05_dvis_f16_phases.md).
//
// Here a tile is SIXTEEN POINTS x ONE direction: points 16 b .. 16 b + 15 (consecutive hit pixels of one 1024-pixel chunk: the same
// direction set, near-equal normals) against direction j of their chunk, kept when ANY of the sixteen faces it (n.d > 1e-6,
// sg_render.py:155); lanes whose point does not are computed and dropped by the reduce pass (3.6 % more tiles than the per-point
// compaction on the synthetic view, tools/pblock_waste.py).  A round = sixteen consecutive kept directions of ONE point block: it needs
// 16 rows of A (the points) and 16 rows of Bd (the directions) -- 32 KB by 32 whole-row LDS-DMA copies per workgroup (eight per wave,
// 1 KB contiguous each) instead of 256 KB by 256 sixteen-line loads -- requested two barrier steps into the second hidden layer of the
// round before, long landed when the conversion reads them: every lane its own point's row (row stride 1040 B: conflict-free b128
// reads), the direction row broadcast.
//   k_dvis_pb_cull    one workgroup per point block (and chunk id in it: blocks at a chunk boundary give two items): the kept
//                     directions in ascending order with their 16-bit point masks, padded to whole rounds; rounds allocated by one
//                     atomic per item;
//   k_dvis_f16p       the persistent grid over the rounds: k_dvis_f16t2's hidden layers and head (same products, same summation order
//                     per pair: the same bits per pair), pair values to pair_vis[item][entry][16 points];
//   k_dvis_pb_reduce  one workgroup per item: the SG-weighted mean per lobe and point in the fixed sample order over the pairs the
//                     point faces (sg_render.py:177-190) -- the sums of k_dvis3_reduce term by term (it adds 0 * w for the others).
// chunk_id must be ASCENDING (the renderer's hit points are: pixel order); a caller that breaks this gets NaN in vis_out, not numbers.
#include "../../include/robir_hip.h"
#include "common.h"
#include "mlp_engine.h"
#include "x6_ring.h"
#include "x6t_engine.h"

namespace rb {

#define RB_TINY 1e-6f
constexpr int FP_WF4 = 512;              // h fragments of a chunk: [kb 8][lane 64] x 16 B = 8 KB
constexpr int FP_TILES = 4;
constexpr int FP_BIAS_F4 = 49 * 4;       // the f16 blob's bias head (packing.pack_vis_f16_head)
constexpr int FP_SLOTS = 8;              // ring slots (8 KB): slot = chunk % 8
constexpr int FP_DIST = FP_SLOTS / 2 - 2;
constexpr int FP_AROW_F4 = 65;           // an A row in the LDS: 1 KB + 16 B (the 16 points of a quarter-wave then read 16 different bank groups)

struct PbRound {
  int item, e0;       // entries e0 .. e0 + 15 of the item's list
  int p0, dir_base;   // first point of the block; first row of its chunk in dirs / Bd
};
struct PbItem {
  int p0, chunk, count, pad;
};

// counters: [0] rounds allocated, [1] facing pairs (statistics), [2] items allocated, [3] != 0: chunk ids not ascending / items overflow
__global__ __launch_bounds__(256) void k_dvis_pb_cull(const float* __restrict__ normals, const int* __restrict__ cid, long n,
                                                       const float* __restrict__ dirs, int LS, int items_max, unsigned* __restrict__ entries,
                                                       PbRound* __restrict__ round_info, PbItem* __restrict__ item_info,
                                                       unsigned long long* __restrict__ counters, unsigned long long* __restrict__ eval_count) {
  __shared__ float s_n[16][3];
  __shared__ int s_c[16];
  __shared__ int s_wcount[4];
  __shared__ int s_item, s_r0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long p0 = (long)blockIdx.x * 16;
  if (tid < 16) {
    const long p = p0 + tid;
    const bool ok = p < n;
    s_c[tid] = ok ? (cid ? cid[p] : 0) : -1;
    for (int k = 0; k < 3; ++k) s_n[tid][k] = ok ? normals[3 * p + k] : 0.f;
    // ascending chunk ids: inside the block and against the last point of the block before
    if (ok && cid) {
      const long q = p > 0 ? p - 1 : 0;
      if (cid[q] > cid[p] || cid[p] < 0) atomicAdd(&counters[3], 1ull);
    }
  }
  __syncthreads();
  for (int f = 0; f < 16; ++f) {
    const int c = s_c[f];                  // workgroup-uniform
    if (c < 0) continue;
    if (f > 0 && s_c[f - 1] == c) continue;   // ascending ids: a chunk's points are adjacent
    unsigned member = 0;
    for (int m = 0; m < 16; ++m)
      if (s_c[m] == c) member |= 1u << m;
    __syncthreads();
    if (tid == 0) s_item = (int)atomicAdd(&counters[2], 1ull);
    __syncthreads();
    const int item = s_item;
    if (item >= items_max) {               // cannot happen with ascending ids (items <= blocks + chunks - 1): reported, never silent
      if (tid == 0) atomicAdd(&counters[3], 1ull);
      continue;
    }
    const long dbase = (long)c * LS;
    unsigned* ent = entries + (long)item * LS;
    int count = 0;
    unsigned pairs = 0;
    for (int j0 = 0; j0 < LS; j0 += 256) {
      const int j = j0 + tid;
      unsigned mask = 0;
      if (j < LS) {
        const float* d = dirs + 3 * (dbase + j);
        const float dx = d[0], dy = d[1], dz = d[2];
#pragma unroll
        for (int m = 0; m < 16; ++m) {
          const float cs = s_n[m][0] * dx + s_n[m][1] * dy + s_n[m][2] * dz;   // sum(n*d): separate mul/add (-ffp-contract=off), as k_dvis3_cull
          if (((member >> m) & 1u) && cs > RB_TINY) mask |= 1u << m;
        }
      }
      const bool keep = mask != 0;
      const unsigned long long bal = __ballot(keep);
      if (lane == 0) s_wcount[wave] = __popcll(bal);
      __syncthreads();
      int base = count;
      for (int w = 0; w < wave; ++w) base += s_wcount[w];
      if (keep) ent[base + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned)j | (mask << 16);
      count += s_wcount[0] + s_wcount[1] + s_wcount[2] + s_wcount[3];
      pairs += __popc(mask);
      __syncthreads();
    }
    const int padded = (count + 15) & ~15;
    for (int i = count + tid; i < padded; i += 256) ent[i] = 0u;      // direction 0, no point: computed, never read
    for (int o = 32; o > 0; o >>= 1) pairs += __shfl_down(pairs, o);
    if (lane == 0 && pairs) {
      atomicAdd(&counters[1], (unsigned long long)pairs);
      if (eval_count) atomicAdd(eval_count, (unsigned long long)pairs);
    }
    const int nr = padded >> 4;
    if (tid == 0) {
      s_r0 = nr ? (int)atomicAdd(&counters[0], (unsigned long long)nr) : 0;
      item_info[item] = PbItem{(int)p0, c, count, 0};
    }
    __syncthreads();
    const int r0 = s_r0;
    for (int i = tid; i < nr; i += 256) round_info[r0 + i] = PbRound{item, i * 16, (int)p0, (int)dbase};
  }
}

struct FpArgs {
  const float *A, *Bd;
  const f4* W;               // [49][4] biases | [49][8][64] h fragments (packing.pack_vis_f16_head)
  int argmax_vis, LS;
  long n;
  const unsigned long long* counters;
  float* pair_vis;
};

typedef _Float16 fp_h2 __attribute__((ext_vector_type(2)));
typedef float fp_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned fp_relu_pack(float a, float b) {      // relu after the truncation, on the pair: max(rtz(x), 0) = rtz(max(x, 0))
  const fp_h2 h = __builtin_bit_cast(fp_h2, __builtin_amdgcn_cvt_pkrtz(a, b));
  const fp_h2 z = fp_h2{(_Float16)0.0f, (_Float16)0.0f};
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(h, z));
}

#ifdef FP_TIMING
#define FP_T(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tacc[i] += t_ - tlast; tlast = t_; }
#else
#define FP_T(i)
#endif

// (entries / round_info as kernel parameters of their own: `const __restrict__` there lets the compiler read them through the scalar
// cache -- wave-uniform addresses -- instead of as vector loads whose waits would drain the weight copies in flight)
__global__ __launch_bounds__(256, 1) void k_dvis_f16p(const FpArgs a, const unsigned* __restrict__ entries, const PbRound* __restrict__ round_info) {
  __shared__ f4 ring[FP_SLOTS * FP_WF4];    // 64 KB
  __shared__ f4 headw[FP_WF4];              // 8 KB: chunk 48 (256 -> 2 head)
  __shared__ f4 bias_tab[49 * 4];
  __shared__ f4 a_lds[16 * FP_AROW_F4];     // the round's sixteen points
  __shared__ f4 b_lds[16 * 64];             // the round's sixteen directions (tile = wave * 4 + t)
  const float* __restrict__ A = a.A;
  const float* __restrict__ Bd = a.Bd;
  const f4* __restrict__ Wb = a.W;
  const f4* __restrict__ Wf = a.W + FP_BIAS_F4;
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = gridDim.x;
  const long total_rounds = (long)a.counters[0];
  for (int i = tid; i < 49 * 4; i += 256) bias_tab[i] = Wb[i];
  for (int i = tid; i < FP_WF4; i += 256) headw[i] = Wf[48L * FP_WF4 + i];
  __syncthreads();
  if ((long)blockIdx.x >= total_rounds) return;           // workgroup-uniform

  const unsigned ring_b = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)ring);
  const unsigned arow_b = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)a_lds);
  const unsigned brow_b = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)b_lds);
  unsigned ring_lane = ring_b + (unsigned)lane * 16u;
  asm volatile("" : "+v"(ring_lane));
  typedef const __attribute__((address_space(3))) u4* lds_u4p;
  // this wave's 2 KB (k-blocks 2 w, 2 w + 1) of chunk c -> slot c % 8: one M0 setting, two copies
  auto copy_chunk = [&](int c, unsigned lv) {
    const f4* src = Wf + (long)c * FP_WF4 + (2 * wave) * 64;
    xt_dma16_imm<0>(src, lv, ring_b + (unsigned)(c % FP_SLOTS) * 8192u + (unsigned)(2 * wave) * 1024u);
    xt_dma16_keep<1024>(src, lv);
  };
  // the rows of a round, eight whole-row copies per wave: the Bd rows of its own four tiles, four of the sixteen A rows
  // (the entries of the wave's four tiles: requested by load_entries a layer before issue_rows uses them)
  typedef unsigned fp_u4 __attribute__((ext_vector_type(4)));
  auto load_entries = [&](const PbRound& R) {
    return *reinterpret_cast<const fp_u4*>(entries + (long)R.item * a.LS + R.e0 + wave * FP_TILES);   // wave-uniform, 16-byte aligned
  };
  auto issue_rows = [&](const PbRound& R, const fp_u4& ent) {
    const unsigned lv = xt_lane16<0>();
#pragma unroll
    for (int t = 0; t < FP_TILES; ++t) {
      const long row = (long)R.dir_base + (long)(__builtin_amdgcn_readfirstlane((int)ent[t]) & 0xFFFF);
      xt_dma16_imm<0>(reinterpret_cast<const f4*>(Bd + row * 256), lv, brow_b + (unsigned)(wave * FP_TILES + t) * 1024u);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      long p = (long)R.p0 + wave * 4 + i;
      if (p >= a.n) p = a.n - 1;                                                        // the last block's padding lanes: masked off
      xt_dma16_imm<0>(reinterpret_cast<const f4*>(A + p * 256), lv, arow_b + (unsigned)(wave * 4 + i) * (unsigned)(FP_AROW_F4 * 16));
    }
  };
  auto lookup = [&](long r) {                                                             // wave-uniform address
    const PbRound R = round_info[r < total_rounds ? r : total_rounds - 1];
    return PbRound{__builtin_amdgcn_readfirstlane(R.item), __builtin_amdgcn_readfirstlane(R.e0), __builtin_amdgcn_readfirstlane(R.p0),
                   __builtin_amdgcn_readfirstlane(R.dir_base)};
  };

  u4 wh[2][8];                    // the fragments of the current chunk and of the next one, by chunk parity
  f4 bias;
  long rd = blockIdx.x;
  PbRound cur = lookup(rd);
  {
    const unsigned lv = xt_lane16<0>();
#pragma unroll
    for (int c = 0; c < 2 * FP_DIST; ++c) copy_chunk(c, lv);          // steps 0 .. FP_DIST - 1
  }
  issue_rows(cur, load_entries(cur));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // once per workgroup
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int k = 0; k < 8; ++k) wh[0][k] = ((lds_u4p)ring_lane)[k * 64];
  bias = bias_tab[g];

  u4 P[FP_TILES][8], Q[FP_TILES][8];
#define FP_MFMA(ACC, WREG, XREG) \
  ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, WREG), __builtin_bit_cast(h8, XREG), ACC, 0, 0, 0)

  // round r + G's record is requested a round early (top of round r - G), its entries at the top of round r: both scalar loads are in
  // flight together and long back when the second layer uses them
  PbRound nxt = lookup(rd + G), nn = nxt;
  fp_u4 ent_n = {0u, 0u, 0u, 0u};
  // one hidden layer: operands X -> outputs Y (the next layer's operands); cb = the layer's first chunk (0, 16, 32).  Steps of two
  // chunks: the copies of step s + FP_DIST at the top of step s, one counted wait + barrier per 64 MFMAs (vis_diffuse_f16t.hip,
  // second generation).  rows: the next round's row copies go out behind the barrier of step 0 -- YOUNGER than the ring copies the
  // wait of step 1 is for (it leaves them in flight: 12 instead of 4), older than those of step 2, whose wait therefore covers them
  auto run_layer = [&](const u4 (&X)[FP_TILES][8], u4 (&Y)[FP_TILES][8], int cb, bool rows) {
    f4 acc[2][FP_TILES];
#pragma unroll
    for (int sb = 0; sb < 8; ++sb) {
      {
        const int n4 = (cb + 2 * sb + 2 * FP_DIST) % 48;
        const unsigned lv = xt_lane16<0>();
        copy_chunk(n4, lv);
        copy_chunk(n4 + 1, lv);
      }
      if (rows && sb == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (FP_DIST - 1) + 8) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (FP_DIST - 1)) : "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (rows && sb == 0) issue_rows(nxt, ent_n);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int jb = 2 * sb + h;
        const lds_u4p nfrag = (lds_u4p)(ring_lane + (unsigned)((cb + jb + 1) % FP_SLOTS) * 8192u);
        const f4 nbias = bias_tab[(cb + jb + 1) * 4 + g];
        const u4 (&wc)[8] = wh[jb & 1];
        u4 (&wn)[8] = wh[(jb & 1) ^ 1];
        f4 (&ac)[FP_TILES] = acc[jb & 1];
        const f4 (&pv)[FP_TILES] = acc[(jb & 1) ^ 1];
#pragma unroll
        for (int k_ = 0; k_ < 8; ++k_) {
          const int k = (jb & 1) ? 7 - k_ : k_;       // k_dvis_f16t's summation order (odd chunks downwards): the same bits per pair
#pragma unroll
          for (int t = 0; t < FP_TILES; ++t) {
            if (k_ == 0) ac[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, wc[k]), __builtin_bit_cast(h8, X[t][k]), bias, 0, 0, 0);
            else FP_MFMA(ac[t], wc[k], X[t][k]);
          }
          // the other work of a k-step behind its four MFMAs: one fragment read of the next chunk, relu + truncation of one value pair
          // of the previous chunk (one instruction behind EACH MFMA instead: measured the same, 74.1 against 74.4 ms)
          wn[k_] = nfrag[k_ * 64];
          if (jb > 0) {
            const int t = k_ >> 1, pj = jb - 1;
            Y[t][pj >> 1][(pj & 1) * 2 + (k_ & 1)] = fp_relu_pack(pv[t][(k_ & 1) * 2], pv[t][(k_ & 1) * 2 + 1]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        bias = nbias;
      }
    }
#pragma unroll
    for (int t = 0; t < FP_TILES; ++t) {
      Y[t][7][2] = fp_relu_pack(acc[1][t][0], acc[1][t][1]);
      Y[t][7][3] = fp_relu_pack(acc[1][t][2], acc[1][t][3]);
    }
  };

#ifdef FP_TIMING
  unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
  int trounds = 0;
#endif
  for (; rd < total_rounds; rd += G) {
    nn = lookup(rd + 2 * (long)G);
    ent_n = load_entries(nxt);
#ifdef FP_TIMING
    ++trounds;
    FP_T(5)
#endif
    // ---- layer 0: relu(A[point of the lane] + Bd[direction of the tile]) truncated to f16, straight into the operand registers
    {
      const f4* arow = a_lds + (lane & 15) * FP_AROW_F4 + g;
      f4 av[16];
#pragma unroll
      for (int kb = 0; kb < 16; ++kb) av[kb] = arow[kb * 4];
#pragma unroll
      for (int t = 0; t < FP_TILES; ++t) {
        const f4* brow = b_lds + (wave * FP_TILES + t) * 64 + g;
        // the sixteen reads of a tile's direction row back to back, ONE wait, then the arithmetic (left to itself the compiler waits
        // for every read where it is used: sixty-four LDS latencies per round)
        f4 bv[16];
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) bv[kb] = brow[kb * 4];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) {
          // two v_pk_add_f32 (IEEE single adds, two per instruction: no matrix instruction in flight here that they would slow down)
          const fp_f2 s0 = fp_f2{av[kb][0], av[kb][1]} + fp_f2{bv[kb][0], bv[kb][1]}, s1 = fp_f2{av[kb][2], av[kb][3]} + fp_f2{bv[kb][2], bv[kb][3]};
          P[t][kb / 2][(kb & 1) * 2] = fp_relu_pack(s0[0], s0[1]);
          P[t][kb / 2][(kb & 1) * 2 + 1] = fp_relu_pack(s1[0], s1[1]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    FP_T(0)
    run_layer(P, Q, 0, false);
    FP_T(1)
    run_layer(Q, P, 16, true);
    FP_T(2)
    run_layer(P, Q, 32, false);
    FP_T(3)
    // ---- head: chunk 48 from its resident LDS copy, operands in Q
    {
      const lds_u4p hw = (lds_u4p)((unsigned)(unsigned long)((__attribute__((address_space(3))) char*)headw) + (unsigned)lane * 16u);
      f4 acc[FP_TILES];
#pragma unroll
      for (int t = 0; t < FP_TILES; ++t) acc[t] = bias;
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) {
        const u4 fh = hw[kb * 64];
#pragma unroll
        for (int t = 0; t < FP_TILES; ++t) FP_MFMA(acc[t], fh, Q[t][kb]);
      }
      bias = bias_tab[g];
      float* out = a.pair_vis + ((long)cur.item * a.LS + cur.e0 + wave * FP_TILES) * 16 + (lane & 15);
#pragma unroll
      for (int t = 0; t < FP_TILES; ++t) {
        const float l0 = acc[t][0], l1 = acc[t][1];
        if (g == 0) {
          float v;
          if (a.argmax_vis) {
            v = l1 > l0 ? 1.f : 0.f;
          } else {
            const float mx = fmaxf(l0, l1);
            const float e0 = expf(l0 - mx), e1 = expf(l1 - mx);
            v = e1 / (e0 + e1);
          }
          out[t * 16] = v;
        }
      }
    }
    cur = nxt;
    nxt = nn;
    FP_T(4)
  }
#ifdef FP_TIMING
  if ((blockIdx.x == 0 || blockIdx.x == 100) && tid == 0)
    printf("f16p wg %d rounds %d cycles/round: conv %llu L1 %llu L2 %llu L3 %llu head %llu lookup %llu\n", (int)blockIdx.x, trounds, tacc[0] / trounds,
           tacc[1] / trounds, tacc[2] / trounds, tacc[3] / trounds, tacc[4] / trounds, tacc[5] / trounds);
#endif
#undef FP_MFMA
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// one workgroup per item: per lobe and point of the block, the SG-weighted mean over the samples the point faces, in sample order
__global__ __launch_bounds__(256) void k_dvis_pb_reduce(const int* __restrict__ cid, long n, const float* __restrict__ wdir,
                                                         const float* __restrict__ wsum, const unsigned* __restrict__ entries,
                                                         const float* __restrict__ pair_vis, const PbItem* __restrict__ item_info,
                                                         const unsigned long long* __restrict__ counters, int L, int nsamp,
                                                         float* __restrict__ vis_out) {
  __shared__ int lobe_lo[256], lobe_hi[256];
  const int tid = threadIdx.x;
  const int item = blockIdx.x;
  if ((unsigned long long)item >= counters[2] || counters[3] != 0) return;   // [3]: ids not ascending -> vis_out keeps its NaN fill
  const PbItem it = item_info[item];
  const int LS = L * nsamp;
  for (int l = tid; l < L; l += 256) lobe_lo[l] = 0, lobe_hi[l] = 0;
  __syncthreads();
  const unsigned* ent = entries + (long)item * LS;
  for (int e = tid; e < it.count; e += 256) {
    const int l = (int)(ent[e] & 0xFFFFu) / nsamp;
    const int lp = e > 0 ? (int)(ent[e - 1] & 0xFFFFu) / nsamp : -1;
    const int ln = e + 1 < it.count ? (int)(ent[e + 1] & 0xFFFFu) / nsamp : -1;
    if (l != lp) lobe_lo[l] = e;
    if (l != ln) lobe_hi[l] = e + 1;
  }
  __syncthreads();
  const float* pv = pair_vis + (long)item * LS * 16;
  for (int idx = tid; idx < 16 * L; idx += 256) {
    const int m = idx & 15, l = idx >> 4;
    const long p = (long)it.p0 + m;
    if (p >= n || (cid ? cid[p] : 0) != it.chunk) continue;
    const float* w = wdir + (long)it.chunk * LS + (long)l * nsamp;
    float acc = 0.f;
    for (int e = lobe_lo[l]; e < lobe_hi[l]; ++e) {
      const unsigned en = ent[e];
      if ((en >> (16 + m)) & 1u) acc += pv[(long)e * 16 + m] * w[(int)(en & 0xFFFFu) - l * nsamp];
    }
    vis_out[p * L + l] = acc / wsum[(long)it.chunk * L + l];
  }
}

}  // namespace rb

using namespace rb;

extern "C" int rb_dvis_pblock_f16(const float* normals, const int* chunk_id, long n, const float* A, const float* Bd, const float* dirs,
                                  const float* wdir, const float* wsum, const float* W49h, int L, int nsamp, int argmax_vis,
                                  int items_max, unsigned* entries, float* pair_vis, int* round_info, int* item_info,
                                  unsigned long long* counters, int n_workgroups, float* vis_out, unsigned long long* eval_count,
                                  rb_stream_t stream) {
  if (n <= 0) return 0;
  RB_REQUIRE(normals && A && Bd && dirs && wdir && wsum && W49h && vis_out, "null pointer");
  RB_REQUIRE(entries && pair_vis && round_info && item_info && counters, "null scratch pointer");
  RB_REQUIRE(L > 0 && L <= 256 && nsamp > 0 && (long)L * nsamp <= 4096 && (L * nsamp) % 16 == 0,
             "need L <= 256, L*nsamp <= 4096 and a multiple of 16");
  const long blocks = (n + 15) / 16;
  RB_REQUIRE(items_max >= blocks && items_max <= RB_MAX_BLOCKS, "items_max: at least ceil(n / 16) (+ the number of chunks - 1), at most 2^22 - 1");
  RB_REQUIRE((long)items_max * (L * nsamp) < (1L << 31), "entry index would overflow 31 bits");
  hipStream_t s = (hipStream_t)stream;
  if (n_workgroups <= 0) n_workgroups = device_cus();
  RB_REQUIRE(n_workgroups > 0, "device query failed");
  if (hipMemsetAsync(counters, 0, 4 * sizeof(unsigned long long), s) != hipSuccess) return rb::fail(__func__, "memset failed");
  // NaN everywhere first: the reduce pass overwrites every point -- unless the chunk ids were not ascending, which must not pass for numbers
  if (hipMemsetAsync(vis_out, 0xFF, (size_t)n * L * sizeof(float), s) != hipSuccess) return rb::fail(__func__, "memset failed");
  hipLaunchKernelGGL(k_dvis_pb_cull, dim3((unsigned)blocks), dim3(256), 0, s, normals, chunk_id, n, dirs, L * nsamp, items_max, entries,
                     reinterpret_cast<PbRound*>(round_info), reinterpret_cast<PbItem*>(item_info), counters, eval_count);
  if (int rc = check_launch("k_dvis_pb_cull")) return rc;
  FpArgs a{};
  a.A = A, a.Bd = Bd, a.W = (const f4*)W49h, a.argmax_vis = argmax_vis, a.LS = L * nsamp, a.n = n;
  a.counters = counters, a.pair_vis = pair_vis;
  hipLaunchKernelGGL(k_dvis_f16p, dim3((unsigned)n_workgroups), dim3(256), 0, s, a, (const unsigned*)entries, reinterpret_cast<const PbRound*>(round_info));
  if (int rc = check_launch("k_dvis_f16p")) return rc;
  hipLaunchKernelGGL(k_dvis_pb_reduce, dim3((unsigned)items_max), dim3(256), 0, s, chunk_id, n, wdir, wsum, entries, pair_vis,
                     reinterpret_cast<const PbItem*>(item_info), counters, L, nsamp, vis_out);
  return check_launch("k_dvis_pb_reduce");
}
