// d sdf / d (encoded input) by reverse mode with EXACT fp32 operands on the f16 matrix pipe, TWO 16-row tiles per wave -- the gradient
// pass of the default precision policy, round 4 (x6t_engine.h has the machine; sdf_back_x6.hip is round 3's one-tile kernel).
//
// In: the sigmoid tiles the value pass stored (k_sdf_x6t<5>: [tile = row / 16][layer 8][chunk 16][lane 64] float4).
// Out: two 64-wide gradient rows per point (layer 0's and the skip connection's share; k_pe_grad_points contracts them with the
// encoding's Jacobian).  The net, back to front, as one cyclic stream of 117 chunks of 16 output rows x K = 256 x 3 pieces (24 KB each:
// W3^T's 208 inputs are padded to 256 here, so every chunk of the stream has one shape):
//   stream layer   0     1     2     3             4            5     6     7
//   matrix         W7^T  W6^T  W5^T  [W4^T]        W3^T         W2^T  W1^T  W0^T
//   chunks         16    16    16    13 + 4 skip   16           16    16    4
//   gate (sigmoid) l6    l5    l4    l3 (x 1/sqrt2; skip rows: x 1/sqrt2, out)   l2  l1  l0   -- (out)
// A layer's operands are dz = dh (.) sigmoid(100 z).  The one-tile kernel loads a layer's sixteen gates per lane at its top (64
// registers); two tiles have no room for that: the gate of a (tile, chunk) -- 1 KB -- rides the weight stream as one more LDS-DMA
// piece per tile and chunk into a wave-private LDS slot (three chunks ahead, like the weights: it comes from HBM), so every chunk
// issues the same eight copies per wave and the counted waits stay exact.  Weights: packing.pack_sdf_back_x6(two_tile=True).
#include "../../include/robir_hip.h"
#include "common.h"
#include "mlp_engine.h"
#include "x6_ring.h"
#include "x6t_engine.h"
#include <type_traits>

namespace rb {

constexpr int BT_SLOT_B = 24 * 1024;
constexpr int BT_NCHUNK = 117;
__host__ __device__ constexpr int bt_nch(int l) { return l == 3 ? 17 : (l == 7 ? 4 : 16); }
__host__ __device__ constexpr int bt_cbase(int l) {
  int n = 0;
  for (int i = 0; i < l; ++i) n += bt_nch(i);
  return n;
}
__host__ __device__ constexpr long bt_coff(int c) { return (long)(c >= BT_NCHUNK ? c - BT_NCHUNK : c) * sx_cf4(256); }

__global__ __launch_bounds__(256, 1) void k_sdf_back_x6t(const f4* __restrict__ sig, long M, const f4* __restrict__ Wt,
                                                          const float* __restrict__ w8row, float* __restrict__ gfeat,
                                                          unsigned* __restrict__ range_word) {
  __shared__ f4 ring[4 * BT_SLOT_B / 16];              // 96 KB
  __shared__ f4 gate_ring[4 * 2 * 6 * 64];             // 48 KB: [wave][tile][slot 6][lane 64] float4
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long nrounds = (M + 127) >> 7;
  if ((long)blockIdx.x >= nrounds) return;

  const float negk = -2048.0f, inv_sqrt2 = 0.70710678118654752440f;
  constexpr float C11 = 1.0f / 2048.0f;
  const unsigned ring_b = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)ring);
  const unsigned gate_b = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)gate_ring) + (unsigned)wave * 12288u;
  unsigned ring_lane = ring_b + (unsigned)lane * 16u;                  // + slot + fragment offset: the fragment reads
  asm volatile("" : "+v"(ring_lane));
  const int first = xt_span_first(256, wave);                          // first 1 KB piece of this wave's span of a chunk copy
  unsigned slot_b[4] = {0u, (unsigned)BT_SLOT_B, 2u * BT_SLOT_B, 3u * BT_SLOT_B};
  unsigned gslot_b[6] = {0u, 1024u, 2048u, 3072u, 4096u, 5120u};      // gate slot of chunk jb of the current layer: gslot_b[jb % 6]
  unsigned sat = 0u;
  XtOps<8> P, Q;
  XtWin win;
  const int rlocal = wave * 16 + (lane & 15);
  long round = 0;
  __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(gfeat, 0, 0, 0x00020000);
  const f4* sig_wave = sig;            // tile `wave` of the round: + t * 4 tiles; a tile = 8 layers x 16 chunks x 64 float4
  const f4* sig_wave_next = sig;       // ... of this workgroup's next round (the last chunks of a round request its first gates)
  constexpr long TILE_F4 = 8L * 16 * 64;

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
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const f4* st = sig_wave + t * 4 * TILE_F4 + lane;
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) {
        f4 p[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int blk = 2 * kb + e;
          const f4 w = *reinterpret_cast<const f4*>(w8row + blk * 16 + 4 * g);
          const f4 s = st[(7 * 16 + blk) * 64];
          p[e] = f4{w[0] * s[0], w[1] * s[1], w[2] * s[2], w[3] * s[3]};
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) put_pair(p[q >> 1][(q & 1) * 2], p[q >> 1][(q & 1) * 2 + 1], P.h[t][kb], P.m[t][kb], P.l[t][kb], q);
      }
    }
  };
  // the gates of (stream layer, chunk) for both tiles -> this wave's gate slots; layer < 0: the next round's stream layer 0 ... which has no
  // gate of its own; none: a dummy copy (every chunk issues the same number of copies)
  // lv: the lane offset of the chunk's copies (derived by the first gate copy, reused by the second and by the weight pieces)
  auto gate_copy = [&](int t, int gate_layer, int chunk, unsigned gslot, bool next_round, unsigned& lv) {
    const f4* base = (next_round ? sig_wave_next : sig_wave) + t * 4 * TILE_F4;
#ifdef BT_ABL_GATE_FIXED              // timing ablation (wrong results): every gate copy reads the same (cache-resident) kilobyte
    const f4* src = sig;
    (void)base;
#else
    const f4* src = gate_layer >= 0 ? base + ((long)gate_layer * 16 + chunk) * 64 : base;
#endif
    if (t == 0) lv = xt_lane16<0>();
    xt_dma16_imm<0>(src, lv, gate_b + (unsigned)t * 6144u + gslot);
  };

  // gate: sigmoid layer that gates this stream layer's outputs (stream layers 0..6: 6 - l); gate_next: ... the next stream layer's
  auto run_layer = [&](auto LI_tag, int cb, int gate, int gate_next) {
    constexpr int LI = decltype(LI_tag)::value;      // 0: hidden (stream layers 0, 1, 2, 4, 5, 6), 3: W4^T (gate + skip rows), 7: W0^T (outputs)
    constexpr int K = 256, NCH = bt_nch(LI), WK = 4, NPART = 2, NFREE = 18;
    constexpr bool OUT = LI == 7, SKIPL = LI == 3;
    SxAcc acc[2], prev[2];
    const f4* wl = Wt + bt_coff(cb);
    asm volatile("" : "+s"(wl));
    auto combine = [&](const SxAcc& a, int r) { return __builtin_fmaf(__builtin_fmaf(a.c2[r], C11, a.c1[r]), C11, a.c0[r]); };
    float z[2][4];
    f4 gt[2];       // the gates of the chunk whose epilogue is running, both tiles: read with the first item -- a read placed beside its use
                    // waits with lgkmcnt(0), i.e. for every fragment request in flight
    auto item_z = [&](int t, const SxAcc& a) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        z[t][r] = combine(a, r);
        if (SKIPL) z[t][r] *= inv_sqrt2;
      }
    };
    // hidden chunk pj, pair i = (tile, register pair): gate it and split it into the next layer's operand registers
    auto read_gates = [&](unsigned gslot) {
      typedef const __attribute__((address_space(3))) f4* lds_f4p;
#pragma unroll
      for (int t = 0; t < 2; ++t) gt[t] = ((lds_f4p)(gate_b + (unsigned)t * 6144u + gslot))[lane];
    };
    auto item_b = [&](int i, int pj) {
      const int t = i >> 1, q = i & 1;
      put_pair(z[t][2 * q] * gt[t][2 * q], z[t][2 * q + 1] * gt[t][2 * q + 1], Q.h[t][pj >> 1], Q.m[t][pj >> 1], Q.l[t][pj >> 1], (pj & 1) * 2 + q);
    };
    // sixteen gradient columns of this lane's row through the round's buffer descriptor (rows >= M are dropped by its bounds check)
    auto output_tile = [&](int t, int col0) {
      int base = ((t * 64 + rlocal) * 128 + 4 * g) * 4;
      asm volatile("" : "+v"(base));
      __builtin_amdgcn_raw_buffer_store_b128(u4{__builtin_bit_cast(unsigned, z[t][0]), __builtin_bit_cast(unsigned, z[t][1]),
                                                __builtin_bit_cast(unsigned, z[t][2]), __builtin_bit_cast(unsigned, z[t][3])},
                                             out_rsrc, base, col0 * 4, 0);
    };
    // epilogue items of chunk pj: [Z(1)] then the four pairs (hidden) or the two tiles' stores (output rows); Z(0) went behind the chunk's own last run
    auto ep_item = [&](int s, int pj, unsigned gslot) {
      const bool outrow = OUT || (SKIPL && pj >= 13);
      if (s == 0) {
        if (!outrow) read_gates(gslot);
        item_z(1, prev[1]);
      } else if (outrow) {
        if (s <= 2) output_tile(s - 1, OUT ? pj * 16 : 64 + (pj - 13) * 16);
      } else {
        item_b(s - 1, pj);
      }
    };
    constexpr int NE = 5;
    // (the transposed layers carry no bias: the chunk heads of the blob are zeros, kept for the shared chunk layout)
#pragma unroll
    for (int jb = 0; jb < NCH; ++jb) {
      // chunk jb+1 (and the gates of chunk jb) have landed once at most the eight copies issued during chunk jb-1 are in flight
      sx_wait<8>();
#ifndef BT_NOBAR
      __builtin_amdgcn_s_barrier();
#endif
      asm volatile("" ::: "memory");
      const f4* src3 = wl + (long)(jb + 3) * sx_cf4(256) + 4 + first * 64;       // the stream is uniform: chunk cb + jb + 3, cyclic
      if (cb + jb + 3 >= BT_NCHUNK) src3 -= (long)BT_NCHUNK * sx_cf4(256);
      const unsigned dst3 = ring_b + slot_b[(jb + 3) & 3] + (unsigned)first * 1024u;
      // the gates of chunk jb+3's outputs (multiplied in during chunk jb+4): the same distance as the weights -- they come from HBM
      const bool nl = jb + 3 >= NCH;                       // chunk jb+3 belongs to the next stream layer
      const int g_layer = nl ? gate_next : ((OUT || (SKIPL && jb + 3 >= 13)) ? -1 : gate);
      const int g_chunk = nl ? jb + 3 - NCH : jb + 3;
      const unsigned g_slot = gslot_b[(jb + 3) % 6];
      acc[0].c0 = acc[1].c0 = acc[0].c1 = acc[0].c2 = acc[1].c1 = acc[1].c2 = f4{0.f, 0.f, 0.f, 0.f};
      const int ne = jb > 0 ? NE : 0, ni = ne + 8;
      const unsigned ep_gslot = gslot_b[(jb + 5) % 6];    // = slot of chunk jb-1
      unsigned lv = 0u;             // lane offset of this chunk's copies, carried from piece to piece (x6t_engine.h)
      auto filler = [&](int pos) {
        const int a = xt_free_index(pos);
        if (a < 0) return;
#pragma unroll
        for (int i = 0; i < 13; ++i)
          if (i < ni && xt_item_slot(i, ni, NFREE) == a) {
            if (i < ne) ep_item(i, jb - 1, ep_gslot);
            else if (i - ne < 2) gate_copy(i - ne, g_layer, g_chunk, g_slot, nl && LI == 7, lv);
            else xt_copy_piece_seq(i - ne - 2, src3, dst3, lv, true);
          }
      };
      auto refill = [&](int piece, int part) {
        const bool down = ((jb * NPART + part) & 1) != 0;
        const int slot = part == 0 ? (jb & 3) : ((jb + 1) & 3), kb_first = part == 0 ? WK : 0;
        xt_request(piece == 0 ? win.h : (piece == 1 ? win.m : win.l), ring_lane + slot_b[slot], kb_first, WK, piece, down);
      };
      xt_chunk<K, 8>(jb * NPART, acc, win, P, filler, refill);
      item_z(0, acc[0]);
      prev[1] = acc[1];
    }
    const unsigned last_gslot = gslot_b[(NCH - 1) % 6];
    {   // slot 0 = the slot of the next layer's first chunk
      constexpr int R = NCH & 3, R6 = NCH % 6;
      unsigned a[4], b[6];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = slot_b[(i + R) & 3];
#pragma unroll
      for (int i = 0; i < 6; ++i) b[i] = gslot_b[(i + R6) % 6];
#pragma unroll
      for (int i = 0; i < 4; ++i) slot_b[i] = a[i];
#pragma unroll
      for (int i = 0; i < 6; ++i) gslot_b[i] = b[i];
    }
#pragma unroll
    for (int s = 0; s < NE; ++s) ep_item(s, NCH - 1, last_gslot);
    if constexpr (!OUT) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
          if (SKIPL && kb >= 6) {      // W3^T takes 208 inputs: the slots beyond them are padding (zero weights; zero operands keep NaNs out)
            P.h[t][kb] = kb == 6 ? u4{Q.h[t][6][0], Q.h[t][6][1], 0u, 0u} : u4{0u, 0u, 0u, 0u};
            P.m[t][kb] = kb == 6 ? u4{Q.m[t][6][0], Q.m[t][6][1], 0u, 0u} : u4{0u, 0u, 0u, 0u};
            P.l[t][kb] = kb == 6 ? u4{Q.l[t][6][0], Q.l[t][6][1], 0u, 0u} : u4{0u, 0u, 0u, 0u};
          } else {
            P.h[t][kb] = Q.h[t][kb];
            P.m[t][kb] = Q.m[t][kb];
            P.l[t][kb] = Q.l[t][kb];
          }
        }
    }
  };

  // ---- prologue: chunks 0, 1, 2 of the stream, the gates of the first round's chunk 0, the first fragment window
  round = blockIdx.x;
  sig_wave = sig + (round * 8 + wave) * TILE_F4;
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int i = 0; i < 6; ++i)
      xt_copy_piece(i, Wt + bt_coff(c) + 4 + first * 64, ring_b + slot_b[c] + (unsigned)first * 1024u);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    unsigned lv0;
    gate_copy(0, 6, c, gslot_b[c], false, lv0);
    gate_copy(1, 6, c, gslot_b[c], false, lv0);
  }
  sx_wait<0>();
  __syncthreads();
  xt_request(win.h, ring_lane + slot_b[0], 0, 4, 0, true);
  xt_request(win.m, ring_lane + slot_b[0], 0, 4, 1, true);
  xt_request(win.l, ring_lane + slot_b[0], 0, 4, 2, true);

  for (; round < nrounds; round += gridDim.x) {
    {
      const long row0 = round * 128, rows = M - row0 < 128 ? M - row0 : 128;
      out_rsrc = __builtin_amdgcn_make_buffer_rsrc(gfeat + row0 * 128, 0, (int)rows * 128 * 4, 0x00020000);
      sig_wave = sig + (round * 8 + wave) * TILE_F4;
      const long nr = round + gridDim.x < nrounds ? round + gridDim.x : round;      // beyond the last round: valid dummy gates
      sig_wave_next = sig + (nr * 8 + wave) * TILE_F4;
    }
    load_layer0();
    // stream layers 0, 1, 2 | 3 (W4^T: gate + skip rows) | 4, 5, 6 (a second copy of the hidden instance) | 7 (W0^T: outputs): straight-line,
    // the register allocator has no loop-carried merge of six instances to satisfy
#pragma unroll 1
    for (int l = 0; l < 3; ++l) run_layer(std::integral_constant<int, 0>{}, 16 * l, 6 - l, 5 - l);
    run_layer(std::integral_constant<int, 3>{}, 48, 3, 2);
#pragma unroll 1
    for (int l = 4; l < 7; ++l) run_layer(std::integral_constant<int, 0>{}, l == 4 ? 65 : 81 + 16 * (l - 5), 6 - l, l == 6 ? -1 : 5 - l);
    run_layer(std::integral_constant<int, 7>{}, 113, -1, 6);
  }
  range_report(sat, range_word);
  sx_wait<0>();
  __syncthreads();
}

// host-side launcher for sdf_back.hip (rb_sdf_value_grad_x6_points)
int launch_sdf_back_x6t(const float* sig, long M, const float* Wt, const float* w8row, float* gfeat, hipStream_t s) {
  const int pg = persistent_grid((M + 127) / 128, 0);
  if (pg <= 0) return rb::fail(__func__, "device query failed");
  hipLaunchKernelGGL(k_sdf_back_x6t, dim3((unsigned)pg), dim3(256), 0, s, (const f4*)sig, M, (const f4*)Wt, w8row, gfeat,
                     range_flags() ? range_flags() + RB_RANGE_SDF : nullptr);
  return check_launch("k_sdf_back_x6t");
}

}  // namespace rb
