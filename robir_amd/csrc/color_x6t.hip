// NeuS colour network (RenderingNetwork.forward, model/neus_model.py:535-560) with EXACT fp32 operands on the f16 matrix pipe, TWO
// 16-row tiles per wave -- round 4 (x6t_engine.h has the machine; color_x6.hip is round 3's one-tile kernel: the arithmetic, the blob
// (packing.pack_color_x6) and the inputs are its own -- the 256 feature columns read where the SDF network wrote them,
// [x | PE4(view) | normal] encoded in the kernel).  The net as one cyclic stream of 65 chunks (K = 320 for the first layer: five parts of
// two k-blocks; K = 256 after: two parts of four) through a 4-slot LDS ring of 30 KB slots, all biases resident in the LDS, rounds of
// 128 rows, the copies of the chunk three ahead first and the previous chunk's relu + exact three-way split behind them, the rgb rows
// through a per-round buffer descriptor.  The products of a class are summed part by part: results agree with k_color_x6 to fp32
// summation order.  The host mirror takes this kernel where it needs fewer than two thirds of the one-tile form's passes (ops.sdf_two_tile).
#include "../../include/robir_hip.h"
#include "common.h"
#include "mlp_engine.h"
#include "x6_ring.h"
#include "x6t_engine.h"
#include <type_traits>

namespace rb {

constexpr int CT_SLOT_B = 30 * 1024 + 512;
constexpr int CT_NCHUNK = 65;
__host__ __device__ constexpr int ct_K(int l) { return l == 0 ? 320 : 256; }
__host__ __device__ constexpr int ct_nch(int l) { return l == 4 ? 1 : 16; }
__host__ __device__ constexpr int ct_layer_of(int c) {      // stream position (cyclic: 65 chunks) -> layer
  if (c >= CT_NCHUNK) c -= CT_NCHUNK;
  return c >> 4;
}
__host__ __device__ constexpr long ct_coff(int c) {
  if (c >= CT_NCHUNK) c -= CT_NCHUNK;
  return c < 16 ? (long)c * sx_cf4(320) : 16 * sx_cf4(320) + (long)(c - 16) * sx_cf4(256);
}
typedef float ct_f4u __attribute__((ext_vector_type(4), aligned(4)));

__global__ __launch_bounds__(256, 1) void k_color_x6t(const float* __restrict__ feat, long feat_stride, float feat_scale,
                                                       const float* __restrict__ pxyz, float x_scale, const float* __restrict__ pview,
                                                       const float* __restrict__ pnormal, long M, const f4* __restrict__ Wp,
                                                       float* __restrict__ rgb, unsigned* __restrict__ range_word) {
  __shared__ f4 ring[4 * CT_SLOT_B / 16];              // 122 KB
  __shared__ f4 bias_tab[CT_NCHUNK * 4];               // 4 KB
  __shared__ float tail_lds[4 * 2 * 16 * 48];          // 24 KB: the encoded tail [x | PE4(view) | normal] of the round's rows
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long nrounds = (M + 127) >> 7;
  if ((long)blockIdx.x >= nrounds) return;

  const float negk = -2048.0f;
  constexpr float C11 = 1.0f / 2048.0f;
  const unsigned ring_b = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)ring);
  unsigned ring_lane = ring_b + (unsigned)lane * 16u;
  asm volatile("" : "+v"(ring_lane));
  const int first320 = xt_span_first(320, wave), first256 = xt_span_first(256, wave);
  auto span_first = [&](int K_) { return K_ == 320 ? first320 : first256; };
  unsigned slot_b[4] = {0u, (unsigned)CT_SLOT_B, 2u * CT_SLOT_B, 3u * CT_SLOT_B};
  unsigned sat = 0u;                   // range sentinel, in the domain of sat_acc_nonneg (>= 0x7bff in a half = out of range)
  XtOps<10> P;                         // operands of the current layer (K <= 320), two tiles
  XtOps<8> Q;                          // ... of the next layer
  XtWin win;
  const int rlocal = wave * 16 + (lane & 15);
  long round = 0;
  __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(rgb, 0, 0, 0x00020000);

  for (int i = tid; i < CT_NCHUNK * 4; i += 256) bias_tab[i] = Wp[ct_coff(i >> 2) + (i & 3)];

  // NONNEG: relu outputs (>= 0: the raw pattern of the h piece orders like the value, one instruction: mlp_engine.h)
  unsigned sat_in = 0u;                // ... of a round's signed inputs (sat_acc's domain): folded into `sat` at the end of the input stage
  auto put_pair = [&](float v0, float v1, u4& dh, u4& dm, u4& dl, int q, auto nonneg) {
    unsigned h, m, l;
    sx_split_pair(v0, v1, negk, h, m, l);
    dh[q] = h;
    dm[q] = m;
    dl[q] = l;
    if constexpr (decltype(nonneg)::value) sat = sat_acc_nonneg(sat, h);
    else sat_in = sat_acc(sat_in, h);
  };
  // this round's rows -> operands of layer 0: features in place (x feat_scale), tail encoded by the four lane groups of a row
  auto load_layer0 = [&]() {
    sat_in = 0u;
    // both tiles' feature rows are requested before either is split: one HBM latency per round, the second tile's under the first's
    // vector work.  Rows beyond M read row 0 (valid memory, finite values): their rgb rows are dropped by the output descriptor.
    ct_f4u fv[2][16];
    long rr[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const long rrow = round * 128 + t * 64 + rlocal;
      rr[t] = rrow < M ? rrow : 0;
      const float* pf = feat + rr[t] * feat_stride + g * 4;
#pragma unroll
      for (int kb = 0; kb < 16; ++kb) fv[t][kb] = *reinterpret_cast<const ct_f4u*>(pf + kb * 16);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float in0[80];
#pragma unroll
      for (int kb = 0; kb < 16; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) in0[kb * 4 + r] = fv[t][kb][r] * feat_scale;
      float pv[3], px[3], pn[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        pv[c] = pview[3 * rr[t] + c];
        px[c] = pxyz[3 * rr[t] + c];
        pn[c] = pnormal[3 * rr[t] + c];
      }
      float* trow = tail_lds + ((wave * 2 + t) * 16 + (lane & 15)) * 48;
      if (g == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          trow[c] = px[c] * x_scale;
          trow[3 + c] = pv[c];
          trow[30 + c] = pn[c];
        }
      } else {
#pragma unroll
        for (int i = 0; i < 5; ++i) trow[33 + (g - 1) * 5 + i] = 0.f;
      }
#pragma unroll 1
      for (int j = g; j < 12; j += 4) {         // write_pe<4>: frequency k = j / 3, axis c = j % 3
        const int k = j / 3, c = j - 3 * k;
        float sn, cs;
        sincosf((c == 0 ? pv[0] : (c == 1 ? pv[1] : pv[2])) * (float)(1 << k), &sn, &cs);
        trow[6 + 6 * k + c] = sn;
        trow[6 + 6 * k + 3 + c] = cs;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // same wave, in-order LDS: all four lane groups have written
      const f4* pt = reinterpret_cast<const f4*>(trow) + g;
#pragma unroll
      for (int kb = 0; kb < 3; ++kb) {
        const f4 v = pt[kb * 4];
#pragma unroll
        for (int r = 0; r < 4; ++r) in0[64 + kb * 4 + r] = v[r];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) in0[76 + r] = 0.f;
#pragma unroll
      for (int kb = 0; kb < 10; ++kb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = (2 * kb + (q >> 1)) * 4 + (q & 1) * 2;
          put_pair(in0[i], in0[i + 1], P.h[t][kb], P.m[t][kb], P.l[t][kb], q, std::false_type{});
        }
    }
    if ((short)(sat_in & 0xffffu) >= 0x7ffe || (short)(sat_in >> 16) >= 0x7ffe) sat = 0x7c007c00u;
  };

  auto run_layer = [&](auto LI_tag, int cb) {
    constexpr int LI = decltype(LI_tag)::value;
    constexpr int K = ct_K(LI), NCH = ct_nch(LI), CB = 16 * LI;
    constexpr int WK = xt_wk(K), NPART = xt_parts(K), NFREE = 9 * NPART;
    constexpr bool OUT = LI == 4;
    constexpr int WKN = xt_wk(ct_K(LI == 4 ? 0 : LI + 1));
    static_assert((NCH * NPART) % 2 == 0, "a layer has an even number of parts");
    SxAcc acc[2], prev[2];
    const f4* wl = Wp + ct_coff(cb);
    const f4* wnext[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) wnext[i] = Wp + ct_coff(cb + NCH + i);
    asm volatile("" : "+s"(wl));
    auto bias_of = [&](int c) {
      const int cc = c >= CT_NCHUNK ? c - CT_NCHUNK : c;
      return bias_tab[cc * 4 + g];
    };
    auto combine = [&](const SxAcc& a, int r) { return __builtin_fmaf(__builtin_fmaf(a.c2[r], C11, a.c1[r]), C11, a.c0[r]); };
    float z[2][4];
    auto item_z = [&](int t, const SxAcc& a) {
#pragma unroll
      for (int r = 0; r < 4; ++r) z[t][r] = fmaxf(combine(a, r), 0.f);
    };
    // epilogue items of hidden chunk pj: the second tile's combine + relu (the first tile's went behind the chunk's own last run), then
    // the exact three-way split of the four value pairs into the next layer's operand registers
    auto ep_item = [&](int s, int pj) {
      if (s == 0) {
        item_z(1, prev[1]);
      } else {
        const int i = s - 1, t = i >> 1, q = i & 1;
        put_pair(z[t][2 * q], z[t][2 * q + 1], Q.h[t][pj >> 1], Q.m[t][pj >> 1], Q.l[t][pj >> 1], (pj & 1) * 2 + q, std::true_type{});
      }
    };
    constexpr int NE = 5;
    f4 bias = bias_of(cb);
#pragma unroll
    for (int jb = 0; jb < NCH; ++jb) {
      // chunk jb+1 has landed once at most this wave's copies of chunk jb+2 are in flight; past the barrier every wave has finished with
      // chunk jb-1, whose slot the copies of chunk jb+3 reuse
      constexpr int dummy = 0;
      (void)dummy;
      const int K2 = jb + 2 < NCH ? K : ct_K(ct_layer_of(CB + jb + 2));
      if (sx_nsw(K2) >= 8) sx_wait<8>();
      else sx_wait<6>();
#ifndef CT_ABL_NOBAR                   // timing ablations (wrong results): no per-chunk barrier | no epilogue | no copies | no fragment reads
      __builtin_amdgcn_s_barrier();
#endif
      asm volatile("" ::: "memory");
      const int K3 = jb + 3 < NCH ? K : ct_K(ct_layer_of(CB + jb + 3));
      const int NC3 = sx_nsw(K3);
      const int f3 = span_first(K3);
      const f4* src3 = (jb + 3 < NCH ? wl + (long)(jb + 3) * sx_cf4(K) : wnext[jb + 3 - NCH < 3 ? jb + 3 - NCH : 0]) + 4 + f3 * 64;
      const unsigned dst3 = ring_b + slot_b[(jb + 3) & 3] + (unsigned)f3 * 1024u;
      f4 nbias;
      acc[0].c0 = bias;
      acc[1].c0 = bias;
      acc[0].c1 = acc[0].c2 = acc[1].c1 = acc[1].c2 = f4{0.f, 0.f, 0.f, 0.f};
#ifdef CT_ABL_NOEP
      const int ne = 0;
#else
      const int ne = (jb > 0 && !OUT) ? NE : 0;
#endif
      unsigned lv = 0u;             // lane offset of this chunk's copies, carried from piece to piece (x6t_engine.h)
      auto filler = [&](int pos) {
        if (pos == 12 * (NPART - 1)) nbias = bias_of(cb + jb + 1);       // before the last part's fragment requests (no lgkmcnt(0) at its use)
        const int a = xt_free_index(pos);
        if (a < 0) return;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#ifndef CT_ABL_NODMA
          if (i < NC3 && i == a) xt_copy_piece_seq(i, src3, dst3, lv);            // the copies first
#endif
#pragma unroll
        for (int i = 0; i < NE; ++i)
          if (i < ne && NC3 + xt_item_slot(i, NE, NFREE - NC3) == a) ep_item(i, jb > 0 ? jb - 1 : 0);
      };
      auto refill = [&](int piece, int part) {
        const bool down = ((jb * NPART + part) & 1) != 0;
        int slot, kb_first, count;
        if (part + 1 < NPART) {
          slot = jb & 3, kb_first = (part + 1) * WK, count = WK;
        } else {
          slot = (jb + 1) & 3, kb_first = 0, count = jb + 1 < NCH ? WK : WKN;
        }
#ifdef CT_ABL_NOREAD
        {
          u4(&d)[4] = piece == 0 ? win.h : (piece == 1 ? win.m : win.l);
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (k < count) asm volatile("" : "+v"(d[k]));
          (void)slot; (void)kb_first; (void)down;
        }
#else
        xt_request(piece == 0 ? win.h : (piece == 1 ? win.m : win.l), ring_lane + slot_b[slot], kb_first, count, piece, down);
#endif
      };
      xt_chunk<K, 10>(jb * NPART, acc, win, P, filler, refill);
      if constexpr (OUT) prev[0] = acc[0];
      else item_z(0, acc[0]);
      prev[1] = acc[1];
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
    if constexpr (OUT) {
      // rgb = sigmoid of the three real outputs: lanes g = 0 hold them; through the round's descriptor (rows >= M dropped)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        int base = g == 0 ? (t * 64 + rlocal) * 12 : -1;
        asm volatile("" : "+v"(base));
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float v = 1.0f / (1.0f + expf(-combine(prev[t], c)));
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), out_rsrc, base, c * 4, 0);
        }
      }
    } else {
#pragma unroll
      for (int s = 0; s < NE; ++s) ep_item(s, NCH - 1);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
          P.h[t][kb] = Q.h[t][kb];
          P.m[t][kb] = Q.m[t][kb];
          P.l[t][kb] = Q.l[t][kb];
        }
    }
  };

  // ---- prologue: chunks 0, 1, 2 of the stream (layer 0: K = 320, eight pieces per wave), the first fragment window
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int i = 0; i < 8; ++i)
      xt_copy_piece(i, Wp + ct_coff(c) + 4 + first320 * 64, ring_b + slot_b[c] + (unsigned)first320 * 1024u);
  sx_wait<0>();
  __syncthreads();
  xt_request(win.h, ring_lane + slot_b[0], 0, 2, 0, true);
  xt_request(win.m, ring_lane + slot_b[0], 0, 2, 1, true);
  xt_request(win.l, ring_lane + slot_b[0], 0, 2, 2, true);

  for (round = blockIdx.x; round < nrounds; round += gridDim.x) {
    {
      const long row0 = round * 128, rows = M - row0 < 128 ? M - row0 : 128;
      out_rsrc = __builtin_amdgcn_make_buffer_rsrc(rgb + row0 * 3, 0, (int)rows * 12, 0x00020000);
    }
#ifdef CT_ABL_LOAD_ONCE               // timing ablation (wrong results): the input stage only in a workgroup's first round
    if (round == (long)blockIdx.x)
#endif
    load_layer0();
    // layer 0 (K = 320) | 1, 2 (one instance) | 3 (followed by the output chunk and the next round's first chunks) | 4: straight-line
    run_layer(std::integral_constant<int, 0>{}, 0);
#pragma unroll 1
    for (int l = 1; l < 3; ++l) run_layer(std::integral_constant<int, 1>{}, 16 * l);
    run_layer(std::integral_constant<int, 3>{}, 48);
    run_layer(std::integral_constant<int, 4>{}, 64);
  }
  range_report<true>(sat, range_word);
  sx_wait<0>();
  __syncthreads();
}

}  // namespace rb

namespace rb {
// the two-tile form of rb_color_x6_points (color_x6.hip holds the entry point; arguments checked there)
int launch_color_x6t(const float* feat, long feat_stride, float feat_scale, const float* x, float x_scale, const float* view,
                     const float* normal, long M, const float* Wp, float* rgb, int n_workgroups, hipStream_t stream) {
  const int pg = persistent_grid((M + 127) / 128, n_workgroups);
  if (pg <= 0) return rb::fail("rb_color_x6_points", "device query failed");
  hipLaunchKernelGGL(k_color_x6t, dim3((unsigned)pg), dim3(256), 0, stream, feat, feat_stride, feat_scale, x, x_scale, view,
                     normal, M, (const f4*)Wp, rgb, range_flags() ? range_flags() + RB_RANGE_COLOR : nullptr);
  return check_launch("k_color_x6t");
}
}  // namespace rb
