// Split-precision (f16x3) forms of the stand-alone MLP kernels of mlp_kernels.hip: same networks, same packed-chunk weight
// stream (an f16x3 chunk has the byte size of the fp32 chunk of the same K), layers on v_mfma_f32_16x16x32_f16 through
// dense_layer_h3 (mlp_engine.h): x*w = xh*wh + xh*wl + xl*wh with fp32 accumulation, operands lifted by a power of two
// before the hi/lo split so that both halves are normal f16 numbers.  A separate translation unit only to keep the
// build parallel.
#include "../../include/robir_hip.h"
#include "common.h"
#include "mlp_engine.h"

namespace rb {

// Same network with split-precision (f16x3) layers: Wp from rb_pack_layer_h3 (all five layers, one scale), `unscale` = 2^-s.
// FUSED: X = points p, Xd = directions d (rep per point): [PE10(p) | PE10(d)] encoded in the kernel (mlp_engine.h: load_features_vis).
template <bool FUSED>
__global__ __launch_bounds__(256, 1) void k_vis_mlp_h3(const float* __restrict__ X, long M, const f4* __restrict__ Wp,
                                                        float unscale, float* __restrict__ Y, unsigned* __restrict__ range_word,
                                                        const float* __restrict__ Xd, int rep) {
  __shared__ f4 lds[2 * chunk_f4(256)];
  __shared__ float pe_scratch[FUSED ? 4 * 2 * 16 * 128 : 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  WStream<256> ws;
  ws.init(lds, tid);
  const f4* wl0 = Wp;
  const f4* wl1 = wl0 + layer_f4<128, 256>();
  const f4* wl4 = wl1 + 3 * layer_f4<256, 256>();
  const long row0 = ((long)blockIdx.x * 4 + wave) * 32 + (lane & 15);
  float z[2][64];
  unsigned xh[2][8][4], xl[2][8][4];
  {
    float in0[2][32];
    if constexpr (FUSED) {
      load_features_vis(X, Xd, rep, row0, M, lane, pe_scratch + (wave * 2 + 0) * 2048, in0[0]);
      load_features_vis(X, Xd, rep, row0 + 16, M, lane, pe_scratch + (wave * 2 + 1) * 2048, in0[1]);
    } else {
      load_features<128>(X, row0, M, lane, in0[0]);
      load_features<128>(X, row0 + 16, M, lane, in0[1]);
    }
    unsigned ih[2][4][4], il[2][4][4];
    split_operands<128, 32, 2>(in0, ih, il);
    ws.prime<chunk_f4(128)>(wl0);
    dense_layer_h3<128, 256, 2, 256>(ws, wl0, wl1, ih, il, z, lane, 1.0f);
  }
#pragma unroll 1
  for (int l = 0; l < 3; ++l) {
    act_split<256, 2, ACT_RELU>(z, unscale, xh, xl);
    const f4* wl = wl1 + l * layer_f4<256, 256>();
    dense_layer_h3<256, 256, 2, 256>(ws, wl, wl + layer_f4<256, 256>(), xh, xl, z, lane, 1.0f);
  }
  act_split<256, 2, ACT_RELU>(z, unscale, xh, xl);
  float o[2][4];
  dense_layer_h3<256, 16, 2, 0>(ws, wl4, nullptr, xh, xl, o, lane, 1.0f);
  range_report(ws.sat, range_word);
  if ((lane >> 4) == 0) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      long row = row0 + 16 * t;
      if (row < M) {
        Y[row * 2] = o[t][0] * unscale;
        Y[row * 2 + 1] = o[t][1] * unscale;
      }
    }
  }
}

// Split-precision (f16x3) form of the same network: Wp from rb_pack_layer_h3 (K padded to multiples of 32: 64, 256, 256,
// 256, 288, 256 x4), `us` = 2^-s.  Tangent rows (forward-mode columns 1..3 of a point) are carried scaled by 2^-6 so that
// their hi halves stay inside the f16 range (PE tangents reach 2^9); the gradient is scaled back on output.
template <int MODE>
__global__ __launch_bounds__(256, 1) void k_sdf_mlp_h3(const float* __restrict__ X, long MR, const f4* __restrict__ Wp,
                                                        float us, float out_scale, float grad_scale,
                                                        float* __restrict__ out0, float* __restrict__ grad, unsigned* __restrict__ range_word) {
  constexpr bool JVP = MODE >= 2;
  constexpr bool FULL = (MODE == 1 || MODE == 3);
  constexpr int NL = FULL ? 272 : 16;
  constexpr float AS = 64.0f, TS = 0.25f;   // operand scales of value rows / tangent rows (powers of two)
  __shared__ f4 lds[2 * chunk_f4(288)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  WStream<288> ws;
  ws.init(lds, tid);
  constexpr long LF = layer_f4<256, 256>();
  const f4* w0 = Wp;
  const f4* w1 = w0 + layer_f4<64, 256>();
  const f4* w3 = w1 + 2 * LF;
  const f4* w4 = w3 + layer_f4<256, 208>();
  const f4* w5 = w4 + layer_f4<288, 256>();
  const f4* w8 = w5 + 3 * LF;
  const long row0 = ((long)blockIdx.x * 4 + wave) * 32 + (lane & 15);
  const bool is_val = JVP ? ((lane & 3) == 0) : true;
  const float asc = is_val ? AS : TS;         // scale of this lane's operands
  const float bm = is_val ? AS : 0.0f;        // bias multiplier (tangent rows carry no bias)
  const float zs = us / asc;                  // un-scaling of an MFMA result
  const float inv_sqrt2 = 0.70710678118654752440f;
  float x0[2][16], z[2][64];
  unsigned xh[2][8][4], xl[2][8][4];
  load_features<64>(X, row0, MR, lane, x0[0]);
  load_features<64>(X, row0 + 16, MR, lane, x0[1]);
  {
    unsigned ih[2][2][4], il[2][2][4];
    split_operands<64, 16, 2>(x0, ih, il, asc);
    ws.prime<chunk_f4(64)>(w0);
    dense_layer_h3<64, 256, 2, 256>(ws, w0, w1, ih, il, z, lane, bm);
  }
  softplus_into<64, 64, JVP, false, true>(z, z, lane, 1.0f, zs);
  split_operands<256, 64, 2>(z, xh, xl, asc);
#pragma unroll 1
  for (int l = 0; l < 2; ++l) {
    dense_layer_h3<256, 256, 2, 256>(ws, w1 + l * LF, w1 + (l + 1) * LF, xh, xl, z, lane, bm);
    softplus_into<64, 64, JVP, false, true>(z, z, lane, 1.0f, zs);
    split_operands<256, 64, 2>(z, xh, xl, asc);
  }
  {
    float hs[2][68];
    {
      float z3[2][52];
      dense_layer_h3<256, 208, 2, 288>(ws, w3, w4, xh, xl, z3, lane, bm);
      softplus_into<52, 68, JVP, false, true>(z3, hs, lane, inv_sqrt2, zs);   // neurons 193..207 are padding: zero weights downstream
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) hs[t][52 + i] = x0[t][i] * inv_sqrt2;
    unsigned sh[2][9][4], sl[2][9][4];
    split_operands<288, 68, 2>(hs, sh, sl, asc);
    dense_layer_h3<288, 256, 2, 256>(ws, w4, w5, sh, sl, z, lane, bm);
  }
  softplus_into<64, 64, JVP, false, true>(z, z, lane, 1.0f, zs);
  split_operands<256, 64, 2>(z, xh, xl, asc);
#pragma unroll 1
  for (int l = 0; l < 3; ++l) {
    dense_layer_h3<256, 256, 2, 256>(ws, w5 + l * LF, w5 + (l + 1) * LF, xh, xl, z, lane, bm);
    softplus_into<64, 64, JVP, false, true>(z, z, lane, 1.0f, zs);
    split_operands<256, 64, 2>(z, xh, xl, asc);
  }
  float zo[2][NL / 4];
  dense_layer_h3<256, NL, 2, 0>(ws, w8, nullptr, xh, xl, zo, lane, bm);
  range_report(ws.sat, range_word);

  const int g = lane >> 4;
  const float os = out_scale * us * (1.0f / AS), gs = grad_scale * us * (1.0f / TS);
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const long row = row0 + 16 * t;
    if (row >= MR) continue;
    if constexpr (!JVP) {
      if constexpr (FULL) {
#pragma unroll
        for (int jb = 0; jb < NL / 16; ++jb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int j = jb * 16 + 4 * g + r;
            if (j < 257) out0[row * 257 + j] = zo[t][jb * 4 + r] * os;
          }
      } else {
        if (g == 0) out0[row] = zo[t][0] * os;
      }
    } else {
      const long m = row >> 2;
      const int c = (int)(row & 3);
      if (c == 0) {
        if constexpr (FULL) {
#pragma unroll
          for (int jb = 0; jb < NL / 16; ++jb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int j = jb * 16 + 4 * g + r;
              if (j < 257) out0[m * 257 + j] = zo[t][jb * 4 + r] * os;
            }
        } else {
          if (g == 0) out0[m] = zo[t][0] * os;
        }
      } else if (g == 0) {
        grad[m * 3 + (c - 1)] = zo[t][0] * gs;
      }
    }
  }
}

// Split-precision (f16x3) form of the colour network: Wp from rb_pack_layer_h3 with k_pad 320 (same column permutation),
// 256 x3, 256; operands lifted by 2^4 before the hi/lo split.
// TWO: the 304 input columns come from two places -- columns 0..255 straight from the SDF net's output rows (feat, any row stride,
// 4-byte aligned: rows of 257 floats), columns 256..303 from a 48-float tail row [x | PE4(view) | normal | 0 x15] (rb_feat_color_tail)
// -- instead of a 1.2 KB row that rb_feat_color first copies together (19 ms of config 2 were that copy).  Same values, same order.
// TWO = 2: the tail is not read either: [x * x_scale | PE4(view) | normal | 0 x15] is ENCODED IN THE KERNEL from pxyz / pview /
// pnormal [M,3] (RenderingNetwork.forward = embedview_fn + layers, model/neus_model.py:535-560): the four lanes that share a row
// evaluate its 12 (frequency, axis) sincosf pairs between them (the calls of k_feat_color_tail: bit-identical columns) and exchange
// them through a 3 KB LDS scratch per tile.
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
template <int TWO>
__global__ __launch_bounds__(256, 1) void k_color_mlp_h3(const float* __restrict__ X, const float* __restrict__ feat, long feat_stride,
                                                          float feat_scale, long M, const f4* __restrict__ Wp, float us,
                                                          float* __restrict__ rgb, unsigned* __restrict__ range_word,
                                                          const float* __restrict__ pxyz, float x_scale,
                                                          const float* __restrict__ pview, const float* __restrict__ pnormal) {
  constexpr float AS = 16.0f;
  __shared__ f4 lds[2 * chunk_f4(320)];
  __shared__ float tail_lds[TWO == 2 ? 4 * 2 * 16 * 48 : 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  WStream<320> ws;
  ws.init(lds, tid);
  constexpr long LF = layer_f4<256, 256>();
  const f4* w0 = Wp;
  const f4* w1 = w0 + layer_f4<320, 256>();
  const f4* w4 = w1 + 3 * LF;
  const long row0 = ((long)blockIdx.x * 4 + wave) * 32 + (lane & 15);
  const float zs = us * (1.0f / AS);
  float z[2][64];
  unsigned xh[2][8][4], xl[2][8][4];
  {
    float in0[2][76];
    if constexpr (TWO != 0) {
      const int g = lane >> 4;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const long row = row0 + 16 * t;
        const bool ok = row < M;
        const float* pf = feat + (ok ? row : 0) * feat_stride + g * 4;
        const f4* pt;
        if constexpr (TWO == 2) {
          float* trow = tail_lds + ((wave * 2 + t) * 16 + (lane & 15)) * 48;
          const long rr = ok ? row : 0;
          const float v[3] = {pview[3 * rr], pview[3 * rr + 1], pview[3 * rr + 2]};
          if (g == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              trow[c] = pxyz[3 * rr + c] * x_scale;
              trow[3 + c] = v[c];
              trow[30 + c] = pnormal[3 * rr + c];
            }
          } else {
#pragma unroll
            for (int i = 0; i < 5; ++i) trow[33 + (g - 1) * 5 + i] = 0.f;
          }
#pragma unroll 1
          for (int j = g; j < 12; j += 4) {         // write_pe<4>: frequency k = j / 3, axis c = j % 3
            const int k = j / 3, c = j - 3 * k;
            float sn, cs;
            sincosf((c == 0 ? v[0] : (c == 1 ? v[1] : v[2])) * (float)(1 << k), &sn, &cs);
            trow[6 + 6 * k + c] = sn;
            trow[6 + 6 * k + 3 + c] = cs;
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // same wave, in-order LDS: all four lane groups have written
          pt = reinterpret_cast<const f4*>(trow) + g;
        } else {
          pt = reinterpret_cast<const f4*>(X + (ok ? row : 0) * 48) + g;
        }
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) {
          const f4u v = ok ? *reinterpret_cast<const f4u*>(pf + kb * 16) : f4u{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int r = 0; r < 4; ++r) in0[t][kb * 4 + r] = v[r] * feat_scale;
        }
#pragma unroll
        for (int kb = 0; kb < 3; ++kb) {
          const f4 v = ok ? pt[kb * 4] : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int r = 0; r < 4; ++r) in0[t][64 + kb * 4 + r] = v[r];
        }
      }
    } else {
      load_features<304>(X, row0, M, lane, in0[0]);
      load_features<304>(X, row0 + 16, M, lane, in0[1]);
    }
    unsigned ih[2][10][4], il[2][10][4];
    split_operands<320, 76, 2>(in0, ih, il, AS);
    ws.prime<chunk_f4(320)>(w0);
    dense_layer_h3<320, 256, 2, 256>(ws, w0, w1, ih, il, z, lane, AS);
  }
#pragma unroll 1
  for (int l = 0; l < 3; ++l) {
    act_split<256, 2, ACT_RELU>(z, zs, xh, xl, AS);
    dense_layer_h3<256, 256, 2, 256>(ws, w1 + l * LF, w1 + (l + 1) * LF, xh, xl, z, lane, AS);
  }
  act_split<256, 2, ACT_RELU>(z, zs, xh, xl, AS);
  float o[2][4];
  dense_layer_h3<256, 16, 2, 0>(ws, w4, nullptr, xh, xl, o, lane, AS);
  range_report(ws.sat, range_word);
  if ((lane >> 4) == 0) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const long row = row0 + 16 * t;
      if (row < M) {
#pragma unroll
        for (int c = 0; c < 3; ++c) rgb[row * 3 + c] = 1.0f / (1.0f + expf(-(o[t][c] * zs)));
      }
    }
  }
}

// Split-precision (f16x3) form of the 512-wide nets: Wp from rb_pack_layer_h3 (64->512, 512->512 x3, 512->NO; output rows
// padded to 16), operands lifted by 2^4 before the hi/lo split.
template <bool ENC, bool FUSED = false>
__global__ __launch_bounds__(256, 1) void k_wide_mlp_h3(const float* __restrict__ X, long M, const f4* __restrict__ Wp,
                                                         float us, float* __restrict__ Y, unsigned* __restrict__ range_word,
                                                         const float* __restrict__ extra = nullptr) {
  constexpr int NO = ENC ? 32 : 144;
  constexpr int ACT = ENC ? ACT_LEAKY02 : ACT_RELU;
  constexpr float AS = 16.0f;
  __shared__ f4 lds[2 * chunk_f4(512)];
  __shared__ float pe_scratch[FUSED ? 4 * 16 * 64 : 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  WStream<512> ws;
  ws.init(lds, tid);
  constexpr long LF = layer_f4<512, 512>();
  const f4* w0 = Wp;
  const f4* w1 = w0 + layer_f4<64, 512>();
  const f4* w4 = w1 + 3 * LF;
  const long row = ((long)blockIdx.x * 4 + wave) * 16 + (lane & 15);
  const float zs = us * (1.0f / AS);
  float z[1][128];
  unsigned xh[1][16][4], xl[1][16][4];
  {
    float in0[1][16];
    if constexpr (FUSED) {
      load_features_pe10x(X, extra, row, M, lane, pe_scratch + wave * 1024, in0[0]);
    } else {
      load_features<64>(X, row, M, lane, in0[0]);
    }
    unsigned ih[1][2][4], il[1][2][4];
    split_operands<64, 16, 1>(in0, ih, il, AS);
    ws.prime<chunk_f4(64)>(w0);
    dense_layer_h3<64, 512, 1, 512>(ws, w0, w1, ih, il, z, lane, AS);
  }
#pragma unroll 1
  for (int l = 0; l < 3; ++l) {
    act_split<512, 1, ACT>(z, zs, xh, xl, AS);
    dense_layer_h3<512, 512, 1, 512>(ws, w1 + l * LF, w1 + (l + 1) * LF, xh, xl, z, lane, AS);
  }
  act_split<512, 1, ACT>(z, zs, xh, xl, AS);
  float o[1][NO / 4];
  dense_layer_h3<512, NO, 1, 0>(ws, w4, nullptr, xh, xl, o, lane, AS);
  range_report(ws.sat, range_word);
  if (row < M) {
    const int g = lane >> 4;
#pragma unroll
    for (int jb = 0; jb < NO / 16; ++jb) {
      f4* dst = reinterpret_cast<f4*>(Y + row * NO + jb * 16) + g;
      *dst = f4{o[0][jb * 4] * zs, o[0][jb * 4 + 1] * zs, o[0][jb * 4 + 2] * zs, o[0][jb * 4 + 3] * zs};
    }
  }
}

// Split-precision (f16x3) form: Wp from rb_pack_layer_h3 (k_pad: K0P, 512, 512, 512, 544 = [N3P | K0P | 16 zero slots],
// 512 x4), operands lifted by 2^6 before the hi/lo split (softplus outputs are small), `us` = 2^-s.
// FUSED: X = the points [.,3]: the 63 encoded columns (model/embedder.py:17-38) are computed in the kernel (mlp_engine.h).
template <int K0P, int N3P, bool ONEHOT, bool FUSED = false>
__global__ __launch_bounds__(256, 1) void k_softplus512_h3(const float* __restrict__ X, long M, int n_label,
                                                            const f4* __restrict__ Wp, float us, int n_out,
                                                            float* __restrict__ Y, unsigned* __restrict__ range_word) {
  constexpr int K4 = N3P + K0P, K4P = 544;
  static_assert(K4 == 528 && K0P % 32 == 0, "both CESR nets give a 528-wide skip layer");
  constexpr float AS = 64.0f;
  __shared__ f4 lds[2 * chunk_f4(K4P)];
  __shared__ float pe_scratch[FUSED ? 4 * 16 * 64 : 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  WStream<K4P> ws;
  ws.init(lds, tid);
  constexpr long LF = layer_f4<512, 512>();
  const f4* w0 = Wp;
  const f4* w1 = w0 + layer_f4<K0P, 512>();
  const f4* w3 = w1 + 2 * LF;
  const f4* w4 = w3 + layer_f4<512, N3P>();
  const f4* w5 = w4 + layer_f4<K4P, 512>();
  const f4* w8 = w5 + 3 * LF;
  const long row = ((long)blockIdx.x * 4 + wave) * 16 + (lane & 15);
  const float inv_sqrt2 = 0.70710678118654752440f, zs = us * (1.0f / AS);
  float x0[1][K0P / 4], z[1][128];
  unsigned xh[1][16][4], xl[1][16][4];
  if constexpr (ONEHOT) {
    const bool ok = row < M;
    const long pt = ok ? row / n_label : 0;
    const int label = ok ? (int)(row % n_label) : -1;
    float enc[16];
    if constexpr (FUSED) load_features_pe10x(X, nullptr, row, M, lane, pe_scratch + wave * 1024, enc, n_label);
    const f4* p = reinterpret_cast<const f4*>(X + (FUSED ? 0 : pt * 64)) + g;
#pragma unroll
    for (int kb = 0; kb < K0P / 16; ++kb) {
      f4 v = f4{0.f, 0.f, 0.f, 0.f};
      if constexpr (FUSED) {
        if (kb < 4) v = f4{enc[kb * 4], enc[kb * 4 + 1], enc[kb * 4 + 2], enc[kb * 4 + 3]};
      } else {
        if (kb < 4 && ok) v = p[kb * 4];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = kb * 16 + 4 * g + r;
        float e = v[r];
        if (k == 63) e = 0.f;                       // column 63 of Xp is padding; the one-hot block starts here
        if (k >= 63 && k - 63 == label) e = 1.f;
        x0[0][kb * 4 + r] = e;
      }
    }
  } else if constexpr (FUSED && K0P == 64) {
    load_features_pe10x(X, nullptr, row, M, lane, pe_scratch + wave * 1024, x0[0]);
  } else {
    load_features<K0P>(X, row, M, lane, x0[0]);
  }
  {
    unsigned ih[1][K0P / 32][4], il[1][K0P / 32][4];
    split_operands<K0P, K0P / 4, 1>(x0, ih, il, AS);
    ws.template prime<chunk_f4(K0P)>(w0);
    dense_layer_h3<K0P, 512, 1, 512>(ws, w0, w1, ih, il, z, lane, AS);
  }
#pragma unroll 1
  for (int l = 0; l < 2; ++l) {
    act_split<512, 1, ACT_SOFTPLUS100_FAST>(z, zs, xh, xl, AS);
    dense_layer_h3<512, 512, 1, 512>(ws, w1 + l * LF, w1 + (l + 1) * LF, xh, xl, z, lane, AS);
  }
  act_split<512, 1, ACT_SOFTPLUS100_FAST>(z, zs, xh, xl, AS);
  {
    float hs[1][K4 / 4];
    {
      float z3[1][N3P / 4];
      dense_layer_h3<512, N3P, 1, K4P>(ws, w3, w4, xh, xl, z3, lane, AS);
#pragma unroll
      for (int i = 0; i < N3P / 4; ++i) hs[0][i] = act_fn<ACT_SOFTPLUS100_FAST>(z3[0][i] * zs) * inv_sqrt2;
    }
#pragma unroll
    for (int i = 0; i < K0P / 4; ++i) hs[0][N3P / 4 + i] = x0[0][i] * inv_sqrt2;
    unsigned sh[1][K4P / 32][4], sl[1][K4P / 32][4];
    split_operands<K4P, K4 / 4, 1>(hs, sh, sl, AS);
    dense_layer_h3<K4P, 512, 1, 512>(ws, w4, w5, sh, sl, z, lane, AS);
  }
#pragma unroll 1
  for (int l = 0; l < 3; ++l) {
    act_split<512, 1, ACT_SOFTPLUS100_FAST>(z, zs, xh, xl, AS);
    dense_layer_h3<512, 512, 1, 512>(ws, w5 + l * LF, w5 + (l + 1) * LF, xh, xl, z, lane, AS);
  }
  act_split<512, 1, ACT_SOFTPLUS100_FAST>(z, zs, xh, xl, AS);
  float o[1][4];
  dense_layer_h3<512, 16, 1, 0>(ws, w8, nullptr, xh, xl, o, lane, AS);
  range_report(ws.sat, range_word);
  if (row < M && g == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (r < n_out) Y[row * n_out + r] = o[0][r] * zs;
  }
}

}  // namespace rb

using namespace rb;

extern "C" {

int rb_vis_mlp_h3(const float* X, long M, const float* Wp, int scale_log2, float* logits, rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(X && Wp && logits, "null pointer");
  hipLaunchKernelGGL(k_vis_mlp_h3<false>, grid1d(M, 128), dim3(256), 0, (hipStream_t)stream, X, M, (const f4*)Wp,
                     ldexpf(1.0f, -scale_log2), logits, range_flags() ? range_flags() + RB_RANGE_VIS : nullptr, nullptr, 1);
  return check_launch("k_vis_mlp_h3");
}

int rb_vis_mlp_h3_points(const float* p, const float* d, long M, int rep, const float* Wp, int scale_log2, float* logits,
                         rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(p && d && Wp && logits, "null pointer");
  RB_REQUIRE(rep >= 1, "rep must be >= 1");
  hipLaunchKernelGGL(k_vis_mlp_h3<true>, grid1d(M, 128), dim3(256), 0, (hipStream_t)stream, p, M, (const f4*)Wp,
                     ldexpf(1.0f, -scale_log2), logits, range_flags() ? range_flags() + RB_RANGE_VIS : nullptr, d, rep);
  return check_launch("k_vis_mlp_h3<points>");
}

int rb_sdf_mlp_h3(const float* X, long M, const float* Wp, int mode, int scale_log2, float out_scale, float grad_scale,
                  float* out0, float* grad, rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(X && Wp && out0, "null pointer");
  RB_REQUIRE(mode >= 0 && mode <= 3, "mode must be 0..3");
  RB_REQUIRE(mode < 2 || grad, "jvp modes need a gradient output");
  const long MR = mode >= 2 ? 4 * M : M;
  dim3 grid = grid1d(MR, 128), block(256);
  hipStream_t s = (hipStream_t)stream;
  const f4* W = (const f4*)Wp;
  const float us = ldexpf(1.0f, -scale_log2);
  switch (mode) {
    case 0: hipLaunchKernelGGL(k_sdf_mlp_h3<0>, grid, block, 0, s, X, MR, W, us, out_scale, grad_scale, out0, grad, range_flags() ? range_flags() + RB_RANGE_SDF : nullptr); break;
    case 1: hipLaunchKernelGGL(k_sdf_mlp_h3<1>, grid, block, 0, s, X, MR, W, us, out_scale, grad_scale, out0, grad, range_flags() ? range_flags() + RB_RANGE_SDF : nullptr); break;
    case 2: hipLaunchKernelGGL(k_sdf_mlp_h3<2>, grid, block, 0, s, X, MR, W, us, out_scale, grad_scale, out0, grad, range_flags() ? range_flags() + RB_RANGE_SDF : nullptr); break;
    default: hipLaunchKernelGGL(k_sdf_mlp_h3<3>, grid, block, 0, s, X, MR, W, us, out_scale, grad_scale, out0, grad, range_flags() ? range_flags() + RB_RANGE_SDF : nullptr); break;
  }
  return check_launch("k_sdf_mlp_h3");
}

int rb_color_mlp_h3(const float* X, long M, const float* Wp, int scale_log2, float* rgb, rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(X && Wp && rgb, "null pointer");
  hipLaunchKernelGGL(k_color_mlp_h3<0>, grid1d(M, 128), dim3(256), 0, (hipStream_t)stream, X, nullptr, 0L, 1.0f, M, (const f4*)Wp,
                     ldexpf(1.0f, -scale_log2), rgb, range_flags() ? range_flags() + RB_RANGE_COLOR : nullptr, nullptr, 1.0f, nullptr,
                     nullptr);
  return check_launch("k_color_mlp_h3");
}

int rb_color_mlp_h3_two(const float* feat, long feat_stride, float feat_scale, const float* tail, long M, const float* Wp,
                        int scale_log2, float* rgb, rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(feat && tail && Wp && rgb, "null pointer");
  hipLaunchKernelGGL(k_color_mlp_h3<1>, grid1d(M, 128), dim3(256), 0, (hipStream_t)stream, tail, feat, feat_stride, feat_scale, M,
                     (const f4*)Wp, ldexpf(1.0f, -scale_log2), rgb, range_flags() ? range_flags() + RB_RANGE_COLOR : nullptr, nullptr, 1.0f,
                     nullptr, nullptr);
  return check_launch("k_color_mlp_h3<two>");
}

int rb_color_mlp_h3_points(const float* feat, long feat_stride, float feat_scale, const float* x, float x_scale, const float* view,
                           const float* normal, long M, const float* Wp, int scale_log2, float* rgb, rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(feat && x && view && normal && Wp && rgb, "null pointer");
  hipLaunchKernelGGL(k_color_mlp_h3<2>, grid1d(M, 128), dim3(256), 0, (hipStream_t)stream, nullptr, feat, feat_stride, feat_scale, M,
                     (const f4*)Wp, ldexpf(1.0f, -scale_log2), rgb, range_flags() ? range_flags() + RB_RANGE_COLOR : nullptr, x, x_scale,
                     view, normal);
  return check_launch("k_color_mlp_h3<points>");
}

int rb_wide_mlp_h3(const float* X, long M, const float* Wp, int encoder, int scale_log2, float* Y, rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(X && Wp && Y, "null pointer");
  const float us = ldexpf(1.0f, -scale_log2);
  if (encoder) {
    hipLaunchKernelGGL(k_wide_mlp_h3<true>, grid1d(M, 64), dim3(256), 0, (hipStream_t)stream, X, M, (const f4*)Wp, us, Y, range_flags() ? range_flags() + RB_RANGE_WIDE : nullptr);
  } else {
    hipLaunchKernelGGL(k_wide_mlp_h3<false>, grid1d(M, 64), dim3(256), 0, (hipStream_t)stream, X, M, (const f4*)Wp, us, Y, range_flags() ? range_flags() + RB_RANGE_WIDE : nullptr);
  }
  return check_launch("k_wide_mlp_h3");
}

int rb_wide_mlp_h3_points(const float* x, const float* extra, long M, const float* Wp, int encoder, int scale_log2, float* Y,
                          rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(x && Wp && Y, "null pointer");
  const float us = ldexpf(1.0f, -scale_log2);
  unsigned* rw = range_flags() ? range_flags() + RB_RANGE_WIDE : nullptr;
  if (encoder) {
    hipLaunchKernelGGL((k_wide_mlp_h3<true, true>), grid1d(M, 64), dim3(256), 0, (hipStream_t)stream, x, M, (const f4*)Wp, us, Y, rw, extra);
  } else {
    hipLaunchKernelGGL((k_wide_mlp_h3<false, true>), grid1d(M, 64), dim3(256), 0, (hipStream_t)stream, x, M, (const f4*)Wp, us, Y, rw, extra);
  }
  return check_launch("k_wide_mlp_h3<points>");
}

int rb_cesr_net_h3_points(const float* x, long M, int kind, int n_label, const float* Wp, int scale_log2, float* Y,
                          rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(x && Wp && Y, "null pointer");
  hipStream_t s = (hipStream_t)stream;
  dim3 grid = grid1d(M, 64), block(256);
  const f4* W = (const f4*)Wp;
  const float us = ldexpf(1.0f, -scale_log2);
  unsigned* rw = range_flags() ? range_flags() + RB_RANGE_SOFTPLUS512 : nullptr;
  switch (kind) {
    case 0: hipLaunchKernelGGL((k_softplus512_h3<64, 464, false, true>), grid, block, 0, s, x, M, 1, W, us, 3, Y, rw); break;
    case 2:
      RB_REQUIRE(n_label >= 1 && n_label <= 128, "n_label must be 1..128");
      hipLaunchKernelGGL((k_softplus512_h3<192, 336, true, true>), grid, block, 0, s, x, M, n_label, W, us, 2, Y, rw);
      break;
    default: return rb::fail("rb_cesr_net_h3_points", "kind: 0 normal_net on PE10(x), 2 shadow_net on (point, one-hot label) rows");
  }
  return check_launch("k_softplus512_h3<points>");
}

int rb_cesr_net_h3(const float* X, long M, int kind, int n_label, const float* Wp, int scale_log2, float* Y,
                   rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(X && Wp && Y, "null pointer");
  hipStream_t s = (hipStream_t)stream;
  dim3 grid = grid1d(M, 64), block(256);
  const f4* W = (const f4*)Wp;
  const float us = ldexpf(1.0f, -scale_log2);
  switch (kind) {
    case 0: hipLaunchKernelGGL((k_softplus512_h3<64, 464, false>), grid, block, 0, s, X, M, 1, W, us, 3, Y, range_flags() ? range_flags() + RB_RANGE_SOFTPLUS512 : nullptr); break;
    case 1: hipLaunchKernelGGL((k_softplus512_h3<192, 336, false>), grid, block, 0, s, X, M, 1, W, us, 2, Y, range_flags() ? range_flags() + RB_RANGE_SOFTPLUS512 : nullptr); break;
    case 2:
      RB_REQUIRE(n_label >= 1 && n_label <= 128, "n_label must be 1..128");
      hipLaunchKernelGGL((k_softplus512_h3<192, 336, true>), grid, block, 0, s, X, M, n_label, W, us, 2, Y, range_flags() ? range_flags() + RB_RANGE_SOFTPLUS512 : nullptr);
      break;
    default: return rb::fail("rb_cesr_net_h3", "kind: 0 normal_net, 1 shadow_net (dense rows), 2 shadow_net (point x one-hot label)");
  }
  return check_launch("k_softplus512_h3");
}

}  // extern "C"
