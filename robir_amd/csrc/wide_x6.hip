// SparseAE encoder / indirect-illumination decoder (64 -> 512 x4 -> 32 | 144; model/sg_envmap_material.py:74-99, 188-247,
// model/implicit_differentiable_renderer.py:199-222) with EXACT fp32 operands on the f16 matrix pipe ("f16x6") -- the default precision
// policy's kernels for the 512-wide ReLU nets, round 3.  The machine of vis_x6.hip / sdf_x6.hip (four waves, one 16-row tile each, three
// operand pieces, six MFMA products per multiply-add in three accumulators by weight class, 4-slot LDS ring, mid-unit barrier, block
// LDS-DMA) with one change of unit: an exact-operand chunk of K = 512 is 48 KB, so the stream moves HALF-chunks -- 16 output neurons x
// 256 of the 512 inputs x 3 pieces = 24 KB -- and a chunk's three accumulators run across its two halves; the activation + three-way
// split of chunk c goes between the MFMAs of chunk c+1's first half.  Units per round: 32 (layer 0, K = 64) + 3 x 64 + 2 x (2 | 9).
// Replaces k_wide_mlp (f32-input MFMA).  Weights: packing.pack_wide_x6.
#include "../../include/robir_hip.h"
#include "common.h"
#include "mlp_engine.h"
#include "x6_ring.h"
#include <type_traits>

namespace rb {

constexpr int WX_SLOT_B = 24 * 1024 + 512;
template <bool ENC>
struct WxNet {
  static constexpr int NO = ENC ? 32 : 144, ACT = ENC ? ACT_LEAKY02 : ACT_RELU;
  // layer -> units (half-chunks; layer 0: whole chunks of K = 64)
  __host__ __device__ static constexpr int nunits(int l) { return l == 0 ? 32 : (l == 4 ? 2 * (NO / 16) : 64); }
  __host__ __device__ static constexpr int total() { return 32 + 3 * 64 + 2 * (NO / 16); }
  __host__ __device__ static constexpr int ubase(int l) {
    int n = 0;
    for (int i = 0; i < l; ++i) n += nunits(i);
    return n;
  }
  __host__ __device__ static constexpr int layer_of(int u) {      // stream position (cyclic) -> layer
    if (u >= total()) u -= total();
    int l = 0, first = 0;
    for (int i = 0; i < 4; ++i) {
      first += nunits(i);
      if (u >= first) l = i + 1;
    }
    return l;
  }
  // float4 offset of the unit's chunk head (bias) + its half's offset: the copy takes bias from here and fragments from here + 4
  __host__ __device__ static constexpr long uoff(int u) {
    if (u >= total()) u -= total();
    const int l = layer_of(u), r = u - ubase(l);
    if (l == 0) return (long)r * sx_cf4(64);
    long off = 32L * sx_cf4(64);
    for (int i = 1; i < l; ++i) off += 32L * sx_cf4(512);
    return off + (long)(r >> 1) * sx_cf4(512) + (long)(r & 1) * 1536;
  }
};

// ROWS: X = feature rows [M,64] (rb_feat_pe10 / rb_feat_ipe: the auto-encoders whose embedded vector is perturbed) instead of points
template <bool ENC, bool ROWS = false>
__global__ __launch_bounds__(256, 1) void k_wide_x6(const float* __restrict__ X, const float* __restrict__ extra, long M,
                                                     const f4* __restrict__ Wp, float* __restrict__ Y, unsigned* __restrict__ range_word) {
  using Net = WxNet<ENC>;
  constexpr int NO = Net::NO;
  __shared__ f4 ring[4 * WX_SLOT_B / 16];              // 98 KB
  __shared__ f4 bias_ring[4 * 16];
  __shared__ float pe_scratch[4 * 16 * 64];            // 16 KB
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long nrounds = (M + 63) >> 6;
  if ((long)blockIdx.x >= nrounds) return;

  const float negk = -2048.0f;
  constexpr float C11 = 1.0f / 2048.0f;
  const unsigned ring_b = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)ring);
  const unsigned bias_b = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)bias_ring);
  const unsigned lane16 = (unsigned)lane * 16u, lane4 = (unsigned)lane * 4u;
  unsigned slot_b[4] = {0u, (unsigned)WX_SLOT_B, 2u * WX_SLOT_B, 3u * WX_SLOT_B};
  unsigned bslot_b[4] = {0u, 256u, 512u, 768u};
  unsigned sat = 0u;
  u4 xh[16], xm[16], xl[16];           // operands of the current layer (K = 512): three pieces, one tile
  u4 yh[16], ym[16], yl[16];           // ... of the next layer
  long rrow = 0;

  auto put_pair = [&](float v0, float v1, u4& dh, u4& dm, u4& dl, int q) {
    unsigned h, m, l;
    sx_split_pair(v0, v1, negk, h, m, l);
    dh[q] = h;
    dm[q] = m;
    dl[q] = l;
    sat = sat_acc(sat, h);
  };
  auto load_layer0 = [&]() {
    float x0[16];
    if constexpr (ROWS) load_features<64>(X, rrow, M, lane, x0);
    else load_features_pe10x(X, extra, rrow, M, lane, pe_scratch + wave * 1024, x0);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = (2 * kb + (q >> 1)) * 4 + (q & 1) * 2;
        put_pair(x0[i], x0[i + 1], xh[kb], xm[kb], xl[kb], q);
      }
  };

  // one layer: NU units of KB k-blocks; a chunk = HV consecutive units (HV = 1 for layer 0, 2 after)
  auto run_layer = [&](auto LI_tag, int ub) {
    constexpr int LI = decltype(LI_tag)::value;
    constexpr int KB = LI == 0 ? 2 : 8, HV = LI == 0 ? 1 : 2, NU = Net::nunits(LI), UB = Net::ubase(LI);
    constexpr int NPU = LI == 0 ? sx_np(64) : sx_np(256);
    constexpr bool OUT = LI == 4;
    constexpr int BS = KB >= 8 ? 2 : 1, DB = 1, D = BS * DB, NB = BS * (DB + 1);
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
      put_pair(act_fn<Net::ACT>(combine(a, 2 * q)), act_fn<Net::ACT>(combine(a, 2 * q + 1)), yh[pj >> 1], ym[pj >> 1], yl[pj >> 1], (pj & 1) * 2 + q);
    };
    auto output_chunk = [&](const SxAcc& a, int pj) {
      if (rrow < M) *(reinterpret_cast<f4*>(Y + rrow * (long)NO + pj * 16) + g) = f4{combine(a, 0), combine(a, 1), combine(a, 2), combine(a, 3)};
    };
    auto epilogue = [&](const SxAcc& a, int pj, int q) {
      if (OUT) {
        if (q == 0) output_chunk(a, pj);
      } else {
        hidden_pair(a, pj, q);
      }
    };
    zero_acc(accs[0], bias_of(0));
#pragma unroll
    for (int i = 0; i < D; ++i)
      if (i < NSTEP) {
        const u4* f = frag_of(i / KB) + (3 * (i % KB)) * 64;
        wfh[i % NB] = f[0];
        wfm[i % NB] = f[64];
        wfl[i % NB] = f[128];
      }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int c = u / HV, hv = u % HV;                // chunk of the layer, half of the chunk
      SxAcc& acc = accs[c & 1];
      if (u > 0 && hv == 0) zero_acc(acc, bnext);
      constexpr int dummy2 = 0;
      (void)dummy2;
      const int L3 = u + 3 < NU ? LI : Net::layer_of(UB + u + 3);
      const int nu3 = L3 == 0 ? sx_units(64) : sx_units(256);
      // source of unit u + 3: inside the layer the two halves of a chunk are 1536 float4 apart, chunks sx_cf4(K) apart
      const long in_layer = LI == 0 ? (long)(u + 3) * sx_cf4(64) : (long)((u + 3) >> 1) * sx_cf4(512) + (long)((u + 3) & 1) * 1536;
      const f4* src3 = u + 3 < NU ? wl + in_layer : wnext[u + 3 - NU < 3 ? u + 3 - NU : 0];
      const int sl3 = (u + 3) & 3;
      const unsigned dst3 = ring_b + slot_b[sl3], bdst3 = bias_b + bslot_b[sl3];
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const int st = u * KB + kb, xk = hv * 8 + kb;   // operand k-block of this step
        if (kb == HB) {   // unit u+1 must have landed: this wave's copies of unit u+2 may still be in flight
          const int L2 = u + 2 < NU ? LI : Net::layer_of(UB + u + 2);
          if ((L2 == 0 ? sx_np(64) : sx_np(256)) >= 7) sx_wait<7>();
          else sx_wait<3>();
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          if ((u + 1) % HV == 0) bnext = bias_of(u + 1);       // the next unit opens a chunk: its bias
        }
        if (st % BS == 0) {
#pragma unroll
          for (int i = BS - 1; i >= 0; --i) {
            const int s2 = st + D + i;
            if (s2 < NSTEP) {
              const u4* f = frag_of(s2 / KB) + (3 * (s2 % KB)) * 64;
              wfl[s2 % NB] = f[128];
              wfm[s2 % NB] = f[64];
              wfh[s2 % NB] = f[0];
            }
          }
        }
        if (st % BS == BS - 1 || kb == KB - 1) {
          const int k0 = (st % BS == BS - 1) ? (kb - (BS - 1) > 0 ? kb - (BS - 1) : 0) : kb - (st % BS);
#define WX_MFMA(ACC, W, X) ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, W), __builtin_bit_cast(h8, X), ACC, 0, 0, 0)
#pragma unroll
          for (int k = k0; k <= kb; ++k) WX_MFMA(acc.c2, wfl[(u * KB + k) % NB], xh[hv * 8 + k]);
#pragma unroll
          for (int k = k0; k <= kb; ++k) WX_MFMA(acc.c2, wfm[(u * KB + k) % NB], xm[hv * 8 + k]);
#pragma unroll
          for (int k = k0; k <= kb; ++k) WX_MFMA(acc.c2, wfh[(u * KB + k) % NB], xl[hv * 8 + k]);
#pragma unroll
          for (int k = k0; k <= kb; ++k) WX_MFMA(acc.c1, wfm[(u * KB + k) % NB], xh[hv * 8 + k]);
#pragma unroll
          for (int k = k0; k <= kb; ++k) WX_MFMA(acc.c1, wfh[(u * KB + k) % NB], xm[hv * 8 + k]);
#pragma unroll
          for (int k = k0; k <= kb; ++k) WX_MFMA(acc.c0, wfh[(u * KB + k) % NB], xh[hv * 8 + k]);
#undef WX_MFMA
        }
        (void)xk;
        if (c > 0 && hv == 0) {               // activation + three-way split (or the store) of chunk c-1
          if (kb == 0) epilogue(accs[(c - 1) & 1], c - 1, 0);
          if (kb == (KB >= 8 ? 3 : 1)) epilogue(accs[(c - 1) & 1], c - 1, 1);
        }
        if (kb >= HB) {
#pragma unroll
          for (int un = 0; un < 3; ++un)
            if (un < nu3 && (un * (KB - HB)) / nu3 == kb - HB) {
              if (L3 == 0) sx_copy_unit<64>(un, src3, lane4, lane16, bdst3, dst3, wave);
              else sx_copy_unit<256>(un, src3, lane4, lane16, bdst3, dst3, wave);
            }
        }
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
    epilogue(last, NCH - 1, 0);
    epilogue(last, NCH - 1, 1);
    if constexpr (!OUT) {
#pragma unroll
      for (int kb = 0; kb < 16; ++kb) {
        xh[kb] = yh[kb];
        xm[kb] = ym[kb];
        xl[kb] = yl[kb];
      }
    }
  };

  // ---- prologue: units 0, 1, 2 of the stream (layer 0: K = 64)
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int un = 0; un < 2; ++un)
      sx_copy_unit<64>(un, Wp + Net::uoff(c), lane4, lane16, bias_b + bslot_b[c], ring_b + slot_b[c], wave);
  sx_wait<0>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  for (long round = blockIdx.x; round < nrounds; round += gridDim.x) {
    rrow = round * 64 + wave * 16 + (lane & 15);
    load_layer0();
    // layer 0 (K = 64) | 1, 2 (one instance) | 3 (followed by the output layer's units) | 4 (followed by the next round's)
#pragma unroll 1
    for (int l = 0; l < 5; ++l) {
      if (l == 0) run_layer(std::integral_constant<int, 0>{}, 0);
      else if (l == 3) run_layer(std::integral_constant<int, 3>{}, Net::ubase(3));
      else if (l == 4) run_layer(std::integral_constant<int, 4>{}, Net::ubase(4));
      else run_layer(std::integral_constant<int, 1>{}, Net::ubase(l));
    }
  }
  range_report(sat, range_word);
  sx_wait<0>();
  __syncthreads();
}

}  // namespace rb

using namespace rb;

extern "C" int rb_wide_x6(const float* X, long M, const float* Wp, int encoder, float* Y, int n_workgroups, rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(X && Wp && Y, "null pointer");
  const int pg = persistent_grid((M + 63) / 64, n_workgroups);
  if (pg <= 0) return rb::fail(__func__, "device query failed");
  const unsigned grid = (unsigned)pg;
  unsigned* rw = range_flags() ? range_flags() + RB_RANGE_WIDE : nullptr;
  if (encoder) hipLaunchKernelGGL((k_wide_x6<true, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, X, (const float*)nullptr, M, (const f4*)Wp, Y, rw);
  else hipLaunchKernelGGL((k_wide_x6<false, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, X, (const float*)nullptr, M, (const f4*)Wp, Y, rw);
  return check_launch("k_wide_x6<rows>");
}

extern "C" int rb_wide_x6_points(const float* x, const float* extra, long M, const float* Wp, int encoder, float* Y, int n_workgroups,
                                 rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(x && Wp && Y, "null pointer");
  const int pg = persistent_grid((M + 63) / 64, n_workgroups);
  if (pg <= 0) return rb::fail(__func__, "device query failed");
  const unsigned grid = (unsigned)pg;
  unsigned* rw = range_flags() ? range_flags() + RB_RANGE_WIDE : nullptr;
  if (encoder) hipLaunchKernelGGL(k_wide_x6<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, extra, M, (const f4*)Wp, Y, rw);
  else hipLaunchKernelGGL(k_wide_x6<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, extra, M, (const f4*)Wp, Y, rw);
  return check_launch("k_wide_x6");
}
