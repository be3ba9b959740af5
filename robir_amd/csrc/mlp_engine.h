// Wave-resident MLP engine for gfx950 (MI355X, CDNA4): exact-fp32 MFMA layers, and a split-precision (f16x3) variant.
//
// Every small MLP on RobIR's per-ray hot path (visibility, SDF, colour, indirect illumination, the sparse
// auto-encoders) is evaluated "transposed":  H_out^T = W . H_in^T  with  v_mfma_f32_16x16x4_f32
//   A operand  = weights      lane l holds W[16*jb + (l&15)][16*kb + 4*(l>>4) + r]        (r = MFMA step 0..3)
//   B operand  = activations  lane l holds h_in[sample (l&15)][16*kb + 4*(l>>4) + r]
//   C/D        = lane l, reg r holds h_out[sample (l&15)][16*jb + 4*(l>>4) + r]
// so the accumulator registers of one layer ARE the B operands of the next: activations of a tile of
// 16 samples never leave the register file between the first and the last layer (no LDS / HBM round trip),
// and the K reduction is an exact fp32 fma chain (f32-input MFMA, bitwise an fmaf loop).
//
// A wave owns NT tiles of 16 samples (NT=2 for 256-wide nets: 2 independent accumulator chains hide the
// 40-cycle dependent MFMA latency and each weight register feeds 2 MFMAs; NT=1 for 512-wide nets, where the
// two chains come from splitting K by block parity).  A workgroup is 4 waves (one per SIMD, up to 512 VGPRs
// each) that consume the same weight stream in lock step: weights are pre-packed on the host side of the C-ABI
// into "chunks" (one per block of 16 output neurons: 16 bias floats + K*16 weights, already in lane order),
// double-buffered through LDS with register staging (global -> VGPR while the previous chunk is being
// multiplied, VGPR -> LDS, one barrier per chunk).
//
// Reference arithmetic restated by the kernels built on this engine: model/neus_model.py:385-417 (SDF),
// :535-560 (colour); model/implicit_differentiable_renderer.py:199-222, 250-258; model/sg_envmap_material.py:74-99.
#pragma once
#include <hip/hip_runtime.h>

namespace rb {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int WG_THREADS = 256;  // 4 waves of 64

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_LEAKY02 = 2, ACT_SOFTPLUS100 = 3, ACT_SOFTPLUS100_FAST = 4 };

__host__ __device__ constexpr int chunk_f4(int K) { return K > 0 ? 4 + K * 4 : 0; }  // float4s per chunk

// torch.nn.Softplus(beta=100, threshold=20): x if beta*x > 20 else log1p(exp(beta*x))/beta, and its derivative
// sigmoid(beta*x) (1 above the threshold, like torch's softplus_backward).  The SDF kernels spend more time here than in
// their MFMAs when this goes through the library expf/log1pf, so it is built from the hardware transcendentals
// (v_exp/v_log/v_rcp, 1 ulp each) with the one step that needs care done exactly:
//   u = fl(1 + e), c = e - (u - 1) (the rounding error of u, exact), log1p(e) = log(u) + c/u.
// Error: a few ulp of the result (tests/test_mlp_gpu.py compares with the oracle), i.e. fp32-rounding level.
// PRECISE = library expf/log1pf and true divisions, bit-compatible with the first implementation: used where the value
// feeds a threshold decision that must reproduce the reference's (octree build), not in the shading passes.
template <bool PRECISE = false>
__device__ __forceinline__ float softplus100(float z, float* dsig) {
  const float bz = 100.0f * z;
  if constexpr (PRECISE) {
    const float ex = expf(bz);
    if (dsig) *dsig = bz > 20.0f ? 1.0f : ex / (ex + 1.0f);
    return bz > 20.0f ? z : log1pf(ex) / 100.0f;
  }
  // round 5: the overflow-free form max(z, 0) + log1p(exp(-|100 z|)) / 100 (see softplus100_stable below for why it is the accurate
  // one), here WITH the log1p correction c / u (u = 1 + e rounds e away below 2^-24: the f32-input kernels are the fp32 yardstick of
  // the test suite and keep it).  NaN propagates through exp2 / log.
  const float e = __builtin_amdgcn_exp2f(-__builtin_fabsf(bz) * 1.44269504088896340736f);     // in (0, 1]
  const float u = 1.0f + e;
  const float r = __builtin_amdgcn_rcpf(u);
  const float c = e - (u - 1.0f);
  const float lg = __builtin_amdgcn_logf(u) * 0.69314718055994530942f;                  // ln(u); v_log_f32 is log2
  if (dsig) *dsig = (z > 0.0f ? 1.0f : e) * r;
  return __builtin_fmaf(lg + c * r, 0.01f, __builtin_fmaxf(z, 0.0f));
}

// Softplus(beta=100) of the SPLIT-PRECISION kernels (22-bit operand pairs): log(1 + e) straight from the hardware log2, without the
// log1p correction above and without clamping the exponent (an overflowing exp2 gives +inf, which only ever reaches the branch
// the select discards; NaN still propagates).  6 vector ops + 2 transcendentals instead of 13 + 3.  The missing correction is an
// ABSOLUTE error <= 2^-24 ln(2) / 100 = 4e-10 on activations whose operand pairs resolve 2^-25 / 64 = 5e-10 anyway: below what
// this arithmetic carries.  The f32-input-MFMA kernels (the precision the reference has) keep softplus100<> above.
constexpr float SP_T_PER_Z = 144.26950408889634074f;      // 100 log2(e): exp(100 z) = exp2(z * this)
constexpr float SP_T_LINEAR = 28.853900817779268147f;     // 20 log2(e): beyond it softplus(z) = z (torch threshold = 20)
constexpr float SP_LN2_OVER_100 = 0.0069314718055994530942f;
__device__ __forceinline__ float softplus100_fast(float z, float* dsig) {
  const float t = z * SP_T_PER_Z;
  const float e = __builtin_amdgcn_exp2f(t);
  const float u = 1.0f + e;
  const float sp = __builtin_amdgcn_logf(u) * SP_LN2_OVER_100;        // v_log_f32 is log2
  const bool lin = t > SP_T_LINEAR;
  if (dsig) *dsig = lin ? 1.0f : e * __builtin_amdgcn_rcpf(u);
  return lin ? z : sp;
}

// Softplus(beta=100) of the EXACT-OPERAND kernels (round 5), in the overflow-free form
//     softplus(z) = max(z, 0) + log(1 + exp(-|100 z|)) / 100
// -- the same function (torch's threshold branch included: beyond 100 z = 20 the second term is < 2.1e-11 and z > 0.2, so the sum
// rounds to z itself), but the transcendental part is a CORRECTION of at most ln(2) / 100 = 0.0069 instead of the whole value: the
// few-ulp errors of v_exp_f32 / v_log_f32 and of the rounded argument (which in softplus100_fast act on log2(1 + e^{100 z}) ~ 144 z, i.e.
// ~2 ulp of every activation with 0 < 100 z < 20) stay below 1e-9 absolute, and what is left is the final addition's half ulp.  Measured
// against float64 on the geometric-initialisation net (tests/test_precision_gpu.py::test_per_net_error_budget): see DESIGN 6.  One vector
// instruction FEWER than the fast form (no compare / select: 4 + 2 transcendentals), the exponent never overflows, NaN propagates
// (exp2 -> log -> fma).  dsig = sigmoid(100 z) = (z > 0 ? 1 : e) / (1 + e), e = exp(-|100 z|).
__device__ __forceinline__ float softplus100_stable(float z, float* dsig) {
  const float e = __builtin_amdgcn_exp2f(-__builtin_fabsf(z) * SP_T_PER_Z);     // in (0, 1]
  const float u = 1.0f + e;
  const float lg = __builtin_amdgcn_logf(u);                                     // v_log_f32 is log2: in (0, 1]
  if (dsig) *dsig = (z > 0.0f ? 1.0f : e) * __builtin_amdgcn_rcpf(u);
  return __builtin_fmaf(lg, SP_LN2_OVER_100, __builtin_fmaxf(z, 0.0f));
}

template <int ACT>
__device__ __forceinline__ float act_fn(float z) {
  if constexpr (ACT == ACT_RELU) {
    return fmaxf(z, 0.0f);
  } else if constexpr (ACT == ACT_LEAKY02) {
    return z > 0.0f ? z : 0.2f * z;
  } else if constexpr (ACT == ACT_SOFTPLUS100) {
    return softplus100<false>(z, nullptr);
  } else if constexpr (ACT == ACT_SOFTPLUS100_FAST) {
    return softplus100_fast(z, nullptr);
  } else {
    return z;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Weight stream: double-buffered LDS chunks with register staging.  MAXK = largest K of any layer streamed.
// ---------------------------------------------------------------------------------------------------------
template <int MAXK>
struct WStream {
  static constexpr int BUF_F4 = chunk_f4(MAXK);
  static constexpr int NST = (BUF_F4 + WG_THREADS - 1) / WG_THREADS;
  f4* lds;  // 2 * BUF_F4 float4s
  int cur;
  int tid;
  unsigned sat;  // running max of |hi half| over every split-precision operand this thread fed to an MFMA (range sentinel)
  f4 stage[NST];

  __device__ __forceinline__ void init(f4* lds_base, int tid_) {
    lds = lds_base;
    cur = 0;
    tid = tid_;
    sat = 0u;
  }
  // Chunk copies are branch-free: a ragged tail (CF4 not a multiple of 256) is handled by clamping the element
  // index, so surplus threads re-copy the last element (same value to the same address: benign).  A per-element
  // `if (idx < CF4)` would make hipcc branch around every load and serialise them (cdna_hip_programming.md 4c).
  template <int CF4>
  __device__ __forceinline__ int elem(int i) const {
    const int idx = i * WG_THREADS + tid;
    if constexpr (CF4 % WG_THREADS == 0) {
      return idx;
    } else {
      return ((i + 1) * WG_THREADS <= CF4) ? idx : (idx < CF4 ? idx : CF4 - 1);
    }
  }
  // Synchronously place the very first chunk of a pass into buffer `cur`.
  template <int CF4>
  __device__ __forceinline__ void prime(const f4* __restrict__ src) {
#pragma unroll
    for (int i = 0; i < (CF4 + WG_THREADS - 1) / WG_THREADS; ++i) {
      const int idx = elem<CF4>(i);
      lds[cur * BUF_F4 + idx] = src[idx];
    }
    __syncthreads();
  }
  template <int CF4>
  __device__ __forceinline__ void prefetch(const f4* __restrict__ src) {
#pragma unroll
    for (int i = 0; i < (CF4 + WG_THREADS - 1) / WG_THREADS; ++i) stage[i] = src[elem<CF4>(i)];
    // keep the loads HERE (ahead of the chunk's MFMAs): without the fence hipcc sinks them next to the ds_writes of
    // commit(), exposing the full L2 latency once per chunk
    __builtin_amdgcn_sched_barrier(0);
  }
  // Store the staged chunk into the other buffer, then make it the current one (one barrier per chunk).
  template <int CF4>
  __device__ __forceinline__ void commit() {
    if constexpr (CF4 > 0) {
#pragma unroll
      for (int i = 0; i < (CF4 + WG_THREADS - 1) / WG_THREADS; ++i) lds[(cur ^ 1) * BUF_F4 + elem<CF4>(i)] = stage[i];
    }
    __syncthreads();
    cur ^= 1;
  }
  __device__ __forceinline__ const f4* chunk() const { return lds + cur * BUF_F4; }
};

// ---------------------------------------------------------------------------------------------------------
// One dense layer  out = W . in + b  for the NT tiles of this wave.
//   K, N       padded layer sizes (multiples of 16);  in: [NT][K/4] regs, out: [NT][N/4] regs (pre-activation)
//   wl         packed chunks of this layer (N/16 chunks of chunk_f4(K) float4s); chunk 0 must already be current
//   wnext      first chunk of whatever is streamed after this layer (NEXTK = its K; 0 = nothing follows)
//   bias_on    per-lane: add the bias (false for tangent columns in forward-mode differentiation)
// ---------------------------------------------------------------------------------------------------------
template <int K, int N, int NT, int NEXTK, class WS>
__device__ __forceinline__ void dense_layer(WS& ws, const f4* __restrict__ wl, const f4* __restrict__ wnext,
                                            const float (&in)[NT][K / 4], float (&out)[NT][N / 4], int lane,
                                            bool bias_on) {
  constexpr int NJB = N / 16, KB = K / 16, CF4 = chunk_f4(K), NCF4 = chunk_f4(NEXTK);
  constexpr int NCH = (NT == 1) ? 2 : 1;  // accumulator chains per tile
  const int g = lane >> 4;
#pragma unroll
  for (int jb = 0; jb < NJB; ++jb) {
    if (jb + 1 < NJB) {
      ws.template prefetch<CF4>(wl + (jb + 1) * CF4);
    } else {
      if constexpr (NCF4 > 0) ws.template prefetch<NCF4>(wnext);
    }
    const f4* cw = ws.chunk();
    f4 bias = cw[g];
    if (!bias_on) bias = f4{0.f, 0.f, 0.f, 0.f};
    f4 acc[NT][NCH];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      acc[t][0] = bias;
      if constexpr (NCH == 2) acc[t][1] = f4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const f4 w = cw[4 + kb * 64 + lane];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int c = (NCH == 2) ? (kb & 1) : 0;
          acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[r], in[t][kb * 4 + r], acc[t][c], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      f4 a = acc[t][0];
      if constexpr (NCH == 2) a = a + acc[t][1];
#pragma unroll
      for (int r = 0; r < 4; ++r) out[t][jb * 4 + r] = a[r];
    }
    if (jb + 1 < NJB) {
      ws.template commit<CF4>();
    } else {
      ws.template commit<NCF4>();
    }
  }
}

template <int N, int NT, int ACT>
__device__ __forceinline__ void activate(const float (&z)[NT][N / 4], float (&h)[NT][N / 4]) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < N / 4; ++i) h[t][i] = act_fn<ACT>(z[t][i]);
}

// ---------------------------------------------------------------------------------------------------------
// Split-precision ("f16x3") dense layer: fp32 operands are carried as hi + lo half pairs,
//   x*w ~= xh*wh + xh*wl + xl*wh        (dropped xl*wl term ~ 2^-22 relative; products exact, fp32 accumulate)
// on v_mfma_f32_16x16x32_f16 (16x the MAC rate of the f32-input MFMA -> ~5x net).  Same transposed chaining:
// the 32 k-slots of one MFMA are (lane group g, slot j) <-> neuron 32*kb + (j<4 ? 4g+j : 16+4g+j-4), i.e. the
// accumulator registers of output blocks 2kb and 2kb+1 of the previous layer, so activations again stay in
// registers (as packed half pairs).  A and B use the same (g, j) -> k map, which is all the instruction requires.
//   chunk jb = [16 bias floats, pre-scaled by 2^s] ++ [kb][hi|lo][lane][8 halves]  (same byte size as the fp32 chunk)
// Weights are pre-scaled by 2^s (power of two: exact) so that their lo parts stay clear of the f16 subnormal
// floor; the caller multiplies the result by 2^-s.
// ---------------------------------------------------------------------------------------------------------
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __fp16 h2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

// ---- activation-range sentinel ------------------------------------------------------------------------------------
// v_cvt_pkrtz rounds toward zero, so a value beyond the f16 range does not become inf: its hi half saturates at 65504
// (0x7BFF) and the pair silently loses precision.  Kernels keep a running packed maximum over the hi halves they consume
// and flag the launch (common.h: range_flags) when one is saturated or infinite.  NaN operands are NOT an overflow: they
// come from NaN inputs (rays exactly parallel to an axis give NaN points in the reference too) and stay visible as NaN
// outputs, so the tracker maps them below everything: key = (|h| + 0x03FF) as a SIGNED 16-bit number -- finite 0..0x7BFE
// -> 0x03FF..0x7FFD, saturated 0x7BFF -> 0x7FFE, inf 0x7C00 -> 0x7FFF, NaN 0x7C01.. wraps negative.
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
typedef short ss2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned sat_acc(unsigned sat, unsigned packed_hi) {
  const us2 key = __builtin_bit_cast(us2, packed_hi & 0x7fff7fffu) + us2{0x03ff, 0x03ff};
  const ss2 m = __builtin_elementwise_max(__builtin_bit_cast(ss2, sat), __builtin_bit_cast(ss2, key));
  return __builtin_bit_cast(unsigned, m);
}
// operands known to be >= 0 and never NaN (ReLU outputs: fmaxf(NaN, 0) = 0): the raw pattern as a signed 16-bit number,
// -0.0 (0x8000) sorts below everything (1 VALU op); saturated / inf = 0x7BFF / 0x7C00
__device__ __forceinline__ unsigned sat_acc_nonneg(unsigned sat, unsigned packed_hi) {
  const ss2 m = __builtin_elementwise_max(__builtin_bit_cast(ss2, sat), __builtin_bit_cast(ss2, packed_hi));
  return __builtin_bit_cast(unsigned, m);
}
// operands >= 0 that CAN be NaN (softplus outputs: a NaN point -- an axis-parallel ray -- stays NaN through the net, and the hardware
// hands it on with whatever sign the instruction sequence leaves): sat_acc's key without the AND (the sign bit is clear for every
// value that matters), so NaN of either sign, -0.0 and anything negative sort below the limit; result in sat_acc's domain (2 VALU ops)
__device__ __forceinline__ unsigned sat_acc_pos(unsigned sat, unsigned packed_hi) {
  const us2 key = __builtin_bit_cast(us2, packed_hi) + us2{0x03ff, 0x03ff};
  const ss2 m = __builtin_elementwise_max(__builtin_bit_cast(ss2, sat), __builtin_bit_cast(ss2, key));
  return __builtin_bit_cast(unsigned, m);
}
template <bool NONNEG = false>
__device__ __forceinline__ void range_report(unsigned sat, unsigned* word) {
  constexpr int lim = NONNEG ? 0x7bff : 0x7ffe;
  const int lo = (short)(sat & 0xffffu), hi = (short)(sat >> 16);
  if (word && (lo >= lim || hi >= lim)) *reinterpret_cast<volatile unsigned*>(word) = 1u;
}

// Two fp32 values -> (packed hi halves, packed lo halves) with lo = v - float(hi).  One v_cvt_pkrtz per pair:
// round-toward-zero only changes how the value is divided between hi and lo (|lo| <= 2^-10 |v|, stored to 2^-21 |v|).
__device__ __forceinline__ void split_pair(float v0, float v1, unsigned& hi, unsigned& lo) {
  const h2 h = __builtin_amdgcn_cvt_pkrtz(v0, v1);
  const h2 l = __builtin_amdgcn_cvt_pkrtz(v0 - (float)h[0], v1 - (float)h[1]);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}

// Operands are kept as 32-bit registers (two halves each): x[t][kb][q], q = 0..3 <-> half slots 2q, 2q+1.
// The layer itself is H3Ring::chunk below (software-pipelined weight ring).

// relu(z * unscale) of a whole layer (C layout, N neurons) -> packed hi/lo operands of the next layer:
// output block jb = 2kb + e, reg r  ->  k-block kb, 32-bit register 2e + r/2.
template <int N, int NT>
__device__ __forceinline__ void relu_split(const float (&z)[NT][N / 4], float unscale, unsigned (&hi)[NT][N / 32][4],
                                           unsigned (&lo)[NT][N / 32][4]) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int kb = 0; kb < N / 32; ++kb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = (2 * kb + (q >> 1)) * 4 + (q & 1) * 2;
        split_pair(fmaxf(z[t][i] * unscale, 0.f), fmaxf(z[t][i + 1] * unscale, 0.f), hi[t][kb][q], lo[t][kb][q]);
      }
}

// hi/lo split of two fp32 values in 3 VALU ops: packed round-toward-zero hi halves, then lo = f16(v - float(hi)) with the
// mixed-precision fma (f16 source read straight from the packed register, result written to one half of `lo`)
__device__ __forceinline__ void split_pair_mix(float v0, float v1, unsigned& hi, unsigned& lo) {
  const h2 h = __builtin_amdgcn_cvt_pkrtz(v0, v1);
  const unsigned hu = __builtin_bit_cast(unsigned, h);
  unsigned l;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(hu), "v"(v0));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(hu), "v"(v1));
  hi = hu;
  lo = l;
}

// fp32 activations of a layer (C layout, KV = 4 * NF valid values per lane-row, zero-padded up to K) -> packed hi/lo
// operands of the next split-precision layer: k-block kb <- output blocks 2kb, 2kb+1 (register pairs 2e + r/2).
// `scale` (a power of two, may differ per lane) lifts the values into the range where both halves are normal f16 numbers:
// an (hi, lo) pair has an absolute resolution of 2^-25, i.e. full fp32-like relative precision only for |v| >~ 0.1.
template <int K, int NF, int NT>
__device__ __forceinline__ void split_operands(const float (&h)[NT][NF], unsigned (&hi)[NT][K / 32][4],
                                               unsigned (&lo)[NT][K / 32][4], float scale = 1.0f) {
  static_assert(K % 32 == 0 && NF * 4 <= K, "K must be a multiple of 32 that covers the activations");
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int kb = 0; kb < K / 32; ++kb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = (2 * kb + (q >> 1)) * 4 + (q & 1) * 2;
        const float v0 = i < NF ? h[t][i < NF ? i : 0] : 0.f, v1 = i + 1 < NF ? h[t][i + 1 < NF ? i + 1 : 0] : 0.f;
        split_pair_mix(v0 * scale, v1 * scale, hi[t][kb][q], lo[t][kb][q]);
      }
}

// act(z * unscale) of a whole layer -> packed operands of the next one (3-op split)
template <int N, int NT, int ACT>
__device__ __forceinline__ void act_split(const float (&z)[NT][N / 4], float unscale, unsigned (&hi)[NT][N / 32][4],
                                          unsigned (&lo)[NT][N / 32][4], float scale = 1.0f) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int kb = 0; kb < N / 32; ++kb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = (2 * kb + (q >> 1)) * 4 + (q & 1) * 2;
        split_pair_mix(act_fn<ACT>(z[t][i] * unscale) * scale, act_fn<ACT>(z[t][i + 1] * unscale) * scale, hi[t][kb][q],
                       lo[t][kb][q]);
      }
}

// ---------------------------------------------------------------------------------------------------------
// Split-precision counterpart of dense_layer: same weight stream (an f16x3 chunk has the byte size of the fp32 chunk of
// the same K), operands as packed hi/lo halves, one fp32 accumulator per tile (two tiles alternate, so consecutive
// MFMAs never share an accumulator).  out = 2^s * (W.x + b): the caller un-scales (rb_pack_layer_h3 scale).
// ---------------------------------------------------------------------------------------------------------
template <int K, int N, int NT, int NEXTK, class WS>
__device__ __forceinline__ void dense_layer_h3(WS& ws, const f4* __restrict__ wl, const f4* __restrict__ wnext,
                                               const unsigned (&xh)[NT][K / 32][4], const unsigned (&xl)[NT][K / 32][4],
                                               float (&out)[NT][N / 4], int lane, float bias_mul) {
  // bias_mul: 0 for rows without bias (tangent columns), otherwise the scale the operands of this lane carry.
  // Accumulators: one chain per tile.  Consecutive MFMAs on ONE accumulator issue back to back and hide up to two other
  // instructions each; spreading a tile's three products over three accumulators (the first version, on the assumption
  // that dependent MFMAs wait on each other) costs 5-6 % here (tools/ubench/mfma_fill.hip, profiles/r01_ubench_mfma_fill.md).
  static_assert(K % 32 == 0 && N % 16 == 0 && (NT == 1 || NT == 2), "split-precision layers: K % 32 == 0, 1 or 2 tiles");
  constexpr int NJB = N / 16, KB = K / 32, CF4 = chunk_f4(K), NCF4 = chunk_f4(NEXTK);
  constexpr int CH = 1;
  const int g = lane >> 4;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int q = 0; q < 4; ++q) ws.sat = sat_acc(ws.sat, xh[t][kb][q]);
#pragma unroll
  for (int jb = 0; jb < NJB; ++jb) {
    if (jb + 1 < NJB) {
      ws.template prefetch<CF4>(wl + (jb + 1) * CF4);
    } else {
      if constexpr (NCF4 > 0) ws.template prefetch<NCF4>(wnext);
    }
    const f4* cw = ws.chunk();
    const f4 bias = cw[g] * bias_mul;
    f4 acc[NT][CH];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      acc[t][0] = bias;
#pragma unroll
      for (int c = 1; c < CH; ++c) acc[t][c] = f4{0.f, 0.f, 0.f, 0.f};
    }
    const u4* cu = reinterpret_cast<const u4*>(cw + 4) + lane;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const h8 wh = __builtin_bit_cast(h8, cu[(2 * kb) * 64]);
      const h8 wlo = __builtin_bit_cast(h8, cu[(2 * kb + 1) * 64]);
      h8 a[NT], b[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        a[t] = __builtin_bit_cast(h8, u4{xh[t][kb][0], xh[t][kb][1], xh[t][kb][2], xh[t][kb][3]});
        b[t] = __builtin_bit_cast(h8, u4{xl[t][kb][0], xl[t][kb][1], xl[t][kb][2], xl[t][kb][3]});
      }
#pragma unroll
      for (int t = 0; t < NT; ++t)
        acc[t][CH == 3 ? 1 : 0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, b[t], acc[t][CH == 3 ? 1 : 0], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, a[t], acc[t][0], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t)
        acc[t][CH == 3 ? 2 : 0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo, a[t], acc[t][CH == 3 ? 2 : 0], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      f4 r = acc[t][0];
      if constexpr (CH == 3) r = r + (acc[t][1] + acc[t][2]);
#pragma unroll
      for (int q = 0; q < 4; ++q) out[t][jb * 4 + q] = r[q];
    }
    if (jb + 1 < NJB) {
      ws.template commit<CF4>();
    } else {
      ws.template commit<NCF4>();
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Software-pipelined weight ring for the split-precision hidden stack (fused light-visibility kernel).
// A chunk = 16 output neurons x K=256 (bias + hi/lo half fragments, chunk_f4(256) float4s).  The chunk stream of the
// whole hidden stack (NCHUNK chunks, cyclic) runs through a 3-slot LDS ring:
//   chunk c's MFMAs (first half)   | ds_write chunk c+1 (global data fetched during chunk c-1) -> slot (c+1)%3
//                                  | issue global loads of chunk c+2
//   MFMAs (third quarter)          | lgkmcnt(0) + s_barrier  (chunk c+1 now visible; its slot was last read for
//                                  |                          chunk c-2, which every wave left before barrier c-1)
//   ds_read first half of c+1      | MFMAs (last quarter)    | ds_read second half of c+1
// so neither the L2 latency of the weight stream, nor the LDS write/read round trip, nor the barrier sit between the
// last MFMA of one chunk and the first MFMA of the next (PMC before: matrix pipe 37 % busy, 37 % parked in waits).
// ---------------------------------------------------------------------------------------------------------
template <int NT, int NCHUNK, bool DMA = false>
struct H3Ring {
  static constexpr int K = 256, KB = K / 32, CF4 = chunk_f4(K);
  static constexpr int NST = (CF4 + WG_THREADS - 1) / WG_THREADS;
  f4* lds;               // 3 * CF4 float4s
  const f4* W;           // NCHUNK chunks, contiguous
  int tid, lane, g;
  int c;                 // chunk whose weights sit in wreg (0..NCHUNK-1)
  int slot;              // LDS slot of chunk c
  u4 wreg[2 * KB];
  f4 bias;
  f4 stage[NST];

  __device__ __forceinline__ int elem(int i) const {
    const int idx = i * WG_THREADS + tid;
    return ((i + 1) * WG_THREADS <= CF4) ? idx : (idx < CF4 ? idx : CF4 - 1);
  }
  __device__ __forceinline__ void load_stage(int chunk) {
    const f4* src = W + (long)chunk * CF4;
#pragma unroll
    for (int i = 0; i < NST; ++i) stage[i] = src[elem(i)];
  }
  // DMA variant: global -> LDS directly (global_load_lds_dwordx4: wave-uniform LDS base + lane*16, 1 KiB per wave
  // instruction), no staging registers and no ds_write.  Completion is tracked with counted s_waitcnt vmcnt by hand.
  static constexpr int NDMA = CF4 / WG_THREADS;          // full passes (4); the 4-float4 tail is copied by wave 0
  // The copy is issued from inline asm on purpose: hipcc's waitcnt pass cannot tell which LDS bytes an LDS-DMA it knows
  // about will write, so it puts `s_waitcnt vmcnt(0)` in front of every later ds_read -- which drained the ring (the
  // chunk c+2 copy issued a few MFMAs earlier) once per chunk and exposed the full L2 latency.  Ordering against the
  // ds_reads of a slot is enforced by hand: counted vmcnt + s_barrier in chunk().
  static __device__ __forceinline__ void dma16(const f4* gsrc, f4* ldst_wave_base) {
    const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)((__attribute__((address_space(3))) char*)ldst_wave_base));
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(l), "v"(gsrc) : "memory");   // m0 is a reserved register: hipcc re-materialises it before each of its own uses
  }
  __device__ __forceinline__ void dma_chunk(int chunk, int s) {
    const f4* src = W + (long)chunk * CF4;
    f4* dst = lds + s * CF4;
    const int wave = tid >> 6;
#pragma unroll
    for (int i = 0; i < NDMA; ++i) dma16(src + i * WG_THREADS + tid, dst + i * WG_THREADS + wave * 64);
    if (tid < CF4 - NDMA * WG_THREADS) dma16(src + NDMA * WG_THREADS + tid, dst + NDMA * WG_THREADS);
  }
  __device__ __forceinline__ void store_stage(int s) {
#pragma unroll
    for (int i = 0; i < NST; ++i) lds[s * CF4 + elem(i)] = stage[i];
  }
  __device__ __forceinline__ void read_w(int s, int first, int count) {
    const u4* ch = reinterpret_cast<const u4*>(lds + s * CF4 + 4);
#pragma unroll
    for (int i = 0; i < 2 * KB; ++i)
      if (i >= first && i < first + count) wreg[i] = ch[i * 64 + lane];
  }
  // chunk 0 -> slot 0 (synchronously), chunk 1 -> stage registers, weights of chunk 0 -> wreg
  __device__ __forceinline__ void start(f4* lds_base, const f4* weights, int tid_) {
    lds = lds_base;
    W = weights;
    tid = tid_;
    lane = tid_ & 63;
    g = lane >> 4;
    c = 0;
    slot = 0;
    if constexpr (DMA) {
      dma_chunk(0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      dma_chunk(1 % NCHUNK, 1);
    } else {
      load_stage(0);
      store_stage(0);
      __syncthreads();
      load_stage(1 % NCHUNK);
    }
    bias = lds[g];
    read_w(0, 0, 2 * KB);
  }
  // One chunk: res[t] (f4, C layout of 16 output neurons) = bias + W_chunk . x
  // CH = number of accumulator chains per tile: 1 (everything into the result register), 2 (hi*hi | corrections),
  // 3 (hi*hi | hi*lo | lo*hi).  Chosen by measurement (tools/prof_dvis.py --variants).
  template <int CH>
  __device__ __forceinline__ void chunk(const unsigned (&xh)[NT][KB][4], const unsigned (&xl)[NT][KB][4], f4 (&res)[NT]) {
    f4 acc[NT][CH];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      acc[t][0] = bias;
#pragma unroll
      for (int i = 1; i < CH; ++i) acc[t][i] = f4{0.f, 0.f, 0.f, 0.f};
    }
    auto mfma_kb = [&](int kb) {
      const h8 wh = __builtin_bit_cast(h8, wreg[kb * 2]);
      const h8 wlo = __builtin_bit_cast(h8, wreg[kb * 2 + 1]);
      h8 a[NT], b[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        a[t] = __builtin_bit_cast(h8, u4{xh[t][kb][0], xh[t][kb][1], xh[t][kb][2], xh[t][kb][3]});
        b[t] = __builtin_bit_cast(h8, u4{xl[t][kb][0], xl[t][kb][1], xl[t][kb][2], xl[t][kb][3]});
      }
      // accumulator of each product: CH 1: all -> 0; 2: {0,1,1}; 3: {0,1,2}
      constexpr int i0 = 0, ia = (CH >= 2 ? 1 : 0), ib = (CH >= 3 ? 2 : (CH >= 2 ? 1 : 0));
      // issue order hi*lo, hi*hi, lo*hi: with two chains the two correction products (same accumulator) are never
      // back to back, also when a wave owns a single tile
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t][ia] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, b[t], acc[t][ia], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t][i0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, a[t], acc[t][i0], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t][ib] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo, a[t], acc[t][ib], 0, 0, 0);
    };
    const int nslot = slot == 2 ? 0 : slot + 1;
    const int c2 = (c + 2) % NCHUNK;
    // ---- phase 1: first half of the MFMAs
#pragma unroll
    for (int kb = 0; kb < KB / 2; ++kb) mfma_kb(kb);
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase 2: stage chunk c+1 into its ring slot, start fetching chunk c+2
    if constexpr (DMA) {
      dma_chunk(c2, nslot == 2 ? 0 : nslot + 1);   // slot of chunk c+2 == slot of chunk c-1: no wave still reads it
    } else {
      store_stage(nslot);
      load_stage(c2);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase 3: third quarter
#pragma unroll
    for (int kb = KB / 2; kb < 3 * KB / 4; ++kb) mfma_kb(kb);
    __builtin_amdgcn_sched_barrier(0);
    // raw barrier: only the LDS writes must have landed; the global loads of chunk c+2 stay in flight across it
    // (__syncthreads() would add s_waitcnt vmcnt(0) and expose the L2 latency every chunk)
    if constexpr (DMA) {
      // chunk c+1's DMA (issued one chunk ago) must have landed; chunk c+2's (NDMA, +1 in wave 0) may stay in flight
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase 4: first half of chunk c+1's weights (their registers are free), bias of c+1
    const f4 nbias = lds[nslot * CF4 + g];
    read_w(nslot, 0, KB);
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase 5: last quarter
#pragma unroll
    for (int kb = 3 * KB / 4; kb < KB; ++kb) mfma_kb(kb);
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase 6: second half of chunk c+1's weights; result
    read_w(nslot, KB, KB);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      f4 r = acc[t][0];
      if constexpr (CH == 2) r = r + acc[t][1];
      if constexpr (CH == 3) r = r + (acc[t][1] + acc[t][2]);
      res[t] = r;
    }
    bias = nbias;
    slot = nslot;
    c = (c + 1 == NCHUNK) ? 0 : c + 1;
  }
};

// Load the B-layout input registers of one 16-sample tile from a row-major feature matrix X[M][KP]
// (KP = padded feature count, multiple of 16).  Rows >= M read as zeros.
template <int KP>
__device__ __forceinline__ void load_features(const float* __restrict__ X, long row, long M, int lane,
                                              float (&in)[KP / 4]) {
  const int g = lane >> 4;
  const bool ok = row < M;
  const f4* p = reinterpret_cast<const f4*>(X + (ok ? row : 0) * KP) + g;
#pragma unroll
  for (int kb = 0; kb < KP / 16; ++kb) {
    f4 v = ok ? p[kb * 4] : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) in[kb * 4 + r] = v[r];
  }
}

// Positional encoding fused into the network kernels: rows are not read but ENCODED IN THE KERNEL from the points xyz[M,3] (x in_scale), cooperatively: the lanes that share a point
// (four for value rows; sixteen when a tile holds the value row and the three tangent rows of four points) evaluate its 30
// (frequency, axis) sincosf pairs between them -- the calls k_feat_pe10 (mlp_kernels.hip; model/embedder.py:17-38) fills the rows with, so the operands are bit-identical --
// and exchange them through a 4 KB LDS scratch per tile (same wave, in-order LDS: no barrier).  8 resp. 2 sincosf per lane and
// tile instead of a 256 B (1 KB with tangent rows) row per point and the encoding launch.
// the 63 encoded columns of one point written by the four lanes (g = 0..3) that share its row: frow[0..62]
__device__ __forceinline__ void pe10_coop_write(float* __restrict__ frow, const float (&a)[3], int g) {
  if (g == 0) {
    frow[0] = a[0];
    frow[1] = a[1];
    frow[2] = a[2];
  }
#pragma unroll 1
  for (int j = g; j < 30; j += 4) {                 // pair j = 3 k + c
    const int k = j / 3, c = j - 3 * k;
    float sn, cs;
    sincosf((c == 0 ? a[0] : (c == 1 ? a[1] : a[2])) * (float)(1 << k), &sn, &cs);
    frow[3 + 6 * k + c] = sn;
    frow[3 + 6 * k + 3 + c] = cs;
  }
}
// [PE10(p) | extra] rows of the 64-input nets (IndirctIllumNetwork: extra = hdr_shift; SparseAE encoders / light-visibility
// first-layer halves: extra = 0), one 16-row tile: model/implicit_differentiable_renderer.py:199-222, model/sg_envmap_material.py:188-247
__device__ __forceinline__ void load_features_pe10x(const float* __restrict__ xyz, const float* __restrict__ extra, long row, long M,
                                                    int lane, float* __restrict__ scratch, float (&in)[16], long rows_per_point = 1) {
  const int n = lane & 15, g = lane >> 4;
  const bool ok = row < M;
  const long i = ok ? row / rows_per_point : 0;      // rows_per_point > 1: consecutive rows share a point (CESR label rows)
  const float a[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
  float* frow = scratch + n * 64;
  pe10_coop_write(frow, a, g);
  if (g == 1) frow[63] = extra ? extra[i] : 0.f;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const f4* fr4 = reinterpret_cast<const f4*>(frow) + g;
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    const f4 v = fr4[kb * 4];
#pragma unroll
    for (int r = 0; r < 4; ++r) in[kb * 4 + r] = ok ? v[r] : 0.f;
  }
}
// [PE10(p) | PE10(d) | 0 0] rows of the visibility MLP (VisNetwork.forward, model/implicit_differentiable_renderer.py:250-256), one
// 16-row tile; point of row i = i / rep (rep consecutive directions per point).  scratch [16][128].
__device__ __forceinline__ void load_features_vis(const float* __restrict__ p, const float* __restrict__ d, int rep, long row, long M,
                                                  int lane, float* __restrict__ scratch, float (&in)[32]) {
  const int n = lane & 15, g = lane >> 4;
  const bool ok = row < M;
  const long i = ok ? row : 0, ip = i / rep;
  const float a[3] = {p[3 * ip], p[3 * ip + 1], p[3 * ip + 2]};
  const float b[3] = {d[3 * i], d[3 * i + 1], d[3 * i + 2]};
  float* frow = scratch + n * 128;
  pe10_coop_write(frow, a, g);
  pe10_coop_write(frow + 63, b, g);
  if (g == 2) {
    frow[126] = 0.f;
    frow[127] = 0.f;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const f4* fr4 = reinterpret_cast<const f4*>(frow) + g;
#pragma unroll
  for (int kb = 0; kb < 8; ++kb) {
    const f4 v = fr4[kb * 4];
#pragma unroll
    for (int r = 0; r < 4; ++r) in[kb * 4 + r] = ok ? v[r] : 0.f;
  }
}

template <bool JVP>
__device__ __forceinline__ void load_features_pe10(const float* __restrict__ xyz, float scale, long row, long MR, int lane,
                                                   float* __restrict__ scratch /* [16][64], private to this wave and tile */,
                                                   float (&in)[16]) {
  const int n = lane & 15, g = lane >> 4;
  const bool ok = row < MR;
  const long i = ok ? (JVP ? row >> 2 : row) : 0;
  const float a[3] = {xyz[3 * i] * scale, xyz[3 * i + 1] * scale, xyz[3 * i + 2] * scale};
  float* frow = scratch + n * 64;
  if constexpr (!JVP) {
    pe10_coop_write(frow, a, g);
    if (g == 1) frow[63] = 0.f;
  } else {
    // rows 4q .. 4q+3 of the tile = (value, d/dx, d/dy, d/dz) of point q: tangent row c is zero except in the columns of axis c
    const int tangent_of = (n & 3) - 1;
#pragma unroll
    for (int e = 0; e < 16; ++e) frow[16 * g + e] = 0.f;
    if (g == 0) {
      if (tangent_of < 0) {
        frow[0] = a[0];
        frow[1] = a[1];
        frow[2] = a[2];
      } else {
        frow[tangent_of] = 1.f;
      }
    }
    float* prow = scratch + (n & ~3) * 64;            // the value row of this lane's point; tangent row c is prow + 64 (c + 1)
#pragma unroll 1
    for (int j = (n & 3) * 4 + g; j < 30; j += 16) {
      const int k = j / 3, c = j - 3 * k;
      const float fr = (float)(1 << k);
      float sn, cs;
      sincosf((c == 0 ? a[0] : (c == 1 ? a[1] : a[2])) * fr, &sn, &cs);
      prow[3 + 6 * k + c] = sn;
      prow[3 + 6 * k + 3 + c] = cs;
      prow[64 * (c + 1) + 3 + 6 * k + c] = fr * cs;
      prow[64 * (c + 1) + 3 + 6 * k + 3 + c] = -fr * sn;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const f4* fr4 = reinterpret_cast<const f4*>(frow) + g;
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    const f4 v = fr4[kb * 4];
#pragma unroll
    for (int r = 0; r < 4; ++r) in[kb * 4 + r] = ok ? v[r] : 0.f;
  }
}

// float4s of one packed layer (N/16 chunks)
template <int K, int N>
__host__ __device__ constexpr long layer_f4() { return (long)(N / 16) * chunk_f4(K); }

// Softplus(beta=100) of a layer's pre-activations (two tiles).  JVP: rows come in groups of four (value row + three tangent
// rows of a point, lane & 3): the value row gets softplus, the tangent rows z' * sigmoid(100 z) with z read from the
// value lane.  `zs` un-scales the MFMA result first (split-precision layers), `scale` scales the output (skip: 1/sqrt 2).
template <int NREG, int HREG, bool JVP, bool PRECISE = false, bool FAST = false>
__device__ __forceinline__ void softplus_into(const float (&z)[2][NREG], float (&h)[2][HREG], int lane, float scale,
                                              float zs = 1.0f) {
  auto sp_fn = [](float v, float* sig) { if constexpr (FAST) return softplus100_fast(v, sig); else return softplus100<PRECISE>(v, sig); };
  if constexpr (JVP) {
    const bool is_val = (lane & 3) == 0;
    const int src = lane & ~3;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < NREG; ++i) {
        const float zi = z[t][i] * zs;
        const float zv = __shfl(zi, src);
        float sig;
        const float sp = sp_fn(zv, &sig);   // value row: softplus; tangent rows: z' * sigmoid(100 z)
        const float v = is_val ? sp : zi * sig;
        h[t][i] = v * scale;
      }
  } else {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < NREG; ++i) h[t][i] = sp_fn(z[t][i] * zs, nullptr) * scale;
  }
}


}  // namespace rb
