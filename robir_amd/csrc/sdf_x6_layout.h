// Stream layout of the exact-operand SDF net (packing.pack_sdf_x6; shared by sdf_x6.hip and sdf_x6t.hip): nine layers as one cyclic
// stream of 142 (all 257 outputs) / 126 (signed distance only) chunks of 16 output neurons x K x 3 pieces.
#pragma once
#include "x6_ring.h"

namespace rb {

typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int SX_SLOT_B = 27 * 1024 + 512;      // K = 288: 27 KB of fragments (+ slack: the slot's last copy may start 1 KB early)
__host__ __device__ constexpr int sx_K(int l) { return l == 0 ? 64 : (l == 4 ? 288 : 256); }
__host__ __device__ constexpr int sx_nch(int l, int last) { return l == 3 ? 13 : (l == 8 ? last : 16); }
__host__ __device__ constexpr int sx_nchunk(int last) { return 16 * 7 + 13 + last; }
__host__ __device__ constexpr int sx_cbase(int l, int last) {
  int n = 0;
  for (int i = 0; i < l; ++i) n += sx_nch(i, last);
  return n;
}
__host__ __device__ constexpr int sx_layer_of(int c, int last) {     // stream position (may run past the end once: cyclic) -> layer
  const int N = sx_nchunk(last);
  if (c >= N) c -= N;
  int l = 0, first = 0;
  for (int i = 0; i < 8; ++i) {
    first += sx_nch(i, last);
    if (c >= first) l = i + 1;
  }
  return l;
}
__host__ __device__ constexpr long sx_coff(int c, int last) {        // float4 offset of chunk c in the packed blob
  const int N = sx_nchunk(last);
  if (c >= N) c -= N;
  long off = 0;
  int first = 0, base = 0, kl = sx_K(0);
  for (int i = 0; i < 8; ++i) {
    first += sx_nch(i, last);
    if (c >= first) {
      off += (long)sx_nch(i, last) * sx_cf4(sx_K(i));
      base = first;
      kl = sx_K(i + 1);
    }
  }
  return off + (long)(c - base) * sx_cf4(kl);
}

}  // namespace rb
