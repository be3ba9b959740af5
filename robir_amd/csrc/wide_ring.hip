// C entry points of the 512-wide chunk-stream kernels (wide_ring.h; instances in wide_ring_{normal,shadow,encoder,decoder}.hip).
#include "wide_ring.h"

using namespace rb;

namespace {
int wr_grid(long M, int n_workgroups) { return persistent_grid((M + 63) / 64, n_workgroups); }
}  // namespace

extern "C" {

int rb_cesr_net_ring_points(const float* x, long M, int kind, int n_label, const float* Wp, int scale_log2, float* Y, int n_workgroups,
                            rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(x && Wp && Y, "null pointer");
  const int grid = wr_grid(M, n_workgroups);
  if (grid <= 0) return rb::fail(__func__, "device query failed");
  const float us = ldexpf(1.0f, -scale_log2);
  unsigned* rw = range_flags() ? range_flags() + RB_RANGE_SOFTPLUS512 : nullptr;
  switch (kind) {
    case 0: return launch_cesr_ring_normal(x, M, (const f4*)Wp, us, Y, rw, grid, (hipStream_t)stream);
    case 2:
      RB_REQUIRE(n_label >= 1 && n_label <= 128, "n_label must be 1..128");
      return launch_cesr_ring_shadow(x, M, n_label, (const f4*)Wp, us, Y, rw, grid, (hipStream_t)stream);
    default: return rb::fail(__func__, "kind: 0 normal_net on PE10(x), 2 shadow_net on (point, one-hot label) rows");
  }
}

int rb_wide_mlp_ring_points(const float* x, const float* extra, long M, const float* Wp, int encoder, int scale_log2, float* Y,
                            int n_workgroups, rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(x && Wp && Y, "null pointer");
  const int grid = wr_grid(M, n_workgroups);
  if (grid <= 0) return rb::fail(__func__, "device query failed");
  const float us = ldexpf(1.0f, -scale_log2);
  unsigned* rw = range_flags() ? range_flags() + RB_RANGE_WIDE : nullptr;
  return encoder ? launch_wide_ring_encoder(x, extra, M, (const f4*)Wp, us, Y, rw, grid, (hipStream_t)stream)
                 : launch_wide_ring_decoder(x, extra, M, (const f4*)Wp, us, Y, rw, grid, (hipStream_t)stream);
}

int rb_wide_mlp_ring(const float* X, long M, const float* Wp, int encoder, int scale_log2, float* Y, int n_workgroups, rb_stream_t stream) {
  if (M <= 0) return 0;
  RB_REQUIRE(X && Wp && Y, "null pointer");
  const int grid = wr_grid(M, n_workgroups);
  if (grid <= 0) return rb::fail(__func__, "device query failed");
  const float us = ldexpf(1.0f, -scale_log2);
  unsigned* rw = range_flags() ? range_flags() + RB_RANGE_WIDE : nullptr;
  return encoder ? launch_wide_ring_encoder_rows(X, M, (const f4*)Wp, us, Y, rw, grid, (hipStream_t)stream)
                 : launch_wide_ring_decoder_rows(X, M, (const f4*)Wp, us, Y, rw, grid, (hipStream_t)stream);
}

}  // extern "C"
