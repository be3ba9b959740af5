// Shared host-side helpers for the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

namespace rb {

// Last error text of the calling thread (returned by rb_last_error()).
char* err_buf();
int fail(const char* what, const char* detail);
int check_launch(const char* kernel);
// compute units of the current device (cached per process; <= 0: the query failed) and the grid of a persistent kernel:
// min(rounds, n_workgroups), n_workgroups <= 0 meaning one workgroup per compute unit
int device_cus();
int persistent_grid(long rounds, int n_workgroups);

// ceil(n / per_block) workgroups.  A dispatch carries its grid size in work-items as a 32-bit number: workgroups x
// workgroup size must stay below 2^32, beyond that the tail of the grid silently never runs (found by the full-size
// render_neus test: 20 M points x 256 threads).  With workgroups of at most 1024 threads, 2^22 - 1 workgroups are always
// safe; a larger request comes back as an empty grid, which the launch rejects and check_launch() reports.
constexpr long RB_MAX_BLOCKS = (1L << 22) - 1;
inline dim3 grid1d(long n, int per_block) {
  const long blocks = (n + per_block - 1) / per_block;
  return dim3(blocks > RB_MAX_BLOCKS ? 0u : (unsigned)blocks);
}

// Activation-range sentinel of the split-precision (f16 hi/lo) kernels: RB_RANGE_WORDS 32-bit words in pinned, mapped host
// memory (allocated on first use; nullptr if that fails -- rb_range_check then reports the failure).  A kernel family stores
// 1 into its word when an operand's hi half saturates (|value * lift| >= 65504: beyond that the pair no longer carries
// fp32-like precision).  Words are only ever SET by kernels; the host reads and clears them in rb_range_check.
enum { RB_RANGE_DVIS = 0, RB_RANGE_VIS, RB_RANGE_SDF, RB_RANGE_COLOR, RB_RANGE_WIDE, RB_RANGE_SOFTPLUS512, RB_RANGE_WORDS = 8 };
unsigned* range_flags();

// reverse-mode SDF gradient on the f32-input MFMA: the two kernels live with the first-generation engine (mlp_kernels.hip), the
// entry point with the other gradient paths (sdf_back.hip)
int launch_sdf_f32_store(const float* xyz, long M, float in_scale, const float* Wp, float out_scale, float* out0, float* sig,
                         hipStream_t s);
int launch_sdf_back_f32(const float* sig, long M, const float* Wt, const float* w8row, float* gfeat, hipStream_t s);
int launch_sdf_back_x6(const float* sig, long M, const float* Wt, const float* w8row, float* gfeat, hipStream_t s);                       // sdf_back_x6.hip
int launch_sdf_x6_store(const float* x, long M, float in_scale, const float* Wp, float out_scale, float* out0, float* sig, hipStream_t s);   // sdf_x6.hip

int launch_color_x6t(const float* feat, long feat_stride, float feat_scale, const float* x, float x_scale, const float* view,
                     const float* normal, long M, const float* Wp, float* rgb, int n_workgroups, hipStream_t stream);                      // color_x6t.hip
int launch_sdf_back_x6t(const float* sig, long M, const float* Wt, const float* w8row, float* gfeat, hipStream_t s);                      // sdf_back_x6t.hip
int launch_sdf_x6t(const float* x, long M, float in_scale, const float* Wp, int mode, float out_scale, float* out0, float* sig,
                   int n_workgroups, hipStream_t s);                                                                                      // sdf_x6t.hip

}  // namespace rb

#define RB_REQUIRE(cond, msg)                     \
  do {                                            \
    if (!(cond)) return rb::fail(__func__, msg);  \
  } while (0)
