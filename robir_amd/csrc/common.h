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

// ceil(n / per_block) workgroups.  A dispatch carries its grid size in work-items as a 32-bit number: workgroups x
// workgroup size must stay below 2^32, beyond that the tail of the grid silently never runs (found by the full-size
// render_neus test: 20 M points x 256 threads).  With workgroups of at most 1024 threads, 2^22 - 1 workgroups are always
// safe; a larger request comes back as an empty grid, which the launch rejects and check_launch() reports.
constexpr long RB_MAX_BLOCKS = (1L << 22) - 1;
inline dim3 grid1d(long n, int per_block) {
  const long blocks = (n + per_block - 1) / per_block;
  return dim3(blocks > RB_MAX_BLOCKS ? 0u : (unsigned)blocks);
}

}  // namespace rb

#define RB_REQUIRE(cond, msg)                     \
  do {                                            \
    if (!(cond)) return rb::fail(__func__, msg);  \
  } while (0)
