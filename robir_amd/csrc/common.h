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

inline dim3 grid1d(long n, int block) { return dim3((unsigned)((n + block - 1) / block)); }

}  // namespace rb

#define RB_REQUIRE(cond, msg)                     \
  do {                                            \
    if (!(cond)) return rb::fail(__func__, msg);  \
  } while (0)
