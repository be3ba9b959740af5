// Throughput of the LDS-DMA weight stream per compute unit: every workgroup (256 threads, one per CU) copies `chunk` bytes per
// iteration from a 2 MB L2-resident buffer into an LDS ring with global_load_lds_dwordx4 / _dword, `depth` chunks in flight,
// one s_barrier per iteration -- the skeleton of the ring kernels without any arithmetic.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_dma_bw.hip -o /tmp/lds_dma_bw && /tmp/lds_dma_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int WIDE>
__device__ __forceinline__ void dma(const void* gbase, unsigned lane_off, unsigned lds) {
  if constexpr (WIDE)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(lane_off), "s"(gbase) : "memory");
  else
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(lds), "v"(lane_off), "s"(gbase) : "memory");
}

// ROWS 1 KB (wide) or 256 B (narrow) rows per wave and iteration; DEPTH iterations in flight
template <int WIDE, int ROWS, int DEPTH>
__global__ __launch_bounds__(256, 1) void k_dma(const char* __restrict__ src, long src_bytes, int iters, float* sink, int same,
                                                 int skew) {
  __shared__ f4 ring[8192];   // 128 KB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned ring_b = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)ring);
  constexpr unsigned ROWB = WIDE ? 1024u : 256u;
  constexpr unsigned ITB = 4u * ROWS * ROWB;           // bytes per iteration and workgroup
  const unsigned lane_off = (unsigned)lane * (WIDE ? 16u : 4u);
  long off = ((long)blockIdx.x * 7919L * ITB) % (src_bytes - (long)ITB * 4);
  off &= ~1023L;
  if (same) off = 0;                                   // every workgroup streams the same addresses (a shared weight blob)
  for (int i = 0; i < skew * (int)((blockIdx.x / 8) % 32); ++i) __builtin_amdgcn_s_sleep(16);   // stagger the workgroups of an XCD
  auto issue = [&](int it) {
    const char* base = src + (off + (long)it * ITB) % (src_bytes - (long)ITB);
    const unsigned slot = (unsigned)(it % (DEPTH + 1)) * ITB;
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
      dma<WIDE>(base + (wave * ROWS + r) * ROWB, lane_off, ring_b + slot + (unsigned)(wave * ROWS + r) * ROWB);
  };
  for (int it = 0; it < DEPTH; ++it) issue(it);
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    issue(it + DEPTH);
    // everything but the DEPTH youngest iterations has landed
    if constexpr (ROWS * DEPTH == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (ROWS * DEPTH == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (ROWS * DEPTH == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (ROWS * DEPTH == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (ROWS * DEPTH == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    else if constexpr (ROWS * DEPTH == 48) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    acc += ring[(it % (DEPTH + 1)) * (ITB / 16) + tid][0];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc == 123.456f) sink[0] = acc;
}

template <int WIDE, int ROWS, int DEPTH>
void run(const char* src, long bytes, float* sink, int grid, const char* label, int same = 0, int skew = 0) {
  const int iters = 20000;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL((k_dma<WIDE, ROWS, DEPTH>), dim3(grid), dim3(256), 0, 0, src, bytes, 2000, sink, same, skew);
  hipEventRecord(a);
  hipLaunchKernelGGL((k_dma<WIDE, ROWS, DEPTH>), dim3(grid), dim3(256), 0, 0, src, bytes, iters, sink, same, skew);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double per_it = 4.0 * ROWS * (WIDE ? 1024 : 256);
  printf("%-44s same %d skew %2d grid %3d: %.3f us per iteration, %.1f GB/s per workgroup, %.2f TB/s aggregate\n", label, same, skew, grid, ms * 1e3 / iters,
         per_it * iters / (ms * 1e-3) / 1e9, per_it * iters * grid / (ms * 1e-3) / 1e12);
}

int main() {
  const long bytes = 2 << 20;
  char* src;
  float* sink;
  hipMalloc(&src, bytes);
  hipMalloc(&sink, 4);
  hipMemset(src, 1, bytes);
  for (int grid : {1, 32, 256}) {
    run<1, 4, 3>(src, bytes, sink, grid, "dwordx4, 16 KB per iteration, 3 in flight");
    run<1, 4, 8>(src, bytes, sink, grid, "dwordx4, 16 KB per iteration, 8 in flight");
    run<1, 2, 4>(src, bytes, sink, grid, "dwordx4, 8 KB per iteration, 4 in flight");
    run<1, 8, 2>(src, bytes, sink, grid, "dwordx4, 32 KB per iteration, 2 in flight");
    run<0, 16, 3>(src, bytes, sink, grid, "dword, 16 KB per iteration, 3 in flight");
  }
  for (int skew : {0, 1, 4, 16}) {
    run<1, 4, 3>(src, bytes, sink, 256, "dwordx4, 16 KB per iteration, 3 in flight", 1, skew);
    run<1, 4, 8>(src, bytes, sink, 256, "dwordx4, 16 KB per iteration, 8 in flight", 1, skew);
  }
  return 0;
}
