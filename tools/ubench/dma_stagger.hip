// What an LDS-DMA copy costs the MFMA stream of a chunk-stream kernel, by WHERE the four waves of a workgroup issue their copies.
// One workgroup per CU, four waves (one per SIMD: 512 registers asked for), per chunk 96 MFMAs (v_mfma_f32_16x16x32_f16 on six
// independent accumulators) + NCOPY global_load_lds_dwordx4 per wave (1 KB each) + optionally one ds_read_b128 per two MFMAs, a counted
// vmcnt wait and one s_barrier per chunk: the skeleton of k_cesr_x6 (NCOPY 12) / the two-tile kernels (NCOPY 6).
//   MODE 0  no copies
//   MODE 1  every wave issues copy i behind the SAME MFMA (96 / NCOPY apart): what the kernels do today
//   MODE 2  wave w issues copy i two MFMAs later than wave w - 1 (32 cycles apart: the texture-address unit takes ~20 per 1 KB)
//   MODE 3  wave w issues all its copies back to back, in its own quarter of the chunk
//   MODE 4  wave w issues its copies two MFMAs apart inside its own quarter of the chunk
//   MODE 5  MODE 2 with run-time tests of the wave id at four candidate positions per copy (one code path for all waves)
//   MODE 6  MODE 1 and after the barrier wave w idles w x 32 cycles
//   MODE 7 / 8 / 9  all waves at the same MFMAs, but clustered: bursts per half chunk / every fourth MFMA / back-to-back pairs
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/dma_stagger.hip -o /tmp/dma_stagger && /tmp/dma_stagger
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
#ifndef RA
#define RA 2               // fragment reads in flight ahead of the MFMA that uses them (one read per two MFMAs); -DRA=6: 192 cycles ahead
#endif
#define NFR (RA + 2)

__device__ __forceinline__ void dma(const void* gbase, unsigned lane_off, unsigned lds) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(lane_off), "s"(gbase) : "memory");
}

template <int NCOPY, int MODE, int READS, int W>
__device__ __forceinline__ void chunk_body(f4 (&acc)[6], h8& a, h8& b, const char* src_chunk, unsigned lane_off, unsigned dst, unsigned rd_addr,
                                           u4 (&frag)[NFR], int wave_rt) {
  constexpr int GAP = 96 / NCOPY;
  int issued = 0;
#pragma unroll
  for (int s = 0; s < 96; ++s) {
    acc[s % 6] = __builtin_amdgcn_mfma_f32_16x16x32_f16(s & 1 ? __builtin_bit_cast(h8, frag[(s / 2) % NFR]) : a, b, acc[s % 6], 0, 0, 0);
    if (READS && (s & 1)) frag[(s / 2 + RA) % NFR] = ((const __attribute__((address_space(3))) u4*)rd_addr)[((s / 2) % 48) * 64];
    if (MODE == 1 || MODE == 6) {
      if (s % GAP == 0) {
        const int i = s / GAP;
        dma(src_chunk + (wave_rt * NCOPY + i) * 1024, lane_off, dst + (unsigned)(wave_rt * NCOPY + i) * 1024u);
      }
    } else if (MODE == 2) {
      if (s >= 2 * W && (s - 2 * W) % GAP == 0 && (s - 2 * W) / GAP < NCOPY) {
        const int i = (s - 2 * W) / GAP;
        dma(src_chunk + (wave_rt * NCOPY + i) * 1024, lane_off, dst + (unsigned)(wave_rt * NCOPY + i) * 1024u);
      }
    } else if (MODE == 3) {
      if (s == 24 * W) {
#pragma unroll
        for (int i = 0; i < NCOPY; ++i) dma(src_chunk + (wave_rt * NCOPY + i) * 1024, lane_off, dst + (unsigned)(wave_rt * NCOPY + i) * 1024u);
      }
    } else if (MODE == 4) {
      constexpr int STEP = 24 / NCOPY >= 1 ? 24 / NCOPY : 1;
      if (s >= 24 * W && (s - 24 * W) % STEP == 0 && (s - 24 * W) / STEP < NCOPY) {
        const int i = (s - 24 * W) / STEP;
        dma(src_chunk + (wave_rt * NCOPY + i) * 1024, lane_off, dst + (unsigned)(wave_rt * NCOPY + i) * 1024u);
      }
    } else if (MODE == 7) {      // all waves alike, per half chunk a burst of 4 and a burst of 2 one k-block later (12 copies: k_cesr_x6 today)
      constexpr int H = NCOPY / 2;                        // copies per half
      const int sh = s % 48, half = s / 48;
      if (sh == 24) {
#pragma unroll
        for (int i = 0; i < (H * 2) / 3; ++i) dma(src_chunk + (wave_rt * NCOPY + half * H + i) * 1024, lane_off, dst + (unsigned)(wave_rt * NCOPY + half * H + i) * 1024u);
      }
      if (sh == 30) {
#pragma unroll
        for (int i = (H * 2) / 3; i < H; ++i) dma(src_chunk + (wave_rt * NCOPY + half * H + i) * 1024, lane_off, dst + (unsigned)(wave_rt * NCOPY + half * H + i) * 1024u);
      }
    } else if (MODE == 8) {      // all waves alike, one copy every FOUR MFMAs from the middle of each half (the two-tile kernels today: consecutive filler positions)
      constexpr int H = NCOPY / 2;
      const int sh = s % 48, half = s / 48;
      if (sh >= 8 && (sh - 8) % 4 == 0 && (sh - 8) / 4 < H) {
        const int i = (sh - 8) / 4;
        dma(src_chunk + (wave_rt * NCOPY + half * H + i) * 1024, lane_off, dst + (unsigned)(wave_rt * NCOPY + half * H + i) * 1024u);
      }
    } else if (MODE == 9) {      // all waves alike, pairs of copies back to back, 2 GAP apart
      if (s % (2 * GAP) == 0) {
        const int i = s / GAP;
        dma(src_chunk + (wave_rt * NCOPY + i) * 1024, lane_off, dst + (unsigned)(wave_rt * NCOPY + i) * 1024u);
        dma(src_chunk + (wave_rt * NCOPY + i + 1) * 1024, lane_off, dst + (unsigned)(wave_rt * NCOPY + i + 1) * 1024u);
      }
    } else if (MODE == 5) {
      // candidate positions of copy i: GAP i + 2 w, w = 0..3 -- one code path, the wave id tested at run time
#pragma unroll
      for (int w = 0; w < 4; ++w)
        if (s >= 2 * w && (s - 2 * w) % GAP == 0 && (s - 2 * w) / GAP < NCOPY) {
          const int i = (s - 2 * w) / GAP;
          if (__builtin_expect(wave_rt == w, 0)) dma(src_chunk + (wave_rt * NCOPY + i) * 1024, lane_off, dst + (unsigned)(wave_rt * NCOPY + i) * 1024u);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  (void)issued;
}

template <int NCOPY, int MODE, int READS>
__global__ __launch_bounds__(256, 1) void k_stream(const char* __restrict__ src, long src_bytes, int iters, float* sink, long* cycles) {
  extern __shared__ f4 ring[];   // 3 slots x NCOPY x 4 KB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned ring_b = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)ring);
  constexpr unsigned CHB = 4u * NCOPY * 1024u;
  const unsigned lane_off = (unsigned)lane * 16u;
  f4 acc[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
  h8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = (_Float16)(0.001f * (lane + i));
    b[i] = (_Float16)(0.002f * (lane - i));
  }
  u4 frag[NFR];
#pragma unroll
  for (int i = 0; i < NFR; ++i) frag[i] = u4{0u, 0u, 0u, 0u};
  // prologue: chunks 0 and 1
  if (MODE != 0)
    for (int c = 0; c < 2; ++c)
      for (int i = 0; i < NCOPY; ++i) dma(src + (long)c * CHB + (wave * NCOPY + i) * 1024, lane_off, ring_b + (unsigned)c * CHB + (wave * NCOPY + i) * 1024u);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  const long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const char* sc = src + (long)((it + 2) & 127) * CHB;   // 128 chunks: 6 MB (12 copies) / 3 MB (6 copies), every workgroup the same stream
    const unsigned dst = ring_b + (unsigned)((it + 2) % 3) * CHB;
    const unsigned rd = ring_b + (unsigned)(it % 3) * CHB + lane * 16u;
    if (MODE == 6)
      for (int i = 0; i < wave; ++i) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    switch (MODE == 5 || MODE == 1 || MODE >= 6 || MODE == 0 ? 0 : wave) {
      case 0: chunk_body<NCOPY, MODE, READS, 0>(acc, a, b, sc, lane_off, dst, rd, frag, wave); break;
      case 1: chunk_body<NCOPY, MODE, READS, 1>(acc, a, b, sc, lane_off, dst, rd, frag, wave); break;
      case 2: chunk_body<NCOPY, MODE, READS, 2>(acc, a, b, sc, lane_off, dst, rd, frag, wave); break;
      default: chunk_body<NCOPY, MODE, READS, 3>(acc, a, b, sc, lane_off, dst, rd, frag, wave); break;
    }
    if (MODE != 0) {
      if (NCOPY == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
  }
  const long t1 = __builtin_readcyclecounter();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < 6; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (r == 123.456f) sink[0] = r + (float)frag[0][0];
  if (tid == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int NCOPY, int MODE, int READS>
void run(const char* src, long bytes, float* sink, long* cycles, const char* label) {
  const int iters = 4000, grid = 256;
  const size_t lds = 3u * 4u * 12 * 1024u;   // 144 KB whatever NCOPY: one workgroup per CU
  hipFuncSetAttribute((const void*)k_stream<NCOPY, MODE, READS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k_stream<NCOPY, MODE, READS>), dim3(grid), dim3(256), lds, 0, src, bytes, 400, sink, cycles);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_stream<NCOPY, MODE, READS>), dim3(grid), dim3(256), lds, 0, src, bytes, iters, sink, cycles);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  long h[256];
  hipMemcpy(h, cycles, sizeof(h), hipMemcpyDeviceToHost);
  double mean = 0;
  for (int i = 0; i < grid; ++i) mean += (double)h[i];
  mean /= grid;
  // s_memtime / readcyclecounter ticks at 100 MHz on this part: use the event time and the nominal clock instead
  printf("copies %2d mode %d reads %d  %-46s %7.3f us per chunk = %6.1f ns per MFMA (16 cycles at 2.4 GHz = 6.67 ns); counter %.0f per chunk\n", NCOPY, MODE,
         READS, label, ms * 1e3 / iters, ms * 1e6 / iters / 96.0, mean / iters);
}

template <int NCOPY, int READS>
void sweep(const char* src, long bytes, float* sink, long* cycles) {
  run<NCOPY, 0, READS>(src, bytes, sink, cycles, "no copies");
  run<NCOPY, 1, READS>(src, bytes, sink, cycles, "all waves at the same MFMA (today)");
  run<NCOPY, 2, READS>(src, bytes, sink, cycles, "waves two MFMAs apart");
  run<NCOPY, 3, READS>(src, bytes, sink, cycles, "a wave's copies back to back, own quarter");
  run<NCOPY, 4, READS>(src, bytes, sink, cycles, "a wave's copies spread over its own quarter");
  run<NCOPY, 5, READS>(src, bytes, sink, cycles, "two MFMAs apart, run-time wave test");
  run<NCOPY, 6, READS>(src, bytes, sink, cycles, "same MFMA + w x 32 idle cycles after the barrier");
  run<NCOPY, 7, READS>(src, bytes, sink, cycles, "alike, bursts of 2/3 + 1/3 per half chunk");
  run<NCOPY, 8, READS>(src, bytes, sink, cycles, "alike, one copy every 4 MFMAs per half chunk");
  run<NCOPY, 9, READS>(src, bytes, sink, cycles, "alike, pairs back to back, evenly spread");
}

int main() {
  const long bytes = 8 << 20;
  char* src;
  float* sink;
  long* cycles;
  hipMalloc(&src, bytes);
  hipMalloc(&sink, 4);
  hipMalloc(&cycles, 256 * sizeof(long));
  hipMemset(src, 0, bytes);
  sweep<12, 0>(src, bytes, sink, cycles);
  sweep<12, 1>(src, bytes, sink, cycles);
  sweep<6, 0>(src, bytes, sink, cycles);
  sweep<6, 1>(src, bytes, sink, cycles);
  return 0;
}
