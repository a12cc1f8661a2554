// dma_rate.hip -- round 6: what one `buffer_load_dwordx4 ... lds` (LDS-DMA) / global_load_dwordx4 costs when its 64 lanes
// read 16 ROWS of 64 bytes (the K / V rows of one head at head_dim 32: a row is half a 128-byte line, token stride 192 B)
// instead of one contiguous KB, from an L2-resident region, by resident waves per CU and requests in flight per wave.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/dma_rate tools/ubench/dma_rate.hip && tools/ubench/dma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, (short)0, 0x7fffffff, 0x00020000);
}
// MODE 0: LDS-DMA, rows scattered; 1: LDS-DMA contiguous KB; 2: global_load to VGPR, rows scattered; 3: VGPR contiguous;
// 4: LDS-DMA, 8 whole 128-byte lines (2.67 token rows of 384 B: K | V of all three heads);
// 5: LDS-DMA, 16 half lines like 0, but the waves of a workgroup come in pairs that request the two halves of the SAME lines
//    at the same time (two heads of one chunk in one workgroup: does the CU's L1 merge them?)
template <int MODE, int INFLIGHT>
__global__ __launch_bounds__(256) void k(const char* src, int iters, int region_rows, int stride, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const __amdgpu_buffer_rsrc_t rs = mk(src);
  char* dst = smem + wave * (INFLIGHT * 1024);
  unsigned seed = blockIdx.x * 977 + (MODE == 5 ? (wave >> 1) : wave) * 131;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int f = 0; f < INFLIGHT; ++f) {
      seed = seed * 1664525u + 1013904223u;
      const int base_row = (seed >> 8) % (region_rows - 64);
      int off;
      if (MODE == 0 || MODE == 2) off = (base_row + (lane >> 2) * 3) * stride + (lane & 3) * 16;      // 16 rows, 3 rows apart
      else if (MODE == 5) off = (base_row + (lane >> 2) * 3) * 128 + (wave & 1) * 64 + (lane & 3) * 16;   // lines of 128 B, this wave's half
      else if (MODE == 4) off = (base_row + (lane / 24) * 5) * 384 + (lane % 24) * 16;              // 2.67 rows of 384 B, 5 rows apart
      else off = base_row * stride + lane * 16;
      if (MODE < 2 || MODE >= 4)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + f * 1024), 16, off, 0, 0, 0);
      else {
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
        acc += __builtin_bit_cast(float, v[0]);
      }
    }
    if (MODE < 2 || MODE >= 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE == 5) __syncthreads();
  }
  if (MODE < 2 || MODE >= 4) acc = ((float*)dst)[lane];
  if (acc == 123.456f) sink[0] = acc;
}
template <int MODE, int INFLIGHT> void run(const char* d, float* sink, int wgs, const char* name, int rows = 4096) {
  const int iters = 2000 / INFLIGHT, stride = 192;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE, INFLIGHT><<<wgs, 256, 4 * INFLIGHT * 1024>>>(d, 10, rows, stride, sink);
  hipEventRecord(a);
  k<MODE, INFLIGHT><<<wgs, 256, 4 * INFLIGHT * 1024>>>(d, iters, rows, stride, sink);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)wgs * 4 * iters * INFLIGHT * 1024;
  printf("%-34s wgs %5d (x4 waves) inflight %d: %7.1f us  %6.2f TB/s  %5.1f B/clk/CU  %6.0f cycles per request per wave\n", name, wgs, INFLIGHT,
         ms * 1e3, bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.4e9, ms * 1e-3 * 2.4e9 / (iters * INFLIGHT));
}
int main() {
  char* d; float* sink;
  const size_t big = (size_t)1100000 * 192 + 65536;
  hipMalloc(&d, big); hipMemset(d, 1, big); hipMalloc(&sink, 64);
  for (int wgs : {256, 768, 1280}) {
    run<0, 2>(d, sink, wgs, "LDS-DMA 16 rows x 64 B");
    run<0, 4>(d, sink, wgs, "LDS-DMA 16 rows x 64 B");
    run<1, 2>(d, sink, wgs, "LDS-DMA contiguous KB");
    run<1, 4>(d, sink, wgs, "LDS-DMA contiguous KB");
    run<2, 2>(d, sink, wgs, "global_load 16 rows x 64 B");
    run<2, 4>(d, sink, wgs, "global_load 16 rows x 64 B");
    run<3, 4>(d, sink, wgs, "global_load contiguous KB");
    run<4, 2>(d, sink, wgs, "LDS-DMA 8 whole lines");
    run<0, 2>(d, sink, wgs, "LDS-DMA 16 rows x 64 B, 200 MB", 1000000);
    run<4, 2>(d, sink, wgs, "LDS-DMA 8 whole lines, 200 MB", 500000);
    run<5, 2>(d, sink, wgs, "LDS-DMA half lines, paired waves");
    run<5, 2>(d, sink, wgs, "LDS-DMA half lines, paired, 200 MB", 1000000);
  }
  return 0;
}
