// microbenchmark: LDS atomic add throughput (f32 vs u32 vs u64) under several address patterns
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %s\n", hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE, int PAT>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  __shared__ float h[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) h[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  int base;
  if (PAT == 0) base = lane;                                   // conflict-free, distinct banks
  else if (PAT == 1) base = (j >> 1) * 40 + 4 * (j & 1) - 4 * g + 64;   // the dq kernel's pattern (2-way same address)
  else if (PAT == 2) base = lane * 32;                         // same bank, distinct addresses (64-way bank conflict)
  else base = lane & 7;                                        // 8-way same address
  float v = 1.0f + lane;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int a = (base + u * 61 + it) & 4095;
      if (MODE == 0) atomicAdd(&h[a], v);
      else if (MODE == 1) atomicAdd((unsigned*)&h[a], (unsigned)v);
      else if (MODE == 3) atomicAdd((unsigned long long*)&h[(a & 2047) * 2], (unsigned long long)v);   // ds_add_u64 (round 4)
      else h[a] = v;
    }
  }
  __syncthreads();
  if (threadIdx.x < 64) out[blockIdx.x * 64 + threadIdx.x] = h[threadIdx.x];
}

template <int MODE, int PAT>
int run(const char* name) {
  float* d; CK(hipMalloc(&d, 4096 * 64 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 2000, grid = 1024;
  k<MODE, PAT><<<grid, 256>>>(d, 10);
  CK(hipEventRecord(e0));
  k<MODE, PAT><<<grid, 256>>>(d, iters);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double winstr = (double)grid * 4 * iters * 16;   // wave-instructions
  // 256 CUs, ~2.4 GHz: cycles per wave-instruction per CU
  const double cyc = ms * 1e-3 * 2.4e9 * 256 / winstr;
  printf("%-28s %8.3f ms  %6.1f CU-cycles / wave-instr\n", name, ms, cyc);
  CK(hipFree(d));
  return 0;
}

int main() {
  run<0, 0>("add_f32 conflict-free");
  run<1, 0>("add_u32 conflict-free");
  run<2, 0>("write_b32 conflict-free");
  run<3, 0>("add_u64 conflict-free");
  run<3, 1>("add_u64 dq-pattern");
  run<3, 3>("add_u64 8-way same addr");
  run<0, 1>("add_f32 dq-pattern");
  run<1, 1>("add_u32 dq-pattern");
  run<0, 3>("add_f32 8-way same addr");
  run<1, 3>("add_u32 8-way same addr");
  run<0, 2>("add_f32 64-way bank");
  run<1, 2>("add_u32 64-way bank");
  return 0;
}
