// valu_rate.hip -- issue rates of the vector instructions the attention loops are made of, on gfx950, as a function of
// resident waves per SIMD: plain VALU (v_fma_f32), transcendental (v_exp_f32), v_cvt_pk_bf16_f32, v_max3_f32,
// v_pk_fma_f32, ds_read_b32 (conflict-free) and v_mfma_f32_16x16x32_bf16, each as 16 independent chains per wave.
// Output: cycles per wave-instruction seen by ONE wave (s_memtime) and chip-level instructions / cycle / SIMD (hipEvent).
// Feeds the `valu` roof of bench.py (review r04, weak item 5).   hipcc --offload-arch=gfx950 -O2 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 1.0f;
  __syncthreads();
  float x[16];
  f32x2 y[8];
  f32x4 acc[4];
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)1.0f; b[e] = (__bf16)0.5f; }
  for (int i = 0; i < 16; ++i) x[i] = 0.001f * (threadIdx.x + i);
  for (int i = 0; i < 8; ++i) y[i] = (f32x2){0.001f * i, 0.002f * threadIdx.x};
  for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  unsigned addr = (threadIdx.x & 63) * 4;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (OP == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[i]));
      if (OP == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
      if (OP == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(x[i]));
      if (OP == 3) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(x[i]));
      if (OP == 4) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(y[i & 7]));
      if (OP == 5) asm volatile("ds_read_b32 %0, %1" : "=v"(x[i]) : "v"(addr));
      if (OP == 6) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(a), "v"(b));
      if (OP == 7) {            // the softmax pair: exp + fma, interleaved
        asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
        asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[(i + 8) & 15]));
      }
      if (OP == 8) {            // MFMA + 4 plain VALU per MFMA (do they overlap inside ONE wave?)
        if ((i & 3) == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[(i >> 2) & 3]) : "v"(a), "v"(b));
        asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[i]));
      }
    }
    if (OP == 5) asm volatile("s_waitcnt lgkmcnt(0)");
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += x[i];
  for (int i = 0; i < 8; ++i) s += y[i][0] + y[i][1];
  for (int i = 0; i < 4; ++i) s += acc[i][0];
  if (s == 123.456f) out[0] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int OP>
void run(const char* name, int per_iter) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 4); hipMalloc(&cyc, 8);
  const int iters = 2000;
  printf("%-28s", name);
  for (int wps : {1, 2, 3, 4, 8}) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<256 * wps, 256>>>(out, cyc, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<256 * wps, 256>>>(out, cyc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n_wave = (double)iters * per_iter;                 // instructions one wave issued
    const double chip_cycles = ms * 1e-3 * 2.4e9;
    printf("  w%d: %6.2f cyc/inst/wave (memtime x%.0f)  %5.3f inst/clk/SIMD |", wps, (double)c / n_wave * (100e6 / 100e6), 1.0,
           n_wave * wps / chip_cycles);
  }
  printf("\n");
  hipFree(out); hipFree(cyc);
}

int main() {
  printf("# per-wave cycles are s_memtime ticks (100 MHz constant clock on gfx9: multiply by clock/100MHz) or shader cycles -- compare rows, and use inst/clk/SIMD (hipEvent, 2.4 GHz assumed)\n");
  run<0>("v_fma_f32", 16);
  run<1>("v_exp_f32", 16);
  run<2>("v_cvt_pk_bf16_f32", 16);
  run<3>("v_max3_f32", 16);
  run<4>("v_pk_fma_f32", 16);
  run<5>("ds_read_b32", 16);
  run<6>("v_mfma_f32_16x16x32_bf16", 16);
  run<7>("exp+fma pairs", 32);
  run<8>("1 mfma + 4 fma", 20);
  return 0;
}
