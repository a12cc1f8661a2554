// probe_layout.hip -- standalone hardware probe: checks, on the GPU it runs on, the
// lane<->element layouts the MFMA kernels rely on (mfma_f32_16x16x32_bf16 A/B/C/D
// fragments, ds_read_b64_tr_b16 transpose semantics).  Prints one PASS/FAIL line each.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <math.h>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// A is 16x32 row-major, B is 32x16 row-major, D = A*B + C (16x16 row-major)
__global__ void k_mfma(const float* A, const float* B, const float* C, float* D) {
  const int l = threadIdx.x, i = l & 15, g = l >> 4;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)A[i * 32 + 8 * g + e]; b[e] = (__bf16)B[(8 * g + e) * 16 + i]; }
  f32x4 c;
  for (int r = 0; r < 4; ++r) c[r] = C[(4 * g + r) * 16 + i];
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + i] = c[r];
}

// LDS holds a 16-row x 16-col bf16 tile, row pitch `pitch` elements; every lane passes the
// address the kernels use: row = 4g + (i>>2), col = 4*(i&3); expects out[lane][e] = T[4g+e][i]
__global__ void k_tr(const float* T, float* out, int pitch) {
  __shared__ __attribute__((aligned(16))) __bf16 lds[16 * 64];
  const int l = threadIdx.x, i = l & 15, g = l >> 4;
  for (int e = l; e < 16 * 16; e += 64) lds[(e / 16) * pitch + (e % 16)] = (__bf16)T[e];
  __syncthreads();
  const int row = 4 * g + (i >> 2), col = 4 * (i & 3);
  const s16x4 t4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (s16x4 __attribute__((address_space(3)))*)(lds + row * pitch + col));
  const bf16x4 tb = __builtin_bit_cast(bf16x4, t4);
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = (float)tb[e];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAIL hip %s\n", hipGetErrorString(e_)); return 2; } } while (0)

int main() {
  int rc = 0;
  std::vector<float> A(16 * 32), B(32 * 16), C(256), D(256), R(256);
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 32; ++k) A[i * 32 + k] = (float)((i * 7 + k * 3) % 11 - 5);
  for (int k = 0; k < 32; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = (float)((k * 5 + j * 13) % 7 - 3);
  for (int i = 0; i < 256; ++i) C[i] = (float)(i % 9) * 0.25f;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
    float s = C[i * 16 + j];
    for (int k = 0; k < 32; ++k) s += A[i * 32 + k] * B[k * 16 + j];
    R[i * 16 + j] = s;
  }
  float *dA, *dB, *dC, *dD;
  CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, 1024)); CK(hipMalloc(&dD, 4096));
  CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dC, C.data(), 1024, hipMemcpyHostToDevice));
  k_mfma<<<1, 64>>>(dA, dB, dC, dD);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost));
  float err = 0; for (int i = 0; i < 256; ++i) err = fmaxf(err, fabsf(D[i] - R[i]));
  printf("%s mfma_f32_16x16x32_bf16 layout (max err %g)\n", err == 0 ? "PASS" : "FAIL", err);
  rc |= err != 0;

  for (int pitch : {16, 32, 64}) {
    std::vector<float> T(256), O(256);
    for (int e = 0; e < 256; ++e) T[e] = (float)e;     // exactly representable in bf16
    CK(hipMemcpy(dA, T.data(), 1024, hipMemcpyHostToDevice));
    k_tr<<<1, 64>>>(dA, dD, pitch);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(O.data(), dD, 1024, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) {
      const int i = l & 15, g = l >> 4;
      if (O[l * 4 + e] != T[(4 * g + e) * 16 + i]) ++bad;
    }
    printf("%s ds_read_b64_tr_b16 transpose semantics, pitch %d (%d mismatches)\n", bad ? "FAIL" : "PASS", pitch, bad);
    if (bad) {
      printf("  dump lane: got[0..3] (expected)\n");
      for (int l = 0; l < 64; l += 5)
        printf("  lane %2d: %g %g %g %g  (%g %g %g %g)\n", l, O[l*4], O[l*4+1], O[l*4+2], O[l*4+3],
               T[(4*(l>>4)+0)*16+(l&15)], T[(4*(l>>4)+1)*16+(l&15)], T[(4*(l>>4)+2)*16+(l&15)], T[(4*(l>>4)+3)*16+(l&15)]);
    }
    rc |= bad != 0;
  }
  return rc;
}
