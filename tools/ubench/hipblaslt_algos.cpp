// Probe: for the model's GEMM shapes, how do the top-N hipBLASLt heuristic algorithms compare with the first one
// (the one PyTorch uses)?   hipcc -O2 tools/ubench/hipblaslt_algos.cpp -lhipblaslt -o /tmp/lt_algos && /tmp/lt_algos
// Row-major  Y[T][N] = X[T][K] * W[N][K]^T   (forward)  ==  column-major  Y^T[N][T] = W^T... (op T on W)
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { auto e_ = (x); if (e_ != 0) { printf("error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 1; } } while (0)

// random bf16 in (-1, 1): constant data flatters the clocks (MI355X guide: DVFS differs on uniform bit patterns)
__global__ void k_fill(unsigned short* p, size_t n, unsigned seed) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
  const float f = ((x >> 8) & 0xffff) / 32768.0f - 1.0f;
  p[i] = (unsigned short)(__float_as_uint(f) >> 16);
}

struct Shape { const char* name; int T, K, N; int kind; };   // kind 0: fwd (Y = X W^T), 1: dgrad (dX = dY W)

int main() {
  hipblasLtHandle_t h; CK(hipblasLtCreate(&h));
  std::vector<Shape> shapes = {
    {"s3 qkv fwd", 25216, 384, 1152, 0}, {"s3 fc1 fwd", 25216, 384, 1536, 0}, {"s3 fc2 fwd", 25216, 1536, 384, 0},
    {"s3 proj fwd", 25216, 384, 384, 0}, {"s3 fc1 dgrad", 25216, 1536, 384, 1}, {"s3 fc2 dgrad", 25216, 384, 1536, 1},
    {"s3 qkv dgrad", 25216, 1152, 384, 1}, {"s2 fc1 fwd", 100480, 192, 768, 0}, {"s2 fc2 fwd", 100480, 768, 192, 0},
    {"s1 fc1 fwd", 401536, 96, 384, 0}, {"s1 fc2 fwd", 401536, 384, 96, 0}, {"s1 fc1 dgrad", 401536, 384, 96, 1},
    // kind 2: weight gradient dW[N][K] = dY[T][N]^T X[T][K]  (contraction over T)
    {"s3 fc1 wgrad", 25216, 384, 1536, 2}, {"s3 fc2 wgrad", 25216, 1536, 384, 2}, {"s3 proj wgrad", 25216, 384, 384, 2},
    {"s2 fc1 wgrad", 100480, 192, 768, 2}, {"s1 fc1 wgrad", 401536, 96, 384, 2},
  };
  size_t wsz = 64u << 20; void* ws; CK(hipMalloc(&ws, wsz));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (auto& sh : shapes) {
    // column-major view: C(N x T) = op(A)(N x K) * B(K x T);  B = X^T view = X row-major (K x T col-major, ld K)
    // fwd:   A = W row-major (N,K) = col-major (K x N) ld K, op T
    // dgrad: dX[T][K'] = dY[T][N'] W[N'][K'];  here sh.K = N' (contraction), sh.N = K' (output cols)
    //        col-major: C(K' x T) = A(K' x N') * B(N' x T), A = W row-major (N',K') = col-major (K' x N') ld K', op N
    int64_t M_ = sh.N, N_ = sh.T, K_ = sh.K;
    if (sh.kind == 2) { M_ = sh.K; N_ = sh.N; K_ = sh.T; }     // C (K x N col-major = dW (N,K) row-major) = X^T-view (K x T) * dY (T x N)
    void *A, *B, *C;
    // generous allocations (every kind touches at most T*max(K,N) elements per operand)
    const size_t big = (size_t)sh.T * (sh.K > sh.N ? sh.K : sh.N) * 2;
    CK(hipMalloc(&A, big)); CK(hipMalloc(&B, big)); CK(hipMalloc(&C, big));
    { size_t na = big / 2, nb = big / 2;
      k_fill<<<(unsigned)((na + 255) / 256), 256>>>((unsigned short*)A, na, 1u);
      k_fill<<<(unsigned)((nb + 255) / 256), 256>>>((unsigned short*)B, nb, 7u); CK(hipDeviceSynchronize()); }
    hipblasLtMatmulDesc_t md; CK(hipblasLtMatmulDescCreate(&md, HIPBLAS_COMPUTE_32F, HIP_R_32F));
    hipblasOperation_t ta = sh.kind == 0 ? HIPBLAS_OP_T : HIPBLAS_OP_N, tb = sh.kind == 2 ? HIPBLAS_OP_T : HIPBLAS_OP_N;
    CK(hipblasLtMatmulDescSetAttribute(md, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta)));
    CK(hipblasLtMatmulDescSetAttribute(md, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb)));
    hipblasLtMatrixLayout_t la, lb, lc;
    if (sh.kind == 0) CK(hipblasLtMatrixLayoutCreate(&la, HIP_R_16BF, K_, M_, K_));
    else CK(hipblasLtMatrixLayoutCreate(&la, HIP_R_16BF, M_, K_, M_));
    if (sh.kind == 2) CK(hipblasLtMatrixLayoutCreate(&lb, HIP_R_16BF, N_, K_, N_));
    else CK(hipblasLtMatrixLayoutCreate(&lb, HIP_R_16BF, K_, N_, K_));
    CK(hipblasLtMatrixLayoutCreate(&lc, HIP_R_16BF, M_, N_, M_));
    hipblasLtMatmulPreference_t pref; CK(hipblasLtMatmulPreferenceCreate(&pref));
    CK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsz, sizeof(wsz)));
    const int REQ = 24;
    hipblasLtMatmulHeuristicResult_t res[REQ]; int got = 0;
    CK(hipblasLtMatmulAlgoGetHeuristic(h, md, la, lb, lc, lc, pref, REQ, res, &got));
    float alpha = 1.f, beta = 0.f;
    std::vector<float> us(got, 1e30f);
    for (int i = 0; i < got; ++i) {
      if (res[i].state != HIPBLAS_STATUS_SUCCESS || res[i].workspaceSize > wsz) continue;
      bool ok = true;
      for (int r = 0; r < 3 && ok; ++r)
        ok = hipblasLtMatmul(h, md, &alpha, A, la, B, lb, &beta, C, lc, C, lc, &res[i].algo, ws, wsz, s) == HIPBLAS_STATUS_SUCCESS;
      if (!ok) continue;
      CK(hipEventRecord(e0, s));
      for (int r = 0; r < 10; ++r)
        hipblasLtMatmul(h, md, &alpha, A, la, B, lb, &beta, C, lc, C, lc, &res[i].algo, ws, wsz, s);
      CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); us[i] = ms * 100.f;
    }
    int best = (int)(std::min_element(us.begin(), us.end()) - us.begin());
    const double fl = 2.0 * sh.T * sh.K * sh.N;
    printf("%-14s T=%6d K=%4d N=%4d  algos %2d  first %7.1f us (%6.1f TF)  best #%d %7.1f us (%6.1f TF)  gain %.2fx\n", sh.name, sh.T, sh.K,
           sh.N, got, us[0], fl / us[0] / 1e6, best, us[best], fl / us[best] / 1e6, us[0] / us[best]);
    hipFree(A); hipFree(B); hipFree(C);
    hipblasLtMatmulPreferenceDestroy(pref); hipblasLtMatrixLayoutDestroy(la); hipblasLtMatrixLayoutDestroy(lb); hipblasLtMatrixLayoutDestroy(lc);
    hipblasLtMatmulDescDestroy(md);
  }
  return 0;
}
