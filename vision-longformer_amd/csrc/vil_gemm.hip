// vil_gemm.hip -- the plain library GEMMs of the projections (forward Y = X W^T + b, input gradient dX = dY W)
// through hipBLASLt with the algorithm SELECTED BY MEASUREMENT per problem.  ("plain library GEMMs
// stay on hipBLASLt"; reference call sites: every nn.Linear of msvit.py / longformer2d.py.)
// hipblasLtMatmulAlgoGetHeuristic returns a ranked list; its first entry (what a framework call gets) loses
// 5-25 % against the best of the top 16 on this model's skinny shapes (tools/ubench/hipblaslt_algos.cpp), and the
// framework's own dispatch was another 10-50 % behind on several of them.  Selection is an EXPLICIT call,
// vil_gemm_tune: it times the candidates on scratch operands with hipEvents (it synchronises -- call it outside
// stream capture, once per problem) and caches the winner; vil_gemm_bf16 itself only launches (asynchronous, never
// synchronises; an untuned problem runs the heuristic's first choice).
// State: the plan cache below ((device, problem) -> descriptors + selected algorithm) is process-global and mutex-guarded; it
// is the library's only mutable state besides the profiling sink of vil_attn_api.hip.
#include "vil_internal.h"
#include <hipblaslt/hipblaslt.h>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

namespace {

struct Plan {
  hipblasLtMatmulDesc_t md = nullptr;
  hipblasLtMatrixLayout_t la = nullptr, lb = nullptr, lc = nullptr;
  hipblasLtMatmulAlgo_t algo;
  size_t ws = 0;
  bool tuned = false, valid = false;
  std::vector<hipblasLtMatmulHeuristicResult_t> cand;
};

typedef std::tuple<int, int, int64_t, int, int, int64_t, int64_t, int> Key;   // device, op, T, K, N, in stride, out stride, bias
std::map<Key, Plan> g_plans;
std::mutex g_mu;
std::map<int, hipblasLtHandle_t> g_handles;      // one handle per device (a plan is tuned on, and valid for, its device)
hipblasLtHandle_t g_handle = nullptr;            // the calling thread's current device's handle (set by get_plan under g_mu)

int make_plan(Plan& pl, int op, int64_t T, int K, int N, int64_t in_rs, int64_t out_rs, bool bias, size_t wsz) {
  // row-major problem restated column-major: C (N x T, ld out_rs) = op(A) (N x K) * B (K x T, ld in_rs)
  //   forward: A = W (N,K) row-major = (K x N) column-major, ld K, transposed;  dgrad: A = W (K,N) row-major =
  //   (N x K) column-major, ld N, not transposed.   (K = contraction length, N = output features)
  //   weight gradient (op 2): C (K x N column-major = dW (N,K) row-major, ld K) = A (K x T: x row-major, ld in_rs)
  //   * B^T (B = dY row-major (T,N) = (N x T) column-major, ld out_rs... passed as `w`); bias = db via BGRADB
  if (hipblasLtMatmulDescCreate(&pl.md, HIPBLAS_COMPUTE_32F, HIP_R_32F) != HIPBLAS_STATUS_SUCCESS) return 1;
  hipblasOperation_t ta = op == 0 ? HIPBLAS_OP_T : HIPBLAS_OP_N, tb = op == 2 ? HIPBLAS_OP_T : HIPBLAS_OP_N;
  hipblasLtMatmulDescSetAttribute(pl.md, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta));
  hipblasLtMatmulDescSetAttribute(pl.md, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb));
  if (bias) {
    hipblasLtEpilogue_t ep = op == 2 ? HIPBLASLT_EPILOGUE_BGRADB : HIPBLASLT_EPILOGUE_BIAS;
    int32_t bt = HIP_R_16BF;
    hipblasLtMatmulDescSetAttribute(pl.md, HIPBLASLT_MATMUL_DESC_EPILOGUE, &ep, sizeof(ep));
    hipblasLtMatmulDescSetAttribute(pl.md, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt));
  }
  hipblasStatus_t st;
  if (op == 2) {
    // here T = contraction length, K = C_in, N = C_out; in_rs = row stride of x, out_rs = row stride of dY
    if (hipblasLtMatrixLayoutCreate(&pl.la, HIP_R_16BF, K, T, in_rs) != HIPBLAS_STATUS_SUCCESS) return 1;
    if (hipblasLtMatrixLayoutCreate(&pl.lb, HIP_R_16BF, N, T, out_rs) != HIPBLAS_STATUS_SUCCESS) return 1;
    if (hipblasLtMatrixLayoutCreate(&pl.lc, HIP_R_16BF, K, N, K) != HIPBLAS_STATUS_SUCCESS) return 1;
  } else {
    if (op == 0) st = hipblasLtMatrixLayoutCreate(&pl.la, HIP_R_16BF, K, N, K);
    else st = hipblasLtMatrixLayoutCreate(&pl.la, HIP_R_16BF, N, K, N);
    if (st != HIPBLAS_STATUS_SUCCESS) return 1;
    if (hipblasLtMatrixLayoutCreate(&pl.lb, HIP_R_16BF, K, T, in_rs) != HIPBLAS_STATUS_SUCCESS) return 1;
    if (hipblasLtMatrixLayoutCreate(&pl.lc, HIP_R_16BF, N, T, out_rs) != HIPBLAS_STATUS_SUCCESS) return 1;
  }
  hipblasLtMatmulPreference_t pref;
  if (hipblasLtMatmulPreferenceCreate(&pref) != HIPBLAS_STATUS_SUCCESS) return 1;
  hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsz, sizeof(wsz));
  const int REQ = 16;
  hipblasLtMatmulHeuristicResult_t res[REQ];
  int got = 0;
  if (bias) {                     // the heuristic wants a non-null bias pointer to rank epilogue kernels
    const void* dummy = (const void*)16;
    hipblasLtMatmulDescSetAttribute(pl.md, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &dummy, sizeof(dummy));
  }
  st = hipblasLtMatmulAlgoGetHeuristic(g_handle, pl.md, pl.la, pl.lb, pl.lc, pl.lc, pref, REQ, res, &got);
  hipblasLtMatmulPreferenceDestroy(pref);
  if (st != HIPBLAS_STATUS_SUCCESS || got == 0) return 1;
  for (int i = 0; i < got; ++i)
    if (res[i].state == HIPBLAS_STATUS_SUCCESS && res[i].workspaceSize <= wsz) pl.cand.push_back(res[i]);
  if (pl.cand.empty()) return 1;
  pl.algo = pl.cand[0].algo; pl.ws = pl.cand[0].workspaceSize;
  pl.valid = true;
  return 0;
}

}  // namespace

extern "C" size_t vil_gemm_workspace_bytes(void) { return (size_t)32 << 20; }

static int get_plan(Plan*& out, int op, int64_t T, int K, int N, int64_t in_rs, int64_t out_rs, bool bias, size_t wsz) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return VIL_E_BACKEND;
  hipblasLtHandle_t& hd = g_handles[dev];
  if (!hd && hipblasLtCreate(&hd) != HIPBLAS_STATUS_SUCCESS) return VIL_E_BACKEND;
  g_handle = hd;
  const Key key(dev, op, T, K, N, in_rs, out_rs, bias ? 1 : 0);
  Plan& pl = g_plans[key];
  if (!pl.valid && make_plan(pl, op, T, K, N, in_rs, out_rs, bias, wsz)) return VIL_E_BACKEND;
  out = &pl;
  return VIL_OK;
}

static hipblasStatus_t run_plan(Plan& pl, const hipblasLtMatmulAlgo_t& a, int op, const void* in, const void* w, void* out,
                                void* workspace, size_t workspace_bytes, hipStream_t s) {
  const float alpha = 1.f, beta = 0.f;
  if (op == 2)      // A = x (`in`), B = dY (`w`)
    return hipblasLtMatmul(g_handle, pl.md, &alpha, in, pl.la, w, pl.lb, &beta, out, pl.lc, out, pl.lc, &a, workspace,
                           workspace_bytes, s);
  return hipblasLtMatmul(g_handle, pl.md, &alpha, w, pl.la, in, pl.lb, &beta, out, pl.lc, out, pl.lc, &a, workspace,
                         workspace_bytes, s);
}

static int check_gemm_args(int op, const void* in, const void* w, const void* bias, const void* out, int64_t T, int K, int N,
                           int64_t in_rs, int64_t out_rs, const void* workspace) {
  if (!in || !w || !out || !workspace) return VIL_E_NULL;
  if (T <= 0 || K <= 0 || N <= 0 || op < 0 || op > 2 || (op == 1 && bias)) return VIL_E_SHAPE;
  if ((K & 7) || (N & 7) || (in_rs & 7) || (out_rs & 7) ||
      (((uintptr_t)in | (uintptr_t)w | (uintptr_t)out | (uintptr_t)workspace) & 15)) return VIL_E_ALIGN;
  return VIL_OK;
}

// Selects the algorithm of one problem by measurement on the caller's operands (`out` is overwritten with the
// product, exactly as vil_gemm_bf16 would).  SYNCHRONISES the stream; returns VIL_E_BACKEND during stream capture.
// Idempotent: a problem that is already tuned returns at once.
extern "C" int vil_gemm_tune(int op, const void* in, const void* w, const void* bias, void* out, int64_t T, int K, int N,
                             int64_t in_row_stride, int64_t out_row_stride, void* workspace, size_t workspace_bytes,
                             void* stream) {
  int e = check_gemm_args(op, in, w, bias, out, T, K, N, in_row_stride, out_row_stride, workspace);
  if (e) return e;
  hipStream_t s = (hipStream_t)stream;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(s, &cs);
  if (cs != hipStreamCaptureStatusNone) return VIL_E_BACKEND;
  std::lock_guard<std::mutex> lock(g_mu);
  Plan* plp = nullptr;
  if ((e = get_plan(plp, op, T, K, N, in_row_stride, out_row_stride, bias != nullptr, workspace_bytes))) return e;
  Plan& pl = *plp;
  if (pl.tuned) return VIL_OK;
  if (bias) hipblasLtMatmulDescSetAttribute(pl.md, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias));
  if (pl.cand.size() > 1) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    // two passes over the candidates, 8 back-to-back launches each, a candidate's time = its better pass: one short
    // measurement per candidate let clock ramp-up and a cold L2 decide between algorithms that differ by a few percent
    std::vector<float> t(pl.cand.size(), 1e30f);
    for (int pass = 0; pass < 2; ++pass)
      for (size_t i = 0; i < pl.cand.size(); ++i) {
        const auto& c = pl.cand[i];
        if (run_plan(pl, c.algo, op, in, w, out, workspace, workspace_bytes, s) != HIPBLAS_STATUS_SUCCESS) continue;   // warm-up / validity
        (void)hipEventRecord(e0, s);
        bool ok = true;
        for (int r = 0; r < 8 && ok; ++r) ok = run_plan(pl, c.algo, op, in, w, out, workspace, workspace_bytes, s) == HIPBLAS_STATUS_SUCCESS;
        (void)hipEventRecord(e1, s);
        if (hipEventSynchronize(e1) != hipSuccess || !ok) continue;
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < t[i]) t[i] = ms;
      }
    float best = 1e30f;
    for (size_t i = 0; i < pl.cand.size(); ++i)
      if (t[i] < best) { best = t[i]; pl.algo = pl.cand[i].algo; pl.ws = pl.cand[i].workspaceSize; }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  }
  pl.tuned = true;
  return VIL_OK;
}

// op 0: out[T][N] = in[T][K] * w[N][K]^T (+ bias[N]);   op 1: out[T][N] = in[T][K] * w[K][N]   (w row-major)
// op 2: out[N][K] = w[T][N]^T * in[T][K]  (weight gradient: in = x, w = dY, `out_row_stride` = row stride of dY),
//       bias != NULL: bias[N] = column sums of dY (bias gradient, written by the GEMM's epilogue)
// bf16 everywhere, fp32 accumulate; row strides in elements (multiples of 8), 16-byte aligned bases.
// Asynchronous on `stream`, never synchronises; runs the algorithm vil_gemm_tune selected for this problem, or the
// heuristic's first choice when the problem was never tuned.
extern "C" int vil_gemm_bf16(int op, const void* in, const void* w, const void* bias, void* out, int64_t T, int K, int N,
                             int64_t in_row_stride, int64_t out_row_stride, void* workspace, size_t workspace_bytes,
                             void* stream) {
  int e = check_gemm_args(op, in, w, bias, out, T, K, N, in_row_stride, out_row_stride, workspace);
  if (e) return e;
  hipStream_t s = (hipStream_t)stream;
  std::lock_guard<std::mutex> lock(g_mu);
  Plan* plp = nullptr;
  if ((e = get_plan(plp, op, T, K, N, in_row_stride, out_row_stride, bias != nullptr, workspace_bytes))) return e;
  Plan& pl = *plp;
  if (bias) hipblasLtMatmulDescSetAttribute(pl.md, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias));
  return run_plan(pl, pl.algo, op, in, w, out, workspace, workspace_bytes, s) == HIPBLAS_STATUS_SUCCESS ? VIL_OK : VIL_E_BACKEND;
}
