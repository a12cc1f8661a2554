// vil_sc2d_op.hip -- the reference's OPERATOR-level surface on HIP kernels: the three products of
// SlidingChunk2D (src/models/layers/slidingchunk_2d.py:26-200) on the reference's own chunked layouts, and
// mask_invalid_locations (:321-357).  These MATERIALISE the (BH, mx, my, W^2, kv) score tensor exactly like the
// reference does: they are the compatibility / parity surface (the reference's test protocol,
// src/tests/test_slidingchunk_2d.py, and the operator-level golden vectors run through them), not the hot path --
// the product's hot path is the fused kernels, which never build that tensor.
//
// Layouts (reference):  t_img  (BH, M, mx, my, W2)   q / k / v / gradient images, chunked
//                       attn   (BH, mx, my, W2, kv)  kv = 9 W2 (mode 0) | W2 (mode -1) | 2 W2 (mode 1..8: [self | nb])
// Neighbours are taken CYCLICALLY (torch.roll in the reference): wrap-around keys exist here and are removed (or
// kept, exact = -1) by the mask, as in the reference.
#include "vil_internal.h"

struct Sc2dParams {
  int BH, M, mx, my, W2, nact, kv;
  int adr[9], adc[9];
  int64_t n;          // output elements
};

static int sc2d_fill(Sc2dParams& p, int BH, int M, int mx, int my, int W, int mode) {
  if (BH <= 0 || M <= 0 || mx <= 0 || my <= 0 || W <= 0) return VIL_E_SHAPE;
  if (mode < -1 || mode > 8) return VIL_E_MODE;
  VilGeom g; vil_geom_init(g, mx * W, my * W, W, 0, mode);
  p.BH = BH; p.M = M; p.mx = mx; p.my = my; p.W2 = W * W; p.nact = g.nact; p.kv = g.nact * p.W2;
  for (int a = 0; a < 9; ++a) { p.adr[a] = g.adr[a]; p.adc[a] = g.adc[a]; }
  return VIL_OK;
}

__device__ __forceinline__ int wrap(int a, int n) { a %= n; return a < 0 ? a + n : a; }

// attn[b,m,n,l,a*W2+t] = sum_c q[b,c,m,n,l] * k[b,c,(m+dr_a) mod mx,(n+dc_a) mod my,t]      (slidingchunk_qk, :26-79)
// T = I/O element type (double, float, __bf16, _Float16), A = accumulation type (double for double, float otherwise: the
// reference's @autocast runs these einsums in fp16 with fp32 accumulation on GPU, slidingchunk_2d.py:203,235)
template <typename T, typename A>
__global__ void k_sc2d_qk(Sc2dParams p, const T* __restrict__ q, const T* __restrict__ k, T* __restrict__ attn) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.n) return;
  int64_t r = i;
  const int s = (int)(r % p.kv); r /= p.kv;
  const int l = (int)(r % p.W2); r /= p.W2;
  const int n = (int)(r % p.my); r /= p.my;
  const int m = (int)(r % p.mx);
  const int b = (int)(r / p.mx);
  const int a = s / p.W2, t = s - a * p.W2;
  const int m2 = wrap(m + p.adr[a], p.mx), n2 = wrap(n + p.adc[a], p.my);
  const int64_t cs = (int64_t)p.mx * p.my * p.W2;
  const T* qp = q + (int64_t)b * p.M * cs + ((int64_t)m * p.my + n) * p.W2 + l;
  const T* kp = k + (int64_t)b * p.M * cs + ((int64_t)m2 * p.my + n2) * p.W2 + t;
  A acc = 0;
  for (int c = 0; c < p.M; ++c) acc += (A)qp[c * cs] * (A)kp[c * cs];
  attn[i] = (T)acc;
}

// out[b,c,m,n,l] = sum_a sum_t attn[b,m,n,l,a*W2+t] * v[b,c,(m+dr_a),(n+dc_a),t]             (slidingchunk_av, :82-130)
template <typename T, typename A>
__global__ void k_sc2d_av(Sc2dParams p, const T* __restrict__ attn, const T* __restrict__ v, T* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.n) return;
  int64_t r = i;
  const int l = (int)(r % p.W2); r /= p.W2;
  const int n = (int)(r % p.my); r /= p.my;
  const int m = (int)(r % p.mx); r /= p.mx;
  const int c = (int)(r % p.M);
  const int b = (int)(r / p.M);
  const T* ap = attn + ((((int64_t)b * p.mx + m) * p.my + n) * p.W2 + l) * p.kv;
  A acc = 0;
  for (int a = 0; a < p.nact; ++a) {
    const int m2 = wrap(m + p.adr[a], p.mx), n2 = wrap(n + p.adc[a], p.my);
    const T* vp = v + ((((int64_t)b * p.M + c) * p.mx + m2) * p.my + n2) * p.W2;
    for (int t = 0; t < p.W2; ++t) acc += (A)ap[a * p.W2 + t] * (A)vp[t];
  }
  out[i] = (T)acc;
}

// grad_t2[b,c,m',n',t] = sum_a sum_l attn[b,m,n,l,a*W2+t] * g[b,c,m,n,l],  (m,n) = (m'-dr_a, n'-dc_a) cyclic
// (slidingchunk_agrad, :132-200: the einsum followed by the REVERSE roll)
template <typename T, typename A>
__global__ void k_sc2d_agrad(Sc2dParams p, const T* __restrict__ attn, const T* __restrict__ g, T* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.n) return;
  int64_t r = i;
  const int t = (int)(r % p.W2); r /= p.W2;
  const int n2 = (int)(r % p.my); r /= p.my;
  const int m2 = (int)(r % p.mx); r /= p.mx;
  const int c = (int)(r % p.M);
  const int b = (int)(r / p.M);
  A acc = 0;
  for (int a = 0; a < p.nact; ++a) {
    const int m = wrap(m2 - p.adr[a], p.mx), n = wrap(n2 - p.adc[a], p.my);
    const T* ap = attn + ((((int64_t)b * p.mx + m) * p.my + n) * p.W2) * p.kv + a * p.W2 + t;
    const T* gp = g + ((((int64_t)b * p.M + c) * p.mx + m) * p.my + n) * p.W2;
    for (int l = 0; l < p.W2; ++l) acc += (A)ap[(int64_t)l * p.kv] * (A)gp[l];
  }
  out[i] = (T)acc;
}

// mask_invalid_locations (:321-357): attn[b,m,n,l,s] = -inf where key slot s of chunk (m,n) is not attended by query l
// (zero / cyclic: independent of l; exact: the (2W+1)^2 window).  Counts the masked (m,n,l,s) of ONE image-head
// into *count (the reference's num_invalid).
template <typename T>
__global__ void k_sc2d_mask(Sc2dParams p, VilGeom g, T* __restrict__ attn, unsigned long long* count) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.n) return;
  int64_t r = i;
  const int s = (int)(r % p.kv); r /= p.kv;
  const int l = (int)(r % p.W2); r /= p.W2;
  const int n = (int)(r % p.my); r /= p.my;
  const int m = (int)(r % p.mx);
  const int b = (int)(r / p.mx);
  const int a = s / p.W2, t = s - a * p.W2, W = g.W;
  int kr, kc;
  int st = vil_key_state(g, m, n, g.adr[a], g.adc[a], t / W, t % W, kr, kc);
  if (st != VIL_KEY_MASKED && g.exact == 1 && !vil_exact_window(W, m * W + l / W, n * W + l % W, kr, kc)) st = VIL_KEY_MASKED;
  if (st == VIL_KEY_MASKED) {
    attn[i] = (T)(-__builtin_huge_valf());
    if (b == 0 && count) atomicAdd(count, 1ull);
  }
}

static int sc2d_dtype_ok(int dtype) {
  return dtype == VIL_DTYPE_F32 || dtype == VIL_DTYPE_F64 || dtype == VIL_DTYPE_BF16 || dtype == VIL_DTYPE_F16;
}
// one launch per I/O dtype
#define SC2D_DISPATCH(KERNEL, ...)                                                                                   \
  switch (dtype) {                                                                                                   \
    case VIL_DTYPE_F32: KERNEL<float, float><<<dim3(grid), dim3(256), 0, s>>>(__VA_ARGS__(float)); break;            \
    case VIL_DTYPE_F64: KERNEL<double, double><<<dim3(grid), dim3(256), 0, s>>>(__VA_ARGS__(double)); break;         \
    case VIL_DTYPE_BF16: KERNEL<__bf16, float><<<dim3(grid), dim3(256), 0, s>>>(__VA_ARGS__(__bf16)); break;         \
    default: KERNEL<_Float16, float><<<dim3(grid), dim3(256), 0, s>>>(__VA_ARGS__(_Float16)); break;                 \
  }

extern "C" int vil_sc2d_qk(const void* q, const void* k, void* attn, int BH, int M, int mx, int my, int W, int mode,
                           int dtype, void* stream) {
  if (!q || !k || !attn) return VIL_E_NULL;
  if (!sc2d_dtype_ok(dtype)) return VIL_E_DTYPE;
  Sc2dParams p; int e = sc2d_fill(p, BH, M, mx, my, W, mode);
  if (e) return e;
  p.n = (int64_t)BH * mx * my * p.W2 * p.kv;
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = (unsigned)((p.n + 255) / 256);
#define QK_ARGS(T_) p, (const T_*)q, (const T_*)k, (T_*)attn
  SC2D_DISPATCH(k_sc2d_qk, QK_ARGS)
  return (int)hipGetLastError();
}

extern "C" int vil_sc2d_av(const void* attn, const void* v, void* out, int BH, int M, int mx, int my, int W, int mode,
                           int dtype, void* stream) {
  if (!attn || !v || !out) return VIL_E_NULL;
  if (!sc2d_dtype_ok(dtype)) return VIL_E_DTYPE;
  Sc2dParams p; int e = sc2d_fill(p, BH, M, mx, my, W, mode);
  if (e) return e;
  p.n = (int64_t)BH * M * mx * my * p.W2;
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = (unsigned)((p.n + 255) / 256);
#define AV_ARGS(T_) p, (const T_*)attn, (const T_*)v, (T_*)out
  SC2D_DISPATCH(k_sc2d_av, AV_ARGS)
  return (int)hipGetLastError();
}

extern "C" int vil_sc2d_agrad(const void* attn, const void* grad, void* out, int BH, int M, int mx, int my, int W, int mode,
                              int dtype, void* stream) {
  if (!attn || !grad || !out) return VIL_E_NULL;
  if (!sc2d_dtype_ok(dtype)) return VIL_E_DTYPE;
  Sc2dParams p; int e = sc2d_fill(p, BH, M, mx, my, W, mode);
  if (e) return e;
  p.n = (int64_t)BH * M * mx * my * p.W2;
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = (unsigned)((p.n + 255) / 256);
#define AG_ARGS(T_) p, (const T_*)attn, (const T_*)grad, (T_*)out
  SC2D_DISPATCH(k_sc2d_agrad, AG_ARGS)
  return (int)hipGetLastError();
}

extern "C" int vil_sc2d_mask(void* attn, int BH, int mx, int my, int padx, int pady, int W, int exact, int mode, int dtype,
                             unsigned long long* count, void* stream) {
  if (!attn) return VIL_E_NULL;
  if (!sc2d_dtype_ok(dtype)) return VIL_E_DTYPE;
  if (exact < -1 || exact > 1 || (exact == 1 && mode != 0)) return VIL_E_EXACT;
  if (padx < 0 || pady < 0 || padx >= W || pady >= W) return VIL_E_SHAPE;
  Sc2dParams p; int e = sc2d_fill(p, BH, 1, mx, my, W, mode);
  if (e) return e;
  VilGeom g; vil_geom_init(g, mx * W - padx, my * W - pady, W, exact, mode);
  p.n = (int64_t)BH * mx * my * p.W2 * p.kv;
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = (unsigned)((p.n + 255) / 256);
  switch (dtype) {
    case VIL_DTYPE_F32: k_sc2d_mask<float><<<dim3(grid), dim3(256), 0, s>>>(p, g, (float*)attn, count); break;
    case VIL_DTYPE_F64: k_sc2d_mask<double><<<dim3(grid), dim3(256), 0, s>>>(p, g, (double*)attn, count); break;
    case VIL_DTYPE_BF16: k_sc2d_mask<__bf16><<<dim3(grid), dim3(256), 0, s>>>(p, g, (__bf16*)attn, count); break;
    default: k_sc2d_mask<_Float16><<<dim3(grid), dim3(256), 0, s>>>(p, g, (_Float16*)attn, count); break;
  }
  return (int)hipGetLastError();
}
