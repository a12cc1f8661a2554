// vil_optim.hip -- the optimizer step of the training loop as ONE multi-tensor launch:
// the reference's AdamW (src/optim/optimization.py:111-193) and QHM (src/optim/qhm.py:8-130) update rules on fp32
// master weights, reading the gradients in the dtype autograd produced them in (bf16 / fp16 / fp32) and writing the
// 16-bit working copy of every GEMM weight in the same pass (no gradient up-cast pass, no master -> working copy pass).
//
// HBM-bound streaming kernel: per element AdamW reads p, m, v (12 B) + g (2-4 B) and writes p, m, v (+2 B working
// copy); every access is a 16-byte (fp32) or 8-byte (16-bit) vector per lane, coalesced along the tensor.
//
// The reference's arithmetic, in its order (fp32 tensors, Python-double scalars rounded to fp32 where PyTorch does):
//   AdamW  m = m*b1 + g*(1-b1);  v = v*b2 + g*g*(1-b2);  denom = sqrt(v) + eps            [eps OUTSIDE the bias correction]
//          step_size = lr * sqrt(1 - b2^t) / (1 - b1^t)    (double)                         [correct_bias]
//          p = p + (-step_size) * m / denom;   then  p = p + (-lr*wd) * p                  [decay AFTER the Adam update]
//   QHM    g' = g + wd*p;  h = h*beta + g'*(1-beta);  d = g'*(1-nu) + h*nu  (nu == 1: d = h; beta == 0 or nu == 0: d = g')
//          p = p + (-lr) * d
#include "vil_internal.h"
#include <string.h>

#define OPT_THREADS 256
#define OPT_VEC 4
#define OPT_ITERS 4
#define OPT_BLOCK_ELEMS (OPT_THREADS * OPT_VEC * OPT_ITERS)

struct OptPlanHeader { int32_t ntensors, nblocks, pad0, pad1; };
struct OptBlock { int32_t tensor, pad; int64_t off; };

struct OptHyper {
  float b1, b2, eps; int correct_bias;     // AdamW
  float momentum, nu;                      // QHM
  int32_t* step_words;                     // [0] completed steps t, [1] arrival ticket of the running launch
  // loss scaling (fp16 training with torch.amp.GradScaler, reference src/engine.py:84-100): device scalars, or null
  const float* inv_scale;                  // gradients are multiplied by *inv_scale (1 / loss scale) on load
  const float* found_inf;                  // != 0: a gradient was non-finite -- the step is skipped, the step count kept
};

typedef float of32x4 __attribute__((ext_vector_type(4)));
typedef uint16_t ou16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float opt_h2f(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ uint16_t opt_f2h(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }

__device__ __forceinline__ float load_grad(const void* g, int dt, int64_t i) {
  if (dt == VIL_DTYPE_F32) return ((const float*)g)[i];
  const uint16_t h = ((const uint16_t*)g)[i];
  return dt == VIL_DTYPE_BF16 ? vil_bf2f(h) : opt_h2f(h);
}
__device__ __forceinline__ of32x4 load_grad4(const void* g, int dt, int64_t i) {
  if (dt == VIL_DTYPE_F32) return *(const of32x4*)((const float*)g + i);
  const ou16x4 h = *(const ou16x4*)((const uint16_t*)g + i);
  of32x4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) r[e] = dt == VIL_DTYPE_BF16 ? vil_bf2f(h[e]) : opt_h2f(h[e]);
  return r;
}
__device__ __forceinline__ void store_low(void* low, int dt, int64_t i, float v) {
  ((uint16_t*)low)[i] = dt == VIL_DTYPE_BF16 ? vil_f2bf(v) : opt_f2h(v);
}
__device__ __forceinline__ void store_low4(void* low, int dt, int64_t i, of32x4 v) {
  ou16x4 h;
#pragma unroll
  for (int e = 0; e < 4; ++e) h[e] = dt == VIL_DTYPE_BF16 ? vil_f2bf(v[e]) : opt_f2h(v[e]);
  *(ou16x4*)((uint16_t*)low + i) = h;
}

// ALGO 0: AdamW, 1: QHM.  Written so that no fused multiply-add contracts two reference operations into one rounding
// (#pragma clang fp contract(off)): the fixtures of tests/golden/optim_reference.npz come from the imported reference
// optimizers on CPU, whose tensor ops round after every operation.
template <int ALGO>
__global__ __launch_bounds__(OPT_THREADS) void k_optim(const char* plan, OptHyper hp) {
#pragma clang fp contract(off)
  const OptPlanHeader* hd = (const OptPlanHeader*)plan;
  const VilOptimTensor* tens = (const VilOptimTensor*)(plan + sizeof(OptPlanHeader));
  const OptBlock* blocks = (const OptBlock*)(tens + hd->ntensors);
  const OptBlock blk = blocks[blockIdx.x];
  const VilOptimTensor t = tens[blk.tensor];
  const float lr = t.lr_dev ? *t.lr_dev : t.lr;
  // GradScaler semantics on the device (no host synchronisation, so the step stays capturable): a non-finite gradient
  // anywhere skips the whole step -- parameters, moments, working copies and the step counter keep their values
  const bool skip = hp.found_inf != nullptr && *hp.found_inf != 0.f;
  const bool unscale = hp.inv_scale != nullptr;
  const float gscale = unscale ? *hp.inv_scale : 1.f;
  // scalars of this step, in double like the reference's Python arithmetic, rounded once to fp32 (what
  // addcdiv_(value=...) / add_(alpha=...) do with a Python float)
  float neg_step = 0.f, neg_lrwd = 0.f;
  if (ALGO == 0) {
    __shared__ float sh_step;
    if (threadIdx.x == 0) {                       // two double-precision pow() per workgroup, not per thread
      const int tstep = hp.step_words[0] + 1;
      double step_size = (double)lr;
      if (hp.correct_bias) {
        const double bc1 = 1.0 - pow((double)hp.b1, (double)tstep);
        const double bc2 = 1.0 - pow((double)hp.b2, (double)tstep);
        step_size = step_size * sqrt(bc2) / bc1;
      }
      sh_step = (float)(-step_size);
    }
    __syncthreads();
    neg_step = sh_step;
    neg_lrwd = (float)(-(double)lr * (double)t.weight_decay);
  }
  const float b1 = ALGO == 0 ? hp.b1 : hp.momentum;
  const float omb1 = (float)(1.0 - (double)b1);                 // `alpha=1.0 - beta1` is a Python double, then fp32
  const float omb2 = (float)(1.0 - (double)hp.b2);
  const float omnu = (float)(1.0 - (double)hp.nu);
  const bool qhm_plain = ALGO == 1 && (fabsf(hp.momentum) < 1e-12f || fabsf(hp.nu) < 1e-12f);
  const bool qhm_nu1 = ALGO == 1 && fabsf(hp.nu - 1.0f) < 1e-12f;
  const float neg_lr = -lr;

  float* P = (float*)t.param;
  float* S1 = (float*)t.state1;
  float* S2 = (float*)t.state2;
  auto update = [&](float p, float g, float& s1, float& s2) -> float {
    if (unscale) g = g * gscale;               // (no multiply without a scaler: the fixtures of the unscaled step stay bit-exact)
    if (ALGO == 0) {
      s1 = s1 * b1 + g * omb1;
      s2 = s2 * hp.b2 + (g * g) * omb2;          // addcmul_(grad, grad, value): value * (t1 * t2)
      const float denom = __builtin_sqrtf(s2) + hp.eps;
      p = p + (neg_step * s1) / denom;           // addcdiv_: self + value * t1 / t2
      if (t.weight_decay > 0.f) p = p + neg_lrwd * p;
      return p;
    } else {
      if (t.weight_decay > 0.f) g = g + t.weight_decay * p;
      float d = g;
      if (!qhm_plain) {
        s1 = s1 * b1 + g * omb1;
        d = qhm_nu1 ? s1 : g * omnu + s1 * hp.nu;
      }
      return p + neg_lr * d;
    }
  };
  const int64_t base = blk.off;
  const int64_t end = base + OPT_BLOCK_ELEMS < t.n ? base + OPT_BLOCK_ELEMS : t.n;
  const bool vec_ok = ((uintptr_t)t.param | (uintptr_t)t.state1 | (uintptr_t)(ALGO == 0 ? t.state2 : t.state1)) % 16 == 0 &&
                      (uintptr_t)t.grad % (t.grad_dtype == VIL_DTYPE_F32 ? 16 : 8) == 0 &&
                      (!t.low || (uintptr_t)t.low % 8 == 0);
  if (skip) {
    // nothing to update
  } else if (vec_ok && base + OPT_BLOCK_ELEMS <= t.n) {
    // full block: all 16-byte loads of the 4 iterations in flight before the first update (the kernel is a stream of
    // ~30 bytes per element; issued one iteration at a time it ran at 3.3 TB/s)
    of32x4 p[OPT_ITERS], s1[OPT_ITERS], s2[OPT_ITERS], g[OPT_ITERS];
#pragma unroll
    for (int it = 0; it < OPT_ITERS; ++it) {
      const int64_t i = base + ((int64_t)it * OPT_THREADS + threadIdx.x) * OPT_VEC;
      p[it] = *(const of32x4*)(P + i);
      s1[it] = (ALGO == 1 && qhm_plain) ? (of32x4){0.f, 0.f, 0.f, 0.f} : *(const of32x4*)(S1 + i);
      s2[it] = ALGO == 0 ? *(const of32x4*)(S2 + i) : (of32x4){0.f, 0.f, 0.f, 0.f};
      g[it] = load_grad4(t.grad, t.grad_dtype, i);
    }
#pragma unroll
    for (int it = 0; it < OPT_ITERS; ++it) {
      const int64_t i = base + ((int64_t)it * OPT_THREADS + threadIdx.x) * OPT_VEC;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = s1[it][e], b = s2[it][e];
        p[it][e] = update(p[it][e], g[it][e], a, b);
        s1[it][e] = a; s2[it][e] = b;
      }
      *(of32x4*)(P + i) = p[it];
      if (!(ALGO == 1 && qhm_plain)) *(of32x4*)(S1 + i) = s1[it];
      if (ALGO == 0) *(of32x4*)(S2 + i) = s2[it];
      if (t.low) store_low4(t.low, t.low_dtype, i, p[it]);
    }
  } else {
    for (int it = 0; it < OPT_ITERS; ++it) {
      const int64_t i = base + ((int64_t)it * OPT_THREADS + threadIdx.x) * OPT_VEC;
      if (i >= end) break;
      for (int64_t j = i; j < i + OPT_VEC && j < end; ++j) {
        float a = (ALGO == 1 && qhm_plain) ? 0.f : S1[j], b = ALGO == 0 ? S2[j] : 0.f;
        const float pj = update(P[j], load_grad(t.grad, t.grad_dtype, j), a, b);
        P[j] = pj;
        if (!(ALGO == 1 && qhm_plain)) S1[j] = a;
        if (ALGO == 0) S2[j] = b;
        if (t.low) store_low(t.low, t.low_dtype, j, pj);
      }
    }
  }
  // The LAST workgroup to finish advances the step counter: every workgroup has read it by then (its ticket add is
  // issued after its read of the counter returned -- the value feeds the step size).  Relaxed atomics only: an
  // agent-scope release here would write back the XCD's L2 once per workgroup; the next launch sees the counter
  // through the kernel boundary.
  __syncthreads();
  if (threadIdx.x == 0) {
    const int prev = __hip_atomic_fetch_add(&hp.step_words[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == (int)gridDim.x - 1) {
      __hip_atomic_store(&hp.step_words[1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (!skip) __hip_atomic_fetch_add(&hp.step_words[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ------------------------------------------------------------------ host side (C ABI, include/vil_attn.h)
static int64_t opt_nblocks(const VilOptimTensor* t, int n) {
  int64_t nb = 0;
  for (int i = 0; i < n; ++i) nb += (t[i].n + OPT_BLOCK_ELEMS - 1) / OPT_BLOCK_ELEMS;
  return nb;
}

extern "C" size_t vil_optim_plan_bytes(const VilOptimTensor* tensors, int ntensors) {
  if (!tensors || ntensors <= 0) return 0;
  return sizeof(OptPlanHeader) + (size_t)ntensors * sizeof(VilOptimTensor) + (size_t)opt_nblocks(tensors, ntensors) * sizeof(OptBlock);
}

extern "C" int vil_optim_plan_build(const VilOptimTensor* tensors, int ntensors, void* host_plan, size_t bytes, int* nblocks) {
  if (!tensors || !host_plan || !nblocks) return VIL_E_NULL;
  if (ntensors <= 0) return VIL_E_SHAPE;
  if (bytes < vil_optim_plan_bytes(tensors, ntensors)) return VIL_E_WORKSPACE;
  for (int i = 0; i < ntensors; ++i) {
    const VilOptimTensor& t = tensors[i];
    if (!t.param || !t.grad || !t.state1) return VIL_E_NULL;
    if (t.n <= 0) return VIL_E_SHAPE;
    if (t.grad_dtype != VIL_DTYPE_F32 && t.grad_dtype != VIL_DTYPE_BF16 && t.grad_dtype != VIL_DTYPE_F16) return VIL_E_DTYPE;
    if (t.low && t.low_dtype != VIL_DTYPE_BF16 && t.low_dtype != VIL_DTYPE_F16) return VIL_E_DTYPE;
  }
  const int64_t nb = opt_nblocks(tensors, ntensors);
  if (nb >= (1ll << 31)) return VIL_E_SHAPE;
  char* out = (char*)host_plan;
  OptPlanHeader hd = {ntensors, (int32_t)nb, 0, 0};
  memcpy(out, &hd, sizeof(hd));
  memcpy(out + sizeof(hd), tensors, (size_t)ntensors * sizeof(VilOptimTensor));
  OptBlock* blocks = (OptBlock*)(out + sizeof(hd) + (size_t)ntensors * sizeof(VilOptimTensor));
  int64_t k = 0;
  for (int i = 0; i < ntensors; ++i)
    for (int64_t off = 0; off < tensors[i].n; off += OPT_BLOCK_ELEMS) blocks[k++] = OptBlock{i, 0, off};
  *nblocks = (int)nb;
  return VIL_OK;
}

static int optim_launch(int algo, const void* plan_dev, int nblocks, const OptHyper& hp, void* stream) {
  if (!plan_dev || !hp.step_words) return VIL_E_NULL;
  if (nblocks <= 0) return VIL_E_SHAPE;
  if ((uintptr_t)plan_dev & 15) return VIL_E_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  if (algo == 0) k_optim<0><<<dim3((unsigned)nblocks), dim3(OPT_THREADS), 0, s>>>((const char*)plan_dev, hp);
  else k_optim<1><<<dim3((unsigned)nblocks), dim3(OPT_THREADS), 0, s>>>((const char*)plan_dev, hp);
  return (int)hipGetLastError();
}

extern "C" int vil_optim_adamw_step_amp(const void* plan_dev, int nblocks, float beta1, float beta2, float eps, int correct_bias,
                                        int32_t* step_words, const float* inv_scale, const float* found_inf, void* stream) {
  if (!(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f) || !(eps >= 0.f)) return VIL_E_SHAPE;
  OptHyper hp = {};
  hp.b1 = beta1; hp.b2 = beta2; hp.eps = eps; hp.correct_bias = correct_bias;
  hp.step_words = step_words; hp.inv_scale = inv_scale; hp.found_inf = found_inf;
  return optim_launch(0, plan_dev, nblocks, hp, stream);
}

extern "C" int vil_optim_adamw_step(const void* plan_dev, int nblocks, float beta1, float beta2, float eps, int correct_bias,
                                    int32_t* step_words, void* stream) {
  return vil_optim_adamw_step_amp(plan_dev, nblocks, beta1, beta2, eps, correct_bias, step_words, nullptr, nullptr, stream);
}

extern "C" int vil_optim_qhm_step_amp(const void* plan_dev, int nblocks, float momentum, float nu, int32_t* step_words,
                                      const float* inv_scale, const float* found_inf, void* stream) {
  if (!(momentum >= 0.f && momentum <= 1.f)) return VIL_E_SHAPE;
  OptHyper hp = {};
  hp.momentum = momentum; hp.nu = nu; hp.step_words = step_words; hp.inv_scale = inv_scale; hp.found_inf = found_inf;
  return optim_launch(1, plan_dev, nblocks, hp, stream);
}

extern "C" int vil_optim_qhm_step(const void* plan_dev, int nblocks, float momentum, float nu, int32_t* step_words, void* stream) {
  return vil_optim_qhm_step_amp(plan_dev, nblocks, momentum, nu, step_words, nullptr, nullptr, stream);
}
