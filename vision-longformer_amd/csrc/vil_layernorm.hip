// vil_layernorm.hip -- fused LayerNorm forward / backward for the blocks either side of the
// hot path ("block glue": `x + drop_path(attn(norm(x), nx, ny))`,
// reference src/models/msvit.py:313-316,336-340).  HBM-bound row kernels:
//   forward  y = (x - mean) * rstd * gamma + beta, x fp32 or bf16, y written directly in the dtype
//            the consumer wants (bf16 for the q/kv/fc1 GEMMs under autocast) -> no separate cast
//            kernels, half the write traffic; mean / rstd kept in fp32 for the backward.
//   backward dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma;
//            dgamma/dbeta: per-lane register partials over a grid-stride row loop, reduced across
//            the wave by shuffles and across workgroups by a second, tiny kernel (deterministic).
// Mapping: LPR lanes per row (16/32/64), 8 contiguous channels per lane and iteration (16-byte bf16 /
// 2 x 16-byte fp32 accesses), 64/LPR rows per wavefront, butterfly reductions inside the LPR lanes.
#include "vil_internal.h"
#include <string.h>

typedef float f32x4_ __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));

template <typename T> struct LNIO;
template <> struct LNIO<float> {
  static __device__ __forceinline__ void ld8(const float* p, float (&v)[8]) {
    const f32x4_ a = *(const f32x4_*)p, b = *(const f32x4_*)(p + 4);
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
  }
  static __device__ __forceinline__ void st8(float* p, const float (&v)[8]) {
    *(f32x4_*)p = (f32x4_){v[0], v[1], v[2], v[3]};
    *(f32x4_*)(p + 4) = (f32x4_){v[4], v[5], v[6], v[7]};
  }
};
template <> struct LNIO<vil_bf16> {
  static __device__ __forceinline__ void ld8(const vil_bf16* p, float (&v)[8]) {
    const u32x4_ a = *(const u32x4_*)p;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = __uint_as_float(a[i] << 16);
      v[2 * i + 1] = __uint_as_float(a[i] & 0xffff0000u);
    }
  }
  static __device__ __forceinline__ void st8(vil_bf16* p, const float (&v)[8]) {
    typedef __bf16 bf16x8_ __attribute__((ext_vector_type(8)));
    bf16x8_ o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (__bf16)v[i];       // v_cvt_pk_bf16_f32, round-to-nearest-even
    *(bf16x8_*)p = o;
  }
};

template <int LPR>
__device__ __forceinline__ float row_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

struct LNParams {
  const void* x; const void* dy; void* y; void* dx;
  const float* gamma; const float* beta;
  float* mean; float* rstd;
  float* parts;            // backward: (nblocks, 2, C)
  float* dgamma; float* dbeta;
  int64_t rows, x_rs, y_rs, dy_rs, dx_rs;
  int C, nblocks;
  float eps;
  // fused residual (vil_resln_*): forward x' = x + rscale[row / rps] * res, y = LN(x'), x' written to xout;
  // backward dx = gres + LNbwd(dy) and gbranch = rscale[row / rps] * dx in the branch's dtype
  const void* res; int res_bf16; int64_t res_rs;
  const float* rscale; int64_t rps;
  unsigned rps_magic;        // ceil(2^32 / rps) when row / rps == umulhi(row, magic) for every row of the call, else 0
  float* xout; int64_t xout_rs;
  const float* gres; int64_t gres_rs;
  void* gbranch; int gb_bf16; int64_t gb_rs;
  // token layout of the normalised side (vil_layernorm_*_tokens): the forward's y / the backward's dy hold tok_gap
  // extra rows (the global tokens) in front of every sample's tok_rps rows: row r lives at r + (r / tok_rps + 1) * tok_gap
  int64_t tok_rps, tok_gap;
  unsigned tok_magic;
};
__device__ __forceinline__ int64_t ln_tok_row(const LNParams& p, int64_t row) {
  if (!p.tok_gap) return row;
  const int64_t q = p.tok_magic ? (int64_t)__umulhi((unsigned)row, p.tok_magic) : row / p.tok_rps;
  return row + (q + 1) * p.tok_gap;
}

template <typename TI, typename TO, int LPR, int NIT>
__global__ __launch_bounds__(256) void k_ln_fwd(LNParams p) {
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, sub = lane % LPR, slot = lane / LPR;
  const int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + slot;
  const bool rok = row < p.rows;
  float v[NIT][8];
  float s = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int e0 = (it * LPR + sub) * 8;
    if (rok && e0 < p.C) {
      LNIO<TI>::ld8((const TI*)p.x + row * p.x_rs + e0, v[it]);
      if (p.res) {                     // fused residual add (stochastic-depth scale per sample)
        float r[8];
        if (p.res_bf16) LNIO<vil_bf16>::ld8((const vil_bf16*)p.res + row * p.res_rs + e0, r);
        else LNIO<float>::ld8((const float*)p.res + row * p.res_rs + e0, r);
        const float sc = p.rscale ? p.rscale[p.rps_magic ? (int64_t)__umulhi((unsigned)row, p.rps_magic) : row / p.rps] : 1.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[it][i] = fmaf(sc, r[i], v[it][i]);
        LNIO<float>::st8(p.xout + row * p.xout_rs + e0, v[it]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[it][i] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[it][i];
  }
  const float mean = row_sum<LPR>(s) / p.C;
  float q = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int e0 = (it * LPR + sub) * 8;
    if (e0 < p.C) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { const float d = v[it][i] - mean; q = fmaf(d, d, q); }
    }
  }
  const float rstd = rsqrtf(row_sum<LPR>(q) / p.C + p.eps);
  if (!rok) return;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int e0 = (it * LPR + sub) * 8;
    if (e0 < p.C) {
      float gm[8], bt[8], o[8];
      LNIO<float>::ld8(p.gamma + e0, gm);
      LNIO<float>::ld8(p.beta + e0, bt);
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = fmaf((v[it][i] - mean) * rstd, gm[i], bt[i]);
      LNIO<TO>::st8((TO*)p.y + ln_tok_row(p, row) * p.y_rs + e0, o);
    }
  }
  if (sub == 0) { p.mean[row] = mean; p.rstd[row] = rstd; }
}

template <typename TI, typename TG, typename TO, int LPR, int NIT>
__global__ __launch_bounds__(256) void k_ln_bwd(LNParams p) {
  constexpr int RPW = 64 / LPR;
  __shared__ float red[4][2][NIT * LPR * 8];
  const int lane = threadIdx.x & 63, sub = lane % LPR, slot = lane / LPR, wave = threadIdx.x >> 6;
  float dg[NIT][8], db[NIT][8], gm[NIT][8];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int e0 = (it * LPR + sub) * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) { dg[it][i] = 0.f; db[it][i] = 0.f; gm[it][i] = 0.f; }
    if (e0 < p.C) LNIO<float>::ld8(p.gamma + e0, gm[it]);
  }
  const int64_t rows_per_pass = (int64_t)gridDim.x * 4 * RPW;
  for (int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * RPW; r0 < p.rows; r0 += rows_per_pass) {
    const int64_t row = r0 + slot;
    const bool rok = row < p.rows;
    const float mean = rok ? p.mean[row] : 0.f, rstd = rok ? p.rstd[row] : 0.f;
    float xh[NIT][8], g[NIT][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e0 = (it * LPR + sub) * 8;
      float xv[8], dyv[8];
      if (rok && e0 < p.C) {
        LNIO<TI>::ld8((const TI*)p.x + row * p.x_rs + e0, xv);
        LNIO<TG>::ld8((const TG*)p.dy + ln_tok_row(p, row) * p.dy_rs + e0, dyv);
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) { xv[i] = mean; dyv[i] = 0.f; }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        xh[it][i] = (xv[i] - mean) * rstd;
        g[it][i] = dyv[i] * gm[it][i];
        s1 += g[it][i];
        s2 = fmaf(g[it][i], xh[it][i], s2);
        dg[it][i] = fmaf(dyv[i], xh[it][i], dg[it][i]);
        db[it][i] += dyv[i];
      }
    }
    const float m1 = row_sum<LPR>(s1) / p.C, m2 = row_sum<LPR>(s2) / p.C;
    if (rok) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int e0 = (it * LPR + sub) * 8;
        if (e0 < p.C) {
          float o[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] = rstd * (g[it][i] - m1 - xh[it][i] * m2);
          if (p.gres) {                // + the gradient arriving on the residual stream
            float r[8];
            LNIO<float>::ld8(p.gres + row * p.gres_rs + e0, r);
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] += r[i];
          }
          LNIO<TO>::st8((TO*)p.dx + row * p.dx_rs + e0, o);
          if (p.gbranch) {             // gradient of the branch that was added in the forward
            const float sc = p.rscale ? p.rscale[p.rps_magic ? (int64_t)__umulhi((unsigned)row, p.rps_magic) : row / p.rps] : 1.0f;
            float b[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) b[i] = o[i] * sc;
            if (p.gb_bf16) LNIO<vil_bf16>::st8((vil_bf16*)p.gbranch + row * p.gb_rs + e0, b);
            else LNIO<float>::st8((float*)p.gbranch + row * p.gb_rs + e0, b);
          }
        }
      }
    }
  }
  // reduce the RPW row slots of the wave, then the 4 waves of the block, then hand the block's
  // partial to the second kernel
#pragma unroll
  for (int it = 0; it < NIT; ++it)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1) {
        dg[it][i] += __shfl_xor(dg[it][i], o, 64);
        db[it][i] += __shfl_xor(db[it][i], o, 64);
      }
      if (slot == 0) {
        red[wave][0][(it * LPR + sub) * 8 + i] = dg[it][i];
        red[wave][1][(it * LPR + sub) * 8 + i] = db[it][i];
      }
    }
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * p.C; e += 256) {
    const int w = e / p.C, c = e % p.C;
    p.parts[((int64_t)blockIdx.x * 2 + w) * p.C + c] = red[0][w][c] + red[1][w][c] + red[2][w][c] + red[3][w][c];
  }
}

// block = 64 columns x 16 partial groups: every thread sums nblocks/16 partials, LDS tree over the groups
__global__ __launch_bounds__(1024) void k_ln_reduce(LNParams p) {
  __shared__ float red[16][64];
  const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + col;
  float s = 0.f;
  if (e < 2 * p.C) {
    const int w = e / p.C, c = e % p.C;
    const float* src = p.parts + (int64_t)w * p.C + c;
    const int64_t st = 2 * (int64_t)p.C;
    int b = grp;
    // eight independent loads in flight per thread (round 4: with four, the 64 partials a thread sums were 16 dependent
    // L2 round trips -- 7.6 us for 3 MB, 29 of these launches per ViL-Small step; with eight 5.6 us, sixteen: no better)
    for (; b + 112 < p.nblocks; b += 128)
      s += ((src[b * st] + src[(b + 16) * st]) + (src[(b + 32) * st] + src[(b + 48) * st])) +
           ((src[(b + 64) * st] + src[(b + 80) * st]) + (src[(b + 96) * st] + src[(b + 112) * st]));
    for (; b + 48 < p.nblocks; b += 64)
      s += (src[b * st] + src[(b + 16) * st]) + (src[(b + 32) * st] + src[(b + 48) * st]);
    for (; b < p.nblocks; b += 16) s += src[b * st];
  }
  red[grp][col] = s;
  __syncthreads();
  if (grp == 0 && e < 2 * p.C) {
#pragma unroll
    for (int g2 = 1; g2 < 16; ++g2) s += red[g2][col];
    const int w = e / p.C, c = e % p.C;
    (w == 0 ? p.dgamma : p.dbeta)[c] = s;
  }
}

// ------------------------------------------------------------------ C ABI
static int ln_check(int64_t rows, int C, int64_t s0, int64_t s1) {
  if (rows <= 0 || C <= 0) return VIL_E_SHAPE;
  if (C % 8 || C > 1024) return VIL_E_HEAD_DIM;
  if ((s0 | s1) & 7) return VIL_E_ALIGN;
  return VIL_OK;
}
static int ln_blocks(int64_t rows, int rpw) {
  int64_t need = (rows + 4 * rpw - 1) / (4 * rpw);
  return (int)(need < 1024 ? need : 1024);
}

extern "C" size_t vil_layernorm_workspace_bytes(int64_t rows, int C) {
  return (size_t)1024 * 2 * (size_t)C * sizeof(float);
}

#define LN_SHAPE_SWITCH(C_, ...)                                              \
  if (C_ <= 128) { constexpr int LPR = 16, NIT = 1; __VA_ARGS__; }            \
  else if (C_ <= 256) { constexpr int LPR = 32, NIT = 1; __VA_ARGS__; }       \
  else if (C_ <= 512) { constexpr int LPR = 64, NIT = 1; __VA_ARGS__; }       \
  else { constexpr int LPR = 64, NIT = 2; __VA_ARGS__; }

static int ln_fwd_launch(LNParams& p, int x_dtype, int y_dtype, hipStream_t s);
static int ln_bwd_launch(LNParams& p, int x_dtype, int dy_dtype, hipStream_t s);
static unsigned ln_rps_magic(int64_t rows, int64_t rps);

extern "C" int vil_layernorm_fwd(const void* x, int x_dtype, const float* gamma, const float* beta,
                                 void* y, int y_dtype, float* mean, float* rstd, int64_t rows, int C,
                                 int64_t x_row_stride, int64_t y_row_stride, float eps, void* stream) {
  if (!x || !gamma || !beta || !y || !mean || !rstd) return VIL_E_NULL;
  int e = ln_check(rows, C, x_row_stride, y_row_stride);
  if (e) return e;
  if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15) return VIL_E_ALIGN;
  LNParams p; memset(&p, 0, sizeof(p));
  p.x = x; p.y = y; p.gamma = gamma; p.beta = beta; p.mean = mean; p.rstd = rstd;
  p.rows = rows; p.C = C; p.x_rs = x_row_stride; p.y_rs = y_row_stride; p.eps = eps;
  return ln_fwd_launch(p, x_dtype, y_dtype, (hipStream_t)stream);
}

static int ln_fwd_launch(LNParams& p, int x_dtype, int y_dtype, hipStream_t s) {
  const int64_t rows = p.rows; const int C = p.C;
  const int key = x_dtype * 2 + y_dtype;
  LN_SHAPE_SWITCH(C, {
    const unsigned grid = (unsigned)((rows + 4 * (64 / LPR) - 1) / (4 * (64 / LPR)));
    switch (key) {
      case 0: k_ln_fwd<float, float, LPR, NIT><<<dim3(grid), dim3(256), 0, s>>>(p); break;
      case 1: k_ln_fwd<float, vil_bf16, LPR, NIT><<<dim3(grid), dim3(256), 0, s>>>(p); break;
      case 2: k_ln_fwd<vil_bf16, float, LPR, NIT><<<dim3(grid), dim3(256), 0, s>>>(p); break;
      case 3: k_ln_fwd<vil_bf16, vil_bf16, LPR, NIT><<<dim3(grid), dim3(256), 0, s>>>(p); break;
      default: return VIL_E_DTYPE;
    }
  });
  return (int)hipGetLastError();
}

extern "C" int vil_layernorm_bwd(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* gamma,
                                 const float* mean, const float* rstd, void* dx, int dx_dtype,
                                 float* dgamma, float* dbeta, void* workspace, int64_t rows, int C,
                                 int64_t dy_row_stride, int64_t x_row_stride, int64_t dx_row_stride, void* stream) {
  if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || !workspace) return VIL_E_NULL;
  int e = ln_check(rows, C, dy_row_stride | x_row_stride, dx_row_stride);
  if (e) return e;
  if (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)gamma) & 15) return VIL_E_ALIGN;
  LNParams p; memset(&p, 0, sizeof(p));
  p.dy = dy; p.x = x; p.gamma = gamma; p.mean = (float*)mean; p.rstd = (float*)rstd; p.dx = dx;
  p.dgamma = dgamma; p.dbeta = dbeta; p.parts = (float*)workspace;
  p.rows = rows; p.C = C; p.dy_rs = dy_row_stride; p.x_rs = x_row_stride; p.dx_rs = dx_row_stride;
  if (dx_dtype != x_dtype) return VIL_E_DTYPE;       // dx has the dtype of x
  return ln_bwd_launch(p, x_dtype, dy_dtype, (hipStream_t)stream);
}

static int ln_bwd_launch(LNParams& p, int x_dtype, int dy_dtype, hipStream_t s) {
  const int64_t rows = p.rows; const int C = p.C;
  int e;
  const int key = x_dtype * 2 + dy_dtype;
  LN_SHAPE_SWITCH(C, {
    p.nblocks = ln_blocks(rows, 64 / LPR);
    switch (key) {
      case 0: k_ln_bwd<float, float, float, LPR, NIT><<<dim3(p.nblocks), dim3(256), 0, s>>>(p); break;
      case 1: k_ln_bwd<float, vil_bf16, float, LPR, NIT><<<dim3(p.nblocks), dim3(256), 0, s>>>(p); break;
      case 2: k_ln_bwd<vil_bf16, float, vil_bf16, LPR, NIT><<<dim3(p.nblocks), dim3(256), 0, s>>>(p); break;
      case 3: k_ln_bwd<vil_bf16, vil_bf16, vil_bf16, LPR, NIT><<<dim3(p.nblocks), dim3(256), 0, s>>>(p); break;
      default: return VIL_E_DTYPE;
    }
  });
  e = (int)hipGetLastError();
  if (e) return e;
  k_ln_reduce<<<dim3((2 * C + 63) / 64), dim3(1024), 0, s>>>(p);
  return (int)hipGetLastError();
}

// ---- LayerNorm of the patch embedding written straight into the stage's token tensor (B, G + N, C): the forward's
// output rows / the backward's dy rows skip the G global-token rows in front of every sample (the reference
// concatenates: torch.cat((cls_tokens, x), dim=1), msvit.py:204-206 -- a full copy each way)
extern "C" int vil_layernorm_fwd_tokens(const void* x, int x_dtype, const float* gamma, const float* beta,
                                        void* y_tokens, int y_dtype, float* mean, float* rstd, int64_t rows, int C,
                                        int64_t x_row_stride, float eps, int64_t rows_per_sample, int64_t gap_rows,
                                        void* stream) {
  if (!x || !gamma || !beta || !y_tokens || !mean || !rstd) return VIL_E_NULL;
  int e = ln_check(rows, C, x_row_stride, C);
  if (e) return e;
  if (rows_per_sample <= 0 || gap_rows < 0 || rows % rows_per_sample) return VIL_E_SHAPE;
  if (((uintptr_t)x | (uintptr_t)y_tokens | (uintptr_t)gamma | (uintptr_t)beta) & 15) return VIL_E_ALIGN;
  LNParams p; memset(&p, 0, sizeof(p));
  p.x = x; p.y = y_tokens; p.gamma = gamma; p.beta = beta; p.mean = mean; p.rstd = rstd;
  p.rows = rows; p.C = C; p.x_rs = x_row_stride; p.y_rs = C; p.eps = eps;
  p.tok_rps = rows_per_sample; p.tok_gap = gap_rows; p.tok_magic = ln_rps_magic(rows, rows_per_sample);
  return ln_fwd_launch(p, x_dtype, y_dtype, (hipStream_t)stream);
}

extern "C" int vil_layernorm_bwd_tokens(const void* dy_tokens, int dy_dtype, const void* x, int x_dtype, const float* gamma,
                                        const float* mean, const float* rstd, void* dx, int dx_dtype,
                                        float* dgamma, float* dbeta, void* workspace, int64_t rows, int C,
                                        int64_t x_row_stride, int64_t dx_row_stride, int64_t rows_per_sample,
                                        int64_t gap_rows, void* stream) {
  if (!dy_tokens || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || !workspace) return VIL_E_NULL;
  int e = ln_check(rows, C, x_row_stride, dx_row_stride);
  if (e) return e;
  if (rows_per_sample <= 0 || gap_rows < 0 || rows % rows_per_sample) return VIL_E_SHAPE;
  if (((uintptr_t)x | (uintptr_t)dy_tokens | (uintptr_t)dx | (uintptr_t)gamma) & 15) return VIL_E_ALIGN;
  if (dx_dtype != x_dtype) return VIL_E_DTYPE;
  LNParams p; memset(&p, 0, sizeof(p));
  p.dy = dy_tokens; p.x = x; p.gamma = gamma; p.mean = (float*)mean; p.rstd = (float*)rstd; p.dx = dx;
  p.dgamma = dgamma; p.dbeta = dbeta; p.parts = (float*)workspace;
  p.rows = rows; p.C = C; p.dy_rs = C; p.x_rs = x_row_stride; p.dx_rs = dx_row_stride;
  p.tok_rps = rows_per_sample; p.tok_gap = gap_rows; p.tok_magic = ln_rps_magic(rows, rows_per_sample);
  return ln_bwd_launch(p, x_dtype, dy_dtype, (hipStream_t)stream);
}

// 64-bit division per row and lane is ~50 VALU instructions; one multiply-high when exact for the call's row range
static unsigned ln_rps_magic(int64_t rows, int64_t rps) {
  if (rps <= 0 || rps >= (1ll << 31) || rows >= (1ll << 32)) return 0;
  const uint64_t magic = ((1ull << 32) + (uint64_t)rps - 1) / (uint64_t)rps;     // ceil(2^32 / rps)
  if (magic >= (1ull << 32)) return 0;
  // floor(n * magic / 2^32) == floor(n / rps) for all n < rows  <=  rows * (magic * rps - 2^32) < 2^32
  const uint64_t e = magic * (uint64_t)rps - (1ull << 32);
  return (e * (uint64_t)rows < (1ull << 32)) ? (unsigned)magic : 0u;
}

// ---- fused residual + LayerNorm on the fp32 residual stream (contiguous rows)
extern "C" int vil_resln_fwd(const float* x, const void* res, int res_dtype, const float* rscale, int64_t rows_per_sample,
                             const float* gamma, const float* beta, float* x_out, void* y, int y_dtype,
                             float* mean, float* rstd, int64_t rows, int C, float eps, void* stream) {
  if (!x || !res || !gamma || !beta || !x_out || !y || !mean || !rstd) return VIL_E_NULL;
  int e = ln_check(rows, C, C, C);
  if (e) return e;
  if (rows_per_sample <= 0) return VIL_E_SHAPE;
  if (((uintptr_t)x | (uintptr_t)res | (uintptr_t)x_out | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15) return VIL_E_ALIGN;
  LNParams p; memset(&p, 0, sizeof(p));
  p.x = x; p.y = y; p.gamma = gamma; p.beta = beta; p.mean = mean; p.rstd = rstd;
  p.rows = rows; p.C = C; p.x_rs = C; p.y_rs = C; p.eps = eps;
  p.res = res; p.res_bf16 = res_dtype == VIL_DTYPE_BF16; p.res_rs = C; p.rscale = rscale; p.rps = rows_per_sample;
  p.xout = x_out; p.xout_rs = C;
  p.rps_magic = ln_rps_magic(rows, rows_per_sample);
  return ln_fwd_launch(p, VIL_DTYPE_F32, y_dtype, (hipStream_t)stream);
}

extern "C" int vil_resln_bwd(const void* dy, int dy_dtype, const float* gres, const float* x, const float* gamma,
                             const float* mean, const float* rstd, const float* rscale, int64_t rows_per_sample,
                             float* dx, void* gbranch, int gb_dtype, float* dgamma, float* dbeta, void* workspace,
                             int64_t rows, int C, void* stream) {
  // gbranch == NULL: no branch was added in the forward (the first block of a stage): dx = gres + LNbwd(dy) only
  if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || !workspace) return VIL_E_NULL;
  int e = ln_check(rows, C, C, C);
  if (e) return e;
  if (rows_per_sample <= 0) return VIL_E_SHAPE;
  if (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)gamma | (uintptr_t)gres | (uintptr_t)gbranch) & 15) return VIL_E_ALIGN;
  LNParams p; memset(&p, 0, sizeof(p));
  p.dy = dy; p.x = x; p.gamma = gamma; p.mean = (float*)mean; p.rstd = (float*)rstd; p.dx = dx;
  p.dgamma = dgamma; p.dbeta = dbeta; p.parts = (float*)workspace;
  p.rows = rows; p.C = C; p.dy_rs = C; p.x_rs = C; p.dx_rs = C;
  p.gres = gres; p.gres_rs = C; p.gbranch = gbranch; p.gb_bf16 = gb_dtype == VIL_DTYPE_BF16; p.gb_rs = C;
  p.rscale = rscale; p.rps = rows_per_sample;
  p.rps_magic = ln_rps_magic(rows, rows_per_sample);
  return ln_bwd_launch(p, VIL_DTYPE_F32, dy_dtype, (hipStream_t)stream);
}
