// vil_attn_glo.hip -- the G global-token QUERY rows of Vision Longformer attention (SURVEY.md 8f
// row 1; reference src/models/layers/longformer2d.py:210-227): each global token attends all
// N = G + Nloc keys with bias g2g[h][g][g'] (global keys) / g2l[0][h][g] (local keys).
//
// G is tiny (1 in every published model), so this is a matrix-VECTOR problem and purely HBM-bound:
// one workgroup per (image, head) streams K and V once (4 lanes x 16 bytes per key row, coalesced),
// every lane keeps an online-softmax partial (m, l, o[M/4 dims]) for each global query, and the
// workgroup merges the partials through LDS.  The backward recomputes p from the saved lse, reduces
// dq the same way and ACCUMULATES the global rows' contribution into dk/dv in place (stream-ordered
// after the local pass wrote them), so autograd never has to add two (B,N,2C) gradient tensors.
#include "vil_internal.h"
#include <string.h>

template <typename T> struct GIO;
template <> struct GIO<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct GIO<vil_bf16> {
  static __device__ __forceinline__ float ld(const vil_bf16* p) { return vil_bf2f(*p); }
  static __device__ __forceinline__ void st(vil_bf16* p, float v) { *p = vil_f2bf(v); }
};
template <> struct GIO<_Float16> {
  static __device__ __forceinline__ float ld(const _Float16* p) { return (float)*p; }
  static __device__ __forceinline__ void st(_Float16* p, float v) { *p = (_Float16)v; }
};

// DPL consecutive elements of one row into registers: 16-byte loads where the layout allows (bf16, DPL % 8 == 0;
// rows are 16-byte aligned by the descriptor checks), element loads otherwise
template <typename T, int DPL>
__device__ __forceinline__ void glo_ld(const T* p, float (&v)[DPL]) {
  if constexpr (sizeof(T) == 2 && DPL % 8 == 0 && !__is_same(T, _Float16)) {     // (bf16 bit layout)
    typedef unsigned gu32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int c = 0; c < DPL / 8; ++c) {
      const gu32x4 a = *(const gu32x4*)(p + c * 8);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[c * 8 + 2 * i] = __uint_as_float(a[i] << 16);
        v[c * 8 + 2 * i + 1] = __uint_as_float(a[i] & 0xffff0000u);
      }
    }
  } else {
#pragma unroll
    for (int d = 0; d < DPL; ++d) v[d] = GIO<T>::ld(p + d);
  }
}

#define GLO_MAXG 4          // global tokens handled per pass (loops over G in chunks)
#define GLO_THREADS 256
// forward: 16 waves per (image, head) and two to four key rows per lane in flight.  Round 3 ran 4 waves per (image, head) with one
// row per lane and no load ahead of its use: 49 (56x56) .. 144 (96x96) serial memory round trips per workgroup, 52 / 89 us
// for 19 / 14 us of HBM time, on 96 .. 384 workgroups.
#define GLO_FWD_THREADS 1024        // sequences of >= 2048 keys; shorter ones (28x28: 785 keys) keep 4 waves: fewer partials to merge
#define GLO_NEG (-1.0e30f)

struct GloParams {
  int B, H, M, G, Nloc;
  float scale;
  int64_t q_sb, q_st, q_sh, k_sb, k_st, k_sh, v_sb, v_st, v_sh, o_sb, o_st, o_sh;
  int64_t do_sb, do_st, do_sh, dq_sb, dq_st, dq_sh, dk_sb, dk_st, dk_sh, dv_sb, dv_st, dv_sh;
  const void* q; const void* k; const void* v; const void* out; const void* dout;
  void* o; void* dq; void* dk; void* dv;
  const float* g2g; const float* g2l0;     // (H,G,G), (H,G) or null
  float* lse;                              // (B,H,G)
  float* dg2g; float* dg2l0;               // accumulated with atomics, or null
};

// lane layout: 4 lanes per key row (each DPL = M/4 consecutive dims), 256 key rows per pass of 1024 threads, the rows of
// GLO_FWD_ROWS passes loaded before any of them is used
// NGT: global queries the instantiation keeps per lane (1: every published model, half the registers; GLO_MAXG otherwise)
template <typename T, int M, int NGT, int NW>
__device__ __forceinline__ void glo_fwd_body(const GloParams& p, int g0) {
  constexpr int DPL = M / 4, RPP = NW * 16;
  constexpr int GLO_FWD_ROWS = (M <= 32 && NGT == 1) ? 4 : 2;      // rows in flight per lane (128 registers per lane at 16 waves)
  __shared__ float s_m[NGT][NW], s_l[NGT][NW];
  __shared__ float s_o[NGT][NW][M];
  // the H heads of an image share K / V cache lines (two heads per 128-byte line at head_dim 32): they run on the SAME
  // XCD (the hardware places block i on XCD i % 8) in consecutive slots, so a line is fetched from HBM once
  // (bh-ordered blocks put the heads of an image on different XCDs: 2.0x the algorithmic bytes, round 3)
  const int slot = blockIdx.x >> 3;
  const int b = (slot / p.H) * 8 + (int)(blockIdx.x & 7), h = slot % p.H;
  if (b >= p.B) return;
  const int bh = b * p.H + h;
  const int tid = threadIdx.x, sub = tid & 3, rowl = tid >> 2;
  const int ng = min(NGT, p.G - g0);
  const int N = p.G + p.Nloc;
  const T* kb = (const T*)p.k + b * p.k_sb + h * p.k_sh;
  const T* vb = (const T*)p.v + b * p.v_sb + h * p.v_sh;
  float q[NGT][DPL], o[NGT][DPL], m[NGT], l[NGT];
#pragma unroll
  for (int g = 0; g < NGT; ++g) {
    m[g] = GLO_NEG; l[g] = 0.f;
#pragma unroll
    for (int d = 0; d < DPL; ++d) {
      o[g][d] = 0.f;
      q[g][d] = g < ng ? GIO<T>::ld((const T*)p.q + b * p.q_sb + (int64_t)(g0 + g) * p.q_st + h * p.q_sh + sub * DPL + d) * p.scale : 0.f;
    }
  }
  for (int j0 = rowl; j0 < N; j0 += RPP * GLO_FWD_ROWS) {
    float kk[GLO_FWD_ROWS][DPL], vv[GLO_FWD_ROWS][DPL];
#pragma unroll
    for (int u = 0; u < GLO_FWD_ROWS; ++u) {
      const int j = min(j0 + u * RPP, N - 1);                  // (rows past the end: loaded, not used)
      glo_ld<T, DPL>(kb + (int64_t)j * p.k_st + sub * DPL, kk[u]);
      glo_ld<T, DPL>(vb + (int64_t)j * p.v_st + sub * DPL, vv[u]);
    }
#pragma unroll
    for (int u = 0; u < GLO_FWD_ROWS; ++u) {
      const int j = j0 + u * RPP;
      if (j >= N) break;
#pragma unroll
      for (int g = 0; g < NGT; ++g) {
        if (g < ng) {
          float s = 0.f;
#pragma unroll
          for (int d = 0; d < DPL; ++d) s = fmaf(q[g][d], kk[u][d], s);
          s += __shfl_xor(s, 1, 64);
          s += __shfl_xor(s, 2, 64);
          if (j < p.G) { if (p.g2g) s += p.g2g[((int64_t)h * p.G + g0 + g) * p.G + j]; }
          else if (p.g2l0) s += p.g2l0[h * p.G + g0 + g];
          if (s > m[g]) {
            const float a = __expf(m[g] - s);
            l[g] *= a;
#pragma unroll
            for (int d = 0; d < DPL; ++d) o[g][d] *= a;
            m[g] = s;
          }
          const float pr = __expf(s - m[g]);
          l[g] += pr;
#pragma unroll
          for (int d = 0; d < DPL; ++d) o[g][d] = fmaf(pr, vv[u][d], o[g][d]);
        }
      }
    }
  }
  // merge: butterfly over the 16 row-groups of each wave, then the waves through LDS
  const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int g = 0; g < NGT; ++g) {
    if (g >= ng) break;                            // (wave-uniform) G is 1 in every published model
#pragma unroll
    for (int off = 4; off < 64; off <<= 1) {
      const float m2 = __shfl_xor(m[g], off, 64), l2 = __shfl_xor(l[g], off, 64);
      const float mm = fmaxf(m[g], m2);
      const float a1 = __expf(m[g] - mm), a2 = __expf(m2 - mm);
      l[g] = l[g] * a1 + l2 * a2;
#pragma unroll
      for (int d = 0; d < DPL; ++d) o[g][d] = o[g][d] * a1 + __shfl_xor(o[g][d], off, 64) * a2;
      m[g] = mm;
    }
    if (lane < 4) {
      if (sub == 0) { s_m[g][wave] = m[g]; s_l[g][wave] = l[g]; }
#pragma unroll
      for (int d = 0; d < DPL; ++d) s_o[g][wave][sub * DPL + d] = o[g][d];
    }
  }
  __syncthreads();
  for (int e = tid; e < ng * M; e += NW * 64) {
    const int g = e / M, d = e % M;
    float mm = GLO_NEG;
    for (int r = 0; r < NW; ++r) mm = fmaxf(mm, s_m[g][r]);
    float ll = 0.f, oo = 0.f;
    for (int r = 0; r < NW; ++r) {
      const float a = __expf(s_m[g][r] - mm);
      ll = fmaf(s_l[g][r], a, ll);
      oo = fmaf(s_o[g][r][d], a, oo);
    }
    GIO<T>::st((T*)p.o + b * p.o_sb + (int64_t)(g0 + g) * p.o_st + h * p.o_sh + d, oo / ll);
    if (d == 0) p.lse[(int64_t)bh * p.G + g0 + g] = mm + __logf(ll);
  }
}

template <typename T, int M>
__global__ __launch_bounds__(GLO_FWD_THREADS) void k_glo_fwd(GloParams p, int g0) { glo_fwd_body<T, M, 1, GLO_FWD_THREADS / 64>(p, g0); }
template <typename T, int M>
__global__ __launch_bounds__(GLO_THREADS) void k_glo_fwd_short(GloParams p, int g0) { glo_fwd_body<T, M, 1, GLO_THREADS / 64>(p, g0); }
template <typename T, int M>
__global__ __launch_bounds__(GLO_THREADS) void k_glo_fwd_multi(GloParams p, int g0) { glo_fwd_body<T, M, GLO_MAXG, GLO_THREADS / 64>(p, g0); }

template <typename T, int M>
__global__ __launch_bounds__(GLO_THREADS) void k_glo_bwd(GloParams p, int g0) {
  constexpr int DPL = M / 4;
  __shared__ float s_dq[GLO_MAXG][4][M];
  __shared__ float s_b[GLO_MAXG][GLO_MAXG + 1][4];   // bias-gradient partials: [g][g' or G=local][wave]
  const int bh = blockIdx.x, b = bh / p.H, h = bh % p.H;
  const int tid = threadIdx.x, sub = tid & 3, rowl = tid >> 2;
  const int ng = min(GLO_MAXG, p.G - g0);
  const int N = p.G + p.Nloc;
  const T* kb = (const T*)p.k + b * p.k_sb + h * p.k_sh;
  const T* vb = (const T*)p.v + b * p.v_sb + h * p.v_sh;
  T* dkb = (T*)p.dk + b * p.dk_sb + h * p.dk_sh;
  T* dvb = (T*)p.dv + b * p.dv_sb + h * p.dv_sh;
  float q[GLO_MAXG][DPL], dO[GLO_MAXG][DPL], dq[GLO_MAXG][DPL], lse[GLO_MAXG], delta[GLO_MAXG];
  float bloc[GLO_MAXG], bglo[GLO_MAXG][GLO_MAXG];
#pragma unroll
  for (int g = 0; g < GLO_MAXG; ++g) {
    float dl = 0.f;
    bloc[g] = 0.f;
#pragma unroll
    for (int g2 = 0; g2 < GLO_MAXG; ++g2) bglo[g][g2] = 0.f;
#pragma unroll
    for (int d = 0; d < DPL; ++d) {
      const int64_t qo = b * p.q_sb + (int64_t)(g0 + g) * p.q_st + h * p.q_sh + sub * DPL + d;
      const int64_t oo = b * p.o_sb + (int64_t)(g0 + g) * p.o_st + h * p.o_sh + sub * DPL + d;
      const int64_t go = b * p.do_sb + (int64_t)(g0 + g) * p.do_st + h * p.do_sh + sub * DPL + d;
      q[g][d] = g < ng ? GIO<T>::ld((const T*)p.q + qo) * p.scale : 0.f;
      dO[g][d] = g < ng ? GIO<T>::ld((const T*)p.dout + go) : 0.f;
      dq[g][d] = 0.f;
      dl = fmaf(dO[g][d], g < ng ? GIO<T>::ld((const T*)p.out + oo) : 0.f, dl);
    }
    dl += __shfl_xor(dl, 1, 64);
    dl += __shfl_xor(dl, 2, 64);
    delta[g] = dl;
    lse[g] = g < ng ? p.lse[(int64_t)bh * p.G + g0 + g] : 0.f;
  }
  for (int j = rowl; j < N; j += GLO_THREADS / 4) {
    float kk[DPL], vv[DPL], dkk[DPL], dvv[DPL];
#pragma unroll
    for (int d = 0; d < DPL; ++d) {
      kk[d] = GIO<T>::ld(kb + (int64_t)j * p.k_st + sub * DPL + d);
      vv[d] = GIO<T>::ld(vb + (int64_t)j * p.v_st + sub * DPL + d);
      dkk[d] = GIO<T>::ld(dkb + (int64_t)j * p.dk_st + sub * DPL + d);
      dvv[d] = GIO<T>::ld(dvb + (int64_t)j * p.dv_st + sub * DPL + d);
    }
#pragma unroll
    for (int g = 0; g < GLO_MAXG; ++g) {
      if (g < ng) {
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int d = 0; d < DPL; ++d) { s = fmaf(q[g][d], kk[d], s); dp = fmaf(dO[g][d], vv[d], dp); }
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64);
        dp += __shfl_xor(dp, 1, 64); dp += __shfl_xor(dp, 2, 64);
        if (j < p.G) { if (p.g2g) s += p.g2g[((int64_t)h * p.G + g0 + g) * p.G + j]; }
        else if (p.g2l0) s += p.g2l0[h * p.G + g0 + g];
        const float pr = __expf(s - lse[g]);
        const float ds = pr * (dp - delta[g]);
#pragma unroll
        for (int d = 0; d < DPL; ++d) {
          dq[g][d] = fmaf(ds, kk[d], dq[g][d]);
          dkk[d] = fmaf(ds, q[g][d], dkk[d]);        // q already carries `scale`
          dvv[d] = fmaf(pr, dO[g][d], dvv[d]);
        }
        if (sub == 0) {
          if (j < p.G) {
#pragma unroll
            for (int g2 = 0; g2 < GLO_MAXG; ++g2) if (g2 == j) bglo[g][g2] += ds;
          } else bloc[g] += ds;
        }
      }
    }
#pragma unroll
    for (int d = 0; d < DPL; ++d) {
      GIO<T>::st(dkb + (int64_t)j * p.dk_st + sub * DPL + d, dkk[d]);
      GIO<T>::st(dvb + (int64_t)j * p.dv_st + sub * DPL + d, dvv[d]);
    }
  }
  const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int g = 0; g < GLO_MAXG; ++g) {
#pragma unroll
    for (int off = 4; off < 64; off <<= 1) {
#pragma unroll
      for (int d = 0; d < DPL; ++d) dq[g][d] += __shfl_xor(dq[g][d], off, 64);
      bloc[g] += __shfl_xor(bloc[g], off, 64);
#pragma unroll
      for (int g2 = 0; g2 < GLO_MAXG; ++g2) bglo[g][g2] += __shfl_xor(bglo[g][g2], off, 64);
    }
    if (lane < 4) {
#pragma unroll
      for (int d = 0; d < DPL; ++d) s_dq[g][wave][sub * DPL + d] = dq[g][d];
      if (sub == 0) {
        s_b[g][GLO_MAXG][wave] = bloc[g];
#pragma unroll
        for (int g2 = 0; g2 < GLO_MAXG; ++g2) s_b[g][g2][wave] = bglo[g][g2];
      }
    }
  }
  __syncthreads();
  for (int e = tid; e < ng * M; e += GLO_THREADS) {
    const int g = e / M, d = e % M;
    const float s = s_dq[g][0][d] + s_dq[g][1][d] + s_dq[g][2][d] + s_dq[g][3][d];
    GIO<T>::st((T*)p.dq + b * p.dq_sb + (int64_t)(g0 + g) * p.dq_st + h * p.dq_sh + d, s * p.scale);
  }
  for (int e = tid; e < ng * (GLO_MAXG + 1); e += GLO_THREADS) {
    const int g = e / (GLO_MAXG + 1), c = e % (GLO_MAXG + 1);
    const float s = s_b[g][c][0] + s_b[g][c][1] + s_b[g][c][2] + s_b[g][c][3];
    if (c == GLO_MAXG) { if (p.dg2l0) atomicAdd(&p.dg2l0[h * p.G + g0 + g], s); }
    else if (c < p.G && p.dg2g) atomicAdd(&p.dg2g[((int64_t)h * p.G + g0 + g) * p.G + c], s);
  }
}

// ------------------------------------------------------------------ C ABI
static int glo_check(const VilAttnDesc* d) {
  if (!d) return VIL_E_NULL;
  if (d->B <= 0 || d->H <= 0 || d->G <= 0 || d->nx <= 0 || d->ny <= 0) return VIL_E_SHAPE;
  if (d->G > GLO_MAXG) return VIL_E_BACKEND;      // bias-gradient bookkeeping is sized for G <= 4
  switch (d->M) { case 8: case 16: case 32: case 48: case 64: break; default: return VIL_E_HEAD_DIM; }
  if (d->dtype != VIL_DTYPE_F32 && d->dtype != VIL_DTYPE_BF16 && d->dtype != VIL_DTYPE_F16) return VIL_E_DTYPE;
  // 16-byte row loads of K / V in the bf16 kernels with M % 32 == 0
  if (d->dtype == VIL_DTYPE_BF16 && d->M % 32 == 0 &&
      ((d->k_st | d->k_sb | d->k_sh | d->v_st | d->v_sb | d->v_sh) & 7)) return VIL_E_ALIGN;
  return VIL_OK;
}

static void glo_fill(GloParams& p, const VilAttnDesc* d) {
  memset(&p, 0, sizeof(p));
  p.B = d->B; p.H = d->H; p.M = d->M; p.G = d->G; p.Nloc = d->nx * d->ny; p.scale = d->scale;
  p.q_sb = d->q_sb; p.q_st = d->q_st; p.q_sh = d->q_sh; p.k_sb = d->k_sb; p.k_st = d->k_st; p.k_sh = d->k_sh;
  p.v_sb = d->v_sb; p.v_st = d->v_st; p.v_sh = d->v_sh; p.o_sb = d->o_sb; p.o_st = d->o_st; p.o_sh = d->o_sh;
  p.do_sb = d->do_sb; p.do_st = d->do_st; p.do_sh = d->do_sh; p.dq_sb = d->dq_sb; p.dq_st = d->dq_st; p.dq_sh = d->dq_sh;
  p.dk_sb = d->dk_sb; p.dk_st = d->dk_st; p.dk_sh = d->dk_sh; p.dv_sb = d->dv_sb; p.dv_st = d->dv_st; p.dv_sh = d->dv_sh;
}

#define GLO_DISPATCH(KERN, ...)                                                      \
  if (d->dtype == VIL_DTYPE_F16) {                                                   \
    switch (d->M) {                                                                  \
      case 8: KERN<_Float16, 8><<<__VA_ARGS__>>>(p, 0); break;                        \
      case 16: KERN<_Float16, 16><<<__VA_ARGS__>>>(p, 0); break;                      \
      case 32: KERN<_Float16, 32><<<__VA_ARGS__>>>(p, 0); break;                      \
      case 48: KERN<_Float16, 48><<<__VA_ARGS__>>>(p, 0); break;                      \
      case 64: KERN<_Float16, 64><<<__VA_ARGS__>>>(p, 0); break;                      \
      default: return VIL_E_HEAD_DIM;                                                \
    }                                                                                \
  } else                                                                             \
  switch (d->M * 2 + (d->dtype == VIL_DTYPE_BF16)) {                                 \
    case 16: KERN<float, 8><<<__VA_ARGS__>>>(p, 0); break;                            \
    case 17: KERN<vil_bf16, 8><<<__VA_ARGS__>>>(p, 0); break;                         \
    case 32: KERN<float, 16><<<__VA_ARGS__>>>(p, 0); break;                           \
    case 33: KERN<vil_bf16, 16><<<__VA_ARGS__>>>(p, 0); break;                        \
    case 64: KERN<float, 32><<<__VA_ARGS__>>>(p, 0); break;                           \
    case 65: KERN<vil_bf16, 32><<<__VA_ARGS__>>>(p, 0); break;                        \
    case 96: KERN<float, 48><<<__VA_ARGS__>>>(p, 0); break;                           \
    case 97: KERN<vil_bf16, 48><<<__VA_ARGS__>>>(p, 0); break;                        \
    case 128: KERN<float, 64><<<__VA_ARGS__>>>(p, 0); break;                          \
    case 129: KERN<vil_bf16, 64><<<__VA_ARGS__>>>(p, 0); break;                       \
    default: return VIL_E_HEAD_DIM;                                                  \
  }

extern "C" int vil_glo_attn_fwd(const VilAttnDesc* d, const void* q_g, const void* k, const void* v,
                                const float* g2g, const float* g2l0, void* out_g, float* lse_g, void* stream) {
  int e = glo_check(d);
  if (e) return e;
  if (!q_g || !k || !v || !out_g || !lse_g) return VIL_E_NULL;
  if (d->dtype == VIL_DTYPE_BF16 && (((uintptr_t)k | (uintptr_t)v) & 15)) return VIL_E_ALIGN;
  GloParams p; glo_fill(p, d);
  p.q = q_g; p.k = k; p.v = v; p.o = out_g; p.lse = lse_g; p.g2g = g2g; p.g2l0 = g2l0;
  hipStream_t s = (hipStream_t)stream;
  vil_prof_tag_desc(d);
  const double e_ = d->dtype == VIL_DTYPE_F32 ? 4 : 2, n_ = (double)d->G + (double)d->nx * d->ny;
  vil_prof_begin(VIL_K_GLO_FWD, s, d->B * (2 * n_ + 2 * d->G) * d->H * d->M * e_, d->B * 4.0 * d->G * n_ * d->H * d->M);
  const dim3 grid((unsigned)(((d->B + 7) / 8) * 8 * d->H));        // (image, head) -> block: see glo_fwd_body
  if (d->G > 1) { GLO_DISPATCH(k_glo_fwd_multi, grid, dim3(GLO_THREADS), 0, s); }
  else if ((int64_t)d->nx * d->ny >= 2048) { GLO_DISPATCH(k_glo_fwd, grid, dim3(GLO_FWD_THREADS), 0, s); }
  else { GLO_DISPATCH(k_glo_fwd_short, grid, dim3(GLO_THREADS), 0, s); }
  vil_prof_end(s);
  return (int)hipGetLastError();
}

extern "C" int vil_glo_attn_bwd(const VilAttnDesc* d, const void* q_g, const void* k, const void* v,
                                const void* out_g, const void* dout_g, const float* lse_g,
                                const float* g2g, const float* g2l0, void* dq_g, void* dk, void* dv,
                                float* dg2g, float* dg2l0, void* stream) {
  int e = glo_check(d);
  if (e) return e;
  if (!q_g || !k || !v || !out_g || !dout_g || !lse_g || !dq_g || !dk || !dv) return VIL_E_NULL;
  GloParams p; glo_fill(p, d);
  p.q = q_g; p.k = k; p.v = v; p.out = out_g; p.dout = dout_g; p.lse = (float*)lse_g; p.g2g = g2g; p.g2l0 = g2l0;
  p.dq = dq_g; p.dk = dk; p.dv = dv; p.dg2g = dg2g; p.dg2l0 = dg2l0;
  hipStream_t s = (hipStream_t)stream;
  vil_prof_tag_desc(d);
  const double e_ = d->dtype == VIL_DTYPE_F32 ? 4 : 2, n_ = (double)d->G + (double)d->nx * d->ny;
  // reads k, v, dk, dv and rewrites dk, dv (the in-place accumulation), + the G query-side rows
  vil_prof_begin(VIL_K_GLO_BWD, s, d->B * (6 * n_ + 4 * d->G) * d->H * d->M * e_, d->B * 10.0 * d->G * n_ * d->H * d->M);
  GLO_DISPATCH(k_glo_bwd, dim3(d->B * d->H), dim3(GLO_THREADS), 0, s);
  vil_prof_end(s);
  return (int)hipGetLastError();
}
