// vil_attn_scalar.hip -- the "scalar" kernel family: one query (forward, dQ) or
// one key (dK/dV) per lane, exact fp32 arithmetic, any W / head_dim / mode /
// mask kind, f32 or bf16 I/O.  It is the fp32 path of the library (tight
// parity with the reference's fp32 tolerances) and the fallback for shapes the
// MFMA family does not cover.  No score tensor is ever materialised: each lane
// runs an online softmax over the keys its query may attend.
//
// Reference semantics restated here (relative to the reference repository):
//   src/models/layers/longformer2d.py:134-204   (local rows of the module forward)
//   src/models/layers/slidingchunk_2d.py:26-246 (sliding-chunk products + backward)
//   src/models/layers/slidingchunk_2d.py:249-357 (masks)
#include "vil_internal.h"

#define KT 64                      // keys (or queries) staged in LDS per tile
#define VIL_NEG_INF (-__builtin_huge_valf())

template <typename T> struct IOT;
template <> struct IOT<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct IOT<vil_bf16> {
  static __device__ __forceinline__ float ld(const vil_bf16* p) { return vil_bf2f(*p); }
  static __device__ __forceinline__ void st(vil_bf16* p, float v) { *p = vil_f2bf(v); }
};
template <> struct IOT<_Float16> {      // the reference's AMP dtype (fp16 autocast)
  static __device__ __forceinline__ float ld(const _Float16* p) { return (float)*p; }
  static __device__ __forceinline__ void st(_Float16* p, float v) { *p = (_Float16)v; }
};

template <typename T, int M>
__device__ __forceinline__ void load_row(const T* p, float (&r)[M]) {
#pragma unroll
  for (int d = 0; d < M; ++d) r[d] = IOT<T>::ld(p + d);
}
template <typename T, int M>
__device__ __forceinline__ void store_row(T* p, const float (&r)[M]) {
#pragma unroll
  for (int d = 0; d < M; ++d) IOT<T>::st(p + d, r[d]);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// LDS tile of staged rows (keys for fwd/dQ, queries for dK/dV)
template <int M>
struct Tile {
  float a[KT][M];      // K rows   | scaled Q rows
  float b[KT][M];      // V rows   | dO rows
  float f0[KT];        //          | lse
  float f1[KT];        //          | delta
  int r[KT], c[KT];    // absolute (row, col) of the staged token
  int x[KT], y[KT];    // in-chunk coordinates
  int state[KT];
};

// ---- stage the keys [t0, t0+KT) of the chunk at active offset a of query chunk (m,n)
template <typename T, int M>
__device__ __forceinline__ int stage_keys(const VilParams& p, Tile<M>& t, int b, int h, int m, int n,
                                          int dr, int dc, int t0) {
  const VilGeom& g = p.g;
  const int cnt = min(KT, g.W2 - t0);
  const T* kp = (const T*)p.k + b * p.k_sb + h * p.k_sh;
  const T* vp = (const T*)p.v + b * p.v_sb + h * p.v_sh;
  for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
    const int tt = t0 + i, xt = tt / g.W, yt = tt % g.W;
    int kr, kc;
    const int st = vil_key_state(g, m, n, dr, dc, xt, yt, kr, kc);
    t.state[i] = st; t.r[i] = kr; t.c[i] = kc; t.x[i] = xt; t.y[i] = yt;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < cnt * M; e += blockDim.x) {
    const int i = e / M, d = e % M;
    float kvv = 0.f, vvv = 0.f;
    if (t.state[i] == VIL_KEY_REAL) {
      const int64_t tok = p.G + (int64_t)t.r[i] * g.ny + t.c[i];
      kvv = IOT<T>::ld(kp + tok * p.k_st + d);
      vvv = IOT<T>::ld(vp + tok * p.v_st + d);
    }
    t.a[i][d] = kvv; t.b[i][d] = vvv;
  }
  __syncthreads();
  return cnt;
}

// ---- stage global keys [g0, g0+KT)
template <typename T, int M>
__device__ __forceinline__ int stage_global_keys(const VilParams& p, Tile<M>& t, int b, int h, int g0) {
  const int cnt = min(KT, p.G - g0);
  const T* kp = (const T*)p.k + b * p.k_sb + h * p.k_sh;
  const T* vp = (const T*)p.v + b * p.v_sb + h * p.v_sh;
  __syncthreads();
  for (int e = threadIdx.x; e < cnt * M; e += blockDim.x) {
    const int i = e / M, d = e % M;
    t.a[i][d] = IOT<T>::ld(kp + (int64_t)(g0 + i) * p.k_st + d);
    t.b[i][d] = IOT<T>::ld(vp + (int64_t)(g0 + i) * p.v_st + d);
  }
  __syncthreads();
  return cnt;
}

struct QueryCtx {
  int b, h, m, n, l, xl, yl, qr, qc;
  bool exists, real;
  int64_t tok;
};

__device__ __forceinline__ QueryCtx make_query(const VilParams& p, int unit, int bh) {
  const VilGeom& g = p.g;
  const int nq = (g.W2 + 63) / 64;
  QueryCtx c;
  c.b = bh / p.H; c.h = bh % p.H;
  const int qp = unit % nq; unit /= nq;
  c.n = unit % g.my; c.m = unit / g.my;
  c.l = qp * 64 + threadIdx.x;
  c.exists = c.l < g.W2;
  c.xl = c.exists ? c.l / g.W : 0; c.yl = c.exists ? c.l % g.W : 0;
  c.qr = c.m * g.W + c.xl; c.qc = c.n * g.W + c.yl;
  c.real = c.exists && c.qr < g.nx && c.qc < g.ny;
  c.tok = c.real ? (int64_t)c.qr * g.ny + c.qc : 0;
  return c;
}

// score of this lane's query against staged key i of the chunk at offset (dr,dc)
template <int M>
__device__ __forceinline__ float local_score(const VilParams& p, const Tile<M>& t, const QueryCtx& c,
                                             const float (&q)[M], int i, int dr, int dc, int& bidx) {
  float s = 0.f;
#pragma unroll
  for (int d = 0; d < M; ++d) s = fmaf(q[d], t.a[i][d], s);
  bidx = -1;
  if (p.has_bias) {
    bidx = vil_bias_index(p.g.W, c.xl, c.yl, dr, dc, t.x[i], t.y[i]);
    s += p.table[(int64_t)bidx * p.H + c.h];
  }
  if (p.g.exact == 1 && !vil_exact_window(p.g.W, c.qr, c.qc, t.r[i], t.c[i])) s = VIL_NEG_INF;
  return s;
}

// ===================================================================== forward
template <typename T, int M>
__global__ __launch_bounds__(64) void k_scalar_fwd(VilParams p) {
  __shared__ Tile<M> t;
  const VilGeom& g = p.g;
  const int nq = (g.W2 + 63) / 64;
  const int units = g.mx * g.my * nq;
  const QueryCtx c = make_query(p, blockIdx.x % units, blockIdx.x / units);
  float q[M], o[M];
  load_row<T, M>((const T*)p.q + c.b * p.q_sb + c.tok * p.q_st + c.h * p.q_sh, q);
#pragma unroll
  for (int d = 0; d < M; ++d) { q[d] *= p.scale; o[d] = 0.f; }
  float mrun = VIL_NEG_INF, lrun = 0.f;

  auto update = [&](float s, const float* vrow) {
    if (s > VIL_NEG_INF) {
      if (s > mrun) {
        const float alpha = __expf(mrun - s);   // mrun = -inf -> 0
        lrun *= alpha;
#pragma unroll
        for (int d = 0; d < M; ++d) o[d] *= alpha;
        mrun = s;
      }
      const float pr = __expf(s - mrun);
      lrun += pr;
#pragma unroll
      for (int d = 0; d < M; ++d) o[d] = fmaf(pr, vrow[d], o[d]);
    }
  };

  for (int g0 = 0; g0 < p.G; g0 += KT) {
    const int cnt = stage_global_keys<T, M>(p, t, c.b, c.h, g0);
    if (c.exists)
      for (int i = 0; i < cnt; ++i) {
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < M; ++d) s = fmaf(q[d], t.a[i][d], s);
        if (p.has_g2l) s += p.g2l[c.h * p.G + g0 + i];
        update(s, t.b[i]);
      }
  }
  if (!p.only_glo) {
    for (int a = 0; a < g.nact; ++a) {
      const int dr = g.adr[a], dc = g.adc[a];
      for (int t0 = 0; t0 < g.W2; t0 += KT) {
        __syncthreads();
        const int cnt = stage_keys<T, M>(p, t, c.b, c.h, c.m, c.n, dr, dc, t0);
        if (c.exists)
          for (int i = 0; i < cnt; ++i) {
            if (t.state[i] == VIL_KEY_MASKED) continue;
            int bidx;
            const float s = local_score<M>(p, t, c, q, i, dr, dc, bidx);
            update(s, t.b[i]);
          }
      }
    }
  }
  if (c.real) {
    const float inv = 1.f / lrun;
#pragma unroll
    for (int d = 0; d < M; ++d) o[d] *= inv;
    store_row<T, M>((T*)p.o + c.b * p.o_sb + c.tok * p.o_st + c.h * p.o_sh, o);
    p.lse[((int64_t)c.b * p.H + c.h) * (g.nx * g.ny) + c.tok] = mrun + __logf(lrun);
  }
}

// ============================================================ delta = rowsum(dO*O)
template <typename T, int M>
__global__ void k_delta(VilParams p) {
  const int Nloc = p.g.nx * p.g.ny;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)p.B * p.H * Nloc) return;
  const int tok = i % Nloc; const int bh = i / Nloc; const int b = bh / p.H, h = bh % p.H;
  const T* op = (const T*)p.out + b * p.o_sb + (int64_t)tok * p.o_st + h * p.o_sh;
  const T* dp = (const T*)p.dout + b * p.do_sb + (int64_t)tok * p.do_st + h * p.do_sh;
  float s = 0.f;
#pragma unroll
  for (int d = 0; d < M; ++d) s = fmaf(IOT<T>::ld(op + d), IOT<T>::ld(dp + d), s);
  p.delta[i] = s;
}

// ============================================================ backward: dQ pass
// lane = query.  Also produces per-workgroup partial sums of d(bias table),
// d(g2l) and the global keys' dK/dV (every local query contributes to them),
// written to p.partials and summed by k_reduce_*.
template <typename T, int M>
__global__ __launch_bounds__(64) void k_scalar_bwd_dq(VilParams p) {
  __shared__ Tile<M> t;
  extern __shared__ float acc[];           // [nbins | G | G*M | G*M]
  const VilGeom& g = p.g;
  const int nq = (g.W2 + 63) / 64;
  const int units = g.mx * g.my * nq;
  const int bh = blockIdx.x / p.parts, part = blockIdx.x % p.parts;
  const int nbins = p.has_bias ? g.tbl * g.tbl : 0;
  float* hist = acc; float* ag2l = acc + nbins; float* akg = ag2l + p.G; float* avg = akg + p.G * M;
  for (int i = threadIdx.x; i < p.part_stride; i += blockDim.x) acc[i] = 0.f;
  __syncthreads();
  const int Nloc = g.nx * g.ny;

  for (int unit = part; unit < units; unit += p.parts) {
    const QueryCtx c = make_query(p, unit, bh);
    float q[M], dO[M], dq[M];
    load_row<T, M>((const T*)p.q + c.b * p.q_sb + c.tok * p.q_st + c.h * p.q_sh, q);
    load_row<T, M>((const T*)p.dout + c.b * p.do_sb + c.tok * p.do_st + c.h * p.do_sh, dO);
#pragma unroll
    for (int d = 0; d < M; ++d) { q[d] *= p.scale; dq[d] = 0.f; }
    const float lse = p.lse[(int64_t)bh * Nloc + c.tok];
    const float delta = p.delta[(int64_t)bh * Nloc + c.tok];

    for (int g0 = 0; g0 < p.G; g0 += KT) {
      const int cnt = stage_global_keys<T, M>(p, t, c.b, c.h, g0);
      for (int i = 0; i < cnt; ++i) {
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int d = 0; d < M; ++d) { s = fmaf(q[d], t.a[i][d], s); dp = fmaf(dO[d], t.b[i][d], dp); }
        if (p.has_g2l) s += p.g2l[c.h * p.G + g0 + i];
        const float pr = c.real ? __expf(s - lse) : 0.f;
        const float ds = pr * (dp - delta);
#pragma unroll
        for (int d = 0; d < M; ++d) dq[d] = fmaf(ds, t.a[i][d], dq[d]);
        const float r0 = wave_sum(ds);
        if (threadIdx.x == 0) ag2l[g0 + i] += r0;
#pragma unroll
        for (int d = 0; d < M; ++d) {
          const float rk = wave_sum(ds * q[d]);
          const float rv = wave_sum(pr * dO[d]);
          if (threadIdx.x == 0) { akg[(g0 + i) * M + d] += rk; avg[(g0 + i) * M + d] += rv; }
        }
      }
    }
    if (!p.only_glo) {
      for (int a = 0; a < g.nact; ++a) {
        const int dr = g.adr[a], dc = g.adc[a];
        for (int t0 = 0; t0 < g.W2; t0 += KT) {
          __syncthreads();
          const int cnt = stage_keys<T, M>(p, t, c.b, c.h, c.m, c.n, dr, dc, t0);
          if (c.real)
            for (int i = 0; i < cnt; ++i) {
              if (t.state[i] == VIL_KEY_MASKED) continue;
              int bidx;
              const float s = local_score<M>(p, t, c, q, i, dr, dc, bidx);
              if (!(s > VIL_NEG_INF)) continue;
              float dp = 0.f;
#pragma unroll
              for (int d = 0; d < M; ++d) dp = fmaf(dO[d], t.b[i][d], dp);
              const float ds = __expf(s - lse) * (dp - delta);
#pragma unroll
              for (int d = 0; d < M; ++d) dq[d] = fmaf(ds, t.a[i][d], dq[d]);
              if (bidx >= 0) atomicAdd(&hist[bidx], ds);
            }
        }
      }
    }
    if (c.real) {
#pragma unroll
      for (int d = 0; d < M; ++d) dq[d] *= p.scale;
      store_row<T, M>((T*)p.dq + c.b * p.dq_sb + c.tok * p.dq_st + c.h * p.dq_sh, dq);
    }
    __syncthreads();
  }
  __syncthreads();
  float* out = p.partials + (int64_t)blockIdx.x * p.part_stride;
  for (int i = threadIdx.x; i < p.part_stride; i += blockDim.x) out[i] = acc[i];
}

// dtable[idx*H+h] = sum_{b,part} partial ; dg2l[h*G+g] likewise
__global__ void k_reduce_bias(VilParams p) {
  const int nbins = p.has_bias ? p.g.tbl * p.g.tbl : 0;
  const int per_h = nbins + p.G;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= per_h * p.H) return;
  const int h = i / per_h, bin = i % per_h;
  float s = 0.f;
  for (int b = 0; b < p.B; ++b)
    for (int pt = 0; pt < p.parts; ++pt)
      s += p.partials[((int64_t)(b * p.H + h) * p.parts + pt) * p.part_stride + bin];
  if (bin < nbins) { if (p.dtable) p.dtable[(int64_t)bin * p.H + h] = s; }
  else if (p.dg2l) p.dg2l[h * p.G + (bin - nbins)] = s;
}

// dk/dv rows of the G global tokens = sum over parts of the dQ pass' partials
template <typename T>
__global__ void k_reduce_glo(VilParams p) {
  const int nbins = p.has_bias ? p.g.tbl * p.g.tbl : 0;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int M = p.M;
  if (i >= p.B * p.H * p.G * M) return;
  const int d = i % M; const int gk = (i / M) % p.G; const int bh = i / (M * p.G);
  const int b = bh / p.H, h = bh % p.H;
  float sk = 0.f, sv = 0.f;
  for (int pt = 0; pt < p.parts; ++pt) {
    const float* rec = p.partials + ((int64_t)bh * p.parts + pt) * p.part_stride + nbins + p.G;
    sk += rec[gk * M + d]; sv += rec[p.G * M + gk * M + d];
  }
  IOT<T>::st((T*)p.dk + b * p.dk_sb + (int64_t)gk * p.dk_st + h * p.dk_sh + d, sk);
  IOT<T>::st((T*)p.dv + b * p.dv_sb + (int64_t)gk * p.dv_st + h * p.dv_sh + d, sv);
}

// ============================================================ backward: dK/dV pass
// lane = key of the owner chunk (mk,nk); streams the query chunks that attend it.
template <typename T, int M>
__global__ __launch_bounds__(64) void k_scalar_bwd_dkdv(VilParams p) {
  __shared__ Tile<M> t;
  const VilGeom& g = p.g;
  const int nq = (g.W2 + 63) / 64;
  int u = blockIdx.x;
  const int kp = u % nq; u /= nq;
  const int nk = u % g.my; u /= g.my;
  const int mk = u % g.mx; u /= g.mx;
  const int h = u % p.H, b = u / p.H;
  const int tk = kp * 64 + threadIdx.x;
  const bool exists = tk < g.W2;
  const int xt = exists ? tk / g.W : 0, yt = exists ? tk % g.W : 0;
  const int kr = mk * g.W + xt, kc = nk * g.W + yt;
  const bool kreal = exists && kr < g.nx && kc < g.ny;
  const int64_t ktok = p.G + (kreal ? (int64_t)kr * g.ny + kc : 0);
  const int Nloc = g.nx * g.ny;
  float kk[M], vv[M], dk[M], dv[M];
  load_row<T, M>((const T*)p.k + b * p.k_sb + ktok * p.k_st + h * p.k_sh, kk);
  load_row<T, M>((const T*)p.v + b * p.v_sb + ktok * p.v_st + h * p.v_sh, vv);
#pragma unroll
  for (int d = 0; d < M; ++d) { dk[d] = 0.f; dv[d] = 0.f; }

  if (!p.only_glo) {
    for (int a = 0; a < g.nact; ++a) {
      const int dr = g.adr[a], dc = g.adc[a];
      int m = mk - dr, n = nk - dc;
      if (g.exact == -1) { m = vil_pmod(m, g.mx); n = vil_pmod(n, g.my); }
      else if (m < 0 || m >= g.mx || n < 0 || n >= g.my) continue;
      for (int l0 = 0; l0 < g.W2; l0 += KT) {
        const int cnt = min(KT, g.W2 - l0);
        __syncthreads();
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
          const int l = l0 + i, xl = l / g.W, yl = l % g.W;
          const int qr = m * g.W + xl, qc = n * g.W + yl;
          const bool real = qr < g.nx && qc < g.ny;
          t.state[i] = real; t.r[i] = qr; t.c[i] = qc; t.x[i] = xl; t.y[i] = yl;
          const int64_t tok = real ? (int64_t)qr * g.ny + qc : 0;
          t.f0[i] = p.lse[((int64_t)b * p.H + h) * Nloc + tok];
          t.f1[i] = p.delta[((int64_t)b * p.H + h) * Nloc + tok];
        }
        __syncthreads();
        for (int e = threadIdx.x; e < cnt * M; e += blockDim.x) {
          const int i = e / M, d = e % M;
          float qv = 0.f, dov = 0.f;
          if (t.state[i]) {
            const int64_t tok = (int64_t)t.r[i] * g.ny + t.c[i];
            qv = IOT<T>::ld((const T*)p.q + b * p.q_sb + tok * p.q_st + h * p.q_sh + d) * p.scale;
            dov = IOT<T>::ld((const T*)p.dout + b * p.do_sb + tok * p.do_st + h * p.do_sh + d);
          }
          t.a[i][d] = qv; t.b[i][d] = dov;
        }
        __syncthreads();
        if (kreal)
          for (int i = 0; i < cnt; ++i) {
            if (!t.state[i]) continue;
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < M; ++d) { s = fmaf(t.a[i][d], kk[d], s); dp = fmaf(t.b[i][d], vv[d], dp); }
            if (p.has_bias)
              s += p.table[(int64_t)vil_bias_index(g.W, t.x[i], t.y[i], dr, dc, xt, yt) * p.H + h];
            if (g.exact == 1 && !vil_exact_window(g.W, t.r[i], t.c[i], kr, kc)) continue;
            const float pr = __expf(s - t.f0[i]);
            const float ds = pr * (dp - t.f1[i]);
#pragma unroll
            for (int d = 0; d < M; ++d) { dv[d] = fmaf(pr, t.b[i][d], dv[d]); dk[d] = fmaf(ds, t.a[i][d], dk[d]); }
          }
      }
    }
  }
  if (kreal) {
    store_row<T, M>((T*)p.dk + b * p.dk_sb + ktok * p.dk_st + h * p.dk_sh, dk);
    store_row<T, M>((T*)p.dv + b * p.dv_sb + ktok * p.dv_st + h * p.dv_sh, dv);
  }
}

// ===================================================================== host side
static int scalar_parts(const VilAttnDesc* d) {
  VilGeom g; vil_geom_init(g, d->nx, d->ny, d->W, d->exact, d->mode);
  const int units = g.mx * g.my * ((g.W2 + 63) / 64);
  int parts = (4096 + d->B * d->H - 1) / (d->B * d->H);
  if (parts > units) parts = units;
  if (parts < 1) parts = 1;
  return parts;
}
static int scalar_part_stride(const VilAttnDesc* d, bool has_bias) {
  const int tbl = 4 * d->W - 1;
  return (has_bias ? tbl * tbl : 0) + d->G + 2 * d->G * d->M;
}

int vil_scalar_supported(const VilAttnDesc* d) {
  switch (d->M) { case 8: case 16: case 32: case 48: case 64: break; default: return VIL_E_HEAD_DIM; }
  if (d->W < 1 || d->W > 32) return VIL_E_WINDOW;
  if (d->dtype != VIL_DTYPE_F32 && d->dtype != VIL_DTYPE_BF16 && d->dtype != VIL_DTYPE_F16) return VIL_E_DTYPE;
  if (d->mode_dev) return VIL_E_BACKEND;                                           // device-side mode: MFMA family only
  if (d->bias_side != 0 && d->bias_side != 4 * d->W - 1) return VIL_E_BACKEND;   // non-default table side: MFMA family only
  return VIL_OK;
}

size_t vil_scalar_workspace(const VilAttnDesc* d, int pass) {
  if (pass == 0) return 0;
  const size_t nloc = (size_t)d->nx * d->ny;
  size_t fl = (size_t)d->B * d->H * nloc;                                     // delta
  fl += (size_t)d->B * d->H * scalar_parts(d) * scalar_part_stride(d, true);  // partials (upper bound)
  return fl * sizeof(float);
}

#define DISPATCH_M(M_, ...)                              \
  switch (M_) {                                          \
    case 8:  { constexpr int MM = 8;  __VA_ARGS__; } break;  \
    case 16: { constexpr int MM = 16; __VA_ARGS__; } break;  \
    case 32: { constexpr int MM = 32; __VA_ARGS__; } break;  \
    case 48: { constexpr int MM = 48; __VA_ARGS__; } break;  \
    case 64: { constexpr int MM = 64; __VA_ARGS__; } break;  \
    default: return VIL_E_HEAD_DIM;                      \
  }

int vil_scalar_fwd(const VilAttnDesc* d, VilParams& p, hipStream_t s) {
  const VilGeom& g = p.g;
  const VilWork w(d);
  vil_prof_begin(VIL_K_SCALAR_FWD, s, w.fwd_bytes(), w.fwd_flops());
  const int nq = (g.W2 + 63) / 64;
  const unsigned grid = (unsigned)(p.B * p.H * g.mx * g.my * nq);
  if (d->dtype == VIL_DTYPE_F32) {
    DISPATCH_M(d->M, k_scalar_fwd<float, MM><<<dim3(grid), dim3(64), 0, s>>>(p));
  } else if (d->dtype == VIL_DTYPE_F16) {
    DISPATCH_M(d->M, k_scalar_fwd<_Float16, MM><<<dim3(grid), dim3(64), 0, s>>>(p));
  } else {
    DISPATCH_M(d->M, k_scalar_fwd<vil_bf16, MM><<<dim3(grid), dim3(64), 0, s>>>(p));
  }
  vil_prof_end(s);
  return (int)hipGetLastError();
}

int vil_scalar_bwd(const VilAttnDesc* d, VilParams& p, hipStream_t s) {
  const VilGeom& g = p.g;
  const int nq = (g.W2 + 63) / 64;
  const int64_t rows = (int64_t)p.B * p.H * g.nx * g.ny;
  p.parts = scalar_parts(d);
  p.part_stride = scalar_part_stride(d, p.has_bias != 0);
  p.partials = p.delta + rows;
  const unsigned gd = (unsigned)((rows + 255) / 256);
  const unsigned gq = (unsigned)(p.B * p.H * p.parts);
  const unsigned gk = (unsigned)(p.B * p.H * g.mx * g.my * nq);
  const size_t accb = (size_t)p.part_stride * sizeof(float);
  int e;
  const VilWork w(d);
  const bool f32 = d->dtype == VIL_DTYPE_F32, f16 = d->dtype == VIL_DTYPE_F16;
  vil_prof_begin(VIL_K_DELTA, s, w.delta_bytes(), 0);
  if (f32) { DISPATCH_M(d->M, k_delta<float, MM><<<dim3(gd), dim3(256), 0, s>>>(p)); }
  else if (f16) { DISPATCH_M(d->M, k_delta<_Float16, MM><<<dim3(gd), dim3(256), 0, s>>>(p)); }
  else { DISPATCH_M(d->M, k_delta<vil_bf16, MM><<<dim3(gd), dim3(256), 0, s>>>(p)); }
  vil_prof_end(s);
  if ((e = (int)hipGetLastError())) return e;
  vil_prof_begin(VIL_K_SCALAR_DQ, s, w.dq_bytes(), w.dq_flops());
  if (f32) { DISPATCH_M(d->M, k_scalar_bwd_dq<float, MM><<<dim3(gq), dim3(64), accb, s>>>(p)); }
  else if (f16) { DISPATCH_M(d->M, k_scalar_bwd_dq<_Float16, MM><<<dim3(gq), dim3(64), accb, s>>>(p)); }
  else { DISPATCH_M(d->M, k_scalar_bwd_dq<vil_bf16, MM><<<dim3(gq), dim3(64), accb, s>>>(p)); }
  vil_prof_end(s);
  if ((e = (int)hipGetLastError())) return e;
  vil_prof_begin(VIL_K_SCALAR_DKDV, s, w.dkdv_bytes(), w.dkdv_flops());
  if (f32) { DISPATCH_M(d->M, k_scalar_bwd_dkdv<float, MM><<<dim3(gk), dim3(64), 0, s>>>(p)); }
  else if (f16) { DISPATCH_M(d->M, k_scalar_bwd_dkdv<_Float16, MM><<<dim3(gk), dim3(64), 0, s>>>(p)); }
  else { DISPATCH_M(d->M, k_scalar_bwd_dkdv<vil_bf16, MM><<<dim3(gk), dim3(64), 0, s>>>(p)); }
  vil_prof_end(s);
  if ((e = (int)hipGetLastError())) return e;
  if (p.G > 0) {
    vil_prof_begin(VIL_K_REDUCE_GLO, s, 0, 0);
    const unsigned gg = (unsigned)((p.B * p.H * p.G * p.M + 255) / 256);
    if (f32) hipLaunchKernelGGL((k_reduce_glo<float>), dim3(gg), dim3(256), 0, s, p);
    else if (f16) hipLaunchKernelGGL((k_reduce_glo<_Float16>), dim3(gg), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((k_reduce_glo<vil_bf16>), dim3(gg), dim3(256), 0, s, p);
    vil_prof_end(s);
    if ((e = (int)hipGetLastError())) return e;
  }
  const int nb = ((p.has_bias ? g.tbl * g.tbl : 0) + p.G) * p.H;
  if (nb > 0 && (p.dtable || p.dg2l)) {
    vil_prof_begin(VIL_K_REDUCE_BIAS, s, 0, 0);
    hipLaunchKernelGGL(k_reduce_bias, dim3((nb + 127) / 128), dim3(128), 0, s, p);
    vil_prof_end(s);
  }
  return (int)hipGetLastError();
}
