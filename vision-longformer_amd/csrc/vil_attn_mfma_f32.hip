// vil_attn_mfma_f32.hip -- the fp32 matrix-core family: the fused local attention with fp32 I/O on
// v_mfma_f32_16x16x4_f32 (exact fp32 products and accumulation at the fp32 vector rate, 155 TFLOP/s on the chip; no
// xf32 / tf32 rounding), for the reference's fp32 protocol (src/tests/benchmark_vil.py:171-251 and its CPU path).
// Same C ABI entry points, same bias-table image, masks and key-slot tables as the 16-bit family
// (vil_mfma_common.h); the one-query-per-lane VALU family (vil_attn_scalar.hip) stays the fallback for what this
// family declines (head_dim 8, W > 32, G > 16).  Round 4: d(bias table) / d(g2l) come from the dQ pass's fixed-point LDS
// histogram like in the 16-bit families (per-workgroup power-of-two scale, one record per workgroup, summed in a fixed
// order in double: bit-reproducible) -- they used to send the whole backward to the VALU family.
//
// Tile algebra (D = A B + C with A 16x4, B 4x16, lane l: A[l%16][l/16], B[l/16][l%16], D[4(l/16)+r][l%16]):
//   S^T (16 keys x 16 queries)  = sum over M/4 MFMAs of K-rows x Q-rows: lane (j, g) feeds key j / query j with head
//        dims g*(M/4) + t -- any bijection of the contraction index works, this one makes a lane's operand a
//        CONTIGUOUS M/4-float piece of the row (16-byte global loads, no LDS staging).
//   O^T (16 dims x 16 queries) += V^T P^T over the tile's 16 keys as 4 MFMAs: MFMA r contracts keys {4g + r}, i.e. B is
//        accumulator register r of the S^T tile as it stands (no cross-lane movement), A = V[key 4g + r][dim] is one
//        float per lane, 16 consecutive dims per 16 lanes (64-byte row segments straight from L1/L2).
// The backward passes have the same two shapes (dP^T = V dO^T like S^T; dQ^T += K^T dS^T, dV^T += dO^T P, dK^T += Q^T dS
// like O^T).  Every kernel is MFMA-bound by construction (a 16-key step of 32 queries at head_dim 64 is 64 MFMAs =
// 2048 pipe cycles), so loads are plain, un-prefetched, and hidden by two to three resident waves.
//
// Reference semantics: src/models/layers/longformer2d.py:134-204, slidingchunk_2d.py:26-246 (see include/vil_attn.h).
#include "vil_mfma_common.h"
#include <type_traits>

#define F32_LSE_PAD 1.0e30f

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t r, int byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}
__device__ __forceinline__ f32x4 buf_load4f(__amdgpu_buffer_rsrc_t r, int byte_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}

struct F32Cfg {
  int QT, HQ, NWP, units_bh, wg_per_bh, wave_lds;        // forward / dQ pass: query tiles per wave, ...
  int KT, kHQ, kNWP, kunits_bh, kwg_per_bh, kwave_lds;   // dK/dV pass: key tiles per wave, ...
  int nqs, nch, gsplit, grows;                            // streamed query slots; global-key owner units and their row share
  unsigned m_NWP, m_HQ, m_kNWP, m_kHQ, m_wgbh, m_kwgbh;
  int2* kv_slots; int* kv_nchunks;
  float* glo_parts;                                       // (B*H, gsplit, G, 2, M) partial dK / dV of the global keys
  unsigned* vnorm;                                        // [0] max_k |v_k|^2 over the whole call (float bits; k_f32_vmax)
  int* hist_parts;                                        // (dQ workgroups, 2 tabsize + 4): hi bins, lo bins, [2 tabsize] = lfx
};

// ------------------------------------------------------------------ forward
template <int MD, int QT>
__global__ __launch_bounds__(256, 2) void k_f32_fwd(VilParams p, MfmaCfg c, F32Cfg fc) {
  constexpr int M = 16 * MD, MQ = M / 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const VilGeom& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lj = lane & 15, lg = lane >> 4;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int b = fdiv(logical, fc.m_wgbh), rem_ = logical - b * (fc.wg_per_bh * p.H);
  const int wgi = fdiv(rem_, c.m_H), h = rem_ - wgi * p.H;
  const int bh = b * p.H + h;
  {
    const f32x4* src = (const f32x4*)(c.tabws + (int64_t)h * c.tabsize);
    for (int i = tid; i < (c.tabsize >> 2); i += blockDim.x) ((f32x4*)smem)[i] = src[i];
  }
  __syncthreads();
  const unsigned tab_lds = lds_addr(smem);
  int* s_koff = (int*)(smem + (size_t)c.tabsize * 4 + (size_t)wave * fc.wave_lds);
  int* s_akey = s_koff + c.NSP;
  const int unit = wgi * 4 + wave;
  if (unit >= fc.units_bh) return;
  const int Nloc = g.nx * g.ny, W = g.W;
  const int kstride_b = (int)p.k_st * 4;
  const unsigned kv_bytes = (unsigned)(p.G + Nloc - 1) * (unsigned)kstride_b + M * 4;
  const __amdgpu_buffer_rsrc_t krs = make_rsrc_n((const float*)p.k + b * p.k_sb + h * p.k_sh, kv_bytes);
  const __amdgpu_buffer_rsrc_t vrs = make_rsrc_n((const float*)p.v + b * p.v_sb + h * p.v_sh, kv_bytes);
  const float* qb = (const float*)p.q + b * p.q_sb + h * p.q_sh;
  float* ob = (float*)p.o + b * p.o_sb + h * p.o_sh;
  const float c1 = p.scale * LOG2E;

  const int ch = fdiv(unit, fc.m_NWP), wp = unit - ch * fc.NWP;
  const int cm = fdiv(ch, c.m_my), cn = ch - cm * g.my;
  const int nslots = load_key_slots(c, ch, lane, s_koff, s_akey);
  const int jj = wp * 16 + lj;
  const int qx = fdiv(jj, fc.m_HQ), qhq = jj - qx * fc.HQ;
  const unsigned aq0b = tab_lds + (min(qx, W - 1) * c.P + QT * qhq) * 4;
  int qtok[QT];
  bool qreal[QT];
  float qf[QT][MQ];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int qy = QT * qhq + qt;
    const int qr = cm * W + qx, qc = cn * W + qy;
    qreal[qt] = qx < W && qy < W && qr < g.nx && qc < g.ny;
    qtok[qt] = qreal[qt] ? qr * g.ny + qc : (cm * W) * g.ny + cn * W;
#pragma unroll
    for (int t = 0; t < MQ; t += 4) {
      const f32x4 v4 = *(const f32x4*)(qb + (int64_t)qtok[qt] * p.q_st + lg * MQ + t);
      qf[qt][t] = v4[0]; qf[qt][t + 1] = v4[1]; qf[qt][t + 2] = v4[2]; qf[qt][t + 3] = v4[3];
    }
  }
  f32x4 o[MD][QT];
  float mrow[QT], lrow[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    mrow[qt] = VIL_M_INIT; lrow[qt] = 0.f;
#pragma unroll
    for (int dt = 0; dt < MD; ++dt) o[dt][qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const int nsteps = nslots >> 4;
  for (int st = 0; st < nsteps; ++st) {
    const int koff = s_koff[st * 16 + lj] + lg * (MQ * 4);
    const i32x4 voff = *(const i32x4*)(s_koff + st * 16 + lg * 4);
    const i32x4 ak = *(const i32x4*)(s_akey + st * 16 + lg * 4);
    float kfr[MQ], vv[MD][4];
#pragma unroll
    for (int t = 0; t < MQ; t += 4) {
      const f32x4 v4 = buf_load4f(krs, koff + t * 4);
      kfr[t] = v4[0]; kfr[t + 1] = v4[1]; kfr[t + 2] = v4[2]; kfr[t + 3] = v4[3];
    }
#pragma unroll
    for (int dt = 0; dt < MD; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) vv[dt][r] = buf_load1(vrs, voff[r] + (dt * 16 + lj) * 4);
    lds_cvf tb[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) tb[r] = lds_f32(aq0b - (unsigned)ak[r]);
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      f32x4 acc = {tb[0][qt], tb[1][qt], tb[2][qt], tb[3][qt]};
#pragma unroll
      for (int t = 0; t < MQ; ++t) acc = mfma4(kfr[t], qf[qt][t], acc);
      // online softmax: the tile's 16 keys of query column j live in the 4 lanes (j, 0..3)
      float mloc = fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3]));
      mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
      const float mn = fmaxf(mrow[qt], mloc);
      const float alpha = __builtin_amdgcn_exp2f((mrow[qt] - mn) * c1);
      mrow[qt] = mn;
      f32x4 pr;
#pragma unroll
      for (int r = 0; r < 4; ++r) pr[r] = __builtin_amdgcn_exp2f((acc[r] - mn) * c1);
      lrow[qt] = lrow[qt] * alpha + ((pr[0] + pr[1]) + (pr[2] + pr[3]));
#pragma unroll
      for (int dt = 0; dt < MD; ++dt) {
        o[dt][qt] *= alpha;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[dt][qt] = mfma4(vv[dt][r], pr[r], o[dt][qt]);
      }
    }
  }
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    float l = lrow[qt];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    if (qreal[qt]) {
#pragma unroll
      for (int dt = 0; dt < MD; ++dt) *(f32x4*)(ob + (int64_t)qtok[qt] * p.o_st + dt * 16 + lg * 4) = o[dt][qt] * inv;
      if (lg == 0) p.lse[(int64_t)bh * Nloc + qtok[qt]] = mrow[qt] * p.scale + __logf(l);
    }
  }
}

// ------------------------------------------------------------------ backward, dQ pass (+ delta = rowsum(dO o O))
// HIST: dS is also accumulated into a per-workgroup LDS histogram laid out like the bias-table image (the bin of a score
// is the table entry its bias was gathered from), fixed point with the workgroup's own power-of-two scale:
// |dS| <= 2 |dO_q| |v_k| (Cauchy-Schwarz) with the maximum of |dO_q|^2 over the workgroup's queries and of |v_k|^2 over
// the call (k_f32_vmax); a bin receives at most one contribution per query.  2^lfx rides on P through lse (exact in
// fp32: a power of two) and leaves dQ in the epilogue.  fp32 tolerances need more than the 22 bits an int32 bin leaves per
// contribution (measured: 1e-3 absolute error on d(table)): a bin is TWO int32 words, x = 65536 hi + lo with
// hi = rint(x / 65536) and lo = x - 65536 hi (exact in fp32), each summed by its own ds_add_u32 -- 44-bit fixed point
// under an fp32 value's 24 significant bits.  One record per workgroup, summed by k_f32_hist_reduce.
template <int MD, int QT, bool HIST>
__global__ __launch_bounds__(256, 2) void k_f32_bwd_dq(VilParams p, MfmaCfg c, F32Cfg fc) {
  constexpr int M = 16 * MD, MQ = M / 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ unsigned s_domax;
  const VilGeom& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lj = lane & 15, lg = lane >> 4;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int b = fdiv(logical, fc.m_wgbh), rem_ = logical - b * (fc.wg_per_bh * p.H);
  const int wgi = fdiv(rem_, c.m_H), h = rem_ - wgi * p.H;
  const int bh = b * p.H + h;
  int* hist = (int*)smem + c.tabsize;                         // HIST: [table | hi bins | lo bins | per-wave slot tables]
  const unsigned hist_off = (unsigned)c.tabsize * 4u;
  {
    const f32x4* src = (const f32x4*)(c.tabws + (int64_t)h * c.tabsize);
    for (int i = tid; i < (c.tabsize >> 2); i += blockDim.x) ((f32x4*)smem)[i] = src[i];
    if (HIST) {
      for (int i = tid; i < 2 * c.tabsize; i += blockDim.x) hist[i] = 0;
      if (tid == 0) s_domax = 0u;
    }
  }
  __syncthreads();
  const unsigned tab_lds = lds_addr(smem);
  int* s_koff = (int*)(smem + (size_t)c.tabsize * (HIST ? 12 : 4) + (size_t)wave * fc.wave_lds);
  int* s_akey = s_koff + c.NSP;
  const int unit = wgi * 4 + wave;
  const bool active = unit < fc.units_bh;
  if (!HIST && !active) return;
  int lfx = 0;
  if (active) {                                                // (HIST: idle waves of the last workgroup still meet the barriers)
  const int Nloc = g.nx * g.ny, W = g.W;
  const int kstride_b = (int)p.k_st * 4;
  const unsigned kv_bytes = (unsigned)(p.G + Nloc - 1) * (unsigned)kstride_b + M * 4;
  const __amdgpu_buffer_rsrc_t krs = make_rsrc_n((const float*)p.k + b * p.k_sb + h * p.k_sh, kv_bytes);
  const __amdgpu_buffer_rsrc_t vrs = make_rsrc_n((const float*)p.v + b * p.v_sb + h * p.v_sh, kv_bytes);
  const float* qb = (const float*)p.q + b * p.q_sb + h * p.q_sh;
  const float* dob = (const float*)p.dout + b * p.do_sb + h * p.do_sh;
  const float* outb = (const float*)p.out + b * p.o_sb + h * p.o_sh;
  float* dqb = (float*)p.dq + b * p.dq_sb + h * p.dq_sh;
  const float c1 = p.scale * LOG2E;

  const int ch = fdiv(unit, fc.m_NWP), wp = unit - ch * fc.NWP;
  const int cm = fdiv(ch, c.m_my), cn = ch - cm * g.my;
  const int nslots = load_key_slots(c, ch, lane, s_koff, s_akey);
  const int jj = wp * 16 + lj;
  const int qx = fdiv(jj, fc.m_HQ), qhq = jj - qx * fc.HQ;
  const unsigned aq0b = tab_lds + (min(qx, W - 1) * c.P + QT * qhq) * 4;
  int qtok[QT];
  bool qreal[QT];
  float qf[QT][MQ], dof[QT][MQ], lse2[QT], ndlt[QT];
  float domax = 0.f;
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int qy = QT * qhq + qt;
    const int qr = cm * W + qx, qc = cn * W + qy;
    qreal[qt] = qx < W && qy < W && qr < g.nx && qc < g.ny;
    qtok[qt] = qreal[qt] ? qr * g.ny + qc : (cm * W) * g.ny + cn * W;
    float dl = 0.f, n2 = 0.f;
#pragma unroll
    for (int t = 0; t < MQ; t += 4) {
      const f32x4 a4 = *(const f32x4*)(qb + (int64_t)qtok[qt] * p.q_st + lg * MQ + t);
      const f32x4 d4 = *(const f32x4*)(dob + (int64_t)qtok[qt] * p.do_st + lg * MQ + t);
      const f32x4 o4 = *(const f32x4*)(outb + (int64_t)qtok[qt] * p.o_st + lg * MQ + t);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        qf[qt][t + e] = a4[e]; dof[qt][t + e] = d4[e]; dl = __builtin_fmaf(d4[e], o4[e], dl);
        if (HIST) n2 = __builtin_fmaf(d4[e], d4[e], n2);
      }
    }
    dl += __shfl_xor(dl, 16, 64);
    dl += __shfl_xor(dl, 32, 64);
    if (HIST) { n2 += __shfl_xor(n2, 16, 64); n2 += __shfl_xor(n2, 32, 64); domax = fmaxf(domax, qreal[qt] ? n2 : 0.f); }
    if (qreal[qt] && lg == 0) p.delta[(int64_t)bh * Nloc + qtok[qt]] = dl;        // the dK/dV pass reads it
    ndlt[qt] = qreal[qt] ? -dl : 0.f;
    lse2[qt] = qreal[qt] ? p.lse[(int64_t)bh * Nloc + qtok[qt]] * LOG2E : F32_LSE_PAD;   // padding slot: p = 0
  }
  if (HIST) {
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) domax = fmaxf(domax, __shfl_xor(domax, o, 64));
    if (lane == 0) __hip_atomic_fetch_max(&s_domax, __float_as_uint(domax), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();                                           // (the idle waves' matching barrier: below)
    const float bound = 2.0f * __builtin_sqrtf(__uint_as_float(s_domax) * __uint_as_float(fc.vnorm[0]));
    if (bound > 0.f && bound < 1e30f) {
      lfx = 44 - (int)ceilf(__log2f(bound * (float)(4 * 16 * QT)));      // <= 16 QT queries per wave, 4 waves per workgroup
      lfx = max(-60, min(60, lfx));
    }
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) if (qreal[qt]) lse2[qt] -= (float)lfx;
  }
  f32x4 dq[MD][QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt)
#pragma unroll
    for (int dt = 0; dt < MD; ++dt) dq[dt][qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int nsteps = nslots >> 4;
  for (int st = 0; st < nsteps; ++st) {
    const int koff = s_koff[st * 16 + lj] + lg * (MQ * 4);
    const i32x4 roff = *(const i32x4*)(s_koff + st * 16 + lg * 4);
    const i32x4 ak = *(const i32x4*)(s_akey + st * 16 + lg * 4);
    float kfr[MQ], vfr[MQ], kk[MD][4];
#pragma unroll
    for (int t = 0; t < MQ; t += 4) {
      const f32x4 a4 = buf_load4f(krs, koff + t * 4), b4 = buf_load4f(vrs, koff + t * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { kfr[t + e] = a4[e]; vfr[t + e] = b4[e]; }
    }
#pragma unroll
    for (int dt = 0; dt < MD; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) kk[dt][r] = buf_load1(krs, roff[r] + (dt * 16 + lj) * 4);
    lds_cvf tb[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) tb[r] = lds_f32(aq0b - (unsigned)ak[r]);
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      f32x4 acc = {tb[0][qt], tb[1][qt], tb[2][qt], tb[3][qt]};
      f32x4 dp = {ndlt[qt], ndlt[qt], ndlt[qt], ndlt[qt]};
#pragma unroll
      for (int t = 0; t < MQ; ++t) { acc = mfma4(kfr[t], qf[qt][t], acc); dp = mfma4(vfr[t], dof[qt][t], dp); }
      f32x4 ds;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ds[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(acc[r], c1, -lse2[qt])) * dp[r];
        if (HIST) {
          const float hi = __builtin_rintf(ds[r] * (1.0f / 65536.0f));
          const float lo = __builtin_fmaf(hi, -65536.0f, ds[r]);                 // exact: |lo| <= 32768
          auto* bin = lds_i32(aq0b - (unsigned)ak[r] + hist_off) + qt;
          __hip_atomic_fetch_add(bin, (int)hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_fetch_add(bin + c.tabsize, (int)lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
#pragma unroll
      for (int dt = 0; dt < MD; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) dq[dt][qt] = mfma4(kk[dt][r], ds[r], dq[dt][qt]);
    }
  }
  const float unscale = HIST ? p.scale * __builtin_amdgcn_exp2f((float)-lfx) : p.scale;
#pragma unroll
  for (int qt = 0; qt < QT; ++qt)
    if (qreal[qt]) {
#pragma unroll
      for (int dt = 0; dt < MD; ++dt) *(f32x4*)(dqb + (int64_t)qtok[qt] * p.dq_st + dt * 16 + lg * 4) = dq[dt][qt] * unscale;
    }
  } else if (HIST) {
    __syncthreads();                                           // an idle wave: the barrier after the |dO|^2 maximum
  }
  if (HIST) {
    // every wave of a workgroup derives the same lfx (same s_domax); an idle wave has none: wave 0 is always active
    __shared__ int s_lfx;
    if (tid == 0) s_lfx = lfx;
    __syncthreads();
    int* rec = fc.hist_parts + (int64_t)logical * (2 * c.tabsize + 4);
    for (int i = tid; i < 2 * c.tabsize; i += blockDim.x) rec[i] = hist[i];
    if (tid == 0) rec[2 * c.tabsize] = s_lfx;
  }
}

// max_k |v_k|^2 over every (image, head, token) row of v -- the second factor of the histogram's bound: 4 lanes per row,
// one atomic maximum (float bits: the values are non-negative) per workgroup
template <int MD>
__global__ __launch_bounds__(256) void k_f32_vmax(VilParams p, unsigned* vnorm) {
  constexpr int M = 16 * MD, MQ = M / 4;
  __shared__ float s_m[4];
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int sub = (int)(t & 3);
  const int64_t row = t >> 2, ntok = (int64_t)p.G + (int64_t)p.g.nx * p.g.ny;
  float n2 = 0.f;
  if (row < (int64_t)p.B * ntok * p.H) {
    const int h = (int)(row % p.H);
    const int64_t bt = row / p.H, tok = bt % ntok, b = bt / ntok;
    const float* vp = (const float*)p.v + b * p.v_sb + tok * p.v_st + h * p.v_sh + sub * MQ;
#pragma unroll
    for (int e = 0; e < MQ; e += 4) {
      const f32x4 v4 = *(const f32x4*)(vp + e);
      n2 = __builtin_fmaf(v4[0], v4[0], __builtin_fmaf(v4[1], v4[1], __builtin_fmaf(v4[2], v4[2], __builtin_fmaf(v4[3], v4[3], n2))));
    }
  }
  n2 += __shfl_xor(n2, 1, 64); n2 += __shfl_xor(n2, 2, 64);
#pragma unroll
  for (int o = 4; o < 64; o <<= 1) n2 = fmaxf(n2, __shfl_xor(n2, o, 64));
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = n2;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(vnorm, __float_as_uint(fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]))));
}

// d(table)[idx*H + h] and d(g2l)[h*G + g] from the dQ workgroups' records.  grid (ceil(trows * P / 64) + G, H), 1024
// threads = 64 bins x 16 record groups; the records of head h are logical workgroups j*H + h.  Every bin is summed in a
// fixed order in double with its record's scale.  d(g2l)[h][g] sums a REGION of gsz bins: the last G blocks of the grid
// own one global token each and walk its region 64 bins at a time, so that sum has a fixed order too (round 4 added one
// float atomic per 64-bin block: the order, and the last bits, varied from run to run -- ADVICE r04).  Bit-reproducible.
__global__ __launch_bounds__(1024) void k_f32_hist_reduce(VilParams p, MfmaCfg c, F32Cfg fc) {
  __shared__ double red[16][64];
  const int ntb = (c.trows * c.P + 63) >> 6;                 // blocks that own table bins
  const int h = blockIdx.y, grp = threadIdx.x >> 6, ln = threadIdx.x & 63;
  const int nrec = p.B * fc.wg_per_bh, stride = 2 * c.tabsize + 4;
  const bool owner_g = (int)blockIdx.x >= ntb;
  const int g_ = (int)blockIdx.x - ntb;
  if (owner_g && !p.dg2l) return;
  const int lo = owner_g ? c.glo0 + g_ * c.gsz : blockIdx.x * 64;
  const int hi = owner_g ? lo + c.gsz : min(lo + 64, c.trows * c.P);
  double s = 0.0;
  for (int base = lo; base < hi; base += 64) {               // (one trip for a table block)
    const int bin = base + ln;
    if (bin < hi)
      for (int j0 = grp; j0 < nrec; j0 += 64) {
        int vh[4], vl[4], lf[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int j = min(j0 + 16 * u, nrec - 1);
          const int* r = fc.hist_parts + ((int64_t)j * p.H + h) * stride;
          const bool ok = j0 + 16 * u < nrec;
          vh[u] = ok ? r[bin] : 0; vl[u] = ok ? r[c.tabsize + bin] : 0;
          lf[u] = r[2 * c.tabsize];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) s += ldexp((double)vh[u] * 65536.0 + (double)vl[u], -lf[u]);
      }
  }
  red[grp][ln] = s;
  __syncthreads();
  if (grp != 0) return;
  double t = 0.0;
#pragma unroll
  for (int u = 0; u < 16; ++u) t += red[u][ln];
  if (owner_g) {
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) t += __shfl_xor(t, o2, 64);     // (fixed butterfly: same order every run)
    if (ln == 0) p.dg2l[h * p.G + g_] = (float)t;
    return;
  }
  const int bin = lo + ln;
  if (bin < hi) {
    const int row = bin / c.P, col = bin % c.P - VIL_CPAD;
    const int dx = row - c.tcen, dy = col - c.tcen, o = p.bias_off;
    if (col >= 0 && col < c.trows && p.dtable && dx >= -o && dx <= o && dy >= -o && dy <= o)
      p.dtable[(int64_t)((dx + o) * p.bias_S + (dy + o)) * p.H + h] = (float)t;
  }
}

// ------------------------------------------------------------------ backward, dK/dV pass (one wave per 16*KT keys of a key chunk;
// the G global keys: gsplit owner units per (image, head), each over a contiguous share of the local queries)
template <int MD, int KT>
__global__ __launch_bounds__(256, 2) void k_f32_bwd_dkdv(VilParams p, MfmaCfg c, F32Cfg fc) {
  constexpr int M = 16 * MD, MQ = M / 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const VilGeom& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lj = lane & 15, lg = lane >> 4;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int b = fdiv(logical, fc.m_kwgbh), rem_ = logical - b * (fc.kwg_per_bh * p.H);
  const int wgi = fdiv(rem_, c.m_H), h = rem_ - wgi * p.H;
  const int bh = b * p.H + h;
  {
    const f32x4* src = (const f32x4*)(c.tabws + (int64_t)h * c.tabsize);
    for (int i = tid; i < (c.tabsize >> 2); i += blockDim.x) ((f32x4*)smem)[i] = src[i];
  }
  __syncthreads();
  const unsigned tab_lds = lds_addr(smem);
  char* wbase = smem + (size_t)c.tabsize * 4 + (size_t)wave * fc.kwave_lds;
  int* s_tok = (int*)wbase;
  int* s_aq = s_tok + fc.nqs;
  float* s_lse = (float*)(s_aq + fc.nqs);
  float* s_dlt = s_lse + fc.nqs;
  const int unit = wgi * 4 + wave;
  if (unit >= fc.kunits_bh) return;
  const int Nloc = g.nx * g.ny, W = g.W, W2 = g.W2;
  const int nown = fc.nch * fc.kNWP;
  const bool glo = unit >= nown;
  const int split = unit - nown;
  const int ch = glo ? 0 : fdiv(unit, fc.m_kNWP), wp = glo ? 0 : unit - ch * fc.kNWP;
  const int km = fdiv(ch, c.m_my), kn = ch - km * g.my;
  const float* lse_bh = p.lse + (int64_t)bh * Nloc;
  const float* dlt_bh = p.delta + (int64_t)bh * Nloc;
  const unsigned q_bytes = (unsigned)(Nloc - 1) * (unsigned)(p.q_st * 4) + M * 4;
  const unsigned do_bytes = (unsigned)(Nloc - 1) * (unsigned)(p.do_st * 4) + M * 4;
  const __amdgpu_buffer_rsrc_t qrs = make_rsrc_n((const float*)p.q + b * p.q_sb + h * p.q_sh, q_bytes);
  const __amdgpu_buffer_rsrc_t drs = make_rsrc_n((const float*)p.dout + b * p.do_sb + h * p.do_sh, do_bytes);
  const float* kb = (const float*)p.k + b * p.k_sb + h * p.k_sh;
  const float* vb = (const float*)p.v + b * p.v_sb + h * p.v_sh;
  const int qstride_b = (int)p.q_st * 4, dostride_b = (int)p.do_st * 4;
  const float c1 = p.scale * LOG2E;

  // ---- streamed query slots: (token, bias address) per slot from the prologue's table of this key chunk + the lse / delta
  // of this (image, head); a global-key owner unit streams the local queries [q0, q1) in token order (no table)
  int nsteps, q0 = 0, q1 = 0;
  if (!glo) {
    const int nchunks = __builtin_amdgcn_readfirstlane(fc.kv_nchunks[ch]);
    const int2* slots = fc.kv_slots + (int64_t)ch * fc.nqs;
    for (int sl = lane; sl < fc.nqs; sl += 64) {
      const int2 e = slots[sl];
      const bool real = e.x >= 0;
      const int t = max(e.x, 0);
      s_tok[sl] = real ? t : VIL_ZERO_OFF;             // padding slot: Q / dO rows read as zeros (bounded descriptors)
      s_aq[sl] = e.y;
      s_lse[sl] = real ? lse_bh[t] * LOG2E : F32_LSE_PAD;
      s_dlt[sl] = real ? dlt_bh[t] : 0.f;
    }
    wave_lds_fence();
    nsteps = (nchunks * W2 + 15) >> 4;
  } else {
    q0 = split * fc.grows; q1 = min(Nloc, q0 + fc.grows);
    nsteps = (max(q1 - q0, 0) + 15) >> 4;
  }
  // ---- this lane's key columns: column j of key tile kt is key (x, y = KT*hq + KT-1 - kt) (the bias gather walks +kt)
  const int jj = wp * 16 + lj;
  const int kx = fdiv(jj, fc.m_kHQ), khq = jj - kx * fc.kHQ;
  const unsigned akl = (unsigned)(glo ? -(c.glo0 + min(lj, max(p.G - 1, 0)) * c.gsz) * 4
                                      : (min(kx, W - 1) * c.P + KT * khq + KT - 1) * 4) - tab_lds;
  int ktok[KT];
  bool kreal[KT];
  float kfr[KT][MQ], vfr[KT][MQ];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    if (glo) {
      kreal[kt] = kt == 0 && lj < p.G;
      ktok[kt] = kreal[kt] ? lj : 0;
    } else {
      const int ky = KT * khq + KT - 1 - kt;
      const int kr = km * W + kx, kc = kn * W + ky;
      kreal[kt] = kx < W && ky < W && kr < g.nx && kc < g.ny;
      ktok[kt] = p.G + (kreal[kt] ? kr * g.ny + kc : (km * W) * g.ny + kn * W);
    }
#pragma unroll
    for (int t = 0; t < MQ; t += 4) {
      const f32x4 a4 = *(const f32x4*)(kb + (int64_t)ktok[kt] * p.k_st + lg * MQ + t);
      const f32x4 b4 = *(const f32x4*)(vb + (int64_t)ktok[kt] * p.v_st + lg * MQ + t);
#pragma unroll
      for (int e = 0; e < 4; ++e) { kfr[kt][t + e] = a4[e]; vfr[kt][t + e] = b4[e]; }
    }
  }
  f32x4 dk[MD][KT], dv[MD][KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int dt = 0; dt < MD; ++dt) { dk[dt][kt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[dt][kt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

  for (int st = 0; st < nsteps; ++st) {
    int tokA;                 // the row this lane feeds to the S / dP products (query slot j of the step)
    i32x4 tok4, aq4;
    f32x4 ls4, nd4;
    if (!glo) {
      tokA = s_tok[st * 16 + lj];
      tok4 = *(const i32x4*)(s_tok + st * 16 + lg * 4);
      aq4 = *(const i32x4*)(s_aq + st * 16 + lg * 4);
      ls4 = *(const f32x4*)(s_lse + st * 16 + lg * 4);
      nd4 = -*(const f32x4*)(s_dlt + st * 16 + lg * 4);
    } else {
      const int ta = q0 + st * 16 + lj;
      tokA = ta < q1 ? ta : VIL_ZERO_OFF;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t = q0 + st * 16 + lg * 4 + r;
        const bool ok = t < q1;
        tok4[r] = ok ? t : VIL_ZERO_OFF;
        aq4[r] = 0;
        ls4[r] = ok ? lse_bh[t] * LOG2E : F32_LSE_PAD;
        nd4[r] = ok ? -dlt_bh[t] : 0.f;
      }
    }
    const int qoffA = tokA == VIL_ZERO_OFF ? VIL_ZERO_OFF : __mul24(tokA, qstride_b) + lg * (MQ * 4);
    const int doffA = tokA == VIL_ZERO_OFF ? VIL_ZERO_OFF : __mul24(tokA, dostride_b) + lg * (MQ * 4);
    float qfr[MQ], dofr[MQ], qq[MD][4], dd[MD][4];
#pragma unroll
    for (int t = 0; t < MQ; t += 4) {
      const f32x4 a4 = buf_load4f(qrs, qoffA + t * 4), d4 = buf_load4f(drs, doffA + t * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { qfr[t + e] = a4[e]; dofr[t + e] = d4[e]; }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool z = tok4[r] == VIL_ZERO_OFF;
      const int qo = z ? VIL_ZERO_OFF : __mul24(tok4[r], qstride_b), dO = z ? VIL_ZERO_OFF : __mul24(tok4[r], dostride_b);
#pragma unroll
      for (int dt = 0; dt < MD; ++dt) {
        qq[dt][r] = buf_load1(qrs, qo + (dt * 16 + lj) * 4);
        dd[dt][r] = buf_load1(drs, dO + (dt * 16 + lj) * 4);
      }
    }
    lds_cvf tb[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) tb[r] = lds_f32((unsigned)aq4[r] - akl);
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      f32x4 acc = {tb[0][kt], tb[1][kt], tb[2][kt], tb[3][kt]};
      f32x4 dp = nd4;
#pragma unroll
      for (int t = 0; t < MQ; ++t) { acc = mfma4(qfr[t], kfr[kt][t], acc); dp = mfma4(dofr[t], vfr[kt][t], dp); }
      f32x4 pr, ds;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pr[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(acc[r], c1, -ls4[r]));
        ds[r] = pr[r] * dp[r];
      }
#pragma unroll
      for (int dt = 0; dt < MD; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dv[dt][kt] = mfma4(dd[dt][r], pr[r], dv[dt][kt]);
          dk[dt][kt] = mfma4(qq[dt][r], ds[r], dk[dt][kt]);
        }
    }
  }
  if (!glo) {
    float* dkb = (float*)p.dk + b * p.dk_sb + h * p.dk_sh;
    float* dvb = (float*)p.dv + b * p.dv_sb + h * p.dv_sh;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
      if (kreal[kt]) {
#pragma unroll
        for (int dt = 0; dt < MD; ++dt) {
          *(f32x4*)(dkb + (int64_t)ktok[kt] * p.dk_st + dt * 16 + lg * 4) = dk[dt][kt] * p.scale;
          *(f32x4*)(dvb + (int64_t)ktok[kt] * p.dv_st + dt * 16 + lg * 4) = dv[dt][kt];
        }
      }
  } else if (kreal[0]) {
    float* out = fc.glo_parts + ((((int64_t)bh * fc.gsplit + split) * p.G + lj) * 2) * M;
#pragma unroll
    for (int dt = 0; dt < MD; ++dt) {
      *(f32x4*)(out + dt * 16 + lg * 4) = dk[dt][0] * p.scale;
      *(f32x4*)(out + M + dt * 16 + lg * 4) = dv[dt][0];
    }
  }
}

// dK / dV rows of the global keys = sum of the owner units' partials: one 2M-thread... (block (b, h, g), thread = column)
__global__ void k_f32_glo_reduce(VilParams p, F32Cfg fc) {
  const int blk = blockIdx.x, gk = blk % p.G, bh = blk / p.G, b = bh / p.H, h = bh % p.H;
  const int col = threadIdx.x, M = p.M;
  if (col >= 2 * M) return;
  float s = 0.f;
  for (int r = 0; r < fc.gsplit; ++r) s += fc.glo_parts[((((int64_t)bh * fc.gsplit + r) * p.G + gk) * 2) * M + col];
  float* dst = col < M ? (float*)p.dk + b * p.dk_sb + (int64_t)gk * p.dk_st + h * p.dk_sh + col
                       : (float*)p.dv + b * p.dv_sb + (int64_t)gk * p.dv_st + h * p.dv_sh + (col - M);
  *dst = s;
}

// ===================================================================== host side
static void f32_cfg(const VilAttnDesc* d, const MfmaCfg& c, F32Cfg& fc) {
  memset(&fc, 0, sizeof(fc));
  VilGeom g; vil_geom_init(g, d->nx, d->ny, d->W, d->exact, d->mode);
  const int W = d->W;
  fc.nch = g.mx * g.my;
  fc.QT = 2;
  fc.HQ = (W + fc.QT - 1) / fc.QT;
  fc.NWP = (W * fc.HQ + 15) / 16;
  fc.units_bh = fc.nch * fc.NWP;
  fc.wg_per_bh = (fc.units_bh + 3) / 4;
  fc.wave_lds = ((c.NSP * 8 + 15) / 16) * 16;
  fc.KT = 2;
  fc.kHQ = (W + fc.KT - 1) / fc.KT;
  fc.kNWP = (W * fc.kHQ + 15) / 16;
  const int64_t nloc = (int64_t)d->nx * d->ny;
  fc.gsplit = d->G > 0 ? (int)((nloc + 1023) / 1024 < 8 ? (nloc + 1023) / 1024 : 8) : 0;
  fc.grows = fc.gsplit ? (int)(((nloc + fc.gsplit - 1) / fc.gsplit + 15) / 16 * 16) : 0;
  fc.kunits_bh = fc.nch * fc.kNWP + fc.gsplit;
  fc.kwg_per_bh = (fc.kunits_bh + 3) / 4;
  fc.nqs = (g.nact * g.W2 + 31) & ~31;
  fc.kwave_lds = fc.nqs * 16;
  fc.m_NWP = vil_magic((unsigned)fc.NWP); fc.m_HQ = vil_magic((unsigned)fc.HQ);
  fc.m_kNWP = vil_magic((unsigned)fc.kNWP); fc.m_kHQ = vil_magic((unsigned)fc.kHQ);
  fc.m_wgbh = vil_magic((unsigned)(fc.wg_per_bh * d->H)); fc.m_kwgbh = vil_magic((unsigned)(fc.kwg_per_bh * d->H));
}
static size_t f32_lds(const MfmaCfg& c, const F32Cfg& fc, bool kv, bool hist = false) {
  return (size_t)c.tabsize * (hist ? 12 : 4) + (size_t)4 * (kv ? fc.kwave_lds : fc.wave_lds);
}

int vil_f32_supported(const VilAttnDesc* d, int pass) {
  if (d->M != 16 && d->M != 32 && d->M != 48 && d->M != 64) return VIL_E_HEAD_DIM;
  if (d->W < 1 || d->W > 32) return VIL_E_WINDOW;
  if (d->G > 16) return VIL_E_BACKEND;
  // 16-byte row pieces: token / batch / head strides must keep rows 16-byte aligned
  if ((d->q_st | d->k_st | d->v_st | d->o_st | d->q_sb | d->k_sb | d->v_sb | d->o_sb | d->q_sh | d->k_sh | d->v_sh | d->o_sh) & 3) return VIL_E_ALIGN;
  const int64_t ntok = (int64_t)d->G + (int64_t)d->nx * d->ny;
  if (d->k_st != d->v_st || d->k_st >= (1 << 21) || ntok >= (1 << 23) || d->k_st * 4 * ntok >= (1ll << 31)) return VIL_E_BACKEND;
  if (pass != 0) {
    if ((d->do_st | d->do_sb | d->do_sh | d->dq_st | d->dq_sb | d->dq_sh | d->dk_st | d->dk_sb | d->dk_sh | d->dv_st | d->dv_sb | d->dv_sh) & 3) return VIL_E_ALIGN;
    for (int64_t st : {d->q_st, d->do_st})
      if (st >= (1 << 21) || st * 4 * (int64_t)d->nx * d->ny >= (1ll << 31)) return VIL_E_BACKEND;
  }
  MfmaCfg c; vil_mfma_make_cfg(d, c);
  F32Cfg fc; f32_cfg(d, c, fc);
  if (f32_lds(c, fc, pass != 0) > 160 * 1024 || f32_lds(c, fc, false, pass != 0) > 160 * 1024) return VIL_E_BACKEND;
  for (uint64_t w : {(uint64_t)fc.wg_per_bh, (uint64_t)fc.kwg_per_bh})
    if ((uint64_t)d->B * d->H * w * (w * d->H) >= (1ull << 32)) return VIL_E_BACKEND;
  return VIL_OK;
}

// floats: [delta | table images | key-slot tables | dK/dV slot tables + counts | global-key partials | v-norm word |
//          histogram records of the dQ workgroups]
static void f32_ws_layout(const VilAttnDesc* d, const MfmaCfg& c, const F32Cfg& fc, size_t off[8]) {
  const size_t rows = (size_t)d->B * d->H * d->nx * d->ny;
  off[0] = 0;
  off[1] = (rows + 3) & ~(size_t)3;
  off[2] = off[1] + (size_t)d->H * c.tabsize;
  off[3] = off[2] + vil_key_slots_floats(c, fc.nch);
  off[4] = off[3] + (size_t)fc.nch * fc.nqs * 2 + (((size_t)fc.nch + 3) & ~(size_t)3);
  off[5] = (off[4] + (size_t)d->B * d->H * fc.gsplit * d->G * 2 * d->M + 3) & ~(size_t)3;
  off[6] = off[5] + 32;
  off[7] = off[6] + (size_t)d->B * d->H * fc.wg_per_bh * (2 * c.tabsize + 4);
}
size_t vil_f32_workspace(const VilAttnDesc* d, int pass) {
  MfmaCfg c; vil_mfma_make_cfg(d, c);
  F32Cfg fc; f32_cfg(d, c, fc);
  size_t off[8]; f32_ws_layout(d, c, fc, off);
  return off[pass != 0 ? 7 : 5] * sizeof(float) + 64;
}

#define F32_SWITCH(...)                                      \
  switch (d->M) {                                            \
    case 16: { constexpr int MD_ = 1; __VA_ARGS__; } break;  \
    case 32: { constexpr int MD_ = 2; __VA_ARGS__; } break;  \
    case 48: { constexpr int MD_ = 3; __VA_ARGS__; } break;  \
    case 64: { constexpr int MD_ = 4; __VA_ARGS__; } break;  \
    default: return VIL_E_HEAD_DIM;                          \
  }

int vil_f32_fwd(const VilAttnDesc* d, VilParams& p, hipStream_t s) {
  MfmaCfg c; vil_mfma_make_cfg(d, c);
  F32Cfg fc; f32_cfg(d, c, fc);
  size_t off[8]; f32_ws_layout(d, c, fc, off);
  float* ws = (float*)p.delta;
  if (((uintptr_t)p.q | (uintptr_t)p.k | (uintptr_t)p.v | (uintptr_t)p.o | (uintptr_t)ws) & 15) return VIL_E_ALIGN;
  c.tabws = ws + off[1];
  c.key_slots = (int2*)(ws + off[2]);
  c.key_nslots = (int*)(c.key_slots + (size_t)fc.nch * c.NSP);
  const VilWork w(d);
  vil_prof_begin(VIL_K_TABLE, s, 0, 0);
  int e = vil_mfma_launch_prep(p, c, (int)p.k_st * 4, s);
  vil_prof_end(s);
  if (e) return e;
  const size_t lds = f32_lds(c, fc, false);
  vil_prof_begin(VIL_K_MFMA_FWD, s, w.fwd_bytes(), w.fwd_flops());
  F32_SWITCH({
    if (int he = vil_ensure_dyn_lds((const void*)k_f32_fwd<MD_, 2>, lds)) return he;
    k_f32_fwd<MD_, 2><<<dim3((unsigned)(p.B * p.H * fc.wg_per_bh)), dim3(256), lds, s>>>(p, c, fc);
  });
  vil_prof_end(s);
  return (int)hipGetLastError();
}

int vil_f32_bwd(const VilAttnDesc* d, VilParams& p, hipStream_t s) {
  if (p.glo_rows) return VIL_E_BACKEND;     // the fused global-query rows (vil_attn_bwd_full) are 16-bit
  MfmaCfg c; vil_mfma_make_cfg(d, c);
  F32Cfg fc; f32_cfg(d, c, fc);
  size_t off[8]; f32_ws_layout(d, c, fc, off);
  const bool hist = p.dtable != nullptr || (p.dg2l != nullptr && p.G > 0);
  float* ws = (float*)p.delta;
  if (((uintptr_t)p.q | (uintptr_t)p.k | (uintptr_t)p.v | (uintptr_t)p.dout | (uintptr_t)p.out | (uintptr_t)p.dq | (uintptr_t)p.dk |
       (uintptr_t)p.dv | (uintptr_t)ws) & 15) return VIL_E_ALIGN;
  p.delta = ws + off[0];
  c.tabws = ws + off[1];
  c.key_slots = (int2*)(ws + off[2]);
  c.key_nslots = (int*)(c.key_slots + (size_t)fc.nch * c.NSP);
  fc.kv_slots = (int2*)(ws + off[3]);
  fc.kv_nchunks = (int*)(fc.kv_slots + (size_t)fc.nch * fc.nqs);
  fc.glo_parts = ws + off[4];
  fc.vnorm = (unsigned*)(ws + off[5]);
  fc.hist_parts = (int*)(ws + off[6]);
  BwdCfg bc; memset(&bc, 0, sizeof(bc));
  bc.nch = fc.nch; bc.nsplit = 0; bc.nqs = fc.nqs; bc.kv_slots = fc.kv_slots; bc.kv_nchunks = fc.kv_nchunks;
  PrepZero zr; memset(&zr, 0, sizeof(zr));
  if (hist) {
    zr.ptr[0] = fc.vnorm; zr.n[0] = 32;
    zr.ptr[1] = (unsigned*)p.dg2l; zr.n[1] = (p.dg2l && p.G > 0) ? p.H * p.G : 0;
    // the LDS image covers only part of the caller's table: the rest of d(table) is 0
    zr.ptr[2] = (unsigned*)p.dtable; zr.n[2] = (p.dtable && c.trows < p.bias_S) ? p.bias_S * p.bias_S * p.H : 0;
    zr.total = zr.n[0] + zr.n[1] + zr.n[2];
  }
  const VilWork w(d);
  vil_prof_begin(VIL_K_TABLE, s, 0, 0);
  int e = vil_mfma_launch_prep_bwd(p, c, bc, (int)p.k_st * 4, zr, s);
  vil_prof_end(s);
  if (e) return e;
  if (hist) {
    const int64_t rows4 = (int64_t)p.B * ((int64_t)p.G + (int64_t)p.g.nx * p.g.ny) * p.H * 4;
    vil_prof_begin(VIL_K_DELTA, s, 0, 0);
    F32_SWITCH((k_f32_vmax<MD_><<<dim3((unsigned)((rows4 + 255) / 256)), dim3(256), 0, s>>>(p, fc.vnorm)));
    vil_prof_end(s);
    if ((e = (int)hipGetLastError())) return e;
  }
  vil_prof_begin(VIL_K_MFMA_DQ, s, w.dq_bytes() + w.delta_bytes(), w.dq_flops());
  {
    const size_t lds = f32_lds(c, fc, false, hist);
    const dim3 grid((unsigned)(p.B * p.H * fc.wg_per_bh));
    F32_SWITCH({
      if (hist) {
        if (int he = vil_ensure_dyn_lds((const void*)k_f32_bwd_dq<MD_, 2, true>, lds)) return he;
        k_f32_bwd_dq<MD_, 2, true><<<grid, dim3(256), lds, s>>>(p, c, fc);
      } else {
        if (int he = vil_ensure_dyn_lds((const void*)k_f32_bwd_dq<MD_, 2, false>, lds)) return he;
        k_f32_bwd_dq<MD_, 2, false><<<grid, dim3(256), lds, s>>>(p, c, fc);
      }
    });
  }
  vil_prof_end(s);
  if ((e = (int)hipGetLastError())) return e;
  vil_prof_begin(VIL_K_MFMA_DKDV, s, w.dkdv_bytes(), w.dkdv_flops());
  {
    const size_t lds = f32_lds(c, fc, true);
    F32_SWITCH({
      if (int he = vil_ensure_dyn_lds((const void*)k_f32_bwd_dkdv<MD_, 2>, lds)) return he;
      k_f32_bwd_dkdv<MD_, 2><<<dim3((unsigned)(p.B * p.H * fc.kwg_per_bh)), dim3(256), lds, s>>>(p, c, fc);
    });
  }
  vil_prof_end(s);
  if ((e = (int)hipGetLastError())) return e;
  if (p.G > 0) {
    vil_prof_begin(VIL_K_REDUCE_GLO, s, 0, 0);
    k_f32_glo_reduce<<<dim3((unsigned)(p.B * p.H * p.G)), dim3(128), 0, s>>>(p, fc);
    vil_prof_end(s);
    if ((e = (int)hipGetLastError())) return e;
  }
  if (hist) {
    vil_prof_begin(VIL_K_REDUCE_BIAS, s, 0, 0);
    k_f32_hist_reduce<<<dim3((unsigned)((c.trows * c.P + 63) / 64 + p.G), (unsigned)p.H), dim3(1024), 0, s>>>(p, c, fc);
    vil_prof_end(s);
    e = (int)hipGetLastError();
  }
  return e;
}
