// vil_attn_mfma_bwd.hip -- MFMA backward of the fused local attention (bf16 I/O,
// fp32 accumulate), FlashAttention-2 style: probabilities are recomputed from the
// saved log-sum-exp, no score tensor exists.  Two passes, no atomics on dQ/dK/dV:
//
//   k_mfma_bwd_dq    one wave per QUERY chunk, same traversal as the forward:
//                    S^T = K Q^T + bias, P^T = exp2(S^T c - lse), dP^T = V dO^T,
//                    dS^T = P^T o (dP^T - delta), dQ^T += K^T dS^T.  K is staged once per
//                    step in the wave's LDS tile and read both ways (ds_read_b128 rows for
//                    the S^T product, ds_read_b64_tr_b16 for K^T).  dS is also accumulated
//                    into a per-workgroup LDS histogram over the bias-table layout, which
//                    yields d(bias table) and d(g2l) after a small cross-workgroup reduce.
//   k_mfma_bwd_dkdv  one wave per KEY chunk (the mirror of slidingchunk_agrad's reverse
//                    rolls, slidingchunk_2d.py:156-190): streams the <= 9 query chunks that
//                    attend it, S = Q K^T + bias, P, dP = dO V^T, dS; dV^T += dO^T P,
//                    dK^T += Q^T dS.  The G global keys (attended by every local query) are
//                    extra owner units that stream all query chunks in splits and leave
//                    fp32 partials for a tiny reduce.
//
// Reference semantics: SlidingChunk2D.backward (src/models/layers/slidingchunk_2d.py:234-246)
// plus the autograd of bias gather / mask / softmax in longformer2d.py:152-200.
#include "vil_mfma_common.h"
#include <type_traits>

#define LSE_PAD 1.0e30f
#ifndef VIL_DQ_INTERLEAVE
#define VIL_DQ_INTERLEAVE 1      // key slots of the dQ pass interleaved inside groups of 8 (key_slot_pos)
#endif

// ===================================================================== dQ pass
// QT = query tiles (of 16 columns) per wave: 4 (64 query slots) for M <= 32, 2 for M >= 48 (two waves per
// SIMD instead of one; same reasoning as KT of the dK/dV pass)
// Tuning switches (see vil_attn_mfma.hip).  ViL-Small stage 1, round-1 kernel 407 us: ring 1 / 3 waves 367 us (default);
// ring 2 / 3 waves 398; ring 1 / 4 waves (128 VGPRs, 36 B scratch) 389; software pipeline (219 VGPRs, 2 waves) 460.
#ifndef VIL_DQ_PIPE
#define VIL_DQ_PIPE 0      // software pipeline over steps at head_dim 32
#endif
#ifndef VIL_DQ_PF
#define VIL_DQ_PF 1        // depth of the K / V prefetch ring at head_dim <= 32
#endif
#ifndef VIL_DQ_WAVES
#define VIL_DQ_WAVES 3     // waves per SIMD of the head_dim 32 instantiation
#endif
constexpr int dq_waves(int MD) { return MD == 2 ? VIL_DQ_WAVES : 2; }
template <typename T, int MD, int QT>
__global__ __launch_bounds__(256, dq_waves(MD)) void k_mfma_bwd_dq(VilParams p, MfmaCfg c, BwdCfg bc) {
  typedef typename V16<T>::x8 X8;
  typedef typename V16<T>::x4 X4;
  constexpr int M = 16 * MD;
  constexpr int MK = (MD + 1) / 2;
  constexpr int VCH = 2 * MD;
  constexpr bool PIPE = VIL_DQ_PIPE && MD == 2;   // software pipeline over steps (two score tiles + two LDS K tiles live; M = 16 runs QT = 4 and would spill)
  constexpr int PF = PIPE ? 2 : (MD <= 2 ? VIL_DQ_PF : 1);   // depth of the K / V prefetch ring
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const VilGeom& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lj = lane & 15, lg = lane >> 4;

  // persistent workgroup bound to one head and one XCD's unit queue: every wave walks its own list of (image, query
  // chunk) units (UnitList, vil_mfma_common.h); the workgroup leaves ONE histogram partial (64-bit bins)
  const int h = (int)(blockIdx.x >> 3) % p.H;
  __shared__ int s_started;                                   // units started by the workgroup's waves (histogram drain trigger)
  __shared__ int s_drained;                                   // ticket of the last COMPLETED drain (a multiple of hist_flush)
  if (tid == 0) { s_started = 0; s_drained = 0; }

  float* tab = (float*)smem;
  int* hist = (int*)(tab + c.tabsize);
  long long* hist64 = (long long*)(hist + c.tabsize);        // (tabsize is a multiple of 4 floats: 8-byte aligned)
  const unsigned tab_lds = lds_addr(smem);
  const unsigned hist_off = (unsigned)c.tabsize * 4u;        // histogram bin = table entry + hist_off (bytes)
  {
    const f32x4* src = (const f32x4*)(c.tabws + (int64_t)h * c.tabsize);
    for (int i = tid; i < (c.tabsize >> 2); i += blockDim.x) ((f32x4*)tab)[i] = src[i];
    if (bc.do_hist)
      for (int i = tid; i < c.tabsize; i += blockDim.x) { hist[i] = 0; hist64[i] = 0; }
  }
  __syncthreads();
  // Fixed-point scale of the bias-gradient histogram.  ds_add_f32 runs ~40x slower than
  // ds_add_u32 on gfx950 (measured: 193 vs 5-8 CU-cycles per wave instruction), so dS is
  // accumulated as int32 in units of 2^-lfx.  |dS| <= 2 max|dO.v| <= 2 |dO|_max |v|_max
  // (Cauchy-Schwarz), a bin gets <= hist_nmax contributions per workgroup, hence 2^lfx =
  // 2^30 / (hist_nmax * 2 * bound) can never overflow.  The power-of-two scale is folded into
  // lse (free) and taken out again in the dQ epilogue and the histogram reduce.
  int lfx = 0;
  if (bc.do_hist) {
    // maxima over the slots (every lane one slot, butterfly over the wave)
    float n0 = __uint_as_float(bc.norm2[32 * (tid & (VIL_NORM_SLOTS - 1))]);
    float n1 = __uint_as_float(bc.norm2[32 * (tid & (VIL_NORM_SLOTS - 1)) + 1]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { n0 = fmaxf(n0, __shfl_xor(n0, o, 64)); n1 = fmaxf(n1, __shfl_xor(n1, o, 64)); }
    const float bound = 2.0f * __builtin_sqrtf(n0 * n1);
    if (bound > 0.f && bound < 1e30f) {
      lfx = 29 - (int)ceilf(__log2f(bound * (float)bc.hist_nmax));
      lfx = max(-60, min(60, lfx));
    }
  }
  // bf16 has fp32's exponent range, so 2^lfx rides on P for free (folded into lse) and leaves dQ in the epilogue;
  // fp16 would overflow (2^lfx dS reaches 2^20): there the scale is applied to the histogram value only
  constexpr bool FOLD = !__is_same(T, _Float16);
  const float hscale = FOLD ? 1.0f : __builtin_amdgcn_exp2f((float)lfx);
  const float unscale = FOLD ? p.scale * __builtin_amdgcn_exp2f((float)-lfx) : p.scale;

  char* wbase = smem + (size_t)c.tabsize * 16 + (size_t)wave * bc.dq_wave_lds;
  int* s_koff = (int*)wbase;
  int* s_akey = s_koff + c.NSP;
  char* s_k = (char*)(s_akey + c.NSP);              // [32][M] bf16 K tile of the current step

  const int Nloc = g.nx * g.ny;
  const int kstride_b = (int)p.k_st * 2;
  const unsigned kv_bytes = (unsigned)(p.G + Nloc - 1) * (unsigned)kstride_b + M * 2;    // (zero keys of cyclic padding: vil_mfma_common.h)
  const float c1 = p.scale * LOG2E;
  const int W = g.W;

  int kst_off[MD], kld_off[MD], ktr_off[2][MD], krow_off[2][MK];
#pragma unroll
  for (int it = 0; it < MD; ++it) {
    const int cid = it * 64 + lane;
    const int row = cid / VCH, chn = cid % VCH;
    kst_off[it] = row * (M * 2) + ((chn * 16) ^ tile_swz<MD>(row));
    kld_off[it] = chn * 16;
  }
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    const int row = hf * 16 + lg * 4 + (lj >> 2);
#pragma unroll
    for (int dt = 0; dt < MD; ++dt)
      ktr_off[hf][dt] = row * (M * 2) + ((dt * 32 + (lj & 3) * 8) ^ tile_swz<MD>(row));
    const int row2 = hf * 16 + lj;
#pragma unroll
    for (int ks = 0; ks < MK; ++ks)
      krow_off[hf][ks] = row2 * (M * 2) + (((ks * 32 + lg * 8) * 2) ^ tile_swz<MD>(row2));
  }
  const int lgo = lg * 16;

  const bool lpt = g.nact == 9 && g.exact != -1;
  UnitList list;
  list.init(bc.uq_dq, p.B, p.H, wave, bc.dq_wpw);
  for (int k = 0;; ++k) {
    const int cur = list.entry(k);
    if (cur < 0) break;
    int started = 0;
    if (bc.do_hist) {
      if (lane == 0) started = __hip_atomic_fetch_add(&s_started, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      started = __builtin_amdgcn_readfirstlane(started);
    }
    if (bc.do_hist) {
      // The bound behind the fixed-point scale (hist_nmax = per-unit contributions x (hist_flush + 2 * waves)) is ENFORCED,
      // not assumed: a wave may not start unit `started` while the last completed drain lies hist_flush + waves or more
      // tickets back.  Then a bin holds, between two exchanges, at most the units started since the last completed drain
      // (< hist_flush + waves) plus the units that were already running when their bins were exchanged (< waves).
      // Drains are thereby serialised (ticket k * flush waits for drain (k-1) * flush, whose owner never waits), so there
      // is no deadlock; in practice a drain is ~25 loop trips against ~40 k cycles per unit and the wait never spins.
      while (started - __hip_atomic_load(&s_drained, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= bc.hist_flush + bc.dq_wpw)
        __builtin_amdgcn_s_sleep(4);
    }
    if (bc.do_hist && started > 0 && started % bc.hist_flush == 0) {
      // Every hist_flush units the workgroup starts, the wave that starts that unit first drains the 32-bit bins into the
      // workgroup's 64-bit bins: an atomic exchange per bin, so the other waves keep adding meanwhile and nothing is lost.
      for (int i = lane; i < c.tabsize; i += 64) {
        const int v = __hip_atomic_exchange(hist + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (v) __hip_atomic_fetch_add(hist64 + i, (long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      wave_lds_fence();
      if (lane == 0) __hip_atomic_store(&s_drained, started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    const int b = fdiv(cur, bc.uq_dq.m_units_bh), urank = cur - b * bc.uq_dq.units_bh;
    const int bh = b * p.H + h;
    const T* qb = (const T*)p.q + b * p.q_sb + h * p.q_sh;
    const __amdgpu_buffer_rsrc_t krs = make_rsrc_n((const T*)p.k + b * p.k_sb + h * p.k_sh, kv_bytes);
    const __amdgpu_buffer_rsrc_t vrs = make_rsrc_n((const T*)p.v + b * p.v_sb + h * p.v_sh, kv_bytes);
    const T* dob = (const T*)p.dout + b * p.do_sb + h * p.do_sh;
    T* dqb = (T*)p.dq + b * p.dq_sb + h * p.dq_sh;
    {
      const int rk = fdiv(urank, bc.m_dq_NWP), wp = urank - rk * bc.dq_NWP;
      const int ch = lpt ? chunk_of_rank(rk, g.mx, g.my) : rk;
      const int unit = ch * bc.dq_NWP + wp;
      const int cm = fdiv(ch, c.m_my), cn = ch - cm * g.my;
      const int jj = wp * 16 + lj;
      const int qx = fdiv(jj, bc.m_dq_HQ), qhq = jj - qx * bc.dq_HQ;
      const unsigned aq0b = tab_lds + (min(qx, W - 1) * c.P + QT * qhq) * 4;
      int qtok[QT];
      bool qreal[QT];
      float lse2[QT], ndlt[QT];
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        const int qy = QT * qhq + qt;
        const int qr = cm * W + qx, qc = cn * W + qy;
        qreal[qt] = qx < W && qy < W && qr < g.nx && qc < g.ny;
        qtok[qt] = qreal[qt] ? qr * g.ny + qc : (cm * W) * g.ny + cn * W;
        // a non-existent query slot gets lse = +big: its probabilities (hence dS) are exactly 0
        lse2[qt] = qreal[qt] ? p.lse[(int64_t)bh * Nloc + qtok[qt]] * LOG2E - (FOLD ? (float)lfx : 0.f) : LSE_PAD;
        ndlt[qt] = qreal[qt] ? -p.delta[(int64_t)bh * Nloc + qtok[qt]] : 0.f;
      }
      X8 qf[MK][QT], dof[MK][QT];
#pragma unroll
      for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int ks = 0; ks < MK; ++ks) {
          const int d0 = ks * 32 + lg * 8;
          X8 z = {};
          qf[ks][qt] = d0 < M ? *(const X8*)(qb + (int64_t)qtok[qt] * p.q_st + d0) : z;
          dof[ks][qt] = d0 < M ? *(const X8*)(dob + (int64_t)qtok[qt] * p.do_st + d0) : z;
        }
      // (requested after this wave's own rows, so that the table's round trip overlaps theirs: see k_mfma_fwd)
      const int nslots = load_key_slots(c, ch, lane, s_koff, s_akey);
      if (bc.glo_from_dq) {
        // dK / dV of the G global keys as a by-product of this pass: the wave holds the Q / dO rows, lse and delta of its
        // queries; its share of dK_g = scale * sum_q dS[q,g] Q[q] and dV_g = sum_q P[q,g] dO[q] is ~250 VALU instructions
        // once per unit, done here while few registers are live (scores by dot products, as the global QUERY rows of
        // the dK/dV pass do).  It replaces owner units of the dK/dV pass that streamed every query chunk through
        // 64-key MFMA tiles with one live key column (11 % of that pass's units at 8x8 chunks).
        const float lfx2 = FOLD ? __builtin_amdgcn_exp2f((float)-lfx) : 1.0f;      // lse2 carries -lfx in the bf16 build
        float* rec = bc.glo_parts + (((int64_t)bh * bc.glo_nrec + unit) * p.G) * 2 * M;
#pragma unroll 1
        for (int r = 0; r < p.G; ++r) {
          X8 kg[MK], vg[MK];
#pragma unroll
          for (int ks = 0; ks < MK; ++ks) {
            X8 z = {};
            const bool ok = (ks * 32 + lg * 8) < M;
            kg[ks] = ok ? buf_load8<T>(krs, r * kstride_b + lgo + ks * 64) : z;
            vg[ks] = ok ? buf_load8<T>(vrs, r * kstride_b + lgo + ks * 64) : z;
          }
          const float bias_g = tab[c.glo0 + r * c.gsz];                    // g2l[h][g] / scale (0 without rpe)
          float ak_[MK][8], av_[MK][8];
#pragma unroll
          for (int ks = 0; ks < MK; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) { ak_[ks][e] = 0.f; av_[ks][e] = 0.f; }
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) {
            float sc = 0.f, dp = 0.f;
#pragma unroll
            for (int ks = 0; ks < MK; ++ks)
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                sc = __builtin_fmaf((float)qf[ks][qt][e], (float)kg[ks][e], sc);
                dp = __builtin_fmaf((float)dof[ks][qt][e], (float)vg[ks][e], dp);
              }
            sc += __shfl_xor(sc, 16, 64); sc += __shfl_xor(sc, 32, 64);
            dp += __shfl_xor(dp, 16, 64); dp += __shfl_xor(dp, 32, 64);
            const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(sc + bias_g, c1, -lse2[qt])) * lfx2;   // 0 for padding slots
            const float ds = pr * (dp + ndlt[qt]);
#pragma unroll
            for (int ks = 0; ks < MK; ++ks)
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                ak_[ks][e] = __builtin_fmaf(ds, (float)qf[ks][qt][e], ak_[ks][e]);
                av_[ks][e] = __builtin_fmaf(pr, (float)dof[ks][qt][e], av_[ks][e]);
              }
          }
#pragma unroll
          for (int o = 1; o < 16; o <<= 1)
#pragma unroll
            for (int ks = 0; ks < MK; ++ks)
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                ak_[ks][e] += __shfl_xor(ak_[ks][e], o, 64);
                av_[ks][e] += __shfl_xor(av_[ks][e], o, 64);
              }
          if (lj == 0) {
#pragma unroll
            for (int ks = 0; ks < MK; ++ks) {
              const int d0 = ks * 32 + lg * 8;
              if (d0 < M) {
                float* rk = rec + (r * 2) * M + d0;
                *(f32x4*)rk = (f32x4){ak_[ks][0], ak_[ks][1], ak_[ks][2], ak_[ks][3]} * p.scale;
                *(f32x4*)(rk + 4) = (f32x4){ak_[ks][4], ak_[ks][5], ak_[ks][6], ak_[ks][7]} * p.scale;
                *(f32x4*)(rk + M) = (f32x4){av_[ks][0], av_[ks][1], av_[ks][2], av_[ks][3]};
                *(f32x4*)(rk + M + 4) = (f32x4){av_[ks][4], av_[ks][5], av_[ks][6], av_[ks][7]};
              }
            }
          }
        }
      }
      f32x4 dq[MD][QT];
#pragma unroll
      for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int dt = 0; dt < MD; ++dt) dq[dt][qt] = (f32x4){0.f, 0.f, 0.f, 0.f};

      const int nsteps = nslots >> 5;
      X8 vf[PF][2][MK];
      u32x4 kr_[PF][MD];
      auto load_step = [&](auto slot_, int st) {
        constexpr int sl = decltype(slot_)::value;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const int off = s_koff[st * 32 + hf * 16 + lj] + lgo;
#pragma unroll
          for (int ks = 0; ks < MK; ++ks) {
            X8 z = {};
            vf[sl][hf][ks] = (ks * 32 + lg * 8) < M ? buf_load8<T>(vrs, off + ks * 64) : z;
          }
        }
#pragma unroll
        for (int it = 0; it < MD; ++it) {
          const int row = (it * 64 + lane) / VCH;
          kr_[sl][it] = __builtin_amdgcn_raw_buffer_load_b128(krs, s_koff[st * 32 + row] + kld_off[it], 0, 0);
        }
      };

      // stage_s(st): K tile of the step -> LDS tile of parity st & 1 (read both ways below), V fragments stay in
      // registers, the ring slot is refilled; S^T = K Q^T + bias and dP^T = V dO^T are issued
      auto stage_s = [&](auto slot_, int st, f32x4 (&sacc)[2][QT], f32x4 (&dpacc)[2][QT], unsigned (&i0)[2][4]) {
        constexpr int sl = decltype(slot_)::value;
        char* sk = s_k + (PIPE ? (st & 1) * (32 * M * 2) : 0);
        X8 vc[2][MK];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
          for (int ks = 0; ks < MK; ++ks) vc[hf][ks] = vf[sl][hf][ks];
#pragma unroll
        for (int it = 0; it < MD; ++it) *(u32x4*)(sk + kst_off[it]) = kr_[sl][it];
        i32x4 ak[2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) ak[hf] = *(const i32x4*)(s_akey + st * 32 + hf * 16 + lg * 4);
        if (st + PF < nsteps) load_step(slot_, st + PF);
        wave_lds_fence();
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          // K rows of this half as the A operand (natural layout, from the LDS tile)
          X8 kc_[MK];
#pragma unroll
          for (int ks = 0; ks < MK; ++ks) {
            X8 z = {};
            kc_[ks] = (ks * 32 + lg * 8) < M ? *(const X8*)(sk + krow_off[hf][ks]) : z;
          }
          lds_cvf tb[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            i0[hf][r] = aq0b - (unsigned)ak[hf][r];
            tb[r] = lds_f32(i0[hf][r]);
          }
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) {
            f32x4 acc = {tb[0][qt], tb[1][qt], tb[2][qt], tb[3][qt]};
            f32x4 dp = {ndlt[qt], ndlt[qt], ndlt[qt], ndlt[qt]};      // dP - delta: the subtraction rides in the accumulator
#pragma unroll
            for (int ks = 0; ks < MK; ++ks) {
              acc = mfma16(kc_[ks], qf[ks][qt], acc);
              dp = mfma16(vc[hf][ks], dof[ks][qt], dp);
            }
            sacc[hf][qt] = acc; dpacc[hf][qt] = dp;
          }
        }
      };
      // dS^T = P^T o (dP^T - delta) (+ the bias-gradient histogram), then dQ^T += K^T dS^T
      auto finish = [&](int st, const f32x4 (&sacc)[2][QT], const f32x4 (&dpacc)[2][QT], const unsigned (&i0)[2][4]) {
        const char* sk = s_k + (PIPE ? (st & 1) * (32 * M * 2) : 0);
        u32x4 dsw[QT];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) {
            float ds[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[hf][qt][r], c1, -lse2[qt]));
              ds[r] = pr * dpacc[hf][qt][r];
              if (bc.do_hist)
                __hip_atomic_fetch_add(lds_i32(i0[hf][r] + hist_off) + qt, f2i_rpi(FOLD ? ds[r] : ds[r] * hscale),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            dsw[qt][hf * 2] = pack2<T>((f32x2){ds[0], ds[1]});
            dsw[qt][hf * 2 + 1] = pack2<T>((f32x2){ds[2], ds[3]});
          }
        X8 dsb[QT];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) dsb[qt] = __builtin_bit_cast(X8, dsw[qt]);
#pragma unroll
        for (int dt = 0; dt < MD; ++dt) {
          X8 kt_;
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            const s16x4 t4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (s16x4 __attribute__((address_space(3)))*)(sk + ktr_off[hf][dt]));
            const X4 tb = __builtin_bit_cast(X4, t4);
#pragma unroll
            for (int e = 0; e < 4; ++e) kt_[hf * 4 + e] = tb[e];
          }
#pragma unroll
          for (int qt = 0; qt < QT; ++qt)
            dq[dt][qt] = mfma16(kt_, dsb[qt], dq[dt][qt]);
        }
      };

      typedef std::integral_constant<int, 0> S0;
      typedef std::integral_constant<int, PF - 1> S1;
      load_step(S0{}, 0);
      if constexpr (PF == 2) { if (nsteps > 1) load_step(S1{}, 1); }
      if constexpr (PIPE) {
        // software pipeline over steps (see k_mfma_fwd): the score / dP MFMAs of step st+1 are issued before the
        // VALU work of step st
        f32x4 sA[2][QT], dA[2][QT], sB[2][QT], dB[2][QT];
        unsigned iA[2][4], iB[2][4];
        stage_s(S0{}, 0, sA, dA, iA);
        for (int st = 0; st < nsteps; st += 2) {
          if (st + 1 < nsteps) stage_s(S1{}, st + 1, sB, dB, iB);
          finish(st, sA, dA, iA);
          if (st + 1 < nsteps) {
            if (st + 2 < nsteps) stage_s(S0{}, st + 2, sA, dA, iA);
            finish(st + 1, sB, dB, iB);
          }
        }
      } else {
        auto one = [&](auto slot_, int st) {
          f32x4 sA[2][QT], dA[2][QT];
          unsigned iA[2][4];
          stage_s(slot_, st, sA, dA, iA);
          finish(st, sA, dA, iA);
          wave_lds_fence();
        };
        for (int st = 0; st < nsteps; st += PF) {
          one(S0{}, st);
          if constexpr (PF == 2) { if (st + 1 < nsteps) one(S1{}, st + 1); }
        }
      }
#pragma unroll
      for (int qt = 0; qt < QT; ++qt)
        if (qreal[qt]) {
#pragma unroll
          for (int dt = 0; dt < MD; ++dt) {
            X4 w;
#pragma unroll
            for (int r = 0; r < 4; ++r) w[r] = (T)(dq[dt][qt][r] * unscale);
            *(X4*)(dqb + (int64_t)qtok[qt] * p.dq_st + dt * 16 + lg * 4) = w;
          }
        }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (bc.do_hist) {
    __syncthreads();
    // record w of head h (w = 0 .. dq_nwg / H - 1): layout reduce_hist_block reads
    const int per = bc.dq_nwg / p.H, w = (int)((blockIdx.x >> 3) / p.H) * 8 + (int)(blockIdx.x & 7);
    long long* out = (long long*)bc.hist_parts + ((int64_t)h * per + w) * c.tabsize;
    for (int i = tid; i < c.tabsize; i += blockDim.x) out[i] = hist64[i] + (long long)hist[i];
    if (blockIdx.x == 0 && tid == 0) ((int*)bc.norm2)[2] = lfx;      // the reduce needs the scale
  }
}

// d(table)[idx*H+h] and d(g2l)[h*G+g] from the per-workgroup histograms.
// One 1024-thread workgroup per (64 bins, head) = 64 bins x 16 partial groups (4 independent loads in flight each);
// a role of k_mfma_post_bwd.
__device__ __forceinline__ void reduce_hist_block(const VilParams& p, const MfmaCfg& c, const BwdCfg& bc, int bx, int h) {
  __shared__ long long red[16][64];
  const int bin = bx * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
  long long si = 0;
  if (bin < c.tabsize) {
    // the records of head h's (persistent) workgroups: (h * n + w) * tabsize, n = dq_nwg / H
    const int n = bc.dq_nwg / p.H;
    const long long* parts = (const long long*)bc.hist_parts + (int64_t)h * n * c.tabsize + bin;
    int i = grp;
    for (; i + 48 < n; i += 64) {
      long long v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = parts[(int64_t)(i + 16 * u) * c.tabsize];
      si += (v[0] + v[1]) + (v[2] + v[3]);
    }
    for (; i < n; i += 16) si += parts[(int64_t)i * c.tabsize];
  }
  red[grp][threadIdx.x & 63] = si;
  __syncthreads();
  if (grp == 0) {            // one whole wave: the 64 bins of this block
    si = 0;
#pragma unroll
    for (int u = 0; u < 16; ++u) si += red[u][threadIdx.x];
    const int lfx = ((const int*)bc.norm2)[2];
    const float s = (float)si * __builtin_amdgcn_ldexpf(1.0f, -lfx) ;
    const int tbl = c.trows;
    int gg = -1;
    if (bin < tbl * c.P) {
      const int row = bin / c.P, col = bin % c.P - VIL_CPAD;
      const int dx = row - c.tcen, dy = col - c.tcen, o = p.bias_off;
      if (col >= 0 && col < tbl && p.dtable && dx >= -o && dx <= o && dy >= -o && dy <= o)
        p.dtable[(int64_t)((dx + o) * p.bias_S + (dy + o)) * p.H + h] = s;
    } else if (bin >= c.glo0 && bin < c.tabsize && p.dg2l) {
      gg = (bin - c.glo0) / c.gsz;
    }
    // every bin of global token g's constant region adds into ONE word: reduce over the wave first
    // (a thousand same-address atomics per head cost ~80 us)
    for (int g_ = 0; g_ < p.G; ++g_) {
      if (!__any(gg == g_)) continue;
      float v = gg == g_ ? s : 0.f;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      if (threadIdx.x == 0 && v != 0.f) atomicAdd(&p.dg2l[h * p.G + g_], v);
    }
  }
}

// ===================================================================== dK/dV pass
// KT = key tiles (of 16 columns) per wave: 4 (64 keys) for M <= 32; 2 (32 keys) for M >= 48, where 64 keys'
// accumulators (128 registers) + K/V fragments (64) pinned the kernel at one latency-bound wave per SIMD.
// ViL-Small stage 1, round-1 kernel 427 us: 4 key tiles / 2 waves 411 us (default); 2 key tiles per wave (two waves per
// 7x7 chunk, each streaming the same Q / dO rows) 510 at 3 waves/SIMD, 506 pipelined, 710 at 4 waves; 4 key tiles held
// to 168 VGPRs spill 404 B and run 1195 us.
#ifndef VIL_KV_KT32
#define VIL_KV_KT32 4      // key tiles per wave at head_dim 32 (4: one wave per 7x7 chunk, 248 registers)
#endif
#ifndef VIL_KV_PIPE
#define VIL_KV_PIPE 0      // software pipeline over steps (head_dim 32 with 2 key tiles per wave only)
#endif
#ifndef VIL_KV_PF
#define VIL_KV_PF 1        // depth of the Q / dO prefetch ring at head_dim 32
#endif
#ifndef VIL_KV_WAVES
#define VIL_KV_WAVES 2     // waves per SIMD of the head_dim 32 instantiation
#endif
#ifndef VIL_KV_ABL_NOTAIL
#define VIL_KV_ABL_NOTAIL 0     // timing ablation only (wrong results): skip the global queries' dq tail
#endif
// Streamed-query slot tables of the dK/dV pass.  Which query rows a key chunk is attended by, and the bias-table address
// term of each, depend on the chunk position only -- not on the (image, head) -- so one wave per key
// chunk (and per global-key split) builds the table ONCE per call (a role of k_mfma_prep_bwd); the dK/dV waves used to rebuild it per (image, head,
// chunk): ~800 instructions, a fifth of a wave's lifetime at ViL-Small stage 1 (tools/kv_timing.py).
//   kv_slots[t][s] = (token, 4 * bias address term)   token = -1: padding slot;   kv_nchunks[t] = query chunks streamed
__device__ __forceinline__ void kv_slots_block(const VilParams& p, const MfmaCfg& c, const BwdCfg& bc, int t, int lane,
                                               char* smem) {
  const VilGeom& g = p.g;
  const bool glo = t >= bc.nch;
  const int split = t - bc.nch, ch = glo ? 0 : t;
  const int kn = ch % g.my, km = ch / g.my;
  const int W = g.W, W2 = g.W2;
  int* s_tok = (int*)smem;
  int* s_aq = s_tok + bc.nqs;
  int adr1, adc1;
  shift_neighbour(p, adr1, adc1);
  for (int s = lane; s < bc.nqs; s += 64) { s_tok[s] = -1; s_aq[s] = glo ? 0 : c.aconst * 4; }
  wave_lds_fence();
  int nchunks;
  if (glo) {
    nchunks = bc.glo_from_dq ? 0 : (bc.nch - split + bc.nsplit - 1) / bc.nsplit;
  } else if (p.only_glo) {
    nchunks = 0;                                   // local keys are attended by nobody
  } else if (g.exact == -1) {
    nchunks = g.nact;                              // cyclic: every neighbour offset reaches a chunk (by wrap-around)
  } else {
    nchunks = 0;
    for (int a = 0; a < g.nact; ++a) {
      const int ar = g.nact == 2 ? (a == 0 ? 0 : adr1) : g.adr[a], ac = g.nact == 2 ? (a == 0 ? 0 : adc1) : g.adc[a];
      const int m_ = km - ar, n_ = kn - ac;
      nchunks += (m_ >= 0 && m_ < g.mx && n_ >= 0 && n_ < g.my);
    }
  }
  for (int rid = lane; rid < nchunks * W; rid += 64) {      // one neighbourhood row per lane
    const int ci = fdiv(rid, c.magicW), xl = rid - ci * W;
    int qm = -1, qn = -1, dr = 0, dc = 0;
    if (glo) {
      const int qch = split + ci * bc.nsplit;
      qm = qch / g.my; qn = qch - qm * g.my;
    } else {
      int cnt = 0;
      for (int a = 0; a < g.nact; ++a) {
        const int a3 = (a * 11) >> 5;
        const int ar = g.nact == 9 ? a3 - 1 : (a == 0 ? 0 : adr1);
        const int ac = g.nact == 9 ? a - 3 * a3 - 1 : (a == 0 ? 0 : adc1);
        int m_ = km - ar, n_ = kn - ac;
        bool ok = m_ >= 0 && m_ < g.mx && n_ >= 0 && n_ < g.my;
        if (g.exact == -1) {                     // cyclic: the query chunk that reaches this key chunk through (ar, ac)
          m_ = m_ < 0 ? m_ + g.mx : (m_ >= g.mx ? m_ - g.mx : m_);
          n_ = n_ < 0 ? n_ + g.my : (n_ >= g.my ? n_ - g.my : n_);
          ok = true;
        }
        if (ok && cnt == ci) { qm = m_; qn = n_; dr = ar; dc = ac; }
        cnt += ok;
      }
    }
    const int qr = qm * W + xl;
    if (qm >= 0 && qr < g.nx) {
      const int qc0 = qn * W;
      const int nvalid = min(W, g.ny - qc0);
      int tok = qr * g.ny + qc0;
      int aq = glo ? 0 : ((xl - dr * W) * c.P - dc * W + c.aconst) * 4;
      // kv_gspare: bit 0 marks the slots of the key chunk's OWN query chunk -- the only ones that count for the global
      // key riding in the unit's spare column (every query must meet the global key exactly once)
      if (bc.kv_gspare && !glo && dr == 0 && dc == 0) aq |= 1;
      int s = ci * W2 + xl * W;
      for (int yl = 0; yl < nvalid; ++yl) {
        s_tok[s] = tok; s_aq[s] = aq;
        ++s; ++tok; aq += glo ? 0 : 4;
      }
    }
  }
  wave_lds_fence();
  // vil_attn_bwd_full: the G global QUERY rows close the stream -- the last G slots of its last 32-slot step (fixed rows
  // 31 - g, which the dK/dV waves rely on).  token = -(2 + g); address term: local-key columns land in region g2l0[g]
  // whatever their key term (<= kv_span), an owner unit's global-key columns (lane gk: -(glo0 + gk * gsz)) in g2g[g][gk].
  // Only the first owner unit streams them (the pair (global query, global key) is counted once).
  if (p.glo_rows && (!glo || split == 0)) {
    const int last = (((nchunks * W2 + p.G + 31) >> 5) << 5) - 1;
    if (lane < p.G) {
      s_tok[last - lane] = -(2 + lane);
      s_aq[last - lane] = glo ? (c.tabsize + (p.G + lane * p.G) * c.gsz - c.glo0) * 4
                              : (c.tabsize + lane * c.gsz + bc.kv_span) * 4;
    }
    wave_lds_fence();
  }
  int2* out = bc.kv_slots + (int64_t)t * bc.nqs;
  for (int s = lane; s < bc.nqs; s += 64) out[s] = make_int2(s_tok[s], s_aq[s]);
  if (lane == 0) bc.kv_nchunks[t] = nchunks;
}

// Backward prologue, ONE launch of 256-thread workgroups with four roles (see k_mfma_prep): bias-table images, key-slot
// tables (dQ pass), streamed-query slot tables (dK/dV pass), and the words that must be zero before the passes run.
// (Kernels instead of hipMemsetAsync for the zeroing: memset nodes captured into a hipGraph were observed to run out of
// order with their neighbouring kernel nodes on replay, tools/graph_op_check.py.)
__global__ __launch_bounds__(256) void k_mfma_prep_bwd(VilParams p, MfmaCfg c, BwdCfg bc, int row_stride_b, int ntx, PrepZero zr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int ntab = ntx * p.H, nkv = bc.nch + bc.nsplit;
  int blk = blockIdx.x;
  if (blk < ntab) {
    const int h = blk / ntx, bx = blk - h * ntx;
    table_element(p, c, (float*)c.tabws, h, bx * 256 + threadIdx.x, 1.0f / p.scale);
    return;
  }
  blk -= ntab;
  if (blk < bc.nch) {
    if (threadIdx.x < 64) key_slots_block(p, c, blk, threadIdx.x, row_stride_b, smem, VIL_DQ_INTERLEAVE != 0);
    return;
  }
  blk -= bc.nch;
  if (blk < nkv) {
    if (threadIdx.x < 64) kv_slots_block(p, c, bc, blk, threadIdx.x, smem);
    return;
  }
  blk -= nkv;
  int i = blk * 256 + threadIdx.x;
#pragma unroll
  for (int r = 0; r < 5; ++r) {
    if (i < zr.n[r]) { zr.ptr[r][i] = 0u; return; }
    i -= zr.n[r];
  }
}

constexpr int kv_waves(int MD) { return MD == 2 ? VIL_KV_WAVES : 2; }
template <typename T, int MD, int KT, int EPRE>
__global__ __launch_bounds__(256, kv_waves(MD)) void k_mfma_bwd_dkdv(VilParams p, MfmaCfg c, BwdCfg bc) {
  typedef typename V16<T>::x8 X8;
  typedef typename V16<T>::x4 X4;
  constexpr int M = 16 * MD;
  constexpr int MK = (MD + 1) / 2;
  constexpr int VCH = 2 * MD;
  constexpr bool PIPE = VIL_KV_PIPE && MD == 2 && KT == 2;   // software pipeline over steps (two score / dP tile sets, two LDS Q / dO tiles)
  constexpr int PF = PIPE ? 2 : (MD == 2 ? VIL_KV_PF : 1);   // depth of the Q / dO prefetch ring
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const VilGeom& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lj = lane & 15, lg = lane >> 4;

  // (image, workgroup-of-chunks, head) order: see k_mfma_fwd.  (Round 4 tried persistent workgroups here as in the dQ
  // pass: same time, but 1.8x instead of 1.5x the algorithmic HBM bytes at 96x96 -- profiles/r04_queue_ablation.txt.)
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int b = fdiv(logical, bc.m_kv_wgbh), rem_ = logical - b * (bc.kv_wg_per_bh * p.H);
  const int wgi = fdiv(rem_, c.m_H), h = rem_ - wgi * p.H;
  const int bh = b * p.H + h;
  const unsigned tab_lds = lds_addr(smem);
  const int Gq = p.glo_rows ? p.G : 0;            // global QUERY rows riding in the stream (vil_attn_bwd_full)
  const int tabx = c.tabsize + bc.kv_xsize;       // bias image + the global queries' constant regions

  float* tab = (float*)smem;
  char* wbase = smem + (size_t)tabx * 4 + (size_t)wave * bc.kv_wave_lds;
  const int nqsa = max((bc.nqs + 63) & ~63, EPRE * 64);   // (arrays hold whole rounds of 64 slots, at least the EPRE rounds the prologue fills without bounds)
  int* s_tok = (int*)wbase;                       // [nqsa] row index of each streamed query slot in the Q / dO descriptors
  int* s_aq = s_tok + nqsa;                       // [nqsa] bias-table address term (bytes)
  float* s_lse = (float*)(s_aq + nqsa);           // [nqsa] lse * log2(e)   (+big for padding slots)
  float* s_dlt = s_lse + nqsa;                    // [nqsa] delta
  char* s_q = (char*)(s_dlt + nqsa);              // [32][M] bf16 Q tile, then the [32][M] dO tile (PIPE: two such pairs)
  float* s_gd = (float*)(s_q + (PIPE ? 4 : 2) * 32 * M * 2);   // [1 or 4][64][4] dS of the step's last query rows 31 - g (the global queries' in the last step)
  int* s_aqg = (int*)(s_gd + (p.G > 1 ? 4 : 1) * 64 * 4);      // [nqs] (kv_gspare) address term of the GLOBAL key's column, relative to its g2l region:
                                                  //       the query's position in the own chunk ((x * P + y) * 4 -- the region is as wide
                                                  //       as that range, like the forward's global key slots), or, for every slot of another
                                                  //       chunk, the distance to the all-masked guard region

  const int Nloc = g.nx * g.ny;
  const int qstride_b = (int)p.q_st * 2, dostride_b = (int)p.do_st * 2;
  const float c1 = p.scale * LOG2E;
  const int W = g.W, W2 = g.W2;
  const int nown = bc.nch * bc.kv_NWP;

  int st_off[MD], ld_off[MD], tr_off[2][MD], row_off[2][MK];
#pragma unroll
  for (int it = 0; it < MD; ++it) {
    const int cid = it * 64 + lane;
    const int row = cid / VCH, chn = cid % VCH;
    st_off[it] = row * (M * 2) + ((chn * 16) ^ tile_swz<MD>(row));
    ld_off[it] = chn * 16;
  }
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    const int row = hf * 16 + lg * 4 + (lj >> 2);
#pragma unroll
    for (int dt = 0; dt < MD; ++dt)
      tr_off[hf][dt] = row * (M * 2) + ((dt * 32 + (lj & 3) * 8) ^ tile_swz<MD>(row));
    const int row2 = hf * 16 + lj;
#pragma unroll
    for (int ks = 0; ks < MK; ++ks)
      row_off[hf][ks] = row2 * (M * 2) + (((ks * 32 + lg * 8) * 2) ^ tile_swz<MD>(row2));
  }

  // With global-query rows the Q / dO descriptors start at token 0 of the all-token tensors (the G global rows; the local
  // rows follow them: vil_attn_bwd_full), and a streamed slot's row index is G + its local token, or g.
  const __amdgpu_buffer_rsrc_t qrs = make_rsrc((Gq ? (const T*)p.q_g : (const T*)p.q) + b * p.q_sb + h * p.q_sh);
  const __amdgpu_buffer_rsrc_t drs = make_rsrc((Gq ? (const T*)p.do_g : (const T*)p.dout) + b * p.do_sb + h * p.do_sh);
  const T* kb = (const T*)p.k + b * p.k_sb + h * p.k_sh;
  const T* vb = (const T*)p.v + b * p.v_sb + h * p.v_sh;
  T* dkb = (T*)p.dk + b * p.dk_sb + h * p.dk_sh;
  T* dvb = (T*)p.dv + b * p.dv_sb + h * p.dv_sh;
  const float* lse_bh = p.lse + (int64_t)bh * Nloc;
  const float* dlt_bh = p.delta + (int64_t)bh * Nloc;
  for (int gi = 0; gi < bc.kv_gpw; ++gi) {
    // A wave whose first unit lies beyond the range still copies its share of the bias image before it leaves
    const int unit_ = (wgi * bc.kv_gpw + gi) * bc.kv_wpw + wave;
    const bool valid = unit_ < bc.units_kv_bh;
    if (!valid && gi > 0) break;
    const int unit = valid ? unit_ : 0;
    const bool glo = unit >= nown;                 // global-key owner unit
    const int split = unit - nown;
    const int ch = glo ? 0 : fdiv(unit, bc.m_kv_NWP), wp = glo ? 0 : unit - ch * bc.kv_NWP;
    const int km = fdiv(ch, c.m_my), kn = ch - km * g.my;
    const int Gu = (!glo || split == 0) ? Gq : 0;  // global queries at the end of THIS unit's stream

    // ---- this lane's key slots: column j of key-tile kt is key (x, y = KT*hq + KT-1 - kt)
    const int jj = wp * 16 + lj;
    const int kx = fdiv(jj, bc.m_kv_HQ), khq = jj - kx * bc.kv_HQ;
    // kv_gspare (G == 1): key tile 0 of the chunk's first unused (x, hq) pair is the GLOBAL key.  Its scores use the g2l
    // region of the table for the own chunk's queries and the guard region (-1e30 -> P = dS = 0) for the other streamed
    // chunks (s_aqg), so every query meets it once over the pass; its dK / dV partial goes to glo_parts[chunk].  The
    // pair's other key tiles are junk columns (finite, never stored): key columns are independent in this pass.
    const bool gcol = bc.kv_gspare && !glo && jj == bc.kv_gjj;
    const unsigned akl = (unsigned)((glo || gcol) ? -(c.glo0 + (gcol ? 0 : min(lj, max(p.G - 1, 0))) * c.gsz) * 4
                                                  : (min(kx, W - 1) * c.P + KT * khq + KT - 1) * 4) - tab_lds;
    const int* aqsel = gcol ? s_aqg : s_aq;
    int ktok[KT];
    bool kreal[KT];
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
        if (gcol && kt == 0) ktok[kt] = 0;            // (kreal stays false: the local epilogue skips it)
      }
    }
    X8 kfb[MK][KT], vfb[MK][KT];
    auto load_own = [&]() {
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int ks = 0; ks < MK; ++ks) {
          const int d0 = ks * 32 + lg * 8;
          X8 z = {};
          kfb[ks][kt] = d0 < M ? *(const X8*)(kb + (int64_t)ktok[kt] * p.k_st + d0) : z;
          vfb[ks][kt] = d0 < M ? *(const X8*)(vb + (int64_t)ktok[kt] * p.v_st + d0) : z;
        }
    };
    // ---- unit prologue in TWO memory round trips (round 5).  It used to be a chain of seven: bias image -> barrier ->
    // global-query rows -> chunk count -> slot entries -> lse / delta gathers (twice at 448 slots) -> own K / V -> first
    // Q / dO rows, ~12-15 % of a unit's life (tools/kv_timing.py, round 2).  Now everything that depends on nothing is
    // requested first -- slot entries (EPRE rounds of 64), own K / V fragments, the chunk count, the global queries'
    // scalars, and LAST the bias image, whose copy loop waits for its own loads and therefore (loads return in order) for
    // all of the above -- then everything that depends on the slot entries: lse / delta gathers and the first Q / dO rows.
    // Same-box A/B at ViL-Small stage 1: 372 -> 358 us.
    // EPRE rounds of 64 slots in one straight-line pass (no branch, so that the compiler's counter waits stay where the
    // data is first used): 7 (448 slots: W <= 7 with the global rows) or 10 (W = 8: 577 slots); longer tables (W = 12)
    // finish in the three-phase loop below
    const int tix = glo ? bc.nch + split : ch;
    const int2* slots = bc.kv_slots + (int64_t)tix * bc.nqs;
    // (a laundered copy of the lane id for the unit's one-off addresses: derived from `lane` they are invariants of the
    //  unit loop, and the compiler hoists all of them out of it and then spills them around the step loop)
    int ln = lane;
    asm volatile("" : "+v"(ln));
    int2 e[EPRE];
#pragma unroll
    for (int u = 0; u < EPRE; ++u) e[u] = slots[min(u * 64 + ln, bc.nqs - 1)];
    load_own();
    const int nchunks = __builtin_amdgcn_readfirstlane(bc.kv_nchunks[tix]);
    if (gi == 0) {
      // the global queries' constant regions behind the image: [G x gsz: g2l[0][h][g] / scale | G*G x gsz: g2g[h][gq][gk] / scale]
      // (scalar loads; placed before the image copy so that they travel with everything else)
      if (bc.kv_xsize) {
        const float inv = 1.0f / p.scale;
        for (int r = 0; r < p.G + p.G * p.G; ++r) {
          float v = 0.f;
          if (r < p.G) { if (p.g2l0) v = p.g2l0[h * p.G + r]; }
          else if (p.g2g) v = p.g2g[(int64_t)h * p.G * p.G + (r - p.G)];
          for (int i = tid; i < c.gsz; i += blockDim.x) tab[c.tabsize + r * c.gsz + i] = v * inv;
        }
      }
      // the bias image, four 16-byte pieces per thread in flight (the plain copy loop waited for every piece in turn)
      const f32x4* src = (const f32x4*)(c.tabws + (int64_t)h * c.tabsize);
      const int n4 = c.tabsize >> 2;
      const int nthr = blockDim.x;
      for (int i0 = tid; i0 < n4; i0 += 4 * nthr) {
        f32x4 t4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) t4[u] = src[min(i0 + u * nthr, n4 - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (i0 + u * nthr < n4) ((f32x4*)tab)[i0 + u * nthr] = t4[u];
      }
    }
    // A wave with no unit (the last workgroup of a launch) leaves HERE, before the barrier that publishes the bias image.
    // That is defined behaviour on the only target this file is built for: the gfx9 / CDNA ISA specifies that S_BARRIER
    // "waits on only the surviving waves" when waves of the workgroup have already terminated.  Meeting the barrier on the
    // way out instead (round 6, the advisor's item) was measured and taken back: the second barrier site changes hipcc's
    // register allocation of the whole kernel -- head_dim 32: 250 -> 256 VGPRs + 76 bytes of scratch per lane, head_dim
    // 64: 245 -> 256 + 168 -- and the pass ran 349 -> 372 us at ViL-Small stage 1 (profiles/r06_attn_ab_vs_round5.txt,
    // first table).  tests/test_gpu_1_parity.py's odd chunk counts (e.g. 3 x 3 chunks, four-wave workgroups) run this path.
    if (!valid) break;
    // LDS copy of the slot table -- row index into the Q / dO descriptors, address terms (the global key's column reads
    // s_aqg: its g2l region for the own chunk's queries, the g2g region for a global query in chunk 0's unit, the guard
    // region for everything else) -- and, in the same sweep, the request for the slot's {lse, delta}: the local token's,
    // or, for a negative token (-1: padding, -(2 + g): global query g), entry G / g of this (image, head)'s xstat row.
    // Nothing is conditional on the slot kind after this point, so no lane masks are carried to the stores below.
    const float* xs_bh = p.xstat + (int64_t)bh * (p.G + 1) * 2;
    auto put_slot = [&](int sl, int2 ev, float& l_, float& d_) {
      const int aq_ = ev.y & ~3;
      const bool neg = ev.x < 0, gq_slot = ev.x <= -2;
      const int xi = gq_slot ? -2 - ev.x : p.G;
      const float* lp = neg ? xs_bh + 2 * xi : lse_bh + ev.x;
      const float* dp = neg ? xs_bh + 2 * xi + 1 : dlt_bh + ev.x;
      l_ = *lp; d_ = *dp;
      s_tok[sl] = gq_slot ? xi : max(ev.x, 0) + Gq;
      s_aq[sl] = aq_;
      if (bc.kv_gspare)
        s_aqg[sl] = gq_slot ? ((ch == 0 ? c.tabsize + p.G * c.gsz : c.guard0) - c.glo0) * 4
                            : ((ev.y & 1) ? aq_ - c.aconst * 4 : (c.guard0 - c.glo0) * 4);
    };
    float l8[EPRE], d8[EPRE];
#pragma unroll
    for (int u = 0; u < EPRE; ++u) put_slot(u * 64 + ln, e[u], l8[u], d8[u]);
    // slots beyond the first EPRE rounds (W = 12; global-key owner units): three phases per four rounds
    const int nq_u = glo ? bc.nqs : bc.nqs_own;
    for (int s0 = EPRE * 64; s0 < nq_u; s0 += 256) {
      int2 e4[4];
      float l4[4], d4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) e4[u] = slots[min(s0 + u * 64 + ln, bc.nqs - 1)];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (s0 + u * 64 < nq_u) put_slot(s0 + u * 64 + ln, e4[u], l4[u], d4[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (s0 + u * 64 < nq_u) { s_lse[s0 + u * 64 + ln] = l4[u] * LOG2E; s_dlt[s0 + u * 64 + ln] = d4[u]; }
    }
    if (gi == 0) __syncthreads(); else wave_lds_fence();      // bias image (whole workgroup) and this wave's s_tok visible
    const int nsteps = (nchunks * W2 + Gu + 31) >> 5;
    f32x4 dk[MD][KT], dv[MD][KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int dt = 0; dt < MD; ++dt) {
        dk[dt][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        dv[dt][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    wave_lds_fence();

    // global -> register prefetch ring of the streamed Q / dO rows, PF steps deep
    u32x4 qr_[PF][MD], dr_[PF][MD];
    auto load_step = [&](auto slot_, int st) {
      constexpr int sl = decltype(slot_)::value;
#pragma unroll
      for (int it = 0; it < MD; ++it) {
        const int row = (it * 64 + lane) / VCH;
        const int tok = s_tok[st * 32 + row];          // Q and dO may have different row strides (fused qkv)
        qr_[sl][it] = __builtin_amdgcn_raw_buffer_load_b128(qrs, __mul24(tok, qstride_b) + ld_off[it], 0, 0);
        dr_[sl][it] = __builtin_amdgcn_raw_buffer_load_b128(drs, __mul24(tok, dostride_b) + ld_off[it], 0, 0);
      }
    };
    constexpr int TILE = 32 * M * 2;
    // stage_s(st): the step's Q / dO rows leave the ring for the LDS tiles of parity st & 1 (read as rows here and
    // transposed in finish), the ring slot is refilled; S = Q K^T + bias and dP = dO V^T are issued
    auto stage_s = [&](auto slot_, int st, f32x4 (&sacc)[2][KT], f32x4 (&dpacc)[2][KT]) {
      constexpr int sl = decltype(slot_)::value;
      char* sq = s_q + (PIPE ? (st & 1) * 2 * TILE : 0);
      char* sd = sq + TILE;
#pragma unroll
      for (int it = 0; it < MD; ++it) {
        *(u32x4*)(sq + st_off[it]) = qr_[sl][it];
        *(u32x4*)(sd + st_off[it]) = dr_[sl][it];
      }
      if (st + PF < nsteps) load_step(slot_, st + PF);
      wave_lds_fence();
      // The bias gathers of the WHOLE step are issued before its first MFMA (32 dword reads in flight, counted waits).
      // Written per tile (gather 4x4, MFMA, next tile) the register allocator reused one C-operand quad for the tiles of
      // the first half and that half walked 4 dependent LDS round trips.  Same-box A/B at ViL-Small stage 1: 375 -> 365 us
      // (the same change in the forward / dQ kernels, which run 3 waves per SIMD: +4 % / +2.5 %, not applied there).
      f32x4 nd4[2];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const i32x4 aq4 = *(const i32x4*)(aqsel + st * 32 + hf * 16 + lg * 4);
        nd4[hf] = -*(const f32x4*)(s_dlt + st * 32 + hf * 16 + lg * 4);   // dP - delta rides in the accumulator
        lds_cvf tb[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) tb[r] = lds_f32((unsigned)aq4[r] - akl);
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          sacc[hf][kt] = (f32x4){tb[0][kt], tb[1][kt], tb[2][kt], tb[3][kt]};
        }
      }
      X8 qa[2][MK], da[2][MK];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int ks = 0; ks < MK; ++ks) {
          X8 z = {};
          qa[hf][ks] = (ks * 32 + lg * 8) < M ? *(const X8*)(sq + row_off[hf][ks]) : z;
          da[hf][ks] = (ks * 32 + lg * 8) < M ? *(const X8*)(sd + row_off[hf][ks]) : z;
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          f32x4 acc = sacc[hf][kt];
          f32x4 dp = nd4[hf];
#pragma unroll
          for (int ks = 0; ks < MK; ++ks) {
            acc = mfma16(qa[hf][ks], kfb[ks][kt], acc);
            dp = mfma16(da[hf][ks], vfb[ks][kt], dp);
          }
          sacc[hf][kt] = acc; dpacc[hf][kt] = dp;
        }
    };
    // P = exp2(S c - lse), dS = P o (dP - delta);  dV^T += dO^T P ; dK^T += Q^T dS
    auto finish = [&](int st, const f32x4 (&sacc)[2][KT], const f32x4 (&dpacc)[2][KT]) {
      const char* sq = s_q + (PIPE ? (st & 1) * 2 * TILE : 0);
      const char* sd = sq + TILE;
      u32x4 pbw[KT], dsw[KT];       // the 8 packed 16-bit operand values of each key tile, as dwords (2 per query half)
      f32x4 gds[4] = {};            // dS of query rows 28 .. 31 of the step, [row - 28][key tile]
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int sb = st * 32 + hf * 16 + lg * 4;
        const f32x4 ls4 = *(const f32x4*)(s_lse + sb);
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          float pr[4];
#pragma unroll
          for (int r = 0; r < 4; ++r)
            pr[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[hf][kt][r], c1, -ls4[r]));
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            const f32x2 p2 = {pr[2 * h2], pr[2 * h2 + 1]};
            const f32x2 d2 = {dpacc[hf][kt][2 * h2], dpacc[hf][kt][2 * h2 + 1]};
            const f32x2 ds2 = p2 * d2;
            pbw[kt][hf * 2 + h2] = pack2<T>(p2);
            dsw[kt][hf * 2 + h2] = pack2<T>(ds2);
            if (hf == 1) { gds[2 * h2][kt] = ds2[0]; gds[2 * h2 + 1][kt] = ds2[1]; }
          }
        }
      }
      // dS (fp32) of the step's last query rows, per key column, to the wave's LDS: in the LAST step rows 31 - g are the
      // global queries (kv_slots_block), whose dq = sum_k dS K the tail forms from them.  One 16-byte store per step for
      // G == 1 (every step: no branch on the step index, no registers carried through the loop).
      *(f32x4*)(s_gd + lane * 4) = gds[3];
      if (Gq > 1) {
#pragma unroll
        for (int r = 0; r < 3; ++r) *(f32x4*)(s_gd + ((3 - r) * 64 + lane) * 4) = gds[r];
      }
      X8 pb[KT], dsb[KT];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) { pb[kt] = __builtin_bit_cast(X8, pbw[kt]); dsb[kt] = __builtin_bit_cast(X8, dsw[kt]); }
#pragma unroll
      for (int dt = 0; dt < MD; ++dt) {
        X8 qt_, dt_;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const X4 tq = __builtin_bit_cast(X4, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (s16x4 __attribute__((address_space(3)))*)(sq + tr_off[hf][dt])));
          const X4 td = __builtin_bit_cast(X4, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (s16x4 __attribute__((address_space(3)))*)(sd + tr_off[hf][dt])));
#pragma unroll
          for (int e = 0; e < 4; ++e) { qt_[hf * 4 + e] = tq[e]; dt_[hf * 4 + e] = td[e]; }
        }
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          dv[dt][kt] = mfma16(dt_, pb[kt], dv[dt][kt]);
          dk[dt][kt] = mfma16(qt_, dsb[kt], dk[dt][kt]);
        }
      }
    };

    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, PF - 1> S1;
    if (nsteps > 0) load_step(S0{}, 0);
    if constexpr (PF == 2) { if (nsteps > 1) load_step(S1{}, 1); }
    // (the lse / delta gathers were requested before the first Q / dO rows: they land first)
#pragma unroll
    for (int u = 0; u < EPRE; ++u) { s_lse[u * 64 + ln] = l8[u] * LOG2E; s_dlt[u * 64 + ln] = d8[u]; }
    wave_lds_fence();
    if constexpr (PIPE) {
      // software pipeline over steps (see k_mfma_fwd): score / dP MFMAs of step st+1 issued before the VALU of step st
      f32x4 sA[2][KT], dA[2][KT], sB[2][KT], dB[2][KT];
      if (nsteps > 0) stage_s(S0{}, 0, sA, dA);
      for (int st = 0; st < nsteps; st += 2) {
        if (st + 1 < nsteps) stage_s(S1{}, st + 1, sB, dB);
        finish(st, sA, dA);
        if (st + 1 < nsteps) {
          if (st + 2 < nsteps) stage_s(S0{}, st + 2, sA, dA);
          finish(st + 1, sB, dB);
        }
      }
    } else {
      auto one = [&](auto slot_, int st) {
        f32x4 sA[2][KT], dA[2][KT];
        stage_s(slot_, st, sA, dA);
        finish(st, sA, dA);
        wave_lds_fence();
      };
      for (int st = 0; st < nsteps; st += PF) {
        one(S0{}, st);
        if constexpr (PF == 2) { if (st + 1 < nsteps) one(S1{}, st + 1); }
      }
    }

    // ---- global-token QUERY rows (vil_attn_bwd_full).  Round 5: they are the last G slots of the unit's query stream, so
    // their P / dS went through the step loop's MFMAs like every other query's and dK / dV already hold their share (the
    // round-4 tail staged their q / dO / out rows, formed the scores by VALU dot products and ran 16 nearly empty MFMAs:
    // 8-11 % of the pass by timing ablation, profiles/r05_attn_ab.txt).  What is left: the unit's share of
    // dq_g = sum_k dS[g, k] K[k] and of d(g2l[0]) = sum_k dS[g, k] over its real local keys, from the dS row the last step
    // left in LDS, to a partial record; d(g2g) from the global key's column.
    if (!VIL_KV_ABL_NOTAIL && Gu) {
      float* rec = bc.gq_parts + ((int64_t)bh * bc.gq_nrec + (glo ? nown : unit)) * p.G * (M + 4);
      const bool gpair = gcol && ch == 0;       // kv_gspare: the (global query, global key) pair is chunk 0's unit's
      for (int gq = 0; gq < p.G; ++gq) {
        const f32x4 d4 = *(const f32x4*)(s_gd + (gq * 64 + 48 + lj) * 4);          // row 31 - gq: written by lane group 3
        float ds[KT], bsum = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          const bool gk_ = kt == 0 && gpair;
          ds[kt] = (kreal[kt] || gk_) ? d4[kt] : 0.f;      // (junk key columns hold finite garbage: never summed)
          if (!glo && !gk_) bsum += ds[kt];
        }
        if (glo && kreal[0] && lg == 0 && p.dg2g) atomicAdd(&p.dg2g[((int64_t)h * p.G + gq) * p.G + lj], ds[0]);
        if (gpair && lg == 0 && p.dg2g) atomicAdd(&p.dg2g[((int64_t)h * p.G + gq) * p.G], ds[0]);
        float dqp[MK][8];
#pragma unroll
        for (int ks = 0; ks < MK; ++ks)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float t = 0.f;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) t = __builtin_fmaf(ds[kt], (float)kfb[ks][kt][e], t);
            dqp[ks][e] = t;
          }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
#pragma unroll
          for (int ks = 0; ks < MK; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) dqp[ks][e] += __shfl_xor(dqp[ks][e], o, 64);
          bsum += __shfl_xor(bsum, o, 64);
        }
        if (lj == 0) {
#pragma unroll
          for (int ks = 0; ks < MK; ++ks) {
            const int d0 = ks * 32 + lg * 8;
            if (d0 < M) {
              *(f32x4*)(rec + gq * (M + 4) + d0) = (f32x4){dqp[ks][0], dqp[ks][1], dqp[ks][2], dqp[ks][3]};
              *(f32x4*)(rec + gq * (M + 4) + d0 + 4) = (f32x4){dqp[ks][4], dqp[ks][5], dqp[ks][6], dqp[ks][7]};
            }
          }
          if (lg == 0) rec[gq * (M + 4) + M] = bsum;
        }
      }
    }

    // ---- epilogue
    // (the store addresses are derived from ktok HERE: left to itself the compiler forms the eight 64-bit row addresses
    //  before the step loop and carries -- or, one register over the budget, spills -- them through it)
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) asm volatile("" : "+v"(ktok[kt]));
    int lge = lg;
    asm volatile("" : "+v"(lge));
    if (!glo) {
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
        if (kreal[kt]) {
#pragma unroll
          for (int dt = 0; dt < MD; ++dt) {
            X4 wk, wv;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              wk[r] = (T)(dk[dt][kt][r] * p.scale);
              wv[r] = (T)dv[dt][kt][r];
            }
            *(X4*)(dkb + (int64_t)ktok[kt] * p.dk_st + dt * 16 + lge * 4) = wk;
            *(X4*)(dvb + (int64_t)ktok[kt] * p.dv_st + dt * 16 + lge * 4) = wv;
          }
        }
      if (gcol) {              // the global key's dK / dV partial of this chunk (reduced by reduce_glo_block)
        float* out = bc.glo_parts + (((int64_t)bh * bc.glo_nrec + ch) * p.G) * 2 * M;
#pragma unroll
        for (int dt = 0; dt < MD; ++dt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            out[dt * 16 + lg * 4 + r] = dk[dt][0][r] * p.scale;
            out[M + dt * 16 + lg * 4 + r] = dv[dt][0][r];
          }
      }
    } else if (kreal[0]) {
      const int rec_ = bc.glo_from_dq ? bc.glo_nrec - 1 : split;       // (the dQ pass wrote records 0 .. glo_nrec-2)
      float* out = bc.glo_parts + ((((int64_t)bh * bc.glo_nrec + rec_) * p.G + lj) * 2) * M;
#pragma unroll
      for (int dt = 0; dt < MD; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          out[dt * 16 + lg * 4 + r] = dk[dt][0][r] * p.scale;
          out[M + dt * 16 + lg * 4 + r] = dv[dt][0][r];
        }
    }
    wave_lds_fence();
  }
}

// One workgroup per (image, head, global token gk):
//   dk/dv rows of global KEY gk   = sum over the splits' fp32 partials (glo_parts)
//   dq row of global QUERY gk     = scale * sum over the units' partials (gq_parts), and its d(g2l[0]) share
// (the first 256 threads of a k_mfma_post_bwd workgroup)
template <typename T>
__device__ __forceinline__ void reduce_glo_block(const VilParams& p, const BwdCfg& bc, int nslots, int blk) {
  __shared__ float red[4][2][64];
  const int M = p.M, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int gk = blk % p.G, bh = blk / p.G;
  const int b = bh / p.H, h = bh % p.H;
  {
    // dk / dv rows of global key gk: sum of glo_nrec records of 2M floats (up to hundreds in by-product mode): the four
    // waves take every fourth record, four loads in flight each, lanes = columns (2M <= 128: one or two per lane)
    const float* rb = bc.glo_parts + (((int64_t)bh * bc.glo_nrec) * p.G + gk) * 2 * M;
    const int64_t rstride = (int64_t)p.G * 2 * M;
    const bool c2 = lane + 64 < 2 * M;
    float b0 = 0.f, b1 = 0.f;
    int s = wv;
    for (; s + 12 < bc.glo_nrec; s += 16) {
      float v0[4], v1[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* r = rb + (s + 4 * u) * rstride;
        v0[u] = lane < 2 * M ? r[lane] : 0.f;
        v1[u] = c2 ? r[lane + 64] : 0.f;
      }
      b0 += (v0[0] + v0[1]) + (v0[2] + v0[3]);
      b1 += (v1[0] + v1[1]) + (v1[2] + v1[3]);
    }
    for (; s < bc.glo_nrec; s += 4) {
      const float* r = rb + s * rstride;
      b0 += lane < 2 * M ? r[lane] : 0.f;
      b1 += c2 ? r[lane + 64] : 0.f;
    }
    red[wv][0][lane] = b0; red[wv][1][lane] = b1;
    __syncthreads();
    if (wv == 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = lane + 64 * j;
        if (col < 2 * M) {
          const float sk = red[0][j][lane] + red[1][j][lane] + red[2][j][lane] + red[3][j][lane];
          T* dst = col < M ? (T*)p.dk + b * p.dk_sb + (int64_t)gk * p.dk_st + h * p.dk_sh + col
                           : (T*)p.dv + b * p.dv_sb + (int64_t)gk * p.dv_st + h * p.dv_sh + (col - M);
          *dst = (T)sk;
        }
      }
    }
    __syncthreads();
  }
  if (!p.glo_rows) return;
  const int RS = M + 4;
  const float* base = bc.gq_parts + ((int64_t)bh * nslots * p.G + gk) * RS;
  const int64_t sstride = (int64_t)p.G * RS;
  const bool c1ok = lane + 64 <= M;                 // second column of this lane (only M = 64: the dS sum)
  float a0 = 0.f, a1 = 0.f;
  int s = wv;
  for (; s + 12 < nslots; s += 16) {
    float v0[4], v1[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float* r = base + (s + 4 * u) * sstride;
      v0[u] = lane <= M ? r[lane] : 0.f;
      v1[u] = c1ok ? r[lane + 64] : 0.f;
    }
    a0 += (v0[0] + v0[1]) + (v0[2] + v0[3]);
    a1 += (v1[0] + v1[1]) + (v1[2] + v1[3]);
  }
  for (; s < nslots; s += 4) {
    const float* r = base + s * sstride;
    a0 += lane <= M ? r[lane] : 0.f;
    a1 += c1ok ? r[lane + 64] : 0.f;
  }
  red[wv][0][lane] = a0; red[wv][1][lane] = a1;
  __syncthreads();
  if (wv == 0) {
    a0 = red[0][0][lane] + red[1][0][lane] + red[2][0][lane] + red[3][0][lane];
    a1 = red[0][1][lane] + red[1][1][lane] + red[2][1][lane] + red[3][1][lane];
    T* dq = (T*)p.dq_g + b * p.dq_sb + (int64_t)gk * p.dq_st + h * p.dq_sh;
    if (lane < M) dq[lane] = (T)(a0 * p.scale);
    const float bs = M == 64 ? a1 : a0;              // column M
    if (p.dg2l0 && lane == (M & 63)) atomicAdd(&p.dg2l0[h * p.G + gk], bs);
  }
}

// Backward epilogue, ONE launch with two roles (separate launches cost ~11 us each for a few microseconds of work):
// [0, nglo) the global-token reductions (first 256 threads; the other waves of the workgroup end at once, and a
// workgroup barrier only waits for waves that are still alive), then (64 bins, head) blocks of the histogram reduce.
template <typename T>
__global__ __launch_bounds__(1024) void k_mfma_post_bwd(VilParams p, MfmaCfg c, BwdCfg bc, int nslots, int nglo, int nhx) {
  const int blk = blockIdx.x;
  if (blk < nglo) {
    if (threadIdx.x < 256) reduce_glo_block<T>(p, bc, nslots, blk);
    return;
  }
  const int hb = blk - nglo;
  const int h = hb / nhx;
  reduce_hist_block(p, c, bc, hb - h * nhx, h);
}

// rowsum(dO * O): LPR lanes per (image, head, token) row, each lane one (or three) 16-byte pieces of the
// row, so a wave reads whole 64..128-byte row segments (one thread per row left 7/8 of every line fetched
// by a load instruction to the other lanes' later loads)
// also: max_q |dO_q|^2 and max_k |v_k|^2 (float bits, atomicMax) for the histogram's fixed-point scale
template <typename T, int MD>
__global__ void k_mfma_delta(VilParams p, unsigned* norm2) {
  typedef typename V16<T>::x8 X8;
  constexpr int M = 16 * MD;
  constexpr int LPR = MD == 4 ? 8 : (MD == 2 ? 4 : 2);
  constexpr int NL = M / (8 * LPR);
  const int Nloc = p.g.nx * p.g.ny;
  // grid (ceil(Nloc*H*LPR/256), B): consecutive lanes walk (token, head, piece) with the piece fastest and the head next,
  // i.e. the H*M contiguous values of a token row: a wave reads whole rows.  (Round 3 gave every head its own blocks:
  // each 128-byte line of out / dout / v -- two heads at head_dim 32 -- was fetched once per head, by blocks that ran at
  // different times: 2.94x the algorithmic bytes at ViL-Small stage 1, profiles/r03_pmc_traffic.json.)
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int sub = t % LPR, th = t / LPR;
  const int tok = th / p.H, h = th - tok * p.H;
  const int b = blockIdx.y, bh = b * p.H + h;
  const bool live = tok < Nloc;
  const int64_t i = (int64_t)bh * Nloc + tok;
  float n_do = 0.f, n_v = 0.f, s = 0.f, n_vg = 0.f;
  if (live) {
    const T* op = (const T*)p.out + b * p.o_sb + (int64_t)tok * p.o_st + h * p.o_sh;
    const T* dp = (const T*)p.dout + b * p.do_sb + (int64_t)tok * p.do_st + h * p.do_sh;
    const T* vp = (const T*)p.v + b * p.v_sb + (int64_t)(p.G + tok) * p.v_st + h * p.v_sh;
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      const int d0 = (sub * NL + l) * 8;
      const X8 a = *(const X8*)(op + d0), c = *(const X8*)(dp + d0), v = *(const X8*)(vp + d0);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        s = __builtin_fmaf((float)a[e], (float)c[e], s);
        n_do = __builtin_fmaf((float)c[e], (float)c[e], n_do);
        n_v = __builtin_fmaf((float)v[e], (float)v[e], n_v);
      }
    }
    if (tok < p.G && sub == 0) {        // the G global rows of v
      const T* vg = (const T*)p.v + b * p.v_sb + (int64_t)tok * p.v_st + h * p.v_sh;
      for (int d = 0; d < M; ++d) n_vg = __builtin_fmaf((float)vg[d], (float)vg[d], n_vg);
      if (p.glo_rows) {                 // ... and {lse, rowsum(dO * O)} of global QUERY row `tok` (vil_attn_bwd_full)
        const T* og = (const T*)p.o_g + b * p.o_sb + (int64_t)tok * p.o_st + h * p.o_sh;
        const T* dg = (const T*)p.do_g + b * p.do_sb + (int64_t)tok * p.do_st + h * p.do_sh;
        float sg = 0.f;
        for (int d = 0; d < M; ++d) sg = __builtin_fmaf((float)og[d], (float)dg[d], sg);
        float* xs = p.xstat + ((int64_t)bh * (p.G + 1) + tok) * 2;
        xs[0] = p.lse_g[(int64_t)bh * p.G + tok]; xs[1] = sg;
      }
    }
    if (tok == 0 && sub == 0) {         // what a padding slot of the dK/dV pass reads: lse = +big (P = 0), delta = 0
      float* xs = p.xstat + ((int64_t)bh * (p.G + 1) + p.G) * 2;
      xs[0] = LSE_PAD / LOG2E; xs[1] = 0.f;
    }
  }
#pragma unroll
  for (int o = 1; o < LPR; o <<= 1) {
    s += __shfl_xor(s, o, 64);
    n_do += __shfl_xor(n_do, o, 64);
    n_v += __shfl_xor(n_v, o, 64);
  }
  n_v = fmaxf(n_v, n_vg);
  if (live && sub == 0) p.delta[i] = s;
  if (norm2) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      n_do = fmaxf(n_do, __shfl_xor(n_do, o, 64));
      n_v = fmaxf(n_v, __shfl_xor(n_v, o, 64));
    }
    // block maximum through LDS, then ONE atomic pair per block into one of VIL_NORM_SLOTS slots 128 bytes
    // apart (same-address atomics serialise at ~10 ns each: one slot for every wave cost 0.17 ms)
    __shared__ float s_n[2][4];
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_n[0][wv] = n_do; s_n[1][wv] = n_v; }
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned a = __float_as_uint(fmaxf(fmaxf(s_n[0][0], s_n[0][1]), fmaxf(s_n[0][2], s_n[0][3])));
      const unsigned b2 = __float_as_uint(fmaxf(fmaxf(s_n[1][0], s_n[1][1]), fmaxf(s_n[1][2], s_n[1][3])));
      unsigned* slot = norm2 + 32 * ((blockIdx.x + blockIdx.y * 7) % VIL_NORM_SLOTS);
      if (a > __hip_atomic_load(&slot[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&slot[0], a);
      if (b2 > __hip_atomic_load(&slot[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&slot[1], b2);
    }
  }
}

// ===================================================================== host side
// (the fp32 family, vil_attn_mfma_f32.hip, builds its tables with the same prologue kernel)
int vil_mfma_launch_prep_bwd(const VilParams& p, const MfmaCfg& c, const BwdCfg& bc, int row_stride_b, const PrepZero& zr, hipStream_t s) {
  const int ntx = (c.tabsize + 255) / 256, nzb = (zr.total + 255) / 256;
  const size_t lds = (size_t)(c.NSP > bc.nqs ? c.NSP : bc.nqs) * 8;
  if (int he = vil_ensure_dyn_lds((const void*)k_mfma_prep_bwd, lds)) return he;
  k_mfma_prep_bwd<<<dim3((unsigned)(ntx * p.H + bc.nch + bc.nch + bc.nsplit + nzb)), dim3(256), lds, s>>>(p, c, bc, row_stride_b, ntx, zr);
  return (int)hipGetLastError();
}

static void bwd_cfg(const VilAttnDesc* d, const MfmaCfg& c, BwdCfg& bc) {
  memset(&bc, 0, sizeof(bc));
  VilGeom g; vil_geom_init(g, d->nx, d->ny, d->W, d->exact, d->mode);
  bc.nch = g.mx * g.my;
  // By-product mode (dK/dV of the global keys out of the dQ pass): a measured alternative, OFF by default.  In-step, one
  // box: ViL-Small stage 1 dK/dV 399 -> 366 us but dQ 350 -> 374 and the record reduce 13 -> 16: whole backward 861 -> 855 us
  // (-0.7 %); ViL-Medium-Deep stage 1 dK/dV 314 -> 295, dQ 274 -> 302: 682 -> 696 us (+2 %); 4x4-chunk and one-chunk stages
  // lose outright (the dQ pass pays ~1.4 us per unit, twice that at head_dim 64).  It moves work between the passes
  // without reducing it: the per-unit VALU reduction over the 16 query columns costs what the owner units' idle MFMA
  // columns cost.  -DVIL_GLO_FROM_DQ=1 enables it for head_dim <= 32 and >= 32 chunks (tests cover both settings).
#ifndef VIL_GLO_FROM_DQ
#define VIL_GLO_FROM_DQ 0
#endif
  bc.glo_from_dq = VIL_GLO_FROM_DQ && d->G > 0 && d->G <= 4 && d->M <= 32 && bc.nch >= 32;
  bc.kv_KT = d->M >= 48 ? 2 : (d->M == 32 ? VIL_KV_KT32 : 4);
  bc.kv_HQ = (g.W + bc.kv_KT - 1) / bc.kv_KT;
  bc.kv_NWP = (g.W * bc.kv_HQ + 15) / 16;
  // Round 5: with ONE global token (every published model) and an unused (x, hq) pair in the chunk's last wave (W = 7:
  // 14 of 16 pairs used), the global key rides in that pair's key tile 0 of every chunk's unit, live only against the
  // chunk's own queries.  The owner units this replaces -- (nch + 8) / 9 per (image, head), each streaming a ninth of
  // all queries through four key tiles with ONE useful column -- were 11-12 % of the pass's units.
  // W = 8 / head_dim 32 (16 of 16 pairs used) and G > 1 keep the owner units.
#ifndef VIL_KV_GSPARE
#define VIL_KV_GSPARE 1
#endif
  bc.kv_gjj = g.W * bc.kv_HQ;
  const bool gspare_ok = VIL_KV_GSPARE && d->G == 1 && !d->only_glo && !bc.glo_from_dq && bc.kv_gjj < 16 * bc.kv_NWP;
  const int gq_rows = (d->G >= 1 && d->G <= 4) ? d->G : 0;      // vil_attn_bwd_full's global-query rows close the stream (G <= 4)
  bc.kv_xsize = gq_rows ? (gq_rows + gq_rows * gq_rows) * c.gsz : 0;
  bc.kv_span = (g.W - 1) * c.P + bc.kv_KT * bc.kv_HQ - 1;
  const int kv_tiles = (VIL_KV_PIPE && d->M == 32 && bc.kv_KT == 2) ? 4 : 2;       // Q + dO tiles (pipelined kernel: two pairs)
  // waves per workgroup.  Two-wave workgroups (a workgroup's slot is held until its longest unit ends) measured 354 -> 350 us
  // at 56x56 and 271 -> 262 us at 96x96 alone (head_dim 32; 160 -> 181 us at head_dim 64) but LOST inside the training step
  // (313 -> 322 us at 56x56: twice the workgroups, twice the bias-image copies next to the other kernels' traffic): four.
#ifndef VIL_KV_WPW
#define VIL_KV_WPW 4
#endif
  // everything that depends on whether the global key rides in a spare column; returns the waves resident on a CU
  auto layout = [&](bool gspare) {
    bc.kv_gspare = gspare;
    bc.nsplit = d->G > 0 ? (gspare ? 0 : (bc.glo_from_dq ? 1 : (bc.nch + 8) / 9)) : 0;
    bc.units_kv_bh = bc.nch * bc.kv_NWP + bc.nsplit;
    bc.gq_nrec = bc.nch * bc.kv_NWP + (gspare ? 0 : 1);
    // streamed query slots: an own-key unit sees <= nact query chunks, a global-key unit its share of all chunks
    const int qch = (d->G > 0 && !bc.glo_from_dq && bc.nsplit > 0) ? (bc.nch + bc.nsplit - 1) / bc.nsplit : 0;
    bc.nqs = ((qch > g.nact ? qch : g.nact) * g.W2 + gq_rows + 31) & ~31;
    // straight-line rounds of 64 slots in the dK/dV prologue (the kernel's template parameter): the whole table where it
    // is short (random-shift training: 2 W^2 slots), 7 rounds at W = 7, 10 at W = 8; W = 12 finishes in a loop
    // ... sized for the chunks' own units (nact * W^2 slots); a global-key owner unit, whose table is a ninth of ALL
    // chunks, finishes in the loop.  (Sized for the longest table, the 2-chunk units of random-shift training at W = 8
    // filled ten rounds for 129 slots: dK/dV 90 -> 117 us at 48x48.)
    bc.nqs_own = (g.nact * g.W2 + gq_rows + 31) & ~31;
    bc.kv_epre = bc.nqs_own <= 3 * 64 ? 3 : (bc.nqs_own <= 7 * 64 ? 7 : 10);
    const int nqsa = ((bc.nqs + 63) & ~63) > bc.kv_epre * 64 ? ((bc.nqs + 63) & ~63) : bc.kv_epre * 64;
    // slot tables (token, address term, lse, delta) + Q / dO tiles + the dS rows of the global queries + (kv_gspare) the global key's address terms
    bc.kv_wave_lds = ((nqsa * 16 + kv_tiles * 32 * d->M * 2 + (d->G > 1 ? 4 : 1) * 1024 + (gspare ? nqsa * 4 : 0) + 15) / 16) * 16;
    bc.kv_wpw = VIL_KV_WPW;
    const size_t tabb = (size_t)(c.tabsize + bc.kv_xsize) * 4;
    while (bc.kv_wpw > 1 && tabb + (size_t)bc.kv_wpw * bc.kv_wave_lds > 160 * 1024) bc.kv_wpw >>= 1;
    const size_t lds = tabb + (size_t)bc.kv_wpw * bc.kv_wave_lds;
    const int wgs = lds > 160 * 1024 ? 0 : (int)((160 * 1024) / lds);
    return wgs * bc.kv_wpw < 8 ? wgs * bc.kv_wpw : 8;           // (two waves per SIMD: the register budget)
  };
  // The spare-column scheme costs nqs * 4 bytes of LDS per wave (the global key's address terms): it is taken only where
  // that does not cost resident waves (W = 12 at head_dim 64: 144 KB -> 164 KB per four-wave workgroup, i.e. two-wave
  // workgroups and half a wave per SIMD -- 340 -> 518 us at 48x48, same-box A/B)
  const int res0 = layout(false);
  if (gspare_ok && layout(true) < res0) layout(false);
  const int groups = (bc.units_kv_bh + bc.kv_wpw - 1) / bc.kv_wpw;
#ifndef VIL_KV_WGS
#define VIL_KV_WGS 8192
#endif
  int gpw = (int)(((int64_t)d->B * d->H * groups) / VIL_KV_WGS);
  if (gpw < 1) gpw = 1;
  if (gpw > groups) gpw = groups;
  bc.kv_gpw = gpw;
  bc.kv_wg_per_bh = (groups + gpw - 1) / gpw;
  bc.dq_wave_lds = ((c.NSP * 8 + ((VIL_DQ_PIPE && d->M == 32) ? 2 : 1) * 32 * d->M * 2 + 15) / 16) * 16;   // slot tables + K tile(s)
  bc.dq_QT = d->M >= 32 ? 2 : 4;
  bc.dq_HQ = (g.W + bc.dq_QT - 1) / bc.dq_QT;
  bc.dq_NWP = (g.W * bc.dq_HQ + 15) / 16;
  bc.dq_units_bh = bc.nch * bc.dq_NWP;
  bc.glo_nrec = bc.kv_gspare ? bc.nch : (bc.glo_from_dq ? bc.dq_units_bh + 1 : bc.nsplit);
  bc.dq_wpw = 4;
  while (bc.dq_wpw > 1 && (size_t)c.tabsize * 16 + (size_t)bc.dq_wpw * bc.dq_wave_lds > 160 * 1024) bc.dq_wpw >>= 1;
  {
    const int dgroups = (bc.dq_units_bh + bc.dq_wpw - 1) / bc.dq_wpw;
#ifndef VIL_DQ_WGS
#define VIL_DQ_WGS 4096
#endif
    int dgpw = (int)(((int64_t)d->B * d->H * dgroups) / VIL_DQ_WGS);
    if (dgpw < 1) dgpw = 1;
    if (dgpw > dgroups) dgpw = dgroups;
    bc.dq_gpw = dgpw;
    bc.dq_wg_per_bh = (dgroups + dgpw - 1) / dgpw;
  }
  // persistent launches (UnitQueue): the workspace is laid out for the largest grid a launch may take; vil_mfma_bwd
  // sets dq_nwg / kv_nwg to the resident number before it launches
  {
    const int step = 8 * d->H;
    const int64_t need_q = (((int64_t)d->B * d->H * bc.dq_units_bh + bc.dq_wpw - 1) / bc.dq_wpw + step - 1) / step * step;
    const int64_t need_k = (((int64_t)d->B * d->H * bc.units_kv_bh + bc.kv_wpw - 1) / bc.kv_wpw + step - 1) / step * step;
    const int64_t cap_wgs = VIL_MAX_PERSISTENT_WGS / step * step > 0 ? VIL_MAX_PERSISTENT_WGS / step * step : step;
    bc.dq_nwg = (int)(need_q < cap_wgs ? need_q : cap_wgs);
    bc.kv_nwg = (int)(need_k < cap_wgs ? need_k : cap_wgs);
  }
  bc.uq_dq.units_bh = bc.dq_units_bh; bc.uq_dq.m_units_bh = vil_magic((unsigned)bc.dq_units_bh);
  bc.uq_kv.units_bh = bc.units_kv_bh; bc.uq_kv.m_units_bh = vil_magic((unsigned)bc.units_kv_bh);
  // dQ histogram: drained every hist_flush list entries; a bin gets at most one contribution per real query of a unit
  bc.hist_flush = 12;
  bc.hist_nmax = (16 * bc.dq_QT < g.W2 ? 16 * bc.dq_QT : g.W2) * (bc.hist_flush + 2 * bc.dq_wpw);
  bc.m_dq_wgbh = vil_magic((unsigned)(bc.dq_wg_per_bh * d->H)); bc.m_dq_NWP = vil_magic((unsigned)bc.dq_NWP);
  bc.m_dq_HQ = vil_magic((unsigned)bc.dq_HQ);
  bc.m_kv_wgbh = vil_magic((unsigned)(bc.kv_wg_per_bh * d->H)); bc.m_kv_NWP = vil_magic((unsigned)bc.kv_NWP);
  bc.m_kv_HQ = vil_magic((unsigned)bc.kv_HQ);
}

static size_t dq_lds(const MfmaCfg& c, const BwdCfg& bc) { return (size_t)c.tabsize * 16 + (size_t)bc.dq_wpw * bc.dq_wave_lds; }   // table, 32-bit bins, 64-bit bins, waves
static size_t kv_lds(const MfmaCfg& c, const BwdCfg& bc) { return (size_t)(c.tabsize + bc.kv_xsize) * 4 + (size_t)bc.kv_wpw * bc.kv_wave_lds; }

int vil_mfma_bwd_supported(const VilAttnDesc* d) {
  if ((d->do_st | d->do_sb | d->do_sh) & 7) return VIL_E_ALIGN;
  // Q and dO rows are addressed with 32-bit byte offsets (24-bit token x stride) in the dK/dV pass
  for (int64_t st : {d->q_st, d->do_st})
    if (st >= (1 << 22) || st * 2 * (int64_t)d->nx * d->ny >= (1ll << 31)) return VIL_E_BACKEND;
  if ((d->dq_st | d->dq_sb | d->dq_sh | d->dk_st | d->dk_sb | d->dk_sh | d->dv_st | d->dv_sb | d->dv_sh) & 3) return VIL_E_ALIGN;
  MfmaCfg c; vil_mfma_make_cfg(d, c);
  BwdCfg bc; bwd_cfg(d, c, bc);
  if (dq_lds(c, bc) > 160 * 1024 || kv_lds(c, bc) > 160 * 1024) return VIL_E_BACKEND;
  if ((uint64_t)d->B * bc.dq_units_bh * (uint64_t)bc.dq_units_bh >= (1ull << 32)) return VIL_E_BACKEND;   // fdiv exactness
  if ((uint64_t)d->B * d->H * bc.kv_wg_per_bh * ((uint64_t)bc.kv_wg_per_bh * d->H) >= (1ull << 32)) return VIL_E_BACKEND;
  return VIL_OK;
}

// floats: [delta | table copies | hist partials | global-key partials | global-query partials | dK/dV slot tables | counts]
static void bwd_ws_layout(const VilAttnDesc* d, const MfmaCfg& c, const BwdCfg& bc, size_t off[9]) {
  const size_t rows = (size_t)d->B * d->H * d->nx * d->ny;
  off[0] = 0;
  off[1] = ((rows + 3) & ~(size_t)3) + (((size_t)d->B * d->H * (d->G + 1) * 2 + 3) & ~(size_t)3) + 32 * VIL_NORM_SLOTS;   // + xstat, norm-maxima slots and the histogram scale
  off[2] = off[1] + (size_t)d->H * c.tabsize;
  off[3] = off[2] + (size_t)bc.dq_nwg * c.tabsize * 2;                              // 64-bit histogram records
  off[4] = off[3] + (size_t)d->B * d->H * bc.glo_nrec * d->G * 2 * d->M;
  off[5] = (off[4] + (size_t)d->B * d->H * bc.gq_nrec * d->G * (d->M + 4) + 3) & ~(size_t)3;
  off[6] = off[5] + (size_t)(bc.nch + bc.nsplit) * bc.nqs * 2;                    // dK/dV slot tables (int2)
  off[7] = off[6] + (((size_t)(bc.nch + bc.nsplit) + 3) & ~(size_t)3);             // their chunk counts
  off[8] = off[7] + vil_key_slots_floats(c, bc.nch);                               // key-slot tables of the dQ pass
}

size_t vil_mfma_bwd_workspace(const VilAttnDesc* d) {
  MfmaCfg c; vil_mfma_make_cfg(d, c);
  BwdCfg bc; bwd_cfg(d, c, bc);
  size_t off[9]; bwd_ws_layout(d, c, bc, off);
  return off[8] * sizeof(float) + 64;
}

int vil_mfma_bwd(const VilAttnDesc* d, VilParams& p, hipStream_t s) {
  if (d->dtype == VIL_DTYPE_F32) return vil_f32_bwd(d, p, s);
  MfmaCfg c; vil_mfma_make_cfg(d, c);
  BwdCfg bc; bwd_cfg(d, c, bc);
  size_t off[9]; bwd_ws_layout(d, c, bc, off);
  float* ws = (float*)p.delta;
  if (((uintptr_t)p.q | (uintptr_t)p.k | (uintptr_t)p.v | (uintptr_t)p.dout | (uintptr_t)p.out | (uintptr_t)ws) & 15)
    return VIL_E_ALIGN;
  if (((uintptr_t)p.dq | (uintptr_t)p.dk | (uintptr_t)p.dv) & 7) return VIL_E_ALIGN;
  p.delta = ws + off[0];
  p.xstat = ws + (((size_t)d->B * d->H * d->nx * d->ny + 3) & ~(size_t)3);
  float* tabws = ws + off[1];
  c.tabws = tabws;
  bc.hist_parts = ws + off[2];
  bc.glo_parts = ws + off[3];
  bc.gq_parts = ws + off[4];
  bc.kv_slots = (int2*)(ws + off[5]);
  bc.kv_nchunks = (int*)(ws + off[6]);
  c.key_slots = (int2*)(ws + off[7]);
  c.key_nslots = (int*)(c.key_slots + (size_t)bc.nch * c.NSP);
  bc.do_hist = (p.dtable != nullptr) || (p.dg2l != nullptr);
  bc.norm2 = (unsigned*)(ws + off[1] - 32 * VIL_NORM_SLOTS);
  const VilWork w(d);
  int e;
#define BWD_SWITCH_T(T_, ...)                    \
  switch (d->M) {                                \
    case 16: { typedef T_ TT_; constexpr int MD_ = 1; __VA_ARGS__; } break; \
    case 32: { typedef T_ TT_; constexpr int MD_ = 2; __VA_ARGS__; } break; \
    case 48: { typedef T_ TT_; constexpr int MD_ = 3; __VA_ARGS__; } break; \
    case 64: { typedef T_ TT_; constexpr int MD_ = 4; __VA_ARGS__; } break; \
    default: return VIL_E_HEAD_DIM;              \
  }
#define BWD_SWITCH(...)                                                   \
  if (d->dtype == VIL_DTYPE_F16) { BWD_SWITCH_T(_Float16, __VA_ARGS__) }  \
  else { BWD_SWITCH_T(__bf16, __VA_ARGS__) }
  {
    PrepZero zr;
    zr.ptr[0] = bc.norm2; zr.n[0] = 32 * VIL_NORM_SLOTS;
    zr.ptr[1] = (unsigned*)p.dg2l; zr.n[1] = p.dg2l ? p.H * p.G : 0;
    // the LDS image covers only part of the caller's table: the rest of d(table) is 0
    zr.ptr[2] = (unsigned*)p.dtable; zr.n[2] = (p.dtable && c.trows < p.bias_S) ? p.bias_S * p.bias_S * p.H : 0;
    // whole-layer backward: the global QUERY rows' bias gradients are accumulated with atomics
    zr.ptr[3] = (unsigned*)p.dg2l0; zr.n[3] = (p.glo_rows && p.dg2l0) ? p.H * p.G : 0;
    zr.ptr[4] = (unsigned*)p.dg2g; zr.n[4] = (p.glo_rows && p.dg2g) ? p.H * p.G * p.G : 0;
    zr.total = zr.n[0] + zr.n[1] + zr.n[2] + zr.n[3] + zr.n[4];
    const int ntx = (c.tabsize + 255) / 256, nzb = (zr.total + 255) / 256;
    const size_t lds = (size_t)(c.NSP > bc.nqs ? c.NSP : bc.nqs) * 8;
    if (int he = vil_ensure_dyn_lds((const void*)k_mfma_prep_bwd, lds)) return he;
    vil_prof_begin(VIL_K_TABLE, s, 0, 0);
    k_mfma_prep_bwd<<<dim3((unsigned)(ntx * p.H + bc.nch + bc.nch + bc.nsplit + nzb)), dim3(256), lds, s>>>(
        p, c, bc, (int)p.k_st * 2, ntx, zr);
    vil_prof_end(s);
    if ((e = (int)hipGetLastError())) return e;
  }
  vil_prof_begin(VIL_K_DELTA, s, w.delta_bytes(), 0);
  BWD_SWITCH((k_mfma_delta<TT_, MD_><<<dim3((unsigned)(((int64_t)p.g.nx * p.g.ny * p.H * (MD_ == 4 ? 8 : (MD_ == 2 ? 4 : 2)) + 255) / 256), p.B), dim3(256), 0, s>>>(
      p, bc.do_hist ? bc.norm2 : nullptr)));
  vil_prof_end(s);
  if ((e = (int)hipGetLastError())) return e;
  {
    const size_t lds = dq_lds(c, bc);
    const int64_t units_total = (int64_t)p.B * p.H * bc.dq_units_bh;
    vil_prof_begin(VIL_K_MFMA_DQ, s, w.dq_bytes(), w.dq_flops());
    BWD_SWITCH({
      const void* kf_ = (const void*)k_mfma_bwd_dq<TT_, MD_, (MD_ >= 2 ? 2 : 4)>;
      if (int he = vil_ensure_dyn_lds(kf_, lds)) return he;
      const int grid = vil_persistent_grid(dq_waves(MD_), bc.dq_wpw, lds, p.H, units_total);
      if (grid < bc.dq_nwg) bc.dq_nwg = grid;              // (never above the grid the workspace was laid out for)
      k_mfma_bwd_dq<TT_, MD_, (MD_ >= 2 ? 2 : 4)><<<dim3((unsigned)bc.dq_nwg), dim3(64 * bc.dq_wpw), lds, s>>>(p, c, bc);
    });
    vil_prof_end(s);
    if ((e = (int)hipGetLastError())) return e;
  }
  {
    const size_t lds = kv_lds(c, bc);
    const unsigned grid = (unsigned)(p.B * p.H * bc.kv_wg_per_bh);
    vil_prof_begin(VIL_K_MFMA_DKDV, s, w.dkdv_bytes(), w.dkdv_flops());
    BWD_SWITCH({
      constexpr int KT_ = (MD_ >= 3 ? 2 : (MD_ == 2 ? VIL_KV_KT32 : 4));
      if (bc.kv_epre == 3) {
        if (int he = vil_ensure_dyn_lds((const void*)k_mfma_bwd_dkdv<TT_, MD_, KT_, 3>, lds)) return he;
        k_mfma_bwd_dkdv<TT_, MD_, KT_, 3><<<dim3(grid), dim3(64 * bc.kv_wpw), lds, s>>>(p, c, bc);
      } else if (bc.kv_epre == 7) {
        if (int he = vil_ensure_dyn_lds((const void*)k_mfma_bwd_dkdv<TT_, MD_, KT_, 7>, lds)) return he;
        k_mfma_bwd_dkdv<TT_, MD_, KT_, 7><<<dim3(grid), dim3(64 * bc.kv_wpw), lds, s>>>(p, c, bc);
      } else {
        if (int he = vil_ensure_dyn_lds((const void*)k_mfma_bwd_dkdv<TT_, MD_, KT_, 10>, lds)) return he;
        k_mfma_bwd_dkdv<TT_, MD_, KT_, 10><<<dim3(grid), dim3(64 * bc.kv_wpw), lds, s>>>(p, c, bc);
      }
    });
    vil_prof_end(s);
    if ((e = (int)hipGetLastError())) return e;
  }
  {
    const int nglo = p.G > 0 ? p.B * p.H * p.G : 0, nhx = (c.tabsize + 63) / 64;
    const int nh = bc.do_hist ? nhx * p.H : 0;
    if (nglo + nh > 0) {
      vil_prof_begin(VIL_K_REDUCE_GLO, s, 0, 0);
      if (d->dtype == VIL_DTYPE_F16)
        k_mfma_post_bwd<_Float16><<<dim3((unsigned)(nglo + nh)), dim3(1024), 0, s>>>(p, c, bc, bc.gq_nrec, nglo, nhx);
      else
        k_mfma_post_bwd<__bf16><<<dim3((unsigned)(nglo + nh)), dim3(1024), 0, s>>>(p, c, bc, bc.gq_nrec, nglo, nhx);
      vil_prof_end(s);
    }
  }
  return (int)hipGetLastError();
}
