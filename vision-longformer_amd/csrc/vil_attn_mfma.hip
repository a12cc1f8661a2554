// vil_attn_mfma.hip -- the MFMA kernel family (bf16 I/O, fp32 accumulate) for
// gfx950 / CDNA4.  Forward: QK^T-within-window, relative-position bias, mask,
// online softmax and .V fused in one pass; nothing but q/k/v/out/lse touches HBM.
//
// Work decomposition.  One 64-lane wavefront owns one query chunk of one
// (image, head) -- W*W queries in 64 "query slots" -- and walks the key slots
// of its 3x3 chunk neighbourhood (+ the G global keys) in steps of 32 keys.
// Waves never synchronise with each other after the workgroup's bias table
// is in LDS: each wave has private LDS for its slot table and its V tile.
//
//   S^T tile (16 keys x 16 queries)  = mfma_16x16x32_bf16(A = K rows, B = Q rows, C = bias)
//        K and Q fragments are loaded straight from HBM/L2 (8 contiguous head
//        dims per lane = the natural row layout, 16-byte buffer loads); the bias (and
//        the -inf of masked / out-of-image keys, and the exact-window mask) enters
//        as the accumulator's initial value, gathered from an LDS copy of this
//        head's bias table straight into the accumulator registers: one v_sub per
//        key, the lane's four y-consecutive queries are immediate offsets.
//   softmax: a lane holds 8 keys x 1 query per step; the row maximum is
//        deferred (only when a score exceeds the running maximum by > 8 nats
//        does the wave re-synchronise maxima across its four lane groups and
//        rescale), p = exp2(fma(s, c, -m c)).
//   O^T tile (16 dims x 16 queries) += mfma(A = V^T, B = P^T): P^T is the S^T
//        accumulator itself (bf16-packed, no cross-lane movement: the key
//        permutation it implies is applied to V^T instead); V^T comes from the
//        wave's LDS V tile through ds_read_b64_tr_b16.  The row sums come from a
//        third MFMA against a constant ones-row, so the VALU never adds them.
//
// Reference semantics: src/models/layers/longformer2d.py:134-204 (see include/vil_attn.h).
#include "vil_mfma_common.h"
#include <type_traits>

// ------------------------------------------------------------------ prologue (tables: see vil_mfma_common.h)
__global__ __launch_bounds__(256) void k_mfma_prep(VilParams p, MfmaCfg c, int row_stride_b, int ntx) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int blk = blockIdx.x, ntab = ntx * p.H;
  if (blk < ntab) {
    const int h = blk / ntx, bx = blk - h * ntx;
    const int e = bx * 256 + threadIdx.x;
    table_element(p, c, (float*)c.tabws, h, e, 1.0f / p.scale);
    if (c.gq_on && e >= c.tabsize && e < c.tabsize + c.gq_ext) {
      // image of the global query's column (vil_attn_fwd_full): entry i stands for the key address term
      // Ak = Amax - i; it carries g2l[0][h][0] / scale where Ak = x * P + y with 0 <= x, y < W -- a key of the chunk's own
      // chunk (neighbour offset (0, 0)) -- and the mask everywhere else: other neighbours, padding and global key slots
      const int W = p.g.W;
      const int amax = (2 * W - 1) * c.P + 2 * W - 1, amin = -W * c.P - W;
      const int t = amax - (e - c.tabsize) - amin;             // Ak - Amin
      float v = VIL_MASK_VAL;
      if (t >= 0) {
        const int xt = t / c.P - W, yt = t % c.P - W;
        if (xt >= 0 && xt < W && yt >= 0 && yt < W) v = p.g2l0 ? p.g2l0[h * p.G] / p.scale : 0.f;
      }
      ((float*)c.tabws)[(int64_t)h * c.tabstride + e] = v;
    }
  } else if (threadIdx.x < 64) {
    key_slots_block(p, c, blk - ntab, threadIdx.x, row_stride_b, smem);
  }
}

// ------------------------------------------------------------------ forward
// Tuning switches of the head_dim <= 32 instantiations (A/B builds: tools/ab/build_variants.sh).  Measured on one
// MI355X box, ViL-Small stage 1 (56x56, W 7, M 32, B 128), round-1 kernel 285 us:
//   pipeline 0, ring 2, 2 waves/SIMD  263 us      pipeline 0, ring 1, 3 waves/SIMD (168 VGPRs)  243 us   <- default
//   pipeline 0, ring 2, 3 waves/SIMD  274 us      pipeline 1 (252 VGPRs, 2 waves)              272 us
//   4 waves/SIMD (128 VGPRs, 256 B of scratch)  686 us
// i.e. occupancy beats both a deeper prefetch ring and software pipelining: a wave issues one instruction per ~5
// cycles whatever it does, so the SIMD's issue rate scales with resident waves until the registers spill.
#ifndef VIL_FWD_PIPE
#define VIL_FWD_PIPE 0     // software pipeline over steps (S^T of step st+1 issued before the softmax of step st)
#endif
#ifndef VIL_FWD_PF
#define VIL_FWD_PF 1       // depth of the K / V prefetch ring
#endif
#ifndef VIL_FWD_WAVES
#define VIL_FWD_WAVES 3    // waves per SIMD the register allocation is held to (2: 256, 3: 168, 4: 128 VGPRs)
#endif
constexpr int fwd_waves(int MD) { return MD <= 2 ? VIL_FWD_WAVES : 2; }
template <typename T, int MD>
__global__ __launch_bounds__(256, fwd_waves(MD)) void k_mfma_fwd(VilParams p, MfmaCfg c) {
  typedef typename V16<T>::x8 X8;
  typedef typename V16<T>::x4 X4;
  constexpr int M = 16 * MD;
  constexpr int MK = (MD + 1) / 2;            // 32-wide K steps over the head dim
  constexpr int VCH = 2 * MD;                 // 16-byte chunks per V row
  constexpr bool PIPE = VIL_FWD_PIPE && MD <= 2;   // software pipeline over steps (two score tiles + two LDS V tiles live)
  constexpr int PF = MD <= 2 ? (PIPE ? 2 : VIL_FWD_PF) : 1;   // depth of the K / V prefetch ring (M = 64 runs at 244 registers)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const VilGeom& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lj = lane & 15, lg = lane >> 4;

  // logical order (image, workgroup-of-chunks, head): the H workgroups that walk the same chunks of one image run
  // back to back on one XCD, so both 64-byte halves (two heads) of every K / V cache line are consumed while the
  // line is L2-resident, and the in-flight K/V footprint of an XCD is H times smaller than with (image, head, ...)
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int b = fdiv(logical, c.m_wgbh), rem_ = logical - b * (c.wg_per_bh * p.H);
  const int wgi = fdiv(rem_, c.m_H), h = rem_ - wgi * p.H;
  const int bh = b * p.H + h;

  float* tab = (float*)smem;
  const int tabx = c.tabsize + (c.gq_on ? c.gq_ext : 0);      // bias image (+ the global query column's image)
  {
    const f32x4* src = (const f32x4*)(c.tabws + (int64_t)h * c.tabstride);
    for (int i = tid; i < (tabx >> 2); i += blockDim.x) ((f32x4*)tab)[i] = src[i];
  }
  __syncthreads();
  const unsigned tab_lds = lds_addr(smem);

  char* wbase = smem + (size_t)tabx * 4 + (size_t)wave * c.wave_lds;
  int* s_koff = (int*)wbase;                       // [NSP] byte offset of each key slot's K/V row
  int* s_akey = s_koff + c.NSP;                    // [NSP] bias-table address term (bytes)
  char* s_v = (char*)(s_akey + c.NSP);             // [32][M] bf16 V tile of the current step

  const int Nloc = g.nx * g.ny;
  const int kstride_b = (int)p.k_st * 2;
  const unsigned kv_bytes = (unsigned)(p.G + Nloc - 1) * (unsigned)kstride_b + M * 2;    // K / V rows of this (image, head)
  const __amdgpu_buffer_rsrc_t krs = make_rsrc_n((const T*)p.k + b * p.k_sb + h * p.k_sh, kv_bytes);
  const __amdgpu_buffer_rsrc_t vrs = make_rsrc_n((const T*)p.v + b * p.v_sb + h * p.v_sh, kv_bytes);
  const T* qb = (const T*)p.q + b * p.q_sb + h * p.q_sh;
  T* ob = (T*)p.o + b * p.o_sb + h * p.o_sh;
  const float c1 = p.scale * LOG2E;               // scores are kept unscaled: s*c1 is log2-domain
  const float thr = 8.0f / p.scale;               // deferred-max threshold (8 nats)
  const int W = g.W;

  // constant A operand whose row 0 is all ones: D[0][j] = sum_k P^T[k][j]
  X8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (T)(lj == 0 ? 1.0f : 0.0f);

  // lane-constant pieces of the V staging / tr-read addresses
  int vst_off[MD], vld_off[MD], vtr_off[2][MD];
#pragma unroll
  for (int it = 0; it < MD; ++it) {
    const int cid = it * 64 + lane;
    const int row = cid / VCH, chn = cid % VCH;
    vst_off[it] = row * (M * 2) + ((chn * 16) ^ tile_swz<MD>(row));
    vld_off[it] = chn * 16;
  }
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    const int row = hf * 16 + lg * 4 + (lj >> 2);
#pragma unroll
    for (int dt = 0; dt < MD; ++dt)
      vtr_off[hf][dt] = row * (M * 2) + ((dt * 32 + (lj & 3) * 8) ^ tile_swz<MD>(row));
  }
  const int lgo = lg * 16;

  for (int gi = 0; gi < c.gpw; ++gi) {
    const int unit = (wgi * c.gpw + gi) * c.wpw + wave;
    if (unit >= c.units_bh) break;
    const int ch = fdiv(unit, c.m_NWP), wp = unit - ch * c.NWP;
    const int cm = fdiv(ch, c.m_my), cn = ch - cm * g.my;

    // ---- query slots of this lane: column j of q-tile qt is query (x, y = 4*hq + qt)
    const int jj = wp * 16 + lj;
    const int qx = fdiv(jj, c.m_HQ), qhq = jj - qx * c.HQ;
    // vil_attn_fwd_full: q-tile 0 of the chunk's first unused (x, hq) pair is the GLOBAL query; its bias comes from the
    // column image behind the table (own-chunk keys: g2l[0], everything else masked)
    const bool gqcol = c.gq_on && jj == c.gq_jj;
    const unsigned aq0b = tab_lds + (gqcol ? c.tabsize + c.gq_a0 : min(qx, W - 1) * c.P + 4 * qhq) * 4;   // LDS byte address; + 4*qt per q-tile
    int qtok[4];
    bool qreal[4];
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      const int qy = 4 * qhq + qt;
      const int qr = cm * W + qx, qc = cn * W + qy;
      qreal[qt] = qx < W && qy < W && qr < g.nx && qc < g.ny;
      qtok[qt] = qreal[qt] ? qr * g.ny + qc : (cm * W) * g.ny + cn * W;
    }
    X8 qf[MK][4];
#pragma unroll
    for (int qt = 0; qt < 4; ++qt)
#pragma unroll
      for (int ks = 0; ks < MK; ++ks) {
        const int d0 = ks * 32 + lg * 8;
        X8 z = {};
        const T* qrow = (gqcol && qt == 0) ? (const T*)p.q_g + b * p.q_sb + h * p.q_sh : qb + (int64_t)qtok[qt] * p.q_st;
        qf[ks][qt] = d0 < M ? *(const X8*)(qrow + d0) : z;
      }
    // (the Q fragments above are requested BEFORE the chunk's key-slot table is fetched: a unit's prologue used to be
    //  table -> Q -> first K/V, three dependent round trips; the table's and Q's now overlap)
    const int nslots = load_key_slots(c, ch, lane, s_koff, s_akey);
    // (round 5 measured the slot count, the slot table and the workgroup's bias image in ONE round trip here: 212 vs 213 us
    //  at 56x56 -- three resident waves per SIMD already hide it -- and a loss on the 2 W^2-slot tables of random-shift training)

    f32x4 o[MD][4], lacc[4];
    float mrow[4];
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      mrow[qt] = VIL_M_INIT;
      lacc[qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int dt = 0; dt < MD; ++dt) o[dt][qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    const int nsteps = nslots >> 5;
    // global -> register prefetch ring, PF steps deep (2 where the registers allow it: one step of compute is
    // shorter than an L2 / HBM round trip under load, so a 1-deep ring left the wave parked at vmcnt(0))
    X8 kf[PF][2][MK];
    u32x4 vr[PF][MD];
    auto load_step = [&](auto slot_, int st) {
      constexpr int sl = decltype(slot_)::value;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int off = s_koff[st * 32 + hf * 16 + lj] + lgo;
#pragma unroll
        for (int ks = 0; ks < MK; ++ks) {
          X8 z = {};
          kf[sl][hf][ks] = (ks * 32 + lg * 8) < M ? buf_load8<T>(krs, off + ks * 64) : z;
        }
      }
#pragma unroll
      for (int it = 0; it < MD; ++it) {
        const int row = (it * 64 + lane) / VCH;
        vr[sl][it] = __builtin_amdgcn_raw_buffer_load_b128(vrs, s_koff[st * 32 + row] + vld_off[it], 0, 0);
      }
    };

    // One step = 32 key slots.  stage_s(st): the step's operands leave the prefetch ring (K fragments stay in
    // registers, the V tile goes to the wave's LDS tile of parity st & 1), the ring slot is refilled PF steps
    // ahead, and S^T = K Q^T + bias is issued (accumulator initialised with the gathered bias).
    auto stage_s = [&](auto slot_, int st, f32x4 (&sc)[2][4]) {
      constexpr int sl = decltype(slot_)::value;
      char* sv = s_v + (PIPE ? (st & 1) * (32 * M * 2) : 0);
      X8 kc_[2][MK];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int ks = 0; ks < MK; ++ks) kc_[hf][ks] = kf[sl][hf][ks];
#pragma unroll
      for (int it = 0; it < MD; ++it) *(u32x4*)(sv + vst_off[it]) = vr[sl][it];
      i32x4 ak[2];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) ak[hf] = *(const i32x4*)(s_akey + st * 32 + hf * 16 + lg * 4);
      if (st + PF < nsteps) load_step(slot_, st + PF);
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        lds_cvf tb[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) tb[r] = lds_f32(aq0b - (unsigned)ak[hf][r]);
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
          f32x4 acc = {tb[0][qt], tb[1][qt], tb[2][qt], tb[3][qt]};
#pragma unroll
          for (int ks = 0; ks < MK; ++ks)
            acc = mfma16(kc_[hf][ks], qf[ks][qt], acc);
          sc[hf][qt] = acc;
        }
      }
    };
    // online softmax with a deferred maximum: ONE (rare) branch per step covers all four query tiles, so the
    // common path of a step is a single basic block the scheduler can interleave with the neighbouring MFMAs
    auto softmax = [&](const f32x4 (&sc)[2][4], X8 (&pb)[4]) {
      float pm[4];
      bool grow = false;
#pragma unroll
      for (int qt = 0; qt < 4; ++qt) {
        pm[qt] = max3f(max3f(max3f(sc[0][qt][0], sc[0][qt][1], sc[0][qt][2]), sc[0][qt][3], sc[1][qt][0]),
                       max3f(sc[1][qt][1], sc[1][qt][2], sc[1][qt][3]), mrow[qt]);     // >= mrow
        grow |= pm[qt] > mrow[qt] + thr;
      }
      if (__any(grow)) {
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
          float mn = fmaxf(pm[qt], __shfl_xor(pm[qt], 16, 64));
          mn = fmaxf(mn, __shfl_xor(mn, 32, 64));
          const float alpha = __builtin_amdgcn_exp2f((mrow[qt] - mn) * c1);   // 1 for the tiles that did not grow
          mrow[qt] = mn;
          lacc[qt] *= alpha;
#pragma unroll
          for (int dt = 0; dt < MD; ++dt) o[dt][qt] *= alpha;
        }
      }
#pragma unroll
      for (int qt = 0; qt < 4; ++qt) {
        const float mc = mrow[qt] * c1;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            pb[qt][hf * 4 + r] = (T)__builtin_amdgcn_exp2f(__builtin_fmaf(sc[hf][qt][r], c1, -mc));
      }
    };
    // O^T += V^T P^T ; row sums via the ones-row
    auto pv = [&](int st, const X8 (&pb)[4]) {
      const char* sv = s_v + (PIPE ? (st & 1) * (32 * M * 2) : 0);
#pragma unroll
      for (int dt = 0; dt < MD; ++dt) {
        X8 vt;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const s16x4 t4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (s16x4 __attribute__((address_space(3)))*)(sv + vtr_off[hf][dt]));
          const X4 tb4 = __builtin_bit_cast(X4, t4);
#pragma unroll
          for (int e = 0; e < 4; ++e) vt[hf * 4 + e] = tb4[e];
        }
#pragma unroll
        for (int qt = 0; qt < 4; ++qt)
          o[dt][qt] = mfma16(vt, pb[qt], o[dt][qt]);
      }
#pragma unroll
      for (int qt = 0; qt < 4; ++qt)
        lacc[qt] = mfma16(ones, pb[qt], lacc[qt]);
    };

    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, PF - 1> S1;
    load_step(S0{}, 0);
    if constexpr (PF == 2) { if (nsteps > 1) load_step(S1{}, 1); }
    if constexpr (PIPE) {
      // software pipeline over steps: S^T of step st+1 is ISSUED before the softmax of step st, so its LDS gathers
      // and MFMAs run under that softmax's VALU work, and the P V MFMAs of step st run under the next step's
      // gathers.  A single wave otherwise walks one long dependency chain per step (LDS gather -> MFMA -> max ->
      // exp -> LDS transpose-read -> MFMA) with two waves per SIMD to cover it: PMC showed the wave issuing 42 %
      // of its cycles, and cutting the VALU instructions per step from 196 to 120 moved the time by only 7 %.
      f32x4 scA[2][4], scB[2][4];
      X8 pb[4];
      stage_s(S0{}, 0, scA);
      for (int st = 0; st < nsteps; st += 2) {
        if (st + 1 < nsteps) stage_s(S1{}, st + 1, scB);
        softmax(scA, pb);
        pv(st, pb);
        if (st + 1 < nsteps) {
          if (st + 2 < nsteps) stage_s(S0{}, st + 2, scA);
          softmax(scB, pb);
          pv(st + 1, pb);
        }
      }
    } else {
      auto one = [&](auto slot_, int st) {
        f32x4 sc[2][4];
        X8 pb[4];
        stage_s(slot_, st, sc);
        softmax(sc, pb);
        wave_lds_fence();
        pv(st, pb);
        wave_lds_fence();
      };
      for (int st = 0; st < nsteps; st += PF) {
        one(S0{}, st);
        if constexpr (PF == 2) { if (st + 1 < nsteps) one(S1{}, st + 1); }
      }
    }

    // ---- epilogue: normalise, store O (4 dims x 8 bytes per d-tile) and LSE
    float l_q0 = 0.f;
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      const float l = __shfl(lacc[qt][0], lj, 64);      // row 0 lives in lane group 0
      if (qt == 0) l_q0 = l;
      const float inv = 1.0f / l;
      if (qreal[qt]) {
#pragma unroll
        for (int dt = 0; dt < MD; ++dt) {
          X4 w;
#pragma unroll
          for (int r = 0; r < 4; ++r) w[r] = (T)(o[dt][qt][r] * inv);
          *(X4*)(ob + (int64_t)qtok[qt] * p.o_st + dt * 16 + lg * 4) = w;
        }
        if (lg == 0)
          p.lse[(int64_t)bh * Nloc + qtok[qt]] = mrow[qt] * p.scale + __logf(l);
      }
    }
    if (gqcol) {      // the global query's partial over this chunk's own keys: O (unnormalised), l, m -> k_gq_merge
      float* part = c.gq_parts + ((int64_t)bh * (g.mx * g.my) + ch) * (M + 4);      // (16-byte records)
#pragma unroll
      for (int dt = 0; dt < MD; ++dt)
        *(f32x4*)(part + dt * 16 + lg * 4) = o[dt][0];
      if (lg == 0) { part[M] = l_q0; part[M + 1] = mrow[0]; }
    }
    wave_lds_fence();
  }
}

// The global token's output row from the chunks' partials (vil_attn_fwd_full, G == 1): one workgroup of 64 threads per
// (image, head).  Partial u: O_u = sum_k 2^((s_k - m_u) c1) v_k over the keys of chunk u, l_u the same sum without v, m_u
// the running maximum (score domain: s = q.k + bias / scale, c1 = scale * log2 e); the global key itself is one more
// partial (m = q_g.k_g + g2g / scale, l = 1, O = v_g).  out_g = sum_u w_u O_u / sum_u w_u l_u with w_u = 2^((m_u - m*) c1),
// lse_g = m* scale + ln(sum_u w_u l_u) -- the definition k_glo_fwd uses (reference longformer2d.py:210-227).
template <typename T>
__global__ __launch_bounds__(1024) void k_gq_merge(VilParams p, MfmaCfg c) {
  // 16 waves: wave w walks partials w, w + 16, ... (four records in flight per lane), lane = head dim; the waves' sums
  // are added in wave order -- a fixed order, the result is bit-reproducible.  (A 64-thread version walked the 64 .. 196
  // records of an (image, head) one dependent round trip after the other: 21 - 43 us.)
  __shared__ float s_o[16][64], s_l[16], s_mx[16];
  const int bh = blockIdx.x, b = bh / p.H, h = bh - b * p.H;
  const int M = p.M, nch = p.g.mx * p.g.my, t = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int RS = M + 4;
  const float c1 = p.scale * LOG2E;
  const float* parts = c.gq_parts + (int64_t)bh * nch * RS;
  const T* qg = (const T*)p.q_g + b * p.q_sb + h * p.q_sh;
  const T* kg = (const T*)p.k + b * p.k_sb + h * p.k_sh;
  const T* vg = (const T*)p.v + b * p.v_sb + h * p.v_sh;
  // the global key's score (every wave computes it: no broadcast needed)
  float sg = t < M ? (float)qg[t] * (float)kg[t] : 0.f;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sg += __shfl_xor(sg, o, 64);
  sg += p.g2g ? p.g2g[(int64_t)h * p.G * p.G] / p.scale : 0.f;
  // maximum over all partials: lane t of wave w reads m of partials w * 64 + t, ...
  float mx = sg;
  for (int u = threadIdx.x; u < nch; u += 1024) mx = fmaxf(mx, parts[(int64_t)u * RS + M + 1]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if (t == 0) s_mx[w] = mx;
  __syncthreads();
  mx = s_mx[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) mx = fmaxf(mx, s_mx[i]);
  float acc = 0.f, L = 0.f;
  for (int u0 = w; u0 < nch; u0 += 64) {
    float mu[4], lu[4], ou[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int u = min(u0 + 16 * k, nch - 1);
      const float* pu = parts + (int64_t)u * RS;
      mu[k] = pu[M + 1]; lu[k] = pu[M]; ou[k] = t < M ? pu[t] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (u0 + 16 * k < nch) {
        const float wt = __builtin_amdgcn_exp2f((mu[k] - mx) * c1);
        L = __builtin_fmaf(wt, lu[k], L);
        acc = __builtin_fmaf(wt, ou[k], acc);
      }
  }
  s_o[w][t] = acc;
  if (t == 0) s_l[w] = L;
  __syncthreads();
  if (w != 0) return;
  acc = 0.f; L = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc += s_o[i][t]; L += s_l[i]; }
  const float wg = __builtin_amdgcn_exp2f((sg - mx) * c1);
  L += wg;
  if (t < M) {
    acc = __builtin_fmaf(wg, (float)vg[t], acc);
    ((T*)p.o_g + b * p.o_sb + h * p.o_sh)[t] = (T)(acc / L);
  }
  if (t == 0) ((float*)p.lse_g)[bh] = mx * p.scale + __logf(L);
}

// ===================================================================== host side
bool vil_mfma_make_cfg(const VilAttnDesc* d, MfmaCfg& c) {
  memset(&c, 0, sizeof(c));
  VilGeom g; vil_geom_init(g, d->nx, d->ny, d->W, d->exact, d->mode);
  const int W = d->W;
  c.HQ = (W + 3) / 4;
  c.NWP = (W * c.HQ + 15) / 16;
  c.trows = d->mode == -1 ? 2 * W - 1 : 4 * W - 1;
  c.tcen = (c.trows - 1) / 2;
  int P = c.trows + 3 + 2 * VIL_CPAD;
  while ((P & 31) != 11) ++P;
  c.P = P;
  const int aqmax = (W - 1) * P + 4 * (c.HQ - 1) + 3;
  c.gsz = ((aqmax + 8) / 4) * 4;
  c.guard0 = ((c.trows * P + 3) / 4) * 4;
  c.glo0 = c.guard0 + c.gsz;
  c.tabsize = ((c.glo0 + d->G * c.gsz + 4 + 3) / 4) * 4;
  c.tabstride = c.tabsize;
  c.aconst = c.tcen * (P + 1) + VIL_CPAD;
  c.magicW = vil_magic((unsigned)W);
  c.magicW2 = vil_magic((unsigned)(W * W));
  c.NS = d->G + g.nact * g.W2;
  c.NSP = (c.NS + 31) & ~31;
  c.units_bh = g.mx * g.my * c.NWP;
  c.wave_lds = ((c.NSP * 8 + ((VIL_FWD_PIPE && d->M <= 32) ? 2 : 1) * 32 * d->M * 2 + 15) / 16) * 16;   // slot tables + V tile(s) (PIPE: two)
#ifndef VIL_FWD_WPW
#define VIL_FWD_WPW 4
#endif
  c.wpw = VIL_FWD_WPW;
  while (c.wpw > 1 && (size_t)c.tabsize * 8 + (size_t)c.wpw * c.wave_lds > 160 * 1024) c.wpw >>= 1;
  const int groups = (c.units_bh + c.wpw - 1) / c.wpw;
#ifndef VIL_FWD_WGS
#define VIL_FWD_WGS 8192   // target workgroup count: a workgroup walks gpw groups of chunks one after the other.
                           // Few long-lived workgroups (2048) spread the chunks in flight on an XCD over 5 images, whose
                           // K/V no longer fit its 4 MB L2; 8192 (one group per workgroup at ViL's sizes): 241 -> 222 us
#endif
  int gpw = (int)(((int64_t)d->B * d->H * groups) / VIL_FWD_WGS);
  if (gpw < 1) gpw = 1;
  if (gpw > groups) gpw = groups;
  c.gpw = gpw;
  c.wg_per_bh = (groups + gpw - 1) / gpw;
  c.m_wgbh = vil_magic((unsigned)(c.wg_per_bh * d->H)); c.m_H = vil_magic((unsigned)d->H);
  c.m_NWP = vil_magic((unsigned)c.NWP); c.m_my = vil_magic((unsigned)g.my); c.m_HQ = vil_magic((unsigned)c.HQ);
  return true;
}

static size_t mfma_lds_bytes(const MfmaCfg& c) { return (size_t)(c.tabsize + (c.gq_on ? c.gq_ext : 0)) * 4 + (size_t)c.wpw * c.wave_lds; }

// vil_attn_fwd_full: can the global token's query row ride in the forward pass?  One global token (every published
// model), local keys attended, and an unused (x, hq) pair in the chunk's last wave (W = 8: all 16 pairs are queries).
static bool gq_fusable(const VilAttnDesc* d, const MfmaCfg& c) {
  return d->G == 1 && !d->only_glo && d->dtype != VIL_DTYPE_F32 && d->W * c.HQ < 16 * c.NWP;
}
static void gq_cfg(const VilAttnDesc* d, MfmaCfg& c) {
  const int W = d->W;
  const int amax = (2 * W - 1) * c.P + 2 * W - 1, amin = -W * c.P - W;
  c.gq_jj = W * c.HQ;
  c.gq_a0 = amax - c.aconst;
  // entry i <-> key address term Amax - i; padding / global key slots (terms below -guard0) land behind the window,
  // inside the extension, on masked entries
  int ext = amax - amin + 4;
  if (amax - c.aconst + c.glo0 + d->G * c.gsz + 4 > ext) ext = amax - c.aconst + c.glo0 + d->G * c.gsz + 4;
  c.gq_ext = (ext + 3) & ~3;
  c.tabstride = c.tabsize + c.gq_ext;
  c.gq_on = 1;
}

int vil_mfma_bwd_supported(const VilAttnDesc* d);
int vil_mfma_launch_gq_merge(const VilAttnDesc* d, const VilParams& p, const MfmaCfg& c, hipStream_t s);
size_t vil_mfma_bwd_workspace(const VilAttnDesc* d);

int vil_mfma_launch_prep(const VilParams& p, const MfmaCfg& c, int row_stride_b, hipStream_t s) {
  const int ntx = (c.tabsize + (c.gq_on ? c.gq_ext : 0) + 255) / 256, nch = p.g.mx * p.g.my;
  if (int he = vil_ensure_dyn_lds((const void*)k_mfma_prep, (size_t)c.NSP * 8)) return he;
  k_mfma_prep<<<dim3((unsigned)(ntx * p.H + nch)), dim3(256), (size_t)c.NSP * 8, s>>>(p, c, row_stride_b, ntx);
  return (int)hipGetLastError();
}

int vil_mfma_supported(const VilAttnDesc* d, int pass) {
  if (d->dtype == VIL_DTYPE_F32) return vil_f32_supported(d, pass);
  if (d->dtype != VIL_DTYPE_BF16 && d->dtype != VIL_DTYPE_F16) return VIL_E_DTYPE;
  if (d->M != 16 && d->M != 32 && d->M != 48 && d->M != 64) return VIL_E_HEAD_DIM;
  if (d->W < 1 || d->W > 32) return VIL_E_WINDOW;
  if (d->G > 16) return VIL_E_BACKEND;
  // 16-byte row loads: token/batch/head strides and M must keep rows 16-byte aligned
  if ((d->q_st | d->k_st | d->v_st | d->q_sb | d->k_sb | d->v_sb | d->q_sh | d->k_sh | d->v_sh) & 7) return VIL_E_ALIGN;
  if ((d->o_st | d->o_sb | d->o_sh) & 3) return VIL_E_ALIGN;
  // K and V rows are addressed through one slot table of 32-bit byte offsets (views of one kv tensor)
  const int64_t ntok = (int64_t)d->G + (int64_t)d->nx * d->ny;
  if (d->k_st != d->v_st || d->k_st >= (1 << 22) || ntok >= (1 << 23) || d->k_st * 2 * ntok >= (1ll << 31))
    return VIL_E_BACKEND;
  MfmaCfg c; vil_mfma_make_cfg(d, c);
  if (mfma_lds_bytes(c) > 160 * 1024) return VIL_E_BACKEND;
  // the workgroup index is decoded with magic-number divisions, exact for index * divisor < 2^32 (fdiv)
  if ((uint64_t)d->B * d->H * c.wg_per_bh * ((uint64_t)c.wg_per_bh * d->H) >= (1ull << 32)) return VIL_E_BACKEND;
  return pass == 0 ? VIL_OK : vil_mfma_bwd_supported(d);
}

size_t vil_mfma_workspace(const VilAttnDesc* d, int pass) {
  if (d->dtype == VIL_DTYPE_F32) return vil_f32_workspace(d, pass);
  if (pass != 0) return vil_mfma_bwd_workspace(d);
  MfmaCfg c; vil_mfma_make_cfg(d, c);
  VilGeom g; vil_geom_init(g, d->nx, d->ny, d->W, d->exact, d->mode);
  size_t fl = (size_t)d->H * c.tabsize + vil_key_slots_floats(c, g.mx * g.my);
  if (gq_fusable(d, c)) {       // vil_attn_fwd_full: wider images + one partial per (image, head, chunk)
    gq_cfg(d, c);
    fl = (size_t)d->H * c.tabstride + vil_key_slots_floats(c, g.mx * g.my) + (size_t)d->B * d->H * g.mx * g.my * (d->M + 4) + 4;
  }
  size_t bytes = fl * sizeof(float);
  if (vil_cw_supported(d, 0) == VIL_OK) { const size_t b2 = vil_cw_workspace(d, 0); if (b2 > bytes) bytes = b2; }
  return bytes;
}

int vil_mfma_fwd(const VilAttnDesc* d, VilParams& p, hipStream_t s) {
  if (d->dtype == VIL_DTYPE_F32) return vil_f32_fwd(d, p, s);
  // round 6: the chunk-workgroup kernels (vil_attn_cw.hip) wherever they apply; VIL_BACKEND_MFMA_WAVE keeps the
  // wave-per-chunk kernels below selectable (A/B, and every shape the new family declines)
  // (measured, same-box, hipEvents: 3x3 neighbourhoods with W <= 8 run 3 - 10 % faster on them; the two-chunk lists of
  //  random-shift training and W = 12 -- five waves per chunk -- slower: those stay on the kernels below unless the caller
  //  asks for the new family by name)
#ifndef VIL_CW_AUTO
#define VIL_CW_AUTO 1      // 0: AUTO / MFMA never take the chunk-workgroup kernels (A/B builds: tools/ab/build_file_variants.sh)
#endif
  const bool cw_wins = VIL_CW_AUTO && d->mode == 0 && d->W <= 8 && !d->only_glo;
  if (d->backend != VIL_BACKEND_MFMA_WAVE && (cw_wins || d->backend == VIL_BACKEND_MFMA_CW) && vil_cw_supported(d, 0) == VIL_OK) {
    const int r = vil_cw_fwd(d, p, s);
    if (r != VIL_E_BACKEND) return r;
  }
  MfmaCfg c; vil_mfma_make_cfg(d, c);
  float* tabws = (float*)p.delta;          // workspace base
  c.tabws = tabws;
  if (p.glo_rows) {                        // vil_attn_fwd_full
    if (!gq_fusable(d, c)) return VIL_E_BACKEND;
    // ... and only where the column's image does not cost resident waves: at W = 12 / head_dim 64 it is 14 KB on top of
    // 79 KB per workgroup -- one workgroup per CU instead of two (forward 140 -> 190 us at 48x48, same-box A/B)
    const int cap = fwd_waves(d->M / 16) * 4;
    const size_t lds0 = mfma_lds_bytes(c);
    const int res0 = (int)((160 * 1024) / lds0) * c.wpw < cap ? (int)((160 * 1024) / lds0) * c.wpw : cap;
    gq_cfg(d, c);
    const size_t lds1 = mfma_lds_bytes(c);
    const int res1 = lds1 > 160 * 1024 ? 0 : ((int)((160 * 1024) / lds1) * c.wpw < cap ? (int)((160 * 1024) / lds1) * c.wpw : cap);
    if (res1 < res0) return VIL_E_BACKEND;
  }
  if (((uintptr_t)p.q | (uintptr_t)p.k | (uintptr_t)p.v | (uintptr_t)tabws) & 15) return VIL_E_ALIGN;
  if ((uintptr_t)p.o & 7) return VIL_E_ALIGN;
  const VilWork w(d);
  const int nch = p.g.mx * p.g.my;
  c.key_slots = (int2*)(tabws + (size_t)p.H * c.tabstride);         // (tabsize / tabstride are multiples of 4 floats)
  c.key_nslots = (int*)(c.key_slots + (size_t)nch * c.NSP);
  if (c.gq_on) c.gq_parts = tabws + (((size_t)p.H * c.tabstride + vil_key_slots_floats(c, nch) + 3) & ~(size_t)3);
  vil_prof_begin(VIL_K_TABLE, s, 0, 0);
  {
    const int ntx = (c.tabsize + (c.gq_on ? c.gq_ext : 0) + 255) / 256;
    if (int he = vil_ensure_dyn_lds((const void*)k_mfma_prep, (size_t)c.NSP * 8)) return he;
    k_mfma_prep<<<dim3((unsigned)(ntx * p.H + nch)), dim3(256), (size_t)c.NSP * 8, s>>>(p, c, (int)p.k_st * 2, ntx);
  }
  vil_prof_end(s);
  int e = (int)hipGetLastError();
  if (e) return e;
  vil_prof_begin(VIL_K_MFMA_FWD, s, w.fwd_bytes(), w.fwd_flops());
  const unsigned grid = (unsigned)(p.B * p.H * c.wg_per_bh);
  const size_t lds = mfma_lds_bytes(c);
#define LAUNCH_FWD(MD_)                                                                              \
  {                                                                                                  \
    if (d->dtype == VIL_DTYPE_F16) {                                                                 \
      if (int he = vil_ensure_dyn_lds((const void*)k_mfma_fwd<_Float16, MD_>, lds)) return he;        \
      k_mfma_fwd<_Float16, MD_><<<dim3(grid), dim3(64 * c.wpw), lds, s>>>(p, c);                      \
    } else {                                                                                         \
      if (int he = vil_ensure_dyn_lds((const void*)k_mfma_fwd<__bf16, MD_>, lds)) return he;          \
      k_mfma_fwd<__bf16, MD_><<<dim3(grid), dim3(64 * c.wpw), lds, s>>>(p, c);                        \
    }                                                                                                \
  }
  switch (d->M) {
    case 16: LAUNCH_FWD(1); break;
    case 32: LAUNCH_FWD(2); break;
    case 48: LAUNCH_FWD(3); break;
    case 64: LAUNCH_FWD(4); break;
    default: return VIL_E_HEAD_DIM;
  }
  vil_prof_end(s);
  if ((e = (int)hipGetLastError())) return e;
  if (c.gq_on) return vil_mfma_launch_gq_merge(d, p, c, s);
  return (int)hipGetLastError();
}

int vil_mfma_launch_gq_merge(const VilAttnDesc* d, const VilParams& p, const MfmaCfg& c, hipStream_t s) {
  vil_prof_begin(VIL_K_GLO_FWD, s, 0, 0);
  if (d->dtype == VIL_DTYPE_F16) k_gq_merge<_Float16><<<dim3((unsigned)(p.B * p.H)), dim3(1024), 0, s>>>(p, c);
  else k_gq_merge<__bf16><<<dim3((unsigned)(p.B * p.H)), dim3(1024), 0, s>>>(p, c);
  vil_prof_end(s);
  return (int)hipGetLastError();
}
