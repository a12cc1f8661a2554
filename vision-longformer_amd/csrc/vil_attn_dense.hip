// vil_attn_dense.hip -- the dense `Attention` of the s0 stages (reference src/models/msvit.py:91-120: every token
// attends every token, Swin-style relative position bias + the global-token bias terms) as its own kernel family for
// gfx950, 16-bit I/O (bf16 / fp16), fp32 accumulation, head_dim 64.
//
// Why not the one-chunk case of the sliding-chunk kernels (vil_attn_mfma*.hip, what these stages ran on through round 3):
// there one wave owns 64 query slots of a chunk and walks 3.5 - 10 key steps, so the per-wave set-up (slot tables, bias
// image, global-row helpers: 40 % of a wave's life at 14x14), five launches per backward and the B*H*ceil(N/64) wave
// quantisation dominate (14x14: 228 us per layer by hipEvents against 29 us of HBM time; 24x24: 333 us).  Here
//   * one workgroup = (image, head, up to 8 row units); one wave = one unit = 16*QT query rows (forward, dQ; S^T
//     orientation) or 16*KT key rows (dK/dV; S orientation) whose fragments come straight from HBM and stay in registers;
//   * the other side (K and V, or Q and dO) streams through a two-slot LDS ring in blocks of 64 rows filled by LDS-DMA
//     (DnDma below) and shared by the workgroup's waves: any sequence length, 32 KB of ring, loads under compute;
//   * the G global tokens are ordinary rows / columns of the same tiles (tokens 0..G-1): no k_glo_* launches.  Their bias
//     terms are constant regions behind the LDS bias table (global KEYS) and a select in the one tile that holds them
//     (global QUERIES; that wave's unit is a single tile so that it does not hold the others up at the block barriers);
//   * delta = rowsum(dO o O) is computed by the dQ pass (it holds dO) and handed to the dK/dV pass: no delta launch;
//   * d(table), d(g2l), d(g2g): int32 fixed-point LDS histogram in the dQ pass with a per-workgroup power-of-two scale
//     derived in the kernel (|dS| <= 2 max|dO_q| max|v_k|: its own dO rows, and max|v_k|^2 left behind the log-sum-exps
//     by the forward), one record per workgroup, summed in a fixed order by k_dense_reduce (bit-reproducible).
// Launches per layer: 1 forward, 3 backward (dQ, dK/dV, reduce).  Same-box hipEvent times per layer, bf16:
//   14x14 B128 H6: 58.8 + 169 us -> 32.5 + 138 us;  24x24 B32 H6: 79.8 + 253 us -> 57.8 + 171 us.
// What limits them now (s_memtime stamps per workgroup): one step of a wave is a ~2 100-cycle dependency chain
// (LDS reads -> MFMA -> max -> exp -> pack -> transposed LDS reads -> MFMA) whether the workgroup has the CU to itself
// or not; 128 registers allow two 7-wave workgroups per CU, and the younger one runs at 2/3 of the older one's pace.
//
// LDS bias table of head h (floats, everything pre-divided by `scale` so that one multiply by scale*log2e serves
// scores and bias):  [ TS = (2nx-1)(2ny-1) table entries | L x mask | G x L x g2l[1][h][g] ],  L = (TS+1)/2.
// With A(t) = x*(2ny-1) + y for local token t = G + x*ny + y, the entry of (query i, key j) is at A(i) - A(j) + L-1
// -- the reference's relative_position_index (msvit.py:74-85) without the index buffer.  Masked (padding) key slots
// and the global keys carry an A that lands every local query in their constant region.
#include "vil_mfma_common.h"
#include <type_traits>

#define DN_LSE_PAD 1.0e30f
#define DN_GGL 80               // floats of the small bias block in LDS: [4 q + k] g2g[h][q][k] / scale, [64 + g] g2l[0][h][g] / scale
#define DN_REC_EXTRA 32        // ints behind the TS bins of a record: [0] lfx, [2 + 2 g] int64 sum of global key g's region, [12 + 5 g] d g2l[0][g], [13 + 5 g + g'] d g2g[g][g']
// workgroup shape per kernel (F forward, Q dQ, K dK/dV): at most MAXW waves, OCC waves per SIMD (4: 128 registers)
#ifndef DN_MAXW_F
#define DN_MAXW_F 8
#endif
#ifndef DN_OCC_F
#define DN_OCC_F 4
#endif
#ifndef DN_MAXW_Q
#define DN_MAXW_Q 8
#endif
#ifndef DN_OCC_Q
#define DN_OCC_Q 4
#endif
// dQ pass with two tiles per wave (151 registers, 3 waves per SIMD, workgroups of <= 4 waves): less LDS traffic and per-step
// overhead per score; measured against the one-tile form (hipEvents, bf16): N 197: 51 vs 65 us, N 145: 22.7 vs 24.2,
// N 50: 27.9 vs 22.3, N 577: 86 vs 82 -- taken for 96 < N <= 384 (dense_dq_tiles)
#ifndef DN_MAXW_Q2
#define DN_MAXW_Q2 4
#endif
#ifndef DN_MAXW_K
#define DN_MAXW_K 8
#endif
#ifndef DN_OCC_K
#define DN_OCC_K 4
#endif
struct DenseCfg {
  int N, NSP, G, nx, ny;
  int P, TS, L, tabsize;
  int unit;                 // rows per wave: 16 * QT (forward, dQ) or 16 * KT (dK/dV)
  int nunits, nwg_bh, wpw;  // row units per (image, head), workgroups per (image, head), waves per workgroup
  int do_hist;
  int rec_base;             // ints of a record before its extras: TS rounded up to even (8-byte aligned int64 fields)
  unsigned m_ny, m_nwg;
  int* parts;               // dQ pass: (B*H*nwg_bh, rec_base + DN_REC_EXTRA) records
  float* delta;             // (B*H, N)
};

// LDS image of a (row, 64 x 16-bit) block: rows of 128 bytes, the 32-byte units of a row XORed with row bits [2:1].
// No padding -- which is what lets the block be filled by LDS-DMA (`buffer_load_dwordx4 ... lds` writes M0 + lane*16: a
// lane-linear image; the swizzle is applied to the SOURCE chunk each lane requests) -- and conflict-free both for the
// natural fragment reads (ds_read_b128: 16 rows x one 16-byte chunk per lane group) and the transposed ones
// (ds_read_b64_tr_b16: 8 rows x 32 bytes per 32-lane group).  Cost: one per-lane address register per k-step (natural)
// / per dim tile (transposed) instead of one per matrix.
__device__ __forceinline__ int dn_off(int row, int colb) { return row * 128 + (colb ^ (((row >> 1) & 3) << 5)); }

__device__ __forceinline__ int dn_A(const DenseCfg& c, int t) {
  if (t < c.G || t >= c.N) return 0;
  const int i = t - c.G, x = (int)fdiv((unsigned)i, c.m_ny);
  return x * c.P + (i - x * c.ny);
}

// entry e of head h's LDS bias table (see the file header) from its raw table value and the head's four g2l[1] terms.
// Branch-free on purpose: with the loads inside a divergent if / else chain the compiler waited for each one in turn
// (the prologue of a workgroup was six serial L2 round trips, ~9 500 cycles).
__device__ __forceinline__ float dn_table_val(const DenseCfg& c, int e, float raw, float g0, float g1, float g2, float g3) {
  const int r0 = c.TS + c.L;
  const float reg = e >= r0 + 3 * c.L ? g3 : (e >= r0 + 2 * c.L ? g2 : (e >= r0 + c.L ? g1 : g0));
  return e < c.TS ? raw : (e < r0 ? VIL_MASK_VAL : reg);
}
// key-slot address terms (4 * (A(slot) - (L-1)); constant regions for the global / padding slots) and the g2g block
__device__ __forceinline__ int dn_akey(const DenseCfg& c, int s) {
  return s < c.G ? -(c.TS + (1 + s) * c.L) * 4 : (s < c.N ? (dn_A(c, s) - (c.L - 1)) * 4 : -c.TS * 4);
}

// ---- streamed matrices.  The "other side" of a wave's rows (K and V in the forward and dQ passes, Q and dO in the dK/dV
// pass) moves through a two-slot LDS ring in blocks of 64 rows, filled by LDS-DMA: block j+1 is requested when block j's
// compute starts and has landed at the workgroup barrier that ends it (hipcc drains vmcnt before s_barrier), so loads
// and compute of ONE workgroup overlap, no register is spent on data in flight, and a sequence of any length needs
// 32 KB of ring.  History (14x14 forward, hipEvents): whole sequence staged up front through registers: every
// co-resident workgroup loads, then every one computes -- 18.8 us + 13.6 us = 32.4 us, no overlap (starting the second
// workgroup of a CU late changes nothing: a slot's own load -> compute chain stays serial); register-staged ring with
// one block in flight: 16 more VGPRs = 3 waves per SIMD, 38.7 us.
// A block is 8 pieces of 8 rows per matrix (one DMA instruction = 64 lanes x 16 bytes = 1 KB); the workgroup's waves take
// pieces round-robin.  Rows >= N read as zeros through the bounded descriptors.
#define DN_BLK 64
#define DN_RING (2 * DN_BLK * 128)       // bytes of one matrix's ring
struct DnDma {
  __amdgpu_buffer_rsrc_t ra, rb;
  int sa_b, sb_b, va0, vb0;
  char *la, *lb;
  int wave, nwaves;
  __device__ __forceinline__ void init(__amdgpu_buffer_rsrc_t ra_, int sa, char* ia, __amdgpu_buffer_rsrc_t rb_, int sb, char* ib,
                                       int lane, int wave_, int nwaves_) {
    ra = ra_; rb = rb_; sa_b = sa; sb_b = sb; la = ia; lb = ib; wave = wave_; nwaves = nwaves_;
    const int row = lane >> 3, chunk = (lane & 7) ^ (((row >> 1) & 3) << 1);     // LDS slot lane%8 of row lane/8 holds this chunk
    va0 = row * sa + chunk * 16; vb0 = row * sb + chunk * 16;
  }
  template <int BLK = DN_BLK> __device__ __forceinline__ void issue(int blk, int slot) {
    constexpr int PPM = BLK / 8;                        // pieces per matrix and block
    for (int pc = wave; pc < 2 * PPM; pc += nwaves) {
      const int m = pc / PPM, piece = pc % PPM;
      const int r0 = blk * BLK + piece * 8;
      char* dst = (m ? lb : la) + slot * (BLK * 128) + piece * 1024;
      if (m) __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)dst, 16, vb0 + r0 * sb_b, 0, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)dst, 16, va0 + r0 * sa_b, 0, 0, 0);
    }
  }
};

// workgroup prologue shared by the three kernels, in two halves: DN_PROLOGUE_LOADS issues every small global load of
// the workgroup back to back (the first 4*nthr bias-table entries, the head's g2l / g2g terms) next to the caller's own
// loads; DN_PROLOGUE_WRITES fills the LDS tables once they have landed.  One memory round trip in all.
#define DN_PROLOGUE_LOADS                                                                                \
  float ga_ = 0.f, gb_ = 0.f, gc_ = 0.f, gd_ = 0.f, ta_ = 0.f, tb_ = 0.f, tc_ = 0.f, td_ = 0.f, gg_ = 0.f, gl0 = 0.f; \
  if (p.has_g2l) {                                                                                       \
    ga_ = p.g2l[h * G]; gb_ = p.g2l[h * G + min(1, G - 1)];                                              \
    gc_ = p.g2l[h * G + min(2, G - 1)]; gd_ = p.g2l[h * G + min(3, G - 1)];                              \
    gl0 = p.g2l0[h * G + min(lj, G - 1)];                                                                \
  }                                                                                                      \
  if (p.has_bias) {                                                                                      \
    ta_ = p.table[(int64_t)min(tid, c.TS - 1) * p.H + h];                                                \
    tb_ = p.table[(int64_t)min(tid + nthr, c.TS - 1) * p.H + h];                                         \
    tc_ = p.table[(int64_t)min(tid + 2 * nthr, c.TS - 1) * p.H + h];                                     \
    td_ = p.table[(int64_t)min(tid + 3 * nthr, c.TS - 1) * p.H + h];                                     \
  }                                                                                                      \
  if (p.g2g && G > 0) gg_ = p.g2g[(h * G + min((tid >> 2) & 15, G - 1)) * G + min(tid & 3, G - 1)];
// (no arithmetic on the loaded values up there: a multiply right behind a load makes the compiler wait for it on the spot)
#define DN_PROLOGUE_WRITES(AKEY)                                                                         \
  {                                                                                                      \
    ga_ *= inv_s; gb_ *= inv_s; gc_ *= inv_s; gd_ *= inv_s; gl0 *= inv_s;                                \
    if (tid < c.tabsize) tab[tid] = dn_table_val(c, tid, ta_ * inv_s, ga_, gb_, gc_, gd_);               \
    if (tid + nthr < c.tabsize) tab[tid + nthr] = dn_table_val(c, tid + nthr, tb_ * inv_s, ga_, gb_, gc_, gd_); \
    if (tid + 2 * nthr < c.tabsize) tab[tid + 2 * nthr] = dn_table_val(c, tid + 2 * nthr, tc_ * inv_s, ga_, gb_, gc_, gd_); \
    if (tid + 3 * nthr < c.tabsize) tab[tid + 3 * nthr] = dn_table_val(c, tid + 3 * nthr, td_ * inv_s, ga_, gb_, gc_, gd_); \
    for (int e = tid + 4 * nthr; e < c.tabsize; e += nthr)                                               \
      tab[e] = dn_table_val(c, e, p.has_bias ? p.table[(int64_t)min(e, c.TS - 1) * p.H + h] * inv_s : 0.f, ga_, gb_, gc_, gd_); \
    int* ak_ = AKEY;                                                                                     \
    if (ak_)                                                                                             \
      for (int s2 = tid; s2 < c.NSP; s2 += nthr) ak_[s2] = dn_akey(c, s2);                               \
    if (tid < 64) ggl[tid] = ((tid >> 2) < G && (tid & 3) < G) ? gg_ * inv_s : 0.f;                      \
    if (tid < 4) ggl[64 + tid] = tid < G ? gl0 : 0.f;                                                    \
  }

// ===================================================================== forward
// lse: (B*H, N + 1) floats -- [N] = max_k |v_k|^2 of the (image, head), which the backward's histogram scale needs
// BLK: rows of a ring block.  64 everywhere but where the 32 KB ring keeps a launch from fitting the chip in one round of
// workgroups (round 4: at 24 x 24 the 576 seven-wave workgroups need 52.7 KB each -- two per CU, 512 slots, a second
// round of 64 workgroups that costs 0.64 of the first; with 32-row blocks they need 36.7 KB, four per CU, one round).
template <typename T, int QT, int BLK>
__global__ __launch_bounds__(64 * DN_MAXW_F, DN_OCC_F) void k_dense_fwd(VilParams p, DenseCfg c) {
  constexpr int RING = 2 * BLK * 128, SPB = BLK / 32;  // bytes of one matrix's ring; 32-key steps per block
  typedef typename V16<T>::x8 X8;
  typedef typename V16<T>::x4 X4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, nthr = blockDim.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lj = lane & 15, lg = lane >> 4;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = (int)fdiv((unsigned)logical, c.m_nwg), wgi = logical - bh * c.nwg_bh;
  const int b = bh / p.H, h = bh - b * p.H;

  float* tab = (float*)smem;
  int* akey = (int*)(tab + c.tabsize);
  float* ggl = (float*)(akey + c.NSP);
  unsigned* misc = (unsigned*)(ggl + DN_GGL);
  char* Kl = (char*)(misc + 16);
  char* Vl = Kl + RING;
  const unsigned tab_lds = lds_addr(smem);

  const int N = c.N, G = c.G;
  const float c1 = p.scale * LOG2E;
  const float thr = 8.0f / p.scale;
  const float inv_s = 1.0f / p.scale;
  // the wave that owns the global-token query rows carries extra work per step (and every wave meets it at each block's
  // barrier): with global tokens its unit is ONE tile (tokens 0..15), all other units are QT tiles starting at token 16
  const int unit = wgi * c.wpw + wave;
  const bool split0 = G > 0 && QT > 1;
  const int q0 = split0 ? (unit == 0 ? 0 : 16 + (unit - 1) * (16 * QT)) : unit * (16 * QT);
  const T* qb = (const T*)p.q + b * p.q_sb + h * p.q_sh;
  T* ob = (T*)p.o + b * p.o_sb + h * p.o_sh;
  const bool glo_wave = G > 0 && unit == 0;
  const bool active = q0 < N;

  // ---- requests first: the first K / V block (LDS-DMA), this wave's Q fragments, the head's table
  DnDma dma;
  {
    const unsigned kbytes = (unsigned)(N - 1) * (unsigned)(p.k_st * 2) + 128u;
    const unsigned vbytes = (unsigned)(N - 1) * (unsigned)(p.v_st * 2) + 128u;
    dma.init(make_rsrc_n((const T*)p.k + b * p.k_sb + h * p.k_sh, kbytes), (int)p.k_st * 2, Kl,
             make_rsrc_n((const T*)p.v + b * p.v_sb + h * p.v_sh, vbytes), (int)p.v_st * 2, Vl, lane, wave, nthr >> 6);
  }
  const int nsteps = c.NSP >> 5, nblk = (nsteps + SPB - 1) / SPB;
  dma.template issue<BLK>(0, 0);
  int qtok[QT];
  X8 qf[2][QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    qtok[qt] = (split0 && unit == 0 && qt > 0) ? N : q0 + qt * 16 + lj;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      X8 z = {};
      qf[ks][qt] = qtok[qt] < N ? *(const X8*)(qb + (int64_t)qtok[qt] * p.q_st + ks * 32 + lg * 8) : z;
    }
  }
  DN_PROLOGUE_LOADS
  if (tid < 16) misc[tid] = 0u;
  DN_PROLOGUE_WRITES(akey)
  float vmax = 0.f;

  unsigned aqb[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) aqb[qt] = tab_lds + (unsigned)dn_A(c, qtok[qt]) * 4u;
  X8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (T)(lj == 0 ? 1.0f : 0.0f);
  int knat[2], vtr[4];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) knat[ks] = dn_off(lj, ks * 64 + lg * 16);
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) vtr[dt] = dn_off(lg * 4 + (lj >> 2), dt * 32 + (lj & 3) * 8);

  f32x4 o[4][QT], lacc[QT];
  float mrow[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    mrow[qt] = VIL_M_INIT;
    lacc[qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt][qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  lds_dma_wait();                                      // this wave's LDS-DMA requests (vil_mfma_common.h)
  __syncthreads();

  // one step = 32 keys of the block in ring slot `slot`.  (One instantiation for the wave that owns the global-token
  // query rows, one for everybody else: a run-time `if (glo_wave)` inside the step became per-score selects in every wave.)
  auto step = [&](auto glo_, int st, int slot) {
    constexpr bool GLO = decltype(glo_)::value;
    constexpr int NQ = (GLO && QT > 1) ? 1 : QT;          // the global-token wave's unit is one tile
    const char* kp = Kl + slot * (BLK * 128) + (st & (SPB - 1)) * (32 * 128);
    const char* vp = Vl + slot * (BLK * 128) + (st & (SPB - 1)) * (32 * 128);
    // ---- S^T = K Q^T + bias: the accumulator starts as the gathered bias
    i32x4 ak[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) ak[hf] = *(const i32x4*)(akey + st * 32 + hf * 16 + lg * 4);
    X8 kf[2][2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) kf[hf][ks] = *(const X8*)(kp + hf * (16 * 128) + knat[ks]);
    f32x4 sc[2][NQ];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int qt = 0; qt < NQ; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r) sc[hf][qt][r] = *lds_f32(aqb[qt] - (unsigned)ak[hf][r]);
    if (GLO) {
      // query tile 0 holds the global-token QUERY rows (tokens 0..G-1): g2g against the global keys, g2l[0] against
      // the local keys (msvit.py:97-100), mask against the padding slots
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int s = st * 32 + hf * 16 + lg * 4 + r;
          const float v = s < G ? ggl[lj * 4 + (s & 3)] : (s < N ? gl0 : VIL_MASK_VAL);
          if (lj < G) sc[hf][0][r] = v;
        }
    }
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int qt = 0; qt < NQ; ++qt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) sc[hf][qt] = mfma16(kf[hf][ks], qf[ks][qt], sc[hf][qt]);

    // ---- online softmax, deferred maximum (vil_attn_mfma.hip)
    float pm[NQ];
    bool grow = false;
#pragma unroll
    for (int qt = 0; qt < NQ; ++qt) {
      pm[qt] = max3f(max3f(max3f(sc[0][qt][0], sc[0][qt][1], sc[0][qt][2]), sc[0][qt][3], sc[1][qt][0]),
                     max3f(sc[1][qt][1], sc[1][qt][2], sc[1][qt][3]), mrow[qt]);
      grow |= pm[qt] > mrow[qt] + thr;
    }
    if (__any(grow)) {
#pragma unroll
      for (int qt = 0; qt < NQ; ++qt) {
        float mn = fmaxf(pm[qt], __shfl_xor(pm[qt], 16, 64));
        mn = fmaxf(mn, __shfl_xor(mn, 32, 64));
        const float alpha = __builtin_amdgcn_exp2f((mrow[qt] - mn) * c1);
        mrow[qt] = mn;
        lacc[qt] *= alpha;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt][qt] *= alpha;
      }
    }
    X8 pb[NQ];
#pragma unroll
    for (int qt = 0; qt < NQ; ++qt) {
      const float mc = mrow[qt] * c1;
      u32x4 w;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const f32x2 e2 = {__builtin_amdgcn_exp2f(__builtin_fmaf(sc[hf][qt][2 * h2], c1, -mc)),
                            __builtin_amdgcn_exp2f(__builtin_fmaf(sc[hf][qt][2 * h2 + 1], c1, -mc))};
          w[hf * 2 + h2] = pack2<T>(e2);
        }
      pb[qt] = __builtin_bit_cast(X8, w);
    }
    // ---- O^T += V^T P^T, row sums through the ones-row
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      X8 vt;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const X4 t4 = __builtin_bit_cast(X4, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (s16x4 __attribute__((address_space(3)))*)(vp + hf * (16 * 128) + vtr[dt])));
#pragma unroll
        for (int e = 0; e < 4; ++e) vt[hf * 4 + e] = t4[e];
      }
#pragma unroll
      for (int qt = 0; qt < NQ; ++qt) o[dt][qt] = mfma16(vt, pb[qt], o[dt][qt]);
    }
#pragma unroll
    for (int qt = 0; qt < NQ; ++qt) lacc[qt] = mfma16(ones, pb[qt], lacc[qt]);
  };

  auto run = [&](auto glo_) {
    for (int j = 0; j < nblk; ++j) {
      if (j + 1 < nblk) dma.template issue<BLK>(j + 1, (j + 1) & 1);      // its slot was released by the barrier that ended block j-1
      {
        // squared row norms of this V block (for the backward's histogram scale): one 16-byte chunk per thread
        const char* vb_ = Vl + (j & 1) * (BLK * 128);
        for (int i = tid; i < BLK * 8; i += nthr) {
          const X8 e = *(const X8*)(vb_ + i * 16);
          float s2 = 0.f;
#pragma unroll
          for (int k = 0; k < 8; ++k) s2 = __builtin_fmaf((float)e[k], (float)e[k], s2);
          s2 += __shfl_xor(s2, 1, 64); s2 += __shfl_xor(s2, 2, 64); s2 += __shfl_xor(s2, 4, 64);
          vmax = fmaxf(vmax, s2);
        }
      }
      if (active) {
#pragma unroll 1
        for (int st = SPB * j; st < min(SPB * j + SPB, nsteps); ++st) step(glo_, st, j & 1);
      }
      lds_dma_wait();                                      // this wave's LDS-DMA requests (vil_mfma_common.h)
      __syncthreads();
    }
  };
  if (glo_wave) run(std::true_type{}); else run(std::false_type{});
  // max_k |v_k|^2 of this (image, head) for the backward (every workgroup of the (image, head) streams all of V)
  {
#pragma unroll
    for (int o2 = 8; o2 < 64; o2 <<= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o2, 64));
    if (lane == 0) __hip_atomic_fetch_max(&misc[1], __float_as_uint(vmax), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
    if (tid == 0) p.lse[(int64_t)bh * (N + 1) + N] = __uint_as_float(misc[1]);
  }
  if (!active) return;

#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const float l = __shfl(lacc[qt][0], lj, 64);
    const float inv = 1.0f / l;
    if (qtok[qt] < N) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        X4 w;
#pragma unroll
        for (int r = 0; r < 4; ++r) w[r] = (T)(o[dt][qt][r] * inv);
        *(X4*)(ob + (int64_t)qtok[qt] * p.o_st + dt * 16 + lg * 4) = w;
      }
      if (lg == 0) p.lse[(int64_t)bh * (N + 1) + qtok[qt]] = mrow[qt] * p.scale + __logf(l);
    }
  }
}

// ===================================================================== backward: dQ (+ delta, + bias gradients)
template <typename T, int QT, bool HIST>
__global__ __launch_bounds__(QT == 1 ? 64 * DN_MAXW_Q : 64 * DN_MAXW_Q2, QT == 1 ? DN_OCC_Q : 3) void k_dense_bwd_dq(VilParams p, DenseCfg c) {
  typedef typename V16<T>::x8 X8;
  typedef typename V16<T>::x4 X4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, nthr = blockDim.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lj = lane & 15, lg = lane >> 4;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = (int)fdiv((unsigned)logical, c.m_nwg), wgi = logical - bh * c.nwg_bh;
  const int b = bh / p.H, h = bh - b * p.H;

  float* tab = (float*)smem;
  int* hist = (int*)(tab + c.tabsize);
  int* akey = hist + c.tabsize;
  float* ggl = (float*)(akey + c.NSP);
  unsigned* misc = (unsigned*)(ggl + DN_GGL);          // [0] max |dO_q|^2 over the workgroup's rows (float bits)
  char* Kl = (char*)(misc + 16);
  char* Vl = Kl + DN_RING;
  const unsigned tab_lds = lds_addr(smem);
  const unsigned hist_off = (unsigned)c.tabsize * 4u;

  const int N = c.N, G = c.G;
  const float c1 = p.scale * LOG2E;
  const float inv_s = 1.0f / p.scale;
  // the wave that owns the global-token query rows carries extra work per step (and every wave meets it at each block's
  // barrier): with global tokens its unit is ONE tile (tokens 0..15), all other units are QT tiles starting at token 16
  const int unit = wgi * c.wpw + wave;
  const bool split0 = G > 0 && QT > 1;
  const int q0 = split0 ? (unit == 0 ? 0 : 16 + (unit - 1) * (16 * QT)) : unit * (16 * QT);
  const T* qb = (const T*)p.q + b * p.q_sb + h * p.q_sh;
  const T* dob = (const T*)p.dout + b * p.do_sb + h * p.do_sh;
  const T* oub = (const T*)p.out + b * p.o_sb + h * p.o_sh;
  T* dqb = (T*)p.dq + b * p.dq_sb + h * p.dq_sh;
  const bool glo_wave = G > 0 && unit == 0;
  const bool active = q0 < N;

  DnDma dma;
  {
    const unsigned kbytes = (unsigned)(N - 1) * (unsigned)(p.k_st * 2) + 128u;
    const unsigned vbytes = (unsigned)(N - 1) * (unsigned)(p.v_st * 2) + 128u;
    dma.init(make_rsrc_n((const T*)p.k + b * p.k_sb + h * p.k_sh, kbytes), (int)p.k_st * 2, Kl,
             make_rsrc_n((const T*)p.v + b * p.v_sb + h * p.v_sh, vbytes), (int)p.v_st * 2, Vl, lane, wave, nthr >> 6);
  }
  const int nsteps = c.NSP >> 5, nblk = (nsteps + 1) >> 1;
  dma.issue(0, 0);
  int qtok[QT];
  X8 qf[2][QT], dof[2][QT], ouf[2][QT];
  float lsev[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    qtok[qt] = (split0 && unit == 0 && qt > 0) ? N : q0 + qt * 16 + lj;
    const bool real = qtok[qt] < N;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      X8 z = {};
      const int64_t d0 = ks * 32 + lg * 8;
      qf[ks][qt] = real ? *(const X8*)(qb + (int64_t)qtok[qt] * p.q_st + d0) : z;
      dof[ks][qt] = real ? *(const X8*)(dob + (int64_t)qtok[qt] * p.do_st + d0) : z;
      ouf[ks][qt] = real ? *(const X8*)(oub + (int64_t)qtok[qt] * p.o_st + d0) : z;
    }
    lsev[qt] = real ? p.lse[(int64_t)bh * (N + 1) + qtok[qt]] : 0.f;
  }
  const float vmax2 = p.lse[(int64_t)bh * (N + 1) + N];
  DN_PROLOGUE_LOADS
  if (tid < 16) misc[tid] = 0u;
  if (HIST)
    for (int i = tid; i < c.tabsize; i += nthr) hist[i] = 0;
  DN_PROLOGUE_WRITES(akey)

  // delta = rowsum(dO o O) of this wave's rows (kept, and written out for the dK/dV pass), |dO_q|^2 for the histogram scale
  float dlt[QT];
  float domax = 0.f;
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    float dl = 0.f, n2 = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        dl = __builtin_fmaf((float)dof[ks][qt][e], (float)ouf[ks][qt][e], dl);
        if (HIST) n2 = __builtin_fmaf((float)dof[ks][qt][e], (float)dof[ks][qt][e], n2);
      }
    dl += __shfl_xor(dl, 16, 64); dl += __shfl_xor(dl, 32, 64);
    dlt[qt] = dl;
    if (HIST) { n2 += __shfl_xor(n2, 16, 64); n2 += __shfl_xor(n2, 32, 64); domax = fmaxf(domax, n2); }
    if (qtok[qt] < N && lg == 0) c.delta[(int64_t)bh * N + qtok[qt]] = dl;
  }
  lds_dma_wait();                                      // this wave's LDS-DMA requests (vil_mfma_common.h)
  __syncthreads();                                     // misc / hist zeroed, tables written
  if (HIST) {
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) domax = fmaxf(domax, __shfl_xor(domax, o, 64));
    if (lane == 0) __hip_atomic_fetch_max(&misc[0], __float_as_uint(domax), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  // fixed-point scale of the bias-gradient histogram (vil_attn_mfma_bwd.hip): |dS| <= 2 |dO|max |v|max, a bin gets at most
  // one contribution per query row of this workgroup
  int lfx = 0;
  if (HIST) {
    const float bound = 2.0f * __builtin_sqrtf(__uint_as_float(misc[0]) * vmax2);
    if (bound > 0.f && bound < 1e30f) {
      lfx = 29 - (int)ceilf(__log2f(bound * (float)(c.wpw * 16 * QT)));
      lfx = max(-60, min(60, lfx));
    }
  }
  constexpr bool FOLD = HIST && !__is_same(T, _Float16);      // bf16: 2^lfx rides on P through lse; fp16 would overflow
  const float hscale = FOLD ? 1.0f : __builtin_amdgcn_exp2f((float)lfx);
  const float unscale = FOLD ? p.scale * __builtin_amdgcn_exp2f((float)-lfx) : p.scale;
  float lse2[QT], ndlt[QT];
  unsigned aqb[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const bool real = qtok[qt] < N;
    lse2[qt] = real ? lsev[qt] * LOG2E - (FOLD ? (float)lfx : 0.f) : DN_LSE_PAD;
    ndlt[qt] = real ? -dlt[qt] : 0.f;
    aqb[qt] = tab_lds + (unsigned)dn_A(c, qtok[qt]) * 4u;
  }

  int nat[2], ktr[4];                               // one layout for both rings: K and V share the natural offsets
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) nat[ks] = dn_off(lj, ks * 64 + lg * 16);
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) ktr[dt] = dn_off(lg * 4 + (lj >> 2), dt * 32 + (lj & 3) * 8);
  f32x4 dq[4][QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[dt][qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float gacc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};        // global query rows: sum of dS over the local keys, and per global key
  const unsigned dump = tab_lds + (unsigned)c.TS * 4u;   // a bin of the mask region

  auto step = [&](auto glo_, int st, int slot) {
    constexpr bool GLO = decltype(glo_)::value;
    constexpr int NQ = (GLO && QT > 1) ? 1 : QT;
    const char* kp = Kl + slot * (DN_BLK * 128) + (st & 1) * (32 * 128);     // (the V ring sits DN_RING bytes behind: an immediate)
    i32x4 ak[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) ak[hf] = *(const i32x4*)(akey + st * 32 + hf * 16 + lg * 4);
    u32x4 dsw[NQ];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      X8 kf[2], vf[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        kf[ks] = *(const X8*)(kp + nat[ks] + hf * (16 * 128));
        vf[ks] = *(const X8*)(kp + nat[ks] + hf * (16 * 128) + DN_RING);
      }
      f32x4 sacc[NQ], dpacc[NQ];
      unsigned i0[NQ][4];
#pragma unroll
      for (int qt = 0; qt < NQ; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          i0[qt][r] = aqb[qt] - (unsigned)ak[hf][r];
          sacc[qt][r] = *lds_f32(i0[qt][r]);
        }
      if (GLO) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int s = st * 32 + hf * 16 + lg * 4 + r;
          const float v = s < G ? ggl[lj * 4 + (s & 3)] : (s < N ? gl0 : VIL_MASK_VAL);
          if (lj < G) { sacc[0][r] = v; i0[0][r] = dump; }
        }
      }
#pragma unroll
      for (int qt = 0; qt < NQ; ++qt) {
        dpacc[qt] = (f32x4){ndlt[qt], ndlt[qt], ndlt[qt], ndlt[qt]};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          sacc[qt] = mfma16(kf[ks], qf[ks][qt], sacc[qt]);
          dpacc[qt] = mfma16(vf[ks], dof[ks][qt], dpacc[qt]);
        }
      }
#pragma unroll
      for (int qt = 0; qt < NQ; ++qt) {
        float ds[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[qt][r], c1, -lse2[qt]));
          ds[r] = pr * dpacc[qt][r];
          if (HIST)
            __hip_atomic_fetch_add(lds_i32(i0[qt][r] + hist_off), f2i_rpi(FOLD ? ds[r] : ds[r] * hscale),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (HIST && qt == 0 && GLO) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int s = st * 32 + hf * 16 + lg * 4 + r;
            if (st == 0 && hf == 0 && lg == 0 && r < G) gacc[1 + r] += ds[r];
            else if (s >= G && s < N) gacc[0] += ds[r];
          }
        }
        dsw[qt][hf * 2] = pack2<T>((f32x2){ds[0], ds[1]});
        dsw[qt][hf * 2 + 1] = pack2<T>((f32x2){ds[2], ds[3]});
      }
    }
    X8 dsb[NQ];
#pragma unroll
    for (int qt = 0; qt < NQ; ++qt) dsb[qt] = __builtin_bit_cast(X8, dsw[qt]);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      X8 kt8;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const X4 t4 = __builtin_bit_cast(X4, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (s16x4 __attribute__((address_space(3)))*)(kp + ktr[dt] + hf * (16 * 128))));
#pragma unroll
        for (int e = 0; e < 4; ++e) kt8[hf * 4 + e] = t4[e];
      }
#pragma unroll
      for (int qt = 0; qt < NQ; ++qt) dq[dt][qt] = mfma16(kt8, dsb[qt], dq[dt][qt]);
    }
  };

  auto run = [&](auto glo_) {
    for (int j = 0; j < nblk; ++j) {
      if (j + 1 < nblk) dma.issue(j + 1, (j + 1) & 1);
      if (active) {
#pragma unroll 1
        for (int st = 2 * j; st < min(2 * j + 2, nsteps); ++st) step(glo_, st, j & 1);
      }
      lds_dma_wait();                                      // this wave's LDS-DMA requests (vil_mfma_common.h)
      __syncthreads();
    }
  };
  if (glo_wave) run(std::true_type{}); else run(std::false_type{});
  if (active) {
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
      if (qtok[qt] < N) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          X4 w;
#pragma unroll
          for (int r = 0; r < 4; ++r) w[r] = (T)(dq[dt][qt][r] * unscale);
          *(X4*)(dqb + (int64_t)qtok[qt] * p.dq_st + dt * 16 + lg * 4) = w;
        }
      }
  }
  if (HIST) {
    // the workgroup's record: [TS table bins][lfx, -, G x int64 region sums (d g2l[1]), 20 floats of the global query rows]
    // (the loop's last barrier ordered every wave's atomics before these reads)
    int* rec = c.parts + (int64_t)logical * (c.rec_base + DN_REC_EXTRA);
    for (int i = tid; i < c.TS; i += nthr) rec[i] = hist[i];
    if (tid == 0) rec[c.rec_base] = lfx;
    if (wave == 0) {
      for (int g = 0; g < G; ++g) {
        long long sum = 0;
        const int* reg = hist + c.TS + (1 + g) * c.L;
        for (int e = lane; e < c.L; e += 64) sum += reg[e];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) sum += __shfl_xor(sum, o, 64);
        if (lane == 0) *(long long*)(rec + c.rec_base + 2 + 2 * g) = sum;
      }
      const float us = FOLD ? __builtin_amdgcn_exp2f((float)-lfx) : 1.0f;
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        float v = gacc[k];
        v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
        if (lg == 0 && lj < 4) ((float*)rec)[c.rec_base + 12 + lj * 5 + k] = (glo_wave && lj < G) ? v * us : 0.f;
      }
    }
  }
}

// ===================================================================== backward: dK, dV
template <typename T, int KT>
__global__ __launch_bounds__(64 * DN_MAXW_K, DN_OCC_K) void k_dense_bwd_dkdv(VilParams p, DenseCfg c) {
  typedef typename V16<T>::x8 X8;
  typedef typename V16<T>::x4 X4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, nthr = blockDim.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lj = lane & 15, lg = lane >> 4;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = (int)fdiv((unsigned)logical, c.m_nwg), wgi = logical - bh * c.nwg_bh;
  const int b = bh / p.H, h = bh - b * p.H;

  float* tab = (float*)smem;
  float* ggl = tab + c.tabsize;
  int* aqs = (int*)(ggl + DN_GGL);                    // [NSP] 4 * A(query slot)
  float* lses = (float*)(aqs + c.NSP);            // [NSP] lse * log2e (padding: +big)
  float* dlts = lses + c.NSP;                     // [NSP] -delta
  char* Ql = (char*)(dlts + c.NSP);
  char* Dl = Ql + DN_RING;
  const unsigned tab_lds = lds_addr(smem);

  const int N = c.N, G = c.G;
  const float c1 = p.scale * LOG2E;
  const float inv_s = 1.0f / p.scale;
  const int unit = wgi * c.wpw + wave;
  const int k0 = unit * (16 * KT);
  const T* kb = (const T*)p.k + b * p.k_sb + h * p.k_sh;
  const T* vb = (const T*)p.v + b * p.v_sb + h * p.v_sh;
  T* dkb = (T*)p.dk + b * p.dk_sb + h * p.dk_sh;
  T* dvb = (T*)p.dv + b * p.dv_sb + h * p.dv_sh;
  const bool active = k0 < N;

  DnDma dma;
  {
    const unsigned qbytes = (unsigned)(N - 1) * (unsigned)(p.q_st * 2) + 128u;
    const unsigned dbytes = (unsigned)(N - 1) * (unsigned)(p.do_st * 2) + 128u;
    dma.init(make_rsrc_n((const T*)p.q + b * p.q_sb + h * p.q_sh, qbytes), (int)p.q_st * 2, Ql,
             make_rsrc_n((const T*)p.dout + b * p.do_sb + h * p.do_sh, dbytes), (int)p.do_st * 2, Dl, lane, wave, nthr >> 6);
  }
  const int nsteps = c.NSP >> 5, nblk = (nsteps + 1) >> 1;
  dma.issue(0, 0);
  int ktok[KT];
  X8 kfb[2][KT], vfb[2][KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    ktok[kt] = k0 + kt * 16 + lj;
    const int t = ktok[kt];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      X8 z = {};
      kfb[ks][kt] = t < N ? *(const X8*)(kb + (int64_t)t * p.k_st + ks * 32 + lg * 8) : z;
      vfb[ks][kt] = t < N ? *(const X8*)(vb + (int64_t)t * p.v_st + ks * 32 + lg * 8) : z;
    }
  }
  float lv[2], dv_[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int s = tid + u * nthr;
    lv[u] = s < N ? p.lse[(int64_t)bh * (N + 1) + s] * LOG2E : DN_LSE_PAD;
    dv_[u] = s < N ? -c.delta[(int64_t)bh * N + s] : 0.f;
  }
  DN_PROLOGUE_LOADS
  DN_PROLOGUE_WRITES((int*)nullptr)
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int s = tid + u * nthr;
    if (s < c.NSP) { aqs[s] = dn_A(c, s) * 4; lses[s] = lv[u]; dlts[s] = dv_[u]; }
  }
  for (int s = tid + 2 * nthr; s < c.NSP; s += nthr) {
    aqs[s] = dn_A(c, s) * 4;
    lses[s] = s < N ? p.lse[(int64_t)bh * (N + 1) + s] * LOG2E : DN_LSE_PAD;
    dlts[s] = s < N ? -c.delta[(int64_t)bh * N + s] : 0.f;
  }

  unsigned akl[KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) akl[kt] = (unsigned)dn_akey(c, ktok[kt]) - tab_lds;
  int nat[2], tr[4];                                // Q and dO share the layout: the dO ring sits DN_RING bytes behind
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) nat[ks] = dn_off(lj, ks * 64 + lg * 16);
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) tr[dt] = dn_off(lg * 4 + (lj >> 2), dt * 32 + (lj & 3) * 8);
  f32x4 dk[4][KT], dv[4][KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      dk[dt][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      dv[dt][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  lds_dma_wait();                                      // this wave's LDS-DMA requests (vil_mfma_common.h)
  __syncthreads();

  auto step = [&](auto first_, int st, int slot) {
    constexpr bool FIRST = decltype(first_)::value;     // step 0 holds the global-token QUERY rows
    const char* sq = Ql + slot * (DN_BLK * 128) + (st & 1) * (32 * 128);
    u32x4 pbw[KT], dsw[KT];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int sb = st * 32 + hf * 16 + lg * 4;
      const i32x4 aq4 = *(const i32x4*)(aqs + sb);
      const f32x4 nd4 = *(const f32x4*)(dlts + sb);
      const f32x4 ls4 = *(const f32x4*)(lses + sb);
      X8 qa[2], da[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        qa[ks] = *(const X8*)(sq + nat[ks] + hf * (16 * 128));
        da[ks] = *(const X8*)(sq + nat[ks] + hf * (16 * 128) + DN_RING);
      }
      f32x4 sacc[KT], dpacc[KT];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) sacc[kt][r] = *lds_f32((unsigned)aq4[r] - akl[kt]);
      if (FIRST && hf == 0) {
        // the global-token QUERY rows are rows 0..G-1 of the first tile (lane group 0)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int t = ktok[kt];
            const float v = t < G ? ggl[r * 4 + (t & 3)] : (t < N ? ggl[64 + r] : VIL_MASK_VAL);
            if (lg == 0 && r < G) sacc[kt][r] = v;
          }
      }
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        dpacc[kt] = nd4;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          sacc[kt] = mfma16(qa[ks], kfb[ks][kt], sacc[kt]);
          dpacc[kt] = mfma16(da[ks], vfb[ks][kt], dpacc[kt]);
        }
      }
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        float pr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) pr[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[kt][r], c1, -ls4[r]));
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const f32x2 p2 = {pr[2 * h2], pr[2 * h2 + 1]};
          const f32x2 d2 = {dpacc[kt][2 * h2], dpacc[kt][2 * h2 + 1]};
          pbw[kt][hf * 2 + h2] = pack2<T>(p2);
          dsw[kt][hf * 2 + h2] = pack2<T>(p2 * d2);
        }
      }
    }
    X8 pb[KT], dsb[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) { pb[kt] = __builtin_bit_cast(X8, pbw[kt]); dsb[kt] = __builtin_bit_cast(X8, dsw[kt]); }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      X8 qt_, dt_;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const X4 tq = __builtin_bit_cast(X4, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (s16x4 __attribute__((address_space(3)))*)(sq + tr[dt] + hf * (16 * 128))));
        const X4 td = __builtin_bit_cast(X4, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (s16x4 __attribute__((address_space(3)))*)(sq + tr[dt] + hf * (16 * 128) + DN_RING)));
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

  for (int j = 0; j < nblk; ++j) {
    if (j + 1 < nblk) dma.issue(j + 1, (j + 1) & 1);
    if (active) {
      if (j == 0 && G > 0) { step(std::true_type{}, 0, 0); if (nsteps > 1) step(std::false_type{}, 1, 0); }
      else {
#pragma unroll 1
        for (int st = 2 * j; st < min(2 * j + 2, nsteps); ++st) step(std::false_type{}, st, j & 1);
      }
    }
    lds_dma_wait();                                      // this wave's LDS-DMA requests (vil_mfma_common.h)
    __syncthreads();
  }
  if (!active) return;
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
    if (ktok[kt] < N) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        X4 wk, wv;
#pragma unroll
        for (int r = 0; r < 4; ++r) { wk[r] = (T)(dk[dt][kt][r] * p.scale); wv[r] = (T)dv[dt][kt][r]; }
        *(X4*)(dkb + (int64_t)ktok[kt] * p.dk_st + dt * 16 + lg * 4) = wk;
        *(X4*)(dvb + (int64_t)ktok[kt] * p.dv_st + dt * 16 + lg * 4) = wv;
      }
    }
}

// ===================================================================== bias gradients from the dQ pass's records
// grid (nbx + 1, H), 1024 threads.  Blocks [0, nbx): 64 table bins x 16 record groups, every load of a thread independent
// of the others.  Last block: the per-record scalars (d g2l[1] region sums as int64, d g2l[0] / d g2g of the global query
// rows).  Every sum runs in a fixed order in double: bit-reproducible.
__global__ __launch_bounds__(1024) void k_dense_reduce(VilParams p, DenseCfg c, int nbx) {
  __shared__ double red[1024];
  const int h = blockIdx.y, bx = blockIdx.x, tid = threadIdx.x;
  const int stride = c.rec_base + DN_REC_EXTRA;
  const int nrec = p.B * c.nwg_bh;                           // records of head h: (b*H + h)*nwg_bh + w
  auto rec_of = [&](int j) { return c.parts + (int64_t)(((j / c.nwg_bh) * p.H + h) * c.nwg_bh + (j % c.nwg_bh)) * stride; };
  if (bx < nbx) {
    const int bin = bx * 64 + (tid & 63), grp = tid >> 6;
    double s = 0.0;
    if (bin < c.TS)
      for (int j0 = grp; j0 < nrec; j0 += 16 * 8) {
        int v[8], lf[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int j = j0 + 16 * u;
          const int* r = rec_of(min(j, nrec - 1));
          v[u] = j < nrec ? r[bin] : 0;
          lf[u] = r[c.rec_base];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += ldexp((double)v[u], -lf[u]);
      }
    red[tid] = s;
    __syncthreads();
    if (grp == 0 && bin < c.TS && p.dtable) {
      double t = 0.0;
      for (int g2 = 0; g2 < 16; ++g2) t += red[g2 * 64 + tid];
      p.dtable[(int64_t)bin * p.H + h] = (float)t;
    }
  } else {
    // value k of a record: [0, 4) region sums (int64, scaled), [4, 24) the 20 floats of the global query rows
    const int k = tid & 31, grp = tid >> 5;
    double s = 0.0;
    if (k < 4) {
      if (k < c.G)
        for (int j = grp; j < nrec; j += 32) {
          const int* r = rec_of(j);
          s += ldexp((double)*(const long long*)(r + c.rec_base + 2 + 2 * k), -r[c.rec_base]);
        }
    } else if (k < 24) {
      for (int b = grp; b < p.B; b += 32) s += (double)((const float*)rec_of(b * c.nwg_bh))[c.rec_base + 12 + k - 4];
    }
    red[tid] = s;
    __syncthreads();
    if (tid < 24) {
      double t = 0.0;
      for (int g2 = 0; g2 < 32; ++g2) t += red[g2 * 32 + tid];
      if (tid < 4) { if (tid < c.G && p.dg2l) p.dg2l[h * c.G + tid] = (float)t; }
      else {
        const int g = (tid - 4) / 5, kk = (tid - 4) % 5;
        if (g < c.G) {
          if (kk == 0) { if (p.dg2l0) p.dg2l0[h * c.G + g] = (float)t; }
          else if (kk - 1 < c.G && p.dg2g) p.dg2g[(h * c.G + g) * c.G + kk - 1] = (float)t;
        }
      }
    }
  }
}

// ===================================================================== host side
static bool dense_cfg(const VilAttnDesc* d, int rows_per_wave, int maxw, DenseCfg& c, bool query_side = false) {
  memset(&c, 0, sizeof(c));
  c.nx = d->nx; c.ny = d->ny; c.G = d->G;
  c.N = d->G + d->nx * d->ny;
  c.NSP = (c.N + 31) & ~31;
  c.P = 2 * d->ny - 1;
  c.TS = (2 * d->nx - 1) * c.P;
  c.L = (c.TS + 1) / 2;
  c.tabsize = ((c.TS + (1 + d->G) * c.L + 3) / 4) * 4;
  c.unit = rows_per_wave;
  c.nunits = (c.N + rows_per_wave - 1) / rows_per_wave;
  if (query_side && d->G > 0 && rows_per_wave > 16)          // unit 0 = tokens 0..15 (k_dense_fwd / _bwd_dq: split0)
    c.nunits = 1 + (c.N > 16 ? (c.N - 16 + rows_per_wave - 1) / rows_per_wave : 0);
  c.nwg_bh = (c.nunits + maxw - 1) / maxw;
  c.wpw = (c.nunits + c.nwg_bh - 1) / c.nwg_bh;
  c.rec_base = (c.TS + 1) & ~1;
  c.m_ny = vil_magic((unsigned)d->ny);
  c.m_nwg = vil_magic((unsigned)c.nwg_bh);
  return true;
}
static size_t dense_lds(const DenseCfg& c, int pass, int blk = DN_BLK) {   // 0 forward, 1 dQ, 2 dK/dV: see the kernels' LDS maps
  if (pass == 0) return (size_t)c.tabsize * 4 + (size_t)c.NSP * 4 + DN_GGL * 4 + 64 + 2 * (size_t)(2 * blk * 128);
  if (pass == 1) return (size_t)c.tabsize * 8 + (size_t)c.NSP * 4 + DN_GGL * 4 + 64 + 2 * DN_RING;
  return (size_t)c.tabsize * 4 + DN_GGL * 4 + 3 * (size_t)c.NSP * 4 + 2 * DN_RING;
}
#ifndef VIL_DENSE_QT
#define VIL_DENSE_QT 2
#endif
static inline int dense_dq_tiles(int N) { return (N > 96 && N <= 384) ? 2 : 1; }
#ifndef VIL_DENSE_KT
#define VIL_DENSE_KT 1
#endif

extern "C" int vil_dense_attn_supported(const VilAttnDesc* d) {
  if (!d) return VIL_E_NULL;
  if (d->B <= 0 || d->H <= 0 || d->nx <= 0 || d->ny <= 0 || d->G < 0) return VIL_E_SHAPE;
  if (d->dtype != VIL_DTYPE_BF16 && d->dtype != VIL_DTYPE_F16) return VIL_E_DTYPE;
  if (d->M != 64) return VIL_E_HEAD_DIM;
  if (d->G > 4) return VIL_E_BACKEND;
  if ((d->q_st | d->k_st | d->v_st | d->q_sb | d->k_sb | d->v_sb | d->q_sh | d->k_sh | d->v_sh) & 7) return VIL_E_ALIGN;
  if ((d->o_st | d->o_sb | d->o_sh) & 3) return VIL_E_ALIGN;
  const int64_t ntok = (int64_t)d->G + (int64_t)d->nx * d->ny;
  const int64_t smax = d->q_st > d->k_st ? (d->q_st > d->v_st ? d->q_st : d->v_st) : (d->k_st > d->v_st ? d->k_st : d->v_st);
  if (smax * 2 * ntok >= (1ll << 31)) return VIL_E_BACKEND;
  DenseCfg c;
  dense_cfg(d, 16, 1, c);
  for (int pass = 0; pass < 3; ++pass)
    if (dense_lds(c, pass) > 160 * 1024) return VIL_E_BACKEND;
  if ((uint64_t)d->B * d->H * c.nwg_bh * (uint64_t)c.nwg_bh >= (1ull << 32)) return VIL_E_BACKEND;
  return VIL_OK;
}

extern "C" size_t vil_dense_attn_workspace_bytes(const VilAttnDesc* d, int pass) {
  if (vil_dense_attn_supported(d) != VIL_OK || pass == 0) return 0;
  DenseCfg c;
  const int dqt = dense_dq_tiles(d->G + d->nx * d->ny);
  dense_cfg(d, 16 * dqt, dqt == 1 ? DN_MAXW_Q : DN_MAXW_Q2, c, true);
  const size_t delta = (((size_t)d->B * d->H * c.N + 3) & ~(size_t)3) * 4;
  return delta + (size_t)d->B * d->H * c.nwg_bh * (c.rec_base + DN_REC_EXTRA) * 4;
}

static void dense_params(VilParams& p, const VilAttnDesc* d, const float* table, const float* g2l, const float* g2g) {
  memset(&p, 0, sizeof(p));
  vil_fill_params(p, d);
  p.table = table; p.has_bias = table != nullptr;
  p.has_g2l = g2l != nullptr && d->G > 0;
  p.g2l = p.has_g2l ? g2l + (size_t)d->H * d->G : nullptr;     // [1]: local query -> global key
  p.g2l0 = p.has_g2l ? g2l : nullptr;                           // [0]: global query -> local key
  p.g2g = d->G > 0 ? g2g : nullptr;
}

#define DN_LAUNCH_T(KERNEL, T, grid, wpw_, lds, s, ...)                                                  \
  {                                                                                                      \
    if (int he = vil_ensure_dyn_lds((const void*)KERNEL(T), lds)) return he;                             \
    KERNEL(T)<<<dim3(grid), dim3(64 * (wpw_)), lds, s>>>(__VA_ARGS__);                                    \
  }
#define DN_LAUNCH(KERNEL, grid, wpw_, lds, s, ...)                                                       \
  {                                                                                                      \
    if (d->dtype == VIL_DTYPE_F16) DN_LAUNCH_T(KERNEL, _Float16, grid, wpw_, lds, s, __VA_ARGS__)         \
    else DN_LAUNCH_T(KERNEL, __bf16, grid, wpw_, lds, s, __VA_ARGS__)                                     \
  }
#define DN_K_FWD(T) k_dense_fwd<T, VIL_DENSE_QT, 64>
#define DN_K_FWD32(T) k_dense_fwd<T, VIL_DENSE_QT, 32>
#define DN_K_DQ1_H(T) k_dense_bwd_dq<T, 1, true>
#define DN_K_DQ1_N(T) k_dense_bwd_dq<T, 1, false>
#define DN_K_DQ2_H(T) k_dense_bwd_dq<T, 2, true>
#define DN_K_DQ2_N(T) k_dense_bwd_dq<T, 2, false>
#define DN_K_DKDV(T) k_dense_bwd_dkdv<T, VIL_DENSE_KT>

// launch shape of the forward: -1 the rule below, 0 always the wide shape, 1 always the narrow one (tests, measurements)
static int g_dense_fwd_shape = -1;
extern "C" int vil_dense_attn_set_fwd_shape(int mode) {
  if (mode < -1 || mode > 1) return VIL_E_SHAPE;
  g_dense_fwd_shape = mode;
  return VIL_OK;
}

extern "C" int vil_dense_attn_fwd(const VilAttnDesc* d, const void* q, const void* k, const void* v,
                                  const float* bias_table, const float* g2l, const float* g2g,
                                  void* out, float* lse, void* stream) {
  int e = vil_dense_attn_supported(d);
  if (e) return e;
  if (!q || !k || !v || !out || !lse) return VIL_E_NULL;
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) return VIL_E_ALIGN;
  if ((uintptr_t)out & 7) return VIL_E_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  VilParams p;
  dense_params(p, d, bias_table, g2l, g2g);
  p.q = q; p.k = k; p.v = v; p.o = out; p.lse = lse;
  DenseCfg c;
  dense_cfg(d, 16 * VIL_DENSE_QT, DN_MAXW_F, c, true);
  vil_prof_tag(d->B, d->H, d->M, d->nx, d->ny, d->nx > d->ny ? d->nx : d->ny, d->G, -1);
  const double n = c.N, ce = (double)d->H * d->M * 2;
  vil_prof_begin(VIL_K_DENSE_FWD, s, d->B * (4 * n * ce + 4.0 * d->H * n), d->B * 4.0 * n * n * d->H * d->M);
  // Two launch shapes: workgroups of up to DN_MAXW_F waves on the 64-row ring, or of up to 4 waves on a 32-row ring.  The
  // kernel runs at 120 registers = 16 waves per CU: two 7-wave workgroups, or four 4-wave ones when their LDS allows
  // (36.7 instead of 52.7 KB at 24 x 24).  Every workgroup streams all of K / V, so the narrow shape (more workgroups per
  // (image, head)) is taken only where it needs fewer ROUNDS of workgroups: 24 x 24, B H = 192: 960 workgroups on 1024
  // slots instead of 576 on 512 -- the second round of 64 cost 0.64 of the first.
  DenseCfg cn;
  dense_cfg(d, 16 * VIL_DENSE_QT, 4, cn, true);
  auto rounds = [&](const DenseCfg& cc, size_t lds_) {
    int per_cu = (int)((155 * 1024) / lds_);
    if (per_cu > 16 / cc.wpw) per_cu = 16 / cc.wpw;
    if (per_cu < 1) per_cu = 1;
    const int64_t slots = (int64_t)per_cu * vil_cu_count(), wgs = (int64_t)d->B * d->H * cc.nwg_bh;
    return (wgs + slots - 1) / slots;
  };
  const size_t lds64 = dense_lds(c, 0, 64), lds32 = dense_lds(cn, 0, 32);
  const bool narrow = g_dense_fwd_shape < 0 ? rounds(cn, lds32) < rounds(c, lds64) : g_dense_fwd_shape == 1;
  if (narrow) {
    DN_LAUNCH(DN_K_FWD32, (unsigned)(d->B * d->H * cn.nwg_bh), cn.wpw, lds32, s, p, cn);
  } else {
    DN_LAUNCH(DN_K_FWD, (unsigned)(d->B * d->H * c.nwg_bh), c.wpw, lds64, s, p, c);
  }
  vil_prof_end(s);
  return (int)hipGetLastError();
}

extern "C" int vil_dense_attn_bwd(const VilAttnDesc* d, const void* q, const void* k, const void* v,
                                  const void* out, const void* dout, const float* lse,
                                  const float* bias_table, const float* g2l, const float* g2g,
                                  void* dq, void* dk, void* dv, float* dbias_table, float* dg2l, float* dg2g,
                                  void* workspace, void* stream) {
  int e = vil_dense_attn_supported(d);
  if (e) return e;
  if (!q || !k || !v || !out || !dout || !lse || !dq || !dk || !dv) return VIL_E_NULL;
  if (bias_table && !dbias_table) return VIL_E_NULL;
  if (g2l && d->G > 0 && !dg2l) return VIL_E_NULL;
  if (g2g && d->G > 0 && !dg2g) return VIL_E_NULL;
  if (!workspace) return VIL_E_WORKSPACE;
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out | (uintptr_t)dout | (uintptr_t)workspace) & 15) return VIL_E_ALIGN;
  if (((uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 7) return VIL_E_ALIGN;
  if ((d->do_st | d->do_sb | d->do_sh | d->o_st | d->o_sb | d->o_sh) & 7) return VIL_E_ALIGN;
  if ((d->dq_st | d->dq_sb | d->dq_sh | d->dk_st | d->dk_sb | d->dk_sh | d->dv_st | d->dv_sb | d->dv_sh) & 3) return VIL_E_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  VilParams p;
  dense_params(p, d, bias_table, g2l, g2g);
  p.q = q; p.k = k; p.v = v; p.out = out; p.dout = dout; p.lse = (float*)lse;
  p.dq = dq; p.dk = dk; p.dv = dv;
  const bool hg = p.has_g2l;
  p.dtable = dbias_table;
  p.dg2l = hg ? dg2l + (size_t)d->H * d->G : nullptr;
  p.dg2l0 = hg ? dg2l : nullptr;
  p.dg2g = (d->G > 0 && g2g) ? dg2g : nullptr;
  DenseCfg cq, ck;
  const int dqt = dense_dq_tiles(d->G + d->nx * d->ny);
  dense_cfg(d, 16 * dqt, dqt == 1 ? DN_MAXW_Q : DN_MAXW_Q2, cq, true);
  dense_cfg(d, 16 * VIL_DENSE_KT, DN_MAXW_K, ck);
  cq.delta = ck.delta = (float*)workspace;
  cq.parts = (int*)((char*)workspace + (((size_t)d->B * d->H * cq.N + 3) & ~(size_t)3) * 4);
  cq.do_hist = (p.dtable || p.dg2l || p.dg2g) ? 1 : 0;
  vil_prof_tag(d->B, d->H, d->M, d->nx, d->ny, d->nx > d->ny ? d->nx : d->ny, d->G, -1);
  const double n = cq.N, ce = (double)d->H * d->M * 2;
  vil_prof_begin(VIL_K_DENSE_DQ, s, d->B * (6 * n * ce + 8.0 * d->H * n), d->B * 6.0 * n * n * d->H * d->M);
  const unsigned gq = (unsigned)(d->B * d->H * cq.nwg_bh);
  if (dqt == 1) {
    if (cq.do_hist) DN_LAUNCH(DN_K_DQ1_H, gq, cq.wpw, dense_lds(cq, 1), s, p, cq)
    else DN_LAUNCH(DN_K_DQ1_N, gq, cq.wpw, dense_lds(cq, 1), s, p, cq)
  } else {
    if (cq.do_hist) DN_LAUNCH(DN_K_DQ2_H, gq, cq.wpw, dense_lds(cq, 1), s, p, cq)
    else DN_LAUNCH(DN_K_DQ2_N, gq, cq.wpw, dense_lds(cq, 1), s, p, cq)
  }
  vil_prof_end(s);
  if ((e = (int)hipGetLastError())) return e;
  vil_prof_begin(VIL_K_DENSE_DKDV, s, d->B * (6 * n * ce + 8.0 * d->H * n), d->B * 8.0 * n * n * d->H * d->M);
  DN_LAUNCH(DN_K_DKDV, (unsigned)(d->B * d->H * ck.nwg_bh), ck.wpw, dense_lds(ck, 2), s, p, ck);
  vil_prof_end(s);
  if ((e = (int)hipGetLastError())) return e;
  if (cq.do_hist) {
    const int nbx = p.dtable ? (cq.TS + 63) / 64 : 0;
    vil_prof_begin(VIL_K_DENSE_REDUCE, s, 0, 0);
    k_dense_reduce<<<dim3((unsigned)(nbx + (d->G > 0 ? 1 : 0)), (unsigned)d->H), dim3(1024), 0, s>>>(p, cq, nbx);
    vil_prof_end(s);
    e = (int)hipGetLastError();
  }
  return e;
}
