// vil_attn_cw.hip -- the chunk-workgroup ("cw") family of the sliding-chunk attention for gfx950 / CDNA4: 16-bit I/O,
// fp32 accumulate, head_dim 32 / 64.  Round 6: the narrow-unit decomposition DESIGN.md section 8.1 asked for.
//
// What is different from the wave-per-chunk family (vil_attn_mfma*.hip), which stays in the library as the fallback and as
// the ablation row (desc.backend = VIL_BACKEND_MFMA_WAVE):
//   * one wave owns 32 query slots (two 16-column tiles: column (x, yp), tile qt <-> query (x, 2 yp + qt)), NOT 64: a
//     chunk is NWP = ceil(W ceil(W/2) / 16) waves (two at W = 7, 8), per-wave state is ~half, 4-6 waves per SIMD instead
//     of 2-3;
//   * the chunk's K and V step tiles are brought into LDS ONCE per chunk by LDS-DMA (buffer_load ... lds: no registers,
//     no ds_write pass), into a two-slot ring, and read by all of the chunk's waves -- the wave-per-chunk family loaded K
//     fragments per wave straight from L2 and staged V through registers;
//   * a workgroup is a COLUMN: NCH chunks of one head walking their key steps in lockstep (one bare s_barrier per 32-key
//     step; chunks are grouped by rank of work so that the steps of a workgroup's chunks are equal in number), bound to the
//     XCD it is dispatched to (blockIdx % 8) and PERSISTENT over that XCD's images: the head's bias image and the chunk
//     positions' slot tables are brought into LDS once per workgroup, not once per (image, chunk);
//   * slot tables are per chunk POSITION (k_cw_prep): DMA row offsets (int32) and the bias address terms as biased 16-bit
//     values in the order a lane reads them; the request cursor runs one tile ahead of the compute and reads them from LDS;
//   * bf16: the softmax is shift-free.  Q is pre-multiplied by scale * log2(e) (one bf16 rounding, like the reference's own
//     `q * scale` under autocast), the bias image is in log2 units, so the MFMA result IS the exponent: p = exp2(s), no
//     running maximum, no subtract, no rescale branch -- 2 VALU instructions per score (v_exp, half a v_cvt_pk) instead
//     of 3.75.  Images with a row sum outside [2^-24, 2^24] (logits beyond ~16 nats, where the rounding of Q' would show)
//     are computed again by the exact kernel launched behind the fast one (same source, SAFE instantiation: unscaled Q,
//     deferred running maximum; its workgroups return at once when their twin flagged nothing; fp16 runs only this one).
// Orientation, operand layouts and the LDS bias image are the wave-per-chunk family's (vil_mfma_common.h): S^T tiles,
// bias as the accumulator's initial value gathered with ds_read_b32 at Aq - Ak, P^T as the next MFMA's B operand, V^T
// through ds_read_b64_tr_b16, row sums from a ones-row MFMA.
//
// Reference semantics: src/models/layers/longformer2d.py:134-204, slidingchunk_2d.py:26-130 (see include/vil_attn.h).
#include "vil_mfma_common.h"
#include <type_traits>
#include <queue>
#include <vector>

#ifndef VIL_CW_OCC32
#define VIL_CW_OCC32 5       // waves per SIMD the head_dim 32 forward is held to (96 VGPRs)
#endif
#ifndef VIL_CW_OCC64
#define VIL_CW_OCC64 3
#endif
#ifndef VIL_CW_DMAX
#define VIL_CW_DMAX 4        // deepest K / V ring
#endif
#ifndef VIL_CW_NCH
#define VIL_CW_NCH 2         // chunks per workgroup (lockstep)
#endif
#ifndef VIL_CW_LATE_V
#define VIL_CW_LATE_V 0          // wait for a step's V tile only before P V (second barrier per step): measured alternative
#endif
#ifndef VIL_CW_CLASS_STREAMS
#define VIL_CW_CLASS_STREAMS 0   // per-class image stream counts (cw_plan_streams): measured, off
#endif
#ifndef VIL_CW_PAIR_ORDER
#define VIL_CW_PAIR_ORDER 1  // a workgroup's two chunks list their shared neighbours at the same positions (k_cw_prep)
#endif
// timing ablations (tools only; the results are wrong): built with -DVIL_CW_ABLATE, selected at run time through
// vil_attn_cw_set_ablation (CwCfg::abl) -- 1 no wait + barrier in the loop, 2 no LDS-DMA in the loop, 4 no bias
// gather, 8 no exponentials, 16 no P V (transposed reads + MFMAs), 32 no K reads / S MFMAs, 64 no step at all
#ifdef VIL_CW_ABLATE
#define CW_ABL(bit) (w.abl & (bit))
#else
#define CW_ABL(bit) 0
#endif
#ifdef VIL_CW_ABLATE
// cycle stamps (ablation builds, abl bit 256): segment sums per wave -> w.dbg[(block * 8 + wave) * 16 + segment]
#define CW_STAMP(k) if (w.abl & 256) { const unsigned long long t_ = __builtin_readcyclecounter(); tacc[k] += t_ - tlast; tlast = t_; }
#else
#define CW_STAMP(k)
#endif
constexpr int cw_occ(int MD, int QT) { return QT == 1 ? (MD <= 2 ? 8 : 5) : (MD <= 2 ? VIL_CW_OCC32 : VIL_CW_OCC64); }

// ------------------------------------------------------------------ prologue launch: tables
// roles: [0, ntx * H) bias images in log2 units (table_element, inv = log2 e), then one workgroup (its first wave) per
// chunk position for the slot tables: cw.koff / cw.akey (NSP per chunk), cw.nslots, cw.nown
__global__ __launch_bounds__(256) void k_cw_prep(VilParams p, MfmaCfg c, CwCfg w, int row_stride_b, int ntx) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int blk = blockIdx.x, ntab = ntx * p.H;
  if (blk < ntab) {
    const int h = blk / ntx, bx = blk - h * ntx;
    table_element(p, c, (float*)c.tabws, h, bx * 256 + threadIdx.x, LOG2E);
    return;
  }
  if (threadIdx.x >= 64) return;
  const int rank = blk - ntab, lane = threadIdx.x;
  const VilGeom& g = p.g;
  const int ch = w.lpt ? chunk_of_rank(rank, g.mx, g.my) : rank;
  const int cn = ch % g.my, cm = ch / g.my;
  int* s_koff = (int*)smem;
  int* s_akey = s_koff + c.NSP;
  int adr1, adc1;
  shift_neighbour(p, adr1, adc1);
  // Order of the neighbours in the list: the own chunk first (the global query column is live there).  With two chunks per
  // workgroup the pair is usually adjacent and shares six of its nine neighbours: those go to the SAME list positions in
  // both chunks -- [own, partner | the four other shared ones | the three private ones] -- so that the two chunks' waves,
  // which walk their lists in lockstep, request the same K / V rows in the same step and the second request finds the
  // lines in the CU's L1 instead of crossing the L2 -> CU path again (VIL_CW_PAIR_ORDER=0: plain own-first order).
  unsigned long long perm = 0ull;
#if VIL_CW_PAIR_ORDER
  if (g.nact == 9 && (rank ^ 1) < w.nch) {      // (whatever NCH is: the result must not depend on the launch shape)
    const int pch = w.lpt ? chunk_of_rank(rank ^ 1, g.mx, g.my) : (rank ^ 1);
    const int dr = pch / g.my - cm, dc = pch % g.my - cn;
    if (dr == 0 && dc == 1) perm = 0x630872154ull;        // positions 0..8 = 4,5,1,2,7,8,0,3,6
    else if (dr == 0 && dc == -1) perm = 0x852761034ull;  // 4,3,0,1,6,7,2,5,8
    else if (dr == 1 && dc == 0) perm = 0x210865374ull;   // 4,7,3,5,6,8,0,1,2
    else if (dr == -1 && dc == 0) perm = 0x876532014ull;  // 4,1,0,2,3,5,6,7,8
  }
#endif
  const int nslots = build_key_slots(p, c, cm, cn, lane, row_stride_b, s_koff, s_akey, adr1, adc1, true, false, perm);
  // akey: 16 bits per slot, biased by w.akb so that every term is non-negative, in the order a lane reads them: the
  // eight keys of lane group lg in step st (tile hf, row r) are the contiguous halfwords st * 32 + lg * 8 + hf * 4 + r
  unsigned short* ak16 = (unsigned short*)w.akey + (int64_t)ch * c.NSP;
  for (int s = lane; s < c.NSP; s += 64) {
    // (entries beyond nslots are never used for compute; they hold a real row so that a prefetch of them is harmless)
    w.koff[(int64_t)ch * c.NSP + s] = s < nslots ? s_koff[s] : 0;
    const int st = s >> 5, hf = (s >> 4) & 1, lg = (s >> 2) & 3, r = s & 3;
    ak16[st * 32 + lg * 8 + hf * 4 + r] = (unsigned short)((s < nslots ? s_akey[s] : -c.guard0 * 4) + w.akb);
  }
  if (lane == 0) {
    w.nslots[ch] = nslots;
    const int rows = min(g.W, g.nx - cm * g.W), cols = min(g.W, g.ny - cn * g.W);
    w.nown[ch] = p.only_glo ? 0 : rows * cols;
  }
}

// one LDS-DMA request: 64 lanes x 16 bytes from rs[voff(lane)] to the 1 KB at dst (lane-linear).  (A __device__ function
// on purpose: with the builtin inside a lambda of the kernel, the HOST pass silently drops the kernel's stub.)
__device__ __forceinline__ void cw_dma16(__amdgpu_buffer_rsrc_t rs, char* dst, int voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, voff, 0, 0, 0);
}

// s_waitcnt vmcnt(n) lgkmcnt(0) for a wave-uniform n, as the BUILTIN with an immediate (so that the compiler's own
// wait-count bookkeeping sees it: behind an opaque asm wait it would wait again, with vmcnt(0), at the next use of
// anything loaded -- i.e. for the tiles just requested).  Counts the table does not hold wait for everything.
#define CW_WAITCNT(N) __builtin_amdgcn_s_waitcnt(((N) & 15) | (((N) >> 4) << 14) | 0x0070)
__device__ __forceinline__ void cw_wait_vm(int n) {
  switch (n) {
    case 1: CW_WAITCNT(1); break;
    case 3: CW_WAITCNT(3); break;
    case 2: CW_WAITCNT(2); break;
    case 4: CW_WAITCNT(4); break;
    case 6: CW_WAITCNT(6); break;
    case 8: CW_WAITCNT(8); break;
    case 12: CW_WAITCNT(12); break;
    case 16: CW_WAITCNT(16); break;
    default: CW_WAITCNT(0); break;
  }
}

// ------------------------------------------------------------------ forward
// A workgroup is bound to one COLUMN -- a head and NCH query chunks of equal rank of work, NCH * NWP waves walking their
// key steps in lockstep -- and to the XCD the hardware places it on (blockIdx % 8); it walks the images of its stream
// (b0, b0 + bstride, ...) one after the other.  What a column keeps for its whole life: the head's bias image and the
// chunk positions' slot tables in LDS.  What streams: the chunks' K / V step tiles, through two-slot LDS rings that run
// ACROSS image boundaries, and the next image's Q rows (plain loads, requested one step before the boundary).  All
// workgroups of an XCD walk the same images in the same order, so the chunks that share a K / V row meet it in that
// XCD's L2 at about the same time.
//
// History of this kernel (ViL-Small stage 1, hipEvents, wave-per-chunk kernel 208 - 224 us on the same boxes):
//   one workgroup per (image, head, chunk pair), 2-slot ring, 20 waves per CU                 192 us  <- the step kept below
//   ... 3 / 4-slot ring (16 / 12 waves per CU: the ring's LDS costs resident waves)           226 / 266 us
//   persistent columns + software-pipelined step (scores of tile t + 1 under the softmax
//   of tile t; 168 VGPRs, 12 waves per CU), Q rows by LDS-DMA, runtime ring slot              244 - 253 us
//   the same with one 64-query wave per chunk (no barrier, 256 VGPRs + scratch)                565 us
// Timing ablations of the first version: no compute at all (requests + barriers) 214 us; compute without requests and
// barriers 189 us; nothing in the loop 96 us.  Counters: no pipe more than 53 % busy (LDS, a third of it bank conflicts),
// a wave-step of ~90 instructions takes ~3 000 cycles whatever the structure; occupancy is the only lever that moved the
// time, and LDS (bias image per head + ring per chunk) is what limits it.  tools/ubench/dma_rate.hip: K / V rows of one
// head at head_dim 32 are HALF cache lines -- the L2 -> CU path saturates at 16.5 TB/s of such rows (27 B/clk/CU) against
// 29 TB/s for whole lines, 3.6 against 7.7 TB/s when they miss L2.
#define CW_D 2
// SAFE = false (bf16): the shift-free softmax; the workgroup leaves the mask of images whose row sums left the safe range in
// w.redo[blockIdx.x].  SAFE = true: the exact classic form -- unscaled Q, p = exp2((s - m) scale log2 e) with a deferred
// running maximum, the wave-per-chunk kernels' arithmetic -- over all images (fp16) or, launched behind the fast kernel with
// redo_only, over the images its twin workgroup flagged (usually none: the workgroup returns at once).
template <typename T, int MD, int QT, bool SAFE>
__global__ __launch_bounds__(512, cw_occ(MD, QT)) void k_cw_fwd(VilParams p, MfmaCfg c, CwCfg w, int redo_only) {
  typedef typename V16<T>::x8 X8;
  typedef typename V16<T>::x4 X4;
  constexpr int D = CW_D;
  constexpr int M = 16 * MD;
  constexpr int MK = MD / 2;                  // 32-wide K steps over the head dim (MD is 2 or 4)
  constexpr int ROWB = M * 2;                 // bytes of a K / V / Q row
  constexpr int TILE = 32 * ROWB;             // one matrix's step tile
  constexpr int SLOTB = 2 * TILE;             // a ring slot: K tile, V tile
  constexpr int LPR = M / 8;                  // lanes (16-byte pieces) per row
  constexpr int RPP = 64 / LPR;               // rows per 1 KB DMA piece
  constexpr int PPM = 32 / RPP;               // pieces per matrix and step: a wave requests the K and the V piece of row
                                              // group sub = wp, wp + NWP, ... (one row offset serves both)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const VilGeom& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lj = lane & 15, lg = lane >> 4;
  const int W = g.W;
  // wave -> (slot of the workgroup = (chunk slot, head slot), query part of the chunk).  The HPW heads of a chunk sit in
  // one workgroup where a K / V row is half a cache line (head_dim 32): their requests for the two halves of a line are
  // issued in the same step and the CU's L1 merges them (tools/ubench/dma_rate.hip: 23 - 27 TB/s instead of 16.7).
  const int slot_ = fdiv(wave, w.m_NWP), wp = wave - slot_ * w.NWP;
  const int cslot = fdiv(slot_, w.m_HPW), hslot = slot_ - cslot * w.HPW;

  // ---- the column and its image stream
  const int xcd = blockIdx.x & 7, kblk = blockIdx.x >> 3;
  int h, grp, b0, bstride, nimg;
  if (w.by_image) {
    // the XCD's list: segments of chunk groups of equal work (cw_make_cfg), each stream-major, then chunk-group-major,
    // head-group-minor: the workgroups of a chunk group's heads are dispatched back to back
    int sc = 0;
    if (w.nseg > 1 && kblk >= w.seg_wg0[1]) sc = 1;
    if (w.nseg > 2 && kblk >= w.seg_wg0[2]) sc = 2;
    if (w.nseg > 3 && kblk >= w.seg_wg0[3]) sc = 3;
    const int idx = kblk - w.seg_wg0[sc], ncol_c = w.seg_ng[sc] * w.NHG;
    const int strm = fdiv(idx, w.seg_m[sc]), col = idx - strm * ncol_c;
    const int gl = fdiv(col, w.m_NHG);
    grp = w.seg_g0[sc] + gl; h = (col - gl * w.NHG) * w.HPW + hslot;
    b0 = xcd + 8 * strm; bstride = 8 * w.seg_ns[sc];
    nimg = b0 < p.B ? (p.B - b0 + bstride - 1) / bstride : 0;
  } else {
    // (image, head group) pairs dealt to the XCDs
    const int strm = fdiv(kblk, w.m_ncolx), col = kblk - strm * w.ncolx;
    (void)strm;
    const int pj = fdiv(col, w.m_ngrp); grp = col - pj * w.ngrp;
    const int pi = xcd + 8 * pj;
    b0 = pi / w.NHG; h = (pi - b0 * w.NHG) * w.HPW + hslot; bstride = 0;
    nimg = pi < p.B * w.NHG ? 1 : 0;
  }
  if (nimg == 0) return;
  unsigned redo_mask = 0u;
  if (SAFE && redo_only) {
    redo_mask = __builtin_amdgcn_readfirstlane(w.redo[blockIdx.x]);
    if (redo_mask == 0u) return;
  }
  const int rank = grp * w.NCH + cslot;
  const bool active = rank < w.nch;
  const int ch = active ? (w.lpt ? chunk_of_rank(rank, g.mx, g.my) : rank) : 0;
  const int cm = fdiv(ch, c.m_my), cn = ch - cm * g.my;
  const int nsteps = active ? (__builtin_amdgcn_readfirstlane(w.nslots[ch]) >> 5) : 0;       // this chunk's steps per image
  int wsteps = 0;                                                                           // the workgroup's (lockstep)
  for (int i = 0; i < w.NCH; ++i) {
    const int r = grp * w.NCH + i;
    if (r < w.nch) wsteps = max(wsteps, __builtin_amdgcn_readfirstlane(w.nslots[w.lpt ? chunk_of_rank(r, g.mx, g.my) : r]) >> 5);
  }
  const int nown = active ? __builtin_amdgcn_readfirstlane(w.nown[ch]) : 0;

  // ---- LDS: [HPW bias images | NCH x HPW rings of D x (K tile, V tile) | NCH x (row offsets | address terms) | flag]
  const unsigned lds0 = lds_addr(smem);
  const int tab_off = hslot * c.tabsize * 4;
  const unsigned tab_lds = lds0 + tab_off;
  const int ring_off = w.HPW * c.tabsize * 4 + slot_ * (D * SLOTB);
  char* ring = smem + ring_off;
  const int koff_lds = w.HPW * c.tabsize * 4 + w.NCH * w.HPW * (D * SLOTB) + cslot * (w.koff_lds + w.ak_lds);   // [NSP] ints (whole 1 KB pieces)
  const int ak_lds = koff_lds + w.koff_lds;                // [NSP] halfwords, per-lane order (k_cw_prep)
  unsigned* flag = (unsigned*)(smem + (size_t)w.HPW * c.tabsize * 4 + (size_t)w.NCH * (w.HPW * (D * SLOTB) + w.koff_lds + w.ak_lds));

  const int Nloc = g.nx * g.ny;
  const int kstride_b = (int)p.k_st * 2;
  const unsigned kv_bytes = (unsigned)(p.G + Nloc - 1) * (unsigned)kstride_b + M * 2;    // (zero keys of cyclic padding: vil_mfma_common.h)
  const float c1 = p.scale * LOG2E;
  const int* koff_ch = w.koff + (int64_t)ch * c.NSP;

  // ---- lane-constant addresses.  DMA: lane L of a piece fills LDS row sub * RPP + L / LPR, 16-byte slot L % LPR -- a
  // lane-linear image -- with the source chunk the tile's XOR swizzle assigns to that slot
  const int drow = lane / LPR;
  const int dchunk = ((lane % LPR) * 16) ^ tile_swz<MD>(drow);        // (rows sub * RPP + drow swizzle like drow: RPP is a multiple of 8)
  // K rows as the A operand: row hf * 16 + lj, 16 bytes at (ks * 64 + lg * 16) ^ swizzle (hf: + 16 rows, an immediate)
  unsigned kaddr[MK];
#pragma unroll
  for (int ks = 0; ks < MK; ++ks)
    kaddr[ks] = lds0 + ring_off + lj * ROWB + (((ks * 32 + lg * 8) * 2) ^ tile_swz<MD>(lj));
  // V^T through the transposed read: row hf * 16 + lg * 4 + lj / 4, 8 bytes at (dt * 32 + (lj & 3) * 8) ^ swizzle
  unsigned vaddr[MD];
#pragma unroll
  for (int dt = 0; dt < MD; ++dt) {
    const int row = lg * 4 + (lj >> 2);
    vaddr[dt] = lds0 + ring_off + TILE + row * ROWB + ((dt * 32 + (lj & 3) * 8) ^ tile_swz<MD>(row));
  }
  const unsigned akaddr = lds0 + ak_lds + lg * 16;            // + st * 64: this lane's eight address terms of step st
  const unsigned kfaddr = lds0 + koff_lds + drow * 4;         // + (st * 32 + sub * RPP) * 4: row offset of a DMA piece's row
  // constant A operand whose row 0 is all ones: D[0][j] = sum_k P^T[k][j]
  X8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (T)(lj == 0 ? 1.0f : 0.0f);

  // ---- this lane's query slots: column jj = (x, yq), tile qt <-> query (x, y = QT yq + qt)
  const int jj = wp * 16 + lj;
  const int qx = fdiv(jj, w.m_HQ), qyq = jj - qx * w.HQ;
  // vil_attn_fwd_full: tile 0 of the chunk's first unused column is the GLOBAL query, live against the chunk's own keys
  // (the first slots of the list: own_first); its partial (O, l) is saved after the last own step
  const bool gq_wave = w.gq_on && active && wp == w.gq_wp;            // (wave-uniform)
  const bool gqcol = gq_wave && lj == w.gq_lj;
  const int gq_steps = gq_wave ? ((p.G + nown + 31) >> 5) : 0;
  const unsigned aqb = tab_lds + (min(qx, W - 1) * c.P + QT * qyq) * 4 + w.akb;   // LDS byte address + the terms' bias; + 4 * qt per tile
  int qtok[QT];
  bool qreal[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int qy = QT * qyq + qt;
    const int qr = cm * W + qx, qc = cn * W + qy;
    qreal[qt] = active && qx < W && qy < W && qr < g.nx && qc < g.ny;
    qtok[qt] = qreal[qt] ? qr * g.ny + qc : (cm * W) * g.ny + cn * W;
  }
  const float g2l0v = (w.gq_on && p.g2l0) ? p.g2l0[h * p.G] * (SAFE ? 1.0f / p.scale : LOG2E) : 0.f;

  // ---- one-time requests: the chunk's slot tables (LDS-DMA, whole 1 KB pieces round-robin over its waves), bias image
  if (active) {
    const __amdgpu_buffer_rsrc_t trs = make_rsrc_n(koff_ch, (unsigned)c.NSP * 4u);
    const __amdgpu_buffer_rsrc_t ars = make_rsrc_n((const unsigned short*)w.akey + (int64_t)ch * c.NSP, (unsigned)c.NSP * 2u);
    const int npk = w.koff_lds >> 10, npa = w.ak_lds >> 10;
    for (int i = hslot * w.NWP + wp; i < npk + npa; i += w.NWP * w.HPW) {      // (one copy per chunk slot: its heads' waves share it)
      if (i < npk) cw_dma16(trs, smem + koff_lds + i * 1024, i * 1024 + lane * 16);
      else cw_dma16(ars, smem + ak_lds + (i - npk) * 1024, (i - npk) * 1024 + lane * 16);
    }
  }
  {
    // the bias images of the workgroup's HPW heads (in log2 units; the exact kernel works in score units: one multiply by
    // scale * log2 e serves scores and bias)
    const int h0 = h - hslot;
    for (int i = tid; i < w.HPW * (c.tabsize >> 2); i += blockDim.x) {
      const int hh = fdiv(i, w.m_tab4), e4 = i - hh * (c.tabsize >> 2);
      const f32x4 v = ((const f32x4*)(c.tabws + (int64_t)(h0 + hh) * c.tabstride))[e4];
      ((f32x4*)smem)[i] = SAFE ? v * (1.0f / c1) : v;
    }
    if (tid == 0) *flag = 0u;
  }

  auto rsrc_k = [&](int b) { return make_rsrc_n((const T*)p.k + (int64_t)b * p.k_sb + h * p.k_sh, kv_bytes); };
  auto rsrc_v = [&](int b) { return make_rsrc_n((const T*)p.v + (int64_t)b * p.v_sb + h * p.v_sh, kv_bytes); };
  // this lane's Q rows of image b (the global query's row is token 0 of q_g)
  auto load_q = [&](int b, X8 (&raw)[MK][QT]) {
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      int tk = qtok[qt];
      asm volatile("" : "+v"(tk));          // (per-image address work stays here)
      const T* qrow = ((gqcol && qt == 0) ? (const T*)p.q_g : (const T*)p.q + (int64_t)tk * p.q_st) + (int64_t)b * p.q_sb + h * p.q_sh + lg * 8;
#pragma unroll
      for (int ks = 0; ks < MK; ++ks) raw[ks][qt] = *(const X8*)(qrow + ks * 32);
    }
  };
  // Q' = Q * scale * log2(e), rounded to the operand type once
  auto scale_q = [&](X8 (&q_)[MK][QT]) {
    if (SAFE) return;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
      for (int ks = 0; ks < MK; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) q_[ks][qt][e] = (T)((float)q_[ks][qt][e] * c1);
  };

  f32x4 o[MD][QT], lacc[QT];
  float mrow[QT];
  X8 qf[MK][QT];

  auto reset_acc = [&]() {
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      mrow[qt] = VIL_M_INIT;
      lacc[qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int dt = 0; dt < MD; ++dt) o[dt][qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };
  // one step = 32 key slots of the tile pair in the ring slot at byte offset sl_b
  auto step = [&](unsigned sl_b, int st, const u32x4& ak, int b, int j, int n_mid) {
    // ---- S^T = K Q'^T + bias (the accumulator starts as the gathered bias)
    f32x4 sc[2][QT];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      X8 kc_[MK];
#pragma unroll
      for (int ks = 0; ks < MK; ++ks)
        kc_[ks] = *(const X8 __attribute__((address_space(3)))*)(size_t)(kaddr[ks] + sl_b + hf * 16 * ROWB);
      lds_cvf tb[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const unsigned a16 = (r & 1) ? (ak[hf * 2 + (r >> 1)] >> 16) : (ak[hf * 2 + (r >> 1)] & 0xffffu);
        tb[r] = lds_f32(aqb - a16);
      }
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        f32x4 acc = {tb[0][qt], tb[1][qt], tb[2][qt], tb[3][qt]};
        if (qt == 0 && st < gq_steps) {
          // the global query's column: g2l[0] against the chunk's own keys (slots [G, G + nown)), masked elsewhere
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int s_ = st * 32 + hf * 16 + lg * 4 + r;
            const float v = (s_ >= p.G && s_ < p.G + nown) ? g2l0v : VIL_MASK_VAL;
            acc[r] = gqcol ? v : acc[r];
          }
        }
#pragma unroll
        for (int ks = 0; ks < MK; ++ks) acc = mfma16(kc_[ks], qf[ks][qt], acc);
        sc[hf][qt] = acc;
      }
    }
    // ---- probabilities
    X8 pb[QT];
    if constexpr (SAFE) {
      // online softmax with a deferred maximum (score units: threshold 8 nats)
      const float thr = 8.0f / p.scale;
      float pm[QT];
      bool grow = false;
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        pm[qt] = max3f(max3f(max3f(sc[0][qt][0], sc[0][qt][1], sc[0][qt][2]), sc[0][qt][3], sc[1][qt][0]),
                       max3f(sc[1][qt][1], sc[1][qt][2], sc[1][qt][3]), mrow[qt]);
        grow |= pm[qt] > mrow[qt] + thr;
      }
      if (__any(grow)) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
          float mn = fmaxf(pm[qt], __shfl_xor(pm[qt], 16, 64));
          mn = fmaxf(mn, __shfl_xor(mn, 32, 64));
          const float alpha = __builtin_amdgcn_exp2f((mrow[qt] - mn) * c1);
          mrow[qt] = mn;
          lacc[qt] *= alpha;
#pragma unroll
          for (int dt = 0; dt < MD; ++dt) o[dt][qt] *= alpha;
        }
      }
    }
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      u32x4 wv;
      const float mq = SAFE ? mrow[qt] * c1 : 0.f;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2)
          wv[hf * 2 + r2] = SAFE ? pack2<T>((f32x2){__builtin_amdgcn_exp2f(__builtin_fmaf(sc[hf][qt][2 * r2], c1, -mq)),
                                                     __builtin_amdgcn_exp2f(__builtin_fmaf(sc[hf][qt][2 * r2 + 1], c1, -mq))})
                                 : pack2<T>((f32x2){__builtin_amdgcn_exp2f(sc[hf][qt][2 * r2]),
                                                     __builtin_amdgcn_exp2f(sc[hf][qt][2 * r2 + 1])});
      pb[qt] = __builtin_bit_cast(X8, wv);
    }
#if VIL_CW_LATE_V
    // the V tile of this step: its requests may still be in flight (only K was waited for at the top of the step)
    asm volatile("" ::: "memory");
    cw_wait_vm(n_mid);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#else
    (void)n_mid;
#endif
    // ---- O^T += V^T P^T ; row sums via the ones-row.  (Transposed reads as inline assembly: the builtin is modelled as
    // an LDS access that may write, and with the next tile's LDS-DMA in flight the compiler would wait for it first.)
#pragma unroll
    for (int d2 = 0; d2 < MD; d2 += 2) {
      s16x4 t00 = lds_tr_issue<0>(vaddr[d2] + sl_b), t10 = lds_tr_issue<16 * ROWB>(vaddr[d2] + sl_b);
      s16x4 t01 = lds_tr_issue<0>(vaddr[d2 + 1] + sl_b), t11 = lds_tr_issue<16 * ROWB>(vaddr[d2 + 1] + sl_b);
      lds_tr_settle(t00, t10, t01, t11);
      const X8 v0 = lds_tr_join<T>(t00, t10), v1 = lds_tr_join<T>(t01, t11);
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        o[d2][qt] = mfma16(v0, pb[qt], o[d2][qt]);
        o[d2 + 1][qt] = mfma16(v1, pb[qt], o[d2 + 1][qt]);
      }
    }
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) lacc[qt] = mfma16(ones, pb[qt], lacc[qt]);
    // ---- the global query's partial over the chunk's own keys: O (unnormalised), l, m -> k_gq_merge
    if (st + 1 == gq_steps) {
      const float l0 = __shfl(lacc[0][0], lj, 64);
      if (gqcol) {
        float* part = c.gq_parts + ((int64_t)(b * p.H + h) * w.nch + ch) * (M + 4);
#pragma unroll
        for (int dt = 0; dt < MD; ++dt) *(f32x4*)(part + dt * 16 + lg * 4) = o[dt][0];
        if (lg == 0) { part[M] = l0; part[M + 1] = SAFE ? mrow[0] : 0.f; }            // (m in the merge's score units)
        if (!SAFE && !(l0 < 1.6e7f)) __hip_atomic_fetch_or(flag, 1u << j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  };
  // image epilogue: normalise, store O (4 dims x 8 bytes per d-tile) and LSE; note the rows whose sums left the safe range
  auto finish_image = [&](int b, int j) {
    T* ob = (T*)p.o + (int64_t)b * p.o_sb + h * p.o_sh;
    const int bh = b * p.H + h;
    bool bad = false;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      const float l = __shfl(lacc[qt][0], lj, 64);        // row 0 lives in lane group 0
      const float inv = 1.0f / l;
      int tk = qtok[qt];
      asm volatile("" : "+v"(tk));          // (the row addresses are per-image work: hoisted out of the stream they are spilled)
      if (qreal[qt]) {
        // (fast kernel: the row's logits must stay within ~16 nats of zero -- beyond that the single bf16 rounding of Q'
        //  shows in the largest probabilities (|s| 2^-9 nats) and the exact kernel takes the image over)
        if (!SAFE) bad |= !(l > 6.0e-8f && l < 1.6e7f);
#pragma unroll
        for (int dt = 0; dt < MD; ++dt) {
          X4 wv;
#pragma unroll
          for (int r = 0; r < 4; ++r) wv[r] = (T)(o[dt][qt][r] * inv);
          *(X4*)(ob + (int64_t)tk * p.o_st + dt * 16 + lg * 4) = wv;
        }
        if (lg == 0)
          p.lse[(int64_t)bh * Nloc + tk] = (SAFE ? mrow[qt] * p.scale : 0.f) + __logf(l);
      }
    }
    if (!SAFE && __any(bad) && lane == 0) __hip_atomic_fetch_or(flag, 1u << j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    reset_acc();
  };
  // tile (cursor) of this wave's chunk has landed for every wave, and every wave is done with the tile before it
  auto step_sync = [&]() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0070);          // vmcnt(0) lgkmcnt(0): the BUILTIN, so that the compiler's own wait-count
    __builtin_amdgcn_s_barrier();                //   bookkeeping sees it (an opaque asm wait would be followed by its vmcnt(0))
    asm volatile("" ::: "memory");
  };

  // ---- the stream over the images whose bit is set in `mask` (bit j <-> image b0 + j * bstride).  Every image is wsteps
  // lockstep steps; a chunk with fewer key steps idles through the rest (groups are formed by rank of work: rare)
  auto stream = [&](unsigned mask) {
    reset_acc();
    int jc = __builtin_ctz(mask);
    unsigned mrem = mask & (mask - 1);                    // images after jc
    // request cursor: tile (jd, sd), one step ahead of the compute cursor (jc, st)
    int jd = jc, sd = 0;
    unsigned mdma = mrem;
    __amdgpu_buffer_rsrc_t krs = rsrc_k(b0 + jd * bstride), vrs = rsrc_v(b0 + jd * bstride);
    bool dma_live = nsteps > 0;
    auto request = [&](unsigned slot_b) -> int {          // request the cursor's tile into the ring slot, advance the cursor;
      int issued = 0;                                     // returns the pieces requested per matrix
      if (dma_live) {
        if (sd < nsteps) {
#if VIL_CW_LATE_V
          int kfv_[PPM];
#pragma unroll
          for (int k = 0; k < PPM; ++k) {                 // all K pieces first: they are waited for first
            const int sub = wp + k * w.NWP;
            if (sub < PPM) {
              kfv_[k] = *lds_i32(kfaddr + (sd * 32 + sub * RPP) * 4);
              cw_dma16(krs, ring + slot_b + sub * 1024, kfv_[k] + dchunk);
              ++issued;
            }
          }
#pragma unroll
          for (int k = 0; k < PPM; ++k) {
            const int sub = wp + k * w.NWP;
            if (sub < PPM) cw_dma16(vrs, ring + slot_b + sub * 1024 + TILE, kfv_[k] + dchunk);
          }
#else
#pragma unroll
          for (int k = 0; k < PPM; ++k) {
            const int sub = wp + k * w.NWP;
            if (sub < PPM) {
              int kfv = *lds_i32(kfaddr + (sd * 32 + sub * RPP) * 4);
              if (CW_ABL(512)) kfv &= 0x3fc0;           // (timing ablation: every request inside one hot 16 KB window)
              char* base = ring + slot_b + sub * 1024;
              cw_dma16(krs, base, kfv + dchunk);
              cw_dma16(vrs, base + TILE, kfv + dchunk);
              ++issued;
            }
          }
#endif
        }
        if (++sd == wsteps) {
          sd = 0;
          if (mdma) { jd = __builtin_ctz(mdma); mdma &= mdma - 1; krs = rsrc_k(b0 + jd * bstride); vrs = rsrc_v(b0 + jd * bstride); }
          else dma_live = false;
        }
      }
      return issued;
    };
    // prologue: slot tables / bias image in LDS (first stream), Q of the first image, tile 0
    load_q(b0 + jc * bstride, qf);
    step_sync();
    scale_q(qf);
    int cur_np = request(0u);
    (void)cur_np;
    u32x4 akc = *(const u32x4 __attribute__((address_space(3)))*)(size_t)(akaddr);
    X8 qn[MK][QT];
    int st = 0;
    unsigned slot_b = 0;
    bool qprev = true;            // ordinary loads behind the current tile's requests (LATE_V: then the top wait is for everything)
    for (;;) {
      // tile (jc, st) has landed; the other slot is free: request the next tile, fetch the next step's address terms
#if VIL_CW_LATE_V
      // ... its K half, that is: the V pieces (requested after the K pieces) may stay in flight until P V needs them
      asm volatile("" ::: "memory");
      cw_wait_vm(qprev ? 0 : cur_np);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const int next_np = request(slot_b ^ SLOTB);
#else
      if (!CW_ABL(1)) step_sync();
      int next_np = 0;
      if (!CW_ABL(2)) next_np = request(slot_b ^ SLOTB);
#endif
      const int sn = st + 1 < nsteps ? st + 1 : 0;
      const u32x4 akn = *(const u32x4 __attribute__((address_space(3)))*)(size_t)(akaddr + sn * 64);
      // one step before the image ends: the next image's Q rows (the wait at the top of the next step covers them)
      const bool qnow = (mrem != 0 && st + 2 == wsteps) || (mrem != 0 && wsteps == 1);
      if (mrem != 0 && st + 2 == wsteps) load_q(b0 + __builtin_ctz(mrem) * bstride, qn);
      // (LATE_V) requests younger than this step's V pieces: the next tile's -- unless ordinary loads are in between
      const int n_mid = (qprev || qnow) ? 0 : 2 * next_np;
      if (st < nsteps && !CW_ABL(64)) step(slot_b, st, akc, b0 + jc * bstride, jc, n_mid);
#if VIL_CW_LATE_V
      else { asm volatile("" ::: "memory"); cw_wait_vm(n_mid); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
      cur_np = next_np; qprev = qnow;
#else
      (void)qprev;
#endif
      akc = akn;
      slot_b ^= SLOTB;
      if (++st == wsteps) {
        if (active) finish_image(b0 + jc * bstride, jc);
        if (!mrem) break;
        jc = __builtin_ctz(mrem); mrem &= mrem - 1;
        if (wsteps == 1) { load_q(b0 + jc * bstride, qn); }             // (one-step images: no step to hide the rows behind)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
          for (int ks = 0; ks < MK; ++ks) qf[ks][qt] = qn[ks][qt];
        scale_q(qf);
        st = 0;
      }
    }
  };

  const unsigned all = nimg >= 32 ? 0xffffffffu : ((1u << nimg) - 1u);
  if constexpr (SAFE) {
    stream(redo_only ? redo_mask : all);
  } else {
    stream(all);
    // images with a row sum outside the safe range (or an overflowed global-query partial) are computed again by the exact
    // kernel launched behind this one (rare: |logit| beyond ~55 nats)
    __syncthreads();
    if (tid == 0) w.redo[blockIdx.x] = *flag;
  }
}

// ===================================================================== host side
// launch-shape override (tools/cw_check.py: image streams per column of the next launches; 0 = the library's own choice).
// Process-global like the other tuning hooks of the library; the product never calls it.
static int g_cw_streams = 0, g_cw_qt = 0, g_cw_nch = 0, g_cw_hpw = 0, g_cw_abl = 0;
extern "C" int vil_attn_cw_set_shape(int streams, int shape_code) {
  // shape_code: chunks per workgroup + 10 * query tiles per wave + 100 * heads per workgroup (0: the library's choice)
  const int nch = shape_code % 10, qt = (shape_code / 10) % 10, hpw = shape_code / 100;
  if (streams < 0 || shape_code < 0 || nch > 4 || (qt != 0 && qt != 1 && qt != 2) || hpw > 8) return VIL_E_SHAPE;
  g_cw_streams = streams; g_cw_nch = nch; g_cw_qt = qt; g_cw_hpw = hpw;
  return VIL_OK;
}
#ifdef VIL_CW_ABLATE
static void* g_cw_dbg = nullptr;
extern "C" int vil_attn_cw_set_ablation(int bits) { g_cw_abl = bits; return VIL_OK; }
extern "C" int vil_attn_cw_set_debug(void* buf) { g_cw_dbg = buf; return VIL_OK; }
#endif

#ifndef VIL_CW_QT
#define VIL_CW_QT 2
#endif
static size_t cw_lds_bytes(const VilAttnDesc* d, const MfmaCfg& c, const CwCfg& w) {
  return (size_t)w.HPW * c.tabsize * 4 + (size_t)w.NCH * (w.HPW * CW_D * 2 * (32 * d->M * 2) + w.koff_lds + w.ak_lds) + 16;
}
// key steps per image of the chunk at rank r (host mirror of build_key_slots' count for the zero-padded 3x3 / own-chunk /
// two-chunk lists; cyclic padding: every chunk sees full neighbours)
static int cw_steps_of_chunk(const VilAttnDesc* d, const VilGeom& g, int ch) {
  if (d->only_glo) return (d->G + 31) / 32;
  const int W = d->W, cm = ch / g.my, cn = ch % g.my;
  int keys = d->G;
  if (g.nact == 9 && g.exact != -1) {
    for (int dr = -1; dr <= 1; ++dr)
      for (int dc = -1; dc <= 1; ++dc) {
        const int rm = cm + dr, rn = cn + dc;
        if (rm < 0 || rm >= g.mx || rn < 0 || rn >= g.my) continue;
        const int rows = W < d->nx - rm * W ? W : d->nx - rm * W, cols = W < d->ny - rn * W ? W : d->ny - rn * W;
        keys += rows * cols;
      }
  } else {
    keys += g.nact * W * W;
  }
  return (keys + 31) / 32;
}
// host mirror of chunk_of_rank (vil_mfma_common.h)
static int cw_chunk_of_rank(int r, int mx, int my) {
  if (mx < 3 || my < 3) return r;
  const int iy = my - 2, ix = mx - 2, ni = ix * iy;
  if (r < ni) { const int a = r / iy; return (1 + a) * my + 1 + (r - a * iy); }
  r -= ni;
  if (r < iy) return 1 + r;
  r -= iy;
  if (r < iy) return (mx - 1) * my + 1 + r;
  r -= iy;
  if (r < ix) return (1 + r) * my;
  r -= ix;
  if (r < ix) return (1 + r) * my + my - 1;
  r -= ix;
  return (r >> 1) * (mx - 1) * my + (r & 1) * (my - 1);
}
// finishing time of `n` workgroups of `work[i]` (dispatched in order) on `cap` slots: every workgroup goes to the slot that
// frees first
static double cw_makespan(const double* work, int n, int cap) {
  std::priority_queue<double, std::vector<double>, std::greater<double>> slots;
  double end = 0.0;
  for (int i = 0; i < n; ++i) {
    double t0 = 0.0;
    if ((int)slots.size() >= cap) { t0 = slots.top(); slots.pop(); }
    const double t1 = t0 + work[i];
    slots.push(t1);
    if (t1 > end) end = t1;
  }
  return end;
}
// Image streams per column (by_image launches).  A column's workgroup walks the images of its stream one after the other, so a
// column of interior chunks (nine neighbour chunks of keys) takes 9 / 4 of the time of a corner column for the same images,
// and with ONE stream count for all columns the corner and edge workgroups finish early.  VIL_CW_CLASS_STREAMS=1 gives every
// class of chunk groups (interior / edge / corner: the segments below) its own stream count -- the fewest streams that bring
// its workgroups under a common target time, the target chosen by a simulated finishing time on the XCD's slots -- so that
// all workgroups of the launch finish together.  MEASURED, NOT ADOPTED (profiles/r06_streams_ab.txt): 28 x 28 at head_dim 64
// 77.6 - 78.0 us against 72.1 - 72.2 us for four streams everywhere (corner columns with two streams instead of four), 56 x 56
// 202 - 206 against 201 - 207 (four streams) / 211 - 215 (two).  The slot model is wrong for these kernels in the same way
// round 5's rank-order ablation was: a workgroup that finishes early does not idle a resource, it speeds its CU's other
// waves up.  The segments stay (dispatch order: heaviest class first); the stream count is one number again.
static void cw_plan_streams(const VilAttnDesc* d, const VilGeom& g, CwCfg& w, int cap, int nimg_x) {
  // segments: the groups of interior, edge and corner chunks (rank order = chunk_of_rank's classes; a group that straddles a
  // class boundary counts with the heavier class); one segment when the chunks are not taken by rank
  int units[4] = {0, 0, 0, 0};
  int first_rank[4] = {0, w.nch, w.nch, w.nch};
  w.nseg = 1;
  if (w.lpt && g.mx >= 3 && g.my >= 3) {
    const int ni = (g.mx - 2) * (g.my - 2);
    first_rank[1] = ni; first_rank[2] = w.nch - 4;
    w.nseg = 3;
  }
  for (int c = 0; c < w.nseg; ++c) {
    const int g0 = c == 0 ? 0 : (first_rank[c] + w.NCH - 1) / w.NCH;              // first group wholly inside class c or lighter
    const int g1 = c + 1 < w.nseg ? (first_rank[c + 1] + w.NCH - 1) / w.NCH : w.ngrp;
    w.seg_g0[c] = g0 < w.ngrp ? g0 : w.ngrp;
    w.seg_ng[c] = (g1 < w.ngrp ? g1 : w.ngrp) - w.seg_g0[c];
    if (w.seg_ng[c] < 0) w.seg_ng[c] = 0;
    for (int gi = w.seg_g0[c]; gi < w.seg_g0[c] + w.seg_ng[c]; ++gi)
      for (int i = 0; i < w.NCH; ++i) {
        const int r = gi * w.NCH + i;
        if (r >= w.nch) break;
        const int st = cw_steps_of_chunk(d, g, w.lpt ? cw_chunk_of_rank(r, g.mx, g.my) : r);
        if (st > units[c]) units[c] = st;
      }
  }
  // (empty segments are dropped)
  {
    int k = 0;
    for (int c = 0; c < w.nseg; ++c)
      if (w.seg_ng[c] > 0) { w.seg_g0[k] = w.seg_g0[c]; w.seg_ng[k] = w.seg_ng[c]; units[k] = units[c]; ++k; }
    w.nseg = k;
  }
  const double pro = 0.35 * units[0];              // a workgroup's prologue in step units of the longest column's image
  int best_ns[4] = {1, 1, 1, 1};
  if (g_cw_streams > 0) {
    int ns = g_cw_streams;
    if (ns > nimg_x) ns = nimg_x;
    for (int c = 0; c < w.nseg; ++c) best_ns[c] = ns;
  } else if (!VIL_CW_CLASS_STREAMS || w.ncolx > cap) {
    // one stream count for all columns: the fewest streams (longest-lived workgroups: tables and bias image loaded once per
    // stream) that fill the XCD's workgroup slots in whole rounds.  Also with more columns than slots (96 x 96: 294 columns,
    // 160 slots): one stream -- every workgroup walks all of its XCD's images -- measured 149 - 151 us against 160 - 167 us
    // for one image per workgroup (profiles/r06_streams_sweep_96.txt), although an XCD's L2 does not hold one image there.
    double best = 1e9;
    int ns = 1;
    for (int t = 1; t <= nimg_x; ++t) {
      const int wgs = w.ncolx * t, rounds = (wgs + cap - 1) / cap;
      const double cost = rounds * ((double)((nimg_x + t - 1) / t) + 0.35);        // rounds x images per workgroup (+ a prologue's worth)
      if (cost < best - 1e-9) { best = cost; ns = t; }
    }
    for (int c = 0; c < w.nseg; ++c) best_ns[c] = ns;
  } else {
    // (measured alternative, off: see the comment above)
    // candidates: every target time "class c split t ways"; per class the fewest streams that meet the target; scored by the
    // simulated finishing time of the launch in its dispatch order.  Plans of more than 1.25 rounds of workgroups are not
    // considered: the later rounds walk images the first one has left behind in L2.
    double best = 1e30;
    std::vector<double> work;
    for (int tc = 0; tc < w.nseg; ++tc)
      for (int t = 1; t <= nimg_x; ++t) {
        const double target = (double)((nimg_x + t - 1) / t) * units[tc];
        int ns[4], wgs = 0;
        for (int c = 0; c < w.nseg; ++c) {
          ns[c] = 1;
          while (ns[c] < nimg_x && (double)((nimg_x + ns[c] - 1) / ns[c]) * units[c] > target + 1e-9) ++ns[c];
          wgs += w.seg_ng[c] * w.NHG * ns[c];
        }
        if (wgs * 4 > cap * 5) continue;
        // dispatch order: segments by decreasing work per workgroup
        int ord[4] = {0, 1, 2, 3};
        for (int i = 0; i < w.nseg; ++i)
          for (int j = i + 1; j < w.nseg; ++j)
            if ((double)((nimg_x + ns[ord[j]] - 1) / ns[ord[j]]) * units[ord[j]] > (double)((nimg_x + ns[ord[i]] - 1) / ns[ord[i]]) * units[ord[i]]) {
              const int x = ord[i]; ord[i] = ord[j]; ord[j] = x;
            }
        work.clear();
        for (int i = 0; i < w.nseg; ++i) {
          const int c = ord[i];
          for (int s_ = 0; s_ < ns[c]; ++s_) {
            const int imgs = (nimg_x - s_ + ns[c] - 1) / ns[c];
            for (int k = 0; k < w.seg_ng[c] * w.NHG; ++k) work.push_back(imgs * (double)units[c] + pro);
          }
        }
        const double cost = cw_makespan(work.data(), (int)work.size(), cap);
        if (cost < best - 1e-9) { best = cost; for (int c = 0; c < w.nseg; ++c) best_ns[c] = ns[c]; }
      }
    if (best > 1e29) {              // (nothing fits: as many columns as slots -- one stream each)
      for (int c = 0; c < w.nseg; ++c) best_ns[c] = 1;
    }
  }
  for (int c = 0; c < w.nseg; ++c)
    while ((nimg_x + best_ns[c] - 1) / best_ns[c] > 32) ++best_ns[c];          // (the redo mask of a workgroup holds 32 images)
  // segments in dispatch order: decreasing work per workgroup
  for (int i = 0; i < w.nseg; ++i)
    for (int j = i + 1; j < w.nseg; ++j) {
      const double wi = (double)((nimg_x + best_ns[i] - 1) / best_ns[i]) * units[i], wj = (double)((nimg_x + best_ns[j] - 1) / best_ns[j]) * units[j];
      if (wj > wi) {
        int x;
        x = best_ns[i]; best_ns[i] = best_ns[j]; best_ns[j] = x;
        x = units[i]; units[i] = units[j]; units[j] = x;
        x = w.seg_g0[i]; w.seg_g0[i] = w.seg_g0[j]; w.seg_g0[j] = x;
        x = w.seg_ng[i]; w.seg_ng[i] = w.seg_ng[j]; w.seg_ng[j] = x;
      }
    }
  // one stream count for every class: ONE segment, i.e. the XCD's list is stream-major over all columns -- all columns of an
  // image are dispatched next to each other and meet its K / V in L2 together.  (Segment-major order with several rounds of
  // workgroups walks every image once per class: in-step HBM traffic of the pass 1.83x the algorithmic bytes at 56 x 56
  // instead of 1.18x, profiles/r06_pmc_traffic.json history in DESIGN.md 4.9.)
  {
    bool uniform = true;
    for (int c = 1; c < w.nseg; ++c) uniform = uniform && best_ns[c] == best_ns[0];
    if (uniform) { w.nseg = 1; w.seg_g0[0] = 0; w.seg_ng[0] = w.ngrp; }
  }
  int wg0 = 0;
  w.NS = 1;
  for (int c = 0; c < w.nseg; ++c) {
    w.seg_ns[c] = best_ns[c];
    w.seg_wg0[c] = wg0;
    w.seg_m[c] = vil_magic((unsigned)(w.seg_ng[c] * w.NHG));
    wg0 += w.seg_ng[c] * w.NHG * best_ns[c];
    if (best_ns[c] > w.NS) w.NS = best_ns[c];
  }
  for (int c = w.nseg; c < 4; ++c) { w.seg_g0[c] = w.ngrp; w.seg_ng[c] = 0; w.seg_ns[c] = 1; w.seg_wg0[c] = wg0; w.seg_m[c] = vil_magic(1u); }
  w.nwgx = wg0;
}
static bool cw_make_cfg(const VilAttnDesc* d, const MfmaCfg& c, CwCfg& w) {
  memset(&w, 0, sizeof(w));
  VilGeom g; vil_geom_init(g, d->nx, d->ny, d->W, d->exact, d->mode);
  const int W = d->W;
  w.QT = g_cw_qt > 0 ? g_cw_qt : VIL_CW_QT;
  w.HQ = (W + w.QT - 1) / w.QT;
  w.NWP = (W * w.HQ + 15) / 16;
  w.nch = g.mx * g.my;
  // heads per workgroup: one.  (Putting the H heads of a chunk into one lockstep workgroup, so that their requests for the two
  // halves of a K / V cache line meet in the CU's L1 -- the micro-benchmark's 23 - 27 TB/s instead of 16.7 -- LOSES in the
  // kernel: 244 - 267 us against 204 us at ViL-Small stage 1, 195 - 221 against 158 us at 96 x 96: six waves per barrier and
  // three bias images per workgroup cost more than the L2 path gives back.  Kept selectable: vil_attn_cw_set_shape.)
  int hpw = 1;
  if (g_cw_hpw > 0 && d->H % g_cw_hpw == 0) hpw = g_cw_hpw;
  if (hpw * w.NWP > 8) hpw = 1;
  w.HPW = hpw; w.NHG = d->H / hpw;
  // chunks per workgroup: two at head_dim 32 with one head per workgroup (one bias image per four waves), else one
  int nchw = g_cw_nch > 0 ? g_cw_nch : ((d->M <= 32 && hpw == 1) ? VIL_CW_NCH : 1);
  while (nchw > 1 && nchw * hpw * w.NWP > 8) nchw >>= 1;
  if (nchw > w.nch) nchw = w.nch;
  w.NCH = nchw < 1 ? 1 : nchw;
  w.ngrp = (w.nch + w.NCH - 1) / w.NCH;
  w.lpt = g.nact == 9 && g.exact != -1;
  w.m_HQ = vil_magic((unsigned)w.HQ);
  w.m_NWP = vil_magic((unsigned)w.NWP);
  w.m_ngrp = vil_magic((unsigned)w.ngrp);
  w.m_HPW = vil_magic((unsigned)w.HPW); w.m_NHG = vil_magic((unsigned)w.NHG); w.m_tab4 = vil_magic((unsigned)(c.tabsize >> 2));
  w.NSP = c.NSP;
  w.abl = g_cw_abl;
#ifdef VIL_CW_ABLATE
  w.dbg = g_cw_dbg;
#endif
  w.koff_lds = ((c.NSP * 4 + 1023) >> 10) << 10;
  w.ak_lds = ((c.NSP * 2 + 1023) >> 10) << 10;
  w.akb = (c.glo0 + d->G * c.gsz) * 4;        // the smallest address term is -(glo0 + (G - 1) gsz + ...) * 4
  // columns of one XCD's list and image streams per column
  w.by_image = d->B >= 8;
  const int nimg_x = w.by_image ? (d->B + 7) / 8 : 1;
  w.ncolx = w.by_image ? w.NHG * w.ngrp : ((d->B * w.NHG + 7) / 8) * w.ngrp;
  w.m_ncolx = vil_magic((unsigned)w.ncolx);
  // workgroups an XCD holds at once: registers (cw_occ waves per SIMD) and LDS
  const size_t lds = cw_lds_bytes(d, c, w);
  int per_cu = cw_occ(d->M / 16, w.QT) * 4 / (w.NWP * w.NCH * w.HPW);
  if (per_cu > (int)((160 * 1024) / lds)) per_cu = (int)((160 * 1024) / lds);
  if (per_cu < 1) per_cu = 1;
  const int cap = per_cu * (vil_cu_count() / 8 > 0 ? vil_cu_count() / 8 : 32);
  w.NS = 1; w.nseg = 0; w.nwgx = w.ncolx;
  if (w.by_image) cw_plan_streams(d, g, w, cap, nimg_x);
  return true;
}
// can the global token's query row ride (vil_attn_fwd_full)?  One global token, local keys attended, an unused column
static bool cw_gq_fusable(const VilAttnDesc* d, const CwCfg& w) {
  return d->G == 1 && !d->only_glo && d->W * w.HQ < 16 * w.NWP;
}

int vil_cw_supported(const VilAttnDesc* d, int pass) {
  if (pass != 0) return VIL_E_BACKEND;
  if (d->dtype != VIL_DTYPE_BF16 && d->dtype != VIL_DTYPE_F16) return VIL_E_DTYPE;
  if (d->M != 32 && d->M != 64) return VIL_E_HEAD_DIM;
  if (d->W < 1 || d->W > 16) return VIL_E_WINDOW;
  if (d->G > 16) return VIL_E_BACKEND;
  if ((d->q_st | d->k_st | d->v_st | d->q_sb | d->k_sb | d->v_sb | d->q_sh | d->k_sh | d->v_sh) & 7) return VIL_E_ALIGN;
  if ((d->o_st | d->o_sb | d->o_sh) & 3) return VIL_E_ALIGN;
  const int64_t ntok = (int64_t)d->G + (int64_t)d->nx * d->ny;
  if (d->k_st != d->v_st || d->k_st >= (1 << 22) || ntok >= (1 << 23) || d->k_st * 2 * ntok >= (1ll << 31))
    return VIL_E_BACKEND;
  MfmaCfg c; vil_mfma_make_cfg(d, c);
  CwCfg w; cw_make_cfg(d, c, w);
  if (w.NWP * w.NCH * w.HPW > 8) return VIL_E_BACKEND;
  if (cw_lds_bytes(d, c, w) > 160 * 1024) return VIL_E_BACKEND;
  if (w.akb + c.tabsize * 4 >= 65536) return VIL_E_BACKEND;                 // 16-bit address terms
  if ((uint64_t)w.nwgx * 8 >= (1ull << 31)) return VIL_E_BACKEND;
  return VIL_OK;
}

// the launch plan of a descriptor (tools and tests: no launch, no device): out24 = {nseg, workgroups per XCD, NS, NCH, NHG,
// chunk groups, by_image, chunks, then per segment (4x): first group, groups, streams, first workgroup}
extern "C" int vil_attn_cw_plan(const VilAttnDesc* d, int32_t* out24) {
  if (!d || !out24) return VIL_E_NULL;
  const int e = vil_cw_supported(d, 0);
  if (e) return e;
  MfmaCfg c; vil_mfma_make_cfg(d, c);
  CwCfg w; cw_make_cfg(d, c, w);
  const int head[8] = {w.nseg, w.nwgx, w.NS, w.NCH, w.NHG, w.ngrp, w.by_image, w.nch};
  for (int i = 0; i < 8; ++i) out24[i] = head[i];
  for (int k = 0; k < 4; ++k) {
    out24[8 + 4 * k] = w.seg_g0[k]; out24[9 + 4 * k] = w.seg_ng[k]; out24[10 + 4 * k] = w.seg_ns[k]; out24[11 + 4 * k] = w.seg_wg0[k];
  }
  return VIL_OK;
}

static size_t cw_align4(size_t x) { return (x + 3) & ~(size_t)3; }
size_t vil_cw_workspace(const VilAttnDesc* d, int pass) {
  if (pass != 0) return 0;
  MfmaCfg c; vil_mfma_make_cfg(d, c);
  CwCfg w; cw_make_cfg(d, c, w);
  size_t fl = cw_align4((size_t)d->H * c.tabsize) + 2 * cw_align4((size_t)w.nch * c.NSP) + 2 * cw_align4((size_t)w.nch) + cw_align4((size_t)8 * w.nwgx);
  if (cw_gq_fusable(d, w)) fl += cw_align4((size_t)d->B * d->H * w.nch * (d->M + 4));
  return fl * sizeof(float);
}

int vil_mfma_launch_gq_merge(const VilAttnDesc* d, const VilParams& p, const MfmaCfg& c, hipStream_t s);

int vil_cw_fwd(const VilAttnDesc* d, VilParams& p, hipStream_t s) {
  MfmaCfg c; vil_mfma_make_cfg(d, c);
  CwCfg w; cw_make_cfg(d, c, w);
  float* ws = (float*)p.delta;            // workspace base
  if (p.glo_rows && !cw_gq_fusable(d, w)) return VIL_E_BACKEND;
  if (((uintptr_t)p.q | (uintptr_t)p.k | (uintptr_t)p.v | (uintptr_t)ws) & 15) return VIL_E_ALIGN;
  if ((uintptr_t)p.o & 7) return VIL_E_ALIGN;
  c.tabws = ws; c.tabstride = c.tabsize;
  size_t off = cw_align4((size_t)d->H * c.tabsize);
  w.koff = (int*)(ws + off); off += cw_align4((size_t)w.nch * c.NSP);
  w.akey = (int*)(ws + off); off += cw_align4((size_t)w.nch * c.NSP);
  w.nslots = (int*)(ws + off); off += cw_align4((size_t)w.nch);
  w.nown = (int*)(ws + off); off += cw_align4((size_t)w.nch);
  w.redo = (unsigned*)(ws + off); off += cw_align4((size_t)8 * w.nwgx);
  if (p.glo_rows) {
    c.gq_parts = ws + off;
    c.gq_on = 1;
    w.gq_on = 1;
    const int gj = d->W * w.HQ;
    w.gq_wp = gj / 16; w.gq_lj = gj % 16;
  }
  const VilWork wk(d);
  vil_prof_begin(VIL_K_TABLE, s, 0, 0);
  {
    const int ntx = (c.tabsize + 255) / 256;
    if (int he = vil_ensure_dyn_lds((const void*)k_cw_prep, (size_t)c.NSP * 8)) return he;
    k_cw_prep<<<dim3((unsigned)(ntx * p.H + w.nch)), dim3(256), (size_t)c.NSP * 8, s>>>(p, c, w, (int)p.k_st * 2, ntx);
  }
  vil_prof_end(s);
  int e = (int)hipGetLastError();
  if (e) return e;
  vil_prof_begin(VIL_K_MFMA_FWD, s, wk.fwd_bytes(), wk.fwd_flops());
  const unsigned grid = 8u * (unsigned)w.nwgx;
  const size_t lds = cw_lds_bytes(d, c, w);
  const unsigned nthr = 64u * (unsigned)(w.NWP * w.NCH * w.HPW);
#define LAUNCH_CW(T_, MD_, QT_, SAFE_, RO_)                                                        \
  {                                                                                                \
    if (int he = vil_ensure_dyn_lds((const void*)k_cw_fwd<T_, MD_, QT_, SAFE_>, lds)) return he;   \
    k_cw_fwd<T_, MD_, QT_, SAFE_><<<dim3(grid), dim3(nthr), lds, s>>>(p, c, w, RO_);                \
  }
#define LAUNCH_CW_Q(T_, SAFE_, RO_)                                                                \
  {                                                                                                \
    if (w.QT == 1) { if (d->M == 32) LAUNCH_CW(T_, 2, 1, SAFE_, RO_) else LAUNCH_CW(T_, 4, 1, SAFE_, RO_) } \
    else { if (d->M == 32) LAUNCH_CW(T_, 2, 2, SAFE_, RO_) else LAUNCH_CW(T_, 4, 2, SAFE_, RO_) }  \
  }
  if (d->dtype == VIL_DTYPE_F16) LAUNCH_CW_Q(_Float16, true, 0)
  else {
    LAUNCH_CW_Q(__bf16, false, 0)
    if ((e = (int)hipGetLastError())) return e;
    LAUNCH_CW_Q(__bf16, true, 1)
  }
  vil_prof_end(s);
  if ((e = (int)hipGetLastError())) return e;
  if (p.glo_rows) return vil_mfma_launch_gq_merge(d, p, c, s);
  return VIL_OK;
}
