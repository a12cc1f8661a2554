// vil_mfma_common.h -- types, launch configuration, LDS bias-table layout and the per-unit
// key-slot table builder shared by the MFMA forward and backward kernels.
#pragma once
#include "vil_internal.h"
#include <string.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
// The MFMA kernels are templated on the 16-bit I/O element type T: __bf16 (the build's training precision) or
// _Float16 (the reference's AMP dtype: fp16 autocast + GradScaler, src/engine.py:84, run_experiment.py:206).
// Accumulation, softmax and every reduction are fp32 for both.
template <typename T> struct V16 {
  typedef T x8 __attribute__((ext_vector_type(8)));
  typedef T x4 __attribute__((ext_vector_type(4)));
  typedef T x2 __attribute__((ext_vector_type(2)));
};
typedef float f32x2 __attribute__((ext_vector_type(2)));
// two fp32 -> one dword of two 16-bit values (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32): written as a vector conversion so
// that the pairing is fixed in the source (left to the SLP vectoriser, scalar conversions next to packed multiplies
// came out paired across the operand dwords and were re-assembled with v_alignbit / v_perm)
template <typename T> __device__ __forceinline__ unsigned pack2(f32x2 v) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, typename V16<T>::x2));
}
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define VIL_CPAD 3   // left pad (floats) of every bias-table row in LDS
#define VIL_MASK_VAL (-1.0e30f)
#define VIL_M_INIT (-1.0e20f)
#define LOG2E 1.4426950408889634f

// LDS bias table of one head (floats):
//   [ trows rows x P : bias/scale at column CPAD + (dy + tcen); exact==1: -1e30 outside the window |
//     gsz x -1e30 : where masked / padded key slots point |
//     G x gsz x g2l[h][g]/scale : where the global key slots point ]
// The table entry of (query (xq,yq), key (X,Y) in the query chunk's frame) is at
//   Aq - Ak + aconst,  Aq = xq*P + yq,  Ak = X*P + Y,  aconst = tcen*(P+1) + CPAD,
// so a lane gathers it with one v_sub (per key) and an immediate offset (per query of its quad).
// P == 11 (mod 32) spreads the 16 query columns of a wave (x*P + 4*hq) over distinct banks.
// ---- persistent workgroups of the dQ pass (round 4).  The sliding-chunk passes launch short-lived workgroups of four
// adjacent chunks in (image, chunk group, head) order; the dQ pass could not (every workgroup leaves a histogram
// partial), ran 4096 long-lived workgroups in (image, head) order and moved 1.83x its algorithmic bytes.  It now launches
// as many workgroups as are resident at once, and every WAVE walks its own list of (image, chunk) units:
//   * a workgroup is bound to one head (the bias-table image and histogram it holds in LDS: head = (blockIdx / 8) % H)
//     and to the XCD the hardware places it on (blockIdx % 8), whose queue covers an eighth of the images --
//     the H workgroups that walk the same images run on the same L2;
//   * entry k of wave w's list (w = the wave's index among the nw waves that serve the queue) is unit
//     k * nw + (w + 37 k) mod nw: at any time the waves of an XCD work inside a window of a few rows of nw units, the four
//     waves of a workgroup sit on ADJACENT chunks (their 3x3 neighbourhoods overlap: the CU's L1 serves part of their
//     K / V reads), and the rotation deals every wave a mix of interior, edge and corner chunks;
//   * inside an image the chunks come in order of decreasing work (chunk_of_rank), so the last entries are short units.
// dQ in the ViL-Small step: HBM traffic 1.83x -> 0.83x of the algorithmic bytes (profiles/r04_pmc_traffic.json), one
// 64-bit histogram record per workgroup (768 instead of 4096: the reduce pass 32 -> 9 us).
// The forward and dK/dV passes keep their short-lived workgroups: with per-wave lists they run at the same speed but the
// waves drift apart over their entries and the L2 window with them (forward 1.05x -> 1.86x of the algorithmic bytes at
// 56x56); handing units out through global atomic tickets -- per unit, per block of four, prefetched or not -- made
// every pass 2 - 3.6x slower (a returning global atomic under these kernels' load stream takes tens of microseconds,
// and loads return in order behind it); a workgroup-level list drawn through an LDS counter loses the adjacency
// (forward at 48x48: 74 us against 64.5 us): profiles/r04_queue_ablation.txt.
struct UnitQueue {
  int units_bh;          // units per (image, head)
  unsigned m_units_bh;   // vil_magic(units_bh)
};
#define VIL_POOL_ROT 37
#define VIL_MAX_PERSISTENT_WGS 2048     // bound of a persistent grid (the dQ pass's histogram partials are laid out for it)

struct MfmaCfg {
  int trows, tcen;   // rows (= columns) of the LDS bias image and its centre: 4W-1 / 2W-1, or 2W-1 / W-1 when the
                     // chunk attends only itself (mode -1: |dx|,|dy| <= W-1 -- a 5x smaller image for the dense stages)
  int P;             // row pitch (floats)
  int tabsize;       // floats per head (multiple of 4)
  int guard0;        // start of the all-masked region
  int glo0;          // start of the per-global-token constant regions
  int gsz;           // size of one such region (>= Aq range)
  int aconst;
  unsigned magicW, magicW2;
  unsigned m_wgbh, m_H, m_NWP, m_my, m_HQ;   // magic reciprocals (vil_magic) of wg_per_bh*H, H, NWP, my, HQ
  int HQ;            // query quads per chunk row = ceil(W/4)
  int NWP;           // waves per chunk = ceil(W*HQ/16)
  int NS;            // real key slots = G + nact*W2
  int NSP;           // padded to a multiple of 32
  int units_bh;      // mx*my*NWP
  int wg_per_bh, gpw;
  int wpw;           // waves per workgroup (4, or fewer when the per-wave LDS is large)
  int wave_lds;      // bytes of private LDS per wave (forward / dQ pass)
  const float* tabws;  // (H, tabsize) prepared bias tables
  int2* key_slots;     // (mx*my, NSP): key-slot table of every query chunk (key_slots_block), .x = K/V row byte offset, .y = bias term
  int* key_nslots;     // (mx*my): padded slot count of each
  UnitQueue uq;        // forward pass units (query chunks x NWP)
  // vil_attn_fwd_full (G == 1): the global token's QUERY row rides in the forward pass as the first unused query column
  // of every chunk's last wave, live against the chunk's OWN keys only (every key has exactly one own chunk); the unit
  // leaves a partial (m, l, O) that k_gq_merge combines with the global key's term.  Its bias comes from a constant
  // image behind the head's table: g2l[0][h][0] / scale where the key address term belongs to the own chunk, masked elsewhere.
  int tabstride;       // floats between two heads' images in tabws (tabsize, or tabsize + gq_ext)
  int gq_on;           // the forward launch carries the global query column
  int gq_jj;           // its (x, hq) pair index within the chunk (= W * HQ)
  int gq_a0;           // the column's address term: table index tabsize + gq_a0 - (Ak - aconst)
  int gq_ext;          // floats of the image extension (multiple of 4)
  float* gq_parts;     // (B*H, chunks, M + 4): O (unnormalised), l, m, pad per chunk
};

// n / d for a run-time divisor without the ~25-instruction integer division sequence: magic = floor(2^32 / d) + 1
// (host: vil_magic), exact for n * d < 2^32 (the host checks the largest n of every use); magic 0 encodes d == 1.
// The kernels decode (image, head, chunk, row, column) from the workgroup index with six such divisions per wave.
__device__ __forceinline__ unsigned fdiv(unsigned n, unsigned magic) { return magic ? __umulhi(n, magic) : n; }
static inline unsigned vil_magic(unsigned d) { return d <= 1 ? 0u : (unsigned)(0x100000000ull / d) + 1u; }

// LDS byte address of a __shared__ object as a plain integer, and a dword read at such an address.  The bias
// gather of the MFMA kernels is `table[Aq - Ak + qt]` for 4 keys (r) x 4 queries (qt) per lane: written through
// generic pointers the compiler pairs the qt-consecutive entries into ds_read2_b32 and then needs 14 v_mov per
// 16 entries to transpose them into the accumulators' (r-major) registers, plus a `v_add 0` per key for the
// relocatable LDS base.  Volatile dword reads at integer addresses land directly in the accumulator registers:
// one v_sub per key, immediate offsets per query (ISA: 4 v_sub + 16 ds_read_b32 instead of 8 VALU + 8
// ds_read2_b32 + 14 v_mov; the LDS array cycles are the same, a ds_read2_b32 counts as two reads).
typedef __attribute__((address_space(3))) const volatile float* lds_cvf;
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
}
__device__ __forceinline__ lds_cvf lds_f32(unsigned a) { return (lds_cvf)(size_t)a; }
__device__ __forceinline__ __attribute__((address_space(3))) int* lds_i32(unsigned a) {
  return (__attribute__((address_space(3))) int*)(size_t)a;
}
// three-operand maximum (v_max3_f32)
__device__ __forceinline__ float max3f(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

// XOR swizzle (bytes, a multiple of 32) of the wave-private [32 rows][16 MD elements] LDS tiles of the sliding-chunk kernels.
// The tiles are read three ways: ds_read_b64_tr_b16 (two 32-lane groups, lane (j, g) -> row 4g + j/4, 8 bytes of a 32-byte
// run), ds_read_b128 along rows (four 16-lane groups), ds_write_b128 (eight 8-lane groups); banks are (a / 4) mod 64
// for all three (MI355X_MICROARCH.md, LDS).  64-byte rows (head_dim 32): rows r and r + 4 share their banks -> flip the
// 32-byte half with bit 2 of the row.  128-byte rows (head_dim 64): rows r and r + 2 share their banks, so the four
// even (odd) rows of a group need four different 32-byte quarters: bits 1-2 of the row (round 4: it had been bit 2 only,
// as for 64-byte rows -- SQ_LDS_BANK_CONFLICT 26 / 43 / 32 % of the forward / dQ / dK,dV LDS cycles at head_dim 64).
// Odd MD (96-byte rows): no swizzle.
template <int MD> __device__ __forceinline__ constexpr int tile_swz(int row) {
#ifdef VIL_TILE_SWZ_OLD
  return (MD % 2 == 0) ? (((row >> 2) & 1) << 5) : 0;
#else
  return MD == 4 ? (((row >> 1) & 3) << 5) : (MD == 2 ? (((row >> 2) & 1) << 5) : 0);
#endif
}

// XCD-aware bijective remap: consecutive logical workgroups (same image/head, neighbouring
// chunks -> shared K/V) land on the same XCD's L2 (hardware places block b on XCD b % 8)
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
}

// The unit list of one wave.  entry(k) returns the k-th unit as an index into head h's image-major unit list, or -1 when
// the list is exhausted (entries grow: once one is beyond the queue, all later ones are).
struct UnitList {
  int q0, nq, w, nw;     // the (XCD, head) queue: first unit, units; this wave's index among the queue's nw waves
  __device__ __forceinline__ void init(const UnitQueue& q, int B, int H, int wave, int waves_per_wg) {
    // the XCD's share of head h's image-major unit list: an eighth of the UNITS (whole images when B is a multiple of 8;
    // with B < 8 -- the reference's operator benchmark runs B = 2 -- image boundaries fall inside a queue)
    const int xcd = blockIdx.x & 7;
    const long long U = (long long)B * q.units_bh;
    q0 = (int)((U * xcd) >> 3); nq = (int)((U * (xcd + 1)) >> 3) - q0;
    w = ((int)(blockIdx.x >> 3) / H) * waves_per_wg + wave; nw = ((int)gridDim.x / (8 * H)) * waves_per_wg;
  }
  __device__ __forceinline__ int entry(int k) const {
    const int u = k * nw + (w + VIL_POOL_ROT * k) % nw;
    return u < nq ? q0 + u : -1;
  }
};
// chunk of rank r when the chunks of an (mx, my) grid are ordered by decreasing work: with the full 3x3 neighbourhood an
// interior chunk sees 9 chunks of keys (and is seen by 9 query chunks), an edge chunk 6, a corner chunk 4
__device__ __forceinline__ int chunk_of_rank(int r, int mx, int my) {
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

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, (short)0, 0x7fffffff, 0x00020000);
}
// bounded descriptor: loads at byte offsets >= nbytes return zeros.  The K / V descriptors of one (image, head) slice
// are bounded so that the ZERO keys of cyclic padding (exact = -1: a zero-padded position reached by wrap-around stays
// in the softmax with k = v = 0, reference slidingchunk_2d.py:249-267) are ordinary key slots whose row offset is
// VIL_ZERO_OFF -- no branch and no zero row in memory.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc_n(const void* p, unsigned nbytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, (short)0, (int)nbytes, 0x00020000);
}
#define VIL_ZERO_OFF 0x7fff0000
template <typename T>
__device__ __forceinline__ typename V16<T>::x8 buf_load8(__amdgpu_buffer_rsrc_t r, int byte_off) {
  return __builtin_bit_cast(typename V16<T>::x8, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// ---- transposed LDS reads next to LDS-DMA.  __builtin_amdgcn_ds_read_tr16_b64 is modelled as an LDS access that may
// WRITE, so with buffer_load ... lds requests in flight the compiler puts s_waitcnt vmcnt(0) in front of it: the block
// a kernel has just requested must land before the current one can be consumed, and the prefetch is gone (ordinary
// ds_read_b128 loads get no such wait).  These helpers issue the read as inline assembly; completion is the caller's:
// lds_tr_settle() = s_waitcnt lgkmcnt(0) carrying the destination registers as operands, so no consumer moves above it.
// (lgkmcnt(0) and not a counted wait: scalar loads share the counter and return out of order.)
__device__ __forceinline__ unsigned lds_addr32(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
}
template <int OFF> __device__ __forceinline__ s16x4 lds_tr_issue(unsigned addr) {
  s16x4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
__device__ __forceinline__ void lds_tr_settle(s16x4& a, s16x4& b, s16x4& c, s16x4& d, s16x4& e, s16x4& f, s16x4& g, s16x4& h) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
}
__device__ __forceinline__ void lds_tr_settle(s16x4& a, s16x4& b, s16x4& c, s16x4& d) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
template <typename T> __device__ __forceinline__ typename V16<T>::x8 lds_tr_join(s16x4 lo, s16x4 hi) {
  typedef typename V16<T>::x4 X4;
  const X4 l = __builtin_bit_cast(X4, lo), h = __builtin_bit_cast(X4, hi);
  typename V16<T>::x8 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) { r[e] = l[e]; r[4 + e] = h[e]; }
  return r;
}

// Completion of this wave's LDS-DMA requests (buffer_load ... lds) is counted by vmcnt like any vector load, but the
// workgroup-scope fence inside __syncthreads() does not wait for it (hipcc 7.2 puts s_waitcnt vmcnt(0) in front of
// s_barrier only where another dependency happens to ask for it -- k_skinny's K = 384 instantiation had none in its tile
// loop and read a slot whose requests were, once in ~10^3 runs, still in flight).  Every barrier that publishes DMA'd
// data is therefore preceded by an explicit wait: N = requests that may stay in flight (issued AFTER the ones needed;
// loads return in order).  Not valid with stores in flight (they share vmcnt and return out of order with loads): place
// the wait before the first store.
template <int N = 0> __device__ __forceinline__ void lds_dma_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// float -> int32 for the fixed-point bias-gradient histograms: floor(x + 0.5) in ONE instruction (v_cvt_rpi_i32_f32).
// __float2int_rn is v_rndne_f32 + v_cvt_i32_f32: one VALU instruction more per score in the dQ passes, whose steps are
// ~90 vector instructions for 16 scores per lane.  Round-half-up instead of round-half-even: ties are measure-zero for
// products of bf16 probabilities and fp32 differences, and the result stays a pure function of its input (the
// histograms' bit-reproducibility does not depend on the rounding rule).
__device__ __forceinline__ int f2i_rpi(float x) {
  int i;
  asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(i) : "v"(x));
  return i;
}

__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Offset (dr, dc) of the one extra neighbour in random-shift mode: from the descriptor, or (hipGraph replay with a
// fresh draw per step) from the device word the descriptor points at (reference slidingchunk_2d.py:15-24)
__device__ __forceinline__ void shift_neighbour(const VilParams& p, int& adr1, int& adc1) {
  adr1 = p.g.adr[1]; adc1 = p.g.adc[1];
  if (p.mode_dev) {
    const int m = __builtin_amdgcn_readfirstlane(*p.mode_dev);
    const int s = m > 4 ? m : m - 1;
    adr1 = s / 3 - 1; adc1 = s - 3 * (s / 3) - 1;
  }
}

// Position of the s-th key of a chunk's list among the key slots.  interleave (the dQ pass's tables, round 6): inside every
// aligned group of 8 slots the i-th key goes to slot 4 (i & 1) + (i >> 1), so that the two lane groups of a 32-lane half
// (rows 4 lg + r of a tile) hold keys ONE apart in the list instead of four.  Consecutive keys of a chunk row differ by one
// in y; the queries of a wave differ by even amounts in y, so the two lane groups' histogram bins (bin = query term - key
// term) can no longer coincide: same-address LDS atomics cost 8.5 instead of 5.0 CU-cycles per wave instruction
// (tools/ubench/lds_atomic.hip), and the bias-gradient histogram is one ds_add per score of that pass.
__device__ __forceinline__ int key_slot_pos(int s, bool interleave) {
  return interleave ? ((s & ~7) | (((s & 1) << 2) | ((s >> 1) & 3))) : s;
}

// Key-slot table of query chunk (cm,cn) in the wave's private LDS:
//   s_koff[s] = byte offset (token * row stride) of key slot s inside the (image, head) K/V slice
//   s_akey[s] = 4 * (Ak - aconst)   (padding slots: -4*guard0; global slot g: -4*(glo0 + g*gsz))
// Slots are COMPACTED: [0,G) global tokens, then only the keys the chunk may attend (neighbour by
// neighbour, row-major), then masked padding up to a multiple of 32.  Out-of-image neighbours and
// zero-padded rows/columns therefore cost no MFMA/softmax work at all (border chunks of a 4x4 chunk
// grid see 4-6 of 9 neighbours).  Rows of the neighbourhood are distributed over lanes (one validity
// test per row); a wave prefix sum places each row's keys.  Returns the padded slot count.
__device__ __forceinline__ int build_key_slots(const VilParams& p, const MfmaCfg& c, int cm, int cn, int lane,
                                               int row_stride_b, int* s_koff, int* s_akey, int adr1, int adc1,
                                               bool own_first = false, bool interleave = false,
                                               unsigned long long perm = 0ull) {
  const VilGeom& g = p.g;
  const int W = g.W;
  const bool cyc = g.exact == -1;
  for (int s = lane; s < p.G; s += 64) {
    s_koff[key_slot_pos(s, interleave)] = __mul24(s, row_stride_b); s_akey[key_slot_pos(s, interleave)] = -(c.glo0 + s * c.gsz) * 4;
  }
  const int nrows = p.only_glo ? 0 : g.nact * W;             // only_glo: the G global keys are the whole key set
  int base = p.G;                                            // wave-uniform running slot offset
  for (int r0 = 0; r0 < nrows; r0 += 64) {
    const int rid = r0 + lane;
    // a neighbourhood row contributes na slots of one kind (real tokens, or -- cyclic padding only -- zero keys),
    // then nb zero-key slots (cyclic: its columns beyond the image, unless this neighbour is reached without wrap)
    int na = 0, nb = 0, off = 0, ak = 0;
    bool real = true;
    if (rid < nrows) {
      const int a_ = fdiv(rid, c.magicW), xt = rid - a_ * W;
      // own_first (chunk-workgroup family): the query chunk's own keys lead the list (neighbour 4 of the 3x3 order), so
      // that the global query column riding in the pass is live in the first steps only
      // perm (chunk-workgroup family, two chunks per workgroup): nibble i = the 3x3 neighbour at list position i, chosen so
      // that the neighbours the two chunks share sit at the SAME positions of both lists (k_cw_prep)
      const int a = g.nact != 9 ? a_ : perm ? (int)((perm >> (4 * a_)) & 15ull)
                                     : own_first ? (a_ == 0 ? 4 : (a_ <= 4 ? a_ - 1 : a_)) : a_;
      const int a3 = (a * 11) >> 5;                           // a / 3 for a in [0, 9)
      const int dr = g.nact == 9 ? a3 - 1 : (a == 0 ? 0 : adr1);
      const int dc = g.nact == 9 ? a - 3 * a3 - 1 : (a == 0 ? 0 : adc1);
      int rm = cm + dr, rn = cn + dc;
      ak = ((dr * W + xt) * c.P + dc * W - c.aconst) * 4;
      if (!cyc) {
        const int kr = rm * W + xt;
        if (rm >= 0 && rm < g.mx && rn >= 0 && rn < g.my && kr < g.nx) {
          const int kc0 = rn * W;
          na = min(W, g.ny - kc0);
          off = __mul24(p.G + kr * g.ny + kc0, row_stride_b);
        }
      } else {
        // reference _get_invalid_locations_mask_cyclic: a padded row / column is masked only when its chunk is the
        // last one reached WITHOUT wrap-around (cm + dr + 1 == mx); reached by wrap it is a zero key
        const bool last_r = rm + 1 == g.mx, last_c = rn + 1 == g.my;
        rm = rm < 0 ? rm + g.mx : (rm >= g.mx ? rm - g.mx : rm);
        rn = rn < 0 ? rn + g.my : (rn >= g.my ? rn - g.my : rn);
        const int kr = rm * W + xt, kc0 = rn * W;
        const bool row_pad = kr >= g.nx;
        if (!(row_pad && last_r)) {
          na = min(W, max(g.ny - kc0, 0));
          nb = last_c ? 0 : W - na;
          real = !row_pad;
          off = real ? __mul24(p.G + kr * g.ny + kc0, row_stride_b) : VIL_ZERO_OFF;
        }
      }
    }
    const int nvalid = na + nb;
    int incl = nvalid;                                        // inclusive prefix sum over the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    int s = base + incl - nvalid;
    for (int yt = 0; yt < na; ++yt) {
      s_koff[key_slot_pos(s, interleave)] = off; s_akey[key_slot_pos(s, interleave)] = ak;
      ++s; off += real ? row_stride_b : 0; ak += 4;
    }
    for (int yt = 0; yt < nb; ++yt) {
      s_koff[key_slot_pos(s, interleave)] = VIL_ZERO_OFF; s_akey[key_slot_pos(s, interleave)] = ak;
      ++s; ak += 4;
    }
    base += __shfl(incl, 63, 64);
  }
  const int total = base, padded = (total + 31) & ~31;
  const int own_off = __mul24(p.G + (cm * W) * g.ny + cn * W, row_stride_b);   // always a real token
  for (int s = total + lane; s < padded; s += 64) { s_koff[key_slot_pos(s, interleave)] = own_off; s_akey[key_slot_pos(s, interleave)] = -c.guard0 * 4; }
  wave_lds_fence();
  return __builtin_amdgcn_readfirstlane(padded);
}

// The key-slot table depends on the chunk position only, not on the (image, head): the prologue kernel builds the table
// of every chunk once per call (key_slots_block: one wave each, build_key_slots into LDS, copied out), and a forward / dQ wave
// fetches its chunk's table with a few independent 8-byte loads.  Each wave used to run build_key_slots itself:
// ~800 instructions and a wave prefix sum per (image, head, chunk) -- 10-20 % of a wave's lifetime (tools/kv_timing.py).
__device__ __forceinline__ int load_key_slots(const MfmaCfg& c, int ch, int lane, int* s_koff, int* s_akey) {
  const int nslots = __builtin_amdgcn_readfirstlane(c.key_nslots[ch]);
  const int2* src = c.key_slots + (int64_t)ch * c.NSP;
  for (int s0 = 0; s0 < nslots; s0 += 512) {
    int2 e[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) e[u] = src[min(s0 + u * 64 + lane, nslots - 1)];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int s = s0 + u * 64 + lane;
      if (s < nslots) { s_koff[s] = e[u].x; s_akey[s] = e[u].y; }
    }
  }
  wave_lds_fence();
  return nslots;
}
// One element e of head h's LDS-image bias table: bias * inv (backward passes: inv = 1 / scale, so that one multiply by
// scale*log2e serves scores and bias alike; forward: inv = log2(e), the scores of its pre-scaled Q are log2-domain; masks and the exact window as VIL_MASK_VAL; one constant region per global token)
__device__ __forceinline__ void table_element(const VilParams& p, const MfmaCfg& c, float* out, int h, int e, float inv) {
  if (e >= c.tabsize) return;
  const int tbl = c.trows, W = p.g.W;
  float v = 0.f;
  if (e < tbl * c.P) {
    const int row = e / c.P, col = e % c.P - VIL_CPAD;
    if (col >= 0 && col < tbl) {
      const int dx = row - c.tcen, dy = col - c.tcen;
      const int o = p.bias_off, S = p.bias_S;      // the caller's table covers |dx|,|dy| <= o
      if (p.has_bias && dx >= -o && dx <= o && dy >= -o && dy <= o)
        v = p.table[(int64_t)((dx + o) * S + (dy + o)) * p.H + h] * inv;
      if (p.g.exact == 1 && (dx > W || dx < -W || dy > W || dy < -W)) v = VIL_MASK_VAL;
    }
  } else if (e < c.glo0) {
    v = VIL_MASK_VAL;
  } else {
    const int g = (e - c.glo0) / c.gsz;
    if (g < p.G && p.has_g2l) v = p.g2l[h * p.G + g] * inv;
  }
  out[(int64_t)h * c.tabstride + e] = v;
}
// Key-slot table of query chunk ch, built by ONE wave in `smem` (NSP * 8 bytes) and copied to c.key_slots
__device__ __forceinline__ void key_slots_block(const VilParams& p, const MfmaCfg& c, int ch, int lane, int row_stride_b,
                                                char* smem, bool interleave = false) {
  const int cn = ch % p.g.my, cm = ch / p.g.my;
  int* s_koff = (int*)smem;
  int* s_akey = s_koff + c.NSP;
  int adr1, adc1;
  shift_neighbour(p, adr1, adc1);
  const int nslots = build_key_slots(p, c, cm, cn, lane, row_stride_b, s_koff, s_akey, adr1, adc1, false, interleave);
  int2* out = c.key_slots + (int64_t)ch * c.NSP;
  for (int s = lane; s < nslots; s += 64) out[s] = make_int2(s_koff[s], s_akey[s]);
  if (lane == 0) c.key_nslots[ch] = nslots;
}
// Forward prologue, ONE launch of 256-thread workgroups with two roles: [0, ntx*H) bias-table images, then one
// workgroup (its first wave) per query chunk for the key-slot tables.  (Separate launches cost ~4 us each under
// hipGraph replay, 24 attention calls per ViL-Small step.)
__global__ void k_mfma_prep(VilParams p, MfmaCfg c, int row_stride_b, int ntx);
// floats of workspace behind the bias tables for the key-slot tables (16-byte multiple)
static inline size_t vil_key_slots_floats(const MfmaCfg& c, int nch) {
  return (size_t)nch * c.NSP * 2 + (((size_t)nch + 3) & ~(size_t)3);
}
// workgroups of a persistent launch: as many as are resident at once (waves per SIMD of the kernel's launch bounds, its
// LDS footprint), a multiple of 8 * H so that every (XCD, head) queue is served by the same number of workgroups, and no
// more than the units can feed
int vil_persistent_grid(int waves_per_simd, int waves_per_wg, size_t lds, int H, int64_t units_total);

// ---- chunk-workgroup family (vil_attn_cw.hip): a chunk is NWP waves of 16 * QT query slots; a workgroup is one COLUMN --
// a head and NCH chunks in lockstep -- bound to an XCD, streaming over that XCD's images
struct CwCfg {
  int QT;              // 16-column query tiles per wave (2 or 4)
  int HQ;              // query groups (y = QT yq .. QT yq + QT - 1) per chunk row = ceil(W / QT)
  int NWP;             // waves per chunk = ceil(W * HQ / 16)
  int NCH;             // chunks per workgroup
  int HPW, NHG;        // heads per workgroup (H at head_dim 32: the halves of a K / V line meet in the CU's L1), head groups = H / HPW
  int nch, ngrp;       // chunks / chunk groups per (image, head)
  int lpt;             // chunks taken in order of decreasing work (chunk_of_rank): groups of equal step counts
  int NSP;
  int by_image;        // B >= 8: XCD x walks images x, x + 8, ...; else (image, head) pairs are dealt to the XCDs, one image per workgroup
  int ncolx;           // columns of one XCD's list: H * ngrp (by_image) or ceil(B H / 8) * ngrp
  int NS;              // image streams per column (stream s of XCD x: images x + 8 s, x + 8 (s + NS), ...): the largest of seg_ns
  // by_image: an XCD's workgroup list is nseg segments of chunk groups with (about) equal work per image -- interior, edge and
  // corner chunks of the 3x3 neighbourhood -- each with its own number of image streams, so that every workgroup of the
  // launch has about the same work whatever its chunks' neighbour count: segment c = groups [seg_g0[c], seg_g0[c] + seg_ng[c])
  // x NHG head groups x seg_ns[c] streams, workgroups [seg_wg0[c], seg_wg0[c + 1]) of the XCD's nwgx
  int nseg, nwgx;
  int seg_g0[4], seg_ng[4], seg_ns[4], seg_wg0[4];
  unsigned seg_m[4];   // vil_magic(seg_ng[c] * NHG)
  int koff_lds, ak_lds;   // LDS bytes of a chunk's two slot tables (whole 1 KB DMA pieces)
  int abl;             // timing-ablation bits (VIL_CW_ABLATE builds only)
  void* dbg;           // cycle-stamp records (VIL_CW_ABLATE builds only)
  int akb;             // bias of the packed 16-bit address terms (bytes, multiple of 4)
  int gq_on, gq_wp, gq_lj;     // vil_attn_fwd_full: the global query's column (wave part, lane column)
  unsigned m_HQ, m_NWP, m_ncolx, m_ngrp, m_HPW, m_NHG, m_tab4;
  int* koff;           // (nch, NSP) K / V row byte offset of every key slot (own chunk's keys first)
  int* akey;           // (nch, NSP) halfwords: bias address term + akb of every key slot, in per-lane order (k_cw_prep)
  int* nslots;         // (nch) padded slot count
  int* nown;           // (nch) key slots of the chunk's own keys (slots [G, G + nown))
  unsigned* redo;      // (workgroups) images the fast kernel's workgroup hands to the exact kernel (bit j: its j-th image)
};
int vil_cw_supported(const VilAttnDesc* d, int pass);
size_t vil_cw_workspace(const VilAttnDesc* d, int pass);
int vil_cw_fwd(const VilAttnDesc* d, VilParams& p, hipStream_t s);

// host: fills the launch configuration for a descriptor
bool vil_mfma_make_cfg(const VilAttnDesc* d, MfmaCfg& c);

// ---- backward configuration (vil_attn_mfma_bwd.hip; the table / slot-table fields are shared with the fp32 family)
#define VIL_NORM_SLOTS 64
struct BwdCfg {
  int nch;            // query/key chunks per (image, head) = mx*my
  int nsplit;         // global-key owner units per (image, head)
  int glo_from_dq;    // G <= 4: dK/dV of the global keys are a by-product of the dQ pass (one owner unit, streaming nothing)
  int glo_nrec;       // partial records per (image, head) in glo_parts: dq units + 1, or nsplit
  int units_kv_bh;    // nch*NWP + (G ? nsplit : 0)
  int kv_wg_per_bh, kv_gpw, kv_wpw;
  int dq_QT, dq_HQ, dq_NWP, dq_units_bh, dq_wg_per_bh, dq_gpw, dq_wpw;   // dQ pass: query tiles per wave, ...
  int kv_KT, kv_HQ, kv_NWP;   // dK/dV pass: key tiles per wave, key quads (pairs) per chunk row, waves per chunk
  int nqs;            // streamed query slots per owner unit (padded to 32)
  int kv_wave_lds, dq_wave_lds;
  int do_hist;
  float* hist_parts;  // (dq workgroups, tabsize) int32
  float* glo_parts;   // (B*H, glo_nrec, G, 2, M)
  float* gq_parts;    // (B*H, nch*NWP + 1, G, M + 4): per-unit partial dq of the global QUERY rows, [M] = sum of dS
  int dq_nwg;
  int hist_nmax;      // upper bound of contributions one histogram bin can receive in one workgroup
  unsigned* norm2;    // VIL_NORM_SLOTS x 32 words; slot k: [0] max ||dO_q||^2, [1] max ||v_k||^2 as float bits
                      // (partial maxima written by k_mfma_delta); word 2 of slot 0: the histogram scale lfx
  unsigned m_dq_wgbh, m_dq_NWP, m_dq_HQ, m_kv_wgbh, m_kv_NWP, m_kv_HQ;   // magic reciprocals (vil_magic, fdiv)
  int2* kv_slots;     // (nch + nsplit, nqs): streamed-query slot tables of the dK/dV pass (kv_slots_block)
  int* kv_nchunks;    // (nch + nsplit)
  UnitQueue uq_dq, uq_kv;   // units of the two passes (persistent workgroups, see UnitQueue)
  int kv_nwg;         // workgroups of the dK/dV launch
  int hist_flush;     // dQ pass: the LDS histogram is drained into the workgroup's 64-bit bins every hist_flush started units
  int kv_gspare;      // dK/dV pass, G == 1: the global key rides in a spare key column of every chunk's last wave (no owner units)
  int kv_gjj;         // ... that column's (x, hq) pair index within the chunk (= W * kv_HQ, the first unused pair)
  int gq_nrec;        // records per (image, head) in gq_parts: nch * kv_NWP (+ 1: the owner unit's, without kv_gspare)
  // vil_attn_bwd_full: the G global QUERY rows are the last G slots of every unit's query stream (round 5).  Their bias
  // (g2l[0][h][g] against local keys, g2g[h][gq][gk] against global keys) comes out of constant regions the dK/dV
  // workgroups append to their LDS copy of the bias image:  [c.tabsize | G x gsz: g2l0 | G*G x gsz: g2g]
  int kv_xsize;       // floats appended (0 without global-query rows)
  int nqs_own;        // slots of a chunk's own unit (nact * W^2 + global rows, padded to 32) <= nqs
  int kv_epre;        // straight-line slot rounds of the dK/dV prologue (7 or 10: the kernel's template parameter)
  int kv_span;        // largest key address term (W-1)*P + KT*HQ - 1: a global-query slot's address term is region + span
};

struct PrepZero { unsigned* ptr[5]; int n[5]; int total; };
// prologue launches (bias-table images + slot tables + zero fill) as host functions callable from the other kernel files
int vil_mfma_launch_prep(const VilParams& p, const MfmaCfg& c, int row_stride_b, hipStream_t s);
int vil_mfma_launch_prep_bwd(const VilParams& p, const MfmaCfg& c, const BwdCfg& bc, int row_stride_b, const PrepZero& zr, hipStream_t s);
// the fp32 matrix-core family (vil_attn_mfma_f32.hip: v_mfma_f32_16x16x4_f32)
int vil_f32_supported(const VilAttnDesc* d, int pass);
size_t vil_f32_workspace(const VilAttnDesc* d, int pass);
int vil_f32_fwd(const VilAttnDesc* d, VilParams& p, hipStream_t s);
int vil_f32_bwd(const VilAttnDesc* d, VilParams& p, hipStream_t s);


