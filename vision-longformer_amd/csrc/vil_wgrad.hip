// vil_wgrad.hip -- fused weight + bias gradient of the projections around the hot path:
//     dW[co][ci] = sum_t dY[t][co] * X[t][ci],     db[co] = sum_t dY[t][co]
// (autograd of nn.Linear; reference msvit.py qkv / proj / Mlp.fc1 / fc2, longformer2d.py query / kv / proj).
// The contraction runs over B*N = 6 k ... 400 k tokens into a tiny (C_out x C_in) output: a library
// GEMM launches C_out*C_in/128^2 = 1..36 workgroups unless the caller splits K by hand (bmm + sum:
// 110-330 TFLOP/s measured on MI355X), and the bias gradient is a second pass over dY.
// Here: grid = output tiles x token splits; a workgroup (4 waves, 128 x 128 tile, one 64 x 64 quadrant per
// wave) streams its token slice 32 rows at a time through a double-buffered LDS tile (row-major, as in HBM:
// 16-byte coalesced loads), both MFMA operands are read TRANSPOSED out of LDS with ds_read_b64_tr_b16
// (rows are the contraction index), fp32 partials per split, then one reduce pass.  db rides on the
// A fragments already in registers.
#include "vil_internal.h"
#include <stdlib.h>

typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wg_bf16x4 __attribute__((ext_vector_type(4)));
typedef short wg_s16x4 __attribute__((ext_vector_type(4)));
typedef float wg_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned wg_u32x4 __attribute__((ext_vector_type(4)));

#define WG_TILE 128
#define WG_ROWS 32
#define WG_PITCH (WG_TILE * 2 + 32)        // bytes; +32: the 4 rows of a transposed read hit distinct banks
#define WG_TILE_BYTES (WG_ROWS * WG_PITCH)

struct WgradParams {
  const __bf16* dy; const __bf16* x;
  int64_t T, sdy, sx;
  int CO, CI, tiles_ci, tiles, nsplit;
  int64_t rows_per_split;
  float* parts;      // (nsplit, CO, CI)
  float* dbparts;    // (nsplit, CO) or null
  void* dw; void* db; int out_bf16;
};

__device__ __forceinline__ wg_bf16x8 wg_tr8(const char* tile, int off0, int off1) {
  const wg_bf16x4 lo = __builtin_bit_cast(wg_bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (wg_s16x4 __attribute__((address_space(3)))*)(tile + off0)));
  const wg_bf16x4 hi = __builtin_bit_cast(wg_bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (wg_s16x4 __attribute__((address_space(3)))*)(tile + off1)));
  wg_bf16x8 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) { r[e] = lo[e]; r[4 + e] = hi[e]; }
  return r;
}

__global__ __launch_bounds__(256, 2) void k_wgrad(WgradParams p) {
  __shared__ __attribute__((aligned(16))) char smem[4 * WG_TILE_BYTES];   // [stage][A | B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lj = lane & 15, lg = lane >> 4;
  // XCD-aware: the hardware deals consecutive workgroups round-robin over the 8 XCDs (8 private L2s).
  // All output tiles of one token slice re-read the same dY / X rows, so a slice lives on ONE XCD
  // (split = xcd + 8 * ...) and its tiles are dispatched back to back: the re-reads hit that L2
  // instead of fetching the slice from HBM once per XCD (measured: 3-12x the algorithmic bytes).
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int split = xcd + 8 * (idx / p.tiles), tile = idx % p.tiles;
  if (split >= p.nsplit) return;
  const int tco = tile / p.tiles_ci, tci = tile % p.tiles_ci;
  const int co0 = tco * WG_TILE, ci0 = tci * WG_TILE;
  const int ch = wave >> 1, cw = wave & 1;
  const int64_t t_begin = (int64_t)split * p.rows_per_split;
  const int64_t t_end = min(p.T, t_begin + p.rows_per_split);
  const int nsteps = (int)((t_end - t_begin + WG_ROWS - 1) / WG_ROWS);

  // staging: 2 x 16-byte chunks of each tile per thread
  int ld_row[2], ld_col[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) { const int c = tid + 256 * u; ld_row[u] = c >> 4; ld_col[u] = (c & 15) * 8; }
  // two steps of global loads in flight per workgroup (register sets 0 / 1): with one, a step's compute
  // (~0.2 us) had to cover a full HBM round trip and the kernel ran latency-bound at ~1 us per step
  wg_u32x4 ra[2][2], rb[2][2];
  auto gload = [&](int st, wg_u32x4 (&qa)[2], wg_u32x4 (&qb)[2]) {
    const int64_t t0 = t_begin + (int64_t)st * WG_ROWS;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int64_t t = t0 + ld_row[u];
      const wg_u32x4 z = {0u, 0u, 0u, 0u};
      qa[u] = (t < t_end && co0 + ld_col[u] < p.CO) ? *(const wg_u32x4*)(p.dy + t * p.sdy + co0 + ld_col[u]) : z;
      qb[u] = (t < t_end && ci0 + ld_col[u] < p.CI) ? *(const wg_u32x4*)(p.x + t * p.sx + ci0 + ld_col[u]) : z;
    }
  };
  auto sstore = [&](int stage, const wg_u32x4 (&qa)[2], const wg_u32x4 (&qb)[2]) {
    char* A = smem + stage * 2 * WG_TILE_BYTES;
    char* B = A + WG_TILE_BYTES;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      *(wg_u32x4*)(A + ld_row[u] * WG_PITCH + ld_col[u] * 2) = qa[u];
      *(wg_u32x4*)(B + ld_row[u] * WG_PITCH + ld_col[u] * 2) = qb[u];
    }
  };
  // transposed-read offsets: lane (lj, lg) gets rows {4*lg + e} (first read) and {16 + 4*lg + e} (second)
  // of column 16*i + lj of its 64-column half
  const int tr_row0 = (lg * 4 + (lj >> 2)) * WG_PITCH + (lj & 3) * 8;
  const int tr_row1 = tr_row0 + 16 * WG_PITCH;
  const int a_col = ch * 64 * 2, b_col = cw * 64 * 2;      // bytes

  wg_f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (wg_f32x4){0.f, 0.f, 0.f, 0.f};
  float dbs[4] = {0.f, 0.f, 0.f, 0.f};
  const bool do_db = p.dbparts != nullptr && tci == 0 && cw == 0;

  auto compute = [&](int stage) {
    const char* A = smem + stage * 2 * WG_TILE_BYTES;
    const char* B = A + WG_TILE_BYTES;
    wg_bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a[i] = wg_tr8(A, tr_row0 + a_col + i * 32, tr_row1 + a_col + i * 32);
      b[i] = wg_tr8(B, tr_row0 + b_col + i * 32, tr_row1 + b_col + i * 32);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    if (do_db) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) dbs[i] += (float)a[i][e];
    }
  };
  // step st computes from LDS stage st&1, stores step st+1 (register set (st+1)&1) and issues the loads of
  // step st+2 into the register set it has just freed
  if (nsteps > 0) gload(0, ra[0], rb[0]);
  if (nsteps > 1) gload(1, ra[1], rb[1]);
  if (nsteps > 0) sstore(0, ra[0], rb[0]);
  __syncthreads();
  for (int st = 0; st < nsteps; st += 2) {
    if (st + 2 < nsteps) gload(st + 2, ra[0], rb[0]);
    compute(0);
    if (st + 1 < nsteps) sstore(1, ra[1], rb[1]);
    __syncthreads();
    if (st + 1 >= nsteps) break;
    if (st + 3 < nsteps) gload(st + 3, ra[1], rb[1]);
    compute(1);
    if (st + 2 < nsteps) sstore(0, ra[0], rb[0]);
    __syncthreads();
  }

  float* out = p.parts + (int64_t)split * p.CO * p.CI;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = co0 + ch * 64 + i * 16 + lg * 4 + r;
      if (co < p.CO) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int ci = ci0 + cw * 64 + j * 16 + lj;
          if (ci < p.CI) out[(int64_t)co * p.CI + ci] = acc[i][j][r];
        }
      }
    }
  if (do_db) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float s = dbs[i];
      s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
      const int co = co0 + ch * 64 + i * 16 + lj;
      if (lg == 0 && co < p.CO) p.dbparts[(int64_t)split * p.CO + co] = s;
    }
  }
}

// dW / db = sum over the splits' partials: block (64, 4): 4 consecutive elements per thread (16-byte loads),
// the splits dealt over the 4 thread groups (four loads in flight each), LDS sum.  (CO*CI % 64 == 0.)
__global__ __launch_bounds__(256) void k_wgrad_reduce(WgradParams p) {
  __shared__ wg_f32x4 red[4][64];
  const int64_t n = (int64_t)p.CO * p.CI, n4 = n >> 2;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + tx;
  wg_f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (i < n4) {
    const wg_f32x4* src = (const wg_f32x4*)p.parts + i;
    int k = ty;
    for (; k + 12 < p.nsplit; k += 16) {
      const wg_f32x4 v0 = src[(int64_t)k * n4], v1 = src[(int64_t)(k + 4) * n4];
      const wg_f32x4 v2 = src[(int64_t)(k + 8) * n4], v3 = src[(int64_t)(k + 12) * n4];
      s += (v0 + v1) + (v2 + v3);
    }
    for (; k < p.nsplit; k += 4) s += src[(int64_t)k * n4];
  } else if (p.dbparts && i < n4 + p.CO) {
    const int c = (int)(i - n4);
    for (int k = ty; k < p.nsplit; k += 4) s[0] += p.dbparts[(int64_t)k * p.CO + c];
  }
  red[ty][tx] = s;
  __syncthreads();
  if (ty != 0) return;
  s = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
  if (i < n4) {
    if (p.out_bf16) {
      wg_bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (__bf16)s[e];
      ((wg_bf16x4*)p.dw)[i] = o;
    } else ((wg_f32x4*)p.dw)[i] = s;
  } else if (p.dbparts && i < n4 + p.CO) {
    const int c = (int)(i - n4);
    if (p.out_bf16) ((vil_bf16*)p.db)[c] = vil_f2bf(s[0]); else ((float*)p.db)[c] = s[0];
  }
}


// =====================================================================================================================
// Second generation (channel counts that are multiples of 96: every ViL-Small / Medium / Medium-Deep layer).
//   * tile 32 MI x 32 NJ (MI, NJ in {3, 6}: 96 or 192 output rows / columns), 2 x 2 waves, wave tile 16 MI x 16 NJ:
//     192 x 192 divides 384 / 768 / 1152 / 1536 / 2304 / 3072 exactly, so tiles x slices can fill 64 workgroup slots of
//     an XCD in ONE round (the 128 x 128 tiling gave 36 tiles x 16 slices = 576 workgroups on 512 slots for stage 3's
//     fc layers: two rounds, the second 12 % full), and a wave reads 341 instead of 512 LDS bytes per MFMA;
//   * the token rows arrive by LDS-DMA (buffer_load ... lds, no staging registers, no ds_write) into a ring of NST
//     stages of 32 rows; the wait in front of the step barrier is s_waitcnt vmcnt(DPW (NST - 2)), not the vmcnt(0)
//     __syncthreads() brings, so NST - 2 stages stay in flight across the barrier.  Rows past the slice's end and the
//     stages past its last step are out of the descriptor's range: they arrive as zeros and keep the count exact;
//   * LDS image of a stage: rows at their natural pitch (384 / 192 bytes), the 32-byte granules of a row XOR-swizzled
//     with the row number so that the four rows a ds_read_b64_tr_b16 group touches fall in four different bank
//     groups; the swizzle is applied to the SOURCE address of each DMA lane (the LDS side of a DMA is linear);
//   * partial records are written in accumulator order (1 KB contiguous per store instruction); the reduce pass
//     un-permutes; db: column sums on the VALU, the 16-row blocks dealt over the waves that hold the same dY rows.
#include "vil_mfma_common.h"
#include <type_traits>

#ifndef WG2_SWZ_OLD
// (round 4: + bit 2 of the row.  Rows r and r + 4 of a stage start on the same bank -- 4 x 192 and 4 x 384 bytes are
// multiples of 256 -- and the two 16-lane groups of a half wave read exactly such a pair: SQ_LDS_BANK_CONFLICT was 50 % of
// SQ_LDS_IDX_ACTIVE for both operand widths.  With the extra term the 8 rows x 32 bytes of a half wave cover 8 different
// 32-byte bank groups, and the 4 rows of a 16-lane group still cover 4.)
template <int W> __device__ __forceinline__ int wg2_swz(int row) {
  return W == 192 ? ((row & 3) ^ ((row >> 2) & 1)) : (((row >> 1) & 1) ^ ((row >> 2) & 1));
}
#else
template <int W> __device__ __forceinline__ int wg2_swz(int row) { return W == 192 ? (row & 3) : ((row >> 1) & 1); }
#endif

__device__ __forceinline__ unsigned wg2_lds_addr(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
}
template <int OFF> __device__ __forceinline__ wg_s16x4 wg2_tr(unsigned addr) {
  wg_s16x4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
// s_waitcnt lgkmcnt(N) that the listed fragments depend on (so that no consumer is scheduled above it)
template <int N> __device__ __forceinline__ void wg2_settle(wg_s16x4& a, wg_s16x4& b) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}
template <int N> __device__ __forceinline__ void wg2_settle(wg_s16x4& a, wg_s16x4& b, wg_s16x4& c, wg_s16x4& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}
template <int N> __device__ __forceinline__ void wg2_settle(wg_s16x4& a, wg_s16x4& b, wg_s16x4& c, wg_s16x4& d, wg_s16x4& e, wg_s16x4& f) {
  asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "n"(N));
}
template <int N> __device__ __forceinline__ void wg2_settle(wg_s16x4& a, wg_s16x4& b, wg_s16x4& c, wg_s16x4& d, wg_s16x4& e, wg_s16x4& f,
                                                            wg_s16x4& g, wg_s16x4& h, wg_s16x4& i, wg_s16x4& j, wg_s16x4& k, wg_s16x4& l) {
  asm volatile("s_waitcnt lgkmcnt(%12)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(i), "+v"(j), "+v"(k), "+v"(l) : "n"(N));
}
__device__ __forceinline__ wg_bf16x8 wg2_join(wg_s16x4 lo, wg_s16x4 hi) {
  const wg_bf16x4 l = __builtin_bit_cast(wg_bf16x4, lo), h = __builtin_bit_cast(wg_bf16x4, hi);
  wg_bf16x8 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) { r[e] = l[e]; r[4 + e] = h[e]; }
  return r;
}

template <int N> __device__ __forceinline__ void wg2_wait_barrier() {
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

template <int MI, int NJ, int NST>
__global__ __launch_bounds__(256, 2) void k_wgrad2(WgradParams p) {
  constexpr int TM = 32 * MI, TN = 32 * NJ, PA = TM * 2, PB = TN * 2, SR = 32;
  constexpr int ABYTES = SR * PA, BBYTES = SR * PB, STAGE = ABYTES + BBYTES;
  constexpr int NW = 4;
  constexpr int NA = ABYTES / 1024, NB = BBYTES / 1024;            // DMA instructions per region and stage
  constexpr int UA = (NA + NW - 1) / NW, UB = (NB + NW - 1) / NW, DPW = UA + UB;
  extern __shared__ __attribute__((aligned(16))) char wsm[];       // NST stages | 1 KB scratch (padding DMAs)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int w4 = wave, ch = w4 >> 1, cw = w4 & 1;
  const int lj = lane & 15, lg = lane >> 4;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int split = xcd + 8 * (idx / p.tiles), tile = idx % p.tiles;
  const int tco = tile / p.tiles_ci, tci = tile % p.tiles_ci;
  const int co0 = tco * TM, ci0 = tci * TN;
  const int64_t t_begin = min(p.T, (int64_t)split * p.rows_per_split);
  const int rows = (int)(min(p.T, t_begin + p.rows_per_split) - t_begin);
  const int nsteps = (rows + SR - 1) / SR;
  const int sdy2 = (int)p.sdy * 2, sx2 = (int)p.sx * 2;
  const __amdgpu_buffer_rsrc_t rsa = make_rsrc_n(p.dy + t_begin * p.sdy + co0, rows > 0 ? (unsigned)((rows - 1) * sdy2 + PA) : 0u);
  const __amdgpu_buffer_rsrc_t rsb = make_rsrc_n(p.x + t_begin * p.sx + ci0, rows > 0 ? (unsigned)((rows - 1) * sx2 + PB) : 0u);

  // DMA lane constants: instruction q of a region fills LDS bytes [1024 q, 1024 q + 1024) of the region
  static_assert(UA <= 3 && UB <= 3, "request arrays");
  int va[3], vb[3], la[3], lb[3];      // (fixed bounds: arrays of template-dependent size are rejected as builtin arguments inside the lambda)
#pragma unroll
  for (int u = 0; u < UA; ++u) {
    const int q = wave + NW * u, P = q * 64 + lane, row = P / (TM / 8), cpos = P % (TM / 8);
    const int g = (cpos >> 1) ^ wg2_swz<TM>(row);
    va[u] = q < NA ? row * sdy2 + (g * 2 + (cpos & 1)) * 16 : 0x7fff0000;
    la[u] = q < NA ? q * 1024 : -1;
  }
#pragma unroll
  for (int u = 0; u < UB; ++u) {
    const int q = wave + NW * u, P = q * 64 + lane, row = P / (TN / 8), cpos = P % (TN / 8);
    const int g = (cpos >> 1) ^ wg2_swz<TN>(row);
    vb[u] = q < NB ? row * sx2 + (g * 2 + (cpos & 1)) * 16 : 0x7fff0000;
    lb[u] = q < NB ? ABYTES + q * 1024 : -1;
  }
  auto issue = [&](int st, int slot) {
    char* base = wsm + slot * STAGE;
    const int adva = st * SR * sdy2, advb = st * SR * sx2;
#pragma unroll
    for (int u = 0; u < UA; ++u)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (__attribute__((address_space(3))) void*)(la[u] >= 0 ? base + la[u] : wsm + NST * STAGE),
                                               16, va[u] + adva, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < UB; ++u)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsb, (__attribute__((address_space(3))) void*)(lb[u] >= 0 ? base + lb[u] : wsm + NST * STAGE),
                                               16, vb[u] + advb, 0, 0, 0);
  };

  // transposed reads: lane (lj, lg) points at row 4 lg + lj / 4 (+ 16), 8 bytes at column 16 i + 4 (lj % 4) of its wave's
  // range and receives rows {4 lg + e} (+ 16) of column 16 i + lj: the same k order for both operands
  int aoff[MI], boff[NJ];
  {
    const int r = lg * 4 + (lj >> 2);
#pragma unroll
    for (int i = 0; i < MI; ++i) aoff[i] = r * PA + (((ch * MI + i) ^ wg2_swz<TM>(r)) << 5) + (lj & 3) * 8;
#pragma unroll
    for (int j = 0; j < NJ; ++j) boff[j] = ABYTES + r * PB + (((cw * NJ + j) ^ wg2_swz<TN>(r)) << 5) + (lj & 3) * 8;
  }

  wg_f32x4 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (wg_f32x4){0.f, 0.f, 0.f, 0.f};
  float dbs[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) dbs[i] = 0.f;
  // the 16-row blocks of this wave's dY rows are shared by 2 tiles_ci waves of the slice: block i belongs to one of them
  const int db_mod = 2 * p.tiles_ci, db_me = p.dbparts ? tci * 2 + cw : -1;

#pragma unroll
  for (int a = 0; a < NST - 1; ++a) issue(a, a);
  // The step loop is unrolled NST times so that a step's ring slot is a compile-time constant: the slot offset rides in the
  // offset field of every transposed read and in the scalar arithmetic of the DMA requests (round 4: the runtime
  // `st % NST` cost 18 vector adds, two scalar multiply-high modulo sequences and five M0 selects per step -- 135 issued
  // instructions per 18 MFMAs, profiles/r04_pmc_k_wgrad2_6_3_3_wgrad_s3_fc1.json).
  const unsigned S0 = wg2_lds_addr(wsm);
  unsigned aad[MI], bad[NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i) aad[i] = S0 + aoff[i];
#pragma unroll
  for (int j = 0; j < NJ; ++j) bad[j] = S0 + boff[j];
  auto step = [&](int st, auto slot_c) {
    constexpr int SLOT = decltype(slot_c)::value;
    wg2_wait_barrier<DPW * (NST - 2)>();                 // stage st has landed everywhere; the slot of st - 1 is free
    issue(st + NST - 1, (SLOT + NST - 1) % NST);
    // The transposed reads are issued as inline assembly: through the builtin the compiler treats them as LDS WRITES
    // that may alias the DMA requests in flight and puts s_waitcnt vmcnt(0) in front of the first one, which
    // serialises every step behind the stage it has just requested.  Order and completion are handled here: the
    // reads return in order, the first wait releases the a fragments and half of b, the second the rest.
    constexpr int SO = SLOT * STAGE;
    wg_s16x4 al[MI], ah[MI], bl[NJ], bh[NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i) { al[i] = wg2_tr<SO>(aad[i]); ah[i] = wg2_tr<SO + 16 * PA>(aad[i]); }
#pragma unroll
    for (int j = 0; j < NJ; ++j) { bl[j] = wg2_tr<SO>(bad[j]); bh[j] = wg2_tr<SO + 16 * PB>(bad[j]); }
    constexpr int NJ0 = NJ / 2 + (NJ & 1), LATE = 2 * (NJ - NJ0);
    if constexpr (MI == 6) wg2_settle<LATE>(al[0], ah[0], al[1], ah[1], al[2], ah[2], al[MI - 3], ah[MI - 3], al[MI - 2], ah[MI - 2], al[MI - 1], ah[MI - 1]);
    else wg2_settle<LATE>(al[0], ah[0], al[1], ah[1], al[2], ah[2]);
    if constexpr (NJ0 == 3) wg2_settle<LATE>(bl[0], bh[0], bl[1], bh[1], bl[2], bh[2]);
    else wg2_settle<LATE>(bl[0], bh[0], bl[1], bh[1]);
    wg_bf16x8 a[MI], b[NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i) a[i] = wg2_join(al[i], ah[i]);
#pragma unroll
    for (int j = 0; j < NJ0; ++j) b[j] = wg2_join(bl[j], bh[j]);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ0; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);                   // keep the first half of the MFMAs above the second wait
    if constexpr (NJ - NJ0 == 3) wg2_settle<0>(bl[NJ - 3], bh[NJ - 3], bl[NJ - 2], bh[NJ - 2], bl[NJ - 1], bh[NJ - 1]);
    else wg2_settle<0>(bl[NJ - 1], bh[NJ - 1]);
#pragma unroll
    for (int j = NJ0; j < NJ; ++j) b[j] = wg2_join(bl[j], bh[j]);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = NJ0; j < NJ; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < MI; ++i)
      if (i % db_mod == db_me) {
#pragma unroll
        for (int e = 0; e < 8; ++e) dbs[i] += (float)a[i][e];
      }
  };
  static_assert(NST == 3, "the step loop is unrolled by hand");
  for (int st = 0; st < nsteps; st += NST) {
    step(st, std::integral_constant<int, 0>{});
    if (st + 1 < nsteps) step(st + 1, std::integral_constant<int, 1>{});
    if (st + 2 < nsteps) step(st + 2, std::integral_constant<int, 2>{});
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the zero-filling requests past the last step

  {
    float* out = p.parts + (int64_t)split * p.CO * p.CI + (int64_t)tile * (TM * TN) + w4 * (MI * NJ * 256) + lane;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[((i * NJ + j) * 4 + r) * 64] = acc[i][j][r];
  }
  if (db_me >= 0) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
      if (i % db_mod == db_me) {
        float s = dbs[i];
        s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
        if (lg == 0) p.dbparts[(int64_t)split * p.CO + co0 + ch * 16 * MI + i * 16 + lj] = s;
      }
  }
}

// sum of the partial records of k_wgrad2 (accumulator order) into row-major dW, and of the db records.  Block (64, NTY):
// 4 consecutive elements per thread, the splits dealt over the NTY thread groups (up to four loads in flight each).
// NTY = 16 from 32 records on: the 96 x 96 outputs of stage 1 have 36 blocks and up to 512 records, so the depth per
// thread is what counts there (33 -> 8 us); with 8-16 records of a large output 4 groups are faster (28 -> 15 us).
template <int NTY>
__global__ __launch_bounds__(64 * NTY) void k_wgrad2_reduce(WgradParams p, int MI, int NJ) {
  __shared__ wg_f32x4 red[NTY][64];
  const int64_t n = (int64_t)p.CO * p.CI, n4 = n >> 2;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + tx;
  wg_f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (i < n4) {
    const wg_f32x4* src = (const wg_f32x4*)p.parts + i;
    int k = ty;
    for (; k + 3 * NTY < p.nsplit; k += 4 * NTY) {
      const wg_f32x4 v0 = src[(int64_t)k * n4], v1 = src[(int64_t)(k + NTY) * n4];
      const wg_f32x4 v2 = src[(int64_t)(k + 2 * NTY) * n4], v3 = src[(int64_t)(k + 3 * NTY) * n4];
      s += (v0 + v1) + (v2 + v3);
    }
    for (; k < p.nsplit; k += NTY) s += src[(int64_t)k * n4];
  } else if (p.dbparts && i < n4 + p.CO) {
    const int c = (int)(i - n4);
    for (int k = ty; k < p.nsplit; k += NTY) s[0] += p.dbparts[(int64_t)k * p.CO + c];
  }
  red[ty][tx] = s;
  __syncthreads();
  if (ty != 0) return;
#pragma unroll
  for (int g = 1; g < NTY; ++g) s += red[g][tx];
  if (i < n4) {
    // float4 index -> (tile, wave, i, j, r, 4 lanes): lanes 4 q .. 4 q + 3 of a 16-lane group are 4 consecutive columns
    int e = (int)i;
    const int l4 = e & 15; e >>= 4;
    const int r = e & 3; e >>= 2;
    const int j = e % NJ; e /= NJ;
    const int ii = e % MI; e /= MI;
    const int w4 = e & 3, tile = e >> 2;
    const int tco = tile / p.tiles_ci, tci = tile % p.tiles_ci;
    const int co = tco * 32 * MI + (w4 >> 1) * 16 * MI + ii * 16 + (l4 >> 2) * 4 + r;
    const int ci = tci * 32 * NJ + (w4 & 1) * 16 * NJ + j * 16 + (l4 & 3) * 4;
    const int64_t o = (int64_t)co * p.CI + ci;
    if (p.out_bf16) {
      wg_bf16x4 ob;
#pragma unroll
      for (int q = 0; q < 4; ++q) ob[q] = (__bf16)s[q];
      *(wg_bf16x4*)((__bf16*)p.dw + o) = ob;
    } else *(wg_f32x4*)((float*)p.dw + o) = s;
  } else if (p.dbparts && i < n4 + p.CO) {
    const int c = (int)(i - n4);
    if (p.out_bf16) ((vil_bf16*)p.db)[c] = vil_f2bf(s[0]); else ((float*)p.db)[c] = s[0];
  }
}

// ---- plans.  gen 1: the 128 x 128 kernel above (any CO, CI multiple of 8); gen 2: k_wgrad2 with tile 32 mi x 32 nj and m
// token slices per XCD (8 m partial records).
struct WgPlan { int gen, mi, nj, m; };

static void wgrad_plan(int64_t T, int CO, int CI, int& tiles_co, int& tiles_ci, int& nsplit, int64_t& rps) {
  tiles_co = (CO + WG_TILE - 1) / WG_TILE; tiles_ci = (CI + WG_TILE - 1) / WG_TILE;
  const int tiles = tiles_co * tiles_ci;
  // ~2 workgroups per CU, whole XCD rounds.  The HBM-bound shapes (>= 64 k tokens into >= 3 output tiles: the fc / qkv
  // layers of stages 1-2) run 8-12 % faster with 3 per CU (round-2 sweep, tools/wgrad_probe2.py); the MFMA-bound ones
  // (stages 3-4) and the one- and two-tile outputs lose 10-25 % there
  const int tgt = (T >= 65536 && tiles >= 3) ? 768 : 512;
  int64_t s = ((tgt + tiles - 1) / tiles + 7) / 8 * 8;
  const int64_t max_by_rows = (T + 4 * WG_ROWS - 1) / (4 * WG_ROWS);   // >= 4 steps per workgroup
  const int64_t max_by_ws = ((int64_t)96 << 20) / ((int64_t)CO * CI * 4);
  if (s > max_by_rows) s = max_by_rows;
  if (s > max_by_ws) s = max_by_ws;
  if (s > 8) s = s / 8 * 8;
  if (s < 1) s = 1;
  rps = ((T + s - 1) / s + WG_ROWS - 1) / WG_ROWS * WG_ROWS;
  nsplit = (int)((T + rps - 1) / rps);
}

static bool wg2_ok(int64_t T, int CO, int CI, int64_t sdy, int64_t sx) {
  return CO % 96 == 0 && CI % 96 == 0 && T * (sdy > sx ? sdy : sx) * 2 < ((int64_t)1 << 31);
}
// slices per XCD a gen-2 plan may use: >= 4 stages per workgroup, partial records within the 96 MB workspace bound
static int wg2_max_m(int64_t T, int CO, int CI) {
  const int64_t by_rows = T / (8 * 4 * 32), by_ws = ((int64_t)96 << 20) / ((int64_t)CO * CI * 4 * 8);
  const int64_t m = by_rows < by_ws ? by_rows : by_ws;
  return m < 1 ? 1 : (m > 64 ? 64 : (int)m);
}
// The candidates the tuner times (and the default picks from): every tile shape that divides the output, with the
// slice count that fills the 64 workgroup slots of an XCD and half of it (fewer, larger partial records).
static int wg2_candidates(int64_t T, int CO, int CI, WgPlan* out) {
  int n = 0;
  const int mx = wg2_max_m(T, CO, CI);
  for (int mi = 6; mi >= 3; mi -= 3)
    for (int nj = 6; nj >= 3; nj -= 3) {
      if (CO % (32 * mi) || CI % (32 * nj)) continue;
      const int tiles = (CO / (32 * mi)) * (CI / (32 * nj));
      int m0 = 64 / tiles; if (m0 < 1) m0 = 1; if (m0 > mx) m0 = mx;
      out[n++] = WgPlan{2, mi, nj, m0};
      if (m0 >= 2) out[n++] = WgPlan{2, mi, nj, m0 / 2};
    }
  return n;
}
// Default (untuned) plan: a cost model fitted to the MI355X sweep of tools/wgrad2_probe.py (profiles/r03_wgrad2_sweep.txt):
// main loop at 1.3 PFLOP/s x tile efficiency x slot fill or at the HBM rate the bytes in flight sustain, plus the partial
// records written once and read once.
static WgPlan wg2_default(int64_t T, int CO, int CI) {
  WgPlan c[8]; const int n = wg2_candidates(T, CO, CI, c);
  double best = 1e30; WgPlan bp = c[0];
  for (int i = 0; i < n; ++i) {
    const int tiles = (CO / (32 * c[i].mi)) * (CI / (32 * c[i].nj)), wgx = tiles * c[i].m;
    const double fill = (double)wgx / (64.0 * ((wgx + 63) / 64));
    const double eff = c[i].mi * c[i].nj == 36 ? 1.0 : (c[i].mi * c[i].nj == 18 ? 0.73 : 0.52);
    const double t_mfma = 2.0 * T * CO * CI / (1.3e15 * eff * fill);
    const double flight = 8.0 * wgx * 2 * 64 * 32 * (c[i].mi + c[i].nj), need = 12.6e6;
    const double t_hbm = 2.0 * T * (CO + CI) / 5.6e12 * (flight < need ? 1.0 + 0.35 * (need / flight - 1.0) : 1.0);
    const double part = 8.0 * c[i].m * CO * CI * 4;
    const double t = (t_mfma > t_hbm ? t_mfma : t_hbm) + part / 5e12 + 5.5e-6 + part / 7.7e12;
    if (t < best) { best = t; bp = c[i]; }
  }
  return bp;
}

#include <map>
#include <mutex>
#include <tuple>
static std::mutex g_wg_mu;
static std::map<std::tuple<int64_t, int, int>, WgPlan> g_wg_plans;      // problems vil_linear_wgrad_tune has measured

static WgPlan wgrad_choose(int64_t T, int CO, int CI, int64_t sdy, int64_t sx) {
  if (!wg2_ok(T, CO, CI, sdy, sx)) return WgPlan{1, 0, 0, 0};
  {
    std::lock_guard<std::mutex> lk(g_wg_mu);
    auto it = g_wg_plans.find(std::make_tuple(T, CO, CI));
    if (it != g_wg_plans.end()) return it->second;
  }
  return wg2_default(T, CO, CI);
}

// Writes (gen 1 / 2) or erases (gen 0) the cached plan of a problem: what vil_linear_wgrad_tune stores.  For restoring
// a selection made earlier, for measurements (tools/wgrad2_probe.py) and for the tests that walk every plan.
extern "C" int vil_linear_wgrad_set_plan(int64_t T, int CO, int CI, int gen, int mi, int nj, int m) {
  if (T <= 0 || CO <= 0 || CI <= 0 || gen < 0 || gen > 2) return VIL_E_SHAPE;
  std::lock_guard<std::mutex> lk(g_wg_mu);
  const auto key = std::make_tuple(T, CO, CI);
  if (gen == 0) { g_wg_plans.erase(key); return 0; }
  if (gen == 1) { g_wg_plans[key] = WgPlan{1, 0, 0, 0}; return 0; }
  if ((mi != 3 && mi != 6) || (nj != 3 && nj != 6) || CO % (32 * mi) || CI % (32 * nj) || m < 1) return VIL_E_SHAPE;
  const int mx = wg2_max_m(T, CO, CI);
  g_wg_plans[key] = WgPlan{2, mi, nj, m > mx ? mx : m};
  return 0;
}

// The plan vil_linear_wgrad would run for a problem with 8-element-aligned strides: the measured one when the problem
// was tuned (plan[4] = 1), the cost model's otherwise (plan[4] = 0).  plan = {gen, mi, nj, m, tuned}.
extern "C" int vil_linear_wgrad_get_plan(int64_t T, int CO, int CI, int* plan) {
  if (!plan) return VIL_E_NULL;
  if (T <= 0 || CO <= 0 || CI <= 0) return VIL_E_SHAPE;
  bool tuned;
  {
    std::lock_guard<std::mutex> lk(g_wg_mu);
    tuned = g_wg_plans.count(std::make_tuple(T, CO, CI)) != 0;
  }
  const WgPlan q = wgrad_choose(T, CO, CI, CO, CI);
  plan[0] = q.gen; plan[1] = q.mi; plan[2] = q.nj; plan[3] = q.m; plan[4] = tuned ? 1 : 0;
  return 0;
}

template <int MI, int NJ>
static int wgrad2_launch(const WgradParams& p, int m, hipStream_t s) {
  const size_t lds = (size_t)3 * 32 * (32 * MI + 32 * NJ) * 2 + 1024;
  const int e = vil_ensure_dyn_lds((const void*)k_wgrad2<MI, NJ, 3>, lds);
  if (e) return e;
  k_wgrad2<MI, NJ, 3><<<dim3(8 * m * p.tiles), dim3(256), lds, s>>>(p);
  return 0;
}

extern "C" size_t vil_linear_wgrad_workspace_bytes(int64_t T, int CO, int CI) {
  if (T <= 0 || CO <= 0 || CI <= 0) return 0;
  int a, b, s; int64_t rps;
  wgrad_plan(T, CO, CI, a, b, s, rps);
  size_t need = ((size_t)s * CO * CI + (size_t)s * CO) * sizeof(float) + 256;
  if (CO % 96 == 0 && CI % 96 == 0) {       // any gen-2 plan the tuner may select
    const size_t m = (size_t)wg2_max_m(T, CO, CI);
    const size_t n2 = ((size_t)8 * m * CO * CI + (size_t)8 * m * CO) * sizeof(float) + 256;
    if (n2 > need) need = n2;
  }
  return need;
}

static int wgrad_run(const WgPlan& q, WgradParams p, hipStream_t s) {
  const double ob = p.out_bf16 ? 2.0 : 4.0;
  vil_prof_tag((int)p.T, p.CO, p.CI, 0, 0, 0, 0, 0);
  // algorithmic traffic of the pair: dY and X read once, dW (+db) written once; the fp32 partials are overhead
  const double bytes = 2.0 * (double)p.T * (p.CO + p.CI) + ob * ((double)p.CO * p.CI + (p.db ? p.CO : 0));
  const int64_t n = (int64_t)p.CO * p.CI / 4 + (p.db ? p.CO : 0);
  if (q.gen == 2) {
    p.tiles_ci = p.CI / (32 * q.nj); p.tiles = (p.CO / (32 * q.mi)) * p.tiles_ci; p.nsplit = 8 * q.m;
    p.rows_per_split = ((p.T + p.nsplit - 1) / p.nsplit + 31) / 32 * 32;
    p.dbparts = p.db ? p.parts + (size_t)p.nsplit * p.CO * p.CI : nullptr;
    vil_prof_begin(VIL_K_WGRAD, s, bytes, 2.0 * (double)p.T * p.CO * p.CI);
    int e = q.mi == 6 ? (q.nj == 6 ? wgrad2_launch<6, 6>(p, q.m, s) : wgrad2_launch<6, 3>(p, q.m, s))
                      : (q.nj == 6 ? wgrad2_launch<3, 6>(p, q.m, s) : wgrad2_launch<3, 3>(p, q.m, s));
    vil_prof_end(s);
    if (!e) e = (int)hipGetLastError();
    if (e) return e;
    vil_prof_begin(VIL_K_WGRAD_REDUCE, s, 0, 0);
    if (p.nsplit >= 32) k_wgrad2_reduce<16><<<dim3((unsigned)((n + 63) / 64)), dim3(1024), 0, s>>>(p, q.mi, q.nj);
    else k_wgrad2_reduce<4><<<dim3((unsigned)((n + 63) / 64)), dim3(256), 0, s>>>(p, q.mi, q.nj);
    vil_prof_end(s);
    return (int)hipGetLastError();
  }
  int tiles_co;
  wgrad_plan(p.T, p.CO, p.CI, tiles_co, p.tiles_ci, p.nsplit, p.rows_per_split);
  p.dbparts = p.db ? p.parts + (size_t)p.nsplit * p.CO * p.CI : nullptr;
  p.tiles = tiles_co * p.tiles_ci;
  vil_prof_begin(VIL_K_WGRAD, s, bytes, 2.0 * (double)p.T * p.CO * p.CI);
  k_wgrad<<<dim3(p.tiles * ((p.nsplit + 7) / 8 * 8)), dim3(256), 0, s>>>(p);
  vil_prof_end(s);
  const int e = (int)hipGetLastError();
  if (e) return e;
  vil_prof_begin(VIL_K_WGRAD_REDUCE, s, 0, 0);
  k_wgrad_reduce<<<dim3((unsigned)((n + 63) / 64)), dim3(256), 0, s>>>(p);
  vil_prof_end(s);
  return (int)hipGetLastError();
}

static int wgrad_args(WgradParams& p, const void* dy, const void* x, int64_t T, int CO, int CI, int64_t dy_stride, int64_t x_stride,
                      void* dw, void* db, int out_bf16, void* workspace) {
  if (!dy || !x || !dw || !workspace) return VIL_E_NULL;
  if (T <= 0 || CO <= 0 || CI <= 0) return VIL_E_SHAPE;
  if ((CO & 7) || (CI & 7) || (dy_stride & 7) || (x_stride & 7) || (((uintptr_t)dy | (uintptr_t)x) & 15)) return VIL_E_ALIGN;
  p.dy = (const __bf16*)dy; p.x = (const __bf16*)x; p.T = T; p.sdy = dy_stride; p.sx = x_stride;
  p.CO = CO; p.CI = CI; p.parts = (float*)workspace; p.dw = dw; p.db = db; p.out_bf16 = out_bf16;
  return 0;
}

extern "C" int vil_linear_wgrad(const void* dy, const void* x, int64_t T, int CO, int CI, int64_t dy_stride,
                                int64_t x_stride, void* dw, void* db, int out_bf16, void* workspace, void* stream) {
  WgradParams p;
  if (const int e = wgrad_args(p, dy, x, T, CO, CI, dy_stride, x_stride, dw, db, out_bf16, workspace)) return e;
  return wgrad_run(wgrad_choose(T, CO, CI, dy_stride, x_stride), p, (hipStream_t)stream);
}

// Times the gen-1 plan and every gen-2 candidate on the caller's operands (two passes of four launches each, the
// minimum of the pass averages) and remembers the fastest for (T, CO, CI).  Synchronises the stream.
extern "C" int vil_linear_wgrad_tune(const void* dy, const void* x, int64_t T, int CO, int CI, int64_t dy_stride,
                                     int64_t x_stride, void* dw, void* db, int out_bf16, void* workspace, void* stream) {
  WgradParams p;
  if (const int e = wgrad_args(p, dy, x, T, CO, CI, dy_stride, x_stride, dw, db, out_bf16, workspace)) return e;
  if (!wg2_ok(T, CO, CI, dy_stride, x_stride)) return 0;
  hipStream_t s = (hipStream_t)stream;
  WgPlan c[9]; int n = wg2_candidates(T, CO, CI, c);
  c[n++] = WgPlan{1, 0, 0, 0};
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return (int)hipGetLastError();
  float best = 1e30f; int bi = -1, err = 0;
  for (int i = 0; i < n && !err; ++i) {
    float t = 1e30f;
    err = wgrad_run(c[i], p, s);                                   // warm-up (raises the LDS limit, faults the workspace in)
    for (int pass = 0; pass < 2 && !err; ++pass) {
      if (hipEventRecord(e0, s) != hipSuccess) err = (int)hipGetLastError();
      for (int k = 0; k < 4 && !err; ++k) err = wgrad_run(c[i], p, s);
      float ms = 0.f;
      if (!err && (hipEventRecord(e1, s) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
                   hipEventElapsedTime(&ms, e0, e1) != hipSuccess)) err = (int)hipGetLastError();
      if (!err && ms < t) t = ms;
    }
    if (!err && t < best) { best = t; bi = i; }
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (err) return err;
  if (bi >= 0) { std::lock_guard<std::mutex> lk(g_wg_mu); g_wg_plans[std::make_tuple(T, CO, CI)] = c[bi]; }
  return 0;
}
