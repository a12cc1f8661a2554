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

extern "C" size_t vil_linear_wgrad_workspace_bytes(int64_t T, int CO, int CI) {
  if (T <= 0 || CO <= 0 || CI <= 0) return 0;
  int a, b, s; int64_t rps;
  wgrad_plan(T, CO, CI, a, b, s, rps);
  return ((size_t)s * CO * CI + (size_t)s * CO) * sizeof(float) + 256;
}

extern "C" int vil_linear_wgrad(const void* dy, const void* x, int64_t T, int CO, int CI, int64_t dy_stride,
                                int64_t x_stride, void* dw, void* db, int out_bf16, void* workspace, void* stream) {
  if (!dy || !x || !dw || !workspace) return VIL_E_NULL;
  if (T <= 0 || CO <= 0 || CI <= 0) return VIL_E_SHAPE;
  if ((CO & 7) || (CI & 7) || (dy_stride & 7) || (x_stride & 7) || (((uintptr_t)dy | (uintptr_t)x) & 15)) return VIL_E_ALIGN;
  WgradParams p;
  p.dy = (const __bf16*)dy; p.x = (const __bf16*)x; p.T = T; p.sdy = dy_stride; p.sx = x_stride;
  p.CO = CO; p.CI = CI;
  int tiles_co;
  wgrad_plan(T, CO, CI, tiles_co, p.tiles_ci, p.nsplit, p.rows_per_split);
  p.parts = (float*)workspace;
  p.dbparts = db ? p.parts + (size_t)p.nsplit * CO * CI : nullptr;
  p.dw = dw; p.db = db; p.out_bf16 = out_bf16;
  hipStream_t s = (hipStream_t)stream;
  p.tiles = tiles_co * p.tiles_ci;
  // algorithmic traffic of the pair: dY and X read once, dW (+db) written once; the fp32 partials are overhead
  const double ob = out_bf16 ? 2.0 : 4.0;
  vil_prof_tag((int)T, CO, CI, 0, 0, 0, 0, 0);
  vil_prof_begin(VIL_K_WGRAD, s, 2.0 * (double)T * (CO + CI) + ob * ((double)CO * CI + (db ? CO : 0)), 2.0 * (double)T * CO * CI);
  k_wgrad<<<dim3(p.tiles * ((p.nsplit + 7) / 8 * 8)), dim3(256), 0, s>>>(p);
  vil_prof_end(s);
  int e = (int)hipGetLastError();
  if (e) return e;
  const int64_t n = (int64_t)CO * CI / 4 + (db ? CO : 0);
  vil_prof_begin(VIL_K_WGRAD_REDUCE, s, 0, 0);
  k_wgrad_reduce<<<dim3((unsigned)((n + 63) / 64)), dim3(256), 0, s>>>(p);
  vil_prof_end(s);
  return (int)hipGetLastError();
}
