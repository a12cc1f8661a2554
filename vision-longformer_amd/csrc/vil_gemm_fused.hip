// vil_gemm_fused.hip -- the one GEMM of the MLP block that is worth writing by hand: the input gradient of `fc2` with the
// backward of the exact (erf) GELU in its epilogue (reference src/models/msvit.py:17-34: fc1 -> nn.GELU -> fc2),
//     dh[t][n] = ( sum_k dy[t][k] * w[k][n] ) * gelu'(h[t][n]),   gelu'(x) = Phi(x) + x phi(x),
// bf16 in / out, fp32 accumulate.  It replaces a hipBLASLt GEMM + ATen's GeluBackward elementwise kernel (read h, read
// da, write dh: 6 bytes per element of the widest tensor of the block; 0.62 ms of a 15 ms ViL-Small step) by one
// launch that reads h and writes dh once.  hipBLASLt's own DGELU epilogue differentiates the tanh approximation, not
// the reference's erf form.  erf by Abramowitz-Stegun 7.1.26 (|error| < 1.5e-7, far below the bf16 result's rounding).
//
// Shape of the problem: T (tokens) is 6 400 ... 400 000, K = C is 96 ... 768, N = 4C -- tall, with a short K loop and a
// wide output: HBM / VALU(epilogue)-bound at stages 1-2, MFMA-bound at stages 3-4.
// One workgroup (4 waves) = a 128 (t) x 128 (n) output tile, computed TRANSPOSED (C^T = W^T dY^T, the S^T orientation of
// the attention kernels): a lane holds 8 consecutive n of one row t per pair of accumulator tiles -- 16-byte loads of h
// and 16-byte stores of dh, no LDS transpose in the epilogue.  A operand = W^T tiles through ds_read_b64_tr_b16 from the
// row-major [k][n] LDS image; B operand = dY rows (two 8-byte reads per fragment: the transposed read delivers k in the
// order {4g + e, 16 + 4g + e}, so the other operand is read in that order too).  K streams in 64-deep blocks through a
// two-slot LDS ring filled by LDS-DMA (the XOR-swizzled image of vil_attn_dense.hip).  Rows t >= T and k >= K read as
// zeros through the bounded descriptors (w); a dY chunk beyond K in the last block is finite data of the next row
// times those zeros.
#include "vil_mfma_common.h"

struct DgParams {
  const void* dy; const void* w; const void* h; void* dh;
  int T, K, N;
  int dy_rs, h_rs, dh_rs;      // row strides, elements
  int nn_tiles;                // N / 128
};

__device__ __forceinline__ int gf_off(int row, int colb) { return row * 128 + (colb ^ (((row >> 1) & 3) << 5)); }

__device__ __forceinline__ float gelu_grad(float x) {
  const float z = x * 0.70710678118654752f, az = __builtin_fabsf(z);
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, az, 1.0f));
  const float e = __builtin_amdgcn_exp2f(-(z * z) * LOG2E);                          // exp(-x^2 / 2)
  float q = __builtin_fmaf(t, 1.061405429f, -1.453152027f);
  q = __builtin_fmaf(q, t, 1.421413741f);
  q = __builtin_fmaf(q, t, -0.284496736f);
  q = __builtin_fmaf(q, t, 0.254829592f);
  const float erfabs = __builtin_fmaf(-(q * t), e, 1.0f);
  const float erfz = __builtin_copysignf(erfabs, z);
  return __builtin_fmaf(0.5f, erfz, 0.5f) + x * (e * 0.39894228040143268f);
}

#define GF_SLOT (32 * 1024)      // bytes of one ring slot: [128 x 64] dY block, then two [64 x 64] halves of the W block

__global__ __launch_bounds__(256, 2) void k_dgrad_dgelu(DgParams p) {
  typedef __bf16 T_;
  typedef typename V16<T_>::x8 X8;
  typedef typename V16<T_>::x4 X4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lj = lane & 15, lg = lane >> 4;
  const int wt = wave & 1, wn = wave >> 1;
  // consecutive workgroups = the n-tiles of one t-tile (they share its dY rows): keep them on one XCD
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_t = logical / p.nn_tiles, tile_n = logical - tile_t * p.nn_tiles;
  const int t0 = tile_t * 128, n0 = tile_n * 128;

  const __amdgpu_buffer_rsrc_t dyr = make_rsrc_n(p.dy, (unsigned)(((int64_t)(p.T - 1) * p.dy_rs + p.K) * 2));
  const __amdgpu_buffer_rsrc_t wr = make_rsrc_n(p.w, (unsigned)((int64_t)p.K * p.N * 2));
  const int drow = lane >> 3, dchunk = (lane & 7) ^ (((drow >> 1) & 3) << 1);
  const int dy_v0 = (t0 + drow) * (p.dy_rs * 2) + dchunk * 16;
  const int w_v0 = drow * (p.N * 2) + n0 * 2 + dchunk * 16;
  auto issue = [&](int kb, int slot) {
    char* base = smem + slot * GF_SLOT;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int pc = wave + 4 * u;                       // 32 one-kilobyte pieces, 8 per wave
      if (pc < 16)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(dyr, (__attribute__((address_space(3))) void*)(base + pc * 1024), 16,
                                                 dy_v0 + pc * 8 * (p.dy_rs * 2) + kb * 128, 0, 0, 0);
      else {
        const int half = (pc - 16) >> 3, p8 = (pc - 16) & 7;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (__attribute__((address_space(3))) void*)(base + pc * 1024), 16,
                                                 w_v0 + (kb * 64 + p8 * 8) * (p.N * 2) + half * 128, 0, 0, 0);
      }
    }
  };

  const int nkb = (p.K + 63) >> 6;
  issue(0, 0);
  // the epilogue's h values: 16 eight-byte loads per lane, in flight during the whole K loop (the first version loaded
  // them tile row by tile row in the epilogue: four dependent HBM round trips per workgroup)
  const T_* hb = (const T_*)p.h;
  X8 h8[4][2];
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) {
    const int t = min(t0 + wt * 64 + tt * 16 + lj, p.T - 1);
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) h8[tt][pr] = *(const X8*)(hb + (int64_t)t * p.h_rs + n0 + wn * 64 + pr * 32 + lg * 8);
  }

  // A tiles come in pairs (pr, hf) over 32 features: MFMA row j of tile (pr, hf) stands for feature 32 pr + 8 (j / 4) +
  // 4 hf + j % 4, so accumulator rows 4 g .. 4 g + 3 of the pair's two tiles are features 8 g .. 8 g + 7 -- one 16-byte
  // load of h and one 16-byte store of dh per lane and pair (8-byte accesses in 32-byte runs ran stage 1 at 2.9 TB/s).
  // The transposed read allows it: the column a lane RECEIVES is (what lane 4 e + j / 4 pointed at) + j % 4, so the
  // loader lane q = j % 4 ... points at feature 32 pr + 8 q + 4 hf instead of 16 nt + 4 q.
  int wtr[4], dnat[2][2];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) wtr[nt] = gf_off(lg * 4 + (lj >> 2), ((nt >> 1) * 32 + (lj & 3) * 8 + (nt & 1) * 4) * 2);
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) dnat[ks][hf] = gf_off(lj, ks * 64 + hf * 32 + lg * 8);
  f32x4 acc[4][4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) acc[nt][tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  lds_dma_wait<8>();                                      // block 0 has landed; the 8 h loads issued after it stay in flight
  __syncthreads();

  for (int kb = 0; kb < nkb; ++kb) {
    if (kb + 1 < nkb) issue(kb + 1, (kb + 1) & 1);
    const char* dyb = smem + (kb & 1) * GF_SLOT + wt * (64 * 128);
    const char* wb = smem + (kb & 1) * GF_SLOT + 16 * 1024 + wn * (64 * 128);
    const int nks = min(2, (p.K - kb * 64) >> 5);
    for (int ks = 0; ks < nks; ++ks) {
      // (transposed reads as inline assembly, lds_tr_issue in vil_mfma_common.h: through the builtin the first read of
      // every K block waited for the NEXT block's LDS-DMA requests -- no prefetch at all from K = 384 on)
      X8 a[4], bq[4];
      s16x4 ar[4][2];
      const unsigned wa = lds_addr32(wb) + ks * (32 * 128);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) { ar[nt][0] = lds_tr_issue<0>(wa + wtr[nt]); ar[nt][1] = lds_tr_issue<16 * 128>(wa + wtr[nt]); }
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const X4 d4 = *(const X4*)(dyb + tt * (16 * 128) + (ks ? dnat[1][hf] : dnat[0][hf]));
#pragma unroll
          for (int e = 0; e < 4; ++e) bq[tt][hf * 4 + e] = d4[e];
        }
      lds_tr_settle(ar[0][0], ar[0][1], ar[1][0], ar[1][1], ar[2][0], ar[2][1], ar[3][0], ar[3][1]);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) a[nt] = lds_tr_join<T_>(ar[nt][0], ar[nt][1]);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) acc[nt][tt] = mfma16(a[nt], bq[tt], acc[nt][tt]);
    }
    lds_dma_wait();                                       // the next block's requests (lds_dma_wait, vil_mfma_common.h)
    __syncthreads();
  }

  // ---- epilogue: dh = acc * gelu'(h); lane (j, g) of tile (nt, tt): row t = .. + 16 tt + j, columns n = .. + 16 nt + 4 g ..+3
  // (h4 was requested before the K loop: the epilogue itself waits for nothing)
  T_* ob = (T_*)p.dh;
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) {
    const int t = t0 + wt * 64 + tt * 16 + lj;
    if (t < p.T) {
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        X8 o8;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          o8[r] = (T_)(acc[2 * pr][tt][r] * gelu_grad((float)h8[tt][pr][r]));
          o8[4 + r] = (T_)(acc[2 * pr + 1][tt][r] * gelu_grad((float)h8[tt][pr][4 + r]));
        }
        *(X8*)(ob + (int64_t)t * p.dh_rs + n0 + wn * 64 + pr * 32 + lg * 8) = o8;
      }
    }
  }
}

// ---- fc1 with nn.GELU() in its epilogue (reference msvit.py:29-31): h = x W^T + b, a = gelu(h), both written once.
// Same tiling and ring as k_dgrad_dgelu; here BOTH operands are k-contiguous (x[t][k], w[n][k]), so both are plain
// 16-byte fragment reads of the swizzled image.  The 8-consecutive-features-per-lane mapping is made by the DMA: LDS row
// 16 nt + j of a wave's 64-row half holds weight row 32 (nt / 2) + 8 (j / 4) + 4 (nt % 2) + j % 4.
struct FgParams {
  const void* x; const void* w; const void* bias; void* h; void* a;
  int T, K, N;
  int x_rs, o_rs;              // row strides, elements
  int nn_tiles;                // N / 128
};

__device__ __forceinline__ float gelu_fwd(float x) {
  const float z = x * 0.70710678118654752f, az = __builtin_fabsf(z);
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, az, 1.0f));
  const float e = __builtin_amdgcn_exp2f(-(z * z) * LOG2E);
  float q = __builtin_fmaf(t, 1.061405429f, -1.453152027f);
  q = __builtin_fmaf(q, t, 1.421413741f);
  q = __builtin_fmaf(q, t, -0.284496736f);
  q = __builtin_fmaf(q, t, 0.254829592f);
  const float hc = 0.5f * (q * t) * e;                    // erfc(|z|) / 2: the negative side has no 1 - erf cancellation
  return x * (x < 0.f ? hc : 1.0f - hc);
}

__global__ __launch_bounds__(256, 2) void k_fwd_gelu(FgParams p) {
  typedef __bf16 T_;
  typedef typename V16<T_>::x8 X8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lj = lane & 15, lg = lane >> 4;
  const int wt = wave & 1, wn = wave >> 1;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_t = logical / p.nn_tiles, tile_n = logical - tile_t * p.nn_tiles;
  const int t0 = tile_t * 128, n0 = tile_n * 128;

  const __amdgpu_buffer_rsrc_t xr = make_rsrc_n(p.x, (unsigned)(((int64_t)(p.T - 1) * p.x_rs + p.K) * 2));
  const __amdgpu_buffer_rsrc_t wr = make_rsrc_n(p.w, (unsigned)((int64_t)p.N * p.K * 2));
  const int drow = lane >> 3, dchunk = (lane & 7) ^ (((drow >> 1) & 3) << 1);
  const int x_v0 = (t0 + drow) * (p.x_rs * 2) + dchunk * 16;
  int w_v[2];                                              // weight row of LDS row 8 q + drow, q even / odd (see above)
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    const int j = par * 8 + drow;                          // row within a 16-row tile
    w_v[par] = (n0 + 8 * (j >> 2) + (j & 3)) * (p.K * 2) + dchunk * 16;
  }
  auto issue = [&](int kb, int slot) {
    char* base = smem + slot * GF_SLOT;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int pc = wave + 4 * u;                         // 32 one-kilobyte pieces, 8 per wave
      if (pc < 16)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (__attribute__((address_space(3))) void*)(base + pc * 1024), 16,
                                                 x_v0 + pc * 8 * (p.x_rs * 2) + kb * 128, 0, 0, 0);
      else {
        const int q = pc - 16, nt = q >> 1;                // LDS rows 8 q .. 8 q + 7 = rows 8 (q & 1) .. of tile nt (0..7)
        const int frow = (nt >> 2) * 64 + ((nt >> 1) & 1) * 32 + (nt & 1) * 4;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (__attribute__((address_space(3))) void*)(base + pc * 1024), 16,
                                                 ((q & 1) ? w_v[1] : w_v[0]) + frow * (p.K * 2) + kb * 128, 0, 0, 0);
      }
    }
  };

  const int nkb = (p.K + 63) >> 6;
  issue(0, 0);
  X8 bias8[2];
#pragma unroll
  for (int pr = 0; pr < 2; ++pr) {
    X8 z = {};
    bias8[pr] = p.bias ? *(const X8*)((const T_*)p.bias + n0 + wn * 64 + pr * 32 + lg * 8) : z;
  }
  int nat[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) nat[ks] = gf_off(lj, ks * 64 + lg * 16);
  f32x4 acc[4][4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) acc[nt][tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  lds_dma_wait();                                         // block 0 (and the two bias loads)
  __syncthreads();

  for (int kb = 0; kb < nkb; ++kb) {
    if (kb + 1 < nkb) issue(kb + 1, (kb + 1) & 1);
    const char* xb = smem + (kb & 1) * GF_SLOT + wt * (64 * 128);
    const char* wb = smem + (kb & 1) * GF_SLOT + 16 * 1024 + wn * (64 * 128);
    const int nks = min(2, (p.K - kb * 64) >> 5);
    for (int ks = 0; ks < nks; ++ks) {
      X8 a[4], bq[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) a[nt] = *(const X8*)(wb + nt * (16 * 128) + (ks ? nat[1] : nat[0]));
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) bq[tt] = *(const X8*)(xb + tt * (16 * 128) + (ks ? nat[1] : nat[0]));
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) acc[nt][tt] = mfma16(a[nt], bq[tt], acc[nt][tt]);
    }
    lds_dma_wait();                                       // the next block's requests (lds_dma_wait, vil_mfma_common.h)
    __syncthreads();
  }

  // ---- epilogue: lane (j, g), pair pr, column tile tt: token t0 + 64 wt + 16 tt + j, features n0 + 64 wn + 32 pr + 8 g .. +7
  T_* hb = (T_*)p.h;
  T_* ab = (T_*)p.a;
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) {
    const int t = t0 + wt * 64 + tt * 16 + lj;
    if (t < p.T) {
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        X8 o8, a8;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          o8[r] = (T_)(acc[2 * pr][tt][r] + (float)bias8[pr][r]);
          o8[4 + r] = (T_)(acc[2 * pr + 1][tt][r] + (float)bias8[pr][4 + r]);
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) a8[r] = (T_)gelu_fwd((float)o8[r]);       // GELU of the rounded pre-activation
        const int64_t o = (int64_t)t * p.o_rs + n0 + wn * 64 + pr * 32 + lg * 8;
        *(X8*)(hb + o) = o8;
        *(X8*)(ab + o) = a8;
      }
    }
  }
}

// h[T][N] = x[T][K] . w[N][K]^T + bias[N], a = gelu(h) (exact erf form, of the rounded bf16 h); bf16, row strides in
// elements (h and a share theirs).  K % 32 == 0, N % 128 == 0; VIL_E_BACKEND outside that contract.
extern "C" int vil_gemm_gelu_bf16(const void* x, const void* w, const void* bias, void* h, void* a, int64_t T, int K, int N,
                                  int64_t x_row_stride, int64_t out_row_stride, void* stream) {
  if (!x || !w || !h || !a) return VIL_E_NULL;
  if (T <= 0 || K <= 0 || N <= 0) return VIL_E_SHAPE;
  if ((K & 31) || (N & 127)) return VIL_E_BACKEND;
  if ((x_row_stride & 7) || (out_row_stride & 7) || (((uintptr_t)x | (uintptr_t)w | (uintptr_t)h | (uintptr_t)a) & 15) ||
      (bias && ((uintptr_t)bias & 15))) return VIL_E_ALIGN;
  if (x_row_stride < K || out_row_stride < N) return VIL_E_SHAPE;
  if ((T + 128) * x_row_stride * 2 >= (1ll << 31) || (int64_t)K * N * 2 >= (1ll << 31) || T * (N / 128) >= (1ll << 30)) return VIL_E_BACKEND;
  FgParams p;
  p.x = x; p.w = w; p.bias = bias; p.h = h; p.a = a;
  p.T = (int)T; p.K = K; p.N = N;
  p.x_rs = (int)x_row_stride; p.o_rs = (int)out_row_stride;
  p.nn_tiles = N / 128;
  const unsigned grid = (unsigned)(((T + 127) / 128) * p.nn_tiles);
  const size_t lds = 2 * GF_SLOT;
  if (int he = vil_ensure_dyn_lds((const void*)k_fwd_gelu, lds)) return he;
  k_fwd_gelu<<<dim3(grid), dim3(256), lds, (hipStream_t)stream>>>(p);
  return (int)hipGetLastError();
}

// dh[T][N] = (dy[T][K] . w[K][N]) * gelu'(h[T][N]); bf16, row strides in elements.  K % 32 == 0, N % 128 == 0, 16-byte
// aligned bases and rows; VIL_E_BACKEND when the problem is outside the kernel's contract (the caller
// then runs the GEMM and the GELU backward separately).
extern "C" int vil_gemm_dgelu_bf16(const void* dy, const void* w, const void* h, void* dh, int64_t T, int K, int N,
                                   int64_t dy_row_stride, int64_t h_row_stride, int64_t dh_row_stride, void* stream) {
  if (!dy || !w || !h || !dh) return VIL_E_NULL;
  if (T <= 0 || K <= 0 || N <= 0) return VIL_E_SHAPE;
  if ((K & 31) || (N & 127)) return VIL_E_BACKEND;
  if ((dy_row_stride & 7) || (h_row_stride & 7) || (dh_row_stride & 7) ||
      (((uintptr_t)dy | (uintptr_t)w | (uintptr_t)h | (uintptr_t)dh) & 15)) return VIL_E_ALIGN;
  if (dy_row_stride < K || h_row_stride < N || dh_row_stride < N) return VIL_E_SHAPE;
  // 32-bit byte offsets inside the descriptors and the tile decode
  if ((T + 128) * dy_row_stride * 2 >= (1ll << 31) || (int64_t)K * N * 2 >= (1ll << 31) || T * (N / 128) >= (1ll << 30)) return VIL_E_BACKEND;
  DgParams p;
  p.dy = dy; p.w = w; p.h = h; p.dh = dh;
  p.T = (int)T; p.K = K; p.N = N;
  p.dy_rs = (int)dy_row_stride; p.h_rs = (int)h_row_stride; p.dh_rs = (int)dh_row_stride;
  p.nn_tiles = N / 128;
  const unsigned grid = (unsigned)(((T + 127) / 128) * p.nn_tiles);
  const size_t lds = 2 * GF_SLOT;
  if (int he = vil_ensure_dyn_lds((const void*)k_dgrad_dgelu, lds)) return he;
  k_dgrad_dgelu<<<dim3(grid), dim3(256), lds, (hipStream_t)stream>>>(p);
  return (int)hipGetLastError();
}
