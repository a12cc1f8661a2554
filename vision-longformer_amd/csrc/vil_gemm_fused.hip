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
// One 128 (t) x 128 (n) output tile at a time per workgroup (4 compute waves + a loader wave, persistent: the schedule
// comment below), computed TRANSPOSED (C^T = W^T dY^T, the S^T orientation of
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
  int ntiles;                  // token tiles x nn_tiles
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

// ---- Schedule of both kernels (round 4): persistent workgroups, a loader wave, static tile lists.
// Round-4 ablations of k_fwd_gelu at stage 3 (T 25 216, K 384, N 1 536; 70 us): without the epilogue's stores 42 us,
// without the K blocks' loads after the first 58 us, with neither loads nor GELU nor stores 31 us.  155 MB of stores do
// not cost 28 us of bandwidth: a workgroup's slot is held until its stores have drained (s_endpgm waits for them), and
// the next workgroup on the slot then waits a full HBM round trip for its first K block -- 2 364 short-lived workgroups
// on 512 slots pay both latencies 4.6 times over.  Now:
//   * 2 workgroups per CU stay resident and walk a STATIC list of tiles (XCD x gets the x-th eighth of the t-major tile
//     order, its 64 workgroups take consecutive tiles of it: the n-tiles of a token tile run side by side on one L2, as
//     xcd_remap arranged before).  No ticket counter: a returning global atomic in a wave's load stream cost the
//     attention kernels 2x (vil_mfma_common.h).
//   * wave 4 is a loader: it issues every LDS-DMA request of the workgroup and owns their completion (s_waitcnt vmcnt
//     in front of the block barrier).  Its vmcnt sees loads only.  The four compute waves never wait for a K block
//     with s_waitcnt vmcnt -- which would also wait for the previous tile's stores (gfx9 counts loads and stores in one
//     counter) -- so those stores drain under the next tile's K loop, and the next tile's first block is requested
//     before the epilogue starts.
//   * block barriers are bare s_barrier instructions: __syncthreads() carries a workgroup-scope release fence, i.e.
//     s_waitcnt vmcnt(0) in front of it once stores are outstanding.
//   * the epilogue's stores are inline assembly: hipcc's waitcnt pass, seeing stores pending over the tile loop's back
//     edge next to the next tile's loads, put s_waitcnt vmcnt(0) at the head of the tile loop -- the drain again.
//     Stores it does not track only ever ADD to the hardware count, so the counted waits it places for this wave's
//     loads wait for at least what they were meant to wait for.
#define GF_THREADS 320
typedef unsigned gf_u32x4 __attribute__((ext_vector_type(4)));
template <typename X8> __device__ __forceinline__ void gf_store16(void* base, unsigned byte_off, const X8& v) {
  // base: uniform (an SGPR pair), byte_off: 32 bits per lane -- no 64-bit address pairs in VGPRs.
  // (s_nop: a VALU write of the data registers of a store of more than 8 bytes needs one wait state; the hazard
  // recognizer does not look into inline assembly)
  asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(byte_off), "v"(__builtin_bit_cast(gf_u32x4, v)), "s"(base) : "memory");
}
__device__ __forceinline__ void gf_barrier() { asm volatile("s_barrier" ::: "memory"); }
__device__ __forceinline__ void gf_tiles(int ntiles, int& first, int& end, int& stride) {
  const int per = (ntiles + 7) >> 3, lo = (int)(blockIdx.x & 7) * per;
  stride = (int)(gridDim.x >> 3);
  end = min(ntiles, lo + per);
  first = lo + (int)(blockIdx.x >> 3);
}

// DGELU = false: the plain input gradient dx = dy . w (vil_gemm_tile_bf16 op 1): no h loads, no derivative.
template <bool DGELU>
__global__ __launch_bounds__(GF_THREADS, 4) void k_dgrad_dgelu(DgParams p) {
  typedef __bf16 T_;
  typedef typename V16<T_>::x8 X8;
  typedef typename V16<T_>::x4 X4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lj = lane & 15, lg = lane >> 4;
  int tile, tile_end, tile_stride;
  gf_tiles(p.ntiles, tile, tile_end, tile_stride);
  const int nkb = (p.K + 63) >> 6;

  if (wave == 4) {                                         // ---- the loader wave
    __builtin_amdgcn_s_setprio(3);
    const __amdgpu_buffer_rsrc_t dyr = make_rsrc_n(p.dy, (unsigned)(((int64_t)(p.T - 1) * p.dy_rs + p.K) * 2));
    const __amdgpu_buffer_rsrc_t wr = make_rsrc_n(p.w, (unsigned)((int64_t)p.K * p.N * 2));
    const int drow = lane >> 3, dchunk = (lane & 7) ^ (((drow >> 1) & 3) << 1);
    auto issue = [&](int tl, int kb, int slot) {           // 32 one-kilobyte pieces of block kb of tile tl
      const int tile_t = tl / p.nn_tiles, tile_n = tl - tile_t * p.nn_tiles;
      const int dy_v0 = (tile_t * 128 + drow) * (p.dy_rs * 2) + dchunk * 16 + kb * 128;
      const int w_v0 = (kb * 64 + drow) * (p.N * 2) + tile_n * 256 + dchunk * 16;
      char* base = smem + slot * GF_SLOT;
#pragma unroll
      for (int pc = 0; pc < 16; ++pc)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(dyr, (__attribute__((address_space(3))) void*)(base + pc * 1024), 16,
                                                 dy_v0 + pc * 8 * (p.dy_rs * 2), 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 16; ++q)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (__attribute__((address_space(3))) void*)(base + (16 + q) * 1024), 16,
                                                 w_v0 + (q & 7) * 8 * (p.N * 2) + (q >> 3) * 128, 0, 0, 0);
    };
    if (tile < tile_end) issue(tile, 0, 0);
    int g = 0;
    for (int tl = tile; tl < tile_end; tl += tile_stride)
      for (int kb = 0; kb < nkb; ++kb, ++g) {
        lds_dma_wait();                                    // block g has landed (this wave's vmcnt counts nothing else)
        gf_barrier();                                      // ... and every compute wave is done with block g - 1
        const bool last = kb + 1 == nkb;
        const int ntl = last ? tl + tile_stride : tl;
        if (ntl < tile_end) issue(ntl, last ? 0 : kb + 1, (g + 1) & 1);
      }
    return;
  }

  // ---- compute waves
  const int wt = wave & 1, wn = wave >> 1;
  // A tiles come in pairs (pr, hf) over 32 features: MFMA row j of tile (pr, hf) stands for feature 32 pr + 8 (j / 4) +
  // 4 hf + j % 4, so accumulator rows 4 g .. 4 g + 3 of the pair's two tiles are features 8 g .. 8 g + 7 -- one 16-byte
  // load of h and one 16-byte store of dh per lane and pair (8-byte accesses in 32-byte runs ran stage 1 at 2.9 TB/s).
  // The transposed read allows it: the column a lane RECEIVES is (what lane 4 e + j / 4 pointed at) + j % 4, so the
  // loader lane q = j % 4 ... points at feature 32 pr + 8 q + 4 hf instead of 16 nt + 4 q.
  // One LDS address per operand: the swizzle XORs bits 5-6 of the byte column, so the other fragments' addresses are
  // wtr0 + 8 (tile nt odd: bit 3), wtr0 ^ 64 (second pair), dnat0 ^ (64 ks + 32 hf) -- an XOR at the point of use instead
  // of six registers that live across the tile loop.
  const int wtr0 = gf_off(lg * 4 + (lj >> 2), (lj & 3) * 16);
  const int dnat0 = gf_off(lj, lg * 8);
  const T_* hb = (const T_*)p.h;
  T_* ob = (T_*)p.dh;
  int g = 0;
  for (; tile < tile_end; tile += tile_stride) {
    const int tile_t = tile / p.nn_tiles, tile_n = tile - tile_t * p.nn_tiles;
    const int t0 = tile_t * 128, n0 = tile_n * 128;
    // the epilogue's h values of token tiles 0 and 1: 4 sixteen-byte loads per lane, in flight during the whole K loop
    // (all 8 were, until round 4: 160 VGPRs.  The hardware places the five waves of a workgroup so that two workgroups
    // per CU need 4 waves per SIMD, i.e. <= 128 registers -- measured: at 160 a launch of one workgroup per CU took
    // the same time.  Token tiles 2 and 3 are requested when the epilogue starts, into the K loop's dead fragments.)
    auto load_h = [&](int tt, int pr) {
      const int t = min(t0 + wt * 64 + tt * 16 + lj, p.T - 1);
      return *(const X8*)((const char*)hb + (unsigned)(t * p.h_rs + n0 + wn * 64 + pr * 32 + lg * 8) * 2u);
    };
    X8 ha[2][2];
    if (DGELU) {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) ha[tt][pr] = load_h(tt, pr);
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) acc[nt][tt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int kb = 0; kb < nkb; ++kb, ++g) {
      gf_barrier();                                        // block g is in its slot (the loader waited for it)
      const char* dyb = smem + (g & 1) * GF_SLOT + wt * (64 * 128);
      const char* wb = smem + (g & 1) * GF_SLOT + 16 * 1024 + wn * (64 * 128);
      const int nks = min(2, (p.K - kb * 64) >> 5);
      for (int ks = 0; ks < nks; ++ks) {
        // (transposed reads as inline assembly, lds_tr_issue in vil_mfma_common.h: through the builtin the compiler
        // orders them behind every LDS-DMA request in flight.)  The W^T fragments come in two halves -- the second
        // half's reads are in flight under the first half's MFMAs -- which keeps 16 registers fewer alive.
        X8 bq[4];
        s16x4 ar0[2][2], ar1[2][2];
        const unsigned wa = lds_addr32(wb) + ks * (32 * 128);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) { ar0[nt][0] = lds_tr_issue<0>(wa + wtr0 + 8 * nt); ar0[nt][1] = lds_tr_issue<16 * 128>(wa + wtr0 + 8 * nt); }
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            const X4 d4 = *(const X4*)(dyb + tt * (16 * 128) + (dnat0 ^ (ks * 64 + hf * 32)));
#pragma unroll
            for (int e = 0; e < 4; ++e) bq[tt][hf * 4 + e] = d4[e];
          }
        lds_tr_settle(ar0[0][0], ar0[0][1], ar0[1][0], ar0[1][1]);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) { ar1[nt][0] = lds_tr_issue<0>(wa + (wtr0 ^ 64) + 8 * nt); ar1[nt][1] = lds_tr_issue<16 * 128>(wa + (wtr0 ^ 64) + 8 * nt); }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const X8 a = lds_tr_join<T_>(ar0[nt][0], ar0[nt][1]);
#pragma unroll
          for (int tt = 0; tt < 4; ++tt) acc[nt][tt] = mfma16(a, bq[tt], acc[nt][tt]);
        }
        lds_tr_settle(ar1[0][0], ar1[0][1], ar1[1][0], ar1[1][1]);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const X8 a = lds_tr_join<T_>(ar1[nt][0], ar1[nt][1]);
#pragma unroll
          for (int tt = 0; tt < 4; ++tt) acc[2 + nt][tt] = mfma16(a, bq[tt], acc[2 + nt][tt]);
        }
      }
    }

    // ---- epilogue: dh = acc * gelu'(h); lane (j, g) of tile (nt, tt): row t = .. + 16 tt + j, columns n = .. + 16 nt + 4 g ..+3
    // Order: request h of token tiles 2, 3 -> results of tiles 0, 1 (held) -> claim the four late loads at once -> stores
    // of tiles 0, 1 -> tiles 2, 3.  No wait hipcc places for a load may follow a store of this tile: its counted
    // waits count the untracked stores too and would drain them.
    X8 hl[2][2];
    if (DGELU) {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) hl[tt][pr] = load_h(2 + tt, pr);
    }
    auto result = [&](int tt, int pr, const X8& h) {
      X8 o8;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        o8[r] = (T_)(DGELU ? acc[2 * pr][tt][r] * gelu_grad((float)h[r]) : acc[2 * pr][tt][r]);
        o8[4 + r] = (T_)(DGELU ? acc[2 * pr + 1][tt][r] * gelu_grad((float)h[4 + r]) : acc[2 * pr + 1][tt][r]);
      }
      return o8;
    };
    auto store = [&](int tt, int pr, const X8& o8) {
      const int t = t0 + wt * 64 + tt * 16 + lj;
      if (t < p.T) gf_store16(ob, (unsigned)(t * p.dh_rs + n0 + wn * 64 + pr * 32 + lg * 8) * 2u, o8);
    };
    X8 oe[2][2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) oe[tt][pr] = result(tt, pr, ha[tt][pr]);
    if (DGELU) {   // (the held results are operands too: otherwise the claim -- and its s_waitcnt vmcnt(0) -- is scheduled above them)
      gf_u32x4 q[4], r[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { q[i] = __builtin_bit_cast(gf_u32x4, hl[i >> 1][i & 1]); r[i] = __builtin_bit_cast(gf_u32x4, oe[i >> 1][i & 1]); }
      asm volatile("" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]));
#pragma unroll
      for (int i = 0; i < 4; ++i) { hl[i >> 1][i & 1] = __builtin_bit_cast(X8, q[i]); oe[i >> 1][i & 1] = __builtin_bit_cast(X8, r[i]); }
    }
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) store(tt, pr, oe[tt][pr]);
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) store(2 + tt, pr, result(2 + tt, pr, hl[tt][pr]));
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
  int ntiles;                  // token tiles x nn_tiles
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

// GELU = false: the plain forward h = x W^T + b (vil_gemm_tile_bf16 op 0): one output.
template <bool GELU>
__global__ __launch_bounds__(GF_THREADS, 3) void k_fwd_gelu(FgParams p) {
  typedef __bf16 T_;
  typedef typename V16<T_>::x8 X8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lj = lane & 15, lg = lane >> 4;
  int tile, tile_end, tile_stride;
  gf_tiles(p.ntiles, tile, tile_end, tile_stride);
  const int nkb = (p.K + 63) >> 6;

  if (wave == 4) {                                         // ---- the loader wave (see the schedule comment above)
    __builtin_amdgcn_s_setprio(3);
    const __amdgpu_buffer_rsrc_t xr = make_rsrc_n(p.x, (unsigned)(((int64_t)(p.T - 1) * p.x_rs + p.K) * 2));
    const __amdgpu_buffer_rsrc_t wr = make_rsrc_n(p.w, (unsigned)((int64_t)p.N * p.K * 2));
    const int drow = lane >> 3, dchunk = (lane & 7) ^ (((drow >> 1) & 3) << 1);
    int w_r[2];                                            // weight row (within the tile) of LDS row 8 q + drow, q even / odd
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      const int j = par * 8 + drow;                        // row within a 16-row tile
      w_r[par] = 8 * (j >> 2) + (j & 3);
    }
    auto issue = [&](int tl, int kb, int slot) {
      const int tile_t = tl / p.nn_tiles, tile_n = tl - tile_t * p.nn_tiles;
      const int x_v0 = (tile_t * 128 + drow) * (p.x_rs * 2) + dchunk * 16 + kb * 128;
      const int w_v0 = tile_n * 128 * (p.K * 2) + dchunk * 16 + kb * 128;
      char* base = smem + slot * GF_SLOT;
#pragma unroll
      for (int pc = 0; pc < 16; ++pc)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (__attribute__((address_space(3))) void*)(base + pc * 1024), 16,
                                                 x_v0 + pc * 8 * (p.x_rs * 2), 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int nt = q >> 1;                             // LDS rows 8 q .. 8 q + 7 = rows 8 (q & 1) .. of tile nt (0..7)
        const int frow = (nt >> 2) * 64 + ((nt >> 1) & 1) * 32 + (nt & 1) * 4;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (__attribute__((address_space(3))) void*)(base + (16 + q) * 1024), 16,
                                                 w_v0 + (frow + ((q & 1) ? w_r[1] : w_r[0])) * (p.K * 2), 0, 0, 0);
      }
    };
    if (tile < tile_end) issue(tile, 0, 0);
    int g = 0;
    for (int tl = tile; tl < tile_end; tl += tile_stride)
      for (int kb = 0; kb < nkb; ++kb, ++g) {
        lds_dma_wait();
        gf_barrier();
        const bool last = kb + 1 == nkb;
        const int ntl = last ? tl + tile_stride : tl;
        if (ntl < tile_end) issue(ntl, last ? 0 : kb + 1, (g + 1) & 1);
      }
    return;
  }

  // ---- compute waves
  const int wt = wave & 1, wn = wave >> 1;
  int nat[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) nat[ks] = gf_off(lj, ks * 64 + lg * 16);
  T_* hb = (T_*)p.h;
  T_* ab = (T_*)p.a;
  int g = 0;
  for (; tile < tile_end; tile += tile_stride) {
    const int tile_t = tile / p.nn_tiles, tile_n = tile - tile_t * p.nn_tiles;
    const int t0 = tile_t * 128, n0 = tile_n * 128;
    X8 bias8[2];
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      X8 z = {};
      bias8[pr] = p.bias ? *(const X8*)((const T_*)p.bias + n0 + wn * 64 + pr * 32 + lg * 8) : z;
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) acc[nt][tt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int kb = 0; kb < nkb; ++kb, ++g) {
      gf_barrier();
      const char* xb = smem + (g & 1) * GF_SLOT + wt * (64 * 128);
      const char* wb = smem + (g & 1) * GF_SLOT + 16 * 1024 + wn * (64 * 128);
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
    }

    // ---- epilogue: lane (j, g), pair pr, column tile tt: token t0 + 64 wt + 16 tt + j, features n0 + 64 wn + 32 pr + 8 g .. +7
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      const int t = t0 + wt * 64 + tt * 16 + lj;
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        X8 o8, a8;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          o8[r] = (T_)(acc[2 * pr][tt][r] + (float)bias8[pr][r]);
          o8[4 + r] = (T_)(acc[2 * pr + 1][tt][r] + (float)bias8[pr][4 + r]);
        }
        if (GELU) {
#pragma unroll
          for (int r = 0; r < 8; ++r) a8[r] = (T_)gelu_fwd((float)o8[r]);       // GELU of the rounded pre-activation
        }
        const unsigned o = (unsigned)(t * p.o_rs + n0 + wn * 64 + pr * 32 + lg * 8) * 2u;
        if (t < p.T) {
          gf_store16(hb, o, o8);
          if (GELU) gf_store16(ab, o, a8);
        }
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
  if ((T + 128) * out_row_stride * 2 >= (1ll << 32)) return VIL_E_BACKEND;        // 32-bit byte offsets of the stores
  FgParams p;
  p.x = x; p.w = w; p.bias = bias; p.h = h; p.a = a;
  p.T = (int)T; p.K = K; p.N = N;
  p.x_rs = (int)x_row_stride; p.o_rs = (int)out_row_stride;
  p.nn_tiles = N / 128;
  p.ntiles = (int)(((T + 127) / 128) * p.nn_tiles);
  const size_t lds = 2 * GF_SLOT;
  const unsigned grid = (unsigned)vil_persistent_grid(3, GF_THREADS / 64, lds, 1, (int64_t)p.ntiles * (GF_THREADS / 64));
  if (int he = vil_ensure_dyn_lds((const void*)k_fwd_gelu<true>, lds)) return he;
  k_fwd_gelu<true><<<dim3(grid), dim3(GF_THREADS), lds, (hipStream_t)stream>>>(p);
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
  if ((T + 128) * h_row_stride * 2 >= (1ll << 32) || (T + 128) * dh_row_stride * 2 >= (1ll << 32)) return VIL_E_BACKEND;   // 32-bit byte offsets
  DgParams p;
  p.dy = dy; p.w = w; p.h = h; p.dh = dh;
  p.T = (int)T; p.K = K; p.N = N;
  p.dy_rs = (int)dy_row_stride; p.h_rs = (int)h_row_stride; p.dh_rs = (int)dh_row_stride;
  p.nn_tiles = N / 128;
  p.ntiles = (int)(((T + 127) / 128) * p.nn_tiles);
  const size_t lds = 2 * GF_SLOT;
  const unsigned grid = (unsigned)vil_persistent_grid(3, GF_THREADS / 64, lds, 1, (int64_t)p.ntiles * (GF_THREADS / 64));
  if (int he = vil_ensure_dyn_lds((const void*)k_dgrad_dgelu<true>, lds)) return he;
  k_dgrad_dgelu<true><<<dim3(grid), dim3(GF_THREADS), lds, (hipStream_t)stream>>>(p);
  return (int)hipGetLastError();
}

// The same two kernels without their activation epilogues: op 0: out[T][N] = in[T][K] . w[N][K]^T (+ bias[N]); op 1:
// out[T][N] = in[T][K] . w[K][N] (the input gradient; w = the weight as stored).  bf16, fp32 accumulate, row strides in
// elements.  K % 32 == 0, N % 128 == 0; VIL_E_BACKEND outside that contract (the caller then uses the library GEMM).
// For the projections of the dense stages (T = 6 k ... 25 k, K, N = 384 ... 3072), where the tuned hipBLASLt kernels run at
// 13 - 31 % of the matrix peak.
extern "C" int vil_gemm_tile_bf16(int op, const void* in, const void* w, const void* bias, void* out, int64_t T, int K, int N,
                                  int64_t in_row_stride, int64_t out_row_stride, void* stream) {
  if (!in || !w || !out) return VIL_E_NULL;
  if (T <= 0 || K <= 0 || N <= 0 || op < 0 || op > 1 || (op == 1 && bias)) return VIL_E_SHAPE;
  if ((K & 31) || (N & 127)) return VIL_E_BACKEND;
  if ((in_row_stride & 7) || (out_row_stride & 7) || (((uintptr_t)in | (uintptr_t)w | (uintptr_t)out) & 15) ||
      (bias && ((uintptr_t)bias & 15))) return VIL_E_ALIGN;
  if (in_row_stride < K || out_row_stride < N) return VIL_E_SHAPE;
  if ((T + 128) * in_row_stride * 2 >= (1ll << 31) || (int64_t)K * N * 2 >= (1ll << 31) || T * (N / 128) >= (1ll << 30)) return VIL_E_BACKEND;
  if ((T + 128) * out_row_stride * 2 >= (1ll << 32)) return VIL_E_BACKEND;
  const size_t lds = 2 * GF_SLOT;
  const int nn_tiles = N / 128, ntiles = (int)(((T + 127) / 128) * nn_tiles);
  const unsigned grid = (unsigned)vil_persistent_grid(3, GF_THREADS / 64, lds, 1, (int64_t)ntiles * (GF_THREADS / 64));
  if (op == 0) {
    FgParams p;
    p.x = in; p.w = w; p.bias = bias; p.h = out; p.a = nullptr;
    p.T = (int)T; p.K = K; p.N = N;
    p.x_rs = (int)in_row_stride; p.o_rs = (int)out_row_stride;
    p.nn_tiles = nn_tiles; p.ntiles = ntiles;
    if (int he = vil_ensure_dyn_lds((const void*)k_fwd_gelu<false>, lds)) return he;
    k_fwd_gelu<false><<<dim3(grid), dim3(GF_THREADS), lds, (hipStream_t)stream>>>(p);
  } else {
    DgParams p;
    p.dy = in; p.w = w; p.h = nullptr; p.dh = out;
    p.T = (int)T; p.K = K; p.N = N;
    p.dy_rs = (int)in_row_stride; p.h_rs = (int)out_row_stride; p.dh_rs = (int)out_row_stride;
    p.nn_tiles = nn_tiles; p.ntiles = ntiles;
    if (int he = vil_ensure_dyn_lds((const void*)k_dgrad_dgelu<false>, lds)) return he;
    k_dgrad_dgelu<false><<<dim3(grid), dim3(GF_THREADS), lds, (hipStream_t)stream>>>(p);
  }
  return (int)hipGetLastError();
}
