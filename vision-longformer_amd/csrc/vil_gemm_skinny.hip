// vil_gemm_skinny.hip -- forward GEMM + bias of the projections whose contraction is short and whose token count is
// huge (stages 1-2 of ViL: T = 100 000 ... 400 000 rows, K = C = 96 / 192, N <= 768):
//     out[t][n] = sum_k in[t][k] * w[n][k] + bias[n]          (nn.Linear forward, bf16 in / out, fp32 accumulate)
// These GEMMs are HBM-bound (read T*K, write T*N 16-bit values for 2*T*K*N flops: 30-70 flop/byte) and hipBLASLt's tiled
// kernels run them at 2.0-2.9 TB/s because every 128/256-wide output tile re-streams its operands (tools/gemm_bench.py:
// 140 us for the 385 MB of stage 1's fc1).  Here the whole weight matrix lives in REGISTERS for the lifetime of a
// persistent workgroup -- wave (wn) holds the A fragments of its 96 output features for every k -- so the only traffic
// is the activations in (LDS-DMA ring, one 128-row tile per slot) and the result out:
//   * C^T orientation (D[n][t] = W rows x in rows): a lane ends up with 8 CONSECUTIVE output features of one token
//     (the weight rows a 16-row MFMA tile covers are chosen so: rows {8g..8g+3} in the even tile of a pair,
//     {8g+4..8g+7} in the odd one) -> one 16-byte store per lane and tile pair, bias added from registers;
//   * the in-tile of a step is read from LDS by every wave (B fragments, ds_read_b128 from the XOR-swizzled 128-byte-row
//     image of vil_attn_dense.hip, one image per 64 k);
//   * one extra wave per workgroup is a LOADER (round 4): it issues every LDS-DMA request of the ring and owns their
//     completion (a counted s_waitcnt vmcnt in front of the barrier that publishes a tile; its vmcnt sees loads only).
//     Until round 4 every wave requested its share of the next tiles and waited for it with s_waitcnt vmcnt(0) before
//     the tile's first store -- which also waits for the PREVIOUS tile's stores (gfx9: one counter for loads and
//     stores), and the __syncthreads() that published a tile carried a release fence, i.e. another vmcnt(0).  With a K
//     loop of ~0.5 us per tile and ~2 us for a store to drain, the waves spent 73 % of their time in those waits
//     (profiles/r04_pipe_utilisation.txt: stage 1's qkv projection ran at 1.96 TB/s).  The compute waves now never
//     execute s_waitcnt vmcnt inside the tile loop and the tile barrier is a bare s_barrier.
#include "vil_mfma_common.h"

__device__ __forceinline__ void sk_barrier() { asm volatile("s_barrier" ::: "memory"); }

struct SkParams {
  const void* in; const void* w; const void* bias; void* out;
  void* out2;                 // GELU instantiations: gelu(out), same layout
  int T, K, N;
  int in_rs, out_rs;          // row strides (elements)
  int n0, n1;                 // the output features [n0, n1) this launch computes (the whole row unless N needs > 7 waves)
  int wn, wt;                 // waves along n (32 * NP features each) x waves along t
  int ntiles;                 // ceil(T / RT)
};

// exact (erf) GELU of nn.GELU() (reference msvit.py:21): x Phi(x), erfc by Abramowitz-Stegun 7.1.26 (|error| < 1.5e-7);
// the negative side is x erfc(|z|) / 2 directly -- no 1 - erf cancellation, the sign of the result is the sign of x
__device__ __forceinline__ float sk_gelu(float x) {
  const float z = x * 0.70710678118654752f, az = __builtin_fabsf(z);
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, az, 1.0f));
  const float e = __builtin_amdgcn_exp2f(-(z * z) * LOG2E);
  float q = __builtin_fmaf(t, 1.061405429f, -1.453152027f);
  q = __builtin_fmaf(q, t, 1.421413741f);
  q = __builtin_fmaf(q, t, -0.284496736f);
  q = __builtin_fmaf(q, t, 0.254829592f);
  const float hc = 0.5f * (q * t) * e;                    // erfc(|z|) / 2
  return x * (x < 0.f ? hc : 1.0f - hc);
}

__device__ __forceinline__ int sk_off(int row, int colb) { return row * 128 + (colb ^ (((row >> 1) & 3) << 5)); }

// KS = K / 32; NP = tile pairs (32 output features each) per wave; TT = 16-row column tiles per inner chunk; RT = token
// rows per tile (ring slot); NSLOT ring slots.  Registers: weights 8 NP KS, accumulators 8 NP TT, (+ 4 TT, + 4 NP).
template <int KS, int NP, int TT, int RT, int NSLOT, bool NN, bool GELU = false>
__global__ __launch_bounds__(512, 2) void k_skinny(SkParams p) {
  typedef __bf16 T_;
  typedef typename V16<T_>::x8 X8;
  constexpr int KB = (KS + 1) / 2;                 // 64-k images per tile
  constexpr int IMG = RT * 128;                    // bytes of one image
  constexpr int SLOT = KB * IMG;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, nthr = blockDim.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = nthr >> 6;   // nwaves - 1 compute waves + the loader
  const int lj = lane & 15, lg = lane >> 4;
  const int wni = wave % p.wn, wti = wave / p.wn;
  const int n_base = p.n0 + wni * (32 * NP);
  const int rows_w = RT / p.wt;                    // token rows of a tile this wave computes
  const int r_base = wti * rows_w;

  // ---- this wave's weight fragments and bias: loaded once
  const T_* wb = (const T_*)p.w;
  X8 afr[NP][2][KS];
  X8 bias8[NP];
  if (!NN) {
    // w[n][k]: a fragment (row n, 8 consecutive k) is one 16-byte load
#pragma unroll
    for (int pr = 0; pr < NP; ++pr)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int n = n_base + 32 * pr + 8 * (lj >> 2) + 4 * hf + (lj & 3);      // the feature MFMA row lj of this tile stands for
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          X8 z = {};
          afr[pr][hf][ks] = n < p.n1 ? *(const X8*)(wb + (int64_t)n * p.K + ks * 32 + lg * 8) : z;
        }
      }
  } else {
    // w[k][n] (the input gradient: w = weight as stored, K = its rows): fragments are k-strided.  The matrix passes
    // through the (not yet used) ring area in chunks of whole 32-k groups, copied linearly by LDS-DMA, and is read
    // with ds_read_b64_tr_b16: lane (g, 4e + q) points at row 8g + 4h + e, features 8q + 4hf .. +3 of its pair -- after
    // the instruction's 4 x 4 exchange lane L holds rows 8g + 4h .. +3 of feature 8 (L/4) + 4hf + L%4, the A layout.
    const int row_b = p.N * 2;
    const int cap_rows = ((NSLOT * SLOT) / row_b - 2) & ~31;              // rows of w the ring area holds (one row of slack)
    const __amdgpu_buffer_rsrc_t wrs = make_rsrc_n(p.w, (unsigned)((int64_t)p.K * row_b));
    for (int k0 = 0; k0 < 32 * KS; k0 += cap_rows) {
      const int rows = min(cap_rows, 32 * KS - k0);
      const int bytes = rows * row_b;
      __syncthreads();
      for (int off = wave * 1024; off < bytes + row_b; off += nwaves * 1024)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)(smem + off), 16,
                                                 k0 * row_b + off + lane * 16, 0, 0, 0);
      lds_dma_wait();
      __syncthreads();
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks * 32 >= k0 && ks * 32 < k0 + rows) {
#pragma unroll
          for (int pr = 0; pr < NP; ++pr)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
              for (int jh = 0; jh < 2; ++jh) {
                const int row = ks * 32 - k0 + 8 * lg + 4 * jh + (lj >> 2);
                const int n = n_base + 32 * pr + 8 * (lj & 3) + 4 * hf;
                const typename V16<T_>::x4 t4 = __builtin_bit_cast(typename V16<T_>::x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (s16x4 __attribute__((address_space(3)))*)(smem + row * row_b + min(n, p.N - 4) * 2)));
#pragma unroll
                for (int e = 0; e < 4; ++e) afr[pr][hf][ks][jh * 4 + e] = t4[e];
              }
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int pr = 0; pr < NP; ++pr) {
    X8 z = {};
    const int n8 = n_base + 32 * pr + 8 * lg;
    bias8[pr] = (p.bias && n8 < p.n1) ? *(const X8*)((const T_*)p.bias + n8) : z;
  }

  // ---- in-tile ring (LDS-DMA): piece = 8 rows x 128 bytes of one 64-k image
  const int first = blockIdx.x;
  if (wave == nwaves - 1) {                                // ---- the loader wave
    __builtin_amdgcn_s_setprio(3);
    constexpr int PIECES = KB * (RT / 8);                  // requests per tile
    static_assert(PIECES <= 63, "counted wait");
    const __amdgpu_buffer_rsrc_t irs = make_rsrc_n(p.in, (unsigned)(((int64_t)(p.T - 1) * p.in_rs + p.K) * 2));
    const int drow = lane >> 3, dslot = lane & 7, dchunk = dslot ^ (((drow >> 1) & 3) << 1);
    const int in_v0 = drow * (p.in_rs * 2) + dchunk * 16;
    auto issue = [&](int tile, int slot) {
      char* base = smem + slot * SLOT;
      const int t0 = tile * RT;
#pragma unroll 4
      for (int pc = 0; pc < PIECES; ++pc) {
        const int kb = pc / (RT / 8), pp = pc - kb * (RT / 8);
        // (the last image of an odd KS holds 32 k: its upper four chunks per row are not requested)
        if (kb * 2 + 1 < KS || dchunk < 4)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(irs, (__attribute__((address_space(3))) void*)(base + kb * IMG + pp * 1024),
                                                   16, in_v0 + (t0 + pp * 8) * (p.in_rs * 2) + kb * 128, 0, 0, 0);
      }
    };
    lds_dma_wait();                                        // (this wave's share of the weight loads above)
    for (int a = 0; a < NSLOT - 1; ++a)
      if (first + a * (int)gridDim.x < p.ntiles) issue(first + a * gridDim.x, a);
    int it = 0;
    for (int tile = first; tile < p.ntiles; tile += gridDim.x, ++it) {
      // tile it has landed: of this wave's requests only those of tile it + 1 (NSLOT - 2 tiles ahead) may be in flight
      if (NSLOT > 2 && tile + (int)gridDim.x < p.ntiles) lds_dma_wait<PIECES*(NSLOT > 2 ? 1 : 0)>();
      else lds_dma_wait();
      sk_barrier();                                        // ... and every compute wave is done with tile it - 1
      const int nxt = tile + (NSLOT - 1) * gridDim.x;
      if (nxt < p.ntiles) issue(nxt, (it + NSLOT - 1) % NSLOT);
    }
    return;
  }
  int bnat[2];
#pragma unroll
  for (int k2 = 0; k2 < 2; ++k2) bnat[k2] = sk_off(lj, k2 * 64 + lg * 16);

  T_* ob = (T_*)p.out;
  int it = 0;
  for (int tile = first; tile < p.ntiles; tile += gridDim.x, ++it) {
    const int slot = it % NSLOT;
    sk_barrier();                                          // tile it is in its slot (the loader waited for it)
    const char* tb = smem + slot * SLOT;
    const int t0 = tile * RT;
    if (n_base < p.n1) {
      for (int r0 = r_base; r0 < r_base + rows_w; r0 += 16 * TT) {     // 16*TT token rows at a time
        const int ntt = min(TT, (r_base + rows_w - r0) >> 4);
        f32x4 acc[NP][2][TT];
#pragma unroll
        for (int pr = 0; pr < NP; ++pr)
#pragma unroll
          for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) acc[pr][hf][tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          X8 bq[TT];
#pragma unroll
          for (int tt = 0; tt < TT; ++tt)
            bq[tt] = *(const X8*)(tb + (ks >> 1) * IMG + (r0 + tt * 16) * 128 + bnat[ks & 1]);
#pragma unroll
          for (int pr = 0; pr < NP; ++pr)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
              for (int tt = 0; tt < TT; ++tt) acc[pr][hf][tt] = mfma16(afr[pr][hf][ks], bq[tt], acc[pr][hf][tt]);
        }
        // epilogue: lane (j, g) of pair pr, column tile tt: token t0 + r0 + 16 tt + j, features n_base + 32 pr + 8 g .. +7
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
          const int t = t0 + r0 + tt * 16 + lj;
          if (tt < ntt && t < p.T) {
#pragma unroll
            for (int pr = 0; pr < NP; ++pr) {
              const int n8 = n_base + 32 * pr + 8 * lg;
              if (n8 < p.n1) {
                X8 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  o[r] = (T_)(acc[pr][0][tt][r] + (float)bias8[pr][r]);
                  o[4 + r] = (T_)(acc[pr][1][tt][r] + (float)bias8[pr][4 + r]);
                }
                *(X8*)(ob + (int64_t)t * p.out_rs + n8) = o;
                if (GELU) {
                  // the activation of the ROUNDED pre-activation, as the unfused pair (bf16 Linear output -> nn.GELU) computes it
                  X8 a;
#pragma unroll
                  for (int r = 0; r < 8; ++r) a[r] = (T_)sk_gelu((float)o[r]);
                  *(X8*)((T_*)p.out2 + (int64_t)t * p.out_rs + n8) = a;
                }
              }
            }
          }
        }
      }
    }
  }
}

template <int KS, int NP, int TT, int RT, int NSLOT, bool NN, bool GELU = false>
static int sk_launch(SkParams& p, int ncu, hipStream_t s) {
  // 7 compute waves + the loader = 8 waves of 256 registers: a row of more than 7 x 32 NP features (N = 768 at 96 per
  // wave) is computed in two launches over the two halves of the features
  const int wn_all = (p.N + 32 * NP - 1) / (32 * NP);
  const int nwin = wn_all > 7 ? 2 : 1;
  if (wn_all > 14) return VIL_E_BACKEND;
  const int wn_first = (wn_all + nwin - 1) / nwin;
  for (int win = 0; win < nwin; ++win) {
    p.n0 = win * wn_first * (32 * NP);
    p.n1 = win + 1 < nwin ? (win + 1) * wn_first * (32 * NP) : p.N;
    p.wn = (p.n1 - p.n0 + 32 * NP - 1) / (32 * NP);
    // waves along t: as many as the tile's 16*TT-row chunks and the 7 compute waves allow (K <= 192: 4 / 2 / 2 / 1 for
    // 1 / 2 / 3 / >= 4 waves along n)
    const int by_rows = RT / (16 * TT), by_waves = 7 / p.wn;
    p.wt = KS <= 6 ? (p.wn == 1 ? 4 : (p.wn <= 3 ? 2 : 1)) : (by_rows < by_waves ? by_rows : by_waves);
    if (p.wt < 1) p.wt = 1;
    p.ntiles = (p.T + RT - 1) / RT;
    const size_t lds = (size_t)NSLOT * ((KS + 1) / 2) * RT * 128;
    // resident workgroups per CU: by LDS, and by waves (the kernel is compiled for 2 waves per SIMD)
    const int wg_per_cu = (lds * 2 <= 160 * 1024 && 2 * (p.wn * p.wt + 1) <= 8) ? 2 : 1;
    int grid = ncu * wg_per_cu;
    if (grid > p.ntiles) grid = p.ntiles;
    if (int he = vil_ensure_dyn_lds((const void*)k_skinny<KS, NP, TT, RT, NSLOT, NN, GELU>, lds)) return he;
    k_skinny<KS, NP, TT, RT, NSLOT, NN, GELU><<<dim3(grid), dim3(64 * (p.wn * p.wt + 1)), lds, s>>>(p);
    if (int he = (int)hipGetLastError()) return he;
  }
  return 0;
}

// CU count of the current device, queried once per device (a device-attribute query is not a legal call while another
// thread's stream capture is in global mode: the persistent grid size must not cost an API call per launch)
static int sk_cu_count() {
  static int cached[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cached[dev] == 0) {
    int v = 0;
    cached[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  return cached[dev];
}

// op 0: out[T][N] = in[T][K] . w[N][K]^T (+ bias[N])   (nn.Linear forward)
// op 1: out[T][N] = in[T][K] . w[K][N]                  (its input gradient: w = the weight as stored, K = out_features)
// bf16, fp32 accumulate.  VIL_E_BACKEND outside the table below (the caller then uses the library GEMM).
//   K     features / wave   rows / tile   ring        stages it serves (ViL: C = 96, 192)
//   96      96                 128        3 x 32 KB   qkv / proj / fc1 forward and proj input gradient at C = 96
//   192     96                 128        3 x 48 KB   the same at C = 192
//   288     32                  64        3 x 40 KB   qkv input gradient at C = 96
//   384     32                  64        3 x 48 KB   fc2 forward, fc1 input gradient at C = 96
//   576     32                  32        3 x 36 KB   qkv input gradient at C = 192
//   768     32                  32        3 x 48 KB   fc2 forward, fc1 input gradient at C = 192
extern "C" int vil_gemm_skinny_bf16(int op, const void* in, const void* w, const void* bias, void* out, int64_t T, int K, int N,
                                    int64_t in_row_stride, int64_t out_row_stride, void* stream) {
  if (!in || !w || !out) return VIL_E_NULL;
  if (T <= 0 || K <= 0 || N <= 0 || op < 0 || op > 1 || (op == 1 && bias)) return VIL_E_SHAPE;
  if ((N & 7) || N > 768) return VIL_E_BACKEND;
  if ((in_row_stride & 7) || (out_row_stride & 7) || (((uintptr_t)in | (uintptr_t)w | (uintptr_t)out) & 15) ||
      (bias && ((uintptr_t)bias & 15))) return VIL_E_ALIGN;
  if ((T + 128) * in_row_stride * 2 >= (1ll << 31)) return VIL_E_BACKEND;
  SkParams p;
  p.in = in; p.w = w; p.bias = bias; p.out = out; p.out2 = nullptr;
  p.T = (int)T; p.K = K; p.N = N;
  p.in_rs = (int)in_row_stride; p.out_rs = (int)out_row_stride;
  const int ncu = sk_cu_count();
  hipStream_t s = (hipStream_t)stream;
#define SK_CASE(KK, KS, NP, TT, RT, NSLOT)                                                   \
  if (K == KK) return op ? sk_launch<KS, NP, TT, RT, NSLOT, true>(p, ncu, s) : sk_launch<KS, NP, TT, RT, NSLOT, false>(p, ncu, s);
  SK_CASE(96, 3, 3, 4, 128, 3)
  SK_CASE(192, 6, 3, 2, 128, 3)
  if (N > 256) return VIL_E_BACKEND;              // (32 features per wave from here on)
  SK_CASE(288, 9, 1, 2, 64, 3)
  SK_CASE(384, 12, 1, 2, 64, 3)
  SK_CASE(576, 18, 1, 2, 32, 3)
  SK_CASE(768, 24, 1, 2, 32, 3)
#undef SK_CASE
  return VIL_E_BACKEND;
}

// out[T][N] = in[T][K] . w[N][K]^T + bias[N] and act[T][N] = gelu(out) (exact erf form, of the rounded bf16 out) in ONE
// launch: fc1 of the MLP block with nn.GELU in its epilogue (reference msvit.py:17-34).  K = 96 / 192 only (the stages
// whose fc1 this kernel family serves); the two outputs share the row stride.
extern "C" int vil_gemm_skinny_gelu_bf16(const void* in, const void* w, const void* bias, void* out, void* act, int64_t T, int K,
                                         int N, int64_t in_row_stride, int64_t out_row_stride, void* stream) {
  if (!in || !w || !out || !act) return VIL_E_NULL;
  if (T <= 0 || K <= 0 || N <= 0) return VIL_E_SHAPE;
  if ((N & 7) || N > 768 || (K != 96 && K != 192)) return VIL_E_BACKEND;
  if ((in_row_stride & 7) || (out_row_stride & 7) || (((uintptr_t)in | (uintptr_t)w | (uintptr_t)out | (uintptr_t)act) & 15) ||
      (bias && ((uintptr_t)bias & 15))) return VIL_E_ALIGN;
  if ((T + 128) * in_row_stride * 2 >= (1ll << 31)) return VIL_E_BACKEND;
  SkParams p;
  p.in = in; p.w = w; p.bias = bias; p.out = out; p.out2 = act;
  p.T = (int)T; p.K = K; p.N = N;
  p.in_rs = (int)in_row_stride; p.out_rs = (int)out_row_stride;
  const int ncu = sk_cu_count();
  hipStream_t s = (hipStream_t)stream;
  return K == 96 ? sk_launch<3, 3, 4, 128, 3, false, true>(p, ncu, s) : sk_launch<6, 3, 2, 128, 3, false, true>(p, ncu, s);
}
