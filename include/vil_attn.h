/* vil_attn.h -- C ABI of libvilattn.so: MI355X (gfx950) kernels for Vision
 * Longformer's 2-D sliding-chunk local+global attention (the `longformerhand`
 * path of microsoft/vision-longformer).
 *
 * The reference has no native code and no FFI (it is pure PyTorch), so this ABI
 * is build-defined.  Each entry point names the reference Python it replaces
 * (paths relative to the reference repository):
 *
 *   vil_attn_fwd   replaces, for the local-query rows of
 *                  Long2DSCSelfAttention.forward (src/models/layers/longformer2d.py:134-204):
 *                  chunk/pad/unfold (:134-149), local->global scores (:152-153),
 *                  slidingchunk_2d QK^T (:157 -> slidingchunk_2d.py:26-79), bias add
 *                  (:159-178), mask_invalid_locations (:180 -> slidingchunk_2d.py:321-357),
 *                  concat+softmax (:183-186), slidingchunk_2d attn.V (:195 ->
 *                  slidingchunk_2d.py:82-130), global-V term (:197-200), un-chunk+crop (:201-204).
 *   vil_attn_bwd   replaces SlidingChunk2D.backward (slidingchunk_2d.py:234-246,
 *                  slidingchunk_agrad :132-200) plus the autograd of the softmax /
 *                  bias gather / mask in between.
 *
 * Conventions: plain pointers and sizes only; all device buffers (inputs,
 * outputs, workspace) are caller-owned; calls are asynchronous on `stream`
 * (a hipStream_t passed as void*), never allocate device memory and never
 * synchronise -- the documented exceptions synchronise by design:
 * vil_gemm_tune, vil_linear_wgrad_tune and vil_attn_profile_end.  Process-global
 * state: the profiling sink, the GEMM and weight-gradient plan caches (described
 * at their entry points) and the record of which kernels already had their
 * dynamic-LDS limit raised.  The library reads no environment variable.
 * Return value: 0 = success, negative = argument error (VIL_E_*), positive =
 * hipError_t of the failing launch.
 *
 * Tensor layout: q is addressed as q[b*q_sb + i*q_st + h*q_sh + d] with
 * i in [0, nx*ny) the local token (row-major r*ny+c), d in [0, M) contiguous.
 * k and v are addressed with token j in [0, G + nx*ny): the G global tokens
 * come first (rows 0..G-1), locals follow -- i.e. the views of the reference's
 * `kv(x)` output are passed without copies.  Strides are in elements.
 */
#ifndef VIL_ATTN_H
#define VIL_ATTN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VIL_ATTN_ABI_VERSION 2

enum { VIL_DTYPE_F32 = 0, VIL_DTYPE_BF16 = 1, VIL_DTYPE_F16 = 2, VIL_DTYPE_F64 = 3 /* operator-level entry points only */ };

enum {
  VIL_OK = 0,
  VIL_E_NULL = -1,        /* required pointer is NULL                      */
  VIL_E_SHAPE = -2,       /* non-positive / inconsistent sizes             */
  VIL_E_HEAD_DIM = -3,    /* head_dim M not supported by any kernel        */
  VIL_E_WINDOW = -4,      /* W not supported                               */
  VIL_E_MODE = -5,        /* mode not in [-1, 8]                           */
  VIL_E_EXACT = -6,       /* exact not in {-1,0,1}, or exact==1 with mode!=0
                             (the reference raises ValueError there,
                             slidingchunk_2d.py:331-343)                    */
  VIL_E_DTYPE = -7,
  VIL_E_ALIGN = -8,       /* pointer/stride alignment unusable             */
  VIL_E_WORKSPACE = -9,   /* workspace NULL although bytes > 0             */
  VIL_E_BACKEND = -10     /* requested kernel family cannot run this desc  */
};

/* kernel family selection (desc.backend) */
enum { VIL_BACKEND_AUTO = 0, VIL_BACKEND_SCALAR = 1, VIL_BACKEND_MFMA = 2,
       VIL_BACKEND_MFMA_WAVE = 3, /* the matrix-core family's wave-per-chunk kernels only (rounds 1-5; the chunk-workgroup
                                    kernels of round 6 are skipped): the ablation row of tools/attn_ab.py */
       VIL_BACKEND_MFMA_CW = 4   /* the matrix-core family with the chunk-workgroup forward kernels wherever they can run
                                    (AUTO / MFMA take them only where they measured faster) */ };

typedef struct VilAttnDesc {
  int32_t B, H, M;          /* images, heads, head_dim                                   */
  int32_t nx, ny;           /* local token grid (rows, cols)                             */
  int32_t W;                /* one-sided window = chunk side (arch flag f)               */
  int32_t G;                /* number of global tokens = leading rows of k/v             */
  int32_t mode;             /* 0: 3x3 chunks, -1: own chunk, 1..8: own + one neighbour   */
  int32_t exact;            /* 0 zero-pad chunks, -1 cyclic chunks, 1 exact window       */
  int32_t dtype;            /* VIL_DTYPE_F32 / _BF16 / _F16 of q/k/v/out/dout/dq/dk/dv    */
  int32_t only_glo;         /* local rows attend the global tokens only                  */
  int32_t backend;          /* VIL_BACKEND_*                                             */
  float   scale;            /* softmax scale; scores = scale*q.k + bias                  */
  int32_t bias_side;        /* side S of the (S*S, H) bias table; 0 = 4W-1 (the sliding-chunk module).
                               S = 2W-1 with mode -1 and W = grid side is the dense `Attention` of the
                               s0 stages (reference msvit.py:37-120): one chunk = the whole image      */
  int64_t q_sb, q_st, q_sh; /* element strides (batch, token, head)                      */
  int64_t k_sb, k_st, k_sh;
  int64_t v_sb, v_st, v_sh;
  int64_t o_sb, o_st, o_sh;       /* out                                                 */
  int64_t do_sb, do_st, do_sh;    /* dout                                                */
  int64_t dq_sb, dq_st, dq_sh;
  int64_t dk_sb, dk_st, dk_sh;
  int64_t dv_sb, dv_st, dv_sh;
  const int32_t* mode_dev;  /* NULL, or (random shift, mode in 1..8) a DEVICE int32 holding the neighbour 1..8 the
                               kernels read at launch time instead of `mode`: lets a captured hipGraph draw a new
                               neighbour per replay (reference longformer2d.py:114-123).  MFMA family only.        */
} VilAttnDesc;

int vil_attn_abi_version(void);
const char* vil_attn_strerror(int code);

/* 0 if the descriptor can be run (by desc->backend, or by any backend for
 * AUTO), else the VIL_E_* the launch would return. */
int vil_attn_check(const VilAttnDesc* d);

/* bytes of scratch the forward (pass=0) / backward (pass=1) needs */
size_t vil_attn_workspace_bytes(const VilAttnDesc* d, int pass);

/* out[b,i,h,:] = softmax_j(scale*q_i.k_j + bias_ij | allowed keys) v_j ;
 * lse[(b*H+h)*Nloc + i] = natural-log-sum-exp of row i (float32).
 * bias_table: float32 (S*S, H) row-major, S = bias_side or 4W-1, entry (dx+(S-1)/2)*S + dy+(S-1)/2, or NULL (rpe off);
 * g2l: float32 (H, G) = g2l_relative_position_bias[1] or NULL. */
int vil_attn_fwd(const VilAttnDesc* d, const void* q, const void* k, const void* v,
                 const float* bias_table, const float* g2l,
                 void* out, float* lse, void* workspace, void* stream);

/* dq/dk/dv are fully overwritten (every row of dk/dv, including the G global
 * rows, which receive the local rows' contribution).  dbias_table ((4W-1)^2,H)
 * and dg2l (H,G) are float32, overwritten; they may be NULL when bias_table /
 * g2l are NULL. */
int vil_attn_bwd(const VilAttnDesc* d, const void* q, const void* k, const void* v,
                 const void* out, const void* dout, const float* lse,
                 const float* bias_table, const float* g2l,
                 void* dq, void* dk, void* dv, float* dbias_table, float* dg2l,
                 void* workspace, void* stream);

/* ---- global-token QUERY rows (reference longformer2d.py:210-227): every one of
 * the G global tokens attends all G+Nloc keys, bias g2g[h][g][g'] on global keys and g2l0[h][g]
 * (= g2l_relative_position_bias[0]) on local keys.  q_g / out_g / dout_g / dq_g point at row 0 of
 * (B, G, H*M) views addressed with the descriptor's q_ / o_ / do_ / dq_ strides; lse_g is (B,H,G).
 * The backward ACCUMULATES into dk / dv (all G+Nloc rows; call it after vil_attn_bwd wrote them)
 * and into dg2g / dg2l0 (caller zero-initialises), and overwrites dq_g.  G <= 4. */
int vil_glo_attn_fwd(const VilAttnDesc* d, const void* q_g, const void* k, const void* v,
                     const float* g2g, const float* g2l0, void* out_g, float* lse_g, void* stream);
int vil_glo_attn_bwd(const VilAttnDesc* d, const void* q_g, const void* k, const void* v,
                     const void* out_g, const void* dout_g, const float* lse_g,
                     const float* g2g, const float* g2l0, void* dq_g, void* dk, void* dv,
                     float* dg2g, float* dg2l0, void* stream);

/* ---- launch-shape hook of the chunk-workgroup kernels (round 6; tools/cw_check.py and the launch-shape invariance test only;
 * process-global, the product never calls it).  streams: image streams per column and XCD (0 = the library's cost model);
 * shape_code = chunks per workgroup (1..4) + 10 * query tiles per wave (1, 2) + 100 * heads per workgroup, any digit 0 = the
 * library's own choice.  Results do not depend on either (tests/test_gpu_1_cw.py); VIL_E_SHAPE for values out of range. */
int vil_attn_cw_set_shape(int streams, int shape_code);
/* The launch plan the chunk-workgroup forward would use for *d (host only: no launch, no device memory).  out24 = {segments,
 * workgroups per XCD, largest stream count, chunks per workgroup, head groups, chunk groups, by_image, chunks per (image,
 * head), then four times (first chunk group, chunk groups, image streams, first workgroup) -- one entry per segment of chunk
 * groups with equal work per image (interior / edge / corner chunks of the 3x3 neighbourhood)}.  tests/test_cabi_cpu.py
 * checks that the plan covers every (image, head, chunk) exactly once. */
int vil_attn_cw_plan(const VilAttnDesc* d, int32_t* out24);

/* ---- whole-layer forward (round 5): local rows AND the global token's query row from ONE pass over K / V.  The global
 * query rides in the forward kernel as a spare query column of every chunk, live against the chunk's own keys; a small
 * merge launch combines the chunks' partials with the global key's term into out_all[:, 0] and lse_g -- what
 * vil_attn_fwd followed by vil_glo_attn_fwd computes (reference longformer2d.py:134-227), without streaming K / V a
 * second time.  q_all / out_all point at TOKEN 0 of (B, G+Nloc, H*M) views with the descriptor's strides; g2l is the
 * reference's (2,H,G) g2l_relative_position_bias, g2g (H,G,G); lse (B,H,Nloc), lse_g (B,H,G).  G == 1, 16-bit I/O, MFMA
 * family, a free query slot in the chunk's last wave (every W but 8 at head_dim <= 32): VIL_E_BACKEND otherwise -- call
 * vil_attn_fwd + vil_glo_attn_fwd instead.  Workspace: vil_attn_workspace_bytes(d, 0). */
int vil_attn_fwd_full(const VilAttnDesc* d, const void* q_all, const void* k, const void* v,
                      const float* bias_table, const float* g2l, const float* g2g,
                      void* out_all, float* lse, float* lse_g, void* workspace, void* stream);

/* ---- whole-layer backward: local rows AND the G global-token query rows in ONE call (MFMA family;
 * VIL_E_BACKEND otherwise -- call vil_attn_bwd + vil_glo_attn_bwd instead).  q_all / out_all /
 * dout_all / dq_all point at TOKEN 0 (the global rows) of (B, G+Nloc, H*M) views with the descriptor's
 * strides; g2l / dg2l are the reference's (2,H,G) g2l_relative_position_bias ([0]: global query ->
 * local keys, [1]: local query -> global keys), g2g / dg2g (H,G,G).  The global rows' backward
 * (autograd of longformer2d.py:210-227) is computed inside the dK/dV pass from the K / V fragments it
 * already holds, so dk / dv are written once.  dg2l and dg2g are overwritten (the call zeroes them in its
 * prologue launch before accumulating).  Workspace:
 * vil_attn_workspace_bytes(d, 1).  1 <= G <= 4. */
int vil_attn_bwd_full(const VilAttnDesc* d, const void* q_all, const void* k, const void* v,
                      const void* out_all, const void* dout_all, const float* lse, const float* lse_g,
                      const float* bias_table, const float* g2l, const float* g2g,
                      void* dq_all, void* dk, void* dv, float* dbias_table, float* dg2l, float* dg2g,
                      void* workspace, void* stream);

/* ---- input gradient of the MLP's second Linear with the backward of the exact (erf) GELU in its epilogue (SURVEY.md 8f
 * row 3; reference src/models/msvit.py:17-34, fc1 -> nn.GELU -> fc2):
 *   dh[t][n] = (sum_k dy[t][k] * w[k][n]) * gelu'(h[t][n]),  gelu'(x) = Phi(x) + x phi(x)
 * bf16 in / out, fp32 accumulate; w = fc2.weight (K = out_features rows, N = hidden columns, row-major), h = fc1's
 * output.  Row strides in elements, multiples of 8; 16-byte aligned bases (VIL_E_ALIGN otherwise).  K % 32 == 0,
 * N % 128 == 0; VIL_E_BACKEND outside that contract. */
int vil_gemm_dgelu_bf16(const void* dy, const void* w, const void* h, void* dh, int64_t T, int K, int N,
                        int64_t dy_row_stride, int64_t h_row_stride, int64_t dh_row_stride, void* stream);
/* The two tile kernels above without their activation epilogues -- op 0: out[T][N] = in[T][K] * w[N][K]^T (+ bias[N]) (the
 * forward of nn.Linear), op 1: out[T][N] = in[T][K] * w[K][N] (its input gradient) -- for the projections of the dense
 * stages (reference msvit.py:91-120 qkv / proj, :17-34 fc2; T = 6 k ... 25 k tokens), where the tuned library GEMM runs
 * at 13-31 % of the matrix peak.  bf16, fp32 accumulate, row strides in elements; K % 32 == 0, N % 128 == 0, 16-byte
 * aligned; VIL_E_BACKEND outside that contract (the caller then uses vil_gemm_bf16). */
int vil_gemm_tile_bf16(int op, const void* in, const void* w, const void* bias, void* out, int64_t T, int K, int N,
                       int64_t in_row_stride, int64_t out_row_stride, void* stream);
/* The forward counterpart: fc1 with nn.GELU() in its epilogue (reference msvit.py:29-31),
 *   h[t][n] = sum_k x[t][k] * w[n][k] + bias[n],   a[t][n] = gelu(h[t][n])  (exact erf form, of the rounded bf16 h)
 * one launch that writes both (the unfused pair writes h, reads it again and writes a).  w = fc1.weight (N rows of K),
 * bias may be NULL; h and a share out_row_stride.  Same contract as vil_gemm_dgelu_bf16. */
int vil_gemm_gelu_bf16(const void* x, const void* w, const void* bias, void* h, void* a, int64_t T, int K, int N,
                       int64_t x_row_stride, int64_t out_row_stride, void* stream);

/* ---- nn.Linear forward (op 0) / input gradient (op 1) of the projections with a huge token count and a small weight
 * matrix (stages 1-2 of ViL; reference msvit.py:17-34, 91-120, layers/longformer2d.py:47-62):
 *   op 0: out[t][n] = sum_k in[t][k] * w[n][k] + bias[n];   op 1: out[t][n] = sum_k in[t][k] * w[k][n]  (w as stored)
 * bf16, fp32 accumulate, the weight matrix held in registers by persistent workgroups, activations through an LDS-DMA
 * ring (csrc/vil_gemm_skinny.hip).  K in {96, 192} with N <= 768, or K in {288, 384, 576, 768} with N <= 256; N % 8 == 0;
 * VIL_E_BACKEND outside that contract.  Row strides in elements; bias may be NULL (and must be for op 1). */
int vil_gemm_skinny_bf16(int op, const void* in, const void* w, const void* bias, void* out, int64_t T, int K, int N,
                         int64_t in_row_stride, int64_t out_row_stride, void* stream);
/* The same forward GEMM with nn.GELU() (exact erf form, reference msvit.py:17-34: fc1 -> GELU) in its epilogue:
 * out = in . w^T + bias, act = gelu(out) evaluated on the rounded bf16 out -- what the unfused Linear -> GELU pair
 * computes -- both (T, N) with row stride out_row_stride.  K = 96 / 192, N <= 768 (VIL_E_BACKEND otherwise). */
int vil_gemm_skinny_gelu_bf16(const void* in, const void* w, const void* bias, void* out, void* act, int64_t T, int K, int N,
                              int64_t in_row_stride, int64_t out_row_stride, void* stream);

/* ---- dense `Attention` of the s0 stages as its own kernel family (reference
 * src/models/msvit.py:91-120): every one of the N = G + nx*ny tokens attends every token,
 *   logit[i][j] = scale * q_i.k_j + bias[h][i][j],
 *   bias = local_relative_position_bias_table[relative_position_index] (local i, local j; msvit.py:74-95),
 *          g2l[1][h][j] (local i, global j), g2l[0][h][i] (global i, local j), g2g[h][i][j] (both global; :97-111).
 * q / k / v / out / dout / dq / dk / dv point at TOKEN 0 of (B, N, H*M) views addressed with the descriptor's strides
 * (three views of one packed qkv tensor in the product); the descriptor's W, mode, exact, bias_side are ignored.
 * bias_table: float32 ((2nx-1)*(2ny-1), H) or NULL (rpe off); g2l (2,H,G), g2g (H,G,G) float32 or NULL.
 * lse: (B,H,N+1) float32 -- N log-sum-exps, then max_k |v_k|^2 of the (image, head) (the backward's fixed-point scale).
 * bf16 / fp16, M == 64, G <= 4.
 * The backward overwrites dq / dk / dv and dbias_table / dg2l / dg2g (bit-reproducible: fixed-point histogram,
 * fixed summation order); workspace >= vil_dense_attn_workspace_bytes(d, 1), 16-byte aligned. */
int vil_dense_attn_supported(const VilAttnDesc* d);
size_t vil_dense_attn_workspace_bytes(const VilAttnDesc* d, int pass);
int vil_dense_attn_fwd(const VilAttnDesc* d, const void* q, const void* k, const void* v,
                       const float* bias_table, const float* g2l, const float* g2g,
                       void* out, float* lse, void* stream);
/* Launch shape of vil_dense_attn_fwd: -1 (default) chooses per problem -- workgroups of up to 8 waves on a 64-row K/V ring,
 * or of up to 4 waves on a 32-row ring where that needs fewer rounds of workgroups on the device (24 x 24 tokens at
 * B H >= 171) --, 0 / 1 force the wide / narrow shape.  Results are bit-identical across shapes (a wave's unit and its
 * 32-key steps do not change).  Process-global; for tests and measurements. */
int vil_dense_attn_set_fwd_shape(int mode);
int vil_dense_attn_bwd(const VilAttnDesc* d, const void* q, const void* k, const void* v,
                       const void* out, const void* dout, const float* lse,
                       const float* bias_table, const float* g2l, const float* g2g,
                       void* dq, void* dk, void* dv, float* dbias_table, float* dg2l, float* dg2g,
                       void* workspace, void* stream);

/* ---- block glue: fused LayerNorm around the attention / MLP blocks
 * (`x + drop_path(attn(norm(x), nx, ny))`, reference src/models/msvit.py:313-316,336-340).
 * x: (rows, C) fp32 or bf16 with a row stride (elements); y is written in y_dtype (bf16 feeds the
 * following GEMM directly); gamma/beta/mean/rstd fp32; C % 8 == 0, C <= 1024. */
size_t vil_layernorm_workspace_bytes(int64_t rows, int C);
int vil_layernorm_fwd(const void* x, int x_dtype, const float* gamma, const float* beta,
                      void* y, int y_dtype, float* mean, float* rstd, int64_t rows, int C,
                      int64_t x_row_stride, int64_t y_row_stride, float eps, void* stream);
/* dx has x's dtype; dgamma/dbeta (C) fp32 are overwritten; workspace >= vil_layernorm_workspace_bytes */
int vil_layernorm_bwd(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* gamma,
                      const float* mean, const float* rstd, void* dx, int dx_dtype,
                      float* dgamma, float* dbeta, void* workspace, int64_t rows, int C,
                      int64_t dy_row_stride, int64_t x_row_stride, int64_t dx_row_stride, void* stream);

/* LayerNorm of the patch embedding written straight into / differentiated straight out of the stage's token tensor
 * (B, gap_rows + rows_per_sample, C) contiguous: row r of the normalised side lives at token row
 * r + (r / rows_per_sample + 1) * gap_rows, i.e. behind the gap_rows global tokens of its sample -- replaces the copy of
 * `torch.cat((cls_tokens, x), dim=1)` (reference src/models/msvit.py:204-206) and of its backward.  The global-token
 * rows are not touched.  rows % rows_per_sample == 0. */
int vil_layernorm_fwd_tokens(const void* x, int x_dtype, const float* gamma, const float* beta,
                             void* y_tokens, int y_dtype, float* mean, float* rstd, int64_t rows, int C,
                             int64_t x_row_stride, float eps, int64_t rows_per_sample, int64_t gap_rows, void* stream);
int vil_layernorm_bwd_tokens(const void* dy_tokens, int dy_dtype, const void* x, int x_dtype, const float* gamma,
                             const float* mean, const float* rstd, void* dx, int dx_dtype,
                             float* dgamma, float* dbeta, void* workspace, int64_t rows, int C,
                             int64_t x_row_stride, int64_t dx_row_stride, int64_t rows_per_sample, int64_t gap_rows,
                             void* stream);

/* ---- stage transition: `x[:, G:].transpose(-2,-1).reshape(B,-1,nx,ny)` + the strided Conv2d of
 * PatchEmbed (reference src/models/msvit.py:500-507, 166-203) as ONE row gather in front of a GEMM:
 *   patches[(b, i', j'), (py, px, c)] = x[b, G + (i' ph + py) ny + (j' pw + px), c] (+ rscale[b] * res[same]),
 * x fp32 (B, G + nx*ny, C) contiguous, res (the pending `drop_path(branch)` of the stage's last block, or NULL) fp32 /
 * bf16 of the same shape, rscale (B) fp32 or NULL, patches (B * nx/ph * ny/pw, ph*pw*C) fp32 / bf16.  C % 8 == 0.
 * _bwd: dx (B, G + nx*ny, C) fp32 = the scattered patch gradient (global-token rows 0) and, when gbranch != NULL,
 * gbranch = rscale[b] * dx in gb_dtype (the pending branch's gradient). */
int vil_patchify_fwd(const float* x, const void* res, int res_dtype, const float* rscale, void* patches, int out_dtype,
                     int B, int G, int nx, int ny, int C, int ph, int pw, void* stream);
int vil_patchify_bwd(const void* dpatches, int dp_dtype, const float* rscale, float* dx, void* gbranch, int gb_dtype,
                     int B, int G, int nx, int ny, int C, int ph, int pw, void* stream);

/* ---- the reference's OPERATOR-level surface (compatibility / parity; the hot path is vil_attn_fwd/_bwd, which never
 * builds the score tensor).  Chunked layouts of the reference: images (BH, M, mx, my, W^2), scores
 * (BH, mx, my, W^2, kv), kv = 9 W^2 (mode 0) | W^2 (mode -1) | 2 W^2 (mode 1..8: [own chunk | neighbour]); neighbours
 * are cyclic (torch.roll).  dtype VIL_DTYPE_F32, _F64, _BF16 or _F16 (16-bit I/O accumulates in fp32); contiguous tensors.
 *   vil_sc2d_qk     SlidingChunk2D.slidingchunk_qk     (src/models/layers/slidingchunk_2d.py:26-79)
 *   vil_sc2d_av     SlidingChunk2D.slidingchunk_av     (:82-130)
 *   vil_sc2d_agrad  SlidingChunk2D.slidingchunk_agrad  (:132-200)
 *   vil_sc2d_mask   mask_invalid_locations             (:321-357): in-place -inf; *count (device, caller-zeroed, may be
 *                   NULL) receives the reference's num_invalid; VIL_E_EXACT where the reference raises ValueError. */
int vil_sc2d_qk(const void* q_img, const void* k_img, void* attn, int BH, int M, int mx, int my, int W, int mode,
                int dtype, void* stream);
int vil_sc2d_av(const void* attn, const void* v_img, void* out_img, int BH, int M, int mx, int my, int W, int mode,
                int dtype, void* stream);
int vil_sc2d_agrad(const void* attn, const void* grad_img, void* out_img, int BH, int M, int mx, int my, int W, int mode,
                   int dtype, void* stream);
int vil_sc2d_mask(void* attn, int BH, int mx, int my, int padx, int pady, int W, int exact, int mode, int dtype,
                  unsigned long long* count, void* stream);

/* ---- optional profiling sink (a measurement aid for bench.py; the ONLY state the
 * library keeps: process-global, not thread-safe).  Between _begin and _end every
 * kernel the library launches is bracketed by hipEventRecord on its launch stream.
 * _end synchronises on the recorded events and returns, per launch in order, the
 * kernel id (name via vil_attn_kernel_name), its duration in ms and the algorithmic
 * HBM bytes / flops of that launch.  Returns the count. */
int vil_attn_profile_begin(int capacity);
int vil_attn_profile_end(int capacity, int* kernel_id, float* ms, double* bytes, double* flops);
/* same, plus the problem tag of every launch: 8 ints per record -- attention kernels {B,H,M,nx,ny,W,G,mode (9 = device
 * word)}, weight-gradient kernels {T,CO,CI,0...} -- so one roofline per SHAPE can be reported */
int vil_attn_profile_end2(int capacity, int* kernel_id, float* ms, double* bytes, double* flops, int* tags);
const char* vil_attn_kernel_name(int kernel_id);

/* ---- host-side geometry helpers (pure CPU, used by the tests to pin the
 * kernels' index/mask logic against the golden masks without a GPU) -------- */

/* Fills mask[(m*my+n)*W2*kv + l*kv + s] (1 = key slot s is NOT attended by
 * query l of chunk (m,n)) for the given grid; kv = 9W^2 / W^2 / 2W^2 by mode,
 * slots in the reference's order.  Mirrors mask_invalid_locations
 * (slidingchunk_2d.py:321-357).  Returns kv, or a VIL_E_* code. */
int vil_geom_mask(int nx, int ny, int W, int exact, int mode, uint8_t* mask);

/* rel[l*kv + s] = index into the ((4W-1)^2) bias table used by query l for
 * key slot s (longformer2d.py:67-100 with the mode's column subset :164-173). */
int vil_geom_bias_index(int W, int mode, int32_t* rel);

/* ---- bias gradient of the projections around the hot path: out[c] = sum_r x[r*row_stride + c] over a
 * (rows x C) bf16 matrix (autograd of nn.Linear's bias; reference msvit.py:91-120, 236-255).  C % 8 == 0,
 * 16-byte aligned rows; out is C floats (out_bf16 = 0) or C bf16 (1); workspace of
 * vil_colsum_workspace_bytes(C) bytes. */
size_t vil_colsum_workspace_bytes(int C);
int vil_colsum_bf16(const void* x, int64_t rows, int C, int64_t row_stride, void* out, int out_bf16,
                    void* workspace, void* stream);
/* same for an fp32 matrix (gradients that live on the fp32 residual stream: cls-token rows) */
int vil_colsum_f32(const void* x, int64_t rows, int C, int64_t row_stride, void* out, int out_bf16,
                   void* workspace, void* stream);

/* ---- fused weight + bias gradient of a projection y = x W^T + b (autograd of nn.Linear; reference
 * msvit.py:91-120, 236-255, longformer2d.py:47-62): dW[co][ci] = sum_t dy[t][co] x[t][ci] (row-major
 * (CO, CI), like nn.Linear.weight) and, when db != NULL, db[co] = sum_t dy[t][co].  dy is (T, CO) and x
 * (T, CI) bf16 with the given row strides (elements); CO, CI and the strides multiples of 8, 16-byte
 * aligned bases.  Outputs bf16 (out_bf16 = 1) or fp32.  Workspace: vil_linear_wgrad_workspace_bytes (covers every
 * plan the tuner may select).  Two kernel generations (csrc/vil_wgrad.hip): 128 x 128 tiles for any channel count,
 * and -- channel counts that are multiples of 96 -- 96/192-wide tiles fed by an LDS-DMA ring; which one runs, with
 * how many token slices, is the plan vil_linear_wgrad_tune measured for (T, CO, CI) (same arguments; times every
 * candidate on the caller's operands, SYNCHRONISES, must be called outside stream capture, idempotent; the plan
 * cache is process-global, mutex-guarded) or a cost model's choice when the problem was never tuned.  Partial sums
 * are added in a fixed order: results are bit-reproducible for a given plan. */
size_t vil_linear_wgrad_workspace_bytes(int64_t T, int CO, int CI);
int vil_linear_wgrad(const void* dy, const void* x, int64_t T, int CO, int CI, int64_t dy_stride, int64_t x_stride,
                     void* dw, void* db, int out_bf16, void* workspace, void* stream);
int vil_linear_wgrad_tune(const void* dy, const void* x, int64_t T, int CO, int CI, int64_t dy_stride, int64_t x_stride,
                          void* dw, void* db, int out_bf16, void* workspace, void* stream);
/* Writes the plan cache entry of a problem directly -- gen 1: the 128 x 128 kernel; gen 2: tile 32 mi x 32 nj (mi, nj in
 * {3, 6}, dividing CO / CI), m token slices per XCD (clamped to what the workspace bound allows); gen 0: erase, back to
 * the cost model.  For restoring an earlier selection, measurements and tests.  VIL_E_SHAPE for a plan that does not
 * fit the problem. */
int vil_linear_wgrad_set_plan(int64_t T, int CO, int CI, int gen, int mi, int nj, int m);
/* Reads the plan vil_linear_wgrad would run: plan = {gen, mi, nj, m, tuned} (tuned = 1: measured by
 * vil_linear_wgrad_tune or written by _set_plan; 0: the cost model's choice).  With _set_plan this lets a host keep the
 * summation order of dW fixed across runs and equal across ranks (linear.export_plans / import_plans; under DDP rank 0's
 * measured plans are broadcast). */
int vil_linear_wgrad_get_plan(int64_t T, int CO, int CI, int* plan);

/* ---- fused residual add + LayerNorm on the fp32 residual stream (block glue of msvit.py:313-316,336-340:
 * `x = x + drop_path(branch)` of one block fused with `norm(x)` of the next).  Contiguous (rows, C) tensors.
 *   forward : x_out = x + rscale[row / rows_per_sample] * res  (rscale NULL: 1);  y = LN(x_out)
 *   backward: dx = gres (NULL: 0) + LNbwd(dy);  gbranch = rscale[...] * dx in the branch's dtype
 * (dx is the gradient of x AND of x_out's producer; gbranch the gradient of res; gbranch NULL: no branch was added --
 * the first block of a stage, where x feeds the norm and the residual stream: dx = gres + LNbwd(dy) in one pass). */
int vil_resln_fwd(const float* x, const void* res, int res_dtype, const float* rscale, int64_t rows_per_sample,
                  const float* gamma, const float* beta, float* x_out, void* y, int y_dtype,
                  float* mean, float* rstd, int64_t rows, int C, float eps, void* stream);
int vil_resln_bwd(const void* dy, int dy_dtype, const float* gres, const float* x, const float* gamma,
                  const float* mean, const float* rstd, const float* rscale, int64_t rows_per_sample,
                  float* dx, void* gbranch, int gb_dtype, float* dgamma, float* dbeta, void* workspace,
                  int64_t rows, int C, void* stream);

/* ---- the plain library GEMMs of the projections (hipBLASLt; reference call sites: every nn.Linear of msvit.py /
 * longformer2d.py).  vil_gemm_bf16 launches asynchronously and never synchronises; the algorithm it runs is the one
 * vil_gemm_tune selected for that problem (same arguments; times the heuristic's candidates on the caller's operands,
 * SYNCHRONISES, must be called outside stream capture, idempotent) or the heuristic's first choice when the problem
 * was never tuned.  The plan cache (problem -> descriptors + algorithm) is process-global, mutex-guarded state.
 *   op 0: out[T][N] = in[T][K] * w[N][K]^T (+ bias[N])      (forward;  w = nn.Linear.weight)
 *   op 1: out[T][N] = in[T][K] * w[K][N]                    (input gradient: in = dY, w = nn.Linear.weight)
 *   op 2: out[N][K] = w[T][N]^T * in[T][K], bias[N] = colsum(w)  (weight / bias gradient: in = x, w = dY,
 *         out_row_stride = row stride of dY; bias (output) may be NULL)
 * bf16 operands, fp32 accumulate; row strides in elements (multiples of 8); workspace of vil_gemm_workspace_bytes(). */
size_t vil_gemm_workspace_bytes(void);
int vil_gemm_bf16(int op, const void* in, const void* w, const void* bias, void* out, int64_t T, int K, int N,
                  int64_t in_row_stride, int64_t out_row_stride, void* workspace, size_t workspace_bytes, void* stream);
int vil_gemm_tune(int op, const void* in, const void* w, const void* bias, void* out, int64_t T, int K, int N,
                  int64_t in_row_stride, int64_t out_row_stride, void* workspace, size_t workspace_bytes, void* stream);

/* ---- the optimizer step of the training loop as ONE multi-tensor launch on fp32 master weights.
 * Replaces torch.optim's per-recipe optimizers of the reference with its own update rules:
 *   vil_optim_adamw_step   src/optim/optimization.py:111-193 (AdamW: denom = sqrt(v) + eps with eps OUTSIDE the bias
 *                          correction, step_size = lr sqrt(1-b2^t)/(1-b1^t), decoupled decay p -= lr wd p AFTER the
 *                          Adam update; the optimizer of config/msvit.yaml:32-47)
 *   vil_optim_qhm_step     src/optim/qhm.py:55-130 (QHM: g += wd p; h = beta h + (1-beta) g; d = (1-nu) g + nu h;
 *                          p -= lr d; the optimizer of config/msvit_384finetune.yaml:28-35)
 * Each tensor is described by a VilOptimTensor: fp32 `param` (updated in place), its gradient in the dtype autograd
 * produced it in, fp32 state tensors, and optionally the 16-bit working copy (`low`) of the parameter, refreshed in
 * the same pass.  A PLAN (header + descriptors + a table of 4096-element blocks) is built on the host
 * (vil_optim_plan_build into a caller buffer of vil_optim_plan_bytes bytes), uploaded by the caller to 16-byte
 * aligned device memory and reused by every step while the tensor addresses are unchanged (hipGraph replay).
 * `step_words` = two device int32: [0] the number of completed steps t (the launch applies step t + 1 and its last
 * workgroup increments the word), [1] an arrival ticket that must be 0 before the first launch.  The learning rate
 * is per tensor (= per param group): a host value, or a device float that a per-iteration schedule updates in place
 * so that a captured step follows it.  Asynchronous on `stream`. */
typedef struct VilOptimTensor {
  void* param;            /* fp32, n elements                                             */
  const void* grad;       /* grad_dtype, n elements                                       */
  void* state1;           /* fp32: AdamW exp_avg / QHM momentum buffer                    */
  void* state2;           /* fp32: AdamW exp_avg_sq; unused by QHM (may be NULL)          */
  void* low;              /* low_dtype working copy written with the updated value, or NULL */
  int64_t n;
  int32_t grad_dtype;     /* VIL_DTYPE_F32 / _BF16 / _F16                                 */
  int32_t low_dtype;      /* VIL_DTYPE_BF16 / _F16                                        */
  float weight_decay;     /* the param group's weight_decay                               */
  float lr;               /* the param group's learning rate, used when lr_dev is NULL     */
  const float* lr_dev;    /* device float holding the learning rate (a captured step follows a per-iteration
                             schedule that updates it in place), or NULL                   */
} VilOptimTensor;
size_t vil_optim_plan_bytes(const VilOptimTensor* tensors, int ntensors);
int vil_optim_plan_build(const VilOptimTensor* tensors, int ntensors, void* host_plan, size_t bytes, int* nblocks);
int vil_optim_adamw_step(const void* plan_dev, int nblocks, float beta1, float beta2, float eps, int correct_bias,
                         int32_t* step_words, void* stream);
int vil_optim_qhm_step(const void* plan_dev, int nblocks, float momentum, float nu, int32_t* step_words, void* stream);
/* The same steps under torch.amp.GradScaler (fp16 training: reference src/engine.py:84-100, src/run_experiment.py:206;
 * what scaler.step(optimizer) does with _amp_foreach_non_finite_check_and_unscale_ + a host-side `if found_inf`):
 * `inv_scale` and `found_inf` are DEVICE floats (either may be NULL).  Gradients are multiplied by *inv_scale as they are
 * loaded (16-bit gradients are unscaled in fp32, never in place); *found_inf != 0 skips the whole step -- parameters,
 * state, working copies and the step count keep their values -- without a host synchronisation, so a captured step
 * (hipGraph) carries the scaler. */
int vil_optim_adamw_step_amp(const void* plan_dev, int nblocks, float beta1, float beta2, float eps, int correct_bias,
                             int32_t* step_words, const float* inv_scale, const float* found_inf, void* stream);
int vil_optim_qhm_step_amp(const void* plan_dev, int nblocks, float momentum, float nu, int32_t* step_words,
                           const float* inv_scale, const float* found_inf, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VIL_ATTN_H */
