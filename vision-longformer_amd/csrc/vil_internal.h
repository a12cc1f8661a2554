// vil_internal.h -- launch parameters shared by the kernel families of libvilattn.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vil_attn.h"
#include "vil_geom.h"

struct VilParams {
  VilGeom g;
  int B, H, M, G;
  int only_glo, has_bias, has_g2l;
  int bias_S, bias_off;      // side of the caller's bias table and its centre offset (S-1)/2
  int parts;                 // backward: workgroups per (b,h) in the dQ pass
  int part_stride;           // floats per partial record
  float scale;
  int64_t q_sb, q_st, q_sh, k_sb, k_st, k_sh, v_sb, v_st, v_sh, o_sb, o_st, o_sh;
  int64_t do_sb, do_st, do_sh, dq_sb, dq_st, dq_sh, dk_sb, dk_st, dk_sh, dv_sb, dv_st, dv_sh;
  const void* q; const void* k; const void* v;
  const void* out; const void* dout;
  const float* table; const float* g2l;
  float* lse;                // fwd: written; bwd: read
  void* o;                   // fwd output
  void* dq; void* dk; void* dv;
  float* dtable; float* dg2l;
  float* delta;              // workspace: rowsum(dO*O), (B*H*Nloc)
  float* partials;           // workspace: per-workgroup partial reductions
  // vil_attn_bwd_full: the backward of the G global-token QUERY rows rides in the dK/dV pass
  int glo_rows;
  const void* q_g; const void* do_g; const void* o_g;   // token 0 of the all-token q / dout / out tensors
  void* dq_g;                                            // token 0 of the all-token dq tensor
  const float* lse_g;        // (B,H,G)
  const float* g2l0;         // (H,G) bias global query -> local keys, or null
  const float* g2g;          // (H,G,G) or null
  float* dg2l0; float* dg2g;
  const int* mode_dev;       // device-side random-shift neighbour (1..8) or null
  float* xstat;              // workspace (B*H, G+1, 2): {lse, rowsum(dO * O)} of the G global-query rows, then {+big, 0} for
                             // padding slots -- what the dK/dV pass's streamed slots with a negative token read (k_mfma_delta)
};

static inline void vil_fill_params(VilParams& p, const VilAttnDesc* d) {
  vil_geom_init(p.g, d->nx, d->ny, d->W, d->exact, d->mode);
  p.B = d->B; p.H = d->H; p.M = d->M; p.G = d->G; p.only_glo = d->only_glo;
  p.scale = d->scale;
  p.mode_dev = d->mode_dev;
  p.bias_S = d->bias_side > 0 ? d->bias_side : 4 * d->W - 1;
  p.bias_off = (p.bias_S - 1) / 2;
  p.q_sb = d->q_sb; p.q_st = d->q_st; p.q_sh = d->q_sh;
  p.k_sb = d->k_sb; p.k_st = d->k_st; p.k_sh = d->k_sh;
  p.v_sb = d->v_sb; p.v_st = d->v_st; p.v_sh = d->v_sh;
  p.o_sb = d->o_sb; p.o_st = d->o_st; p.o_sh = d->o_sh;
  p.do_sb = d->do_sb; p.do_st = d->do_st; p.do_sh = d->do_sh;
  p.dq_sb = d->dq_sb; p.dq_st = d->dq_st; p.dq_sh = d->dq_sh;
  p.dk_sb = d->dk_sb; p.dk_st = d->dk_st; p.dk_sh = d->dk_sh;
  p.dv_sb = d->dv_sb; p.dv_st = d->dv_st; p.dv_sh = d->dv_sh;
}

// ---- bf16 <-> f32 (raw bits; round-to-nearest-even on store) ----
typedef uint16_t vil_bf16;
__host__ __device__ __forceinline__ float vil_bf2f(vil_bf16 h) {
  union { uint32_t u; float f; } c; c.u = (uint32_t)h << 16; return c.f;
}
__host__ __device__ __forceinline__ vil_bf16 vil_f2bf(float f) {
  union { uint32_t u; float f; } c; c.f = f;
  if ((c.u & 0x7fffffffu) > 0x7f800000u) return (vil_bf16)((c.u >> 16) | 0x40);  // quiet NaN
  c.u += 0x7fffu + ((c.u >> 16) & 1u);
  return (vil_bf16)(c.u >> 16);
}

// ---- optional profiling sink (vil_attn_profile_begin/_end): brackets every kernel
// launch with hipEvents on the launch stream and records its algorithmic bytes/flops
enum { VIL_K_TABLE = 0, VIL_K_MFMA_FWD, VIL_K_SCALAR_FWD, VIL_K_DELTA, VIL_K_SCALAR_DQ, VIL_K_SCALAR_DKDV,
       VIL_K_REDUCE_GLO, VIL_K_REDUCE_BIAS, VIL_K_MFMA_DQ, VIL_K_MFMA_DKDV, VIL_K_GLO_FWD, VIL_K_GLO_BWD,
       VIL_K_WGRAD, VIL_K_WGRAD_REDUCE, VIL_K_DENSE_FWD, VIL_K_DENSE_DQ, VIL_K_DENSE_DKDV, VIL_K_DENSE_REDUCE, VIL_K_COUNT };
void vil_prof_begin(int kid, hipStream_t s, double bytes, double flops);
void vil_prof_end(hipStream_t s);
// problem tag attached to the records that follow (attention: B,H,M,nx,ny,W,G,mode; wgrad: T,CO,CI): lets bench.py
// report one roofline per SHAPE instead of one average over different problems
void vil_prof_tag(int a0, int a1, int a2, int a3, int a4, int a5, int a6, int a7);
static inline void vil_prof_tag_desc(const VilAttnDesc* d) {
  vil_prof_tag(d->B, d->H, d->M, d->nx, d->ny, d->W, d->G, d->mode_dev ? 9 : d->mode);
}

// Raises a kernel's dynamic-LDS limit above 64 KB once per (device, kernel, size) instead of on every launch
// (hipFuncSetAttribute is a driver call; the remembered maxima are an idempotent process-global cache).
int vil_ensure_dyn_lds(const void* kernel, size_t bytes);
int vil_cu_count();                                  // CUs of the current device (cached per device)

// algorithmic (minimum) HBM bytes / flops of one launch over the whole batch; SURVEY.md 8(d)
struct VilWork {
  double nloc, n, c, e, k, h, b, tbl;
  explicit VilWork(const VilAttnDesc* d) {
    VilGeom g; vil_geom_init(g, d->nx, d->ny, d->W, d->exact, d->mode);
    nloc = (double)d->nx * d->ny; n = nloc + d->G; c = (double)d->H * d->M;
    e = d->dtype == VIL_DTYPE_F32 ? 4 : 2; h = d->H; b = d->B;
    k = d->only_glo ? d->G : (double)g.nact * g.W2 + d->G; tbl = d->bias_side > 0 ? (double)d->bias_side * d->bias_side : (double)g.tbl * g.tbl;
  }
  double fwd_bytes() const { return b * ((2 * nloc + 2 * n) * c * e + 4 * h * nloc + 4 * h * tbl); }
  double fwd_flops() const { return b * 4 * nloc * k * c; }
  double delta_bytes() const { return b * (2 * nloc * c * e + 4 * h * nloc); }
  double dq_bytes() const { return b * ((3 * nloc + 2 * n) * c * e + 8 * h * nloc); }
  double dq_flops() const { return b * 6 * nloc * k * c; }
  double dkdv_bytes() const { return b * ((2 * nloc + 4 * n) * c * e + 8 * h * nloc); }
  double dkdv_flops() const { return b * 8 * nloc * k * c; }
};

// kernel families (each returns 0 or a hipError_t / VIL_E_*)
int vil_scalar_supported(const VilAttnDesc* d);
size_t vil_scalar_workspace(const VilAttnDesc* d, int pass);
int vil_scalar_fwd(const VilAttnDesc* d, VilParams& p, hipStream_t s);
int vil_scalar_bwd(const VilAttnDesc* d, VilParams& p, hipStream_t s);

int vil_mfma_supported(const VilAttnDesc* d, int pass);
size_t vil_mfma_workspace(const VilAttnDesc* d, int pass);
int vil_mfma_fwd(const VilAttnDesc* d, VilParams& p, hipStream_t s);
int vil_mfma_bwd(const VilAttnDesc* d, VilParams& p, hipStream_t s);
