// vil_patchify.hip -- the stage transition of the multi-scale model around the hot path:
//     x = x[:, G:].transpose(-2, -1).reshape(B, C, nx, ny);  PatchEmbed: Conv2d(C, C', kernel = stride = (ph, pw))
// (reference src/models/msvit.py:500-507, 166-203).  The strided convolution of non-overlapping patches is a GEMM over
// (py, px, c) patch vectors, so the transition is ONE row-gather: drop the G global tokens, regroup the tokens of every
// ph x pw patch into one vector, cast to the GEMM's dtype -- and, fused in, the residual add of the stage's last block
// that was still pending (x + drop_path(branch)).  The backward scatters the patch-vector gradient home, zero-fills the
// global-token rows and emits the pending branch's gradient (stochastic-depth scale and cast included).  Replaces, per
// transition, add + slice + permute/cast copy (forward) and permute copy + fill + slice copy + mul + cast (backward).
//
//   patches[(b, i', j'), (py, px, c)] = x[b, G + (i' ph + py) ny + (j' pw + px), c] + rscale[b] * res[same]
#include "vil_internal.h"
#include <cstring>

struct PatchParams {
  const float* x; const void* res; const float* rscale;
  void* patches;                 // forward: output; backward: the gradient w.r.t. the patch vectors (input)
  float* dx; void* gbranch;
  int res_bf16, p_bf16, gb_bf16;
  int B, G, nx, ny, C8, ph, pw;  // C8 = C / 8
  int nyp;                       // ny / pw
  unsigned m_C8, m_ny, m_ph, m_pw;    // magic reciprocals (0: divisor 1)
  int64_t per_sample;            // (G + nx*ny) * C8 work items per sample
  int64_t rows_ps;               // G + nx*ny token rows per sample
};

__device__ __forceinline__ unsigned pdiv(unsigned n, unsigned magic) { return magic ? __umulhi(n, magic) : n; }
static unsigned pmagic(unsigned d) { return d <= 1 ? 0u : (unsigned)(0x100000000ull / d) + 1u; }

__device__ __forceinline__ void ld8f(const float* p, float (&v)[8]) {
  const float4 a = ((const float4*)p)[0], b = ((const float4*)p)[1];
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void st8f(float* p, const float (&v)[8]) {
  ((float4*)p)[0] = make_float4(v[0], v[1], v[2], v[3]);
  ((float4*)p)[1] = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void ld8b(const vil_bf16* p, float (&v)[8]) {
  const uint4 r = *(const uint4*)p;
  const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
}
__device__ __forceinline__ void st8b(vil_bf16* p, const float (&v)[8]) {
  unsigned w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) w[i] = (unsigned)vil_f2bf(v[2 * i]) | ((unsigned)vil_f2bf(v[2 * i + 1]) << 16);
  *(uint4*)p = make_uint4(w[0], w[1], w[2], w[3]);
}

// element offset (in units of 8 channels) of local token `tok`, channel group c8, inside sample b's patch matrix
__device__ __forceinline__ int64_t patch_slot(const PatchParams& p, int tok, int c8) {
  const int r = pdiv(tok, p.m_ny), c = tok - r * p.ny;
  const int ip = pdiv(r, p.m_ph), py = r - ip * p.ph;
  const int jp = pdiv(c, p.m_pw), px = c - jp * p.pw;
  return ((int64_t)(ip * p.nyp + jp) * (p.ph * p.pw) + (py * p.pw + px)) * p.C8 + c8;
}

// grid (ceil(per_sample / 256), B): one thread per 8 channels of one token of one sample
__global__ __launch_bounds__(256) void k_patchify_fwd(PatchParams p) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (i >= p.per_sample) return;
  const int row = pdiv((unsigned)i, p.m_C8), c8 = (int)i - row * p.C8;
  if (row < p.G) return;                                   // the stage's global tokens are dropped
  const int64_t src = ((int64_t)b * p.rows_ps + row) * p.C8 + c8;
  float v[8];
  ld8f(p.x + src * 8, v);
  if (p.res) {
    float r[8];
    if (p.res_bf16) ld8b((const vil_bf16*)p.res + src * 8, r); else ld8f((const float*)p.res + src * 8, r);
    const float sc = p.rscale ? p.rscale[b] : 1.0f;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaf(sc, r[e], v[e]);
  }
  const int64_t dst = (int64_t)b * ((int64_t)p.nx * p.ny * p.C8) + patch_slot(p, row - p.G, c8);
  if (p.p_bf16) st8b((vil_bf16*)p.patches + dst * 8, v); else st8f((float*)p.patches + dst * 8, v);
}

__global__ __launch_bounds__(256) void k_patchify_bwd(PatchParams p) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (i >= p.per_sample) return;
  const int row = pdiv((unsigned)i, p.m_C8), c8 = (int)i - row * p.C8;
  const int64_t dst = ((int64_t)b * p.rows_ps + row) * p.C8 + c8;
  float v[8];
  if (row < p.G) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
  } else {
    const int64_t src = (int64_t)b * ((int64_t)p.nx * p.ny * p.C8) + patch_slot(p, row - p.G, c8);
    if (p.p_bf16) ld8b((const vil_bf16*)p.patches + src * 8, v); else ld8f((const float*)p.patches + src * 8, v);
  }
  st8f(p.dx + dst * 8, v);
  if (p.gbranch) {
    const float sc = p.rscale ? p.rscale[b] : 1.0f;
    float g[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] = v[e] * sc;
    if (p.gb_bf16) st8b((vil_bf16*)p.gbranch + dst * 8, g); else st8f((float*)p.gbranch + dst * 8, g);
  }
}

static int patch_fill(PatchParams& p, int B, int G, int nx, int ny, int C, int ph, int pw) {
  if (B <= 0 || G < 0 || nx <= 0 || ny <= 0 || C <= 0 || ph <= 0 || pw <= 0) return VIL_E_SHAPE;
  if (nx % ph || ny % pw) return VIL_E_SHAPE;
  if (C % 8) return VIL_E_ALIGN;
  memset(&p, 0, sizeof(p));
  p.B = B; p.G = G; p.nx = nx; p.ny = ny; p.C8 = C / 8; p.ph = ph; p.pw = pw; p.nyp = ny / pw;
  p.rows_ps = (int64_t)G + (int64_t)nx * ny;
  p.per_sample = p.rows_ps * p.C8;
  // the index arithmetic is 32-bit per sample and the divisions are magic multiplies, exact for n * d < 2^32
  if (p.per_sample >= (1ll << 31) || (uint64_t)p.per_sample * (uint64_t)p.C8 >= (1ull << 32) ||
      (uint64_t)nx * ny * (uint64_t)ny >= (1ull << 32))
    return VIL_E_SHAPE;
  p.m_C8 = pmagic((unsigned)p.C8); p.m_ny = pmagic((unsigned)ny); p.m_ph = pmagic((unsigned)ph); p.m_pw = pmagic((unsigned)pw);
  return VIL_OK;
}

extern "C" int vil_patchify_fwd(const float* x, const void* res, int res_dtype, const float* rscale, void* patches,
                                int out_dtype, int B, int G, int nx, int ny, int C, int ph, int pw, void* stream) {
  if (!x || !patches) return VIL_E_NULL;
  if ((res && res_dtype != VIL_DTYPE_F32 && res_dtype != VIL_DTYPE_BF16) ||
      (out_dtype != VIL_DTYPE_F32 && out_dtype != VIL_DTYPE_BF16)) return VIL_E_DTYPE;
  if (((uintptr_t)x | (uintptr_t)res | (uintptr_t)patches) & 15) return VIL_E_ALIGN;
  PatchParams p; int e = patch_fill(p, B, G, nx, ny, C, ph, pw);
  if (e) return e;
  p.x = x; p.res = res; p.res_bf16 = res_dtype == VIL_DTYPE_BF16; p.rscale = res ? rscale : nullptr;
  p.patches = patches; p.p_bf16 = out_dtype == VIL_DTYPE_BF16;
  k_patchify_fwd<<<dim3((unsigned)((p.per_sample + 255) / 256), (unsigned)B), dim3(256), 0, (hipStream_t)stream>>>(p);
  return (int)hipGetLastError();
}

extern "C" int vil_patchify_bwd(const void* dpatches, int dp_dtype, const float* rscale, float* dx, void* gbranch,
                                int gb_dtype, int B, int G, int nx, int ny, int C, int ph, int pw, void* stream) {
  if (!dpatches || !dx) return VIL_E_NULL;
  if ((dp_dtype != VIL_DTYPE_F32 && dp_dtype != VIL_DTYPE_BF16) ||
      (gbranch && gb_dtype != VIL_DTYPE_F32 && gb_dtype != VIL_DTYPE_BF16)) return VIL_E_DTYPE;
  if (((uintptr_t)dpatches | (uintptr_t)dx | (uintptr_t)gbranch) & 15) return VIL_E_ALIGN;
  PatchParams p; int e = patch_fill(p, B, G, nx, ny, C, ph, pw);
  if (e) return e;
  p.patches = (void*)dpatches; p.p_bf16 = dp_dtype == VIL_DTYPE_BF16; p.dx = dx;
  p.gbranch = gbranch; p.gb_bf16 = gb_dtype == VIL_DTYPE_BF16; p.rscale = gbranch ? rscale : nullptr;
  k_patchify_bwd<<<dim3((unsigned)((p.per_sample + 255) / 256), (unsigned)B), dim3(256), 0, (hipStream_t)stream>>>(p);
  return (int)hipGetLastError();
}
