// vil_attn_api.hip -- the C ABI of libvilattn.so (see include/vil_attn.h):
// argument validation, kernel-family dispatch, and the host-side geometry
// helpers the CPU tests use to pin the kernels' mask / bias-index logic.
#include "vil_internal.h"
#include "vil_mfma_common.h"
#include <string.h>
#include <vector>

extern "C" int vil_attn_abi_version(void) { return VIL_ATTN_ABI_VERSION; }

extern "C" const char* vil_attn_strerror(int code) {
  switch (code) {
    case VIL_OK: return "ok";
    case VIL_E_NULL: return "required pointer is NULL";
    case VIL_E_SHAPE: return "invalid or inconsistent sizes";
    case VIL_E_HEAD_DIM: return "unsupported head_dim";
    case VIL_E_WINDOW: return "unsupported window size";
    case VIL_E_MODE: return "mode must be in [-1, 8]";
    case VIL_E_EXACT: return "longsc exact should be in [0,1,-1] (exact=1 requires mode 0)";
    case VIL_E_DTYPE: return "unsupported dtype";
    case VIL_E_ALIGN: return "pointer or stride alignment not supported";
    case VIL_E_WORKSPACE: return "workspace is NULL but workspace_bytes > 0";
    case VIL_E_BACKEND: return "requested backend cannot run this descriptor";
    default: break;
  }
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "unknown error";
}

static int check_common(const VilAttnDesc* d) {
  if (!d) return VIL_E_NULL;
  if (d->B <= 0 || d->H <= 0 || d->M <= 0 || d->nx <= 0 || d->ny <= 0 || d->W <= 0 || d->G < 0)
    return VIL_E_SHAPE;
  if (d->only_glo && d->G < 1) return VIL_E_SHAPE;
  if (d->mode < -1 || d->mode > 8) return VIL_E_MODE;
  if (d->exact < -1 || d->exact > 1) return VIL_E_EXACT;
  if (d->exact == 1 && d->mode != 0 && !d->only_glo) return VIL_E_EXACT;
  if (d->dtype != VIL_DTYPE_F32 && d->dtype != VIL_DTYPE_BF16 && d->dtype != VIL_DTYPE_F16) return VIL_E_DTYPE;
  if (d->mode_dev && (d->mode < 1 || d->mode > 8)) return VIL_E_MODE;
  if (d->bias_side < 0 || (d->bias_side > 0 && (d->bias_side % 2 == 0 || d->bias_side > 4 * d->W - 1))) return VIL_E_SHAPE;
  return VIL_OK;
}

static int pick_backend(const VilAttnDesc* d, int pass) {
  const int want = d->backend;
  if (want == VIL_BACKEND_SCALAR) return vil_scalar_supported(d) == VIL_OK ? VIL_BACKEND_SCALAR : 0;
  if (want == VIL_BACKEND_MFMA || want == VIL_BACKEND_MFMA_WAVE || want == VIL_BACKEND_MFMA_CW) return vil_mfma_supported(d, pass) == VIL_OK ? VIL_BACKEND_MFMA : 0;
  if (vil_mfma_supported(d, pass) == VIL_OK) return VIL_BACKEND_MFMA;
  if (vil_scalar_supported(d) == VIL_OK) return VIL_BACKEND_SCALAR;
  return 0;
}

extern "C" int vil_attn_check(const VilAttnDesc* d) {
  int e = check_common(d);
  if (e) return e;
  if (d->backend == VIL_BACKEND_SCALAR) return vil_scalar_supported(d);
  if (d->backend == VIL_BACKEND_MFMA || d->backend == VIL_BACKEND_MFMA_WAVE || d->backend == VIL_BACKEND_MFMA_CW) {
    e = vil_mfma_supported(d, 0);
    return e ? e : vil_mfma_supported(d, 1);
  }
  if (d->backend != VIL_BACKEND_AUTO) return VIL_E_BACKEND;
  return vil_scalar_supported(d) == VIL_OK ? VIL_OK : vil_mfma_supported(d, 0);
}

extern "C" size_t vil_attn_workspace_bytes(const VilAttnDesc* d, int pass) {
  if (check_common(d)) return 0;
  // the union of both families, so the caller may switch backend per call
  size_t a = vil_scalar_supported(d) == VIL_OK ? vil_scalar_workspace(d, pass) : 0;
  size_t b = vil_mfma_supported(d, pass) == VIL_OK ? vil_mfma_workspace(d, pass) : 0;
  return a > b ? a : b;
}

extern "C" int vil_attn_fwd(const VilAttnDesc* d, const void* q, const void* k, const void* v,
                            const float* bias_table, const float* g2l,
                            void* out, float* lse, void* workspace, void* stream) {
  int e = check_common(d);
  if (e) return e;
  vil_prof_tag_desc(d);
  if (!q || !k || !v || !out || !lse) return VIL_E_NULL;
  const int be = pick_backend(d, 0);
  if (!be) return d->backend == VIL_BACKEND_AUTO ? vil_scalar_supported(d) : VIL_E_BACKEND;
  if (!workspace && vil_attn_workspace_bytes(d, 0) > 0) return VIL_E_WORKSPACE;
  VilParams p; memset(&p, 0, sizeof(p));
  vil_fill_params(p, d);
  p.q = q; p.k = k; p.v = v; p.o = out; p.lse = lse;
  p.table = bias_table; p.g2l = g2l;
  p.has_bias = bias_table != nullptr; p.has_g2l = (g2l != nullptr) && d->G > 0;
  p.delta = (float*)workspace;
  return be == VIL_BACKEND_MFMA ? vil_mfma_fwd(d, p, (hipStream_t)stream)
                                : vil_scalar_fwd(d, p, (hipStream_t)stream);
}

// Whole-layer forward: local rows AND the global token's row in the forward pass's own launch (round 5).  Returns
// VIL_E_BACKEND where the row cannot ride (G != 1, only_glo, fp32 / scalar family, W = 8 at head_dim 32): the caller then
// runs vil_attn_fwd + vil_glo_attn_fwd.
extern "C" int vil_attn_fwd_full(const VilAttnDesc* d, const void* q_all, const void* k, const void* v,
                                 const float* bias_table, const float* g2l, const float* g2g,
                                 void* out_all, float* lse, float* lse_g, void* workspace, void* stream) {
  int e = check_common(d);
  if (e) return e;
  vil_prof_tag_desc(d);
  if (!q_all || !k || !v || !out_all || !lse || !lse_g) return VIL_E_NULL;
  if (d->G != 1 || d->only_glo) return VIL_E_BACKEND;
  if (d->backend == VIL_BACKEND_SCALAR || d->dtype == VIL_DTYPE_F32 || vil_mfma_supported(d, 0) != VIL_OK) return VIL_E_BACKEND;
  if (!workspace) return VIL_E_WORKSPACE;
  const int64_t es = 2;                                   // the MFMA family: bf16 or fp16
  VilParams p; memset(&p, 0, sizeof(p));
  vil_fill_params(p, d);
  const int64_t HG = (int64_t)d->H * d->G;
  p.q = (const char*)q_all + d->G * d->q_st * es; p.o = (char*)out_all + d->G * d->o_st * es;
  p.k = k; p.v = v; p.lse = lse;
  p.table = bias_table; p.g2l = g2l ? g2l + HG : nullptr;          // (2, H, G): [1] local query -> global key
  p.has_bias = bias_table != nullptr; p.has_g2l = g2l != nullptr;
  p.glo_rows = 1;
  p.q_g = q_all; p.o_g = out_all; p.lse_g = lse_g; p.g2l0 = g2l; p.g2g = g2g;
  p.delta = (float*)workspace;
  return vil_mfma_fwd(d, p, (hipStream_t)stream);
}

extern "C" int vil_attn_bwd(const VilAttnDesc* d, const void* q, const void* k, const void* v,
                            const void* out, const void* dout, const float* lse,
                            const float* bias_table, const float* g2l,
                            void* dq, void* dk, void* dv, float* dbias_table, float* dg2l,
                            void* workspace, void* stream) {
  int e = check_common(d);
  if (e) return e;
  vil_prof_tag_desc(d);
  if (!q || !k || !v || !out || !dout || !lse || !dq || !dk || !dv) return VIL_E_NULL;
  if (bias_table && !dbias_table) return VIL_E_NULL;
  if (g2l && d->G > 0 && !dg2l) return VIL_E_NULL;
  const int be = pick_backend(d, 1);
  if (!be) return d->backend == VIL_BACKEND_AUTO ? vil_scalar_supported(d) : VIL_E_BACKEND;
  if (!workspace && vil_attn_workspace_bytes(d, 1) > 0) return VIL_E_WORKSPACE;
  VilParams p; memset(&p, 0, sizeof(p));
  vil_fill_params(p, d);
  p.q = q; p.k = k; p.v = v; p.out = out; p.dout = dout; p.lse = (float*)lse;
  p.table = bias_table; p.g2l = g2l;
  p.has_bias = bias_table != nullptr; p.has_g2l = (g2l != nullptr) && d->G > 0;
  p.dq = dq; p.dk = dk; p.dv = dv; p.dtable = dbias_table; p.dg2l = dg2l;
  p.delta = (float*)workspace;
  return be == VIL_BACKEND_MFMA ? vil_mfma_bwd(d, p, (hipStream_t)stream)
                                : vil_scalar_bwd(d, p, (hipStream_t)stream);
}

extern "C" int vil_attn_bwd_full(const VilAttnDesc* d, const void* q_all, const void* k, const void* v,
                                 const void* out_all, const void* dout_all, const float* lse, const float* lse_g,
                                 const float* bias_table, const float* g2l, const float* g2g,
                                 void* dq_all, void* dk, void* dv, float* dbias_table, float* dg2l, float* dg2g,
                                 void* workspace, void* stream) {
  int e = check_common(d);
  if (e) return e;
  vil_prof_tag_desc(d);
  if (!q_all || !k || !v || !out_all || !dout_all || !lse || !lse_g || !dq_all || !dk || !dv) return VIL_E_NULL;
  if (bias_table && !dbias_table) return VIL_E_NULL;
  if ((g2l && !dg2l) || (g2g && !dg2g)) return VIL_E_NULL;
  if (d->G < 1 || d->G > 4 || d->only_glo) return VIL_E_BACKEND;
  if (d->backend == VIL_BACKEND_SCALAR || d->dtype == VIL_DTYPE_F32 || vil_mfma_supported(d, 1) != VIL_OK) return VIL_E_BACKEND;
  if (!workspace) return VIL_E_WORKSPACE;
  const int64_t es = 2;                                   // the MFMA family: bf16 or fp16
  VilParams p; memset(&p, 0, sizeof(p));
  vil_fill_params(p, d);
  const int64_t HG = (int64_t)d->H * d->G;
  p.q = (const char*)q_all + d->G * d->q_st * es; p.out = (const char*)out_all + d->G * d->o_st * es;
  p.dout = (const char*)dout_all + d->G * d->do_st * es; p.dq = (char*)dq_all + d->G * d->dq_st * es;
  p.k = k; p.v = v; p.lse = (float*)lse;
  p.table = bias_table; p.g2l = g2l ? g2l + HG : nullptr;
  p.has_bias = bias_table != nullptr; p.has_g2l = g2l != nullptr;
  p.dk = dk; p.dv = dv; p.dtable = dbias_table; p.dg2l = dg2l ? dg2l + HG : nullptr;
  p.glo_rows = 1;
  p.q_g = q_all; p.o_g = out_all; p.do_g = dout_all; p.dq_g = dq_all;
  p.lse_g = lse_g; p.g2l0 = g2l; p.g2g = g2g; p.dg2l0 = dg2l; p.dg2g = dg2g;
  p.delta = (float*)workspace;
  return vil_mfma_bwd(d, p, (hipStream_t)stream);
}

// ------------------------------------------------------------ host geometry helpers
extern "C" int vil_geom_mask(int nx, int ny, int W, int exact, int mode, uint8_t* mask) {
  if (!mask) return VIL_E_NULL;
  if (nx <= 0 || ny <= 0 || W <= 0) return VIL_E_SHAPE;
  if (mode < -1 || mode > 8) return VIL_E_MODE;
  if (exact < -1 || exact > 1 || (exact == 1 && mode != 0)) return VIL_E_EXACT;
  VilGeom g; vil_geom_init(g, nx, ny, W, exact, mode);
  const int kv = g.nact * g.W2;
  for (int m = 0; m < g.mx; ++m)
    for (int n = 0; n < g.my; ++n)
      for (int l = 0; l < g.W2; ++l) {
        const int qr = m * W + l / W, qc = n * W + l % W;
        uint8_t* row = mask + ((size_t)(m * g.my + n) * g.W2 + l) * kv;
        for (int a = 0; a < g.nact; ++a)
          for (int t = 0; t < g.W2; ++t) {
            int kr, kc;
            int st = vil_key_state(g, m, n, g.adr[a], g.adc[a], t / W, t % W, kr, kc);
            if (st != VIL_KEY_MASKED && exact == 1 && !vil_exact_window(W, qr, qc, kr, kc))
              st = VIL_KEY_MASKED;
            row[a * g.W2 + t] = st == VIL_KEY_MASKED;
          }
      }
  return kv;
}

extern "C" int vil_geom_bias_index(int W, int mode, int32_t* rel) {
  if (!rel) return VIL_E_NULL;
  if (W <= 0) return VIL_E_SHAPE;
  if (mode < -1 || mode > 8) return VIL_E_MODE;
  VilGeom g; vil_geom_init(g, W, W, W, 0, mode);
  const int kv = g.nact * g.W2;
  for (int l = 0; l < g.W2; ++l)
    for (int a = 0; a < g.nact; ++a)
      for (int t = 0; t < g.W2; ++t)
        rel[l * kv + a * g.W2 + t] = vil_bias_index(W, l / W, l % W, g.adr[a], g.adc[a], t / W, t % W);
  return kv;
}

// ------------------------------------------------------------ dynamic-LDS limits
#include <map>
#include <tuple>
#include <mutex>
int vil_ensure_dyn_lds(const void* kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return 0;
  static std::map<std::pair<int, const void*>, size_t> done;
  static std::mutex mu;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(mu);
  size_t& have = done[std::make_pair(dev, kernel)];
  if (have >= bytes) return 0;
  hipError_t he = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (he != hipSuccess) return (int)he;
  have = bytes;
  return 0;
}

// CU count of the current device, queried once per device (a device-attribute query is not a legal call while another
// thread's stream capture is in global mode: grid sizes must not cost an API call per launch)
int vil_cu_count() {
  static int cus_of[64] = {0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  int cus = (dev >= 0 && dev < 64) ? cus_of[dev] : 0;
  if (cus <= 0) {
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    if (dev >= 0 && dev < 64) cus_of[dev] = cus;
  }
  return cus;
}

int vil_persistent_grid(int waves_per_simd, int waves_per_wg, size_t lds, int H, int64_t units_total) {
  // resident workgroups per CU from the kernel's own launch bounds (waves per SIMD the register allocation was held
  // to: 4 SIMDs per CU) and its LDS footprint (160 KB per CU) -- deterministic, so that workspace layouts and the
  // fixed-point scale of the dQ histogram do not depend on a driver query; the CU count is the device's
  const int cus = vil_cu_count();
  int per_cu = waves_per_simd * 4 / (waves_per_wg > 0 ? waves_per_wg : 1);
  const int by_lds = lds > 0 ? (int)((160 * 1024) / lds) : per_cu;
  if (per_cu > by_lds) per_cu = by_lds;
  if (per_cu < 1) per_cu = 1;
  const int step = 8 * (H > 0 ? H : 1);
  int64_t need = (units_total + waves_per_wg - 1) / waves_per_wg;
  need = (need + step - 1) / step * step;
  int64_t n = (int64_t)per_cu * cus / step * step;
  if (n < step) n = step;
  if (n > need) n = need;
  if (n > VIL_MAX_PERSISTENT_WGS / step * step && VIL_MAX_PERSISTENT_WGS >= step) n = VIL_MAX_PERSISTENT_WGS / step * step;
  return (int)n;
}

// ------------------------------------------------------------ profiling sink
// Process-global and not thread-safe by design: a measurement aid for bench.py, the
// only state the library keeps.  Events are created once and reused.
static bool g_on = false;
struct ProfRec { int kid; double bytes, flops; int tag[8]; };
static int g_tag[8] = {0, 0, 0, 0, 0, 0, 0, 0};
void vil_prof_tag(int a0, int a1, int a2, int a3, int a4, int a5, int a6, int a7) {
  if (!g_on) return;
  g_tag[0] = a0; g_tag[1] = a1; g_tag[2] = a2; g_tag[3] = a3; g_tag[4] = a4; g_tag[5] = a5; g_tag[6] = a6; g_tag[7] = a7;
}
static std::vector<hipEvent_t> g_ev;
static std::vector<ProfRec> g_rec;
static size_t g_cap = 0;
static bool g_open = false;

void vil_prof_begin(int kid, hipStream_t s, double bytes, double flops) {
  if (!g_on || g_rec.size() >= g_cap) return;
  ProfRec r; r.kid = kid; r.bytes = bytes; r.flops = flops;
  for (int i = 0; i < 8; ++i) r.tag[i] = g_tag[i];
  g_rec.push_back(r);
  g_open = true;
  (void)hipEventRecord(g_ev[2 * (g_rec.size() - 1)], s);
}
void vil_prof_end(hipStream_t s) {
  if (!g_on || !g_open) return;
  g_open = false;
  (void)hipEventRecord(g_ev[2 * (g_rec.size() - 1) + 1], s);
}

extern "C" const char* vil_attn_kernel_name(int kid) {
  static const char* names[VIL_K_COUNT] = {"k_mfma_table", "k_mfma_fwd", "k_scalar_fwd", "k_delta", "k_scalar_bwd_dq",
                                           "k_scalar_bwd_dkdv", "k_reduce_glo", "k_reduce_bias", "k_mfma_bwd_dq",
                                           "k_mfma_bwd_dkdv", "k_glo_fwd", "k_glo_bwd", "k_wgrad", "k_wgrad_reduce",
                                           "k_dense_fwd", "k_dense_bwd_dq", "k_dense_bwd_dkdv", "k_dense_reduce"};
  return (kid >= 0 && kid < VIL_K_COUNT) ? names[kid] : "?";
}

extern "C" int vil_attn_profile_begin(int capacity) {
  if (capacity <= 0) return VIL_E_SHAPE;
  while (g_ev.size() < 2 * (size_t)capacity) {
    hipEvent_t e;
    hipError_t he = hipEventCreate(&e);
    if (he != hipSuccess) return (int)he;
    g_ev.push_back(e);
  }
  g_rec.clear(); g_cap = (size_t)capacity; g_on = true;
  return VIL_OK;
}

static int profile_drain(int cap, int* kid, float* ms, double* bytes, double* flops, int* tags);
extern "C" int vil_attn_profile_end(int cap, int* kid, float* ms, double* bytes, double* flops) {
  return profile_drain(cap, kid, ms, bytes, flops, nullptr);
}
extern "C" int vil_attn_profile_end2(int cap, int* kid, float* ms, double* bytes, double* flops, int* tags) {
  return profile_drain(cap, kid, ms, bytes, flops, tags);
}
static int profile_drain(int cap, int* kid, float* ms, double* bytes, double* flops, int* tags) {
  g_on = false;
  const int n = (int)(g_rec.size() < (size_t)cap ? g_rec.size() : (size_t)cap);
  for (int i = 0; i < n; ++i) {
    (void)hipEventSynchronize(g_ev[2 * i + 1]);
    float t = 0.f;
    (void)hipEventElapsedTime(&t, g_ev[2 * i], g_ev[2 * i + 1]);
    if (kid) kid[i] = g_rec[i].kid;
    if (ms) ms[i] = t;
    if (bytes) bytes[i] = g_rec[i].bytes;
    if (flops) flops[i] = g_rec[i].flops;
    if (tags) for (int j = 0; j < 8; ++j) tags[8 * i + j] = g_rec[i].tag[j];
  }
  g_rec.clear();
  return n;
}
