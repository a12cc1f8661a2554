"""ctypes binding of libvilattn.so (C ABI: include/vil_attn.h).

The product path has no CPU fallback: if the shared library has not been built
(``python -c 'import __graft_entry__ as g; g.build()'``) every entry point
raises."""
import ctypes
import os

# torch must be imported BEFORE the library is dlopen'ed: PyTorch-ROCm ships its own
# libamdhip64; loading ours first would bind libvilattn.so to a second HIP runtime
# (/opt/rocm) that does not share torch's device context and streams.
import torch  # noqa: F401,E402

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvilattn.so")
_AB_LIBRARY = False          # set only by use_library_for_ab() (tools/): entry points an older build lacks stay unbound


def use_library_for_ab(path):
    """tools/ only (A/B kernel timing against a library built from another revision): must be called before the
    first lib().  The product never calls this; it loads the in-tree libvilattn.so or raises."""
    global LIB_PATH, _AB_LIBRARY
    if _lib is not None:
        raise RuntimeError("use_library_for_ab() after the library was loaded")
    LIB_PATH, _AB_LIBRARY = path, True

DTYPE_F32, DTYPE_BF16, DTYPE_F16, DTYPE_F64 = 0, 1, 2, 3
BACKEND_AUTO, BACKEND_SCALAR, BACKEND_MFMA, BACKEND_MFMA_WAVE, BACKEND_MFMA_CW = 0, 1, 2, 3, 4
ABI_VERSION = 2

EXPORTS = ("vil_attn_abi_version", "vil_attn_strerror", "vil_attn_check", "vil_attn_workspace_bytes",
           "vil_attn_fwd", "vil_attn_bwd", "vil_geom_mask", "vil_geom_bias_index",
           "vil_attn_profile_begin", "vil_attn_profile_end", "vil_attn_profile_end2", "vil_attn_kernel_name",
           "vil_layernorm_workspace_bytes", "vil_layernorm_fwd", "vil_layernorm_bwd",
           "vil_layernorm_fwd_tokens", "vil_layernorm_bwd_tokens", "vil_patchify_fwd", "vil_patchify_bwd",
           "vil_glo_attn_fwd", "vil_glo_attn_bwd", "vil_attn_bwd_full", "vil_attn_fwd_full",
           "vil_dense_attn_supported", "vil_dense_attn_workspace_bytes", "vil_dense_attn_fwd", "vil_dense_attn_bwd", "vil_dense_attn_set_fwd_shape",
           "vil_colsum_workspace_bytes", "vil_colsum_bf16", "vil_colsum_f32",
           "vil_linear_wgrad_workspace_bytes", "vil_linear_wgrad", "vil_linear_wgrad_tune", "vil_linear_wgrad_set_plan", "vil_linear_wgrad_get_plan",
           "vil_resln_fwd", "vil_resln_bwd", "vil_gemm_workspace_bytes", "vil_gemm_bf16", "vil_gemm_tune", "vil_gemm_dgelu_bf16", "vil_gemm_gelu_bf16", "vil_gemm_tile_bf16", "vil_gemm_skinny_bf16", "vil_gemm_skinny_gelu_bf16",
           "vil_sc2d_qk", "vil_sc2d_av", "vil_sc2d_agrad", "vil_sc2d_mask",
           "vil_optim_plan_bytes", "vil_optim_plan_build", "vil_optim_adamw_step", "vil_optim_qhm_step",
           "vil_optim_adamw_step_amp", "vil_optim_qhm_step_amp", "vil_attn_cw_set_shape", "vil_attn_cw_plan")


class VilAttnDesc(ctypes.Structure):
    _fields_ = ([(n, ctypes.c_int32) for n in
                 ("B", "H", "M", "nx", "ny", "W", "G", "mode", "exact", "dtype", "only_glo", "backend")] +
                [("scale", ctypes.c_float), ("bias_side", ctypes.c_int32)] +
                [(t + s, ctypes.c_int64) for t in ("q", "k", "v", "o", "do", "dq", "dk", "dv")
                 for s in ("_sb", "_st", "_sh")] +
                [("mode_dev", ctypes.c_void_p)])


VIL_E_BACKEND = -10


class VilOptimTensor(ctypes.Structure):
    _fields_ = [("param", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("state1", ctypes.c_void_p),
                ("state2", ctypes.c_void_p), ("low", ctypes.c_void_p), ("n", ctypes.c_int64),
                ("grad_dtype", ctypes.c_int32), ("low_dtype", ctypes.c_int32), ("weight_decay", ctypes.c_float),
                ("lr", ctypes.c_float), ("lr_dev", ctypes.c_void_p)]


class VilAttnError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libvilattn error {code}: {msg}")
        self.code = code


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run __graft_entry__.build()).  There is no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        vp, fp = ctypes.c_void_p, ctypes.c_void_p
        dp = ctypes.POINTER(VilAttnDesc)
        L.vil_attn_abi_version.restype = ctypes.c_int
        L.vil_attn_strerror.restype = ctypes.c_char_p
        L.vil_attn_strerror.argtypes = [ctypes.c_int]
        L.vil_attn_check.restype = ctypes.c_int
        L.vil_attn_check.argtypes = [dp]
        L.vil_attn_workspace_bytes.restype = ctypes.c_size_t
        L.vil_attn_workspace_bytes.argtypes = [dp, ctypes.c_int]
        L.vil_attn_fwd.restype = ctypes.c_int
        L.vil_attn_fwd.argtypes = [dp, vp, vp, vp, fp, fp, vp, fp, vp, vp]
        L.vil_attn_bwd.restype = ctypes.c_int
        L.vil_attn_bwd.argtypes = [dp, vp, vp, vp, vp, vp, fp, fp, fp, vp, vp, vp, fp, fp, vp, vp]
        L.vil_geom_mask.restype = ctypes.c_int
        L.vil_geom_mask.argtypes = [ctypes.c_int] * 5 + [ctypes.c_void_p]
        L.vil_geom_bias_index.restype = ctypes.c_int
        L.vil_geom_bias_index.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.vil_attn_profile_begin.restype = ctypes.c_int
        L.vil_attn_profile_begin.argtypes = [ctypes.c_int]
        L.vil_attn_profile_end.restype = ctypes.c_int
        L.vil_attn_profile_end.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 4
        L.vil_attn_profile_end2.restype = ctypes.c_int
        L.vil_attn_profile_end2.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 5
        L.vil_attn_kernel_name.restype = ctypes.c_char_p
        L.vil_attn_kernel_name.argtypes = [ctypes.c_int]
        i64 = ctypes.c_int64
        L.vil_glo_attn_fwd.restype = ctypes.c_int
        L.vil_glo_attn_fwd.argtypes = [dp] + [vp] * 8
        L.vil_glo_attn_bwd.restype = ctypes.c_int
        L.vil_glo_attn_bwd.argtypes = [dp] + [vp] * 14
        L.vil_attn_bwd_full.restype = ctypes.c_int
        L.vil_attn_bwd_full.argtypes = [dp] + [vp] * 18
        if hasattr(L, "vil_attn_fwd_full"):          # (absent from an older A/B library: tools/)
            L.vil_attn_fwd_full.restype = ctypes.c_int
            L.vil_attn_fwd_full.argtypes = [dp] + [vp] * 11
        L.vil_gemm_dgelu_bf16.restype = ctypes.c_int
        L.vil_gemm_dgelu_bf16.argtypes = [vp, vp, vp, vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int64,
                                          ctypes.c_int64, ctypes.c_int64, vp]
        L.vil_gemm_skinny_bf16.restype = ctypes.c_int
        L.vil_gemm_skinny_bf16.argtypes = [ctypes.c_int, vp, vp, vp, vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_int64, ctypes.c_int64, vp]
        L.vil_gemm_gelu_bf16.restype = ctypes.c_int
        L.vil_gemm_gelu_bf16.argtypes = [vp, vp, vp, vp, vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int64,
                                         ctypes.c_int64, vp]
        if hasattr(L, "vil_gemm_tile_bf16"):
            L.vil_gemm_tile_bf16.restype = ctypes.c_int
            L.vil_gemm_tile_bf16.argtypes = [ctypes.c_int, vp, vp, vp, vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_int64, ctypes.c_int64, vp]
        L.vil_gemm_skinny_gelu_bf16.restype = ctypes.c_int
        L.vil_gemm_skinny_gelu_bf16.argtypes = [vp, vp, vp, vp, vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int64,
                                                ctypes.c_int64, vp]
        L.vil_dense_attn_supported.restype = ctypes.c_int
        L.vil_dense_attn_supported.argtypes = [dp]
        L.vil_dense_attn_workspace_bytes.restype = ctypes.c_size_t
        L.vil_dense_attn_workspace_bytes.argtypes = [dp, ctypes.c_int]
        L.vil_dense_attn_fwd.restype = ctypes.c_int
        L.vil_dense_attn_fwd.argtypes = [dp] + [vp] * 9
        if hasattr(L, "vil_dense_attn_set_fwd_shape"):
            L.vil_dense_attn_set_fwd_shape.restype = ctypes.c_int
            L.vil_dense_attn_set_fwd_shape.argtypes = [ctypes.c_int]
        L.vil_dense_attn_bwd.restype = ctypes.c_int
        L.vil_dense_attn_bwd.argtypes = [dp] + [vp] * 17
        L.vil_colsum_workspace_bytes.restype = ctypes.c_size_t
        L.vil_colsum_workspace_bytes.argtypes = [ctypes.c_int]
        L.vil_colsum_bf16.restype = ctypes.c_int
        L.vil_colsum_bf16.argtypes = [vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int64, vp, ctypes.c_int, vp, vp]
        L.vil_colsum_f32.restype = ctypes.c_int
        L.vil_colsum_f32.argtypes = [vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int64, vp, ctypes.c_int, vp, vp]
        L.vil_linear_wgrad_workspace_bytes.restype = ctypes.c_size_t
        L.vil_linear_wgrad_workspace_bytes.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int]
        L.vil_linear_wgrad.restype = ctypes.c_int
        L.vil_linear_wgrad.argtypes = [vp, vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                       vp, vp, ctypes.c_int, vp, vp]
        L.vil_linear_wgrad_tune.restype = ctypes.c_int
        L.vil_linear_wgrad_tune.argtypes = L.vil_linear_wgrad.argtypes
        L.vil_linear_wgrad_set_plan.restype = ctypes.c_int
        L.vil_linear_wgrad_set_plan.argtypes = [ctypes.c_int64] + [ctypes.c_int] * 6
        if hasattr(L, "vil_linear_wgrad_get_plan"):
            L.vil_linear_wgrad_get_plan.restype = ctypes.c_int
            L.vil_linear_wgrad_get_plan.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
        L.vil_resln_fwd.restype = ctypes.c_int
        L.vil_resln_fwd.argtypes = [vp, vp, ctypes.c_int, vp, ctypes.c_int64, vp, vp, vp, vp, ctypes.c_int, vp, vp,
                                    ctypes.c_int64, ctypes.c_int, ctypes.c_float, vp]
        L.vil_resln_bwd.restype = ctypes.c_int
        L.vil_resln_bwd.argtypes = [vp, ctypes.c_int, vp, vp, vp, vp, vp, vp, ctypes.c_int64, vp, vp, ctypes.c_int, vp, vp, vp,
                                    ctypes.c_int64, ctypes.c_int, vp]
        L.vil_gemm_workspace_bytes.restype = ctypes.c_size_t
        L.vil_gemm_workspace_bytes.argtypes = []
        L.vil_gemm_bf16.restype = ctypes.c_int
        L.vil_gemm_bf16.argtypes = [ctypes.c_int, vp, vp, vp, vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int64,
                                    ctypes.c_int64, vp, ctypes.c_size_t, vp]
        ci = ctypes.c_int
        for fn in (L.vil_sc2d_qk, L.vil_sc2d_av, L.vil_sc2d_agrad):
            fn.restype = ci
            fn.argtypes = [vp, vp, vp] + [ci] * 7 + [vp]
        L.vil_sc2d_mask.restype = ci
        L.vil_sc2d_mask.argtypes = [vp] + [ci] * 9 + [vp, vp]
        L.vil_gemm_tune.restype = ctypes.c_int
        L.vil_gemm_tune.argtypes = L.vil_gemm_bf16.argtypes
        if not (_AB_LIBRARY and not hasattr(L, "vil_optim_plan_bytes")):
            otp = ctypes.POINTER(VilOptimTensor)
            L.vil_optim_plan_bytes.restype = ctypes.c_size_t
            L.vil_optim_plan_bytes.argtypes = [otp, ci]
            L.vil_optim_plan_build.restype = ci
            L.vil_optim_plan_build.argtypes = [otp, ci, vp, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int)]
            L.vil_optim_adamw_step.restype = ci
            L.vil_optim_adamw_step.argtypes = [vp, ci, ctypes.c_float, ctypes.c_float, ctypes.c_float, ci, vp, vp]
            L.vil_optim_qhm_step.restype = ci
            L.vil_optim_qhm_step.argtypes = [vp, ci, ctypes.c_float, ctypes.c_float, vp, vp]
            # (required: optim._launch always takes the _amp entry points -- an A/B library without them cannot run the
            #  optimizer step, so there is nothing to guard)
            L.vil_optim_adamw_step_amp.restype = ci
            L.vil_optim_adamw_step_amp.argtypes = [vp, ci, ctypes.c_float, ctypes.c_float, ctypes.c_float, ci, vp, vp, vp, vp]
            L.vil_optim_qhm_step_amp.restype = ci
            L.vil_optim_qhm_step_amp.argtypes = [vp, ci, ctypes.c_float, ctypes.c_float, vp, vp, vp, vp]
        L.vil_layernorm_workspace_bytes.restype = ctypes.c_size_t
        L.vil_layernorm_workspace_bytes.argtypes = [i64, ctypes.c_int]
        L.vil_layernorm_fwd.restype = ctypes.c_int
        L.vil_layernorm_fwd.argtypes = [vp, ctypes.c_int, vp, vp, vp, ctypes.c_int, vp, vp, i64, ctypes.c_int,
                                        i64, i64, ctypes.c_float, vp]
        L.vil_layernorm_bwd.restype = ctypes.c_int
        L.vil_layernorm_bwd.argtypes = [vp, ctypes.c_int, vp, ctypes.c_int, vp, vp, vp, vp, ctypes.c_int, vp, vp, vp,
                                        i64, ctypes.c_int, i64, i64, i64, vp]
        L.vil_layernorm_fwd_tokens.restype = ctypes.c_int
        L.vil_layernorm_fwd_tokens.argtypes = [vp, ctypes.c_int, vp, vp, vp, ctypes.c_int, vp, vp, i64, ctypes.c_int,
                                               i64, ctypes.c_float, i64, i64, vp]
        L.vil_layernorm_bwd_tokens.restype = ctypes.c_int
        L.vil_layernorm_bwd_tokens.argtypes = [vp, ctypes.c_int, vp, ctypes.c_int, vp, vp, vp, vp, ctypes.c_int, vp, vp,
                                               vp, i64, ctypes.c_int, i64, i64, i64, i64, vp]
        L.vil_patchify_fwd.restype = ci
        L.vil_patchify_fwd.argtypes = [vp, vp, ci, vp, vp, ci] + [ci] * 7 + [vp]
        L.vil_patchify_bwd.restype = ci
        L.vil_patchify_bwd.argtypes = [vp, ci, vp, vp, vp, ci] + [ci] * 7 + [vp]
        if L.vil_attn_abi_version() != ABI_VERSION:
            raise RuntimeError("libvilattn.so ABI version mismatch; rebuild it")
        _lib = L
    return _lib


def source_fingerprint():
    """sha256 (16 hex digits) over the kernel sources and the ABI header: ties a PMC traffic file collected under
    rocprofv3 to the build it was collected on (bench.py attaches `traffic` only when they match)."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(_HERE, "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h")))
    files.append(os.path.join(os.path.dirname(_HERE), "include", "vil_attn.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def check(code):
    if code != 0:
        raise VilAttnError(code, lib().vil_attn_strerror(code).decode())


def profile_begin(capacity=8192):
    check(lib().vil_attn_profile_begin(int(capacity)))


def profile_end_tagged(capacity=8192):
    """Returns a list of (kernel_name, ms, algorithmic_bytes, algorithmic_flops, tag) per launch; tag = the 8-int
    problem tag (attention: B,H,M,nx,ny,W,G,mode; weight gradient: T,CO,CI,0,...)."""
    import numpy as np
    kid = np.zeros(capacity, dtype=np.int32)
    ms = np.zeros(capacity, dtype=np.float32)
    by = np.zeros(capacity, dtype=np.float64)
    fl = np.zeros(capacity, dtype=np.float64)
    tg = np.zeros((capacity, 8), dtype=np.int32)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    L = lib()
    n = L.vil_attn_profile_end2(int(capacity), vp(kid), vp(ms), vp(by), vp(fl), vp(tg))
    return [(L.vil_attn_kernel_name(int(kid[i])).decode(), float(ms[i]), float(by[i]), float(fl[i]),
             tuple(int(v) for v in tg[i])) for i in range(n)]


def profile_end(capacity=8192):
    """Returns a list of (kernel_name, ms, algorithmic_bytes, algorithmic_flops), one per launch."""
    import numpy as np
    kid = np.zeros(capacity, dtype=np.int32)
    ms = np.zeros(capacity, dtype=np.float32)
    by = np.zeros(capacity, dtype=np.float64)
    fl = np.zeros(capacity, dtype=np.float64)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    L = lib()
    n = L.vil_attn_profile_end(int(capacity), vp(kid), vp(ms), vp(by), vp(fl))
    return [(L.vil_attn_kernel_name(int(kid[i])).decode(), float(ms[i]), float(by[i]), float(fl[i])) for i in range(n)]
