"""GPU micro-benchmark: fused weight/bias-gradient kernel vs the library paths (see tools/gemm_probe.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vision_longformer_amd.linear import _wgrad, _GEMM_WS, _colsum


# the library-GEMM alternative (vil_gemm_bf16 op 2: dW by hipBLASLt, db by its BGRADB epilogue) -- measured 2-20x
# slower than the fused kernel on this stack (the bias-gradient epilogue kernels are poor), so it is not used
def _wgrad_lt(dy2, x2, want_db):
    """(dW, db) through hipBLASLt (vil_gemm_bf16 op 2: measured algorithm, bias gradient in the epilogue), or None."""
    T, co = dy2.shape
    ci = x2.shape[1]
    if not (dy2.is_cuda and dy2.dtype == torch.bfloat16 and x2.dtype == torch.bfloat16 and co % 8 == 0 and ci % 8 == 0
            and dy2.stride(1) == 1 and x2.stride(1) == 1 and dy2.stride(0) % 8 == 0 and x2.stride(0) % 8 == 0
            and dy2.data_ptr() % 16 == 0 and x2.data_ptr() % 16 == 0):
        return None
    import ctypes
    from vision_longformer_amd import _lib
    L = _lib.lib()
    ws = _GEMM_WS.get(dy2.device)
    if ws is None:
        ws = _GEMM_WS[dy2.device] = torch.empty(L.vil_gemm_workspace_bytes(), dtype=torch.uint8, device=dy2.device)
    dw = torch.empty(co, ci, dtype=torch.bfloat16, device=dy2.device)
    db = torch.empty(co, dtype=torch.bfloat16, device=dy2.device) if want_db else None
    vp = ctypes.c_void_p
    rc = L.vil_gemm_bf16(2, vp(x2.data_ptr()), vp(dy2.data_ptr()), vp(db.data_ptr()) if want_db else None,
                         vp(dw.data_ptr()), T, ci, co, x2.stride(0), dy2.stride(0), vp(ws.data_ptr()), ws.numel(),
                         vp(torch.cuda.current_stream(dy2.device).cuda_stream))
    if rc == _lib.VIL_E_BACKEND:
        return None
    _lib.check(rc)
    return dw, db



dev = torch.device("cuda:0")
def bench(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
T3, T4, T2, T1 = 128 * 197, 128 * 50, 128 * 785, 128 * 3137
shapes = {"s3 qkv": (T3, 384, 1152), "s3 proj": (T3, 384, 384), "s3 fc1": (T3, 384, 1536), "s3 fc2": (T3, 1536, 384),
          "s4 fc1": (T4, 768, 3072), "s4 fc2": (T4, 3072, 768), "s2 fc1": (T2, 192, 768), "s2 fc2": (T2, 768, 192),
          "s2 kv": (T2, 192, 384), "s1 fc1": (T1, 96, 384), "s1 fc2": (T1, 384, 96), "s1 kv": (T1, 96, 192)}
for name, (T, ci, co) in shapes.items():
    x = torch.randn(T, ci, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(T, co, device=dev, dtype=torch.bfloat16)
    t_f = bench(lambda: _wgrad(dy, x, True))
    r = _wgrad_lt(dy, x, True)
    if r is None:
        t_lt = float("nan"); err = float("nan")
    else:
        a = _wgrad(dy, x, True)
        err = max(float((r[0].float() - a[0].float()).abs().max() / a[0].float().abs().max()),
                  float((r[1].float() - a[1].float()).abs().max() / a[1].float().abs().max()))
        t_lt = bench(lambda: _wgrad_lt(dy, x, True))
    # hipBLASLt for dW only (tuned: vil_gemm_tune through _gemm-like call below) + the HBM-rate column-sum kernel for db
    import ctypes
    from vision_longformer_amd import _lib
    L = _lib.lib()
    ws = _GEMM_WS[dev]
    dwl = torch.empty(co, ci, dtype=torch.bfloat16, device=dev)
    vp = ctypes.c_void_p
    args = (2, vp(x.data_ptr()), vp(dy.data_ptr()), None, vp(dwl.data_ptr()), T, ci, co, x.stride(0), dy.stride(0),
            vp(ws.data_ptr()), ws.numel(), vp(torch.cuda.current_stream(dev).cuda_stream))
    rc = L.vil_gemm_tune(*args)
    def lt_nobias():
        L.vil_gemm_bf16(*args); return _colsum(dy)
    t_ltn = bench(lt_nobias) if rc == 0 else float("nan")
    t_cs = bench(lambda: _colsum(dy))
    S = 8; Tp = (T // S) * S
    def splitk():
        return torch.bmm(dy[:Tp].view(S, Tp // S, co).transpose(1, 2), x[:Tp].view(S, Tp // S, ci)).sum(0), dy.sum(0)
    t_sk = bench(splitk)
    fl = 2 * T * ci * co
    print(f"{name:8s} T={T:6d} {ci:4d}->{co:4d}  fused dW+db {t_f:7.1f} us ({fl/t_f/1e6:6.1f} TF)   hipBLASLt+BGRADB {t_lt:7.1f} us (rel.diff {err:.1e})   hipBLASLt(dW, tuned)+colsum {t_ltn:7.1f} us (colsum alone {t_cs:5.1f})   bmm splitK8 + sum {t_sk:7.1f} us")
