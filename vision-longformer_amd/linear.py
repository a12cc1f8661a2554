"""`VilLinear` / `vil_linear`: the projections around the hot path (every nn.Linear of msvit.py / longformer2d.py,
and the patch embedding restated as a Linear), same parameters / state-dict keys as nn.Linear.

On device tensors each of the three GEMMs of a projection goes through libvilattn.so:
  forward  Y = X W^T + b  and  input gradient dX = dY W :  hipBLASLt with the algorithm selected by measurement
                                                          per problem (`vil_gemm_bf16`);
  weight + bias gradient  dW = dY^T X, db = colsum(dY)  :  one fused MFMA kernel (`vil_linear_wgrad`: the contraction
           runs over 6 k ... 400 k tokens into a tiny output, which a library GEMM handles with 9-144 workgroups
           on a 256-CU part; measured 20-35 TFLOP/s on the stage-1 shapes, tools/gemm_probe.py), or for small
           token counts a library GEMM + the column-sum kernel.
No bias / cls-token gradient is left to PyTorch's multi-block reductions (they replay wrongly under hipGraph on this
stack, see `_colsum`).  Operands that do not fit a kernel's contract (fp32, odd sizes, CPU) take the PyTorch path;
the `torch.bmm` split-K fallback is what the fused kernel replaced (3-4x faster than the unsplit library call)."""
import os

import torch
from torch import nn
import torch.nn.functional as F


def _pick_split(T):
    s = 1
    while s < 64 and T % (2 * s) == 0 and T // (2 * s) >= 2048:
        s *= 2
    return s


_CS_WS = {}


def _colsum(dy2):
    """dy2.sum(0) through libvilattn's HBM-rate column-sum kernels (bf16 or fp32 rows, C % 8 == 0).  Also the
    only column sum that is safe inside a captured hipGraph on this stack: PyTorch's multi-block reductions
    zero a semaphore buffer with hipMemsetAsync, and memset nodes were observed to run out of order on replay
    (tools/graph_reduce_check.py: wrong sums in 199 of 200 replays for a (1600, 2304) bias gradient)."""
    T, co = dy2.shape
    if not (dy2.is_cuda and dy2.dtype in (torch.bfloat16, torch.float32) and co % 8 == 0 and dy2.stride(1) == 1
            and dy2.stride(0) % 8 == 0 and dy2.data_ptr() % 16 == 0 and T >= 1):
        return dy2.sum(0)
    import ctypes
    from . import _lib
    L = _lib.lib()
    key = (dy2.device, co)
    ws = _CS_WS.get(key)
    if ws is None:
        ws = _CS_WS[key] = torch.empty(L.vil_colsum_workspace_bytes(co) // 4, dtype=torch.float32, device=dy2.device)
    f32 = dy2.dtype == torch.float32
    out = torch.empty(co, dtype=dy2.dtype, device=dy2.device)
    fn = L.vil_colsum_f32 if f32 else L.vil_colsum_bf16
    _lib.check(fn(ctypes.c_void_p(dy2.data_ptr()), T, co, dy2.stride(0), ctypes.c_void_p(out.data_ptr()),
                  0 if f32 else 1, ctypes.c_void_p(ws.data_ptr()),
                  ctypes.c_void_p(torch.cuda.current_stream(dy2.device).cuda_stream)))
    return out


_WG_WS = {}
_WG_TUNED = set()

# Plan selection by measurement (the weight gradient's kernel / slice count, hipBLASLt's algorithm) makes the summation
# order of dW depend on a one-off timing.  deterministic_plans(True) -- or VIL_DETERMINISTIC_PLANS=1 in the
# environment -- switches the timing off: the library's cost model / hipBLASLt's first heuristic choice run, the same in
# every run and on every rank.  With measurement left on, export_plans() / import_plans() carry the selection across a
# resume (put the dict next to the checkpoint), and under torch.distributed with world size > 1 every rank takes rank 0's
# measured weight-gradient plans through sync_plans() -- ONE collective outside autograd, after the eager warm-up steps.
_DETERMINISTIC_PLANS = os.environ.get("VIL_DETERMINISTIC_PLANS", "0") not in ("", "0")


def deterministic_plans(enable=True):
    """No timing-based plan selection from here on (problems already tuned keep their plan; see import_plans)."""
    global _DETERMINISTIC_PLANS
    _DETERMINISTIC_PLANS = bool(enable)


def _wg_get_plan(T, co, ci):
    import ctypes
    from . import _lib
    buf = (ctypes.c_int * 5)()
    _lib.check(_lib.lib().vil_linear_wgrad_get_plan(T, co, ci, buf))
    return tuple(buf)


def export_plans():
    """{(T, C_out, C_in): (gen, mi, nj, m)} of every weight-gradient problem whose plan was measured or imported."""
    out = {}
    for (_, T, co, ci) in _WG_TUNED:
        gen, mi, nj, m, tuned = _wg_get_plan(T, co, ci)
        if tuned:
            out[(T, co, ci)] = (gen, mi, nj, m)
    return out


def import_plans(plans, device=None):
    """Restores export_plans(): the problems are marked as tuned, so no timing run replaces the imported plan."""
    from . import _lib
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    for (T, co, ci), (gen, mi, nj, m) in plans.items():
        _lib.check(_lib.lib().vil_linear_wgrad_set_plan(T, co, ci, gen, mi, nj, m))
        _WG_TUNED.add((dev, T, co, ci))


def sync_plans(src=0, device=None):
    """Every rank takes rank `src`'s weight-gradient plans (same dW summation order on every rank, and -- with
    export_plans() saved next to the checkpoint -- across a resume).  A COLLECTIVE on the default process group, called
    outside autograd at a point every rank reaches together: engine.sync_replicas() and GraphedTrainStep call it after
    the warm-up steps.  (Round 4 broadcast a plan from inside the backward pass whenever a rank tuned a new problem; a
    rank that had imported its plans, or saw another token count, would have skipped that collective and hung the
    others.)  import_plans() / deterministic_plans() must be applied identically on every rank; problems only some
    ranks have met keep their local plan."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return
    box = [export_plans() if dist.get_rank() == src else None]
    dist.broadcast_object_list(box, src=src)
    if dist.get_rank() != src and box[0]:
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        import_plans(box[0], device)


def _wgrad(dy2, x2, want_db):
    """(dW, db) of y = x W^T + b through libvilattn's fused MFMA weight/bias-gradient kernel, or None
    when the operands do not fit its contract (then the caller uses library GEMMs)."""
    T, co = dy2.shape
    ci = x2.shape[1]
    if not (dy2.is_cuda and dy2.dtype == torch.bfloat16 and x2.dtype == torch.bfloat16 and co % 8 == 0 and ci % 8 == 0
            and dy2.stride(1) == 1 and x2.stride(1) == 1 and dy2.stride(0) % 8 == 0 and x2.stride(0) % 8 == 0
            and dy2.data_ptr() % 16 == 0 and x2.data_ptr() % 16 == 0 and T >= 1024):
        return None
    import ctypes
    from . import _lib
    L = _lib.lib()
    need = L.vil_linear_wgrad_workspace_bytes(T, co, ci)
    ws = _WG_WS.get(dy2.device)
    if ws is None or ws.numel() * 4 < need:
        ws = _WG_WS[dy2.device] = torch.empty(max(need // 4 + 64, 1 << 22), dtype=torch.float32, device=dy2.device)
    dw = torch.empty(co, ci, dtype=torch.bfloat16, device=dy2.device)
    db = torch.empty(co, dtype=torch.bfloat16, device=dy2.device) if want_db else None
    vp = ctypes.c_void_p
    args = (vp(dy2.data_ptr()), vp(x2.data_ptr()), T, co, ci, dy2.stride(0), x2.stride(0),
            vp(dw.data_ptr()), vp(db.data_ptr()) if want_db else None, 1, vp(ws.data_ptr()),
            vp(torch.cuda.current_stream(dy2.device).cuda_stream))
    key = (dy2.device, T, co, ci)
    if key not in _WG_TUNED and not _DETERMINISTIC_PLANS and not torch.cuda.is_current_stream_capturing():
        # one-off plan selection per problem by measurement (synchronises; never inside a captured region)
        _WG_TUNED.add(key)
        _lib.check(L.vil_linear_wgrad_tune(*args))
    _lib.check(L.vil_linear_wgrad(*args))
    return dw, db


_SKINNY_MIN_T = 8192          # tokens from which the weights-in-registers forward kernel is taken (tools/gemm_bench.py)
_SKINNY_FORCE = False
_GELU_EPILOGUE_MAX_K = 96     # fc1 + GELU in one launch up to this K (0: never).  In-step A/B (tools/ab_bench.py, profiles/r03_ab_gelu_epilogue.txt):
                              # K = 96 gains 0.04 (ViL-Small) / 0.1 ms (Medium-Deep) per step; with K = 192 included the step is 0.05 ms slower


def _gemm_skinny(op, inp2, w, bias):
    """op 0: inp2 @ w.T (+ bias); op 1: inp2 @ w -- through vil_gemm_skinny_bf16 (small weight matrix, huge T), or None
    outside its contract."""
    T, K = inp2.shape
    N = w.shape[0] if op == 0 else w.shape[1]
    ok_shape = (K in (96, 192) and N <= 768) or (K in (288, 384, 576, 768) and N <= 256)
    if not (ok_shape and N % 8 == 0 and (T >= _SKINNY_MIN_T or _SKINNY_FORCE)
            and inp2.is_cuda and inp2.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and w.is_contiguous()
            and (w.shape[1] if op == 0 else w.shape[0]) == K and inp2.stride(1) == 1 and inp2.stride(0) % 8 == 0
            and inp2.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0
            and (bias is None or (op == 0 and bias.dtype == torch.bfloat16 and bias.is_contiguous() and bias.data_ptr() % 16 == 0))):
        return None
    import ctypes
    from . import _lib
    out = torch.empty(T, N, dtype=torch.bfloat16, device=inp2.device)
    vp = ctypes.c_void_p
    rc = _lib.lib().vil_gemm_skinny_bf16(op, vp(inp2.data_ptr()), vp(w.data_ptr()), vp(bias.data_ptr()) if bias is not None else None,
                                         vp(out.data_ptr()), T, K, N, inp2.stride(0), N,
                                         vp(torch.cuda.current_stream(inp2.device).cuda_stream))
    if rc == _lib.VIL_E_BACKEND:
        return None
    _lib.check(rc)
    return out


def _gemm_skinny_gelu(inp2, w, bias):
    """(h, gelu(h)) with h = inp2 @ w.T + bias in one launch (vil_gemm_skinny_gelu_bf16: fc1 of the MLP block with the
    exact GELU in its epilogue), or None outside the kernel's contract (K = 96 / 192, N <= 768, many tokens)."""
    T, K = inp2.shape
    N = w.shape[0]
    if not (K <= _GELU_EPILOGUE_MAX_K and K in (96, 192) and N <= 768 and N % 8 == 0 and (T >= _SKINNY_MIN_T or _SKINNY_FORCE)
            and inp2.is_cuda and inp2.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and w.is_contiguous()
            and w.shape[1] == K and inp2.stride(1) == 1 and inp2.stride(0) % 8 == 0
            and inp2.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0
            and (bias is None or (bias.dtype == torch.bfloat16 and bias.is_contiguous() and bias.data_ptr() % 16 == 0))):
        return None
    import ctypes
    from . import _lib
    both = torch.empty(2, T, N, dtype=torch.bfloat16, device=inp2.device)
    vp = ctypes.c_void_p
    rc = _lib.lib().vil_gemm_skinny_gelu_bf16(vp(inp2.data_ptr()), vp(w.data_ptr()), vp(bias.data_ptr()) if bias is not None else None,
                                              vp(both[0].data_ptr()), vp(both[1].data_ptr()), T, K, N, inp2.stride(0), N,
                                              vp(torch.cuda.current_stream(inp2.device).cuda_stream))
    if rc == _lib.VIL_E_BACKEND:
        return None
    _lib.check(rc)
    return both[0], both[1]


_GELU_TILE_MIN_K = 384        # fc1 + GELU through the 128 x 128 tile kernel (vil_gemm_gelu_bf16) from this K on (0: never); in-step A/B:
                              # profiles/r03_ab_gelu_epilogue.txt


def _gemm_tile_gelu(inp2, w, bias):
    """(h, gelu(h)) with h = inp2 @ w.T + bias in one launch of the tiled kernel (vil_gemm_gelu_bf16, csrc/vil_gemm_fused.hip),
    or None outside its contract."""
    T, K = inp2.shape
    N = w.shape[0]
    if not (_GELU_TILE_MIN_K and K >= _GELU_TILE_MIN_K and K % 32 == 0 and N % 128 == 0 and T >= 1024
            and inp2.is_cuda and inp2.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and w.is_contiguous()
            and w.shape[1] == K and inp2.stride(1) == 1 and inp2.stride(0) % 8 == 0
            and inp2.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0
            and (bias is None or (bias.dtype == torch.bfloat16 and bias.is_contiguous() and bias.data_ptr() % 16 == 0))):
        return None
    import ctypes
    from . import _lib
    both = torch.empty(2, T, N, dtype=torch.bfloat16, device=inp2.device)
    vp = ctypes.c_void_p
    rc = _lib.lib().vil_gemm_gelu_bf16(vp(inp2.data_ptr()), vp(w.data_ptr()), vp(bias.data_ptr()) if bias is not None else None,
                                       vp(both[0].data_ptr()), vp(both[1].data_ptr()), T, K, N, inp2.stride(0), N,
                                       vp(torch.cuda.current_stream(inp2.device).cuda_stream))
    if rc == _lib.VIL_E_BACKEND:
        return None
    _lib.check(rc)
    return both[0], both[1]


_GEMM_WS = {}
_GEMM_TUNED = set()


def _gemm(op, inp2, w, bias):
    """op 0: inp2 @ w.T (+ bias);  op 1: inp2 @ w -- hipBLASLt through libvilattn with the algorithm chosen by
    measurement per problem (vil_gemm_bf16).  None when the operands do not fit the contract."""
    T, K = inp2.shape
    N = w.shape[0] if op == 0 else w.shape[1]
    if not (inp2.is_cuda and inp2.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and w.is_contiguous()
            and (w.shape[1] if op == 0 else w.shape[0]) == K and K % 8 == 0 and N % 8 == 0 and T >= 1
            and inp2.stride(1) == 1 and inp2.stride(0) % 8 == 0 and inp2.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0
            and (bias is None or (bias.dtype == torch.bfloat16 and bias.is_contiguous()))):
        return None
    import ctypes
    from . import _lib
    L = _lib.lib()
    ws = _GEMM_WS.get(inp2.device)
    if ws is None:
        ws = _GEMM_WS[inp2.device] = torch.empty(L.vil_gemm_workspace_bytes(), dtype=torch.uint8, device=inp2.device)
    out = torch.empty(T, N, dtype=torch.bfloat16, device=inp2.device)
    vp = ctypes.c_void_p
    args = (op, vp(inp2.data_ptr()), vp(w.data_ptr()), vp(bias.data_ptr()) if bias is not None else None,
            vp(out.data_ptr()), T, K, N, inp2.stride(0), N, vp(ws.data_ptr()), ws.numel(),
            vp(torch.cuda.current_stream(inp2.device).cuda_stream))
    key = (inp2.device, op, T, K, N, inp2.stride(0), bias is not None)
    if key not in _GEMM_TUNED and not _DETERMINISTIC_PLANS and not torch.cuda.is_current_stream_capturing():
        # explicit, one-off algorithm selection per problem (synchronises; never inside a captured region):
        # the launch call itself stays asynchronous
        _GEMM_TUNED.add(key)
        rc = L.vil_gemm_tune(*args)
        if rc not in (0, _lib.VIL_E_BACKEND):        # (BACKEND: no algorithm for this problem -> the caller's fallback)
            _lib.check(rc)
    rc = L.vil_gemm_bf16(*args)
    if rc == _lib.VIL_E_BACKEND:
        return None
    _lib.check(rc)
    return out


_TILE_GEMM = True             # plain projections of the dense stages through vil_gemm_tile_bf16 where it beats the tuned library GEMM
                              # (tools/tile_gemm_probe.py, profiles/r04_tile_gemm_probe.txt; in-step A/B: tools/ab_bench.py)


def _tile_gemm_takes(op, K, N):
    """op 0: K = in_features, N = out_features; op 1: K = out_features (the contraction), N = in_features.  The shapes the
    persistent tile kernel wins at the dense stages' token counts: forward with a contraction <= 384 (qkv, proj of stage 3)
    or <= 768 into <= 768 features (proj of stage 4); input gradient of the square projections (<= 768 x 768).  The long
    contractions (fc2 forward, qkv / fc1 input gradients) stay with hipBLASLt: 34-37 us against 45-56 here."""
    if op == 0:
        return K <= 384 or (K <= 768 and N <= 768)
    return K <= 768 and N <= 768


def _gemm_tile(op, inp2, w, bias):
    """vil_gemm_tile_bf16 (csrc/vil_gemm_fused.hip: the loader-wave tile kernels without their activation epilogues), or
    None outside its contract / the shapes it is taken for."""
    T, K = inp2.shape
    N = w.shape[0] if op == 0 else w.shape[1]
    if not (_TILE_GEMM and _tile_gemm_takes(op, K, N) and K % 32 == 0 and N % 128 == 0 and T >= 1024
            and inp2.is_cuda and inp2.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and w.is_contiguous()
            and (w.shape[1] if op == 0 else w.shape[0]) == K and inp2.stride(1) == 1 and inp2.stride(0) % 8 == 0
            and inp2.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0
            and (bias is None or (op == 0 and bias.dtype == torch.bfloat16 and bias.is_contiguous() and bias.data_ptr() % 16 == 0))):
        return None
    import ctypes
    from . import _lib
    out = torch.empty(T, N, dtype=torch.bfloat16, device=inp2.device)
    vp = ctypes.c_void_p
    rc = _lib.lib().vil_gemm_tile_bf16(op, vp(inp2.data_ptr()), vp(w.data_ptr()), vp(bias.data_ptr()) if bias is not None else None,
                                       vp(out.data_ptr()), T, K, N, inp2.stride(0), N,
                                       vp(torch.cuda.current_stream(inp2.device).cuda_stream))
    if rc == _lib.VIL_E_BACKEND:
        return None
    _lib.check(rc)
    return out


def _fwd_gemm(x2, weight, bias):
    """x2 @ weight.T (+ bias): the weights-in-registers kernel for the short-K / huge-T projections of stages 1-2
    (csrc/vil_gemm_skinny.hip), the tuned library GEMM otherwise; None when neither takes the operands."""
    y = _gemm_skinny(0, x2, weight, bias)
    if y is None:
        y = _gemm_tile(0, x2, weight, bias)
    return y if y is not None else _gemm(0, x2, weight, bias)


def _bwd_gemm(dy2, weight):
    """dy2 @ weight (the input gradient): the weights-in-registers kernel where it applies, else the library GEMM"""
    dx = _gemm_skinny(1, dy2, weight, None)
    if dx is None:
        dx = _gemm_tile(1, dy2, weight, None)
    return dx if dx is not None else _gemm(1, dy2, weight, None)


class _SplitKLinearFn(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        x2 = x.reshape(-1, x.shape[-1])
        y = _fwd_gemm(x2, weight, bias) if x.is_cuda else None
        if y is None:
            return F.linear(x, weight, bias)
        return y.view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        co, ci = weight.shape
        dy2 = dy.reshape(-1, co)
        x2 = x.reshape(-1, ci)
        T = x2.shape[0]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _bwd_gemm(dy2, weight) if dy2.is_cuda else None
            dx = (dx if dx is not None else dy2 @ weight).view(x.shape)
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            fused = _wgrad(dy2, x2, want_db)
            if fused is not None:
                return dx, fused[0].to(weight.dtype), (fused[1] if want_db else None)
            S = _pick_split(T) if T >= 4096 else 1
            if S > 1 and dy2.is_contiguous() and x2.is_contiguous():
                parts = torch.bmm(dy2.view(S, T // S, co).transpose(1, 2), x2.view(S, T // S, ci))
                dw = parts.sum(0, dtype=torch.float32).to(weight.dtype)
            else:
                dw = dy2.t() @ x2
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = _colsum(dy2)
        return dx, dw, db


class _LinearGeluOutFn(torch.autograd.Function):
    """(h, a) = (x W^T + b, gelu(h)): fc1 of the MLP block with nn.GELU() in the GEMM's epilogue where the
    weights-in-registers kernel serves the shape (stages 1-2), Linear + F.gelu otherwise.  a is NOT differentiable here:
    it feeds _GeluLinearFn, whose backward returns the gradient with respect to h (GELU derivative included)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        x2 = x.reshape(-1, x.shape[-1])
        both = _gemm_skinny_gelu(x2, weight, bias) if x.is_cuda else None
        if both is None and x.is_cuda:
            both = _gemm_tile_gelu(x2, weight, bias)
        if both is not None:
            h, a = (t.view(*x.shape[:-1], weight.shape[0]) for t in both)
        else:
            h = _fwd_gemm(x2, weight, bias) if x.is_cuda else None
            h = F.linear(x, weight, bias) if h is None else h.view(*x.shape[:-1], weight.shape[0])
            a = F.gelu(h)
        ctx.mark_non_differentiable(a)
        ctx.set_materialize_grads(False)      # (otherwise autograd hands backward a zero tensor the size of a: a fill kernel per block)
        return h, a

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dh, _da):
        return _SplitKLinearFn.backward(ctx, dh)


# The fused kernel re-streams its operands per 128 x 128 tile through a two-slot ring.  Kernels alone (tools/mlp_bench.py) it
# beats library GEMM + gelu_backward at K = 96 / 192 (195 vs 371 us at ViL-Small's 401 536 tokens, 115 vs 161 us), is level
# at K = 384 and behind at K = 768; INSIDE the training step it wins at every K (tools/ab_bench.py, same box, alternating:
# -0.20 ms per ViL-Small step from K = 384, -0.05 ms more from K = 768; profiles/r03_ab_dgelu_k.txt).
_DGELU_MAX_K = 768
_DGELU_FORCE = False          # tools / tests: run the fused kernel at every shape it accepts


def _dgrad_dgelu(dy2, weight, h2):
    """(dy2 @ weight) * gelu'(h2) in one launch (vil_gemm_dgelu_bf16), or None outside the kernel's contract."""
    T, K = dy2.shape
    N = weight.shape[1]
    if K > _DGELU_MAX_K and not _DGELU_FORCE:
        return None
    if not (dy2.is_cuda and dy2.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and h2.dtype == torch.bfloat16
            and weight.is_contiguous() and weight.shape[0] == K and h2.shape == (T, N) and K % 32 == 0 and N % 128 == 0
            and dy2.stride(1) == 1 and h2.stride(1) == 1 and dy2.stride(0) % 8 == 0 and h2.stride(0) % 8 == 0
            and dy2.data_ptr() % 16 == 0 and weight.data_ptr() % 16 == 0 and h2.data_ptr() % 16 == 0):
        return None
    import ctypes
    from . import _lib
    dh = torch.empty(T, N, dtype=torch.bfloat16, device=dy2.device)
    vp = ctypes.c_void_p
    rc = _lib.lib().vil_gemm_dgelu_bf16(vp(dy2.data_ptr()), vp(weight.data_ptr()), vp(h2.data_ptr()), vp(dh.data_ptr()), T, K, N,
                                        dy2.stride(0), h2.stride(0), N, vp(torch.cuda.current_stream(dy2.device).cuda_stream))
    if rc == _lib.VIL_E_BACKEND:
        return None
    _lib.check(rc)
    return dh


class _GeluLinearFn(torch.autograd.Function):
    """y = gelu(h) W^T + b (the tail of the MLP block, reference msvit.py:29-33) with ONE launch for the input gradient:
    dh = (dy W) * gelu'(h) -- fc2's dgrad GEMM with the exact-erf GELU backward in its epilogue -- instead of a library
    GEMM + an elementwise kernel that re-reads h and the GEMM's output."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, h, weight, bias, a=None):
        if a is None:
            a = F.gelu(h)
        ctx.save_for_backward(h, a, weight)
        ctx.has_bias = bias is not None
        a2 = a.reshape(-1, a.shape[-1])
        y = _fwd_gemm(a2, weight, bias) if a.is_cuda else None
        if y is None:
            return F.linear(a, weight, bias)
        return y.view(*a.shape[:-1], weight.shape[0])

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        h, a, weight = ctx.saved_tensors
        co, ci = weight.shape
        dy2 = dy.reshape(-1, co)
        a2, h2 = a.reshape(-1, ci), h.reshape(-1, ci)
        dh = dw = db = None
        if ctx.needs_input_grad[0]:
            dh = _dgrad_dgelu(dy2, weight, h2)
            if dh is None:
                da = _gemm(1, dy2, weight, None)
                da = da if da is not None else dy2 @ weight
                dh = torch.ops.aten.gelu_backward(da, h2)
            dh = dh.view(h.shape)
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            fused = _wgrad(dy2, a2, want_db)
            if fused is not None:
                return dh, fused[0].to(weight.dtype), (fused[1] if want_db else None), None
            dw = dy2.t() @ a2
        if want_db:
            db = _colsum(dy2)
        return dh, dw, db, None


def vil_gelu_linear(h, weight, bias, a=None):
    """F.linear(F.gelu(h), weight, bias) (exact GELU) whose backward fuses the GELU derivative into the input-gradient
    GEMM; same autocast semantics as vil_linear.  a: gelu(h) when the caller already has it (vil_linear_gelu)."""
    if h.is_cuda:                   # (also under no_grad: the evaluation forward runs the same kernels)
        if torch.is_autocast_enabled("cuda"):
            dt = torch.get_autocast_dtype("cuda")
            h, w = h.to(dt), weight.to(dt)
            b = bias.to(dt) if bias is not None else None
            with torch.autocast("cuda", enabled=False):
                return _GeluLinearFn.apply(h, w, b, a.to(dt) if a is not None else None)
        return _GeluLinearFn.apply(h, weight, bias, a)
    return F.linear(F.gelu(h) if a is None else a, weight, bias)


def vil_linear_gelu(x, weight, bias):
    """(h, gelu(h)) with h = F.linear(x, weight, bias): the head of the MLP block (reference msvit.py:29-31) as one
    autograd node -- one launch where the weights-in-registers GEMM serves the shape.  Feed both to vil_gelu_linear."""
    if x.is_cuda:
        if torch.is_autocast_enabled("cuda"):
            dt = torch.get_autocast_dtype("cuda")
            x, w = x.to(dt), weight.to(dt)
            b = bias.to(dt) if bias is not None else None
            with torch.autocast("cuda", enabled=False):
                return _LinearGeluOutFn.apply(x, w, b)
        return _LinearGeluOutFn.apply(x, weight, bias)
    h = F.linear(x, weight, bias)
    return h, F.gelu(h)


def vil_linear(x, weight, bias):
    """F.linear with the library's weight / bias gradient on device tensors (any number of tokens: every bias
    gradient must stay off PyTorch's multi-block reductions, see _colsum)."""
    if x.is_cuda:
        if torch.is_autocast_enabled("cuda"):
            dt = torch.get_autocast_dtype("cuda")
            # autocast semantics of nn.Linear: inputs and parameters in the autocast dtype
            x, w = x.to(dt), weight.to(dt)
            b = bias.to(dt) if bias is not None else None
            with torch.autocast("cuda", enabled=False):
                return _SplitKLinearFn.apply(x, w, b)
        return _SplitKLinearFn.apply(x, weight, bias)
    return F.linear(x, weight, bias)


def _adjacent(a, b):
    """b starts where a ends, in ONE storage (rows of one packed matrix)"""
    return (a.dtype == b.dtype and a.is_contiguous() and b.is_contiguous() and a.device == b.device
            and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()
            and a.data_ptr() + a.numel() * a.element_size() == b.data_ptr())


class _PairLinearFn(torch.autograd.Function):
    """y = x [W1; W2]^T + [b1; b2] for two projections of the same input whose parameters are ROWS OF ONE PACKED MATRIX
    (pack_pair): one forward GEMM, one input-gradient GEMM, one weight-gradient launch; the four parameter gradients
    are row slices of the packed gradient (views: no cat in the forward, no split copies in the backward)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, w1, b1, w2, b2):
        co1, ci = w1.shape
        co = co1 + w2.shape[0]
        w = w1.as_strided((co, ci), (ci, 1))
        b = b1.as_strided((co,), (1,)) if b1 is not None else None
        ctx.save_for_backward(x, w)
        ctx.co1, ctx.has_bias = co1, b is not None
        y = _fwd_gemm(x.reshape(-1, ci), w, b)
        if y is None:
            y = F.linear(x.reshape(-1, ci), w, b)
        return y.view(*x.shape[:-1], co)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        co, ci = w.shape
        dy2, x2 = dy.reshape(-1, co), x.reshape(-1, ci)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _bwd_gemm(dy2, w)
            dx = (dx if dx is not None else dy2 @ w).view(x.shape)
        fused = _wgrad(dy2, x2, ctx.has_bias)
        if fused is not None:
            dw, db = fused[0].to(w.dtype), fused[1]
        else:
            dw = (dy2.t() @ x2).to(w.dtype)
            db = _colsum(dy2) if ctx.has_bias else None
        c = ctx.co1
        return dx, dw[:c], (db[:c] if db is not None else None), dw[c:], (db[c:] if db is not None else None)


def pack_pair(lin1, lin2):
    """Re-homes the parameters of two Linears over the same input as rows of one packed (co1 + co2, ci) matrix (and
    one packed bias): same Parameter objects, names, shapes and values (state dict unchanged), new storage.  Done
    once; repeated only when something re-allocated a parameter (.to(), a dtype change)."""
    with torch.no_grad():
        w1, w2 = lin1.weight, lin2.weight
        pw = torch.empty(w1.shape[0] + w2.shape[0], w1.shape[1], dtype=w1.dtype, device=w1.device)
        pw[:w1.shape[0]].copy_(w1); pw[w1.shape[0]:].copy_(w2)
        w1.data, w2.data = pw[:w1.shape[0]], pw[w1.shape[0]:]
        if lin1.bias is not None:
            b1, b2 = lin1.bias, lin2.bias
            pb = torch.empty(b1.numel() + b2.numel(), dtype=b1.dtype, device=b1.device)
            pb[:b1.numel()].copy_(b1); pb[b1.numel():].copy_(b2)
            b1.data, b2.data = pb[:b1.numel()], pb[b1.numel():]


def vil_linear_pair(x, lin1, lin2):
    """lin1(x) and lin2(x) as ONE projection (columns [lin1 | lin2]); None when the pair cannot run packed (the
    caller then applies the two Linears separately)."""
    w1, w2, b1, b2 = lin1.weight, lin2.weight, lin1.bias, lin2.bias
    if not (x.is_cuda and w1.dtype == w2.dtype and w1.dtype == x.dtype == torch.bfloat16
            and (b1 is None) == (b2 is None) and w1.shape[1] == w2.shape[1]):
        return None
    if not (_adjacent(w1, w2) and (b1 is None or _adjacent(b1, b2))):
        if torch.cuda.is_current_stream_capturing():
            return None
        pack_pair(lin1, lin2)
    with torch.autocast("cuda", enabled=False):
        return _PairLinearFn.apply(x, w1, b1, w2, b2)


class VilLinear(nn.Linear):
    def forward(self, x):
        return vil_linear(x, self.weight, self.bias)


class _ExpandRows(torch.autograd.Function):
    """t (1, G, C) -> (B, G, C) (the cls tokens prepended to every image, msvit.py:198); the backward's sum
    over the batch is a column sum of the (B, G*C) gradient."""

    @staticmethod
    def forward(ctx, t, B):
        ctx.shape = t.shape
        return t.expand(B, -1, -1)

    @staticmethod
    def backward(ctx, g):
        B = g.shape[0]
        g2 = g.reshape(B, -1)
        if g2.stride(-1) != 1:
            g2 = g2.contiguous()
        return _colsum(g2).view(ctx.shape), None


def expand_rows(t, B):
    return _ExpandRows.apply(t, B) if t.is_cuda else t.expand(B, -1, -1)
