"""Fused local attention of Vision Longformer as a torch.autograd.Function over
the C ABI of libvilattn.so (include/vil_attn.h).

Replaces everything between the q/kv projections and the output projection of
the reference module: the local-query rows (src/models/layers/longformer2d.py:134-204
and the SlidingChunk2D autograd function, src/models/layers/slidingchunk_2d.py:202-246),
the global-token query rows (longformer2d.py:210-227; `vil_full_attention`) and, as the
one-chunk case, the dense `Attention` of the s0 stages (msvit.py:91-120;
`vil_dense_attention`): no chunk/pad/roll copies, no score or (H,N,N) bias tensor;
the backward recomputes probabilities from the saved log-sum-exp.

PyTorch is plumbing here (device memory, the current HIP stream); the compute
is the HIP kernels.  There is no eager fallback: CPU tensors or a missing
library raise."""
import ctypes

import torch

from . import _lib

_DT = {torch.float32: _lib.DTYPE_F32, torch.bfloat16: _lib.DTYPE_BF16, torch.float16: _lib.DTYPE_F16}
_BACKENDS = {"auto": _lib.BACKEND_AUTO, "scalar": _lib.BACKEND_SCALAR, "mfma": _lib.BACKEND_MFMA,
             "mfma_wave": _lib.BACKEND_MFMA_WAVE,     # the wave-per-chunk kernels only (rounds 1-5): A/B row
             "mfma_cw": _lib.BACKEND_MFMA_CW}         # the chunk-workgroup forward wherever it can run (round 6)

# process-wide default kernel family ("auto" | "scalar" | "mfma"); tests override per call
DEFAULT_BACKEND = "auto"


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _strides(t, M):
    """(batch, token, head) element strides of a (B, T, H*M) tensor whose last
    dim is contiguous: heads are M apart inside a token row."""
    return t.stride(0), t.stride(1), M


def _last_contig(t):
    return t if t.stride(-1) == 1 else t.contiguous()


def _make_desc(q, k, v, out, cfg, backend):
    B, Nloc, C = q.shape
    H, M = cfg["H"], C // cfg["H"]
    d = _lib.VilAttnDesc()
    d.B, d.H, d.M = B, H, M
    d.nx, d.ny, d.W, d.G = cfg["nx"], cfg["ny"], cfg["W"], cfg["G"]
    d.mode, d.exact = cfg["mode"], cfg["exact"]
    d.dtype = _DT[q.dtype]
    d.only_glo = int(cfg["only_glo"])
    d.backend = _BACKENDS[backend]
    d.scale = float(cfg["scale"])
    d.bias_side = int(cfg.get("bias_side", 0))
    d.mode_dev = cfg.get("mode_dev") or None      # device int32 holding the random-shift neighbour (graph replay)
    d.q_sb, d.q_st, d.q_sh = _strides(q, M)
    d.k_sb, d.k_st, d.k_sh = _strides(k, M)
    d.v_sb, d.v_st, d.v_sh = _strides(v, M)
    d.o_sb, d.o_st, d.o_sh = _strides(out, M)
    return d


_WS = {}


def _workspace(d, pass_, device, query=None):
    """Scratch of one library call.  The library uses it only between the call's own launches, and launches on one
    stream are ordered, so ONE buffer per (device, stream) is reused by every call on that stream (grown on demand)
    instead of a torch.empty per call (host cost in the eager step).  During stream capture a fresh allocation is
    taken from the graph's private pool, as before."""
    n = max(int((query or _lib.lib().vil_attn_workspace_bytes)(ctypes.byref(d), pass_)), 4) // 4 + 1
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(n, dtype=torch.float32, device=device)
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _WS.get(key)
    if ws is None or ws.numel() < n:
        ws = _WS[key] = torch.empty(int(n * 1.25) + 1024, dtype=torch.float32, device=device)
    return ws


class _VilLocalAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, kv, table, g2l, cfg, backend):
        if not q.is_cuda:
            raise RuntimeError("vil_local_attention needs device tensors: the product path is the HIP "
                               "kernels (libvilattn.so); there is no CPU fallback")
        if q.dtype not in _DT or kv.dtype != q.dtype:
            raise TypeError(f"vil_local_attention supports float32 / bfloat16 / float16 q,kv of one dtype; got {q.dtype}, {kv.dtype}")
        L = _lib.lib()
        q = _last_contig(q)
        kv = _last_contig(kv)
        B, Nloc, C = q.shape
        assert kv.shape[0] == B and kv.shape[2] == 2 * C and kv.shape[1] == cfg["G"] + Nloc
        assert Nloc == cfg["nx"] * cfg["ny"]
        k, v = kv[..., :C], kv[..., C:]
        out = torch.empty(B, Nloc, C, dtype=q.dtype, device=q.device)
        lse = torch.empty(B, cfg["H"], Nloc, dtype=torch.float32, device=q.device)
        tab = table.detach().float().contiguous() if table is not None else None
        g2 = g2l.detach().float().contiguous() if g2l is not None else None
        d = _make_desc(q, k, v, out, cfg, backend)
        ws = _workspace(d, 0, q.device)
        stream = ctypes.c_void_p(torch.cuda.current_stream(q.device).cuda_stream)
        with torch.cuda.device(q.device):
            _lib.check(L.vil_attn_fwd(ctypes.byref(d), _ptr(q), _ptr(k), _ptr(v), _ptr(tab), _ptr(g2),
                                      _ptr(out), _ptr(lse), _ptr(ws), stream))
        ctx.save_for_backward(q, kv, out, lse, tab, g2)
        ctx.cfg, ctx.backend = cfg, backend
        ctx.table_dtype = table.dtype if table is not None else None
        ctx.g2l_dtype = g2l.dtype if g2l is not None else None
        return out

    @staticmethod
    def backward(ctx, dout):
        q, kv, out, lse, tab, g2 = ctx.saved_tensors
        cfg = ctx.cfg
        L = _lib.lib()
        B, Nloc, C = q.shape
        dout = _last_contig(dout)
        if dout.dtype != q.dtype:
            dout = dout.to(q.dtype)
        k, v = kv[..., :C], kv[..., C:]
        dq = torch.empty(B, Nloc, C, dtype=q.dtype, device=q.device)
        dkv = torch.empty(B, kv.shape[1], 2 * C, dtype=q.dtype, device=q.device)
        dk, dv = dkv[..., :C], dkv[..., C:]
        dtab = torch.empty_like(tab) if tab is not None else None
        dg2 = torch.empty_like(g2) if g2 is not None else None
        M = C // cfg["H"]
        d = _make_desc(q, k, v, out, cfg, ctx.backend)
        d.do_sb, d.do_st, d.do_sh = _strides(dout, M)
        d.dq_sb, d.dq_st, d.dq_sh = _strides(dq, M)
        d.dk_sb, d.dk_st, d.dk_sh = _strides(dk, M)
        d.dv_sb, d.dv_st, d.dv_sh = _strides(dv, M)
        ws = _workspace(d, 1, q.device)
        stream = ctypes.c_void_p(torch.cuda.current_stream(q.device).cuda_stream)
        with torch.cuda.device(q.device):
            _lib.check(L.vil_attn_bwd(ctypes.byref(d), _ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(dout),
                                      _ptr(lse), _ptr(tab), _ptr(g2), _ptr(dq), _ptr(dk), _ptr(dv),
                                      _ptr(dtab), _ptr(dg2), _ptr(ws), stream))
        if dtab is not None and ctx.table_dtype != dtab.dtype:
            dtab = dtab.to(ctx.table_dtype)
        if dg2 is not None and ctx.g2l_dtype != dg2.dtype:
            dg2 = dg2.to(ctx.g2l_dtype)
        return dq, dkv, dtab, dg2, None, None


def vil_local_attention(q, kv, bias_table, g2l_bias, *, nx, ny, w, nglo, num_heads, mode=0, exact=0,
                        scale=None, only_glo=False, backend=None, _debug=0):
    """Local-query rows of Vision Longformer attention.

    q:  (B, nx*ny, C)       unscaled local queries (output of the `query` Linear)
    kv: (B, nglo+nx*ny, 2C) output of the `kv` Linear: [..., :C] keys, [..., C:] values,
                            global tokens first
    bias_table: ((4w-1)^2, H) or None (rpe off); g2l_bias: (H, nglo) or None.
    Returns (B, nx*ny, C), heads concatenated, ready for the `proj` Linear."""
    if exact not in (0, 1, -1) or (exact == 1 and mode != 0 and not only_glo):
        # the reference raises from mask_invalid_locations (slidingchunk_2d.py:331-343)
        raise ValueError("longsc exact should be in [0,1,-1]!")
    C = q.shape[-1]
    cfg = dict(nx=int(nx), ny=int(ny), W=int(w), G=int(nglo), H=int(num_heads), mode=int(mode),
               exact=int(exact), only_glo=bool(only_glo),
               scale=float(scale) if scale is not None else (C // num_heads) ** -0.5, debug=int(_debug))
    if nglo == 0:
        g2l_bias = None
    return _VilLocalAttention.apply(q, kv, bias_table, g2l_bias, cfg, backend or DEFAULT_BACKEND)


def _full_fwd(q_all, k, v, tab, g2l_f, g2g_f, cfg, backend):
    """local rows + (G > 0) global rows; q_all/k/v are (B, N, C) views (last dim contiguous)."""
    L = _lib.lib()
    B, N, C = q_all.shape
    G, H = cfg["G"], cfg["H"]
    Nloc = N - G
    out_all = torch.empty(B, N, C, dtype=q_all.dtype, device=q_all.device)
    lse = torch.empty(B, H, Nloc, dtype=torch.float32, device=q_all.device)
    lse_g = torch.empty(B, H, max(G, 1), dtype=torch.float32, device=q_all.device)
    q_loc, out_loc = q_all[:, G:], out_all[:, G:]
    d = _make_desc(q_loc, k, v, out_loc, cfg, backend)
    ws = _workspace(d, 0, q_all.device)
    stream = ctypes.c_void_p(torch.cuda.current_stream(q_all.device).cuda_stream)
    with torch.cuda.device(q_all.device):
        if G == 1 and backend != "scalar" and hasattr(L, "vil_attn_fwd_full"):
            # one pass over K / V: the global token's row rides in the forward kernel (round 5)
            rc = L.vil_attn_fwd_full(ctypes.byref(d), _ptr(q_all), _ptr(k), _ptr(v), _ptr(tab), _ptr(g2l_f), _ptr(g2g_f),
                                     _ptr(out_all), _ptr(lse), _ptr(lse_g), _ptr(ws), stream)
            if rc == 0:
                return out_all, lse, lse_g
            if rc != _lib.VIL_E_BACKEND:
                _lib.check(rc)
        _lib.check(L.vil_attn_fwd(ctypes.byref(d), _ptr(q_loc), _ptr(k), _ptr(v), _ptr(tab),
                                  _ptr(g2l_f[1]) if g2l_f is not None else None,
                                  _ptr(out_loc), _ptr(lse), _ptr(ws), stream))
        if G > 0:
            _lib.check(L.vil_glo_attn_fwd(ctypes.byref(d), _ptr(q_all), _ptr(k), _ptr(v), _ptr(g2g_f),
                                          _ptr(g2l_f[0]) if g2l_f is not None else None,
                                          _ptr(out_all), _ptr(lse_g), stream))
    return out_all, lse, lse_g


def _full_bwd(q_all, k, v, out_all, dout_all, lse, lse_g, tab, g2l_f, g2g_f, dq_all, dk, dv, cfg, backend):
    L = _lib.lib()
    B, N, C = q_all.shape
    G, H = cfg["G"], cfg["H"]
    M = C // H
    dtab = torch.empty_like(tab) if tab is not None else None
    # (vil_attn_bwd_full zeroes dg2l / dg2g in its own prologue launch; the two-call fallback below needs them zero)
    dg2l = torch.empty_like(g2l_f) if g2l_f is not None else None
    dg2g = torch.empty_like(g2g_f) if g2g_f is not None else None
    q_loc, out_loc, do_loc, dq_loc = q_all[:, G:], out_all[:, G:], dout_all[:, G:], dq_all[:, G:]
    d = _make_desc(q_loc, k, v, out_loc, cfg, backend)
    d.do_sb, d.do_st, d.do_sh = _strides(do_loc, M)
    d.dq_sb, d.dq_st, d.dq_sh = _strides(dq_loc, M)
    d.dk_sb, d.dk_st, d.dk_sh = _strides(dk, M)
    d.dv_sb, d.dv_st, d.dv_sh = _strides(dv, M)
    ws = _workspace(d, 1, q_all.device)
    stream = ctypes.c_void_p(torch.cuda.current_stream(q_all.device).cuda_stream)
    with torch.cuda.device(q_all.device):
        if G > 0 and backend != "scalar":
            # one call: the global rows' backward rides in the dK/dV pass (MFMA family)
            rc = L.vil_attn_bwd_full(ctypes.byref(d), _ptr(q_all), _ptr(k), _ptr(v), _ptr(out_all), _ptr(dout_all),
                                     _ptr(lse), _ptr(lse_g), _ptr(tab), _ptr(g2l_f), _ptr(g2g_f),
                                     _ptr(dq_all), _ptr(dk), _ptr(dv), _ptr(dtab), _ptr(dg2l), _ptr(dg2g), _ptr(ws), stream)
            if rc == 0:
                return dtab, dg2l, dg2g
            if rc != _lib.VIL_E_BACKEND:
                _lib.check(rc)
        for t in (dg2l, dg2g):
            if t is not None:
                t.zero_()
        _lib.check(L.vil_attn_bwd(ctypes.byref(d), _ptr(q_loc), _ptr(k), _ptr(v), _ptr(out_loc), _ptr(do_loc),
                                  _ptr(lse), _ptr(tab), _ptr(g2l_f[1]) if g2l_f is not None else None,
                                  _ptr(dq_loc), _ptr(dk), _ptr(dv), _ptr(dtab),
                                  _ptr(dg2l[1]) if dg2l is not None else None, _ptr(ws), stream))
        if G > 0:
            _lib.check(L.vil_glo_attn_bwd(ctypes.byref(d), _ptr(q_all), _ptr(k), _ptr(v), _ptr(out_all), _ptr(dout_all),
                                          _ptr(lse_g), _ptr(g2g_f), _ptr(g2l_f[0]) if g2l_f is not None else None,
                                          _ptr(dq_all), _ptr(dk), _ptr(dv), _ptr(dg2g),
                                          _ptr(dg2l[0]) if dg2l is not None else None, stream))
    return dtab, dg2l, dg2g


def _check_dev(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name} needs device tensors: the product path is the HIP kernels "
                           "(libvilattn.so); there is no CPU fallback")
    if t.dtype not in _DT:
        raise TypeError(f"{name} supports float32 / bfloat16 / float16 tensors; got {t.dtype}")


def _f32c(t):
    return t.detach().float().contiguous() if t is not None else None


class _VilFullAttention(torch.autograd.Function):
    """Local rows (vil_attn_fwd/_bwd) AND the G global-token query rows (vil_glo_attn_fwd/_bwd) of one
    Long2DSCSelfAttention layer on shared q / kv tensors: q_all (B, G+Nloc, C) -> out_all (B, G+Nloc, C).
    The global rows' dk/dv contribution is accumulated in place after the local pass."""

    @staticmethod
    def forward(ctx, q_all, kv, table, g2l, g2g, cfg, backend):
        _check_dev(q_all, "vil_full_attention")
        q_all, kv = _last_contig(q_all), _last_contig(kv)
        B, N, C = q_all.shape
        assert kv.shape == (B, N, 2 * C) and kv.dtype == q_all.dtype and N - cfg["G"] == cfg["nx"] * cfg["ny"]
        tab, g2l_f, g2g_f = _f32c(table), _f32c(g2l), _f32c(g2g)
        out_all, lse, lse_g = _full_fwd(q_all, kv[..., :C], kv[..., C:], tab, g2l_f, g2g_f, cfg, backend)
        ctx.save_for_backward(q_all, kv, out_all, lse, lse_g, tab, g2l_f, g2g_f)
        ctx.cfg, ctx.backend = cfg, backend
        ctx.dts = tuple(t.dtype if t is not None else None for t in (table, g2l, g2g))
        return out_all

    @staticmethod
    def backward(ctx, dout_all):
        q_all, kv, out_all, lse, lse_g, tab, g2l_f, g2g_f = ctx.saved_tensors
        B, N, C = q_all.shape
        dout_all = _last_contig(dout_all).to(q_all.dtype)
        dq_all = torch.empty(B, N, C, dtype=q_all.dtype, device=q_all.device)
        dkv = torch.empty(B, N, 2 * C, dtype=q_all.dtype, device=q_all.device)
        dtab, dg2l, dg2g = _full_bwd(q_all, kv[..., :C], kv[..., C:], out_all, dout_all, lse, lse_g, tab, g2l_f, g2g_f,
                                     dq_all, dkv[..., :C], dkv[..., C:], ctx.cfg, ctx.backend)
        tdt, ldt, gdt = ctx.dts
        return (dq_all, dkv, dtab.to(tdt) if dtab is not None else None,
                dg2l.to(ldt) if dg2l is not None else None, dg2g.to(gdt) if dg2g is not None else None, None, None)


class _VilQKVAttention(torch.autograd.Function):
    """Same op on a packed (B, N, 3C) projection [q | k | v] (the dense `Attention` of the s0 stages,
    reference msvit.py:91-120): the gradient comes back as ONE (B, N, 3C) tensor for the qkv GEMM."""

    @staticmethod
    def forward(ctx, qkv, table, g2l, g2g, cfg, backend):
        _check_dev(qkv, "vil_qkv_attention")
        qkv = _last_contig(qkv)
        B, N, C3 = qkv.shape
        C = C3 // 3
        assert N - cfg["G"] == cfg["nx"] * cfg["ny"]
        tab, g2l_f, g2g_f = _f32c(table), _f32c(g2l), _f32c(g2g)
        out_all, lse, lse_g = _full_fwd(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], tab, g2l_f, g2g_f, cfg, backend)
        ctx.save_for_backward(qkv, out_all, lse, lse_g, tab, g2l_f, g2g_f)
        ctx.cfg, ctx.backend = cfg, backend
        ctx.dts = tuple(t.dtype if t is not None else None for t in (table, g2l, g2g))
        return out_all

    @staticmethod
    def backward(ctx, dout_all):
        qkv, out_all, lse, lse_g, tab, g2l_f, g2g_f = ctx.saved_tensors
        B, N, C3 = qkv.shape
        C = C3 // 3
        dout_all = _last_contig(dout_all).to(qkv.dtype)
        dqkv = torch.empty_like(qkv)
        dtab, dg2l, dg2g = _full_bwd(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], out_all, dout_all, lse, lse_g,
                                     tab, g2l_f, g2g_f, dqkv[..., :C], dqkv[..., C:2 * C], dqkv[..., 2 * C:],
                                     ctx.cfg, ctx.backend)
        tdt, ldt, gdt = ctx.dts
        return (dqkv, dtab.to(tdt) if dtab is not None else None, dg2l.to(ldt) if dg2l is not None else None,
                dg2g.to(gdt) if dg2g is not None else None, None, None)


FULL_MAX_G = 4      # vil_glo_attn_* bookkeeping limit


class _VilGlobalRows(torch.autograd.Function):
    """The G global-token QUERY rows alone (reference longformer2d.py:210-227) for layers whose global rows do
    not share the local rows' projections (`sharew=False`), `only_glo` layers, or any layer outside the fused
    whole-layer op: q_g (B, G, C) from `query_global`, kv (B, N, 2C) from `kv_global` -> (B, G, C).
    vil_glo_attn_fwd / vil_glo_attn_bwd; the backward accumulates into a zero-initialised dkv."""

    @staticmethod
    def forward(ctx, q_g, kv, g2g, g2l0, cfg):
        _check_dev(q_g, "vil_global_attention")
        q_g, kv = _last_contig(q_g), _last_contig(kv)
        L = _lib.lib()
        B, G, C = q_g.shape
        H = cfg["H"]
        M = C // H
        assert kv.dtype == q_g.dtype and kv.shape[0] == B and kv.shape[2] == 2 * C and G == cfg["G"]
        k, v = kv[..., :C], kv[..., C:]
        out = torch.empty_like(q_g)
        lse = torch.empty(B, H, G, dtype=torch.float32, device=q_g.device)
        g2g_f, g2l_f = _f32c(g2g), _f32c(g2l0)
        d = _make_desc(q_g, k, v, out, cfg, "auto")
        stream = ctypes.c_void_p(torch.cuda.current_stream(q_g.device).cuda_stream)
        with torch.cuda.device(q_g.device):
            _lib.check(L.vil_glo_attn_fwd(ctypes.byref(d), _ptr(q_g), _ptr(k), _ptr(v), _ptr(g2g_f), _ptr(g2l_f),
                                          _ptr(out), _ptr(lse), stream))
        ctx.save_for_backward(q_g, kv, out, lse, g2g_f, g2l_f)
        ctx.cfg = cfg
        ctx.dts = tuple(t.dtype if t is not None else None for t in (g2g, g2l0))
        return out

    @staticmethod
    def backward(ctx, dout):
        q_g, kv, out, lse, g2g_f, g2l_f = ctx.saved_tensors
        cfg = ctx.cfg
        L = _lib.lib()
        B, G, C = q_g.shape
        M = C // cfg["H"]
        dout = _last_contig(dout).to(q_g.dtype)
        dq = torch.empty_like(q_g)
        dkv = torch.zeros_like(kv)
        k, v, dk, dv = kv[..., :C], kv[..., C:], dkv[..., :C], dkv[..., C:]
        dg2g = torch.zeros_like(g2g_f) if g2g_f is not None else None
        dg2l = torch.zeros_like(g2l_f) if g2l_f is not None else None
        d = _make_desc(q_g, k, v, out, cfg, "auto")
        d.do_sb, d.do_st, d.do_sh = _strides(dout, M)
        d.dq_sb, d.dq_st, d.dq_sh = _strides(dq, M)
        d.dk_sb, d.dk_st, d.dk_sh = _strides(dk, M)
        d.dv_sb, d.dv_st, d.dv_sh = _strides(dv, M)
        stream = ctypes.c_void_p(torch.cuda.current_stream(q_g.device).cuda_stream)
        with torch.cuda.device(q_g.device):
            _lib.check(L.vil_glo_attn_bwd(ctypes.byref(d), _ptr(q_g), _ptr(k), _ptr(v), _ptr(out), _ptr(dout), _ptr(lse),
                                          _ptr(g2g_f), _ptr(g2l_f), _ptr(dq), _ptr(dk), _ptr(dv), _ptr(dg2g), _ptr(dg2l),
                                          stream))
        gdt, ldt = ctx.dts
        return (dq, dkv, dg2g.to(gdt) if dg2g is not None else None, dg2l.to(ldt) if dg2l is not None else None, None)


def vil_global_attention(q_g, kv, g2g_bias, g2l0_bias, *, nx, ny, nglo, num_heads, scale=None):
    """Global-token query rows only: q_g (B, nglo, C) UNSCALED output of `query_global`, kv (B, nglo+nx*ny, 2C)
    output of `kv_global`; g2g_bias (H, nglo, nglo), g2l0_bias (H, nglo) = g2l_relative_position_bias[0], or None.
    Returns (B, nglo, C)."""
    assert 1 <= nglo <= FULL_MAX_G
    cfg = _cfg(q_g.shape[-1], nx, ny, 1, nglo, num_heads, 0, 0, scale)
    return _VilGlobalRows.apply(q_g, kv, g2g_bias, g2l0_bias, cfg)



def _cfg(q_last_dim, nx, ny, w, nglo, num_heads, mode, exact, scale, bias_side=0):
    return dict(nx=int(nx), ny=int(ny), W=int(w), G=int(nglo), H=int(num_heads), mode=int(mode), exact=int(exact),
                only_glo=False, scale=float(scale) if scale is not None else (q_last_dim // num_heads) ** -0.5,
                debug=0, bias_side=int(bias_side))


def vil_full_attention(q_all, kv, bias_table, g2l_bias, g2g_bias, *, nx, ny, w, nglo, num_heads, mode=0, exact=0,
                       scale=None, backend=None, mode_dev=None):
    """All rows of one layer: q_all (B, nglo+nx*ny, C) from ONE query projection, kv (B, N, 2C);
    g2l_bias (2, H, nglo) and g2g_bias (H, nglo, nglo) or None.  Returns (B, N, C)."""
    if exact not in (0, 1, -1) or (exact == 1 and mode != 0):
        raise ValueError("longsc exact should be in [0,1,-1]!")
    assert 1 <= nglo <= FULL_MAX_G
    cfg = _cfg(q_all.shape[-1], nx, ny, w, nglo, num_heads, mode, exact, scale)
    if mode_dev is not None:            # (1,) int32 device tensor, kept alive by the caller (the module's buffer)
        assert mode > 0 and mode_dev.dtype == torch.int32 and mode_dev.is_cuda
        cfg["mode_dev"] = mode_dev.data_ptr()
    return _VilFullAttention.apply(q_all, kv, bias_table, g2l_bias, g2g_bias, cfg, backend or DEFAULT_BACKEND)


def vil_full_attention_qkv(qkv, bias_table, g2l_bias, g2g_bias, *, nx, ny, w, nglo, num_heads, mode=0, exact=0,
                           scale=None, backend=None, mode_dev=None):
    """vil_full_attention on ONE packed (B, N, 3C) projection [q | k | v] (the layer's query and kv weights applied as a
    single GEMM): q, k, v are strided views for the kernels and the gradient comes back as one (B, N, 3C) tensor, so the
    input gradient of the projection is one GEMM instead of two plus an accumulation pass."""
    if exact not in (0, 1, -1) or (exact == 1 and mode != 0):
        raise ValueError("longsc exact should be in [0,1,-1]!")
    assert 1 <= nglo <= FULL_MAX_G
    cfg = _cfg(qkv.shape[-1] // 3, nx, ny, w, nglo, num_heads, mode, exact, scale)
    if mode_dev is not None:
        assert mode > 0 and mode_dev.dtype == torch.int32 and mode_dev.is_cuda
        cfg["mode_dev"] = mode_dev.data_ptr()
    return _VilQKVAttention.apply(qkv, bias_table, g2l_bias, g2g_bias, cfg, backend or DEFAULT_BACKEND)


class _VilDenseAttention(torch.autograd.Function):
    """The dense `Attention` of the s0 stages on its own kernel family (vil_dense_attn_fwd / _bwd, csrc/vil_attn_dense.hip):
    packed (B, N, 3C) projection in, (B, N, C) out, the gradient back as ONE (B, N, 3C) tensor; the global tokens are
    ordinary rows / columns of the same kernels.  1 forward launch, 3 backward launches."""

    @staticmethod
    def forward(ctx, qkv, table, g2l, g2g, cfg):
        _check_dev(qkv, "vil_dense_attention")
        qkv = _last_contig(qkv)
        B, N, C3 = qkv.shape
        C = C3 // 3
        H = cfg["H"]
        q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
        tab, g2l_f, g2g_f = _f32c(table), _f32c(g2l), _f32c(g2g)
        out = torch.empty(B, N, C, dtype=qkv.dtype, device=qkv.device)
        lse = torch.empty(B, H, N + 1, dtype=torch.float32, device=qkv.device)     # [..., N]: max |v_k|^2 (see include/vil_attn.h)
        d = _make_desc(q, k, v, out, cfg, "auto")
        stream = ctypes.c_void_p(torch.cuda.current_stream(qkv.device).cuda_stream)
        with torch.cuda.device(qkv.device):
            _lib.check(_lib.lib().vil_dense_attn_fwd(ctypes.byref(d), _ptr(q), _ptr(k), _ptr(v), _ptr(tab), _ptr(g2l_f),
                                                     _ptr(g2g_f), _ptr(out), _ptr(lse), stream))
        ctx.save_for_backward(qkv, out, lse, tab, g2l_f, g2g_f)
        ctx.cfg = cfg
        ctx.dts = tuple(t.dtype if t is not None else None for t in (table, g2l, g2g))
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse, tab, g2l_f, g2g_f = ctx.saved_tensors
        cfg = ctx.cfg
        B, N, C3 = qkv.shape
        C = C3 // 3
        M = C // cfg["H"]
        L = _lib.lib()
        dout = _last_contig(dout).to(qkv.dtype)
        dqkv = torch.empty_like(qkv)
        q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
        dq, dk, dv = dqkv[..., :C], dqkv[..., C:2 * C], dqkv[..., 2 * C:]
        dtab = torch.empty_like(tab) if tab is not None else None
        dg2l = torch.empty_like(g2l_f) if g2l_f is not None else None
        dg2g = torch.empty_like(g2g_f) if g2g_f is not None else None
        d = _make_desc(q, k, v, out, cfg, "auto")
        d.do_sb, d.do_st, d.do_sh = _strides(dout, M)
        d.dq_sb, d.dq_st, d.dq_sh = _strides(dq, M)
        d.dk_sb, d.dk_st, d.dk_sh = _strides(dk, M)
        d.dv_sb, d.dv_st, d.dv_sh = _strides(dv, M)
        ws = _workspace(d, 1, qkv.device, L.vil_dense_attn_workspace_bytes)
        stream = ctypes.c_void_p(torch.cuda.current_stream(qkv.device).cuda_stream)
        with torch.cuda.device(qkv.device):
            _lib.check(L.vil_dense_attn_bwd(ctypes.byref(d), _ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(dout), _ptr(lse),
                                            _ptr(tab), _ptr(g2l_f), _ptr(g2g_f), _ptr(dq), _ptr(dk), _ptr(dv),
                                            _ptr(dtab), _ptr(dg2l), _ptr(dg2g), _ptr(ws), stream))
        tdt, ldt, gdt = ctx.dts
        return (dqkv, dtab.to(tdt) if dtab is not None else None, dg2l.to(ldt) if dg2l is not None else None,
                dg2g.to(gdt) if dg2g is not None else None, None)


def dense_family_supported(qkv, nx, ny, nglo, num_heads):
    """True when the dedicated dense kernels (csrc/vil_attn_dense.hip) take this problem: bf16 / fp16, head_dim 64,
    nglo <= 4, K and V of one (image, head) within the LDS."""
    if not qkv.is_cuda or qkv.dtype not in (torch.bfloat16, torch.float16) or qkv.stride(-1) != 1:
        return False
    C = qkv.shape[-1] // 3
    cfg = _cfg(C, nx, ny, max(int(nx), int(ny)), nglo, num_heads, -1, 0, None)
    d = _make_desc(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], qkv[..., :C], cfg, "auto")
    d.o_sb, d.o_st = qkv.shape[1] * C, C                 # the (B, N, C) output the call allocates
    return _lib.lib().vil_dense_attn_supported(ctypes.byref(d)) == 0


def vil_dense_attention(qkv, bias_table, g2l_bias, g2g_bias, *, nx, ny, nglo, num_heads, scale=None, backend=None):
    """Dense attention over an (nglo + nx*ny)-token sequence with the Swin-style relative position bias
    of the `s0` stages (reference msvit.py:37-120).  qkv: (B, N, 3C) packed projection; bias_table
    ((2nx-1)*(2ny-1), H) or None.  Returns (B, N, C).

    backend None / "auto" / "dense": the dedicated dense kernel family (csrc/vil_attn_dense.hip) when it takes the
    problem (bf16 / fp16, head_dim 64, nglo <= 4, sequence within the LDS); otherwise -- and with backend "mfma" /
    "scalar" -- the ONE-CHUNK case of the sliding-chunk kernels (chunk side w = max(nx, ny), mode -1, bias table side
    2w-1; square grids)."""
    backend = backend or DEFAULT_BACKEND
    if nglo == 0:
        g2l_bias = g2g_bias = None
    if backend in ("auto", "dense"):
        ok = (bias_table is None or bias_table.shape[0] == (2 * int(nx) - 1) * (2 * int(ny) - 1)) and \
            dense_family_supported(qkv, nx, ny, nglo, num_heads)
        if ok:
            cfg = _cfg(qkv.shape[-1] // 3, nx, ny, max(int(nx), int(ny)), nglo, num_heads, -1, 0, scale)
            return _VilDenseAttention.apply(qkv, bias_table, g2l_bias, g2g_bias, cfg)
        if backend == "dense":
            raise RuntimeError("vil_dense_attention: the dense kernel family does not take this problem")
        backend = "auto"
    w = max(int(nx), int(ny))
    assert 0 <= nglo <= FULL_MAX_G
    side = 2 * w - 1 if bias_table is not None else 0
    if bias_table is not None:
        assert bias_table.shape[0] == side * side, "dense bias table must be ((2w-1)^2, H) with w = max(nx, ny)"
    cfg = _cfg(qkv.shape[-1] // 3, nx, ny, w, nglo, num_heads, -1, 0, scale, bias_side=side)
    return _VilQKVAttention.apply(qkv, bias_table, g2l_bias, g2g_bias, cfg, backend)
