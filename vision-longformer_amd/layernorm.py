"""Fused LayerNorm (HIP, libvilattn.so: vil_layernorm_fwd/_bwd) for the blocks around the hot path.

`VilLayerNorm` is an `nn.LayerNorm` (same parameters / state-dict keys).  On device tensors it runs one
fused kernel per direction; under bf16 autocast with `cast_output=True` it writes its output directly
in bf16 -- numerically identical to PyTorch's fp32 LayerNorm followed by autocast's cast for the next
Linear, without the cast kernels and with half the write traffic.  CPU tensors fall through to
`nn.LayerNorm.forward` (this layer is glue, not the hot path; the CPU baseline model uses it that way)."""
import ctypes

import torch
from torch import nn

from . import _lib

_DT = {torch.float32: _lib.DTYPE_F32, torch.bfloat16: _lib.DTYPE_BF16}


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


class _FusedLN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps, out_dtype):
        L = _lib.lib()
        C = x.shape[-1]
        x2 = x.reshape(-1, C)
        if x2.stride(-1) != 1:
            x2 = x2.contiguous()
        rows = x2.shape[0]
        w = weight.detach().float().contiguous()
        b = bias.detach().float().contiguous()
        y = torch.empty(rows, C, dtype=out_dtype, device=x.device)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        stream = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        with torch.cuda.device(x.device):
            _lib.check(L.vil_layernorm_fwd(_p(x2), _DT[x2.dtype], _p(w), _p(b), _p(y), _DT[out_dtype], _p(mean), _p(rstd),
                                           rows, C, x2.stride(0), y.stride(0), float(eps), stream))
        ctx.save_for_backward(x2, w, mean, rstd)
        ctx.x_shape = x.shape
        ctx.wdtype = weight.dtype
        return y.view(*x.shape[:-1], C)

    @staticmethod
    def backward(ctx, dy):
        x2, w, mean, rstd = ctx.saved_tensors
        L = _lib.lib()
        rows, C = x2.shape
        dy2 = dy.reshape(rows, C)
        if dy2.stride(-1) != 1:
            dy2 = dy2.contiguous()
        if dy2.dtype not in _DT:
            dy2 = dy2.float()
        dx = torch.empty(rows, C, dtype=x2.dtype, device=x2.device)
        dgamma = torch.empty(C, dtype=torch.float32, device=x2.device)
        dbeta = torch.empty(C, dtype=torch.float32, device=x2.device)
        ws = torch.empty(L.vil_layernorm_workspace_bytes(rows, C) // 4, dtype=torch.float32, device=x2.device)
        stream = ctypes.c_void_p(torch.cuda.current_stream(x2.device).cuda_stream)
        with torch.cuda.device(x2.device):
            _lib.check(L.vil_layernorm_bwd(_p(dy2), _DT[dy2.dtype], _p(x2), _DT[x2.dtype], _p(w), _p(mean), _p(rstd),
                                           _p(dx), _DT[dx.dtype], _p(dgamma), _p(dbeta), _p(ws), rows, C,
                                           dy2.stride(0), x2.stride(0), dx.stride(0), stream))
        return dx.view(ctx.x_shape), dgamma.to(ctx.wdtype), dbeta.to(ctx.wdtype), None, None


class _TokensLN(torch.autograd.Function):
    """out[:, :G] = cls (broadcast over the batch);  out[:, G:] = LayerNorm(x)   for x (B, N, C), cls (1, G, C):
    `torch.cat((cls_tokens, norm(x)), dim=1)` of the reference's PatchEmbed (msvit.py:204-206) with the LayerNorm
    writing straight into the token tensor (vil_layernorm_fwd_tokens) and its backward reading straight out of the
    token tensor's gradient (vil_layernorm_bwd_tokens): no concatenation copy in either direction."""

    @staticmethod
    def forward(ctx, x, cls, weight, bias, eps, out_dtype):
        L = _lib.lib()
        B, N, C = x.shape
        G = cls.shape[1]
        x2 = x.reshape(-1, C)
        if x2.stride(-1) != 1:
            x2 = x2.contiguous()
        rows = x2.shape[0]
        w = weight.detach().float().contiguous()
        b = bias.detach().float().contiguous()
        out = torch.empty(B, G + N, C, dtype=out_dtype, device=x.device)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        stream = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        with torch.cuda.device(x.device):
            _lib.check(L.vil_layernorm_fwd_tokens(_p(x2), _DT[x2.dtype], _p(w), _p(b), _p(out), _DT[out_dtype], _p(mean),
                                                  _p(rstd), rows, C, x2.stride(0), float(eps), N, G, stream))
        out[:, :G] = cls.to(out_dtype)
        ctx.save_for_backward(x2, w, mean, rstd)
        ctx.x_shape, ctx.G, ctx.wdtype, ctx.cdtype, ctx.cshape = x.shape, G, weight.dtype, cls.dtype, cls.shape
        return out

    @staticmethod
    def backward(ctx, g):
        from .linear import _colsum
        x2, w, mean, rstd = ctx.saved_tensors
        L = _lib.lib()
        rows, C = x2.shape
        B, N, _ = ctx.x_shape
        G = ctx.G
        if not g.is_contiguous():
            g = g.contiguous()
        if g.dtype not in _DT:
            g = g.float()
        dx = torch.empty(rows, C, dtype=x2.dtype, device=x2.device)
        dgamma = torch.empty(C, dtype=torch.float32, device=x2.device)
        dbeta = torch.empty(C, dtype=torch.float32, device=x2.device)
        ws = torch.empty(L.vil_layernorm_workspace_bytes(rows, C) // 4, dtype=torch.float32, device=x2.device)
        stream = ctypes.c_void_p(torch.cuda.current_stream(x2.device).cuda_stream)
        with torch.cuda.device(x2.device):
            _lib.check(L.vil_layernorm_bwd_tokens(_p(g), _DT[g.dtype], _p(x2), _DT[x2.dtype], _p(w), _p(mean), _p(rstd),
                                                  _p(dx), _DT[dx.dtype], _p(dgamma), _p(dbeta), _p(ws), rows, C,
                                                  x2.stride(0), dx.stride(0), N, G, stream))
        dcls = None
        if ctx.needs_input_grad[1]:
            # the batch sum of the global-token rows: a column sum of the (B, G*C) leading block of every sample
            dcls = _colsum(g.view(B, (G + N) * C)[:, :G * C]).view(ctx.cshape).to(ctx.cdtype)
        return dx.view(ctx.x_shape), dcls, dgamma.to(ctx.wdtype), dbeta.to(ctx.wdtype), None, None


def tokens_layernorm_ok(x, cls, norm):
    C = x.shape[-1]
    return (isinstance(norm, VilLayerNorm) and x.is_cuda and x.dim() == 3 and x.dtype in _DT and cls is not None
            and cls.shape[1] >= 1 and C % 8 == 0 and C <= 1024 and norm.elementwise_affine and norm.bias is not None)


def tokens_layernorm(x, cls, norm):
    """cat((cls.expand(B), norm(x)), dim=1) in one LayerNorm kernel each way; output dtype as VilLayerNorm.forward."""
    out_dtype = x.dtype
    if torch.is_autocast_enabled("cuda"):
        ac = torch.get_autocast_dtype("cuda")
        out_dtype = ac if (norm.cast_output and ac in _DT) else torch.float32
    return _TokensLN.apply(x, cls, norm.weight, norm.bias, norm.eps, out_dtype)


class _ResLN(torch.autograd.Function):
    """x_new = x + rscale[b] * branch;  y = LayerNorm(x_new)   (vil_resln_fwd / _bwd).
    One kernel each way for the residual add of one block and the norm of the next; the backward also
    emits the branch gradient (stochastic-depth scale and cast included), so the add / mul / cast kernels
    of the unfused graph disappear."""

    @staticmethod
    def forward(ctx, x, branch, rscale, weight, bias, eps, out_dtype):
        L = _lib.lib()
        C = x.shape[-1]
        x2 = x.reshape(-1, C)
        br2 = branch.reshape(-1, C)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        if not br2.is_contiguous():
            br2 = br2.contiguous()
        rows = x2.shape[0]
        rps = rows // x.shape[0]
        w = weight.detach().float().contiguous()
        b = bias.detach().float().contiguous()
        xn = torch.empty(rows, C, dtype=torch.float32, device=x.device)
        y = torch.empty(rows, C, dtype=out_dtype, device=x.device)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        stream = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        _lib.check(L.vil_resln_fwd(_p(x2), _p(br2), _DT[br2.dtype], _p(rscale) if rscale is not None else None, rps,
                                   _p(w), _p(b), _p(xn), _p(y), _DT[out_dtype], _p(mean), _p(rstd), rows, C,
                                   float(eps), stream))
        ctx.save_for_backward(xn, w, mean, rstd, rscale)
        ctx.shape, ctx.wdtype, ctx.bdtype, ctx.rps = x.shape, weight.dtype, branch.dtype, rps
        return xn.view(x.shape), y.view(x.shape)

    @staticmethod
    def backward(ctx, g_x, g_y):
        xn, w, mean, rstd, rscale = ctx.saved_tensors
        L = _lib.lib()
        rows, C = xn.shape
        if g_y is None:
            g_y = torch.zeros(rows, C, dtype=torch.bfloat16, device=xn.device)
        dy2 = g_y.reshape(rows, C)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        if dy2.dtype not in _DT:
            dy2 = dy2.float()
        gres = None
        if g_x is not None:
            gres = g_x.reshape(rows, C)
            if gres.dtype != torch.float32 or not gres.is_contiguous():
                gres = gres.float().contiguous()
        dx = torch.empty(rows, C, dtype=torch.float32, device=xn.device)
        gb = torch.empty(rows, C, dtype=ctx.bdtype, device=xn.device)
        dgamma = torch.empty(C, dtype=torch.float32, device=xn.device)
        dbeta = torch.empty(C, dtype=torch.float32, device=xn.device)
        ws = torch.empty(L.vil_layernorm_workspace_bytes(rows, C) // 4, dtype=torch.float32, device=xn.device)
        stream = ctypes.c_void_p(torch.cuda.current_stream(xn.device).cuda_stream)
        _lib.check(L.vil_resln_bwd(_p(dy2), _DT[dy2.dtype], _p(gres) if gres is not None else None, _p(xn), _p(w),
                                   _p(mean), _p(rstd), _p(rscale) if rscale is not None else None, ctx.rps,
                                   _p(dx), _p(gb), _DT[gb.dtype], _p(dgamma), _p(dbeta), _p(ws), rows, C, stream))
        return (dx.view(ctx.shape), gb.view(ctx.shape), None, dgamma.to(ctx.wdtype), dbeta.to(ctx.wdtype), None, None)


class _PassLN(torch.autograd.Function):
    """(x, LayerNorm(x)) for the FIRST block of a stage, where x feeds both the norm and the residual stream: the
    backward adds the two gradients inside the LayerNorm-backward kernel (vil_resln_bwd with no branch) instead of an
    autograd accumulation pass over the fp32 stream."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, out_dtype):
        L = _lib.lib()
        C = x.shape[-1]
        x2 = x.reshape(-1, C)
        rows = x2.shape[0]
        w = weight.detach().float().contiguous()
        b = bias.detach().float().contiguous()
        y = torch.empty(rows, C, dtype=out_dtype, device=x.device)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        stream = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        with torch.cuda.device(x.device):
            _lib.check(L.vil_layernorm_fwd(_p(x2), _DT[x2.dtype], _p(w), _p(b), _p(y), _DT[out_dtype], _p(mean), _p(rstd),
                                           rows, C, x2.stride(0), y.stride(0), float(eps), stream))
        ctx.save_for_backward(x2, w, mean, rstd)
        ctx.shape, ctx.wdtype = x.shape, weight.dtype
        return x.view_as(x), y.view(x.shape)

    @staticmethod
    def backward(ctx, g_x, g_y):
        x2, w, mean, rstd = ctx.saved_tensors
        L = _lib.lib()
        rows, C = x2.shape
        if g_y is None:
            g_y = torch.zeros(rows, C, dtype=torch.bfloat16, device=x2.device)
        dy2 = g_y.reshape(rows, C)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        if dy2.dtype not in _DT:
            dy2 = dy2.float()
        gres = None
        if g_x is not None:
            gres = g_x.reshape(rows, C)
            if gres.dtype != torch.float32 or not gres.is_contiguous():
                gres = gres.float().contiguous()
        dx = torch.empty(rows, C, dtype=torch.float32, device=x2.device)
        dgamma = torch.empty(C, dtype=torch.float32, device=x2.device)
        dbeta = torch.empty(C, dtype=torch.float32, device=x2.device)
        ws = torch.empty(L.vil_layernorm_workspace_bytes(rows, C) // 4, dtype=torch.float32, device=x2.device)
        stream = ctypes.c_void_p(torch.cuda.current_stream(x2.device).cuda_stream)
        with torch.cuda.device(x2.device):
            _lib.check(L.vil_resln_bwd(_p(dy2), _DT[dy2.dtype], _p(gres) if gres is not None else None, _p(x2), _p(w),
                                       _p(mean), _p(rstd), None, rows, _p(dx), None, _DT[torch.float32], _p(dgamma),
                                       _p(dbeta), _p(ws), rows, C, stream))
        return dx.view(ctx.shape), dgamma.to(ctx.wdtype), dbeta.to(ctx.wdtype), None, None


def pass_layernorm_ok(x, norm):
    C = x.shape[-1]
    return (isinstance(norm, VilLayerNorm) and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
            and C % 8 == 0 and C <= 1024 and norm.elementwise_affine and norm.bias is not None)


def pass_layernorm(x, norm):
    """(x, norm(x)) with the gradient of both uses of x summed inside the LayerNorm backward; see _PassLN"""
    out_dtype = x.dtype
    if torch.is_autocast_enabled("cuda"):
        ac = torch.get_autocast_dtype("cuda")
        out_dtype = ac if (norm.cast_output and ac in _DT) else torch.float32
    return _PassLN.apply(x, norm.weight, norm.bias, norm.eps, out_dtype)


def res_layernorm_ok(x, branch, norm):
    C = x.shape[-1]
    return (isinstance(norm, VilLayerNorm) and x.is_cuda and x.dtype == torch.float32 and branch.dtype in _DT
            and branch.shape == x.shape and C % 8 == 0 and C <= 1024 and norm.elementwise_affine and norm.bias is not None)


def res_layernorm(x, branch, rscale, norm):
    """(x + rscale * branch, norm(x + rscale * branch)) with `norm` a VilLayerNorm; rscale (B,) fp32 or None."""
    out_dtype = x.dtype
    if torch.is_autocast_enabled("cuda"):
        ac = torch.get_autocast_dtype("cuda")
        out_dtype = ac if (norm.cast_output and ac in _DT) else torch.float32
    return _ResLN.apply(x, branch, rscale, norm.weight, norm.bias, norm.eps, out_dtype)


class VilLayerNorm(nn.LayerNorm):
    """cast_output=True: under autocast emit the autocast dtype (the consumer is a GEMM);
    False: keep the residual-stream dtype (PatchEmbed's norm_embed)."""

    def __init__(self, normalized_shape, eps=1e-5, cast_output=True, **kw):
        super().__init__(normalized_shape, eps=eps, **kw)
        self.cast_output = cast_output

    def forward(self, x):
        C = x.shape[-1]
        if (not x.is_cuda) or x.dtype not in _DT or C % 8 or C > 1024 or not self.elementwise_affine or self.bias is None:
            return super().forward(x)
        out_dtype = x.dtype
        if torch.is_autocast_enabled("cuda"):
            ac = torch.get_autocast_dtype("cuda")
            # PyTorch autocast runs LayerNorm in fp32 and the next GEMM casts its input to `ac`
            out_dtype = ac if (self.cast_output and ac in _DT) else torch.float32
        return _FusedLN.apply(x, self.weight, self.bias, self.eps, out_dtype)
