"""Operator-level drop-in of the reference's ``src/models/layers/slidingchunk_2d.py`` on HIP kernels
(C ABI ``vil_sc2d_*`` of libvilattn.so): same names, argument meaning, layouts and error behaviour

    slidingchunk_2d(t1, t2, is_t1_diagonaled=False, mode=0)      (= SlidingChunk2D.apply, :368)
    slidingchunk_2dautograd(t1, t2, is_t1_diagonaled, mode)      (:360-365)
    mask_invalid_locations(input_tensor, nx, ny, padx, pady, w, exact, mode=0) -> num_invalid   (:321-357)

so the reference's own test protocol (src/tests/test_slidingchunk_2d.py) and the operator-level golden vectors run
against the HIP path.  This surface MATERIALISES the (BH, mx, my, W^2, kv) score tensor, as the reference does: it is
for parity / compatibility.  The hot path of the product is ``ops.vil_full_attention`` (fused, no score tensor).

Layouts: images (BH, M, mx, my, W^2); scores (BH, mx, my, W^2, kv) with kv = 9 W^2 (mode 0), W^2 (mode -1) or
2 W^2 (mode 1..8: [own chunk | that neighbour], slidingchunk_2d.py:15-24).  float64, float32, bfloat16 and float16
tensors run natively (16-bit I/O with fp32 accumulation: what the reference's @autocast einsums do on GPU,
slidingchunk_2d.py:203,235)."""
import ctypes
import math

import torch

from . import _lib

_DT = {torch.float32: _lib.DTYPE_F32, torch.float64: _lib.DTYPE_F64, torch.bfloat16: _lib.DTYPE_BF16,
       torch.float16: _lib.DTYPE_F16}


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _prep(*ts):
    t0 = ts[0]
    if not t0.is_cuda:
        raise RuntimeError("slidingchunk_2d needs device tensors: the product path is the HIP kernels "
                           "(libvilattn.so); there is no CPU fallback")
    dt = t0.dtype if t0.dtype in _DT else torch.float32
    return dt, [t.to(dt).contiguous() for t in ts]


def _w_of(w2):
    w = int(round(math.sqrt(w2)))
    assert w * w == w2, "last image dimension must be W^2"
    return w


def _kv(mode, w2):
    return 9 * w2 if mode == 0 else (w2 if mode == -1 else 2 * w2)


def _qk(q_img, k_img, mode):
    dt, (q, k) = _prep(q_img, k_img)
    BH, M, mx, my, w2 = q.shape
    assert k.shape == q.shape
    attn = torch.empty(BH, mx, my, w2, _kv(mode, w2), dtype=dt, device=q.device)
    with torch.cuda.device(q.device):
        _lib.check(_lib.lib().vil_sc2d_qk(q.data_ptr(), k.data_ptr(), attn.data_ptr(), BH, M, mx, my, _w_of(w2), mode,
                                         _DT[dt], _stream(q)))
    return attn.to(q_img.dtype)


def _av(attn, v_img, mode):
    dt, (a, v) = _prep(attn, v_img)
    BH, M, mx, my, w2 = v.shape
    assert a.shape == (BH, mx, my, w2, _kv(mode, w2))
    out = torch.empty_like(v)
    with torch.cuda.device(v.device):
        _lib.check(_lib.lib().vil_sc2d_av(a.data_ptr(), v.data_ptr(), out.data_ptr(), BH, M, mx, my, _w_of(w2), mode,
                                         _DT[dt], _stream(v)))
    return out.to(v_img.dtype)


def _agrad(attn, grad_img, mode):
    dt, (a, g) = _prep(attn, grad_img)
    BH, M, mx, my, w2 = g.shape
    assert a.shape == (BH, mx, my, w2, _kv(mode, w2))
    out = torch.empty_like(g)
    with torch.cuda.device(g.device):
        _lib.check(_lib.lib().vil_sc2d_agrad(a.data_ptr(), g.data_ptr(), out.data_ptr(), BH, M, mx, my, _w_of(w2), mode,
                                            _DT[dt], _stream(g)))
    return out.to(grad_img.dtype)


class SlidingChunk2D(torch.autograd.Function):
    """Same contract as the reference class (slidingchunk_2d.py:11-246): `forward` dispatches on
    `is_t1_diagonaled`, saves (t1, t2); `backward` returns (grad_t1, grad_t2, None, None)."""
    slidingchunk_qk = staticmethod(_qk)
    slidingchunk_av = staticmethod(_av)
    slidingchunk_agrad = staticmethod(_agrad)

    @staticmethod
    def forward(ctx, t1, t2, is_t1_diagonaled=False, mode=0):
        ctx.save_for_backward(t1, t2)
        ctx.is_t1_diagonaled = is_t1_diagonaled
        ctx.mode = mode
        return _av(t1, t2, mode) if is_t1_diagonaled else _qk(t1, t2, mode)

    @staticmethod
    def backward(ctx, grad_output):
        t1, t2 = ctx.saved_tensors
        mode = ctx.mode
        if ctx.is_t1_diagonaled:
            grad_t1 = _qk(grad_output, t2, mode)
            grad_t2 = _agrad(t1, grad_output, mode)
        else:
            grad_t1 = _av(grad_output, t2, mode)
            grad_t2 = _agrad(grad_output, t1, mode)
        return grad_t1, grad_t2, None, None


def slidingchunk_2dautograd(t1, t2, is_t1_diagonaled=False, mode=0):
    """Reference :360-365 -- the variant differentiated by autograd through the primitive products (each product is
    its own autograd node here)."""
    return _QKNode.apply(t1, t2, mode) if not is_t1_diagonaled else _AVNode.apply(t1, t2, mode)


class _QKNode(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, mode):
        ctx.save_for_backward(q, k)
        ctx.mode = mode
        return _qk(q, k, mode)

    @staticmethod
    def backward(ctx, g):
        q, k = ctx.saved_tensors
        return _av(g, k, ctx.mode), _agrad(g, q, ctx.mode), None


class _AVNode(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, v, mode):
        ctx.save_for_backward(a, v)
        ctx.mode = mode
        return _av(a, v, mode)

    @staticmethod
    def backward(ctx, g):
        a, v = ctx.saved_tensors
        return _qk(g, v, ctx.mode), _agrad(a, g, ctx.mode), None


slidingchunk_2d = SlidingChunk2D.apply


def mask_invalid_locations(input_tensor, nx, ny, padx, pady, w, exact, mode=0):
    """In-place -inf on the key slots a chunk does not attend (reference :321-357; nx, ny are the CHUNK counts, as
    there).  Returns num_invalid as a 0-dim int64 device tensor.  exact=1 with mode != 0, or exact outside
    {0, 1, -1}, raises ValueError like the reference."""
    if exact not in (0, 1, -1) or (exact == 1 and mode != 0):
        raise ValueError("longsc exact should be in [0,1,-1]!")
    if not input_tensor.is_cuda:
        raise RuntimeError("mask_invalid_locations needs a device tensor (HIP kernels; no CPU fallback)")
    t = input_tensor
    BH, mx, my, w2, kv = t.shape
    assert (mx, my) == (nx, ny) and w2 == w * w and kv == _kv(mode, w2)
    work = t if (t.dtype in _DT and t.is_contiguous()) else t.to(torch.float32).contiguous()
    count = torch.zeros(1, dtype=torch.int64, device=t.device)
    with torch.cuda.device(t.device):
        _lib.check(_lib.lib().vil_sc2d_mask(work.data_ptr(), BH, mx, my, padx, pady, w, exact, mode, _DT[work.dtype],
                                           count.data_ptr(), _stream(t)))
    if work is not t:
        t.copy_(work)
    return count[0]
