"""GPU (-m gpu): the block-glue kernels around the hot path against fp64 references: fused
LayerNorm / residual-LayerNorm, bias-gradient column sums, fused weight+bias gradient, library GEMM wrapper."""
import math
import os
import subprocess

import numpy as np
import pytest
import torch

import golden_cases as GC
from gpu_common import (ROOT, report, case, cid, make_inputs, run_oracle, run_hip, compare, rms,
                        F32_TOL, BF16_TOL, LOW_TOL, SMALL)
from oracle import vil_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "the gpu-marked tests need a GPU"
    return torch.device("cuda:0")


# ---------------------------------------------------------------- block glue: fused LayerNorm
@pytest.mark.parametrize("C,rows", [(96, 1000), (48, 333), (192, 4097), (384, 777), (768, 130), (16, 70)])
@pytest.mark.parametrize("mode", ["fp32", "fp32_to_bf16", "bf16"])
def test_fused_layernorm_vs_torch(dev, C, rows, mode):
    from vision_longformer_amd.layernorm import VilLayerNorm
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(rows, C, generator=g) * 2 + 0.5)
    dy = torch.randn(rows, C, generator=g)
    ln = VilLayerNorm(C, eps=1e-6).to(dev)
    with torch.no_grad():
        ln.weight.normal_(1.0, 0.3); ln.bias.normal_(0, 0.3)
    ref = torch.nn.LayerNorm(C, eps=1e-6).double()
    ref.weight.data.copy_(ln.weight.detach().double().cpu()); ref.bias.data.copy_(ln.bias.detach().double().cpu())
    xin_dtype = torch.bfloat16 if mode == "bf16" else torch.float32
    xd = x.to(xin_dtype).to(dev).requires_grad_(True)
    xr = x.to(xin_dtype).double().requires_grad_(True)
    if mode == "fp32_to_bf16":
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = ln(xd)
        assert y.dtype == torch.bfloat16
    else:
        y = ln(xd)
        assert y.dtype == xin_dtype
    dyd = dy.to(y.dtype).to(dev)
    y.backward(dyd)
    yr = ref(xr)
    yr.backward(dyd.double().cpu())
    torch.cuda.synchronize()
    lo = mode != "fp32"
    torch.testing.assert_close(y.double().cpu(), yr.detach(), atol=3e-2 if lo else 2e-5, rtol=2e-2 if lo else 1e-5)
    torch.testing.assert_close(xd.grad.double().cpu(), xr.grad, atol=3e-2 if mode == "bf16" else 2e-4, rtol=2e-2 if mode == "bf16" else 1e-4)
    gs = max(1.0, float(ref.weight.grad.abs().max()))
    torch.testing.assert_close(ln.weight.grad.double().cpu(), ref.weight.grad, atol=2e-3 * gs, rtol=2e-3)
    torch.testing.assert_close(ln.bias.grad.double().cpu(), ref.bias.grad, atol=2e-3 * gs, rtol=2e-3)


@pytest.mark.parametrize("rows,C", [(25216, 384), (1000, 96), (6400, 3072), (777, 1152), (5, 8)])
def test_colsum_bias_gradient(dev, rows, C):
    """db of the projections (vil_colsum_bf16) against an fp64 column sum."""
    from vision_longformer_amd.linear import _colsum
    g = torch.Generator().manual_seed(5)
    x = torch.randn(rows, C, generator=g).bfloat16()
    got = _colsum(x.to(dev)).float().cpu().double()
    want = x.double().sum(0)
    err = (got - want).abs().max().item()
    tol = 4e-3 * max(1.0, want.abs().max().item())          # bf16 output rounding
    assert err <= tol, (err, tol)
    # strided view (a column slice of a wider matrix), as dY of a fused qkv projection would be
    wide = torch.randn(rows, 2 * C, generator=g).bfloat16()
    got = _colsum(wide.to(dev)[:, C:]).float().cpu().double()
    want = wide[:, C:].double().sum(0)
    assert (got - want).abs().max().item() <= 4e-3 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("T,CO,CI", [(25216, 1536, 384), (25216, 384, 1536), (6272, 3072, 768), (100480, 192, 576),
                                      (40000, 96, 288), (1031, 40, 72), (4096, 1000, 768)])
def test_linear_wgrad_fused(dev, T, CO, CI):
    """dW = dY^T X and db = colsum(dY) (vil_linear_wgrad) against fp64 on the same bf16 inputs."""
    from vision_longformer_amd.linear import _wgrad
    g = torch.Generator().manual_seed(9)
    dy = (torch.randn(T, CO, generator=g) * 0.1).bfloat16()
    x = torch.randn(T, CI, generator=g).bfloat16()
    res = _wgrad(dy.to(dev), x.to(dev), True)
    assert res is not None
    dw, db = res
    torch.cuda.synchronize()
    sub = slice(0, min(CO, 256))                               # fp64 reference on a slab of output rows (CPU time)
    want = dy[:, sub].double().t() @ x.double()
    got = dw[sub].float().cpu().double()
    scale = want.abs().max().item()
    assert (got - want).abs().max().item() <= 6e-3 * scale, ((got - want).abs().max().item(), scale)
    wdb = dy.double().sum(0)
    assert (db.float().cpu().double() - wdb).abs().max().item() <= 6e-3 * max(1.0, wdb.abs().max().item())
    # strided operands: column slices of wider matrices (dY of a packed projection)
    wide = (torch.randn(T, 2 * CO, generator=g) * 0.1).bfloat16()
    res = _wgrad(wide.to(dev)[:, CO:], x.to(dev), False)
    want = wide[:, CO:][:, sub].double().t() @ x.double()
    got = res[0][sub].float().cpu().double()
    assert (got - want).abs().max().item() <= 6e-3 * want.abs().max().item()


def _wgrad_f32(dy, x, want_db=True):
    """vil_linear_wgrad with fp32 outputs, straight through the C ABI (the plan is whatever the plan cache says)."""
    import ctypes
    from vision_longformer_amd import _lib
    L = _lib.lib()
    T, co = dy.shape
    ci = x.shape[1]
    ws = torch.empty(L.vil_linear_wgrad_workspace_bytes(T, co, ci) // 4 + 64, dtype=torch.float32, device=dy.device)
    dw = torch.empty(co, ci, dtype=torch.float32, device=dy.device)
    db = torch.empty(co, dtype=torch.float32, device=dy.device)
    vp = ctypes.c_void_p
    _lib.check(L.vil_linear_wgrad(vp(dy.data_ptr()), vp(x.data_ptr()), T, co, ci, dy.stride(0), x.stride(0), vp(dw.data_ptr()),
                                  vp(db.data_ptr()) if want_db else None, 0, vp(ws.data_ptr()),
                                  vp(torch.cuda.current_stream(dy.device).cuda_stream)))
    torch.cuda.synchronize()
    return dw, db


@pytest.mark.parametrize("T,CO,CI", [(20011, 384, 192), (9000, 192, 96), (5003, 96, 96), (12345, 96, 384), (3100, 768, 576), (1500, 288, 480)])
def test_linear_wgrad_every_plan(dev, T, CO, CI):
    """Every plan of the second-generation weight-gradient kernel (tile 96/192 x 96/192, 1 .. max token slices per XCD:
    LDS-DMA ring with out-of-range stages, swizzled transposed reads, accumulator-order partial records, the reduce
    pass) and the 128 x 128 kernel: fp32 outputs against fp64 at fp32-accumulation tolerance, ragged last slice,
    strided dY; the same plan twice is bit-identical."""
    g = torch.Generator().manual_seed(13)
    wide = (torch.randn(T, CO + 64, generator=g) * 0.1).bfloat16().to(dev)
    dy = wide[:, 64:]
    x = torch.randn(T, CI, generator=g).bfloat16().to(dev)
    want = (dy.double().t() @ x.double()).cpu()
    wdb = dy.double().sum(0).cpu()
    scale, sdb = want.abs().max().item(), max(1.0, wdb.abs().max().item())
    from vision_longformer_amd import _lib
    L = _lib.lib()
    plans = [(1, 0, 0, 0)] + [(2, mi, nj, m) for mi in (3, 6) for nj in (3, 6) if CO % (32 * mi) == 0 and CI % (32 * nj) == 0
                              for m in (1, 3, 64)]
    assert len(plans) > 3
    assert L.vil_linear_wgrad_set_plan(T, CO, CI, 2, 5, 3, 1) == -2            # VIL_E_SHAPE
    for plan in plans:
        _lib.check(L.vil_linear_wgrad_set_plan(T, CO, CI, *plan))
        dw, db = _wgrad_f32(dy, x)
        assert (dw.double().cpu() - want).abs().max().item() <= 2e-5 * scale, plan
        assert (db.double().cpu() - wdb).abs().max().item() <= 2e-5 * sdb, plan
        dw2, db2 = _wgrad_f32(dy, x)
        assert torch.equal(dw, dw2) and torch.equal(db, db2), plan
        dw3, _ = _wgrad_f32(dy, x, want_db=False)
        assert torch.equal(dw, dw3), plan
    _lib.check(L.vil_linear_wgrad_set_plan(T, CO, CI, 0, 0, 0, 0))


def test_linear_wgrad_tune_selects_a_plan(dev):
    """vil_linear_wgrad_tune measures the candidates on the caller's operands and the launch that follows runs the
    selected plan: the result still matches fp64, and tuning twice is harmless"""
    import ctypes
    from vision_longformer_amd import _lib
    L = _lib.lib()
    T, co, ci = 30000, 384, 192
    g = torch.Generator().manual_seed(2)
    dy = (torch.randn(T, co, generator=g) * 0.1).bfloat16().to(dev)
    x = torch.randn(T, ci, generator=g).bfloat16().to(dev)
    ws = torch.empty(L.vil_linear_wgrad_workspace_bytes(T, co, ci) // 4 + 64, dtype=torch.float32, device=dev)
    dw = torch.empty(co, ci, dtype=torch.float32, device=dev)
    db = torch.empty(co, dtype=torch.float32, device=dev)
    vp = ctypes.c_void_p
    args = (vp(dy.data_ptr()), vp(x.data_ptr()), T, co, ci, co, ci, vp(dw.data_ptr()), vp(db.data_ptr()), 0, vp(ws.data_ptr()),
            vp(torch.cuda.current_stream(dev).cuda_stream))
    for _ in range(2):
        _lib.check(L.vil_linear_wgrad_tune(*args))
        _lib.check(L.vil_linear_wgrad(*args))
        torch.cuda.synchronize()
        want = dy.double().t() @ x.double()
        assert (dw.double() - want).abs().max().item() <= 2e-5 * want.abs().max().item()
        assert (db.double() - dy.double().sum(0)).abs().max().item() <= 2e-5 * max(1.0, dy.double().sum(0).abs().max().item())


@pytest.mark.parametrize("B,N,C,bdt", [(4, 197, 384, torch.bfloat16), (2, 50, 768, torch.bfloat16), (3, 785, 192, torch.float32),
                                        (2, 3137, 96, torch.bfloat16)])
def test_residual_layernorm_fused(dev, B, N, C, bdt):
    """vil_resln_fwd/_bwd: (x + s*branch, LN(x + s*branch)) and all gradients against fp64."""
    from vision_longformer_amd.layernorm import VilLayerNorm, res_layernorm
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, N, C, generator=g)
    br = torch.randn(B, N, C, generator=g).to(bdt).float()
    sc = torch.tensor([0.0, 1.25, 1.25, 0.0][:B])
    ln = VilLayerNorm(C, eps=1e-6).to(dev)
    with torch.no_grad():
        ln.weight.copy_(1 + 0.1 * torch.randn(C, generator=g)); ln.bias.copy_(0.1 * torch.randn(C, generator=g))
    gx = torch.randn(B, N, C, generator=g)
    gy = torch.randn(B, N, C, generator=g).bfloat16().float()
    # fp64 reference
    xr, brr = x.double().requires_grad_(True), br.double().requires_grad_(True)
    wr, b_r = ln.weight.detach().double().cpu().requires_grad_(True), ln.bias.detach().double().cpu().requires_grad_(True)
    xn_r = xr + sc.double().view(B, 1, 1) * brr
    y_r = torch.nn.functional.layer_norm(xn_r, (C,), wr, b_r, 1e-6)
    ((xn_r * gx.double()).sum() + (y_r * gy.double()).sum()).backward()
    # fused
    xd, brd = x.to(dev).requires_grad_(True), br.to(dev, bdt).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        xn, y = res_layernorm(xd, brd, sc.to(dev), ln)
    assert y.dtype == torch.bfloat16 and xn.dtype == torch.float32
    ((xn * gx.to(dev)).sum() + (y.float() * gy.to(dev)).sum()).backward()
    torch.cuda.synchronize()
    torch.testing.assert_close(xn.detach().double().cpu(), xn_r.detach(), atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(y.detach().double().cpu(), y_r.detach(), atol=3e-2, rtol=1e-2)
    torch.testing.assert_close(xd.grad.double().cpu(), xr.grad, atol=1e-4, rtol=1e-4)
    tolb = dict(atol=2e-2, rtol=1e-2) if bdt == torch.bfloat16 else dict(atol=1e-4, rtol=1e-4)
    torch.testing.assert_close(brd.grad.double().cpu(), brr.grad, **tolb)
    gs = float(wr.grad.abs().max())
    torch.testing.assert_close(ln.weight.grad.double().cpu(), wr.grad, atol=2e-3 * gs, rtol=2e-3)
    torch.testing.assert_close(ln.bias.grad.double().cpu(), b_r.grad, atol=2e-3 * float(b_r.grad.abs().max()), rtol=2e-3)


@pytest.mark.parametrize("B,G,nx,ny,C,ph,pw,bdt,odt", [(3, 1, 56, 56, 96, 2, 2, torch.bfloat16, torch.bfloat16),
                                                       (2, 1, 28, 28, 192, 2, 2, torch.bfloat16, torch.bfloat16),
                                                       (2, 2, 12, 8, 48, 2, 4, torch.float32, torch.float32),
                                                       (4, 0, 14, 14, 384, 2, 2, None, torch.bfloat16),
                                                       (2, 1, 6, 9, 64, 3, 3, torch.bfloat16, torch.float32)])
def test_patchify_stage_transition(dev, B, G, nx, ny, C, ph, pw, bdt, odt):
    """vil_patchify_fwd / _bwd against the reference's op sequence in fp64: x + s*branch, x[:, G:], the image view and
    the (py, px, c) patch vectors a strided Conv2d consumes (msvit.py:500-507, 166-203), and both gradients."""
    from vision_longformer_amd.msvit import _Patchify
    g = torch.Generator().manual_seed(9)
    N = nx * ny
    x = torch.randn(B, G + N, C, generator=g)
    br = torch.randn(B, G + N, C, generator=g).to(bdt).float() if bdt is not None else None
    sc = torch.tensor([0.0, 1.25, 1.0, 1.25][:B]) if bdt is not None else None
    nxp, nyp = nx // ph, ny // pw
    gout = torch.randn(B * nxp * nyp, ph * pw * C, generator=g).to(odt).float()
    xr = x.double().requires_grad_(True)
    brr = br.double().requires_grad_(True) if br is not None else None
    xs = xr + sc.double().view(B, 1, 1) * brr if br is not None else xr
    ref = xs[:, G:].reshape(B, nxp, ph, nyp, pw, C).permute(0, 1, 3, 2, 4, 5).reshape(B * nxp * nyp, ph * pw * C)
    (ref * gout.double()).sum().backward()
    xd = x.to(dev).requires_grad_(True)
    brd = br.to(dev, bdt).requires_grad_(True) if br is not None else None
    out = _Patchify.apply(xd, brd, sc.to(dev) if sc is not None else None, G, nx, ny, ph, pw, odt)
    assert out.dtype == odt and out.shape == ref.shape
    (out.float() * gout.to(dev)).sum().backward()
    torch.cuda.synchronize()
    tol = dict(atol=2e-2, rtol=1e-2) if odt == torch.bfloat16 else dict(atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(out.detach().double().cpu(), ref.detach(), **tol)
    torch.testing.assert_close(xd.grad.double().cpu(), xr.grad, atol=1e-6, rtol=1e-6)
    if G:
        assert float(xd.grad[:, :G].abs().max()) == 0.0
    if br is not None:
        tolb = dict(atol=2e-2, rtol=1e-2) if bdt == torch.bfloat16 else dict(atol=1e-6, rtol=1e-6)
        torch.testing.assert_close(brd.grad.double().cpu(), brr.grad, **tolb)


@pytest.mark.parametrize("B,N,C,G,xdt", [(4, 196, 384, 1, torch.bfloat16), (3, 784, 192, 1, torch.bfloat16),
                                          (2, 3136, 96, 1, torch.bfloat16), (5, 49, 768, 2, torch.float32)])
def test_tokens_layernorm_fused(dev, B, N, C, G, xdt):
    """vil_layernorm_fwd_tokens / _bwd_tokens: cat((cls.expand(B), LayerNorm(x)), dim=1) and every gradient (x, cls,
    gamma, beta) against the fp64 concatenation (reference PatchEmbed, msvit.py:204-206)."""
    from vision_longformer_amd.layernorm import VilLayerNorm, tokens_layernorm, tokens_layernorm_ok
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, N, C, generator=g).to(xdt).float()
    cls = 0.5 * torch.randn(1, G, C, generator=g)
    ln = VilLayerNorm(C, eps=1e-6, cast_output=False).to(dev)
    with torch.no_grad():
        ln.weight.copy_(1 + 0.1 * torch.randn(C, generator=g)); ln.bias.copy_(0.1 * torch.randn(C, generator=g))
    gout = torch.randn(B, G + N, C, generator=g)
    xr, cr = x.double().requires_grad_(True), cls.double().requires_grad_(True)
    wr, b_r = ln.weight.detach().double().cpu().requires_grad_(True), ln.bias.detach().double().cpu().requires_grad_(True)
    out_r = torch.cat((cr.expand(B, -1, -1), torch.nn.functional.layer_norm(xr, (C,), wr, b_r, 1e-6)), dim=1)
    (out_r * gout.double()).sum().backward()
    xd = x.to(dev, xdt).requires_grad_(True)
    cd = cls.to(dev).requires_grad_(True)
    assert tokens_layernorm_ok(xd, cd, ln)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=xdt == torch.bfloat16):
        out = tokens_layernorm(xd, cd, ln)
    assert out.dtype == torch.float32 and out.shape == (B, G + N, C) and out.is_contiguous()
    (out * gout.to(dev)).sum().backward()
    torch.cuda.synchronize()
    torch.testing.assert_close(out.detach().double().cpu(), out_r.detach(), atol=1e-4, rtol=1e-4)
    tolx = dict(atol=2e-2, rtol=1e-2) if xdt == torch.bfloat16 else dict(atol=1e-4, rtol=1e-4)
    torch.testing.assert_close(xd.grad.double().cpu(), xr.grad, **tolx)
    torch.testing.assert_close(cd.grad.double().cpu(), cr.grad, atol=1e-4 * float(cr.grad.abs().max()) + 1e-5, rtol=1e-4)
    gs = float(wr.grad.abs().max())
    torch.testing.assert_close(ln.weight.grad.double().cpu(), wr.grad, atol=2e-3 * gs, rtol=2e-3)
    torch.testing.assert_close(ln.bias.grad.double().cpu(), b_r.grad, atol=2e-3 * float(b_r.grad.abs().max()), rtol=2e-3)


@pytest.mark.parametrize("T,K,N", [(25216, 384, 1536), (25216, 1536, 384), (6400, 768, 3072), (100480, 192, 576),
                                    (4000, 96, 288), (197, 768, 1000), (8, 48, 96)])
def test_library_gemm_selected_algorithm(dev, T, K, N):
    """vil_gemm_bf16 (hipBLASLt, measured algorithm choice): forward with bias and input gradient vs fp64."""
    from vision_longformer_amd.linear import _gemm
    g = torch.Generator().manual_seed(13)
    x = torch.randn(T, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) * 0.05).bfloat16()
    b = torch.randn(N, generator=g).bfloat16()
    dy = torch.randn(T, N, generator=g).bfloat16()
    rows = slice(0, min(T, 512))
    for rep in range(2):                                       # first call tunes, second uses the cached plan
        y = _gemm(0, x.to(dev), w.to(dev), b.to(dev))
        assert y is not None and y.shape == (T, N)
        want = x[rows].double() @ w.double().t() + b.double()
        err = (y[rows].float().cpu().double() - want).abs().max().item()
        assert err <= 2e-2 * max(1.0, want.abs().max().item()), err
        dx = _gemm(1, dy.to(dev), w.to(dev), None)
        want = dy[rows].double() @ w.double()
        err = (dx[rows].float().cpu().double() - want).abs().max().item()
        assert err <= 2e-2 * max(1.0, want.abs().max().item()), err
    # strided input rows (a column slice of a wider matrix)
    wide = torch.randn(T, 2 * K, generator=g).bfloat16()
    y = _gemm(0, wide.to(dev)[:, K:], w.to(dev), None)
    want = wide[rows, K:].double() @ w.double().t()
    assert (y[rows].float().cpu().double() - want).abs().max().item() <= 2e-2 * max(1.0, want.abs().max().item())


# ---------------------------------------------------------------- MLP tail: fc2's input gradient with the GELU backward fused
def _gelu_grad64(x):
    return 0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)


@pytest.mark.parametrize("T,K,N", [(25216, 384, 1536), (6400, 768, 3072), (4001, 96, 384), (130, 192, 768), (7, 32, 128),
                                    (401536, 96, 384)])
def test_fused_dgrad_dgelu_vs_fp64(dev, T, K, N):
    """vil_gemm_dgelu_bf16: dh = (dy W) * gelu'(h), exact-erf GELU (reference msvit.py:17-34), against fp64 on sampled rows;
    includes a K that is not a multiple of the 64-deep block, a ragged last row tile and the stage-1 token count"""
    from vision_longformer_amd import linear
    from vision_longformer_amd.linear import _dgrad_dgelu
    linear._DGELU_FORCE = True                   # every shape the kernel accepts, not only the ones the product routes to it
    g = torch.Generator().manual_seed(29)
    dy = torch.randn(T, K, generator=g).bfloat16().to(dev)
    w = (torch.randn(K, N, generator=g) * 0.05).bfloat16().to(dev)
    h = (torch.randn(T, N, generator=g) * 1.5).bfloat16().to(dev)
    dh = _dgrad_dgelu(dy, w, h)
    assert dh is not None and dh.shape == (T, N) and dh.dtype == torch.bfloat16
    rows = torch.cat([torch.arange(0, min(T, 200)), torch.arange(max(T - 200, 0), T), torch.randint(0, T, (200,), generator=g)]).unique()
    want = (dy[rows].double() @ w.double()) * _gelu_grad64(h[rows].double())
    got = dh[rows].double()
    err = (got - want).abs().max().item()
    assert err <= 1.2e-2 * max(1.0, want.abs().max().item()), err          # bf16 output rounding (2^-8 relative)
    # strided dy (a column slice of a wider gradient) and strided h
    wide = torch.randn(T, 2 * K, generator=g).bfloat16().to(dev)
    hw = (torch.randn(T, N + 64, generator=g)).bfloat16().to(dev)
    dh2 = _dgrad_dgelu(wide[:, K:], w, hw[:, :N])
    want = (wide[rows][:, K:].double() @ w.double()) * _gelu_grad64(hw[rows][:, :N].double())
    assert (dh2[rows].double() - want).abs().max().item() <= 1.2e-2 * max(1.0, want.abs().max().item())
    linear._DGELU_FORCE = False


@pytest.mark.parametrize("B,N,C", [(4, 197, 384), (2, 3137, 96), (3, 50, 768)])
def test_mlp_block_fused_tail_matches_unfused(dev, B, N, C):
    """msvit.Mlp (fc1 -> exact GELU -> fc2) through vil_gelu_linear against the same module evaluated with plain torch
    ops in fp64: output, input gradient and every parameter gradient under bf16 autocast"""
    from vision_longformer_amd.msvit import Mlp
    torch.manual_seed(5)
    m = Mlp(C, 4 * C).to(dev)
    x = torch.randn(B, N, C, device=dev, requires_grad=True)
    dout = torch.randn(B, N, C, device=dev)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = m(x)
    y.float().backward(dout)
    torch.cuda.synchronize()
    m64 = Mlp(C, 4 * C).double().cpu()
    m64.load_state_dict({k: v.double().cpu() for k, v in m.state_dict().items()})
    x64 = x.detach().double().cpu().requires_grad_(True)
    y64 = torch.nn.functional.linear(torch.nn.functional.gelu(torch.nn.functional.linear(x64, m64.fc1.weight, m64.fc1.bias)),
                                     m64.fc2.weight, m64.fc2.bias)
    y64.backward(dout.double().cpu())
    def close(a, b, what):
        a, b = a.detach().double().cpu(), b.detach()
        e = (a - b).abs().max().item()
        assert e <= 3e-2 * max(b.abs().max().item(), 1e-3) + 0.1 * rms(b), (what, e, b.abs().max().item())
    close(y, y64, "y")
    close(x.grad, x64.grad, "dx")
    for (n, p), (_, p64) in zip(m.named_parameters(), m64.named_parameters()):
        close(p.grad, p64.grad, n)


@pytest.mark.parametrize("T,K,N", [(401536, 96, 384), (100480, 192, 768), (50001, 96, 288), (777, 192, 200), (9, 96, 8)])
def test_skinny_gemm_gelu_epilogue(dev, T, K, N):
    """vil_gemm_skinny_gelu_bf16: h = x W^T + b and gelu(h) (exact erf form) in one launch.  h is bit-identical to the plain
    kernel's output; the activation is the GELU of the ROUNDED h (what Linear -> nn.GELU computes) to bf16 rounding plus
    the 4e-7 absolute error of the erfc approximation, and never has the wrong sign"""
    from vision_longformer_amd import linear
    from vision_longformer_amd.linear import _gemm_skinny, _gemm_skinny_gelu
    linear._SKINNY_FORCE = True
    cap, linear._GELU_EPILOGUE_MAX_K = linear._GELU_EPILOGUE_MAX_K, 192
    try:
        g = torch.Generator().manual_seed(37)
        x = torch.randn(T, K, generator=g).bfloat16().to(dev)
        w = (torch.randn(N, K, generator=g) * 0.25).bfloat16().to(dev)
        b = torch.randn(N, generator=g).bfloat16().to(dev)
        for bias in (b, None):
            h, a = _gemm_skinny_gelu(x, w, bias)
            assert torch.equal(h, _gemm_skinny(0, x, w, bias))
            want = torch.nn.functional.gelu(h.double())
            err = (a.double() - want).abs()
            assert bool((err <= want.abs() * 2.0 ** -8 + 4e-7).all()), err.max().item()
            assert bool((a.float() * h.float() >= 0).all())
            assert (h.float().min().item() < -4.0 and h.float().max().item() > 4.0) or T < 1000     # both tails exercised
    finally:
        linear._SKINNY_FORCE = False
        linear._GELU_EPILOGUE_MAX_K = cap


@pytest.mark.parametrize("T,K,N", [(25216, 384, 1536), (6400, 768, 3072), (100480, 192, 768), (40001, 96, 384), (1031, 160, 128)])
def test_tile_gemm_gelu_epilogue_vs_fp64(dev, T, K, N):
    """vil_gemm_gelu_bf16 (128 x 128 tiles, weight rows permuted by the DMA so that a lane owns 8 consecutive features):
    h = x W^T + b against fp64 on sampled rows, a = gelu of the rounded h to bf16 rounding + the erfc approximation's
    4e-7; ragged last tile, K not a multiple of the 64-deep ring block, strided x, no bias"""
    from vision_longformer_amd import linear
    from vision_longformer_amd.linear import _gemm_tile_gelu
    old, linear._GELU_TILE_MIN_K = linear._GELU_TILE_MIN_K, 32
    try:
        g = torch.Generator().manual_seed(41)
        wide = torch.randn(T, K + 40, generator=g).bfloat16().to(dev)
        x = wide[:, 40:]
        w = (torch.randn(N, K, generator=g) * 0.15).bfloat16().to(dev)
        b = torch.randn(N, generator=g).bfloat16().to(dev)
        rows = torch.cat([torch.arange(0, 300), torch.arange(T - 300, T), torch.randint(0, T, (200,), generator=g)]).unique()
        for bias in (b, None):
            h, a = _gemm_tile_gelu(x, w, bias)
            want = x[rows].double() @ w.double().t() + (bias.double() if bias is not None else 0)
            assert (h[rows].double() - want).abs().max().item() <= 1.2e-2 * max(1.0, want.abs().max().item())
            ga = torch.nn.functional.gelu(h.double())
            assert bool(((a.double() - ga).abs() <= ga.abs() * 2.0 ** -8 + 4e-7).all())
            assert bool((a.float() * h.float() >= 0).all())
    finally:
        linear._GELU_TILE_MIN_K = old


@pytest.mark.parametrize("T,K,N", [(25216, 384, 1152), (25216, 384, 384), (6400, 768, 768), (18464, 384, 384), (1031, 160, 128),
                                   (4099, 1536, 384), (2048, 96, 256)])
def test_plain_tile_gemm_vs_fp64(dev, T, K, N):
    """vil_gemm_tile_bf16 (the loader-wave tile kernels without their activation epilogues): forward with / without bias
    and the input gradient against fp64 on sampled rows; ragged last tile, K not a multiple of the 64-deep ring block,
    strided input, and the routing rule of linear._gemm_tile"""
    import ctypes
    from vision_longformer_amd import _lib, linear
    L = _lib.lib()
    g = torch.Generator().manual_seed(47)
    wide = torch.randn(T, K + 24, generator=g).bfloat16().to(dev)
    x = wide[:, 24:]
    w = (torch.randn(N, K, generator=g) * 0.1).bfloat16().to(dev)
    b = torch.randn(N, generator=g).bfloat16().to(dev)
    dy = torch.randn(T, N, generator=g).bfloat16().to(dev)
    rows = torch.cat([torch.arange(0, 200), torch.arange(T - 200, T), torch.randint(0, T, (200,), generator=g)]).unique()
    vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for bias in (b, None):
        out = torch.full((T, N), float("nan"), dtype=torch.bfloat16, device=dev)
        _lib.check(L.vil_gemm_tile_bf16(0, vp(x), vp(w), vp(bias), vp(out), T, K, N, x.stride(0), N, st))
        want = x[rows].double() @ w.double().t() + (bias.double() if bias is not None else 0)
        assert (out[rows].double() - want).abs().max().item() <= 1.2e-2 * max(1.0, want.abs().max().item())
        assert bool(torch.isfinite(out.float()).all())
    dx = torch.full((T, K), float("nan"), dtype=torch.bfloat16, device=dev)
    if K % 128 == 0:
        _lib.check(L.vil_gemm_tile_bf16(1, vp(dy), vp(w), None, vp(dx), T, N, K, N, K, st))
        want = dy[rows].double() @ w.double()
        assert (dx[rows].double() - want).abs().max().item() <= 1.2e-2 * max(1.0, want.abs().max().item())
        assert bool(torch.isfinite(dx.float()).all())
    else:
        assert L.vil_gemm_tile_bf16(1, vp(dy), vp(w), None, vp(dx), T, N, K, N, K, st) == _lib.VIL_E_BACKEND
    # contract violations
    assert L.vil_gemm_tile_bf16(1, vp(dy), vp(w), vp(b), vp(dx), T, N, K, N, K, st) != 0           # a bias with the input gradient
    assert L.vil_gemm_tile_bf16(2, vp(x), vp(w), None, vp(dx), T, K, N, x.stride(0), N, st) != 0
    # the host routes exactly the shapes of its rule here
    got = linear._gemm_tile(0, x, w, b)
    assert (got is not None) == (linear._tile_gemm_takes(0, K, N) and K % 32 == 0 and N % 128 == 0 and T >= 1024)


@pytest.mark.parametrize("B,N,C", [(3, 3137, 96), (2, 785, 192)])
def test_mlp_block_with_gelu_epilogue_matches_unfused(dev, B, N, C):
    """msvit.Mlp with fc1 + GELU as ONE launch (vil_linear_gelu -> vil_gemm_skinny_gelu_bf16) against the same module
    with the kernel family switched off: identical h, so outputs and gradients agree to the rounding of the activation"""
    from vision_longformer_amd import linear
    from vision_longformer_amd.msvit import Mlp
    torch.manual_seed(6)
    m = Mlp(C, 4 * C).to(dev)
    x0 = torch.randn(B, N, C, device=dev)
    dout = torch.randn(B, N, C, device=dev)
    res = []
    for force, min_t in ((True, linear._SKINNY_MIN_T), (False, 1 << 60)):
        linear._SKINNY_FORCE, old, cap = force, linear._SKINNY_MIN_T, linear._GELU_EPILOGUE_MAX_K
        linear._SKINNY_MIN_T, linear._GELU_EPILOGUE_MAX_K = min_t, 192
        try:
            x = x0.clone().requires_grad_(True)
            m.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = m(x)
            y.float().backward(dout)
            res.append([y.detach().float(), x.grad.float()] + [p.grad.float() for p in m.parameters()])
        finally:
            linear._SKINNY_FORCE, linear._SKINNY_MIN_T, linear._GELU_EPILOGUE_MAX_K = False, old, cap
    for a, b in zip(*res):
        assert (a - b).abs().max().item() <= 2e-2 * max(b.abs().max().item(), 1e-3) + 0.05 * rms(b)


@pytest.mark.parametrize("op,T,K,N", [(0, 401536, 96, 384), (0, 100480, 192, 576), (0, 100480, 192, 768), (0, 50001, 96, 288),
                                       (0, 4001, 96, 96), (0, 777, 192, 192), (0, 130, 96, 200), (0, 9, 192, 8),
                                       (0, 100003, 384, 96), (0, 50001, 768, 192), (1, 401536, 288, 96), (1, 100003, 384, 96),
                                       (1, 100480, 576, 192), (1, 50001, 768, 192), (1, 4001, 96, 96), (1, 777, 192, 192),
                                       (1, 130, 288, 72), (1, 33, 576, 136)])
def test_skinny_gemm_vs_fp64(dev, op, T, K, N):
    """vil_gemm_skinny_bf16 (weights in registers, LDS-DMA ring of activation tiles, csrc/vil_gemm_skinny.hip): the
    forward of an nn.Linear with bias (op 0) and its input gradient (op 1: weight read k-strided through transposed LDS
    reads) against fp64 on sampled rows; ragged last tile, N not a multiple of the 32-feature pair, strided input rows"""
    from vision_longformer_amd import linear
    from vision_longformer_amd.linear import _gemm_skinny
    linear._SKINNY_FORCE = True
    try:
        g = torch.Generator().manual_seed(31)
        x = torch.randn(T, K, generator=g).bfloat16().to(dev)
        w = (torch.randn(*((N, K) if op == 0 else (K, N)), generator=g) * 0.1).bfloat16().to(dev)
        b = torch.randn(N, generator=g).bfloat16().to(dev)
        wd = w.double().t() if op == 0 else w.double()
        rows = torch.cat([torch.arange(0, min(T, 300)), torch.arange(max(T - 300, 0), T), torch.randint(0, T, (200,), generator=g)]).unique()
        for bias in ((b, None) if op == 0 else (None,)):
            y = _gemm_skinny(op, x, w, bias)
            assert y is not None and y.shape == (T, N)
            want = x[rows].double() @ wd + (bias.double() if bias is not None else 0)
            err = (y[rows].double() - want).abs().max().item()
            assert err <= 1.2e-2 * max(1.0, want.abs().max().item()), err
        wide = torch.randn(T, 2 * K, generator=g).bfloat16().to(dev)
        y = _gemm_skinny(op, wide[:, K:], w, b if op == 0 else None)
        want = wide[rows][:, K:].double() @ wd + (b.double() if op == 0 else 0)
        assert (y[rows].double() - want).abs().max().item() <= 1.2e-2 * max(1.0, want.abs().max().item())
    finally:
        linear._SKINNY_FORCE = False
