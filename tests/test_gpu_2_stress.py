"""Soak tests of every LDS-DMA kernel (`buffer_load ... lds` rings): the round-3 race (k_skinny K = 384 read a ring slot
whose DMA requests were still in flight, about one launch in 10^3 -- `__syncthreads()` does not wait for LDS-DMA,
DESIGN.md section 4.3) was found by an intermittent failure of a test that ran each shape ONCE.  Here every LDS-DMA kernel is
launched >= 3000 times per shape on fixed inputs and every output is compared bit for bit with the first launch, while a
second stream streams a large buffer (a memory hog that perturbs the arrival order of the DMA requests).

  k_skinny       every K instantiation (96 ... 768), forward (op 0) and input gradient (op 1)
  k_skinny_gelu  fc1 + GELU in the weights-in-registers kernel
  k_wgrad2       every (tile, slices) plan + the 128x128 kernel
  k_dgrad_dgelu, k_fwd_gelu   the 128x128 tile kernels of the MLP
  k_dense_fwd / k_dense_bwd_dq / k_dense_bwd_dkdv   the dense-stage attention family

The comparison runs on the device (one xor-accumulate per launch, one read-back per shape), so 3000 launches of a shape
cost about a second."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu

LAUNCHES = 3000


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


class _Hog:
    """A side stream that keeps HBM and the L2s busy with large copies while the kernel under test runs."""

    def __init__(self, dev, mb=192):
        self.stream = torch.cuda.Stream(device=dev)
        n = mb * (1 << 20) // 4
        self.a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
        self.b = torch.empty_like(self.a)

    def poke(self, n=2):
        with torch.cuda.stream(self.stream):
            for _ in range(n):
                self.b.copy_(self.a, non_blocking=True)
                self.a.add_(self.b, alpha=1e-9)

    def done(self):
        self.stream.synchronize()


def _bits(t):
    t = t.contiguous()
    return t.view(torch.int32) if t.element_size() == 4 else t.view(torch.int16)


def _soak(dev, launch, outputs, launches=LAUNCHES, poke_every=40):
    """launch(): runs the kernel(s) once into the same output tensors; outputs(): the tensors to compare.  Returns the
    number of launches whose output differed from the first launch in any bit."""
    hog = _Hog(dev)
    launch()
    torch.cuda.synchronize()
    ref = [_bits(t).clone() for t in outputs()]
    bad = torch.zeros(1, dtype=torch.int64, device=dev)
    for i in range(launches):
        if i % poke_every == 0:
            hog.poke()
        launch()
        for r, t in zip(ref, outputs()):
            bad += (_bits(t) != r).any()
    torch.cuda.synchronize()
    hog.done()
    return int(bad.item())


def _vp(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


@pytest.mark.parametrize("op,T,K,N", [(0, 30011, 96, 384), (0, 20003, 192, 576), (0, 12007, 288, 96), (0, 12007, 384, 96),
                                       (0, 9001, 576, 192), (0, 9001, 768, 192),
                                       (1, 30011, 96, 96), (1, 20003, 192, 192), (1, 12007, 288, 96), (1, 12007, 384, 96),
                                       (1, 9001, 576, 192), (1, 9001, 768, 192)])
def test_soak_skinny_gemm(dev, op, T, K, N):
    from vision_longformer_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(T, K, generator=g).bfloat16().to(dev)
    w = (torch.randn(*((N, K) if op == 0 else (K, N)), generator=g) * 0.1).bfloat16().to(dev)
    b = torch.randn(N, generator=g).bfloat16().to(dev) if op == 0 else None
    out = torch.empty(T, N, dtype=torch.bfloat16, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def launch():
        _lib.check(L.vil_gemm_skinny_bf16(op, _vp(x), _vp(w), _vp(b), _vp(out), T, K, N, x.stride(0), N, st))

    want = x.double() @ (w.double().t() if op == 0 else w.double()) + (b.double() if b is not None else 0)
    nbad = _soak(dev, launch, lambda: [out])
    assert nbad == 0, f"{nbad} of {LAUNCHES} launches differ"
    assert (out.double() - want).abs().max().item() <= 1.2e-2 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("T,K,N", [(30011, 96, 384), (20003, 192, 768)])
def test_soak_skinny_gemm_gelu(dev, T, K, N):
    from vision_longformer_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(6)
    x = torch.randn(T, K, generator=g).bfloat16().to(dev)
    w = (torch.randn(N, K, generator=g) * 0.1).bfloat16().to(dev)
    b = torch.randn(N, generator=g).bfloat16().to(dev)
    both = torch.empty(2, T, N, dtype=torch.bfloat16, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def launch():
        _lib.check(L.vil_gemm_skinny_gelu_bf16(_vp(x), _vp(w), _vp(b), _vp(both[0]), _vp(both[1]), T, K, N, x.stride(0), N, st))

    nbad = _soak(dev, launch, lambda: [both])
    assert nbad == 0, f"{nbad} of {LAUNCHES} launches differ"


@pytest.mark.parametrize("T,K,N", [(6400, 384, 1536), (3200, 768, 3072), (12007, 96, 384)])
def test_soak_tile_gemm_gelu(dev, T, K, N):
    """k_fwd_gelu (vil_gemm_gelu_bf16): fc1 + GELU on the 128 x 128 tile kernel"""
    from vision_longformer_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(7)
    x = torch.randn(T, K, generator=g).bfloat16().to(dev)
    w = (torch.randn(N, K, generator=g) * 0.1).bfloat16().to(dev)
    b = torch.randn(N, generator=g).bfloat16().to(dev)
    both = torch.empty(2, T, N, dtype=torch.bfloat16, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def launch():
        _lib.check(L.vil_gemm_gelu_bf16(_vp(x), _vp(w), _vp(b), _vp(both[0]), _vp(both[1]), T, K, N, x.stride(0), N, st))

    nbad = _soak(dev, launch, lambda: [both])
    assert nbad == 0, f"{nbad} of {LAUNCHES} launches differ"


@pytest.mark.parametrize("op,T,K,N", [(0, 6400, 384, 1152), (0, 9001, 384, 384), (0, 3200, 768, 768), (1, 6400, 384, 384), (1, 3200, 768, 768),
                                      (1, 9001, 1152, 384)])
def test_soak_plain_tile_gemm(dev, op, T, K, N):
    """vil_gemm_tile_bf16 (k_fwd_gelu<false> / k_dgrad_dgelu<false>): the persistent loader-wave tile kernels as plain GEMMs"""
    from vision_longformer_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(12)
    x = torch.randn(T, K, generator=g).bfloat16().to(dev)
    w = (torch.randn(*((N, K) if op == 0 else (K, N)), generator=g) * 0.1).bfloat16().to(dev)
    b = torch.randn(N, generator=g).bfloat16().to(dev) if op == 0 else None
    out = torch.empty(T, N, dtype=torch.bfloat16, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def launch():
        _lib.check(L.vil_gemm_tile_bf16(op, _vp(x), _vp(w), _vp(b), _vp(out), T, K, N, K, N, st))

    nbad = _soak(dev, launch, lambda: [out])
    assert nbad == 0, f"{nbad} of {LAUNCHES} launches differ"
    want = x.double() @ (w.double().t() if op == 0 else w.double()) + (b.double() if b is not None else 0)
    assert (out.double() - want).abs().max().item() <= 1.2e-2 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("T,K,N", [(6400, 384, 1536), (3200, 768, 3072), (12007, 96, 384), (9001, 192, 768)])
def test_soak_dgrad_dgelu(dev, T, K, N):
    """k_dgrad_dgelu (vil_gemm_dgelu_bf16): fc2's input gradient with the GELU backward in its epilogue"""
    from vision_longformer_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(8)
    dy = torch.randn(T, K, generator=g).bfloat16().to(dev)
    w = (torch.randn(K, N, generator=g) * 0.1).bfloat16().to(dev)
    h = torch.randn(T, N, generator=g).bfloat16().to(dev)
    dh = torch.empty(T, N, dtype=torch.bfloat16, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def launch():
        _lib.check(L.vil_gemm_dgelu_bf16(_vp(dy), _vp(w), _vp(h), _vp(dh), T, K, N, dy.stride(0), h.stride(0), N, st))

    nbad = _soak(dev, launch, lambda: [dh])
    assert nbad == 0, f"{nbad} of {LAUNCHES} launches differ"


@pytest.mark.parametrize("T,CO,CI", [(20011, 384, 192), (9000, 192, 96), (6272, 768, 384)])
def test_soak_wgrad_every_plan(dev, T, CO, CI):
    """k_wgrad2 under every (tile, slices) plan and the 128 x 128 kernel: LAUNCHES launches split over the plans"""
    from vision_longformer_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(9)
    dy = (torch.randn(T, CO, generator=g) * 0.1).bfloat16().to(dev)
    x = torch.randn(T, CI, generator=g).bfloat16().to(dev)
    ws = torch.empty(L.vil_linear_wgrad_workspace_bytes(T, CO, CI) // 4 + 64, dtype=torch.float32, device=dev)
    dw = torch.empty(CO, CI, dtype=torch.float32, device=dev)
    db = torch.empty(CO, dtype=torch.float32, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    plans = [(1, 0, 0, 0)] + [(2, mi, nj, m) for mi in (3, 6) for nj in (3, 6) if CO % (32 * mi) == 0 and CI % (32 * nj) == 0
                              for m in (1, 3, 64)]

    def launch():
        _lib.check(L.vil_linear_wgrad(_vp(dy), _vp(x), T, CO, CI, dy.stride(0), x.stride(0), _vp(dw), _vp(db), 0, _vp(ws), st))

    per_plan = -(-LAUNCHES // len(plans))
    try:
        for plan in plans:
            _lib.check(L.vil_linear_wgrad_set_plan(T, CO, CI, *plan))
            nbad = _soak(dev, launch, lambda: [dw, db], launches=per_plan)
            assert nbad == 0, f"plan {plan}: {nbad} of {per_plan} launches differ"
    finally:
        _lib.check(L.vil_linear_wgrad_set_plan(T, CO, CI, 0, 0, 0, 0))


@pytest.mark.parametrize("nx,ny,G,H,B", [(14, 14, 1, 6, 8), (24, 24, 1, 6, 2), (7, 7, 1, 12, 8), (12, 12, 0, 12, 4)])
def test_soak_dense_attention_family(dev, nx, ny, G, H, B):
    """k_dense_fwd, k_dense_bwd_dq, k_dense_bwd_dkdv (+ the reduce): forward and backward through the C ABI, every
    output and every gradient (incl. the fixed-point bias gradients) bit-identical over the soak"""
    from vision_longformer_amd import _lib
    from vision_longformer_amd.ops import _VilDenseAttention  # noqa: F401  (the op the module calls)
    from vision_longformer_amd.ops import vil_dense_attention
    M = 64
    g = torch.Generator().manual_seed(10)
    N, C = G + nx * ny, H * M
    qkv = torch.randn(B, N, 3 * C, generator=g).bfloat16().to(dev).requires_grad_(True)
    table = (torch.randn((2 * nx - 1) * (2 * ny - 1), H, generator=g) * 0.3).to(dev).requires_grad_(True)
    g2l = (torch.randn(2, H, G, generator=g) * 0.3).to(dev).requires_grad_(True) if G else None
    g2g = (torch.randn(H, G, G, generator=g) * 0.3).to(dev).requires_grad_(True) if G else None
    dout = torch.randn(B, N, C, generator=g).bfloat16().to(dev)
    leaves = [t for t in (qkv, table, g2l, g2g) if t is not None]
    res = {}

    def launch():
        for t in leaves:
            t.grad = None
        out = vil_dense_attention(qkv, table, g2l, g2g, nx=nx, ny=ny, nglo=G, num_heads=H, scale=M ** -0.5, backend="dense")
        out.backward(dout)
        res["o"] = [out.detach()] + [t.grad for t in leaves]

    nbad = _soak(dev, launch, lambda: res["o"], launches=LAUNCHES // 2, poke_every=20)
    assert nbad == 0, f"{nbad} of {LAUNCHES // 2} forward+backward passes differ"
    # the forward's narrow launch shape (4-wave workgroups, 32-row ring blocks; round 4) under the same soak
    L = _lib.lib()
    try:
        _lib.check(L.vil_dense_attn_set_fwd_shape(1))
        nbad = _soak(dev, launch, lambda: res["o"], launches=LAUNCHES // 2, poke_every=20)
    finally:
        _lib.check(L.vil_dense_attn_set_fwd_shape(-1))
    assert nbad == 0, f"narrow forward: {nbad} of {LAUNCHES // 2} forward+backward passes differ"
