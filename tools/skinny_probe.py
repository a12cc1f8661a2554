import os, sys, torch
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd())
from vision_longformer_amd import linear
from vision_longformer_amd.linear import _gemm_skinny, _gemm
linear._SKINNY_FORCE = True
dev = torch.device("cuda:0")
def timed(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
CASES = [(0, 1000, 96, 384), (0, 401536, 96, 384), (0, 401536, 96, 288), (0, 401536, 96, 96), (0, 130, 96, 288),
         (0, 100480, 192, 576), (0, 100480, 192, 192), (0, 100480, 192, 768), (0, 401536, 384, 96), (0, 100480, 768, 192),
         (1, 401536, 96, 96), (1, 401536, 288, 96), (1, 401536, 384, 96), (1, 100480, 192, 192), (1, 100480, 576, 192),
         (1, 100480, 768, 192), (1, 4001, 288, 96), (1, 777, 768, 192), (0, 294944, 96, 384), (1, 73760, 576, 192)]
for op, T, K, N in CASES:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(T, K, generator=g).bfloat16().to(dev)
    w = (torch.randn(*((N, K) if op == 0 else (K, N)), generator=g) * 0.1).bfloat16().to(dev)
    b = torch.randn(N, generator=g).bfloat16().to(dev) if op == 0 else None
    y = _gemm_skinny(op, x, w, b)
    assert y is not None, (op, T, K, N)
    rows = torch.cat([torch.arange(0, min(T, 300)), torch.arange(max(T - 300, 0), T)]).unique()
    want = x[rows].double() @ (w.double().t() if op == 0 else w.double()) + (b.double() if b is not None else 0)
    err = (y[rows].double() - want).abs().max().item()
    ok = err <= 2e-2 * max(1.0, want.abs().max().item())
    ts = timed(lambda: _gemm_skinny(op, x, w, b)); tl = timed(lambda: _gemm(op, x, w, b))
    byts = 2.0 * T * (K + N)
    print(f"op {op} T={T} K={K} N={N}: err {err:.3e} {'OK' if ok else 'FAIL'} | skinny {ts:7.1f} us ({byts/ts/1e3:6.0f} GB/s)  library {tl:7.1f} us", flush=True)
