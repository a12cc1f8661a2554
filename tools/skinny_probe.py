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
for T, K, N in [(1000, 96, 384), (401536, 96, 384), (401536, 96, 288), (401536, 96, 96), (4001, 96, 96), (130, 96, 288),
                (100480, 192, 576), (100480, 192, 192), (100480, 192, 768), (777, 192, 576), (294944, 96, 384), (73760, 192, 768)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(T, K, generator=g).bfloat16().to(dev)
    w = (torch.randn(N, K, generator=g) * 0.1).bfloat16().to(dev)
    b = torch.randn(N, generator=g).bfloat16().to(dev)
    y = _gemm_skinny(x, w, b)
    assert y is not None
    rows = torch.cat([torch.arange(0, min(T, 300)), torch.arange(max(T - 300, 0), T)]).unique()
    want = x[rows].double() @ w.double().t() + b.double()
    err = (y[rows].double() - want).abs().max().item()
    ok = err <= 2e-2 * max(1.0, want.abs().max().item())
    ts = timed(lambda: _gemm_skinny(x, w, b)); tl = timed(lambda: _gemm(0, x, w, b))
    byts = 2.0 * T * (K + N)
    print(f"T={T} K={K} N={N}: err {err:.3e} {'OK' if ok else 'FAIL'} | skinny {ts:7.1f} us ({byts/ts/1e3:6.0f} GB/s)  library {tl:7.1f} us", flush=True)
