"""Diagnostic: does every element of the outputs get written?  Outputs are pre-filled with NaN."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vision_longformer_amd import ops
dev = torch.device("cuda:0")
orig_empty, orig_empty_like = torch.empty, torch.empty_like
def nan_empty(*a, **k):
    t = orig_empty(*a, **k)
    if t.is_floating_point() and t.is_cuda:
        t.fill_(float("nan"))
    return t
def nan_empty_like(x, **k):
    t = orig_empty_like(x, **k)
    if t.is_floating_point() and t.is_cuda:
        t.fill_(float("nan"))
    return t
torch.empty, torch.empty_like = nan_empty, nan_empty_like
g = torch.Generator().manual_seed(1)
def rep(tag, **ts):
    torch.cuda.synchronize()
    for n, t in ts.items():
        if t is not None and not torch.isfinite(t).all():
            bad = (~torch.isfinite(t)).nonzero()
            print(f"{tag}: {n} has {bad.shape[0]} non-finite of {t.numel()}, first idx {bad[0].tolist()}")
for (B, H, M, W, G) in [(32, 12, 64, 7, 1), (128, 12, 64, 7, 1), (32, 6, 64, 14, 1), (2, 2, 16, 5, 2), (4, 3, 32, 7, 1), (4, 3, 48, 9, 1)]:
    C = H * M; N = G + W * W
    qkv = torch.randn(B, N, 3 * C, generator=g).to(dev, torch.bfloat16).requires_grad_(True)
    table = (torch.randn((2 * W - 1) ** 2, H, generator=g) * 0.02).to(dev).requires_grad_(True)
    g2l = (torch.randn(2, H, G, generator=g) * 0.02).to(dev).requires_grad_(True)
    g2g = (torch.randn(H, G, G, generator=g) * 0.02).to(dev).requires_grad_(True)
    out = ops.vil_dense_attention(qkv, table, g2l, g2g, nx=W, ny=W, nglo=G, num_heads=H, scale=M ** -0.5)
    rep(f"dense fwd B{B} H{H} M{M} W{W} G{G}", out=out)
    out.backward(torch.randn(out.shape, generator=g).to(dev, torch.bfloat16))
    rep(f"dense bwd B{B} H{H} M{M} W{W} G{G}", dqkv=qkv.grad, dtable=table.grad, dg2l=g2l.grad, dg2g=g2g.grad)
for (B, H, M, W, nx, G, mode) in [(8, 3, 32, 7, 56, 1, 0), (8, 3, 64, 7, 28, 1, 0), (4, 3, 64, 7, 30, 1, 0), (4, 3, 64, 8, 28, 2, 3)]:
    C = H * M; N = G + nx * nx
    q = torch.randn(B, N, C, generator=g).to(dev, torch.bfloat16).requires_grad_(True)
    kv = torch.randn(B, N, 2 * C, generator=g).to(dev, torch.bfloat16).requires_grad_(True)
    table = (torch.randn((4 * W - 1) ** 2, H, generator=g) * 0.02).to(dev).requires_grad_(True)
    g2l = (torch.randn(2, H, G, generator=g) * 0.02).to(dev).requires_grad_(True)
    g2g = (torch.randn(H, G, G, generator=g) * 0.02).to(dev).requires_grad_(True)
    out = ops.vil_full_attention(q, kv, table, g2l, g2g, nx=nx, ny=nx, w=W, nglo=G, num_heads=H, mode=mode)
    rep(f"full fwd nx{nx} M{M} W{W} mode{mode}", out=out)
    out.backward(torch.randn(out.shape, generator=g).to(dev, torch.bfloat16))
    rep(f"full bwd nx{nx} M{M} W{W} mode{mode}", dq=q.grad, dkv=kv.grad, dtable=table.grad, dg2l=g2l.grad, dg2g=g2g.grad)
print("unwritten check done")
