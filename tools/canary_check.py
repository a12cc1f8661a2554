"""Diagnostic: out-of-bounds writes of the library's kernels into their workspace tails.
Every workspace handed to the library gets a canary-filled tail that is verified after the call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vision_longformer_amd import ops
dev = torch.device("cuda:0")
PAD = 1 << 16
live = []
orig = ops._workspace
def ws_canary(d, pass_, device):
    n = ops._lib.lib().vil_attn_workspace_bytes(__import__("ctypes").byref(d), pass_)
    n4 = max(int(n), 4) // 4 + 1
    t = torch.full((n4 + PAD,), 12345.0, dtype=torch.float32, device=device)
    live.append((t, n4, pass_, (d.B, d.H, d.M, d.nx, d.ny, d.W, d.G, d.mode)))
    return t
ops._workspace = ws_canary
def check(tag):
    torch.cuda.synchronize()
    for t, n4, pass_, desc in live:
        tail = t[n4:]
        bad = (tail != 12345.0).nonzero()
        if bad.numel():
            print(f"{tag}: OOB write pass {pass_} desc {desc}: {bad.numel()} words, first at +{int(bad[0])} (ws {n4} words)")
    live.clear()
g = torch.Generator().manual_seed(1)
for (B, H, M, W, G) in [(32, 12, 64, 7, 1), (128, 12, 64, 7, 1), (32, 6, 64, 14, 1), (128, 6, 64, 14, 1), (2, 2, 16, 5, 2)]:
    C = H * M; N = G + W * W
    qkv = torch.randn(B, N, 3 * C, generator=g).to(dev, torch.bfloat16).requires_grad_(True)
    table = (torch.randn((2 * W - 1) ** 2, H, generator=g) * 0.02).to(dev).requires_grad_(True)
    g2l = (torch.randn(2, H, G, generator=g) * 0.02).to(dev).requires_grad_(True)
    g2g = (torch.randn(H, G, G, generator=g) * 0.02).to(dev).requires_grad_(True)
    out = ops.vil_dense_attention(qkv, table, g2l, g2g, nx=W, ny=W, nglo=G, num_heads=H, scale=M ** -0.5)
    check(f"dense fwd B{B} H{H} W{W}")
    out.backward(torch.randn_like(out))
    check(f"dense bwd B{B} H{H} W{W}")
for (B, H, M, W, nx, G) in [(32, 3, 32, 7, 56, 1), (32, 3, 64, 7, 28, 1), (128, 3, 64, 7, 28, 1)]:
    C = H * M; N = G + nx * nx
    q = torch.randn(B, N, C, generator=g).to(dev, torch.bfloat16).requires_grad_(True)
    kv = torch.randn(B, N, 2 * C, generator=g).to(dev, torch.bfloat16).requires_grad_(True)
    table = (torch.randn((4 * W - 1) ** 2, H, generator=g) * 0.02).to(dev).requires_grad_(True)
    g2l = (torch.randn(2, H, G, generator=g) * 0.02).to(dev).requires_grad_(True)
    g2g = (torch.randn(H, G, G, generator=g) * 0.02).to(dev).requires_grad_(True)
    out = ops.vil_full_attention(q, kv, table, g2l, g2g, nx=nx, ny=nx, w=W, nglo=G, num_heads=H)
    check(f"full fwd B{B} H{H} nx{nx}")
    out.backward(torch.randn_like(out))
    check(f"full bwd B{B} H{H} nx{nx}")
print("canary check done")
