"""Diagnostic: the library's own op (which issues hipMemsetAsync nodes) under hipGraph replay vs eager."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vision_longformer_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
B, H, M, W, nx, G = 16, 3, 64, 7, 28, 1
C = H * M; N = G + nx * nx
q = torch.randn(B, N, C, generator=g).to(dev, torch.bfloat16).requires_grad_(True)
kv = torch.randn(B, N, 2 * C, generator=g).to(dev, torch.bfloat16).requires_grad_(True)
table = (torch.randn((4 * W - 1) ** 2, H, generator=g) * 0.02).to(dev).requires_grad_(True)
g2l = (torch.randn(2, H, G, generator=g) * 0.02).to(dev).requires_grad_(True)
g2g = (torch.randn(H, G, G, generator=g) * 0.02).to(dev).requires_grad_(True)
dout = torch.randn(B, N, C, generator=g).to(dev, torch.bfloat16)
leaves = [q, kv, table, g2l, g2g]
def step():
    for t in leaves:
        t.grad = None
    out = ops.vil_full_attention(q, kv, table, g2l, g2g, nx=nx, ny=nx, w=W, nglo=G, num_heads=H)
    out.backward(dout)
    return out
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2): step()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    out_g = step()
grads_g = [t.grad for t in leaves]
names = ["dq", "dkv", "dtable", "dg2l", "dg2g"]
nbad = 0
for it in range(50):
    with torch.no_grad():
        q.copy_(torch.randn(B, N, C, device=dev)); kv.copy_(torch.randn(B, N, 2 * C, device=dev) * (1 + it % 3))
        dout.copy_(torch.randn(B, N, C, device=dev) * (0.1 + (it % 5)))
    gr.replay(); torch.cuda.synchronize()
    got = [t.clone() for t in grads_g]; og = out_g.clone()
    oe = step(); torch.cuda.synchronize()
    exp = [t.grad.clone() for t in leaves]
    for t, gg in zip(leaves, grads_g):
        t.grad = gg
    msgs = []
    if not torch.equal(og, oe): msgs.append("out")
    for n, a, b in zip(names, got, exp):
        if not torch.equal(a, b):
            msgs.append(f"{n} (max diff {float((a.float() - b.float()).abs().max()):.3e})")
    if msgs:
        nbad += 1
        if nbad <= 5: print(f"replay {it}: mismatch in {msgs}")
print(f"{nbad} of 50 replays differ from eager")
