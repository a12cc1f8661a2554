"""Host-side cost of one fused attention call (eager): tiny shapes so the GPU is never the bottleneck."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vision_longformer_amd.longformer2d import Long2DSCSelfAttention
from vision_longformer_amd.msvit import Attention
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = Long2DSCSelfAttention(64, num_heads=2, qkv_bias=True, w=4, sharew=True, nglo=1, rpe=True).to(dev).bfloat16().train()
x = torch.randn(2, 1 + 64, 64, device=dev, dtype=torch.bfloat16, requires_grad=True)
def step():
    y = m(x, 8, 8)
    y.backward(y)
for _ in range(20): step()
torch.cuda.synchronize()
n = 300
t = time.perf_counter()
for _ in range(n): step()
t1 = time.perf_counter() - t
torch.cuda.synchronize()
print(f"Long2DSCSelfAttention fwd+bwd host time: {t1 / n * 1e6:.1f} us per call (incl. 3 Linear fwd+bwd)")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(100): step()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(18)

# ---- break-down of the backward callbacks
import vision_longformer_amd.ops as ops, vision_longformer_amd.linear as lin
acc = {}
def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter(); r = fn(*a, **k); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0; return r
    return w
ops._full_bwd = timed("_full_bwd", ops._full_bwd)
ops._full_fwd = timed("_full_fwd", ops._full_fwd)
lin._wgrad = timed("_wgrad", lin._wgrad)
lin._colsum = timed("_colsum", lin._colsum)
ob = lin._SplitKLinearFn.backward
lin._SplitKLinearFn.backward = staticmethod(timed("linear.backward", ob))
fb = ops._VilFullAttention.backward
ops._VilFullAttention.backward = staticmethod(timed("attn.backward", fb))
for _ in range(200): step()
torch.cuda.synchronize()
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"{k:20s} {v / 200 * 1e6:8.1f} us per step")
