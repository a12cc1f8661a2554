"""GPU micro-benchmark: fused weight/bias-gradient kernel vs the library paths (see tools/gemm_probe.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vision_longformer_amd.linear import _wgrad
dev = torch.device("cuda:0")
def bench(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
T3, T4, T2, T1 = 128 * 197, 128 * 50, 128 * 785, 128 * 3137
shapes = {"s3 qkv": (T3, 384, 1152), "s3 proj": (T3, 384, 384), "s3 fc1": (T3, 384, 1536), "s3 fc2": (T3, 1536, 384),
          "s4 fc1": (T4, 768, 3072), "s4 fc2": (T4, 3072, 768), "s2 fc1": (T2, 192, 768), "s2 fc2": (T2, 768, 192),
          "s2 kv": (T2, 192, 384), "s1 fc1": (T1, 96, 384), "s1 fc2": (T1, 384, 96), "s1 kv": (T1, 96, 192)}
for name, (T, ci, co) in shapes.items():
    x = torch.randn(T, ci, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(T, co, device=dev, dtype=torch.bfloat16)
    t_f = bench(lambda: _wgrad(dy, x, True))
    S = 8; Tp = (T // S) * S
    def splitk():
        return torch.bmm(dy[:Tp].view(S, Tp // S, co).transpose(1, 2), x[:Tp].view(S, Tp // S, ci)).sum(0), dy.sum(0)
    t_sk = bench(splitk)
    fl = 2 * T * ci * co
    print(f"{name:8s} T={T:6d} {ci:4d}->{co:4d}  fused dW+db {t_f:7.1f} us ({fl/t_f/1e6:6.1f} TF)   bmm splitK8 + sum {t_sk:7.1f} us ({fl/t_sk/1e6:6.1f} TF)")
