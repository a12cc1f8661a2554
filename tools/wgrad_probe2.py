"""GPU micro-benchmark of vil_linear_wgrad on every (tokens, C_out, C_in) of a ViL-Small step: us per call (both
kernels), GB/s of the algorithmic bytes, TFLOP/s.  VIL_WGRAD_WGS=<n> (read once at first use) changes the planner's
workgroup target."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vision_longformer_amd.linear import _wgrad

dev = torch.device("cuda:0")
T1, T2, T3, T4 = 128 * 3137, 128 * 785, 128 * 197, 128 * 49
shapes = [("s1 qkv?", T1, 192, 96), ("s1 proj", T1, 96, 96), ("s1 fc1", T1, 384, 96), ("s1 fc2", T1, 96, 384),
          ("s2 kv", T2, 384, 192), ("s2 proj", T2, 192, 192), ("s2 fc1", T2, 768, 192), ("s2 fc2", T2, 192, 768),
          ("s3 qkv", T3, 1152, 384), ("s3 proj", T3, 384, 384), ("s3 fc1", T3, 1536, 384), ("s3 fc2", T3, 384, 1536),
          ("s4 qkv", T4, 2304, 768), ("s4 proj", T4, 768, 768), ("s4 fc1", T4, 3072, 768), ("s4 fc2", T4, 768, 3072)]
tot = 0.0
for name, T, co, ci in shapes:
    x = torch.randn(T, ci, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(T, co, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        _wgrad(dy, x, True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 30
    e0.record()
    for _ in range(n):
        _wgrad(dy, x, True)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    by = 2.0 * T * (co + ci); fl = 2.0 * T * co * ci
    tot += us
    print(f"{name:8s} T={T:6d} co={co:4d} ci={ci:4d}  {us:7.1f} us  {by/us/1e3:7.1f} GB/s  {fl/us/1e6:7.1f} TF/s", flush=True)
print(f"sum {tot:.1f} us  (VIL_WGRAD_WGS={os.environ.get('VIL_WGRAD_WGS','default')})")
