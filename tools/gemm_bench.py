"""The hipBLASLt GEMMs of one training step (forward with bias, input gradient) at ViL-Small's shapes, with the algorithm
vil_gemm_tune selected: per-shape microseconds and the sum.  VIL_ATTN_LIB selects an A/B build (tools/ab/)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vision_longformer_amd import _lib   # noqa: E402
if os.environ.get("VIL_ATTN_LIB"):
    _lib.use_library_for_ab(os.environ["VIL_ATTN_LIB"])
from vision_longformer_amd.linear import _gemm   # noqa: E402

# (name, T, K (in), N (out), layers)
SHAPES = [("s1 qkv", 401536, 96, 288, 1), ("s1 proj", 401536, 96, 96, 1), ("s1 fc1", 401536, 96, 384, 1), ("s1 fc2", 401536, 384, 96, 1),
          ("s2 qkv", 100480, 192, 576, 2), ("s2 proj", 100480, 192, 192, 2), ("s2 fc1", 100480, 192, 768, 2), ("s2 fc2", 100480, 768, 192, 2),
          ("s3 qkv", 25216, 384, 1152, 8), ("s3 proj", 25216, 384, 384, 8), ("s3 fc1", 25216, 384, 1536, 8), ("s3 fc2", 25216, 1536, 384, 8),
          ("s4 qkv", 6400, 768, 2304, 1), ("s4 proj", 6400, 768, 768, 1), ("s4 fc1", 6400, 768, 3072, 1), ("s4 fc2", 6400, 3072, 768, 1)]


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    dev = torch.device("cuda:0")
    tot_f = tot_b = 0.0
    for name, T, K, N, L in SHAPES:
        x = torch.randn(T, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        b = torch.randn(N, device=dev).bfloat16()
        dy = torch.randn(T, N, device=dev).bfloat16()
        f = timed(lambda: _gemm(0, x, w, b))
        g = timed(lambda: _gemm(1, dy, w, None))
        tot_f += f * L; tot_b += g * L
        fl = 2.0 * T * K * N
        print(f"{name:8s} T={T:6d} {K:4d}->{N:4d} x{L}: fwd {f:7.1f} us ({fl / f / 1e6:6.0f} TF)  dgrad {g:7.1f} us ({fl / g / 1e6:6.0f} TF)", flush=True)
    print(f"per step: forward {tot_f / 1e3:.3f} ms, input gradients {tot_b / 1e3:.3f} ms, sum {(tot_f + tot_b) / 1e3:.3f} ms")


if __name__ == "__main__":
    main()
