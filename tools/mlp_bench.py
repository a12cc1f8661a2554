"""fc2 input gradient + GELU backward: the fused launch (vil_gemm_dgelu_bf16) against the library GEMM + ATen's
gelu_backward, at the MLP shapes of ViL-Small / Medium-Deep (hipEvents over 20 repetitions)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vision_longformer_amd import linear   # noqa: E402
from vision_longformer_amd.linear import _dgrad_dgelu, _gemm   # noqa: E402
linear._DGELU_FORCE = True

SHAPES = [("small s1", 401536, 96), ("small s2", 100480, 192), ("small s3", 25216, 384), ("small s4", 6400, 768),
          ("meddeep s1", 294944, 96), ("meddeep s3", 18464, 384)]


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
    for name, T, C in SHAPES:
        K, N = C, 4 * C
        dy = torch.randn(T, K, device=dev).bfloat16()
        w = (torch.randn(K, N, device=dev) * 0.05).bfloat16()
        h = torch.randn(T, N, device=dev).bfloat16()
        fused = timed(lambda: _dgrad_dgelu(dy, w, h))
        gemm = timed(lambda: _gemm(1, dy, w, None))
        da = _gemm(1, dy, w, None)
        gelu = timed(lambda: torch.ops.aten.gelu_backward(da, h))
        flops, byts = 2.0 * T * K * N, 2.0 * T * (K + 2 * N)
        print(f"{name:11s} T={T:6d} K={K:4d} N={N:5d}: fused {fused:7.1f} us ({flops / fused / 1e6:6.1f} TF, {byts / fused / 1e3:6.0f} GB/s)"
              f" | library GEMM {gemm:7.1f} + gelu_backward {gelu:7.1f} = {gemm + gelu:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
