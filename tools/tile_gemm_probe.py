"""The plain tile GEMM (vil_gemm_tile_bf16: the loader-wave kernels of csrc/vil_gemm_fused.hip without their activation
epilogues) against the tuned hipBLASLt GEMM at the projections of the dense stages, with an fp64 check on sampled rows.
    python tools/tile_gemm_probe.py        (us per launch: forward with bias and input gradient)"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vision_longformer_amd import _lib   # noqa: E402
from vision_longformer_amd.linear import _gemm   # noqa: E402

SHAPES = [("small s3 qkv", 25216, 384, 1152), ("small s3 proj", 25216, 384, 384), ("small s3 fc1", 25216, 384, 1536), ("small s3 fc2", 25216, 1536, 384),
          ("small s4 qkv", 6400, 768, 2304), ("small s4 proj", 6400, 768, 768), ("small s4 fc1", 6400, 768, 3072), ("small s4 fc2", 6400, 3072, 768),
          ("meddeep s3 qkv", 18464, 384, 1152), ("meddeep s3 proj", 18464, 384, 384), ("meddeep s3 fc2", 18464, 1536, 384),
          ("s2 qkv", 100480, 192, 576), ("s2 fc2", 100480, 768, 192)]


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


def tile(op, x, w, bias, out):
    vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    rc = _lib.lib().vil_gemm_tile_bf16(op, vp(x), vp(w), vp(bias), vp(out), x.shape[0], x.shape[1], out.shape[1], x.stride(0), out.stride(0),
                                       ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
    return rc


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    for name, T, K, N in SHAPES:
        x = torch.randn(T, K, generator=g).bfloat16().to(dev)
        w = (torch.randn(N, K, generator=g) * 0.05).bfloat16().to(dev)
        b = torch.randn(N, generator=g).bfloat16().to(dev)
        dy = torch.randn(T, N, generator=g).bfloat16().to(dev)
        out_f = torch.empty(T, N, dtype=torch.bfloat16, device=dev)
        out_b = torch.empty(T, K, dtype=torch.bfloat16, device=dev)
        rows = torch.randint(0, T, (64,), generator=g)
        line = f"{name:16s} T={T:6d} {K:5d}->{N:5d}:"
        for label, op, a, bb, o in (("fwd", 0, x, b, out_f), ("dgrad", 1, dy, None, out_b)):
            rc = tile(op, a, w, bb, o)
            if rc != 0:
                line += f"  {label} tile: rc {rc}"
                continue
            torch.cuda.synchronize()
            want = (a[rows].double() @ (w.double().t() if op == 0 else w.double())) + (bb.double() if bb is not None else 0)
            err = (o[rows].double() - want).abs().max().item() / max(1.0, want.abs().max().item())
            t_tile = timed(lambda: tile(op, a, w, bb, o))
            t_lib = timed(lambda: _gemm(op, a, w, bb))
            line += f"  {label} tile {t_tile:6.1f} us | hipBLASLt {t_lib:6.1f} us (err {err:.1e})"
        print(line, flush=True)


if __name__ == "__main__":
    main()
