"""Dense-stage attention alone: the dedicated kernel family (csrc/vil_attn_dense.hip) against the one-chunk case of the
sliding-chunk kernels, forward and backward, hipEvent times over the op (all launches of a call).
    python tools/dense_bench.py [--iters 30]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vision_longformer_amd.ops import vil_dense_attention   # noqa: E402
from vision_longformer_amd import _lib   # noqa: E402
if os.environ.get("VIL_ATTN_LIB"):          # A/B against a library built with other switches (tools/ab/)
    _lib.use_library_for_ab(os.environ["VIL_ATTN_LIB"])

SHAPES = [("small s3 14x14", 14, 1, 6, 128), ("small s4 7x7", 7, 0, 12, 128), ("meddeep s3 24x24", 24, 1, 6, 32),
          ("meddeep s4 12x12", 12, 0, 12, 32)]


def timed(fn, iters):
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


def kernel_times(fn, iters):
    """{kernel: avg us} from the library's own hipEvents around each launch"""
    fn(); torch.cuda.synchronize()
    _lib.profile_begin(iters * 16)
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    acc = {}
    for name, ms, by, fl, tag in _lib.profile_end_tagged(iters * 16):
        a = acc.setdefault(name, [0, 0.0]); a[0] += 1; a[1] += ms
    return {k: v[1] / iters * 1e3 for k, v in acc.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--dense-only", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    for name, nx, G, H, B in SHAPES:
        M = 64
        N, C = G + nx * nx, H * M
        g = torch.Generator().manual_seed(1)
        qkv = torch.randn(B, N, 3 * C, generator=g).to(dev, dt).requires_grad_(True)
        dout = torch.randn(B, N, C, generator=g).to(dev, dt)
        tab = (torch.randn((2 * nx - 1) ** 2, H, generator=g) * 0.3).to(dev).requires_grad_(True)
        g2l = (torch.randn(2, H, G, generator=g) * 0.3).to(dev).requires_grad_(True) if G else None
        g2g = (torch.randn(H, G, G, generator=g) * 0.3).to(dev).requires_grad_(True) if G else None
        row = {}
        for be in (("dense",) if args.dense_only else ("mfma", "dense")):
            try:
                f = lambda: vil_dense_attention(qkv, tab, g2l, g2g, nx=nx, ny=nx, nglo=G, num_heads=H, backend=be)
                with torch.no_grad():
                    tf = timed(f, args.iters)
                out = f()
                tb = timed(lambda: torch.autograd.grad(out, [qkv, tab] + ([g2l, g2g] if G else []), dout, retain_graph=True),
                           args.iters)
                row[be] = (tf, tb)
                with torch.no_grad():
                    kf = kernel_times(f, 10)
                kb = kernel_times(lambda: torch.autograd.grad(out, [qkv, tab] + ([g2l, g2g] if G else []), dout, retain_graph=True), 10)
                print(f"   {be:6s} kernels fwd: " + ", ".join(f"{k} {v:.1f}" for k, v in kf.items()) + f" = {sum(kf.values()):.1f} us")
                print(f"   {be:6s} kernels bwd: " + ", ".join(f"{k} {v:.1f}" for k, v in kb.items()) + f" = {sum(kb.values()):.1f} us")
            except RuntimeError as e:
                row[be] = (float("nan"), float("nan"))
                print("  ", be, "failed:", str(e)[:100])
        hbm_f = B * 4 * N * C * 2 / 8e12 * 1e6
        hbm_b = B * 8 * N * C * 2 / 8e12 * 1e6
        row.setdefault("mfma", (float("nan"), float("nan")))
        print(f"{name:18s} B{B} H{H} N{N}: fwd one-chunk {row['mfma'][0]:7.1f} us  dense {row['dense'][0]:7.1f} us  (HBM floor {hbm_f:5.1f}) |"
              f" bwd one-chunk {row['mfma'][1]:7.1f} us  dense {row['dense'][1]:7.1f} us  (HBM floor {hbm_b:5.1f})", flush=True)


if __name__ == "__main__":
    main()
