"""The reference's operator micro-benchmark protocol (src/tests/benchmark_vil.py:171-224, the data behind
docs/attn_time.png / attn_memory.png): forward + backward wall time (host clock around a synchronised call, mean of the
repetitions after the first 10) and peak allocated memory of the sliding-chunk attention on an n x n feature map,
B=2, H=12, M=64, W=8, seed 300, q/k/v ~ N(0,1), no bias, no global token, `context.sum().backward()`.

Rows printed per n (GPU):
  fused bf16     the product path: ops.vil_local_attention, MFMA kernels, bf16 I/O (no score tensor exists)
  fused fp32     the same op with fp32 I/O on the fp32 matrix-core family (v_mfma_f32_16x16x4_f32)
  fused fp32 valu  ... on the one-query-per-lane VALU family (the fp32 path of rounds 1-2)
  operator fp32  the reference's own pipeline on the operator-level HIP surface (slidingchunk_2d ->
                 mask_invalid_locations -> softmax -> slidingchunk_2d), which materialises the score tensor like the
                 reference's `scwbackward` method does
and, read off the reference's figure (unstated CUDA GPU, fp32), the published `scwbackward` curve.

    python tools/op_benchmark.py [--sizes 48 96 144 192 240 288] [--reps 30]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from einops import rearrange  # noqa: E402
import torch.nn.functional as F  # noqa: E402

PUBLISHED_S = {48: 0.03, 96: 0.07, 144: 0.135, 192: 0.235, 240: 0.39, 288: 0.685}   # docs/attn_time.png, seconds


def run(method, n, reps, B=2, H=12, M=64, W=8):
    from vision_longformer_amd.ops import vil_local_attention
    from vision_longformer_amd.slidingchunk_2d import slidingchunk_2d, mask_invalid_locations
    dev = torch.device("cuda:0")
    N = n * n
    cost = []
    torch.cuda.reset_peak_memory_stats()
    for i in range(reps):
        query = torch.randn(B * H * N * M, device=dev).view(B, H, N, M).requires_grad_(True)
        key = torch.randn(B * H * N * M, device=dev).flip(dims=(0,)).view(B, H, N, M).requires_grad_(True)
        value = torch.randn(B * H * N * M, device=dev).view(B, H, N, M).requires_grad_(True)
        if method.startswith("fused"):
            dt = torch.bfloat16 if method == "fused bf16" else torch.float32
            be = {"fused bf16": "mfma", "fused fp32": "mfma", "fused fp32 valu": "scalar"}[method]
            # the product's layout: (B, N, H*M) token-major projections (the layout change is setup, not the op)
            q = query.detach().transpose(1, 2).reshape(B, N, H * M).to(dt).requires_grad_(True)
            kv = torch.cat([key.detach().transpose(1, 2).reshape(B, N, H * M),
                            value.detach().transpose(1, 2).reshape(B, N, H * M)], -1).to(dt).requires_grad_(True)
        torch.cuda.synchronize()
        t0 = time.time()
        if method.startswith("fused"):
            ctx = vil_local_attention(q, kv, None, None, nx=n, ny=n, w=W, nglo=0, num_heads=H, mode=0, exact=0, scale=1.0,
                                      backend=be)
        else:
            q_img, k_img, v_img = (rearrange(t, "b h (x y) c -> (b h) c x y", x=n) for t in (query, key, value))
            pad = (W - n % W) % W
            m = (n + pad) // W
            q_img, k_img, v_img = (rearrange(F.pad(t, (0, pad, 0, pad)), "b c (m x) (n y) -> b c m n (x y)", x=W, y=W)
                                   for t in (q_img, k_img, v_img))
            a = slidingchunk_2d(q_img, k_img, False)
            mask_invalid_locations(a, m, m, pad, pad, W, exact=0)
            ctx = slidingchunk_2d(torch.softmax(a, dim=-1), v_img, True)
            ctx = rearrange(ctx, "b c m n (x y) -> b (m x) (n y) c", x=W)[:, :n, :n].reshape(B, H, N, M)
        ctx.sum().backward()
        torch.cuda.synchronize()
        cost.append(time.time() - t0)
    skip = min(10, reps // 3)
    kept = sorted(cost[skip:])
    # median of the timed repetitions: the protocol allocates fresh operands per repetition, and one caching-allocator
    # refill among 20 (cudaMalloc of ~350 MB tensors at 240^2) doubled the MEAN of an otherwise 2 ms op
    return kept[len(kept) // 2] * 1e3, torch.cuda.max_memory_allocated() / 2 ** 20


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", type=int, nargs="*", default=[48, 96, 144, 192, 240, 288])
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--methods", nargs="*", default=["fused bf16", "fused fp32", "fused fp32 valu", "operator fp32"])
    a = ap.parse_args()
    torch.manual_seed(300)
    rows = []
    for n in a.sizes:
        row = {"n": n, "published_scwbackward_ms": PUBLISHED_S.get(n, 0) * 1e3 or None}
        for mth in a.methods:
            if mth == "operator fp32" and n > 192:
                continue                     # the materialised score tensor: 2*12*(n/8)^2*64*576*4 B (7.6 GB at 288^2) x several
            ms, mb = run(mth, n, a.reps if mth != "operator fp32" else max(6, a.reps // 5))
            row[mth] = {"fwd_bwd_ms": round(ms, 3), "peak_MB": round(mb, 1)}
        rows.append(row)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
