"""Micro-benchmark of the hot-path op alone (GPU): forward / backward of
vil_local_attention at a BASELINE shape, per-kernel times from the library's
hipEvent profiling sink.  Used under rocprofv3 for PMC counters.

    python tools/kernel_bench.py small_s1[,small_s2,...] [--reps 20] [--fwd-only] [--backend mfma]
"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vision_longformer_amd import _lib
if os.environ.get("VIL_ATTN_LIB"):          # A/B against a library built from another revision (tools/ab/)
    _lib.use_library_for_ab(os.environ["VIL_ATTN_LIB"])
from vision_longformer_amd.ops import vil_local_attention, vil_dense_attention

SHAPES = {  # H, M, W, nx, ny, G, mode, B
    "small_s1": (3, 32, 7, 56, 56, 1, 0, 128),
    "small_s2": (3, 64, 7, 28, 28, 1, 0, 128),
    "tiny_s1": (1, 48, 7, 56, 56, 1, 0, 128),
    "meddeep_s1_f7": (3, 32, 7, 96, 96, 1, 0, 32),
    "meddeep_s2_f7": (3, 64, 7, 48, 48, 1, 0, 32),
    "meddeep_s1_f8": (3, 32, 8, 96, 96, 1, 0, 32),
    "meddeep_s2_f12": (3, 64, 12, 48, 48, 1, 0, 32),
    "basedeep_s1_f6_rs": (3, 32, 6, 96, 96, 1, 3, 32),
    "basedeep_s2_f8_rs": (3, 64, 8, 48, 48, 1, 5, 32),
    # the reference's operator benchmark protocol (tools/op_benchmark.py): B 2, H 12, M 64, W 8, no global tokens
    "opbench_192": (12, 64, 8, 192, 192, 0, 0, 2),
    "opbench_240": (12, 64, 8, 240, 240, 0, 0, 2),
    "opbench_288": (12, 64, 8, 288, 288, 0, 0, 2),
    # dense s0 stages (one chunk = the whole grid, mode -1, (2W-1)^2 table): vil_dense_attention on fused qkv
    "small_s3_dense": (6, 64, 14, 14, 14, 1, -1, 128),
    "small_s4_dense": (12, 64, 7, 7, 7, 1, -1, 128),
    "meddeep_s3_dense": (6, 64, 24, 24, 24, 1, -1, 32),
    "meddeep_s4_dense": (12, 64, 12, 12, 12, 1, -1, 32),
}

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("shape")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--fwd-only", action="store_true")
    ap.add_argument("--backend", default="auto")
    ap.add_argument("--batch", type=int, default=0)
    a = ap.parse_args()
    for shape in a.shape.split(","):
        run(a, shape)


def run(a, shape):
    H, M, W, nx, ny, G, mode, B = SHAPES[shape]
    B = a.batch or B
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(300)
    C = H * M
    q = (torch.randn(B, nx * ny, C, generator=g) ).to(dev, torch.bfloat16).requires_grad_(not a.fwd_only)
    kv = torch.randn(B, G + nx * ny, 2 * C, generator=g).to(dev, torch.bfloat16).requires_grad_(not a.fwd_only)
    table = (torch.randn((4 * W - 1) ** 2, H, generator=g) * 0.02).to(dev).requires_grad_(not a.fwd_only)
    g2l = (torch.randn(H, G, generator=g) * 0.02).to(dev).requires_grad_(not a.fwd_only)
    dout = torch.randn(B, nx * ny, C, generator=g).to(dev, torch.bfloat16)
    kw = dict(nx=nx, ny=ny, w=W, nglo=G, num_heads=H, mode=mode, backend=a.backend)
    dense = shape.endswith("_dense")
    if dense:
        rg = not a.fwd_only
        qkv = torch.randn(B, G + nx * ny, 3 * C, generator=g).to(dev, torch.bfloat16).requires_grad_(rg)
        table = (torch.randn((2 * W - 1) ** 2, H, generator=g) * 0.02).to(dev).requires_grad_(rg)
        g2l2 = (torch.randn(2, H, G, generator=g) * 0.02).to(dev).requires_grad_(rg)
        g2g = (torch.randn(H, G, G, generator=g) * 0.02).to(dev).requires_grad_(rg)
        dout = torch.randn(B, G + nx * ny, C, generator=g).to(dev, torch.bfloat16)
    def step():
        if dense:
            out = vil_dense_attention(qkv, table, g2l2, g2g, nx=nx, ny=ny, nglo=G, num_heads=H, scale=M ** -0.5)
        else:
            out = vil_local_attention(q, kv, table, g2l, **kw)
        if not a.fwd_only:
            out.backward(dout)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    _lib.profile_begin(a.reps * 16)
    for _ in range(a.reps):
        step()
    torch.cuda.synchronize()
    recs = _lib.profile_end(a.reps * 16)
    agg = {}
    for n, ms, by, fl in recs:
        x = agg.setdefault(n, [0, 0.0, 0.0, 0.0]); x[0] += 1; x[1] += ms; x[2] += by; x[3] += fl
    res = {n: dict(avg_ms=round(x[1] / x[0], 5), GBps=round(x[2] / x[1] / 1e6, 1), TFLOPs=round(x[3] / x[1] / 1e9, 2))
           for n, x in agg.items()}
    print(json.dumps({"shape": shape, "B": B, "kernels": res}))

if __name__ == "__main__":
    main()
