"""Per-kernel times of ONE full sliding-chunk attention layer (local + global rows, forward + backward through
vil_attn_bwd_full: what the model's layers launch) at the BASELINE shapes, from the library's hipEvent sink.
One process per library; tools/attn_ab.sh alternates libraries on one box.

    [VIL_ATTN_LIB=tools/ab/libvilattn_<name>.so] python tools/attn_ab.py small_s1[,meddeep_s1_f7,...] [--reps 10]
"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vision_longformer_amd import _lib
if os.environ.get("VIL_ATTN_LIB"):
    _lib.use_library_for_ab(os.environ["VIL_ATTN_LIB"])
from vision_longformer_amd.ops import vil_full_attention

SHAPES = {  # H, M, W, nx, ny, G, mode, B
    "small_s1": (3, 32, 7, 56, 56, 1, 0, 128),
    "small_s2": (3, 64, 7, 28, 28, 1, 0, 128),
    "meddeep_s1_f7": (3, 32, 7, 96, 96, 1, 0, 32),
    "meddeep_s2_f7": (3, 64, 7, 48, 48, 1, 0, 32),
    "meddeep_s1_f8": (3, 32, 8, 96, 96, 1, 0, 32),
    "meddeep_s2_f12": (3, 64, 12, 48, 48, 1, 0, 32),
    "basedeep_s1_f6_rs": (3, 32, 6, 96, 96, 1, 3, 32),
    "basedeep_s2_f8_rs": (3, 64, 8, 48, 48, 1, 5, 32),
}


def run(shape, reps):
    H, M, W, nx, ny, G, mode, B = SHAPES[shape]
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(300)
    C = H * M
    q = torch.randn(B, G + nx * ny, C, generator=g).to(dev, torch.bfloat16).requires_grad_(True)
    kv = torch.randn(B, G + nx * ny, 2 * C, generator=g).to(dev, torch.bfloat16).requires_grad_(True)
    table = (torch.randn((4 * W - 1) ** 2, H, generator=g) * 0.02).to(dev).requires_grad_(True)
    g2l = (torch.randn(2, H, G, generator=g) * 0.02).to(dev).requires_grad_(True)
    g2g = (torch.randn(H, G, G, generator=g) * 0.02).to(dev).requires_grad_(True)
    dout = torch.randn(B, G + nx * ny, C, generator=g).to(dev, torch.bfloat16)

    def step():
        out = vil_full_attention(q, kv, table, g2l, g2g, nx=nx, ny=ny, w=W, nglo=G, num_heads=H, mode=mode)
        out.backward(dout)
        return out
    for _ in range(3):
        out = step()
    torch.cuda.synchronize()
    chk = [float(out.float().abs().sum()), float(q.grad.float().abs().sum()), float(kv.grad.float().abs().sum()),
           float(table.grad.abs().sum())]
    _lib.profile_begin(reps * 24)
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    recs = _lib.profile_end(reps * 24)
    agg = {}
    for n, ms, by, fl in recs:
        x = agg.setdefault(n, [0, 0.0]); x[0] += 1; x[1] += ms
    us = {n: round(x[1] / reps * 1e3, 1) for n, x in agg.items()}        # us per layer call (a kernel launched twice counts twice)
    bwd = sum(v for k, v in us.items() if k not in ("k_mfma_fwd", "k_glo_fwd", "k_table")) 
    print(json.dumps({"shape": shape, "lib": os.path.basename(os.environ.get("VIL_ATTN_LIB", "HEAD")), "us": us,
                      "fwd_us": round(us.get("k_mfma_fwd", 0) + us.get("k_glo_fwd", 0), 1), "total_us": round(sum(us.values()), 1),
                      "checksums": [round(c, 3) for c in chk]}))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("shapes")
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    for s in a.shapes.split(","):
        run(s, a.reps)
