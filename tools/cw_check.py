"""Round 6: the chunk-workgroup kernels (backend "mfma") against the wave-per-chunk kernels (backend "mfma_wave") on the
same inputs, one process: max |difference| of every output and per-kernel hipEvent times of both (the library's sink).

    python tools/cw_check.py [shape,shape,...] [--reps 10] [--bwd] [--full]

Shapes are tools/kernel_bench.py's names, or H,M,W,nx,ny,G,mode,B tuples.  --full runs vil_full_attention (global rows
ride in the passes), --bwd adds the backward."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vision_longformer_amd import _lib
if os.environ.get("VIL_ATTN_LIB"):          # an A/B / ablation build (tools/ab/)
    _lib.use_library_for_ab(os.environ["VIL_ATTN_LIB"])
from vision_longformer_amd.ops import vil_local_attention, vil_full_attention
from kernel_bench import SHAPES


def inputs(shape, full, dev, exact=0):
    H, M, W, nx, ny, G, mode, B = shape
    g = torch.Generator(device="cpu").manual_seed(300)
    C = H * M
    nq = nx * ny + (G if full else 0)
    q = torch.randn(B, nq, C, generator=g).to(dev, torch.bfloat16)
    kv = torch.randn(B, G + nx * ny, 2 * C, generator=g).to(dev, torch.bfloat16)
    table = (torch.randn((4 * W - 1) ** 2, H, generator=g) * 0.5).to(dev)
    g2l = (torch.randn(2 if full else 1, H, G, generator=g) * 0.5).to(dev)
    if not full:
        g2l = g2l[0]
    g2g = (torch.randn(H, G, G, generator=g) * 0.5).to(dev)
    dout = torch.randn(B, nq, C, generator=g).to(dev, torch.bfloat16)
    return q, kv, table, g2l, g2g, dout


def run(shape, backend, full, bwd, reps, exact=0):
    H, M, W, nx, ny, G, mode, B = shape
    dev = torch.device("cuda:0")
    q, kv, table, g2l, g2g, dout = inputs(shape, full, dev)
    leaves = [q, kv, table, g2l] + ([g2g] if full else [])
    if bwd:
        for t in leaves:
            t.requires_grad_(True)
    kw = dict(nx=nx, ny=ny, w=W, nglo=G, num_heads=H, mode=mode, exact=exact, backend=backend)

    def step():
        for t in leaves:
            t.grad = None
        if full:
            out = vil_full_attention(q, kv, table, g2l, g2g, **kw)
        else:
            out = vil_local_attention(q, kv, table, g2l if G else None, **kw)
        if bwd:
            out.backward(dout)
        return out
    out = step()
    torch.cuda.synchronize()
    res = [out.detach().float()] + ([t.grad.detach().float() for t in leaves if t.grad is not None] if bwd else [])
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    _lib.profile_begin(reps * 24)
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    recs = _lib.profile_end(reps * 24)
    agg = {}
    for n, ms, by, fl in recs:
        x = agg.setdefault(n, [0, 0.0]); x[0] += 1; x[1] += ms
    return res, {n: round(1e3 * x[1] / x[0], 1) for n, x in agg.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("shapes", nargs="?", default="small_s1,small_s2,meddeep_s1_f8,basedeep_s1_f6_rs")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--bwd", action="store_true")
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--cw", default="", help="launch shapes to try: streams,shape_code[,ablation bits][;...] (vil_attn_cw_set_shape: code = chunks per workgroup + 10 * query tiles per wave + 100 * heads per workgroup)")
    a = ap.parse_args()
    names = ["out", "dq", "dkv", "dtable", "dg2l", "dg2g"]
    for name in a.shapes.split(","):
        shape = list(SHAPES[name]) if name in SHAPES else [int(v) for v in name.split(":")]
        if a.batch:
            shape[7] = a.batch
        full = a.full and shape[5] >= 1
        r_old, t_old = run(shape, "mfma_wave", full, a.bwd, a.reps)
        for cw in (a.cw.split(";") if a.cw else [""]):
            if cw:
                v = [int(x) for x in cw.split(",")]
                _lib.check(_lib.lib().vil_attn_cw_set_shape(v[0], v[1] if len(v) > 1 else 0))
                if len(v) > 2:          # timing ablation bits (a -DVIL_CW_ABLATE build loaded through VIL_ATTN_LIB)
                    _lib.lib().vil_attn_cw_set_ablation(v[2])
            r_new, t_new = run(shape, "mfma_cw", full, a.bwd, a.reps)
            report(name, shape, full, cw, r_new, r_old, t_new, t_old)


def report(name, shape, full, cw, r_new, r_old, t_new, t_old):
    names = ["out", "dq", "dkv", "dtable", "dg2l", "dg2g"]
    if True:
        diff = {}
        for nm, x, y in zip(names, r_new, r_old):
            d = (x - y).abs()
            diff[nm] = dict(max=float(d.max()), rms_ref=float(y.pow(2).mean().sqrt()), finite=bool(torch.isfinite(x).all()))
        print(json.dumps({"shape": name, "dims": shape, "full": full, "cw": cw, "diff_vs_wave": diff, "us_cw": t_new, "us_wave": t_old}), flush=True)


if __name__ == "__main__":
    main()
