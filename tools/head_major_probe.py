"""Round 6: what would a HEAD-MAJOR layout of q / k / v / out (and their gradients) be worth to the attention kernels?
In the reference's layout a head's row is head_dim * 2 bytes inside a (B, N, H * head_dim) token row: 64 bytes at head_dim 32 --
half a cache line -- and the L2 -> CU path moves whole lines (tools/ubench/dma_rate.hip).  The C ABI takes element strides, so
the same kernels run on (B, H, N, head_dim) tensors: consecutive tokens of one head are contiguous.  This probe runs the local
attention (vil_attn_fwd + vil_attn_bwd) on the same values in both layouts and prints per-kernel hipEvent times and the
largest output difference (expected 0: only addresses change).  The product keeps the reference's layout -- the projections
on both sides of the attention read and write token-major rows.

    python tools/head_major_probe.py [small_s1,meddeep_s1_f7,...] [--reps 10] [--backend mfma|mfma_wave]"""
import argparse, ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vision_longformer_amd import _lib, ops
from kernel_bench import SHAPES


def run(shape, layout, backend, reps):
    H, M, W, nx, ny, G, mode, B = shape
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(300)
    C, N = H * M, nx * ny
    dt = torch.bfloat16
    vals = {n: torch.randn(B, (G if n in "kv" else 0) + N, H, M, generator=g).to(dev, dt) for n in ("q", "k", "v", "do")}
    table = (torch.randn((4 * W - 1) ** 2, H, generator=g) * 0.02).to(dev)
    g2l = (torch.randn(H, G, generator=g) * 0.02).to(dev)

    def lay(t):            # (B, T, H, M) values -> storage in the wanted layout; returns (storage, strides (sb, st, sh))
        Bq, T = t.shape[0], t.shape[1]
        if layout == "token":
            s = t.contiguous()
            return s, (T * C, C, M)
        s = t.permute(0, 2, 1, 3).contiguous()          # (B, H, T, M)
        return s, (H * T * M, M, T * M)
    q, sq = lay(vals["q"]); k, sk = lay(vals["k"]); v, sv = lay(vals["v"]); do, sdo = lay(vals["do"])
    out = torch.zeros_like(q); dq = torch.zeros_like(q); dk = torch.zeros_like(k); dv = torch.zeros_like(v)
    lse = torch.zeros(B, H, N, device=dev)
    dtab = torch.zeros_like(table); dg2 = torch.zeros_like(g2l)
    d = _lib.VilAttnDesc()
    d.B, d.H, d.M, d.nx, d.ny, d.W, d.G, d.mode, d.exact = B, H, M, nx, ny, W, G, mode, 0
    d.dtype, d.only_glo, d.backend, d.scale = _lib.DTYPE_BF16, 0, ops._BACKENDS[backend], float(M) ** -0.5
    for pre, s in (("q", sq), ("k", sk), ("v", sv), ("o", sq), ("do", sdo), ("dq", sq), ("dk", sk), ("dv", sv)):
        setattr(d, pre + "_sb", s[0]); setattr(d, pre + "_st", s[1]); setattr(d, pre + "_sh", s[2])
    L = _lib.lib()
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    P = ops._ptr

    def step():
        ws = ops._workspace(d, 0, dev)
        _lib.check(L.vil_attn_fwd(ctypes.byref(d), P(q), P(k), P(v), P(table), P(g2l), P(out), P(lse), P(ws), st))
        ws = ops._workspace(d, 1, dev)
        _lib.check(L.vil_attn_bwd(ctypes.byref(d), P(q), P(k), P(v), P(out), P(do), P(lse), P(table), P(g2l), P(dq), P(dk), P(dv),
                                  P(dtab), P(dg2), P(ws), st))
    for _ in range(3):
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
    us = {n: round(1e3 * x[1] / reps, 1) for n, x in agg.items()}
    back = (lambda t: t) if layout == "token" else (lambda t: t.permute(0, 2, 1, 3))
    res = [back(out).float().cpu(), back(dq).float().cpu(), back(dk).float().cpu(), back(dv).float().cpu(), dtab.float().cpu()]
    return us, res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("shapes", nargs="?", default="small_s1,meddeep_s1_f7,small_s2")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--backend", default="mfma")
    a = ap.parse_args()
    for name in a.shapes.split(","):
        u_t, r_t = run(SHAPES[name], "token", a.backend, a.reps)
        u_h, r_h = run(SHAPES[name], "head", a.backend, a.reps)
        diff = [float((x - y).abs().max()) for x, y in zip(r_t, r_h)]
        print(json.dumps({"shape": name, "backend": a.backend, "us_token_major": u_t, "us_head_major": u_h,
                          "max_diff_out_dq_dk_dv_dtable": diff}), flush=True)


if __name__ == "__main__":
    main()
