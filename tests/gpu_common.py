"""Shared helpers of the GPU (-m gpu) test files: seeded inputs, the oracle and HIP runners, the comparison
with its tolerances, and the parity report written under gpurun_out/.

Tolerances
  fp32 I/O (scalar family): the reference test's own contract
      (src/tests/test_slidingchunk_2d.py:159-166): context atol 1e-4 / rtol 1e-5 is
      stated for unit-variance random data; here out: atol 2e-5 + rtol 1e-4,
      grads: atol 1e-4 + rtol 1e-3.
  bf16 / fp16 I/O: against the fp64 oracle evaluated on the SAME rounded inputs:
      out atol 2e-2 / rtol 5e-2 (the reference's fp16 profiling tolerance is 2e-2 / 1e-1, :167-175);
      every gradient is bounded RELATIVE TO THE REFERENCE TENSOR'S RMS, |err| <= k * rms(ref) + rtol * |ref|:
      q / kv gradients k = 0.10, rtol 5e-2; bias-table / g2l / g2g gradients (sums of rounded dS over up to
      B * Nloc terms; heavy-tailed: corner bins get one term, centre bins hundreds) k = 8e-2, rtol 3e-2.  An absolute bound would pass a wrong-bin bug on small-gradient cases.
"""
import os

import torch

import golden_cases as GC
from oracle import vil_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = os.path.join(ROOT, "gpurun_out", "parity_report.txt")


def report(line):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, "a") as f:
        f.write(line + "\n")


def case(H, M, W, nx, ny, G, mode=0, exact=0, rpe=True, only_glo=False, B=2):
    return dict(H=H, M=M, W=W, nx=nx, ny=ny, G=G, mode=mode, exact=exact, rpe=rpe, only_glo=only_glo, B=B)


def cid(c):
    return "H{H}M{M}W{W}_{nx}x{ny}_G{G}_m{mode}_e{exact}{r}{o}_B{B}".format(
        r="" if c["rpe"] else "_norpe", o="_oglo" if c["only_glo"] else "", **c)


def make_inputs(c, dtype, seed=GC.SEED):
    g = torch.Generator().manual_seed(seed)
    B, H, M, G = c["B"], c["H"], c["M"], c["G"]
    C = H * M
    Nloc = c["nx"] * c["ny"]
    q = torch.randn(B, Nloc, C, generator=g)
    kv = torch.randn(B, G + Nloc, 2 * C, generator=g)
    table = torch.randn((4 * c["W"] - 1) ** 2, H, generator=g) * 0.5 if c["rpe"] else None
    g2l = torch.randn(H, G, generator=g) * 0.5 if (c["rpe"] and G > 0) else None
    dout = torch.randn(B, Nloc, C, generator=g)
    # round to the I/O dtype so that HIP and oracle see identical inputs
    q, kv, dout = (t.to(dtype).float() for t in (q, kv, dout))
    return q, kv, table, g2l, dout


def run_oracle(c, q, kv, table, g2l, dout):
    B, H, M, G = c["B"], c["H"], c["M"], c["G"]
    C = H * M
    q = q.double().requires_grad_(True)
    kv = kv.double().requires_grad_(True)
    tab = table.double().requires_grad_(True) if table is not None else None
    g2 = g2l.double().requires_grad_(True) if g2l is not None else None
    Nloc = q.shape[1]
    qh = q.view(B, Nloc, H, M).transpose(1, 2)
    kvh = kv.view(B, G + Nloc, 2, H, M).permute(2, 0, 3, 1, 4)
    out = O.local_attention(qh, kvh[0], kvh[1], c["nx"], c["ny"], c["W"], G, mode=c["mode"], exact=c["exact"],
                            bias_table=tab, g2l_bias=g2, only_glo=c["only_glo"])
    out = out.transpose(1, 2).reshape(B, Nloc, C)
    (out * dout.double()).sum().backward()
    return dict(out=out.detach(), dq=q.grad, dkv=kv.grad,
                dtable=tab.grad if tab is not None else None, dg2l=g2.grad if g2 is not None else None)


def run_hip(c, q, kv, table, g2l, dout, dtype, backend, dev, debug=0):
    from vision_longformer_amd.ops import vil_local_attention
    qd = q.to(dev, dtype).requires_grad_(True)
    kvd = kv.to(dev, dtype).requires_grad_(True)
    tab = table.to(dev).requires_grad_(True) if table is not None else None
    g2 = g2l.to(dev).requires_grad_(True) if g2l is not None else None
    out = vil_local_attention(qd, kvd, tab, g2, nx=c["nx"], ny=c["ny"], w=c["W"], nglo=c["G"],
                              num_heads=c["H"], mode=c["mode"], exact=c["exact"], only_glo=c["only_glo"],
                              backend=backend, _debug=debug)
    out.backward(dout.to(dev, dtype))
    torch.cuda.synchronize()
    f = lambda t: t.detach().double().cpu() if t is not None else None
    return dict(out=f(out), dq=f(qd.grad), dkv=f(kvd.grad), dtable=f(tab.grad if tab is not None else None),
                dg2l=f(g2.grad if g2 is not None else None))



def rms(t):
    return float(t.double().pow(2).mean().sqrt())


def compare(tag, got, ref, tols):
    """tols[name] = (atol, rtol) or ("rms", k, rtol): |err| <= atol (or k * rms(ref)) + rtol * |ref| elementwise."""
    worst = []
    ok = True
    for k, tol in tols.items():
        if ref.get(k) is None:
            continue
        a, b = got[k], ref[k]
        assert a is not None, f"{tag}: {k} missing"
        assert torch.isfinite(a).all(), f"{tag}: {k} has non-finite values"
        if tol[0] == "rms":
            atol, rtol = tol[1] * max(rms(b), 1e-4), tol[2]      # (floor: an identically-zero reference, e.g. dq of only_glo)
        else:
            atol, rtol = tol
        err = (a - b).abs()
        lim = atol + rtol * b.abs()
        bad = int((err > lim).sum())
        worst.append(f"{k}:{err.max().item():.2e}[{float((err / lim).max()):.2f}]" + (f"(!{bad})" if bad else ""))
        ok &= bad == 0
    report(f"{'ok  ' if ok else 'FAIL'} {tag}  " + " ".join(worst))
    assert ok, f"{tag}: " + " ".join(worst)


F32_TOL = dict(out=(2e-5, 1e-4), dq=(1e-4, 1e-3), dkv=(1e-4, 1e-3), dtable=(5e-4, 1e-3), dg2l=(5e-4, 1e-3))
LOW_TOL = dict(out=(2e-2, 5e-2), dq=("rms", 0.06, 5e-2), dkv=("rms", 0.06, 5e-2), dqkv=("rms", 0.06, 5e-2),
               dtable=("rms", 8e-2, 3e-2), dg2l=("rms", 6e-2, 3e-2), dg2g=("rms", 6e-2, 3e-2))
BF16_TOL = LOW_TOL
# Round 5: tightened to what is observed.  profiles/r04_parity_report.txt (err / bound per tensor over ~410 report
# lines) has the q / kv gradients at <= 0.42 of the round-4 bound (0.10 rms): now 0.06 rms.  d(g2l) / d(g2g) were at
# 0.20 / 0.04 of 0.08 rms: now 0.06.  d(table) stays at 0.08 rms: its worst case sits at 0.79 of it (a 240-entry table
# whose gradient sums ~10^5 bf16-rounded products per bin), so 0.06 would be red without being wrong.

SMALL = [
    case(2, 16, 4, 8, 8, 1), case(2, 16, 4, 8, 8, 1, rpe=False), case(2, 16, 4, 10, 9, 1),
    case(2, 16, 4, 10, 9, 1, exact=1), case(2, 16, 4, 10, 9, 1, exact=-1), case(3, 16, 3, 7, 7, 2),
    case(3, 16, 3, 7, 7, 2, mode=2), case(3, 16, 3, 7, 7, 2, mode=7), case(3, 16, 3, 7, 7, 2, mode=-1),
    case(2, 16, 4, 10, 10, 0), case(2, 16, 4, 8, 8, 1, only_glo=True), case(2, 16, 4, 5, 6, 1, mode=5, exact=-1),
    case(2, 32, 7, 14, 14, 1), case(2, 32, 7, 16, 15, 1, mode=3), case(2, 64, 8, 20, 20, 1, B=1),
    case(1, 48, 7, 15, 14, 1), case(3, 32, 6, 13, 12, 1, mode=1), case(2, 8, 2, 5, 4, 1),
    case(2, 64, 12, 24, 25, 1, B=1), case(2, 32, 7, 9, 30, 3, exact=1), case(2, 32, 7, 7, 7, 1),
    case(1, 16, 4, 3, 2, 1), case(2, 32, 8, 16, 16, 0, mode=8),
]
