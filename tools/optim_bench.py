"""Times the optimizer step alone on the parameter set of a ViL configuration (GPU): the HIP multi-tensor kernel
(bf16 gradients in, fp32 master + state + bf16 working copy out) against torch.optim.AdamW(fused) + the two foreach
copy passes it replaced.

    python tools/optim_bench.py [vil_small_224]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vision_longformer_amd.engine import build_vil, MasterWeightOptimizer, recipe_of


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(n):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / n


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "vil_small_224"
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    for kind in ("adamw", "qhm"):
        model = build_vil(cfg).to(dev)
        opt = MasterWeightOptimizer(model, kind=kind)
        for p in opt.low:
            p.grad = torch.randn_like(p) * 1e-3
        for p in opt.direct:
            p.grad = torch.randn_like(p) * 1e-3
        n = sum(p.numel() for p in opt.low) + sum(p.numel() for p in opt.direct)
        ms = timeit(opt.step)
        by = n * ((28 if kind == "adamw" else 20))
        print(f"{cfg} {kind}: {n/1e6:.1f} M parameters, HIP multi-tensor step {ms*1e3:.1f} us, {by/ms/1e6:.0f} GB/s of ~{by/1e6:.0f} MB")
        if kind == "adamw":
            masters = [m.detach().clone().requires_grad_(True) for m in opt.master]
            for m in masters:
                m.grad = torch.zeros_like(m)
            topt = torch.optim.AdamW(masters + opt.direct, lr=1e-3, fused=True)

            @torch.no_grad()
            def old():
                torch._foreach_copy_([m.grad for m in masters], [p.grad for p in opt.low])
                topt.step()
                torch._foreach_copy_(opt.low, masters)
            print(f"   torch fused AdamW + 2 foreach copies: {timeit(old)*1e3:.1f} us")


if __name__ == "__main__":
    main()
