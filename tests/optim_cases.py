"""Seeded optimizer cases shared by tools/gen_golden_optim.py (reference -> fixtures), the CPU oracle test and the GPU
parity test.  Inputs are regenerated from the seed; tests/golden/optim_reference.npz holds the reference's OUTPUTS."""
import torch

SHAPES = [(1000,), (37,), (64, 65), (4099,), (3, 5, 7), (8192,), (1,)]     # vector bodies, scalar tails, odd sizes
NSTEPS = 5
CASES = {
    # name: (kind, hyper-parameters, per-group weight decay [group 0: even tensors, group 1: odd tensors])
    "adamw_recipe": ("adamw", dict(lr=5e-4, betas=(0.9, 0.999), eps=1e-8, correct_bias=True), (0.05, 0.0)),
    "adamw_defaults": ("adamw", dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-6, correct_bias=True), (0.01, 0.01)),
    "adamw_nobias": ("adamw", dict(lr=2e-3, betas=(0.8, 0.95), eps=1e-6, correct_bias=False), (0.0, 0.1)),
    "qhm_recipe": ("qhm", dict(lr=0.01, momentum=0.9, qhm_nu=1.0), (0.0, 0.0)),
    "qhm_nu07_wd": ("qhm", dict(lr=0.05, momentum=0.9, qhm_nu=0.7), (1e-4, 0.0)),
    "qhm_plain_sgd": ("qhm", dict(lr=0.1, momentum=0.0, qhm_nu=1.0), (1e-3, 1e-3)),
}


def make_inputs(seed=300):
    """initial parameters and NSTEPS gradients per tensor (fp32; bf16-representable gradients so that the same values
    can be fed as bf16 on the GPU)"""
    g = torch.Generator(device="cpu").manual_seed(seed)
    params = [torch.randn(s, generator=g) * 0.1 for s in SHAPES]
    grads = [[(torch.randn(s, generator=g) * (10.0 ** (k - 2))).bfloat16().float() for s in SHAPES] for k in range(NSTEPS)]
    return params, grads


def run_case(make_opt, name):
    """steps the optimizer built by make_opt(kind, groups, hyper) on the seeded inputs; returns the parameters after
    every step: list over steps of list of tensors"""
    kind, hyper, wds = CASES[name]
    params, grads = make_inputs()
    ps = [torch.nn.Parameter(p.clone()) for p in params]
    groups = [{"params": ps[0::2], "weight_decay": wds[0]}, {"params": ps[1::2], "weight_decay": wds[1]}]
    opt = make_opt(kind, groups, hyper)
    out = []
    for k in range(NSTEPS):
        for p, g in zip(ps, grads[k]):
            p.grad = g.clone()
        opt.step()
        out.append([p.detach().clone() for p in ps])
    return out
