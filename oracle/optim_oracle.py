"""CPU oracle of the reference's optimizers (AdamW with the weight-decay fix, QHM).

TEST / MEASUREMENT INFRASTRUCTURE ONLY (see oracle/vil_oracle.py): tests, smoke() and bench.py's cpu_baseline leg.
The product's optimizer is the HIP multi-tensor kernel behind vision_longformer_amd.optim and has no CPU path.

Restated as pure functions on (parameter, gradient, state) tensors, fp32 torch CPU ops in the reference's operation
order, plus thin torch.optim.Optimizer shells so that CPU host-logic tests (DDP over gloo, checkpoint round trips)
and the CPU baseline can step a model with the reference's update rules.  Pinned by tools/gen_golden.py: the imported
reference optimizers (src/optim/optimization.py:111-193, src/optim/qhm.py:8-130) and these functions produce
bit-identical tensors on the seeded cases frozen in tests/golden/optim_reference.npz.
"""
import math

import torch
from torch.optim import Optimizer


def adamw_update(p, g, exp_avg, exp_avg_sq, step, lr, beta1=0.9, beta2=0.999, eps=1e-6, weight_decay=0.0,
                 correct_bias=True):
    """One AdamW step in place; `step` is the 1-based step count of THIS update (optimization.py:150-191)."""
    exp_avg.mul_(beta1).add_(g, alpha=1.0 - beta1)
    exp_avg_sq.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    denom = exp_avg_sq.sqrt().add_(eps)                       # eps outside the bias correction
    step_size = lr
    if correct_bias:
        step_size = step_size * math.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
    p.addcdiv_(exp_avg, denom, value=-step_size)
    if weight_decay > 0.0:                                    # decoupled decay, after the Adam update
        p.add_(p, alpha=-lr * weight_decay)


def qhm_update(p, g, momentum_buffer, lr, momentum=0.0, qhm_nu=1.0, weight_decay=0.0):
    """One QHM step in place (qhm.py:72-130); `g` is modified like the reference modifies p.grad.  `momentum_buffer`
    may be None when momentum == 0 or nu == 0 (plain SGD)."""
    if weight_decay > 0:
        g.add_(p, alpha=weight_decay)
    if abs(momentum) < 1e-12 or abs(qhm_nu) < 1e-12:
        d = g
    else:
        momentum_buffer.mul_(momentum).add_(g, alpha=1 - momentum)
        if abs(qhm_nu - 1) < 1e-12:
            d = momentum_buffer
        else:
            d = g.mul_(1 - qhm_nu).add_(momentum_buffer, alpha=qhm_nu)
    p.add_(d, alpha=-lr)


class AdamW(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, correct_bias=correct_bias))

    @torch.no_grad()
    def step(self, closure=None):
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"], st["exp_avg"], st["exp_avg_sq"] = 0, torch.zeros_like(p), torch.zeros_like(p)
                st["step"] += 1
                adamw_update(p, p.grad, st["exp_avg"], st["exp_avg_sq"], st["step"], float(group["lr"]),
                             group["betas"][0], group["betas"][1], group["eps"], group["weight_decay"],
                             group["correct_bias"])


class QHM(Optimizer):
    def __init__(self, params, lr=-1, momentum=0, qhm_nu=1, weight_decay=0):
        super().__init__(params, dict(lr=lr, momentum=momentum, qhm_nu=qhm_nu, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        for group in self.param_groups:
            plain = abs(group["momentum"]) < 1e-12 or abs(group["qhm_nu"]) < 1e-12
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not plain and "momentum_buffer" not in st:
                    st["momentum_buffer"] = torch.zeros_like(p)
                qhm_update(p, p.grad, st.get("momentum_buffer"), float(group["lr"]), group["momentum"],
                           group["qhm_nu"], group["weight_decay"])
