"""Diagnostic: per-step loss / gradient norm of the bench workload (eager), to tell optimisation
divergence from a numerical fault.  usage: python tools/diverge_check.py [lr] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vision_longformer_amd.engine import build_vil, MasterWeightAdamW, SyntheticBatches, soft_target_cross_entropy
lr = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = build_vil("vil_small_224").to(dev).train()
opt = MasterWeightAdamW(model, lr=lr)
data = SyntheticBatches(128, 224, dev, 0)
for st in range(steps):
    x, t = data.next()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(x)
        loss = soft_target_cross_entropy(out, t)
    opt.zero_grad()
    loss.backward()
    gn = torch.sqrt(sum((p.grad.float() ** 2).sum() for p in model.parameters() if p.grad is not None))
    bad = [n for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    print(f"step {st:3d} loss {float(loss):10.4f} |logits|max {float(out.abs().max()):9.2f} gradnorm {float(gn):10.3e} nonfinite grads: {bad[:4]}")
    if bad or not torch.isfinite(loss):
        break
    opt.step()
