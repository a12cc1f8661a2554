"""Diagnostic: per-step losses of the eager and the graphed step from the same initial state.
usage: python tools/graph_check.py [drop_path] [steps] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vision_longformer_amd.engine import (build_vil, MasterWeightAdamW, SyntheticBatches, train_step, GraphedTrainStep)
dp = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
dev = torch.device("cuda:0")

def run(graphed):
    torch.manual_seed(0)
    model = build_vil("vil_small_224", drop_path_rate=dp).to(dev).train()
    opt = MasterWeightAdamW(model, lr=1e-3, capturable=graphed)
    data = SyntheticBatches(B, 224, dev, 0)
    losses = []
    if graphed:
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        msd = [m.clone() for m in opt.master]
        gs = GraphedTrainStep(model, opt, *data.next(), warmup=2)
        with torch.no_grad():
            for k, v in model.state_dict().items():
                v.copy_(sd[k])
            for m, v in zip(opt.master, msd):
                m.copy_(v)
        for st in opt.opt.state.values():
            for k, v in st.items():
                if torch.is_tensor(v):
                    v.zero_()
        data = SyntheticBatches(B, 224, dev, 0)
        for _ in range(steps):
            losses.append(float(gs(*data.next())))
    else:
        for _ in range(steps):
            losses.append(float(train_step(model, opt, *data.next())))
    return losses

le = run(False)
lg = run(True)
for i, (a, b) in enumerate(zip(le, lg)):
    print(f"step {i:3d} eager {a:9.4f} graph {b:9.4f}")
