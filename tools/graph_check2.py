import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vision_longformer_amd.engine import (build_vil, MasterWeightAdamW, make_optimizer, SyntheticBatches, GraphedTrainStep)
master = int(sys.argv[1]) if len(sys.argv) > 1 else 1
B = 32
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = build_vil("vil_small_224", drop_path_rate=0.0).to(dev).train()
opt = MasterWeightAdamW(model, lr=1e-3, capturable=True) if master else make_optimizer(model, lr=1e-3, capturable=True)
data = SyntheticBatches(B, 224, dev, 0)
gs = GraphedTrainStep(model, opt, *data.next(), warmup=2)
names = {id(p): n for n, p in model.named_parameters()}
for it in range(4):
    loss = float(gs(*data.next()))
    torch.cuda.synchronize()
    badg = [names[id(p)] for p in model.parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    badp = [names[id(p)] for p in model.parameters() if not torch.isfinite(p).all()]
    print(f"iter {it} loss {loss:.4f} nonfinite grads {len(badg)} {badg[:6]} params {len(badp)} {badp[:6]}")
