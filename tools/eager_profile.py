"""Host-side profile of the eager training step (cProfile): where the Python time of `--graph off` goes."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vision_longformer_amd.engine import build_vil, MasterWeightAdamW, SyntheticBatches, train_step
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = build_vil("vil_small_224").to(dev).train()
opt = MasterWeightAdamW(model)
data = SyntheticBatches(128, 224, dev, 0)
for _ in range(5): train_step(model, opt, *data.next())
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10): train_step(model, opt, *data.next())
t_issue = (time.perf_counter() - t) / 10
torch.cuda.synchronize()
t_all = (time.perf_counter() - t) / 10
print(f"eager step: host issue time {t_issue * 1e3:.2f} ms, wall {t_all * 1e3:.2f} ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(5): train_step(model, opt, *data.next())
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
