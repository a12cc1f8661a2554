"""Diagnostic: are multi-block torch reductions (which zero a semaphore buffer with hipMemsetAsync) and the
library's own memset nodes reproduced faithfully by hipGraph replay on this stack?"""
import torch
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (R, C) in [(1600, 2304), (1600, 3072), (6400, 768), (25216, 384), (1600, 768)]:
    x = torch.randn(R, C, device=dev).bfloat16()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            y = x.sum(0)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        # some unrelated work before and after, as in a training step
        a = x.float() * 2.0
        y = x.sum(0)
        z = x.float().sum(0)
        b = a + 1.0
    bad = 0
    for it in range(200):
        x.copy_(torch.randn(R, C, device=dev))
        g.replay()
        torch.cuda.synchronize()
        ref = x.float().sum(0)
        e1 = (y.float() - ref).abs().max().item()
        e2 = (z - ref).abs().max().item()
        if not (e1 < 2.0 and e2 < 1e-2) or not torch.isfinite(y).all():
            bad += 1
            if bad <= 3:
                print(f"  ({R},{C}) replay {it}: bf16 sum err {e1:.3e}, fp32 sum err {e2:.3e}, finite {bool(torch.isfinite(y).all())}")
    print(f"({R},{C}): {bad} bad replays of 200")
