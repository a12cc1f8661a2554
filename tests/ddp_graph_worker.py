"""Worker of tests/test_gpu_3_engine.py::test_graphed_step_two_processes_one_gpu: one rank of a world-2 job whose ranks
SHARE cuda:0 (VIL_SHARE_DEVICE=1, collectives over gloo).  Runs the product's multi-rank training step --
GraphedTrainStep(world=2): segment graphs, flat-gradient all-reduce per segment, optimizer graph; the HIP kernels, not
an oracle stand-in -- on this rank's half of a fixed batch and writes rank 0's parameters to argv[1]."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ARCH = "l1,h1,d32,n1,s1,g1,p4,f4,a0_l2,h2,d64,n2,s1,g1,p2,f4,a0_l3,h2,d64,n1,s0,g1,p2,f7,a0"


def batches(dev):
    g = torch.Generator().manual_seed(5)
    xs = [torch.randn(8, 3, 64, 64, generator=g).to(dev) for _ in range(3)]
    ts = [torch.softmax(torch.randn(8, 16, generator=g), -1).to(dev) for _ in range(3)]
    return xs, ts


def build(dev, seed=0):
    from vision_longformer_amd.msvit import MsViT
    torch.manual_seed(seed)
    return MsViT(ARCH, img_size=64, num_classes=16, drop_path_rate=0.0, norm_embed=True, sharew=True).to(dev).train()


def main():
    from vision_longformer_amd.engine import init_distributed, MasterWeightAdamW, GraphedTrainStep
    rank, local_rank, world, dev = init_distributed()
    assert world == 2 and dev.type == "cuda"
    per_rank = os.environ.get("VIL_TEST_SEED_PER_RANK") == "1"    # replicas that start from DIFFERENT weights: the engine must fix that
    m = build(dev, seed=100 + rank if per_rank else 0)
    opt = MasterWeightAdamW(m, lr=1e-3, capturable=True)
    xs, ts = batches(dev)
    half = slice(rank * 4, rank * 4 + 4)
    f0 = torch.cat([p.detach().float().reshape(-1) for p in m.parameters()])
    b0 = [torch.zeros_like(f0) for _ in range(world)]
    dist.all_gather(b0, f0)
    differed_before = not torch.equal(b0[0], b0[1])
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    msd = [mm.clone() for mm in opt.master]
    gs = GraphedTrainStep(m, opt, xs[0][half], ts[0][half], world=2, warmup=2)
    if not per_rank:
        with torch.no_grad():                       # undo the warm-up updates (in place: the graphs hold the buffers)
            for k, v in m.state_dict().items():
                v.copy_(sd[k])
            for mm, v in zip(opt.master, msd):
                mm.copy_(v)
        opt.reset_state()
    losses = [float(gs(x[half], t[half])) for x, t in zip(xs, ts)]
    torch.cuda.synchronize()
    flat = torch.cat([p.detach().float().reshape(-1) for p in m.parameters()])
    both = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    lt = torch.tensor(losses, device=dev)
    lall = [torch.zeros_like(lt) for _ in range(world)]
    dist.all_gather(lall, lt)
    if rank == 0:
        torch.save({"params": flat.cpu(), "same": bool(torch.equal(both[0], both[1])),
                    "seeded_per_rank": per_rank, "differed_before": differed_before,
                    "losses": torch.stack(lall).mean(0).cpu(), "comm": gs.comm_summary(), "ngraphs": len(gs.graphs)}, sys.argv[1])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
