"""Worker of tests/test_gpu_3_engine.py::test_segmented_step_on_rccl_single_rank: ONE process, WORLD_SIZE=1, a real RCCL
process group (backend "nccl") with a collective timeout.  Runs the product's multi-GPU training step --
GraphedTrainStep(force_segments=True): three segment graphs, `all_reduce(AVG, async_op=True)` of every segment's flat
gradient buffers on the process group's stream between the replays, optimizer graph -- for five steps and compares it
with the single-graph step of an identically initialised model.  Prints one JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import ddp_graph_worker as W
    from vision_longformer_amd.engine import init_distributed, MasterWeightAdamW, GraphedTrainStep
    rank, local_rank, world, dev = init_distributed(single_rank_group=True)
    assert world == 1 and dev.type == "cuda" and dist.is_initialized() and dist.get_backend() == "nccl"
    xs, ts = W.batches(dev)
    xs, ts = xs + xs[:2], ts + ts[:2]                   # five steps

    def run(force):
        m = W.build(dev)
        opt = MasterWeightAdamW(m, lr=1e-3, capturable=True)
        sd = {k: v.clone() for k, v in m.state_dict().items()}
        msd = [mm.clone() for mm in opt.master]
        gs = GraphedTrainStep(m, opt, xs[0], ts[0], world=1, warmup=2, force_segments=force)
        with torch.no_grad():                           # undo the warm-up updates in place
            for k, v in m.state_dict().items():
                v.copy_(sd[k])
            for mm, v in zip(opt.master, msd):
                mm.copy_(v)
        opt.reset_state()
        losses = [float(gs(x, t)) for x, t in zip(xs, ts)]
        torch.cuda.synchronize()
        flat = torch.cat([p.detach().float().reshape(-1) for p in m.parameters()]).cpu()
        return losses, flat, gs

    l1, p1, _ = run(False)
    l3, p3, gs = run(True)
    out = {"backend": dist.get_backend(), "ngraphs": len(gs.graphs), "has_opt_graph": gs.opt_graph is not None,
           "losses_single": l1, "losses_segmented": l3, "max_dparam": float((p1 - p3).abs().max()),
           "comm": gs.comm_summary()}
    print("RCCL_WORKER " + json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
