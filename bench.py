#!/usr/bin/env python
"""bench.py -- training throughput of ViL with the MI355X-native longformerhand path.

    python bench.py --gpus 1 --steps K --warmup W                      (single GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W      (one rank per GPU, RCCL)

One "step" = forward + backward + AdamW step of the named ViL on one synthetic
ImageNet-shape batch already resident in HBM (bf16 autocast, fp32 master weights).
W untimed warm-up steps, then exactly K steps bracketed by barrier +
torch.cuda.synchronize(); the elapsed time is the MAX over ranks; rank 0 prints ONE
JSON line.  `value` = images of all ranks / that time (weak scaling: fixed per-GPU batch).

Extra objects on the same line:
  roofline      the dominant hot-path kernel over the timed region: per-launch durations
                come from hipEvents the library records around each of its launches on
                the launch stream (vil_attn_profile_begin/_end); achieved = sum of the
                launches' ALGORITHMIC bytes / sum of their durations.
  kernels       the same statistics for every hot-path kernel.
  cpu_baseline  the oracle (CPU restatement of the reference path) inside the same host
                model, timed on this box's host cores on a bounded sample (rank 0, N=1).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy rate)
MFMA_BF16_PEAK_TFLOPS = 2500.0
F32_VALU_PEAK_TFLOPS = 157.3


def cpu_baseline(config, seconds):
    """Bounded CPU sample: the same host model with every hot-path layer computed by the
    oracle (fp32, B=2), fwd + bwd + AdamW, on this box's host cores."""
    from oracle.cpu_model import build_cpu_baseline_model
    from vision_longformer_amd.engine import CONFIGS, make_optimizer, SyntheticBatches, train_step
    cores = len(os.sched_getaffinity(0))
    threads = min(cores, 64)
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    img = CONFIGS[config][1]
    B = 2
    model = build_cpu_baseline_model(config).train()
    opt = make_optimizer(model)
    data = SyntheticBatches(B, img, torch.device("cpu"))
    train_step(model, opt, *data.next(), amp_dtype=None)           # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        train_step(model, opt, *data.next(), amp_dtype=None)
        n += 1
        el = time.perf_counter() - t0
        if el >= seconds or n >= 50:
            break
    cpu = "?"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": round(B * n / el, 3), "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"{n} train steps (fwd+bwd+AdamW) of {config}, batch {B}, fp32, oracle hot path, "
                      f"{threads} threads of {cores} logical cores, {cpu}, {el:.1f} s"}


def kernel_stats(recs):
    agg = {}
    for name, ms, by, fl in recs:
        a = agg.setdefault(name, dict(launches=0, ms=0.0, bytes=0.0, flops=0.0))
        a["launches"] += 1
        a["ms"] += ms
        a["bytes"] += by
        a["flops"] += fl
    out = {}
    for name, a in agg.items():
        t = a["ms"] * 1e-3
        out[name] = {"launches": a["launches"], "avg_ms": round(a["ms"] / a["launches"], 5),
                     "total_ms": round(a["ms"], 3),
                     "GBps": round(a["bytes"] / t / 1e9, 1) if t > 0 else 0.0,
                     "TFLOPs": round(a["flops"] / t / 1e12, 2) if t > 0 else 0.0}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="vil_small_224")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the config's)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="auto", choices=["auto", "scalar", "mfma"])
    ap.add_argument("--master-weights", default="on", choices=["on", "off"],
                    help="bf16 working weights + fp32 master (engine.MasterWeightAdamW) instead of per-call autocast casts")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="run the step as hipGraphs (auto = on; random-shift neighbours are device-side words refreshed per replay); "
                         "off = eager step under DDP (bucketed all-reduce overlapped with backward)")
    args = ap.parse_args()

    from vision_longformer_amd import _lib, ops
    from vision_longformer_amd.engine import (CONFIGS, init_distributed, build_vil, make_optimizer, wrap_ddp,
                                             SyntheticBatches, train_step, GraphedTrainStep, MasterWeightAdamW)
    rank, local_rank, world, device = init_distributed()
    if device.type != "cuda":
        raise SystemExit("bench.py needs a GPU: the product path is HIP-only (no CPU fallback)")
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    _lib.lib()                                  # fail loudly if the extension is missing
    ops.DEFAULT_BACKEND = args.backend
    fam, img, cfg_batch, f1, f2, mode = CONFIGS[args.config]
    B = args.batch or cfg_batch
    torch.manual_seed(0)
    model = build_vil(args.config).to(device).train()
    use_graph = args.graph in ("on", "auto")
    use_master = args.master_weights == "on"
    opt = MasterWeightAdamW(model, capturable=use_graph) if use_master else make_optimizer(model, capturable=use_graph)
    data = SyntheticBatches(B, img, device, rank)
    if use_graph:
        def agree(ok):                   # every rank takes the same branch (a lone eager rank would dead-lock the others)
            if world > 1:
                flag = torch.tensor([ok], device=device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                ok = int(flag.item())
            return ok
        ok = 1
        try:
            if world > 1:      # same initial weights on every rank (what DDP's constructor would do)
                for p_ in model.parameters():
                    dist.broadcast(p_.data, 0)
                if use_master:
                    for m_ in opt.master:
                        dist.broadcast(m_, 0)
            gstep = GraphedTrainStep(model, opt, *data.next(), world=world)
        except Exception as exc:          # capture refused: every rank falls back to the eager DDP step together
            print(f"[rank {rank}] hipGraph capture failed ({exc!r}); falling back to the eager step", file=sys.stderr)
            ok = 0
        ok = agree(ok)
        if ok:
            # pre-flight: a replayed step must behave like a training step (finite, sane loss) -- on this stack
            # hipGraph replay mis-orders hipMemsetAsync nodes, which broke PyTorch's multi-block reductions
            # before every such reduction was moved onto the library's kernels
            pre = [float(gstep(*data.next())) for _ in range(4)]
            good = all(v == v and 0.0 < v < 30.0 for v in pre)
            if not good:
                print(f"[rank {rank}] graph replay pre-flight losses {pre}; falling back to the eager step", file=sys.stderr)
            ok = agree(int(good))
        if not ok:
            use_graph = False
            torch.cuda.synchronize()
            torch.manual_seed(0)
            model = build_vil(args.config).to(device).train()
            opt = MasterWeightAdamW(model) if use_master else make_optimizer(model)
    if use_graph:
        step_fn = lambda xb, tb: gstep(xb, tb)
    else:
        ddp = wrap_ddp(model, device, world)
        step_fn = lambda xb, tb: train_step(ddp, opt, xb, tb)

    for _ in range(args.warmup):
        step_fn(*data.next())
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    cap = max(1024, args.steps * 512)
    if not use_graph:                  # (a replayed graph makes no library calls: its kernels are profiled below)
        _lib.profile_begin(cap)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step_fn(*data.next())
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if use_graph:
        # per-kernel hipEvent timing needs the library's own launches: run the SAME step eagerly,
        # right after the timed region, on the same weights / shapes (not part of `value`)
        _lib.profile_begin(cap)
        for _ in range(min(args.steps, 5)):
            gstep._body(eager=True)
        torch.cuda.synchronize()
    recs = _lib.profile_end(cap)
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss_val = float(loss.item())

    if rank == 0:
        ks = kernel_stats(recs)
        nprof = min(args.steps, 5) if use_graph else args.steps
        hot_ms = sum(k["total_ms"] for k in ks.values())
        dom = max(ks, key=lambda n: ks[n]["total_ms"]) if ks else None
        roofline = None
        if dom:
            k = ks[dom]
            tot_b = sum(r[2] for r in recs if r[0] == dom)
            tot_f = sum(r[3] for r in recs if r[0] == dom)
            ai = tot_f / tot_b if tot_b else 0.0
            # the fused kernels sit just below the bf16 ridge (2.5 PF / 8 TB/s = 312 F/B) -> HBM roof
            traffic = None
            try:       # PMC HBM bytes per launch, collected offline with rocprofv3 --pmc (profiles/)
                pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
                if pm["config"] == args.config and pm["per_gpu_batch"] == B and dom in pm["kernels"]:
                    traffic = round(pm["kernels"][dom]["hbm_bytes_per_launch"])
            except (OSError, KeyError, ValueError):
                pass
            roofline = {"kernel": dom, "bound": "hbm", "achieved": k["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(k["GBps"] / HBM_PEAK_GBS, 4), "traffic": traffic,
                        "algorithmic_bytes_per_launch": round(tot_b / k["launches"]),
                        "avg_launch_ms": k["avg_ms"], "launches": k["launches"],
                        "arith_intensity_flop_per_byte": round(ai, 1),
                        "achieved_tflops": k["TFLOPs"],
                        "frac_of_bf16_mfma_peak": round(k["TFLOPs"] / MFMA_BF16_PEAK_TFLOPS, 4),
                        "share_of_hot_path_time": round(k["total_ms"] / hot_ms, 3) if hot_ms else None}
        out = {
            "metric": "images/sec (train) ViL-Small@224" if args.config == "vil_small_224"
                      else f"images/sec (train) {args.config}",
            "value": round(B * world * args.steps / elapsed, 2), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{args.config}: ViL ({fam}) ATTN_TYPE=longformerhand rpe, {img}x{img}, "
                                   f"windows f{f1}/f{f2}, train step fwd+bwd+AdamW, random-init weights",
                       "global_batch": B * world, "per_gpu_batch": B, "parallelism": f"dp{world}",
                       "backend": args.backend, "random_shift_mode": mode,
                       "precision": "bf16 compute, fp32 master weights + fp32 AdamW state"
                                    + (" (bf16 working copy, foreach refresh)" if use_master else " (autocast casts)"),
                       "launch": "hipGraph replay (fwd+bwd" + ("+AdamW)" if world == 1 else "), flat-gradient RCCL all-reduce, "
                                                                     "hipGraph replay (AdamW)")
                                 if use_graph else "eager (DDP bucketed all-reduce)"},
            "roofline": roofline,
            "hot_path_ms_per_step": round(hot_ms / nprof, 3),
            "kernels": ks,
            "final_loss": round(loss_val, 4),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.config, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
