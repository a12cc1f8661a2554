"""GPU (-m gpu), collected LAST: the training engine (hipGraph step, multi-rank flavour) on top of the kernels."""
import math
import os
import subprocess

import numpy as np
import pytest
import torch

import golden_cases as GC
from gpu_common import (ROOT, report, case, cid, make_inputs, run_oracle, run_hip, compare, rms,
                        F32_TOL, BF16_TOL, LOW_TOL, SMALL)
from oracle import vil_oracle as O

pytestmark = pytest.mark.gpu


def _reset(opt):
    """optimizer state back to "never stepped" IN PLACE (moments and the device-side step count): a captured step keeps
    reading the same tensors"""
    if hasattr(opt, "reset_state"):
        opt.reset_state()
        return
    for st in opt.state.values():
        for v in st.values():
            if torch.is_tensor(v):
                v.zero_()
    opt.reset_step_count()


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "the gpu-marked tests need a GPU"
    return torch.device("cuda:0")


# ---------------------------------------------------------------- hipGraph training step
@pytest.mark.parametrize("master", [False, True])
def test_graphed_train_step_equals_eager(dev, master):
    """The captured fwd+bwd+AdamW graph must walk the same trajectory as the eager step
    (plain fused AdamW, and bf16 working weights + fp32 master AdamW)."""
    from vision_longformer_amd.engine import make_optimizer, train_step, GraphedTrainStep, MasterWeightAdamW
    from vision_longformer_amd.msvit import MsViT
    arch = "l1,h1,d32,n1,s1,g1,p4,f4,a0_l2,h2,d64,n2,s1,g1,p2,f4,a0_l3,h2,d64,n1,s0,g1,p2,f7,a0"
    g = torch.Generator().manual_seed(5)
    xs = [torch.randn(8, 3, 64, 64, generator=g).to(dev) for _ in range(3)]
    ts = [torch.softmax(torch.randn(8, 10, generator=g), -1).to(dev) for _ in range(3)]

    def run(graphed):
        torch.manual_seed(0)
        m = MsViT(arch, img_size=64, num_classes=10, drop_path_rate=0.0, norm_embed=True, sharew=True).to(dev).train()
        opt = MasterWeightAdamW(m, lr=1e-3, capturable=graphed) if master else make_optimizer(m, lr=1e-3, capturable=graphed)
        losses = []
        if graphed:
            sd = {k: v.clone() for k, v in m.state_dict().items()}
            msd = [mm.clone() for mm in opt.master] if master else []
            gs = GraphedTrainStep(m, opt, xs[0], ts[0], warmup=2)
            with torch.no_grad():                       # undo the warm-up updates (in place: the graph holds the buffers)
                for k, v in m.state_dict().items():
                    v.copy_(sd[k])
                for mm, v in zip(opt.master if master else [], msd):
                    mm.copy_(v)
            _reset(opt)
            for x, t in zip(xs, ts):
                losses.append(float(gs(x, t)))
        else:
            for x, t in zip(xs, ts):
                losses.append(float(train_step(m, opt, x, t)))
        torch.cuda.synchronize()
        return losses, torch.cat([p.detach().float().reshape(-1) for p in m.parameters()]).cpu()

    le, pe = run(False)
    lg, pg = run(True)
    report(f"     graph-vs-eager losses {le} {lg}  max|dparam| {float((pe - pg).abs().max()):.3e}")
    assert max(abs(a - b) for a, b in zip(le, lg)) < 2e-2
    assert float((pe - pg).abs().max()) < 5e-3


def test_graphed_train_step_vil_small_shapes(dev):
    """hipGraph replay of the real ViL-Small step (batch 32, no DropPath) must follow the eager trajectory.
    Regression test: with PyTorch's multi-block bias-gradient reductions or hipMemsetAsync nodes inside the
    capture, replay produced NaN gradients on this stack from the second step on."""
    from vision_longformer_amd.engine import build_vil, MasterWeightAdamW, SyntheticBatches, train_step, GraphedTrainStep
    B, steps = 32, 5

    def run(graphed):
        torch.manual_seed(0)
        model = build_vil("vil_small_224", drop_path_rate=0.0).to(dev).train()
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
            _reset(opt)
            data = SyntheticBatches(B, 224, dev, 0)
            for _ in range(steps):
                losses.append(float(gs(*data.next())))
        else:
            for _ in range(steps):
                losses.append(float(train_step(model, opt, *data.next())))
        return losses

    le, lg = run(False), run(True)
    report(f"     ViL-Small graph-vs-eager losses {[round(v, 3) for v in le]} {[round(v, 3) for v in lg]}")
    assert all(math.isfinite(v) for v in lg)
    # bf16 training from the same state: the trajectories separate slowly (atomics order), not by O(1)
    assert abs(le[0] - lg[0]) < 1e-2 and max(abs(a - b) for a, b in zip(le, lg)) < 0.5


def test_graphed_train_step_multi_rank_path(dev, monkeypatch):
    """The world > 1 flavour of the graphed step (graph A: fwd+bwd+pack into flat buffers, all-reduce, graph B:
    AdamW) on one GPU with the collective stubbed out (the mean over one rank is the identity): must follow the
    eager trajectory exactly like the single-graph flavour."""
    import vision_longformer_amd.engine as E
    calls = []
    monkeypatch.setattr(E.dist, "all_reduce", lambda t, op=None, async_op=False: calls.append(t.numel()))
    arch = "l1,h1,d32,n1,s1,g1,p4,f4,a0_l2,h2,d64,n2,s1,g1,p2,f4,a0_l3,h2,d64,n1,s0,g1,p2,f7,a0"
    g = torch.Generator().manual_seed(5)
    xs = [torch.randn(8, 3, 64, 64, generator=g).to(dev) for _ in range(3)]
    ts = [torch.softmax(torch.randn(8, 16, generator=g), -1).to(dev) for _ in range(3)]

    def run(graphed):
        torch.manual_seed(0)
        m = E.MsViT(arch, img_size=64, num_classes=16, drop_path_rate=0.0, norm_embed=True, sharew=True).to(dev).train()
        opt = E.MasterWeightAdamW(m, lr=1e-3, capturable=graphed)
        losses = []
        if graphed:
            sd = {k: v.clone() for k, v in m.state_dict().items()}
            msd = [mm.clone() for mm in opt.master]
            gs = E.GraphedTrainStep(m, opt, xs[0], ts[0], world=2, warmup=2)
            assert gs.opt_graph is not None and len(gs.graphs) == 3 and len(gs.flats) >= 3      # three backward segments
            with torch.no_grad():
                for k, v in m.state_dict().items():
                    v.copy_(sd[k])
                for mm, v in zip(opt.master, msd):
                    mm.copy_(v)
            _reset(opt)
            n0 = len(calls)
            for x, t in zip(xs, ts):
                losses.append(float(gs(x, t)))
            assert len(calls) - n0 == 3 * len(gs.flats)          # one collective per flat buffer and step
        else:
            for x, t in zip(xs, ts):
                losses.append(float(E.train_step(m, opt, x, t)))
        torch.cuda.synchronize()
        return losses, torch.cat([p.detach().float().reshape(-1) for p in m.parameters()]).cpu()

    le, pe = run(False)
    lg, pg = run(True)
    report(f"     graph(world>1 path)-vs-eager losses {le} {lg}  max|dparam| {float((pe - pg).abs().max()):.3e}")
    assert max(abs(a - b) for a, b in zip(le, lg)) < 2e-2
    assert float((pe - pg).abs().max()) < 2e-2


def test_graphed_step_two_processes_one_gpu(dev, tmp_path):
    """The DEFAULT multi-GPU path of bench.py (segment graphs -> asynchronous flat-gradient all-reduce per segment ->
    optimizer graph) with a REAL process group: two processes share cuda:0 (VIL_SHARE_DEVICE=1, gloo), each trains on
    its half of the batch through the HIP kernels for three steps; the ranks must end bit-identical, and equal (bf16
    tolerance) to one process training eagerly on the concatenated batch."""
    import socket
    import sys
    import ddp_graph_worker as W
    from vision_longformer_amd.engine import MasterWeightAdamW, train_step
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    out = str(tmp_path / "rank0.pt")
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), VIL_SHARE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "ddp_graph_worker.py"), out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p_ in procs:
        try:
            o, _ = p_.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    assert all(p_.returncode == 0 for p_ in procs), "\n".join(logs)[-4000:]
    got = torch.load(out)
    assert got["same"], "ranks diverged"
    assert got["ngraphs"] == 3 and got["comm"]["exposed_bytes"] < got["comm"]["total_bytes"]
    # single process, eager, whole batch
    m = W.build(dev)
    opt = MasterWeightAdamW(m, lr=1e-3)
    xs, ts = W.batches(dev)
    le = [float(train_step(m, opt, x, t)) for x, t in zip(xs, ts)]
    torch.cuda.synchronize()
    pe = torch.cat([p.detach().float().reshape(-1) for p in m.parameters()]).cpu()
    dl = max(abs(a - float(b)) for a, b in zip(le, got["losses"]))
    dp = float((pe - got["params"]).abs().max())
    report(f"     2-process graphed step (shared GPU, gloo) vs 1-process eager: max|dloss| {dl:.3e} max|dparam| {dp:.3e}; comm {got['comm']['segments_bytes']}")
    assert dl < 2e-2 and dp < 2e-2


def test_segmented_step_on_rccl_single_rank(dev):
    """RCCL under the N > 1 code path before any multi-GPU run: a WORLD_SIZE=1 process group with backend "nccl" (= RCCL)
    and a 120 s collective timeout; three segment graphs with a real asynchronous all_reduce(AVG) of each segment's flat
    gradient buffers on the process group's stream between the replays, optimizer graph; five steps must reproduce the
    single-graph step (same kernels, same reduction order: equal up to the float atomics of two bias gradients)."""
    import json
    import socket
    import sys
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0", VIL_DIST_TIMEOUT_S="120")
    env.pop("VIL_SHARE_DEVICE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_graph_worker.py")], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RCCL_WORKER ")][-1]
    got = json.loads(line[len("RCCL_WORKER "):])
    assert got["backend"] == "nccl" and got["ngraphs"] == 3 and got["has_opt_graph"]
    assert got["comm"]["exposed_bytes"] < got["comm"]["total_bytes"]
    dl = max(abs(a - b) for a, b in zip(got["losses_single"], got["losses_segmented"]))
    report(f"     segmented step on RCCL (world 1) vs single graph: max|dloss| {dl:.3e} max|dparam| {got['max_dparam']:.3e}; "
           f"comm {got['comm']['segments_bytes']}")
    assert dl < 1e-3 and got["max_dparam"] < 1e-3


def test_graphed_step_observes_lr_schedule(dev):
    """A captured optimizer step must follow a per-iteration learning-rate schedule (the reference's warm-up + cosine,
    src/engine.py): with capturable=True the lr is a device tensor that engine.set_lr updates in place.  fp32 step (no
    bf16 noise): graphed and eager under the same schedule must agree closely, and the schedule must matter."""
    from vision_longformer_amd.engine import make_optimizer, train_step, GraphedTrainStep, set_lr
    from vision_longformer_amd.msvit import MsViT
    arch = "l1,h1,d32,n1,s1,g1,p4,f4,a0_l2,h2,d64,n1,s1,g1,p2,f4,a0_l3,h2,d64,n1,s0,g1,p2,f7,a0"
    g = torch.Generator().manual_seed(7)
    xs = [torch.randn(8, 3, 64, 64, generator=g).to(dev) for _ in range(4)]
    ts = [torch.softmax(torch.randn(8, 10, generator=g), -1).to(dev) for _ in range(4)]
    lrs = [1e-3, 4e-3, 2e-2, 5e-4]

    def run(graphed, schedule):
        torch.manual_seed(0)
        m = MsViT(arch, img_size=64, num_classes=10, drop_path_rate=0.0, norm_embed=True, sharew=True).to(dev).train()
        opt = make_optimizer(m, lr=lrs[0], capturable=graphed)
        if graphed:
            sd = {k: v.clone() for k, v in m.state_dict().items()}
            gs = GraphedTrainStep(m, opt, xs[0], ts[0], warmup=2, amp_dtype=None)
            with torch.no_grad():
                for k, v in m.state_dict().items():
                    v.copy_(sd[k])
            _reset(opt)
        for i, (x, t) in enumerate(zip(xs, ts)):
            if schedule:
                set_lr(opt, lrs[i])
            gs(x, t) if graphed else train_step(m, opt, x, t, amp_dtype=None)
        torch.cuda.synchronize()
        return torch.cat([p.detach().float().reshape(-1) for p in m.parameters()]).cpu()

    pe, pg, pg_const = run(False, True), run(True, True), run(True, False)
    d_sched = float((pe - pg).abs().max())
    d_const = float((pg - pg_const).abs().max())
    report(f"     lr schedule under hipGraph (fp32 step): |graph - eager| {d_sched:.3e}; |scheduled - constant lr| {d_const:.3e}")
    assert d_sched < 2e-3
    assert d_const > 1e-2 and d_const > 5 * d_sched          # the captured step really read the new lr


def test_graphed_step_follows_python_float_lr_assignments(dev):
    """a torch LR scheduler assigns param_group["lr"] = float between iterations (the reference's
    pattern).  With a Python-float lr (capturable=False) the captured optimizer launch reads a device mirror of it;
    GraphedTrainStep refreshes the mirror before every replay -- no engine.set_lr call here."""
    from vision_longformer_amd.engine import make_optimizer, train_step, GraphedTrainStep
    from vision_longformer_amd.msvit import MsViT
    arch = "l1,h1,d32,n1,s1,g1,p4,f4,a0_l2,h2,d64,n1,s1,g1,p2,f4,a0_l3,h2,d64,n1,s0,g1,p2,f7,a0"
    g = torch.Generator().manual_seed(7)
    xs = [torch.randn(8, 3, 64, 64, generator=g).to(dev) for _ in range(4)]
    ts = [torch.softmax(torch.randn(8, 10, generator=g), -1).to(dev) for _ in range(4)]
    lrs = [1e-3, 4e-3, 2e-2, 5e-4]

    def run(graphed, schedule):
        torch.manual_seed(0)
        m = MsViT(arch, img_size=64, num_classes=10, drop_path_rate=0.0, norm_embed=True, sharew=True).to(dev).train()
        opt = make_optimizer(m, lr=lrs[0], capturable=False)          # Python-float lr in the param groups
        assert not torch.is_tensor(opt.param_groups[0]["lr"])
        if graphed:
            sd = {k: v.clone() for k, v in m.state_dict().items()}
            gs = GraphedTrainStep(m, opt, xs[0], ts[0], warmup=2, amp_dtype=None)
            with torch.no_grad():
                for k, v in m.state_dict().items():
                    v.copy_(sd[k])
            _reset(opt)
        for i, (x, t) in enumerate(zip(xs, ts)):
            if schedule:
                for gr in opt.param_groups:
                    gr["lr"] = lrs[i]                                 # what torch.optim.lr_scheduler does
            gs(x, t) if graphed else train_step(m, opt, x, t, amp_dtype=None)
        torch.cuda.synchronize()
        return torch.cat([p.detach().float().reshape(-1) for p in m.parameters()]).cpu()

    pe, pg, pg_const = run(False, True), run(True, True), run(True, False)
    d_sched = float((pe - pg).abs().max())
    d_const = float((pg - pg_const).abs().max())
    report(f"     float-lr schedule under hipGraph (fp32 step): |graph - eager| {d_sched:.3e}; |scheduled - constant lr| {d_const:.3e}")
    assert d_sched < 2e-3
    assert d_const > 1e-2 and d_const > 5 * d_sched


def test_master_weight_step_captured_without_warmup(dev):
    """allocate() must also create the state of fp32 MASTERS (requires_grad=False; their gradient is
    the bound bf16 copy's).  GraphedTrainStep(warmup=0) with MasterWeightAdamW used to raise 'optimizer state would be
    created inside a stream capture'.  The un-warmed graph must train like the eager step."""
    from vision_longformer_amd.engine import MasterWeightAdamW, train_step, GraphedTrainStep
    from vision_longformer_amd.msvit import MsViT
    from vision_longformer_amd import linear
    arch = "l1,h1,d32,n1,s1,g1,p4,f4,a0_l2,h2,d64,n1,s1,g1,p2,f4,a0_l3,h2,d64,n1,s0,g1,p2,f7,a0"
    g = torch.Generator().manual_seed(11)
    xs = [torch.randn(8, 3, 64, 64, generator=g).to(dev) for _ in range(3)]
    ts = [torch.softmax(torch.randn(8, 10, generator=g), -1).to(dev) for _ in range(3)]

    def run(graphed):
        torch.manual_seed(0)
        m = MsViT(arch, img_size=64, num_classes=10, drop_path_rate=0.0, norm_embed=True, sharew=True).to(dev).train()
        opt = MasterWeightAdamW(m, lr=1e-3, capturable=graphed)
        if graphed:
            gs = GraphedTrainStep(m, opt, xs[0], ts[0], warmup=0)
            inner = opt.opt
            assert all("exp_avg" in inner.state[mp_] for mp_ in opt.master)
        losses = [float(gs(x, t) if graphed else train_step(m, opt, x, t)) for x, t in zip(xs, ts)]
        torch.cuda.synchronize()
        return losses, torch.cat([p.detach().float().reshape(-1) for p in m.parameters()]).cpu()

    was = linear._DETERMINISTIC_PLANS
    linear.deterministic_plans(True)           # (no timing-based plan selection: warmup=0 never reaches a tuning call anyway)
    try:
        le, pe = run(False)
        lg, pg = run(True)
    finally:
        linear.deterministic_plans(was)
    dl = max(abs(a - b) for a, b in zip(le, lg))
    dp = float((pe - pg).abs().max())
    report(f"     MasterWeightAdamW captured with warmup=0 vs eager: max|dloss| {dl:.3e} max|dparam| {dp:.3e}")
    assert dl < 2e-2 and dp < 2e-2


def test_graphed_step_makes_differently_seeded_replicas_identical(dev, tmp_path):
    """Review r04 item 7: replica consistency is the engine's job.  Two ranks (sharing cuda:0 over gloo) build their
    models from DIFFERENT seeds and construct GraphedTrainStep(world=2) with no broadcast of their own; after three
    steps their parameters must be bit-identical (rank 0's initial weights, the same averaged gradients)."""
    import socket
    import sys
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    out = str(tmp_path / "rank0.pt")
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), VIL_SHARE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", VIL_TEST_SEED_PER_RANK="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "ddp_graph_worker.py"), out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p_ in procs:
        try:
            o, _ = p_.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    assert all(p_.returncode == 0 for p_ in procs), "\n".join(logs)[-4000:]
    got = torch.load(out)
    assert got["seeded_per_rank"] and got["differed_before"], "the ranks did not start from different weights"
    assert got["same"], "ranks diverged although the engine synchronised them"
    report(f"     differently seeded replicas after GraphedTrainStep(world=2) + 3 steps: identical = {got['same']}")


def test_fp16_grad_scaler_step_graphed_equals_eager(dev):
    """The reference's AMP recipe (src/engine.py:84-100, src/run_experiment.py:206: fp16 autocast + GradScaler) in the
    build's own loop: engine.train_step(scaler=...) eagerly and GraphedTrainStep(scaler=...) as ONE hipGraph -- loss
    scaling, the non-finite check over the fp16 gradients of the working copies, the device-side skip inside the HIP
    optimizer launch and the scale update are all captured.  Same trajectory both ways; an overflowing loss scale (2^24:
    fp16 gradients overflow) must skip steps and back the scale off, under replay, without a host decision."""
    from vision_longformer_amd.engine import MasterWeightOptimizer, train_step, GraphedTrainStep
    from vision_longformer_amd.msvit import MsViT
    arch = "l1,h1,d32,n1,s1,g1,p4,f4,a0_l2,h2,d64,n1,s1,g1,p2,f4,a0_l3,h2,d64,n1,s0,g1,p2,f7,a0"
    g = torch.Generator().manual_seed(13)
    xs = [torch.randn(8, 3, 64, 64, generator=g).to(dev) for _ in range(4)]
    ts = [torch.softmax(torch.randn(8, 10, generator=g), -1).to(dev) for _ in range(4)]

    def run(graphed, init_scale):
        torch.manual_seed(0)
        m = MsViT(arch, img_size=64, num_classes=10, drop_path_rate=0.0, norm_embed=True, sharew=True).to(dev).train()
        opt = MasterWeightOptimizer(m, kind="adamw", low_dtype=torch.float16, lr=1e-3, capturable=graphed)
        scaler = torch.amp.GradScaler("cuda", init_scale=init_scale, growth_interval=1000)
        if graphed:
            sd = {k: v.clone() for k, v in m.state_dict().items()}
            msd = [mm.clone() for mm in opt.master]
            gs = GraphedTrainStep(m, opt, xs[0], ts[0], warmup=1, amp_dtype=torch.float16, scaler=scaler)
            with torch.no_grad():
                for k, v in m.state_dict().items():
                    v.copy_(sd[k])
                for mm, v in zip(opt.master, msd):
                    mm.copy_(v)
            opt.reset_state()
            scaler._scale.fill_(init_scale)
            scaler._growth_tracker.zero_()
        losses = [float(gs(x, t) if graphed else train_step(m, opt, x, t, amp_dtype=torch.float16, scaler=scaler))
                  for x, t in zip(xs, ts)]
        torch.cuda.synchronize()
        return losses, torch.cat([mm.detach().float().reshape(-1) for mm in opt.master]).cpu(), float(scaler.get_scale())

    le, pe, se = run(False, 1024.0)
    lg, pg, sg = run(True, 1024.0)
    dl = max(abs(a - b) for a, b in zip(le, lg))
    dp = float((pe - pg).abs().max())
    report(f"     fp16 + GradScaler: graphed vs eager max|dloss| {dl:.3e} max|dmaster| {dp:.3e}; scale {se} / {sg}")
    assert all(v == v and 0.0 < v < 30.0 for v in le + lg)
    assert dl < 2e-2 and dp < 2e-2 and se == sg == 1024.0
    # a loss scale that overflows fp16 gradients: steps are skipped on the device and the scale backs off, replay after replay
    lo, po, so = run(True, 2.0 ** 24)
    assert so < 2.0 ** 24, "the scale never backed off under graph replay"
    assert all(v == v for v in lo)
