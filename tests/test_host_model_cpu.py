"""CPU: host-model plumbing that needs no kernel: arch parser, state-dict contract of
the whole MsViT against the reference's key list, reset_vil_mode, parameter groups,
and the N>1 data-parallel path over gloo (world_size 2) with the oracle standing in
for the HIP op (tests may use the oracle; the product path may not)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vision_longformer_amd.engine import build_vil, make_optimizer, param_groups, CONFIGS
from oracle import optim_oracle as OO        # the CPU stand-in for the HIP optimizer kernels (host-logic tests only)
from vision_longformer_amd.msvit import MsViT, parse_arch, vil_arch
from vision_longformer_amd.longformer2d import Long2DSCSelfAttention

SMALL_ARCH = ("l1,h1,d16,n1,s1,g1,p4,f4,a0_l2,h2,d32,n2,s1,g1,p2,f2,a0_"
              "l3,h2,d32,n1,s0,g1,p2,f7,a0")


def test_parse_arch_defaults_and_published_archs():
    cfgs = parse_arch("l1,h3,d96,n1,s1,g1,p4,f7,a0_l2,h3,d192,n2")
    assert cfgs[0] == dict(l=1, h=3, d=96, n=1, s=1, g=1, p=4, f=7, a=0)
    assert cfgs[1] == dict(l=2, h=3, d=192, n=2, s=1, g=1, p=2, f=7, a=1)
    assert vil_arch("small").startswith("l1,h3,d96,n1,s1,g1,p4,f7,a0_l2,h3,d192,n2,s1,g1,p2,f7,a0_l3,h6,d384,n8,s0")


def test_param_counts_match_readme():
    # README.md:77-95 / SURVEY 2c: 6.72 M, 24.66 M
    assert sum(p.numel() for p in build_vil("vil_tiny_224").parameters()) == 6719698
    assert sum(p.numel() for p in build_vil("vil_small_224").parameters()) == 24657328


def test_state_dict_keys_equal_reference(golden_dir):
    gold = np.load(os.path.join(golden_dir, "model_tiny.npz"))
    mine = build_vil("vil_tiny_224", drop_path_rate=0.0)
    assert sorted(mine.state_dict().keys()) == [str(k) for k in gold["state_keys"]]


def test_hot_path_layer_placement_and_mode_switch():
    m = build_vil("vil_base_deep_384_rs")
    hot = [x for x in m.modules() if isinstance(x, Long2DSCSelfAttention)]
    assert len(hot) == 9 and all(h.mode == 1 for h in hot)          # 9 of 34 attention layers
    assert [h.attention_window for h in hot] == [6] + [8] * 8
    m.reset_vil_mode(0)
    assert all(h.mode == 0 for h in hot)
    m = build_vil("vil_small_224")
    assert sum(isinstance(x, Long2DSCSelfAttention) for x in m.modules()) == 3


def test_random_shift_rng_draw_matches_reference_contract():
    import random
    a = Long2DSCSelfAttention(32, num_heads=2, w=4, nglo=1, rpe=True, mode=1)
    random.seed(7)
    expect = [random.randrange(1, 9) for _ in range(5)]
    random.seed(7)
    a.train()
    got = [a._resolve_mode() for _ in range(5)]
    assert got == expect                       # one draw per training forward from the global RNG
    a.eval()
    assert a._resolve_mode() == 0              # evaluation: full 3x3
    a.mode = -1
    assert a._resolve_mode() == -1


def test_no_weight_decay_groups():
    m = MsViT(SMALL_ARCH, img_size=32, num_classes=10)
    groups = param_groups(m, 0.05)
    nd = {id(p) for p in groups[1]["params"]}
    skip = m.no_weight_decay()
    assert {"pos_embed", "cls_token", "norm.weight", "norm.bias", "norm_embed", "head.bias", "relative_position"} == set(skip)
    for n, p in m.named_parameters():
        # the reference's rule (optim/__init__.py:24-37): NAME contains one of the no_weight_decay() substrings;
        # plain Linear biases (qkv / proj / fc) are decayed
        assert (id(p) in nd) == any(s in n for s in skip), n
    assert any(n.endswith("proj.bias") and id(p) not in nd for n, p in m.named_parameters())
    assert sum(len(g["params"]) for g in groups) == len(list(m.parameters()))


def _ddp_worker(rank, world, port, ret):
    import types
    from oracle.cpu_model import _oracle_forward
    from vision_longformer_amd.engine import init_distributed, wrap_ddp, train_step
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    rank, local_rank, world, device = init_distributed()
    torch.manual_seed(0)
    model = MsViT(SMALL_ARCH, img_size=32, num_classes=10, drop_path_rate=0.0, norm_embed=True, sharew=True)
    for mod in model.modules():
        if isinstance(mod, Long2DSCSelfAttention):
            mod.forward = types.MethodType(_oracle_forward, mod)
    opt = make_optimizer(model, lr=1e-2, optimizer_module=OO)
    ddp = wrap_ddp(model, device, world)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4, 3, 32, 32, generator=g)
    t = torch.softmax(torch.randn(4, 10, generator=g), -1)
    half = slice(rank * 2, rank * 2 + 2)
    for _ in range(2):
        train_step(ddp, opt, x[half], t[half], amp_dtype=None)
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    if rank == 0:
        ret["same"] = bool(torch.equal(gathered[0], gathered[1]))
        ret["flat"] = flat.clone()
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_gloo_world2_matches_single_process():
    """Sharding the batch over 2 ranks (gradient all-reduce, mean) == one process on the
    whole batch: the only collective of the path is DDP's, the op itself needs none."""
    import types
    from oracle.cpu_model import _oracle_forward
    from vision_longformer_amd.engine import train_step
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_ddp_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret["same"], "ranks diverged"
    torch.manual_seed(0)
    model = MsViT(SMALL_ARCH, img_size=32, num_classes=10, drop_path_rate=0.0, norm_embed=True, sharew=True)
    for mod in model.modules():
        if isinstance(mod, Long2DSCSelfAttention):
            mod.forward = types.MethodType(_oracle_forward, mod)
    opt = make_optimizer(model, lr=1e-2, optimizer_module=OO)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4, 3, 32, 32, generator=g)
    t = torch.softmax(torch.randn(4, 10, generator=g), -1)
    for _ in range(2):
        train_step(model, opt, x, t, amp_dtype=None)
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    # AdamW normalises tiny gradients, so fp32 summation-order noise shows up at ~1e-5 * lr scale
    torch.testing.assert_close(flat, ret["flat"], rtol=1e-3, atol=2e-4)


def test_master_weight_adamw_matches_plain_adamw_in_fp32():
    """With a float32 'low' dtype the master-weight wrapper must be exactly AdamW."""
    from vision_longformer_amd.engine import MasterWeightAdamW, train_step
    import types
    from oracle.cpu_model import _oracle_forward

    def make():
        torch.manual_seed(0)
        m = MsViT(SMALL_ARCH, img_size=32, num_classes=10, drop_path_rate=0.0, norm_embed=True, sharew=True)
        for mod in m.modules():
            if isinstance(mod, Long2DSCSelfAttention):
                mod.forward = types.MethodType(_oracle_forward, mod)
        return m
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4, 3, 32, 32, generator=g)
    t = torch.softmax(torch.randn(4, 10, generator=g), -1)
    a, b = make(), make()
    oa = make_optimizer(a, lr=1e-2, optimizer_module=OO)
    ob = MasterWeightAdamW(b, lr=1e-2, low_dtype=torch.float32, optimizer_module=OO)
    assert len(ob.low) > 10 and len(ob.direct) > 5
    for _ in range(2):
        train_step(a, oa, x, t, amp_dtype=None)
        train_step(b, ob, x, t, amp_dtype=None)
    fa = torch.cat([p.detach().reshape(-1) for p in a.parameters()])
    fb = torch.cat([p.detach().reshape(-1) for p in b.parameters()])
    torch.testing.assert_close(fa, fb, rtol=0, atol=0)


def test_master_weight_adamw_state_dict_roundtrip_and_fp32_export():
    """The fp32 masters and Adam moments checkpoint through MasterWeightAdamW.state_dict(); the exported model state
    dict is fp32 (the reference checkpoint format, utils/checkpoint.py:170-180), shared aliases included."""
    from vision_longformer_amd.engine import MasterWeightAdamW, set_lr
    torch.manual_seed(0)
    m = MsViT(SMALL_ARCH, img_size=32, num_classes=10, sharew=True)
    opt = MasterWeightAdamW(m, lr=1e-2, optimizer_module=OO)
    for p in opt.low:
        p.grad = torch.randn_like(p)
    for p in opt.direct:
        p.grad = torch.randn_like(p)
    opt.step()
    sd = opt.state_dict()
    fp32 = opt.export_fp32_state_dict(m)
    assert all(v.dtype != torch.bfloat16 for v in fp32.values())
    assert set(fp32.keys()) == set(m.state_dict().keys())
    n0 = opt._low_names[0]
    assert torch.equal(fp32[n0], sd["master"][n0])
    # resume into a fresh model / optimizer
    torch.manual_seed(1)
    m2 = MsViT(SMALL_ARCH, img_size=32, num_classes=10, sharew=True)
    opt2 = MasterWeightAdamW(m2, lr=1e-2, optimizer_module=OO)
    opt2.load_state_dict(sd)
    for a, b in zip(opt.master, opt2.master):
        assert torch.equal(a, b)
    for a, b in zip(opt.low, opt2.low):
        assert torch.equal(a, b) and b.dtype == torch.bfloat16
    s1, s2 = opt.opt.state_dict()["state"], opt2.opt.state_dict()["state"]
    assert s1.keys() == s2.keys() and all(torch.equal(s1[k]["exp_avg"], s2[k]["exp_avg"]) for k in s1)
    # the schedule must reach the INNER optimizer after a resume (its param_groups dicts are replaced by the load)
    set_lr(opt2, 5e-4)
    assert all(float(g["lr"]) == 5e-4 for g in opt2.opt.param_groups)
    w0 = opt2.master[0].detach().clone()
    for p in opt2.low:
        p.grad = torch.ones_like(p)
    for p in opt2.direct:
        p.grad = torch.ones_like(p)
    opt2.step()
    # Adam's first resumed update of a weight with |m/sqrt(v)| ~ 1 moves it by about lr, not by the checkpointed 1e-2
    assert float((opt2.master[0] - w0).abs().max()) < 5e-3


def test_cpu_optimizer_is_refused_without_the_oracle_module():
    m = MsViT(SMALL_ARCH, img_size=32, num_classes=10, sharew=True)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        make_optimizer(m)


def test_attn_drop_in_training_takes_the_materialised_path_not_the_fused_one():
    """attn_drop > 0 in training drops attention PROBABILITIES (reference longformer2d.py:186,224): the module must not
    silently run the fused kernels (which never materialise them).  It runs the operator-level HIP path -- which, like
    everything else, has no CPU fallback."""
    a = Long2DSCSelfAttention(32, num_heads=2, w=4, nglo=1, sharew=True, attn_drop=0.1)
    a.train()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        a(torch.zeros(1, 1 + 16, 32), 4, 4)


def test_mlp_autograd_nodes_match_plain_torch_on_cpu():
    """The two autograd nodes of the MLP block -- (h, gelu(h)) = fc1 with its activation as a NON-differentiable second
    output, then gelu + fc2 as one node whose backward returns the gradient with respect to h -- against plain torch on
    CPU tensors (the host logic only: on CPU both nodes take their torch fallbacks).  Checks that the activation handed
    over is not differentiated twice, that no gradient is materialised for it, and every parameter gradient."""
    from vision_longformer_amd.linear import _LinearGeluOutFn, _GeluLinearFn
    torch.manual_seed(3)
    x = torch.randn(5, 7, 16, dtype=torch.float64, requires_grad=True)
    w1 = torch.randn(64, 16, dtype=torch.float64, requires_grad=True)
    b1 = torch.randn(64, dtype=torch.float64, requires_grad=True)
    w2 = torch.randn(16, 64, dtype=torch.float64, requires_grad=True)
    b2 = torch.randn(16, dtype=torch.float64, requires_grad=True)
    h, a = _LinearGeluOutFn.apply(x, w1, b1)
    assert h.requires_grad and not a.requires_grad
    y = _GeluLinearFn.apply(h, w2, b2, a)
    g = torch.randn_like(y)
    y.backward(g)
    got = [t.grad.clone() for t in (x, w1, b1, w2, b2)]
    for t in (x, w1, b1, w2, b2):
        t.grad = None
    ref = torch.nn.functional.linear(torch.nn.functional.gelu(torch.nn.functional.linear(x, w1, b1)), w2, b2)
    ref.backward(g)
    assert torch.allclose(y, ref, rtol=1e-12, atol=1e-12)
    for a_, t in zip(got, (x, w1, b1, w2, b2)):
        assert torch.allclose(a_, t.grad, rtol=1e-10, atol=1e-10)
    # without the pre-computed activation the second node computes it itself: same result
    h2, _ = _LinearGeluOutFn.apply(x.detach(), w1.detach(), b1.detach())
    assert torch.equal(_GeluLinearFn.apply(h2, w2.detach(), b2.detach()), _GeluLinearFn.apply(h2, w2.detach(), b2.detach(), _))


def test_wgrad_plan_export_import_and_deterministic_mode():
    """The plan cache of the weight gradient through the C ABI (no GPU compute): get / set, export / import across a
    'resume', and the switch that turns timing-based selection off (dW summation order must be reproducible)."""
    from vision_longformer_amd import _lib, linear
    L = _lib.lib()
    T, co, ci = 25216, 1536, 384
    _lib.check(L.vil_linear_wgrad_set_plan(T, co, ci, 0, 0, 0, 0))
    default = linear._wg_get_plan(T, co, ci)
    assert default[4] == 0 and default[0] in (1, 2)                       # the cost model's choice, not tuned
    assert linear._wg_get_plan(T, co, ci) == default                      # ... and it is a pure function of the problem
    linear.import_plans({(T, co, ci): (2, 6, 3, 1)}, device="cpu")
    assert linear._wg_get_plan(T, co, ci) == (2, 6, 3, 1, 1)
    assert linear.export_plans()[(T, co, ci)] == (2, 6, 3, 1)
    assert (torch.device("cpu"), T, co, ci) in linear._WG_TUNED            # no timing run will replace it
    _lib.check(L.vil_linear_wgrad_set_plan(T, co, ci, 0, 0, 0, 0))
    assert (T, co, ci) not in linear.export_plans()
    linear._WG_TUNED.discard((torch.device("cpu"), T, co, ci))
    # a plan that does not divide the problem is refused, and the reader checks its arguments
    assert L.vil_linear_wgrad_set_plan(T, 100, ci, 2, 3, 3, 1) != 0
    import ctypes
    assert L.vil_linear_wgrad_get_plan(T, co, ci, None) != 0
    was = linear._DETERMINISTIC_PLANS
    try:
        linear.deterministic_plans(True)
        assert linear._DETERMINISTIC_PLANS is True
    finally:
        linear.deterministic_plans(was)


def _plan_share_worker(rank, world, port, ret):
    import torch.distributed as dist
    from vision_longformer_amd import linear
    from vision_longformer_amd.engine import sync_replicas, assert_replicas_identical, replica_checksum
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    T, co, ci = 25216, 1536, 384
    mine = (2, 6, 3, 1) if rank == 0 else (2, 3, 6, 2)                     # what each rank's own timing "selected"
    plans = {(T, co, ci): mine}
    if rank == 1:
        plans[(T, 768, 384)] = (2, 3, 6, 2)                                # a problem only this rank has met
    linear.import_plans(plans, device="cpu")
    # replicas built from different seeds (a caller that forgot to seed): the engine's sync makes them rank 0's
    torch.manual_seed(100 + rank)
    model = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.LayerNorm(8))
    opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9)
    model(torch.randn(4, 8)).sum().backward()
    opt.step()                                                             # momentum buffers exist and differ per rank
    lo, hi = replica_checksum(model)
    differed = not torch.equal(lo, hi)
    sync_replicas(model, opt)                                              # parameters, optimizer state, plans: ONE entry point
    assert_replicas_identical(model)
    mom = torch.cat([opt.state[p]["momentum_buffer"].reshape(-1) for p in model.parameters()])
    both = [torch.zeros_like(mom) for _ in range(world)]
    dist.all_gather(both, mom)
    ret[rank] = (linear._wg_get_plan(T, co, ci), linear._wg_get_plan(T, 768, 384), differed, bool(torch.equal(both[0], both[1])))
    # a diverged replica is reported, by every rank
    if rank == 1:
        with torch.no_grad():
            next(model.parameters()).add_(1e-3)
    try:
        assert_replicas_identical(model, what="test")
        ret[f"raised{rank}"] = False
    except RuntimeError:
        ret[f"raised{rank}"] = True
    dist.barrier()
    dist.destroy_process_group()


def test_replica_sync_and_rank0_plans_over_gloo():
    """engine.sync_replicas (what GraphedTrainStep(world > 1) runs itself): two ranks that differ in parameters,
    optimizer state and measured weight-gradient plans end up on rank 0's; a problem only one rank has met keeps its
    local plan and the collective still completes (round 4 broadcast from inside backward); a replica that
    diverges afterwards is detected on every rank."""
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_plan_share_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret[0][0] == (2, 6, 3, 1, 1) and ret[1][0] == (2, 6, 3, 1, 1)
    assert ret[1][1] == (2, 3, 6, 2, 1) and ret[0][1][4] == 0              # rank 1 keeps its own; rank 0 never had one
    assert ret[0][2] and ret[1][2] and ret[0][3] and ret[1][3]
    assert ret["raised0"] and ret["raised1"]
