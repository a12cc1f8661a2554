"""GPU (-m gpu): the optimizer step -- the reference's AdamW / QHM update rules on the HIP multi-tensor
kernel (csrc/vil_optim.hip through vision_longformer_amd.optim) against the parameters the imported reference
optimizers produced (tests/golden/optim_reference.npz), with fp32 and 16-bit gradients, the fused 16-bit working copy,
hipGraph replay and checkpoint round trips."""
import os

import numpy as np
import pytest
import torch

import optim_cases as OC

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "optim_reference.npz")
# fp32 tensors, same operation order as the reference; the CPU reference contracts `a + alpha*b` into an FMA in its
# vectorised loops and not in its scalar tails, the kernel never contracts: a few ulp OF THE LARGEST TERM per step (a
# parameter that ends near zero after steps of size ~1 carries the absolute rounding of those steps)
RTOL, ATOL_REL = 2e-6, 3e-7


def _close(t, ref, what):
    atol = ATOL_REL * max(float(ref.abs().max()), 1e-3)
    torch.testing.assert_close(t, ref, rtol=RTOL, atol=atol, msg=lambda m: f"{what}: {m}")
    return float(((t - ref).abs() / (atol + RTOL * ref.abs())).max())


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "the gpu-marked tests need a GPU"
    return torch.device("cuda:0")


def _hip_opt(kind, groups, hyper):
    from vision_longformer_amd import optim
    return optim.AdamW(groups, **hyper) if kind == "adamw" else optim.QHM(groups, **hyper)


def _run(dev, name, grad_dtype, bind_low):
    """the seeded case on the GPU; returns (params after every step, working copies after the last step)"""
    kind, hyper, wds = OC.CASES[name]
    params, grads = OC.make_inputs()
    ps = [torch.nn.Parameter(p.clone().to(dev)) for p in params]
    lows = [torch.nn.Parameter(p.clone().to(dev, torch.bfloat16)) for p in params] if bind_low else None
    groups = [{"params": ps[0::2], "weight_decay": wds[0]}, {"params": ps[1::2], "weight_decay": wds[1]}]
    opt = _hip_opt(kind, groups, hyper)
    if bind_low:
        for p, l in zip(ps, lows):
            opt.bind_working_copy(p, l)
    out = []
    for k in range(OC.NSTEPS):
        for i, (p, g) in enumerate(zip(ps, grads[k])):
            if bind_low:
                lows[i].grad = g.to(dev, grad_dtype)
            else:
                p.grad = g.to(dev, grad_dtype)
        opt.step()
        out.append([p.detach().float().cpu().clone() for p in ps])
    return out, lows, ps, opt


@pytest.mark.parametrize("name", list(OC.CASES))
@pytest.mark.parametrize("grad_dtype", [torch.float32, torch.bfloat16])
def test_optimizer_kernel_vs_reference_fixtures(dev, name, grad_dtype):
    gold = np.load(GOLD)
    # 16-bit gradients belong to 16-bit working copies bound to the fp32 masters (autograd's dtype rule)
    got, _, _, _ = _run(dev, name, grad_dtype, bind_low=grad_dtype != torch.float32)
    worst = 0.0
    for k in (0, OC.NSTEPS - 1):
        for i, t in enumerate(got[k]):
            ref = torch.from_numpy(gold[f"{name}/step{k + 1}/p{i}"])
            worst = max(worst, _close(t, ref, f"{name} step {k + 1} tensor {i}"))
    assert worst <= 1.0


@pytest.mark.parametrize("name", ["adamw_recipe", "qhm_recipe", "qhm_nu07_wd"])
def test_optimizer_kernel_writes_the_working_copy_in_the_same_pass(dev, name):
    gold = np.load(GOLD)
    got, lows, ps, _ = _run(dev, name, torch.bfloat16, bind_low=True)
    for i, t in enumerate(got[-1]):
        ref = torch.from_numpy(gold[f"{name}/step{OC.NSTEPS}/p{i}"])
        _close(t, ref, f"{name} tensor {i}")
        assert torch.equal(lows[i].detach(), ps[i].detach().to(torch.bfloat16)), i      # bf16(master), written by the kernel


def test_optimizer_kernel_under_hipgraph_replay_with_device_lr(dev):
    """One captured launch replayed N times == N eager steps: the plan of the captured addresses is uploaded on a
    side stream, the step counter advances on the device, the lr is a device tensor the schedule updates in place."""
    from vision_longformer_amd import optim
    kind, hyper, wds = OC.CASES["adamw_recipe"]
    params, grads = OC.make_inputs()

    def make(lr):
        ps = [torch.nn.Parameter(p.clone().to(dev)) for p in params]
        h = dict(hyper); h["lr"] = lr
        return ps, optim.AdamW([{"params": ps[0::2], "weight_decay": wds[0]}, {"params": ps[1::2], "weight_decay": wds[1]}], **h)

    lrs = [5e-4, 4e-4, 3e-4, 2e-4, 1e-4]
    pe, oe = make(lrs[0])
    for k in range(OC.NSTEPS):
        for g_ in oe.param_groups:
            g_["lr"] = lrs[k]
        for p, g in zip(pe, grads[k]):
            p.grad = g.to(dev)
        oe.step()
    lr_dev = torch.tensor(lrs[0], device=dev)
    pg, og = make(lr_dev)
    static = [torch.zeros_like(p) for p in pg]
    for p, s in zip(pg, static):
        p.grad = s
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="allocate"):
        with torch.cuda.graph(torch.cuda.CUDAGraph()):
            og.step()
    torch.cuda.synchronize()
    og.allocate()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        og.step()
    og.after_capture()
    for k in range(OC.NSTEPS):
        lr_dev.fill_(lrs[k])
        for s, g in zip(static, grads[k]):
            s.copy_(g)
        graph.replay()
    torch.cuda.synchronize()
    for a, b in zip(pe, pg):
        assert torch.equal(a.detach(), b.detach())


def test_optimizer_state_dict_round_trip_in_place(dev):
    """load_state_dict copies the moments INTO the existing tensors (a captured step keeps reading them), keeps the lr
    object and restores the step count: the resumed run continues bit-identically."""
    from vision_longformer_amd import optim
    kind, hyper, wds = OC.CASES["adamw_defaults"]
    params, grads = OC.make_inputs()

    def make():
        ps = [torch.nn.Parameter(p.clone().to(dev)) for p in params]
        return ps, optim.AdamW([{"params": ps[0::2], "weight_decay": wds[0]}, {"params": ps[1::2], "weight_decay": wds[1]}], **hyper)

    pa, oa = make()
    for k in range(3):
        for p, g in zip(pa, grads[k]):
            p.grad = g.to(dev)
        oa.step()
    sd = oa.state_dict()
    assert sd["vil_steps"] == [3]
    pb, ob = make()
    for p, g in zip(pb, grads[0]):
        p.grad = g.to(dev)
    ob.step()                                        # state tensors exist before the load
    ptr = ob.state[pb[0]]["exp_avg"].data_ptr()
    with torch.no_grad():
        for a, b in zip(pa, pb):
            b.copy_(a)
    ob.load_state_dict(sd)
    assert ob.state[pb[0]]["exp_avg"].data_ptr() == ptr
    for k in range(3, OC.NSTEPS):
        for p, q, g in zip(pa, pb, grads[k]):
            p.grad = g.to(dev); q.grad = g.to(dev)
        oa.step(); ob.step()
    for a, b in zip(pa, pb):
        assert torch.equal(a.detach(), b.detach())


def test_optimizer_rejects_cpu_tensors():
    from vision_longformer_amd import optim
    p = torch.nn.Parameter(torch.zeros(8))
    p.grad = torch.ones(8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        optim.AdamW([p], lr=1e-3).step()
    with pytest.raises(ValueError):
        optim.AdamW([p], lr=-1.0)
    with pytest.raises(ValueError):
        optim.QHM([p], lr=0.1, momentum=1.5)


def test_capture_without_an_eager_step_keeps_the_moments(dev):
    """allocate() must create the optimizer state eagerly.  A step captured with NO eager step before
    (GraphedTrainStep(warmup=0)) used to create exp_avg / exp_avg_sq inside the capture -- the zero-fill became a graph
    node and re-zeroed the moments at every replay while the device step counter kept advancing.  Here nothing runs
    before allocate() + capture, and the replays must reproduce the reference fixtures."""
    from vision_longformer_amd import optim
    gold = np.load(GOLD)
    kind, hyper, wds = OC.CASES["adamw_recipe"]
    params, grads = OC.make_inputs()
    ps = [torch.nn.Parameter(p.clone().to(dev)) for p in params]
    opt = optim.AdamW([{"params": ps[0::2], "weight_decay": wds[0]}, {"params": ps[1::2], "weight_decay": wds[1]}], **hyper)
    static = [torch.zeros_like(p) for p in ps]
    for p, s in zip(ps, static):
        p.grad = s
    opt.allocate()                                   # BEFORE any capture attempt, no eager step
    assert all("exp_avg" in opt.state[p] for p in ps)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        opt.step()
    opt.after_capture()
    for k in range(OC.NSTEPS):
        for s, g in zip(static, grads[k]):
            s.copy_(g)
        graph.replay()
    torch.cuda.synchronize()
    for i, p in enumerate(ps):
        _close(p.detach().float().cpu(), torch.from_numpy(gold[f"adamw_recipe/step{OC.NSTEPS}/p{i}"]), f"tensor {i}")


def test_state_created_inside_a_capture_raises(dev):
    from vision_longformer_amd import optim
    p = torch.nn.Parameter(torch.zeros(64, device=dev))
    p.grad = torch.ones(64, device=dev)
    opt = optim.AdamW([p], lr=1e-3)
    with pytest.raises(RuntimeError, match="allocate"):
        with torch.cuda.graph(torch.cuda.CUDAGraph()):
            opt.step()
    torch.cuda.synchronize()
    assert "exp_avg" not in opt.state[p]             # nothing was created in the aborted capture's pool


def test_checkpoint_is_interchangeable_with_the_reference_adamw(dev):
    """the reference keeps state[p]['step'] per parameter (optimization.py:155-165).  Our state_dict
    carries it, and a checkpoint WITHOUT 'vil_steps' (one written by the reference class) restores the bias-correction
    step from the per-parameter values instead of restarting at t = 1 on warm moments."""
    from vision_longformer_amd import optim
    kind, hyper, wds = OC.CASES["adamw_defaults"]
    params, grads = OC.make_inputs()

    def make():
        ps = [torch.nn.Parameter(p.clone().to(dev)) for p in params]
        return ps, optim.AdamW([{"params": ps[0::2], "weight_decay": wds[0]}, {"params": ps[1::2], "weight_decay": wds[1]}], **hyper)

    pa, oa = make()
    for k in range(3):
        for p, g in zip(pa, grads[k]):
            p.grad = g.to(dev)
        oa.step()
    sd = oa.state_dict()
    assert all(st["step"] == 3 for st in sd["state"].values())          # what the reference's step() will increment
    import copy
    ref_style = copy.deepcopy({"state": sd["state"], "param_groups": sd["param_groups"]})   # no 'vil_steps': a reference checkpoint (as read from disk)
    pb, ob = make()
    with torch.no_grad():
        for a, b in zip(pa, pb):
            b.copy_(a)
    ob.load_state_dict(ref_style)                    # before any step of `ob`: its launch bucket does not exist yet
    for k in range(3, OC.NSTEPS):
        for p, q, g in zip(pa, pb, grads[k]):
            p.grad = g.to(dev); q.grad = g.to(dev)
        oa.step(); ob.step()
    for a, b in zip(pa, pb):
        assert torch.equal(a.detach(), b.detach())
    assert ob.state_dict()["vil_steps"] == [OC.NSTEPS]


def test_python_float_lr_schedule_needs_no_plan_rebuild_and_reaches_a_captured_step(dev):
    """A per-iteration schedule that assigns Python floats to group['lr'] (the reference's engine does): the lr lives in
    a device scalar the optimizer owns, so the plan key does not change from step to step, and a CAPTURED step follows
    the schedule through sync_lr() / engine.set_lr -- it used to bake the float of capture time into the graph."""
    from vision_longformer_amd import optim
    from vision_longformer_amd.engine import set_lr
    kind, hyper, wds = OC.CASES["adamw_recipe"]
    params, grads = OC.make_inputs()
    lrs = [5e-4, 4e-4, 3e-4, 2e-4, 1e-4]

    def make():
        ps = [torch.nn.Parameter(p.clone().to(dev)) for p in params]
        return ps, optim.AdamW([{"params": ps[0::2], "weight_decay": wds[0]}, {"params": ps[1::2], "weight_decay": wds[1]}], **hyper)

    pe, oe = make()
    static_e = [torch.zeros_like(p) for p in pe]
    for p, s in zip(pe, static_e):
        p.grad = s
    keys = []
    for k in range(OC.NSTEPS):
        for g_ in oe.param_groups:
            g_["lr"] = lrs[k]
        for s, g in zip(static_e, grads[k]):
            s.copy_(g)
        oe.step()
        keys.append(next(iter(oe._plans.values())).eager.key)
    assert all(k_ == keys[0] for k_ in keys)         # same addresses, same plan: no rebuild, no host synchronisation
    pg, og = make()
    static = [torch.zeros_like(p) for p in pg]
    for p, s in zip(pg, static):
        p.grad = s
    og.allocate()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        og.step()
    og.after_capture()
    for k in range(OC.NSTEPS):
        set_lr(og, lrs[k])
        for s, g in zip(static, grads[k]):
            s.copy_(g)
        graph.replay()
    torch.cuda.synchronize()
    for a, b in zip(pe, pg):
        assert torch.equal(a.detach(), b.detach())


def test_second_capture_with_other_addresses_raises(dev):
    """every capture reads the bucket's one graph plan (nblocks baked into the node): re-capturing with other gradient
    addresses would silently retarget the first graph -- it raises instead"""
    from vision_longformer_amd import optim
    p = torch.nn.Parameter(torch.zeros(4096, device=dev))
    opt = optim.AdamW([p], lr=torch.tensor(1e-3, device=dev))
    p.grad = torch.ones(4096, device=dev)
    opt.allocate()
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1):
        opt.step()
    opt.after_capture()
    p.grad = torch.ones(4096, device=dev)            # another gradient tensor: another address
    with pytest.raises(RuntimeError, match="already captured"):
        with torch.cuda.graph(torch.cuda.CUDAGraph()):
            opt.step()
    torch.cuda.synchronize()


@pytest.mark.parametrize("name", ["adamw_recipe", "qhm_recipe"])
def test_loss_scaled_step_matches_the_reference_fixtures_and_skips_on_inf(name, dev):
    """fp16 training under torch.amp.GradScaler (reference src/engine.py:84-100): the optimizer declares
    _step_supports_amp_scaling, so scaler.step() hands it `grad_scale` / `found_inf` device tensors and the HIP launch
    unscales on load and skips on the device.  Gradients multiplied by 1024 and fed with grad_scale = 1024 reproduce the
    reference fixtures bit for bit (a power of two: the unscaling is exact); a step with found_inf = 1 in between changes
    nothing -- parameters, moments, step count -- so the trajectory afterwards is still the fixtures'."""
    gold = np.load(GOLD)
    kind, hyper, wds = OC.CASES[name]
    params, grads = OC.make_inputs()
    ps = [torch.nn.Parameter(p.clone().to(dev)) for p in params]
    opt = _hip_opt(kind, [{"params": ps[0::2], "weight_decay": wds[0]}, {"params": ps[1::2], "weight_decay": wds[1]}], hyper)
    assert getattr(opt, "_step_supports_amp_scaling", False)
    scale = torch.tensor(1024.0, device=dev)
    for k in range(OC.NSTEPS):
        if k == 2:                                   # an overflowed step in the middle: skipped entirely
            before = [p.detach().clone() for p in ps]
            for p in ps:
                p.grad = torch.full_like(p, float("inf"))
            opt.grad_scale, opt.found_inf = scale, torch.ones(1, device=dev)
            opt.step()
            torch.cuda.synchronize()
            assert all(torch.equal(a, p.detach()) for a, p in zip(before, ps)), "a skipped step moved a parameter"
        for p, g in zip(ps, grads[k]):
            p.grad = (g * 1024.0).to(dev)
        opt.grad_scale, opt.found_inf = scale, torch.zeros(1, device=dev)
        opt.step()
        del opt.grad_scale, opt.found_inf
        if f"{name}/step{k + 1}/p0" in gold.files:          # (the fixtures hold steps 1 and NSTEPS)
            for i, p in enumerate(ps):
                _close(p.detach().float().cpu(), torch.from_numpy(gold[f"{name}/step{k + 1}/p{i}"]), f"{name} step {k + 1} tensor {i}")


def test_grad_scaler_drives_the_master_weight_optimizer(dev):
    """torch.amp.GradScaler.step(MasterWeightOptimizer): fp16 working copies, fp16 gradients living on them.  The
    optimizer's own non-finite check must see those gradients (GradScaler's walk over param_groups would not), the scale
    must back off after an overflow, and a clean step must equal the unscaled step."""
    from vision_longformer_amd.engine import MasterWeightOptimizer
    torch.manual_seed(0)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.fc = torch.nn.Linear(64, 64)
            self.norm = torch.nn.LayerNorm(64)

        def no_weight_decay(self):
            return set()

    def build():
        torch.manual_seed(0)
        m = Net().to(dev)
        return m, MasterWeightOptimizer(m, kind="adamw", low_dtype=torch.float16, lr=1e-2)

    g = torch.Generator().manual_seed(3)
    # fp16-representable gradients: scaling by 256 and back is then exact, so the two runs must agree to rounding
    gw = (torch.randn(64, 64, generator=g) * 0.05).half().float().to(dev)
    gb = (torch.randn(64, generator=g) * 0.05).half().float().to(dev)
    # reference run: unscaled gradients, plain step
    m0, o0 = build()
    m0.fc.weight.grad, m0.fc.bias.grad = gw.half(), gb.half()
    m0.norm.weight.grad, m0.norm.bias.grad = gb.clone(), gb.clone()
    o0.step()
    # scaled run through GradScaler.step
    m1, o1 = build()
    scaler = torch.amp.GradScaler("cuda", init_scale=256.0, growth_interval=1000)
    scaler.scale(torch.zeros((), device=dev))                 # (creates the scaler's state, as scale(loss) does in a loop)
    m1.fc.weight.grad, m1.fc.bias.grad = (gw * 256).half(), (gb * 256).half()
    m1.norm.weight.grad, m1.norm.bias.grad = gb * 256, gb * 256
    scaler.step(o1)
    scaler.update()
    torch.cuda.synchronize()
    for a, b in zip(o0.master, o1.master):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-7)
    assert float(scaler.get_scale()) == 256.0
    # overflow in a 16-bit gradient only: step skipped, scale halved
    before = [mm.clone() for mm in o1.master]
    m1.fc.weight.grad = torch.full_like(m1.fc.weight, float("inf"))
    m1.fc.bias.grad = (gb * 256).half()
    m1.norm.weight.grad, m1.norm.bias.grad = gb * 256, gb * 256
    scaler.step(o1)
    scaler.update()
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(before, o1.master)), "the overflowed step was not skipped"
    assert float(scaler.get_scale()) == 128.0, "the scale did not back off"
