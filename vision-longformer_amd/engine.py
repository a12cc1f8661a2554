"""Data-parallel training step for ViL on synthetic ImageNet-shape batches.

Counterpart of the reference's loop (src/engine.py:60-195 driven by
src/run_experiment.py:142-262) reduced to what the throughput metric needs, and
MI355X-first: one process per GPU, `torch.distributed` backend "nccl" (= RCCL
over xGMI on ROCm); the step is captured as hipGraphs (`GraphedTrainStep`: forward +
backward, one all-reduce per flat gradient buffer, AdamW) because the eager step is
host-bound; the eager fallback is DistributedDataParallel with `broadcast_buffers=False`
(the reference re-broadcasts the constant int64 relative_position_index buffers every
step: 3-66 MB of pure waste) and `gradient_as_bucket_view=True`.
bf16 compute with fp32 master weights (`MasterWeightAdamW`; no GradScaler needed), and no
per-step host synchronisation (the reference's meters call .item() every step).
The hot path has no collective of its own: (image, head, chunk) units are
independent, so the batch is simply sharded across ranks."""
import os

import torch
import torch.distributed as dist
import torch.nn.functional as F

from .msvit import MsViT, vil_arch

CONFIGS = {
    # name: (arch family, image size, per-GPU batch, stage-1 window, stage-2 window, random-shift mode)
    "vil_tiny_224": ("tiny", 224, 2, 7, 7, 0),
    "vil_small_224": ("small", 224, 128, 7, 7, 0),
    "vil_medium_deep_384": ("medium_deep", 384, 32, 7, 7, 0),
    "vil_medium_deep_384_f8f12": ("medium_deep", 384, 32, 8, 12, 0),
    "vil_base_deep_384_rs": ("base_deep", 384, 32, 6, 8, 1),
}


def recipe_of(config):
    """optimizer of the reference's recipe for a configuration: AdamW for the 224 training recipe (config/msvit.yaml),
    QHM for the 384 fine-tuning recipe (config/msvit_384finetune.yaml)"""
    return "qhm" if CONFIGS[config][1] == 384 else "adamw"


def init_distributed(single_rank_group=False):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun / torch.distributed.run).  Collectives carry a
    timeout (VIL_DIST_TIMEOUT_S, default 600 s) so that a mis-ordered stream or a missing rank fails loudly instead of
    hanging.  single_rank_group: also create the (RCCL) process group when WORLD_SIZE == 1 -- the multi-GPU code path
    (segment graphs + asynchronous all-reduce) then runs for real on one GPU (bench.py --force-segments, tests)."""
    import datetime
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = torch.cuda.is_available()
    # VIL_SHARE_DEVICE=1 (tests on a 1-GPU box): every rank uses cuda:0 and the collectives go over gloo
    share = use_cuda and os.environ.get("VIL_SHARE_DEVICE") == "1"
    device = torch.device("cuda", 0 if share else local_rank) if use_cuda else torch.device("cpu")
    if use_cuda:
        torch.cuda.set_device(device)
    if (world > 1 or single_rank_group) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        tmo = datetime.timedelta(seconds=float(os.environ.get("VIL_DIST_TIMEOUT_S", "600")))
        if use_cuda and not share:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device, timeout=tmo)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=tmo)
    return rank, local_rank, world, device


def build_vil(config, drop_path_rate=0.1, num_classes=1000, **overrides):
    fam, img, _, f1, f2, mode = CONFIGS[config]
    kw = dict(img_size=img, num_classes=num_classes, drop_path_rate=drop_path_rate, norm_embed=True,
              sharew=True, attn_type="longformerhand", mode=mode)
    kw.update(overrides)
    return MsViT(vil_arch(fam, f1, f2), **kw)


def param_groups(model, weight_decay):
    """No weight decay exactly on the parameters whose NAME contains one of the substrings
    MsViT.no_weight_decay() lists -- the reference's rule (optim/__init__.py:24-37): plain Linear
    biases (qkv / proj / fc) ARE decayed there, so they are here."""
    skip = model.no_weight_decay()
    decay, no_decay = [], []
    for n, p in model.named_parameters():
        if not p.requires_grad:
            continue
        (no_decay if _no_decay(n, skip) else decay).append(p)
    return [{"params": decay, "weight_decay": weight_decay}, {"params": no_decay, "weight_decay": 0.0}]


def _no_decay(name, skip):
    return any(s in name for s in skip)


def _lr_value(lr, device, capturable):
    """A captured optimizer step bakes a Python-float lr into the graph as a kernel scalar; a
    per-iteration schedule (reference engine.py: warm-up + cosine every iteration) would then be
    silently ignored on replay.  With capturable=True the lr is therefore a DEVICE tensor that the
    schedule updates in place (`set_lr`)."""
    return torch.tensor(float(lr), dtype=torch.float32, device=device) if capturable else float(lr)


def set_lr(optimizer, lr):
    """In-place learning-rate update that a captured (hipGraph) optimizer step observes."""
    for g in optimizer.param_groups:
        if torch.is_tensor(g["lr"]):
            g["lr"].fill_(float(lr))
        else:
            g["lr"] = float(lr)
    for o in (optimizer, getattr(optimizer, "opt", None)):
        if hasattr(o, "sync_lr"):
            o.sync_lr()                  # Python-float lr: refresh the device scalar the (captured) HIP step reads


# The reference's recipes (src/config/msvit.yaml:32-47: AdamW lr 5e-4, wd 0.05, betas (0.9, 0.999), eps 1e-8 through
# get_opt; src/config/msvit_384finetune.yaml:28-35: QHM lr 0.01, momentum 0.9, nu 1, wd 0)
RECIPES = {"adamw": dict(lr=5e-4, weight_decay=0.05, betas=(0.9, 0.999), eps=1e-8, correct_bias=True),
           "qhm": dict(lr=0.01, weight_decay=0.0, momentum=0.9, qhm_nu=1.0)}


def _optimizer_classes(device, optimizer_module):
    """The optimizer implementation: the HIP multi-tensor kernels (optim.py) on the GPU.  There is no CPU product path;
    CPU host-logic tests and bench.py's cpu_baseline leg pass oracle.optim_oracle explicitly."""
    if optimizer_module is not None:
        return optimizer_module
    if device.type != "cuda":
        raise RuntimeError("the optimizer step is a HIP kernel (vision_longformer_amd.optim): no CPU fallback; "
                           "CPU tests pass optimizer_module=oracle.optim_oracle")
    from . import optim
    return optim


def _build(mod, kind, groups, hyper):
    hyper = dict(hyper)
    hyper.pop("weight_decay", None)                    # per group
    return mod.AdamW(groups, **hyper) if kind == "adamw" else mod.QHM(groups, **hyper)


def make_optimizer(model, kind="adamw", capturable=False, optimizer_module=None, **overrides):
    """The reference's optimizer of the recipe `kind` on the model's fp32 parameters, with get_opt's two weight-decay
    groups (src/optim/__init__.py:24-37)."""
    p0 = next(model.parameters())
    mod = _optimizer_classes(p0.device, optimizer_module)
    hyper = dict(RECIPES[kind]); hyper.update(overrides)
    hyper["lr"] = _lr_value(hyper["lr"], p0.device, bool(capturable and p0.is_cuda))
    return _build(mod, kind, param_groups(model, hyper["weight_decay"]), hyper)


class MasterWeightOptimizer:
    """The reference's optimizer (`kind`: "adamw" | "qhm") on fp32 master weights; the module holds 16-bit working
    copies of every GEMM / conv parameter.  Numerically this is bf16 autocast (weights rounded to bf16 once per step
    from the fp32 master) without the per-call cast kernels, and DDP all-reduces the bf16 gradients (half the bytes
    over xGMI).  On the GPU the whole step is ONE kernel launch (csrc/vil_optim.hip): it reads the 16-bit gradients
    autograd produced, updates master + state, and writes the working copy in the same pass."""

    # fp16 training under torch.amp.GradScaler (the reference's loop: src/engine.py:84-100): scaler.step(this) hands the
    # scaler to step(), which checks EVERY gradient it consumes for non-finite values (the 16-bit gradients live on the
    # working copies, where GradScaler's own walk over param_groups would not look), and the HIP launch unscales in fp32
    # and skips on the device
    _step_supports_amp_scaling = True

    def __init__(self, model, kind="adamw", low_dtype=torch.bfloat16, capturable=False, optimizer_module=None, **overrides):
        skip = model.no_weight_decay()
        names = {id(p): n for n, p in model.named_parameters()}
        masters = {}
        for m in model.modules():
            if isinstance(m, (torch.nn.Linear, torch.nn.Conv2d)):
                for p in m.parameters(recurse=False):
                    if id(p) not in masters and p.dtype == torch.float32:
                        masters[id(p)] = torch.nn.Parameter(p.detach().clone(), requires_grad=False)
                        p.data = p.data.to(low_dtype)
        self.low, self.master, self.direct = [], [], []
        decay, no_decay = [], []
        for p in model.parameters():
            if not p.requires_grad:
                continue
            n = names[id(p)]
            if id(p) in masters:
                mp = masters[id(p)]
                self.low.append(p); self.master.append(mp)
                tgt = mp
            else:
                self.direct.append(p)
                tgt = p
            (no_decay if _no_decay(n, skip) else decay).append(tgt)
        p0 = next(model.parameters())
        mod = _optimizer_classes(p0.device, optimizer_module)
        self.kind = kind
        hyper = dict(RECIPES[kind]); hyper.update(overrides)
        hyper["lr"] = _lr_value(hyper["lr"], p0.device, bool(capturable and p0.is_cuda))
        self.opt = _build(mod, kind, [{"params": decay, "weight_decay": hyper["weight_decay"]},
                                      {"params": no_decay, "weight_decay": 0.0}], hyper)
        self._fused = hasattr(self.opt, "bind_working_copy")
        if self._fused:
            for mp, p in zip(self.master, self.low):
                self.opt.bind_working_copy(mp, p)
        self._low_names = [names[id(p)] for p in self.low]

    @property
    def param_groups(self):          # always the inner optimizer's live list (load_state_dict replaces the dicts)
        return self.opt.param_groups

    # ---- checkpointing: the module's state_dict holds bf16-rounded working copies of the GEMM weights, so the
    # fp32 masters and the optimizer state live here (reference checkpoints: utils/checkpoint.py:170-180 saves
    # {net, optimizer}; `export_fp32_state_dict` gives the `net` entry in the reference's fp32 format)
    def state_dict(self):
        return {"master": {n: m.detach().clone() for n, m in zip(self._low_names, self.master)},
                "opt": self.opt.state_dict()}

    @torch.no_grad()
    def load_state_dict(self, sd):
        for n, m, p in zip(self._low_names, self.master, self.low):
            m.copy_(sd["master"][n])
            p.copy_(m)                              # refresh the working copy from the master
        if self._fused:
            self.opt.load_state_dict(sd["opt"])     # in place: state tensors and lr objects keep their addresses
        else:
            lrs = [g["lr"] for g in self.opt.param_groups]
            self.opt.load_state_dict(sd["opt"])
            for g, lr in zip(self.opt.param_groups, lrs):
                if torch.is_tensor(lr):
                    lr.fill_(float(g["lr"]))
                    g["lr"] = lr

    def export_fp32_state_dict(self, model):
        """model.state_dict() with every 16-bit working weight replaced by its fp32 master."""
        out = {k: v.detach().clone() for k, v in model.state_dict().items()}
        master_of = {id(p): m for p, m in zip(self.low, self.master)}
        for n, p in model.named_parameters(remove_duplicate=False):   # shared (sharew) weights appear under every alias
            if id(p) in master_of:
                out[n] = master_of[id(p)].detach().clone()
        return out

    def zero_grad(self, set_to_none=True):
        for p in self.low:
            p.grad = None
        for p in self.direct:
            p.grad = None

    @torch.no_grad()
    def reset_state(self):
        """zeroes the optimizer state in place (moments / momentum buffers and the step count): the tensors a captured
        step reads keep their addresses"""
        for st in self.opt.state.values():
            for k, v in st.items():
                if torch.is_tensor(v):
                    v.zero_()
                elif k == "step":
                    st[k] = 0
        if self._fused:
            self.opt.reset_step_count()

    def settle(self):
        """waits for a pending plan upload (call before and after stream capture)"""
        if self._fused:
            self.opt.allocate()
            self.opt.after_capture()

    def _amp_found_inf(self):
        """device float: non-zero iff a gradient this optimizer is about to consume is inf / NaN (no host synchronisation)"""
        grads = [p.grad for p in self.low + self.direct if p.grad is not None]
        if not grads:                              # nothing to consume: nothing can be non-finite
            ps = self.low + self.direct
            return torch.zeros(1, dtype=torch.float32, device=ps[0].device if ps else "cpu")
        dev = grads[0].device
        found = torch.zeros(1, dtype=torch.float32, device=dev)
        one = torch.ones(1, dtype=torch.float32, device=dev)
        by_dtype = {}
        for g in grads:
            by_dtype.setdefault(g.dtype, []).append(g)
        for gl in by_dtype.values():
            torch._amp_foreach_non_finite_check_and_unscale_(gl, found, one)      # scale 1: a pure check, gradients untouched
        return found

    @torch.no_grad()
    def step(self, grad_scaler=None):
        if self._fused:
            if grad_scaler is not None:
                # GradScaler.step() passed itself (its contract with optimizers that declare _step_supports_amp_scaling):
                # record the verdict where scaler.update() reads it, hand scale and verdict to the launch
                st = grad_scaler._per_optimizer_states[id(self)]
                if getattr(st.get("stage"), "name", "") == "UNSCALED":
                    # scaler.unscale_(opt) ran first (the gradient-clipping pattern): it reaches only the gradients it can
                    # see through param_groups -- the fp32 `direct` ones; the 16-bit gradients of the working copies would
                    # be unscaled once by the launch and the direct ones twice.  There is no consistent reading: refuse.
                    raise RuntimeError("MasterWeightOptimizer: scaler.unscale_(optimizer) before scaler.step(optimizer) is not "
                                       "supported -- the unscale is part of the HIP optimizer launch (clip on the scaled "
                                       "gradients with max_norm * scaler.get_scale(), or unscale after the step)")
                found = self._amp_found_inf()
                st["found_inf_per_device"] = {found.device: found}
                self.opt.grad_scale, self.opt.found_inf = grad_scaler._get_scale_async(), found
                try:
                    self.opt.step()
                finally:
                    del self.opt.grad_scale, self.opt.found_inf
                return
            gs, fi = getattr(self, "grad_scale", None), getattr(self, "found_inf", None)
            if gs is not None:
                # the attribute protocol (GradScaler registers grad_scale / found_inf on the optimizer it was given -- this
                # wrapper -- instead of passing itself): forward them to the launch, with this wrapper's own check of the
                # 16-bit gradients OR-ed into the verdict
                found = self._amp_found_inf()
                if fi is not None:
                    found = torch.maximum(found, fi.to(found.dtype).reshape(1))
                    fi.copy_(found.reshape(fi.shape).to(fi.dtype))
                self.opt.grad_scale, self.opt.found_inf = gs, found
                try:
                    self.opt.step()
                finally:
                    del self.opt.grad_scale, self.opt.found_inf
                return
            self.opt.step()                         # one launch: bf16 grads in, master + state + working copy out
            return
        if grad_scaler is not None:
            raise RuntimeError("loss scaling is part of the HIP optimizer step (no CPU path)")
        # CPU host-logic tests (oracle optimizer): gradients up-cast, step, working copies refreshed
        for p, m in zip(self.low, self.master):
            m.grad = None if p.grad is None else p.grad.float()
        self.opt.step()
        for p, m in zip(self.low, self.master):
            p.copy_(m)


def to_working_precision(model, low_dtype=torch.bfloat16):
    """16-bit working copies of every GEMM / conv weight, in place -- what MasterWeightOptimizer does to a model it
    trains -- for an inference-only model (no masters kept): the forward then runs without per-call autocast casts."""
    for m in model.modules():
        if isinstance(m, (torch.nn.Linear, torch.nn.Conv2d)):
            for p in m.parameters(recurse=False):
                if p.dtype == torch.float32:
                    p.data = p.data.to(low_dtype)
    return model


class GraphedEvalStep:
    """The evaluation forward of the reference's `validate` loop (src/engine.py:198-327: model.eval(), no_grad, AMP
    autocast, logits out) as ONE hipGraph replay on a static batch: `logits = step(images)`.  model.eval() puts the
    random-shift layers into mode 0 (longformer2d.py:114-123), so nothing is drawn per replay."""

    def __init__(self, model, images, amp_dtype=torch.bfloat16, warmup=2):
        self.model, self.amp = model.eval(), amp_dtype
        dev = images.device
        self.x = torch.empty_like(images)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            self.x.copy_(images)
            for _ in range(warmup):
                self._fwd()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        mode = "thread_local" if dist.is_available() and dist.is_initialized() else "global"
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode=mode):
            self.out = self._fwd()

    def _fwd(self):
        with torch.no_grad(), torch.autocast("cuda", dtype=self.amp, enabled=self.amp is not None):
            return self.model(self.x)

    def __call__(self, images):
        self.x.copy_(images, non_blocking=True)
        self.graph.replay()
        return self.out


def MasterWeightAdamW(model, **kw):
    return MasterWeightOptimizer(model, kind="adamw", **kw)


def wrap_ddp(model, device, world):
    if world <= 1:
        return model
    ids = [device.index] if device.type == "cuda" else None
    return torch.nn.parallel.DistributedDataParallel(
        model, device_ids=ids, broadcast_buffers=False, gradient_as_bucket_view=True, bucket_cap_mb=32)


class SyntheticBatches:
    """ImageNet-shape batches resident on the device: N(0,1) images, label-smoothed
    one-hot soft targets (what the reference's mixup path feeds the loss)."""

    def __init__(self, batch, img_size, device, rank=0, num_classes=1000, n_distinct=8, smoothing=0.1):
        g = torch.Generator(device="cpu").manual_seed(1234 + rank)
        self.items = []
        for _ in range(n_distinct):
            x = torch.randn(batch, 3, img_size, img_size, generator=g)
            y = torch.randint(0, num_classes, (batch,), generator=g)
            t = torch.full((batch, num_classes), smoothing / num_classes)
            t.scatter_(1, y[:, None], 1.0 - smoothing + smoothing / num_classes)
            self.items.append((x.to(device), t.to(device)))
        self.i = 0

    def next(self):
        self.i = (self.i + 1) % len(self.items)
        return self.items[self.i]


def soft_target_cross_entropy(logits, target):
    return torch.sum(-target * F.log_softmax(logits.float(), dim=-1), dim=-1).mean()


def train_step(model, optimizer, images, targets, amp_dtype=torch.bfloat16, scaler=None):
    """forward + backward (DDP all-reduce overlaps) + optimizer step; returns the
    loss tensor without synchronising.  `scaler`: a torch.amp.GradScaler for fp16 autocast -- the reference's loop
    (src/engine.py:84-100: scaler.scale(loss).backward(); scaler.step(optimizer); scaler.update())."""
    dev_type = images.device.type
    with torch.autocast(dev_type, dtype=amp_dtype, enabled=amp_dtype is not None):
        loss = soft_target_cross_entropy(model(images), targets)
    optimizer.zero_grad(set_to_none=True)
    if scaler is not None:
        scaler.scale(loss).backward()
        scaler.step(optimizer)
        scaler.update()
    else:
        loss.backward()
        optimizer.step()
    return loss.detach()


def _dist_on(world=None):
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() if world is None else world) > 1


def _optimizer_tensors(optimizer):
    """every tensor a replica's optimizer owns, in a rank-independent order: fp32 masters, then the state tensors of the
    parameters in param-group order (moments / momentum buffers), then tensor learning rates"""
    out = []
    inner = getattr(optimizer, "opt", optimizer)
    out += list(getattr(optimizer, "master", []))
    for group in inner.param_groups:
        for p in group["params"]:
            st = inner.state.get(p, {})
            out += [st[k] for k in sorted(st) if torch.is_tensor(st[k])]
        if torch.is_tensor(group.get("lr")):
            out.append(group["lr"])
    for bucket in getattr(inner, "_plans", {}).values():       # device step counters of the HIP optimizer
        out.append(bucket.steps)
    return out


@torch.no_grad()
def sync_replicas(model, optimizer=None, src=0):
    """What DistributedDataParallel's constructor does for the reference (src/run_experiment.py:146-153), for the
    graphed step: every rank takes rank `src`'s parameters, buffers, fp32 masters, optimizer state and tuned
    weight-gradient plans.  A collective: every rank of the default process group must call it at the same point."""
    if not _dist_on():
        return
    tensors = [p.data for p in model.parameters()] + [b for b in model.buffers()]
    if optimizer is not None:
        tensors += _optimizer_tensors(optimizer)
    # every rank must issue the SAME sequence of broadcasts: lazily created optimizer state (one rank resumed a checkpoint,
    # another already stepped) would make the lists differ and the ranks hang in mismatched collectives.  Agree on the
    # list's shape first (count and total elements, min == max over ranks) and fail loudly instead.
    dev = tensors[0].device if tensors else torch.device("cpu")
    sig = torch.tensor([len(tensors), sum(int(t.numel()) for t in tensors)], dtype=torch.int64, device=dev)
    lo, hi = sig.clone(), sig.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    if not torch.equal(lo, hi):
        raise RuntimeError(f"sync_replicas: the ranks hold different tensor lists (count / elements {sig.tolist()} here, "
                           f"{lo.tolist()} .. {hi.tolist()} over ranks): create the optimizer state on every rank first "
                           "(optimizer.allocate() / one step) before synchronising")
    for t in tensors:
        dist.broadcast(t, src)
    from . import linear
    linear.sync_plans(src)


@torch.no_grad()
def replica_checksum(model, optimizer=None):
    """(min over ranks, max over ranks) of an fp64 checksum of this replica's parameters (+ masters): equal <=> the
    replicas hold the same values (bit for bit up to checksum collisions)."""
    ts = [p.data for p in model.parameters()] + list(getattr(optimizer, "master", []) if optimizer is not None else [])
    dev = ts[0].device
    acc = torch.zeros(2, dtype=torch.float64, device=dev)
    for i, t in enumerate(ts):
        f = t.detach().double().reshape(-1)
        acc[0] += f.sum()
        acc[1] += (f * f).sum() * (1.0 + (i % 7))
    lo, hi = acc.clone(), acc.clone()
    if _dist_on():
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return lo, hi


def assert_replicas_identical(model, optimizer=None, what="parameters"):
    lo, hi = replica_checksum(model, optimizer)
    if not torch.equal(lo, hi):
        raise RuntimeError(f"data-parallel replicas diverged ({what}): checksum range {lo.tolist()} .. {hi.tolist()}")


class GraphedTrainStep:
    """The training step as hipGraphs (HIP streams + graphs instead of per-op eager launches).

    A ViL-Small step is ~600 kernel launches; the host, not the GPU, bounds the eager step.
      * world == 1: ONE graph: zero-grad + forward + backward + optimizer step (fp32-master AdamW or the capturable
        fused AdamW) on static buffers; autograd writes every gradient into the graph's private memory pool
        (p.grad is None on entry, so nothing is zeroed or accumulated).
      * world > 1: the backward is cut into SEGMENTS at stage boundaries -- [last stage + norm + head], [stage
        before it], [the first stages] -- each captured as its own graph (segment 0 also holds the forward); a
        segment's graph ends by packing that segment's gradients into one flat buffer per dtype (16-byte aligned views
        that become p.grad).  Per step: replay segment 0, start the RCCL all-reduce (mean) of its flat buffers
        asynchronously (the collective runs on the process group's own stream, ordered after the replay), replay
        segment 1 while that all-reduce is on the xGMI links, ... ; wait for the collectives; replay the optimizer
        graph.  No graph contains a collective.  Only the LAST segment's all-reduce (the first stages: 1.3 MB of the
        49 MB of bf16 gradients of ViL-Small) has no backward compute left to hide under: `comm_summary()`.
    Random-shift training (mode > 0) stays graphable: the neighbour of each layer is a device word read by the
    kernels (VilAttnDesc.mode_dev), refreshed from the host before every replay."""

    def __init__(self, model, optimizer, images, targets, world=1, amp_dtype=torch.bfloat16, warmup=3, segments=3,
                 force_segments=False, sync=True, scaler=None):
        self.model, self.opt, self.world, self.amp = model, optimizer, world, amp_dtype
        # fp16 + torch.amp.GradScaler (the reference's AMP recipe): loss scaling, the non-finite check, the skipped step
        # and the scale update are all device-side, so they are captured with the step (a skipped step costs a replay,
        # not a host round trip).  The scaler's tensors must exist before capture: at least one warm-up step.
        self.scaler = scaler
        if scaler is not None and warmup < 1:
            raise ValueError("GraphedTrainStep(scaler=...) needs warmup >= 1 (the scaler creates its state lazily)")
        # Replica consistency is the step's own job (DistributedDataParallel's constructor does it for the reference,
        # src/run_experiment.py:146-153): rank 0's parameters, buffers, masters and optimizer state before the warm-up
        # steps; rank 0's tuned plans after them; and a checksum over ranks before anything is captured.
        self._sync = bool(sync) and _dist_on(world)
        if self._sync:
            sync_replicas(model, optimizer)
        # force_segments: the multi-rank structure (segment graphs, flat-gradient all-reduce per segment on the process
        # group's stream, optimizer graph) with world == 1 -- exercises the RCCL path on a single GPU
        self.segmented = world > 1 or bool(force_segments)
        dev = images.device
        self.x = torch.empty_like(images)
        self.t = torch.empty_like(targets)
        self.params = [p for p in model.parameters() if p.requires_grad]
        # ---- backward segments (world > 1): parameters by the stage they belong to, last stage first
        self.seg_params, self.cut_stages = [self.params], []
        self.flats, self.views, self.seg_flats = [], {}, [[]]
        if self.segmented:
            L = model.num_layers
            nseg = max(1, min(int(segments), L))
            stage_of = {}
            for li in range(L):
                for p in getattr(model, "layer%d" % (li + 1)).parameters():
                    stage_of[id(p)] = li
            # segment k covers stages [lo_k, hi_k]; cuts at the inputs of the last (nseg - 1) stages
            self.cut_stages = list(range(L - 1, L - nseg, -1))          # e.g. L=4, nseg=3 -> [3, 2] (0-based stage index)
            bounds = [L] + self.cut_stages + [0]
            self.seg_params = []
            for k in range(nseg):                                       # (norm / head: with the last stage)
                lo, hi = bounds[k + 1], bounds[k]
                self.seg_params.append([p for p in self.params if lo <= stage_of.get(id(p), L - 1) < hi])
            assert sum(len(ps) for ps in self.seg_params) == len(self.params)
            self.seg_flats = []
            for ps in self.seg_params:                                 # one flat buffer per (segment, dtype)
                by_dt, fl = {}, []
                for p in ps:
                    by_dt.setdefault(p.dtype, []).append(p)
                for dt, pl in by_dt.items():
                    sizes = [(p.numel() + 7) // 8 * 8 for p in pl]
                    flat = torch.zeros(sum(sizes), dtype=dt, device=dev)
                    off = 0
                    for p, n in zip(pl, sizes):
                        self.views[p] = flat[off:off + p.numel()].view_as(p)
                        off += n
                    fl.append(flat)
                self.seg_flats.append(fl)
                self.flats += fl
        # random-shift layers: the neighbour of every layer is a device word the kernels read at launch time;
        # it is drawn on the host before each replay with the reference's RNG call (one random.randrange(1, 9)
        # per layer forward, in layer order: longformer2d.py:114-123) and uploaded with one small copy
        from .longformer2d import Long2DSCSelfAttention
        self.rs_layers = [m for m in model.modules() if isinstance(m, Long2DSCSelfAttention) and m.mode > 0]
        if self.rs_layers:
            self.modes_dev = torch.ones(len(self.rs_layers), dtype=torch.int32, device=dev)
            # the host runs ahead of the device: a ring of pinned staging buffers, each guarded by an event recorded
            # after its copy, so a later step's draws never overwrite words an earlier (pending) copy still reads
            self.modes_ring = [torch.ones(len(self.rs_layers), dtype=torch.int32).pin_memory() for _ in range(4)]
            self.modes_evt = [None] * len(self.modes_ring)
            self.modes_i = 0
            for i, m in enumerate(self.rs_layers):
                if not m.fused_path_ok():
                    raise RuntimeError("random-shift layer outside the fused path cannot read a device-side neighbour "
                                       "word (mode_dev); run it with the eager step")
                m.mode_dev = self.modes_dev[i:i + 1]
        self.loss = None
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            self.x.copy_(images); self.t.copy_(targets)
            for _ in range(warmup):
                self._body(eager=True)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        if self._sync:
            from . import linear
            linear.sync_plans(0)                  # plans measured during the warm-up steps: rank 0's, on every rank
            assert_replicas_identical(model, optimizer, "after the warm-up steps, before capture")
        settle = getattr(self.opt, "settle", None) or getattr(self.opt, "after_capture", None)
        if hasattr(self.opt, "allocate"):
            self.opt.allocate()
        if settle:
            settle()
        self.graphs, self.opt_graph = [], None
        # capture_error_mode "thread_local": with a process group alive, its watchdog thread polls events (hipEventQuery)
        # at any time; under the default "global" mode such a call from ANOTHER thread invalidates the capture
        # ("operation not permitted when stream is capturing") -- a race that depends on how long the capture takes
        mode = "thread_local" if dist.is_available() and dist.is_initialized() else "global"
        if not self.segmented:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode=mode):
                self._segment(0, None)
                self._opt_step()
            self.graphs.append(g)
        else:
            state, pool = None, None
            for k in range(len(self.seg_params)):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool, capture_error_mode=mode):
                    state = self._segment(k, state)
                pool = pool or g.pool()
                self.graphs.append(g)
            del state
            self.opt_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.opt_graph, pool=pool, capture_error_mode=mode):
                self._opt_step()
        self.graph = self.graphs[0]
        if settle:
            settle()                                  # the optimizer's plan of the captured addresses is on the device
        torch.cuda.synchronize(dev)

    def _draw_modes(self):
        if self.rs_layers and self.model.training:
            import random
            k = self.modes_i
            self.modes_i = (k + 1) % len(self.modes_ring)
            if self.modes_evt[k] is not None:
                self.modes_evt[k].synchronize()          # the copy that last read this staging buffer has finished
            host = self.modes_ring[k]
            for i in range(len(self.rs_layers)):
                host[i] = random.randrange(1, 9)
            self.modes_dev.copy_(host, non_blocking=True)
            if self.modes_evt[k] is None:
                self.modes_evt[k] = torch.cuda.Event()
            self.modes_evt[k].record()

    def _segment(self, k, state):
        """Segment k of the step.  k == 0: forward + loss + the backward of the last stage(s) (the whole backward when
        there is one segment); k > 0: the backward from the previous cut down to the next one.  `state` carries the
        cut activations and the gradient flowing into the current cut.  p.grad is WRITTEN (never accumulated)."""
        nseg = len(self.seg_params)
        if k == 0:
            for p in self.params:
                p.grad = None
            self.model._seg_points = {} if nseg > 1 else None
            with torch.autocast("cuda", dtype=self.amp, enabled=self.amp is not None):
                loss = soft_target_cross_entropy(self.model(self.x), self.t)
            self.loss = loss.detach()
            cuts = self.model._seg_points
            self.model._seg_points = None
            if self.scaler is not None:
                loss = self.scaler.scale(loss)
            if nseg == 1:
                loss.backward()
                return None
            outs, gouts = [loss], None
        else:
            cuts, outs, gouts = state["cuts"], [state["cut_act"]], [state["cut_grad"]]
        ps = self.seg_params[k]
        nxt = cuts[self.cut_stages[k]] if k < nseg - 1 else None          # activation entering this segment's first stage
        ins = ([nxt] if nxt is not None else []) + ps
        grads = torch.autograd.grad(outs, ins, grad_outputs=gouts, allow_unused=True)
        gp = grads[1:] if nxt is not None else grads
        with torch.no_grad():                       # pack for the all-reduce (multi-tensor copy); views become p.grad
            pairs = [(self.views[p], g) for p, g in zip(ps, gp) if g is not None]
            if pairs:
                torch._foreach_copy_([a for a, _ in pairs], [b for _, b in pairs])
            for p, g in zip(ps, gp):
                p.grad = self.views[p] if g is not None else None
        return {"cuts": cuts, "cut_act": nxt, "cut_grad": grads[0]} if nxt is not None else None

    def _allreduce(self, flats, async_op=False):
        # gloo (single-device tests) has no AVG; without a process group (unit test with the collective
        # stubbed) there is no backend to ask
        avg = dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl"
        works = []
        for f in flats:
            if avg:
                w = dist.all_reduce(f, op=dist.ReduceOp.AVG, async_op=async_op)
            else:
                w = dist.all_reduce(f, async_op=False)
                if self.world > 1:
                    f.div_(self.world)
            if async_op and w is not None:
                works.append(w)
        return works

    def comm_summary(self):
        """Bytes all-reduced per step and segment, and the bytes whose all-reduce has no backward compute left to run
        under (the last segment's)."""
        seg = [sum(f.numel() * f.element_size() for f in fl) for fl in self.seg_flats]
        return {"collective": "all-reduce (mean) of one flat gradient buffer per (backward segment, dtype)",
                "segments_bytes": seg, "total_bytes": sum(seg), "exposed_bytes": seg[-1] if seg else 0,
                "overlap": "segment k's all-reduce runs on the process group's stream under segment k+1's graph replay"}

    def measure_allreduce(self, reps=5):
        """milliseconds of one segment's all-reduce ALONE (nothing else on the device), per segment: the cost the
        overlap has to hide.  Leaves the flat buffers averaged `reps` times over (gradients are rewritten every step)."""
        if not (self.segmented and dist.is_available() and dist.is_initialized()):
            return None
        out = []
        for fl in self.seg_flats:
            self._allreduce(fl)
            torch.cuda.synchronize()
            dist.barrier()
            t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(reps):
                self._allreduce(fl)
            t1.record()
            torch.cuda.synchronize()
            out.append(round(t0.elapsed_time(t1) / reps, 4))
        return out

    def _opt_step(self):
        if self.scaler is not None:
            self.scaler.step(self.opt)
            self.scaler.update()
        else:
            self.opt.step()

    def _body(self, eager):
        """the same step launched op by op (warm-up, and the per-kernel profile of bench.py)"""
        self._draw_modes()
        state = None
        for k in range(len(self.seg_params)):
            state = self._segment(k, state)
            if self.segmented:
                self._allreduce(self.seg_flats[k])
        self._opt_step()

    def _sync_lr(self):
        """A scheduler that assigns param_group["lr"] = float between replays (the reference's pattern,
        src/engine.py:60-135 with its per-iteration scheduler step) is followed: changed Python-float learning rates are
        written into the device scalars the captured optimizer launch reads (one tiny fill per changed group)."""
        for o in (self.opt, getattr(self.opt, "opt", None)):
            if hasattr(o, "sync_lr"):
                o.sync_lr()

    def __call__(self, images, targets):
        self.x.copy_(images, non_blocking=True)
        self.t.copy_(targets, non_blocking=True)
        self._sync_lr()
        self._draw_modes()
        if self.opt_graph is None:
            self.graphs[0].replay()
            return self.loss
        works = []
        for k, g in enumerate(self.graphs):
            g.replay()
            # asynchronous: the collective is ordered after this replay on the process group's stream and overlaps
            # the next segment's replay on the compute stream
            works += self._allreduce(self.seg_flats[k], async_op=True)
        for w in works:
            w.wait()                              # compute stream waits for the collectives (no host block with nccl)
        self.opt_graph.replay()
        return self.loss
