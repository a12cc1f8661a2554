"""The reference's optimizers on the HIP multi-tensor kernel (csrc/vil_optim.hip, C ABI vil_optim_*).

`AdamW` and `QHM` keep the reference's constructor signatures, defaults, argument checks and update rules
(src/optim/optimization.py:111-193, src/optim/qhm.py:8-130) and are `torch.optim.Optimizer`s, so the reference's
schedulers, `get_opt` parameter groups (src/optim/__init__.py:24-37) and checkpoint code work on them unchanged.
One launch per step updates every parameter (all tensors of all groups that share betas / eps; lr and weight decay
are per tensor).  Device tensors only -- there is no CPU fallback.

Mixed precision: `bind_working_copy(master, low_param)` makes the step read `low_param.grad` (the 16-bit gradient
autograd produced) as the gradient of the fp32 `master` and write the updated value to `low_param` in the same pass.
`MasterWeightOptimizer` does that binding for every GEMM / conv weight of a model.

hipGraph capture: the step is one kernel launch reading a device-resident plan (tensor addresses); when `step()` runs
under stream capture the plan of the captured addresses is uploaded on a side stream, and `after_capture()` must be
called once the capture has ended (engine.GraphedTrainStep does).  `lr` may be a device tensor updated in place by the
schedule (`engine.set_lr`); a Python-float `lr` is mirrored into a device scalar the optimizer owns, so a per-iteration
schedule never rebuilds the plan and a captured step follows `group["lr"]` assignments made between replays only
through `sync_lr()` (GraphedTrainStep calls it before every replay).

Step count: the reference keeps `state[p]["step"]` per parameter (optimization.py:155-165); here ONE device counter per
launch bucket is advanced by the kernel.  The two agree whenever every parameter of a bucket has a gradient at every
step (always, in this model).  `state_dict()` writes the bucket's count into every `state[p]["step"]`, so a checkpoint
loads into the reference AdamW and vice versa (`load_state_dict` takes the largest per-parameter `step` when the
`vil_steps` entry is absent).
"""
import ctypes

import torch
from torch.optim import Optimizer

from . import _lib

_DT = {torch.float32: _lib.DTYPE_F32, torch.bfloat16: _lib.DTYPE_BF16, torch.float16: _lib.DTYPE_F16}


class _Plan:
    """device-resident descriptor table of one launch (tensors that share lr / hyper-parameters)"""

    def __init__(self, device, max_tensors, numels):
        L = _lib.lib()
        arr = (_lib.VilOptimTensor * max_tensors)()
        for i, n in enumerate(numels):
            arr[i].n = n
            arr[i].param = arr[i].grad = arr[i].state1 = 1     # (sizes only)
        self.bytes = int(L.vil_optim_plan_bytes(arr, max_tensors))
        self.dev = torch.zeros((self.bytes + 15) // 16 * 16, dtype=torch.uint8, device=device)
        self.host = torch.zeros(self.bytes, dtype=torch.uint8).pin_memory()
        self.key, self.nblocks = None, 0
        self.side = torch.cuda.Stream(device=device)
        self.pending = None
        self.captured = False       # a captured launch reads this plan at every replay: its addresses are frozen


class _Bucket:
    """the tensors that share lr / hyper-parameters: one launch per step.  Two plans: the one eager steps rebuild
    whenever autograd reallocated a gradient, and the one a captured launch reads at every replay (never touched by
    eager steps made after the capture, e.g. a profiling pass between replays).  Both are allocated by the first eager
    step; the step counter is shared."""

    def __init__(self, device, max_tensors, numels):
        self.eager = _Plan(device, max_tensors, numels)
        self.graph = _Plan(device, max_tensors, numels)
        self.steps = torch.zeros(2, dtype=torch.int32, device=device)     # [completed steps, arrival ticket]


class _VilOptimizer(Optimizer):
    _ALGO = None
    _HAS_STEP = False          # the reference class keeps a per-parameter `step` in its state (AdamW: yes, QHM: no)
    # torch.amp.GradScaler (the reference trains with fp16 autocast + GradScaler: src/engine.py:84-100): scaler.step()
    # sets `grad_scale` / `found_inf` (device tensors) on an optimizer that declares this, and leaves unscaling and the
    # skip-on-inf decision to its step() -- here they happen inside the HIP launch, without a host synchronisation
    _step_supports_amp_scaling = True

    def __init__(self, params, defaults):
        super().__init__(params, defaults)
        self._bound = {}           # id(master) -> low parameter (gradient source + 16-bit working copy)
        self._plans = {}
        self._lr_dev = {}          # id(param group) -> [device scalar mirroring a Python-float lr, its last value]

    # ---- mixed precision
    def bind_working_copy(self, master, low_param):
        if master.dtype != torch.float32 or low_param.dtype not in (torch.bfloat16, torch.float16):
            raise ValueError("bind_working_copy(master fp32, low bf16/fp16)")
        if master.shape != low_param.shape:
            raise ValueError("working copy shape mismatch")
        self._bound[id(master)] = low_param

    def _grad_of(self, p):
        low = self._bound.get(id(p))
        return (low.grad if low is not None else p.grad), low

    def _state_for(self, p, group):
        raise NotImplementedError

    def _launch(self, plan, group, stream, inv_scale, found_inf):
        raise NotImplementedError

    def _bucket_key(self, group):
        raise NotImplementedError

    def _amp_scalars(self, device):
        """(1 / loss scale, found_inf) as device float pointers, or (None, None) outside a GradScaler step"""
        gs, fi = getattr(self, "grad_scale", None), getattr(self, "found_inf", None)
        vp = ctypes.c_void_p
        inv = None
        if gs is not None:
            inv = gs.to(device=device, dtype=torch.float64).reciprocal().to(torch.float32).reshape(1)
        if fi is not None and torch.is_tensor(fi):
            fi = fi.to(device=device, dtype=torch.float32).reshape(-1)[:1].contiguous()
        else:
            fi = None
        self._amp_keep = (inv, fi)                   # alive until the launch has been enqueued (and for a captured graph's pool)
        return (vp(inv.data_ptr()) if inv is not None else None), (vp(fi.data_ptr()) if fi is not None else None)

    def _buckets(self):
        """param groups that share the kernel-wide hyper-parameters (one launch each; lr and weight decay are per
        tensor, so the usual two weight-decay groups are ONE launch)"""
        buckets = {}
        for group in self.param_groups:
            buckets.setdefault(self._bucket_key(group), []).append(group)
        return buckets

    def allocate(self):
        """Creates, eagerly, everything a step needs that must not be created inside a stream capture: the plan buffers
        of every launch bucket, every parameter's state tensors (zero moments) and the device mirrors of Python-float
        learning rates.  Call before capturing a step that was never run eagerly (GraphedTrainStep(warmup=0))."""
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("optimizer.allocate() inside a stream capture")
        for key, groups in self._buckets().items():
            ps = [p for gr in groups for p in gr["params"]]
            if not ps:
                continue
            if key not in self._plans:
                self._plans[key] = _Bucket(ps[0].device, len(ps), [p.numel() for p in ps])
                self._new_plan_steps(self._plans[key])
            for gr in groups:
                self._lr_mirror(gr, ps[0].device)
                for p in gr["params"]:
                    # (fp32 masters are requires_grad=False: their gradient is the bound 16-bit copy's -- the same
                    # rule step() applies through _grad_of)
                    if p.requires_grad or id(p) in self._bound:
                        self._state_for(p, gr)

    def _new_state(self, p):
        """zero state tensor of a parameter; never inside a capture (the zero-fill would become a graph node that
        re-zeroes the moments at every replay while the device step counter keeps advancing)"""
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("optimizer state would be created inside a stream capture: call optimizer.allocate() "
                               "(or run one eager step) before capturing the step")
        return torch.zeros_like(p, memory_format=torch.preserve_format)

    def _lr_mirror(self, group, device):
        """device scalar that carries a Python-float lr of a param group (refreshed by sync_lr); a tensor lr is its own"""
        lr = group["lr"]
        if torch.is_tensor(lr):
            return lr
        m = self._lr_dev.get(id(group))
        if m is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("call optimizer.allocate() (or run one eager step) before capturing the step")
            m = self._lr_dev[id(group)] = [torch.full((), float(lr), dtype=torch.float32, device=device), float(lr)]
        return m[0]

    def sync_lr(self):
        """writes changed Python-float learning rates into their device mirrors (one tiny fill per changed group; no host
        synchronisation, no plan rebuild).  step() calls it; under graph replay call it before the replay."""
        for group in self.param_groups:
            lr = group["lr"]
            m = self._lr_dev.get(id(group))
            if m is not None and not torch.is_tensor(lr) and float(lr) != m[1]:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("learning rate changed inside a stream capture")
                m[0].fill_(float(lr))
                m[1] = float(lr)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        buckets = self._buckets()
        for key, groups in buckets.items():
            entries = []
            for group in groups:
                for p in group["params"]:
                    g, low = self._grad_of(p)
                    if g is None:
                        continue
                    if g.is_sparse:
                        raise RuntimeError("sparse gradients are not supported")
                    if not p.is_cuda:
                        raise RuntimeError("vision_longformer_amd.optim runs on the GPU only (no CPU fallback)")
                    if p.dtype != torch.float32 or not p.is_contiguous() or not g.is_contiguous() or g.dtype not in _DT:
                        raise RuntimeError("optimizer tensors must be contiguous; parameters fp32, gradients fp32/bf16/fp16")
                    s1, s2 = self._state_for(p, group)
                    entries.append((p, g, s1, s2, low, float(group["weight_decay"]), self._lr_mirror(group, p.device)))
            if not entries:
                continue
            dev = entries[0][0].device
            bucket = self._plans.get(key)
            capturing = torch.cuda.is_current_stream_capturing()
            if not capturing:
                self.sync_lr()
            if bucket is None:
                if capturing:
                    raise RuntimeError("call optimizer.allocate() (or run one eager step) before capturing the step: "
                                       "the plan buffers cannot be allocated inside a capture")
                nall = [p.numel() for gr in groups for p in gr["params"]]
                bucket = self._plans[key] = _Bucket(dev, len(nall), nall)
                self._new_plan_steps(bucket)
            plan = bucket.graph if capturing else bucket.eager
            plan.steps = bucket.steps
            self._refresh(plan, entries)
            stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            inv_scale, found_inf = self._amp_scalars(dev)
            _lib.check(self._launch(plan, groups[0], stream, inv_scale, found_inf))
        return loss

    def _refresh(self, plan, entries):
        """(re)build and upload the plan when a tensor address changed (eager autograd reallocates gradients)"""
        key = tuple((p.data_ptr(), g.data_ptr(), low.data_ptr() if low is not None else 0, wd, lr.data_ptr())
                    for p, g, _, _, low, wd, lr in entries)
        capturing = torch.cuda.is_current_stream_capturing()
        if plan.captured and key != plan.key:
            # an earlier captured launch replays from this plan buffer (nblocks is baked into its node): overwriting it
            # would silently retarget that graph to the new addresses
            raise RuntimeError("this optimizer's step was already captured with other tensor addresses; a second capture "
                               "(another GraphedTrainStep, another batch shape) needs its own optimizer instance")
        if capturing:
            plan.captured = True
        if key == plan.key:
            return
        L = _lib.lib()
        arr = (_lib.VilOptimTensor * len(entries))()
        for i, (p, g, s1, s2, low, wd, lr) in enumerate(entries):
            a = arr[i]
            a.param, a.grad, a.state1 = p.data_ptr(), g.data_ptr(), s1.data_ptr()
            a.state2 = s2.data_ptr() if s2 is not None else None
            a.low = low.data_ptr() if low is not None else None
            a.n, a.grad_dtype = p.numel(), _DT[g.dtype]
            a.low_dtype = _DT[low.dtype] if low is not None else 0
            a.weight_decay = wd
            a.lr, a.lr_dev = 0.0, lr.data_ptr()        # always a device scalar: the group's tensor lr or its mirror
        if plan.pending is not None:
            plan.pending.synchronize()                 # the previous upload still reads the pinned buffer
            plan.pending = None
        nb = ctypes.c_int(0)
        _lib.check(L.vil_optim_plan_build(arr, len(entries), ctypes.c_void_p(plan.host.data_ptr()), plan.bytes, ctypes.byref(nb)))
        plan.nblocks, plan.key = nb.value, key
        cur = torch.cuda.current_stream(plan.dev.device)
        if torch.cuda.is_current_stream_capturing():
            # a captured launch only RECORDS the kernel; the plan it will read at replay is uploaded on a side stream
            # (no memcpy node in the graph); after_capture() waits for it
            with torch.cuda.stream(plan.side):
                plan.dev[:plan.bytes].copy_(plan.host, non_blocking=True)
                plan.pending = torch.cuda.Event()
                plan.pending.record(plan.side)
        else:
            plan.dev[:plan.bytes].copy_(plan.host, non_blocking=True)
            plan.pending = torch.cuda.Event()
            plan.pending.record(cur)

    def reset_step_count(self, value=0):
        for bucket in self._plans.values():
            bucket.steps[0] = int(value)
            bucket.steps[1] = 0

    def after_capture(self):
        for bucket in self._plans.values():
            for plan in (bucket.eager, bucket.graph):
                if plan.pending is not None:
                    plan.pending.synchronize()
                    plan.pending = None

    # ---- checkpointing: loaded moments are copied INTO the existing state tensors (a captured step keeps reading
    # the same addresses) and the lr objects (possibly device tensors a graph reads) are kept
    def _bucket_of(self, group):
        return self._plans.get(self._bucket_key(group))

    def state_dict(self):
        # the reference's per-parameter `step` (optimization.py:155-165) = the launch bucket's device counter
        if self._HAS_STEP:
            for group in self.param_groups:
                b = self._bucket_of(group)
                if b is None:
                    continue
                n = int(b.steps[0].item())
                for p in group["params"]:
                    if p in self.state and len(self.state[p]):
                        self.state[p]["step"] = n
        sd = super().state_dict()
        sd["vil_steps"] = [int(b.steps[0].item()) for b in self._plans.values()]     # completed steps per launch bucket
        return sd

    @torch.no_grad()
    def load_state_dict(self, state_dict):
        steps = state_dict.get("vil_steps")
        old_state = {id(p): dict(self.state[p]) for g in self.param_groups for p in g["params"] if p in self.state}
        lrs = [g["lr"] for g in self.param_groups]
        old_gids = [id(g) for g in self.param_groups]
        super().load_state_dict({k: v for k, v in state_dict.items() if k != "vil_steps"})
        for g, lr in zip(self.param_groups, lrs):
            if torch.is_tensor(lr):
                lr.fill_(float(g["lr"]))
                g["lr"] = lr
        # (Optimizer.load_state_dict replaces the param-group dicts: carry the lr mirrors over by position)
        self._lr_dev = {id(g): self._lr_dev[k] for g, k in zip(self.param_groups, old_gids) if k in self._lr_dev}
        for g in self.param_groups:
            for p in g["params"]:
                old, new = old_state.get(id(p)), self.state.get(p)
                if new and not old:
                    # no state tensors of our own yet: take copies (Optimizer.load_state_dict keeps the checkpoint's
                    # tensor objects when device and dtype already match -- two optimizers would share moments)
                    for k, v in list(new.items()):
                        if torch.is_tensor(v):
                            new[k] = v.clone()
                if not old or not new:
                    continue
                for k, v in list(new.items()):
                    if torch.is_tensor(v) and torch.is_tensor(old.get(k)) and old[k].shape == v.shape:
                        old[k].copy_(v)
                        new[k] = old[k]
        if steps is None:
            # a checkpoint of the reference optimizer: per-parameter `step` -> the bucket's counter (the largest one; they
            # are equal unless some parameter had no gradient on some steps)
            per_bucket = {}
            for g in self.param_groups:
                k = self._bucket_key(g)
                for p in g["params"]:
                    st = self.state.get(p)
                    if st and "step" in st:
                        per_bucket[k] = max(per_bucket.get(k, 0), int(st["step"]))
            self._loaded_steps_by_key = per_bucket
            self._loaded_steps = None
            for k, bucket in self._plans.items():
                bucket.steps[0] = int(per_bucket.get(k, 0))
                bucket.steps[1] = 0
        else:
            steps = list(steps)
            self._loaded_steps, self._loaded_steps_by_key = steps, None
            for bucket, s_ in zip(self._plans.values(), steps):
                bucket.steps[0] = int(s_)
                bucket.steps[1] = 0
        self.sync_lr()

    def _new_plan_steps(self, bucket):
        """a bucket created after load_state_dict starts from the loaded step count (creation order, or -- for a
        checkpoint of the reference optimizer -- the bucket's key)"""
        by_key = getattr(self, "_loaded_steps_by_key", None)
        if by_key:
            for k, b in self._plans.items():
                if b is bucket:
                    bucket.steps[0] = int(by_key.get(k, 0))
            return
        loaded = getattr(self, "_loaded_steps", None)
        idx = len(self._plans) - 1
        if loaded and idx < len(loaded):
            bucket.steps[0] = int(loaded[idx])


class AdamW(_VilOptimizer):
    """Adam with the reference's weight-decay fix (src/optim/optimization.py:111-193): `denom = sqrt(v) + eps`, eps
    outside the bias correction (default 1e-6), decoupled decay `p -= lr * wd * p` after the Adam update."""

    _HAS_STEP = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True):
        if not torch.is_tensor(lr) and lr < 0.0:
            raise ValueError("Invalid learning rate: {} - should be >= 0.0".format(lr))
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError("Invalid beta parameter: {} - should be in [0.0, 1.0[".format(betas[0]))
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError("Invalid beta parameter: {} - should be in [0.0, 1.0[".format(betas[1]))
        if not 0.0 <= eps:
            raise ValueError("Invalid epsilon value: {} - should be >= 0.0".format(eps))
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, correct_bias=correct_bias))

    def _bucket_key(self, group):
        return (tuple(group["betas"]), float(group["eps"]), bool(group["correct_bias"]))

    def _state_for(self, p, group):
        st = self.state[p]
        if "exp_avg" not in st:
            st["exp_avg"] = self._new_state(p)
            st["exp_avg_sq"] = self._new_state(p)
        return st["exp_avg"], st["exp_avg_sq"]

    def _launch(self, plan, group, stream, inv_scale=None, found_inf=None):
        b1, b2 = group["betas"]
        return _lib.lib().vil_optim_adamw_step_amp(ctypes.c_void_p(plan.dev.data_ptr()), plan.nblocks, b1, b2, group["eps"],
                                                    int(bool(group["correct_bias"])),
                                                    ctypes.c_void_p(plan.steps.data_ptr()), inv_scale, found_inf, stream)


class QHM(_VilOptimizer):
    """Quasi-hyperbolic momentum SGD (src/optim/qhm.py:8-130): g += wd p; h = beta h + (1 - beta) g;
    d = (1 - nu) g + nu h; p -= lr d.  nu = 1 (the 384 fine-tuning recipe, config/msvit_384finetune.yaml:28-35) is SGD
    with dampened momentum."""

    def __init__(self, params, lr=-1, momentum=0, qhm_nu=1, weight_decay=0):
        if not torch.is_tensor(lr) and lr <= 0:
            raise ValueError("Invalid value for learning rate (>0): {}".format(lr))
        if momentum < 0 or momentum > 1:
            raise ValueError("Invalid value for momentum [0,1): {}".format(momentum))
        if weight_decay < 0:
            raise ValueError("Invalid value for weight_decay (>=0): {}".format(weight_decay))
        super().__init__(params, dict(lr=lr, momentum=momentum, qhm_nu=qhm_nu, weight_decay=weight_decay))

    def _bucket_key(self, group):
        return (float(group["momentum"]), float(group["qhm_nu"]))

    def _state_for(self, p, group):
        st = self.state[p]
        if abs(group["momentum"]) < 1e-12 or abs(group["qhm_nu"]) < 1e-12:      # plain SGD: no buffer (reference)
            return p, None          # (state1 must be a valid address; never touched by the kernel in this case)
        if "momentum_buffer" not in st:
            st["momentum_buffer"] = self._new_state(p)
        return st["momentum_buffer"], None

    def _launch(self, plan, group, stream, inv_scale=None, found_inf=None):
        return _lib.lib().vil_optim_qhm_step_amp(ctypes.c_void_p(plan.dev.data_ptr()), plan.nblocks, group["momentum"],
                                                  group["qhm_nu"], ctypes.c_void_p(plan.steps.data_ptr()), inv_scale, found_inf, stream)
