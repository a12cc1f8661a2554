"""GPU (-m gpu), collected FIRST: the HIP path, called through the C ABI (ctypes -> libvilattn.so), against the CPU
oracle on the same seeded inputs, against the golden fixtures frozen from the reference, and -- at BASELINE's full
sizes -- through size-independent properties.  Nothing in this file depends on the training engine: an engine or
glue regression (tests/test_gpu_2_glue.py, test_gpu_3_engine.py) cannot hide oracle parity under `-x`.
Tolerances: tests/gpu_common.py."""
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


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "the gpu-marked tests need a GPU"
    return torch.device("cuda:0")


def test_layout_probe(dev):
    """Hardware check of the MFMA fragment / ds_read_b64_tr_b16 layouts the kernels assume."""
    exe = os.path.join(ROOT, "vision-longformer_amd", "probe_layout")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    report("probe_layout:\n" + r.stdout + r.stderr)
    assert r.returncode == 0 and "FAIL" not in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("c", SMALL, ids=cid)
def test_scalar_f32_vs_oracle(c, dev):
    inp = make_inputs(c, torch.float32)
    ref = run_oracle(c, *inp)
    got = run_hip(c, *inp, torch.float32, "scalar", dev)
    compare("scalar/f32 " + cid(c), got, ref, F32_TOL)


@pytest.mark.parametrize("c", SMALL, ids=cid)
def test_scalar_bf16_vs_oracle(c, dev):
    inp = make_inputs(c, torch.bfloat16)
    ref = run_oracle(c, *inp)
    got = run_hip(c, *inp, torch.bfloat16, "scalar", dev)
    compare("scalar/bf16 " + cid(c), got, ref, BF16_TOL)


# >= 32 chunks at head_dim <= 32 (also the shapes of the backward's optional by-product mode, -DVIL_GLO_FROM_DQ=1)
MANY_CHUNKS = [case(2, 16, 2, 12, 13, 2), case(1, 32, 3, 18, 20, 4), case(2, 32, 2, 16, 16, 1, exact=-1),
               case(2, 32, 3, 19, 18, 3, mode=5), case(2, 16, 2, 12, 12, 2, only_glo=True), case(3, 32, 4, 24, 23, 1, exact=1, B=1)]
MFMA_CASES = [c for c in SMALL if c["M"] in (16, 32, 48, 64)] + MANY_CHUNKS      # incl. cyclic padding (exact=-1) and only_glo


@pytest.mark.parametrize("c", MFMA_CASES, ids=cid)
def test_mfma_bf16_vs_oracle(c, dev):
    """MFMA forward AND backward (backend forced: unsupported shapes would raise) on bf16 I/O."""
    inp = make_inputs(c, torch.bfloat16)
    ref = run_oracle(c, *inp)
    got = run_hip(c, *inp, torch.bfloat16, "mfma", dev)
    compare("mfma/bf16 " + cid(c), got, ref, BF16_TOL)


F32_MFMA_CASES = [c for c in SMALL if c["M"] in (16, 32, 48, 64)] + [
    case(12, 64, 8, 24, 24, 0, rpe=False), case(3, 32, 7, 28, 28, 1, B=3), case(2, 64, 12, 30, 25, 2, B=1), case(1, 16, 5, 11, 13, 3),
]


@pytest.mark.parametrize("c", F32_MFMA_CASES, ids=cid)
def test_f32_matrix_core_family_vs_oracle(c, dev):
    """fp32 I/O on v_mfma_f32_16x16x4_f32 (exact fp32 products): forward and backward for every case (bias table, masks,
    modes, cyclic padding, global keys) on the matrix cores -- round 4: including d(bias table) / d(g2l), from the dQ
    pass's fixed-point histogram (no k_scalar_* launch any more).  fp32 tolerances."""
    from vision_longformer_amd import _lib
    q, kv, table, g2l, dout = make_inputs(c, torch.float32)
    ref = run_oracle(c, q, kv, table, g2l, dout)
    # (1) forced matrix-core family, no bias parameters: forward + backward
    c0 = dict(c, rpe=False)
    ref0 = run_oracle(c0, q, kv, None, None, dout)
    _lib.profile_begin(64)
    got0 = run_hip(c0, q, kv, None, None, dout, torch.float32, "mfma", dev)
    names = [r[0] for r in _lib.profile_end(64)]
    assert "k_mfma_fwd" in names and "k_mfma_bwd_dq" in names and "k_mfma_bwd_dkdv" in names, names
    compare("f32 matrix-core fwd+bwd (no bias) " + cid(c0), got0, ref0, F32_TOL)
    # (2) AUTO with the case's bias parameters: the matrix-core family end to end, bias gradients included
    _lib.profile_begin(64)
    got = run_hip(c, q, kv, table, g2l, dout, torch.float32, "auto", dev)
    names = [r[0] for r in _lib.profile_end(64)]
    assert "k_mfma_fwd" in names and "k_mfma_bwd_dq" in names and "k_mfma_bwd_dkdv" in names, names
    assert not any(n.startswith("k_scalar") for n in names), names
    compare("f32 matrix-core fwd+bwd with bias gradients " + cid(c), got, ref, F32_TOL)
    if c["rpe"]:
        again = run_hip(c, q, kv, table, g2l, dout, torch.float32, "auto", dev)
        assert torch.equal(got["dtable"], again["dtable"]), "fp32 d(table) is not bit-reproducible"


def test_fp16_backward_propagates_non_finite_gradients(dev):
    """fp16 training (the reference's AMP: autocast + GradScaler, src/engine.py:84, run_experiment.py:206) relies on the
    inf / NaN of an overflowed scaled gradient reaching the parameter gradients, so that the scaler skips the step.
    An inf in dout must come out as non-finite dq / dkv (the backward kernels are built WITH NaN semantics), and the
    finite case must stay finite."""
    from vision_longformer_amd.ops import vil_local_attention
    c = case(2, 32, 7, 14, 14, 1)
    q, kv, table, g2l, dout = make_inputs(c, torch.float16)
    for poison in (None, float("inf"), float("nan")):
        qd = q.to(dev, torch.float16).requires_grad_(True)
        kvd = kv.to(dev, torch.float16).requires_grad_(True)
        tab = table.to(dev).requires_grad_(True)
        g2 = g2l.to(dev).requires_grad_(True)
        out = vil_local_attention(qd, kvd, tab, g2, nx=c["nx"], ny=c["ny"], w=c["W"], nglo=c["G"], num_heads=c["H"],
                                  mode=0, exact=0, backend="mfma")
        d = dout.to(dev, torch.float16).clone()
        if poison is not None:
            d[1, 77, 5] = poison
        out.backward(d)
        torch.cuda.synchronize()
        finite = bool(torch.isfinite(qd.grad).all() and torch.isfinite(kvd.grad).all())
        assert finite == (poison is None), f"poison {poison}: gradients finite = {finite}"


def test_mfma_forced_rescale_branch(dev):
    """The deferred-max rescale is rare on random data: force it with a spiked key
    (cdna guide 5.4 rule 26) late in the key order and check against the oracle."""
    c = case(2, 32, 7, 14, 14, 1)
    q, kv, table, g2l, dout = make_inputs(c, torch.bfloat16)
    C = c["H"] * c["M"]
    # token (13,13) is visited last by chunk (1,1); align its key with query (8,8)
    qi = 8 * 14 + 8
    ki = 1 + 13 * 14 + 13
    kv[:, ki, :C] = (q[:, qi] * 6).bfloat16().float()
    ref = run_oracle(c, q, kv, table, g2l, dout)
    got = run_hip(c, q, kv, table, g2l, dout, torch.bfloat16, "mfma", dev)
    # (the spiked key concentrates the softmax on one bf16-rounded probability: looser q / kv gradient bound)
    compare("mfma spike " + cid(c), got, ref, dict(BF16_TOL, dq=("rms", 0.2, 5e-2), dkv=("rms", 0.2, 5e-2)))


def _fuzz_cases(n=64, seed=20250926):
    import random as _r
    rng = _r.Random(seed)
    cases = []
    while len(cases) < n:
        W = rng.choice([2, 3, 4, 5, 6, 7, 8, 9, 12])
        M = rng.choice([16, 32, 48, 64])
        H = rng.choice([1, 2, 3])
        nx = rng.randint(max(1, W - 1), int(3.5 * W))
        ny = rng.randint(max(1, W - 1), int(3.5 * W))
        G = rng.choice([0, 1, 1, 2, 3, 4])
        mode = rng.choice([0, 0, 0, -1, 1, 2, 3, 4, 5, 6, 7, 8])
        exact = rng.choice([0, 0, 1, -1]) if mode == 0 else rng.choice([0, 0, -1])
        only_glo = G > 0 and rng.random() < 0.08
        cases.append(case(H, M, W, nx, ny, G, mode=mode, exact=exact, rpe=rng.random() < 0.8, only_glo=only_glo,
                          B=rng.choice([1, 2, 3])))
    return cases


@pytest.mark.parametrize("c", _fuzz_cases(), ids=cid)
def test_mfma_bf16_fuzz_vs_oracle(c, dev):
    """Seeded random walk over (heads, head_dim, window, ragged grids, global tokens, modes, exact window / cyclic
    padding, only_glo, batch): MFMA forward and backward (backend forced) against the oracle."""
    inp = make_inputs(c, torch.bfloat16, seed=GC.SEED + 1)
    ref = run_oracle(c, *inp)
    got = run_hip(c, *inp, torch.bfloat16, "mfma", dev)
    compare("fuzz mfma/bf16 " + cid(c), got, ref, BF16_TOL)


def _fuzz_full_cases(n=24, seed=777):
    import random as _r
    rng = _r.Random(seed)
    cases = []
    while len(cases) < n:
        W = rng.choice([2, 3, 4, 5, 6, 7, 8])
        M = rng.choice([16, 32, 48, 64])
        H = rng.choice([1, 2, 3])
        nx = rng.randint(max(1, W - 1), 3 * W)
        ny = rng.randint(max(1, W - 1), 3 * W)
        G = rng.choice([1, 1, 2, 3, 4])
        mode = rng.choice([0, 0, -1, 1, 3, 6, 8])
        cases.append(case(H, M, W, nx, ny, G, mode=mode, exact=0, rpe=rng.random() < 0.8, B=rng.choice([1, 2])))
    # >= 32 chunks, head_dim <= 32, with the global QUERY rows (also the shapes of -DVIL_GLO_FROM_DQ=1)
    cases += [case(2, 32, 2, 14, 13, 1, B=2), case(1, 16, 3, 18, 19, 3, mode=3, B=1), case(2, 32, 3, 20, 18, 4, B=1)]
    return cases


@pytest.mark.parametrize("c", _fuzz_full_cases(), ids=cid)
def test_full_attention_fuzz_vs_oracle(c, dev):
    """vil_full_attention (local rows + global-token query rows, backward through vil_attn_bwd_full) against the
    oracle's local rows plus a direct fp64 statement of the global rows (reference longformer2d.py:210-227)."""
    from vision_longformer_amd.ops import vil_full_attention
    B, H, M, G, W, nx, ny = c["B"], c["H"], c["M"], c["G"], c["W"], c["nx"], c["ny"]
    C, Nloc = H * M, nx * ny
    N = G + Nloc
    g = torch.Generator().manual_seed(GC.SEED + 2)
    rt = lambda t: t.bfloat16().float()
    q, kv, dout = rt(torch.randn(B, N, C, generator=g)), rt(torch.randn(B, N, 2 * C, generator=g)), rt(torch.randn(B, N, C, generator=g))
    table = torch.randn((4 * W - 1) ** 2, H, generator=g) * 0.5 if c["rpe"] else None
    g2l = torch.randn(2, H, G, generator=g) * 0.5 if c["rpe"] else None
    g2g = torch.randn(H, G, G, generator=g) * 0.5 if c["rpe"] else None
    scale = M ** -0.5
    # ---- fp64 reference
    L = [t.double().requires_grad_(True) if t is not None else None for t in (q, kv, table, g2l, g2g)]
    qh = L[0].view(B, N, H, M).transpose(1, 2)                          # (B,H,N,M)
    kvh = L[1].view(B, N, 2, H, M).permute(2, 0, 3, 1, 4)
    loc = O.local_attention(qh[:, :, G:], kvh[0], kvh[1], nx, ny, W, G, mode=c["mode"], exact=0,
                            bias_table=L[2], g2l_bias=L[3][1] if L[3] is not None else None)
    sg = scale * (qh[:, :, :G] @ kvh[0].transpose(-1, -2))              # (B,H,G,N)
    if L[3] is not None:
        sg = sg + torch.cat([L[4], L[3][0].unsqueeze(-1).expand(-1, -1, Nloc)], dim=-1).unsqueeze(0)
    glo = sg.softmax(-1) @ kvh[1]
    ref_out = torch.cat([glo, loc], dim=2).transpose(1, 2).reshape(B, N, C)
    (ref_out * dout.double()).sum().backward()
    ref = dict(out=ref_out.detach(), dq=L[0].grad, dkv=L[1].grad, dtable=L[2].grad if L[2] is not None else None,
               dg2l=L[3].grad if L[3] is not None else None, dg2g=L[4].grad if L[4] is not None else None)
    # ---- HIP
    D = [t.to(dev, torch.bfloat16 if i < 2 else torch.float32).requires_grad_(True) if t is not None else None
         for i, t in enumerate((q, kv, table, g2l, g2g))]
    out = vil_full_attention(D[0], D[1], D[2], D[3], D[4], nx=nx, ny=ny, w=W, nglo=G, num_heads=H, mode=c["mode"],
                             backend="mfma")
    out.backward(dout.to(dev, torch.bfloat16))
    torch.cuda.synchronize()
    f = lambda t: t.detach().double().cpu() if t is not None else None
    got = dict(out=f(out), dq=f(D[0].grad), dkv=f(D[1].grad), dtable=f(D[2].grad if D[2] is not None else None),
               dg2l=f(D[3].grad if D[3] is not None else None), dg2g=f(D[4].grad if D[4] is not None else None))
    tol = BF16_TOL
    compare("fuzz full mfma/bf16 " + cid(c), got, ref, tol)


@pytest.mark.parametrize("shape", [(2, 3, 32, 7, 21, 20, 0), (2, 2, 64, 7, 14, 14, 0), (1, 3, 32, 6, 18, 18, 3), (2, 2, 16, 4, 9, 10, 0),
                                   (1, 2, 64, 12, 24, 24, 0)],
                         ids=lambda s: "B%d_H%dM%d_W%d_%dx%d_m%d" % s)
def test_forward_full_equals_two_call_path(shape, dev):
    """vil_attn_fwd_full (round 5: the global token's query row as a spare query column of the forward pass + k_gq_merge)
    against the two calls it replaces, vil_attn_fwd + vil_glo_attn_fwd (reference longformer2d.py:134-227), through the
    C ABI on the same inputs: local rows within one bf16 step and their log-sum-exps to 5e-3 (the deferred-maximum
    rescale is a wave-wide decision, so the extra column can move WHEN a local column rescales -- the probabilities are
    rounded to bf16 against another maximum, not another set of keys), the
    global row and its lse within bf16 output tolerance (another summation order).  And VIL_E_BACKEND exactly where
    the row cannot ride: two global tokens, W = 8 at head_dim 32 (no free query slot)."""
    import ctypes
    from vision_longformer_amd import _lib, ops
    B, H, M, W, nx, ny, mode = shape
    G, C, Nloc = 1, H * M, nx * ny
    g = torch.Generator().manual_seed(GC.SEED + 9)
    q = torch.randn(B, G + Nloc, C, generator=g).to(dev, torch.bfloat16)
    kv = torch.randn(B, G + Nloc, 2 * C, generator=g).to(dev, torch.bfloat16)
    tab = (torch.randn((4 * W - 1) ** 2, H, generator=g) * 0.5).to(dev)
    g2l = (torch.randn(2, H, G, generator=g) * 0.5).to(dev)
    g2g = (torch.randn(H, G, G, generator=g) * 0.5).to(dev)
    cfg = ops._cfg(C, nx, ny, W, G, H, mode, 0, None)
    k, v = kv[..., :C], kv[..., C:]
    L = _lib.lib()
    res = {}
    for which in ("full", "two"):
        out = torch.zeros(B, G + Nloc, C, dtype=torch.bfloat16, device=dev)
        lse = torch.zeros(B, H, Nloc, device=dev)
        lse_g = torch.zeros(B, H, G, device=dev)
        d = ops._make_desc(q[:, G:], k, v, out[:, G:], cfg, "mfma")
        ws = ops._workspace(d, 0, dev)
        st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        P = ops._ptr
        if which == "full":
            rc = L.vil_attn_fwd_full(ctypes.byref(d), P(q), P(k), P(v), P(tab), P(g2l), P(g2g), P(out), P(lse), P(lse_g), P(ws), st)
            if W == 12 and M == 64:       # the column's image would cost resident waves there: declined, the caller makes two calls
                assert rc == _lib.VIL_E_BACKEND
                return
            _lib.check(rc)
        else:
            _lib.check(L.vil_attn_fwd(ctypes.byref(d), P(q[:, G:]), P(k), P(v), P(tab), P(g2l[1]), P(out[:, G:]), P(lse), P(ws), st))
            _lib.check(L.vil_glo_attn_fwd(ctypes.byref(d), P(q), P(k), P(v), P(g2g), P(g2l[0]), P(out), P(lse_g), st))
        torch.cuda.synchronize()
        res[which] = (out.float().cpu(), lse.cpu(), lse_g.cpu())
    (of, lf, lgf), (ot, lt, lgt) = res["full"], res["two"]
    dloc, dlse = float((of[:, G:] - ot[:, G:]).abs().max()), float((lf - lt).abs().max())
    assert dloc <= 1.6e-2 and dlse <= 5e-3, f"local rows changed: {dloc:.2e} {dlse:.2e}"
    eg = float((of[:, :G] - ot[:, :G]).abs().max())
    el = float((lgf - lgt).abs().max())
    report(f"     fwd_full vs fwd + glo_fwd {shape}: global row max|d| {eg:.2e}, lse_g max|d| {el:.2e}")
    assert eg < 2e-2 and el < 2e-3
    # where the row cannot ride
    for (G2, W2_, M2) in ((2, 7, 32), (1, 8, 32)):
        cfg2 = ops._cfg(H * M2, 16, 16, W2_, G2, H, 0, 0, None)
        q2 = torch.zeros(1, G2 + 256, H * M2, dtype=torch.bfloat16, device=dev)
        kv2 = torch.zeros(1, G2 + 256, 2 * H * M2, dtype=torch.bfloat16, device=dev)
        o2 = torch.zeros_like(q2)
        d2 = ops._make_desc(q2[:, G2:], kv2[..., :H * M2], kv2[..., H * M2:], o2[:, G2:], cfg2, "mfma")
        ws2 = ops._workspace(d2, 0, dev)
        rc = L.vil_attn_fwd_full(ctypes.byref(d2), ops._ptr(q2), ops._ptr(kv2[..., :H * M2]), ops._ptr(kv2[..., H * M2:]), None, None, None,
                                 ops._ptr(o2), ops._ptr(torch.zeros(1, H, 256, device=dev)), ops._ptr(torch.zeros(1, H, G2, device=dev)),
                                 ops._ptr(ws2), ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        assert rc == _lib.VIL_E_BACKEND, (G2, W2_, M2, rc)


# ---------------------------------------------------------------- module level vs golden
def _load_module(c, dev, dtype):
    from vision_longformer_amd.longformer2d import Long2DSCSelfAttention
    params, x, dout = GC.module_inputs(c, dtype=torch.float64)
    mod = Long2DSCSelfAttention(c["dim"], num_heads=c["H"], qkv_bias=True, w=c["W"], sharew=c["sharew"],
                                nglo=c["G"], only_glo=c["only_glo"], exact=c["exact"], rpe=c["rpe"],
                                mode=(1 if c["mode"] > 0 else c["mode"]))
    sd = mod.state_dict()
    for k in sd:
        if k in params:
            sd[k] = params[k].to(sd[k].dtype)
    mod.load_state_dict(sd)
    return mod.to(dev), x, dout


@pytest.mark.parametrize("c", GC.MODULE_CASES, ids=lambda c: c["name"])
def test_module_fp32_vs_golden(c, dev, golden_dir):
    import random
    gold = np.load(os.path.join(golden_dir, "module_cases.npz"))
    mod, x, dout = _load_module(c, dev, torch.float32)
    mod.backend = "scalar"
    mod.train()
    orig = random.randrange
    random.randrange = (lambda a, b=None, _m=c["mode"]: _m)
    try:
        xd = x.float().to(dev).requires_grad_(True)
        out = mod(xd, c["nx"], c["ny"])
        out.backward(dout.float().to(dev))
    finally:
        random.randrange = orig
    torch.cuda.synchronize()
    pre = c["name"] + "/"

    def check(nm, t, atol, rtol):
        t = t.detach().double().cpu()
        if pre + nm in gold.files:
            ref = torch.from_numpy(gold[pre + nm])
            torch.testing.assert_close(t, ref, atol=atol, rtol=rtol, msg=lambda m: f"{pre}{nm}: {m}")
        else:
            s, _ = GC.sample_big(t)
            ref = torch.from_numpy(gold[pre + nm + "@sample"])
            torch.testing.assert_close(s, ref, atol=atol, rtol=rtol, msg=lambda m: f"{pre}{nm}: {m}")

    check("out", out, 1e-4, 1e-4)
    check("dx", xd.grad, 3e-4, 1e-3)
    for n, p_ in mod.named_parameters():
        if p_.grad is not None and ((pre + "d_" + n) in gold.files or (pre + "d_" + n + "@sample") in gold.files):
            scale = max(1.0, float(p_.grad.abs().max()))
            check("d_" + n, p_.grad, 2e-3 * scale, 2e-3)
    report("ok   module/f32 " + c["name"])


@pytest.mark.parametrize("c", GC.MODULE_CASES, ids=lambda c: c["name"])
def test_module_bf16_autocast_vs_golden(c, dev, golden_dir):
    """bf16 autocast through the module (MFMA family wherever it applies) against the REFERENCE's fp64 fixtures:
    the forward output AND, element-wise, dx and every parameter gradient (not just norms: a permutation or sign
    error inside a gradient tensor would keep its norm)."""
    import random
    gold = np.load(os.path.join(golden_dir, "module_cases.npz"))
    mod, x, dout = _load_module(c, dev, torch.float32)
    mod.train()
    orig = random.randrange
    random.randrange = (lambda a, b=None, _m=c["mode"]: _m)
    try:
        xd = x.float().to(dev).requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = mod(xd, c["nx"], c["ny"])
        out.backward(dout.to(dev, out.dtype))
    finally:
        random.randrange = orig
    torch.cuda.synchronize()
    pre = c["name"] + "/"

    def ref_of(nm, t):
        t = t.detach().double().cpu()
        if pre + nm in gold.files:
            return t, torch.from_numpy(gold[pre + nm])
        if pre + nm + "@sample" in gold.files:
            return GC.sample_big(t)[0], torch.from_numpy(gold[pre + nm + "@sample"])
        return None, None

    t, ref = ref_of("out", out)
    err = (t - ref).abs().max().item()
    report(f"     module/bf16-autocast {c['name']} max|err|={err:.3e} (ref max {ref.abs().max().item():.2f})")
    assert err < 0.06 * max(1.0, ref.abs().max().item())
    # backward: bf16 GEMMs + bf16 attention I/O against fp64; bounds relative to each tensor's rms
    worst = []
    got, want, tols = {}, {}, {}
    t, ref = ref_of("dx", xd.grad)
    got["dx"], want["dx"], tols["dx"] = t, ref, ("rms", 0.08, 5e-2)
    for n, p_ in mod.named_parameters():
        if p_.grad is None:
            continue
        t, ref = ref_of("d_" + n, p_.grad)
        if ref is not None:
            got["d_" + n], want["d_" + n] = t, ref
            tols["d_" + n] = ("rms", 0.1, 5e-2)
    assert len(tols) >= 4, "golden gradients missing"
    compare("module/bf16-autocast backward " + c["name"], got, want, tols)


ATTN_DROP_CASES = [c for c in GC.MODULE_CASES if c["name"] in (
    "d32h2w4_8x8_g1", "d32h2w4_10x9_g1", "d32h2w4_10x9_g1_exact1", "d32h2w4_10x9_g1_cyclic", "d48h3w3_7x7_g2_mode2",
    "d48h3w3_7x7_g2_self", "d32h2w4_10x10_g0", "d32h2w4_8x8_g1_onlyglo", "d32h2w4_8x8_g1_nosharew", "d64h2w7_16x15_g1_mode3")]


@pytest.mark.parametrize("c", ATTN_DROP_CASES, ids=lambda c: c["name"])
def test_module_attn_drop_path_vs_golden(c, dev, golden_dir):
    """attn_drop > 0 in training runs the reference's algorithm on the operator-level HIP kernels (scores materialised,
    probabilities dropped: longformer2d.py:134-229).  With a drop probability of 1e-12 nothing is dropped, so the path
    must reproduce the reference's fixtures (output, dx, every parameter gradient); with p = 0.5 it must stay finite,
    differ from the undropped output, and keep its mean (inverted dropout)."""
    import random
    gold = np.load(os.path.join(golden_dir, "module_cases.npz"))
    mod, x, dout = _load_module(c, dev, torch.float32)
    mod.train()
    mod.attn_drop.p = 1e-12
    orig = random.randrange
    random.randrange = (lambda a, b=None, _m=c["mode"]: _m)
    try:
        xd = x.float().to(dev).requires_grad_(True)
        out = mod(xd, c["nx"], c["ny"])
        out.backward(dout.float().to(dev))
        torch.cuda.synchronize()
        pre = c["name"] + "/"

        def check(nm, t, atol, rtol):
            t = t.detach().double().cpu()
            if pre + nm in gold.files:
                ref = torch.from_numpy(gold[pre + nm])
            elif pre + nm + "@sample" in gold.files:
                t, ref = GC.sample_big(t)[0], torch.from_numpy(gold[pre + nm + "@sample"])
            else:
                return 0
            torch.testing.assert_close(t, ref, atol=atol * max(1.0, float(ref.abs().max())), rtol=rtol,
                                       msg=lambda m: f"{pre}{nm}: {m}")
            return 1

        assert check("out", out, 1e-4, 1e-4) and check("dx", xd.grad, 3e-4, 1e-3)
        n_checked = sum(check("d_" + n, p_.grad, 2e-3, 2e-3) for n, p_ in mod.named_parameters() if p_.grad is not None)
        assert n_checked >= 4
        mod.attn_drop.p = 0.5
        torch.manual_seed(1)
        outs = torch.stack([mod(xd.detach(), c["nx"], c["ny"]).detach() for _ in range(16)])
        assert torch.isfinite(outs).all()
        assert float((outs[0] - out.detach()).abs().max()) > 1e-3
        if not c["only_glo"]:
            err = float((outs.mean(0) - out.detach()).abs().mean() / out.detach().abs().mean())
            assert err < 0.5, err
    finally:
        random.randrange = orig
    report("ok   module attn_drop (materialised operator path) " + c["name"])


@pytest.mark.parametrize("amp", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("c", GC.DENSE_CASES, ids=lambda c: c["name"])
def test_dense_attention_module_vs_golden(c, amp, dev, golden_dir):
    """The package's dense `Attention` (the s0 stages; csrc/vil_attn_dense.hip, or the one-chunk case of the fused kernels
    when head_dim != 64) under bf16 and fp16 autocast
    against the REFERENCE module's fixtures (tools/gen_golden.py, src/models/msvit.py:37-120): output, dx and every
    parameter gradient element-wise."""
    from vision_longformer_amd.msvit import Attention
    gold = np.load(os.path.join(golden_dir, "dense_cases.npz"))
    params, x, dout = GC.dense_inputs(c)
    mod = Attention(c["dim"], num_heads=c["H"], qkv_bias=True, rpe=True, wx=c["nx"], wy=c["nx"], nglo=c["G"])
    sd = mod.state_dict()
    for k in sd:
        if k in params:
            sd[k] = params[k].to(sd[k].dtype)
    mod.load_state_dict(sd)
    mod = mod.to(dev).train()
    xd = x.float().to(dev).requires_grad_(True)
    from vision_longformer_amd import _lib
    _lib.profile_begin(64)
    with torch.autocast("cuda", dtype=amp):
        out = mod(xd, c["nx"], c["nx"])
    out.backward(dout.to(dev, out.dtype))
    torch.cuda.synchronize()
    names = {r[0] for r in _lib.profile_end(64)}
    # the HIP path ran: the dense kernel family for head_dim 64, the one-chunk case of the sliding-chunk kernels otherwise
    fam = "k_dense_fwd" if c["dim"] // c["H"] == 64 else "k_mfma_fwd"
    assert any(n.startswith(fam) for n in names) and any("dkdv" in n for n in names), names
    pre = c["name"] + "/"

    def ref_of(nm, t):
        t = t.detach().double().cpu()
        if pre + nm in gold.files:
            return t, torch.from_numpy(gold[pre + nm])
        return GC.sample_big(t)[0], torch.from_numpy(gold[pre + nm + "@sample"])

    t, ref = ref_of("out", out)
    err = (t - ref).abs().max().item()
    assert err < 0.06 * max(1.0, ref.abs().max().item()), err
    got, want, tols = {}, {}, {}
    got["dx"], want["dx"] = ref_of("dx", xd.grad)
    tols["dx"] = ("rms", 0.08, 5e-2)
    for n, p_ in mod.named_parameters():
        got["d_" + n], want["d_" + n] = ref_of("d_" + n, p_.grad)
        tols["d_" + n] = ("rms", 0.1, 5e-2)
    compare(f"dense module/{'bf16' if amp == torch.bfloat16 else 'fp16'}-autocast " + c["name"], got, want, tols)


def test_bias_gradients_at_the_bench_batch(dev):
    """d(table) / d(g2l) are accumulated in int32 fixed point whose power-of-two scale comes from batch-wide maxima and
    from the number of contributions a bin can receive in one workgroup: check them at the BENCH batch (ViL-Small stage
    1, B = 128) against the fp64 oracle (summed over the batch in slices of 8 images), and twice for bit-reproducibility."""
    c = case(3, 32, 7, 56, 56, 1, B=128)
    q, kv, table, g2l, dout = make_inputs(c, torch.bfloat16, seed=23)
    from vision_longformer_amd.ops import vil_local_attention
    runs = []
    for _ in range(2):
        qd = q.to(dev, torch.bfloat16).requires_grad_(True)
        kvd = kv.to(dev, torch.bfloat16).requires_grad_(True)
        tab, g2 = table.to(dev).requires_grad_(True), g2l.to(dev).requires_grad_(True)
        out = vil_local_attention(qd, kvd, tab, g2, nx=56, ny=56, w=7, nglo=1, num_heads=3, mode=0, exact=0, backend="mfma")
        out.backward(dout.to(dev, torch.bfloat16))
        torch.cuda.synchronize()
        runs.append((tab.grad.clone(), g2.grad.clone()))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1]), "bias gradients not bit-reproducible"
    dt, dg = torch.zeros_like(table, dtype=torch.float64), torch.zeros_like(g2l, dtype=torch.float64)
    for b0 in range(0, 128, 8):
        cs = dict(c, B=8)
        r = run_oracle(cs, q[b0:b0 + 8], kv[b0:b0 + 8], table, g2l, dout[b0:b0 + 8])
        dt += r["dtable"]; dg += r["dg2l"]
    got = dict(dtable=runs[0][0].double().cpu(), dg2l=runs[0][1].double().cpu())
    compare("bias gradients at the bench batch (B=128, small_s1)", got, dict(dtable=dt, dg2l=dg),
            dict(dtable=BF16_TOL["dtable"], dg2l=BF16_TOL["dg2l"]))


# ---------------------------------------------------------------- full-size properties
FULL = [
    ("small_s1", case(3, 32, 7, 56, 56, 1, B=8)),
    ("small_s2", case(3, 64, 7, 28, 28, 1, B=8)),
    ("meddeep_s1_f7", case(3, 32, 7, 96, 96, 1, B=2)),
    ("meddeep_s1_f8", case(3, 32, 8, 96, 96, 1, B=2)),       # the 384 fine-tuning recipe's stage 1 (f8 / f12, reference README.md:296-301): W^2 = 64, no spare column
    ("meddeep_s2_f12", case(3, 64, 12, 48, 48, 1, B=2)),
    ("basedeep_s1_f6_rs", case(3, 32, 6, 96, 96, 1, B=2, mode=4)),
    ("basedeep_s2_f8_rs", case(3, 64, 8, 48, 48, 1, B=2, mode=6)),
]


@pytest.mark.parametrize("name,c", FULL, ids=[n for n, _ in FULL])
def test_full_size_properties(name, c, dev):
    """BASELINE shapes: (1) softmax rows sum to one: v == const -> out == const;
    (2) linearity in v; (3) MFMA forward agrees with the scalar fp32-math family;
    (4) one sampled image against the oracle."""
    from vision_longformer_amd.ops import vil_local_attention
    q, kv, table, g2l, dout = make_inputs(c, torch.bfloat16, seed=11)
    C = c["H"] * c["M"]
    kw = dict(nx=c["nx"], ny=c["ny"], w=c["W"], nglo=c["G"], num_heads=c["H"], mode=c["mode"], exact=c["exact"])
    qd, kvd = q.to(dev, torch.bfloat16), kv.to(dev, torch.bfloat16)
    tab, g2 = table.to(dev), g2l.to(dev)
    with torch.no_grad():
        kv1 = kvd.clone(); kv1[..., C:] = 0.75
        o1 = vil_local_attention(qd, kv1, tab, g2, **kw)
        assert (o1.float() - 0.75).abs().max().item() < 8e-3
        kv2 = kvd.clone(); kv2[..., C:] = kv2[..., C:] * 2
        oa = vil_local_attention(qd, kvd, tab, g2, **kw).float()
        ob = vil_local_attention(qd, kv2, tab, g2, **kw).float()
        assert (ob - 2 * oa).abs().max().item() < 4e-2
        os_ = vil_local_attention(qd, kvd, tab, g2, backend="scalar", **kw).float()
        d = (oa - os_).abs().max().item()
        report(f"     full {name}: |mfma - scalar| = {d:.3e}")
        assert d < 3e-2
    c1 = dict(c, B=1)
    ref = run_oracle(c1, q[:1], kv[:1], table, g2l, dout[:1])
    got = run_hip(c1, q[:1], kv[:1], table, g2l, dout[:1], torch.bfloat16, "mfma", dev)
    compare("full " + name, got, ref, BF16_TOL)


# ---------------------------------------------------------------- dense attention (s0 stages)
@pytest.mark.parametrize("nx,G,H,M,B,rpe", [(14, 1, 6, 64, 2, True), (7, 0, 12, 64, 2, True), (24, 1, 6, 64, 1, True),
                                             (12, 0, 12, 64, 1, True), (5, 2, 2, 16, 2, True), (14, 1, 3, 32, 2, False),
                                             (9, 1, 2, 48, 2, True)])
def test_dense_attention_one_chunk_vs_reference(dev, nx, G, H, M, B, rpe):
    from vision_longformer_amd.ops import vil_dense_attention
    g = torch.Generator().manual_seed(17)
    N, C = G + nx * nx, H * M
    qkv = torch.randn(B, N, 3 * C, generator=g).bfloat16().float()
    dout = torch.randn(B, N, C, generator=g).bfloat16().float()
    table = torch.randn((2 * nx - 1) ** 2, H, generator=g) * 0.5 if rpe else None
    g2l = torch.randn(2, H, G, generator=g) * 0.5 if (rpe and G) else None
    g2g = torch.randn(H, G, G, generator=g) * 0.5 if (rpe and G) else None
    scale = M ** -0.5
    leaves = [t.double().requires_grad_(True) if t is not None else None for t in (qkv, table, g2l, g2g)]
    ref = O.dense_attention(leaves[0], leaves[1], leaves[2], leaves[3], nx, nx, G, H, scale)      # oracle, pinned by dense_cases.npz
    (ref * dout.double()).sum().backward()
    dl = [t.to(dev, torch.bfloat16 if i == 0 else torch.float32).requires_grad_(True) if t is not None else None
          for i, t in enumerate((qkv, table, g2l, g2g))]
    out = vil_dense_attention(dl[0], dl[1], dl[2], dl[3], nx=nx, ny=nx, nglo=G, num_heads=H, scale=scale, backend="mfma")
    out.backward(dout.to(dev, torch.bfloat16))
    torch.cuda.synchronize()
    got = dict(out=out.detach().double().cpu(), dqkv=dl[0].grad.double().cpu())
    want = dict(out=ref.detach(), dqkv=leaves[0].grad)
    for nm, i in (("dtable", 1), ("dg2l", 2), ("dg2g", 3)):
        if leaves[i] is not None:
            got[nm] = dl[i].grad.double().cpu(); want[nm] = leaves[i].grad
    tol = {k: BF16_TOL[k] for k in ("out", "dqkv", "dtable", "dg2l", "dg2g")}
    compare(f"dense one-chunk nx{nx} G{G} H{H} M{M}", got, want, tol)


DENSE_FAMILY = [  # nx, ny, G, H, B, rpe, dtype
    (14, 14, 1, 6, 2, True, torch.bfloat16), (7, 7, 0, 12, 2, True, torch.bfloat16), (12, 12, 0, 12, 1, True, torch.bfloat16),
    (14, 14, 1, 6, 2, True, torch.float16), (5, 5, 2, 2, 3, True, torch.bfloat16), (9, 11, 4, 2, 2, True, torch.bfloat16),
    (14, 14, 1, 3, 2, False, torch.bfloat16), (3, 2, 1, 1, 2, True, torch.float16), (16, 16, 3, 2, 1, True, torch.bfloat16),
    (15, 17, 1, 2, 1, True, torch.float16), (1, 1, 1, 2, 2, True, torch.bfloat16), (14, 14, 0, 2, 5, True, torch.bfloat16),
    (21, 19, 2, 2, 1, True, torch.float16), (24, 24, 1, 6, 1, True, torch.bfloat16), (32, 32, 1, 1, 1, True, torch.bfloat16),
]


def _dense_family_case(dev, nx, ny, G, H, B, rpe, dtype, seed=23):
    from vision_longformer_amd.ops import vil_dense_attention
    M = 64
    g = torch.Generator().manual_seed(seed)
    N, C = G + nx * ny, H * M
    qkv = torch.randn(B, N, 3 * C, generator=g).to(dtype).float()
    dout = torch.randn(B, N, C, generator=g).to(dtype).float()
    table = torch.randn((2 * nx - 1) * (2 * ny - 1), H, generator=g) * 0.5 if rpe else None
    g2l = torch.randn(2, H, G, generator=g) * 0.5 if (rpe and G) else None
    g2g = torch.randn(H, G, G, generator=g) * 0.5 if (rpe and G) else None
    scale = M ** -0.5
    leaves = [t.double().requires_grad_(True) if t is not None else None for t in (qkv, table, g2l, g2g)]
    ref = O.dense_attention(leaves[0], leaves[1], leaves[2], leaves[3], nx, ny, G, H, scale)
    (ref * dout.double()).sum().backward()
    dl = [t.to(dev, dtype if i == 0 else torch.float32).requires_grad_(True) if t is not None else None
          for i, t in enumerate((qkv, table, g2l, g2g))]
    out = vil_dense_attention(dl[0], dl[1], dl[2], dl[3], nx=nx, ny=ny, nglo=G, num_heads=H, scale=scale, backend="dense")
    out.backward(dout.to(dev, dtype))
    torch.cuda.synchronize()
    got = dict(out=out.detach().double().cpu(), dqkv=dl[0].grad.double().cpu())
    want = dict(out=ref.detach(), dqkv=leaves[0].grad)
    for nm, i in (("dtable", 1), ("dg2l", 2), ("dg2g", 3)):
        if leaves[i] is not None:
            got[nm] = dl[i].grad.double().cpu(); want[nm] = leaves[i].grad
    return got, want, dl


@pytest.mark.parametrize("nx,ny,G,H,B,rpe,dtype", DENSE_FAMILY)
def test_dense_family_vs_oracle(dev, nx, ny, G, H, B, rpe, dtype):
    """csrc/vil_attn_dense.hip (the s0 stages' own kernels: global tokens as ordinary rows / columns, delta and the
    bias-gradient histogram inside the dQ pass) against the fp64 oracle of msvit.py:91-120, forward and every gradient"""
    got, want, _ = _dense_family_case(dev, nx, ny, G, H, B, rpe, dtype)
    tol = {k: LOW_TOL[k] for k in ("out", "dqkv", "dtable", "dg2l", "dg2g")}
    compare(f"dense family {nx}x{ny} G{G} H{H} {dtype}", got, want, tol)


def test_dense_family_bias_gradients_bit_reproducible(dev):
    """fixed-point histogram + fixed-order reduce: two runs give identical bits for every output"""
    a, _, _ = _dense_family_case(dev, 14, 14, 1, 6, 4, True, torch.bfloat16)
    b, _, _ = _dense_family_case(dev, 14, 14, 1, 6, 4, True, torch.bfloat16)
    for k in a:
        assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("nx,ny,G,H,B", [(24, 24, 1, 6, 2), (14, 14, 1, 6, 3), (21, 19, 2, 2, 1), (7, 7, 0, 12, 2), (32, 32, 1, 1, 1)])
def test_dense_forward_launch_shapes_agree_bit_for_bit(dev, nx, ny, G, H, B):
    """k_dense_fwd runs in two launch shapes (<= 8 waves per workgroup on a 64-row K/V ring; <= 4 waves on a 32-row ring,
    taken by itself only where it saves a round of workgroups: 24 x 24 at B H >= 171).  A wave's unit and its 32-key steps
    are the same in both, so outputs, lse and everything the backward derives from them must be identical bits -- and the
    narrow shape is checked against the oracle like the wide one."""
    from vision_longformer_amd import _lib
    L = _lib.lib()
    res = {}
    try:
        for mode in (0, 1):
            _lib.check(L.vil_dense_attn_set_fwd_shape(mode))
            res[mode] = _dense_family_case(dev, nx, ny, G, H, B, True, torch.bfloat16)
    finally:
        _lib.check(L.vil_dense_attn_set_fwd_shape(-1))
    for k in res[0][0]:
        assert torch.equal(res[0][0][k], res[1][0][k]), k
    tol = {k: LOW_TOL[k] for k in ("out", "dqkv", "dtable", "dg2l", "dg2g")}
    compare(f"dense family, narrow forward {nx}x{ny} G{G} H{H}", res[1][0], res[1][1], tol)
    assert L.vil_dense_attn_set_fwd_shape(2) != 0


@pytest.mark.parametrize("nx,W,M,H", [(16, 4, 32, 2), (20, 7, 64, 3), (21, 6, 32, 2)])
def test_device_side_random_shift_mode(dev, nx, W, M, H):
    """VilAttnDesc.mode_dev: the neighbour read from a device word must give exactly the result of the same
    neighbour passed in the descriptor (forward and every gradient), for all 8 neighbours."""
    from vision_longformer_amd.ops import vil_full_attention
    g = torch.Generator().manual_seed(11)
    B, G, C = 2, 1, H * M
    N = G + nx * nx
    q0 = torch.randn(B, N, C, generator=g).to(dev, torch.bfloat16)
    kv0 = torch.randn(B, N, 2 * C, generator=g).to(dev, torch.bfloat16)
    tab0 = (torch.randn((4 * W - 1) ** 2, H, generator=g) * 0.3).to(dev)
    g2l0 = (torch.randn(2, H, G, generator=g) * 0.3).to(dev)
    g2g0 = (torch.randn(H, G, G, generator=g) * 0.3).to(dev)
    dout = torch.randn(B, N, C, generator=g).to(dev, torch.bfloat16)
    word = torch.zeros(1, dtype=torch.int32, device=dev)

    def run(mode, mode_dev):
        leaves = [t.clone().requires_grad_(True) for t in (q0, kv0, tab0, g2l0, g2g0)]
        out = vil_full_attention(*leaves, nx=nx, ny=nx, w=W, nglo=G, num_heads=H, mode=mode, mode_dev=mode_dev)
        out.backward(dout)
        torch.cuda.synchronize()
        return [out.detach()] + [t.grad for t in leaves]

    for m in range(1, 9):
        ref = run(m, None)
        word.fill_(m)
        got = run(1 if m != 1 else 2, word)              # the descriptor's static mode must be ignored
        for name, a, b in zip(("out", "dq", "dkv", "dtable", "dg2l", "dg2g"), got, ref):
            if name in ("dg2l", "dg2g"):                  # float atomics: order-dependent in the last bits
                torch.testing.assert_close(a, b, atol=1e-4, rtol=1e-4, msg=f"mode {m} {name}")
            else:
                assert torch.equal(a, b), f"mode {m}: {name} differs"


# ---------------------------------------------------------------- model level (BASELINE config 1)
def test_model_vil_tiny_vs_reference_logits(dev, golden_dir):
    """ViL-Tiny 224, B=2: the build's MsViT (HIP hot path, fp32) against logits / loss /
    gradient norms produced by the REFERENCE MsViT loaded with the same state dict."""
    from vision_longformer_amd.engine import build_vil
    gold = np.load(os.path.join(golden_dir, "model_tiny.npz"))
    torch.manual_seed(0)
    model = build_vil("vil_tiny_224", drop_path_rate=0.0).double()
    with torch.no_grad():
        for n, p_ in model.named_parameters():
            if "relative_position" in n:
                p_.normal_(0, 0.3)
    model = model.float().to(dev).train()
    g = torch.Generator().manual_seed(GC.SEED)
    img = torch.randn(2, 3, 224, 224, generator=g, dtype=torch.float64).float().to(dev)
    tgt = torch.tensor([3, 977], device=dev)
    # fp32 end to end on the matrix-core family (round 4): the sliding-chunk stages INCLUDING their bias-table / g2l
    # gradients (no k_scalar_* launch), the dense stages as the one-chunk case of the same kernels (no
    # scaled_dot_product_attention with a materialised (H, N, N) bias)
    from vision_longformer_amd import _lib
    import torch.nn.functional as F_
    sdpa = F_.scaled_dot_product_attention

    def no_sdpa(*a, **k):
        raise AssertionError("fp32 dense Attention fell back to scaled_dot_product_attention")
    F_.scaled_dot_product_attention = no_sdpa
    try:
        _lib.profile_begin(4096)
        logits = model(img)
        loss = torch.nn.functional.cross_entropy(logits, tgt)
        loss.backward()
        torch.cuda.synchronize()
        names = {r[0] for r in _lib.profile_end(4096)}
    finally:
        F_.scaled_dot_product_attention = sdpa
    assert not any(n.startswith("k_scalar") for n in names), names
    assert {"k_mfma_fwd", "k_mfma_bwd_dq", "k_mfma_bwd_dkdv"} <= names, names
    ref = torch.from_numpy(gold["logits"])
    err = (logits.detach().double().cpu() - ref).abs().max().item()
    report(f"     model ViL-Tiny fp32: max|logit err| = {err:.3e}, loss {loss.item():.6f} vs {float(gold['loss']):.6f}")
    assert err < 2e-3
    assert abs(loss.item() - float(gold["loss"])) < 1e-4
    gn = dict(zip([str(s) for s in gold["grad_names"]], gold["grad_norms"]))
    def sampled(tag, k, rtol, cos_min):
        """sampled gradient ELEMENTS against the reference's (norms are blind to permutation / sign errors)"""
        worst_r, worst_c = 0.0, 1.0
        for n, p_ in model.named_parameters():
            if n in gn:
                ref_s = torch.from_numpy(gold["gsample/" + n])
                got_s = GC.model_grad_sample(p_.grad)
                err = (got_s - ref_s).abs()
                # (head.weight is heavy-tailed: two target-class rows carry the norm, and an element of such a row is
                # 0.5 * feature, whose bf16 LayerNorm error is absolute, ~4e-3 of an O(1) feature -- hence the max-abs term)
                lim = k * max(rms(ref_s), 1e-12) + rtol * ref_s.abs() + k * float(ref_s.abs().max())
                cos = float((got_s * ref_s).sum() / (got_s.norm() * ref_s.norm()).clamp_min(1e-30))
                worst_r, worst_c = max(worst_r, float((err / lim).max())), min(worst_c, cos)
                assert bool((err <= lim).all()), (tag, n, float(err.max()), rms(ref_s))
                assert cos > cos_min, (tag, n, cos)
        report(f"     model ViL-Tiny {tag}: sampled gradient elements worst err/lim {worst_r:.2f}, worst cosine {worst_c:.6f}")

    for n, p_ in model.named_parameters():
        if n in gn:
            assert abs(p_.grad.norm().item() - gn[n]) < 2e-3 * max(1.0, gn[n]), n
    sampled("fp32", 2e-3, 2e-3, 0.999999)
    # bf16 autocast, MFMA forward
    for m in model.modules():
        if hasattr(m, "backend"):
            m.backend = None
    with torch.autocast("cuda", dtype=torch.bfloat16), torch.no_grad():
        lb = model(img)
    errb = (lb.double().cpu() - ref).abs().max().item()
    report(f"     model ViL-Tiny bf16 autocast: max|logit err| = {errb:.3e} (logit range {ref.abs().max():.2f})")
    assert errb < 0.1
    # bf16 autocast TRAINING step: every fused backward (MFMA dQ / dK/dV with the global rows, dense one-chunk
    # attention, fused weight/bias gradients, tuned GEMMs, residual-LayerNorm) against the reference's gradient norms
    model.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        lossb = torch.nn.functional.cross_entropy(model(img).float(), tgt)
    lossb.backward()
    torch.cuda.synchronize()
    assert abs(lossb.item() - float(gold["loss"])) < 3e-2
    worst = 0.0
    for n, p_ in model.named_parameters():
        if n in gn:
            rel = abs(p_.grad.float().norm().item() - gn[n]) / max(gn[n], 1e-3)
            worst = max(worst, rel)
            assert rel < 8e-2, (n, p_.grad.float().norm().item(), gn[n])
    report(f"     model ViL-Tiny bf16 autocast backward: loss {lossb.item():.5f}, worst gradient-norm rel. err {worst:.3e}")
    sampled("bf16 autocast", 0.3, 0.1, 0.995)


# ---------------------------------------------------------------- operator level: the reference's own surface on HIP
@pytest.mark.parametrize("opcase", GC.OP_CASES, ids=lambda c: c[0])
def test_operator_level_golden(opcase, dev, golden_dir):
    """slidingchunk_2d / mask_invalid_locations with the reference's signatures over the HIP kernels
    (vision_longformer_amd.slidingchunk_2d), run through the protocol that froze tests/golden/op_cases.npz from the
    REAL reference (tools/gen_golden.py): scores, output and all three input gradients, every mode."""
    from vision_longformer_amd.slidingchunk_2d import slidingchunk_2d, slidingchunk_2dautograd, mask_invalid_locations
    gold = np.load(os.path.join(golden_dir, "op_cases.npz"))
    name, BH, M, mx, my, W = opcase
    for mode in GC.MODES:
        q, k, v = GC.op_inputs(opcase)
        g = torch.Generator().manual_seed(GC.SEED + 1)
        gout = torch.randn(q.shape, generator=g, dtype=torch.float64)
        for sc in (slidingchunk_2d, slidingchunk_2dautograd):
            qq, kk, vv = (t.clone().to(dev).requires_grad_(True) for t in (q, k, v))
            attn = sc(qq, kk, False, mode)
            a2 = attn.clone()
            mask_invalid_locations(a2, mx, my, 0, 0, W, 0, mode)
            out = sc(torch.softmax(a2, dim=-1), vv, True, mode)
            (out * gout.to(dev)).sum().backward()
            torch.cuda.synchronize()
            pre = f"{name}_m{mode}_"
            for nm, t in zip(("attn", "out", "dq", "dk", "dv"), (attn, out, qq.grad, kk.grad, vv.grad)):
                ref = torch.from_numpy(gold[pre + nm]).double()
                # fixtures are stored in fp32: compare at fp32 resolution
                torch.testing.assert_close(t.detach().cpu(), ref, rtol=2e-6, atol=2e-6, msg=lambda m_: f"{pre}{nm}: {m_}")
    report(f"ok   operator-level golden {name} (10 modes x 2 autograd flavours, fp64 on HIP)")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_operator_level_16bit_io(dtype, dev, golden_dir):
    """The operator surface in the reference's AMP dtypes (its SlidingChunk2D runs under @autocast on GPU,
    slidingchunk_2d.py:203,235): 16-bit I/O, fp32 accumulation inside the HIP kernels, against the reference's fp64
    fixtures at 16-bit tolerances (the reference's own profiling tolerances, test_slidingchunk_2d.py:167-175)."""
    from vision_longformer_amd.slidingchunk_2d import slidingchunk_2d, mask_invalid_locations
    gold = np.load(os.path.join(golden_dir, "op_cases.npz"))
    opcase = GC.OP_CASES[1]
    name, BH, M, mx, my, W = opcase
    for mode in (0, -1, 3, 6):
        q, k, v = (t.to(dtype) for t in GC.op_inputs(opcase))
        g = torch.Generator().manual_seed(GC.SEED + 1)
        gout = torch.randn(q.shape, generator=g, dtype=torch.float64)
        qq, kk, vv = (t.clone().to(dev).requires_grad_(True) for t in (q, k, v))
        attn = slidingchunk_2d(qq, kk, False, mode)
        assert attn.dtype == dtype
        a2 = attn.float()
        mask_invalid_locations(a2, mx, my, 0, 0, W, 0, mode)
        out = slidingchunk_2d(torch.softmax(a2, dim=-1).to(dtype), vv, True, mode)
        (out.float() * gout.float().to(dev)).sum().backward()
        torch.cuda.synchronize()
        pre = f"{name}_m{mode}_"
        k8 = 1 if dtype == torch.float16 else 4          # bf16 carries 3 mantissa bits less than the fp16 the reference bounds are for
        for nm, t, atol, rtol in (("attn", attn, 2e-2 * k8, 1e-1), ("out", out, 2e-2 * k8, 1e-1), ("dq", qq.grad, 5e-2 * k8, 2e-1),
                                  ("dk", kk.grad, 5e-2 * k8, 2e-1), ("dv", vv.grad, 5e-2 * k8, 2e-1)):
            ref = torch.from_numpy(gold[pre + nm]).double()
            torch.testing.assert_close(t.detach().double().cpu(), ref, rtol=rtol, atol=atol, msg=lambda m_: f"{pre}{nm}: {m_}")


@pytest.mark.parametrize("grid", GC.MASK_GRIDS, ids=lambda g: "g%dx%dp%dx%dw%d" % g)
def test_operator_mask_invalid_locations_bit_exact(grid, dev, golden_dir):
    """mask_invalid_locations on the device: the -inf pattern and num_invalid against the reference's masks."""
    from vision_longformer_amd.slidingchunk_2d import mask_invalid_locations
    gold = np.load(os.path.join(golden_dir, "masks.npz"))
    mx, my, padx, pady, W = grid
    W2 = W * W
    for exact in (0, -1, 1):
        for mode in GC.MODES:
            if exact == 1 and mode != 0:
                with pytest.raises(ValueError):
                    mask_invalid_locations(torch.zeros(1, mx, my, W2, 2 * W2, device=dev), mx, my, padx, pady, W, exact, mode)
                continue
            kv = {0: 9 * W2, -1: W2}.get(mode, 2 * W2)
            key = f"g{mx}x{my}p{padx}x{pady}w{W}e{exact}m{mode}"
            n = mx * my * W2 * kv
            ref = np.unpackbits(gold[key])[:n].astype(bool).reshape(mx, my, W2, kv)
            t = torch.zeros(2, mx, my, W2, kv, device=dev)
            ninv = mask_invalid_locations(t, mx, my, padx, pady, W, exact, mode)
            got = torch.isinf(t).cpu().numpy()
            assert np.array_equal(got[0], ref) and np.array_equal(got[1], ref), key
            assert bool((t[torch.isinf(t)] < 0).all())
            assert int(ninv) == int(gold[key + "_n"]), key


def _dense_masks(nx, ny, w, exact):
    """The reference test's dense masks (src/tests/test_slidingchunk_2d.py:14-35), vectorised."""
    i = torch.arange(nx * ny)
    r, c = i // ny, i % ny
    if exact:
        return ((r[:, None] - r[None, :]).abs() > w) | ((c[:, None] - c[None, :]).abs() > w)
    return ((r[:, None] // w - r[None, :] // w).abs() > 1) | ((c[:, None] // w - c[None, :] // w).abs() > 1)


@pytest.mark.parametrize("exact_sliding", [0, 1])
def test_reference_test_protocol(dev, exact_sliding):
    """The reference's own parity protocol (src/tests/test_slidingchunk_2d.py:54-183): 40x40 feature map, M=64, W=8,
    B=2, H=12, fp32, seed 300; sliding-chunk attention against dense masked attention, tolerances of :159-166
    (context atol 1e-4 / rtol 1e-5; gradients atol 1e-4 / rtol 1e-3).  Run twice: through the operator-level
    surface exactly as the reference test is written, and through the FUSED op (fp32 kernels, scale 1, no bias,
    no global token) that replaces that pipeline in the product."""
    from einops import rearrange
    import torch.nn.functional as F
    from vision_longformer_amd.slidingchunk_2d import slidingchunk_2d, mask_invalid_locations
    from vision_longformer_amd.ops import vil_local_attention
    torch.manual_seed(300)
    nx = ny = 40
    N, M, W, B, H = nx * ny, 64, 8, 2, 12
    mask = _dense_masks(nx, ny, W, exact_sliding).to(dev)[None, None]
    for it in range(2):
        query = torch.randn(B * H * N * M, device=dev).view(B, H, N, M).requires_grad_(True)
        key = torch.randn(B * H * N * M, device=dev).flip(dims=(0,)).view(B, H, N, M).requires_grad_(True)
        value = torch.randn(B * H * N * M, device=dev).view(B, H, N, M).requires_grad_(True)
        # dense reference (naive2d_matmul_qk)
        a2 = (query @ key.transpose(-2, -1)).masked_fill(mask, float("-inf"))
        c2 = torch.softmax(a2, dim=-1) @ value
        g2 = torch.autograd.grad(c2.sum(), (query, key, value))
        # (1) operator-level surface, as the reference test is written
        q_img, k_img, v_img = (rearrange(t, "b h (x y) c -> (b h) c x y", x=nx) for t in (query, key, value))
        padx, pady = (W - nx % W) % W, (W - ny % W) % W
        mx, my = (nx + padx) // W, (ny + pady) // W
        q_img, k_img, v_img = (rearrange(F.pad(t, (0, pady, 0, padx)), "b c (m x) (n y) -> b c m n (x y)", x=W, y=W)
                               for t in (q_img, k_img, v_img))
        a1 = slidingchunk_2d(q_img, k_img, False)
        mask_invalid_locations(a1, mx, my, padx, pady, W, exact=exact_sliding)
        c1 = slidingchunk_2d(torch.softmax(a1, dim=-1), v_img, True)
        c1 = rearrange(c1, "b c m n (x y) -> b (m x) (n y) c", x=W)[:, :nx, :ny].reshape(B, H, N, M)
        g1 = torch.autograd.grad(c1.sum(), (query, key, value))
        # (2) the fused op on the same q, k, v: (B, N, H*M) token-major views
        qf = query.detach().transpose(1, 2).reshape(B, N, H * M).requires_grad_(True)
        kvf = torch.cat([key.detach().transpose(1, 2).reshape(B, N, H * M),
                         value.detach().transpose(1, 2).reshape(B, N, H * M)], dim=-1).requires_grad_(True)
        c3 = vil_local_attention(qf, kvf, None, None, nx=nx, ny=ny, w=W, nglo=0, num_heads=H, mode=0,
                                 exact=exact_sliding, scale=1.0, backend="scalar")
        c3.sum().backward()
        torch.cuda.synchronize()
        c3h = c3.view(B, N, H, M).transpose(1, 2)
        g3 = (qf.grad.view(B, N, H, M).transpose(1, 2), kvf.grad[..., :H * M].reshape(B, N, H, M).transpose(1, 2),
              kvf.grad[..., H * M:].reshape(B, N, H, M).transpose(1, 2))
        for tag, c, g in (("operator", c1, g1), ("fused", c3h, g3)):
            assert torch.allclose(c, c2, atol=1e-4, rtol=1e-5), f"{tag} context it={it}: {(c - c2).abs().max().item():.3e}"
            for nm, a, b in zip(("query_grad", "key_grad", "value_grad"), g, g2):
                assert torch.allclose(a, b, atol=1e-4, rtol=1e-3), f"{tag} {nm} it={it}: {(a - b).abs().max().item():.3e}"
        report(f"ok   reference test protocol exact={exact_sliding} it={it}: operator |dc| {(c1 - c2).abs().max().item():.2e}, "
               f"fused |dc| {(c3h - c2).abs().max().item():.2e}")


# ---------------------------------------------------------------- fp16 I/O: the reference's actual AMP dtype
F16_CASES = [c for c in SMALL if c["M"] in (16, 32, 48, 64)][::2] + [case(3, 32, 7, 56, 56, 1, B=1), case(3, 64, 7, 28, 28, 1, B=1)]


@pytest.mark.parametrize("backend", ["mfma", "scalar"])
@pytest.mark.parametrize("c", F16_CASES, ids=cid)
def test_f16_vs_oracle(c, backend, dev):
    """float16 q / kv / dout (the reference trains under fp16 autocast + GradScaler, src/engine.py:84): both kernel
    families against the fp64 oracle on the same fp16-rounded inputs; output bound = the reference's own fp16
    tolerance (src/tests/test_slidingchunk_2d.py:167-175: atol 2e-2 / rtol 1e-1), gradients rms-relative as for bf16
    (fp16 has 3 more mantissa bits than bf16, so the bf16 bounds hold with margin)."""
    inp = make_inputs(c, torch.float16)
    ref = run_oracle(c, *inp)
    got = run_hip(c, *inp, torch.float16, backend, dev)
    compare(f"{backend}/f16 " + cid(c), got, ref, dict(LOW_TOL, out=(2e-2, 1e-1)))


@pytest.mark.parametrize("c", [c for c in GC.MODULE_CASES if c["name"] in
                               ("d32h2w4_8x8_g1", "d32h2w4_8x8_g1_nosharew", "d48h3w3_7x7_g2_mode7", "d32h2w4_10x9_g1_cyclic",
                                "d64h2w7_16x15_g1_mode3", "small_s1_d96h3w7_56x56_g1")], ids=lambda c: c["name"])
def test_module_fp16_autocast_vs_golden(c, dev, golden_dir):
    """The drop-in module under torch.autocast(float16) -- what the reference's engine.train does unedited
    (src/engine.py:84) -- against the reference's fp64 fixtures: output, dx and every parameter gradient."""
    import random
    gold = np.load(os.path.join(golden_dir, "module_cases.npz"))
    mod, x, dout = _load_module(c, dev, torch.float32)
    mod.train()
    orig = random.randrange
    random.randrange = (lambda a, b=None, _m=c["mode"]: _m)
    try:
        xd = x.float().to(dev).requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.float16):
            out = mod(xd, c["nx"], c["ny"])
        assert out.dtype == torch.float16
        out.backward(dout.to(dev, out.dtype))
    finally:
        random.randrange = orig
    torch.cuda.synchronize()
    pre = c["name"] + "/"

    def ref_of(nm, t):
        t = t.detach().double().cpu()
        if pre + nm in gold.files:
            return t, torch.from_numpy(gold[pre + nm])
        if pre + nm + "@sample" in gold.files:
            return GC.sample_big(t)[0], torch.from_numpy(gold[pre + nm + "@sample"])
        return None, None

    got, want, tols = {}, {}, {}
    for nm, t in [("out", out), ("dx", xd.grad)] + [("d_" + n, p_.grad) for n, p_ in mod.named_parameters() if p_.grad is not None]:
        a, b = ref_of(nm, t)
        if b is not None:
            got[nm], want[nm] = a, b
            tols[nm] = (2e-2, 1e-1) if nm == "out" else ("rms", 0.2 if nm == "dx" else 0.1, 5e-2)
    assert len(tols) >= 6
    compare("module/fp16-autocast " + c["name"], got, want, tols)
