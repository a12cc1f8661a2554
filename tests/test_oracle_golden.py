"""CPU: the oracle restatement (oracle/vil_oracle.py) against the golden vectors
frozen from the real reference by tools/gen_golden.py.  This is what pins the
oracle on machines where /root/reference does not exist."""
import os

import numpy as np
import pytest
import torch

import golden_cases as GC
from oracle import vil_oracle as O


@pytest.fixture(scope="module")
def gold(golden_dir):
    return {n: np.load(os.path.join(golden_dir, n + ".npz")) for n in
            ("rel_index", "masks", "op_cases", "module_cases", "dense_cases")}


@pytest.mark.parametrize("W", [2, 3, 4, 6, 7, 8, 12])
def test_relative_position_index(gold, W):
    ref = torch.from_numpy(gold["rel_index"][f"W{W}"].astype(np.int64))
    assert torch.equal(ref, O.relative_position_index(W))


@pytest.mark.parametrize("grid", GC.MASK_GRIDS, ids=lambda g: "g%dx%dp%dx%dw%d" % g)
def test_masks_bit_exact(gold, grid):
    mx, my, padx, pady, W = grid
    W2 = W * W
    for exact in (0, -1, 1):
        for mode in GC.MODES:
            if exact == 1 and mode != 0:
                continue
            kv = {0: 9 * W2, -1: W2}.get(mode, 2 * W2)
            key = f"g{mx}x{my}p{padx}x{pady}w{W}e{exact}m{mode}"
            n = mx * my * W2 * kv
            ref = np.unpackbits(gold["masks"][key])[:n].astype(bool).reshape(mx, my, W2, kv)
            m, ninv = O.invalid_mask(mx, my, padx, pady, W, exact, mode)
            mine = (m.view(mx, my, W2, kv) if m.dim() == 3 else m.view(mx, my, 1, kv).expand(mx, my, W2, kv))
            assert np.array_equal(ref, mine.numpy()), key
            assert int(gold["masks"][key + "_n"]) == ninv, key


def test_exact_with_mode_raises():
    t = torch.zeros(1, 2, 2, 16, 32)
    with pytest.raises(ValueError):
        O.mask_invalid_locations(t, 2, 2, 0, 0, 4, 1, 3)
    with pytest.raises(ValueError):
        O.mask_invalid_locations(t, 2, 2, 0, 0, 4, 2, 0)


@pytest.mark.parametrize("case", GC.OP_CASES, ids=lambda c: c[0])
def test_operator_level(gold, case):
    name, BH, M, mx, my, W = case
    for mode in GC.MODES:
        q, k, v = GC.op_inputs(case)
        g = torch.Generator().manual_seed(GC.SEED + 1)
        gout = torch.randn(q.shape, generator=g, dtype=torch.float64)
        qq, kk, vv = (t.clone().requires_grad_(True) for t in (q, k, v))
        attn = O.slidingchunk_2d(qq, kk, False, mode)
        a2 = attn.clone()
        O.mask_invalid_locations(a2, mx, my, 0, 0, W, 0, mode)
        out = O.slidingchunk_2d(torch.softmax(a2, dim=-1), vv, True, mode)
        (out * gout).sum().backward()
        pre = f"{name}_m{mode}_"
        for nm, t in zip(("attn", "out", "dq", "dk", "dv"), (attn, out, qq.grad, kk.grad, vv.grad)):
            ref = torch.from_numpy(gold["op_cases"][pre + nm]).double()
            # fixtures are stored in fp32: compare at fp32 resolution
            torch.testing.assert_close(t.detach(), ref, rtol=2e-6, atol=2e-6, msg=pre + nm)


def _check(gold, pre, nm, t, rtol=1e-9, atol=1e-10):
    mc = gold["module_cases"]
    if pre + nm in mc.files:
        torch.testing.assert_close(t, torch.from_numpy(mc[pre + nm]), rtol=rtol, atol=atol, msg=pre + nm)
    else:
        s, sums = GC.sample_big(t)
        torch.testing.assert_close(s, torch.from_numpy(mc[pre + nm + "@sample"]), rtol=rtol, atol=atol, msg=pre + nm)
        torch.testing.assert_close(sums, torch.from_numpy(mc[pre + nm + "@sums"]), rtol=1e-8, atol=1e-8, msg=pre + nm)


@pytest.mark.parametrize("c", GC.MODULE_CASES, ids=lambda c: c["name"])
def test_module_level(gold, c):
    params, x, dout = GC.module_inputs(c)
    op = {k: v.clone().requires_grad_(True) for k, v in params.items() if "_global" not in k or not c["sharew"]}
    full = dict(op)
    if c["G"] >= 1 and c["sharew"]:
        for nm in ("query", "kv", "proj"):
            full[nm + "_global.weight"] = op[nm + ".weight"]
            full[nm + "_global.bias"] = op[nm + ".bias"]
    xo = x.clone().requires_grad_(True)
    out = O.long2dsc_forward(full, xo, c["nx"], c["ny"], num_heads=c["H"], w=c["W"], nglo=c["G"],
                             rpe=c["rpe"], exact=c["exact"], mode=c["mode"], only_glo=c["only_glo"])
    (out * dout).sum().backward()
    pre = c["name"] + "/"
    _check(gold, pre, "out", out.detach())
    _check(gold, pre, "dx", xo.grad)
    mc = gold["module_cases"]
    checked = 0
    for n, p in op.items():
        if (pre + "d_" + n) in mc.files or (pre + "d_" + n + "@sample") in mc.files:
            _check(gold, pre, "d_" + n, p.grad)
            checked += 1
    assert checked >= 6


@pytest.mark.parametrize("c", GC.DENSE_CASES, ids=lambda c: c["name"])
def test_dense_attention_module_level(gold, c):
    """oracle.dense_module_forward (the s0 stages' Attention, msvit.py:37-120) against the reference module's outputs
    and gradients frozen in tests/golden/dense_cases.npz."""
    params, x, dout = GC.dense_inputs(c)
    op = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    xo = x.clone().requires_grad_(True)
    out = O.dense_module_forward(op, xo, c["nx"], c["nx"], num_heads=c["H"], nglo=c["G"], rpe=True)
    (out * dout).sum().backward()
    dc = gold["dense_cases"]
    pre = c["name"] + "/"

    def chk(nm, t):
        if pre + nm in dc.files:
            torch.testing.assert_close(t, torch.from_numpy(dc[pre + nm]), rtol=1e-9, atol=1e-10, msg=pre + nm)
        else:
            s_, sums = GC.sample_big(t)
            torch.testing.assert_close(s_, torch.from_numpy(dc[pre + nm + "@sample"]), rtol=1e-9, atol=1e-10, msg=pre + nm)
            torch.testing.assert_close(sums, torch.from_numpy(dc[pre + nm + "@sums"]), rtol=1e-8, atol=1e-8, msg=pre + nm)

    chk("out", out.detach())
    chk("dx", xo.grad)
    for n, p in op.items():
        chk("d_" + n, p.grad)


@pytest.mark.parametrize("c", [c for c in GC.MODULE_CASES if c["name"] not in GC.BIG_CASES
                               and c["exact"] in (0, 1) and not c["only_glo"]],
                         ids=lambda c: c["name"])
def test_dense_closed_form_matches_chunked(c):
    """Two independent statements of the same function."""
    g = torch.Generator().manual_seed(7)
    B, H, M, G = 2, c["H"], c["dim"] // c["H"], c["G"]
    Nloc = c["nx"] * c["ny"]
    q = torch.randn(B, H, Nloc, M, generator=g, dtype=torch.float64)
    k = torch.randn(B, H, G + Nloc, M, generator=g, dtype=torch.float64)
    v = torch.randn(B, H, G + Nloc, M, generator=g, dtype=torch.float64)
    table = torch.randn((4 * c["W"] - 1) ** 2, H, generator=g, dtype=torch.float64) * 0.5
    g2l = torch.randn(H, G, generator=g, dtype=torch.float64) * 0.5
    kw = dict(mode=c["mode"], exact=c["exact"], bias_table=table, g2l_bias=g2l if G else None)
    a = O.local_attention(q, k, v, c["nx"], c["ny"], c["W"], G, **kw)
    b = O.local_attention_dense(q, k, v, c["nx"], c["ny"], c["W"], G, **kw)
    torch.testing.assert_close(a, b, rtol=1e-10, atol=1e-12)


def test_optimizer_oracle_reproduces_the_reference_fixtures():
    """oracle/optim_oracle.py (AdamW with the reference's eps / decay placement, QHM) against the parameters the
    imported reference optimizers produced after 1 and 5 steps (tools/gen_golden_optim.py): bit-identical on CPU."""
    import optim_cases as OC
    from oracle import optim_oracle as OO
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "optim_reference.npz"))

    def ora(kind, groups, hyper):
        return OO.AdamW(groups, **hyper) if kind == "adamw" else OO.QHM(groups, **hyper)

    torch.set_num_threads(1)
    for name in OC.CASES:
        got = OC.run_case(ora, name)
        for k in (0, OC.NSTEPS - 1):
            for i, t in enumerate(got[k]):
                assert np.array_equal(t.numpy(), gold[f"{name}/step{k + 1}/p{i}"]), (name, k, i)


def test_reference_adamw_differs_from_torch_adamw():
    """Why the build does not use torch.optim.AdamW: eps placement and decay order differ measurably."""
    import optim_cases as OC
    from oracle import optim_oracle as OO
    kind, hyper, wds = OC.CASES["adamw_recipe"]
    ref = OC.run_case(lambda k, g, h: OO.AdamW(g, **h), "adamw_recipe")
    tor = OC.run_case(lambda k, g, h: torch.optim.AdamW(g, lr=h["lr"], betas=h["betas"], eps=h["eps"]), "adamw_recipe")
    assert max(float((a - b).abs().max()) for a, b in zip(ref[-1], tor[-1])) > 1e-7
