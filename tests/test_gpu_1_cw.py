"""Round 6: the chunk-workgroup forward kernels (csrc/vil_attn_cw.hip) through the C ABI, selected by name (backend "mfma_cw":
AUTO / "mfma" take them only for 3x3 neighbourhoods with W <= 8, so the shapes they decline there are exercised here).

Checked against the fp64 oracle on the same rounded inputs (tests/gpu_common.py tolerances, the forward's bound: atol 2e-2 +
rtol 5e-2, log-sum-exp to 2e-2), against the wave-per-chunk kernels, and -- the hand-off to the exact kernel -- on inputs
whose logits leave the fast kernel's range.  Reference semantics: src/models/layers/longformer2d.py:134-227."""
import ctypes

import pytest
import torch

import golden_cases as GC
from gpu_common import BF16_TOL, case, cid, compare, make_inputs, report, run_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a device (no CPU fallback in the product path)"
    return torch.device("cuda:0")


# every key-list kind the family has to walk: 3x3 neighbourhoods (borders, padded grids), the own chunk only, the two-chunk
# lists of random shift, cyclic padding (zero keys), the exact window mask, no / several global keys, W = 12 (five waves per
# chunk), one-step lists (only_glo), both head dims, fp16 (exact kernel only)
CW_CASES = [
    case(2, 32, 7, 14, 14, 1), case(3, 32, 7, 21, 20, 1, B=3), case(2, 64, 7, 16, 15, 1), case(2, 32, 8, 16, 16, 1),
    case(2, 64, 8, 20, 20, 0, B=1), case(2, 32, 7, 9, 30, 3, exact=1), case(2, 32, 6, 13, 12, 1, mode=1),
    case(2, 32, 7, 16, 15, 1, mode=3), case(2, 32, 8, 16, 16, 0, mode=8), case(2, 64, 4, 10, 9, 1, exact=-1),
    case(2, 32, 4, 5, 6, 1, mode=5, exact=-1), case(3, 32, 3, 7, 7, 2, mode=-1), case(2, 64, 12, 24, 25, 1, B=1),
    case(2, 32, 4, 8, 8, 1, only_glo=True), case(2, 32, 7, 7, 7, 1), case(1, 32, 4, 3, 2, 1), case(2, 32, 5, 11, 17, 4, rpe=False),
    case(3, 32, 7, 28, 28, 1, B=9),          # B >= 8: persistent columns over an XCD's images, several streams
    case(2, 64, 7, 14, 14, 1, B=17),
    case(6, 32, 7, 14, 14, 1, B=8), case(12, 64, 8, 16, 16, 1, B=2),     # more heads than any ViL stage-1/2 (columns = chunk x head)
]


def _fwd(c, q, kv, table, g2l, dtype, backend, dev):
    """forward only through vil_attn_fwd: (out, lse) as fp64 CPU tensors"""
    from vision_longformer_amd import _lib, ops
    B, H, M, G = c["B"], c["H"], c["M"], c["G"]
    C, Nloc = H * M, c["nx"] * c["ny"]
    qd, kvd = q.to(dev, dtype), kv.to(dev, dtype)
    tab = table.to(dev).float().contiguous() if table is not None else None
    g2 = g2l.to(dev).float().contiguous() if g2l is not None else None
    out = torch.zeros(B, Nloc, C, dtype=dtype, device=dev)
    lse = torch.zeros(B, H, Nloc, device=dev)
    cfg = dict(nx=c["nx"], ny=c["ny"], W=c["W"], G=G, H=H, mode=c["mode"], exact=c["exact"], only_glo=c["only_glo"],
               scale=M ** -0.5, debug=0)
    k, v = kvd[..., :C], kvd[..., C:]
    d = ops._make_desc(qd, k, v, out, cfg, backend)
    ws = ops._workspace(d, 0, dev)
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(_lib.lib().vil_attn_fwd(ctypes.byref(d), ops._ptr(qd), ops._ptr(k), ops._ptr(v), ops._ptr(tab), ops._ptr(g2),
                                       ops._ptr(out), ops._ptr(lse), ops._ptr(ws), st))
    torch.cuda.synchronize()
    return out.double().cpu(), lse.double().cpu()


def _oracle_lse(c, q, kv, table, g2l):
    """natural-log-sum-exp of every row from the oracle's pieces (fp64)"""
    from oracle import vil_oracle as O
    B, H, M, G = c["B"], c["H"], c["M"], c["G"]
    Nloc = q.shape[1]
    qh = q.double().view(B, Nloc, H, M).transpose(1, 2)
    kvh = kv.double().view(B, G + Nloc, 2, H, M).permute(2, 0, 3, 1, 4)
    ones = torch.ones_like(kvh[1])
    return qh, kvh, ones


@pytest.mark.parametrize("c", CW_CASES, ids=cid)
def test_cw_forward_vs_oracle_and_wave_kernels(c, dev):
    q, kv, table, g2l, dout = make_inputs(c, torch.bfloat16)
    ref = run_oracle(c, q, kv, table, g2l, dout)
    out, lse = _fwd(c, q, kv, table, g2l, torch.bfloat16, "mfma_cw", dev)
    compare("cw fwd bf16 " + cid(c), dict(out=out), ref, dict(out=BF16_TOL["out"]))
    out_w, lse_w = _fwd(c, q, kv, table, g2l, torch.bfloat16, "mfma_wave", dev)
    # same math, other rounding points (Q' = Q scale log2 e rounded to bf16 once; P rounded against no running maximum)
    assert float((out - out_w).abs().max()) <= 3.2e-2, float((out - out_w).abs().max())
    assert float((lse - lse_w).abs().max()) <= 2e-2, float((lse - lse_w).abs().max())


@pytest.mark.parametrize("c", [case(2, 32, 7, 14, 14, 1), case(2, 64, 8, 16, 16, 1), case(2, 32, 6, 13, 12, 1, mode=1)], ids=cid)
def test_cw_forward_fp16_is_the_exact_kernel(c, dev):
    q, kv, table, g2l, dout = make_inputs(c, torch.float16)
    ref = run_oracle(c, q, kv, table, g2l, dout)
    out, lse = _fwd(c, q, kv, table, g2l, torch.float16, "mfma_cw", dev)
    compare("cw fwd fp16 " + cid(c), dict(out=out), ref, dict(out=BF16_TOL["out"]))
    out_w, lse_w = _fwd(c, q, kv, table, g2l, torch.float16, "mfma_wave", dev)
    assert float((lse - lse_w).abs().max()) <= 2e-3          # the same arithmetic as the wave-per-chunk kernels


def test_cw_large_logits_go_to_the_exact_kernel(dev):
    """Rows whose sums leave [2^-24, 2^24] are computed again by the exact kernel launched behind the fast one: spiked keys
    (cdna guide 5.4 rule 26) in some images, ordinary data in the others; both kinds must match the oracle, and the
    log-sum-exps must agree with the wave-per-chunk kernels' to fp32 rounding where the exact kernel ran."""
    c = case(3, 32, 7, 21, 20, 1, B=10)
    q, kv, table, g2l, dout = make_inputs(c, torch.bfloat16)
    C = c["H"] * c["M"]
    spiked = [1, 4, 9]
    for b in spiked:
        qi, ki = 8 * c["ny"] + 8, 1 + 13 * c["ny"] + 13
        kv[b, ki, :C] = (q[b, qi] * 6).bfloat16().float()
        kv[b, 1 + 2 * c["ny"] + 3, :C] = (q[b, 3 * c["ny"] + 2] * -7).bfloat16().float()      # and a very negative one
    ref = run_oracle(c, q, kv, table, g2l, dout)
    out, lse = _fwd(c, q, kv, table, g2l, torch.bfloat16, "mfma_cw", dev)
    compare("cw fwd spike " + cid(c), dict(out=out), ref, dict(out=BF16_TOL["out"]))
    out_w, lse_w = _fwd(c, q, kv, table, g2l, torch.bfloat16, "mfma_wave", dev)
    # the exact kernel takes over the COLUMN (head, chunk pair) that flagged the image, not the whole image: the spiked query
    # rows themselves must carry the wave-per-chunk kernels' arithmetic, every other row either one (2e-2 as everywhere)
    qa, qb = 8 * c["ny"] + 8, 3 * c["ny"] + 2
    d_spiked = float((lse[spiked][:, :, [qa, qb]] - lse_w[spiked][:, :, [qa, qb]]).abs().max())
    assert d_spiked <= 6e-3, d_spiked            # (bf16-rounded probabilities summed against another running maximum: own keys first)
    assert float((lse - lse_w).abs().max()) <= 3.5e-2
    report(f"     cw exact hand-off: lse of the spiked rows agrees with the wave-per-chunk kernels to {d_spiked:.1e}")


@pytest.mark.parametrize("shape", [(2, 3, 32, 7, 21, 20, 0), (2, 2, 64, 7, 14, 14, 0), (9, 3, 32, 7, 28, 28, 0)],
                         ids=lambda s: "B%d_H%dM%d_W%d_%dx%d_m%d" % s)
def test_cw_forward_full_global_row(shape, dev):
    """vil_attn_fwd_full on the chunk-workgroup kernels (the global query in a spare column, live in the own chunk's first
    steps: own keys lead the slot list) against vil_attn_fwd + vil_glo_attn_fwd."""
    from vision_longformer_amd import _lib, ops
    B, H, M, W, nx, ny, mode = shape
    G, C, Nloc = 1, H * M, nx * ny
    g = torch.Generator().manual_seed(GC.SEED + 19)
    q = torch.randn(B, G + Nloc, C, generator=g).to(dev, torch.bfloat16)
    kv = torch.randn(B, G + Nloc, 2 * C, generator=g).to(dev, torch.bfloat16)
    tab = (torch.randn((4 * W - 1) ** 2, H, generator=g) * 0.5).to(dev)
    g2l = (torch.randn(2, H, G, generator=g) * 0.5).to(dev)
    g2g = (torch.randn(H, G, G, generator=g) * 0.5).to(dev)
    cfg = ops._cfg(C, nx, ny, W, G, H, mode, 0, None)
    k, v = kv[..., :C], kv[..., C:]
    L = _lib.lib()
    P = ops._ptr
    res = {}
    for which in ("full", "two"):
        out = torch.zeros(B, G + Nloc, C, dtype=torch.bfloat16, device=dev)
        lse = torch.zeros(B, H, Nloc, device=dev)
        lse_g = torch.zeros(B, H, G, device=dev)
        d = ops._make_desc(q[:, G:], k, v, out[:, G:], cfg, "mfma_cw" if which == "full" else "mfma_wave")
        ws = ops._workspace(d, 0, dev)
        st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        if which == "full":
            _lib.check(L.vil_attn_fwd_full(ctypes.byref(d), P(q), P(k), P(v), P(tab), P(g2l), P(g2g), P(out), P(lse), P(lse_g), P(ws), st))
        else:
            _lib.check(L.vil_attn_fwd(ctypes.byref(d), P(q[:, G:]), P(k), P(v), P(tab), P(g2l[1]), P(out[:, G:]), P(lse), P(ws), st))
            _lib.check(L.vil_glo_attn_fwd(ctypes.byref(d), P(q), P(k), P(v), P(g2g), P(g2l[0]), P(out), P(lse_g), st))
        torch.cuda.synchronize()
        res[which] = (out.float().cpu(), lse.cpu(), lse_g.cpu())
    (of, lf, lgf), (ot, lt, lgt) = res["full"], res["two"]
    assert float((of[:, G:] - ot[:, G:]).abs().max()) <= 3.2e-2 and float((lf - lt).abs().max()) <= 2e-2
    eg, el = float((of[:, :G] - ot[:, :G]).abs().max()), float((lgf - lgt).abs().max())
    report(f"     cw fwd_full vs fwd + glo_fwd {shape}: global row max|d| {eg:.2e}, lse_g max|d| {el:.2e}")
    assert eg < 2.5e-2 and el < 1e-2


def test_cw_launch_shapes_agree(dev):
    """The tuning hook (streams per column, chunks per workgroup, query tiles per wave, heads per workgroup) changes the
    decomposition, never the result beyond the order-free parts: bit-identical outputs for every shape (each query row's
    keys are walked in the same order with the same arithmetic)."""
    from vision_longformer_amd import _lib
    c = case(3, 32, 7, 28, 28, 1, B=11)
    q, kv, table, g2l, dout = make_inputs(c, torch.bfloat16)
    L = _lib.lib()
    base = None
    try:
        for streams, code in ((0, 0), (1, 2), (2, 1), (9, 2), (0, 12), (3, 11), (0, 321), (1, 122)):
            _lib.check(L.vil_attn_cw_set_shape(streams, code))
            out, lse = _fwd(c, q, kv, table, g2l, torch.bfloat16, "mfma_cw", dev)
            if base is None:
                base = (out, lse)
            else:
                assert torch.equal(out, base[0]) and torch.equal(lse, base[1]), (streams, code)
    finally:
        L.vil_attn_cw_set_shape(0, 0)


@pytest.mark.parametrize("c,family", [
    (case(3, 32, 7, 21, 21, 1, B=2), "mfma_cw"), (case(2, 64, 8, 16, 16, 1), "mfma_cw"), (case(2, 32, 7, 14, 14, 1, exact=1), "mfma_cw"),
    (case(2, 64, 12, 24, 25, 1, B=1), "mfma_wave"), (case(2, 32, 6, 13, 12, 1, mode=3), "mfma_wave"),
    (case(2, 32, 4, 8, 8, 1, only_glo=True), "mfma_wave"), (case(2, 32, 7, 14, 14, 1, mode=-1), "mfma_wave"),
], ids=lambda v: v if isinstance(v, str) else cid(v))
def test_default_backend_takes_the_family_the_dispatch_rule_names(c, family, dev):
    """desc.backend = MFMA (what the module uses): the chunk-workgroup forward for 3x3 neighbourhoods with W <= 8, the
    wave-per-chunk kernels for everything else -- observed through the bits of the result, which differ
    between the families (other rounding points) and are reproducible within one."""
    q, kv, table, g2l, dout = make_inputs(c, torch.bfloat16)
    out, lse = _fwd(c, q, kv, table, g2l, torch.bfloat16, "mfma", dev)
    out_f, lse_f = _fwd(c, q, kv, table, g2l, torch.bfloat16, family, dev)
    assert torch.equal(out, out_f) and torch.equal(lse, lse_f)
    other = "mfma_wave" if family == "mfma_cw" else "mfma_cw"
    out_o, lse_o = _fwd(c, q, kv, table, g2l, torch.bfloat16, other, dev)
    assert not (torch.equal(out, out_o) and torch.equal(lse, lse_o))
