"""Case lists and deterministic input builders shared by tools/gen_golden.py
(which runs the real reference on them, in the build container only) and the
tests (which compare the oracle / the HIP path with the frozen outputs).

Inputs are regenerated from seeds with torch's CPU generator (deterministic
for a given torch build; the GPU box runs the same image), so the fixtures
only hold the reference's *outputs*.  Seed 300 is the reference tests' seed
(src/tests/test_slidingchunk_2d.py:56-60)."""
import torch

SEED = 300

# (mx, my, padx, pady, W)
MASK_GRIDS = [
    (2, 2, 0, 0, 4), (3, 3, 2, 2, 4), (3, 3, 3, 1, 4), (4, 4, 0, 0, 3),
    (8, 8, 0, 0, 7), (2, 2, 0, 0, 7), (3, 3, 4, 4, 8), (1, 1, 0, 0, 4),
    (1, 3, 2, 0, 3), (14, 14, 2, 2, 7), (7, 7, 1, 1, 7), (4, 4, 0, 0, 12),
    (2, 3, 1, 5, 6),
]
MODES = [0, -1, 1, 2, 3, 4, 5, 6, 7, 8]

# operator level: (name, BH, M, mx, my, W)
OP_CASES = [
    ("g2x2w4", 2, 4, 2, 2, 4),
    ("g3x3w4", 2, 4, 3, 3, 4),
    ("g4x4w3", 3, 5, 4, 4, 3),
    ("g1x3w3", 2, 4, 1, 3, 3),
    ("g3x2w2", 2, 3, 3, 2, 2),
]

# module level: dict(name, dim, H, W, nx, ny, G, rpe, sharew, only_glo, exact, mode, B)
def _mc(name, dim, H, W, nx, ny, G, rpe=True, sharew=True, only_glo=False, exact=0, mode=0, B=2):
    return dict(name=name, dim=dim, H=H, W=W, nx=nx, ny=ny, G=G, rpe=rpe, sharew=sharew,
                only_glo=only_glo, exact=exact, mode=mode, B=B)


MODULE_CASES = [
    _mc("d32h2w4_8x8_g1", 32, 2, 4, 8, 8, 1),
    _mc("d32h2w4_8x8_g1_norpe", 32, 2, 4, 8, 8, 1, rpe=False),
    _mc("d32h2w4_8x8_g1_nosharew", 32, 2, 4, 8, 8, 1, sharew=False),
    _mc("d32h2w4_10x9_g1", 32, 2, 4, 10, 9, 1),
    _mc("d32h2w4_10x9_g1_exact1", 32, 2, 4, 10, 9, 1, exact=1),
    _mc("d32h2w4_10x9_g1_cyclic", 32, 2, 4, 10, 9, 1, exact=-1),
    _mc("d48h3w3_7x7_g2", 48, 3, 3, 7, 7, 2),
    _mc("d48h3w3_7x7_g2_mode2", 48, 3, 3, 7, 7, 2, mode=2),
    _mc("d48h3w3_7x7_g2_mode7", 48, 3, 3, 7, 7, 2, mode=7),
    _mc("d48h3w3_7x7_g2_self", 48, 3, 3, 7, 7, 2, mode=-1),
    _mc("d32h2w4_10x10_g0", 32, 2, 4, 10, 10, 0),
    _mc("d32h2w4_8x8_g1_onlyglo", 32, 2, 4, 8, 8, 1, only_glo=True),
    _mc("d32h2w4_5x6_g1_mode5_cyclic", 32, 2, 4, 5, 6, 1, exact=-1, mode=5),
    _mc("d64h2w7_14x14_g1", 64, 2, 7, 14, 14, 1),
    _mc("d64h2w7_16x15_g1_mode3", 64, 2, 7, 16, 15, 1, mode=3),
    _mc("d128h2w8_20x20_g1", 128, 2, 8, 20, 20, 1, B=1),
    # BASELINE shapes (summarised in the fixture: strided sample + sums)
    _mc("small_s1_d96h3w7_56x56_g1", 96, 3, 7, 56, 56, 1),
    _mc("tiny_s1_d48h1w7_56x56_g1", 48, 1, 7, 56, 56, 1),
    _mc("small_s2_d192h3w7_28x28_g1", 192, 3, 7, 28, 28, 1),
    _mc("meddeep_s2_d192h3w7_48x48_g1", 192, 3, 7, 48, 48, 1, B=1),
    # the other windows / grids of BASELINE configs 4-5 (round 3): W = 6 with a random-shift neighbour on the 96x96 grid
    # (16x16 chunks), W = 8 random shift and W = 12 on 48x48, and the 96x96 grid zero-padded to 98x98 under W = 7
    _mc("basedeep_s1_d96h3w6_96x96_g1_mode4", 96, 3, 6, 96, 96, 1, mode=4, B=1),
    _mc("basedeep_s2_d192h3w8_48x48_g1_mode6", 192, 3, 8, 48, 48, 1, mode=6, B=1),
    _mc("meddeep_s2_d192h3w12_48x48_g1", 192, 3, 12, 48, 48, 1, B=1),
    _mc("meddeep_s1_d96h3w7_96x96_g1_pad98", 96, 3, 7, 96, 96, 1, B=1),
]
BIG_CASES = {"small_s1_d96h3w7_56x56_g1", "tiny_s1_d48h1w7_56x56_g1",
             "small_s2_d192h3w7_28x28_g1", "meddeep_s2_d192h3w7_48x48_g1",
             "basedeep_s1_d96h3w6_96x96_g1_mode4", "basedeep_s2_d192h3w8_48x48_g1_mode6",
             "meddeep_s2_d192h3w12_48x48_g1", "meddeep_s1_d96h3w7_96x96_g1_pad98"}   # too slow for O(N^2) checks
SAMPLE_ABOVE = 4096   # tensors with more elements are stored as strided sample + sums


# dense `Attention` module (the s0 stages, reference msvit.py:37-120): dict(name, dim, H, nx, G, B)
DENSE_CASES = [
    dict(name="dense_d384h6_14x14_g1", dim=384, H=6, nx=14, G=1, B=2),       # ViL-Small stage 3
    dict(name="dense_d384h6_24x24_g1", dim=384, H=6, nx=24, G=1, B=1),       # ViL-Medium-Deep@384 stage 3
    dict(name="dense_d128h2_14x14_g0", dim=128, H=2, nx=14, G=0, B=2),
    dict(name="dense_d256h4_7x7_g0", dim=256, H=4, nx=7, G=0, B=2),
    dict(name="dense_d32h2_5x5_g2", dim=32, H=2, nx=5, G=2, B=2),
    dict(name="dense_d96h2_9x9_g1", dim=96, H=2, nx=9, G=1, B=2),            # head_dim 48
]


def dense_inputs(c, dtype=torch.float64):
    """(params, x, dout) of a dense Attention case; weights ~ N(0, 1/sqrt(dim)), bias tables ~ N(0, 0.5)"""
    g = torch.Generator().manual_seed(SEED + 1)
    dim, H, nx, G = c["dim"], c["H"], c["nx"], c["G"]
    shapes = [("qkv.weight", (3 * dim, dim)), ("qkv.bias", (3 * dim,)), ("proj.weight", (dim, dim)), ("proj.bias", (dim,)),
              ("local_relative_position_bias_table", ((2 * nx - 1) ** 2, H))]
    if G >= 1:
        shapes += [("g2l_relative_position_bias", (2, H, G)), ("g2g_relative_position_bias", (H, G, G))]
    params = {}
    for name, shape in shapes:
        t = torch.randn(*shape, generator=g, dtype=torch.float64)
        t = t * (0.5 if "relative_position" in name else (dim ** -0.5 if name.endswith(".weight") else 0.1))
        params[name] = t.to(dtype)
    N = G + nx * nx
    x = torch.randn(c["B"], N, dim, generator=g, dtype=torch.float64).to(dtype)
    dout = torch.randn(c["B"], N, dim, generator=g, dtype=torch.float64).to(dtype)
    return params, x, dout


def op_inputs(case, dtype=torch.float64):
    name, BH, M, mx, my, W = case
    g = torch.Generator().manual_seed(SEED)
    W2 = W * W
    q = torch.randn(BH, M, mx, my, W2, generator=g, dtype=dtype)
    k = torch.randn(BH, M, mx, my, W2, generator=g, dtype=dtype)
    v = torch.randn(BH, M, mx, my, W2, generator=g, dtype=dtype)
    return q, k, v


def module_param_shapes(c):
    dim, H, W, G = c["dim"], c["H"], c["W"], c["G"]
    shapes = [("query.weight", (dim, dim)), ("query.bias", (dim,)),
              ("kv.weight", (2 * dim, dim)), ("kv.bias", (2 * dim,)),
              ("proj.weight", (dim, dim)), ("proj.bias", (dim,))]
    if G >= 1 and not c["sharew"]:
        shapes += [("query_global.weight", (dim, dim)), ("query_global.bias", (dim,)),
                   ("kv_global.weight", (2 * dim, dim)), ("kv_global.bias", (2 * dim,)),
                   ("proj_global.weight", (dim, dim)), ("proj_global.bias", (dim,))]
    if c["rpe"]:
        shapes += [("local_relative_position_bias_table", ((4 * W - 1) ** 2, H))]
        if G >= 1:
            shapes += [("g2l_relative_position_bias", (2, H, G)),
                       ("g2g_relative_position_bias", (H, G, G))]
    return shapes


def module_inputs(c, dtype=torch.float64):
    """Returns (params, x, dout).  Weights ~ N(0, 1/sqrt(dim)) so scores are O(1);
    the bias tables ~ N(0, 0.5) so that bias errors are visible."""
    g = torch.Generator().manual_seed(SEED)
    dim = c["dim"]
    params = {}
    for name, shape in module_param_shapes(c):
        t = torch.randn(*shape, generator=g, dtype=torch.float64)
        if "relative_position" in name:
            t = t * 0.5
        elif name.endswith(".weight"):
            t = t * dim ** -0.5
        else:
            t = t * 0.1
        params[name] = t.to(dtype)
    if c["G"] >= 1 and c["sharew"]:
        for nm in ("query", "kv", "proj"):
            params[nm + "_global.weight"] = params[nm + ".weight"]
            params[nm + "_global.bias"] = params[nm + ".bias"]
    N = c["G"] + c["nx"] * c["ny"]
    x = torch.randn(c["B"], N, dim, generator=g, dtype=torch.float64).to(dtype)
    dout = torch.randn(c["B"], N, dim, generator=g, dtype=torch.float64).to(dtype)
    return params, x, dout


def sample_big(t):
    """Deterministic strided sample + sums used to summarise big outputs."""
    flat = t.reshape(-1).to(torch.float64)
    step = max(1, flat.numel() // 4096)
    return flat[::step].clone(), torch.stack([flat.sum(), flat.abs().sum(), (flat * flat).sum()])


def model_grad_sample(g, n=512):
    """Deterministic strided sample (<= n elements) of a model-level gradient tensor."""
    flat = g.detach().reshape(-1).to(torch.float64).cpu()
    step = max(1, flat.numel() // n)
    return flat[::step][:n].clone()
