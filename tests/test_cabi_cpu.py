"""CPU: the C-ABI library loads, exports every symbol include/vil_attn.h declares,
validates arguments like the reference does, and its host-side geometry helpers
(the SAME inline functions the kernels use for masks / bias indices) reproduce
the golden masks and the reference's relative_position_index bit for bit.
No compute call is made here (no GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import golden_cases as GC
from vision_longformer_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    hdr = open(os.path.join(ROOT, "include", "vil_attn.h")).read()
    declared = set(re.findall(r"\b(vil_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name
    assert L.vil_attn_abi_version() == _lib.ABI_VERSION == 2
    assert L.vil_attn_strerror(0) == b"ok"
    assert b"exact" in L.vil_attn_strerror(-6)


def test_desc_struct_matches_header_size():
    # 12 int32 + float + int32 + 24 int64 + 1 pointer (mode_dev)
    assert ctypes.sizeof(_lib.VilAttnDesc) == 14 * 4 + 24 * 8 + 8


def _desc(**kw):
    d = _lib.VilAttnDesc()
    base = dict(B=2, H=2, M=16, nx=8, ny=8, W=4, G=1, mode=0, exact=0, dtype=_lib.DTYPE_F32, only_glo=0,
                backend=_lib.BACKEND_AUTO, scale=0.25)
    base.update(kw)
    for k, v in base.items():
        setattr(d, k, v)
    return d


def test_argument_validation():
    L = _lib.lib()
    assert L.vil_attn_check(ctypes.byref(_desc())) == 0
    assert L.vil_attn_check(ctypes.byref(_desc(mode=9))) == -5
    assert L.vil_attn_check(ctypes.byref(_desc(exact=2))) == -6
    # exact=1 with mode!=0: the reference raises ValueError (slidingchunk_2d.py:331-343)
    assert L.vil_attn_check(ctypes.byref(_desc(exact=1, mode=3))) == -6
    assert L.vil_attn_check(ctypes.byref(_desc(M=7, backend=_lib.BACKEND_SCALAR))) == -3
    assert L.vil_attn_check(ctypes.byref(_desc(dtype=5))) == -7
    assert L.vil_attn_check(ctypes.byref(_desc(B=0))) == -2
    # NULL pointers are rejected before any launch
    d = _desc()
    assert L.vil_attn_fwd(ctypes.byref(d), None, None, None, None, None, None, None, None, None) == -1
    assert L.vil_attn_workspace_bytes(ctypes.byref(d), 1) > 0


def test_glue_entry_points_validate_before_launching():
    """The glue kernels' entry points (LayerNorm, column sum, weight gradient, library GEMM) reject NULL / bad
    shapes / misalignment with VIL_E_* codes before touching the device (so this runs without a GPU)."""
    L = _lib.lib()
    vp = ctypes.c_void_p
    a16 = vp(4096)                       # a 16-byte aligned fake address: never dereferenced on these paths
    assert L.vil_colsum_bf16(None, 8, 8, 8, a16, 1, a16, None) == -1
    assert L.vil_colsum_bf16(a16, 8, 12, 16, a16, 1, a16, None) == -8          # C % 8: VIL_E_ALIGN
    assert L.vil_colsum_f32(a16, 0, 8, 8, a16, 0, a16, None) == -2
    assert L.vil_colsum_workspace_bytes(384) == 512 * 384 * 4
    assert L.vil_linear_wgrad(None, a16, 64, 8, 8, 8, 8, a16, None, 1, a16, None) == -1
    assert L.vil_linear_wgrad(a16, a16, 64, 8, 12, 8, 16, a16, None, 1, a16, None) == -8
    assert L.vil_linear_wgrad(a16, vp(4100), 64, 8, 8, 8, 8, a16, None, 1, a16, None) == -8
    assert L.vil_linear_wgrad_workspace_bytes(25216, 1536, 384) > 1536 * 384 * 4
    assert L.vil_linear_wgrad_tune(None, a16, 64, 8, 8, 8, 8, a16, None, 1, a16, None) == -1
    assert L.vil_linear_wgrad_set_plan(4096, 192, 96, 2, 6, 3, 4) == 0 and L.vil_linear_wgrad_set_plan(4096, 192, 96, 0, 0, 0, 0) == 0
    assert L.vil_linear_wgrad_set_plan(4096, 192, 96, 2, 6, 6, 4) == L.vil_linear_wgrad_set_plan(4096, 192, 96, 3, 0, 0, 0) < 0
    assert L.vil_linear_wgrad_tune(a16, a16, 64, 8, 12, 8, 16, a16, None, 1, a16, None) == -8
    assert L.vil_gemm_bf16(0, None, a16, None, a16, 8, 8, 8, 8, 8, a16, 1 << 20, None) == -1
    assert L.vil_gemm_bf16(3, a16, a16, None, a16, 8, 8, 8, 8, 8, a16, 1 << 20, None) == -2
    assert L.vil_gemm_bf16(1, a16, a16, a16, a16, 8, 8, 8, 8, 8, a16, 1 << 20, None) == -2   # no bias on the input gradient
    assert L.vil_gemm_bf16(0, a16, a16, None, a16, 8, 12, 8, 16, 8, a16, 1 << 20, None) == -8
    assert L.vil_gemm_workspace_bytes() >= 1 << 20
    assert L.vil_resln_fwd(None, a16, 1, None, 1, a16, a16, a16, a16, 1, a16, a16, 8, 8, 1e-6, None) == -1
    assert L.vil_resln_fwd(a16, a16, 1, None, 0, a16, a16, a16, a16, 1, a16, a16, 8, 8, 1e-6, None) == -2
    assert L.vil_resln_bwd(a16, 1, None, a16, a16, a16, a16, None, 1, None, None, 1, a16, a16, a16, 8, 8, None) == -1   # dx (gbranch may be NULL)
    d = _desc(mode=3, dtype=_lib.DTYPE_BF16)
    d.mode_dev = 4096
    assert L.vil_attn_check(ctypes.byref(d)) in (0, -10)        # accepted (MFMA) or backend-declined, never a crash
    d = _desc(mode=0, dtype=_lib.DTYPE_BF16)
    d.mode_dev = 4096
    assert L.vil_attn_check(ctypes.byref(d)) == -5              # a device-side neighbour needs a random-shift mode


@pytest.mark.parametrize("grid", GC.MASK_GRIDS, ids=lambda g: "g%dx%dp%dx%dw%d" % g)
def test_geometry_masks_bit_exact(golden_dir, grid):
    L = _lib.lib()
    gold = np.load(os.path.join(golden_dir, "masks.npz"))
    mx, my, padx, pady, W = grid
    nx, ny = mx * W - padx, my * W - pady
    W2 = W * W
    for exact in (0, -1, 1):
        for mode in GC.MODES:
            if exact == 1 and mode != 0:
                assert L.vil_geom_mask(nx, ny, W, exact, mode, ctypes.c_void_p(1)) == -6
                continue
            kv = {0: 9 * W2, -1: W2}.get(mode, 2 * W2)
            buf = np.zeros(mx * my * W2 * kv, dtype=np.uint8)
            assert L.vil_geom_mask(nx, ny, W, exact, mode, buf.ctypes.data_as(ctypes.c_void_p)) == kv
            key = f"g{mx}x{my}p{padx}x{pady}w{W}e{exact}m{mode}"
            ref = np.unpackbits(gold[key])[:buf.size]
            assert np.array_equal(ref, buf), key


@pytest.mark.parametrize("W", [2, 3, 4, 6, 7, 8, 12])
def test_geometry_bias_index(golden_dir, W):
    L = _lib.lib()
    gold = np.load(os.path.join(golden_dir, "rel_index.npz"))[f"W{W}"]
    W2 = W * W
    for mode in GC.MODES:
        kv = {0: 9 * W2, -1: W2}.get(mode, 2 * W2)
        buf = np.zeros(W2 * kv, dtype=np.int32)
        assert L.vil_geom_bias_index(W, mode, buf.ctypes.data_as(ctypes.c_void_p)) == kv
        if mode == 0:
            cols = np.arange(9 * W2)
        elif mode == -1:
            cols = np.arange(4 * W2, 5 * W2)
        else:
            cid = mode if mode > 4 else mode - 1
            cols = np.concatenate([np.arange(4 * W2, 5 * W2), np.arange(cid * W2, (cid + 1) * W2)])
        assert np.array_equal(buf.reshape(W2, kv), gold[:, cols]), (W, mode)


def test_module_state_dict_contract(golden_dir):
    """Parameter / buffer names and shapes of the drop-in module."""
    from vision_longformer_amd.longformer2d import Long2DSCSelfAttention, build_relative_position_index
    m = Long2DSCSelfAttention(96, num_heads=3, qkv_bias=True, w=7, sharew=True, nglo=1, rpe=True)
    sd = m.state_dict()
    expect = {
        "query.weight": (96, 96), "query.bias": (96,), "kv.weight": (192, 96), "kv.bias": (192,),
        "proj.weight": (96, 96), "proj.bias": (96,),
        "query_global.weight": (96, 96), "query_global.bias": (96,), "kv_global.weight": (192, 96),
        "kv_global.bias": (192,), "proj_global.weight": (96, 96), "proj_global.bias": (96,),
        "local_relative_position_bias_table": (729, 3), "g2l_relative_position_bias": (2, 3, 1),
        "g2g_relative_position_bias": (3, 1, 1), "relative_position_index": (49, 441),
    }
    assert {k: tuple(v.shape) for k, v in sd.items()} == expect
    assert sd["relative_position_index"].dtype == torch.int64
    # sharew aliases are de-duplicated in parameters()
    assert len(list(m.parameters())) == 9
    m2 = Long2DSCSelfAttention(32, num_heads=2, qkv_bias=True, w=4, sharew=False, nglo=1, rpe=True)
    assert len(list(m2.parameters())) == 15
    gold = np.load(os.path.join(golden_dir, "rel_index.npz"))
    for W in (2, 3, 4, 6, 7, 8, 12):
        assert np.array_equal(build_relative_position_index(W).numpy(), gold[f"W{W}"].astype(np.int64))
    for attr in ("mode", "Nglo", "num_heads", "head_dim", "attention_window", "only_glo", "exact", "rpe", "scale"):
        assert hasattr(m, attr)


def test_product_path_has_no_cpu_fallback():
    from vision_longformer_amd.longformer2d import Long2DSCSelfAttention
    m = Long2DSCSelfAttention(32, num_heads=2, qkv_bias=True, w=4, nglo=1, rpe=True)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.randn(1, 65, 32), 8, 8)
    with pytest.raises(AssertionError):
        m(torch.randn(1, 64, 32), 8, 8)   # "Global dimension does not match!"


def test_bench_refuses_to_run_without_a_gpu():
    """bench.py measures the HIP path only: on a box without a GPU it must stop with a clear message instead of
    timing a CPU fallback (and `--help` must work anywhere)."""
    import subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "--gpus" in r.stdout and "--steps" in r.stdout and "--warmup" in r.stdout
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert r.returncode != 0 and "needs a GPU" in (r.stderr + r.stdout)


def test_product_package_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under the product package may import it."""
    pkg = os.path.join(ROOT, "vision-longformer_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), fn
    for fn in ("__init__.py",):
        src = open(os.path.join(ROOT, "vision_longformer_amd", fn)).read()
        assert "oracle" not in src


def test_round2_entry_points_validate_before_launching():
    """Operator-level entry points (vil_sc2d_*), vil_gemm_tune and the fp16 dtype: argument errors come back as
    VIL_E_* codes without touching a device."""
    L = _lib.lib()
    vp = ctypes.c_void_p
    a16 = vp(4096)
    F32, F64, BF16, F16 = _lib.DTYPE_F32, _lib.DTYPE_F64, _lib.DTYPE_BF16, _lib.DTYPE_F16
    assert L.vil_sc2d_qk(None, a16, a16, 2, 4, 2, 2, 4, 0, F32, None) == -1
    assert L.vil_sc2d_qk(a16, a16, a16, 2, 4, 2, 2, 4, 0, 7, None) == -7           # unknown dtype
    assert L.vil_sc2d_av(a16, a16, a16, 2, 4, 2, 2, 4, 9, F64, None) == -5         # mode out of range
    assert L.vil_sc2d_agrad(a16, a16, a16, 0, 4, 2, 2, 4, 0, F32, None) == -2
    assert L.vil_sc2d_mask(a16, 2, 2, 2, 0, 0, 4, 1, 3, F32, None, None) == -6     # exact=1 with mode != 0: the reference's ValueError
    assert L.vil_sc2d_mask(a16, 2, 2, 2, 0, 0, 4, 2, 0, F32, None, None) == -6
    assert L.vil_sc2d_mask(a16, 2, 2, 2, 4, 0, 4, 0, 0, F32, None, None) == -2     # padding must be < W
    assert L.vil_gemm_tune(0, None, a16, None, a16, 8, 8, 8, 8, 8, a16, 1 << 20, None) == -1
    assert L.vil_gemm_tune(0, a16, a16, None, a16, 8, 12, 8, 16, 8, a16, 1 << 20, None) == -8
    # fp16 I/O is a supported dtype of both kernel families (the reference's AMP dtype)
    assert L.vil_attn_check(ctypes.byref(_desc(dtype=F16))) == 0
    assert L.vil_attn_check(ctypes.byref(_desc(dtype=F16, backend=_lib.BACKEND_MFMA, M=32))) == 0
    assert L.vil_attn_check(ctypes.byref(_desc(dtype=F64))) == -7                  # fp64: operator-level entry points only
    # cyclic padding and only_glo no longer fall back to the scalar family
    assert L.vil_attn_check(ctypes.byref(_desc(dtype=BF16, backend=_lib.BACKEND_MFMA, M=32, exact=-1))) == 0
    assert L.vil_attn_check(ctypes.byref(_desc(dtype=BF16, backend=_lib.BACKEND_MFMA, M=32, only_glo=1))) == 0


def test_token_layernorm_entry_points_validate_before_launching():
    """vil_layernorm_fwd_tokens / _bwd_tokens (LayerNorm written behind the global-token rows of the token tensor):
    argument errors without touching a device."""
    L = _lib.lib()
    vp = ctypes.c_void_p
    a16 = vp(4096)
    F32, BF16 = _lib.DTYPE_F32, _lib.DTYPE_BF16
    f = L.vil_layernorm_fwd_tokens
    assert f(None, BF16, a16, a16, a16, F32, a16, a16, 64, 96, 96, 1e-5, 16, 1, None) == -1
    assert f(a16, BF16, a16, a16, a16, F32, a16, a16, 64, 100, 100, 1e-5, 16, 1, None) == -3       # C % 8 (VIL_E_HEAD_DIM)
    assert f(a16, BF16, a16, a16, a16, F32, a16, a16, 64, 96, 96, 1e-5, 0, 1, None) == -2           # rows per sample
    assert f(a16, BF16, a16, a16, a16, F32, a16, a16, 64, 96, 96, 1e-5, 24, 1, None) == -2          # 64 % 24 != 0
    assert f(a16, BF16, a16, a16, a16, F32, a16, a16, 64, 96, 96, 1e-5, 16, -1, None) == -2
    b = L.vil_layernorm_bwd_tokens
    assert b(a16, F32, a16, BF16, a16, a16, a16, a16, BF16, a16, a16, None, 64, 96, 96, 96, 16, 1, None) == -1
    assert b(a16, F32, a16, BF16, a16, a16, a16, a16, F32, a16, a16, a16, 64, 96, 96, 96, 16, 1, None) == -7   # dx dtype = x dtype
    assert b(a16, F32, a16, BF16, a16, a16, a16, a16, BF16, a16, a16, a16, 60, 96, 96, 96, 16, 1, None) == -2


def test_patchify_entry_points_validate_before_launching():
    """vil_patchify_fwd / _bwd (stage transition as one row gather): argument errors without touching a device."""
    L = _lib.lib()
    vp = ctypes.c_void_p
    a16 = vp(4096)
    F32, BF16, F16 = _lib.DTYPE_F32, _lib.DTYPE_BF16, _lib.DTYPE_F16
    f, b = L.vil_patchify_fwd, L.vil_patchify_bwd
    assert f(None, a16, BF16, a16, a16, BF16, 2, 1, 8, 8, 96, 2, 2, None) == -1
    assert f(a16, a16, F16, a16, a16, BF16, 2, 1, 8, 8, 96, 2, 2, None) == -7           # branch dtype
    assert f(a16, None, 0, None, a16, F16, 2, 1, 8, 8, 96, 2, 2, None) == -7             # output dtype
    assert f(a16, None, 0, None, a16, BF16, 2, 1, 9, 8, 96, 2, 2, None) == -2            # nx % ph
    assert f(a16, None, 0, None, a16, BF16, 2, 1, 8, 8, 100, 2, 2, None) == -8           # C % 8
    assert f(a16, None, 0, None, vp(4100), BF16, 2, 1, 8, 8, 96, 2, 2, None) == -8       # alignment
    assert b(a16, BF16, a16, None, a16, BF16, 2, 1, 8, 8, 96, 2, 2, None) == -1
    assert b(a16, BF16, a16, a16, a16, F16, 2, 1, 8, 8, 96, 2, 2, None) == -7
    assert b(a16, BF16, a16, a16, None, 0, 0, 1, 8, 8, 96, 2, 2, None) == -2


def test_optimizer_entry_points_plan_and_validate_on_the_host():
    """vil_optim_* (the reference's AdamW / QHM as one multi-tensor launch): the descriptor struct matches the header,
    the plan builder lays out header + descriptors + 4096-element blocks, and argument errors come back as VIL_E_*
    codes without touching a device."""
    L = _lib.lib()
    vp = ctypes.c_void_p
    assert ctypes.sizeof(_lib.VilOptimTensor) == 5 * 8 + 8 + 4 * 4 + 8
    T = _lib.VilOptimTensor
    arr = (T * 3)()
    for i, n in enumerate((10000, 37, 4096)):
        arr[i].param, arr[i].grad, arr[i].state1, arr[i].state2 = 4096, 8192, 12288, 16384
        arr[i].n, arr[i].grad_dtype, arr[i].low_dtype, arr[i].weight_decay = n, _lib.DTYPE_BF16, _lib.DTYPE_BF16, 0.05
    nbytes = L.vil_optim_plan_bytes(arr, 3)
    assert nbytes == 16 + 3 * ctypes.sizeof(T) + (3 + 1 + 1) * 16
    buf = (ctypes.c_char * nbytes)()
    nb = ctypes.c_int(0)
    assert L.vil_optim_plan_build(arr, 3, buf, nbytes, ctypes.byref(nb)) == 0 and nb.value == 5
    hdr = np.frombuffer(buf, dtype=np.int32, count=4)
    assert hdr[0] == 3 and hdr[1] == 5
    blocks = np.frombuffer(buf, dtype=np.int64, offset=16 + 3 * ctypes.sizeof(T)).reshape(5, 2)
    assert [int(b[0] & 0xffffffff) for b in blocks] == [0, 0, 0, 1, 2] and [int(b[1]) for b in blocks] == [0, 4096, 8192, 0, 0]
    assert L.vil_optim_plan_build(arr, 3, buf, nbytes - 1, ctypes.byref(nb)) == -9
    arr[1].grad_dtype = _lib.DTYPE_F64
    assert L.vil_optim_plan_build(arr, 3, buf, nbytes, ctypes.byref(nb)) == -7
    arr[1].grad_dtype = _lib.DTYPE_F32
    arr[2].n = 0
    assert L.vil_optim_plan_build(arr, 3, buf, nbytes, ctypes.byref(nb)) == -2
    a16 = vp(4096)
    assert L.vil_optim_adamw_step(None, 5, 0.9, 0.999, 1e-6, 1, a16, None) == -1
    assert L.vil_optim_adamw_step(a16, 5, 1.0, 0.999, 1e-6, 1, a16, None) == -2      # beta1 in [0, 1)
    assert L.vil_optim_adamw_step(vp(4100), 5, 0.9, 0.999, 1e-6, 1, a16, None) == -8
    assert L.vil_optim_qhm_step(a16, 0, 0.9, 1.0, a16, None) == -2
    assert L.vil_optim_qhm_step(a16, 5, 1.5, 1.0, a16, None) == -2


def test_dense_family_entry_points_validate_on_the_host():
    """vil_dense_attn_* (csrc/vil_attn_dense.hip): descriptor checks, workspace size and NULL rejection run before any
    launch, so they are testable without a GPU"""
    L = _lib.lib()

    def dd(**kw):
        base = dict(B=4, H=6, M=64, nx=14, ny=14, W=14, G=1, mode=-1, dtype=_lib.DTYPE_BF16)
        base.update(kw)
        d = _desc(**base)
        C = d.H * d.M
        N = d.G + d.nx * d.ny
        for pre in ("q", "k", "v", "do", "dq", "dk", "dv"):
            setattr(d, pre + "_sb", N * 3 * C); setattr(d, pre + "_st", 3 * C); setattr(d, pre + "_sh", d.M)
        d.o_sb, d.o_st, d.o_sh = N * C, C, d.M
        return d

    ok = dd()
    assert L.vil_dense_attn_supported(ctypes.byref(ok)) == 0
    assert L.vil_dense_attn_supported(ctypes.byref(dd(nx=24, ny=24))) == 0          # any sequence length
    assert L.vil_dense_attn_supported(ctypes.byref(dd(nx=9, ny=17, G=4))) == 0       # rectangular grids, up to 4 global tokens
    assert L.vil_dense_attn_supported(ctypes.byref(dd(M=32))) == _lib.VIL_E_HEAD_DIM if hasattr(_lib, "VIL_E_HEAD_DIM") else -3
    assert L.vil_dense_attn_supported(ctypes.byref(dd(dtype=_lib.DTYPE_F32))) == -7
    assert L.vil_dense_attn_supported(ctypes.byref(dd(G=5))) == _lib.VIL_E_BACKEND
    assert L.vil_dense_attn_supported(ctypes.byref(dd(B=0))) == -2
    assert L.vil_dense_attn_workspace_bytes(ctypes.byref(ok), 0) == 0
    # backward: delta (B*H*N floats) + one histogram record per dQ workgroup
    N, TS = 197, 27 * 27
    nwg = 2                                     # 13 row units of 16 queries -> two workgroups of <= 8 waves
    want = ((4 * 6 * N + 3) // 4 * 4) * 4 + 4 * 6 * nwg * (((TS + 1) // 2 * 2) + 32) * 4
    assert L.vil_dense_attn_workspace_bytes(ctypes.byref(ok), 1) == want
    assert L.vil_dense_attn_fwd(ctypes.byref(ok), None, None, None, None, None, None, None, None, None) == -1
    assert L.vil_dense_attn_bwd(ctypes.byref(ok), *([None] * 17)) == -1


def test_hot_kernels_stay_inside_their_register_budgets():
    """hipcc's own resource report of the last build (csrc/build/*.resources.txt, written by __graft_entry__.build()): the
    sliding-chunk kernels of BASELINE's shapes hold no scratch and keep the residency their launch bounds ask for.  The
    16-bit wave-per-chunk kernels sit a handful of registers under their limits; a harmless-looking edit on an exit path
    cost the dK/dV pass 6.5 % in round 6 (256 VGPRs + 76 bytes of scratch instead of 250 + 0) and no parity test can see that."""
    import sys
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.build()
    res = g.kernel_resources()
    assert len(res) > 60, len(res)

    def pick(pattern):
        sel = {k: v for k, v in res.items() if re.search(pattern, k)}
        assert sel, pattern
        return sel
    # every 16-bit instantiation of the wave-per-chunk passes: no scratch, and the residency of their launch bounds
    for pat, occ in ((r"k_mfma_fwdIDF16.Li[12]E", 3), (r"k_mfma_fwdIDF16.Li[34]E", 2), (r"k_mfma_bwd_dqIDF16", 2),
                     (r"k_mfma_bwd_dqIDF16.Li2ELi2E", 3), (r"k_mfma_bwd_dkdvIDF16", 2)):
        for k, v in pick(pat).items():
            assert v["scratch"] == 0, (k, v)
            assert v["occupancy"] >= occ, (k, v)
    # the chunk-workgroup forward (<T, head_dim / 16, query tiles per wave, SAFE>): the fast kernels hold no scratch; head_dim
    # 32 at <= 96 registers = five waves per SIMD, head_dim 64 with two query tiles three; the exact kernels (flagged
    # columns and fp16 only) may keep their few spilled dwords
    for k, v in pick(r"k_cw_fwdIDF16.Li\dELi\dELb0EEv").items():
        assert v["scratch"] == 0, (k, v)
    for k, v in pick(r"k_cw_fwdIDF16.Li2ELi\dELb0EEv").items():
        assert v["vgprs"] <= 96 and v["occupancy"] >= 5, (k, v)
    for k, v in pick(r"k_cw_fwdIDF16.Li\dELi\dELb1EEv").items():
        assert v["scratch"] <= 16 and v["occupancy"] >= 3, (k, v)
    # dense stages (<T, head_dim / 32, ...>): four waves per SIMD at head_dim 32, three at head_dim 64
    for k, v in pick(r"k_dense_(fwd|bwd_dq|bwd_dkdv)IDF16.Li1E").items():
        assert v["occupancy"] >= 4 and v["scratch"] <= 40, (k, v)
    for k, v in pick(r"k_dense_(fwd|bwd_dq|bwd_dkdv)IDF16.Li2E").items():
        assert v["occupancy"] >= 3 and v["scratch"] == 0, (k, v)


def _cw_plan(B, H, M, nx, ny, W, G=1, mode=0):
    d = _lib.VilAttnDesc()
    C, N = H * M, G + nx * ny
    for k, v in dict(B=B, H=H, M=M, nx=nx, ny=ny, W=W, G=G, mode=mode, exact=0, dtype=_lib.DTYPE_BF16, only_glo=0,
                     backend=_lib.BACKEND_MFMA_CW, scale=M ** -0.5).items():
        setattr(d, k, v)
    for pre, st, rows in (("q", C, nx * ny), ("k", 2 * C, N), ("v", 2 * C, N), ("o", C, nx * ny)):
        setattr(d, pre + "_st", st); setattr(d, pre + "_sb", st * rows); setattr(d, pre + "_sh", M)
    out = (ctypes.c_int32 * 24)()
    L = _lib.lib()
    L.vil_attn_cw_plan.restype = ctypes.c_int
    assert L.vil_attn_cw_plan(ctypes.byref(d), out) == 0
    return list(out)


@pytest.mark.parametrize("shape", [(128, 3, 32, 56, 56, 7), (128, 3, 64, 28, 28, 7), (32, 3, 32, 96, 96, 7), (32, 3, 64, 48, 48, 7),
                                   (32, 3, 32, 96, 96, 8), (9, 2, 32, 21, 20, 7), (17, 2, 64, 14, 14, 7), (8, 6, 32, 14, 14, 7),
                                   (100, 3, 32, 30, 9, 7), (8, 1, 32, 3, 2, 4), (600, 1, 32, 14, 14, 7), (2048, 1, 64, 7, 7, 7)], ids=str)
def test_cw_launch_plan_covers_every_image_head_and_chunk_group_once(shape):
    """The chunk-workgroup forward's launch plan (vil_attn_cw_plan: host only) decoded the way the kernel decodes blockIdx
    (csrc/vil_attn_cw.hip, `by_image` branch): every (image, head group, chunk group) belongs to exactly one workgroup, no
    workgroup walks more than the 32 images its redo mask can name, and the plan's totals are consistent."""
    B, H = shape[0], shape[1]
    o = _cw_plan(*shape)
    nseg, nwgx, NS, NCH, NHG, ngrp, by_image, nch = o[:8]
    segs = [o[8 + 4 * k: 12 + 4 * k] for k in range(nseg)]
    assert by_image == 1 and 1 <= nseg <= 4 and ngrp == -(-nch // NCH) and NHG == H
    assert sorted(g for g0, ng, ns, wg0 in segs for g in range(g0, g0 + ng)) == list(range(ngrp))       # a partition of the groups
    assert sum(ng * NHG * ns for g0, ng, ns, wg0 in segs) == nwgx and max(ns for _, _, ns, _ in segs) == NS
    seen = {}
    for xcd in range(8):
        for kblk in range(nwgx):
            sc = max(k for k in range(nseg) if kblk >= segs[k][3])
            g0, ng, ns, wg0 = segs[sc]
            idx = kblk - wg0
            assert 0 <= idx < ng * NHG * ns
            strm, col = divmod(idx, ng * NHG)
            gl, h = divmod(col, NHG)
            imgs = list(range(xcd + 8 * strm, B, 8 * ns))
            assert len(imgs) <= 32
            for b in imgs:
                key = (b, h, g0 + gl)
                assert key not in seen, (key, seen[key], (xcd, kblk))
                seen[key] = (xcd, kblk)
    assert len(seen) == B * NHG * ngrp


def test_cw_launch_shape_hook_validates_its_arguments():
    """vil_attn_cw_set_shape (tools / tests only): out-of-range values are refused with VIL_E_SHAPE and leave the library's own
    choice in place; (0, 0) restores it."""
    L = _lib.lib()
    E_SHAPE = -2
    assert L.vil_attn_strerror(E_SHAPE) != b"ok"
    plan0 = _cw_plan(32, 3, 32, 28, 28, 7)
    for bad in ((-1, 0), (0, -1), (0, 5), (0, 30), (0, 900)):       # streams < 0, code < 0, 5 chunks, 3 query tiles, 9 heads
        assert L.vil_attn_cw_set_shape(*bad) == E_SHAPE, bad
        assert _cw_plan(32, 3, 32, 28, 28, 7) == plan0
    try:
        assert L.vil_attn_cw_set_shape(2, 1) == 0           # two streams, one chunk per workgroup
        o = _cw_plan(32, 3, 32, 28, 28, 7)
        assert o[2] == 2 and o[3] == 1 and o != plan0
    finally:
        assert L.vil_attn_cw_set_shape(0, 0) == 0
    assert _cw_plan(32, 3, 32, 28, 28, 7) == plan0
