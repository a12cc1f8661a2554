"""GPU micro-benchmark of vil_linear_wgrad on the layer shapes of ViL-Small / Medium-Deep: time per call (weight + bias
gradient and the reduce pass) and the error against fp64 on a column sample, with the plan vil_linear_wgrad_tune selects;
--sweep pins every plan in turn (vil_linear_wgrad_set_plan: the 128 x 128 kernel, tile 32 mi x 32 nj with m slices per
XCD) and times it per shape (kernel and reduce pass separately, from the library's hipEvent sink) next to the
plan vil_linear_wgrad_tune selects."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctypes
from vision_longformer_amd import linear
from vision_longformer_amd.linear import _wgrad

dev = torch.device("cuda:0")


def bench(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


T3, T4, T2, T1 = 128 * 197, 128 * 50, 128 * 785, 128 * 3137
shapes = {"s1 q": (T1, 96, 96), "s1 kv": (T1, 96, 192), "s1 fc1": (T1, 96, 384), "s1 fc2": (T1, 384, 96),
          "s2 q": (T2, 192, 192), "s2 kv": (T2, 192, 384), "s2 fc1": (T2, 192, 768), "s2 fc2": (T2, 768, 192),
          "s3 qkv": (T3, 384, 1152), "s3 proj": (T3, 384, 384), "s3 fc1": (T3, 384, 1536), "s3 fc2": (T3, 1536, 384),
          "s4 qkv": (T4, 768, 2304), "s4 proj": (T4, 768, 768), "s4 fc1": (T4, 768, 3072), "s4 fc2": (T4, 3072, 768),
          "ragged": (4001, 192, 96), "tiny": (77, 96, 96)}
from vision_longformer_amd import _lib
only = [a for a in sys.argv[1:] if a != "--sweep"]
sweep = "--sweep" in sys.argv


def split_times(f, n=10):
    """(main kernel us, reduce us) per call from the library's hipEvent sink"""
    _lib.profile_begin()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    rec = _lib.profile_end()
    main = sum(r[1] for r in rec if r[0] == "k_wgrad") / n * 1e3
    red = sum(r[1] for r in rec if r[0] == "k_wgrad_reduce") / n * 1e3
    return main, red


if sweep:
    for name, (T, ci, co) in shapes.items():
        if only and not any(name.startswith(o) for o in only):
            continue
        g = torch.Generator().manual_seed(5)
        x = torch.randn(T, ci, generator=g).bfloat16().to(dev)
        dy = torch.randn(T, co, generator=g).bfloat16().to(dev)
        _wgrad(dy, x, True)                                     # (tunes this problem; the selection is restored below)
        L = _lib.lib()
        _lib.check(L.vil_linear_wgrad_set_plan(T, co, ci, 1, 0, 0, 0))
        ref = _wgrad(dy, x, True)
        t0 = bench(lambda: _wgrad(dy, x, True))
        k0, r0 = split_times(lambda: _wgrad(dy, x, True))
        print(f"{name:8s} T={T:7d} {ci:5d}->{co:5d}  gen1 {t0:6.1f} us (kernel {k0:5.1f} + reduce {r0:4.1f})")
        res = []
        for mi in ([6, 3] if co % 192 == 0 else [3]):
            for nj in ([6, 3] if ci % 192 == 0 else [3]):
                tiles = (co // (32 * mi)) * (ci // (32 * nj))
                m0 = max(1, 64 // tiles)
                for m in sorted({max(1, m0 // 4), max(1, m0 // 2), m0, m0 * 2}):
                    if 8 * m * co * ci * 4 > (96 << 20) or T // (8 * m * 32) < 4 or m > 64:
                        continue
                    _lib.check(L.vil_linear_wgrad_set_plan(T, co, ci, 2, mi, nj, m))
                    out = _wgrad(dy, x, True)
                    err = float((out[0].float() - ref[0].float()).abs().max() / ref[0].float().abs().max())
                    t = bench(lambda: _wgrad(dy, x, True), n=10)
                    k, r = split_times(lambda: _wgrad(dy, x, True), n=5)
                    res.append((t, m, mi, nj, k, r, err))
        res.sort()
        for t, m, mi, nj, k, r, err in res[:4]:
            print(f"           m {m:3d} tile {32*mi:3d}x{32*nj:3d}  {t:6.1f} us (kernel {k:5.1f} + reduce {r:4.1f})  diff {err:.1e}")
        _lib.check(L.vil_linear_wgrad_tune(ctypes.c_void_p(dy.data_ptr()), ctypes.c_void_p(x.data_ptr()), T, co, ci, co, ci,
                                           ctypes.c_void_p(ref[0].data_ptr()), ctypes.c_void_p(ref[1].data_ptr()), 1,
                                           ctypes.c_void_p(linear._WG_WS[dy.device].data_ptr()),
                                           ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        t = bench(lambda: _wgrad(dy, x, True), n=10)
        print(f"           tuned plan                {t:6.1f} us")
    sys.exit(0)

tot = 0.0
for name, (T, ci, co) in shapes.items():
    if only and not any(name.startswith(o) for o in only):
        continue
    g = torch.Generator().manual_seed(5)
    x = torch.randn(T, ci, generator=g).bfloat16().to(dev)
    dy = torch.randn(T, co, generator=g).bfloat16().to(dev)
    dw, db = _wgrad(dy, x, True)
    cols = torch.arange(0, co, max(1, co // 24))
    want = dy[:, cols].double().t() @ x.double()
    err = float((dw[cols].double() - want).abs().max() / want.abs().max())
    errb = float((db.double() - dy.double().sum(0)).abs().max() / dy.double().sum(0).abs().max())
    t = bench(lambda: _wgrad(dy, x, True))
    tot += t if name[0] == "s" else 0
    fl = 2.0 * T * ci * co / t / 1e6
    by = 2.0 * T * (ci + co) / t / 1e6
    print(f"{name:8s} T={T:7d} {ci:5d}->{co:5d}  {t:7.1f} us  {fl:7.1f} TFLOP/s  {by:6.2f} TB/s   rel.err dW {err:.1e} db {errb:.1e}"
          f"  {'OK' if err < 8e-3 and errb < 8e-3 else 'MISMATCH'}")
print(f"sum over the model shapes: {tot:.1f} us")
