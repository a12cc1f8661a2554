"""Per-segment cycle budget of one k_mfma_bwd_dkdv wave (diagnostics build -DVIL_KV_TIMING=1 of libvilattn.so,
tools/ab/build_variants.sh kvtiming "-DVIL_KV_TIMING=1"; run with VIL_ATTN_LIB pointing at it).  s_memtime stamps
(each one drains lgkmcnt and fences the scheduler, so the instrumented kernel is a few % slower than the product).

    VIL_ATTN_LIB=$PWD/tools/ab/libvilattn_kvtiming.so python tools/kv_timing.py small_s1
"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vision_longformer_amd import _lib
from vision_longformer_amd.ops import vil_local_attention, vil_dense_attention
from tools.kernel_bench import SHAPES

NAMES = ["slot table build (+ global-query row staging)", "key slots + K/V fragment loads issued", "step: Q/dO ring -> LDS tile",
         "step: next loads issued + LDS fence", "step: fragment reads, bias gathers, S / dP MFMAs issued", "step: softmax (exp, dS, packs)",
         "step: transposed reads + dV / dK MFMAs issued", "whole step loop (incl. first loads)", "global-query rows", "epilogue stores issued",
         "unit total", "workgroup prologue (table image -> LDS, barrier)", "units", "steps", "wave lifetime", "waves"]


def main():
    shape = sys.argv[1] if len(sys.argv) > 1 else "small_s1"
    H, M, W, nx, ny, G, mode, B = SHAPES[shape]
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(300)
    C = H * M
    q = torch.randn(B, nx * ny, C, generator=g).to(dev, torch.bfloat16).requires_grad_(True)
    kv = torch.randn(B, G + nx * ny, 2 * C, generator=g).to(dev, torch.bfloat16).requires_grad_(True)
    dense = shape.endswith("_dense")
    if dense:
        qkv = torch.randn(B, G + nx * ny, 3 * C, generator=g).to(dev, torch.bfloat16).requires_grad_(True)
        table = (torch.randn((2 * W - 1) ** 2, H, generator=g) * 0.02).to(dev).requires_grad_(True)
        g2l2 = (torch.randn(2, H, G, generator=g) * 0.02).to(dev).requires_grad_(True)
        g2g = (torch.randn(H, G, G, generator=g) * 0.02).to(dev).requires_grad_(True)
        dout = torch.randn(B, G + nx * ny, C, generator=g).to(dev, torch.bfloat16)
    else:
        table = (torch.randn((4 * W - 1) ** 2, H, generator=g) * 0.02).to(dev).requires_grad_(True)
        g2l = (torch.randn(H, G, generator=g) * 0.02).to(dev).requires_grad_(True)
        dout = torch.randn(B, nx * ny, C, generator=g).to(dev, torch.bfloat16)
    L = _lib.lib()
    if not hasattr(L, "vil_debug_kv_timing"):
        raise SystemExit("this library was not built with -DVIL_KV_TIMING=1")
    L.vil_debug_kv_timing.argtypes = [ctypes.c_void_p, ctypes.c_int]
    buf = (ctypes.c_ulonglong * 16)()

    def run():
        if dense:
            out = vil_dense_attention(qkv, table, g2l2, g2g, nx=nx, ny=ny, nglo=G, num_heads=H, scale=M ** -0.5)
        else:
            out = vil_local_attention(q, kv, table, g2l, nx=nx, ny=ny, w=W, nglo=G, num_heads=H, mode=mode, backend="mfma")
        out.backward(dout)
        torch.cuda.synchronize()

    for _ in range(3):
        run()
    L.vil_debug_kv_timing(None, 1)
    reps = 5
    _lib.profile_begin(reps * 16)
    for _ in range(reps):
        run()
    recs = _lib.profile_end(reps * 16)
    kms = [ms for n, ms, by, fl in recs if n == "k_mfma_bwd_dkdv"]
    L.vil_debug_kv_timing(buf, 0)
    print(f"instrumented k_mfma_bwd_dkdv: {sum(kms) / len(kms) * 1e3:.1f} us per launch")
    v = [int(x) for x in buf]
    units, steps, waves = v[12], v[13], v[15]
    print(f"{shape}: {reps} launches, {units // reps} timed wave-units and {waves // reps} timed waves per launch, {steps / max(units, 1):.2f} steps per unit")
    print(f"{'segment':60s} {'cycles/unit':>12s} {'cycles/step':>12s} {'% of unit':>10s}")
    for i in (0, 1, 7, 2, 3, 4, 5, 6, 8, 9, 10):
        per_unit = v[i] / max(units, 1)
        per_step = v[i] / max(steps, 1) if i in (2, 3, 4, 5, 6, 7) else float("nan")
        print(f"{NAMES[i]:60s} {per_unit:12.0f} {per_step:12.0f} {100.0 * v[i] / max(v[10], 1):10.1f}")
    print(f"{NAMES[11]:60s} {v[11] / max(waves, 1):12.0f} cycles/wave")
    print(f"{NAMES[14]:60s} {v[14] / max(waves, 1):12.0f} cycles/wave  ({units / max(waves, 1):.2f} units per wave)")


if __name__ == "__main__":
    main()
