"""One process that launches every hand-written hot kernel at its bench shapes, for `rocprofv3 --pmc` passes
(tools/pmc_all.sh): the sliding-chunk attention family at 56x56 (M 32) and 28x28 / 48x48 (M 64), the dense family at 14x14
and 24x24, the weights-in-registers GEMM (forward, input gradient), the weight gradient, fc2's input gradient with the GELU
backward, fc1 with the GELU epilogue, the same tile kernels as plain GEMMs.  Between two cases a marker kernel (a torch.sign whose grid size grows with the case
index) is launched, so that tools/pmc_all_summary.py can cut the dispatch stream into cases without any other side channel.

    python tools/pmc_all.py [--reps 2] [--only case,case]      (prints the case list as JSON on the last line)
"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from vision_longformer_amd import _lib  # noqa: E402
if os.environ.get("VIL_ATTN_LIB"):
    _lib.use_library_for_ab(os.environ["VIL_ATTN_LIB"])
from vision_longformer_amd.ops import vil_full_attention, vil_dense_attention  # noqa: E402

MARK_UNIT = 1 << 20

ATTN = {  # H, M, W, nx, ny, G, mode, B
    "sc_56x56_m32": (3, 32, 7, 56, 56, 1, 0, 128),
    "sc_28x28_m64": (3, 64, 7, 28, 28, 1, 0, 128),
    "sc_48x48_m64": (3, 64, 7, 48, 48, 1, 0, 32),
    "sc_96x96_m32": (3, 32, 7, 96, 96, 1, 0, 32),
    "sc_48x48_w12_m64": (3, 64, 12, 48, 48, 1, 0, 32),          # the one MFMA-bound shape (AI 648): Medium-Deep f8 / f12 stage 2
    "sc_96x96_w6_m32_rs": (3, 32, 6, 96, 96, 1, 3, 32),         # Base-Deep stage 1, random shift
    "dense_14x14": (6, 64, 14, 14, 14, 1, -1, 128),
    "dense_24x24": (6, 64, 24, 24, 24, 1, -1, 32),
}
GEMM = {  # kind, T, K, N
    "skinny_fwd_s1_qkv": ("skinny0", 401536, 96, 288),
    "skinny_dgrad_s1_qkv": ("skinny1", 401536, 288, 96),
    "skinny_fwd_s2_fc2": ("skinny0", 100480, 768, 192),
    "fwd_gelu_s3": ("gelu", 25216, 384, 1536),
    "dgrad_dgelu_s3": ("dgelu", 25216, 384, 1536),
    "dgrad_dgelu_s1": ("dgelu", 401536, 96, 384),
    "tile_fwd_s3_qkv": ("tile0", 25216, 384, 1152),
    "tile_dgrad_s3_proj": ("tile1", 25216, 384, 384),
    "wgrad_s1_fc1": ("wgrad", 401536, 384, 96),
    "wgrad_s3_fc1": ("wgrad", 25216, 1536, 384),
}


def vp(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def attn_case(name, dev, fwd_only=False):
    H, M, W, nx, ny, G, mode, B = ATTN[name]
    g = torch.Generator(device="cpu").manual_seed(300)
    C = H * M
    if name.startswith("dense"):
        qkv = torch.randn(B, G + nx * ny, 3 * C, generator=g).to(dev, torch.bfloat16).requires_grad_(True)
        table = (torch.randn((2 * W - 1) ** 2, H, generator=g) * 0.02).to(dev).requires_grad_(True)
        g2l2 = (torch.randn(2, H, G, generator=g) * 0.02).to(dev).requires_grad_(True)
        g2g = (torch.randn(H, G, G, generator=g) * 0.02).to(dev).requires_grad_(True)
        dout = torch.randn(B, G + nx * ny, C, generator=g).to(dev, torch.bfloat16)

        def step():
            out = vil_dense_attention(qkv, table, g2l2, g2g, nx=nx, ny=ny, nglo=G, num_heads=H, scale=M ** -0.5)
            if not fwd_only:
                out.backward(dout)
        return step
    # the whole layer as the model runs it (round 5): local AND global rows, vil_attn_fwd_full / vil_attn_bwd_full
    q = torch.randn(B, G + nx * ny, C, generator=g).to(dev, torch.bfloat16).requires_grad_(True)
    kv = torch.randn(B, G + nx * ny, 2 * C, generator=g).to(dev, torch.bfloat16).requires_grad_(True)
    table = (torch.randn((4 * W - 1) ** 2, H, generator=g) * 0.02).to(dev).requires_grad_(True)
    g2l = (torch.randn(2, H, G, generator=g) * 0.02).to(dev).requires_grad_(True)
    g2g = (torch.randn(H, G, G, generator=g) * 0.02).to(dev).requires_grad_(True)
    dout = torch.randn(B, G + nx * ny, C, generator=g).to(dev, torch.bfloat16)

    def step():
        out = vil_full_attention(q, kv, table, g2l, g2g, nx=nx, ny=ny, w=W, nglo=G, num_heads=H, mode=mode)
        if not fwd_only:
            out.backward(dout)
    return step


def gemm_case(name, dev):
    kind, T, K, N = GEMM[name]
    L = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(301)
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    if kind in ("skinny0", "skinny1"):
        op = int(kind[-1])
        x = torch.randn(T, K, generator=g).bfloat16().to(dev)
        w = (torch.randn(*((N, K) if op == 0 else (K, N)), generator=g) * 0.1).bfloat16().to(dev)
        b = torch.randn(N, generator=g).bfloat16().to(dev) if op == 0 else None
        out = torch.empty(T, N, dtype=torch.bfloat16, device=dev)
        return lambda: _lib.check(L.vil_gemm_skinny_bf16(op, vp(x), vp(w), vp(b), vp(out), T, K, N, x.stride(0), N, st()))
    if kind in ("tile0", "tile1"):
        op = int(kind[-1])
        x = torch.randn(T, K, generator=g).bfloat16().to(dev)
        w = (torch.randn(*((N, K) if op == 0 else (K, N)), generator=g) * 0.1).bfloat16().to(dev)
        b = torch.randn(N, generator=g).bfloat16().to(dev) if op == 0 else None
        out = torch.empty(T, N, dtype=torch.bfloat16, device=dev)
        return lambda: _lib.check(L.vil_gemm_tile_bf16(op, vp(x), vp(w), vp(b), vp(out), T, K, N, x.stride(0), N, st()))
    if kind == "gelu":
        x = torch.randn(T, K, generator=g).bfloat16().to(dev)
        w = (torch.randn(N, K, generator=g) * 0.1).bfloat16().to(dev)
        b = torch.randn(N, generator=g).bfloat16().to(dev)
        both = torch.empty(2, T, N, dtype=torch.bfloat16, device=dev)
        return lambda: _lib.check(L.vil_gemm_gelu_bf16(vp(x), vp(w), vp(b), vp(both[0]), vp(both[1]), T, K, N, x.stride(0), N, st()))
    if kind == "dgelu":
        dy = torch.randn(T, K, generator=g).bfloat16().to(dev)
        w = (torch.randn(K, N, generator=g) * 0.1).bfloat16().to(dev)
        h = torch.randn(T, N, generator=g).bfloat16().to(dev)
        dh = torch.empty(T, N, dtype=torch.bfloat16, device=dev)
        return lambda: _lib.check(L.vil_gemm_dgelu_bf16(vp(dy), vp(w), vp(h), vp(dh), T, K, N, dy.stride(0), h.stride(0), N, st()))
    if kind == "wgrad":
        CO, CI = K, N
        dy = (torch.randn(T, CO, generator=g) * 0.1).bfloat16().to(dev)
        x = torch.randn(T, CI, generator=g).bfloat16().to(dev)
        ws = torch.empty(L.vil_linear_wgrad_workspace_bytes(T, CO, CI) // 4 + 64, dtype=torch.float32, device=dev)
        dw = torch.empty(CO, CI, dtype=torch.float32, device=dev)
        db = torch.empty(CO, dtype=torch.float32, device=dev)
        return lambda: _lib.check(L.vil_linear_wgrad(vp(dy), vp(x), T, CO, CI, dy.stride(0), x.stride(0), vp(dw), vp(db), 0, vp(ws), st()))
    raise KeyError(kind)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--only", default="")
    ap.add_argument("--fwd-only", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    names = [n for n in list(ATTN) + list(GEMM) if not a.only or n in a.only.split(",")]
    mark = torch.zeros(MARK_UNIT * (len(names) + 2), dtype=torch.int32, device=dev)
    times = {}
    for i, n in enumerate(names):
        step = attn_case(n, dev, a.fwd_only) if n in ATTN else gemm_case(n, dev)
        torch.sign(mark[:MARK_UNIT * (i + 1)], out=mark[:MARK_UNIT * (i + 1)])   # marker: the only sign kernel of the process, grid size grows with i
        step()                                   # warm-up (workspace allocation); its dispatches count into the case's means
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            step()
        e1.record()
        torch.cuda.synchronize()
        times[n] = e0.elapsed_time(e1) / a.reps * 1e3
        del step
        torch.cuda.empty_cache()
    torch.sign(mark[:MARK_UNIT * (len(names) + 1)], out=mark[:MARK_UNIT * (len(names) + 1)])
    torch.cuda.synchronize()
    print(json.dumps({"cases": names, "reps": a.reps, "us_per_rep": {k: round(v, 1) for k, v in times.items()}}))


if __name__ == "__main__":
    main()
