"""Round 6 diagnostic: cycle stamps of the chunk-workgroup forward's pipelined step (a -DVIL_CW_ABLATE build loaded through
VIL_ATTN_LIB): per-segment cycles per step, averaged over the waves of every workgroup.

    VIL_ATTN_LIB=tools/ab/libvilattn_abl.so python tools/cw_stamps.py small_s1 [extra ablation bits]"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vision_longformer_amd import _lib
_lib.use_library_for_ab(os.environ["VIL_ATTN_LIB"])
from vision_longformer_amd.ops import vil_local_attention
sys.path.insert(0, os.path.join(ROOT, "tools"))
from kernel_bench import SHAPES
from cw_check import inputs

name = sys.argv[1] if len(sys.argv) > 1 else "small_s1"
extra = int(sys.argv[2]) if len(sys.argv) > 2 else 0
shape = SHAPES[name]
H, M, W, nx, ny, G, mode, B = shape
dev = torch.device("cuda:0")
q, kv, table, g2l, g2g, dout = inputs(shape, False, dev)
L = _lib.lib()
nrec = 8 * 4096 * 16 * 8
dbg = torch.zeros(nrec * 16, dtype=torch.int64, device=dev)
L.vil_attn_cw_set_debug(ctypes.c_void_p(dbg.data_ptr()))
kw = dict(nx=nx, ny=ny, w=W, nglo=G, num_heads=H, mode=mode, backend="mfma")
for _ in range(2):
    vil_local_attention(q, kv, table, g2l if G else None, **kw)
L.vil_attn_cw_set_ablation(256 | extra)
dbg.zero_()
torch.cuda.synchronize()
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record(); vil_local_attention(q, kv, table, g2l if G else None, **kw); t1.record()
torch.cuda.synchronize()
r = dbg.view(-1, 16).cpu().double()
r = r[r.sum(1) > 0]
names = ["loop ctl", "wait+barrier", "request(+q)", "LDS reads issued", "probs", "PV mfma", "S mfma", "gq", "boundary", "read q / slot", "-", "-", "prologue", "tail", "-", "-"]
tot = r.sum(1).mean()
print(f"{name}: kernel {t0.elapsed_time(t1) * 1e3:.1f} us, {r.shape[0]} waves, cycles per wave {tot:.0f}")
steps = None
for k in range(16):
    v = r[:, k].mean()
    if v > 0:
        print(f"  {names[k]:18s} {v:12.0f} cycles per wave  {100 * v / tot:5.1f} %")
