import sys, os, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import torch
from vision_longformer_amd import _lib
from vision_longformer_amd.ops import vil_local_attention, vil_full_attention, vil_global_attention
from kernel_bench import SHAPES
from cw_check import inputs
dev = torch.device("cuda:0")
for name in sys.argv[1].split(","):
    shape = SHAPES[name]
    H, M, W, nx, ny, G, mode, B = shape
    q, kv, table, g2l, g2g, dout = inputs(shape, True, dev)
    for t in (q, kv, table, g2l, g2g): t.requires_grad_(True)
    kw = dict(nx=nx, ny=ny, w=W, nglo=G, num_heads=H, mode=mode)
    def full():
        out = vil_full_attention(q, kv, table, g2l, g2g, **kw); out.backward(dout)
    def two():
        ol = vil_local_attention(q[:, G:], kv, table, g2l[1], **kw)
        og = vil_global_attention(q[:, :G], kv, g2g, g2l[0], nx=nx, ny=ny, nglo=G, num_heads=H)
        torch.cat([og, ol], 1).backward(dout)
    res = {}
    for nm, fn in (("full", full), ("two", two)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        _lib.profile_begin(4000)
        for _ in range(10): fn()
        torch.cuda.synchronize()
        recs = _lib.profile_end(4000)
        agg = {}
        for n, ms, by, fl in recs:
            agg[n] = agg.get(n, 0) + ms
        res[nm] = {k: round(v * 100, 1) for k, v in agg.items()}
        res[nm]["sum"] = round(sum(agg.values()) * 100, 1)
    print(name, json.dumps(res))
