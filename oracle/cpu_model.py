"""CPU baseline model for bench.py's `cpu_baseline` leg: the package's MsViT built on
the CPU with every hot-path layer's forward replaced by the ORACLE restatement
(oracle/vil_oracle.long2dsc_forward).  TEST / MEASUREMENT INFRASTRUCTURE ONLY -- the
product path never imports this (and has no CPU path at all)."""
import random
import types

import torch

from . import vil_oracle as O


def _oracle_forward(self, x, nx, ny):
    params = {n: p for n, p in self.named_parameters(remove_duplicate=False)}
    mode = self.mode
    if self.mode > 0:
        mode = random.randrange(1, 9) if self.training else 0
    return O.long2dsc_forward(params, x, nx, ny, num_heads=self.num_heads, w=self.attention_window,
                              nglo=self.Nglo, rpe=self.rpe, exact=self.exact, mode=mode,
                              only_glo=self.only_glo, qk_scale=self.scale)


def build_cpu_baseline_model(config, **kw):
    from vision_longformer_amd.engine import build_vil
    from vision_longformer_amd.longformer2d import Long2DSCSelfAttention
    model = build_vil(config, **kw)
    for m in model.modules():
        if isinstance(m, Long2DSCSelfAttention):
            m.forward = types.MethodType(_oracle_forward, m)
    return model
