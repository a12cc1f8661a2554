"""Drop-in ``Long2DSCSelfAttention`` for MsViT blocks (ATTN_TYPE='longformerhand').

Same constructor signature, ``forward(x, nx, ny)`` contract, public attributes,
parameter / buffer names and random-shift RNG draw as the reference module
(src/models/layers/longformer2d.py:12-229), so checkpoints load unchanged and
``MsViT.reset_vil_mode`` keeps working.  The local-attention core is one call
into the HIP kernels (ops.vil_local_attention) instead of the reference's
chunk / roll / einsum / mask / softmax / einsum pipeline.

Differences by design (all numerically neutral):
  * with ``sharew=True`` the reference runs the kv GEMM twice on identical
    weights (longformer2d.py:127,211); here the first result is reused;
  * ``relative_position_index`` is still registered (state-dict compatibility)
    but the kernels compute the index arithmetically and never read it.
"""
import random

import torch
from torch import nn

from .ops import vil_local_attention, vil_full_attention, vil_full_attention_qkv, vil_global_attention, FULL_MAX_G
from .linear import VilLinear, vil_linear, vil_linear_pair


def _trunc_normal_(t, std):
    return nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2.0, b=2.0)


def build_relative_position_index(w):
    """(w^2, 9w^2) int64: for query l=(xl,yl) of the centre chunk and key t=(xt,yt)
    of neighbour nb=(dr+1)*3+(dc+1): (xl-dr*w-xt+2w-1)*(4w-1) + (yl-dc*w-yt+2w-1)
    (longformer2d.py:67-100)."""
    side = 4 * w - 1
    pos = torch.arange(w * w)
    px, py = pos // w, pos % w
    blocks = []
    for nb in range(9):
        dr, dc = nb // 3 - 1, nb % 3 - 1
        rx = px[:, None] - (dr * w + px)[None, :] + 2 * w - 1
        ry = py[:, None] - (dc * w + py)[None, :] + 2 * w - 1
        blocks.append(rx * side + ry)
    return torch.cat(blocks, dim=1)


class Long2DSCSelfAttention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0., w=7, d=1,
                 autoregressive=False, sharew=False, nglo=1, only_glo=False, exact=0, autograd=False, rpe=False,
                 mode=0):
        super().__init__()
        assert d == 1, "Dilation is not supported!"
        assert not autoregressive, "Autoregressive is not supported yet!"
        if only_glo:
            assert nglo >= 1, "Nglo == 0 in the only global mode!"
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = qk_scale or self.head_dim ** -0.5
        self.Nglo = nglo
        self.only_glo = only_glo
        self.attention_window = w
        self.attention_dilation = d
        self.autoregressive = autoregressive
        self.exact = exact
        self.autograd = autograd          # accepted for API parity; the fused kernels have one backward
        self.rpe = rpe
        self.mode = mode                  # 0: 3x3 chunks; -1: own chunk; >0: random-shift training
        self.backend = None               # kernel family override ("scalar" / "mfma"); None = library default
        self.mode_dev = None              # (1,) int32 device tensor: the random-shift neighbour is read by the
                                          # kernels at launch time (set by engine.GraphedTrainStep, which draws
                                          # the neighbour on the host before every hipGraph replay)

        self.query = VilLinear(dim, dim, bias=qkv_bias)
        self.kv = VilLinear(dim, dim * 2, bias=qkv_bias)
        self.proj = VilLinear(dim, dim)
        if nglo >= 1:
            if sharew:
                self.query_global, self.kv_global, self.proj_global = self.query, self.kv, self.proj
            else:
                self.query_global = VilLinear(dim, dim, bias=qkv_bias)
                self.kv_global = VilLinear(dim, dim * 2, bias=qkv_bias)
                self.proj_global = VilLinear(dim, dim)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj_drop = nn.Dropout(proj_drop)

        if rpe:
            self.local_relative_position_bias_table = nn.Parameter(torch.zeros((4 * w - 1) ** 2, num_heads))
            _trunc_normal_(self.local_relative_position_bias_table, .02)
            if nglo >= 1:
                self.g2l_relative_position_bias = nn.Parameter(torch.zeros(2, num_heads, nglo))
                self.g2g_relative_position_bias = nn.Parameter(torch.zeros(num_heads, nglo, nglo))
                _trunc_normal_(self.g2l_relative_position_bias, .02)
                _trunc_normal_(self.g2g_relative_position_bias, .02)
            self.register_buffer("relative_position_index", build_relative_position_index(w))

    def _resolve_mode(self):
        """longformer2d.py:114-123: one draw from Python's global `random` per
        training forward when mode > 0; evaluation always uses the full 3x3."""
        if self.mode > 0 and self.training and self.mode_dev is not None:
            return self.mode             # the neighbour comes from self.mode_dev (drawn by the graphed step)
        if self.mode > 0:
            return random.randrange(1, 9) if self.training else 0
        return self.mode

    def fused_path_ok(self):
        """One query / kv / proj GEMM over all tokens + one fused op for local and global rows: the shared-weight
        layers of every published ViL."""
        G, H, M = self.Nglo, self.num_heads, self.head_dim
        return (1 <= G <= FULL_MAX_G and self.query_global is self.query and self.kv_global is self.kv
                and self.proj_global is self.proj and not self.only_glo and M in (8, 16, 32, 48, 64))

    def _packed_projection_ok(self, x):
        q, kv = self.query, self.kv
        return (x.is_cuda and isinstance(q, VilLinear) and isinstance(kv, VilLinear) and q.weight.dtype == kv.weight.dtype
                and (q.bias is None) == (kv.bias is None))

    def forward(self, x, nx, ny):
        B, N, C = x.shape
        G, H, M = self.Nglo, self.num_heads, self.head_dim
        Nloc = nx * ny
        assert G + Nloc == N, "Global dimension does not match!"
        if self.training and self.attn_drop.p > 0.0:
            # the reference drops attention PROBABILITIES of local and global rows (longformer2d.py:186,224); the fused
            # kernels never materialise them.  No published config sets ATTN_DROP > 0: that case runs the reference's
            # own algorithm on the operator-level HIP kernels (scores materialised, as there)
            return self._forward_materialised(x, nx, ny)
        mode = self._resolve_mode()
        rs_dev = self.mode_dev is not None and self.mode > 0 and self.training
        table = self.local_relative_position_bias_table if self.rpe else None
        g2l = self.g2l_relative_position_bias if (self.rpe and G >= 1) else None
        g2g = self.g2g_relative_position_bias if (self.rpe and G >= 1) else None

        if self.fused_path_ok():
            # shared weights (every published ViL): ONE query / kv / proj GEMM over all N tokens and one
            # fused op for local + global rows -- no token slicing, no concatenation, and the two
            # gradient contributions to kv are summed inside the kernels
            kw = dict(nx=nx, ny=ny, w=self.attention_window, nglo=G, num_heads=H, mode=(1 if rs_dev else mode),
                      exact=self.exact, scale=self.scale, backend=self.backend, mode_dev=self.mode_dev if rs_dev else None)
            qkv = None
            if self._packed_projection_ok(x):
                # query and kv as ONE (3C, C) GEMM over the shared input: x is read once, and the backward is one input-
                # gradient GEMM and one weight-gradient kernel instead of two of each plus an accumulation pass.  With
                # 16-bit working weights (engine.MasterWeightOptimizer) the two Linears' parameters are rows of one packed
                # matrix (linear.pack_pair, once): no concatenation per forward, no split copies per backward.  Under
                # per-call autocast casts the casted copies are concatenated instead.
                qkv = vil_linear_pair(x, self.query, self.kv)
                if qkv is None:
                    wq, wkv = self.query.weight, self.kv.weight
                    bias = torch.cat([self.query.bias, self.kv.bias]) if self.query.bias is not None else None
                    qkv = vil_linear(x, torch.cat([wq, wkv], dim=0), bias)
            if qkv is not None:
                out = vil_full_attention_qkv(qkv, table, g2l, g2g, **kw)
            else:
                out = vil_full_attention(self.query(x), self.kv(x), table, g2l, g2g, **kw)
            return self.proj_drop(self.proj(out))

        if rs_dev:
            raise RuntimeError("mode_dev (device-side random-shift neighbour) is only read by the fused path; this "
                               "layer (sharew=False / only_glo / nglo outside 1..4) must draw its neighbour on the host")
        q = self.query(x[:, G:])                  # (B, Nloc, C), unscaled: the kernel applies `scale`
        kv = self.kv(x)                           # (B, N, 2C): [..., :C] keys, [..., C:] values
        x1 = vil_local_attention(q, kv, table, g2l[1] if g2l is not None else None, nx=nx, ny=ny,
                                 w=self.attention_window, nglo=G, num_heads=H, mode=mode, exact=self.exact,
                                 scale=self.scale, only_glo=self.only_glo, backend=self.backend)
        x1 = self.proj(x1)
        if G == 0:
            return self.proj_drop(x1)

        # global-token rows: full attention over all N keys (longformer2d.py:210-227) with the global projections
        qg = self.query_global(x[:, :G])          # unscaled
        kvg = kv if self.kv_global is self.kv else self.kv_global(x)
        if G <= FULL_MAX_G and M in (8, 16, 32, 48, 64):
            x0 = vil_global_attention(qg, kvg, g2g, g2l[0] if g2l is not None else None,
                                      nx=nx, ny=ny, nglo=G, num_heads=H, scale=self.scale)
        else:
            # nglo > 4 (no published model): the global-row kernels keep their bias-gradient partials for at most
            # four global tokens; this one case stays on library ops
            qg = (self.scale * qg).view(B, G, H, M)
            kvh = kvg.view(B, N, 2, H, M)
            a0 = torch.einsum('bghm,bnhm->bhgn', qg, kvh[:, :, 0])
            if self.rpe:
                a0 = a0 + torch.cat([g2g, g2l[0].unsqueeze(-1).expand(-1, -1, Nloc)], dim=-1).unsqueeze(0)
            a0 = torch.softmax(a0.float(), dim=-1).to(kvh.dtype)
            x0 = torch.einsum('bhgn,bnhm->bghm', a0, kvh[:, :, 1]).reshape(B, G, C)
        x0 = self.proj_global(x0)
        return self.proj_drop(torch.cat((x0, x1.to(x0.dtype)), dim=1))

    def _forward_materialised(self, x, nx, ny):
        """attn_drop > 0 in training: sliding-chunk scores are materialised by the operator-level HIP kernels
        (slidingchunk_2d.slidingchunk_2d / mask_invalid_locations, the reference's SlidingChunk2D surface), the softmax
        and the dropout of the probabilities are tensor ops -- the semantics of longformer2d.py:134-229 with dropout."""
        from .slidingchunk_2d import slidingchunk_2d, mask_invalid_locations
        import torch.nn.functional as F
        B, N, C = x.shape
        G, H, M, W = self.Nglo, self.num_heads, self.head_dim, self.attention_window
        Nloc, W2 = nx * ny, W * W
        mode = self._resolve_mode()
        q = (self.scale * self.query(x[:, G:])).float().view(B, Nloc, H, M)
        kv = self.kv(x).float().view(B, N, 2, H, M)
        k, v = kv[:, :, 0], kv[:, :, 1]                                   # (B, N, H, M)
        kg, vg = k[:, :G].permute(0, 2, 1, 3), v[:, :G].permute(0, 2, 1, 3)      # (B, H, G, M)
        if self.only_glo:
            s1 = torch.einsum("bnhm,bhgm->bhng", q, kg)
            if self.rpe:
                s1 = s1 + self.g2l_relative_position_bias[1][None, :, None, :]
            p1 = self.attn_drop(torch.softmax(s1, dim=-1))
            x1 = torch.einsum("bhng,bhgm->bnhm", p1, vg).reshape(B, Nloc, C)
        else:
            padx, pady = (W - nx % W) % W, (W - ny % W) % W
            mx, my = (nx + padx) // W, (ny + pady) // W

            def chunked(t):                                              # (B, Nloc, H, M) -> (B*H, M, mx, my, W^2)
                t = t.view(B, nx, ny, H, M).permute(0, 3, 4, 1, 2)
                t = F.pad(t, (0, pady, 0, padx))
                return t.reshape(B * H, M, mx, W, my, W).permute(0, 1, 2, 4, 3, 5).reshape(B * H, M, mx, my, W2).contiguous()

            qi, ki, vi = chunked(q), chunked(k[:, G:]), chunked(v[:, G:])
            s11 = slidingchunk_2d(qi, ki, False, mode)                   # (BH, mx, my, W^2, kv)
            kvn = s11.shape[-1]
            if self.rpe:
                if mode == 0:
                    idx = self.relative_position_index
                elif mode == -1:
                    idx = self.relative_position_index[:, 4 * W2:5 * W2]
                else:
                    cid = mode if mode > 4 else mode - 1
                    idx = torch.cat([self.relative_position_index[:, 4 * W2:5 * W2],
                                     self.relative_position_index[:, cid * W2:(cid + 1) * W2]], dim=-1)
                bias = self.local_relative_position_bias_table[idx.reshape(-1)].view(W2, kvn, H).permute(2, 0, 1)
                s11 = (s11.view(B, H, mx, my, W2, kvn) + bias[None, :, None, None]).view(B * H, mx, my, W2, kvn)
            mask_invalid_locations(s11, mx, my, padx, pady, W, self.exact, mode)
            if G >= 1:
                s10 = torch.einsum("bcmnl,bgc->bmnlg", qi, kg.reshape(B * H, G, M))
                if self.rpe:
                    s10 = s10 + self.g2l_relative_position_bias[1].repeat(B, 1)[:, None, None, None, :]
                s1 = torch.cat((s10, s11), dim=-1)
            else:
                s1 = s11
            p1 = self.attn_drop(torch.softmax(s1, dim=-1))
            x1 = slidingchunk_2d(p1[..., G:].contiguous(), vi, True, mode)          # (BH, M, mx, my, W^2)
            if G >= 1:
                x1 = x1 + torch.einsum("bmnlg,bgc->bcmnl", p1[..., :G], vg.reshape(B * H, G, M))
            x1 = x1.view(B, H, M, mx, my, W, W).permute(0, 3, 5, 4, 6, 1, 2).reshape(B, mx * W, my * W, C)
            x1 = x1[:, :nx, :ny].reshape(B, Nloc, C)
        x1 = self.proj(x1.to(x.dtype))
        if G == 0:
            return self.proj_drop(x1)
        qg = (self.scale * self.query_global(x[:, :G])).float().view(B, G, H, M)
        kvg = self.kv_global(x).float().view(B, N, 2, H, M)
        s0 = torch.einsum("bghm,bnhm->bhgn", qg, kvg[:, :, 0])
        if self.rpe:
            s0 = s0 + torch.cat([self.g2g_relative_position_bias,
                                 self.g2l_relative_position_bias[0].unsqueeze(-1).expand(-1, -1, Nloc)], dim=-1)[None]
        p0 = self.attn_drop(torch.softmax(s0, dim=-1))
        x0 = torch.einsum("bhgn,bnhm->bghm", p0, kvg[:, :, 1]).reshape(B, G, C)
        x0 = self.proj_global(x0.to(x.dtype))
        return self.proj_drop(torch.cat((x0, x1.to(x0.dtype)), dim=1))

    @staticmethod
    def compute_macs(module, input, output):
        """MACs of one forward, same accounting as the reference hook body
        (longformer2d.py:231-279): attention products + projections."""
        _, T, C = input[0].shape
        G, W = module.Nglo, module.attention_window
        kq = (C - G) * G * C if module.only_glo else (C - G) * (9 * W ** 2) * C + (C - G) * G * C
        kq += G * T * C
        macs = 2 * kq
        for lin in (module.query, module.kv, module.proj):
            macs += sum(p.numel() for p in lin.parameters()) * T
        module.__flops__ += macs
