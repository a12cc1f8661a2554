"""Host model: multi-scale Vision Longformer (MsViT) around the HIP hot path.  On device tensors every layer
type runs on libvilattn.so: sliding-chunk and dense attention (ops.py), the projections and the patch embedding
(linear.py: hipBLASLt GEMMs with measured algorithm choice, fused weight/bias gradient), LayerNorm fused with the
preceding residual add (layernorm.py; `MsViT._run_stage`).  GELU and the loss are stock PyTorch kernels.

Only what the `longformerhand` path needs to be exercised end to end is here:
the arch-string parser, PatchEmbed, the dense `Attention` of the `s0` stages,
MlpBlock, AttnBlock (the drop-in boundary: ``x + drop_path(attn(norm(x), nx, ny))``,
reference src/models/msvit.py:313-316) and the MsViT container with
``reset_vil_mode``.  Module / parameter names follow the reference
(src/models/msvit.py) so that its checkpoints load unchanged; other attention
types of the reference (linformer, performer, srformer) are out of scope.
"""
from functools import partial

import torch
import torch.nn.functional as F
from torch import nn

from .longformer2d import Long2DSCSelfAttention, _trunc_normal_
from .ops import vil_dense_attention, dense_family_supported, FULL_MAX_G
from .layernorm import (VilLayerNorm, res_layernorm, res_layernorm_ok, tokens_layernorm, tokens_layernorm_ok,
                        pass_layernorm, pass_layernorm_ok)
from .linear import VilLinear, vil_linear, vil_gelu_linear, vil_linear_gelu, expand_rows


class DropPath(nn.Module):
    """Stochastic depth per sample."""

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = float(drop_prob)

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = torch.empty((x.shape[0],) + (1,) * (x.dim() - 1), dtype=x.dtype, device=x.device).bernoulli_(keep)
        return x * (mask / keep)


class _DropPathAdd(torch.autograd.Function):
    """x + y * mask (mask = per-sample keep / keep_prob) as ONE kernel each way: the residual add of
    AttnBlock / MlpBlock with stochastic depth folded in (instead of mul, then add, then their backwards)."""

    @staticmethod
    def forward(ctx, x, y, mask):
        ctx.save_for_backward(mask)
        ctx.ydtype = y.dtype
        return torch.addcmul(x, y, mask)

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return g, (g * mask).to(ctx.ydtype), None


class _Patchify(torch.autograd.Function):
    """Stage transition as one row gather each way (vil_patchify_fwd / _bwd): the pending `x + drop_path(branch)` of the
    stage's last block, `x[:, G:]`, the regrouping of every ph x pw patch's tokens into one (py, px, c) vector and the cast
    to the GEMM dtype; the backward scatters the patch gradient home (global-token rows zero) and emits the branch's
    gradient.  Reference msvit.py:500-507 + the strided Conv2d of PatchEmbed (:166-203)."""

    @staticmethod
    def forward(ctx, x, branch, rscale, G, nx, ny, ph, pw, out_dtype):
        from . import _lib
        import ctypes
        L = _lib.lib()
        B, _, C = x.shape
        x = x.contiguous()
        if branch is not None:
            branch = branch.contiguous()
        patches = torch.empty(B * (nx // ph) * (ny // pw), ph * pw * C, dtype=out_dtype, device=x.device)
        vp = ctypes.c_void_p
        dt = {torch.float32: _lib.DTYPE_F32, torch.bfloat16: _lib.DTYPE_BF16}
        with torch.cuda.device(x.device):
            _lib.check(L.vil_patchify_fwd(vp(x.data_ptr()), vp(branch.data_ptr()) if branch is not None else None,
                                          dt[branch.dtype] if branch is not None else 0,
                                          vp(rscale.data_ptr()) if rscale is not None else None, vp(patches.data_ptr()),
                                          dt[out_dtype], B, G, nx, ny, C, ph, pw,
                                          vp(torch.cuda.current_stream(x.device).cuda_stream)))
        ctx.save_for_backward(rscale)
        ctx.cfg = (B, G, nx, ny, C, ph, pw, x.shape, branch.dtype if branch is not None else None)
        return patches

    @staticmethod
    def backward(ctx, dp):
        from . import _lib
        import ctypes
        L = _lib.lib()
        (rscale,) = ctx.saved_tensors
        B, G, nx, ny, C, ph, pw, xshape, bdtype = ctx.cfg
        dt = {torch.float32: _lib.DTYPE_F32, torch.bfloat16: _lib.DTYPE_BF16}
        dp = dp.contiguous()
        if dp.dtype not in dt:
            dp = dp.float()
        dx = torch.empty(xshape, dtype=torch.float32, device=dp.device)
        gb = torch.empty(xshape, dtype=bdtype, device=dp.device) if bdtype is not None else None
        vp = ctypes.c_void_p
        with torch.cuda.device(dp.device):
            _lib.check(L.vil_patchify_bwd(vp(dp.data_ptr()), dt[dp.dtype], vp(rscale.data_ptr()) if rscale is not None else None,
                                          vp(dx.data_ptr()), vp(gb.data_ptr()) if gb is not None else None,
                                          dt[bdtype] if bdtype is not None else 0, B, G, nx, ny, C, ph, pw,
                                          vp(torch.cuda.current_stream(dp.device).cuda_stream)))
        return dx, gb, None, None, None, None, None, None, None


def residual_drop_path(x, y, drop_prob, training):
    """x + drop_path(y): stochastic depth per sample (reference msvit.py:313-316,336-340 with timm's DropPath)."""
    if drop_prob == 0.0 or not training:
        return x + y
    keep = 1.0 - drop_prob
    mask = torch.empty((x.shape[0],) + (1,) * (x.dim() - 1), dtype=x.dtype, device=x.device).bernoulli_(keep).div_(keep)
    return _DropPathAdd.apply(x, y, mask)


class _TableGather(torch.autograd.Function):
    """table[index] whose backward is an atomic index_add_ instead of PyTorch's sort-based
    index-put backward (a radix sort + merge kernels per dense attention layer and step)."""

    @staticmethod
    def forward(ctx, table, index):
        ctx.save_for_backward(index)
        ctx.shape = table.shape
        return table[index]

    @staticmethod
    def backward(ctx, g):
        (index,) = ctx.saved_tensors
        out = torch.zeros(ctx.shape, dtype=g.dtype, device=g.device)
        out.index_add_(0, index, g)
        return out, None


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        self.fc1 = VilLinear(in_features, hidden_features or in_features)
        self.act = act_layer()
        self.fc2 = VilLinear(hidden_features or in_features, out_features or in_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        if isinstance(self.act, nn.GELU) and self.act.approximate == "none" and (self.drop.p == 0.0 or not self.training):
            # fc1 with the GELU in its epilogue where the weights-in-registers GEMM serves the shape (csrc/vil_gemm_skinny.hip),
            # fc2 with the GELU in front of it as one autograd node: its input gradient is ONE launch (dgrad GEMM with the
            # erf-GELU backward in the epilogue, csrc/vil_gemm_fused.hip)
            h, a = vil_linear_gelu(x, self.fc1.weight, self.fc1.bias)
            return vil_gelu_linear(h, self.fc2.weight, self.fc2.bias, a)
        return self.drop(self.fc2(self.drop(self.act(self.fc1(x)))))


class Attention(nn.Module):
    """Dense multi-head attention with optional relative position bias over an
    (nglo + wx*wy)-token sequence: the `s0` stages (reference msvit.py:37-120)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0.,
                 rpe=False, wx=14, wy=14, nglo=1):
        super().__init__()
        self.num_heads = num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.qkv = VilLinear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = VilLinear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.rpe = rpe
        if rpe:
            self.wx, self.wy, self.nglo = wx, wy, nglo
            self.local_relative_position_bias_table = nn.Parameter(torch.zeros((2 * wx - 1) * (2 * wy - 1), num_heads))
            _trunc_normal_(self.local_relative_position_bias_table, .02)
            if nglo >= 1:
                self.g2l_relative_position_bias = nn.Parameter(torch.zeros(2, num_heads, nglo))
                self.g2g_relative_position_bias = nn.Parameter(torch.zeros(num_heads, nglo, nglo))
                _trunc_normal_(self.g2l_relative_position_bias, .02)
                _trunc_normal_(self.g2g_relative_position_bias, .02)
            ix, iy = torch.meshgrid(torch.arange(wx), torch.arange(wy), indexing="ij")
            ix, iy = ix.reshape(-1), iy.reshape(-1)
            rel = (ix[:, None] - ix[None, :] + wx - 1) * (2 * wy - 1) + (iy[:, None] - iy[None, :] + wy - 1)
            self.register_buffer("relative_position_index", rel)

    def _bias(self, N):
        """(H, N, N) additive bias: [g2g | g2l[0]] rows for global queries,
        [g2l[1] | table gather] rows for local ones (msvit.py:88-111)."""
        L = self.wx * self.wy
        assert N == self.nglo + L, "For relative position, N != self.nglo + self.wx*self.wy!"
        loc = _TableGather.apply(self.local_relative_position_bias_table, self.relative_position_index.reshape(-1))
        loc = loc.view(L, L, -1).permute(2, 0, 1)
        if self.nglo == 0:
            return loc
        top = torch.cat([self.g2g_relative_position_bias,
                         self.g2l_relative_position_bias[0].unsqueeze(-1).expand(-1, -1, L)], dim=-1)
        bot = torch.cat([self.g2l_relative_position_bias[1].unsqueeze(1).expand(-1, L, -1), loc], dim=-1)
        return torch.cat([top, bot], dim=1)

    def forward(self, x, nx=None, ny=None):
        B, N, C = x.shape
        H = self.num_heads
        qkv = self.qkv(x)
        M = C // H
        nglo = self.nglo if self.rpe else (N - nx * ny if nx is not None else -1)
        gx, gy = (self.wx, self.wy) if self.rpe else (nx, ny)
        # fp32 (the reference's CPU / test precision) takes the one-chunk case of the fp32 matrix-core kernels (round 4; it
        # used to fall to scaled_dot_product_attention with a materialised (H, N, N) bias)
        ok = (qkv.is_cuda and qkv.dtype in (torch.bfloat16, torch.float16, torch.float32) and gx is not None and 0 <= nglo
              and nglo + gx * gy == N and (self.attn_drop.p == 0.0 or not self.training))
        if ok and (dense_family_supported(qkv, gx, gy, nglo, H)
                   or (gx == gy and gx <= 32 and nglo <= FULL_MAX_G and M in (16, 32, 48, 64))):
            # the dense attention of the s0 stages on its own HIP kernels (csrc/vil_attn_dense.hip:
            # head_dim 64, any grid) or, for the other head sizes, as the one-chunk case of the fused sliding-chunk kernels
            # -- bias table gathered in-kernel, no (H,N,N) bias tensor, no SDPA mask
            out = vil_dense_attention(qkv, self.local_relative_position_bias_table if self.rpe else None,
                                      self.g2l_relative_position_bias if (self.rpe and nglo > 0) else None,
                                      self.g2g_relative_position_bias if (self.rpe and nglo > 0) else None,
                                      nx=gx, ny=gy, nglo=nglo, num_heads=H, scale=self.scale)
            return self.proj_drop(self.proj(out))
        q, k, v = qkv.view(B, N, 3, H, C // H).permute(2, 0, 3, 1, 4)
        mask = self._bias(N).unsqueeze(0).to(q.dtype) if self.rpe else None
        p = self.attn_drop.p if self.training else 0.0
        x = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, dropout_p=p, scale=self.scale)
        x = x.transpose(1, 2).reshape(B, N, C)
        return self.proj_drop(self.proj(x))


class PatchEmbed(nn.Module):
    """Strided-conv patch embedding + optional LayerNorm, global (cls) tokens
    prepended, optional absolute 2-D position embedding (msvit.py:159-224)."""

    def __init__(self, patch_size, nx, ny, in_chans=3, embed_dim=768, nglo=1, norm_layer=nn.LayerNorm,
                 norm_embed=True, drop_rate=0.0, ape=True):
        super().__init__()
        ps = patch_size if isinstance(patch_size, tuple) else (patch_size, patch_size)
        self.patch_size = ps
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=ps, stride=ps)
        self.norm_embed = norm_layer(embed_dim) if norm_embed else None
        if isinstance(self.norm_embed, VilLayerNorm):
            self.norm_embed.cast_output = False       # feeds the fp32 residual stream, not a GEMM
        self.nx, self.ny, self.Nglo = nx, ny, nglo
        if nglo >= 1:
            self.cls_token = nn.Parameter(torch.zeros(1, nglo, embed_dim))
            _trunc_normal_(self.cls_token, .02)
        else:
            self.cls_token = None
        self.ape = ape
        if ape:
            self.cls_pos_embed = nn.Parameter(torch.zeros(1, nglo, embed_dim))
            self.x_pos_embed = nn.Parameter(torch.zeros(1, nx, embed_dim // 2))
            self.y_pos_embed = nn.Parameter(torch.zeros(1, ny, embed_dim // 2))
            for t in (self.cls_pos_embed, self.x_pos_embed, self.y_pos_embed):
                _trunc_normal_(t, .02)
        self.pos_drop = nn.Dropout(p=drop_rate)

    def _embed(self, img):
        """The strided conv of non-overlapping patches IS a Linear over (c, py, px) patch vectors: on the device
        it runs as one GEMM with the library's weight/bias gradient (token-major output, no transpose; and no
        multi-block bias-gradient reduction, which hipGraph replay gets wrong on this stack)."""
        ph, pw = self.patch_size
        B, Cin, Hh, Ww = img.shape
        if not img.is_cuda or Hh % ph or Ww % pw or (Cin * ph * pw) % 8 or self.proj.out_channels % 8:
            x = self.proj(img)
            nx, ny = x.shape[2], x.shape[3]
            return x.flatten(2).transpose(1, 2), nx, ny
        nx, ny = Hh // ph, Ww // pw
        if torch.is_autocast_enabled("cuda"):
            img = img.to(torch.get_autocast_dtype("cuda"))
        patches = img.view(B, Cin, nx, ph, ny, pw).permute(0, 2, 4, 1, 3, 5).reshape(B * nx * ny, Cin * ph * pw)
        y = vil_linear(patches, self.proj.weight.view(self.proj.out_channels, -1), self.proj.bias)
        return y.view(B, nx * ny, -1), nx, ny

    def _embed_tokens(self, xt, pnx, pny):
        """Same projection fed from the previous stage's TOKEN-major output (B, pnx*pny, Cin): the patch
        vectors are gathered in (py, px, c) order with one cast+permute copy (contiguous Cin-chunks) and the
        weight is permuted to match, instead of materialising the (B, Cin, pnx, pny) image first."""
        ph, pw = self.patch_size
        B, _, Cin = xt.shape
        nx, ny = pnx // ph, pny // pw
        dt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else xt.dtype
        pv = xt.reshape(B, nx, ph, ny, pw, Cin).permute(0, 1, 3, 2, 4, 5)
        patches = pv.contiguous() if xt.dtype == dt else pv.to(dtype=dt, memory_format=torch.contiguous_format)
        w = self.proj.weight.permute(0, 2, 3, 1).reshape(self.proj.out_channels, ph * pw * Cin)
        y = vil_linear(patches.reshape(B * nx * ny, ph * pw * Cin), w, self.proj.bias)
        return y.view(B, nx * ny, -1), nx, ny

    def tokens_ok(self, xt, pnx, pny):
        ph, pw = self.patch_size
        return (xt.is_cuda and pnx % ph == 0 and pny % pw == 0 and (xt.shape[-1] * ph * pw) % 8 == 0
                and self.proj.out_channels % 8 == 0)

    def _embed_patches(self, patches, pnx, pny):
        """Projection of ready-made (py, px, c) patch vectors (B * nx * ny, ph * pw * Cin) -- MsViT's fused stage transition"""
        ph, pw = self.patch_size
        nx, ny = pnx // ph, pny // pw
        Cin = patches.shape[1] // (ph * pw)
        w = self.proj.weight.permute(0, 2, 3, 1).reshape(self.proj.out_channels, ph * pw * Cin)
        y = vil_linear(patches, w, self.proj.bias)
        return y.view(-1, nx * ny, self.proj.out_channels), nx, ny

    def forward(self, xtuple, tokens=False):
        if tokens == "patches":
            x, nx, ny = self._embed_patches(*xtuple)
        elif tokens:
            x, nx, ny = self._embed_tokens(*xtuple)
        else:
            x, nx, ny = self._embed(xtuple[0])
        B = x.shape[0]
        assert nx == self.nx and ny == self.ny, "Fix input size!"
        if (self.norm_embed is not None and self.cls_token is not None
                and tokens_layernorm_ok(x, self.cls_token, self.norm_embed)):
            # the LayerNorm writes behind the global-token rows of the stage's token tensor: no concatenation copy
            x = tokens_layernorm(x, self.cls_token, self.norm_embed)
        else:
            if self.norm_embed is not None:
                x = self.norm_embed(x)
            if self.cls_token is not None:
                x = torch.cat((expand_rows(self.cls_token, B).to(x.dtype), x), dim=1)
        if self.ape:
            grid = torch.cat([self.x_pos_embed.unsqueeze(2).expand(-1, -1, ny, -1),
                              self.y_pos_embed.unsqueeze(1).expand(-1, nx, -1, -1)], dim=-1).flatten(1, 2)
            x = x + torch.cat([self.cls_pos_embed, grid], dim=1)
        return self.pos_drop(x), nx, ny


class AttnBlock(nn.Module):
    def __init__(self, dim, num_heads, qkv_bias=False, qk_scale=None, drop=0., attn_drop=0., drop_path=0.,
                 norm_layer=nn.LayerNorm, attn_type='full', w=7, d=1, sharew=False, nglo=1, only_glo=False,
                 seq_len=None, num_feats=256, share_kv=False, sw_exact=0, rratio=2, rpe=False, wx=14, wy=14,
                 mode=0):
        super().__init__()
        self.norm = norm_layer(dim)
        if attn_type == 'full':
            self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale,
                                  attn_drop=attn_drop, proj_drop=drop, rpe=rpe, wx=wx, wy=wy, nglo=nglo)
        elif attn_type in ('longformerhand', 'longformerauto'):
            self.attn = Long2DSCSelfAttention(
                dim, exact=sw_exact, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale,
                attn_drop=attn_drop, proj_drop=drop, w=w, d=d, sharew=sharew, nglo=nglo, only_glo=only_glo,
                autograd=(attn_type == 'longformerauto'), rpe=rpe, mode=mode)
        else:
            raise ValueError("Not supported attention type {}".format(attn_type))
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()

    def forward(self, xtuple):
        x, nx, ny = xtuple
        p = self.drop_path.drop_prob if isinstance(self.drop_path, DropPath) else 0.0
        return residual_drop_path(x, self.attn(self.norm(x), nx, ny), p, self.training), nx, ny


class MlpBlock(nn.Module):
    def __init__(self, dim, out_dim=None, mlp_ratio=4., drop=0., drop_path=0., act_layer=nn.GELU,
                 norm_layer=nn.LayerNorm):
        super().__init__()
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), out_dim, act_layer=act_layer, drop=drop)
        self.shortcut = nn.Identity()
        if out_dim is not None and out_dim != dim:
            self.shortcut = nn.Sequential(VilLinear(dim, out_dim), nn.Dropout(drop))

    def forward(self, xtuple):
        x, nx, ny = xtuple
        p = self.drop_path.drop_prob if isinstance(self.drop_path, DropPath) else 0.0
        return residual_drop_path(self.shortcut(x), self.mlp(self.norm(x)), p, self.training), nx, ny


def parse_arch(arch):
    """'l1,h3,d96,n1,s1,g1,p4,f7,a0_l2,...' -> list of per-stage dicts
    (l stage id, h heads, d dim, n blocks, s sparse(1)/full(0), g global tokens,
    p patch size, f window, a absolute(1)/relative(0) position; msvit.py:402-410)."""
    stages = []
    for part in arch.split('_'):
        cfg = dict(l=1, h=3, d=192, n=1, s=1, g=1, p=2, f=7, a=1)
        for tok in part.split(','):
            cfg[tok[0]] = int(tok[1:])
        stages.append(cfg)
    return stages


class MsViT(nn.Module):
    def __init__(self, arch, img_size=512, in_chans=3, num_classes=1000, qkv_bias=True, qk_scale=None,
                 drop_rate=0., attn_drop_rate=0., drop_path_rate=0., norm_layer=partial(VilLayerNorm, eps=1e-6),
                 norm_embed=False, w=7, d=1, sharew=False, only_glo=False, share_kv=False,
                 attn_type='longformerhand', sw_exact=0, mode=0, **args):
        super().__init__()
        self.num_classes = num_classes
        self.drop_path_rate = drop_path_rate
        self.attn_type = attn_type
        self.layer_cfgs = parse_arch(arch)
        self.num_layers = len(self.layer_cfgs)
        if self.num_layers not in (3, 4):
            raise ValueError("Numer of layers {} not implemented yet!".format(self.num_layers))
        self.depth = sum(c['n'] for c in self.layer_cfgs)
        self.out_planes = self.layer_cfgs[-1]['d']
        self.Nglos = [c['g'] for c in self.layer_cfgs]
        self.avg_pool = args.get('avg_pool', False)
        self._dp_scales = None
        self._dp_rows = {}

        attn_args = dict(attn_type=attn_type, qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop_rate,
                         attn_drop=attn_drop_rate, w=w, d=d, sharew=sharew, only_glo=only_glo,
                         share_kv=share_kv, sw_exact=sw_exact, norm_layer=norm_layer, mode=mode)
        dprs = torch.linspace(0, drop_path_rate, self.depth).split([c['n'] for c in self.layer_cfgs])
        side = img_size
        in_dim = in_chans
        for i, cfg in enumerate(self.layer_cfgs):
            assert cfg['l'] == i + 1, "Error in _make_layer: layerid {} does not equal to layer_id {}".format(i + 1, cfg['l'])
            side //= cfg['p']
            attn_args.update(nglo=cfg['g'], num_feats=cfg['f'], rratio=cfg['f'], w=cfg['f'])
            if cfg['s'] == 0:
                attn_args['attn_type'] = 'full'          # sticky, as in the reference (msvit.py:460-461)
            blocks = [PatchEmbed(cfg['p'], side, side, in_chans=in_dim, embed_dim=cfg['d'], ape=bool(cfg['a']),
                                 nglo=cfg['g'], norm_layer=norm_layer, norm_embed=norm_embed, drop_rate=drop_rate)]
            for dpr in dprs[i]:
                blocks.append(AttnBlock(cfg['d'], cfg['h'], drop_path=float(dpr), seq_len=side * side + cfg['g'],
                                        rpe=not cfg['a'], wx=side, wy=side, **attn_args))
                blocks.append(MlpBlock(cfg['d'], drop_path=float(dpr), mlp_ratio=4.0, norm_layer=norm_layer,
                                       act_layer=nn.GELU, drop=drop_rate))
            setattr(self, "layer%d" % (i + 1), nn.Sequential(*blocks))
            in_dim = cfg['d']
        if self.num_layers == 3:
            self.layer4 = None
        self.norm = norm_layer(self.out_planes)
        self.head = VilLinear(self.out_planes, num_classes) if num_classes > 0 else nn.Identity()
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            _trunc_normal_(m.weight, .02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'pos_embed', 'cls_token', 'norm.weight', 'norm.bias', 'norm_embed', 'head.bias', 'relative_position'}

    def get_classifier(self):
        return self.head

    def _drop_scale(self, blk, B, device):
        """per-sample stochastic-depth scale of one block, {0, 1/keep}: a row of ONE (blocks, B) draw made at the top
        of the forward (three kernels per step instead of two per block: 44 tiny launches less in ViL-Small)"""
        p = blk.drop_path.drop_prob if isinstance(blk.drop_path, DropPath) else 0.0
        if p == 0.0 or not blk.training:
            return None
        tab = self._dp_scales
        row = self._dp_rows.get(id(blk)) if tab is not None else None
        if row is None or tab.shape[1] != B or tab.device != device:
            return torch.empty(B, dtype=torch.float32, device=device).bernoulli_(1.0 - p).div_(1.0 - p)
        return tab[row]

    def _draw_drop_scales(self, B, device):
        if not self.training or self.drop_path_rate <= 0.0 or device.type != "cuda":
            self._dp_scales = None
            return
        # rows are keyed by (stage, block index) and the cache by the tuple of drop probabilities, so a stochastic-depth
        # schedule that changes drop_prob (or a deepcopy of the model) rebuilds it
        blks, keys = [], []
        for i in range(self.num_layers):
            for j, blk in enumerate(list(getattr(self, "layer%d" % (i + 1)))[1:]):
                if isinstance(getattr(blk, "drop_path", None), DropPath) and blk.drop_path.drop_prob > 0.0:
                    blks.append(blk); keys.append((i, j, float(blk.drop_path.drop_prob)))
        sig = (tuple(keys), device)
        if getattr(self, "_dp_sig", None) != sig:
            self._dp_sig = sig
            self._dp_rows = {id(blk): r for r, blk in enumerate(blks)}
            self._dp_keep = (1.0 - torch.tensor([k[2] for k in keys], dtype=torch.float32, device=device)).view(-1, 1)
        keep = self._dp_keep
        if keep.numel() == 0:
            self._dp_scales = None
            return
        u = torch.rand(keep.shape[0], B, device=device)
        self._dp_scales = (u < keep).to(torch.float32) / keep

    @staticmethod
    def _fused_transition_ok(x, pend, embed, nx, ny):
        if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 3):
            return False
        ph, pw = embed.patch_size
        dt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else x.dtype
        if dt not in (torch.float32, torch.bfloat16) or not embed.tokens_ok(x, nx, ny) or x.shape[-1] % 8:
            return False
        if pend is not None and (pend[0].shape != x.shape or pend[0].dtype not in (torch.float32, torch.bfloat16)):
            return False
        return True

    @staticmethod
    def _settle(x, pend):
        """materialise a deferred `x + drop_path(branch)`"""
        if pend is None:
            return x
        br, sc = pend
        return x + br if sc is None else _DropPathAdd.apply(x, br, sc.view(-1, 1, 1).to(x.dtype))

    def _run_stage(self, layer, xtuple, tokens=False):
        """One stage with the residual add of every block DEFERRED into the next block's LayerNorm
        (vil_resln_*: one kernel each way instead of add + norm, and norm-backward + add + mask-mul + cast).
        Same arithmetic as `layer(xtuple)` (msvit.py:313-316,336-340); returns the last add still pending."""
        x, nx, ny = layer[0](xtuple, tokens=tokens)
        B = x.shape[0]
        pend = None
        for blk in list(layer)[1:]:
            attn = isinstance(blk, AttnBlock)
            if not (attn or (isinstance(blk, MlpBlock) and isinstance(blk.shortcut, nn.Identity))):
                x, nx, ny = blk((self._settle(x, pend), nx, ny))
                pend = None
                continue
            if pend is not None and res_layernorm_ok(x, pend[0], blk.norm):
                x, y = res_layernorm(x, pend[0], pend[1], blk.norm)
            elif pend is None and (x.requires_grad or not torch.is_grad_enabled()) and pass_layernorm_ok(x, blk.norm):
                x, y = pass_layernorm(x, blk.norm)      # first block of a stage: x feeds the norm AND the residual stream
            else:
                x = self._settle(x, pend)
                y = blk.norm(x)
            br = blk.attn(y, nx, ny) if attn else blk.mlp(y)
            pend = (br, self._drop_scale(blk, B, x.device))
        return x, pend, nx, ny

    def forward_features(self, x):
        B = x.shape[0]
        nx = ny = None
        pend = None
        self._draw_drop_scales(B, x.device)
        for i in range(self.num_layers):
            layer = getattr(self, "layer%d" % (i + 1))
            tokens = False
            if i > 0 and self._fused_transition_ok(x, pend, layer[0], nx, ny):
                # one row gather: pending residual add + drop the global tokens + (py, px, c) patch vectors + cast
                ph, pw = layer[0].patch_size
                dt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else x.dtype
                br, sc = pend if pend is not None else (None, None)
                x = _Patchify.apply(x, br, sc, self.Nglos[i - 1], nx, ny, ph, pw, dt)
                if getattr(self, "_seg_points", None) is not None:
                    self._seg_points[i] = x          # activation entering stage i: a backward-segment cut (engine.GraphedTrainStep)
                tokens = "patches"
            elif i > 0:   # drop the previous stage's global tokens, back to an image
                x = self._settle(x, pend)
                if getattr(self, "_seg_points", None) is not None:
                    self._seg_points[i] = x          # activation entering stage i: a backward-segment cut (engine.GraphedTrainStep)
                x = x[:, self.Nglos[i - 1]:]
                tokens = layer[0].tokens_ok(x, nx, ny)
                if not tokens:
                    x = x.transpose(-2, -1).reshape(B, -1, nx, ny)
            if x.is_cuda:
                x, pend, nx, ny = self._run_stage(layer, (x, nx, ny), tokens)
            else:
                x, nx, ny = layer((x, nx, ny))
                pend = None
        if pend is not None and res_layernorm_ok(x, pend[0], self.norm):
            _, x = res_layernorm(x, pend[0], pend[1], self.norm)
        else:
            x = self.norm(self._settle(x, pend))
        if self.Nglos[-1] > 0 and not self.avg_pool:
            return x[:, 0]
        return torch.mean(x, dim=1)

    def reset_vil_mode(self, mode):
        """Switch every sliding-chunk layer between random-shift and full mode
        (msvit.py:532-541)."""
        for m in self.modules():
            if isinstance(m, Long2DSCSelfAttention) and m.mode != mode:
                m.mode = mode

    def forward(self, x):
        return self.head(self.forward_features(x))


# arch strings of the published ViL models (reference README.md:63-67), relative-position (a0) variants
def vil_arch(name, f1=7, f2=7):
    n = {"tiny": (1, 1, 9, 1), "small": (1, 2, 8, 1), "medium_deep": (1, 4, 16, 1), "base_deep": (1, 8, 24, 1)}[name]
    dims = {"tiny": ((1, 48), (3, 96), (3, 192), (6, 384)), "small": ((3, 96), (3, 192), (6, 384), (12, 768)),
            "medium_deep": ((3, 96), (3, 192), (6, 384), (12, 768)),
            "base_deep": ((3, 96), (3, 192), (6, 384), (12, 768))}[name]
    ps = (4, 2, 2, 2)
    ss = (1, 1, 0, 0)
    gs = (1, 1, 1, 0)
    fs = (f1, f2, 7, 7)
    return "_".join("l%d,h%d,d%d,n%d,s%d,g%d,p%d,f%d,a0" % (i + 1, dims[i][0], dims[i][1], n[i], ss[i], gs[i], ps[i], fs[i])
                    for i in range(4))
