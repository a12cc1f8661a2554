"""CPU oracle for Vision Longformer's 2-D sliding-chunk local+global attention.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it, and only as the checker / the timed CPU
baseline.  The product path (``vision-longformer_amd/``) never imports it and
fails loudly when the HIP library is missing.

This file is the build's own restatement (plain PyTorch CPU ops, fp32/fp64) of
the algorithm of the reference's ``longformerhand`` path.  It is *pinned*: the
script ``tools/gen_golden.py`` imports the real reference from
``/root/reference/src`` in the build container, checks every function below
against it (operator level, mask level, module level, fwd + grads) and freezes
golden vectors under ``tests/golden/``; ``tests/test_oracle_golden.py`` re-checks
the oracle against those vectors wherever the tests run.

Reference functions restated here (paths relative to /root/reference):
  * relative_position_index          src/models/layers/longformer2d.py:67-100
  * invalid-location masks           src/models/layers/slidingchunk_2d.py:249-318
  * mask_invalid_locations           src/models/layers/slidingchunk_2d.py:321-357
  * slidingchunk_qk / _av / _agrad   src/models/layers/slidingchunk_2d.py:26-200
  * Long2DSCSelfAttention.forward    src/models/layers/longformer2d.py:106-229

Index conventions: tokens are [G global | nx*ny local],
local token i = r*ny + c; zero padding is appended at the bottom/right; chunk
(m, n) holds rows [mW, (m+1)W) x cols [nW, (n+1)W); in-chunk index l = x*W + y;
key slot = nb*W^2 + t with nb = (dr+1)*3 + (dc+1).
"""
import math
from functools import lru_cache

import torch
import torch.nn.functional as F

# neighbour order of the 3x3 chunk neighbourhood (slidingchunk_2d.py:37-66)
NEIGHBOURS = [(-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 0), (0, 1), (1, -1), (1, 0), (1, 1)]
# random-shift mode m -> neighbour offset (slidingchunk_2d.py:15-24; the dict
# there stores roll shifts, the neighbour is minus the shift)
MODE_NEIGHBOUR = {1: (-1, -1), 2: (-1, 0), 3: (-1, 1), 4: (0, -1),
                  5: (0, 1), 6: (1, -1), 7: (1, 0), 8: (1, 1)}


def active_neighbours(mode):
    """Neighbour offsets whose keys a query chunk sees, in key-slot order
    (slidingchunk_2d.py:37-79): mode 0 -> all 9; -1 -> self; i>0 -> [self, nb_i]."""
    if mode == 0:
        return list(NEIGHBOURS)
    if mode == -1:
        return [(0, 0)]
    if 1 <= mode <= 8:
        return [(0, 0), MODE_NEIGHBOUR[mode]]
    raise ValueError("mode must be in [-1, 8]")


def neighbour_slot(offset):
    """Index 0..8 of a neighbour offset inside the 9-chunk key axis."""
    return (offset[0] + 1) * 3 + (offset[1] + 1)


def pad_amounts(nx, ny, W):
    """(padx, pady, mx, my) as in longformer2d.py:141-143."""
    padx = (W - nx % W) % W
    pady = (W - ny % W) % W
    return padx, pady, (nx + padx) // W, (ny + pady) // W


@lru_cache(maxsize=None)
def relative_position_index(W):
    """(W^2, 9W^2) int64 index into the ((4W-1)^2, H) bias table.

    idx[l, nb*W^2+t] = (xl - (dr*W+xt) + 2W-1) * (4W-1) + (yl - (dc*W+yt) + 2W-1)
    (longformer2d.py:67-100)."""
    W2 = W * W
    l = torch.arange(W2)
    xl, yl = l // W, l % W
    cols = []
    for (dr, dc) in NEIGHBOURS:
        xt = dr * W + xl  # key coordinates in the centre-chunk frame
        yt = dc * W + yl
        relx = xl[:, None] - xt[None, :] + 2 * W - 1
        rely = yl[:, None] - yt[None, :] + 2 * W - 1
        cols.append(relx * (4 * W - 1) + rely)
    return torch.cat(cols, dim=1)


def mode_columns(W, mode):
    """Columns of the 9W^2 key axis kept for a mode (longformer2d.py:164-173,
    slidingchunk_2d.py:344-353)."""
    W2 = W * W
    cols = []
    for off in active_neighbours(mode):
        s = neighbour_slot(off)
        cols.append(torch.arange(s * W2, (s + 1) * W2))
    return torch.cat(cols)


# --------------------------------------------------------------------------
# masks (vectorised closed forms of the reference's Python list comprehensions)
# --------------------------------------------------------------------------
def _slot_geometry(mx, my, W):
    W2 = W * W
    i = torch.arange(mx * my)
    ci, cj = i // my, i % my                      # chunk row / col
    j = torch.arange(9 * W2)
    nbr, nbc = (j // W2) // 3, (j // W2) % 3      # 0..2
    xt, yt = (j % W2) // W, (j % W2) % W
    return ci[:, None], cj[:, None], nbr[None, :], nbc[None, :], xt[None, :], yt[None, :]


def invalid_mask_zero(mx, my, padx, pady, W):
    """(mx*my, 9W^2) bool; slidingchunk_2d.py:270-290."""
    ci, cj, nbr, nbc, xt, yt = _slot_geometry(mx, my, W)
    r = ci + nbr - 1
    c = cj + nbc - 1
    return ((r < 0) | (r >= mx) | (r * W + xt >= mx * W - padx) |
            (c < 0) | (c >= my) | (c * W + yt >= my * W - pady))


def invalid_mask_cyclic(mx, my, padx, pady, W):
    """(mx*my, 9W^2) bool; slidingchunk_2d.py:249-267."""
    ci, cj, nbr, nbc, xt, yt = _slot_geometry(mx, my, W)
    return (((ci + nbr == mx) & ((mx - 1) * W + xt >= mx * W - padx)) |
            ((cj + nbc == my) & ((my - 1) * W + yt >= my * W - pady)))


def invalid_mask_exact(mx, my, padx, pady, W):
    """(mx*my, W^2, 9W^2) bool; slidingchunk_2d.py:293-318."""
    W2 = W * W
    ci, cj, nbr, nbc, xt, yt = _slot_geometry(mx, my, W)
    kr = ((ci + nbr - 1) * W + xt)[:, None, :]            # key abs row
    kc = ((cj + nbc - 1) * W + yt)[:, None, :]
    l = torch.arange(W2)
    ql, qc = (l // W)[None, :, None], (l % W)[None, :, None]
    ci3, cj3 = ci[:, :, None], cj[:, :, None]
    nx_max = mx * W - 1 - padx
    ny_max = my * W - 1 - pady
    lo_r = torch.clamp((ci3 - 1) * W + ql, min=0)
    hi_r = torch.clamp((ci3 + 1) * W + ql, max=nx_max)
    lo_c = torch.clamp((cj3 - 1) * W + qc, min=0)
    hi_c = torch.clamp((cj3 + 1) * W + qc, max=ny_max)
    return (kr < lo_r) | (kr > hi_r) | (kc < lo_c) | (kc > hi_c)


def invalid_mask(mx, my, padx, pady, W, exact, mode=0):
    """Mask + num_invalid exactly as mask_invalid_locations picks/slices them
    (slidingchunk_2d.py:321-357).  Returns (mask, num_invalid) with mask shaped
    (mx*my, kv) for exact in {0,-1} and (mx*my, W^2, 9W^2) for exact == 1."""
    W2 = W * W
    if exact == 1 and mode == 0:
        m = invalid_mask_exact(mx, my, padx, pady, W)
        return m, int(m.sum())
    if exact == 0:
        m = invalid_mask_zero(mx, my, padx, pady, W)
    elif exact == -1:
        m = invalid_mask_cyclic(mx, my, padx, pady, W)
    else:
        raise ValueError("longsc exact should be in [0,1,-1]!")
    if mode != 0:
        m = m[:, mode_columns(W, mode)]
    return m, int(W2 * m.sum())


def mask_invalid_locations(input_tensor, mx, my, padx, pady, W, exact, mode=0):
    """In-place -inf fill with the reference's signature (slidingchunk_2d.py:321)."""
    m, num_invalid = invalid_mask(mx, my, padx, pady, W, exact, mode)
    if m.dim() == 3:
        m = m.view(1, mx, my, W * W, -1)
    else:
        m = m.view(1, mx, my, 1, -1)
    input_tensor.masked_fill_(m.to(input_tensor.device).expand(input_tensor.size()), -float('inf'))
    return num_invalid


# --------------------------------------------------------------------------
# operator level: the three sliding-chunk products, by gather instead of roll
# --------------------------------------------------------------------------
def _neighbour_chunks(t_img, offsets):
    """t_img (BH, M, mx, my, W2) -> (BH, M, mx, my, len(offsets)*W2) holding, for
    every chunk, the chunks at the cyclic neighbour offsets (torch.roll semantics
    of slidingchunk_2d.py:37-66: neighbour (dr,dc) of chunk (m,n) is chunk
    ((m+dr) mod mx, (n+dc) mod my))."""
    BH, M, mx, my, W2 = t_img.shape
    mi = torch.arange(mx)
    ni = torch.arange(my)
    outs = []
    for (dr, dc) in offsets:
        rows = (mi + dr) % mx
        cols = (ni + dc) % my
        outs.append(t_img[:, :, rows][:, :, :, cols])
    return torch.cat(outs, dim=-1)


def slidingchunk_qk(q_img, k_img, mode=0):
    """attn[b,m,n,l,nb*W2+t] = sum_c q[b,c,m,n,l] k[b,c,(m+dr)%mx,(n+dc)%my,t]
    (slidingchunk_2d.py:26-79)."""
    kn = _neighbour_chunks(k_img, active_neighbours(mode))
    return torch.einsum('bcmnl,bcmnt->bmnlt', q_img, kn)


def slidingchunk_av(attn, v_img, mode=0):
    """out[b,c,m,n,l] = sum_{nb,t} attn[b,m,n,l,nb*W2+t] v[b,c,(m+dr),(n+dc),t]
    (slidingchunk_2d.py:82-130)."""
    vn = _neighbour_chunks(v_img, active_neighbours(mode))
    return torch.einsum('bmnlt,bcmnt->bcmnl', attn, vn)


def slidingchunk_agrad(attn, grad_x, mode=0):
    """grad_t2[b,c,m',n',t] = sum over (m,n,nb) with (m+dr,n+dc)=(m',n') of
    sum_l attn[b,m,n,l,nb*W2+t] g[b,c,m,n,l]  (slidingchunk_2d.py:132-200)."""
    BH, M, mx, my, W2 = grad_x.shape
    out = torch.zeros_like(grad_x)
    mi = torch.arange(mx)
    ni = torch.arange(my)
    for s, (dr, dc) in enumerate(active_neighbours(mode)):
        part = torch.einsum('bmnlt,bcmnl->bcmnt', attn[..., s * W2:(s + 1) * W2], grad_x)
        rows = (mi + dr) % mx
        cols = (ni + dc) % my
        # scatter chunk (m,n)'s contribution home to chunk (rows[m], cols[n])
        idx = (rows[:, None] * my + cols[None, :]).reshape(-1)
        flat = out.view(BH, M, mx * my, W2)
        flat.index_add_(2, idx, part.reshape(BH, M, mx * my, W2))
    return out


class _SlidingChunk2D(torch.autograd.Function):
    """Hand-written backward with the reference's structure
    (slidingchunk_2d.py:202-246)."""

    @staticmethod
    def forward(ctx, t1, t2, is_t1_diagonaled=False, mode=0):
        ctx.save_for_backward(t1, t2)
        ctx.is_t1_diagonaled = is_t1_diagonaled
        ctx.mode = mode
        if is_t1_diagonaled:
            return slidingchunk_av(t1, t2, mode)
        return slidingchunk_qk(t1, t2, mode)

    @staticmethod
    def backward(ctx, grad_output):
        t1, t2 = ctx.saved_tensors
        mode = ctx.mode
        if ctx.is_t1_diagonaled:
            g1 = slidingchunk_qk(grad_output, t2, mode)
            g2 = slidingchunk_agrad(t1, grad_output, mode)
        else:
            g1 = slidingchunk_av(grad_output, t2, mode)
            g2 = slidingchunk_agrad(grad_output, t1, mode)
        return g1, g2, None, None


def slidingchunk_2d(t1, t2, is_t1_diagonaled=False, mode=0):
    return _SlidingChunk2D.apply(t1, t2, is_t1_diagonaled, mode)


# --------------------------------------------------------------------------
# fused local attention (what the HIP kernels implement): chunked form
# --------------------------------------------------------------------------
def chunk_tokens(t, nx, ny, W):
    """(B, H, nx*ny, M) -> (B*H, M, mx, my, W2), zero padded bottom/right
    (longformer2d.py:134-149)."""
    B, H, Nloc, M = t.shape
    padx, pady, mx, my = pad_amounts(nx, ny, W)
    img = t.reshape(B * H, nx, ny, M).permute(0, 3, 1, 2)
    if padx or pady:
        img = F.pad(img, (0, pady, 0, padx))
    img = img.reshape(B * H, M, mx, W, my, W).permute(0, 1, 2, 4, 3, 5)
    return img.reshape(B * H, M, mx, my, W * W)


def unchunk_tokens(t_img, B, H, nx, ny, W):
    """(B*H, M, mx, my, W2) -> (B, H, nx*ny, M), cropping the padding
    (longformer2d.py:201-202)."""
    BH, M, mx, my, W2 = t_img.shape
    img = t_img.reshape(BH, M, mx, my, W, W).permute(0, 2, 4, 3, 5, 1)
    img = img.reshape(BH, mx * W, my * W, M)[:, :nx, :ny]
    return img.reshape(B, H, nx * ny, M)


def local_attention(q, k, v, nx, ny, W, G, mode=0, exact=0, scale=None,
                    bias_table=None, g2l_bias=None, only_glo=False, return_lse=False):
    """Local-query rows of Long2DSCSelfAttention (longformer2d.py:126-204, minus
    the Linear layers).

    q: (B, H, Nloc, M) *unscaled* local queries; k, v: (B, H, G+Nloc, M) with
    the G global tokens first.  bias_table: ((4W-1)^2, H) or None (rpe off);
    g2l_bias: (H, G) = g2l_relative_position_bias[1] or None.
    Returns out (B, H, Nloc, M) [and lse (B, H, Nloc), natural log]."""
    B, H, Nloc, M = q.shape
    assert Nloc == nx * ny and k.shape[2] == G + Nloc
    if scale is None:
        scale = M ** -0.5
    W2 = W * W
    padx, pady, mx, my = pad_amounts(nx, ny, W)
    qs = q * scale
    kg, vg = k[:, :, :G], v[:, :, :G]
    if only_glo:
        s = torch.einsum('bhlc,bhgc->bhlg', qs, kg)
        if g2l_bias is not None:
            s = s + g2l_bias[None, :, None, :]
        lse = torch.logsumexp(s, dim=-1)
        p = torch.softmax(s - s.max(dim=-1, keepdim=True)[0], dim=-1)
        out = torch.einsum('bhlg,bhgc->bhlc', p, vg)
        return (out, lse) if return_lse else out
    if exact == 1 and mode != 0:
        raise ValueError("longsc exact should be in [0,1,-1]!")
    q_img = chunk_tokens(qs, nx, ny, W)
    k_img = chunk_tokens(k[:, :, G:], nx, ny, W)
    v_img = chunk_tokens(v[:, :, G:], nx, ny, W)
    attn11 = slidingchunk_qk(q_img, k_img, mode)                    # (BH,mx,my,W2,kv)
    kv = attn11.shape[-1]
    if bias_table is not None:
        idx = relative_position_index(W)[:, mode_columns(W, mode)]     # (W2, kv)
        bias = bias_table[idx.reshape(-1)].view(W2, kv, H).permute(2, 0, 1)  # (H,W2,kv)
        attn11 = attn11.view(B, H, mx, my, W2, kv) + bias[None, :, None, None]
        attn11 = attn11.reshape(B * H, mx, my, W2, kv)
    m, _ = invalid_mask(mx, my, padx, pady, W, exact, mode)
    m = m.view(1, mx, my, W2, kv) if m.dim() == 3 else m.view(1, mx, my, 1, kv)
    attn11 = attn11.masked_fill(m, -float('inf'))
    if G > 0:
        attn10 = torch.einsum('bcmnl,bgc->bmnlg', q_img, kg.reshape(B * H, G, M))
        if g2l_bias is not None:
            attn10 = attn10 + g2l_bias.repeat(B, 1).view(B * H, 1, 1, 1, G)
        attn1 = torch.cat((attn10, attn11), dim=-1)
    else:
        attn1 = attn11
    lse = torch.logsumexp(attn1, dim=-1)                            # (BH,mx,my,W2)
    p = torch.softmax(attn1 - attn1.max(dim=-1, keepdim=True)[0], dim=-1)
    x1 = slidingchunk_av(p[..., G:], v_img, mode)
    if G > 0:
        x1 = x1 + torch.einsum('bmnlg,bgc->bcmnl', p[..., :G], vg.reshape(B * H, G, M))
    out = unchunk_tokens(x1, B, H, nx, ny, W)
    if return_lse:
        lse = unchunk_tokens(lse.unsqueeze(1), B, H, nx, ny, W).squeeze(-1)
        return out, lse
    return out


# --------------------------------------------------------------------------
# dense closed form -- O(N^2), small shapes only.  An
# independent statement of the same function, used to cross-check the chunked
# form for exact in {0, 1}.
# --------------------------------------------------------------------------
def local_attention_dense(q, k, v, nx, ny, W, G, mode=0, exact=0, scale=None,
                          bias_table=None, g2l_bias=None):
    B, H, Nloc, M = q.shape
    if scale is None:
        scale = M ** -0.5
    if exact not in (0, 1):
        raise ValueError("dense closed form covers exact in {0, 1}")
    i = torch.arange(Nloc)
    r, c = i // ny, i % ny
    cr, cc = r // W, c // W
    dr = cr[None, :] - cr[:, None]          # key chunk row - query chunk row
    dc = cc[None, :] - cc[:, None]
    if exact == 1:
        if mode != 0:
            raise ValueError("longsc exact should be in [0,1,-1]!")
        allowed = ((r[None, :] - r[:, None]).abs() <= W) & ((c[None, :] - c[:, None]).abs() <= W)
    else:
        allowed = torch.zeros(Nloc, Nloc, dtype=torch.bool)
        for (a, b) in active_neighbours(mode):
            allowed |= (dr == a) & (dc == b)
    s = torch.einsum('bhic,bhjc->bhij', q * scale, k)                 # (B,H,Nloc,N)
    if bias_table is not None:
        relx = r[:, None] - r[None, :] + 2 * W - 1
        rely = c[:, None] - c[None, :] + 2 * W - 1
        idx = (relx * (4 * W - 1) + rely).clamp(0, (4 * W - 1) ** 2 - 1)
        lb = bias_table[idx.reshape(-1)].view(Nloc, Nloc, H).permute(2, 0, 1)
        s[..., G:] = s[..., G:] + lb[None]
        if G > 0 and g2l_bias is not None:
            s[..., :G] = s[..., :G] + g2l_bias[None, :, None, :]
    full_allowed = torch.cat([torch.ones(Nloc, G, dtype=torch.bool), allowed], dim=1)
    s = s.masked_fill(~full_allowed, -float('inf'))
    return torch.einsum('bhij,bhjc->bhic', torch.softmax(s, dim=-1), v)


# --------------------------------------------------------------------------
# module level: functional restatement of Long2DSCSelfAttention.forward
# --------------------------------------------------------------------------
def long2dsc_forward(params, x, nx, ny, *, num_heads, w, nglo=1, rpe=False,
                     exact=0, mode=0, only_glo=False, qk_scale=None):
    """params: state-dict style mapping with the reference's key names
    (query.weight, kv.bias, query_global.*, local_relative_position_bias_table,
    g2l_relative_position_bias, g2g_relative_position_bias, ...).  `mode` is
    the *resolved* mode for this call (the random draw of longformer2d.py:114-123
    is made by the caller)."""
    B, N, C = x.shape
    H, G, W = num_heads, nglo, w
    M = C // H
    Nloc = nx * ny
    assert G + Nloc == N, "Global dimension does not match!"
    scale = qk_scale or M ** -0.5

    def lin(name, t):
        return F.linear(t, params[name + '.weight'], params.get(name + '.bias'))

    q = lin('query', x[:, G:]).reshape(B, Nloc, H, M).transpose(1, 2)
    kv = lin('kv', x).reshape(B, N, 2, H, M).permute(2, 0, 3, 1, 4)
    k, v = kv[0], kv[1]
    table = params['local_relative_position_bias_table'] if rpe else None
    g2l1 = params['g2l_relative_position_bias'][1] if (rpe and G > 0) else None
    x1 = local_attention(q, k, v, nx, ny, W, G, mode=mode, exact=exact, scale=scale,
                         bias_table=table, g2l_bias=g2l1, only_glo=only_glo)
    x1 = lin('proj', x1.transpose(1, 2).reshape(B, Nloc, C))
    if G == 0:
        return x1
    # global-token rows: vanilla attention over all N keys (longformer2d.py:210-227)
    qg = scale * lin('query_global', x[:, :G]).reshape(B, G, H, M).transpose(1, 2)
    kvg = lin('kv_global', x).reshape(B, N, 2, H, M).permute(2, 0, 3, 1, 4)
    kg, vg = kvg[0], kvg[1]
    a0 = torch.einsum('bhgc,bhnc->bhgn', qg, kg)
    if rpe:
        gb = torch.cat([params['g2g_relative_position_bias'],
                        params['g2l_relative_position_bias'][0].unsqueeze(-1).expand(-1, -1, Nloc)],
                       dim=-1)
        a0 = a0 + gb[None]
    a0 = torch.softmax(a0 - a0.max(dim=-1, keepdim=True)[0], dim=-1)
    x0 = torch.einsum('bhgn,bhnc->bhgc', a0, vg).transpose(1, 2).reshape(B, G, C)
    x0 = lin('proj_global', x0)
    return torch.cat((x0, x1), dim=1)


# --------------------------------------------------------------------------
# the dense `Attention` of the s0 stages (reference src/models/msvit.py:37-120): full attention over
# [G global | nx*ny local] tokens with the Swin-style relative position bias of a (2nx-1) x (2ny-1) table,
# the g2l / g2g biases of the global tokens.  Pinned by tools/gen_golden.py against the reference module
# (tests/golden/dense_cases.npz).
# --------------------------------------------------------------------------
def dense_relative_position_index(nx, ny):
    """msvit.py:71-82: index into the ((2nx-1)(2ny-1), H) table for every (query, key) pair of the grid"""
    ix, iy = torch.meshgrid(torch.arange(nx), torch.arange(ny), indexing="ij")
    ix, iy = ix.reshape(-1), iy.reshape(-1)
    return (ix[:, None] - ix[None, :] + nx - 1) * (2 * ny - 1) + (iy[:, None] - iy[None, :] + ny - 1)


def dense_attention(qkv, table, g2l, g2g, nx, ny, G, H, scale=None):
    """qkv (B, N, 3C) packed projection -> (B, N, C); msvit.py:84-120 (softmax over all N keys)."""
    B, N, C3 = qkv.shape
    C = C3 // 3
    M = C // H
    if scale is None:
        scale = M ** -0.5
    q, k, v = qkv.view(B, N, 3, H, M).permute(2, 0, 3, 1, 4)
    attn = (q @ k.transpose(-2, -1)) * scale
    if table is not None:
        L = nx * ny
        assert N == G + L, "For relative position, N != self.nglo + self.wx*self.wy!"
        loc = table[dense_relative_position_index(nx, ny).reshape(-1)].view(L, L, H).permute(2, 0, 1)
        if G > 0:
            top = torch.cat([g2g, g2l[0].unsqueeze(-1).expand(-1, -1, L)], dim=-1)
            bot = torch.cat([g2l[1].unsqueeze(1).expand(-1, L, -1), loc], dim=-1)
            bias = torch.cat([top, bot], dim=1)
        else:
            bias = loc
        attn = attn + bias.unsqueeze(0)
    attn = attn.softmax(dim=-1)
    return (attn @ v).transpose(1, 2).reshape(B, N, C)


def dense_module_forward(params, x, nx, ny, *, num_heads, nglo=1, rpe=True, qk_scale=None):
    """functional restatement of Attention.forward: qkv Linear -> dense_attention -> proj Linear"""
    C = x.shape[-1]
    qkv = F.linear(x, params["qkv.weight"], params.get("qkv.bias"))
    out = dense_attention(qkv, params["local_relative_position_bias_table"] if rpe else None,
                          params.get("g2l_relative_position_bias") if (rpe and nglo > 0) else None,
                          params.get("g2g_relative_position_bias") if (rpe and nglo > 0) else None,
                          nx, ny, nglo, num_heads, qk_scale or (C // num_heads) ** -0.5)
    return F.linear(out, params["proj.weight"], params.get("proj.bias"))
