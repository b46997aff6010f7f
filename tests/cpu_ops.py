"""CPU stand-in for frameino_amd.ops, used ONLY by the gloo multi-process tests to exercise the sharding /
collective logic of frameino_amd/parallel.py where no GPU exists.  Each function restates the operator contract
of include/frameino_hip.h in plain torch (same semantics as the references in tests/test_kernels_gpu.py).
It is test infrastructure: the product never imports it."""
import math

import torch
import torch.nn.functional as F

EPI_NONE, EPI_GELU_TANH, EPI_RESIDUAL, EPI_GATED_RESIDUAL = 0, 1, 2, 3


def _g(table, sel):
    if table is None:
        return None
    if table.dim() == 1:
        return table
    return table[sel.long()] if sel is not None else table[0]


def adaln_modulate(x, shift, scale, sel=None, eps=1e-6, out=None):
    n = F.layer_norm(x.float(), (x.shape[-1],), None, None, eps)
    y = (n * (1 + _g(scale, sel)) + _g(shift, sel)).to(x.dtype)
    return y if out is None else out.copy_(y)


def layernorm(x, weight=None, bias=None, eps=1e-5, out=None):
    y = F.layer_norm(x.float(), (x.shape[-1],), weight, bias, eps).to(x.dtype)
    return y if out is None else out.copy_(y)


def gated_residual(x, y, gate=None, sel=None, out=None):
    r = (x.float() + y.float() * _g(gate, sel)).to(x.dtype) if gate is not None else x + y
    return r if out is None else out.copy_(r)


SCALE_FOLDED = -1.0
LOG2E = 1.4426950408889634


def rmsnorm_rope_(x, weight, eps, cos=None, sin=None, head_dim=0, out_scale=1.0):
    y = x
    if weight is not None:
        var = x.float().pow(2).mean(-1, keepdim=True)
        y = x * torch.rsqrt(var + eps)
        if weight.dtype in (torch.float16, torch.bfloat16):
            y = y.to(weight.dtype)
        y = y * weight
    if cos is not None:
        rows, dim = y.shape
        yh = y.reshape(rows, dim // head_dim, head_dim // 2, 2)
        x1, x2 = yh[..., 0], yh[..., 1]
        c, s = cos[:, None, :], sin[:, None, :]
        o = torch.empty_like(yh)
        o[..., 0] = x1 * c - x2 * s
        o[..., 1] = x1 * s + x2 * c
        y = o.reshape(rows, dim)
    x.copy_((y * out_scale).to(x.dtype) if out_scale != 1.0 else y.to(x.dtype))
    return x


def qkv_rmsnorm_rope_(qkv, dim, q_weight, q_eps, k_weight, k_eps, cos, sin, head_dim, q_out_scale=1.0, out=None,
                      head_off=None, head_ld=None):
    heads = dim // head_dim
    if out is None:
        rmsnorm_rope_(qkv[:, :dim], q_weight, q_eps, cos, sin, head_dim, q_out_scale)
        rmsnorm_rope_(qkv[:, dim:2 * dim], k_weight, k_eps, cos, sin, head_dim)
        return qkv
    rmsnorm_rope_scatter(qkv[:, :dim], q_weight, q_eps, cos, sin, head_dim, out, head_off[:heads], head_ld, q_out_scale)
    rmsnorm_rope_scatter(qkv[:, dim:2 * dim], k_weight, k_eps, cos, sin, head_dim, out, head_off[heads:2 * heads], head_ld)
    rmsnorm_rope_scatter(qkv[:, 2 * dim:], None, 0.0, None, None, head_dim, out, head_off[2 * heads:], head_ld)
    return out


def rmsnorm_rope_scatter(x, weight, eps, cos, sin, head_dim, out, head_off, head_ld, out_scale=1.0):
    y = rmsnorm_rope_(x.clone(), weight, eps, cos, sin, head_dim, out_scale)
    rows = y.shape[0]
    flat = out.view(-1)
    r = torch.arange(rows, device=y.device)[:, None]
    c = torch.arange(head_dim, device=y.device)[None, :]
    for hd in range(y.shape[1] // head_dim):
        flat[int(head_off[hd]) + r * int(head_ld[hd]) + c] = y[:, hd * head_dim:(hd + 1) * head_dim]
    return out


def attention(q, k, v, heads, out=None, scale=None):
    b, lq, hd = q.shape
    dh = hd // heads
    f = lambda t: t.reshape(b, -1, heads, dh).transpose(1, 2)      # noqa: E731
    # SCALE_FOLDED: q carries softmax_scale * log2(e); softmax(ln2 * q.k) = 2^(q.k) normalised
    sc = math.log(2.0) if scale == SCALE_FOLDED else scale
    o = F.scaled_dot_product_attention(f(q), f(k), f(v), scale=sc).transpose(1, 2).reshape(b, lq, hd).to(q.dtype)
    return o if out is None else out.copy_(o)


def attention_partial(q, k, v, heads, out=None, scale=None):
    """(unnormalised O, row max m, row sum l) of attention over one key range (contract of fino_attn_partial)"""
    b, lq, hd = q.shape
    dh = hd // heads
    f = lambda t: t.reshape(b, -1, heads, dh).transpose(1, 2).float()      # noqa: E731
    s_ = f(q) @ f(k).transpose(2, 3) * (dh ** -0.5 if scale is None else math.log(2.0) if scale == SCALE_FOLDED else scale)
    m = s_.amax(dim=-1, keepdim=True)
    p_ = torch.exp(s_ - m)
    return p_ @ f(v), m, p_.sum(dim=-1, keepdim=True)


def attention_merge(parts, batch, lq, heads, head_dim, dtype, out=None):
    m = torch.stack([p[1] for p in parts]).amax(dim=0)
    num = sum(p[0] * torch.exp(p[1] - m) for p in parts)
    den = sum(p[2] * torch.exp(p[1] - m) for p in parts)
    o = (num / den).transpose(1, 2).reshape(batch, lq, heads * head_dim).to(dtype)
    return o if out is None else out.copy_(o)


def gemm(a, w, bias=None, epilogue=EPI_NONE, residual=None, gate=None, sel=None, out=None, out2=None, split=0, tile_m=0):
    y = F.linear(a, w, bias)
    if out2 is not None:                                  # fino_gemm_split_n: columns [split, N) to a second buffer
        assert epilogue == EPI_NONE and residual is None
        out.copy_(y[:, :split])
        out2.copy_(y[:, split:])
        return out, out2
    if epilogue == EPI_GELU_TANH:
        y = F.gelu(y, approximate="tanh")
    elif epilogue == EPI_RESIDUAL:
        y = residual + y
    elif epilogue == EPI_GATED_RESIDUAL:
        y = (residual.float() + y.float() * _g(gate, sel)).to(a.dtype)
    return y if out is None else out.copy_(y)


def gemm_blocked_a(a_blocks, rows, w, bias, residual, gate, sel, out, tile_m=0):
    if a_blocks.dim() == 4:                               # [groups, peers, rows_pad, bk]: K block j * groups + g = [g, j]
        a_blocks = a_blocks.permute(1, 0, 2, 3).reshape(-1, a_blocks.shape[2], a_blocks.shape[3])
    a = a_blocks[:, :rows].permute(1, 0, 2).reshape(rows, -1)
    return gemm(a, w, bias, EPI_GATED_RESIDUAL, residual, gate, sel, out=out)


def skinny_linear(x, w, b=None, silu_input=False):
    if silu_input:
        x = F.silu(x)
    return F.linear(x, w.float(), None if b is None else b.float())


def patchify(x, patch, out=None):
    c, f, h, w = x.shape
    pt, ph, pw = patch
    a = x.reshape(c, f // pt, pt, h // ph, ph, w // pw, pw).permute(1, 3, 5, 0, 2, 4, 6)
    a = a.reshape((f // pt) * (h // ph) * (w // pw), c * pt * ph * pw)
    return a if out is None else out.copy_(a)


def unpatchify(y, cout, frames, height, width, patch, out=None):
    pt, ph, pw = patch
    t = y.reshape(1, frames // pt, height // ph, width // pw, pt, ph, pw, cout).permute(0, 7, 1, 4, 2, 5, 3, 6)
    t = t.flatten(6, 7).flatten(4, 5).flatten(2, 3)[0]
    return t if out is None else out.copy_(t)


def wan_model_input(lat, cond, id_lat, traj, dtype, out=None):
    c, fg = lat.shape[:2]
    blend = torch.cat([cond, lat[:, 1:]], dim=1)
    top = blend if id_lat is None else torch.cat([blend, id_lat], dim=1)
    x = torch.cat([top, traj], dim=0).to(dtype)
    return x if out is None else out.copy_(x)


def cfg_euler_step_(lat, cond_pred, uncond_pred, guidance, dt_dev, round_out=True):
    fg = lat.shape[1]
    n = cond_pred if uncond_pred is None else uncond_pred + guidance * (cond_pred - uncond_pred)
    v = lat + dt_dev * n[:, :fg]
    if round_out:
        v = v.to(cond_pred.dtype).float()
    lat.copy_(v)
    return lat
