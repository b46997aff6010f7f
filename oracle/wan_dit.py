"""Oracle (CPU restatement) of the Wan2.2 DiT forward used by FrameINO.  Test infrastructure.

Functional code over a flat state-dict `sd` whose keys are the reference's parameter names
(`blocks.3.attn1.to_q.weight`, ...).  Follows /root/reference/architecture/transformer_wan.py;
line numbers below cite that file unless another file is named.  Rounding points (where the
reference casts back to the activation dtype) are reproduced so that running the oracle in
bf16 on CPU mirrors the reference's bf16 numerics, and in fp32 gives the reference precision.
"""
import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- small pieces
def linear(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


def timestep_sinusoid(timesteps, dim, flip_sin_to_cos=True, downscale_freq_shift=0.0, max_period=10000):
    """architecture/embeddings.py:27-78 (get_timestep_embedding), as configured by
    transformer_wan.py:158 (flip_sin_to_cos=True, downscale_freq_shift=0)."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32, device=timesteps.device)
    exponent = exponent / (half - downscale_freq_shift)
    emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


def rope_1d(dim, max_len, theta=10000.0):
    """architecture/embeddings.py:1199-1207 with use_real, repeat_interleave_real, fp64 freqs."""
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float64)[: dim // 2] / dim))
    ang = torch.outer(torch.arange(max_len).to(torch.float64), freqs)
    return ang.cos().repeat_interleave(2, dim=1).float(), ang.sin().repeat_interleave(2, dim=1).float()


def wan_rope(head_dim, max_seq_len, ppf, pph, ppw, device=None):
    """WanRotaryPosEmbed (:192-253): t/h/w split of the head dim, tables [1,1,L,head_dim] (built on the CPU in fp64
    like the reference's registered buffers, then moved)."""
    h_dim = w_dim = 2 * (head_dim // 6)
    t_dim = head_dim - h_dim - w_dim
    tabs = [tuple(t.to(device) for t in rope_1d(d, max_seq_len)) for d in (t_dim, h_dim, w_dim)]

    def build(i):
        f = tabs[0][i][:ppf].view(ppf, 1, 1, -1).expand(ppf, pph, ppw, -1)
        h = tabs[1][i][:pph].view(1, pph, 1, -1).expand(ppf, pph, ppw, -1)
        w = tabs[2][i][:ppw].view(1, 1, ppw, -1).expand(ppf, pph, ppw, -1)
        return torch.cat([f, h, w], dim=-1).reshape(1, 1, ppf * pph * ppw, -1)

    return build(0), build(1)


def apply_wan_rope(x, cos, sin):
    """:75-87 -- adjacent-pair rotation; cos from even slots, sin from odd slots; result in x dtype."""
    xp = x.reshape(*x.shape[:-1], -1, 2)
    x1, x2 = xp[..., 0], xp[..., 1]
    c = cos[..., 0::2]
    s = sin[..., 1::2]
    out = torch.empty_like(x)
    out[..., 0::2] = x1 * c - x2 * s
    out[..., 1::2] = x1 * s + x2 * c
    return out.type_as(x)


def rms_norm(x, weight, eps):
    """diffusers RMSNorm (third-party, restated): fp32 statistics, cast to the weight's
    dtype when that is half precision, then multiply by the weight."""
    var = x.float().pow(2).mean(-1, keepdim=True)
    y = x * torch.rsqrt(var + eps)
    if weight.dtype in (torch.float16, torch.bfloat16):
        y = y.to(weight.dtype)
    return y * weight


def fp32_layer_norm(x, weight, bias, eps):
    """diffusers FP32LayerNorm (third-party, restated): layer_norm in fp32, cast back."""
    d = x.shape[-1]
    return F.layer_norm(
        x.float(), (d,), None if weight is None else weight.float(), None if bias is None else bias.float(), eps
    ).to(x.dtype)


# ----------------------------------------------------------------------------- attention
def sdpa(q, k, v):
    """F.scaled_dot_product_attention(q, k, v) (no mask, no dropout, non-causal, scale 1/sqrt(d)).  When the oracle is
    run on a GPU as the full-size checker (tests/test_fullsize_oracle_gpu.py) the [H, Lq, Lk] score tensor of the
    whole call would be tens of GB, so the same arithmetic is done a few heads at a time (heads are independent)."""
    if not q.is_cuda or q.shape[1] * q.shape[2] * k.shape[2] <= (1 << 28):
        return F.scaled_dot_product_attention(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False)
    step = max(1, (1 << 31) // (q.shape[2] * k.shape[2]))            # <= 8 GiB of fp32 scores per chunk
    out = torch.empty(q.shape[:-1] + (v.shape[-1],), dtype=q.dtype, device=q.device)
    scale = q.shape[-1] ** -0.5
    for bi in range(q.shape[0]):
        for h0 in range(0, q.shape[1], step):
            s = torch.matmul(q[bi, h0:h0 + step].float(), k[bi, h0:h0 + step].float().transpose(1, 2)) * scale
            out[bi, h0:h0 + step] = torch.matmul(torch.softmax(s, dim=-1), v[bi, h0:h0 + step].float()).to(q.dtype)
            del s
    return out


def wan_attention(sd, prefix, heads, eps, hidden_states, encoder_hidden_states=None, rotary_emb=None):
    """WanAttnProcessor2_0.__call__ (:43-119) with add_k_proj=None (TI2V-5B)."""
    ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
    q = linear(sd, prefix + ".to_q", hidden_states)
    k = linear(sd, prefix + ".to_k", ctx)
    v = linear(sd, prefix + ".to_v", ctx)
    q = rms_norm(q, sd[prefix + ".norm_q.weight"], eps)   # across all heads (:64-67)
    k = rms_norm(k, sd[prefix + ".norm_k.weight"], eps)
    q = q.unflatten(2, (heads, -1)).transpose(1, 2)
    k = k.unflatten(2, (heads, -1)).transpose(1, 2)
    v = v.unflatten(2, (heads, -1)).transpose(1, 2)
    if rotary_emb is not None:
        q = apply_wan_rope(q, *rotary_emb)
        k = apply_wan_rope(k, *rotary_emb)
    o = sdpa(q, k, v)
    o = o.transpose(1, 2).flatten(2, 3).type_as(q)
    return linear(sd, prefix + ".to_out.0", o)


def feed_forward(sd, prefix, x):
    """diffusers FeedForward("gelu-approximate") (third-party, restated)."""
    h = F.gelu(linear(sd, prefix + ".net.0.proj", x), approximate="tanh")
    return linear(sd, prefix + ".net.2", h)


# ----------------------------------------------------------------------------- block
def wan_block(sd, prefix, cfg, hidden_states, encoder_hidden_states, temb, rotary_emb):
    """WanTransformerBlock.forward (:308-350).  temb is [B,6,D] or [B,L,6,D]."""
    heads, eps = cfg["num_attention_heads"], cfg["eps"]
    table = sd[prefix + ".scale_shift_table"]
    if temb.ndim == 4:
        mods = (table.unsqueeze(0) + temb.float()).chunk(6, dim=2)
        shift_msa, scale_msa, gate_msa, c_shift, c_scale, c_gate = [m.squeeze(2) for m in mods]
    else:
        shift_msa, scale_msa, gate_msa, c_shift, c_scale, c_gate = (table + temb.float()).chunk(6, dim=1)

    x = hidden_states
    n = (fp32_layer_norm(x.float(), None, None, eps) * (1 + scale_msa) + shift_msa).type_as(x)          # :334
    a = wan_attention(sd, prefix + ".attn1", heads, eps, n, None, rotary_emb)
    x = (x.float() + a * gate_msa).type_as(x)                                                          # :336

    if cfg.get("cross_attn_norm", True):
        n = fp32_layer_norm(x.float(), sd[prefix + ".norm2.weight"], sd[prefix + ".norm2.bias"], eps).type_as(x)
    else:
        n = x.float().type_as(x)
    a = wan_attention(sd, prefix + ".attn2", heads, eps, n, encoder_hidden_states, None)
    x = x + a                                                                                          # :341

    n = (fp32_layer_norm(x.float(), None, None, eps) * (1 + c_scale) + c_shift).type_as(x)             # :344
    f = feed_forward(sd, prefix + ".ffn", n)
    x = (x.float() + f.float() * c_gate).type_as(x)                                                    # :348
    return x


# ----------------------------------------------------------------------------- model
def wan_condition_embedder(sd, cfg, timestep, encoder_hidden_states, timestep_seq_len=None):
    """WanTimeTextImageEmbedding.forward (:168-189), image branch absent (image_dim=None)."""
    p = "condition_embedder"
    t = timestep_sinusoid(timestep, cfg["freq_dim"])
    if timestep_seq_len is not None:
        t = t.unflatten(0, (1, timestep_seq_len))
    te_dtype = sd[p + ".time_embedder.linear_1.weight"].dtype
    t = t.to(te_dtype)
    temb = linear(sd, p + ".time_embedder.linear_2", F.silu(linear(sd, p + ".time_embedder.linear_1", t)))
    temb = temb.type_as(encoder_hidden_states)
    timestep_proj = linear(sd, p + ".time_proj", F.silu(temb))
    txt = linear(sd, p + ".text_embedder.linear_1", encoder_hidden_states)
    txt = linear(sd, p + ".text_embedder.linear_2", F.gelu(txt, approximate="tanh"))
    return temb, timestep_proj, txt


def wan_forward(sd, cfg, hidden_states, timestep, encoder_hidden_states):
    """WanTransformer3DModel.forward (:454-552).  Returns [B, out_channels, F, H, W]."""
    b, c, nf, hh, ww = hidden_states.shape
    pt, ph, pw = cfg["patch_size"]
    ppf, pph, ppw = nf // pt, hh // ph, ww // pw
    rotary = wan_rope(cfg["attention_head_dim"], cfg["rope_max_seq_len"], ppf, pph, ppw, hidden_states.device)  # :484

    x = F.conv3d(hidden_states, sd["patch_embedding.weight"], sd["patch_embedding.bias"], stride=(pt, ph, pw))
    x = x.flatten(2).transpose(1, 2)                                                          # :486-487

    if timestep.ndim == 2:
        ts_len = timestep.shape[1]
        timestep = timestep.flatten()
    else:
        ts_len = None
    temb, tproj, txt = wan_condition_embedder(sd, cfg, timestep, encoder_hidden_states, ts_len)
    tproj = tproj.unflatten(2, (6, -1)) if ts_len is not None else tproj.unflatten(1, (6, -1))

    for i in range(cfg["num_layers"]):
        x = wan_block(sd, f"blocks.{i}", cfg, x, txt, tproj, rotary)

    table = sd["scale_shift_table"]
    if temb.ndim == 3:
        shift, scale = (table.unsqueeze(0) + temb.unsqueeze(2)).chunk(2, dim=2)               # :522
        shift, scale = shift.squeeze(2), scale.squeeze(2)
    else:
        shift, scale = (table + temb.unsqueeze(1)).chunk(2, dim=1)                            # :527
    x = (fp32_layer_norm(x.float(), None, None, cfg["eps"]) * (1 + scale) + shift).type_as(x)  # :536
    x = linear(sd, "proj_out", x)
    x = x.reshape(b, ppf, pph, ppw, pt, ph, pw, -1).permute(0, 7, 1, 4, 2, 5, 3, 6)           # :539-542
    return x.flatten(6, 7).flatten(4, 5).flatten(2, 3)


# ----------------------------------------------------------------------------- random weights
WAN22_5B_CFG = dict(
    patch_size=(1, 2, 2), num_attention_heads=24, attention_head_dim=128, in_channels=96, out_channels=48,
    text_dim=4096, freq_dim=256, ffn_dim=14336, num_layers=30, cross_attn_norm=True, eps=1e-6,
    rope_max_seq_len=1024,
)

FP32_KEEP = ("time_embedder", "scale_shift_table", "norm1", "norm2", "norm3")   # :393 _keep_in_fp32_modules


def wan_param_shapes(cfg):
    d = cfg["num_attention_heads"] * cfg["attention_head_dim"]
    pt, ph, pw = cfg["patch_size"]
    f = cfg["ffn_dim"]
    s = {
        "patch_embedding.weight": (d, cfg["in_channels"], pt, ph, pw), "patch_embedding.bias": (d,),
        "condition_embedder.time_embedder.linear_1.weight": (d, cfg["freq_dim"]),
        "condition_embedder.time_embedder.linear_1.bias": (d,),
        "condition_embedder.time_embedder.linear_2.weight": (d, d),
        "condition_embedder.time_embedder.linear_2.bias": (d,),
        "condition_embedder.time_proj.weight": (6 * d, d), "condition_embedder.time_proj.bias": (6 * d,),
        "condition_embedder.text_embedder.linear_1.weight": (d, cfg["text_dim"]),
        "condition_embedder.text_embedder.linear_1.bias": (d,),
        "condition_embedder.text_embedder.linear_2.weight": (d, d),
        "condition_embedder.text_embedder.linear_2.bias": (d,),
        "scale_shift_table": (1, 2, d),
        "proj_out.weight": (cfg["out_channels"] * pt * ph * pw, d), "proj_out.bias": (cfg["out_channels"] * pt * ph * pw,),
    }
    for i in range(cfg["num_layers"]):
        p = f"blocks.{i}"
        s[p + ".scale_shift_table"] = (1, 6, d)
        for a in ("attn1", "attn2"):
            for n in ("to_q", "to_k", "to_v", "to_out.0"):
                s[f"{p}.{a}.{n}.weight"] = (d, d)
                s[f"{p}.{a}.{n}.bias"] = (d,)
            s[f"{p}.{a}.norm_q.weight"] = (d,)
            s[f"{p}.{a}.norm_k.weight"] = (d,)
        if cfg.get("cross_attn_norm", True):
            s[p + ".norm2.weight"] = (d,)
            s[p + ".norm2.bias"] = (d,)
        s[p + ".ffn.net.0.proj.weight"] = (f, d)
        s[p + ".ffn.net.0.proj.bias"] = (f,)
        s[p + ".ffn.net.2.weight"] = (d, f)
        s[p + ".ffn.net.2.bias"] = (d,)
    return s


def wan_random_state_dict(cfg, seed=0, dtype=torch.float32, std=0.02):
    """Seeded random weights (no checkpoints offline).  Norm gains ~1, everything else N(0, std^2);
    the fp32 islands of :393 stay fp32 whatever `dtype` is."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in wan_param_shapes(cfg).items():
        if name.endswith("norm_q.weight") or name.endswith("norm_k.weight") or name.endswith("norm2.weight"):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif "scale_shift_table" in name:
            t = torch.randn(shape, generator=g) / shape[-1] ** 0.5
        else:
            t = std * torch.randn(shape, generator=g)
        keep32 = any(k in name for k in FP32_KEEP)
        sd[name] = t.to(torch.float32 if keep32 else dtype)
    return sd
