"""Oracle (CPU restatement) of the CogVideoX-5B DiT forward as FrameINO uses it (FrameIn variant, learned positional
embedding + 3D RoPE).  Test infrastructure.  Follows /root/reference/architecture/cogvideox_transformer_3d.py
(`CogVideoXTransformer3DModel.forward` :446-562, `CogVideoXBlock.forward` :122-161),
architecture/attention_processor.py (`CogVideoXAttnProcessor2_0.__call__` :2815-2877) and
architecture/embeddings.py (`CogVideoXPatchEmbed.forward` :718-805, `apply_rotary_emb` :1219-1258).
diffusers pieces (CogVideoXLayerNormZero, AdaLayerNorm, FeedForward, nn.LayerNorm qk-norm) are restated (unpinned).
"""
import torch
import torch.nn.functional as F

from .wan_dit import linear, sdpa, timestep_sinusoid


def cog_rope(x, cos, sin):
    """embeddings.py:1239-1258 (use_real, unbind_dim=-1): x [B,H,L,Dh], cos/sin [L,Dh]."""
    xr, xi = x.reshape(*x.shape[:-1], -1, 2).unbind(-1)
    rot = torch.stack([-xi, xr], dim=-1).flatten(3)
    return (x.float() * cos[None, None] + rot.float() * sin[None, None]).to(x.dtype)


def cog_attention(sd, p, heads, eps, hidden_states, encoder_hidden_states, rotary):
    lt = encoder_hidden_states.size(1)
    hs = torch.cat([encoder_hidden_states, hidden_states], dim=1)
    b = hs.shape[0]
    q, k, v = (linear(sd, f"{p}.{n}", hs) for n in ("to_q", "to_k", "to_v"))
    dh = q.shape[-1] // heads
    q, k, v = (t.view(b, -1, heads, dh).transpose(1, 2) for t in (q, k, v))
    q = F.layer_norm(q, (dh,), sd[p + ".norm_q.weight"], sd[p + ".norm_q.bias"], eps)
    k = F.layer_norm(k, (dh,), sd[p + ".norm_k.weight"], sd[p + ".norm_k.bias"], eps)
    if rotary is not None:
        q = torch.cat([q[:, :, :lt], cog_rope(q[:, :, lt:], *rotary)], dim=2)
        k = torch.cat([k[:, :, :lt], cog_rope(k[:, :, lt:], *rotary)], dim=2)
    o = sdpa(q, k, v).transpose(1, 2).reshape(b, -1, heads * dh)
    o = linear(sd, p + ".to_out.0", o)
    return o[:, lt:], o[:, :lt]


def layer_norm_zero(sd, p, eps, h, e, temb):
    """diffusers CogVideoXLayerNormZero (restated)."""
    shift, scale, gate, e_shift, e_scale, e_gate = linear(sd, p + ".linear", F.silu(temb)).chunk(6, dim=1)
    d = h.shape[-1]
    w, b = sd.get(p + ".norm.weight"), sd.get(p + ".norm.bias")
    hn = F.layer_norm(h, (d,), w, b, eps) * (1 + scale)[:, None, :] + shift[:, None, :]
    en = F.layer_norm(e, (d,), w, b, eps) * (1 + e_scale)[:, None, :] + e_shift[:, None, :]
    return hn, en, gate[:, None, :], e_gate[:, None, :]


def cog_block(sd, p, cfg, h, e, temb, rotary):
    lt = e.size(1)
    eps = cfg["norm_eps"]
    hn, en, g, eg = layer_norm_zero(sd, p + ".norm1", eps, h, e, temb)
    ah, ae = cog_attention(sd, p + ".attn1", cfg["num_attention_heads"], 1e-6, hn, en, rotary)
    h = h + g * ah
    e = e + eg * ae
    hn, en, g, eg = layer_norm_zero(sd, p + ".norm2", eps, h, e, temb)
    x = torch.cat([en, hn], dim=1)
    ff = linear(sd, p + ".ff.net.2", F.gelu(linear(sd, p + ".ff.net.0.proj", x), approximate="tanh"))
    h = h + g * ff[:, lt:]
    e = e + eg * ff[:, :lt]
    return h, e


def cog_pos_embeds(sd, cfg, num_frames, height, width, text_len, dtype):
    """embeddings.py:764-802: FrameIn appends the first frame's PE for the ID frame; trilinear resize off-default."""
    pos = sd["patch_embed.pos_embedding"]
    ps, tcr = cfg["patch_size"], cfg.get("temporal_compression_ratio", 4)
    maxt = cfg["max_text_seq_length"]
    post_frames = (cfg["sample_frames"] - 1) // tcr + 1
    pph, ppw = cfg["sample_height"] // ps, cfg["sample_width"] // ps
    seq = height * width * num_frames // (ps * ps)
    if cfg.get("use_FrameIn", False):
        first = (pos.shape[1] - maxt) // (num_frames - 1)
        pos = torch.cat([pos, pos[:, text_len:text_len + first].clone()], dim=1)
    pre_frames = (num_frames - 1) * tcr + 1
    if cfg["sample_height"] != height or cfg["sample_width"] != width or cfg["sample_frames"] != pre_frames:
        if cfg.get("use_FrameIn", False):
            post_frames += 1
        d = pos.shape[-1]
        pw = pos[:, text_len:].view(1, post_frames, pph, ppw, d).permute(0, 4, 1, 2, 3)
        pw = F.interpolate(pw, size=[post_frames, height // ps, width // ps], mode="trilinear", align_corners=False)
        pw = pw.permute(0, 2, 3, 4, 1).reshape(1, -1, d)
        pos = torch.cat([pos[:, :text_len], pw], dim=1)[:, :text_len + seq]
    return pos.to(dtype)


def cog_forward(sd, cfg, hidden_states, encoder_hidden_states, timestep, image_rotary_emb):
    b, nf, c, hh, ww = hidden_states.shape
    inner = cfg["num_attention_heads"] * cfg["attention_head_dim"]
    ps = cfg["patch_size"]
    t_emb = timestep_sinusoid(timestep, inner, cfg.get("flip_sin_to_cos", True), cfg.get("freq_shift", 0))
    t_emb = t_emb.to(hidden_states.dtype)
    emb = linear(sd, "time_embedding.linear_2", F.silu(linear(sd, "time_embedding.linear_1", t_emb)))
    # patch embed (:718-805)
    txt = linear(sd, "patch_embed.text_proj", encoder_hidden_states)
    lt = txt.shape[1]
    img = F.conv2d(hidden_states.reshape(-1, c, hh, ww), sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"],
                   stride=ps)
    img = img.view(b, nf, *img.shape[1:]).flatten(3).transpose(2, 3).flatten(1, 2)
    x = torch.cat([txt, img], dim=1).contiguous()
    x = x + cog_pos_embeds(sd, cfg, nf, hh, ww, lt, x.dtype)
    e, h = x[:, :lt], x[:, lt:]
    for i in range(cfg["num_layers"]):
        h, e = cog_block(sd, f"transformer_blocks.{i}", cfg, h, e, emb, image_rotary_emb)
    x = torch.cat([e, h], dim=1)
    x = F.layer_norm(x, (inner,), sd.get("norm_final.weight"), sd.get("norm_final.bias"), cfg["norm_eps"])[:, lt:]
    # AdaLayerNorm(chunk_dim=1) (restated)
    shift, scale = linear(sd, "norm_out.linear", F.silu(emb)).chunk(2, dim=1)
    x = F.layer_norm(x, (inner,), sd.get("norm_out.norm.weight"), sd.get("norm_out.norm.bias"), cfg["norm_eps"])
    x = x * (1 + scale[:, None, :]) + shift[:, None, :]
    x = linear(sd, "proj_out", x)
    out = x.reshape(b, nf, hh // ps, ww // ps, -1, ps, ps).permute(0, 1, 4, 2, 5, 3, 6).flatten(5, 6).flatten(3, 4)
    return out
