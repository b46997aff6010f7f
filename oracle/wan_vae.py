"""Oracle (CPU restatement) of the Wan 3D causal VAE (Wan2.2 residual variant) -- test infrastructure.

Follows /root/reference/architecture/autoencoder_kl_wan.py (line numbers cite that file), but in the
**whole-sequence** form the MI355X build uses: the reference streams the time axis in chunks (1 frame, then 1 latent
frame = 4 pixel frames at a time) and carries the last CACHE_T=2 input frames of every causal conv in `feat_cache`
(:34, :350-358).  Because every temporal conv is causal, chunking is only a schedule; the same outputs are obtained by
running each layer once over the whole sequence with these rules (checked against the reference's chunked run in
tests/test_oracle_golden.py):

  * WanCausalConv3d (:134-176): zero-pad 2*pad_t frames at the FRONT of the sequence.
  * upsample3d (:267-291): the first frame is never temporally upsampled ("Rep" sentinel); frames 1.. go through
    time_conv as a sequence whose history BEFORE frame 1 is zeros (the reference feeds zeros, not frame 0);
    the 2C output channels are interleaved into time.
  * downsample3d (:297-307): frame 0 passes; out[k] = time_conv(f[2k-2], f[2k-1], f[2k]) for k >= 1 (stride 2).
  * DupUp3D (:90-131): frame 0 keeps only its last temporal copy (first_chunk), later frames all factor_t copies.
  * AvgDown3D (:37-87): one zero frame in front, then non-overlapping groups (the reference pads the 1-frame chunk).

State-dict keys are the reference's parameter names (encoder.*, decoder.*, quant_conv.*, post_quant_conv.*).
"""
import torch
import torch.nn.functional as F


def _conv3d(xp, w, b, stride=(1, 1, 1)):
    """F.conv3d(xp, w, b, stride) on an already padded input.  On the host: the library call.  On a GPU in fp32 (the
    full-size checks of tests/test_wan_vae_gpu.py run this oracle on the device): the same sum written as one matrix
    product per kernel tap, out += W[:, :, tap] @ x[tap-shifted] -- the vendor library's fp32 3-D convolution takes
    minutes per layer at 704x1280, a matmul does not; only the fp32 summation order differs."""
    if not (xp.is_cuda and xp.dtype == torch.float32):
        return F.conv3d(xp, w, b, stride=stride)
    bsz, ci, tp, hp, wp = xp.shape
    co, _, kt, kh, kw = w.shape
    st, sh, sw = stride
    to, ho, wo = (tp - kt) // st + 1, (hp - kh) // sh + 1, (wp - kw) // sw + 1
    out = torch.zeros(bsz, co, to * ho * wo, dtype=xp.dtype, device=xp.device)
    for dt in range(kt):
        for dh in range(kh):
            for dw in range(kw):
                xs = xp[:, :, dt:dt + st * (to - 1) + 1:st, dh:dh + sh * (ho - 1) + 1:sh, dw:dw + sw * (wo - 1) + 1:sw]
                out.baddbmm_(w[None, :, :, dt, dh, dw].expand(bsz, -1, -1), xs.reshape(bsz, ci, -1))
    out = out.view(bsz, co, to, ho, wo)
    return out if b is None else out + b.view(1, -1, 1, 1, 1)


def _conv2d(x, w, b, stride=1, padding=0):
    """F.conv2d on [N, C, H, W] (frames as the batch) through the same switch"""
    if not (x.is_cuda and x.dtype == torch.float32):
        return F.conv2d(x, w, b, stride=stride, padding=padding)
    xp = F.pad(x, (padding, padding, padding, padding)) if padding else x
    s = stride if isinstance(stride, int) else stride[0]
    y = _conv3d(xp.permute(1, 0, 2, 3)[None], w[:, :, None], b, (1, s, s))           # frames on the (stride-1) time axis
    return y[0].permute(1, 0, 2, 3)


def causal_conv3d(x, w, b, stride=(1, 1, 1)):
    """WanCausalConv3d.forward (:169-176) over a whole sequence: pad (kw//2, kh//2) spatially as the module was built
    (padding = k//2 for the 3x3x3 convs, 0 for 1x1x1 and for the strided time_conv), 2*pad_t zeros at the front."""
    kt, kh, kw = w.shape[2:]
    return _conv3d(F.pad(x, (kw // 2, kw // 2, kh // 2, kh // 2, 0, 0)), w, b, stride) if kt == 1 else \
        _conv3d(F.pad(x, (kw // 2, kw // 2, kh // 2, kh // 2, kt - 1, 0)), w, b, stride)


def rms_norm(x, gamma, channel_dim=1):
    """WanRMS_norm.forward (:201-202): F.normalize over channels * sqrt(C) * gamma (bias=False everywhere used)."""
    c = x.shape[channel_dim]
    return F.normalize(x, dim=channel_dim) * (c ** 0.5) * gamma


def res_block(sd, p, x):
    """WanResidualBlock.forward (:342-382)."""
    h = causal_conv3d(x, sd[p + ".conv_shortcut.weight"], sd[p + ".conv_shortcut.bias"]) \
        if (p + ".conv_shortcut.weight") in sd else x
    y = F.silu(rms_norm(x, sd[p + ".norm1.gamma"]))
    y = causal_conv3d(y, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"])
    y = F.silu(rms_norm(y, sd[p + ".norm2.gamma"]))
    y = causal_conv3d(y, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"])
    return y + h


def attention_block(sd, p, x):
    """WanAttentionBlock.forward (:402-427): single-head spatial attention per frame."""
    b, c, t, h, w = x.shape
    y = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    y = rms_norm(y, sd[p + ".norm.gamma"])
    qkv = _conv2d(y, sd[p + ".to_qkv.weight"], sd[p + ".to_qkv.bias"])
    qkv = qkv.reshape(b * t, 1, c * 3, -1).permute(0, 1, 3, 2).contiguous()
    q, k, v = qkv.chunk(3, dim=-1)
    y = F.scaled_dot_product_attention(q, k, v)
    y = y.squeeze(1).permute(0, 2, 1).reshape(b * t, c, h, w)
    y = _conv2d(y, sd[p + ".proj.weight"], sd[p + ".proj.bias"])
    return y.view(b, t, c, h, w).permute(0, 2, 1, 3, 4) + x


def mid_block(sd, p, x):
    x = res_block(sd, p + ".resnets.0", x)
    x = attention_block(sd, p + ".attentions.0", x)
    return res_block(sd, p + ".resnets.1", x)


def _per_frame_conv2d(x, w, b, **kw):
    bsz, c, t, h, ww = x.shape
    y = _conv2d(x.permute(0, 2, 1, 3, 4).reshape(bsz * t, c, h, ww), w, b, **kw)
    return y.view(bsz, t, *y.shape[1:]).permute(0, 2, 1, 3, 4)


def upsample(sd, p, x, temporal):
    """WanResample.forward, modes upsample3d / upsample2d (:265-295), whole-sequence form."""
    if temporal and x.shape[2] > 1:
        rest = x[:, :, 1:]
        y = causal_conv3d(rest, sd[p + ".time_conv.weight"], sd[p + ".time_conv.bias"])       # [B, 2C, T-1, H, W]
        b, c2, t, h, w = y.shape
        y = y.reshape(b, 2, c2 // 2, t, h, w)
        y = torch.stack((y[:, 0], y[:, 1]), 3).reshape(b, c2 // 2, 2 * t, h, w)                 # :289-291
        x = torch.cat([x[:, :, :1], y], dim=2)
    b, c, t, h, w = x.shape
    y = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    y = F.interpolate(y.float(), scale_factor=(2.0, 2.0), mode="nearest-exact").type_as(y)     # :216-217
    y = _conv2d(y, sd[p + ".resample.1.weight"], sd[p + ".resample.1.bias"], padding=1)
    return y.view(b, t, *y.shape[1:]).permute(0, 2, 1, 3, 4)


def downsample(sd, p, x, temporal):
    """WanResample.forward, modes downsample3d / downsample2d (:293-308), whole-sequence form."""
    y = _per_frame_conv2d(F.pad(x, (0, 1, 0, 1)), sd[p + ".resample.1.weight"], sd[p + ".resample.1.bias"], stride=2)
    if temporal and y.shape[2] > 1:
        # out[k] = conv(f[2k-2], f[2k-1], f[2k]), k >= 1  == stride-2 conv over the sequence without extra padding
        z = _conv3d(y, sd[p + ".time_conv.weight"], sd[p + ".time_conv.bias"], (2, 1, 1))
        y = torch.cat([y[:, :, :1], z], dim=2)
    return y


def dup_up3d(x, out_channels, factor_t, factor_s):
    """DupUp3D.forward (:109-131); frame 0 is the reference's first chunk (keeps its last temporal copy only)."""
    b, c, t, h, w = x.shape
    factor = factor_t * factor_s * factor_s
    rep = out_channels * factor // c
    y = x.repeat_interleave(rep, dim=1).view(b, out_channels, factor_t, factor_s, factor_s, t, h, w)
    y = y.permute(0, 1, 5, 2, 6, 3, 7, 4).reshape(b, out_channels, t * factor_t, h * factor_s, w * factor_s)
    return y[:, :, factor_t - 1:]


def avg_down3d(x, out_channels, factor_t, factor_s):
    """AvgDown3D.forward (:55-87) over the whole sequence: the reference pads the 1-frame first chunk at the front."""
    pad_t = (factor_t - x.shape[2] % factor_t) % factor_t
    x = F.pad(x, (0, 0, 0, 0, pad_t, 0))
    b, c, t, h, w = x.shape
    factor = factor_t * factor_s * factor_s
    y = x.view(b, c, t // factor_t, factor_t, h // factor_s, factor_s, w // factor_s, factor_s)
    y = y.permute(0, 1, 3, 5, 7, 2, 4, 6).reshape(b, c * factor, t // factor_t, h // factor_s, w // factor_s)
    return y.view(b, out_channels, c * factor // out_channels, t // factor_t, h // factor_s, w // factor_s).mean(dim=2)


def vae_dims(cfg):
    mult = list(cfg["dim_mult"])
    enc = [cfg["base_dim"] * u for u in [1] + mult]
    dec = [cfg["decoder_base_dim"] * u for u in [mult[-1]] + mult[::-1]]
    return enc, dec


def vae_patchify(x, p):
    """:912-932"""
    b, c, f, h, w = x.shape
    x = x.view(b, c, f, h // p, p, w // p, p).permute(0, 1, 6, 4, 2, 3, 5).contiguous()
    return x.view(b, c * p * p, f, h // p, w // p)


def vae_unpatchify(x, p):
    """:935-952"""
    b, cp, f, h, w = x.shape
    c = cp // (p * p)
    x = x.view(b, c, p, p, f, h, w).permute(0, 1, 4, 5, 3, 6, 2).contiguous()
    return x.view(b, c, f, h * p, w * p)


def wan_vae_decode(sd, cfg, z):
    """AutoencoderKLWan._decode (:1198-1227) with WanDecoder3d.forward (:874-909), is_residual=True.
    z [B, z_dim, T, h, w] -> video [B, 3, 1+4(T-1), 16h, 16w] clamped to [-1, 1]."""
    assert cfg.get("is_residual", True), "only the Wan2.2 residual VAE is on FrameINO's path"
    _, dec = vae_dims(cfg)
    tup = list(cfg["temperal_downsample"])[::-1]
    nres = cfg["num_res_blocks"]
    x = causal_conv3d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    x = causal_conv3d(x, sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"])
    x = mid_block(sd, "decoder.mid_block", x)
    nb = len(cfg["dim_mult"])
    for i in range(nb):
        p = f"decoder.up_blocks.{i}"
        up_flag = i != nb - 1
        temporal = bool(tup[i]) if up_flag else False
        x_copy = x
        for r in range(nres + 1):
            x = res_block(sd, f"{p}.resnets.{r}", x)
        if up_flag:
            x = upsample(sd, p + ".upsampler", x, temporal)
            x = x + dup_up3d(x_copy, dec[i + 1], 2 if temporal else 1, 2)                     # :706-709
    x = F.silu(rms_norm(x, sd["decoder.norm_out.gamma"]))
    x = causal_conv3d(x, sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"])
    if cfg.get("patch_size"):
        x = vae_unpatchify(x, cfg["patch_size"])
    return torch.clamp(x, min=-1.0, max=1.0)


def wan_vae_encode(sd, cfg, x):
    """AutoencoderKLWan._encode (:1145-1169) + quant_conv; returns the moments [B, 2*z_dim, T', h, w]
    (DiagonalGaussianDistribution.mode() is the first z_dim channels)."""
    assert cfg.get("is_residual", True)
    enc, _ = vae_dims(cfg)
    tdown = list(cfg["temperal_downsample"])
    nres = cfg["num_res_blocks"]
    if cfg.get("patch_size"):
        x = vae_patchify(x, cfg["patch_size"])
    x = causal_conv3d(x, sd["encoder.conv_in.weight"], sd["encoder.conv_in.bias"])
    nb = len(cfg["dim_mult"])
    for i in range(nb):
        p = f"encoder.down_blocks.{i}"
        down_flag = i != nb - 1
        temporal = bool(tdown[i]) if down_flag else False
        x_copy = x
        for r in range(nres):
            x = res_block(sd, f"{p}.resnets.{r}", x)
        if down_flag:
            x = downsample(sd, p + ".downsampler", x, temporal)
        x = x + avg_down3d(x_copy, enc[i + 1], 2 if temporal else 1, 2 if down_flag else 1)   # :479-502
    x = mid_block(sd, "encoder.mid_block", x)
    x = F.silu(rms_norm(x, sd["encoder.norm_out.gamma"]))
    x = causal_conv3d(x, sd["encoder.conv_out.weight"], sd["encoder.conv_out.bias"])
    return causal_conv3d(x, sd["quant_conv.weight"], sd["quant_conv.bias"])
