"""Oracle (CPU restatement) of diffusers' `AutoencoderKLCogVideoX` -- the VAE the CogVideoX FrameINO pipeline calls
(`vae.encode(x).latent_dist.sample()`, `vae.decode(z).sample`; call sites
pipelines/pipeline_cogvideox_i2v_motion_FrameINO.py:380-396, :426-431, :809-826 and
train_code/train_cogvideox_motion_FrameINO.py:493-546).  Test infrastructure.

THIRD-PARTY: the class lives in `diffusers` (unpinned by the reference, requirements.txt:12) and has no source under
/root/reference, so this file restates its published algorithm from the library's documented structure -- **parity
UNPINNED** (no golden vector can exist offline).  What is restated:

  * `CogVideoXCausalConv3d` (pad_mode "constant"): spatial zero padding k//2; temporal padding = (k_t - 1) copies of the
    FIRST frame of the sequence (or, when streaming, the last k_t - 1 input frames of the previous frame batch =
    `conv_cache`).
  * `CogVideoXResnetBlock3D`: norm1 -> SiLU -> conv1 -> norm2 -> SiLU -> conv2 (+ 1x1x1 `conv_shortcut` when the widths
    differ) + input;  norm = `nn.GroupNorm(32, C, eps 1e-6)` in the encoder, `CogVideoXSpatialNorm3D` in the decoder.
  * `CogVideoXSpatialNorm3D(f, zq)`: GroupNorm(f) * conv_y(zq') + conv_b(zq'), zq' = the latent nearest-interpolated to
    f's (T, H, W) -- first frame separately when T > 1 is odd; conv_y / conv_b are 1x1x1.
  * `CogVideoXDownsample3D`: (compress_time: average frame pairs, the first frame of an odd-length batch kept as is),
    zero-pad right/bottom by one, Conv2d 3x3 stride 2.  `CogVideoXUpsample3D`: nearest 2x in space (and in time when
    compress_time; the first frame of an odd-length batch only in space), Conv2d 3x3 pad 1.
  * Encoder: conv_in, 4 down blocks (3 resnets each; downsample except the last; compress_time in the first
    log2(temporal_compression_ratio) blocks), mid block (2 resnets), GroupNorm, SiLU, conv_out -> 2 x latent channels.
    Decoder: conv_in, mid block, 4 up blocks (4 resnets each; upsample except the last), SpatialNorm, SiLU, conv_out.
  * FRAME BATCHING of `_encode` / `_decode` (8 sample frames / 2 latent frames per batch, the remainder in the first
    batch, conv caches carried across batches).  GroupNorm statistics are therefore per frame batch: results depend
    on the batching, which is reproduced here.

  * TILING (`enable_tiling()`; round 5): `tiled_encode` / `tiled_decode` -- overlapping spatial tiles of half the config's
    sample size, overlap factors 1/6 and 1/5, every tile with its own frame batches and conv caches, `blend_v` / `blend_h`
    (the same loops as the in-tree Wan VAE's, architecture/autoencoder_kl_wan.py:1254-1268) -- restated from the library's
    documented behaviour like everything else here: parity UNPINNED, except the two blend loops, which are checked bit for bit
    against the in-tree ones (tests/golden/vae_blend.npz, tests/test_oracle_golden.py).

State-dict keys are diffusers' parameter names (`encoder.down_blocks.0.resnets.0.conv1.conv.weight`, ...)."""
import torch
import torch.nn.functional as F


def causal_conv3d(sd, name, x, cache):
    """-> (y, new_cache).  x [B, C, T, H, W]; `cache`: the previous batch's last (k_t - 1) input frames or None."""
    w, b = sd[name + ".conv.weight"], sd[name + ".conv.bias"]
    kt, kh, kw = w.shape[2:]
    if kt > 1:
        front = [cache] if cache is not None else [x[:, :, :1]] * (kt - 1)
        x = torch.cat(front + [x], dim=2)
    new_cache = x[:, :, -(kt - 1):].clone() if kt > 1 else None
    return F.conv3d(x, w, b, padding=(0, kh // 2, kw // 2)), new_cache


def group_norm(sd, name, x, groups, eps=1e-6):
    return F.group_norm(x, groups, sd[name + ".weight"], sd[name + ".bias"], eps)


def spatial_norm(sd, name, f, zq, groups, caches, key):
    if f.shape[2] > 1 and f.shape[2] % 2 == 1:
        z_first = F.interpolate(zq[:, :, :1], size=(1,) + tuple(f.shape[-2:]))
        z_rest = F.interpolate(zq[:, :, 1:], size=(f.shape[2] - 1,) + tuple(f.shape[-2:]))
        z = torch.cat([z_first, z_rest], dim=2)
    else:
        z = F.interpolate(zq, size=tuple(f.shape[-3:]))
    y, _ = causal_conv3d(sd, name + ".conv_y", z, None)            # kernel 1: no temporal context, no cache
    b, _ = causal_conv3d(sd, name + ".conv_b", z, None)
    return group_norm(sd, name + ".norm_layer", f, groups) * y + b


def resnet(sd, p, x, zq, groups, caches, new):
    h = x
    if zq is not None:
        h = spatial_norm(sd, p + ".norm1", h, zq, groups, caches, p + ".norm1")
    else:
        h = group_norm(sd, p + ".norm1", h, groups)
    h, new[p + ".conv1"] = causal_conv3d(sd, p + ".conv1", F.silu(h), caches.get(p + ".conv1"))
    if zq is not None:
        h = spatial_norm(sd, p + ".norm2", h, zq, groups, caches, p + ".norm2")
    else:
        h = group_norm(sd, p + ".norm2", h, groups)
    h, new[p + ".conv2"] = causal_conv3d(sd, p + ".conv2", F.silu(h), caches.get(p + ".conv2"))
    if (p + ".conv_shortcut.weight") in sd:
        x = F.conv3d(x, sd[p + ".conv_shortcut.weight"], sd[p + ".conv_shortcut.bias"])
    return h + x


def downsample(sd, p, x, compress_time):
    if compress_time:
        b, c, t, hh, ww = x.shape
        y = x.permute(0, 3, 4, 1, 2).reshape(b * hh * ww, c, t)
        if t % 2 == 1:
            first, rest = y[..., 0], y[..., 1:]
            if rest.shape[-1] > 0:
                rest = F.avg_pool1d(rest, kernel_size=2, stride=2)
            y = torch.cat([first[..., None], rest], dim=-1)
        else:
            y = F.avg_pool1d(y, kernel_size=2, stride=2)
        x = y.reshape(b, hh, ww, c, y.shape[-1]).permute(0, 3, 4, 1, 2)
    x = F.pad(x, (0, 1, 0, 1))
    b, c, t, hh, ww = x.shape
    y = F.conv2d(x.permute(0, 2, 1, 3, 4).reshape(b * t, c, hh, ww), sd[p + ".conv.weight"], sd[p + ".conv.bias"],
                 stride=2)
    return y.reshape(b, t, *y.shape[1:]).permute(0, 2, 1, 3, 4)


def upsample(sd, p, x, compress_time):
    if compress_time:
        if x.shape[2] > 1 and x.shape[2] % 2 == 1:
            first = F.interpolate(x[:, :, 0], scale_factor=2.0)
            rest = F.interpolate(x[:, :, 1:], scale_factor=2.0)
            x = torch.cat([first[:, :, None], rest], dim=2)
        elif x.shape[2] > 1:
            x = F.interpolate(x, scale_factor=2.0)
        else:
            x = F.interpolate(x.squeeze(2), scale_factor=2.0)[:, :, None]
    else:
        b, c, t, hh, ww = x.shape
        y = F.interpolate(x.permute(0, 2, 1, 3, 4).reshape(b * t, c, hh, ww), scale_factor=2.0)
        x = y.reshape(b, t, c, *y.shape[2:]).permute(0, 2, 1, 3, 4)
    b, c, t, hh, ww = x.shape
    y = F.conv2d(x.permute(0, 2, 1, 3, 4).reshape(b * t, c, hh, ww), sd[p + ".conv.weight"], sd[p + ".conv.bias"],
                 padding=1)
    return y.reshape(b, t, *y.shape[1:]).permute(0, 2, 1, 3, 4)


def _levels(cfg):
    n = len(cfg["block_out_channels"])
    tlevel = int(round(torch.log2(torch.tensor(float(cfg["temporal_compression_ratio"]))).item()))
    return n, tlevel


def encoder(sd, cfg, x, caches):
    new = {}
    g = cfg["norm_num_groups"]
    n, tlevel = _levels(cfg)
    h, new["encoder.conv_in"] = causal_conv3d(sd, "encoder.conv_in", x, caches.get("encoder.conv_in"))
    for i in range(n):
        for r in range(cfg["layers_per_block"]):
            h = resnet(sd, f"encoder.down_blocks.{i}.resnets.{r}", h, None, g, caches, new)
        if i != n - 1:
            h = downsample(sd, f"encoder.down_blocks.{i}.downsamplers.0", h, i < tlevel)
    for r in range(2):
        h = resnet(sd, f"encoder.mid_block.resnets.{r}", h, None, g, caches, new)
    h = F.silu(group_norm(sd, "encoder.norm_out", h, g))
    h, new["encoder.conv_out"] = causal_conv3d(sd, "encoder.conv_out", h, caches.get("encoder.conv_out"))
    return h, new


def decoder(sd, cfg, z, caches):
    new = {}
    g = cfg["norm_num_groups"]
    n, tlevel = _levels(cfg)
    h, new["decoder.conv_in"] = causal_conv3d(sd, "decoder.conv_in", z, caches.get("decoder.conv_in"))
    for r in range(2):
        h = resnet(sd, f"decoder.mid_block.resnets.{r}", h, z, g, caches, new)
    for i in range(n):
        for r in range(cfg["layers_per_block"] + 1):
            h = resnet(sd, f"decoder.up_blocks.{i}.resnets.{r}", h, z, g, caches, new)
        if i != n - 1:
            h = upsample(sd, f"decoder.up_blocks.{i}.upsamplers.0", h, i < tlevel)
    h = F.silu(spatial_norm(sd, "decoder.norm_out", h, z, g, caches, "decoder.norm_out"))
    h, new["decoder.conv_out"] = causal_conv3d(sd, "decoder.conv_out", h, caches.get("decoder.conv_out"))
    return h, new


def frame_batches(num_frames, batch):
    """the slices `_encode` / `_decode` walk: the remainder rides in the first batch"""
    nb = max(num_frames // batch, 1)
    rem = num_frames % batch
    return [(batch * i + (0 if i == 0 else rem), batch * (i + 1) + rem) for i in range(nb)]


def encode_moments(sd, cfg, x, sample_batch=8):
    """AutoencoderKLCogVideoX._encode -> the posterior parameters [B, 2 * latent, T', h, w] (mean | logvar)."""
    caches, outs = {}, []
    for s, e in frame_batches(x.shape[2], sample_batch):
        y, caches = encoder(sd, cfg, x[:, :, s:e], caches)
        outs.append(y)
    return torch.cat(outs, dim=2)


def decode(sd, cfg, z, latent_batch=2):
    """AutoencoderKLCogVideoX._decode -> [B, 3, T, H, W]."""
    caches, outs = {}, []
    for s, e in frame_batches(z.shape[2], latent_batch):
        y, caches = decoder(sd, cfg, z[:, :, s:e], caches)
        outs.append(y)
    return torch.cat(outs, dim=2)


# ------------------------------------------------------------------------------------------------ tiling (enable_tiling)
def tiling_params(cfg, tile_sample_min_height=None, tile_sample_min_width=None, tile_overlap_factor_height=None,
                  tile_overlap_factor_width=None):
    """diffusers' `AutoencoderKLCogVideoX.__init__` / `enable_tiling` bookkeeping: the minimum tile is HALF the sample size the
    config names (480 x 720 -> 240 x 360), its latent size that divided by 2^(blocks - 1); overlap factors 1/6 (height) and 1/5
    (width).  The canonical CogVideoX caller switches it on (reference test_code/run_cogvideox_FrameIn_mass_evaluation.py:95-96)."""
    sh = tile_sample_min_height or cfg.get("sample_height", 480) // 2
    sw = tile_sample_min_width or cfg.get("sample_width", 720) // 2
    down = 2 ** (len(cfg["block_out_channels"]) - 1)
    return dict(sample_h=sh, sample_w=sw, latent_h=int(sh / down), latent_w=int(sw / down),
                of_h=tile_overlap_factor_height or 1 / 6, of_w=tile_overlap_factor_width or 1 / 5)


def blend_v(a, b, blend_extent):
    """the loop of diffusers' blend_v (identical to the in-tree architecture/autoencoder_kl_wan.py:1254-1260), in place on b"""
    blend_extent = min(a.shape[3], b.shape[3], blend_extent)
    for y in range(blend_extent):
        b[:, :, :, y, :] = a[:, :, :, -blend_extent + y, :] * (1 - y / blend_extent) + b[:, :, :, y, :] * (y / blend_extent)
    return b


def blend_h(a, b, blend_extent):
    """... and blend_h (:1262-1268)"""
    blend_extent = min(a.shape[4], b.shape[4], blend_extent)
    for x in range(blend_extent):
        b[:, :, :, :, x] = a[:, :, :, :, -blend_extent + x] * (1 - x / blend_extent) + b[:, :, :, :, x] * (x / blend_extent)
    return b


def _blend_rows(rows, blend_h_extent, blend_w_extent, limit_h, limit_w):
    """the second half of tiled_encode / tiled_decode: every tile blended IN PLACE with its upper and left neighbour (which have
    been blended already: row-major order), cropped to the stride, concatenated"""
    result_rows = []
    for i, row in enumerate(rows):
        result_row = []
        for j, tile in enumerate(row):
            if i > 0:
                tile = blend_v(rows[i - 1][j], tile, blend_h_extent)
            if j > 0:
                tile = blend_h(row[j - 1], tile, blend_w_extent)
            result_row.append(tile[:, :, :, :limit_h, :limit_w])
        result_rows.append(torch.cat(result_row, dim=4))
    return torch.cat(result_rows, dim=3)


def tiled_encode_moments(sd, cfg, x, tp=None, sample_batch=8):
    """AutoencoderKLCogVideoX.tiled_encode: overlapping spatial tiles, each walked in the frame batches of `_encode` with its own
    conv caches, blended over the overlap.  (`_encode` takes this path when tiling is on and the frame is larger than a tile.)"""
    tp = tp or tiling_params(cfg)
    height, width = x.shape[3], x.shape[4]
    overlap_h, overlap_w = int(tp["sample_h"] * (1 - tp["of_h"])), int(tp["sample_w"] * (1 - tp["of_w"]))
    blend_eh, blend_ew = int(tp["latent_h"] * tp["of_h"]), int(tp["latent_w"] * tp["of_w"])
    limit_h, limit_w = tp["latent_h"] - blend_eh, tp["latent_w"] - blend_ew
    rows = []
    for i in range(0, height, overlap_h):
        row = []
        for j in range(0, width, overlap_w):
            caches, outs = {}, []
            for s, e in frame_batches(x.shape[2], sample_batch):
                y, caches = encoder(sd, cfg, x[:, :, s:e, i:i + tp["sample_h"], j:j + tp["sample_w"]], caches)
                outs.append(y)
            row.append(torch.cat(outs, dim=2))
        rows.append(row)
    return _blend_rows(rows, blend_eh, blend_ew, limit_h, limit_w)


def tiled_decode(sd, cfg, z, tp=None, latent_batch=2):
    """AutoencoderKLCogVideoX.tiled_decode"""
    tp = tp or tiling_params(cfg)
    height, width = z.shape[3], z.shape[4]
    overlap_h, overlap_w = int(tp["latent_h"] * (1 - tp["of_h"])), int(tp["latent_w"] * (1 - tp["of_w"]))
    blend_eh, blend_ew = int(tp["sample_h"] * tp["of_h"]), int(tp["sample_w"] * tp["of_w"])
    limit_h, limit_w = tp["sample_h"] - blend_eh, tp["sample_w"] - blend_ew
    rows = []
    for i in range(0, height, overlap_h):
        row = []
        for j in range(0, width, overlap_w):
            caches, outs = {}, []
            for s, e in frame_batches(z.shape[2], latent_batch):
                y, caches = decoder(sd, cfg, z[:, :, s:e, i:i + tp["latent_h"], j:j + tp["latent_w"]], caches)
                outs.append(y)
            row.append(torch.cat(outs, dim=2))
        rows.append(row)
    return _blend_rows(rows, blend_eh, blend_ew, limit_h, limit_w)


def uses_tiling_encode(x, tp):
    return x.shape[4] > tp["sample_w"] or x.shape[3] > tp["sample_h"]


def uses_tiling_decode(z, tp):
    return z.shape[4] > tp["latent_w"] or z.shape[3] > tp["latent_h"]


# ------------------------------------------------------------------------------------------------ shapes / random weights
COGVIDEOX_VAE_CFG = dict(in_channels=3, out_channels=3, block_out_channels=(128, 256, 256, 512), latent_channels=16,
                         layers_per_block=3, norm_eps=1e-6, norm_num_groups=32, temporal_compression_ratio=4,
                         scaling_factor=0.7, invert_scale_latents=False)


def cog_vae_param_shapes(cfg):
    ch = list(cfg["block_out_channels"])
    n = len(ch)
    zc = cfg["latent_channels"]
    s = {}

    def conv3(name, co, ci, k=3):
        s[name + ".conv.weight"] = (co, ci, k, k, k)
        s[name + ".conv.bias"] = (co,)

    def res(name, ci, co, zq):
        for nm, c in (("norm1", ci), ("norm2", co)):
            if zq:
                s[f"{name}.{nm}.norm_layer.weight"] = (c,)
                s[f"{name}.{nm}.norm_layer.bias"] = (c,)
                conv3(f"{name}.{nm}.conv_y", c, zc, 1)
                conv3(f"{name}.{nm}.conv_b", c, zc, 1)
            else:
                s[f"{name}.{nm}.weight"] = (c,)
                s[f"{name}.{nm}.bias"] = (c,)
        conv3(name + ".conv1", co, ci)
        conv3(name + ".conv2", co, co)
        if ci != co:
            s[name + ".conv_shortcut.weight"] = (co, ci, 1, 1, 1)
            s[name + ".conv_shortcut.bias"] = (co,)

    conv3("encoder.conv_in", ch[0], cfg["in_channels"])
    ci = ch[0]
    for i in range(n):
        for r in range(cfg["layers_per_block"]):
            res(f"encoder.down_blocks.{i}.resnets.{r}", ci, ch[i], False)
            ci = ch[i]
        if i != n - 1:
            s[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"] = (ch[i], ch[i], 3, 3)
            s[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"] = (ch[i],)
    for r in range(2):
        res(f"encoder.mid_block.resnets.{r}", ch[-1], ch[-1], False)
    s["encoder.norm_out.weight"] = (ch[-1],)
    s["encoder.norm_out.bias"] = (ch[-1],)
    conv3("encoder.conv_out", 2 * zc, ch[-1])
    rev = ch[::-1]
    conv3("decoder.conv_in", rev[0], zc)
    for r in range(2):
        res(f"decoder.mid_block.resnets.{r}", rev[0], rev[0], True)
    ci = rev[0]
    for i in range(n):
        for r in range(cfg["layers_per_block"] + 1):
            res(f"decoder.up_blocks.{i}.resnets.{r}", ci, rev[i], True)
            ci = rev[i]
        if i != n - 1:
            s[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"] = (rev[i], rev[i], 3, 3)
            s[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"] = (rev[i],)
    s["decoder.norm_out.norm_layer.weight"] = (rev[-1],)
    s["decoder.norm_out.norm_layer.bias"] = (rev[-1],)
    conv3("decoder.norm_out.conv_y", rev[-1], zc, 1)
    conv3("decoder.norm_out.conv_b", rev[-1], zc, 1)
    conv3("decoder.conv_out", cfg["out_channels"], rev[-1])
    return s


def cog_vae_random_state_dict(cfg, seed=0, device="cpu"):
    g = torch.Generator(device=device).manual_seed(seed)
    sd = {}
    for k, shp in cog_vae_param_shapes(cfg).items():
        if k.endswith("bias"):
            sd[k] = 0.02 * torch.randn(shp, generator=g, device=device)
        elif len(shp) == 1:                                       # norm gains
            sd[k] = 1.0 + 0.1 * torch.randn(shp, generator=g, device=device)
        elif ".conv_y." in k:                                     # multiplicative modulation ~ 1
            fan = shp[1] * shp[2] * shp[3] * shp[4]
            sd[k] = torch.randn(shp, generator=g, device=device) * (0.3 / fan ** 0.5)
        else:
            fan = 1
            for d in shp[1:]:
                fan *= d
            sd[k] = torch.randn(shp, generator=g, device=device) / fan ** 0.5
    for k in list(sd):
        if ".conv_y.conv.bias" in k:
            sd[k] = 1.0 + sd[k]
    return sd
