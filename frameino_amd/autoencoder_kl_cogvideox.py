"""MI355X-native CogVideoX 3D causal VAE -- the `vae` argument of the CogVideoX FrameINO pipelines
(/root/reference/pipelines/pipeline_cogvideox_i2v_motion_FrameINO.py:380-396 `vae.encode(...).latent_dist.sample()`,
:426-431 `vae.decode(z).sample`, :809-826 trajectory / identity encodes; train_code/train_cogvideox_motion_FrameINO.py
:493-546).  The class the reference loads is diffusers' `AutoencoderKLCogVideoX`: third-party, no source in the
reference tree -- restated from its published structure (oracle/cog_vae.py lists what), **parity unpinned**.

Same surface: `.encode(x).latent_dist.sample(generator) / .mode()`, `.decode(z).sample`, `.config.scaling_factor`,
`.config.invert_scale_latents`, `.dtype`, diffusers' parameter names (`load_reference_state_dict`).

MI355X design: channels-last activations [T, H, W, Cpad]; every convolution (3x3x3 causal, the 3x3 Conv2d of the
down / up samplers with the zero pad or the nearest 2x upsample folded into the gather, 1x1x1 shortcuts) is the
implicit-GEMM MFMA kernel `fino_conv3d` with the residual add as its epilogue; GroupNorm (+ the SpatialNorm3D
modulation, evaluated once at latent resolution and read through the nearest-neighbour map, + SiLU) is one fused
HBM-bound kernel chain (`fino_groupnorm_cl`).  The frame batching of diffusers' `_encode` / `_decode` (8 sample / 2
latent frames per batch, conv caches carried across) is kept: GroupNorm statistics are per batch there, so the schedule
is part of the result -- and it bounds the activation memory to one batch."""
import math
from types import SimpleNamespace

import torch

from . import ops
from .loading import FromPretrainedMixin
from .autoencoder_kl_wan import DiagonalGaussianDistribution, _Config, cpad


def cog_vae_param_shapes(cfg):
    """diffusers' parameter names and shapes."""
    ch = list(cfg["block_out_channels"])
    n, zc = len(ch), cfg["latent_channels"]
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


def frame_batches(num_frames, batch):
    """the slices diffusers' `_encode` / `_decode` walk: the remainder rides in the first batch"""
    nb = max(num_frames // batch, 1)
    rem = num_frames % batch
    return [(batch * i + (0 if i == 0 else rem), batch * (i + 1) + rem) for i in range(nb)]


class AutoencoderKLCogVideoX(FromPretrainedMixin):
    _loader_name = "load_cogvideox_vae"

    num_latent_frames_batch_size = 2
    num_sample_frames_batch_size = 8

    def __init__(self, in_channels=3, out_channels=3, block_out_channels=(128, 256, 256, 512), latent_channels=16,
                 layers_per_block=3, norm_eps=1e-6, norm_num_groups=32, temporal_compression_ratio=4,
                 scaling_factor=0.7, invert_scale_latents=False, sample_height=480, sample_width=720, **unused):
        self.config = _Config(in_channels=in_channels, out_channels=out_channels,
                              block_out_channels=tuple(block_out_channels), latent_channels=latent_channels,
                              layers_per_block=layers_per_block, norm_eps=norm_eps, norm_num_groups=norm_num_groups,
                              temporal_compression_ratio=temporal_compression_ratio, scaling_factor=scaling_factor,
                              invert_scale_latents=invert_scale_latents, sample_height=sample_height,
                              sample_width=sample_width)
        self.use_slicing = self.use_tiling = False
        # diffusers' tiling bookkeeping (AutoencoderKLCogVideoX.__init__): minimum tile = half the sample size of the config
        self.tile_sample_min_height = sample_height // 2
        self.tile_sample_min_width = sample_width // 2
        self.tile_overlap_factor_height = 1 / 6
        self.tile_overlap_factor_width = 1 / 5
        self._tile_latent()
        for c in block_out_channels:
            if cpad(c) & (cpad(c) - 1):
                raise ValueError("block_out_channels must pad to powers of two (GroupNorm kernel layout)")
        self._sd, self._pk = None, None
        self._dtype, self._device = torch.bfloat16, torch.device("cpu")
        self._tlevel = int(round(math.log2(temporal_compression_ratio)))

    # ---- module-like surface ----
    @property
    def dtype(self):
        return self._dtype

    @property
    def device(self):
        return self._device

    def eval(self):
        return self

    # ---- diffusers' memory switches (test_code/run_cogvideox_FrameIn_mass_evaluation.py:95-96) ----
    def enable_slicing(self):
        """accepted: one video of the batch at a time is what this mirror does anyway (same results)"""
        self.use_slicing = True

    def disable_slicing(self):
        self.use_slicing = False

    def _tile_latent(self):
        down = 2 ** (len(self.config.block_out_channels) - 1)
        self.tile_latent_min_height = int(self.tile_sample_min_height / down)
        self.tile_latent_min_width = int(self.tile_sample_min_width / down)

    def enable_tiling(self, tile_sample_min_height=None, tile_sample_min_width=None, tile_overlap_factor_height=None,
                      tile_overlap_factor_width=None):
        """diffusers' `enable_tiling` (what the canonical caller switches on: reference
        test_code/run_cogvideox_FrameIn_mass_evaluation.py:95-96): frames larger than a tile -- half the config's sample size,
        240 x 360 -- are encoded / decoded as OVERLAPPING spatial tiles, each walked in the frame batches with its own conv
        caches, and blended over the overlap (`tiled_encode` / `tiled_decode`; overlap factors 1/6 and 1/5).  The result
        differs from the un-tiled one (GroupNorm statistics are per tile, the overlap is a linear cross-fade), and at the
        default 480 x 720 tiling is ACTIVE -- so a drop-in must reproduce it: round 5 implements it on the HIP convolutions
        (fino_vae_blend_tiles) against the oracle's restatement (oracle/cog_vae.py; third-party: parity unpinned)."""
        self.use_tiling = True
        self.tile_sample_min_height = tile_sample_min_height or self.tile_sample_min_height
        self.tile_sample_min_width = tile_sample_min_width or self.tile_sample_min_width
        self.tile_overlap_factor_height = tile_overlap_factor_height or self.tile_overlap_factor_height
        self.tile_overlap_factor_width = tile_overlap_factor_width or self.tile_overlap_factor_width
        self._tile_latent()

    def disable_tiling(self):
        self.use_tiling = False

    def to(self, device=None, dtype=None):
        if isinstance(device, torch.dtype):
            device, dtype = None, device
        if dtype is not None:
            self._dtype = dtype
        if device is not None:
            self._device = torch.device(device)
        if self._sd is not None:
            self._sd = {k: v.to(self._device) for k, v in self._sd.items()}
        self._pk = None
        return self

    def state_dict(self):
        return dict(self._sd)

    def load_reference_state_dict(self, sd, dtype=torch.bfloat16):
        shapes = cog_vae_param_shapes(self.config)
        missing = [k for k in shapes if k not in sd]
        if missing:
            raise KeyError(f"CogVideoX VAE state-dict is missing {missing[:5]}")
        for k, shp in shapes.items():
            if tuple(sd[k].shape) != tuple(shp):
                raise ValueError(f"{k}: expected {shp}, got {tuple(sd[k].shape)}")
        self._sd = {k: sd[k].detach().to(self._device).float() for k in shapes}
        self._dtype, self._pk = dtype, None
        return self

    def random_init_(self, seed=0, device=None, dtype=torch.bfloat16):
        if device is not None:
            self._device = torch.device(device)
        g = torch.Generator(device=self._device).manual_seed(seed)
        sd = {}
        for k, shp in cog_vae_param_shapes(self.config).items():
            if k.endswith("bias"):
                t = 0.02 * torch.randn(shp, generator=g, device=self._device)
                sd[k] = 1.0 + t if ".conv_y." in k else t
            elif len(shp) == 1:
                sd[k] = 1.0 + 0.1 * torch.randn(shp, generator=g, device=self._device)
            else:
                fan = math.prod(shp[1:])
                sd[k] = torch.randn(shp, generator=g, device=self._device) * ((0.3 if ".conv_y." in k else 1.0) / fan ** 0.5)
        self._sd, self._dtype, self._pk = sd, dtype, None
        return self

    # ---- packing ----
    def _pack(self):
        sd, dt, pk = self._sd, self._dtype, {}
        for k in sd:
            if not k.endswith(".weight"):
                continue
            n = k[:-7]
            w = sd[k]
            if w.dim() == 1:                                             # GroupNorm affine
                c = w.numel()
                g = torch.zeros(cpad(c), device=w.device)
                b = torch.zeros(cpad(c), device=w.device)
                g[:c], b[:c] = w, sd[n + ".bias"]
                pk[n] = SimpleNamespace(g=g, b=b, c=c)
                continue
            if w.dim() == 4:
                w = w.unsqueeze(2)                                       # Conv2d -> kt = 1
            co, ci, kt, kh, kw = w.shape
            cip, cop = cpad(ci), cpad(co)
            w2 = torch.zeros(cop, kt * kh * kw, cip, device=w.device)
            w2[:co, :, :ci] = w.permute(0, 2, 3, 4, 1).reshape(co, kt * kh * kw, ci)
            b2 = torch.zeros(cop, device=w.device)
            b2[:co] = sd[n + ".bias"]
            pk[n] = SimpleNamespace(w=w2.reshape(cop, -1).to(dt).contiguous(), b=b2.to(dt), k=(kt, kh, kw), co=co,
                                    cop=cop, cip=cip)
        self._pk = pk
        return pk

    # ---- layers (channels-last [T, H, W, Cpad]) ----
    def _conv3(self, x, name, caches, new, residual=None):
        """CogVideoXCausalConv3d: (k_t - 1) copies of the first frame -- or the previous batch's last frames -- in front."""
        e = self._pk[name + ".conv"]
        kt = e.k[0]
        if kt > 1:
            front = caches.get(name)
            if front is None:
                front = x[:1].expand(kt - 1, -1, -1, -1)
            xin = torch.cat([front, x], dim=0)
            new[name] = xin[-(kt - 1):].clone()
        else:
            xin = x
        return ops.conv3d_cl(xin, e.w, e.b, e.k, (1, 1, 1), (0, e.k[1] // 2, e.k[2] // 2), residual=residual)

    def _norm(self, x, name, zmod, silu=True):
        e = self._pk[name + ".norm_layer"] if zmod is not None else self._pk[name]
        mod = None
        if zmod is not None:
            z, zshape = zmod
            ey, eb = self._pk[name + ".conv_y.conv"], self._pk[name + ".conv_b.conv"]
            rows = z.view(-1, z.shape[-1])
            mod = (ops.gemm(rows, ey.w, ey.b).view(*zshape, ey.cop), ops.gemm(rows, eb.w, eb.b).view(*zshape, eb.cop))
        return ops.groupnorm_cl(x, e.c, self.config.norm_num_groups, e.g, e.b, self.config.norm_eps, mod, silu)

    def _res(self, x, p, zmod, caches, new):
        h = self._norm(x, p + ".norm1", zmod)
        h = self._conv3(h, p + ".conv1", caches, new)
        h = self._norm(h, p + ".norm2", zmod)
        sc = x
        if (p + ".conv_shortcut") in self._pk:
            e = self._pk[p + ".conv_shortcut"]
            sc = ops.conv3d_cl(x, e.w, e.b, e.k)
        return self._conv3(h, p + ".conv2", caches, new, residual=sc)

    def _down(self, x, p, compress_time):
        if compress_time and x.shape[0] > 1:
            x = ops.avg_pool_time2(x)
        e = self._pk[p + ".conv"]
        t, h, w, _ = x.shape
        # F.pad(0, 1, 0, 1) + Conv2d(3, stride 2): taps past the bottom / right edge read zeros
        return ops.conv3d_cl(x, e.w, e.b, e.k, (1, 2, 2), (0, 0, 0), out_thw=(t, (h + 1 - 3) // 2 + 1, (w + 1 - 3) // 2 + 1))

    def _up(self, x, p, compress_time):
        e = self._pk[p + ".conv"]
        y = ops.conv3d_cl(x, e.w, e.b, e.k, (1, 1, 1), (0, 1, 1), upsample2x=True)       # nearest 2x folded into the gather
        t = x.shape[0]
        if compress_time and t > 1:
            # the Conv2d is per frame, so the temporal nearest upsample is a duplication of ITS output frames:
            # every frame twice, except the first frame of an odd-length batch
            idx = torch.arange(t, device=x.device).repeat_interleave(2)
            if t % 2 == 1:
                idx = idx[1:]
            y = y.index_select(0, idx)
        return y

    # ---- encoder / decoder over one frame batch ----
    def _encoder(self, x, caches):
        new, cfg = {}, self.config
        n = len(cfg.block_out_channels)
        h = self._conv3(x, "encoder.conv_in", caches, new)
        for i in range(n):
            for r in range(cfg.layers_per_block):
                h = self._res(h, f"encoder.down_blocks.{i}.resnets.{r}", None, caches, new)
            if i != n - 1:
                h = self._down(h, f"encoder.down_blocks.{i}.downsamplers.0", i < self._tlevel)
        for r in range(2):
            h = self._res(h, f"encoder.mid_block.resnets.{r}", None, caches, new)
        h = self._norm(h, "encoder.norm_out", None)
        return self._conv3(h, "encoder.conv_out", caches, new), new

    def _decoder(self, z, caches):
        new, cfg = {}, self.config
        n = len(cfg.block_out_channels)
        zmod = (z, tuple(z.shape[:3]))
        h = self._conv3(z, "decoder.conv_in", caches, new)
        for r in range(2):
            h = self._res(h, f"decoder.mid_block.resnets.{r}", zmod, caches, new)
        for i in range(n):
            for r in range(cfg.layers_per_block + 1):
                h = self._res(h, f"decoder.up_blocks.{i}.resnets.{r}", zmod, caches, new)
            if i != n - 1:
                h = self._up(h, f"decoder.up_blocks.{i}.upsamplers.0", i < self._tlevel)
        h = self._norm(h, "decoder.norm_out", zmod)
        return self._conv3(h, "decoder.conv_out", caches, new), new

    def _to_cl(self, x, c):
        """[1, C, T, H, W] -> channels-last [T, H, W, cpad(C)] of the VAE dtype"""
        _, _, t, h, w = x.shape
        y = torch.zeros(t, h, w, cpad(c), dtype=self._dtype, device=x.device)
        y[..., :c] = x[0].permute(1, 2, 3, 0).to(self._dtype)
        return y

    def _tiles(self, xs, tile_h, tile_w, stride_h, stride_w, batch, run):
        """rows of tiles as tiled_encode / tiled_decode build them: tile (i, j) = frames walked in `batch`-frame batches (own conv
        caches) over xs[:, i : i + tile_h, j : j + tile_w]; channels-last [T', h', w', Cpad] each"""
        rows = []
        for i in range(0, xs.shape[1], stride_h):
            row = []
            for j in range(0, xs.shape[2], stride_w):
                caches, outs = {}, []
                for s, e in frame_batches(xs.shape[0], batch):
                    y, caches = run(xs[s:e, i:i + tile_h, j:j + tile_w].contiguous(), caches)
                    outs.append(y)
                row.append(torch.cat(outs, dim=0))
            rows.append(row)
        return rows

    @staticmethod
    def _blend(rows, extent_h, extent_w, limit_h, limit_w):
        """every tile blended IN PLACE with its (already blended) upper and left neighbour, cropped to the stride, concatenated"""
        result_rows = []
        for i, row in enumerate(rows):
            result_row = []
            for j, tile in enumerate(row):
                if i > 0 and extent_h > 0:
                    ops.vae_blend_tiles_(rows[i - 1][j], tile, extent_h, 0)
                if j > 0 and extent_w > 0:
                    ops.vae_blend_tiles_(row[j - 1], tile, extent_w, 1)
                result_row.append(tile[:, :limit_h, :limit_w])
            result_rows.append(torch.cat(result_row, dim=2))
        return torch.cat(result_rows, dim=1)

    @torch.no_grad()
    def _encode(self, x):
        self._pk or self._pack()
        cfg = self.config
        xs = self._to_cl(x, cfg.in_channels)
        if self.use_tiling and (xs.shape[2] > self.tile_sample_min_width or xs.shape[1] > self.tile_sample_min_height):
            th, tw = self.tile_sample_min_height, self.tile_sample_min_width
            fh, fw = self.tile_overlap_factor_height, self.tile_overlap_factor_width
            eh, ew = int(self.tile_latent_min_height * fh), int(self.tile_latent_min_width * fw)
            rows = self._tiles(xs, th, tw, int(th * (1 - fh)), int(tw * (1 - fw)), self.num_sample_frames_batch_size, self._encoder)
            m = self._blend(rows, eh, ew, self.tile_latent_min_height - eh, self.tile_latent_min_width - ew)
            return m[..., :2 * cfg.latent_channels].permute(3, 0, 1, 2)[None].contiguous()
        caches, outs = {}, []
        for s, e in frame_batches(xs.shape[0], self.num_sample_frames_batch_size):
            y, caches = self._encoder(xs[s:e].contiguous(), caches)
            outs.append(y[..., :2 * cfg.latent_channels])
        m = torch.cat(outs, dim=0)
        return m.permute(3, 0, 1, 2)[None].contiguous()                      # [1, 2 * latent, T', h, w] of the VAE dtype

    def encode(self, x, return_dict=True):
        moments = torch.cat([self._encode(x[i:i + 1]) for i in range(x.shape[0])])
        post = DiagonalGaussianDistribution(moments)
        return (post,) if not return_dict else SimpleNamespace(latent_dist=post)

    @torch.no_grad()
    def _decode(self, z):
        self._pk or self._pack()
        cfg = self.config
        zs = self._to_cl(z, cfg.latent_channels)
        if self.use_tiling and (zs.shape[2] > self.tile_latent_min_width or zs.shape[1] > self.tile_latent_min_height):
            th, tw = self.tile_latent_min_height, self.tile_latent_min_width
            fh, fw = self.tile_overlap_factor_height, self.tile_overlap_factor_width
            eh, ew = int(self.tile_sample_min_height * fh), int(self.tile_sample_min_width * fw)
            rows = self._tiles(zs, th, tw, int(th * (1 - fh)), int(tw * (1 - fw)), self.num_latent_frames_batch_size, self._decoder)
            v = self._blend(rows, eh, ew, self.tile_sample_min_height - eh, self.tile_sample_min_width - ew)
            return v[..., :cfg.out_channels].permute(3, 0, 1, 2)[None].contiguous()
        caches, outs = {}, []
        for s, e in frame_batches(zs.shape[0], self.num_latent_frames_batch_size):
            y, caches = self._decoder(zs[s:e].contiguous(), caches)
            outs.append(y[..., :cfg.out_channels])
        v = torch.cat(outs, dim=0)
        return v.permute(3, 0, 1, 2)[None].contiguous()                      # [1, 3, T, H, W]

    def decode(self, z, return_dict=True):
        out = torch.cat([self._decode(z[i:i + 1]) for i in range(z.shape[0])])
        return (out,) if not return_dict else SimpleNamespace(sample=out)
