"""MI355X-native Wan 3D causal VAE (Wan2.2 residual variant) -- mirror of
/root/reference/architecture/autoencoder_kl_wan.py::AutoencoderKLWan for the calls the FrameINO pipeline makes:
`vae.encode(x).latent_dist.mode()/sample()`, `vae.decode(z, return_dict=False)[0]`, `vae.config.*`, `vae.dtype`.

MI355X design:
  * whole-sequence execution: the reference streams the time axis in chunks through 34 (decoder) / 26 (encoder)
    `feat_cache` slots because a 24-80 GB GPU cannot hold the activations; every temporal conv is causal, so chunking
    is only a schedule.  With 288 GB of HBM each layer runs ONCE over all frames (rules in oracle/wan_vae.py, checked
    against the reference's streaming run): M = T.H.W output positions per conv launch, no cache bookkeeping.
  * channels-last activations [T, H, W, Cpad] (Cpad = multiple of 64): every conv -- 3x3x3 causal, (3,1,1) time
    convs, the 3x3 Conv2d of WanResample with its nearest-exact 2x upsample folded into the gather, 1x1x1 shortcuts --
    is an implicit GEMM on the MFMA GEMM kernel (fino_conv3d), the residual add is its epilogue.
  * RMS-norm+SiLU, DupUp3D/AvgDown3D shortcuts, patchify/unpatchify+clamp are fused HBM-bound kernels.
Weights are stored under the reference's parameter names (`load_reference_state_dict`), packed once for the kernels.
"""
import math
import warnings
from types import SimpleNamespace

import torch

from . import ops
from .loading import FromPretrainedMixin


def cpad(c):
    return (c + 63) // 64 * 64


class _Config(dict):
    __getattr__ = dict.__getitem__


class _Planes:
    """an activation already cut into the bf16 planes a split-bf16 convolution reads ([T, H, W, planes * Cpad], hi | mid | lo
    side by side): what the fp32-compute mode's norm kernel emits, so that the convolution behind it needs no split pass"""
    __slots__ = ("t",)

    def __init__(self, t):
        self.t = t


class DiagonalGaussianDistribution:
    """diffusers' posterior object for the two calls the pipelines make (mode / sample)."""

    def __init__(self, parameters):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def mode(self):
        return self.mean

    def sample(self, generator=None):
        gdev = generator.device if generator is not None else self.mean.device
        noise = torch.randn(self.mean.shape, generator=generator, device=gdev, dtype=self.mean.dtype)
        return self.mean + self.std * noise.to(self.mean.device)


def wan_vae_param_shapes(cfg):
    """Parameter names and shapes of the reference module tree (is_residual=True)."""
    mult = list(cfg["dim_mult"])
    enc = [cfg["base_dim"] * u for u in [1] + mult]
    dec = [cfg["decoder_base_dim"] * u for u in [mult[-1]] + mult[::-1]]
    z, nres = cfg["z_dim"], cfg["num_res_blocks"]
    tdown = list(cfg["temperal_downsample"])
    tup = tdown[::-1]
    s = {}

    def conv(name, co, ci, k):
        s[name + ".weight"] = (co, ci) + tuple(k)
        s[name + ".bias"] = (co,)

    def res(name, ci, co):
        s[name + ".norm1.gamma"] = (ci, 1, 1, 1)
        conv(name + ".conv1", co, ci, (3, 3, 3))
        s[name + ".norm2.gamma"] = (co, 1, 1, 1)
        conv(name + ".conv2", co, co, (3, 3, 3))
        if ci != co:
            conv(name + ".conv_shortcut", co, ci, (1, 1, 1))

    def mid(name, c):
        res(name + ".resnets.0", c, c)
        s[name + ".attentions.0.norm.gamma"] = (c, 1, 1)
        conv(name + ".attentions.0.to_qkv", 3 * c, c, (1, 1))
        conv(name + ".attentions.0.proj", c, c, (1, 1))
        res(name + ".resnets.1", c, c)

    conv("encoder.conv_in", enc[0], cfg["in_channels"], (3, 3, 3))
    nb = len(mult)
    for i in range(nb):
        p = f"encoder.down_blocks.{i}"
        ci = enc[i]
        for r in range(nres):
            res(f"{p}.resnets.{r}", ci, enc[i + 1])
            ci = enc[i + 1]
        if i != nb - 1:
            conv(f"{p}.downsampler.resample.1", enc[i + 1], enc[i + 1], (3, 3))
            if tdown[i]:
                conv(f"{p}.downsampler.time_conv", enc[i + 1], enc[i + 1], (3, 1, 1))
    mid("encoder.mid_block", enc[-1])
    s["encoder.norm_out.gamma"] = (enc[-1], 1, 1, 1)
    conv("encoder.conv_out", 2 * z, enc[-1], (3, 3, 3))
    conv("quant_conv", 2 * z, 2 * z, (1, 1, 1))
    conv("post_quant_conv", z, z, (1, 1, 1))
    conv("decoder.conv_in", dec[0], z, (3, 3, 3))
    mid("decoder.mid_block", dec[0])
    for i in range(nb):
        p = f"decoder.up_blocks.{i}"
        ci = dec[i]
        for r in range(nres + 1):
            res(f"{p}.resnets.{r}", ci, dec[i + 1])
            ci = dec[i + 1]
        if i != nb - 1:
            conv(f"{p}.upsampler.resample.1", dec[i + 1], dec[i + 1], (3, 3))
            if tup[i]:
                conv(f"{p}.upsampler.time_conv", 2 * dec[i + 1], dec[i + 1], (3, 1, 1))
    s["decoder.norm_out.gamma"] = (dec[-1], 1, 1, 1)
    conv("decoder.conv_out", cfg["out_channels"], dec[-1], (3, 3, 3))
    return s


class AutoencoderKLWan(FromPretrainedMixin):
    _loader_name = "load_wan_vae"

    def __init__(self, base_dim=96, decoder_base_dim=None, z_dim=16, dim_mult=(1, 2, 4, 4), num_res_blocks=2,
                 attn_scales=(), temperal_downsample=(False, True, True), dropout=0.0, latents_mean=None,
                 latents_std=None, is_residual=False, in_channels=3, out_channels=3, patch_size=None,
                 scale_factor_temporal=4, scale_factor_spatial=8):
        if not is_residual:
            raise NotImplementedError("FrameINO's Wan path uses the Wan2.2 residual VAE (is_residual=True)")
        self.config = _Config(base_dim=base_dim, decoder_base_dim=decoder_base_dim or base_dim, z_dim=z_dim,
                              dim_mult=list(dim_mult), num_res_blocks=num_res_blocks, attn_scales=list(attn_scales),
                              temperal_downsample=list(temperal_downsample), dropout=dropout,
                              latents_mean=list(latents_mean or [0.0] * z_dim),
                              latents_std=list(latents_std or [1.0] * z_dim), is_residual=is_residual,
                              in_channels=in_channels, out_channels=out_channels, patch_size=patch_size,
                              scale_factor_temporal=scale_factor_temporal, scale_factor_spatial=scale_factor_spatial)
        self._sd = None
        self._pk = None
        self._dtype = torch.bfloat16           # MFMA operand dtype of the convolutions (bf16 | fp16)
        self._io_dtype = None                  # what `.dtype` reports when fp32 was asked for (reference app.py:157)
        self._planes = 0                       # 0: convolutions in `_dtype`; 2 | 3: fp32-compute mode on split-bf16 products
        self._span_limit = 1 << 31             # bytes a split convolution's gather offsets cover (tests lower it: strip path)
        self._device = torch.device("cpu")
        self.use_slicing = self.use_tiling = False
        self.tile_sample_min_height = self.tile_sample_min_width = 256            # reference :1070-1075 (recorded only)
        self.tile_sample_stride_height = self.tile_sample_stride_width = 192
        # frames per time chunk of the decoder's tail: results do not depend on it.  0 / None: whole sequence (fastest:
        # 0.73 s and 36 GiB for 49 f 704x1280), n: chunks of n frames (8: 0.80 s, 20 GiB), "auto": whole sequence while
        # its estimated activation peak stays under `decode_memory_budget_gib`, chunks of 8 above (1024x1792, 81 f).
        self.decode_chunk_frames = "auto"
        self.decode_memory_budget_gib = 48.0

    _warned_fp32 = False
    _warned_tiling = False

    # ---- module-like surface ----
    @property
    def dtype(self):
        return self._io_dtype or self._dtype

    @property
    def compute_dtype(self):
        """the precision the convolutions compute in: bf16 | fp16 (MFMA operands of that type, fp32 accumulation; differs from
        `.dtype` when fp32 was asked for), or float32 after `set_compute_dtype(torch.float32)`"""
        return torch.float32 if self._planes else self._dtype

    @compute_dtype.setter
    def compute_dtype(self, dtype):
        self.set_compute_dtype(dtype)

    def set_compute_dtype(self, dtype, planes=3):
        """`torch.float32`: compute like the fp32 the reference app runs this VAE in (app.py:157, decode :1198-1227).  Activations
        stay fp32 between layers, norms / SiLU / the mid-block softmax are fp32 arithmetic, and every convolution and matrix
        product is a SPLIT-BF16 product on the matrix pipe: each fp32 operand is three bf16 planes (hi + mid + lo, 24 significant
        bits), the six cross terms >= 2^-16 relative are accumulated in the MFMA's fp32 accumulator (fino_conv3d_split;
        include/frameino_hip.h has the arithmetic) -- fp32-faithful at 1/6 of the bf16 rate (the fp32-input MFMA: 1/16).
        `planes=2`: hi + lo, three terms, ~2^-16 relative, 1/3 of the bf16 rate.  `torch.bfloat16` / `torch.float16`: back to
        16-bit convolutions (the default).  The interface dtype (`.dtype`, inputs, outputs) is not touched.
        Measured (tests/test_fullsize_oracle_gpu.py, tests/test_wan_vae_gpu.py): see DESIGN.md section 4.8."""
        if dtype == torch.float32:
            if planes not in (2, 3):
                raise ValueError("planes must be 2 (hi + lo) or 3 (hi + mid + lo)")
            self._planes, self._dtype = int(planes), torch.bfloat16
        elif dtype in (torch.bfloat16, torch.float16):
            self._planes, self._dtype = 0, dtype
        else:
            raise ValueError(f"compute dtype {dtype}: float32, bfloat16 or float16")
        self._pk = None
        return self

    @property
    def device(self):
        return self._device

    def eval(self):
        return self

    def _set_dtype(self, dtype):
        """fp32 (the precision the reference app runs this VAE in, app.py:157) is accepted as the INTERFACE dtype: master
        weights stay fp32, inputs / outputs are fp32, `.dtype` says fp32 -- and the convolutions compute in bf16 with fp32
        accumulation (MFMA has no fp32 path worth the name: 157 TFLOP/s against 2.5 P)."""
        if dtype == torch.float32:
            self._io_dtype, self._dtype = torch.float32, torch.bfloat16
            if not AutoencoderKLWan._warned_fp32:
                AutoencoderKLWan._warned_fp32 = True
                warnings.warn("AutoencoderKLWan(torch_dtype=float32): inputs, outputs and `.dtype` are fp32 as asked "
                              "(reference app.py:157), but the convolutions compute in bf16 with fp32 accumulation "
                              "(`.compute_dtype`); measured against an fp32 decode at 704x1280: PSNR 50.5 dB, rel-RMS "
                              "1.1e-2 (tests/test_fullsize_oracle_gpu.py).  `vae.set_compute_dtype(torch.float32)` computes "
                              "like fp32 (split-bf16 products, ~6x the decode time)", RuntimeWarning, stacklevel=3)
        else:
            self._io_dtype, self._dtype = None, dtype
            self._planes = 0

    # ---- diffusers' memory switches (architecture/autoencoder_kl_wan.py:1084-1133) ----
    def enable_slicing(self):
        """accepted: this mirror already encodes / decodes one video of the batch at a time (same results)"""
        self.use_slicing = True

    def disable_slicing(self):
        self.use_slicing = False

    def enable_tiling(self, tile_sample_min_height=None, tile_sample_min_width=None, tile_sample_stride_height=None,
                      tile_sample_stride_width=None):
        """The reference's tiling (:1270-1397) cuts the frame into overlapping spatial tiles and BLENDS them.  On the VAE
        FrameINO's Wan path uses -- Wan2.2: `patch_size=2`, `is_residual=True`, the only configuration this mirror builds
        -- those reference paths do not produce a result at all: `tiled_encode` feeds un-patchified 3-channel tiles to an
        encoder whose conv_in expects 12 (:1304-1316, RuntimeError), `tiled_decode` runs the decoder without
        `first_chunk=True`, so the residual up-blocks' shortcut and main path disagree on the frame count (:1376,
        RuntimeError), and would skip unpatchify and clamp.  Verified by running the reference (tools/golden/
        probe_vae_tiling.py, profiles/r04_ref_vae_tiling_probe.txt; tests/test_reference_plugin_cpu.py repeats it in the
        build container).  There is therefore no reference output to match: this mirror returns the UN-TILED result, says
        so once, and turns on the memory saver it does have, which is result-IDENTICAL: the decoder's tail in time chunks
        of 8 frames with the causal convs' two carried frames (`enable_time_chunking`; tests/test_fullsize_gpu.py:
        torch.equal to the whole-sequence decode, 36 -> 20 GiB at 49 f 704x1280).  The tile arguments are recorded."""
        self.use_tiling = True
        self.tile_sample_min_height = tile_sample_min_height or self.tile_sample_min_height
        self.tile_sample_min_width = tile_sample_min_width or self.tile_sample_min_width
        self.tile_sample_stride_height = tile_sample_stride_height or self.tile_sample_stride_height
        self.tile_sample_stride_width = tile_sample_stride_width or self.tile_sample_stride_width
        if not AutoencoderKLWan._warned_tiling:
            AutoencoderKLWan._warned_tiling = True
            warnings.warn("AutoencoderKLWan.enable_tiling(): the reference's tiled_encode / tiled_decode raise on the "
                          "Wan2.2 VAE (patch_size=2, is_residual=True), so there is no tiled reference result; this "
                          "mirror returns the UN-TILED encode / decode and switches on its result-identical memory saver "
                          "(time-chunked decoder tail, `enable_time_chunking`)", RuntimeWarning, stacklevel=2)
        self.enable_time_chunking(8)

    def disable_tiling(self):
        self.use_tiling = False
        self.disable_time_chunking()

    def enable_time_chunking(self, frames=8):
        """the decoder's tail (everything after the last temporal upsampling) in chunks of `frames` frames: bit-identical
        output, activation memory of one chunk (36 -> 20 GiB at 49 f 704x1280 for +9 % time)"""
        self.decode_chunk_frames = int(frames)

    def disable_time_chunking(self):
        self.decode_chunk_frames = "auto"

    def to(self, device=None, dtype=None):
        if isinstance(device, torch.dtype):
            device, dtype = None, device
        if dtype is not None:
            self._set_dtype(dtype)
        if device is not None:
            self._device = torch.device(device)
        if self._sd is not None:
            self._sd = {k: v.to(self._device) for k, v in self._sd.items()}
            self._pk = None
        return self

    def state_dict(self):
        return dict(self._sd)

    def load_reference_state_dict(self, sd, dtype=torch.bfloat16):
        shapes = wan_vae_param_shapes(self.config)
        missing = [k for k in shapes if k not in sd]
        if missing:
            raise KeyError(f"VAE state-dict is missing {missing[:5]}")
        for k, shp in shapes.items():
            if tuple(sd[k].shape) != tuple(shp):
                raise ValueError(f"{k}: expected {shp}, got {tuple(sd[k].shape)}")
        self._sd = {k: sd[k].detach().to(self._device).float() for k in shapes}
        self._set_dtype(dtype)
        self._pk = None
        return self

    def random_init_(self, seed=0, std=0.03, device=None, dtype=torch.bfloat16):
        """Seeded random weights of the right shapes (no checkpoints offline)."""
        if device is not None:
            self._device = torch.device(device)
        g = torch.Generator(device=self._device).manual_seed(seed)
        sd = {}
        for k, shp in wan_vae_param_shapes(self.config).items():
            if k.endswith("gamma"):
                sd[k] = 1.0 + 0.05 * torch.randn(shp, generator=g, device=self._device)
            elif k.endswith("bias"):
                sd[k] = 0.01 * torch.randn(shp, generator=g, device=self._device)
            else:
                fan = math.prod(shp[1:])
                sd[k] = torch.randn(shp, generator=g, device=self._device) * (1.0 / math.sqrt(fan))
        self._sd, self._pk = sd, None
        self._set_dtype(dtype)
        return self

    # ---- packing ----
    def _pack(self):
        sd, dt = self._sd, self._dtype
        pk = {}

        def pack_conv(name, halves=1):
            w = sd[name + ".weight"]
            b = sd[name + ".bias"]
            if w.dim() == 4:
                w = w.unsqueeze(2)                           # Conv2d -> kt = 1
            co, ci, kt, kh, kw = w.shape
            cip = cpad(ci)
            coh = co // halves
            cop = cpad(coh)
            w2 = torch.zeros(halves * cop, kt * kh * kw, cip, device=w.device, dtype=torch.float32)
            b2 = torch.zeros(halves * cop, device=w.device, dtype=torch.float32)
            wt = w.permute(0, 2, 3, 4, 1).reshape(co, kt * kh * kw, ci)      # [co, tap, ci], tap = (dt*kh+dh)*kw+dw
            for s in range(halves):
                w2[s * cop:s * cop + coh, :, :ci] = wt[s * coh:(s + 1) * coh]
                b2[s * cop:s * cop + coh] = b[s * coh:(s + 1) * coh]
            if self._planes:
                # fp32-compute mode: one bf16 weight plane per PRODUCT of the split expansion, per tap ([Cout, tap, product,
                # Cin]: the order the convolution's K walk visits, ops.SPLIT_PRODUCTS); bias stays fp32.  1 x 1 layers used as
                # plain matrix products (the mid-block attention) also keep the A-side expansion of their weight.
                w6 = ops.split_bf16(w2.reshape(halves * cop * kt * kh * kw, cip).contiguous(), "W", self._planes)
                e = SimpleNamespace(w=w6.reshape(halves * cop, -1), b=b2.contiguous(), k=(kt, kh, kw), ci=ci, co=coh, cip=cip,
                                    cop=cop)
                if kt * kh * kw == 1:
                    e.w_a = ops.split_bf16(w2.reshape(halves * cop, cip).contiguous(), "A", self._planes)
                pk[name] = e
                return
            pk[name] = SimpleNamespace(w=w2.reshape(halves * cop, -1).to(dt).contiguous(), b=b2.to(dt), k=(kt, kh, kw),
                                       ci=ci, co=coh, cip=cip, cop=cop)

        def pack_gamma(name):
            g = sd[name].reshape(-1)
            out = torch.zeros(cpad(g.numel()), device=g.device, dtype=torch.float32)
            out[:g.numel()] = g
            pk[name] = SimpleNamespace(g=out, c=g.numel())

        for k in sd:
            if k.endswith(".weight"):
                n = k[:-7]
                if n.endswith("upsampler.time_conv"):
                    pack_conv(n, halves=2)
                elif n.endswith("to_qkv"):
                    pack_conv(n, halves=3)
                else:
                    pack_conv(n)
            elif k.endswith("gamma"):
                pack_gamma(k)
        self._pk = pk
        return pk

    # ---- layer helpers (all on channels-last [T, H, W, Cpad]) ----
    def _conv(self, x, name, pad, stride=(1, 1, 1), up=False, residual=None, out_thw=None):
        e = self._pk[name]
        if self._planes:
            xp = x.t if isinstance(x, _Planes) else ops.split_bf16(x, "planes", self._planes)
            t, h, w, cw = xp.shape
            kt, kh, kw = e.k
            # The convolution kernel gathers with 32-bit byte offsets from the first input frame a 256-row tile can touch:
            # (2 stride_t + k_t) frames of `planes` planes must stay under 2 GiB.  ONE layer of the 704 x 1280 decode is over
            # (up_blocks.3.resnets.0.conv1: 512 channels x 3 planes at 352 x 640 = 692 MB per frame): such a stride-1 "same"
            # convolution runs as horizontal strips with one halo row each side -- every output row sees the same taps.
            lim = self._span_limit
            span = (((255 + h * w - 1) // (h * w) + 1) * stride[0] + kt) * h * w * cw * 2        # (fino_conv3d_split's own formula)
            if span >= lim and tuple(stride) == (1, 1, 1) and not up and out_thw is None and pad[1] == kh // 2 \
                    and pad[2] == kw // 2 and h >= 8:
                # the strip count from the kernel's own span formula evaluated on a STRIP (its rows + the kh - 1 halo rows; the
                # ceil((255 + h w - 1) / (h w)) term grows once a strip's h w drops under 256): the first n that fits
                def strip_span(hs):
                    return (((255 + hs * w - 1) // (hs * w) + 1) * stride[0] + kt) * hs * w * cw * 2
                n = 2
                while n < h and strip_span(-(-h // n) + kh - 1) >= lim:
                    n += 1
                rows = -(-h // n)
                outs = []
                for r0 in range(0, h, rows):
                    r1 = min(h, r0 + rows)
                    a0, a1 = max(0, r0 - pad[1]), min(h, r1 + (kh - 1 - pad[1]))
                    rs = None if residual is None else residual[:, a0:a1].contiguous()
                    ys = ops.conv3d_split_cl(xp[:, a0:a1].contiguous(), e.w, e.b, e.k, self._planes, stride, pad, None, False, rs)
                    outs.append(ys[:, r0 - a0:r0 - a0 + (r1 - r0)])
                return torch.cat(outs, dim=1)
            return ops.conv3d_split_cl(xp, e.w, e.b, e.k, self._planes, stride, pad, out_thw, up, residual)
        return ops.conv3d_cl(x, e.w, e.b, e.k, stride, pad, out_thw, up, residual)

    def _causal3(self, x, name, residual=None, caches=None):
        """WanCausalConv3d over the whole sequence (kt - 1 zero frames in front) or, with `caches` (time-chunked tail
        of the decoder), over one chunk whose history is the last kt - 1 input frames of the previous chunk: the same
        arithmetic per output element either way (the reference streams exactly like this, :350-358)."""
        e = self._pk[name]
        kt, kh, kw = e.k
        if caches is None or kt == 1:
            return self._conv(x, name, (kt - 1, kh // 2, kw // 2), residual=residual)
        if self._planes and not isinstance(x, _Planes):          # (the history is kept as the planes the convolution reads)
            x = _Planes(ops.split_bf16(x, "planes", self._planes))
        xt = x.t if isinstance(x, _Planes) else x
        prev = caches.get(name)
        if prev is None:                               # first chunk: the history is zeros (same as the front padding)
            prev = torch.zeros((kt - 1,) + tuple(xt.shape[1:]), dtype=xt.dtype, device=xt.device)
        xin = torch.cat([prev, xt], dim=0)
        caches[name] = xin[-(kt - 1):].clone()
        return self._conv(_Planes(xin) if isinstance(x, _Planes) else xin, name, (0, kh // 2, kw // 2), residual=residual)

    def _norm(self, x, name, silu=True, split="planes"):
        e = self._pk[name]
        if self._planes:
            y = ops.rmsnorm_silu_cl_f32(x, e.g, e.c, silu, split, self._planes)
            return _Planes(y) if split == "planes" else y
        return ops.rmsnorm_silu_cl(x, e.g, e.c, silu)

    def _res(self, x, p, caches=None):
        h = self._causal3(x, p + ".conv_shortcut") if (p + ".conv_shortcut") in self._pk else x
        y = self._causal3(self._norm(x, p + ".norm1.gamma"), p + ".conv1", caches=caches)
        return self._causal3(self._norm(y, p + ".norm2.gamma"), p + ".conv2", residual=h, caches=caches)

    def _attn_f32(self, x, p):
        """WanAttentionBlock (:402-427) in the fp32-compute mode: every matrix product a split-bf16 product with fp32 output
        (ops.gemm_f32 on operands expanded by ops.split_bf16), norm and softmax in fp32."""
        t, h, w, cp = x.shape
        e, pr, np_ = self._pk[p + ".to_qkv"], self._pk[p + ".proj"], self._planes
        nprod = len(ops.SPLIT_PRODUCTS[np_])
        c, hw = e.co, h * w
        hwp = (hw + 63) // 64 * 64                               # K of the P.V product: whole 64-wide K-tiles
        n = self._norm(x, p + ".norm.gamma", silu=False, split=None).view(t, hw, cp)
        k6 = nprod * cp
        wq, wk = e.w[:cp], e.w[cp:2 * cp]                         # W-side expansions [cp, products * cp]
        wv_a = e.w_a[2 * cp:]                                     # A-side expansion of W_v: V^T = W_v . n^T
        bq, bk, bv = e.b[:cp], e.b[cp:2 * cp], e.b[2 * cp:]
        assert wq.shape[1] == k6
        out = torch.empty_like(x).view(t, hw, cp)
        scale = c ** -0.5
        for f in range(t):
            nf = n[f]
            if hwp != hw:
                nf = torch.zeros(hwp, cp, dtype=torch.float32, device=x.device)
                nf[:hw] = n[f]
            n_a, n_w = ops.split_bf16(nf, "A", np_), ops.split_bf16(nf, "W", np_)
            q = ops.gemm_f32(n_a[:hw], wq, bq)                                       # [hw, cp]
            k = ops.gemm_f32(n_a, wk, bk)                                            # [hwp, cp]
            if hwp != hw:
                k[hw:] = 0
            vt = ops.gemm_f32(wv_a, n_w)                                             # V^T [cp, hwp]; bias after P.V (rows of P sum to 1)
            s_ = ops.gemm_f32(ops.split_bf16(q, "A", np_), ops.split_bf16(k, "W", np_))      # [hw, hwp]
            ops.softmax_rows_(s_, hw, scale)
            if hwp != hw:
                s_[:, hw:] = 0
            o = ops.gemm_f32(ops.split_bf16(s_, "A", np_), ops.split_bf16(vt, "W", np_), bv)
            ops.gemm_f32(ops.split_bf16(o, "A", np_), pr.w, pr.b, residual=x.view(t, hw, cp)[f], out=out[f])
        return out.view(t, h, w, cp)

    def _attn(self, x, p):
        """WanAttentionBlock (:402-427): one head of dim C over the h.w tokens of each frame."""
        if self._planes:
            return self._attn_f32(x, p)
        t, h, w, cp = x.shape
        e = self._pk[p + ".to_qkv"]
        c, hw = e.co, h * w
        hwp = (hw + 7) // 8 * 8
        n = self._norm(x, p + ".norm.gamma", silu=False).view(t, hw, cp)
        wq, wk, wv = e.w[:cp], e.w[cp:2 * cp], e.w[2 * cp:]
        bq, bk, bv = e.b[:cp], e.b[cp:2 * cp], e.b[2 * cp:]
        out = torch.empty_like(n)
        scale = c ** -0.5
        pr = self._pk[p + ".proj"]
        for f in range(t):
            nf = n[f]
            if hwp != hw:                                     # tiny test shapes only: pad tokens to a multiple of 8
                nf = torch.zeros(hwp, cp, dtype=x.dtype, device=x.device)
                nf[:hw] = n[f]
            q = ops.gemm(nf[:hw], wq, bq)
            k = ops.gemm(nf, wk, bk)
            if hwp != hw:
                k[hw:] = 0
            vt = ops.gemm(wv, nf)                             # V^T [Cpad, hwp] (bias added after P.V: rows of P sum to 1)
            s = ops.gemm(q, k)                                # [hw, hwp]
            ops.softmax_rows_(s, hw, scale)
            if hwp != hw:
                s[:, hw:] = 0
            o = ops.gemm(s, vt, bv)
            ops.gemm(o, pr.w, pr.b, ops.EPI_RESIDUAL, residual=x.view(t, hw, cp)[f], out=out[f])
        return out.view(t, h, w, cp)

    def _mid(self, x, p):
        x = self._res(x, p + ".resnets.0")
        x = self._attn(x, p + ".attentions.0")
        return self._res(x, p + ".resnets.1")

    # ---- decode (reference :1198-1227, whole-sequence) ----
    def _tail_halo_rows(self, first, nb):
        """(input rows of halo one side of a slab of the decoder tail `up_blocks[first:]` + head needs, output rows per input row):
        walking the layers from the output back, a stride-1 3 x 3 convolution adds one row at its resolution, a 2x nearest
        upsample halves (rounding up) what lies behind it."""
        cfg = self.config
        ps = cfg.patch_size or 1
        need, scale = 1, ps                                   # conv_out (3 x 3 x 3); unpatchify is pointwise
        for i in range(nb - 1, first - 1, -1):
            if i != nb - 1:                                   # this block ends in [2x upsample, 3 x 3 conv]: the conv, then halve
                need = -(-(need + 1) // 2)
                scale *= 2
            need += 2 * (cfg.num_res_blocks + 1)              # its resnets: two 3 x 3 (x 3) convolutions each
        return need, scale

    @torch.no_grad()
    def decode_slab(self, z, index, count):
        """rows of the decoded video that slab `index` of `count` holds, computed WITHOUT the other slabs' tail work:
        -> (video rows [1, C, T, rows, W] or None for an empty slab, (row0, row1, rows_per_slab, H)).  The blocks up to the last
        temporal upsampling (26 % of the decode's FLOPs at 704 x 1280) run whole on every caller; the tail runs on the slab +
        its halo.  Bit-identical to the same rows of `decode(z)` (tests/test_wan_vae_gpu.py)."""
        return self.decode(z, return_dict="slab", slab=(index, count))

    @torch.no_grad()
    def decode(self, z, return_dict=True, slab=None):
        if z.shape[0] != 1:
            if slab is not None:
                raise NotImplementedError("decode_slab decodes one video per call")
            outs = [self.decode(z[i:i + 1], return_dict=False)[0] for i in range(z.shape[0])]
            out = torch.cat(outs)
            return (out,) if not return_dict else SimpleNamespace(sample=out)
        pk = self._pk or self._pack()
        cfg, dt = self.config, self._dtype
        mult = cfg.dim_mult
        dec = [cfg.decoder_base_dim * u for u in [mult[-1]] + mult[::-1]]
        tup = cfg.temperal_downsample[::-1]
        zc = cfg.z_dim
        _, _, t, h, w = z.shape
        if self._planes:
            dt = torch.float32                          # fp32-compute mode: activations stay fp32 between the layers
        x = torch.zeros(t, h, w, cpad(zc), dtype=dt, device=z.device)
        x[..., :zc] = z[0].permute(1, 2, 3, 0).to(dt)
        x = self._causal3(x, "post_quant_conv")
        x = self._causal3(x, "decoder.conv_in")
        x = self._mid(x, "decoder.mid_block")
        nb = len(mult)
        ps = cfg.patch_size or 1

        def up_block(x, i, caches=None):
            """one WanResidualUpBlock (:626-716); with `caches`: on a time chunk (blocks without temporal upsampling)"""
            p = f"decoder.up_blocks.{i}"
            up_flag = i != nb - 1
            temporal = bool(tup[i]) if up_flag else False
            x_copy = x
            for r in range(cfg.num_res_blocks + 1):
                x = self._res(x, f"{p}.resnets.{r}", caches)
            if up_flag:
                if temporal and x.shape[0] > 1:
                    assert caches is None
                    e = pk[p + ".upsampler.time_conv"]
                    tt, hh, ww, cp = x.shape
                    y = self._causal3(x[1:].contiguous(), p + ".upsampler.time_conv")     # [T-1, H, W, 2*Cpad]
                    xn = torch.empty(1 + 2 * (tt - 1), hh, ww, cp, dtype=dt, device=x.device)
                    xn[0] = x[0]
                    xn[1:].view(tt - 1, 2, hh, ww, cp).copy_(y.view(tt - 1, hh, ww, 2, e.cop).permute(0, 3, 1, 2, 4))
                    x = xn
                x = self._conv(x, p + ".upsampler.resample.1", (0, 1, 1), up=True)
                x = ops.dup_up3d_add(x, x_copy, dec[i], dec[i + 1], 2 if temporal else 1, 2)
            return x

        def head(x, caches=None):
            x = self._norm(x, "decoder.norm_out.gamma")
            x = self._causal3(x, "decoder.conv_out", caches=caches)
            return ops.vae_unpatchify_clamp(x, cfg.out_channels // ps ** 2, ps)              # [C, T, H*ps, W*ps] fp32

        # The blocks up to and including the last temporal upsampling run once over the whole sequence; the tail --
        # where the activations are largest (49 x 352 x 640 x 512 channels = 11 GB per tensor at 704x1280) and the frame
        # count no longer changes -- runs in time chunks with the last two input frames of every causal conv carried
        # over: identical results (each output element sees the same taps), activation memory of one chunk.
        last_temporal = max([i for i in range(nb - 1) if tup[i]], default=-1)
        for i in range(last_temporal + 1):
            x = up_block(x, i)

        def tail(x):
            chunk = self.decode_chunk_frames
            tt = x.shape[0]
            if chunk == "auto":
                # measured: the whole-sequence peak is ~6.4 tensors of [T, H/2, W/2, decoder_base_dim] at the output size
                sp = 2 ** (nb - 1 - (last_temporal + 1))
                est = 6.4 * tt * x.shape[1] * sp * x.shape[2] * sp * cpad(cfg.decoder_base_dim) * 2 / 2 ** 30
                if self._planes:
                    est *= 2 + self._planes                 # fp32 activations + their bf16 planes
                chunk = 8 if est > self.decode_memory_budget_gib else 0
            if not chunk or chunk >= tt or last_temporal + 1 >= nb:
                for i in range(last_temporal + 1, nb):
                    x = up_block(x, i)
                return head(x)
            caches, out = {}, None
            for t0 in range(0, tt, chunk):
                xc = x[t0:t0 + chunk]
                for i in range(last_temporal + 1, nb):
                    xc = up_block(xc, i, caches)
                v = head(xc, caches)
                if out is None:
                    out = torch.empty((v.shape[0], tt) + tuple(v.shape[2:]), dtype=v.dtype, device=v.device)
                out[:, t0:t0 + v.shape[1]] = v
                del xc, v
            return out

        if slab is None or last_temporal + 1 >= nb:
            out = tail(x)[None]
        else:
            # ---- the tail on ONE horizontal slab of the frame (round 6: N ranks decode N slabs, frameino_amd/parallel.py) ----
            # From here on every layer is local in space except the 3 x 3 taps of its convolutions (RMS norms are per pixel, the
            # mid-block attention is behind us, DupUp3D and nearest upsampling are pointwise): rows [a, b) of this tensor become
            # rows [a, b) x (output rows per input row) of the video, and need input rows [a - halo, b + halo) -- one row per
            # stride-1 3 x 3 convolution at this resolution, half a row per convolution behind a 2x upsample.  The slab is cut
            # WITH that halo, runs through the unchanged layers (whose zero padding is right at the frame's own border and
            # wrong at a cut: the error creeps in one row per convolution and ends exactly at the halo), and the halo's output
            # is cropped.  Every kept element sees the same taps in the same K order as in the un-cut decode: bit-identical.
            si, sn = int(slab[0]), int(slab[1])
            hrows = x.shape[1]
            halo, scale = self._tail_halo_rows(last_temporal + 1, nb)
            per = -(-hrows // sn)
            a, b = min(hrows, si * per), min(hrows, (si + 1) * per)
            if b <= a:
                out = None
            else:
                s0, s1 = max(0, a - halo), min(hrows, b + halo)
                v = tail(x[:, s0:s1].contiguous())
                del x
                out = v[:, :, (a - s0) * scale:(b - s0) * scale].contiguous()[None]
            if return_dict == "slab":
                return out, (a * scale, b * scale, per * scale, hrows * scale)
        return (out,) if not return_dict else SimpleNamespace(sample=out)

    # ---- encode (reference :1145-1169, whole-sequence) ----
    _ENC_SLAB_BLOCKS = 2          # conv_in + down_blocks.0 / .1 (73 % of the encoder's FLOPs at 704 x 1280) run on a slab; see _enc_halo_rows

    def _enc_halo_rows(self, k):
        """(rows of halo above, below) -- at the resolution of conv_in's input -- that rows [a, b) of the output of `down_blocks[k - 1]`
        need, rounded up to multiples of 2^k so that a slab starts on the same phase of every stride-2 convolution as the whole frame
        does (and the asymmetric ZeroPad2d((0, 1, 0, 1)) lands on the frame's own bottom edge only).  Walking back from the output: a
        stride-2 3 x 3 convolution reads input rows 2r .. 2r + 2 (and the AvgDown3D shortcut 2r, 2r + 1), a stride-1 one r - 1 .. r + 1."""
        nrb = self.config.num_res_blocks
        lo = hi = 0
        for _ in range(k):
            lo, hi = 2 * lo, 2 * hi + 1                       # the block's downsampler
            lo, hi = lo + 2 * nrb, hi + 2 * nrb               # its resnets (two 3 x 3 x 3 convolutions each)
        lo, hi = lo + 1, hi + 1                               # conv_in
        q = 2 ** k
        return -(-lo // q) * q, -(-hi // q) * q

    @torch.no_grad()
    def _encode(self, x, slab=None, resume=None):
        """`slab=(index, count)`: run conv_in and the first `_ENC_SLAB_BLOCKS` down blocks on one horizontal slab of the frame (+ halo)
        and return (rows [a, b) of that block's output [T', b - a, W', Cpad], (a, b, rows per slab, H')) -- bit-identical to those rows of
        the whole encode.  `resume=activation`: continue from that block's (gathered) output to the moments.  (round 6:
        frameino_amd/parallel.py::sharded_vae_encode)"""
        pk = self._pk or self._pack()
        cfg, dt = self.config, self._dtype
        mult = cfg.dim_mult
        enc = [cfg.base_dim * u for u in [1] + mult]
        tdown = cfg.temperal_downsample
        ps = cfg.patch_size or 1
        if self._planes:
            dt = torch.float32
        nb = len(mult)
        k = min(self._ENC_SLAB_BLOCKS, nb - 1)
        crop = None
        if resume is not None:
            x, first = resume, k
        else:
            x = ops.vae_patchify(x[0].float().contiguous(), cpad(cfg.in_channels), ps, dt)
            first = 0
            if slab is not None:
                si, sn = int(slab[0]), int(slab[1])
                q = 2 ** k
                h0 = x.shape[1]
                if h0 % q:
                    raise ValueError(f"encode_slab: the frame's {h0 * ps} rows must be a multiple of {q * ps}")
                hk = h0 // q
                per = -(-hk // sn)
                a, b = min(hk, si * per), min(hk, (si + 1) * per)
                if b <= a:
                    return None, (a, b, per, hk)
                lo, hi = self._enc_halo_rows(k)
                s0, s1 = max(0, a * q - lo), min(h0, b * q + hi)
                x = x[:, s0:s1].contiguous()
                crop = (a - s0 // q, b - s0 // q, (a, b, per, hk))
            x = self._causal3(x, "encoder.conv_in")
        for i in range(first, nb):
            p = f"encoder.down_blocks.{i}"
            down_flag = i != nb - 1
            temporal = bool(tdown[i]) if down_flag else False
            x_copy = x
            for r in range(cfg.num_res_blocks):
                x = self._res(x, f"{p}.resnets.{r}")
            if down_flag:
                tt, hh, ww, _ = x.shape
                # ZeroPad2d((0,1,0,1)) + Conv2d(3, stride 2): taps past the bottom/right edge read zeros
                x = self._conv(x, p + ".downsampler.resample.1", (0, 0, 0), stride=(1, 2, 2),
                               out_thw=(tt, hh // 2, ww // 2))
                if temporal and x.shape[0] > 1:
                    tt = x.shape[0]
                    zc = self._conv(x, p + ".downsampler.time_conv", (0, 0, 0), stride=(2, 1, 1),
                                    out_thw=((tt - 1) // 2, x.shape[1], x.shape[2]))
                    x = torch.cat([x[:1], zc], dim=0)
            x = ops.avg_down3d_add(x, x_copy, enc[i], enc[i + 1], 2 if temporal else 1, 2 if down_flag else 1)
            if crop is not None and i == k - 1:
                return x[:, crop[0]:crop[1]].contiguous(), crop[2]
        x = self._mid(x, "encoder.mid_block")
        x = self._norm(x, "encoder.norm_out.gamma")
        x = self._causal3(x, "encoder.conv_out")
        x = self._causal3(x, "quant_conv")
        z2 = 2 * cfg.z_dim
        return x[..., :z2].permute(3, 0, 1, 2).float()[None].contiguous()             # [1, 2z, T', h, w]

    def encode_slab(self, x, index, count):
        """-> (rows [a, b) of the activation behind `encoder.down_blocks[_ENC_SLAB_BLOCKS - 1]` or None, (a, b, rows per slab, H'))"""
        if x.shape[0] != 1:
            raise NotImplementedError("encode_slab encodes one video per call")
        return self._encode(x, slab=(index, count))

    def encode_resume(self, activation):
        """the rest of the encoder on the (gathered) activation `encode_slab` returns pieces of -> latent_dist"""
        return SimpleNamespace(latent_dist=DiagonalGaussianDistribution(self._encode(None, resume=activation)))

    def encode(self, x, return_dict=True):
        moments = torch.cat([self._encode(x[i:i + 1]) for i in range(x.shape[0])])
        post = DiagonalGaussianDistribution(moments)
        return (post,) if not return_dict else SimpleNamespace(latent_dist=post)
