"""MI355X-native CogVideoXTransformer3DModel (CogVideoX-5B-I2V as FrameINO fine-tunes it: 48 input channels,
`use_FrameIn=True`, learned positional embedding + 3D RoPE).

Mirror of /root/reference/architecture/cogvideox_transformer_3d.py: same constructor arguments, parameter names,
`forward(hidden_states, encoder_hidden_states, timestep, timestep_cond=None, ofs=None, image_rotary_emb=None,
attention_kwargs=None, return_dict=True)`, `attn_processors` / `set_attn_processor` / `fuse_qkv_projections` /
`unfuse_qkv_projections` (:346-444).  Arithmetic runs in libframeino_hip.so.

Layout: text and video tokens of a batch element live in ONE row-major buffer `[B, Lt+Lv, D]` (text first) -- the
reference concatenates them for attention and for the feed-forward in every block anyway (:155, attention_processor
:2826).  Per-row modulation uses a `[2B, D]` table (row 2b = video params, 2b+1 = text params) + an int32 selector, so
LayerNormZero / gates are single kernels over all rows; the gate multiply + residual is the GEMM epilogue.
"""
import math
import os
from types import SimpleNamespace

import torch
import torch.nn.functional as F
from torch import nn

from . import ops
from .attention_processor import Attention, MI355CogVideoXAttnProcessor, MI355FusedCogVideoXAttnProcessor
from .loading import FromPretrainedMixin
from .transformer_wan import FeedForward, _Config, _MLP2


class _LayerNormZero(nn.Module):
    def __init__(self, cond_dim, dim, affine, eps):
        super().__init__()
        self.linear = nn.Linear(cond_dim, 6 * dim)
        self.norm = nn.LayerNorm(dim, eps=eps, elementwise_affine=affine)


class _AdaLayerNorm(nn.Module):
    def __init__(self, cond_dim, out_dim, affine, eps):
        super().__init__()
        self.linear = nn.Linear(cond_dim, out_dim)
        self.norm = nn.LayerNorm(out_dim // 2, eps, affine)


class CogVideoXPatchEmbed(nn.Module):
    def __init__(self, patch_size, in_channels, embed_dim, text_embed_dim, bias, pos_shape, persistent):
        super().__init__()
        self.proj = nn.Conv2d(in_channels, embed_dim, kernel_size=(patch_size, patch_size), stride=patch_size, bias=bias)
        self.text_proj = nn.Linear(text_embed_dim, embed_dim)
        self.register_buffer("pos_embedding", torch.zeros(pos_shape), persistent=persistent)


class CogVideoXBlock(nn.Module):
    def __init__(self, dim, heads, head_dim, time_embed_dim, attention_bias, affine, eps):
        super().__init__()
        self.norm1 = _LayerNormZero(time_embed_dim, dim, affine, eps)
        self.attn1 = Attention(dim, heads=heads, dim_head=head_dim, qk_norm="layer_norm", eps=1e-6,
                               bias=attention_bias, out_bias=True, processor=MI355CogVideoXAttnProcessor())
        self.norm2 = _LayerNormZero(time_embed_dim, dim, affine, eps)
        self.ff = FeedForward(dim, 4 * dim)


def _sincos_1d(dim, pos):
    omega = 1.0 / 10000 ** (torch.arange(dim // 2, dtype=torch.float64) / (dim / 2.0))
    out = torch.outer(pos.reshape(-1).double(), omega)
    return torch.cat([torch.sin(out), torch.cos(out)], dim=1).float()


def cog_sincos_pos_embed(embed_dim, pw, ph, frames, spatial_scale, temporal_scale):
    """architecture/embeddings.py:81-150 (get_3d_sincos_pos_embed, pt path) -> [frames*ph*pw, embed_dim]."""
    ds, dtm = 3 * embed_dim // 4, embed_dim // 4
    gh = torch.arange(ph, dtype=torch.float32) / spatial_scale
    gw = torch.arange(pw, dtype=torch.float32) / spatial_scale
    grid = torch.stack(torch.meshgrid(gw, gh, indexing="xy"), dim=0).reshape(2, 1, ph, pw)
    emb_h = _sincos_1d(ds // 2, grid[0])
    emb_w = _sincos_1d(ds // 2, grid[1])
    spatial = torch.cat([emb_h, emb_w], dim=1)                                       # [ph*pw, ds]
    temporal = _sincos_1d(dtm, torch.arange(frames, dtype=torch.float32) / temporal_scale)
    spatial = spatial[None].repeat_interleave(frames, dim=0)
    temporal = temporal[:, None].repeat_interleave(ph * pw, dim=1)
    return torch.cat([temporal, spatial], dim=-1).flatten(0, 1)


class CogVideoXTransformer3DModel(nn.Module, FromPretrainedMixin):
    _loader_name = "load_cogvideox_transformer"

    def __init__(self, num_attention_heads=30, attention_head_dim=64, in_channels=16, out_channels=16,
                 flip_sin_to_cos=True, freq_shift=0, time_embed_dim=512, ofs_embed_dim=None, text_embed_dim=4096,
                 num_layers=30, dropout=0.0, attention_bias=True, sample_width=90, sample_height=60, sample_frames=49,
                 patch_size=2, patch_size_t=None, temporal_compression_ratio=4, max_text_seq_length=226,
                 activation_fn="gelu-approximate", timestep_activation_fn="silu", norm_elementwise_affine=True,
                 norm_eps=1e-5, spatial_interpolation_scale=1.875, temporal_interpolation_scale=1.0,
                 use_rotary_positional_embeddings=False, use_learned_positional_embeddings=False, patch_bias=True,
                 extra_encoder_cond_channels=-1, use_FrameIn=False):
        super().__init__()
        if patch_size_t is not None or ofs_embed_dim:
            raise NotImplementedError("CogVideoX-1.5 (patch_size_t / ofs) is not on FrameINO's path")
        if not use_rotary_positional_embeddings:
            raise NotImplementedError("FrameINO fine-tunes CogVideoX-5B (RoPE); the 2B variant is not on its path")
        inner = num_attention_heads * attention_head_dim
        self.inner_dim = inner
        self.config = _Config(num_attention_heads=num_attention_heads, attention_head_dim=attention_head_dim,
                              in_channels=in_channels, out_channels=out_channels, flip_sin_to_cos=flip_sin_to_cos,
                              freq_shift=freq_shift, time_embed_dim=time_embed_dim, ofs_embed_dim=ofs_embed_dim,
                              text_embed_dim=text_embed_dim, num_layers=num_layers, sample_width=sample_width,
                              sample_height=sample_height, sample_frames=sample_frames, patch_size=patch_size,
                              patch_size_t=patch_size_t, temporal_compression_ratio=temporal_compression_ratio,
                              max_text_seq_length=max_text_seq_length, norm_elementwise_affine=norm_elementwise_affine,
                              norm_eps=norm_eps, spatial_interpolation_scale=spatial_interpolation_scale,
                              temporal_interpolation_scale=temporal_interpolation_scale,
                              use_rotary_positional_embeddings=use_rotary_positional_embeddings,
                              use_learned_positional_embeddings=use_learned_positional_embeddings,
                              use_FrameIn=use_FrameIn)
        pph, ppw = sample_height // patch_size, sample_width // patch_size
        ptf = (sample_frames - 1) // temporal_compression_ratio + 1
        self.patch_embed = CogVideoXPatchEmbed(patch_size, in_channels, inner, text_embed_dim, patch_bias,
                                               (1, max_text_seq_length + pph * ppw * ptf, inner),
                                               use_learned_positional_embeddings)
        if not use_learned_positional_embeddings:
            pe = cog_sincos_pos_embed(inner, ppw, pph, ptf, spatial_interpolation_scale, temporal_interpolation_scale)
            self.patch_embed.pos_embedding[:, max_text_seq_length:].copy_(pe)
        self.time_embedding = _MLP2(inner, time_embed_dim)
        self.transformer_blocks = nn.ModuleList([
            CogVideoXBlock(inner, num_attention_heads, attention_head_dim, time_embed_dim, attention_bias,
                           norm_elementwise_affine, norm_eps) for _ in range(num_layers)])
        self.norm_final = nn.LayerNorm(inner, norm_eps, norm_elementwise_affine)
        self.norm_out = _AdaLayerNorm(time_embed_dim, 2 * inner, norm_elementwise_affine, norm_eps)
        self.proj_out = nn.Linear(inner, patch_size * patch_size * out_channels)
        self._packed = None
        self._pos_cache = {}
        self._fp8 = {}
        self._fp8_pending = False
        self.fp8_attention = False           # see enable_fp8_attention
        # q leaves its LayerNorm + RoPE kernel multiplied by head_dim**-0.5 * log2(e) and the attention kernels take q.k as
        # the exp2 argument (FINO_ATTN_SCALE_FOLDED): at head_dim 64 that selects the 4-wave kernel with the running maximum
        # folded into its MFMAs (-5 % per step).  Video rows: one rounding of q.c instead of q; the 226 text rows (LayerNorm
        # only, already rounded) are rounded twice.  False: q as the reference rounds it, scale applied to the logits.
        self.fold_softmax_scale = True
        # in the last block, rows nobody reads afterwards (text rows; frames beyond forward(live_frames=)) are keys / values only
        self.skip_dead_rows = os.environ.get("FINO_SKIP_DEAD_ROWS", "1") != "0"
        self.original_attn_processors = None

    # ---- reference surface (:346-444) ----
    @property
    def dtype(self):
        return self.proj_out.weight.dtype

    @property
    def device(self):
        return self.proj_out.weight.device

    @property
    def attn_processors(self):
        return {f"transformer_blocks.{i}.attn1.processor": b.attn1.processor
                for i, b in enumerate(self.transformer_blocks)}

    def set_attn_processor(self, processor):
        count = len(self.transformer_blocks)
        if isinstance(processor, dict) and len(processor) != count:
            raise ValueError(f"A dict of processors was passed, but the number of processors {len(processor)} does not "
                             f"match the number of attention layers: {count}. Please make sure to pass {count} "
                             f"processor classes.")
        for i, b in enumerate(self.transformer_blocks):
            b.attn1.set_processor(processor[f"transformer_blocks.{i}.attn1.processor"] if isinstance(processor, dict)
                                  else processor)
        self._packed = None

    def fuse_qkv_projections(self):
        self.original_attn_processors = self.attn_processors
        for b in self.transformer_blocks:
            b.attn1.fuse_projections(fuse=True)
        self.set_attn_processor(MI355FusedCogVideoXAttnProcessor())

    def unfuse_qkv_projections(self):
        if self.original_attn_processors is not None:
            self.set_attn_processor(self.original_attn_processors)

    def load_reference_state_dict(self, sd, dtype=None):
        own = self.state_dict()
        missing = [k for k in own if k not in sd]
        if missing:
            raise KeyError(f"state-dict is missing {missing[:5]}")
        with torch.no_grad():
            for k, t in list(self.named_parameters()) + list(self.named_buffers()):
                t.data = sd[k].to(dtype or sd[k].dtype).to(t.device).contiguous()
        self.reset_caches()
        return self

    def reset_caches(self):
        """Drop everything derived from the parameters (packed / fused copies, MXFP8 weights, positional tables);
        called whenever the parameters may have changed or moved."""
        had_fp8 = bool(self._fp8) or self._fp8_pending
        self._packed = None
        self._fp8 = {}
        self._fp8_pending = had_fp8      # re-quantised lazily by the next forward, from wherever the parameters are then
        self._pos_cache.clear()
        return had_fp8

    def _apply(self, fn, *args, **kwargs):          # .to() / .cuda() / .half()
        out = super()._apply(fn, *args, **kwargs)
        if hasattr(self, "_pos_cache"):
            self.reset_caches()
        return out

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self.reset_caches()
        return out

    # ---- packing ----
    # ------------------------------------------------------------------ MXFP8 linears (BASELINE config 5)
    def enable_mxfp8_linears(self, enabled=True):
        """QKV, attention out, FFN up and FFN down of every block on the MXFP8 path (OCP e4m3 + one e8m0 scale per 32
        K-elements, fp32 accumulate; `fino_quantize_mxfp8` / `fino_gemm_mxfp8`); weights are quantised once here.
        Attention, norms, embeddings and the output head stay in the model dtype.  No reference counterpart (SURVEY
        F11): compared with this model's own bf16 forward."""
        self._fp8 = {}
        self._fp8_pending = False
        if not enabled:
            return self
        pk = self._packed or self._pack()
        for li, (blk, e) in enumerate(zip(self.transformer_blocks, pk.layers)):
            for key, w in (("qkv", e.wqkv), ("out", blk.attn1.to_out[0].weight), ("ff1", blk.ff.net[0].proj.weight),
                           ("ff2", blk.ff.net[2].weight)):
                self._fp8[(li, key)] = ops.quantize_mxfp8(w.detach().contiguous())
        return self

    def enable_fp8_attention(self, enabled=True, p_mode=None):
        """The joint text + video self-attention (attention_processor.py:2863 of the reference, head_dim 64: 55 % of the
        CogVideoX-5B step) with fp8 e4m3 matrix operands -- K / V quantised per call with one scale per 32 elements, Q and
        P in registers, both products on the block-scaled fp8 MFMA, fp32 softmax and accumulation
        (`fino_attn_fwd_fp8`).  With `enable_mxfp8_linears()` this is BASELINE config 5's "fp8 MFMA path" end to end.
        No reference counterpart: tolerance stated in tests/test_attention_fp8_gpu.py and
        tests/test_fullsize_oracle_gpu.py against fp32 and against this model's own bf16 forward.
        p_mode: "exp2" | "ramp" -- how a softmax weight becomes its e4m3 byte (ops.FP8_P_*; None = ops.FP8_P_DEFAULT)."""
        self.fp8_attention = bool(enabled)
        self.fp8_p_mode = p_mode
        return self

    def _lin(self, li, key, x, w, b, epi=0, xq=None, **kw):
        wq = self._fp8.get((li, key)) if self._fp8 else None
        if wq is None:
            return ops.gemm(x, w, b, epi, **kw)
        xq, xs = xq if xq is not None else ops.quantize_mxfp8(x)
        return ops.gemm_mxfp8(xq, xs, wq[0], wq[1], b, epi, **kw)

    def _lnz_q(self, li, key, x2, w, b, shift, scale, sel, eps):
        """CogVideoXLayerNormZero in front of linear (li, key) emitted directly as that linear's MXFP8 activations
        (fino_ln_mxfp8 mode 2: one pass instead of norm -> bf16 -> quantise); None when the linear is not on the MXFP8 path"""
        if not self._fp8 or (li, key) not in self._fp8 or not hasattr(ops, "ln_mxfp8") or os.environ.get("FINO_NO_LN_MXFP8"):
            return None
        return ops.ln_mxfp8(2, x2, weight=w, bias=b, shift=shift, scale=scale, sel=sel, eps=eps)

    def _default_processors(self):
        return all(type(b.attn1.processor) in (MI355CogVideoXAttnProcessor, MI355FusedCogVideoXAttnProcessor)
                   for b in self.transformer_blocks)

    def _pack(self):
        pk = SimpleNamespace(layers=[])
        f32 = lambda t: None if t is None else t.detach().float().contiguous()     # noqa: E731
        for b in self.transformer_blocks:
            a = b.attn1
            e = SimpleNamespace()
            e.wqkv = torch.cat([a.to_q.weight, a.to_k.weight, a.to_v.weight]).detach().contiguous()
            e.bqkv = torch.cat([a.to_q.bias, a.to_k.bias, a.to_v.bias]).detach().contiguous() \
                if a.to_q.bias is not None else None
            e.n1w, e.n1b = f32(b.norm1.norm.weight), f32(b.norm1.norm.bias)
            e.n2w, e.n2b = f32(b.norm2.norm.weight), f32(b.norm2.norm.bias)
            pk.layers.append(e)
        # all 2*layers LayerNormZero projections + norm_out as ONE skinny GEMM weight
        ws = [m.linear.weight for b in self.transformer_blocks for m in (b.norm1, b.norm2)] + [self.norm_out.linear.weight]
        bs = [m.linear.bias for b in self.transformer_blocks for m in (b.norm1, b.norm2)] + [self.norm_out.linear.bias]
        pk.wmod = torch.cat(ws).detach().contiguous()
        pk.bmod = torch.cat(bs).detach().contiguous()
        pk.w_patch = self.patch_embed.proj.weight.detach().reshape(self.inner_dim, -1).contiguous()
        pk.nfw, pk.nfb = f32(self.norm_final.weight), f32(self.norm_final.bias)
        pk.now, pk.nob = f32(self.norm_out.norm.weight), f32(self.norm_out.norm.bias)
        self._packed = pk
        return pk

    def _pos_embeds(self, num_frames, height, width, text_len, dtype):
        """architecture/embeddings.py:764-802 (FrameIn first-frame PE reuse, trilinear resize off the default size).
        Step-invariant: built once per geometry (the reference rebuilds and re-interpolates it every forward)."""
        key = (num_frames, height, width, text_len, dtype)
        if key in self._pos_cache:
            return self._pos_cache[key]
        c = self.config
        pos = self.patch_embed.pos_embedding
        ps, tcr, maxt = c.patch_size, c.temporal_compression_ratio, c.max_text_seq_length
        post = (c.sample_frames - 1) // tcr + 1
        pph, ppw = c.sample_height // ps, c.sample_width // ps
        seq = height * width * num_frames // (ps * ps)
        if c.use_FrameIn:
            first = (pos.shape[1] - maxt) // (num_frames - 1)
            pos = torch.cat([pos, pos[:, text_len:text_len + first].clone()], dim=1)
        if c.sample_height != height or c.sample_width != width or c.sample_frames != (num_frames - 1) * tcr + 1:
            if c.use_FrameIn:
                post += 1
            d = pos.shape[-1]
            pw_ = pos[:, text_len:].view(1, post, pph, ppw, d).permute(0, 4, 1, 2, 3)
            pw_ = F.interpolate(pw_, size=[post, height // ps, width // ps], mode="trilinear", align_corners=False)
            pw_ = pw_.permute(0, 2, 3, 4, 1).reshape(1, -1, d)
            pos = torch.cat([pos[:, :text_len], pw_], dim=1)[:, :text_len + seq]
        out = pos[0].to(dtype).contiguous()
        self._pos_cache[key] = out
        return out

    # ---- forward (:446-562) ----
    @torch.no_grad()
    def forward(self, hidden_states, encoder_hidden_states, timestep, timestep_cond=None, ofs=None,
                image_rotary_emb=None, attention_kwargs=None, return_dict=True, live_frames=None):
        """`live_frames=k` (round 6): the caller reads the prediction of the first k latent frames only -- the FrameINO loop drops
        the identity frame appended on the frame axis (pipeline_cogvideox_i2v_motion_FrameINO.py:866-881, :896).  In the LAST block
        the other frames' tokens then serve as keys / values only, and so do the TEXT rows on every call (the model returns video
        rows only, :531-542): their attention queries, out-projection and feed-forward are skipped; dropped frames come back ZERO.
        Every returned row is computed exactly as without it."""
        if timestep_cond is not None:
            raise NotImplementedError("timestep_cond is never passed on the FrameINO path")
        if attention_kwargs is not None:
            attention_kwargs = dict(attention_kwargs)
            attention_kwargs.pop("scale", None)
        if self._fp8_pending:
            self.enable_mxfp8_linears()
        pk = self._packed or self._pack()
        default_procs = self._default_processors()
        c = self.config
        b, nf, ch, hh, ww = hidden_states.shape
        d, heads, dh, ps = self.inner_dim, c.num_attention_heads, c.attention_head_dim, c.patch_size
        dev, dt = hidden_states.device, hidden_states.dtype
        lt = encoder_hidden_states.shape[1]
        lv = nf * (hh // ps) * (ww // ps)
        L = lt + lv

        # 1. time embedding (:477-483): sinusoid fp32 -> T -> MLP in T
        half = d // 2
        expo = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=dev) / (half - c.freq_shift))
        ang = timestep.reshape(-1).float()[:, None] * expo[None]
        t_emb = torch.cat([torch.sin(ang), torch.cos(ang)], dim=-1)
        if c.flip_sin_to_cos:
            t_emb = torch.cat([t_emb[:, half:], t_emb[:, :half]], dim=-1)
        te = self.time_embedding
        h1 = ops.skinny_linear(t_emb.to(dt).float().contiguous(), te.linear_1.weight, te.linear_1.bias).to(dt)
        emb = ops.skinny_linear(F.silu(h1).float().contiguous(), te.linear_2.weight, te.linear_2.bias).to(dt)   # [B, E]
        # every LayerNormZero / AdaLayerNorm projection of silu(emb) in one launch -> [B, (2*layers*6 + 2) * D] in T
        mods = ops.skinny_linear(F.silu(emb).float().contiguous(), pk.wmod, pk.bmod).to(dt).float()
        nl = c.num_layers
        lnz = mods[:, :2 * nl * 6 * d].view(b, 2 * nl, 2, 3, d)                      # [B, norm, video|text, (shift,scale,gate), D]
        tables = lnz.permute(1, 0, 2, 3, 4).reshape(2 * nl, 2 * b, 3, d).contiguous()   # row 2b = video, 2b+1 = text
        out_mod = mods[:, 2 * nl * 6 * d:].view(b, 2, d).contiguous()                # (shift, scale) of norm_out
        sel = torch.zeros(b, L, dtype=torch.int32, device=dev)
        sel[:, :lt] = 1
        sel += 2 * torch.arange(b, device=dev, dtype=torch.int32)[:, None]
        sel = sel.reshape(-1).contiguous()

        # 2. patch embedding (:494) into the joint buffer [B, Lt+Lv, D]
        x = torch.empty(b, L, d, dtype=dt, device=dev)
        pe = self.patch_embed
        pos = self._pos_embeds(nf, hh, ww, lt, dt)
        for i in range(b):
            ops.gemm(encoder_hidden_states[i], pe.text_proj.weight, pe.text_proj.bias, out=x[i, :lt])
            a = ops.patchify(hidden_states[i].permute(1, 0, 2, 3).contiguous(), (1, ps, ps))
            ops.gemm(a, pk.w_patch, pe.proj.bias, out=x[i, lt:])
            ops.gated_residual(x[i], pos, out=x[i])
        x2 = x.view(b * L, d)
        cos = sin = None
        if image_rotary_emb is not None:
            cos, sin = (t.to(dev).float().contiguous() for t in image_rotary_emb)

        fold = self.fold_softmax_scale and hasattr(ops, "SCALE_FOLDED")
        qfold = {"out_scale": dh ** -0.5 * ops.LOG2E} if fold else {}
        afold = {"scale": ops.SCALE_FOLDED} if fold else {}
        # rows of the joint sequence whose output anyone reads after the last block: the video rows of the live frames
        tpf = (hh // ps) * (ww // ps)
        kf = nf if live_frames is None else max(1, min(int(live_frames), nf))
        r0, r1 = lt, lt + kf * tpf
        skip_dead = (self.skip_dead_rows and default_procs and not self._fp8 and not self.fp8_attention and len(self.transformer_blocks) > 1
                     and (r1 - r0) < L)
        if not skip_dead:
            kf, r0, r1 = nf, lt, L
        # 3. blocks (:503-529)
        for li, (blk, e) in enumerate(zip(self.transformer_blocks, pk.layers)):
            t1, t2 = tables[2 * li], tables[2 * li + 1]                              # [2B, 3, D]
            if skip_dead and li == len(self.transformer_blocks) - 1:
                # LAST block: every row is a key / value, only rows [r0, r1) of each sample are queries and go on
                n = ops.layernorm_zero(x2, e.n1w, e.n1b, t1[:, 0], t1[:, 1], sel, c.norm_eps)
                qkv = self._lin(li, "qkv", n, e.wqkv, e.bqkv).view(b, L, 3 * d)
                nq, nk = blk.attn1.norm_q, blk.attn1.norm_k
                ops.headnorm_rope_(qkv[:, :, :d], heads, dh, nq.weight, nq.bias, nq.eps, cos, sin, rope_row0=lt, **qfold)
                ops.headnorm_rope_(qkv[:, :, d:2 * d], heads, dh, nk.weight, nk.bias, nk.eps, cos, sin, rope_row0=lt)
                att = ops.attention(qkv[:, r0:r1, :d], qkv[:, :, d:2 * d], qkv[:, :, 2 * d:], heads, **afold)      # [B, r1 - r0, D]
                for i in range(b):
                    xs, ss = x2[i * L + r0:i * L + r1], sel[i * L + r0:i * L + r1]
                    self._lin(li, "out", att[i], blk.attn1.to_out[0].weight, blk.attn1.to_out[0].bias,
                              ops.EPI_GATED_RESIDUAL_STAGED, residual=xs, gate=t1[:, 2], sel=ss, out=xs)
                    n = ops.layernorm_zero(xs, e.n2w, e.n2b, t2[:, 0], t2[:, 1], ss, c.norm_eps)
                    ff = self._lin(li, "ff1", n, blk.ff.net[0].proj.weight, blk.ff.net[0].proj.bias, ops.EPI_GELU_TANH)
                    self._lin(li, "ff2", ff, blk.ff.net[2].weight, blk.ff.net[2].bias, ops.EPI_GATED_RESIDUAL_STAGED,
                              residual=xs, gate=t2[:, 2], sel=ss, out=xs)
                continue
            xq1 = self._lnz_q(li, "qkv", x2, e.n1w, e.n1b, t1[:, 0], t1[:, 1], sel, c.norm_eps) if default_procs else None
            n = None if xq1 is not None else ops.layernorm_zero(x2, e.n1w, e.n1b, t1[:, 0], t1[:, 1], sel, c.norm_eps)
            if default_procs:
                qkv = self._lin(li, "qkv", n, e.wqkv, e.bqkv, xq=xq1).view(b, L, 3 * d)
                nq, nk = blk.attn1.norm_q, blk.attn1.norm_k
                ops.headnorm_rope_(qkv[:, :, :d], heads, dh, nq.weight, nq.bias, nq.eps, cos, sin, rope_row0=lt, **qfold)
                ops.headnorm_rope_(qkv[:, :, d:2 * d], heads, dh, nk.weight, nk.bias, nk.eps, cos, sin, rope_row0=lt)
                if self.fp8_attention and dh == 64:
                    att = ops.attention_fp8(qkv[:, :, :d], qkv[:, :, d:2 * d], qkv[:, :, 2 * d:], heads,
                                            p_mode=getattr(self, "fp8_p_mode", None), **afold)
                else:
                    att = ops.attention(qkv[:, :, :d], qkv[:, :, d:2 * d], qkv[:, :, 2 * d:], heads, **afold)
                self._lin(li, "out", att.view(b * L, d), blk.attn1.to_out[0].weight, blk.attn1.to_out[0].bias,
                          ops.EPI_GATED_RESIDUAL_STAGED, residual=x2, gate=t1[:, 2], sel=sel, out=x2)
            else:
                n3 = n.view(b, L, d)
                ah, ae = blk.attn1(hidden_states=n3[:, lt:], encoder_hidden_states=n3[:, :lt],
                                   image_rotary_emb=image_rotary_emb, **(attention_kwargs or {}))
                y = torch.cat([ae, ah], dim=1).reshape(b * L, d)
                ops.gated_residual(x2, y, t1[:, 2], sel, out=x2, staged=True)
            fp8 = self._fp8
            w1q, w2q = fp8.get((li, "ff1")), fp8.get((li, "ff2"))
            xq2 = (self._lnz_q(li, "ff1", x2, e.n2w, e.n2b, t2[:, 0], t2[:, 1], sel, c.norm_eps)
                   if (w1q is not None and w2q is not None) else None)
            n = None if xq2 is not None else ops.layernorm_zero(x2, e.n2w, e.n2b, t2[:, 0], t2[:, 1], sel, c.norm_eps)
            if w1q is not None and w2q is not None:
                hq = ops.gemm_mxfp8_q(*(xq2 if xq2 is not None else ops.quantize_mxfp8(n)), w1q[0], w1q[1],
                                      blk.ff.net[0].proj.bias, ops.EPI_GELU_TANH)
                ops.gemm_mxfp8(hq[0], hq[1], w2q[0], w2q[1], blk.ff.net[2].bias, ops.EPI_GATED_RESIDUAL_STAGED,
                               residual=x2, gate=t2[:, 2], sel=sel, out=x2)
            else:
                ff = self._lin(li, "ff1", n, blk.ff.net[0].proj.weight, blk.ff.net[0].proj.bias, ops.EPI_GELU_TANH)
                self._lin(li, "ff2", ff, blk.ff.net[2].weight, blk.ff.net[2].bias, ops.EPI_GATED_RESIDUAL_STAGED,
                          residual=x2, gate=t2[:, 2], sel=sel, out=x2)

        # 4. final norms + projection (:531-542) on the video rows
        outs = []
        for i in range(b):
            v = ops.layernorm(x[i, lt:r1], pk.nfw, pk.nfb, c.norm_eps)
            v = ops.layernorm_zero(v, pk.now, pk.nob, out_mod[i, 0], out_mod[i, 1], None, c.norm_eps)
            y = ops.gemm(v, self.proj_out.weight, self.proj_out.bias)                # [Lv, p*p*Cout] (c, dh, dw) columns
            y = y.view(kf, hh // ps, ww // ps, c.out_channels, ps, ps).permute(0, 3, 1, 4, 2, 5)
            outs.append(y.reshape(kf, c.out_channels, hh, ww))
        out = torch.stack(outs)
        if kf < nf:                                                                  # frames nobody reads: zeros
            out = torch.cat([out, out.new_zeros((b, nf - kf) + tuple(out.shape[2:]))], dim=1)
        if not return_dict:
            return (out,)
        return SimpleNamespace(sample=out)
