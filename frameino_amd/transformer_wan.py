"""MI355X-native WanTransformer3DModel (Wan2.2-TI2V-5B DiT as used by FrameINO).

Host-side mirror of /root/reference/architecture/transformer_wan.py: same class / parameter names (HF checkpoints
load by key), same `forward(hidden_states, timestep, encoder_hidden_states, encoder_hidden_states_image=None,
return_dict=True, attention_kwargs=None)` signature, same `blocks[i].attn1/attn2` + processor plugin surface,
`cache_context(name)`, `.config`, `.dtype`.  All arithmetic runs in the HIP kernels of libframeino_hip.so.

What is different by design (results identical):
  * the per-token timestep embedding is de-duplicated: R distinct timestep values -> R rows through the time MLP
    and an [R, layers, 6, D] fp32 modulation table + an int32 per-token selector, instead of the reference's
    [1, L, 6, D] fp32 tensors (908 MB per block at L=12320);
  * RoPE tables are built once per (frames, height, width) and kept compact ([L, 64] cos/sin);
  * text K/V of the cross-attention (step-invariant) are cached per `cache_context` name;
  * Q/K/V are one fused GEMM; norm+RoPE run in place on the fused buffer; attention reads it in place;
    bias / GELU / gated-residual are GEMM epilogues.
"""
import contextlib
import math
import os
from types import SimpleNamespace

import torch
from torch import nn

from . import ops
from .attention_processor import Attention, MI355WanAttnProcessor
from .loading import FromPretrainedMixin


class _Config(dict):
    __getattr__ = dict.__getitem__


class _FP32LayerNormParams(nn.Module):
    """Parameter holder for diffusers' FP32LayerNorm (weight/bias stay fp32: reference :393)."""

    def __init__(self, dim, eps, elementwise_affine):
        super().__init__()
        self.eps = eps
        if elementwise_affine:
            self.weight = nn.Parameter(torch.ones(dim))
            self.bias = nn.Parameter(torch.zeros(dim))
        else:
            self.register_parameter("weight", None)
            self.register_parameter("bias", None)


class _GELUProj(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out)


class FeedForward(nn.Module):
    """Parameter layout of diffusers FeedForward("gelu-approximate"): net.0.proj, net.2."""

    def __init__(self, dim, inner_dim):
        super().__init__()
        self.net = nn.ModuleList([_GELUProj(dim, inner_dim), nn.Dropout(0.0), nn.Linear(inner_dim, dim)])


class _MLP2(nn.Module):
    def __init__(self, d_in, d_hidden, d_out=None):
        super().__init__()
        self.linear_1 = nn.Linear(d_in, d_hidden)
        self.linear_2 = nn.Linear(d_hidden, d_out or d_hidden)


class WanTimeTextImageEmbedding(nn.Module):
    """Parameters of reference :146-166 (image_embedder absent: image_dim=None for TI2V-5B)."""

    def __init__(self, dim, time_freq_dim, time_proj_dim, text_embed_dim):
        super().__init__()
        self.time_embedder = _MLP2(time_freq_dim, dim)
        self.time_proj = nn.Linear(dim, time_proj_dim)
        self.text_embedder = _MLP2(text_embed_dim, dim)


def wan_rope_tables(head_dim, max_seq_len, ppf, pph, ppw, theta=10000.0):
    """Compact RoPE tables [L, head_dim/2] (cos, sin) equal to WanRotaryPosEmbed (reference :192-253) sampled at
    the slots the processor reads (:82-83).  fp64 angle tables, stored fp32, as the reference."""
    h_dim = w_dim = 2 * (head_dim // 6)
    t_dim = head_dim - h_dim - w_dim

    def axis(dim):
        freqs = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float64)[: dim // 2] / dim))
        ang = torch.outer(torch.arange(max_seq_len, dtype=torch.float64), freqs)
        return ang.cos().float(), ang.sin().float()          # [S, dim/2] (pairwise-repeated values deduplicated)

    (ct, st), (ch, sh), (cw, sw) = axis(t_dim), axis(h_dim), axis(w_dim)

    def build(t, h, w):
        a = t[:ppf].view(ppf, 1, 1, -1).expand(ppf, pph, ppw, -1)
        b = h[:pph].view(1, pph, 1, -1).expand(ppf, pph, ppw, -1)
        c = w[:ppw].view(1, 1, ppw, -1).expand(ppf, pph, ppw, -1)
        return torch.cat([a, b, c], dim=-1).reshape(ppf * pph * ppw, -1).contiguous()

    return build(ct, ch, cw), build(st, sh, sw)


class WanTransformerBlock(nn.Module):
    def __init__(self, dim, ffn_dim, num_heads, qk_norm="rms_norm_across_heads", cross_attn_norm=False, eps=1e-6):
        super().__init__()
        self.attn1 = Attention(dim, heads=num_heads, dim_head=dim // num_heads, qk_norm=qk_norm, eps=eps, bias=True,
                               out_bias=True, processor=MI355WanAttnProcessor())
        self.attn2 = Attention(dim, heads=num_heads, dim_head=dim // num_heads, qk_norm=qk_norm, eps=eps, bias=True,
                               out_bias=True, processor=MI355WanAttnProcessor())
        self.norm2 = _FP32LayerNormParams(dim, eps, True) if cross_attn_norm else None
        self.ffn = FeedForward(dim, ffn_dim)
        self.scale_shift_table = nn.Parameter(torch.randn(1, 6, dim) / dim ** 0.5)


class WanTransformer3DModel(nn.Module, FromPretrainedMixin):
    _loader_name = "load_wan_transformer"
    _keep_in_fp32_modules = ["time_embedder", "scale_shift_table", "norm1", "norm2", "norm3"]   # reference :393

    def __init__(self, patch_size=(1, 2, 2), num_attention_heads=40, attention_head_dim=128, in_channels=16,
                 out_channels=16, text_dim=4096, freq_dim=256, ffn_dim=13824, num_layers=40, cross_attn_norm=True,
                 qk_norm="rms_norm_across_heads", eps=1e-6, image_dim=None, added_kv_proj_dim=None,
                 rope_max_seq_len=1024, pos_embed_seq_len=None):
        super().__init__()
        if image_dim is not None or added_kv_proj_dim is not None:
            raise NotImplementedError("Wan2.1 image-embedding branch is outside FrameINO's TI2V-5B path")
        self.config = _Config(patch_size=tuple(patch_size), num_attention_heads=num_attention_heads,
                              attention_head_dim=attention_head_dim, in_channels=in_channels,
                              out_channels=out_channels or in_channels, text_dim=text_dim, freq_dim=freq_dim,
                              ffn_dim=ffn_dim, num_layers=num_layers, cross_attn_norm=cross_attn_norm, qk_norm=qk_norm,
                              eps=eps, image_dim=image_dim, added_kv_proj_dim=added_kv_proj_dim,
                              rope_max_seq_len=rope_max_seq_len, pos_embed_seq_len=pos_embed_seq_len)
        inner = num_attention_heads * attention_head_dim
        self.inner_dim = inner
        self.patch_embedding = nn.Conv3d(in_channels, inner, kernel_size=patch_size, stride=patch_size)
        self.condition_embedder = WanTimeTextImageEmbedding(inner, freq_dim, inner * 6, text_dim)
        self.blocks = nn.ModuleList([WanTransformerBlock(inner, ffn_dim, num_attention_heads, qk_norm, cross_attn_norm,
                                                         eps) for _ in range(num_layers)])
        self.proj_out = nn.Linear(inner, self.config.out_channels * math.prod(patch_size))
        self.scale_shift_table = nn.Parameter(torch.randn(1, 2, inner) / inner ** 0.5)
        self._packed = None
        self._rope_cache = {}
        self._text_cache = {}
        self._ws = {}
        self._ctx_name = None
        self.ops = ops            # kernel front end (tests of the sharding logic inject a CPU stand-in)
        self.parallel = None      # frameino_amd.parallel.TokenShard or None
        self._fp8 = {}            # (layer, linear) -> (e4m3 weight bytes, MX scales); see enable_mxfp8_linears
        self._fp8_pending = False # MXFP8 was on when the parameters last moved / changed: re-quantise at the next forward
        self.dedup_shared_prefix = True   # A/B knob: CFG-batched call computes the branch-invariant prefix once
        # True: q of the self-attention leaves its norm + RoPE kernel already multiplied by head_dim**-0.5 * log2(e) (fp32,
        # one rounding) and the attention kernels take q.k as the exp2 argument (FINO_ATTN_SCALE_FOLDED), which lets the
        # 4-wave kernel fold the running maximum into its MFMAs.  False (default): q rounded where the reference rounds it,
        # scale applied to the logits -- inside the power-capped step the fold buys nothing.
        self.fold_softmax_scale = False
        self.fp8_attention = False        # see enable_fp8_attention
        # The text cross-attention over a ZERO-PADDED prompt (pipeline_wan...FrameINO.py:235-238 pads every prompt to 512 tokens
        # with zero rows): all padding tokens yield the same K and V row, so the real tokens + ONE key that stands for the run
        # give the same softmax (ops.attention_tail: softmax(q.[K; k x M]) [V; v x M] = softmax(q.[K; k] + [0; ln M]) [V; v]).
        # Found per prompt from the embeddings themselves (trailing all-zero rows), default processors only; False = attend to
        # all 512 rows as the reference does.
        self.dedup_text_padding = os.environ.get("FINO_TEXT_FOLD", "1") != "0"      # (the environment switch: A/B timing)
        # ... and then the text cross-attention's out-projection re-associated as P.(V W_o^T) (see _text_out_weights)
        self.reassociate_text_out = os.environ.get("FINO_TEXT_REASSOC", "1") != "0"
        # honour `forward(live_rows=...)`: in the last block, rows the caller discards contribute K | V only (round 6)
        self.skip_dead_rows = os.environ.get("FINO_SKIP_DEAD_ROWS", "1") != "0"

    # ------------------------------------------------------------------ diffusers-style surface
    @property
    def dtype(self):
        return self.proj_out.weight.dtype

    @property
    def device(self):
        return self.proj_out.weight.device

    @contextlib.contextmanager
    def cache_context(self, name):
        prev, self._ctx_name = self._ctx_name, name
        try:
            yield
        finally:
            self._ctx_name = prev

    @property
    def attn_processors(self):
        out = {}
        for i, blk in enumerate(self.blocks):
            out[f"blocks.{i}.attn1.processor"] = blk.attn1.processor
            out[f"blocks.{i}.attn2.processor"] = blk.attn2.processor
        return out

    def load_reference_state_dict(self, sd, dtype=None):
        """Load weights keyed by the reference's parameter names; non-fp32-island tensors are cast to `dtype`."""
        own = self.state_dict()
        missing = [k for k in own if k not in sd]
        extra = [k for k in sd if k not in own]
        if missing or extra:
            raise KeyError(f"state-dict mismatch: missing {missing[:5]} unexpected {extra[:5]}")
        with torch.no_grad():
            for k, p in self.named_parameters():
                keep32 = any(s in k for s in self._keep_in_fp32_modules)
                t = sd[k].to(torch.float32 if keep32 else (dtype or sd[k].dtype))
                p.data = t.to(p.device).contiguous()
        self.reset_caches()
        return self

    # ------------------------------------------------------------------ derived state
    def reset_caches(self):
        """Drop everything derived from the parameters or the prompt: packed/fused weight copies, MXFP8 weights, text
        K/V, RoPE tables, workspaces.  Called whenever the parameters may have changed or moved."""
        had_fp8 = bool(self._fp8) or self._fp8_pending
        self._packed = None
        self._fp8 = {}
        # the MXFP8 weights are re-quantised lazily, by the next forward, from wherever the parameters are then: a
        # `.to("cpu")` / `.float()` in between must not run the GPU quantiser on host tensors or leave the module half moved
        self._fp8_pending = had_fp8
        self._text_cache.clear()
        self._rope_cache.clear()
        self._ws.clear()
        return had_fp8

    def _apply(self, fn, *args, **kwargs):          # .to() / .cuda() / .half() / .float(): parameters move or change
        out = super()._apply(fn, *args, **kwargs)
        if hasattr(self, "_text_cache"):
            self.reset_caches()
        return out

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self.reset_caches()
        return out

    def _default_processors(self):
        # exactly our processor (not a subclass): the fused path in forward_steps IS that processor's body; anything
        # else the user installed (set_processor, at any time) is called through the plugin protocol instead.
        return all(type(b.attn1.processor) is MI355WanAttnProcessor and type(b.attn2.processor) is MI355WanAttnProcessor
                   for b in self.blocks)

    def _pack(self):
        """One-time repack for the fused kernels: fused QKV / KV weights, flat patch-embed weight, the 30 per-block
        scale_shift_tables stacked (fp32)."""
        d = self.inner_dim
        pk = SimpleNamespace(layers=[])
        for blk in self.blocks:
            a1, a2 = blk.attn1, blk.attn2
            e = SimpleNamespace()
            e.wqkv = torch.cat([a1.to_q.weight, a1.to_k.weight, a1.to_v.weight]).detach().contiguous()
            e.bqkv = torch.cat([a1.to_q.bias, a1.to_k.bias, a1.to_v.bias]).detach().contiguous()
            e.wkv2 = torch.cat([a2.to_k.weight, a2.to_v.weight]).detach().contiguous()
            e.bkv2 = torch.cat([a2.to_k.bias, a2.to_v.bias]).detach().contiguous()
            pk.layers.append(e)
        pk.sst = torch.stack([b.scale_shift_table.detach().float()[0] for b in self.blocks])        # [layers, 6, D]
        pk.w_patch = self.patch_embedding.weight.detach().reshape(d, -1).contiguous()
        self._packed = pk
        return pk

    # ------------------------------------------------------------------ MXFP8 linears (optional; BASELINE config 5)
    def enable_mxfp8_linears(self, enabled=True):
        """Run the six large linears of every block (QKV, attention out, cross-attention q / out, FFN up / down) on the
        MXFP8 path (`fino_quantize_mxfp8` + `fino_gemm_mxfp8`: OCP e4m3 with one e8m0 scale per 32 K-elements, fp32
        accumulate): weights are quantised once here, activations per call.  Attention, norms, modulation, embeddings
        and the output head stay in the model dtype.  There is no reference counterpart: the result is
        compared with this model's own bf16 forward (tests/test_mxfp8_gpu.py)."""
        self._fp8 = {}
        self._fp8_pending = False
        if not enabled:
            return self
        pk = self._packed or self._pack()
        o = self.ops
        for li, (blk, e) in enumerate(zip(self.blocks, pk.layers)):
            for key, w in (("qkv", e.wqkv), ("out", blk.attn1.to_out[0].weight), ("q2", blk.attn2.to_q.weight),
                           ("out2", blk.attn2.to_out[0].weight), ("ff1", blk.ffn.net[0].proj.weight),
                           ("ff2", blk.ffn.net[2].weight)):
                self._fp8[(li, key)] = o.quantize_mxfp8(w.detach().contiguous())
        return self

    def enable_fp8_attention(self, enabled=True, p_mode=None):
        """The 3-D self-attention (transformer_wan.py:108) with fp8 (e4m3) matrix operands -- q, k, v and P on the
        block-scaled fp8 MFMA, softmax and accumulation fp32 (fino_attn_fwd_fp8, head_dim 128 as two 64-channel sub-heads).
        Opt-in, single-GPU forward only; no reference counterpart: rel-RMS ~5e-2 per attention output on N(0, 1) inputs
        (tests/test_attention_fp8_gpu.py).  The text cross-attention stays bf16.
        p_mode: "exp2" | "ramp" -- how a softmax weight becomes its e4m3 byte (ops.FP8_P_*; None = ops.FP8_P_DEFAULT)."""
        self.fp8_attention = bool(enabled)
        self.fp8_p_mode = p_mode
        return self

    def _ln_q(self, li, key, mode, x, **ln):
        """the LayerNorm in front of linear (li, key) emitted directly as that linear's MXFP8 activations (fino_ln_mxfp8: one
        pass instead of norm -> bf16 -> quantise), or None when the linear is not on the MXFP8 path"""
        o = self.ops
        if not self._fp8 or (li, key) not in self._fp8 or not hasattr(o, "ln_mxfp8") or _NO_LN_MXFP8:
            return None
        return o.ln_mxfp8(mode, x, **ln)

    def _lin(self, li, key, x, w, b, epi=0, xq=None, **kw):
        """one of a block's large linears: MXFP8 when enabled (and K is a multiple of 128), else the model-dtype GEMM.
        xq: the activations already quantised (_ln_q)"""
        wq = self._fp8.get((li, key)) if self._fp8 else None
        o = self.ops
        if wq is None and self._fp8 and key in ("kv", "q"):
            # token shards project K|V and Q separately: quantise those row blocks of the fused weight on first use
            # (MX scales are per output row, so this equals slicing the quantised fused weight)
            wq = self._fp8[(li, key)] = o.quantize_mxfp8(w.detach().contiguous())
        if wq is None:
            return o.gemm(x, w, b, epi, **kw)
        kw.pop("tile_m", None)                                  # (the MXFP8 GEMM has one tile height)
        xq, xs = xq if xq is not None else o.quantize_mxfp8(x)
        return o.gemm_mxfp8(xq, xs, wq[0], wq[1], b, epi, **kw)

    def _workspace(self, L, dtype, device):
        # one workspace per (shape, cache_context name): the two CFG branches may run concurrently on two streams
        key = (L, dtype, str(device), self._ctx_name)
        ws = self._ws.get(key)
        if ws is None:
            d, f = self.inner_dim, self.config.ffn_dim
            mk = lambda *s: torch.empty(*s, dtype=dtype, device=device)          # noqa: E731
            ws = SimpleNamespace(x=mk(L, d), n=mk(L, d), qkv=mk(L, 3 * d), att=mk(L, d), q2=mk(L, d), ff=mk(L, f),
                                 a=mk(L, self.config.in_channels * math.prod(self.config.patch_size)),
                                 po=mk(L, self.config.out_channels * math.prod(self.config.patch_size)))
            self._ws = {k: v for k, v in self._ws.items() if k[:3] == key[:3]}      # drop other shapes, keep branches
            self._ws[key] = ws
        return ws

    def _rope(self, ppf, pph, ppw, device):
        key = (ppf, pph, ppw, str(device))
        if key not in self._rope_cache:
            cos, sin = wan_rope_tables(self.config.attention_head_dim, self.config.rope_max_seq_len, ppf, pph, ppw)
            self._rope_cache[key] = (cos.to(device), sin.to(device))
        return self._rope_cache[key]

    # ------------------------------------------------------------------ conditioning
    def _time_rows(self, t_rows, act_dtype):
        """Reference :175-183 on R distinct timestep values.  Returns temb [R, D] (act dtype) and
        timestep_proj [R, 6, D] (act dtype)."""
        ce = self.condition_embedder
        half = self.config.freq_dim // 2
        expo = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=t_rows.device) / half)
        ang = t_rows.float()[:, None] * expo[None, :]
        sinus = torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)               # flip_sin_to_cos=True
        te = ce.time_embedder
        wd = te.linear_1.weight.dtype                     # fp32 when the reference's fp32 islands are kept (:393)
        silu = torch.nn.functional.silu
        outs = []
        for i in range(0, sinus.shape[0], 16):                                    # skinny kernel: <= 16 rows/call
            s = sinus[i:i + 16].to(wd).float().contiguous()                       # :179-181
            h = self.ops.skinny_linear(s, te.linear_1.weight, te.linear_1.bias).to(wd)
            h = silu(h).float().contiguous()
            e = self.ops.skinny_linear(h, te.linear_2.weight, te.linear_2.bias).to(wd)
            temb = e.to(act_dtype)                                                # .type_as(encoder_hidden_states) :182
            a = silu(temb).float().contiguous()                                   # act_fn(temb) in the act dtype :183
            tp = self.ops.skinny_linear(a, ce.time_proj.weight, ce.time_proj.bias).to(ce.time_proj.weight.dtype)
            outs.append((temb, tp))
        temb = torch.cat([o[0] for o in outs])
        tproj = torch.cat([o[1] for o in outs]).unflatten(1, (6, -1))
        return temb, tproj

    def _qk_norm_rope(self, blk, qkv, d, cos, sin, dh, qfold):
        """norm_q / norm_k ("rms_norm_across_heads", :64-67) + RoPE (:73-90) on the q and k columns of the fused projection,
        in place: one launch where the ops offer it (the arithmetic of the two rmsnorm_rope_ calls, bit for bit)"""
        o, a = self.ops, blk.attn1
        if hasattr(o, "qkv_rmsnorm_rope_"):
            o.qkv_rmsnorm_rope_(qkv, d, a.norm_q.weight, a.norm_q.eps, a.norm_k.weight, a.norm_k.eps, cos, sin, dh,
                                q_out_scale=qfold.get("out_scale", 1.0))
        else:
            o.rmsnorm_rope_(qkv[:, :d], a.norm_q.weight, a.norm_q.eps, cos, sin, dh, **qfold)
            o.rmsnorm_rope_(qkv[:, d:2 * d], a.norm_k.weight, a.norm_k.eps, cos, sin, dh)

    def _text_tail(self, ehs, lq):
        """(rows to keep, per-sample key counts, per-sample multiplicities) when every sample of the prompt batch ends in a run of
        all-zero rows worth folding into one key, else None.  One host read per prompt (the result is cached with the K / V)."""
        o = self.ops
        b, lt, _ = ehs.shape
        if not (self.dedup_text_padding and hasattr(o, "attention_tail")):
            return None
        live = (ehs != 0).any(dim=-1)                                                  # [b, lt]
        last = torch.where(live.any(dim=1), lt - 1 - live.flip(1).int().argmax(dim=1), torch.full((b,), -1, device=ehs.device))
        n_real = [int(x) + 1 for x in last.tolist()]                                   # tokens before the zero run
        if max(n_real) >= lt:                                                          # a prompt without padding
            return None
        keep = max(128, -(-(max(n_real) + 1) // 64) * 64)                              # >= 2 key tiles for the walking kernel
        heads = self.config.num_attention_heads
        if keep > lt - 64 or not o.attention_tail_supported(b, heads, lq, keep, self.inner_dim // heads):
            return None
        return keep, [n + 1 for n in n_real], [float(lt - n) for n in n_real]

    def _text_out_weights(self, val, b, lq, pk):
        """(P.V) W_o^T = P.(V W_o^T): with the padding run folded a sample's text is lk <= ~130 keys, so per layer and sample
        W2 = [W_o,h V_h^T]_h  ([D, heads x kp], kp = lk rounded up to 8) is a per-prompt constant and the out-projection of the
        text cross-attention (:117 after :108) a GEMM over K = heads x kp (1728 / 384 at 64 / 8 prompt tokens) instead of 3072,
        fed by the attention PROBABILITIES (ops.attention_probs).  One GEMM per layer and sample builds it: W_o against the
        block-diagonal arrangement of V (row h kp + j = V[j] restricted to head h's channels)."""
        o, d = self.ops, self.inner_dim
        heads = self.config.num_attention_heads
        dh = d // heads
        lk_b = val.tail[0]
        kps = [-(-lk // 8) * 8 for lk in lk_b]
        dev, dt = val.txt.device, val.txt.dtype
        hidx = torch.arange(heads, device=dev)
        val.kp = kps
        val.pbuf = [torch.empty((lq, heads * kp), dtype=dt, device=dev) for kp in kps]
        val.rrms = torch.empty(b * lq, dtype=torch.float32, device=dev)
        val.w2 = []
        for li, blk in enumerate(self.blocks):
            kv = val.kv[li].view(b, val.lt, 2 * d)
            per_sample = []
            for i, kp in enumerate(kps):
                vblk = torch.zeros((heads, kp, heads, dh), dtype=dt, device=dev)
                vblk[hidx, :, hidx, :] = kv[i, :kp, d:].reshape(kp, heads, dh).permute(1, 0, 2)
                per_sample.append(o.gemm(blk.attn2.to_out[0].weight, vblk.view(heads * kp, d)))       # [D, heads*kp]
            val.w2.append(per_sample)
        if dt == torch.float16:
            # fp16 only (round 6): V W_o^T is STORED in the model dtype.  A text key with a large value vector and a vanishing
            # probability is harmless in the reference's order -- (P.V) stays finite -- but an element of V W_o^T beyond 65504
            # is inf here and P.inf = nan for every query.  One host read per prompt (this runs when the per-prompt cache is
            # built, never inside a captured step): if anything left the range, this prompt keeps the ordinary out-projection.
            finite = torch.stack([torch.isfinite(w).all() for layer in val.w2 for w in layer]).all()
            if not bool(finite):
                val.w2, val.pbuf, val.rrms = None, None, None

    def _text_kv(self, encoder_hidden_states, pk, lq=0, may_fold=False):
        """text_embedder (:185) + the 30 layers' attn2 K (after norm_k) and V: step-invariant, cached per
        cache_context name for as long as the caller passes the SAME prompt tensor object, unmodified.  `may_fold`: the
        zero-padded tail of the prompt may be folded into one key (`dedup_text_padding`): then only the kept rows are embedded
        and projected (row-wise operations: the kept rows come out bit-identical) and `.tail` carries (key counts, multiplicities)."""
        hit = self._text_cache.get(self._ctx_name)
        if hit is not None and hit[0] is encoder_hidden_states and hit[1] == encoder_hidden_states._version \
                and hit[3] == (bool(may_fold), int(lq) if may_fold else 0, bool(self._fp8) or self._fp8_pending, self.reassociate_text_out,
                               self.dedup_text_padding):
            return hit[2]
        ce = self.condition_embedder
        d = self.inner_dim
        tail = self._text_tail(encoder_hidden_states, lq) if may_fold else None
        kept = encoder_hidden_states if tail is None else encoder_hidden_states[:, :tail[0]].contiguous()
        ctx = kept.reshape(-1, kept.shape[-1])
        h = self.ops.gemm(ctx, ce.text_embedder.linear_1.weight, ce.text_embedder.linear_1.bias, self.ops.EPI_GELU_TANH)
        txt = self.ops.gemm(h, ce.text_embedder.linear_2.weight, ce.text_embedder.linear_2.bias)
        kvs = []
        for blk, e in zip(self.blocks, pk.layers):
            kv = self.ops.gemm(txt, e.wkv2, e.bkv2)                                     # [Lt, 2D]
            self.ops.rmsnorm_rope_(kv[:, :d], blk.attn2.norm_k.weight, blk.attn2.norm_k.eps)
            kvs.append(kv)
        val = SimpleNamespace(txt=txt, kv=kvs, lt=kept.shape[1], tail=None if tail is None else (tail[1], tail[2]), w2=None)
        # P.(V W_o^T) pays only while its K = heads x keys stays well under D (K = 1728 / 384 at 64 / 8 prompt tokens against
        # 3072): from ~77 kept keys on (24 heads) it costs the FLOPs it saves, plus the rrms / probabilities passes, a GEMM per
        # sample and 30 cached [D, heads x kp] matrices per sample (0.32 GB for a 64-token prompt, per cache context).  Longer prompts take the
        # folded attention + the ordinary out-projection.
        worth = tail is not None and self.config.num_attention_heads * (-(-max(tail[1]) // 8) * 8) <= 0.6 * d
        if worth and self.reassociate_text_out and not self._fp8 and not self._fp8_pending \
                and hasattr(self.ops, "attention_probs") \
                and self.ops.attention_probs_supported(1, self.config.num_attention_heads, lq, kept.shape[1],
                                                       d // self.config.num_attention_heads):
            self._text_out_weights(val, kept.shape[0], lq, pk)
        # the entry holds the prompt tensor itself: identity (`is`) + in-place version, never its address -- a freed
        # prompt's address is handed to the next same-shape prompt by the caching allocator
        self._text_cache[self._ctx_name] = (encoder_hidden_states, encoder_hidden_states._version, val,
                                            (bool(may_fold), int(lq) if may_fold else 0, bool(self._fp8) or self._fp8_pending,
                                             self.reassociate_text_out, self.dedup_text_padding))
        return val

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, hidden_states, timestep, encoder_hidden_states, encoder_hidden_states_image=None,
                return_dict=True, attention_kwargs=None, timestep_rows=None, live_rows=None):
        """`timestep_rows=(values [R], selector int32 [L])` is the de-duplicated form of a per-token timestep; when a
        2-D `timestep` is given instead it is de-duplicated here (torch.unique: host sync, eager only).
        `live_rows=(lo, hi)` (round 6): the caller will only read the output of token rows [lo, hi) -- the FrameINO loop drops the
        ID frame's prediction (pipeline_wan_i2v_motion_FrameINO.py:884-885) and re-imposes the first latent frame from the
        condition before every forward and at the end (:829, :913).  In the LAST block the other rows then only feed the
        self-attention as keys and values: their q projection's use, attention queries, out-projection, text branch, FFN and
        output head are skipped, and their part of the returned tensor is ZERO.  Every kept row is computed exactly as without it.
        A bare `transformer(...)` call (None) keeps the full output."""
        gen = self.forward_steps(hidden_states, timestep, encoder_hidden_states, encoder_hidden_states_image,
                                 return_dict, attention_kwargs, timestep_rows, live_rows=live_rows)
        while True:
            try:
                next(gen)
            except StopIteration as done:
                return done.value

    def forward_steps(self, hidden_states, timestep, encoder_hidden_states, encoder_hidden_states_image=None,
                      return_dict=True, attention_kwargs=None, timestep_rows=None, shard=None, live_rows=None):
        """The forward as a generator that yields after the embedding stage and after every block, so that a caller
        can interleave two independent forwards (the CFG branches) kernel-stream by kernel-stream
        (frameino_amd/parallel.py: one branch's K|V all-gather then flies under the other branch's compute).  Every
        piece of per-call state (workspace, text K/V, shard) is bound at the first `next()`; `shard` overrides
        `self.parallel` for this call.  Returns (StopIteration.value) what `forward` returns."""
        if encoder_hidden_states_image is not None:
            raise NotImplementedError("encoder_hidden_states_image: Wan2.1 branch, outside the TI2V-5B path")
        if attention_kwargs is not None:
            attention_kwargs = dict(attention_kwargs)
            attention_kwargs.pop("scale", None)                                    # LoRA scale (:463-476): no PEFT here
        b = hidden_states.shape[0]
        if self._fp8_pending:
            self.enable_mxfp8_linears()
        pk = self._packed or self._pack()
        default_procs = self._default_processors()
        o = self.ops
        cfg = self.config
        _, c, nf, hh, ww = hidden_states.shape
        pt, ph, pw = cfg.patch_size
        ppf, pph, ppw = nf // pt, hh // ph, ww // pw
        L = ppf * pph * ppw
        d, heads, dh = self.inner_dim, cfg.num_attention_heads, cfg.attention_head_dim
        dev, dt = hidden_states.device, hidden_states.dtype
        cos1, sin1 = self._rope(ppf, pph, ppw, dev)

        # ---- token shard of this rank (frameino_amd/parallel.py); single GPU: the whole sequence ----
        sh = shard if shard is not None else self.parallel
        sh = sh if (sh is not None and sh.active) else None
        if sh is not None:
            if b != 1:
                raise NotImplementedError("token-sharded execution runs one sample per call")
            lo, n, lpad = sh.rows(L)              # first row, valid rows, padded shard length (equal on all ranks)
            cos1, sin1 = cos1[lo:lo + n].contiguous(), sin1[lo:lo + n].contiguous()
        else:
            lo, n, lpad = 0, L, L
        # the shard's tile height for its GEMM calls (a per-call argument of the C ABI; {} = the library's planner)
        tk = {}
        if sh is not None and getattr(sh, "gemm_tile_m", 0):
            tk = {"tile_m": sh.tile_m_for(n) if hasattr(sh, "tile_m_for") else sh.gemm_tile_m}
        # Batch elements are extra ROWS of the token-major buffers ([B*n, D]): every GEMM / norm is one launch over
        # both CFG branches (twice the tiles per launch, weights read once); attention and RoPE index rows per batch.
        nr = b * n
        ws = self._workspace(b * lpad, dt, dev)
        if b == 1:
            cos, sin = cos1, sin1
        else:
            key = ("rope_b", ppf, pph, ppw, b, str(dev))
            if key not in self._rope_cache:
                self._rope_cache[key] = (cos1.repeat(b, 1).contiguous(), sin1.repeat(b, 1).contiguous())
            cos, sin = self._rope_cache[key]

        # ---- timestep rows + selector (F7) ----
        # `same_rows`: every batch element sees the same modulation rows BY CONSTRUCTION (the selector of one element
        # repeated, or one scalar timestep) -- the precondition of the shared prefix below
        same_rows = False
        if timestep_rows is not None:
            t_rows, sel = timestep_rows
            same_rows = True
            if sel is not None and sh is not None:
                sel = sel[lo:lo + n].contiguous()
            if sel is not None and b > 1:
                sel = sel.repeat(b)
        elif timestep.ndim == 2:
            t_rows, inv = torch.unique(timestep.reshape(-1), return_inverse=True)
            sel = inv.to(torch.int32).contiguous()
            if sh is not None:
                sel = sel[lo:lo + n].contiguous()
            elif timestep.shape[0] == 1 and b > 1:                                 # one row of per-token values: broadcast
                sel = sel.repeat(b)
                same_rows = True
        else:
            t_rows = timestep.reshape(-1)
            sel = None if b == 1 else torch.arange(b, device=dev, dtype=torch.int32).repeat_interleave(n)
            if t_rows.numel() == 1 and b > 1:
                sel = None
                same_rows = True
        temb, tproj = self._time_rows(t_rows.to(dev), encoder_hidden_states.dtype)       # [R,D], [R,6,D]
        # per-layer modulation tables: scale_shift_table + temb.float()  (:317-319)  -> [R, layers, 6, D] fp32
        mod = (pk.sst[None] + tproj.float()[:, None]).contiguous()
        head = (self.scale_shift_table.float() + temb.float()[:, None]).contiguous()     # [R, 2, D]  (:522/:527)

        text = self._text_kv(encoder_hidden_states, pk, lq=n, may_fold=default_procs)      # (token shards too: n = the shard's rows)
        lt = text.lt

        # ---- patch embedding (:486-487): gather + GEMM (the gather is 9 MB; every rank builds it, keeps its rows) ----
        # `shared`: the batch elements are the SAME latent (the pipeline's CFG-batched call passes x.expand(2, ...)) under
        # the same timestep rows: everything up to the first text cross-attention is identical for them, so the patch
        # embedding and layer 0's self-attention branch run once and their result is copied (exactly what each element
        # would have computed).  A per-sample timestep ([b] or [b, L] values) takes the general path.
        shared = (self.dedup_shared_prefix and b > 1 and sh is None and hidden_states.stride(0) == 0 and default_procs
                  and same_rows)
        if b == 1 or shared:
            a_rows = o.patchify(hidden_states[0], cfg.patch_size)[lo:lo + n]
        else:
            a_rows = torch.cat([o.patchify(hidden_states[i], cfg.patch_size) for i in range(b)])
        x = ws.x[:nr]
        o.gemm(a_rows, pk.w_patch, self.patch_embedding.bias, out=x[:n] if shared else x, **tk)
        nrm, att, q2, ff = ws.n[:nr], ws.att[:nr], ws.q2[:nr], ws.ff[:nr]
        fold = self.fold_softmax_scale and hasattr(o, "SCALE_FOLDED")
        qfold = {"out_scale": dh ** -0.5 * o.LOG2E} if fold else {}
        afold = {"scale": o.SCALE_FOLDED} if fold else {}
        attend = o.attention
        if self.fp8_attention and sh is None and hasattr(o, "attention_fp8"):
            def attend(q_, k_, v_, heads_, **kw_):
                return o.attention_fp8(q_, k_, v_, heads_, p_mode=getattr(self, "fp8_p_mode", None), **kw_)
        yield

        # rows whose output the caller reads (`live_rows`): honoured in the last block on the single-GPU default-processor path
        live = None
        if (live_rows is not None and self.skip_dead_rows and default_procs and sh is None and len(self.blocks) > 1
                and not self._fp8 and attend is o.attention):
            l0, l1 = int(live_rows[0]), int(live_rows[1])
            if 0 <= l0 < l1 <= n and (l1 - l0) < n:
                live = (l0, l1)
        segs = [(0, nr)]                 # global row ranges the per-token operations of a block run on
        for li, (blk, e) in enumerate(zip(self.blocks, pk.layers)):
            m = mod[:, li]                                                        # [R, 6, D] view, row stride = layers*6*D
            last_live = live is not None and li == len(self.blocks) - 1
            if last_live:
                segs = [(bi * n + live[0], bi * n + live[1]) for bi in range(b)]
            # 1. self-attention (:334-336)
            once = shared and li == 0
            xq1 = None
            if not once:
                if default_procs and sh is None:
                    xq1 = self._ln_q(li, "qkv", 0, x, shift=m[:, 0], scale=m[:, 1], sel=sel, eps=cfg.eps)
                if xq1 is None:
                    o.adaln_modulate(x, m[:, 0], m[:, 1], sel, cfg.eps, out=nrm)
            if not default_procs:
                if sh is not None:
                    raise NotImplementedError("token-sharded execution needs the built-in MI355WanAttnProcessor")
                rot = _CompactRope((cos1, sin1))
                a = blk.attn1(nrm.view(b, n, d), rotary_emb=rot, **(attention_kwargs or {}))
                o.gated_residual(x, a.reshape(nr, d), m[:, 2], sel, out=x)
            elif sh is None and shared and li == 0:
                n1, q1, sel1 = nrm[:n], ws.qkv[:n], (None if sel is None else sel[:n])
                o.adaln_modulate(x[:n], m[:, 0], m[:, 1], sel1, cfg.eps, out=n1)
                self._lin(li, "qkv", n1, e.wqkv, e.bqkv, out=q1)
                self._qk_norm_rope(blk, q1, d, cos1, sin1, dh, qfold)
                q3 = q1.view(1, n, 3 * d)
                attend(q3[:, :, :d], q3[:, :, d:2 * d], q3[:, :, 2 * d:], heads, out=att[:n].view(1, n, d), **afold)
                self._lin(li, "out", att[:n], blk.attn1.to_out[0].weight, blk.attn1.to_out[0].bias, o.EPI_GATED_RESIDUAL,
                          residual=x[:n], gate=m[:, 2], sel=sel1, out=x[:n])
                for bi in range(1, b):
                    x[bi * n:(bi + 1) * n].copy_(x[:n])
            elif sh is None:
                qkv = ws.qkv[:nr]
                self._lin(li, "qkv", nrm, e.wqkv, e.bqkv, xq=xq1, out=qkv)
                self._qk_norm_rope(blk, qkv, d, cos, sin, dh, qfold)
                q3 = qkv.view(b, n, 3 * d)
                if last_live:            # every row is a key; only the live rows are queries
                    attend(q3[:, live[0]:live[1], :d], q3[:, :, d:2 * d], q3[:, :, 2 * d:], heads,
                           out=att.view(b, n, d)[:, live[0]:live[1]], **afold)
                else:
                    attend(q3[:, :, :d], q3[:, :, d:2 * d], q3[:, :, 2 * d:], heads, out=att.view(b, n, d), **afold)
            elif sh.heads_exchange_ok(heads):
                # heads exchange (frameino_amd/parallel.py): q | k | v of MY tokens -> all-to-all -> all tokens of MY
                # heads -> one attention launch over the whole sequence -> all-to-all back to the token owners
                ways = sh.ways
                hp = heads // ways
                dp = hp * dh
                qkv = ws.qkv[:n]
                self._lin(li, "qkv", nrm, e.wqkv, e.bqkv, out=qkv, **tk)
                # the heads travel in groups, every group its own all-to-all on the communicator's stream: while group g
                # is attended to, group g+1 arrives and group g-1's outputs leave
                lay = sh.heads_send_layout(heads, dh, lpad, dt, dev) if hasattr(o, "qkv_rmsnorm_rope_") else None
                if lay is not None:
                    # ONE launch: RMSNorm + RoPE write q and k straight into the send buffers (slice j: heads of rank j) and
                    # v follows as a scattering copy -- no permute copy of q | k | v afterwards
                    o.qkv_rmsnorm_rope_(qkv, d, blk.attn1.norm_q.weight, blk.attn1.norm_q.eps, blk.attn1.norm_k.weight,
                                        blk.attn1.norm_k.eps, cos, sin, dh, q_out_scale=qfold.get("out_scale", 1.0),
                                        out=lay.flat, head_off=lay.off_qkv, head_ld=lay.ld)
                else:
                    o.rmsnorm_rope_(qkv[:, :d], blk.attn1.norm_q.weight, blk.attn1.norm_q.eps, cos, sin, dh, **qfold)
                    o.rmsnorm_rope_(qkv[:, d:2 * d], blk.attn1.norm_k.weight, blk.attn1.norm_k.eps, cos, sin, dh)
                    q4 = qkv.view(n, 3, ways, dp)
                inflight = []
                for gi, (h0, h1) in enumerate(sh.head_ranges(hp)):
                    dg = (h1 - h0) * dh
                    if lay is not None:
                        send = lay.views[gi]
                    else:
                        send = sh.a2a_buffer(f"qkv_send{gi}", (ways, lpad, 3, dg), dt, dev)   # slice j: heads of rank j
                        send[:, :n].copy_(q4[:, :, :, h0 * dh:h1 * dh].permute(2, 0, 1, 3))
                    inflight.append((gi, h0, h1, dg) + sh.all_to_all(f"qkv_recv{gi}", send, async_op=True))
                # the outputs return into ONE flat [group][rank j's heads][token] buffer when the groups are equal: the
                # out-projection then reads it as K blocks (fino_gemm_blocked_a) instead of a permute copy into [token, D]
                orl = (sh.heads_recv_layout(heads, dh, lpad, dt, dev)
                       if (not self._fp8 and hasattr(o, "gemm_blocked_a") and hasattr(sh, "heads_recv_layout")) else None)
                back = []
                for gi, h0, h1, dg, recv, work in inflight:                                # slice j: tokens of rank j
                    if work is not None:
                        work.wait()
                    r3 = recv.view(1, ways * lpad, 3 * dg)[:, :L]
                    oh = sh.a2a_buffer(f"o_send{gi}", (ways, lpad, dg), dt, dev)
                    o.attention(r3[:, :, :dg], r3[:, :, dg:2 * dg], r3[:, :, 2 * dg:], h1 - h0,
                                out=oh.view(1, ways * lpad, dg)[:, :L], **afold)
                    back.append((h0, h1) + sh.all_to_all(f"o_recv{gi}", oh, async_op=True))
                if orl is not None and all(orv.data_ptr() == orl[gi].data_ptr() for gi, (_, _, orv, _) in enumerate(back)):
                    for _, _, _, work in back:
                        if work is not None:
                            work.wait()
                    o.gemm_blocked_a(orl, n, blk.attn1.to_out[0].weight, blk.attn1.to_out[0].bias, x, m[:, 2], sel, out=x,
                                     **tk)
                    once = True                                     # the common out-projection below is done
                else:
                    a3 = att.view(n, ways, dp)
                    for h0, h1, orv, work in back:                                         # slice j: heads of rank j
                        if work is not None:
                            work.wait()
                        a3[:, :, h0 * dh:h1 * dh].copy_(orv[:, :n].permute(1, 0, 2))
            elif (hasattr(sh, "kv_groups") and sh.kv_groups(heads) > 1 and hasattr(o, "rmsnorm_rope_scatter")
                  and hasattr(sh, "kv_group_layout")):
                # K|V all-gather in HEAD GROUPS (TokenShard.kv_head_groups): the attention of group g runs while group g+1 is
                # still on the wire.  k (norm_k + RoPE) is scattered by head into [group][token][k_g | v_g] send blocks, v
                # follows as a scattering copy; per group one all-gather and one attention launch over that group's heads.
                kv_loc = sh.kv_local(lpad, 2 * d, dt, dev)
                lay = sh.kv_group_layout(heads, dh, lpad, dt, dev)
                if sh.fused_qkv_ok() and not self._fp8:
                    o.gemm(nrm, e.wqkv, e.bqkv, out=q2, out2=kv_loc[:n], split=d, **tk)
                else:
                    self._lin(li, "kv", nrm, e.wqkv[d:], e.bqkv[d:], out=kv_loc[:n], **tk)
                o.rmsnorm_rope_scatter(kv_loc[:n, :d], blk.attn1.norm_k.weight, blk.attn1.norm_k.eps, cos, sin, dh, lay.flat,
                                       lay.off_k, lay.ld)
                o.rmsnorm_rope_scatter(kv_loc[:n, d:], None, 0.0, None, None, dh, lay.flat, lay.off_v, lay.ld)
                flying = [sh._all_gather(f"kv_all_g{gi}", v_, True) for gi, v_ in enumerate(lay.views)]
                if not (sh.fused_qkv_ok() and not self._fp8):
                    self._lin(li, "q", nrm, e.wqkv[:d], e.bqkv[:d], out=q2, **tk)
                o.rmsnorm_rope_(q2, blk.attn1.norm_q.weight, blk.attn1.norm_q.eps, cos, sin, dh, **qfold)
                q3, a3 = q2.view(1, n, d), att.view(1, n, d)
                for (h0, h1), (kv_all, work) in zip(lay.ranges, flying):
                    if work is not None:
                        work.wait()
                    dg = (h1 - h0) * dh
                    kv3 = kv_all.view(1, -1, 2 * dg)[:, :L]
                    o.attention(q3[:, :, h0 * dh:h1 * dh], kv3[:, :, :dg], kv3[:, :, dg:], h1 - h0,
                                out=a3[:, :, h0 * dh:h1 * dh], **afold)
            else:
                kv_loc = sh.kv_local(lpad, 2 * d, dt, dev)
                if sh.fused_qkv_ok() and not self._fp8:
                    # ONE q | k | v GEMM whose k | v columns land in the all-gather's send buffer (fino_gemm_split_n): the
                    # interleaved plan, where the OTHER branch's compute is what the gather flies under
                    o.gemm(nrm, e.wqkv, e.bqkv, out=q2, out2=kv_loc[:n], split=d, **tk)
                    o.rmsnorm_rope_(kv_loc[:n, :d], blk.attn1.norm_k.weight, blk.attn1.norm_k.eps, cos, sin, dh)
                    kv_all, work = sh.all_gather_kv(kv_loc)
                else:
                    # K|V of the local tokens first, so that their all-gather (RCCL over xGMI) overlaps the Q projection
                    self._lin(li, "kv", nrm, e.wqkv[d:], e.bqkv[d:], out=kv_loc[:n], **tk)
                    o.rmsnorm_rope_(kv_loc[:n, :d], blk.attn1.norm_k.weight, blk.attn1.norm_k.eps, cos, sin, dh)
                    kv_all, work = sh.all_gather_kv(kv_loc)
                    self._lin(li, "q", nrm, e.wqkv[:d], e.bqkv[:d], out=q2, **tk)
                o.rmsnorm_rope_(q2, blk.attn1.norm_q.weight, blk.attn1.norm_q.eps, cos, sin, dh, **qfold)
                if sh.local_first():
                    # local keys first -- nothing of it waits for the wire -- then what the gather delivered before /
                    # after the own chunk; the (O, m, l) partials are merged (same softmax up to fp32 summation order)
                    qv = q2.view(1, n, d)
                    pf = o.attention_partial_floats(1, heads, n, dh) if hasattr(o, "attention_partial_floats") else 0
                    parts = [o.attention_partial(qv, kv_loc[:n, :d][None], kv_loc[:n, d:][None], heads,
                                                 out=sh.partial_buf(0, pf, dev), **afold)]
                    if work is not None:
                        work.wait()
                    kv3 = kv_all.view(1, -1, 2 * d)
                    for pi, (k0, k1) in enumerate(((0, lo), (lo + lpad, L))):
                        if k1 > k0:
                            parts.append(o.attention_partial(qv, kv3[:, k0:k1, :d], kv3[:, k0:k1, d:], heads,
                                                             out=sh.partial_buf(1 + pi, pf, dev), **afold))
                    o.attention_merge(parts, 1, n, heads, dh, dt, out=att.view(1, n, d))
                else:
                    if work is not None:
                        work.wait()
                    kv3 = kv_all.view(1, -1, 2 * d)[:, :L]
                    o.attention(q2.view(1, n, d), kv3[:, :, :d], kv3[:, :, d:], heads, out=att.view(1, n, d), **afold)
            if default_procs and not once:
                for r0, r1 in segs:
                    self._lin(li, "out", att[r0:r1], blk.attn1.to_out[0].weight, blk.attn1.to_out[0].bias, o.EPI_GATED_RESIDUAL,
                              residual=x[r0:r1], gate=m[:, 2], sel=None if sel is None else sel[r0:r1], out=x[r0:r1], **tk)
            # 2. cross-attention (:339-341): text K/V are replicated, nothing to exchange
            n2 = blk.norm2
            xq2 = self._ln_q(li, "q2", 1, x, weight=n2.weight, bias=n2.bias, eps=cfg.eps) if (n2 is not None and
                                                                                                default_procs) else None
            if xq2 is not None:
                pass
            elif n2 is not None:
                for r0, r1 in segs:
                    o.layernorm(x[r0:r1], n2.weight, n2.bias, cfg.eps, out=nrm[r0:r1])
            else:
                nrm.copy_(x)
            if default_procs:
                for r0, r1 in segs:
                    self._lin(li, "q2", nrm[r0:r1], blk.attn2.to_q.weight, blk.attn2.to_q.bias, xq=xq2, out=q2[r0:r1], **tk)
                kv = text.kv[li].view(b, lt, 2 * d)
                s0, s1 = (live if last_live else (0, n))                       # the rows of every batch element that run
                if text.w2 is not None:
                    # norm_q's statistic only (the probabilities kernel normalises q while it loads it: same rounding points, no
                    # pass that rewrites q); probabilities per sample, then x += P.(V W_o^T) + b: K = heads x keys instead of D
                    rr = text.rrms
                    for r0, r1 in segs:
                        o.row_rrms(q2[r0:r1], blk.attn2.norm_q.eps, out=rr[r0:r1])
                    for i in range(b):
                        r0, r1 = i * n + s0, i * n + s1
                        pr = o.attention_probs(q2[r0:r1].view(1, s1 - s0, d), kv[i:i + 1, :, :d], heads, text.tail[0][i:i + 1],
                                               text.tail[1][i:i + 1], text.kp[i], out=text.pbuf[i][:s1 - s0].view(1, s1 - s0, -1),
                                               q_rrms=rr[r0:r1].view(1, s1 - s0), q_weight=blk.attn2.norm_q.weight)
                        xi = x[r0:r1]
                        o.gemm(pr.view(s1 - s0, -1), text.w2[li][i], blk.attn2.to_out[0].bias, o.EPI_RESIDUAL, residual=xi, out=xi, **tk)
                else:
                    for r0, r1 in segs:
                        o.rmsnorm_rope_(q2[r0:r1], blk.attn2.norm_q.weight, blk.attn2.norm_q.eps)
                    qv, av = q2.view(b, n, d)[:, s0:s1], att.view(b, n, d)[:, s0:s1]
                    if text.tail is not None:
                        o.attention_tail(qv, kv[:, :, :d], kv[:, :, d:], heads, text.tail[0], text.tail[1], out=av)
                    else:
                        o.attention(qv, kv[:, :, :d], kv[:, :, d:], heads, out=av)
                    for r0, r1 in segs:
                        self._lin(li, "out2", att[r0:r1], blk.attn2.to_out[0].weight, blk.attn2.to_out[0].bias, o.EPI_RESIDUAL,
                                  residual=x[r0:r1], out=x[r0:r1], **tk)
            else:
                a = blk.attn2(nrm.view(b, n, d), encoder_hidden_states=text.txt.view(b, lt, d),
                              **(attention_kwargs or {}))
                o.gated_residual(x, a.reshape(nr, d), out=x)
            # 3. feed-forward (:344-348)
            w1q, w2q = self._fp8.get((li, "ff1")), self._fp8.get((li, "ff2"))
            xq3 = (self._ln_q(li, "ff1", 0, x, shift=m[:, 3], scale=m[:, 4], sel=sel, eps=cfg.eps)
                   if (w1q is not None and w2q is not None) else None)
            if xq3 is None:
                for r0, r1 in segs:
                    o.adaln_modulate(x[r0:r1], m[:, 3], m[:, 4], None if sel is None else sel[r0:r1], cfg.eps, out=nrm[r0:r1])
            if w1q is not None and w2q is not None:
                # MXFP8: the adaLN emits the FFN input already quantised, and the GELU epilogue the hidden activations (no
                # bf16 round trips)
                hq = o.gemm_mxfp8_q(*(xq3 if xq3 is not None else o.quantize_mxfp8(nrm)), w1q[0], w1q[1],
                                    blk.ffn.net[0].proj.bias, o.EPI_GELU_TANH)
                o.gemm_mxfp8(hq[0], hq[1], w2q[0], w2q[1], blk.ffn.net[2].bias, o.EPI_GATED_RESIDUAL, residual=x,
                             gate=m[:, 5], sel=sel, out=x)
            else:
                for r0, r1 in segs:
                    self._lin(li, "ff1", nrm[r0:r1], blk.ffn.net[0].proj.weight, blk.ffn.net[0].proj.bias, o.EPI_GELU_TANH,
                              out=ff[r0:r1], **tk)
                    self._lin(li, "ff2", ff[r0:r1], blk.ffn.net[2].weight, blk.ffn.net[2].bias, o.EPI_GATED_RESIDUAL,
                              residual=x[r0:r1], gate=m[:, 5], sel=None if sel is None else sel[r0:r1], out=x[r0:r1], **tk)
            yield

        # ---- output head (:519-543) ----
        for r0, r1 in segs:
            o.adaln_modulate(x[r0:r1], head[:, 0], head[:, 1], None if sel is None else sel[r0:r1], cfg.eps, out=nrm[r0:r1])
        if sh is None:
            po = ws.po[:nr]
            if live is not None:
                po.zero_()               # rows nobody computed: the caller asked not to read them -- zeros, not the last call's values
            for r0, r1 in segs:
                o.gemm(nrm[r0:r1], self.proj_out.weight, self.proj_out.bias, out=po[r0:r1])
        else:
            po_loc = sh.out_local(lpad, ws.po.shape[1], dt, dev)
            o.gemm(nrm, self.proj_out.weight, self.proj_out.bias, out=po_loc[:n], **tk)
            po = sh.all_gather_out(po_loc)[:L]
        if b == 1:
            out = o.unpatchify(po, cfg.out_channels, nf, hh, ww, cfg.patch_size)[None]
        else:
            po3 = po.view(b, L, -1)
            out = torch.stack([o.unpatchify(po3[i], cfg.out_channels, nf, hh, ww, cfg.patch_size) for i in range(b)])
        if not return_dict:
            return (out,)
        return SimpleNamespace(sample=out)


_NO_LN_MXFP8 = bool(os.environ.get("FINO_NO_LN_MXFP8"))       # A/B knob: LayerNorm -> bf16 -> quantise as two passes


class _CompactRope(tuple):
    """(cos, sin) already in the compact [L, Dh/2] layout; recognised by MI355WanAttnProcessor."""
    compact = True
