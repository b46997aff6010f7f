"""Attention-processor plugin surface (mirror of the reference's architecture/attention_processor.py protocol)
with MI355X-native processors.

* `Attention` is the container the reference gets from diffusers (to_q/to_k/to_v/to_out/norm_q/norm_k/heads,
  `set_processor` / `get_processor`, kwarg filtering by the processor's signature -- protocol documented at
  /root/reference/architecture/attention_processor.py:522-600).  Parameter names equal diffusers', so HF
  checkpoints load by key.
* `MI355WanAttnProcessor` has the call signature of WanAttnProcessor2_0
  (/root/reference/architecture/transformer_wan.py:43-50) and `MI355CogVideoXAttnProcessor` /
  `MI355FusedCogVideoXAttnProcessor` those of CogVideoXAttnProcessor2_0 / FusedCogVideoXAttnProcessor2_0
  (/root/reference/architecture/attention_processor.py:2815-2822, :2890-2897).  They can be installed on any
  Attention-like module (including diffusers' own) and run the whole processor body on HIP kernels:
  fused-QKV MFMA GEMM -> RMSNorm/LayerNorm + RoPE in place -> flash attention reading the fused buffer in place
  -> output-projection GEMM.  There is no SDPA / eager fallback.
"""
import inspect

import torch
from torch import nn

from . import ops


class RMSNorm(nn.Module):
    """Parameter holder for diffusers' RMSNorm (weight only); arithmetic runs in fino_rmsnorm_rope."""

    def __init__(self, dim, eps):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))


class Attention(nn.Module):
    def __init__(self, query_dim, heads=8, kv_heads=None, dim_head=64, bias=False, qk_norm=None, eps=1e-5,
                 out_bias=True, cross_attention_dim=None, added_kv_proj_dim=None, processor=None, dropout=0.0,
                 **unused):
        super().__init__()
        if added_kv_proj_dim is not None:
            raise NotImplementedError("added_kv_proj_dim (Wan2.1 image branch) is outside FrameINO's TI2V-5B path")
        self.inner_dim = dim_head * heads
        self.heads = heads
        self.head_dim = dim_head
        self.scale = dim_head ** -0.5
        self.use_bias = bias
        self.is_cross_attention = cross_attention_dim is not None
        self.qk_norm = qk_norm
        self.eps = eps
        kv_in = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.to_q = nn.Linear(query_dim, self.inner_dim, bias=bias)
        self.to_k = nn.Linear(kv_in, self.inner_dim, bias=bias)
        self.to_v = nn.Linear(kv_in, self.inner_dim, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(self.inner_dim, query_dim, bias=out_bias), nn.Dropout(dropout)])
        if qk_norm is None:
            self.norm_q = self.norm_k = None
        elif qk_norm == "rms_norm_across_heads":
            self.norm_q = RMSNorm(self.inner_dim, eps)
            self.norm_k = RMSNorm(self.inner_dim, eps)
        elif qk_norm == "layer_norm":
            self.norm_q = nn.LayerNorm(dim_head, eps=eps)
            self.norm_k = nn.LayerNorm(dim_head, eps=eps)
        else:
            raise ValueError(f"unsupported qk_norm {qk_norm}")
        self.add_k_proj = self.add_v_proj = self.norm_added_k = None
        self.fused_projections = False
        self.processor = processor

    # ---- plugin protocol (reference attention_processor.py:522-600) ----
    def set_processor(self, processor):
        self.processor = processor

    def get_processor(self, return_deprecated_lora=False):
        return self.processor

    @torch.no_grad()
    def fuse_projections(self, fuse=True):
        """reference attention_processor.py:769-820: materialise to_qkv = cat(to_q, to_k, to_v)."""
        w = torch.cat([self.to_q.weight.data, self.to_k.weight.data, self.to_v.weight.data])
        self.to_qkv = nn.Linear(w.shape[1], w.shape[0], bias=self.use_bias, device=w.device, dtype=w.dtype)
        self.to_qkv.weight.copy_(w)
        if self.use_bias:
            self.to_qkv.bias.copy_(torch.cat([self.to_q.bias.data, self.to_k.bias.data, self.to_v.bias.data]))
        self.fused_projections = fuse

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **cross_attention_kwargs):
        params = set(inspect.signature(self.processor.__call__).parameters.keys())
        kwargs = {k: v for k, v in cross_attention_kwargs.items() if k in params}
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask, **kwargs)


# ------------------------------------------------------------------------------------------------ helpers
def _fused_weight(attn, names, key):
    """cat of nn.Linear weights/biases, cached on the module and rebuilt when a source tensor is replaced (`.to()`,
    a new Parameter) or modified in place (`load_state_dict`, optimiser step).  The entry keeps the source tensors
    themselves: identity + in-place version, never an address the allocator may hand to another tensor."""
    mods = [getattr(attn, n) for n in names]
    srcs = [t for m in mods for t in (m.weight, m.bias) if t is not None]
    cache = attn.__dict__.setdefault("_fino_cache", {})
    hit = cache.get(key)
    fresh = hit is not None and len(hit[0]) == len(srcs) and all(
        a is t and v == t._version and a.device == t.device for (a, v), t in zip(hit[0], srcs))
    if not fresh:
        w = torch.cat([m.weight.data for m in mods]).contiguous()
        b = torch.cat([m.bias.data for m in mods]).contiguous() if mods[0].bias is not None else None
        cache[key] = ([(t, t._version) for t in srcs], w, b)
        hit = cache[key]
    return hit[1], hit[2]


def compact_rope(rotary_emb):
    """(cos, sin) as the reference passes them ([1,1,L,Dh] fp32, values repeated pairwise) ->
    contiguous [L, Dh/2] tables with the slots transformer_wan.py:82-83 reads (cos 0::2, sin 1::2)."""
    cos, sin = rotary_emb
    return (cos.reshape(-1, cos.shape[-1])[:, 0::2].float().contiguous(),
            sin.reshape(-1, sin.shape[-1])[:, 1::2].float().contiguous())


class MI355WanAttnProcessor:
    """HIP implementation of WanAttnProcessor2_0.__call__ (transformer_wan.py:43-119)."""

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, rotary_emb=None):
        if attention_mask is not None:
            raise NotImplementedError("attention_mask is never passed on the FrameINO path")
        if getattr(attn, "add_k_proj", None) is not None:
            raise NotImplementedError("image-KV branch (Wan2.1 I2V) is outside the TI2V-5B path")
        b, lq, d = hidden_states.shape
        heads = attn.heads
        dh = d // heads
        eps = attn.norm_q.eps if attn.norm_q is not None else 0.0
        x = hidden_states.reshape(b * lq, d)
        rope = None
        if rotary_emb is not None:
            rope = rotary_emb if getattr(rotary_emb, "compact", False) else compact_rope(rotary_emb)
        if encoder_hidden_states is None:
            w, bias = _fused_weight(attn, ("to_q", "to_k", "to_v"), "qkv")
            qkv = ops.gemm(x, w, bias)                                   # [B*L, 3D]
            q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
            lk = lq
        else:
            lk = encoder_hidden_states.shape[1]
            ctx = encoder_hidden_states.reshape(b * lk, -1)
            q = ops.gemm(x, attn.to_q.weight, attn.to_q.bias)
            w, bias = _fused_weight(attn, ("to_k", "to_v"), "kv")
            kv = ops.gemm(ctx, w, bias)
            k, v = kv[:, :d], kv[:, d:]
        for i in range(b):
            qs, ks = q[i * lq:(i + 1) * lq], k[i * lk:(i + 1) * lk]
            cs = rope if rope is not None else (None, None)
            if attn.norm_q is not None or rope is not None:
                ops.rmsnorm_rope_(qs, attn.norm_q.weight if attn.norm_q is not None else None, eps, cs[0], cs[1], dh)
                ops.rmsnorm_rope_(ks, attn.norm_k.weight if attn.norm_k is not None else None, eps, cs[0], cs[1], dh)
        o = ops.attention(q.unflatten(0, (b, lq)), k.unflatten(0, (b, lk)), v.unflatten(0, (b, lk)), heads)
        out = ops.gemm(o.reshape(b * lq, d), attn.to_out[0].weight, attn.to_out[0].bias)
        return out.view(b, lq, -1)


class MI355CogVideoXAttnProcessor:
    """HIP implementation of CogVideoXAttnProcessor2_0.__call__ (attention_processor.py:2815-2877):
    joint [text|video] attention, per-head LayerNorm on q/k, RoPE on the video tokens only."""

    fused = False

    def __call__(self, attn, hidden_states, encoder_hidden_states, attention_mask=None, image_rotary_emb=None):
        if attention_mask is not None:
            raise NotImplementedError("attention_mask is never passed on the FrameINO path")
        lt = encoder_hidden_states.size(1)
        hs = torch.cat([encoder_hidden_states, hidden_states], dim=1)
        b, l, d = hs.shape
        heads = attn.heads
        dh = d // heads
        if self.fused and getattr(attn, "to_qkv", None) is not None:
            w, bias = attn.to_qkv.weight, attn.to_qkv.bias                # attention_processor.py:2908
        else:
            w, bias = _fused_weight(attn, ("to_q", "to_k", "to_v"), "qkv")
        qkv = ops.gemm(hs.reshape(b * l, d), w, bias).view(b, l, 3 * d)
        q, k, v = qkv[:, :, :d], qkv[:, :, d:2 * d], qkv[:, :, 2 * d:]
        cos = sin = None
        if image_rotary_emb is not None:
            cos, sin = (t.float().contiguous() for t in image_rotary_emb)
        nq, nk = attn.norm_q, attn.norm_k
        ops.headnorm_rope_(q, heads, dh, None if nq is None else nq.weight, None if nq is None else nq.bias,
                           0.0 if nq is None else nq.eps, cos, sin, rope_row0=lt)
        if not attn.is_cross_attention:
            ops.headnorm_rope_(k, heads, dh, None if nk is None else nk.weight, None if nk is None else nk.bias,
                               0.0 if nk is None else nk.eps, cos, sin, rope_row0=lt)
        o = ops.attention(q, k, v, heads)
        out = ops.gemm(o.reshape(b * l, d), attn.to_out[0].weight, attn.to_out[0].bias).view(b, l, -1)
        return out[:, lt:], out[:, :lt]


class MI355FusedCogVideoXAttnProcessor(MI355CogVideoXAttnProcessor):
    """FusedCogVideoXAttnProcessor2_0 (attention_processor.py:2880-2948): uses attn.to_qkv after
    `fuse_projections()`; results equal the unfused processor (same GEMM, same K order)."""

    fused = True


AttentionProcessor = (MI355WanAttnProcessor, MI355CogVideoXAttnProcessor, MI355FusedCogVideoXAttnProcessor)
