import inspect

import torch
from torch import nn

from .normalization import RMSNorm


class Attention(nn.Module):
    """Stand-in for diffusers.models.attention_processor.Attention: only the container,
    the projection/norm sub-modules and the kwarg-filtering dispatch to `processor`."""

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, kv_heads=None, dim_head=64, dropout=0.0,
                 bias=False, qk_norm=None, added_kv_proj_dim=None, added_proj_bias=True, out_bias=True,
                 eps=1e-5, processor=None, out_dim=None, elementwise_affine=True, **unused):
        super().__init__()
        self.inner_dim = out_dim if out_dim is not None else dim_head * heads
        self.inner_kv_dim = self.inner_dim if kv_heads is None else dim_head * kv_heads
        self.query_dim = query_dim
        self.is_cross_attention = cross_attention_dim is not None
        self.cross_attention_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.out_dim = out_dim if out_dim is not None else query_dim
        self.heads = out_dim // dim_head if out_dim is not None else heads
        self.scale = dim_head ** -0.5
        self.added_kv_proj_dim = added_kv_proj_dim
        self.fused_projections = False
        self.use_bias = bias

        if qk_norm is None:
            self.norm_q = None
            self.norm_k = None
        elif qk_norm == "layer_norm":
            self.norm_q = nn.LayerNorm(dim_head, eps=eps, elementwise_affine=elementwise_affine)
            self.norm_k = nn.LayerNorm(dim_head, eps=eps, elementwise_affine=elementwise_affine)
        elif qk_norm == "rms_norm_across_heads":
            self.norm_q = RMSNorm(dim_head * heads, eps=eps)
            self.norm_k = RMSNorm(dim_head * (kv_heads if kv_heads is not None else heads), eps=eps)
        else:
            raise ValueError(qk_norm)

        self.to_q = nn.Linear(query_dim, self.inner_dim, bias=bias)
        self.to_k = nn.Linear(self.cross_attention_dim, self.inner_kv_dim, bias=bias)
        self.to_v = nn.Linear(self.cross_attention_dim, self.inner_kv_dim, bias=bias)
        self.add_k_proj = None
        self.add_v_proj = None
        self.norm_added_k = None
        if added_kv_proj_dim is not None:
            self.add_k_proj = nn.Linear(added_kv_proj_dim, self.inner_kv_dim, bias=added_proj_bias)
            self.add_v_proj = nn.Linear(added_kv_proj_dim, self.inner_kv_dim, bias=added_proj_bias)
            self.norm_added_k = RMSNorm(dim_head * heads, eps=eps)
        self.to_out = nn.ModuleList([nn.Linear(self.inner_dim, self.out_dim, bias=out_bias), nn.Dropout(dropout)])
        self.processor = processor

    def set_processor(self, processor):
        self.processor = processor

    def get_processor(self):
        return self.processor

    @torch.no_grad()
    def fuse_projections(self, fuse=True):
        device = self.to_q.weight.data.device
        dtype = self.to_q.weight.data.dtype
        w = torch.cat([self.to_q.weight.data, self.to_k.weight.data, self.to_v.weight.data])
        self.to_qkv = nn.Linear(w.shape[1], w.shape[0], bias=self.use_bias, device=device, dtype=dtype)
        self.to_qkv.weight.copy_(w)
        if self.use_bias:
            self.to_qkv.bias.copy_(torch.cat([self.to_q.bias.data, self.to_k.bias.data, self.to_v.bias.data]))
        self.fused_projections = fuse

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **cross_attention_kwargs):
        attn_parameters = set(inspect.signature(self.processor.__call__).parameters.keys())
        cross_attention_kwargs = {k: w for k, w in cross_attention_kwargs.items() if k in attn_parameters}
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask, **cross_attention_kwargs)
