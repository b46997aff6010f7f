#!/usr/bin/env python3
"""Workload for rocprofv3: the head_dim-64 attention at the CogVideoX-5B shape through the 8-wave kernel (plain scale) and
the 4-wave kernel (folded scale), three launches each (tools/README.md has the rocprofv3 command lines)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frameino_amd import ops
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
b, heads, L = 2, 48, 19126
d = heads * 64
qkv = torch.randn(b, L, 3 * d, device=dev, generator=g).bfloat16()
q, k, v = qkv[:, :, :d], qkv[:, :, d:2 * d], qkv[:, :, 2 * d:]
qs = (q.float() * (64 ** -0.5 * ops.LOG2E)).bfloat16()
out = torch.empty(b, L, d, device=dev, dtype=torch.bfloat16)
for _ in range(3):
    ops.attention(q, k, v, heads, out=out)                                  # 8-wave kernel
    ops.attention(qs, k, v, heads, out=out, scale=ops.SCALE_FOLDED)         # 4-wave kernel (default for the folded scale)
torch.cuda.synchronize()
