#!/usr/bin/env python3
"""head_dim-64 attention at the CogVideoX-5B shape (B = 2, 48 heads, L = 19126): us and TFLOP/s of the loaded library
(FINO_LIB_PATH selects another build for an A/B on one box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frameino_amd import ops
b, L, H, D = 2, 19126, 48, 3072
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(b, L, 3 * D, device="cuda", generator=g).bfloat16()
o = torch.empty(b, L, D, device="cuda", dtype=torch.bfloat16)
f = lambda: ops.attention(qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:], H, out=o)
for _ in range(3): f()
ts = []
for _ in range(5):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): f()
    e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) / 5 * 1e3)
ts.sort(); t = ts[len(ts) // 2]
print(f"{os.environ.get('FINO_LIB_PATH', 'default lib')}: d64 attention B={b} L={L}: {t:.1f} us, {4.0 * b * L * L * D / t / 1e6:.0f} TFLOP/s")
