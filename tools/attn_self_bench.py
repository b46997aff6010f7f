#!/usr/bin/env python3
"""Self-attention at the bench shape (B = 2, 24 heads x 128, L = 12320; FINO_LIB_PATH selects another build): median us."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frameino_amd import ops
b, L, heads, dh = 2, int(sys.argv[1]) if len(sys.argv) > 1 else 12320, 24, 128
d = heads * dh
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(b, L, 3 * d, device="cuda", generator=g).bfloat16()
o = torch.empty(b, L, d, device="cuda", dtype=torch.bfloat16)
f = lambda: ops.attention(qkv[:, :, :d], qkv[:, :, d:2 * d], qkv[:, :, 2 * d:], heads, out=o)
f(); f()
ts = []
for _ in range(7):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): f()
    e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) / 5 * 1e3)
us = statistics.median(ts)
print(f"self-attention B {b} L {L}: {us:8.1f} us  {4.0 * b * L * L * d / us / 1e6:6.0f} TFLOP/s  ({os.environ.get('FINO_LIB_PATH', 'product')})")
