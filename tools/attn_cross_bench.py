#!/usr/bin/env python3
"""Text cross-attention at the bench shape (B = 2, Lq = 12320, Lk = 512, 24 heads x 128): median us per launch and TFLOP/s.
(the one-barrier loop FINO_ATTN_PP=0 used to select left the library in round 5)."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frameino_amd import ops
b, lq, lk, heads, dh = 2, 12320, int(sys.argv[1]) if len(sys.argv) > 1 else 512, 24, 128
d = heads * dh
g = torch.Generator(device="cuda").manual_seed(0)
q = torch.randn(b, lq, d, device="cuda", generator=g).bfloat16()
kv = torch.randn(b, lk, 2 * d, device="cuda", generator=g).bfloat16()
o = torch.empty_like(q)
f = lambda: ops.attention(q, kv[:, :, :d], kv[:, :, d:], heads, out=o)
f(); f()
ts = []
for _ in range(7):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): f()
    e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) / 10 * 1e3)
us = statistics.median(ts)
print(f"cross-attention Lq {lq} x Lk {lk}, B {b}: {us:7.1f} us  {4.0 * b * lq * lk * d / us / 1e6:6.0f} TFLOP/s")
