#!/usr/bin/env python3
"""fino_attn_probs at the bench shape (24 heads x 128, Lq = 12320; prompts of 64 / 8 tokens = 65 / 9 keys with the padding run
folded): microseconds per launch, and the bytes it has to move (q read once, P written once)."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frameino_amd import ops
heads, L = 24, int(sys.argv[1]) if len(sys.argv) > 1 else 12320
d = heads * 128
g = torch.Generator(device="cuda").manual_seed(0)
q = torch.randn(1, L, d, device="cuda", generator=g).bfloat16()
k = torch.randn(1, 128, d, device="cuda", generator=g).bfloat16()
for lk in (65, 9, 128):
    kp = -(-lk // 8) * 8
    out = torch.empty(1, L, heads * kp, device="cuda", dtype=torch.bfloat16)
    f = lambda: ops.attention_probs(q, k, heads, [lk], [512.0 - lk + 1], kp, out=out)
    f(); f()
    t = []
    for _ in range(7):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): f()
        e.record(); torch.cuda.synchronize(); t.append(s.elapsed_time(e) / 10 * 1e3)
    us = statistics.median(t)
    mb = (q.numel() + out.numel()) * 2 / 1e6
    print(f"{lk:4d} keys (kp {kp:3d}): {us:7.1f} us   {mb:6.1f} MB -> {mb / us:5.2f} TB/s")
