#!/usr/bin/env python3
"""Per-block fixed cost of the self-attention kernel: time(Lk) at fixed Lq / batch / heads is a + b.Lk; the intercept a
(prologue, epilogue, ramp, tail) against the slope (steady loop).  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frameino_amd import ops
b, H, D = 2, 24, 3072
Lq = 12320
g = torch.Generator(device="cuda").manual_seed(0)
q = torch.randn(b, Lq, D, device="cuda", generator=g).bfloat16()
o = torch.empty_like(q)
res = []
for Lk in (1536, 3072, 6144, 9216, 12288, 12320):
    kv = torch.randn(b, Lk, 2 * D, device="cuda", generator=g).bfloat16()
    f = lambda: ops.attention(q, kv[:, :, :D], kv[:, :, D:], H, out=o)
    for _ in range(3): f()
    ts = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): f()
        e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) / 5 * 1e3)
    ts.sort(); t = ts[2]
    res.append((Lk, t))
    print(f"Lk={Lk:6d}: {t:8.1f} us  {4.0 * b * Lq * Lk * D / t / 1e6:6.0f} TFLOP/s", flush=True)
(x0, y0), (x1, y1) = res[1], res[4]
slope = (y1 - y0) / (x1 - x0)
print(f"slope {slope * 64:.2f} us per 64-key tile step over all blocks; intercept {y0 - slope * x0:.1f} us "
      f"({(y0 - slope * x0) / res[-1][1] * 100:.1f} % of the Lk = 12320 launch); steady-state rate "
      f"{4.0 * b * Lq * D / slope / 1e6:.0f} TFLOP/s")
