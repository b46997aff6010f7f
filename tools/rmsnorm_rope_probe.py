#!/usr/bin/env python3
"""Why does the in-place q | k RMSNorm + RoPE run at 3.7 TB/s where the out-of-place adaLN does 4.5 on the same bytes?
Times, at the bench shape (M = 24640 rows of a fused [M, 9216] projection): the in-place one-launch kernel, two in-place launches,
and the same arithmetic written OUT of place through the scatter tables into a separate [M, 2 D] buffer."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frameino_amd import ops
from frameino_amd.transformer_wan import wan_rope_tables
dev = torch.device("cuda")
M, D, H, dh = 24640, 3072, 24, 128
g = torch.Generator(device=dev).manual_seed(0)
qkv = torch.randn(M, 3 * D, device=dev, generator=g).bfloat16()
w = torch.ones(D, device=dev, dtype=torch.bfloat16)
c1, s1 = wan_rope_tables(dh, 1024, 14, 22, 40)
cos, sin = c1.repeat(2, 1).contiguous().to(dev), s1.repeat(2, 1).contiguous().to(dev)
qk2 = torch.empty(M, 2 * D, device=dev, dtype=torch.bfloat16)
off_q = torch.tensor([h * dh for h in range(H)], dtype=torch.int64, device=dev)
off_k = torch.tensor([D + h * dh for h in range(H)], dtype=torch.int64, device=dev)
ld = torch.full((H,), 2 * D, dtype=torch.int64, device=dev)


def t(f, n=20):
    for _ in range(3): f()
    ts = []
    for _ in range(7):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): f()
        e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) / n * 1e3)
    return statistics.median(ts)


nbytes = 4.0 * M * D * 2
a = t(lambda: ops.qkv_rmsnorm_rope_(qkv, D, w, 1e-6, w, 1e-6, cos, sin, dh))
print(f"in place, one launch (product):            {a:7.1f} us  {nbytes / a / 1e6:5.2f} TB/s")
b = t(lambda: (ops.rmsnorm_rope_(qkv[:, :D], w, 1e-6, cos, sin, dh), ops.rmsnorm_rope_(qkv[:, D:2 * D], w, 1e-6, cos, sin, dh)))
print(f"in place, two launches:                    {b:7.1f} us  {nbytes / b / 1e6:5.2f} TB/s")
c = t(lambda: (ops.rmsnorm_rope_scatter(qkv[:, :D], w, 1e-6, cos, sin, dh, qk2, off_q, ld),
               ops.rmsnorm_rope_scatter(qkv[:, D:2 * D], w, 1e-6, cos, sin, dh, qk2, off_k, ld)))
print(f"out of place into [M, 2D], two launches:   {c:7.1f} us  {nbytes / c / 1e6:5.2f} TB/s")
x = torch.randn(M, D, device=dev, generator=g).bfloat16()
y = torch.empty_like(x)
mod = torch.randn(2, 2, D, device=dev, generator=g)
sel = (torch.arange(M, device=dev) % 2).to(torch.int32)
d = t(lambda: ops.adaln_modulate(x, mod[:, 0], mod[:, 1], sel, 1e-6, out=y))
print(f"adaLN out of place (control, half the bytes): {d:7.1f} us  {2.0 * M * D * 2 / d / 1e6:5.2f} TB/s")
e = t(lambda: ops.adaln_modulate(x, mod[:, 0], mod[:, 1], sel, 1e-6, out=x))
print(f"adaLN IN place (control):                  {e:7.1f} us  {2.0 * M * D * 2 / e / 1e6:5.2f} TB/s")
