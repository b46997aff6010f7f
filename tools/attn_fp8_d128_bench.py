#!/usr/bin/env python3
"""head_dim-128 self-attention at the Wan2.2-5B bench shape (B = 2, 24 heads, L = 12320): bf16 8-wave kernel vs the fp8-operand
kernel (fino_attn_fwd_fp8: K/V quantisation pre-pass + attn_fp8_d128_kernel), accuracy on sampled rows then interleaved timing."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frameino_amd import ops
b, heads, L = 2, 24, int(sys.argv[1]) if len(sys.argv) > 1 else 12320
d = heads * 128
g = torch.Generator(device="cuda").manual_seed(0)
q = torch.randn(b, L, d, device="cuda", generator=g).bfloat16()
kv = torch.randn(b, L, 2 * d, device="cuda", generator=g).bfloat16()
k, v = kv[:, :, :d], kv[:, :, d:]
if os.environ.get("FINO_BENCH_ZERO"):        # all-zero operands: nothing toggles -- how much of the time is the power cap
    q.zero_(); kv.zero_()
o = torch.empty_like(q)
runs = {"bf16 8-wave": lambda: ops.attention(q, k, v, heads, out=o), "fp8, P = exp2": lambda: ops.attention_fp8(q, k, v, heads, out=o, p_mode="exp2"),
        "fp8, P = ramp": lambda: ops.attention_fp8(q, k, v, heads, out=o, p_mode="ramp")}
rows = torch.tensor(sorted(set(torch.randint(0, L, (24,)).tolist()) | {0, L - 1}), device="cuda")
qh = q[0, rows].float().view(len(rows), heads, 128).transpose(0, 1)
kh, vh = k[0].float().view(L, heads, 128).transpose(0, 1), v[0].float().view(L, heads, 128).transpose(0, 1)
ref = (torch.softmax(qh @ kh.transpose(1, 2) * 128 ** -0.5, -1) @ vh).transpose(0, 1).reshape(len(rows), d)
t = {n: [] for n in runs}
for n, f in runs.items():
    f(); f()
    e = ((o[0, rows].float() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    print(f"{n:14s} rel-RMS vs fp32 SDPA (sampled rows): {e:.4f}")
for _ in range(5):
    for n, f in runs.items():
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(3): f()
        e.record(); torch.cuda.synchronize(); t[n].append(s.elapsed_time(e) / 3 * 1e3)
fl = 4.0 * b * L * L * d
for n in runs:
    us = statistics.median(t[n])
    print(f"{n:14s} {us:9.1f} us  {fl / us / 1e6:7.0f} TFLOP/s-equivalent")
