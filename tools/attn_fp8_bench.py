#!/usr/bin/env python3
"""head_dim-64 self-attention at the CogVideoX-5B shape (B = 2, 48 heads, L = 19126): the library's bf16 kernels (8-wave with
the plain scale, 4-wave with the folded scale) and the fp8-operand kernel (fino_attn_fwd_fp8: K / V quantisation pre-pass +
main kernel), interleaved, median of 5 x 3 launches; accuracy of each against fp32 SDPA on sampled rows."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frameino_amd import ops
dev = "cuda"
b, heads, L = 2, 48, int(sys.argv[1]) if len(sys.argv) > 1 else 19126
d = heads * 64
g = torch.Generator(device=dev).manual_seed(0)
q = torch.randn(b, L, d, device=dev, generator=g).bfloat16()
kv = torch.randn(b, L, 2 * d, device=dev, generator=g).bfloat16()
k, v = kv[:, :, :d], kv[:, :, d:]
c = 64 ** -0.5 * ops.LOG2E
qs = (q.float() * c).bfloat16()
o = torch.empty_like(q)
runs = {"bf16 8-wave": lambda: ops.attention(q, k, v, heads, out=o),
        "bf16 4-wave folded": lambda: ops.attention(qs, k, v, heads, out=o, scale=ops.SCALE_FOLDED),
        "fp8 operands, P = exp2": lambda: ops.attention_fp8(q, k, v, heads, out=o, p_mode="exp2"),
        "fp8 operands, P = ramp": lambda: ops.attention_fp8(q, k, v, heads, out=o, p_mode="ramp")}
if os.environ.get("FINO_FP8_KERNEL"):         # 1: the 8-wave ping-pong kernel instead of the free-running 4-wave one
    from frameino_amd import _lib
    _lib.lib().fino_tune_set(5, int(os.environ["FINO_FP8_KERNEL"]))
if os.environ.get("FINO_FP8_ONLY"):           # timing-experiment builds (tools/attn_fp8_variants.sh): that kernel alone
    runs = {n: f for n, f in runs.items() if n.startswith("fp8")}
rows = torch.tensor(sorted(set(torch.randint(0, L, (24,)).tolist()) | {0, L - 1}), device=dev)
qh = q[0, rows].float().view(len(rows), heads, 64).transpose(0, 1)
kh, vh = k[0].float().view(L, heads, 64).transpose(0, 1), v[0].float().view(L, heads, 64).transpose(0, 1)
ref = (torch.softmax(qh @ kh.transpose(1, 2) * 0.125, -1) @ vh).transpose(0, 1).reshape(len(rows), d)
t = {n: [] for n in runs}
for n, f in runs.items():
    f(); f()
    e = ((o[0, rows].float() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    print(f"{n:24s} rel-RMS vs fp32 SDPA (sampled rows): {e:.4f}")
for _ in range(5):
    for n, f in runs.items():
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(3): f()
        e.record(); torch.cuda.synchronize(); t[n].append(s.elapsed_time(e) / 3 * 1e3)
fl = 4.0 * b * L * L * d
for n in runs:
    us = statistics.median(t[n])
    print(f"{n:24s} {us:9.1f} us  {fl / us / 1e6:7.0f} TFLOP/s-equivalent")
