#!/usr/bin/env python3
"""Micro-benchmarks of the hot kernels at the Wan2.2-5B 704x1280x49f shapes (L=12320) -- GPU box only.
Prints achieved TFLOP/s (or GB/s) per kernel against the gfx950 peaks (2.5 PF bf16 dense, 8 TB/s HBM)."""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frameino_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--L", type=int, default=12320)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--only", default="")
a = ap.parse_args()
dev = "cuda"
L, D, H, F = a.L, 3072, 24, 14336
torch.manual_seed(0)


def timeit(fn, iters=a.iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def report(name, t, flops=None, bytes_=None):
    msg = f"{name:34s} {t*1e6:10.1f} us"
    if flops:
        msg += f"  {flops/t/1e12:8.1f} TFLOP/s ({flops/t/2.5e15*100:5.1f}% of 2.5PF)"
    if bytes_:
        msg += f"  {bytes_/t/1e9:8.0f} GB/s ({bytes_/t/8e12*100:5.1f}% of 8TB/s)"
    print(msg, flush=True)


def want(n):
    return not a.only or any(k in n for k in a.only.split(","))


x = torch.randn(L, D, device=dev).bfloat16()
if want("attn"):
    qkv = torch.randn(1, L, 3 * D, device=dev).bfloat16()
    o = torch.empty(1, L, D, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: ops.attention(qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:], H, out=o))
    report("attn self L=%d" % L, t, flops=4.0 * L * L * D)
    kv = torch.randn(1, 512, 2 * D, device=dev).bfloat16()
    t = timeit(lambda: ops.attention(qkv[:, :, :D], kv[:, :, :D], kv[:, :, D:], H, out=o))
    report("attn cross Lk=512", t, flops=4.0 * L * 512 * D)
if want("gemm"):
    for (n, k, epi, nm) in [(3 * D, D, 0, "qkv"), (D, D, 3, "out+gate"), (F, D, 1, "ffn-up+gelu"), (D, F, 3, "ffn-down+gate")]:
        w = (torch.randn(n, k, device=dev) * 0.02).bfloat16()
        b = torch.randn(n, device=dev).bfloat16()
        ain = torch.randn(L, k, device=dev).bfloat16()
        out = torch.empty(L, n, device=dev, dtype=torch.bfloat16)
        res = torch.randn(L, n, device=dev).bfloat16() if epi == 3 else None
        gate = torch.randn(2, n, device=dev) if epi == 3 else None
        sel = (torch.arange(L, device=dev) < 880).to(torch.int32) if epi == 3 else None
        t = timeit(lambda: ops.gemm(ain, w, b, epi, res, gate, sel, out=out))
        report(f"gemm {nm} {L}x{n}x{k}", t, flops=2.0 * L * n * k)
        t = timeit(lambda: torch.nn.functional.linear(ain, w, b))
        report(f"  (hipBLASLt F.linear same shape)", t, flops=2.0 * L * n * k)
if want("elem"):
    table = torch.randn(2, 6, D, device=dev)
    sel = (torch.arange(L, device=dev) < 880).to(torch.int32)
    y = torch.empty_like(x)
    t = timeit(lambda: ops.adaln_modulate(x, table[:, 0], table[:, 1], sel, out=y))
    report("adaln_modulate", t, bytes_=2 * L * D * 2)
    t = timeit(lambda: ops.gated_residual(x, y, table[:, 2], sel, out=y))
    report("gated_residual", t, bytes_=3 * L * D * 2)
    qkv = torch.randn(L, 3 * D, device=dev).bfloat16()
    w = torch.ones(D, device=dev).bfloat16()
    cos, sin = torch.rand(L, 64, device=dev), torch.rand(L, 64, device=dev)
    t = timeit(lambda: ops.rmsnorm_rope_(qkv[:, :D], w, 1e-6, cos, sin, 128))
    report("rmsnorm_rope (q)", t, bytes_=2 * L * D * 2)
