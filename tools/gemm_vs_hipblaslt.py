#!/usr/bin/env python3
"""Plain-epilogue (bias only) comparison of the ping-pong GEMM with hipBLASLt (torch F.linear) on the four block
shapes at M = 24640, interleaved rounds in one process, median.  Separates the main loops from the fused epilogues."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frameino_amd import ops
M, D, F = 24640, 3072, 14336
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)


def timed(fn, iters=6):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for n, k, nm in [(3 * D, D, "qkv"), (D, D, "out"), (F, D, "ffn-up"), (D, F, "ffn-down")]:
    A = torch.randn(M, k, device=dev, generator=g).bfloat16()
    W = (torch.randn(n, k, device=dev, generator=g) * 0.02).bfloat16()
    b = torch.randn(n, device=dev, generator=g).bfloat16()
    out = torch.empty(M, n, device=dev, dtype=torch.bfloat16)
    ours = lambda: ops.gemm(A, W, b, 0, out=out)
    gelu = lambda: ops.gemm(A, W, b, 1, out=out)
    hbl = lambda: torch.nn.functional.linear(A, W, b)
    for f in (ours, gelu, hbl):
        timed(f, 2)
    r = {"ours bias": [], "ours bias+gelu": [], "hipBLASLt bias": []}
    for _ in range(7):
        r["ours bias"].append(timed(ours)); r["ours bias+gelu"].append(timed(gelu)); r["hipBLASLt bias"].append(timed(hbl))
    fl = 2.0 * M * n * k
    print(f"{nm:9s} {M}x{n}x{k}: " + "  ".join(f"{kk}: {statistics.median(v):7.1f} us ({fl / statistics.median(v) / 1e6:5.0f} TF)" for kk, v in r.items()))
