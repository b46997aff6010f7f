#!/usr/bin/env python3
"""GEMM timing at the Wan block shapes (GPU box): `FINO_GEMM_PP=0|1 python tools/gemm_ab.py [L]`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frameino_amd import ops
L = int(sys.argv[1]) if len(sys.argv) > 1 else 12320
D, F = 3072, 14336
tot = 0.0
for (n, k, epi, nm) in [(3 * D, D, 0, "qkv"), (D, D, 3, "out+gate"), (F, D, 1, "ffn-up+gelu"), (D, F, 3, "ffn-down+gate")]:
    a = torch.randn(L, k, device="cuda").bfloat16(); w = (torch.randn(n, k, device="cuda") * 0.02).bfloat16()
    b = torch.randn(n, device="cuda").bfloat16(); out = torch.empty(L, n, device="cuda", dtype=torch.bfloat16)
    res = torch.randn(L, n, device="cuda").bfloat16() if epi == 3 else None
    gate = torch.randn(2, n, device="cuda") if epi == 3 else None
    sel = (torch.arange(L, device="cuda") < 880).to(torch.int32) if epi == 3 else None
    f = lambda: ops.gemm(a, w, b, epi, res, gate, sel, out=out)
    for _ in range(3): f()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(20): f()
    e.record(); torch.cuda.synchronize(); t = s.elapsed_time(e) / 20 * 1e-3; tot += t
    print("pp=%s %-14s %dx%dx%d: %7.1f us %5.0f TF" % (os.environ.get("FINO_GEMM_PP", "1"), nm, L, n, k, t * 1e6, 2 * L * n * k / t / 1e12))
print("pp=%s sum %.1f us" % (os.environ.get("FINO_GEMM_PP", "1"), tot * 1e6))
