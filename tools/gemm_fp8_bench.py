#!/usr/bin/env python3
"""MXFP8 GEMM vs the bf16 ping-pong GEMM at the Wan / CogVideoX block shapes (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frameino_amd import ops
L = int(sys.argv[1]) if len(sys.argv) > 1 else 24640
D, F = 3072, 14336 if L != 38252 else 12288


def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e-3


tot8 = tot16 = totq = 0.0
for (n, k, epi, nm) in [(3 * D, D, 0, "qkv"), (D, D, 3, "out+gate"), (F, D, 1, "ffn-up+gelu"), (D, F, 3, "ffn-down+gate")]:
    a = torch.randn(L, k, device="cuda").bfloat16(); w = (torch.randn(n, k, device="cuda") * 0.02).bfloat16()
    b = torch.randn(n, device="cuda").bfloat16(); out = torch.empty(L, n, device="cuda", dtype=torch.bfloat16)
    res = torch.randn(L, n, device="cuda").bfloat16() if epi == 3 else None
    gate = torch.randn(2, n, device="cuda") if epi == 3 else None
    sel = (torch.arange(L, device="cuda") < 880).to(torch.int32) if epi == 3 else None
    aq, sa = ops.quantize_mxfp8(a); wq, sw = ops.quantize_mxfp8(w)
    t16 = timeit(lambda: ops.gemm(a, w, b, epi, res, gate, sel, out=out))
    t8 = timeit(lambda: ops.gemm_mxfp8(aq, sa, wq, sw, b, epi, res, gate, sel, out=out))
    tq = timeit(lambda: ops.quantize_mxfp8(a, out=(aq, sa)))
    fl = 2.0 * L * n * k
    tot8 += t8; tot16 += t16; totq += tq
    print(f"{nm:14s} {L}x{n}x{k}: bf16 {t16*1e6:7.1f} us {fl/t16/1e12:5.0f} TF | mxfp8 {t8*1e6:7.1f} us {fl/t8/1e12:5.0f} TF "
          f"({t16/t8:.2f}x) | quantise A {tq*1e6:6.1f} us ({(2+1.03)*L*k/tq/1e9:5.0f} GB/s)")
print(f"sum: bf16 {tot16*1e6:.0f} us, mxfp8 {tot8*1e6:.0f} us + quantise {totq*1e6:.0f} us")
