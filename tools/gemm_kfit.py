#!/usr/bin/env python3
"""Fixed (prologue + epilogue + launch) vs per-K cost of the GEMM: time at several K, same M x N (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frameino_amd import ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 24640
for n, epi in ((3072, 3), (3072, 2), (3072, 0), (9216, 0), (14336, 1)):
    pts = []
    for k in (768, 1536, 3072, 6144):
        a = torch.randn(M, k, device="cuda").bfloat16(); w = (torch.randn(n, k, device="cuda") * 0.02).bfloat16()
        b = torch.randn(n, device="cuda").bfloat16(); out = torch.empty(M, n, device="cuda", dtype=torch.bfloat16)
        res = torch.randn(M, n, device="cuda").bfloat16() if epi >= 2 else None
        gate = torch.randn(2, n, device="cuda") if epi == 3 else None
        sel = (torch.arange(M, device="cuda") < 880).to(torch.int32) if epi == 3 else None
        f = lambda: ops.gemm(a, w, b, epi, res, gate, sel, out=out)
        for _ in range(3): f()
        torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True); s.record()
        for _ in range(20): f()
        e.record(); torch.cuda.synchronize(); pts.append((k, s.elapsed_time(e) / 20 * 1e3))
    (k0, t0), (k1, t1) = pts[0], pts[-1]
    slope = (t1 - t0) / (k1 - k0)
    fixed = t0 - slope * k0
    print(f"M={M} N={n} epi={epi}: " + " ".join(f"K={k}:{t:.0f}us" for k, t in pts) +
          f" | fixed {fixed:.0f} us, slope {slope*1000:.1f} us/1000K -> marginal {2*M*n/(slope*1e-6)/1e12:.0f} TF, fixed share at K=3072 {fixed/(fixed+slope*3072)*100:.0f}%")
