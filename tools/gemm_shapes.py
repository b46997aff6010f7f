#!/usr/bin/env python3
"""The Wan2.2-5B block GEMMs (B = 2 x 12320 tokens) with the epilogues the forward uses: median us and TFLOP/s.
A/B of two library builds on one box: run it once per FINO_LIB_PATH (tools/README.md)."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frameino_amd import _lib, ops
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 24640
TILE_M = int(sys.argv[2]) if len(sys.argv) > 2 else 0          # the per-call tile height (0 = planned, 8 = 256-row tiles only, ...)
SHAPES = [("qkv", 9216, 3072, 0), ("out-proj", 3072, 3072, 3), ("q2", 3072, 3072, 0), ("out2", 3072, 3072, 2),
          ("ffn-up", 14336, 3072, 1), ("ffn-down", 3072, 14336, 3)]
tot = 0.0
print(os.path.basename(_lib.LIB_PATH))
for nm, n, k, epi in SHAPES:
    A = torch.randn(M, k, device=dev, generator=g).bfloat16()
    W = (torch.randn(n, k, device=dev, generator=g) * 0.02).bfloat16()
    b = torch.randn(n, device=dev, generator=g).bfloat16()
    res = torch.randn(M, n, device=dev, generator=g).bfloat16() if epi >= 2 else None
    gate = torch.randn(2, n, device=dev, generator=g) if epi == 3 else None
    # FrameINO's selector: the first-frame tokens (880 of every 12320) see timestep row 0, the rest row 1 (FINO_SEL_ALT=1: the
    # worst case instead, a selector that changes on every row)
    sel = None
    if epi == 3:
        ar = torch.arange(M, device=dev)
        sel = ((ar % 2) if os.environ.get("FINO_SEL_ALT") else ((ar % 12320) >= 880)).to(torch.int32)
    out = torch.empty(M, n, device=dev, dtype=torch.bfloat16)
    f = lambda: ops.gemm(A, W, b, epi, res, gate, sel, out=out, tile_m=TILE_M)
    f(); f()
    r = []
    for _ in range(7):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): f()
        e.record(); torch.cuda.synchronize(); r.append(s.elapsed_time(e) / 5 * 1e3)
    t = statistics.median(r)
    tot += t
    print(f"  {nm:9s} {M}x{n}x{k} epi {epi}: {t:7.1f} us  {2.0 * M * n * k / t / 1e6:5.0f} TF")
print(f"  sum {tot:.1f} us")
