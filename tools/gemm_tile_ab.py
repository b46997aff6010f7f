#!/usr/bin/env python3
"""Tile heights of fino_gemm at every block GEMM of a Wan2.2-5B layer: 256-row tiles only (FINO_TUNE_GEMM_TILE_M = 8, the
round-2 behaviour), every forced single height, and the planner's choice (0), interleaved on one box, median of 5 x 5
launches.  usage: gemm_tile_ab.py [--all-heights] [rows ...]"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frameino_amd import _lib, ops
lib = _lib.lib()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
D, FF = 3072, 14336
args = [a for a in sys.argv[1:] if not a.startswith("--")]
all_h = "--all-heights" in sys.argv
rows = [int(x) for x in args] or [24640, 12320, 6160, 3080, 1540]
shapes = [("qkv", 3 * D, D, 0), ("kv", 2 * D, D, 0), ("q/q2", D, D, 0), ("out", D, D, 3), ("out2", D, D, 2),
          ("ffn_up", FF, D, 1), ("ffn_down", D, FF, 3)]
for M in rows:
    tot = {}
    for nm, n, k, epi in shapes:
        A = torch.randn(M, k, device=dev, generator=g).bfloat16()
        W = (torch.randn(n, k, device=dev, generator=g) * 0.02).bfloat16()
        b = torch.randn(n, device=dev, generator=g).bfloat16()
        res = torch.randn(M, n, device=dev, generator=g).bfloat16() if epi >= 2 else None
        gate = torch.randn(2, n, device=dev, generator=g) if epi >= 3 else None
        sel = (torch.arange(M, device=dev) % 2).to(torch.int32) if epi >= 3 else None
        out = torch.empty(M, n, device=dev, dtype=torch.bfloat16)
        f = lambda: ops.gemm(A, W, b, epi, res, gate, sel, out=out)
        modes = [8, 0] + ([2, 3, 4, 5, 6, 7] if all_h else [])
        r = {m_: [] for m_ in modes}
        for mode in modes:
            lib.fino_tune_set(3, mode); f(); f()
        for _ in range(5):
            for mode in modes:
                lib.fino_tune_set(3, mode)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(5): f()
                e.record(); torch.cuda.synchronize(); r[mode].append(s.elapsed_time(e) / 5 * 1e3)
        lib.fino_tune_set(3, 0)
        t = {m_: statistics.median(v) for m_, v in r.items()}
        fl = 2.0 * M * n * k
        r256, rest = ops.gemm_plan(M, n)
        for m_ in (8, 0):
            tot[m_] = tot.get(m_, 0.0) + (t[m_] if nm != "kv" else 0.0)
        extra = "  ".join(f"{32 * m_}: {t[m_]:6.1f}" for m_ in modes if m_ not in (0, 8))
        print(f"M={M:6d} {nm:9s} N={n:5d} K={k:5d}: 256-row {t[8]:7.1f} us ({fl / t[8] / 1e6:5.0f} TF)  planned "
              f"{t[0]:7.1f} us ({fl / t[0] / 1e6:5.0f} TF) [{r256} rows x 256 + rest x {rest}]  {extra}", flush=True)
    print(f"M={M:6d} layer (qkv + q2 + out + out2 + ffn_up + ffn_down): 256-row {tot[8]:8.1f} us  planned {tot[0]:8.1f} us "
          f"({100 * (tot[0] / tot[8] - 1):+.1f} %)", flush=True)
