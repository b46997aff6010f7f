#!/usr/bin/env python3
"""Stream-K of fino_gemm_ws forced off / on / default policy (FINO_TUNE_GEMM_STREAM_K = 1 / 2 / 0) at every block GEMM of a
Wan2.2-5B layer, for the one-GPU row counts (24640 = both CFG branches, 12320) and the token-shard ones (6160 / 3080 /
1540), interleaved on one box, median of 7 x 5 launches.  usage: gemm_sk_ab.py [rows ...]"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frameino_amd import _lib, ops
lib = _lib.lib()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
D, FF = 3072, 14336
rows = [int(x) for x in sys.argv[1:]] or [24640, 12320, 6160, 3080, 1540]
shapes = [("qkv", 3 * D, D, 0), ("kv", 2 * D, D, 0), ("q/q2", D, D, 0), ("out", D, D, 3), ("out2", D, D, 2),
          ("ffn_up", FF, D, 1), ("ffn_down", D, FF, 3)]
tot = {}
for M in rows:
    for nm, n, k, epi in shapes:
        A = torch.randn(M, k, device=dev, generator=g).bfloat16()
        W = (torch.randn(n, k, device=dev, generator=g) * 0.02).bfloat16()
        b = torch.randn(n, device=dev, generator=g).bfloat16()
        res = torch.randn(M, n, device=dev, generator=g).bfloat16() if epi >= 2 else None
        gate = torch.randn(2, n, device=dev, generator=g) if epi >= 3 else None
        sel = (torch.arange(M, device=dev) % 2).to(torch.int32) if epi >= 3 else None
        out = torch.empty(M, n, device=dev, dtype=torch.bfloat16)
        f = lambda: ops.gemm(A, W, b, epi, res, gate, sel, out=out)
        r = {0: [], 1: [], 2: []}
        for mode in (1, 2, 0):
            lib.fino_tune_set(3, mode); f(); f()
        for _ in range(7):
            for mode in (1, 2, 0):
                lib.fino_tune_set(3, mode)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(5): f()
                e.record(); torch.cuda.synchronize(); r[mode].append(s.elapsed_time(e) / 5 * 1e3)
        lib.fino_tune_set(3, 2); ws2 = lib.fino_gemm_workspace_bytes(M, n, k)
        lib.fino_tune_set(3, 0); ws0 = lib.fino_gemm_workspace_bytes(M, n, k)
        t = {m_: statistics.median(v) for m_, v in r.items()}
        tiles = -(-M // 256) * -(-n // 256)
        fl = 2.0 * M * n * k
        for m_ in t:
            tot[(M, m_)] = tot.get((M, m_), 0.0) + t[m_]
        print(f"M={M:6d} {nm:9s} N={n:5d} K={k:5d} tiles {tiles:5d} ({tiles / 256:5.2f} rounds): whole {t[1]:7.1f} us "
              f"({fl / t[1] / 1e6:5.0f} TF)  stream-K {t[2]:7.1f} us ({fl / t[2] / 1e6:5.0f} TF){'' if ws2 else ' [n/a]'}  "
              f"default {t[0]:7.1f} us [{'sk' if ws0 else 'whole'}]", flush=True)
    print(f"M={M:6d} layer sum (qkv + q/q2 + out + out2 + ffn_up + ffn_down, kv not counted): whole "
          f"{tot[(M, 1)] - 0:8.1f} us  stream-K {tot[(M, 2)]:8.1f} us  default {tot[(M, 0)]:8.1f} us  (all seven lines summed)")
