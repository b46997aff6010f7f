#!/usr/bin/env python3
"""Launch forms of the ping-pong GEMM for the block GEMMs of a Wan2.2-5B layer with the epilogues the forward uses, interleaved
on one box (median of 7 x 5 launches); results must be bit-identical.  FINO_TUNE_GEMM_RASTER: bit 0 = PERSISTENT workgroups
(one per CU) that chain their tiles (the next tile's prologue DMA issued in front of the epilogue), value >> 4 = L2 warm-up
distance in K-tiles (each wave touches the next tile's first A K-tiles that many K-tiles before its loop ends).
usage: [FINO_AB_MODES=0,1,48,49] gemm_persist_ab.py [rows ...]"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frameino_amd import _lib, ops
lib = _lib.lib()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
D, FF = 3072, 14336
rows = [int(x) for x in sys.argv[1:]] or [24640, 12320, 3080]
shapes = [("qkv", 3 * D, D, 0), ("q2", D, D, 0), ("out", D, D, 3), ("out2", D, D, 2), ("ffn_up", FF, D, 1), ("ffn_down", D, FF, 3)]
for M in rows:
    MODES = [int(x) for x in os.environ.get("FINO_AB_MODES", "0,1,48,49,96,97").split(",")]
    tot = {m_: 0.0 for m_ in MODES}
    for nm, n, k, epi in shapes:
        A = torch.randn(M, k, device=dev, generator=g).bfloat16()
        W = (torch.randn(n, k, device=dev, generator=g) * 0.02).bfloat16()
        b = torch.randn(n, device=dev, generator=g).bfloat16()
        res = torch.randn(M, n, device=dev, generator=g).bfloat16() if epi >= 2 else None
        gate = torch.randn(2, n, device=dev, generator=g) if epi >= 3 else None
        sel = ((torch.arange(M, device=dev) % 12320) >= 880).to(torch.int32) if epi >= 3 else None       # FrameINO's selector
        outs = {}
        r = {m_: [] for m_ in MODES}
        for mode in MODES:
            lib.fino_tune_set(1, mode)
            outs[mode] = ops.gemm(A, W, b, epi, res, gate, sel)
        same = all(torch.equal(outs[MODES[0]], outs[m_]) for m_ in MODES)
        out = torch.empty(M, n, device=dev, dtype=torch.bfloat16)
        f = lambda: ops.gemm(A, W, b, epi, res, gate, sel, out=out)      # noqa: E731
        for _ in range(7):
            for mode in MODES:
                lib.fino_tune_set(1, mode)
                f()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(5):
                    f()
                e.record(); torch.cuda.synchronize(); r[mode].append(s.elapsed_time(e) / 5 * 1e3)
        lib.fino_tune_set(1, 0)
        t = {m_: statistics.median(v) for m_, v in r.items()}
        fl = 2.0 * M * n * k
        for m_ in MODES:
            tot[m_] += t[m_]
        base = t[MODES[0]]
        print(f"M={M:6d} {nm:9s} N={n:5d} K={k:5d} epi={epi}: " + "  ".join(
            f"[{m_}] {t[m_]:7.1f} us {100 * (t[m_] / base - 1):+5.1f}%" for m_ in MODES) + f"  bit-identical={same}", flush=True)
    print(f"M={M:6d} layer sum: " + "  ".join(f"[{m_}] {tot[m_]:8.1f} us {100 * (tot[m_] / tot[MODES[0]] - 1):+5.1f}%" for m_ in MODES),
          flush=True)
