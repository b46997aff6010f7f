#!/usr/bin/env python3
"""Is the GEMM's per-tile fixed cost (prologue + epilogue) bound by the CHIP (every CU reaches its epilogue in the same
microsecond: a 32-MB read + 32-MB write burst per round of tiles) or by the CU (latency / issue of its own 128 + 128 KB)?
Time against K at N = 3072 for tile counts that fill a quarter, a half and all of ONE round of the 256 CUs, and the bench
shape (4.55 rounds): the K = 0 intercept per round is the fixed cost of a tile.  If it shrinks with fewer concurrent tiles
the burst is chip-bound and taking the CUs out of lock step would hide it; if not, it is the CU's own (GPU box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from frameino_amd import ops  # noqa: E402

N = 3072
print(f"# N = {N}; tiles = ceil(M / 256) x 12; time(K) = fixed + slope x K fitted on K = 768 / 1536 / 3072 / 6144")
for m in (1280, 2560, 5376, 24640):
    tiles = (m + 255) // 256 * 12
    rounds = max(1.0, tiles / 256)
    for epi, name in ((0, "bias only"), (2, "residual"), (3, "gated residual")):
        pts = []
        for k in (768, 1536, 3072, 6144):
            a = torch.randn(m, k, device="cuda").bfloat16()
            w = (torch.randn(N, k, device="cuda") * 0.02).bfloat16()
            b = torch.randn(N, device="cuda").bfloat16()
            out = torch.empty(m, N, device="cuda", dtype=torch.bfloat16)
            res = torch.randn(m, N, device="cuda").bfloat16() if epi >= 2 else None
            gate = torch.randn(2, N, device="cuda") if epi == 3 else None
            sel = (torch.arange(m, device="cuda") < 880).to(torch.int32) if epi == 3 else None
            f = lambda: ops.gemm(a, w, b, epi, res, gate, sel, out=out, tile_m=8)      # noqa: E731  (256-row tiles only)
            for _ in range(5):
                f()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(40):
                f()
            e.record()
            torch.cuda.synchronize()
            pts.append((k, s.elapsed_time(e) / 40 * 1e3))
        n_ = len(pts)
        sx = sum(k for k, _ in pts); sy = sum(t for _, t in pts)
        sxx = sum(k * k for k, _ in pts); sxy = sum(k * t for k, t in pts)
        slope = (n_ * sxy - sx * sy) / (n_ * sxx - sx * sx)
        fixed = (sy - slope * sx) / n_
        print(f"M={m:6d} ({tiles:4d} tiles, {tiles / 256:4.2f} rounds) {name:15s}: " + " ".join(f"K={k}:{t:6.1f}us" for k, t in pts) +
              f" | fixed {fixed:6.1f} us = {fixed / (int(rounds) + (1 if rounds % 1 else 0)):5.1f} us per round of tiles")
