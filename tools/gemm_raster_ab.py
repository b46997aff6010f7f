#!/usr/bin/env python3
"""How much does the GEMM's L2-fill (fabric) traffic cost?  The four block GEMMs of a Wan layer at the bench's M = 24640
rows, timed under different raster group heights (FINO_TUNE_GEMM_GROUP_M: an XCD's 32 concurrent tiles form a
G-row x 32/G-column window, so one round of an XCD fetches G + 32/G operand panels: 33 / 18 / 12 / 12 / 18 / 33 for
G = 1 / 2 / 4 / 8 / 16 / 32).  Interleaved rounds in one process (median of `--rounds`).  GPU box only.

    python tools/gemm_raster_ab.py                      # A/B table
    python tools/gemm_raster_ab.py --group-m 1 --pmc    # few launches of one setting, for a rocprofv3 --pmc pass
"""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from frameino_amd import _lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=24640)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--iters", type=int, default=6)
ap.add_argument("--group-m", type=int, nargs="*", default=[1, 2, 4, 8, 16, 32])
ap.add_argument("--raster", type=int, nargs="*", default=[0])
ap.add_argument("--pmc", action="store_true", help="3 launches per shape of the first --group-m only")
a = ap.parse_args()
lib = _lib.lib()
M, D, F = a.rows, 3072, 14336
dev = "cuda"
shapes = [(3 * D, D, 0, "qkv"), (D, D, 3, "out+gate"), (F, D, 1, "ffn-up+gelu"), (D, F, 3, "ffn-down+gate")]
g = torch.Generator(device=dev).manual_seed(0)
cases = []
for n, k, epi, nm in shapes:
    A = torch.randn(M, k, device=dev, generator=g).bfloat16()
    W = (torch.randn(n, k, device=dev, generator=g) * 0.02).bfloat16()
    b = torch.randn(n, device=dev, generator=g).bfloat16()
    out = torch.empty(M, n, device=dev, dtype=torch.bfloat16)
    res = torch.randn(M, n, device=dev, generator=g).bfloat16() if epi == 3 else None
    gate = torch.randn(2, n, device=dev, generator=g) if epi == 3 else None
    sel = (torch.arange(M, device=dev) % (M // 2) >= 880).to(torch.int32) if epi == 3 else None
    cases.append((nm, n, k, lambda A=A, W=W, b=b, epi=epi, res=res, gate=gate, sel=sel, out=out:
                  ops.gemm(A, W, b, epi, res, gate, sel, out=out)))


def timed(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


if a.pmc:
    lib.fino_tune_set(0, a.group_m[0])
    lib.fino_tune_set(1, a.raster[0])
    for nm, n, k, fn in cases:
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    sys.exit(0)

settings = [(r, gm) for r in a.raster for gm in a.group_m]
res = {(nm, s): [] for nm, *_ in cases for s in settings}
for nm, n, k, fn in cases:
    for s in settings:
        lib.fino_tune_set(1, s[0]); lib.fino_tune_set(0, s[1])
        timed(fn, 2)
    for _ in range(a.rounds):
        for s in settings:
            lib.fino_tune_set(1, s[0]); lib.fino_tune_set(0, s[1])
            res[(nm, s)].append(timed(fn, a.iters))
lib.fino_tune_set(0, 0); lib.fino_tune_set(1, 0)
print(f"M = {M} rows; median us (TFLOP/s) per setting (raster, group_m)")
tot = {s: 0.0 for s in settings}
for nm, n, k, fn in cases:
    line = f"{nm:14s} {M}x{n}x{k}: "
    for s in settings:
        t = statistics.median(res[(nm, s)])
        tot[s] += t
        line += f" r{s[0]}g{s[1]}: {t:7.1f} ({2.0 * M * n * k / t / 1e6:5.0f})"
    print(line)
print("sum".ljust(14) + " " * 20 + " ".join(f" r{s[0]}g{s[1]}: {tot[s]:7.1f}        " for s in settings))
