#!/usr/bin/env python3
"""FFN-up -> FFN-down as the forward runs them (M = 24640), with the FFN-down tile rows walked first-to-last (product) or last-to-first
(the default since round 6; FINO_TUNE_GEMM_RASTER = 2 restores first-to-last): does FFN-down find the rows FFN-up wrote LAST still in the Infinity Cache?  (its 706 MB A operand does not fit the
256 MiB cache; profiles/r06_gemm_energy_dropone.txt: staging is 25 % of FFN-down's joules against 17 % elsewhere).  Interleaved rounds, one box."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frameino_amd import _lib, ops
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
M, D, F, L = 24640, 3072, 14336, 12320
bf = torch.bfloat16
rn = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).to(bf)
x, w1, b1, w2, b2 = rn(M, D), rn(F, D, sc=0.02), rn(F), rn(D, F, sc=0.02), rn(D)
mod = torch.randn(2, 6, D, device=dev, generator=g) * 0.1
sel = (torch.arange(M, device=dev) % L >= 880).to(torch.int32)
ff, out = torch.empty(M, F, device=dev, dtype=bf), torch.empty(M, D, device=dev, dtype=bf)
lib = _lib.lib()
def pair():
    ops.gemm(x, w1, b1, ops.EPI_GELU_TANH, out=ff)
    ops.gemm(ff, w2, b2, ops.EPI_GATED_RESIDUAL, x, mod[:, 5], sel, out=out)
res = {0: [], 1: []}
ref = None
for rnd in range(6):
    for knob in (0, 1):
        lib.fino_tune_set(1, 2 if knob == 0 else 0)
        pair(); pair()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): pair()
        e.record(); torch.cuda.synchronize()
        res[knob].append(s.elapsed_time(e) / 10 * 1e3)
        if ref is None: ref = out.clone()
        assert torch.equal(out, ref)
lib.fino_tune_set(1, 0)
for knob in (0, 1):
    print(f"FFN-up + FFN-down pair, FFN-down rows {'last to first' if knob else 'first to last'}: median {statistics.median(res[knob]):.1f} us  ({' '.join(f'{v:.0f}' for v in res[knob])})")
