#!/usr/bin/env python3
"""Two measurements behind profiles/r05_energy_ledger.txt (VERDICT r4 item 6 named them as candidates for the step's largest
non-MFMA energy lines): what do (a) the clamped rows of the self-attention's last q-block and (b) the partial last round of the
N = 3072 GEMMs cost, in time and in joules?

(a) L = 12320 is 48 q-blocks of 256 rows + one block with 32 valid rows; its other 224 rows are ZERO operands (never stored).
    Compared: the same launch at L = 12288 (48 whole blocks) -- per valid FLOP.
(b) the gated-residual out-projection (N = K = 3072) at row counts that make 3.0 / 3.75 / 4.5 / 4.55 (bench) / 4.69 / 5.0 rounds
    of 256 x 256 tiles on 256 CUs (the planner may lower the last rows' tiles: fino_gemm_plan is printed) -- per FLOP."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from energy_ledger import idle_power, measure  # noqa: E402
from frameino_amd import ops  # noqa: E402

dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
bf = torch.bfloat16
p_idle = idle_power()
print(f"# idle {p_idle:.0f} W")
H, D = 24, 3072
for L in (12288, 12320, 12544):
    qkv = torch.randn(2, L, 3 * D, device=dev, generator=g).to(bf)
    out = torch.empty(2, L, D, device=dev, dtype=bf)
    f = lambda: ops.attention(qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:], H, out=out)      # noqa: E731
    us, w = measure(f, batch=4)
    fl = 4.0 * 2 * L * L * D
    print(f"self-attention L = {L:5d} ({-(-L // 256)} q-blocks, {L % 256 or 256:3d} rows in the last): {us:8.1f} us  {fl / us / 1e6:6.0f} TFLOP/s  "
          f"{w:5.0f} W  {(w - p_idle) * us * 1e-6 / fl * 1e12:.4f} pJ/FLOP dynamic", flush=True)
    del qkv, out
K = N = 3072
w_ = (torch.randn(N, K, device=dev, generator=g) * 0.02).to(bf)
b_ = torch.randn(N, device=dev, generator=g).to(bf)
gate = torch.randn(2, N, device=dev, generator=g)
for M in (16384, 20480, 24576, 24640, 25600, 27306):
    a = torch.randn(M, K, device=dev, generator=g).to(bf)
    r = torch.randn(M, N, device=dev, generator=g).to(bf)
    sel = (torch.arange(M, device=dev) % 2).to(torch.int32)
    o = torch.empty(M, N, device=dev, dtype=bf)
    f = lambda: ops.gemm(a, w_, b_, ops.EPI_GATED_RESIDUAL, r, gate, sel, out=o)      # noqa: E731
    us, w = measure(f)
    fl = 2.0 * M * N * K
    plan = ops.gemm_plan(M, N)
    tiles = -(-M // 256) * 12
    print(f"out-projection M = {M:5d} ({tiles} tiles of 256 = {tiles / 256:.2f} rounds; plan {plan}): {us:7.1f} us  {fl / us / 1e6:6.0f} TFLOP/s  "
          f"{w:5.0f} W  {(w - p_idle) * us * 1e-6 / fl * 1e12:.4f} pJ/FLOP dynamic", flush=True)
    del a, r, o
