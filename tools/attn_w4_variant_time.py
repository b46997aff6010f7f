#!/usr/bin/env python3
"""Timing of the 4-wave folded-scale attention kernel from the library named by FINO_LIB_PATH (experiment builds:
`make -C frameino_amd/csrc variant NAME=x VFLAGS="-DFINO_EXPERIMENT -DW4_X_..."`, run with FINO_ALLOW_EXPERIMENT=1; their results are wrong by design, only the time counts).
usage: attn_w4_variant_time.py [head_dim=64] [L=19126] [heads=48]"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frameino_amd import _lib, ops
hd, L, heads = (int(x) for x in (sys.argv[1:4] + ["64", "19126", "48"][len(sys.argv) - 1:]))
lib = _lib.lib()
g = torch.Generator(device="cuda").manual_seed(0)
d = heads * hd
qkv = torch.randn(2, L, 3 * d, device="cuda", generator=g).bfloat16()
qs = (qkv[:, :, :d].float() * (hd ** -0.5 * ops.LOG2E)).bfloat16()
k, v = qkv[:, :, d:2 * d], qkv[:, :, 2 * d:]
out = torch.empty(2, L, d, device="cuda", dtype=torch.bfloat16)
lib.fino_tune_set(4, 2)
for _ in range(3): ops.attention(qs, k, v, heads, out=out, scale=ops.SCALE_FOLDED)
ts = []
for _ in range(7):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(2): ops.attention(qs, k, v, heads, out=out, scale=ops.SCALE_FOLDED)
    e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) / 2 * 1e3)
us = statistics.median(ts)
print(f"{os.path.basename(os.environ.get('FINO_LIB_PATH', 'libframeino_hip.so')):28s} head_dim {hd} L {L}: {us:8.1f} us  "
      f"{4.0 * 2 * heads * L * L * hd / us / 1e6:5.0f} TFLOP/s", flush=True)
