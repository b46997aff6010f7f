#!/usr/bin/env python3
"""Stream-K tail of fino_gemm_ws forced on / off (FINO_TUNE_GEMM_STREAM_K = 2 / 1) at the FFN-down shapes, interleaved,
median.  (The default, 0, splits only when the tail fills <= 40 % of a round.)"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frameino_amd import _lib, ops
lib = _lib.lib()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
for M, n, k, nm in [(24640, 3072, 14336, "Wan FFN-down B=2"), (12320, 3072, 14336, "Wan FFN-down B=1"),
                    (38252, 3072, 12288, "Cog FFN-down B=2"),
                    # token shards (bench.py --gpus 4 / 8): fewer tiles than CUs, dealt whole by default (0)
                    (6160, 3072, 14336, "FFN-down 2 shards"), (3080, 3072, 14336, "FFN-down 4 shards"),
                    (1540, 3072, 14336, "FFN-down 8 shards")]:
    A = torch.randn(M, k, device=dev, generator=g).bfloat16()
    W = (torch.randn(n, k, device=dev, generator=g) * 0.02).bfloat16()
    b = torch.randn(n, device=dev, generator=g).bfloat16()
    res = torch.randn(M, n, device=dev, generator=g).bfloat16()
    gate = torch.randn(2, n, device=dev, generator=g)
    sel = (torch.arange(M, device=dev) % 2).to(torch.int32)
    out = torch.empty(M, n, device=dev, dtype=torch.bfloat16)
    f = lambda: ops.gemm(A, W, b, 3, res, gate, sel, out=out)
    r = {2: [], 1: []}
    outs = {}
    lib.fino_tune_set(3, 0); f(); f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): f()
    e.record(); torch.cuda.synchronize(); t_def = s.elapsed_time(e) / 10 * 1e3
    for off in (1, 2):
        lib.fino_tune_set(3, off); f(); f(); outs[off] = out.clone()
    for _ in range(7):
        for off in (1, 2):
            lib.fino_tune_set(3, off)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5): f()
            e.record(); torch.cuda.synchronize(); r[off].append(s.elapsed_time(e) / 5 * 1e3)
    lib.fino_tune_set(3, 0)
    d = (outs[2].float() - outs[1].float()).abs().max().item()
    t1, t0 = statistics.median(r[1]), statistics.median(r[2])
    print(f"{nm:18s} {M}x{n}x{k}: whole tiles {t1:7.1f} us ({2.0*M*n*k/t1/1e6:5.0f} TF)  stream-K tail {t0:7.1f} us "
          f"({2.0*M*n*k/t0/1e6:5.0f} TF)  ws {lib.fino_gemm_workspace_bytes(M, n, k) / 2**20:.0f} MiB  max|diff| {d:.4f}  default {t_def:7.1f} us")
