#!/usr/bin/env python3
"""Self-attention A/B on the GPU box: tail split on/off at the bench shapes (single GPU batch-2 / batch-1, 4-way token shard)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frameino_amd import ops

D, H = 3072, 24
torch.manual_seed(0)


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


for b, lq, lk in [(2, 12320, 12320), (1, 12320, 12320), (1, 3080, 12320), (2, 3080, 12320), (1, 1540, 12320), (1, 25088, 25088)]:
    q = torch.randn(b, lq, D, device="cuda").bfloat16()
    kv = torch.randn(b, lk, 2 * D, device="cuda").bfloat16()
    o = torch.empty_like(q)
    res = []
    for split in (False, True, False, True):  # noqa
        ops.SPLIT_ATTENTION_TAIL = split
        t = timeit(lambda: ops.attention(q, kv[:, :, :D], kv[:, :, D:], H, out=o))
        res.append(4.0 * b * lq * lk * D / t / 1e12)
    print(f"B={b} Lq={lq} Lk={lk}: unsplit {res[0]:.0f}/{res[2]:.0f} TF  split {res[1]:.0f}/{res[3]:.0f} TF", flush=True)
