#!/usr/bin/env python3
"""Self-attention kernel A/B on the GPU box: FINO_TUNE_ATTN_KERNEL values (default 1 = register-staged ping-pong vs 4 = the
LDS-DMA-staged one) at the bench shapes, alternating, same buffers.  usage: attn_kernel_ab.py [kernel ids ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frameino_amd import _lib, ops

D, H = 3072, int(os.environ.get("FINO_AB_HEADS", 24))       # FINO_AB_HEADS=48: head_dim 64 (the CogVideoX-5B shape)
ids = [int(x) for x in sys.argv[1:]] or [1, 4]
lib = _lib.lib()
torch.manual_seed(0)


def timeit(fn, iters=20, warm=3):
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


SHAPES = [(2, 12320, 12320), (1, 12320, 12320), (2, 3080, 12320), (1, 1540, 12320), (2, 25088, 25088)]
if os.environ.get("FINO_AB_CROSS"):      # the text cross-attention shapes (1 GPU, 4- and 8-way token shards)
    SHAPES = [(2, 12320, 512), (1, 12320, 512), (2, 3080, 512), (2, 1540, 512), (2, 25088, 512)]
if os.environ.get("FINO_AB_SHAPES"):     # "b,lq,lk;b,lq,lk;..."
    SHAPES = [tuple(int(x) for x in t.split(",")) for t in os.environ["FINO_AB_SHAPES"].split(";")]
for b, lq, lk in SHAPES:
    qkv = torch.randn(b, max(lq, lk), 3 * D, device="cuda").bfloat16()
    q, k, v = qkv[:, :lq, :D], qkv[:, :lk, D:2 * D], qkv[:, :lk, 2 * D:]
    o = torch.empty(b, lq, D, device="cuda", dtype=torch.bfloat16)
    res = {i: [] for i in ids}
    for rep in range(3):
        for i in ids:
            lib.fino_tune_set(4, i)
            t = timeit(lambda: ops.attention(q, k, v, H, out=o))
            res[i].append(4.0 * b * lq * lk * D / t / 1e12)
    lib.fino_tune_set(4, 0)
    print(f"B={b} Lq={lq} Lk={lk}: " + "   ".join(f"kernel {i}: " + "/".join(f"{x:.0f}" for x in res[i]) + " TF" for i in ids), flush=True)
