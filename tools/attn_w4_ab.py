#!/usr/bin/env python3
"""8-wave ping-pong attention kernel vs the 4-wave one-wave-per-SIMD kernel (FINO_TUNE_ATTN_KERNEL = 1 / 2), head_dim 128:
agreement on ragged shapes, then interleaved timing at the bench shape."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frameino_amd import _lib, ops
lib = _lib.lib()
KEY = 4
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)


def rel(a, b):
    return ((a.float() - b.float()).pow(2).mean().sqrt() / b.float().pow(2).mean().sqrt()).item()


def ref(q, k, v, heads):
    b, lq, d = q.shape
    dh = d // heads
    f = lambda t: t.view(b, -1, heads, dh).transpose(1, 2).float()
    return torch.nn.functional.scaled_dot_product_attention(f(q), f(k), f(v)).transpose(1, 2).reshape(b, lq, d)


for (b, heads, lq, lk, qscale) in [(1, 2, 64, 64, 1.0), (1, 2, 300, 500, 1.0), (2, 3, 257, 129, 1.0), (1, 8, 1000, 77, 1.0),
                                   (1, 2, 513, 2000, 8.0), (2, 24, 1024, 1024, 1.0)]:
    d = heads * 128
    q = (torch.randn(b, lq, d, device=dev, generator=g) * qscale).bfloat16()
    kv = torch.randn(b, lk, 2 * d, device=dev, generator=g).bfloat16()
    k, v = kv[:, :, :d], kv[:, :, d:]
    lib.fino_tune_set(KEY, 1); o1 = ops.attention(q, k, v, heads)
    lib.fino_tune_set(KEY, 2); o2 = ops.attention(q, k, v, heads)
    lib.fino_tune_set(KEY, 0)
    r = ref(q, k, v, heads)
    print(f"B{b} H{heads} Lq{lq} Lk{lk} qx{qscale}: w8 vs fp32 {rel(o1, r):.5f}  w4 vs fp32 {rel(o2, r):.5f}  w4 vs w8 {rel(o2, o1):.5f}"
          f"  finite {bool(torch.isfinite(o2.float()).all())}", flush=True)

for (b, heads, L) in [(2, 24, 12288), (2, 24, 12320)]:
    d = heads * 128
    qkv = torch.randn(b, L, 3 * d, device=dev, generator=g).bfloat16()
    q, k, v = qkv[:, :, :d], qkv[:, :, d:2 * d], qkv[:, :, 2 * d:]
    out = torch.empty(b, L, d, device=dev, dtype=torch.bfloat16)
    t = {1: [], 2: []}
    for kk in (1, 2):
        lib.fino_tune_set(KEY, kk); ops.attention(q, k, v, heads, out=out)
    for _ in range(5):
        for kk in (1, 2):
            lib.fino_tune_set(KEY, kk)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(3): ops.attention(q, k, v, heads, out=out)
            e.record(); torch.cuda.synchronize(); t[kk].append(s.elapsed_time(e) / 3 * 1e3)
    lib.fino_tune_set(KEY, 0)
    fl = 4.0 * b * heads * L * L * 128
    a, c = statistics.median(t[1]), statistics.median(t[2])
    print(f"B{b} H{heads} L{L}: w8 {a:8.1f} us {fl / a / 1e6:6.0f} TF   w4 {c:8.1f} us {fl / c / 1e6:6.0f} TF", flush=True)
