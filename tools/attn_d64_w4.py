#!/usr/bin/env python3
"""head_dim 64 (CogVideoX): the 8-wave kernel vs the 4-wave kernel with the folded softmax scale (FINO_TUNE_ATTN_KERNEL = 2,
scale = FINO_ATTN_SCALE_FOLDED, q pre-multiplied): agreement with fp32 SDPA, then interleaved timing at the CogVideoX-5B
shape (B = 2, 48 heads, L = 226 + 18900)."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frameino_amd import _lib, ops
lib = _lib.lib(); dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
c = 64 ** -0.5 * ops.LOG2E


def rel(a, b):
    return ((a.float() - b.float()).pow(2).mean().sqrt() / b.float().pow(2).mean().sqrt()).item()


def ref(q, k, v, heads):
    b, lq, d = q.shape
    f = lambda t: t.view(b, -1, heads, d // heads).transpose(1, 2).float()
    return torch.nn.functional.scaled_dot_product_attention(f(q), f(k), f(v)).transpose(1, 2).reshape(b, lq, d)


for (b, heads, lq, lk) in [(1, 2, 64, 64), (1, 3, 300, 500), (2, 4, 257, 129), (1, 8, 1000, 77), (2, 48, 1024, 1024)]:
    d = heads * 64
    qf = torch.randn(b, lq, d, device=dev, generator=g)
    kv = torch.randn(b, lk, 2 * d, device=dev, generator=g).bfloat16()
    k, v = kv[:, :, :d], kv[:, :, d:]
    q, qs = qf.bfloat16(), (qf * c).bfloat16()
    o8 = ops.attention(q, k, v, heads)
    lib.fino_tune_set(4, 2); o4 = ops.attention(qs, k, v, heads, scale=ops.SCALE_FOLDED); lib.fino_tune_set(4, 0)
    print(f"B{b} H{heads} Lq{lq} Lk{lk}: 8-wave vs SDPA(q) {rel(o8, ref(q, k, v, heads)):.5f}   4-wave folded vs SDPA(q~/c) "
          f"{rel(o4, ref(qs.float() / c, k, v, heads)):.5f}  finite {bool(torch.isfinite(o4.float()).all())}", flush=True)

b, heads, L = 2, 48, 19126
d = heads * 64
qkv = torch.randn(b, L, 3 * d, device=dev, generator=g).bfloat16()
q, k, v = qkv[:, :, :d], qkv[:, :, d:2 * d], qkv[:, :, 2 * d:]
qs = (q.float() * c).bfloat16()
out = torch.empty(b, L, d, device=dev, dtype=torch.bfloat16)
runs = {"8-wave": (0, q, None), "8-wave folded scale": (0, qs, ops.SCALE_FOLDED), "4-wave folded scale": (2, qs, ops.SCALE_FOLDED)}
t = {n: [] for n in runs}
for n, (tk, qq, sc) in runs.items():
    lib.fino_tune_set(4, tk); ops.attention(qq, k, v, heads, out=out, scale=sc)
for _ in range(5):
    for n, (tk, qq, sc) in runs.items():
        lib.fino_tune_set(4, tk)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(2): ops.attention(qq, k, v, heads, out=out, scale=sc)
        e.record(); torch.cuda.synchronize(); t[n].append(s.elapsed_time(e) / 2 * 1e3)
lib.fino_tune_set(4, 0)
fl = 4.0 * b * heads * L * L * 64
print(f"B{b} H{heads} L{L} head_dim 64: " + "   ".join(f"{n} {statistics.median(v_):8.1f} us {fl / statistics.median(v_) / 1e6:5.0f} TF" for n, v_ in t.items()))
