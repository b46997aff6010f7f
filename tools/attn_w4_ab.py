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
    qf = torch.randn(b, lq, d, device=dev, generator=g) * qscale          # "fp32 q" as the norm + RoPE kernel holds it
    q = qf.bfloat16()
    qs = (qf * (128 ** -0.5 * ops.LOG2E)).bfloat16()                        # what rmsnorm_rope_(out_scale=...) writes
    kv = torch.randn(b, lk, 2 * d, device=dev, generator=g).bfloat16()
    k, v = kv[:, :, :d], kv[:, :, d:]
    lib.fino_tune_set(KEY, 1); o1 = ops.attention(q, k, v, heads); o1f = ops.attention(qs, k, v, heads, scale=ops.SCALE_FOLDED)
    lib.fino_tune_set(KEY, 2); o2 = ops.attention(q, k, v, heads); o2f = ops.attention(qs, k, v, heads, scale=ops.SCALE_FOLDED)
    lib.fino_tune_set(KEY, 0)
    r = ref(qf, k, v, heads)
    print(f"B{b} H{heads} Lq{lq} Lk{lk} qx{qscale} vs SDPA(fp32 q): w8 {rel(o1, r):.5f}  w8 folded-scale {rel(o1f, r):.5f}  "
          f"w4 {rel(o2, r):.5f}  w4 folded-scale (MFMA fold) {rel(o2f, r):.5f}  finite {bool(torch.isfinite(o2f.float()).all())}", flush=True)

# partials over key ranges + merge through the 4-wave kernel
for (b, heads, lq, splits) in [(2, 3, 300, (0, 64, 500)), (1, 24, 3080, (0, 3080, 6160, 12320))]:
    d = heads * 128
    q = torch.randn(b, lq, d, device=dev, generator=g).bfloat16()
    kv = torch.randn(b, splits[-1], 2 * d, device=dev, generator=g).bfloat16()
    k, v = kv[:, :, :d], kv[:, :, d:]
    lib.fino_tune_set(KEY, 1); full = ops.attention(q, k, v, heads)
    lib.fino_tune_set(KEY, 2)
    parts = [ops.attention_partial(q, k[:, a:c], v[:, a:c], heads) for a, c in zip(splits[:-1], splits[1:])]
    mer = ops.attention_merge(parts, b, lq, heads, 128, q.dtype)
    lib.fino_tune_set(KEY, 0)
    print(f"partials B{b} H{heads} Lq{lq} ranges {splits}: merged(w4) vs single pass(w8) {rel(mer, full):.5f}", flush=True)

for (b, heads, L, lkv) in [(2, 24, 12288, 0), (2, 24, 12320, 0), (1, 24, 12320, 0), (2, 24, 25088, 0), (1, 24, 3080, 12320),
                           (2, 24, 12320, 512)]:
    d = heads * 128
    lkv = lkv or L
    qkv = torch.randn(b, max(L, lkv), 3 * d, device=dev, generator=g).bfloat16()
    q, k, v = qkv[:, :L, :d], qkv[:, :lkv, d:2 * d], qkv[:, :lkv, 2 * d:]
    out = torch.empty(b, L, d, device=dev, dtype=torch.bfloat16)
    qsc = (q.float() * (128 ** -0.5 * ops.LOG2E)).bfloat16()      # the folded-scale runs get q pre-multiplied, as the model does
    t = {1: [], 2: [], 3: []}
    runs = {1: (1, None), 2: (2, None), 3: (2, ops.SCALE_FOLDED)}
    for kk, (tk, sc) in runs.items():
        lib.fino_tune_set(KEY, tk); ops.attention(q if sc is None else qsc, k, v, heads, out=out, scale=sc)
    for _ in range(5):
        for kk, (tk, sc) in runs.items():
            lib.fino_tune_set(KEY, tk)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(3): ops.attention(q if sc is None else qsc, k, v, heads, out=out, scale=sc)
            e.record(); torch.cuda.synchronize(); t[kk].append(s.elapsed_time(e) / 3 * 1e3)
    lib.fino_tune_set(KEY, 0)
    fl = 4.0 * b * heads * L * lkv * 128
    a, c, f3 = statistics.median(t[1]), statistics.median(t[2]), statistics.median(t[3])
    print(f"B{b} H{heads} Lq{L} Lk{lkv}: w8 {a:8.1f} us {fl / a / 1e6:6.0f} TF   w4 {c:8.1f} us {fl / c / 1e6:6.0f} TF   "
          f"w4 folded scale {f3:8.1f} us {fl / f3 / 1e6:6.0f} TF", flush=True)
