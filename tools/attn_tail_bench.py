#!/usr/bin/env python3
"""Text cross-attention at the bench shape (B = 2 CFG branches, 24 heads x 128, Lq = 12320, prompts of 64 / 8 tokens zero-padded to
512): the plain launch over all 512 keys against fino_attn_fwd_tail over the real keys + ONE key that stands for the padding run."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frameino_amd import ops
b, heads, L, total = 2, 24, int(sys.argv[1]) if len(sys.argv) > 1 else 12320, 512
n_real = (64, 8)
d = heads * 128
g = torch.Generator(device="cuda").manual_seed(0)
q = torch.randn(b, L, d, device="cuda", generator=g).bfloat16()
kv = torch.randn(b, total, 2 * d, device="cuda", generator=g).bfloat16()
for i, n in enumerate(n_real):
    kv[i, n:] = kv[i, n].clone()
k, v = kv[:, :, :d], kv[:, :, d:]
o = torch.empty_like(q)
runs = {"plain, 512 keys": lambda: ops.attention(q, k, v, heads, out=o)}
for lc in (128, 192):
    runs[f"tail, {lc} rows allocated"] = (lambda lc=lc: ops.attention_tail(q, k[:, :lc], v[:, :lc], heads, [n + 1 for n in n_real],
                                                                            [total - n for n in n_real], out=o))
t = {n: [] for n in runs}
for n, f in runs.items():
    f(); f()
for _ in range(7):
    for n, f in runs.items():
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): f()
        e.record(); torch.cuda.synchronize(); t[n].append(s.elapsed_time(e) / 5 * 1e3)
for n in runs:
    print(f"{n:28s} {statistics.median(t[n]):8.1f} us")
