#!/usr/bin/env python3
"""Diagnostic: does the K / V row pitch matter to the self-attention kernel?  The same attention with K / V rows (a) inside the
fused q | k | v projection output (row pitch 3*H*Dh*2 = 18432 B, what the model runs), (b) in their own [L, H*Dh] tensors
(6144 B), (c) per head contiguous [H][L][Dh] (256 B: a 64-key tile is 16 KiB of consecutive bytes), (d) fused with the row
pitch padded by 128 / 256 / 512 B."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frameino_amd import _lib, ops
from frameino_amd.ops import _p, _dt, _stream

D, H, DH = 3072, 24, 128
B, L = 2, 12320
lib = _lib.lib()
torch.manual_seed(0)


def run(q, k, v, o, k_strides, v_strides):
    ws, wsb = ops._attention_workspace(B, H, L, L, DH, q.device)
    _lib.check(lib.fino_attn_fwd_ws(_p(q), _p(k), _p(v), _p(o), B, H, L, L, DH, q.stride(0), q.stride(1), DH,
                                    *k_strides, *v_strides, o.stride(0), o.stride(1), DH, float(DH ** -0.5), _dt(q),
                                    _p(ws), wsb, _stream()), "attn")


def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


src = torch.randn(B, L, 3 * D, device="cuda").bfloat16()
o = torch.empty(B, L, D, device="cuda", dtype=torch.bfloat16)
cases = {}
q = src[:, :, :D]
cases["fused q|k|v rows (18432 B)"] = (q, src[:, :, D:2 * D], src[:, :, 2 * D:], None, None)
kc, vc = src[:, :, D:2 * D].contiguous(), src[:, :, 2 * D:].contiguous()
cases["own tensors (6144 B)"] = (q, kc, vc, None, None)
kh = src[:, :, D:2 * D].view(B, L, H, DH).permute(0, 2, 1, 3).contiguous()      # [B, H, L, Dh]
vh = src[:, :, 2 * D:].view(B, L, H, DH).permute(0, 2, 1, 3).contiguous()
cases["per head contiguous (256 B)"] = (q, kh, vh, (H * L * DH, DH, L * DH), (H * L * DH, DH, L * DH))
for pad in (64, 128, 256):      # elements
    wide = torch.zeros(B, L, 3 * D + pad, device="cuda", dtype=torch.bfloat16)
    wide[:, :, :3 * D] = src
    cases[f"fused, row pitch + {2 * pad} B"] = (wide[:, :, :D], wide[:, :, D:2 * D], wide[:, :, 2 * D:3 * D], None, None)
ref = None
for rnd in range(2):
    for name, (qq, k, v, ks, vs) in cases.items():
        ks = ks or (k.stride(0), k.stride(1), DH)
        vs = vs or (v.stride(0), v.stride(1), DH)
        t = timeit(lambda: run(qq, k, v, o, ks, vs))
        if ref is None: ref = o.clone()
        assert torch.equal(o, ref), name
        print(f"{name:36s} {t * 1e6:8.1f} us  {4.0 * B * L * L * D / t / 1e12:7.0f} TFLOP/s", flush=True)
