#!/usr/bin/env python3
"""Per-layer pieces of the two self-attention exchanges at the Wan2.2-5B shape on one GPU (no wire): what rank 0 of P
token shards launches per layer and CFG branch.  kv: K|V + Q projections, local-first partials + merge; heads: fused QKV
projection, pack, attention over H/P heads x all tokens, unpack."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frameino_amd import ops

dev = "cuda"
L, d, heads, dh = 12320, 3072, 24, 128
g = torch.Generator(device=dev).manual_seed(0)


def t_us(fn, reps=20):
    for _ in range(3): fn()
    ts = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps): fn()
        e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) / reps * 1e3)
    return statistics.median(ts)


for P in (2, 4, 8):
    lpad = (L + P - 1) // P
    n = lpad
    hp, dp = heads // P, (heads // P) * dh
    x = torch.randn(n, d, device=dev, generator=g).bfloat16()
    wqkv = (torch.randn(3 * d, d, device=dev, generator=g) * 0.02).bfloat16()
    bqkv = torch.zeros(3 * d, device=dev).bfloat16()
    qkv = torch.empty(n, 3 * d, device=dev, dtype=torch.bfloat16)
    send = torch.zeros(P, lpad, 3, dp, device=dev, dtype=torch.bfloat16)
    recv = torch.randn(P, lpad, 3, dp, device=dev, generator=g).bfloat16()
    oh = torch.zeros(P, lpad, dp, device=dev, dtype=torch.bfloat16)
    att = torch.empty(n, d, device=dev, dtype=torch.bfloat16)
    r3 = recv.view(1, P * lpad, 3 * dp)[:, :L]
    kv_all = torch.randn(1, P * lpad, 2 * d, device=dev, generator=g).bfloat16()
    q2 = torch.randn(1, n, d, device=dev, generator=g).bfloat16()
    kvl = torch.empty(n, 2 * d, device=dev, dtype=torch.bfloat16)
    from frameino_amd.parallel import TokenShard
    sh = TokenShard.__new__(TokenShard); sh.ways, sh._buf, sh.head_groups = P, {}, 1
    lay = sh.heads_send_layout(heads, dh, lpad, torch.bfloat16, torch.device(dev))
    wn = torch.ones(d, device=dev).bfloat16()
    wo = (torch.randn(d, d, device=dev, generator=g) * 0.02).bfloat16()
    xres = torch.zeros(n, d, device=dev).bfloat16()
    gt = torch.zeros(2, d, device=dev)
    sl = (torch.arange(n, device=dev) % 2).to(torch.int32)
    ang = torch.rand(n, dh // 2, device=dev, generator=g)
    cs, sn = torch.cos(ang).contiguous(), torch.sin(ang).contiguous()
    res = {
        "gemm qkv fused": t_us(lambda: ops.gemm(x, wqkv, bqkv, out=qkv)),
        "gemm kv + gemm q": t_us(lambda: (ops.gemm(x, wqkv[d:], bqkv[d:], out=kvl), ops.gemm(x, wqkv[:d], bqkv[:d], out=att))),
        "pack (torch permute copy)": t_us(lambda: send[:, :n].copy_(qkv.view(n, 3, P, dp).permute(2, 0, 1, 3))),
        "rmsnorm+rope of q and k in place (what either way runs)": t_us(lambda: (
            ops.rmsnorm_rope_(qkv[:, :d], wn, 1e-6, cs, sn, dh), ops.rmsnorm_rope_(qkv[:, d:2 * d], wn, 1e-6, cs, sn, dh))),
        "q | k norm + rope + scatter of q, k, v into the send buffer, ONE launch (round 3: replaces the two lines above)": t_us(
            lambda: ops.qkv_rmsnorm_rope_(qkv, d, wn, 1e-6, wn, 1e-6, cs, sn, dh, out=lay.flat, head_off=lay.off_qkv,
                                          head_ld=lay.ld)),
        "q | k norm + rope in place, ONE launch (round 3, the 1-GPU forward)": t_us(
            lambda: ops.qkv_rmsnorm_rope_(qkv, d, wn, 1e-6, wn, 1e-6, cs, sn, dh)),
        "out-projection (gated residual) on [token, D]": t_us(lambda: ops.gemm(att, wo, bqkv[:d], ops.EPI_GATED_RESIDUAL, xres, gt, sl, out=xres)),
        "out-projection reading the returned blocks (fino_gemm_blocked_a; replaces unpack + the line above)": t_us(
            lambda: ops.gemm_blocked_a(oh, n, wo, bqkv[:d], xres, gt, sl, out=xres)),
        "unpack (torch permute copy)": t_us(lambda: att.view(n, P, dp).copy_(oh[:, :n].permute(1, 0, 2))),
        "attention H/P heads x L x L": t_us(lambda: ops.attention(r3[:, :, :dp], r3[:, :, dp:2 * dp], r3[:, :, 2 * dp:], hp,
                                                                  out=oh.view(1, P * lpad, dp)[:, :L])),
        "attention n x L, all heads, one pass": t_us(lambda: ops.attention(q2, kv_all[:, :L, :d], kv_all[:, :L, d:], heads,
                                                                            out=att.view(1, n, d))),
    }
    parts = lambda: ops.attention_merge(   # noqa: E731
        [ops.attention_partial(q2, kv_all[:, :n, :d], kv_all[:, :n, d:], heads),
         ops.attention_partial(q2, kv_all[:, lpad:L, :d], kv_all[:, lpad:L, d:], heads)], 1, n, heads, dh, torch.bfloat16,
        out=att.view(1, n, d))
    res["attention n x L local-first (2 partials + merge)"] = t_us(parts)
    print(f"P = {P} token shards (n = {n} tokens per rank):")
    for k_, v_ in res.items():
        print(f"    {k_:50s} {v_:8.1f} us")
