"""Unit parity of every HIP kernel (through the C ABI) against a plain PyTorch fp32 restatement of the same
operator with the reference's rounding points.  Needs a real MI355X: `pytest -m gpu`."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from frameino_amd import ops as o
    return o


def rnd(*shape, dtype=torch.bfloat16, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(DEV)


def rel_rms(a, b):
    a, b = a.float(), b.float()
    return ((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-12)).item()


def ulp_close(a, b, dtype, max_ulp_frac=0.01):
    """bf16/fp16 outputs of an fp32 computation: allow a tiny fraction of 1-ulp flips (reduction order)."""
    a, b = a.float(), b.float()
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    bad = (a - b).abs() > (2 * eps * b.abs() + 1e-6)
    assert bad.float().mean().item() <= max_ulp_frac, f"{bad.float().mean().item():.4f} of elements off by >2ulp"
    assert rel_rms(a, b) < 2 * eps


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rows,dim", [(7, 48), (1001, 3072), (33, 1536)])
def test_adaln_modulate(ops, dtype, rows, dim):
    x = rnd(rows, dim, dtype=dtype, seed=1)
    table = rnd(2, 6, dim, dtype=torch.float32, seed=2, scale=0.5)
    sel = (torch.arange(rows) % 3 == 0).to(torch.int32).to(DEV)
    shift, scale = table[:, 0], table[:, 1]
    y = ops.adaln_modulate(x, shift, scale, sel, eps=1e-6)
    n = F.layer_norm(x.float(), (dim,), None, None, 1e-6)
    ref = (n * (1 + scale[sel.long()]) + shift[sel.long()]).to(dtype)
    ulp_close(y, ref, dtype)
    # no selector -> row 0
    y0 = ops.adaln_modulate(x, shift[0], scale[0], None, eps=1e-6)
    ref0 = (n * (1 + scale[0]) + shift[0]).to(dtype)
    ulp_close(y0, ref0, dtype)


@pytest.mark.parametrize("affine", [True, False])
def test_layernorm(ops, affine):
    x = rnd(517, 3072, seed=3)
    w = rnd(3072, dtype=torch.float32, seed=4) if affine else None
    b = rnd(3072, dtype=torch.float32, seed=5) if affine else None
    y = ops.layernorm(x, w, b, eps=1e-6)
    ref = F.layer_norm(x.float(), (3072,), w, b, 1e-6).to(x.dtype)
    ulp_close(y, ref, x.dtype)


@pytest.mark.parametrize("gated", [True, False])
def test_gated_residual(ops, gated):
    x, y = rnd(300, 3072, seed=6), rnd(300, 3072, seed=7)
    table = rnd(2, 6, 3072, dtype=torch.float32, seed=8)
    sel = (torch.arange(300) % 2).to(torch.int32).to(DEV)
    if gated:
        out = ops.gated_residual(x, y, table[:, 2], sel)
        ref = (x.float() + y.float() * table[:, 2][sel.long()]).to(x.dtype)
    else:
        out = ops.gated_residual(x, y)
        ref = x + y
    assert torch.equal(out, ref)


@pytest.mark.parametrize("dim,head_dim,rope", [(3072, 128, True), (48, 24, True), (3072, 128, False)])
def test_rmsnorm_rope_inplace_on_fused_qkv(ops, dim, head_dim, rope):
    rows = 203
    qkv = rnd(rows, 3 * dim, seed=9)
    w = (1 + 0.1 * torch.randn(dim)).to(torch.bfloat16).to(DEV)
    cos = torch.rand(rows, head_dim // 2).to(DEV) if rope else None
    sin = torch.rand(rows, head_dim // 2).to(DEV) if rope else None
    ref_in = qkv[:, dim:2 * dim].clone()
    before = qkv.clone()
    ops.rmsnorm_rope_(qkv[:, dim:2 * dim], w, 1e-6, cos, sin, head_dim)
    # reference: diffusers RMSNorm semantics + transformer_wan.py:75-87
    var = ref_in.float().pow(2).mean(-1, keepdim=True)
    y = (ref_in * torch.rsqrt(var + 1e-6)).to(torch.bfloat16) * w
    if rope:
        yh = y.view(rows, dim // head_dim, head_dim // 2, 2)
        x1, x2 = yh[..., 0], yh[..., 1]
        c, s = cos[:, None, :], sin[:, None, :]
        out = torch.empty_like(yh)
        out[..., 0] = x1 * c - x2 * s
        out[..., 1] = x1 * s + x2 * c
        y = out.view(rows, dim)
    ulp_close(qkv[:, dim:2 * dim], y, torch.bfloat16)
    assert torch.equal(qkv[:, :dim], before[:, :dim]) and torch.equal(qkv[:, 2 * dim:], before[:, 2 * dim:])


def test_headnorm_rope_cog(ops):
    b, lt, lv, heads, hd = 2, 5, 37, 3, 64
    x = rnd(b, lt + lv, heads * hd, seed=10)
    w = (1 + 0.1 * torch.randn(hd)).to(torch.bfloat16).to(DEV)
    bb = (0.1 * torch.randn(hd)).to(torch.bfloat16).to(DEV)
    cos, sin = torch.rand(lv, hd).to(DEV), torch.rand(lv, hd).to(DEV)
    ref = x.view(b, lt + lv, heads, hd).transpose(1, 2)
    ref = F.layer_norm(ref, (hd,), w, bb, 1e-6)
    xr, xi = ref[:, :, lt:].reshape(b, heads, lv, hd // 2, 2).unbind(-1)
    rot = torch.stack([-xi, xr], dim=-1).flatten(3)
    ref[:, :, lt:] = (ref[:, :, lt:].float() * cos + rot.float() * sin).to(x.dtype)
    ref = ref.transpose(1, 2).reshape(b, lt + lv, heads * hd)
    ops.headnorm_rope_(x, heads, hd, w, bb, 1e-6, cos, sin, rope_row0=lt)
    ulp_close(x, ref, torch.bfloat16)


def sdpa_ref(q, k, v, heads):
    b, lq, hd = q.shape
    dh = hd // heads
    qh = q.float().view(b, lq, heads, dh).transpose(1, 2)
    kh = k.float().view(b, -1, heads, dh).transpose(1, 2)
    vh = v.float().view(b, -1, heads, dh).transpose(1, 2)
    o = F.scaled_dot_product_attention(qh, kh, vh)
    return o.transpose(1, 2).reshape(b, lq, hd)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("b,heads,dh,lq,lk", [
    (1, 2, 128, 1000, 1000),   # ragged q-block and kv tile
    (1, 3, 128, 300, 77),      # cross-attention like, Lk < one tile multiple
    (1, 24, 128, 513, 512),    # all heads -> exercises the XCD mapping
    (2, 5, 64, 700, 700),      # CogVideoX head size, batch 2, ragged
    (1, 1, 64, 31, 1),         # single key
    (1, 2, 128, 1, 64),        # one query, exactly one full key tile
    (1, 2, 128, 257, 65),      # two tiles, one key in the second; query block boundary + 1
    (2, 3, 64, 256, 128),      # exactly two tiles, exactly one query block
    (1, 9, 128, 40, 193),      # odd head count, three tiles + 1 key
])
def test_attention_vs_fp32_sdpa(ops, dtype, b, heads, dh, lq, lk):
    q = rnd(b, lq, heads * dh, dtype=dtype, seed=11)
    k = rnd(b, lk, heads * dh, dtype=dtype, seed=12)
    v = rnd(b, lk, heads * dh, dtype=dtype, seed=13)
    o = ops.attention(q, k, v, heads)
    ref = sdpa_ref(q, k, v, heads)
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    # P is rounded to the operand dtype before P.V (as every flash kernel does): tolerance 4 ulp rel-RMS
    assert rel_rms(o, ref) < 4 * eps, rel_rms(o, ref)
    assert (o.float() - ref).abs().max().item() < 0.05


@pytest.mark.parametrize("b,heads,dh,lq,lk", [
    (1, 8, 128, 2000, 2000),      # 8 blocks per XCD -> 4 key ranges each
    (1, 3, 128, 1100, 3000),      # heads not a multiple of 8: some XCDs own nothing; 5 blocks -> 6 ranges (ragged keys)
    (2, 12, 64, 777, 1500),       # head_dim 64, 3 heads-batches x 4 blocks = 12 per XCD -> 2 ranges
    (1, 24, 128, 3080, 3080),     # the 4-way token shard's query count (39 blocks per XCD: 32 whole + 7 split)
])
def test_attention_tail_split_equals_unsplit(ops, b, heads, dh, lq, lk):
    """fino_attn_fwd_ws (key-range partials + merge) against the one-pass kernel and fp32 SDPA."""
    from frameino_amd import _lib
    q = rnd(b, lq, heads * dh, seed=21)
    k = rnd(b, lk, heads * dh, seed=22)
    v = rnd(b, lk, heads * dh, seed=23)
    k[0, lk - 3] = q[0, 7] * 3.0                       # a late spike: the ranges end with different running maxima
    assert _lib.lib().fino_attn_workspace_bytes(b, heads, lq, lk, dh) > 0
    ops.SPLIT_ATTENTION_TAIL = False
    try:
        o1 = ops.attention(q, k, v, heads)
    finally:
        ops.SPLIT_ATTENTION_TAIL = True
    o2 = ops.attention(q, k, v, heads)
    ref = sdpa_ref(q, k, v, heads)
    assert rel_rms(o2, ref) < 2.0 ** -6
    assert rel_rms(o2, o1.float()) < 2.0 ** -8        # same P rounding; only the fp32 merge order differs
    # blocks that run whole are bit-identical
    full_rows = 256
    assert torch.equal(o1[:, :full_rows], o2[:, :full_rows]) or rel_rms(o2[:, :full_rows], o1[:, :full_rows].float()) < 2.0 ** -8


def test_attention_workspace_too_small_is_an_error(ops):
    from frameino_amd import _lib
    q = rnd(1, 2000, 8 * 128, seed=24)
    out = torch.empty_like(q)
    ws = torch.empty(64, dtype=torch.float32, device=DEV)
    rc = _lib.lib().fino_attn_fwd_ws(q.data_ptr(), q.data_ptr(), q.data_ptr(), out.data_ptr(), 1, 8, 2000, 2000, 128,
                                    q.stride(0), q.stride(1), 128, q.stride(0), q.stride(1), 128, q.stride(0),
                                    q.stride(1), 128, out.stride(0), out.stride(1), 128, 0.088, 0, ws.data_ptr(), 256,
                                    0)
    assert rc != 0 and b"workspace" in _lib.lib().fino_last_error()


def test_attention_reads_fused_qkv_in_place(ops):
    L, heads, dh = 450, 4, 128
    d = heads * dh
    qkv = rnd(1, L, 3 * d, seed=14)
    o = ops.attention(qkv[:, :, :d], qkv[:, :, d:2 * d], qkv[:, :, 2 * d:], heads)
    ref = sdpa_ref(qkv[:, :, :d].contiguous(), qkv[:, :, d:2 * d].contiguous(), qkv[:, :, 2 * d:].contiguous(), heads)
    assert rel_rms(o, ref) < 2.0 ** -6


def test_attention_large_scores_online_softmax(ops):
    """Forces the running max to jump late in the sequence (rescale branch) and -inf-free masking of the tail."""
    L, heads, dh = 333, 1, 128
    q = rnd(1, L, dh, seed=15)
    k = rnd(1, L, dh, seed=16)
    v = rnd(1, L, dh, seed=17)
    k[0, 300] = q[0, 5] * 4.0          # spike for query 5 at key 300
    o = ops.attention(q, k, v, heads)
    ref = sdpa_ref(q, k, v, heads)
    assert rel_rms(o, ref) < 2.0 ** -6
    assert torch.isfinite(o.float()).all()


def gemm_ref(a, w, bias, epi, residual=None, gate=None, sel=None):
    y = a.float() @ w.float().t()
    if bias is not None:
        y = y + bias.float()
    y = y.to(a.dtype)
    if epi == 1:
        y = F.gelu(y.float(), approximate="tanh").to(a.dtype)
    elif epi == 2:
        y = residual + y
    elif epi == 3:
        g = gate[sel.long()] if sel is not None else gate
        y = (residual.float() + y.float() * g).to(a.dtype)
    elif epi == 4:                                   # staged: the gated product is rounded to T first (CogVideoX)
        g = gate[sel.long()] if sel is not None else gate
        y = (residual.float() + (y.float() * g).to(a.dtype).float()).to(a.dtype)
    return y


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,n,k", [
    (300, 512, 256),     # fast path, ragged M
    (1000, 3072, 3072),  # model shape (reduced M)
    (257, 192, 384),     # N < tile (proj_out / patch-embed shapes)
    (72, 48, 48),        # generic path: K not a multiple of 64 (tiny test models)
    (5, 3072, 64),       # tiny M
    (256, 264, 128),     # exactly one row tile, ragged N (multiple of 8), two K-tiles
    (513, 256, 192),     # three K-tiles (odd count through the ping-pong phases), M = 2 tiles + 1 row
])
@pytest.mark.parametrize("epi", [0, 1, 2, 3])
def test_gemm_epilogues(ops, dtype, m, n, k, epi):
    a = rnd(m, k, dtype=dtype, seed=20)
    w = rnd(n, k, dtype=dtype, seed=21, scale=1.0 / math.sqrt(k))
    bias = rnd(n, dtype=dtype, seed=22)
    residual = rnd(m, n, dtype=dtype, seed=23) if epi >= 2 else None
    table = rnd(2, 6, n, dtype=torch.float32, seed=24) if epi == 3 else None
    sel = (torch.arange(m) % 2).to(torch.int32).to(DEV) if epi == 3 else None
    gate = table[:, 5] if epi == 3 else None
    out = ops.gemm(a, w, bias, epi, residual, gate, sel)
    ref = gemm_ref(a, w, bias, epi, residual, gate, sel)
    ulp_close(out, ref, dtype, max_ulp_frac=0.02)


def test_gemm_strided_views_and_in_place_residual(ops):
    m, d = 260, 512
    big = rnd(m, 3 * d, seed=25)
    w = rnd(d, d, seed=26, scale=0.05)
    x = rnd(m, d, seed=27)
    ref = gemm_ref(big[:, d:2 * d], w, None, 2, x.clone())
    ops.gemm(big[:, d:2 * d], w, None, 2, residual=x, out=x)      # C aliases R, A is a column slice
    ulp_close(x, ref, torch.bfloat16, max_ulp_frac=0.02)


def test_gemm_exact_small_integers(ops):
    """Layout check with exactly representable data: any fragment/swizzle mix-up gives O(1) errors."""
    m, n, k = 256, 256, 128
    g = torch.Generator().manual_seed(3)
    a = torch.randint(-3, 4, (m, k), generator=g).to(torch.bfloat16).to(DEV)
    w = torch.randint(-3, 4, (n, k), generator=g).to(torch.bfloat16).to(DEV)
    out = ops.gemm(a, w)
    assert torch.equal(out.float(), (a.float() @ w.float().t()).to(torch.bfloat16).float())


def test_skinny_linear(ops):
    x = rnd(2, 256, dtype=torch.float32, seed=30)
    w = rnd(3072, 256, dtype=torch.float32, seed=31, scale=0.05)
    b = rnd(3072, dtype=torch.float32, seed=32)
    torch.testing.assert_close(ops.skinny_linear(x, w, b), F.linear(x, w, b), atol=1e-4, rtol=1e-4)
    wb = w.to(torch.bfloat16)
    bb = b.to(torch.bfloat16)
    ref = F.linear(F.silu(x), wb.float(), bb.float())
    torch.testing.assert_close(ops.skinny_linear(x, wb, bb, silu_input=True), ref, atol=1e-4, rtol=1e-4)


def test_patchify_gemm_equals_conv3d_and_unpatchify(ops):
    c, f, h, w_, d, cout = 8, 3, 8, 12, 48, 4
    x = rnd(c, f, h, w_, seed=40)
    wt = rnd(d, c, 1, 2, 2, seed=41, scale=0.2)
    a = ops.patchify(x, (1, 2, 2))
    ref_a = x.view(c, f, 1, h // 2, 2, w_ // 2, 2).permute(1, 3, 5, 0, 2, 4, 6).reshape(f * (h // 2) * (w_ // 2), c * 4)
    assert torch.equal(a, ref_a)
    y = ops.gemm(a, wt.view(d, -1))
    conv = F.conv3d(x[None].float(), wt.float(), stride=(1, 2, 2)).flatten(2).transpose(1, 2)[0]
    assert rel_rms(y, conv) < 2.0 ** -7
    # unpatchify (transformer_wan.py:539-543)
    t = rnd(f * (h // 2) * (w_ // 2), 4 * cout, seed=42)
    out = ops.unpatchify(t, cout, f, h, w_, (1, 2, 2))
    ref = t.reshape(1, f, h // 2, w_ // 2, 1, 2, 2, cout).permute(0, 7, 1, 4, 2, 5, 3, 6)
    ref = ref.flatten(6, 7).flatten(4, 5).flatten(2, 3)[0]
    assert torch.equal(out, ref)


def test_sampler_glue_matches_reference_arithmetic(ops):
    c, fg, nid, h, w_ = 4, 3, 1, 4, 6
    lat = rnd(c, fg, h, w_, dtype=torch.float32, seed=50)
    cond = rnd(c, 1, h, w_, dtype=torch.float32, seed=51)
    idl = rnd(c, nid, h, w_, dtype=torch.float32, seed=52)
    traj = rnd(c, fg + nid, h, w_, dtype=torch.float32, seed=53)
    x = ops.wan_model_input(lat, cond, idl, traj, torch.bfloat16)
    mask = torch.ones(1, fg, 1, 1, device=DEV)
    mask[:, 0] = 0
    blend = ((1 - mask) * cond + mask * lat).to(torch.bfloat16)
    ref = torch.cat([torch.cat([blend, idl.to(torch.bfloat16)], 1), traj.to(torch.bfloat16)], 0)
    assert torch.equal(x, ref)

    pc, pu = rnd(c, fg + nid, h, w_, seed=54), rnd(c, fg + nid, h, w_, seed=55)
    dt = torch.tensor([-0.037], device=DEV)
    lat2 = lat.clone()
    ops.cfg_euler_step_(lat2, pc, pu, 5.0, dt, round_out=True)
    noise = (pu + 5.0 * (pc - pu))[:, :fg]
    ref = (lat + dt * noise).to(torch.bfloat16).float()
    assert torch.equal(lat2, ref)
    lat3 = lat.clone()
    ops.cfg_euler_step_(lat3, pc, None, 1.0, dt, round_out=False)
    assert torch.equal(lat3, lat + dt * pc[:, :fg])


def test_layernorm_zero_staged_rounding(ops):
    """CogVideoXLayerNormZero executed in bf16: every tensor op rounds (cogvideox_transformer_3d.py:134-136)."""
    b, L, d = 2, 37, 3072
    x = rnd(b * L, d, seed=60)
    # affine parameters are T tensors in the reference: bf16-representable values, passed to the kernel as fp32 copies
    w, bb = (rnd(d, seed=61).float() * 0.1 + 1).bfloat16().float(), (rnd(d, seed=62).float() * 0.1).bfloat16().float()
    tab = rnd(2 * b, 3, d, seed=63, scale=0.3).float()            # bf16-representable values
    sel = (torch.arange(b * L, device=DEV) % 4).to(torch.int32)
    y = ops.layernorm_zero(x, w, bb, tab[:, 0], tab[:, 1], sel, 1e-5)
    n = F.layer_norm(x, (d,), w.bfloat16(), bb.bfloat16(), 1e-5)
    sc, sh = (1 + tab[:, 1].bfloat16())[sel.long()], tab[:, 0].bfloat16()[sel.long()]
    ref = n * sc + sh
    # a 1-ulp flip of the normalised value (reduction order) propagates through the product; with cancellation in
    # "+ shift" the bound is 2 ulp of the LARGER intermediate, not of the result
    mag = (n.float() * sc.float()).abs() + sh.float().abs()
    bad = (y.float() - ref.float()).abs() > 2 * 2.0 ** -8 * mag + 1e-6
    assert bad.float().mean().item() < 0.01
    assert rel_rms(y, ref) < 2.0 ** -7


def test_gated_residual_staged_and_gemm_epilogue(ops):
    m, n, k = 300, 512, 256
    x, y = rnd(m, n, seed=64), rnd(m, n, seed=65)
    gate = rnd(4, 3, n, seed=66).float()
    sel = (torch.arange(m, device=DEV) % 4).to(torch.int32)
    out = ops.gated_residual(x, y, gate[:, 2], sel, staged=True)
    ref = x + gate[:, 2].bfloat16()[sel.long()] * y
    assert torch.equal(out, ref)
    a, w = rnd(m, k, seed=67), rnd(n, k, seed=68, scale=0.06)
    bias = rnd(n, seed=69)
    out = ops.gemm(a, w, bias, ops.EPI_GATED_RESIDUAL_STAGED, residual=x, gate=gate[:, 2], sel=sel)
    lin = (a.float() @ w.float().t() + bias.float()).bfloat16()
    ref = x + gate[:, 2].bfloat16()[sel.long()] * lin
    ulp_close(out, ref, torch.bfloat16, max_ulp_frac=0.02)


def test_cfg_vpred_step(ops):
    f, ft, c, h, w_ = 3, 4, 2, 4, 6
    lat = rnd(f, c, h, w_, seed=70)
    pred = rnd(2, ft, c, h, w_, seed=71)
    coef = torch.tensor([0.83, 0.55, 0.91, 0.12, 6.0], device=DEV)
    lat2 = lat.clone()
    ops.cfg_vpred_step_(lat2, pred, coef, True)
    p = pred.float()[:, :f]
    v = p[0] + 6.0 * (p[1] - p[0])
    x0 = (coef[0] * lat.float()).bfloat16().float() - coef[1] * v
    ref = ((coef[2] * lat.float()).bfloat16().float() + coef[3] * x0).bfloat16()
    assert torch.equal(lat2, ref)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("has_uncond", [True, False])
def test_cfg_unipc_step(dtype, has_uncond):
    """fino_cfg_unipc_step: the coefficient-folded corrector+predictor on a [C, Fg(+ID), H, W] prediction."""
    from frameino_amd import ops
    g = torch.Generator().manual_seed(5)
    c, fg, ft, h, w = 4, 3, 4, 6, 10
    pc = torch.randn(c, ft, h, w, generator=g).to(dtype)
    pu = torch.randn(c, ft, h, w, generator=g).to(dtype) if has_uncond else None
    x, last, m0, m1 = (torch.randn(c, fg, h, w, generator=g) for _ in range(4))
    for use_corr in (0.0, 1.0):
        coef = torch.tensor([5.0, 0.83, use_corr, 0.9, 0.31, -0.07, 0.12, 0.8, 0.25, -0.05])
        v = pc[:, :fg]
        if has_uncond:
            u = pu[:, :fg]
            v = u + 5.0 * (v - u)                               # every op rounds to the model dtype (:882)
        # `sigma_t * model_output` as the sampler evaluates it: a CPU 0-dim fp32 sigma times a device T tensor (the
        # exact product rounded once to T -- a CPU emulation through fp32 double-rounds ties differently)
        mt = x - (coef[1] * v.to(DEV)).float().cpu()
        xc = coef[3] * last + coef[4] * m0 + coef[5] * m1 + coef[6] * mt if use_corr else x
        xn = coef[7] * xc + coef[8] * mt + coef[9] * m0
        bx, bl, b0, b1 = (t.clone().to(DEV) for t in (x, last, m0, m1))
        ops.cfg_unipc_step_(bx, bl, b0, b1, pc.to(DEV), None if pu is None else pu.to(DEV), coef.to(DEV))
        torch.testing.assert_close(b0.cpu(), mt, atol=1e-6, rtol=1e-6)
        torch.testing.assert_close(b1.cpu(), m0, atol=0, rtol=0)
        torch.testing.assert_close(bl.cpu(), xc, atol=2e-6, rtol=2e-6)
        torch.testing.assert_close(bx.cpu(), xn, atol=2e-6, rtol=2e-6)


@pytest.mark.parametrize("heads,dh,lq,splits", [(2, 128, 300, (0, 64, 500)), (3, 64, 77, (0, 200, 201, 650)),
                                                (24, 128, 3080, (0, 3080, 6160))])
def test_attention_partials_over_key_ranges_merge_to_the_full_attention(heads, dh, lq, splits):
    """fino_attn_partial + fino_attn_merge: attention over 1-3 disjoint key ranges merged through (O, m, l) equals the
    single pass over all keys up to fp32 summation order (what the token-sharded forward does with its own K/V chunk
    and the gathered ones)."""
    from frameino_amd import ops
    g = torch.Generator(device=DEV).manual_seed(17)
    d = heads * dh
    lk = splits[-1]
    q = torch.randn(2, lq, d, device=DEV, generator=g).bfloat16()
    kv = (torch.randn(2, lk, 2 * d, device=DEV, generator=g) * 1.5).bfloat16()
    k, v = kv[:, :, :d], kv[:, :, d:]
    full = ops.attention(q, k, v, heads)
    parts = [ops.attention_partial(q, k[:, a:b], v[:, a:b], heads) for a, b in zip(splits[:-1], splits[1:])]
    merged = ops.attention_merge(parts, 2, lq, heads, dh, q.dtype)
    assert torch.isfinite(merged.float()).all()
    assert rel_rms(merged, full.float()) < 2.0 ** -8, rel_rms(merged, full.float())
    one = ops.attention_merge([ops.attention_partial(q, k, v, heads)], 2, lq, heads, dh, q.dtype)
    assert rel_rms(one, full.float()) < 2.0 ** -9


def _sdpa_f32(q, k, v, heads, scale=None):
    b, lq, d = q.shape
    dh = d // heads
    f = lambda t: t.view(b, -1, heads, dh).transpose(1, 2).float()      # noqa: E731
    return torch.nn.functional.scaled_dot_product_attention(f(q), f(k), f(v), scale=scale).transpose(1, 2).reshape(b, lq, d)


@pytest.mark.parametrize("b,heads,lq,lk", [(1, 2, 64, 64), (2, 3, 257, 129), (1, 2, 300, 500), (1, 8, 1000, 77),
                                           (2, 24, 1024, 1024), (1, 24, 2000, 4500)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("dh", [128, 64])
def test_four_wave_attention_kernel_and_folded_scale(b, heads, lq, lk, dtype, dh):
    """FINO_TUNE_ATTN_KERNEL = 2 (fino_attention_w4.hip): whole blocks, ragged last tile, tail split, and the
    FINO_ATTN_SCALE_FOLDED path (q pre-multiplied by scale * log2(e), running maximum folded into the MFMAs), each against
    fp32 SDPA on the q it was given (tolerance 2^-8 relative RMS); the folded q is ONE rounding of the fp32 q like the plain
    one, so both are equally far from SDPA on the un-rounded q.  head_dim 64 has the 4-wave kernel for the folded path only
    (the plain call falls back to the 8-wave kernel)."""
    from frameino_amd import _lib, ops
    g = torch.Generator(device=DEV).manual_seed(3)
    d = heads * dh
    qf = torch.randn(b, lq, d, device=DEV, generator=g) * 2.0
    kv = torch.randn(b, lk, 2 * d, device=DEV, generator=g).to(dtype)
    k, v = kv[:, :, :d], kv[:, :, d:]
    c = dh ** -0.5 * ops.LOG2E
    q, qs = qf.to(dtype), (qf * c).to(dtype)                 # as rmsnorm_rope_ / headnorm_rope_ write q without / with out_scale
    ref, ref_s = _sdpa_f32(q, k, v, heads), _sdpa_f32(qs.float() / c, k, v, heads)
    lib = _lib.lib()
    try:
        lib.fino_tune_set(4, 2)
        plain = ops.attention(q, k, v, heads)
        folded = ops.attention(qs, k, v, heads, scale=ops.SCALE_FOLDED)
        lib.fino_tune_set(4, 1)
        folded8 = ops.attention(qs, k, v, heads, scale=ops.SCALE_FOLDED)
    finally:
        lib.fino_tune_set(4, 0)
    for name, out, want in (("4-wave", plain, ref), ("4-wave folded", folded, ref_s), ("8-wave folded", folded8, ref_s)):
        assert torch.isfinite(out.float()).all(), name
        assert rel_rms(out, want) < 2.0 ** -8, (name, rel_rms(out, want))
    # and both roundings of q are equally far from the un-rounded one
    truth = _sdpa_f32(qf, k, v, heads)
    assert rel_rms(folded, truth) < 1.25 * rel_rms(plain, truth) + 1e-4


def test_four_wave_partials_merge_and_peaky_rows():
    """the 4-wave kernel's (O, m, l) partials are in the 8-wave layout: merged key ranges == one pass; and rows whose
    maximum keeps growing by far more than the deferred-rescale threshold (keys sorted by logit) stay exact"""
    from frameino_amd import _lib, ops
    g = torch.Generator(device=DEV).manual_seed(5)
    heads, lq, lk = 3, 300, 1500
    d = heads * 128
    q = (torch.randn(1, lq, d, device=DEV, generator=g) * 6.0).bfloat16()
    kv = torch.randn(1, lk, 2 * d, device=DEV, generator=g).bfloat16()
    # order the keys by their logit against query 0 of head 0: the running maximum of that row rises all the way
    order = (kv[0, :, :128].float() @ q[0, 0, :128].float()).argsort()
    kv = kv[:, order].contiguous()
    k, v = kv[:, :, :d], kv[:, :, d:]
    ref = _sdpa_f32(q, k, v, heads)
    lib = _lib.lib()
    try:
        lib.fino_tune_set(4, 2)
        one = ops.attention(q, k, v, heads)
        qs = (q.float() * (128 ** -0.5 * ops.LOG2E)).bfloat16()
        fold = ops.attention(qs, k, v, heads, scale=ops.SCALE_FOLDED)
        parts = [ops.attention_partial(q, k[:, a:c], v[:, a:c], heads) for a, c in ((0, 700), (700, 701), (701, lk))]
        merged = ops.attention_merge(parts, 1, lq, heads, 128, q.dtype)
    finally:
        lib.fino_tune_set(4, 0)
    assert rel_rms(one, ref) < 2.0 ** -8 and rel_rms(merged, ref) < 2.0 ** -8, (rel_rms(one, ref), rel_rms(merged, ref))
    assert rel_rms(fold, _sdpa_f32(qs.float() / (128 ** -0.5 * ops.LOG2E), k, v, heads)) < 2.0 ** -8


def test_rmsnorm_rope_out_scale_is_one_rounding():
    from frameino_amd import ops
    g = torch.Generator(device=DEV).manual_seed(9)
    rows, heads, dh = 333, 4, 128
    d = heads * dh
    x = torch.randn(rows, d, device=DEV, generator=g).bfloat16()
    w = (1 + 0.1 * torch.randn(d, device=DEV, generator=g)).bfloat16()
    ang = torch.rand(rows, dh // 2, device=DEV, generator=g) * 6.28
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
    c = dh ** -0.5 * ops.LOG2E
    plain = ops.rmsnorm_rope_(x.clone(), w, 1e-6, cos, sin, dh)
    scaled = ops.rmsnorm_rope_(x.clone(), w, 1e-6, cos, sin, dh, out_scale=c)
    # fp32 restatement of the kernel's arithmetic (same rounding points up to the RoPE, then x c, one rounding)
    xf = x.float()
    y = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6)).bfloat16().float() * w.float()
    y = y.bfloat16().float().view(rows, heads, dh // 2, 2)
    o = torch.stack((y[..., 0] * cos[:, None] - y[..., 1] * sin[:, None], y[..., 0] * sin[:, None] + y[..., 1] * cos[:, None]), -1)
    want = (o.reshape(rows, d) * c).bfloat16()
    assert (scaled.float() - want.float()).abs().max() <= 2.0 ** -8 * want.float().abs().max()
    assert (scaled != want).float().mean() < 0.02                     # a few 1-ulp flips from rsqrt / fma contraction
    assert rel_rms(scaled.float() / c, plain.float()) < 2.0 ** -8     # and it is the plain result, scaled


def test_headnorm_rope_out_scale():
    """fino_headnorm_rope_scaled: rows that get RoPE are (norm + rope) x c rounded once; the others the rounded norm x c"""
    from frameino_amd import ops
    g = torch.Generator(device=DEV).manual_seed(10)
    b, rows, heads, dh, lt = 2, 50, 3, 64, 7
    x = torch.randn(b, rows, heads * dh, device=DEV, generator=g).bfloat16()
    w = (1 + 0.1 * torch.randn(dh, device=DEV, generator=g)).bfloat16()
    bi = (0.1 * torch.randn(dh, device=DEV, generator=g)).bfloat16()
    ang = torch.rand(rows - lt, dh, device=DEV, generator=g) * 6.28
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
    c = dh ** -0.5 * ops.LOG2E
    plain = ops.headnorm_rope_(x.clone(), heads, dh, w, bi, 1e-6, cos, sin, rope_row0=lt)
    scaled = ops.headnorm_rope_(x.clone(), heads, dh, w, bi, 1e-6, cos, sin, rope_row0=lt, out_scale=c)
    assert rel_rms(scaled.float() / c, plain.float()) < 2.0 ** -8
    # text rows: exactly the rounded plain result times c, rounded again
    assert torch.equal(scaled[:, :lt], (plain[:, :lt].float() * c).bfloat16())


def test_diag_mfma_peak_runs_and_counts_flops():
    """fino_diag_mfma_peak (tools/mfma_peak.py): launches, reports the FLOPs it launched, and sustains a plausible rate"""
    import ctypes
    from frameino_amd import _lib
    lib = _lib.lib()
    scratch = torch.zeros(64 + 2 * 256 * 4, device=DEV)
    fl = ctypes.c_double()
    for kind in (0, 1):
        _lib.check(lib.fino_diag_mfma_peak(kind, 2, 200, scratch.data_ptr(), ctypes.byref(fl), None), "diag")
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        _lib.check(lib.fino_diag_mfma_peak(kind, 2, 20000, scratch.data_ptr(), ctypes.byref(fl), None), "diag")
        e.record()
        torch.cuda.synchronize()
        cus = torch.cuda.get_device_properties(0).multi_processor_count
        assert fl.value == cus * 2 * 4.0 * 20000 * 524288.0
        tf = fl.value / (s.elapsed_time(e) * 1e-3) / 1e12
        assert 500.0 < tf < 2600.0, tf
    assert lib.fino_diag_mfma_peak(7, 2, 10, scratch.data_ptr(), ctypes.byref(fl), None) != 0      # bad kind -> error code


@pytest.mark.parametrize("ways,groups,n,lpad", [(4, 1, 3080, 3080), (4, 2, 1000, 1024), (8, 2, 1540, 1540), (2, 1, 777, 784)])
def test_rmsnorm_rope_scatter_equals_inplace_kernel_plus_permute_copy(ways, groups, n, lpad):
    """fino_rmsnorm_rope_scatter: q, k (RMSNorm + RoPE) and v (plain copy) of a token shard written straight into the heads
    exchange's send buffers == the in-place kernel followed by the permute copy it replaces, bit for bit"""
    from frameino_amd import ops
    from frameino_amd.parallel import TokenShard
    heads, dh = 24, 128
    d = heads * dh
    g = torch.Generator(device=DEV).manual_seed(n)
    qkv = torch.randn(n, 3 * d, device=DEV, generator=g).bfloat16()
    wq = (1 + 0.1 * torch.randn(d, device=DEV, generator=g)).bfloat16()
    wk = (1 + 0.1 * torch.randn(d, device=DEV, generator=g)).bfloat16()
    ang = torch.rand(n, dh // 2, device=DEV, generator=g) * 6.28
    cos, sin = torch.cos(ang).contiguous(), torch.sin(ang).contiguous()
    sh = TokenShard.__new__(TokenShard)
    sh.ways, sh._buf, sh.head_groups = ways, {}, groups
    lay = sh.heads_send_layout(heads, dh, lpad, torch.bfloat16, torch.device(DEV))
    c = dh ** -0.5 * ops.LOG2E
    ops.rmsnorm_rope_scatter(qkv[:, :d], wq, 1e-6, cos, sin, dh, lay.flat, lay.off[0], lay.ld, out_scale=c)
    ops.rmsnorm_rope_scatter(qkv[:, d:2 * d], wk, 1e-6, cos, sin, dh, lay.flat, lay.off[1], lay.ld)
    ops.rmsnorm_rope_scatter(qkv[:, 2 * d:], None, 0.0, None, None, dh, lay.flat, lay.off[2], lay.ld)
    ref = qkv.clone()
    ops.rmsnorm_rope_(ref[:, :d], wq, 1e-6, cos, sin, dh, out_scale=c)
    ops.rmsnorm_rope_(ref[:, d:2 * d], wk, 1e-6, cos, sin, dh)
    hp = heads // ways
    q4 = ref.view(n, 3, ways, hp * dh)
    assert len(lay.views) == len(sh.head_ranges(hp))
    for view, (h0, h1) in zip(lay.views, sh.head_ranges(hp)):
        want = q4[:, :, :, h0 * dh:h1 * dh].permute(2, 0, 1, 3)
        assert torch.equal(view[:, :n], want)
        assert not view[:, n:].any()
    assert torch.equal(qkv[:, 2 * d:], ref[:, 2 * d:])          # the source is left alone
    # the one-launch forms: scattered (a fresh layout) and in place
    sh2 = TokenShard.__new__(TokenShard)
    sh2.ways, sh2._buf, sh2.head_groups = ways, {}, groups
    lay2 = sh2.heads_send_layout(heads, dh, lpad, torch.bfloat16, torch.device(DEV))
    before = qkv.clone()
    ops.qkv_rmsnorm_rope_(qkv, d, wq, 1e-6, wk, 1e-6, cos, sin, dh, q_out_scale=c, out=lay2.flat, head_off=lay2.off_qkv,
                          head_ld=lay2.ld)
    assert torch.equal(lay2.flat, lay.flat) and torch.equal(qkv, before)
    ops.qkv_rmsnorm_rope_(qkv, d, wq, 1e-6, wk, 1e-6, cos, sin, dh, q_out_scale=c)
    assert torch.equal(qkv, ref)


@pytest.mark.parametrize("b,heads,lq,lk,dh", [(2, 24, 1000, 512, 128), (1, 3, 300, 77, 128), (2, 2, 129, 1024, 128),
                                              (1, 24, 3080, 12320, 128)])      # (head_dim 128 only since round 5)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_free_running_attention_kernel_equals_the_ping_pong_kernel(b, heads, lq, lk, dh, dtype):
    """attn_fr_kernel (4 waves, two workgroups per CU, LDS-DMA staging; FINO_TUNE_ATTN_KERNEL = 3, the default for the text
    cross-attention at head_dim 128): the same arithmetic in the same order as the 8-wave ping-pong kernel -- bit-identical
    outputs on ragged shapes, strided q / k / v, rows past Lq untouched"""
    from frameino_amd import _lib, ops
    d = heads * dh
    g = torch.Generator(device=DEV).manual_seed(lq * 7 + lk)
    q = torch.randn(b, lq, d + 64, device=DEV, generator=g).to(dtype)[:, :, :d]
    kv = torch.randn(b, lk, 2 * d + 64, device=DEV, generator=g).to(dtype)
    k, v = kv[:, :, :d], kv[:, :, d:2 * d]
    lib = _lib.lib()
    split = ops.SPLIT_ATTENTION_TAIL
    try:
        lib.fino_tune_set(4, 1)
        ops.SPLIT_ATTENTION_TAIL = False       # whole blocks, as the free-running kernel runs them (the tail split merges
        want = ops.attention(q, k, v, heads)   # key-range partials: the same numbers up to one rounding of the merge)
        ops.SPLIT_ATTENTION_TAIL = split
        lib.fino_tune_set(4, 3)
        out = torch.zeros(b, lq + 5, d, device=DEV, dtype=dtype)
        got = ops.attention(q, k, v, heads, out=out[:, :lq])
    finally:
        lib.fino_tune_set(4, 0)
        ops.SPLIT_ATTENTION_TAIL = split
    assert torch.isfinite(got.float()).all() and not out[:, lq:].any()
    assert torch.equal(got, want), (got.float() - want.float()).abs().max().item()


@pytest.mark.parametrize("b,heads,lq,lk", [(2, 24, 1000, 512), (1, 3, 300, 77), (2, 2, 129, 1024), (1, 8, 700, 200),
                                           (1, 5, 64, 64), (1, 24, 3080, 12320), (2, 24, 2100, 4097)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("dh", [128, 64])
def test_dma_staged_ping_pong_kernel_equals_the_register_staged_one(b, heads, lq, lk, dtype, dh):
    """attn_ppd_kernel (FINO_TUNE_ATTN_KERNEL = 4: K / V by LDS-DMA into rings of four slots, issued 2 - 4 tiles ahead):
    the arithmetic of attn_pp_kernel in the same order -- bit-identical outputs with and without the tail split, as key-range
    partials, on ragged shapes and strided q / k / v; rows past Lq untouched"""
    from frameino_amd import _lib, ops
    d = heads * dh
    g = torch.Generator(device=DEV).manual_seed(lq * 7 + lk)
    q = torch.randn(b, lq, d + 64, device=DEV, generator=g).to(dtype)[:, :, :d]
    kv = torch.randn(b, lk, 2 * d + 64, device=DEV, generator=g).to(dtype)
    k, v = kv[:, :, :d], kv[:, :, d:2 * d]
    lib = _lib.lib()
    split = ops.SPLIT_ATTENTION_TAIL
    try:
        for sp in (False, True):
            ops.SPLIT_ATTENTION_TAIL = sp
            lib.fino_tune_set(4, 1)
            want = ops.attention(q, k, v, heads)
            wantp = ops.attention_partial(q, k, v, heads).clone()
            lib.fino_tune_set(4, 4)
            out = torch.zeros(b, lq + 5, d, device=DEV, dtype=dtype)
            got = ops.attention(q, k, v, heads, out=out[:, :lq])
            gotp = ops.attention_partial(q, k, v, heads)
            assert torch.isfinite(got.float()).all() and not out[:, lq:].any()
            assert torch.equal(got, want), (sp, (got.float() - want.float()).abs().max().item())
            assert torch.equal(gotp, wantp), sp
    finally:
        lib.fino_tune_set(4, 0)
        ops.SPLIT_ATTENTION_TAIL = split


@pytest.mark.parametrize("b,heads,lq,lk,grid", [(2, 24, 3000, 512, 0), (1, 3, 300, 77, 2), (2, 2, 129, 1024, 1), (1, 8, 700, 200, 3),
                                                (2, 5, 520, 128, 7), (1, 24, 1540, 512, 5), (2, 3, 1000, 640, 4), (1, 2, 64, 130, 1)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_walking_attention_kernel_equals_the_ping_pong_kernel(b, heads, lq, lk, grid, dtype):
    """attn_ppw_kernel (FINO_TUNE_ATTN_KERNEL = 6: one workgroup walks a run of q-blocks, K / V rings streaming across block
    boundaries, the next block's Q rows prefetched, stores counted into the vmcnt bookkeeping): bit-identical to the one-block
    ping-pong kernel -- with runs that cross heads and batches (FINO_TUNE_ATTN_WALK_GRID caps the workgroup count), ragged
    last key tiles, strided q / k / v, rows past Lq untouched"""
    from frameino_amd import _lib, ops
    dh = 128
    d = heads * dh
    g = torch.Generator(device=DEV).manual_seed(lq * 7 + lk)
    q = torch.randn(b, lq, d + 64, device=DEV, generator=g).to(dtype)[:, :, :d]
    kv = torch.randn(b, lk, 2 * d + 64, device=DEV, generator=g).to(dtype)
    k, v = kv[:, :, :d], kv[:, :, d:2 * d]
    lib = _lib.lib()
    split = ops.SPLIT_ATTENTION_TAIL
    try:
        ops.SPLIT_ATTENTION_TAIL = False
        lib.fino_tune_set(4, 1)
        want = ops.attention(q, k, v, heads)
        lib.fino_tune_set(4, 6)
        lib.fino_tune_set(6, grid)
        for _ in range(2):
            out = torch.zeros(b, lq + 5, d, device=DEV, dtype=dtype)
            got = ops.attention(q, k, v, heads, out=out[:, :lq])
            assert torch.isfinite(got.float()).all() and not out[:, lq:].any()
            assert torch.equal(got, want), (got.float() - want.float()).abs().max().item()
    finally:
        lib.fino_tune_set(4, 0)
        lib.fino_tune_set(6, 0)
        ops.SPLIT_ATTENTION_TAIL = split
