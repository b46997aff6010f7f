"""MXFP8 linear path (BASELINE config 5): quantiser against a torch restatement of the MX rule, the scaled-MFMA GEMM
against fp32 matmul on the DEQUANTISED operands (exact up to fp32 summation order), and the end-to-end error of
quantise + GEMM against the bf16 GEMM it replaces."""
import pytest
import torch

from tests.parity import rel_rms

pytestmark = pytest.mark.gpu
DEV = "cuda"


def dequant(q, scales, rows, cols):
    """inverse of the layout fino_quantize_mxfp8 writes: -> fp32 [rows, cols]"""
    rp = (rows + 255) // 256 * 256
    s = scales.view(cols // 128, rp // 256, 4, 16, 16).cpu()          # [kt][rt][g][row&15][row>>4]
    e = s.permute(1, 4, 3, 0, 2).reshape(rp, cols // 32)[:rows]       # [row][kt*4+g]  (row = (row>>4)*16 + (row&15))
    val = q.cpu().view(torch.float8_e4m3fn).float()
    return val * torch.exp2(e.float() - 127).repeat_interleave(32, dim=1)


@pytest.mark.parametrize("rows,cols", [(300, 256), (77, 1024), (512, 3072)])
def test_quantize_matches_mx_rule(rows, cols):
    from frameino_amd import ops
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(rows, cols, generator=g) * torch.exp2(torch.randint(-6, 6, (rows, 1), generator=g).float())).bfloat16()
    x[0, :32] = 0                                                     # an all-zero block
    q, s = ops.quantize_mxfp8(x.to(DEV))
    xd = dequant(q, s, rows, cols)
    xf = x.float()
    blk = xf.view(rows, cols // 32, 32)
    amax = blk.abs().amax(-1, keepdim=True)
    # every block's amax is scaled into (224, 448]: nothing saturates; e4m3's half-ulp there is 16, so the error
    # of any element is at most amax * 16 / 224
    err = (xd.view(rows, cols // 32, 32) - blk).abs()
    assert (err <= amax / 14 + 1e-30).all()
    assert (xd[0, :32] == 0).all()
    assert rel_rms(xd, xf) < 0.04


@pytest.mark.parametrize("m,n,k,epi", [(256, 256, 128, 0), (300, 520, 384, 0), (1000, 768, 1024, 1), (513, 256, 2048, 3),
                                       (2048, 3072, 3072, 2)])
def test_gemm_mxfp8_vs_dequantised_fp32(m, n, k, epi):
    from frameino_amd import ops
    from tests.test_kernels_gpu import gemm_ref
    g = torch.Generator().manual_seed(2)
    a = torch.randn(m, k, generator=g).bfloat16().to(DEV)
    w = (torch.randn(n, k, generator=g) * 0.05).bfloat16().to(DEV)
    bias = torch.randn(n, generator=g).bfloat16().to(DEV)
    res = torch.randn(m, n, generator=g).bfloat16().to(DEV) if epi >= 2 else None
    gate = torch.randn(2, n, generator=g).to(DEV) if epi == 3 else None
    sel = (torch.arange(m) % 2).to(torch.int32).to(DEV) if epi == 3 else None
    aq, sa = ops.quantize_mxfp8(a)
    wq, sw = ops.quantize_mxfp8(w)
    out = ops.gemm_mxfp8(aq, sa, wq, sw, bias, epi, res, gate, sel)
    ad = dequant(aq, sa, m, k).to(DEV)
    wd = dequant(wq, sw, n, k).to(DEV)
    ref = gemm_ref(ad, wd, bias, epi, res, gate, sel)                 # fp32 matmul of exactly what the MFMA multiplies
    assert rel_rms(out, ref.float()) < 2.0 ** -7                      # bf16 output rounding only
    # and against the bf16 GEMM it replaces: MX quantisation noise (two e4m3 operands)
    full = ops.gemm(a, w, bias, epi, res, gate, sel)
    r = rel_rms(out, full.float())
    print(f"[{m}x{n}x{k} epi {epi}] mxfp8 vs bf16 GEMM rel-RMS {r:.4f}")
    assert r < 0.06


def test_wan_model_with_mxfp8_linears_vs_own_bf16():
    """`enable_mxfp8_linears()`: the same forward with the six large linears per block on the MXFP8 path; compared with
    this model's own bf16 forward (there is no reference fp8 path)."""
    from oracle import wan_dit as W
    from tests.parity import hip_wan_model
    cfg = dict(W.WAN22_5B_CFG, num_attention_heads=4, attention_head_dim=128, in_channels=16, out_channels=8,
               text_dim=256, ffn_dim=1024, num_layers=3)
    sd = W.wan_random_state_dict(cfg, seed=7, dtype=torch.float32, std=0.04)
    g = torch.Generator().manual_seed(8)
    x = torch.randn(1, 16, 5, 16, 20, generator=g).to(DEV).bfloat16()
    txt = torch.randn(1, 77, 256, generator=g).to(DEV).bfloat16()
    ts = torch.full((1, 5 * 8 * 10), 811.0)
    ts[0, :80] = 0.0
    m = hip_wan_model(cfg, sd, DEV)
    ref = m(x, ts.to(DEV), txt, return_dict=False)[0]
    m.enable_mxfp8_linears()
    out = m(x, ts.to(DEV), txt, return_dict=False)[0]
    m.enable_mxfp8_linears(False)
    back = m(x, ts.to(DEV), txt, return_dict=False)[0]
    r = rel_rms(out, ref.float())
    print(f"mxfp8-linears forward vs own bf16 forward: rel-RMS {r:.4f}")
    assert 1e-4 < r < 0.1 and torch.equal(back, ref)


def test_cog_model_with_mxfp8_linears_vs_own_bf16(golden):
    from frameino_amd.cogvideox_transformer_3d import CogVideoXTransformer3DModel
    from tests.test_oracle_golden import _cog_cfg
    cfg, sd, a = golden("cog_dit_tiny")
    cfg = _cog_cfg(cfg)
    m = CogVideoXTransformer3DModel(**cfg).to(DEV)
    m.load_reference_state_dict(sd, dtype=torch.bfloat16)
    m = m.eval()
    run = lambda: m(hidden_states=a["x_def"].to(DEV).bfloat16(), encoder_hidden_states=a["txt_def"].to(DEV).bfloat16(),   # noqa: E731
                    timestep=a["ts_def"].to(DEV), image_rotary_emb=(a["cos_def"].to(DEV), a["sin_def"].to(DEV)),
                    return_dict=False)[0]
    ref = run()
    m.enable_mxfp8_linears()
    out = run()
    r = rel_rms(out, ref.float())
    print(f"cog mxfp8-linears forward vs own bf16 forward: rel-RMS {r:.4f}")
    assert r < 0.12


@pytest.mark.parametrize("m,n,k,epi", [(300, 512, 256, 0), (1000, 1024, 384, 1)])
def test_quantised_output_epilogue_equals_two_passes(m, n, k, epi):
    """fino_gemm_mxfp8_q == fino_quantize_mxfp8(fino_gemm_mxfp8(...)), byte for byte."""
    from frameino_amd import ops
    g = torch.Generator().manual_seed(3)
    a = torch.randn(m, k, generator=g).bfloat16().to(DEV)
    w = (torch.randn(n, k, generator=g) * 0.05).bfloat16().to(DEV)
    bias = torch.randn(n, generator=g).bfloat16().to(DEV)
    aq, sa = ops.quantize_mxfp8(a)
    wq, sw = ops.quantize_mxfp8(w)
    q1, s1 = ops.gemm_mxfp8_q(aq, sa, wq, sw, bias, epi)
    q2, s2 = ops.quantize_mxfp8(ops.gemm_mxfp8(aq, sa, wq, sw, bias, epi))
    assert torch.equal(q1, q2)
    rp = (m + 255) // 256 * 256
    v1 = s1.view(n // 128, rp // 256, 4, 16, 16).permute(1, 4, 3, 0, 2).reshape(rp, n // 32)[:m]
    v2 = s2.view(n // 128, rp // 256, 4, 16, 16).permute(1, 4, 3, 0, 2).reshape(rp, n // 32)[:m]
    assert torch.equal(v1, v2)


def _mx_cases(n, seed):
    import random
    rng = random.Random(seed)
    return [(rng.choice([rng.randint(1, 700), 256 * rng.randint(1, 5), 256 * rng.randint(1, 5) + 3]),
             8 * rng.randint(1, 200), 128 * rng.randint(1, 20), rng.randint(0, 4)) for _ in range(n)]


@pytest.mark.parametrize("case", _mx_cases(16, 99), ids=lambda c: "m%d_n%d_k%d_e%d" % c)
def test_gemm_mxfp8_random_shapes(case):
    from frameino_amd import ops
    from tests.test_kernels_gpu import gemm_ref
    m, n, k, epi = case
    g = torch.Generator(device=DEV).manual_seed(sum(case))
    a = torch.randn(m, k, device=DEV, generator=g).bfloat16()
    w = (torch.randn(n, k, device=DEV, generator=g) * 0.05).bfloat16()
    bias = torch.randn(n, device=DEV, generator=g).bfloat16()
    res = torch.randn(m, n, device=DEV, generator=g).bfloat16() if epi >= 2 else None
    gate = torch.randn(2, n, device=DEV, generator=g) if epi >= 3 else None
    sel = torch.randint(0, 2, (m,), device=DEV, generator=g).to(torch.int32) if epi >= 3 else None
    aq, sa = ops.quantize_mxfp8(a)
    wq, sw = ops.quantize_mxfp8(w)
    out = ops.gemm_mxfp8(aq, sa, wq, sw, bias, epi, res, gate, sel)
    ref = gemm_ref(dequant(aq, sa, m, k).to(DEV), dequant(wq, sw, n, k).to(DEV), bias, epi, res, gate, sel)
    assert rel_rms(out, ref.float()) < 2.0 ** -7


def test_token_sharded_branch_uses_the_mxfp8_projections():
    """with `enable_mxfp8_linears()` the token-sharded forward must run its separate K|V and Q projections on
    the MXFP8 path too (row blocks of the fused QKV weight, quantised on first use): same result as the unsharded MXFP8
    forward, where the fused projection is one MXFP8 GEMM (MX scales are per output row, so the row blocks quantise
    identically)."""
    import torch.distributed as dist
    from frameino_amd.configs import WAN22_5B_CFG
    from frameino_amd.parallel import TokenShard
    from frameino_amd.random_init import random_wan_model
    cfg = dict(WAN22_5B_CFG, num_attention_heads=2, num_layers=2, ffn_dim=512, text_dim=128, in_channels=8,
               out_channels=4)
    m = random_wan_model(cfg, torch.device(DEV), seed=5).enable_mxfp8_linears()
    g = torch.Generator(device=DEV).manual_seed(6)
    x = torch.randn(1, 8, 3, 16, 16, device=DEV, generator=g).bfloat16()
    txt = torch.randn(1, 32, 128, device=DEV, generator=g).bfloat16()
    ts = torch.tensor([500.0], device=DEV)
    base = m(hidden_states=x, timestep=ts, encoder_hidden_states=txt, return_dict=False)[0]
    own_group = not dist.is_initialized()
    if own_group:
        dist.init_process_group("gloo", store=dist.HashStore(), rank=0, world_size=1)
    try:
        m.parallel = TokenShard(0, 1, None, force=True)
        sharded = m(hidden_states=x, timestep=ts, encoder_hidden_states=txt, return_dict=False)[0]
        assert (0, "kv") in m._fp8 and (0, "q") in m._fp8           # the sharded projections were quantised
    finally:
        m.parallel = None
        if own_group:
            dist.destroy_process_group()
    assert rel_rms(sharded, base.float()) < 2e-3, rel_rms(sharded, base.float())


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("rows", [24640, 1000, 257])
def test_layernorm_emitting_mxfp8_equals_layernorm_then_quantise(mode, rows):
    """fino_ln_mxfp8: adaLN / affine LayerNorm / LayerNormZero with the result as MXFP8 activations == the same LayerNorm to
    bf16 followed by fino_quantize_mxfp8, byte for byte (elements and scales)"""
    from frameino_amd import ops
    d = 3072
    g = torch.Generator(device=DEV).manual_seed(rows + mode)
    x = torch.randn(rows, d, device=DEV, generator=g).bfloat16()
    tab = torch.randn(2, 2, d, device=DEV, generator=g) * 0.3
    sel = (torch.arange(rows, device=DEV) % 2).to(torch.int32)
    w = 1 + 0.1 * torch.randn(d, device=DEV, generator=g)
    b = 0.1 * torch.randn(d, device=DEV, generator=g)
    if mode == 0:
        y = ops.adaln_modulate(x, tab[:, 0], tab[:, 1], sel, 1e-6)
        q, s = ops.ln_mxfp8(0, x, shift=tab[:, 0], scale=tab[:, 1], sel=sel, eps=1e-6)
    elif mode == 1:
        y = ops.layernorm(x, w, b, 1e-6)
        q, s = ops.ln_mxfp8(1, x, weight=w, bias=b, eps=1e-6)
    else:
        y = ops.layernorm_zero(x, w, b, tab[:, 0], tab[:, 1], sel, 1e-5)
        q, s = ops.ln_mxfp8(2, x, weight=w, bias=b, shift=tab[:, 0], scale=tab[:, 1], sel=sel, eps=1e-5)
    q0, s0 = ops.quantize_mxfp8(y)
    assert torch.equal(q, q0) and torch.equal(s, s0)
