"""fino_attn_fwd_fp8 (csrc/fino_attention_fp8.hip): self-attention with fp8 (OCP e4m3) matrix operands, head_dim 64 --
BASELINE config 5's "fp8 MFMA path" on the SDPA of architecture/attention_processor.py:2863.  There is no reference
counterpart for the precision; what is stated and checked:

  * against fp32 SDPA: rel-RMS <= 8e-2 on N(0, 1) q / k / v (an emulation of the same quantisation in torch -- block-scaled
    e4m3 q, k along the head, v along 32-key blocks, P = e4m3(exp2(s - m) 2^6) -- measures 5.5e-2: the kernel must not be
    worse than 1.2 x that emulation), and within 1.5 x of it on peaky logits;
  * both ways a softmax weight becomes its e4m3 byte (include/frameino_hip.h: FINO_FP8_P_EXP2 = exp2 + rounding, FINO_FP8_P_RAMP = the
    byte written as rne(8 (s - m) + 55.5): exp2 with a piecewise-linear mantissa), each against its own torch emulation;
  * exact properties the quantisation cannot break: V = 1 gives O = 1 (l sums the SAME rounded P that multiply V), keys past
    Lk and query rows past Lq never leak, batch / head strides, ragged tails."""
import math

import pytest
import torch

from tests.parity import record, rel_rms

pytestmark = pytest.mark.gpu
DEV = "cuda"
LOG2E = 1.4426950408889634


@pytest.fixture(params=[0, 1], ids=["free-running", "ping-pong"], autouse=True)
def fp8_kernel(request):
    """both main kernels of fino_attn_fwd_fp8: the default (4 waves, three workgroups per CU) and FINO_TUNE_ATTN_FP8_KERNEL = 1"""
    from frameino_amd import _lib
    _lib.lib().fino_tune_set(5, request.param)
    yield request.param
    _lib.lib().fino_tune_set(5, 0)


@pytest.fixture(params=["exp2", "ramp"], autouse=True)
def p_mode(request):
    """both P modes of fino_attn_fwd_fp8 as the default of ops.attention_fp8"""
    from frameino_amd import ops
    old = ops.FP8_P_DEFAULT
    ops.FP8_P_DEFAULT = ops.FP8_P_MODES[request.param]
    yield request.param
    ops.FP8_P_DEFAULT = old


def _mxq(x, dim=-1, block=32):
    x = x.transpose(dim, -1)
    shp = x.shape
    xb = x.reshape(*shp[:-1], shp[-1] // block, block)
    amax = xb.abs().amax(-1, keepdim=True).clamp_min(1e-30)
    s = torch.exp2(torch.ceil(torch.log2(amax / 448.0)))
    q = (xb / s).to(torch.float8_e4m3fn).float() * s
    return q.reshape(shp).transpose(dim, -1)


def _sdpa(q, k, v, heads):
    b, lq, hd = q.shape
    dh = hd // heads
    qh, kh, vh = (t.float().view(b, -1, heads, dh).transpose(1, 2) for t in (q, k, v))
    p = torch.softmax(qh @ kh.transpose(2, 3) * dh ** -0.5, dim=-1)
    return (p @ vh).transpose(1, 2).reshape(b, lq, hd)


def _emulated(q, k, v, heads, p_mode="exp2"):
    """the kernel's quantisation in plain torch (fp32 everywhere else, exact running maximum)"""
    b, lq, hd = q.shape
    dh = hd // heads
    lk = k.shape[1]
    pad = (-lk) % 32
    qh, kh, vh = (t.float().view(b, -1, heads, dh).transpose(1, 2) for t in (q, k, v))
    q8 = _mxq(qh * (dh ** -0.5 * LOG2E))
    k8 = _mxq(kh)
    vp = torch.nn.functional.pad(vh, (0, 0, 0, pad))
    v8 = _mxq(vp, dim=2)[:, :, :lk]
    s = q8 @ k8.transpose(2, 3)
    if p_mode == "ramp":
        # the byte IS rne(8 (s - m) + 55.5) with m a whole number of octaves (the maximum lands near 2^6); 0 below, <= 0x7e above
        m = torch.round(s.amax(-1, keepdim=True) - 6)
        byte = torch.round(8 * (s - m) + 55.5).clamp(0, 126).to(torch.uint8)
        p8 = byte.view(torch.float8_e4m3fn).float()
    else:
        p = torch.exp2(s - s.amax(-1, keepdim=True))
        p8 = (p * 64).to(torch.float8_e4m3fn).float() / 64
    return ((p8 @ v8) / p8.sum(-1, keepdim=True)).transpose(1, 2).reshape(b, lq, hd)


@pytest.mark.parametrize("b,heads,lq,lk", [(1, 2, 256, 256), (2, 3, 300, 1000), (1, 8, 1000, 777), (2, 48, 512, 2048),
                                           (1, 1, 33, 65)])
@pytest.mark.parametrize("dh", [64, 128])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_fp8_attention_vs_fp32_and_vs_the_emulated_quantisation(b, heads, lq, lk, dtype, dh, fp8_kernel, p_mode):
    from frameino_amd import ops
    if dh == 128:
        if fp8_kernel == 1:
            pytest.skip("head_dim 128 has one kernel (attn_fp8_d128_kernel)")
        heads = max(1, heads // 2)
    d = heads * dh
    g = torch.Generator(device=DEV).manual_seed(lq + lk + heads)
    q = torch.randn(b, lq, d, device=DEV, generator=g).to(dtype)
    kv = torch.randn(b, lk, 2 * d + 64, device=DEV, generator=g).to(dtype)          # row-strided k | v views
    k, v = kv[:, :, :d], kv[:, :, d:2 * d]
    out = torch.zeros(b, lq + 7, d, device=DEV, dtype=dtype)
    o = ops.attention_fp8(q, k, v, heads, out=out[:, :lq])
    assert torch.isfinite(o.float()).all() and not out[:, lq:].any()
    ref = _sdpa(q, k, v, heads)
    emu = _emulated(q, k, v, heads, p_mode)
    r, re = rel_rms(o, ref), rel_rms(emu, ref)
    record(f"attention_fp8[{p_mode}-b{b}-h{heads}x{dh}-lq{lq}-lk{lk}-{str(dtype)[6:]}]", f"rel_rms vs fp32 SDPA (torch emulation of the "
           f"quantisation: {re:.4f})", r, 8e-2)
    assert r < 8e-2 and r < 1.2 * re + 2e-3, (r, re)
    # V = 1: every row of P8 / sum(P8) sums to one whatever the rounding
    o1 = ops.attention_fp8(q, k, torch.ones_like(v), heads)
    assert (o1.float() - 1).abs().max().item() < 4e-3


def test_fp8_attention_peaky_logits_and_the_rescale_branch(p_mode):
    from frameino_amd import ops
    b, heads, lq, lk = 1, 4, 512, 3000
    d = heads * 64
    g = torch.Generator(device=DEV).manual_seed(5)
    q = (torch.randn(b, lq, d, device=DEV, generator=g) * 4).bfloat16()
    k = torch.randn(b, lk, d, device=DEV, generator=g).bfloat16()
    k[:, 1500:] *= 1.5                                       # the running maximum keeps growing along the keys
    v = torch.randn(b, lk, d, device=DEV, generator=g).bfloat16()
    o = ops.attention_fp8(q, k, v, heads)
    ref, emu = _sdpa(q, k, v, heads), _emulated(q, k, v, heads, p_mode)
    r, re = rel_rms(o, ref), rel_rms(emu, ref)
    record(f"attention_fp8[{p_mode}, peaky q x4]", f"rel_rms vs fp32 SDPA (emulation: {re:.4f})", r, 0.2)
    assert torch.isfinite(o.float()).all() and r < 1.5 * re + 5e-3, (r, re)


def test_fp8_attention_full_size_config5_sampled_rows(p_mode):
    """CogVideoX-5B FrameINO, 49 f 480x720: [2, 19126, 48 x 64]; sampled query rows against fp32 on the device, and against
    the library's own bf16 kernel"""
    from frameino_amd import ops
    b, heads, L = 2, 48, 19126
    d = heads * 64
    g = torch.Generator(device=DEV).manual_seed(9)
    q = torch.randn(b, L, d, device=DEV, generator=g).bfloat16()
    kv = torch.randn(b, L, 2 * d, device=DEV, generator=g).bfloat16()
    k, v = kv[:, :, :d], kv[:, :, d:]
    o = ops.attention_fp8(q, k, v, heads)
    ob = ops.attention(q, k, v, heads)
    rows = torch.tensor(sorted({0, 1, 255, 256, L - 1, L - 2, (L // 256) * 256, L // 2} |
                               set(torch.randint(0, L, (20,)).tolist())), device=DEV)
    for bi in range(b):
        ref = _sdpa(q[bi:bi + 1, rows], k[bi:bi + 1], v[bi:bi + 1], heads)[0]
        r, rb = rel_rms(o[bi, rows], ref), rel_rms(ob[bi, rows], ref)
        record(f"attention_fp8_full_size_config5[{p_mode}, batch {bi}]", f"rel_rms sampled rows vs fp32 SDPA (own bf16 kernel: {rb:.4f})",
               r, 8e-2)
        assert r < 8e-2, r
    assert torch.isfinite(o.float()).all()
    assert (ops.attention_fp8(q, k, torch.ones_like(v), heads).float() - 1).abs().max().item() < 4e-3


def test_fp8_attention_full_size_wan_shape_sampled_rows(fp8_kernel, p_mode):
    """Wan2.2-5B self-attention, 49 f 704x1280: [2, 12320, 24 x 128] (the CFG-batched launch); sampled query rows against fp32
    on the device, and the library's own bf16 kernel beside it"""
    if fp8_kernel == 1:
        pytest.skip("head_dim 128 has one kernel")
    from frameino_amd import ops
    b, heads, L = 2, 24, 12320
    d = heads * 128
    g = torch.Generator(device=DEV).manual_seed(10)
    q = torch.randn(b, L, d, device=DEV, generator=g).bfloat16()
    kv = torch.randn(b, L, 2 * d, device=DEV, generator=g).bfloat16()
    k, v = kv[:, :, :d], kv[:, :, d:]
    o = ops.attention_fp8(q, k, v, heads)
    ob = ops.attention(q, k, v, heads)
    rows = torch.tensor(sorted({0, 1, 255, 256, L - 1, L - 2, (L // 256) * 256, L // 2} |
                               set(torch.randint(0, L, (20,)).tolist())), device=DEV)
    for bi in range(b):
        ref = _sdpa(q[bi:bi + 1, rows], k[bi:bi + 1], v[bi:bi + 1], heads)[0]
        r, rb = rel_rms(o[bi, rows], ref), rel_rms(ob[bi, rows], ref)
        record(f"attention_fp8_full_size_wan[{p_mode}, batch {bi}]", f"rel_rms sampled rows vs fp32 SDPA (own bf16 kernel: {rb:.4f})", r, 8e-2)
        assert r < 8e-2, r
    assert torch.isfinite(o.float()).all()
    assert (ops.attention_fp8(q, k, torch.ones_like(v), heads).float() - 1).abs().max().item() < 4e-3

