"""Seeded random-shape sweeps of the two MFMA kernels against fp32 torch on the device: the shape-dependent control
paths (attention tail-split planning: blocks per XCD, key ranges that straddle block boundaries, head counts that
leave XCDs empty; GEMM ragged M / N, K-tile counts, every epilogue) are exercised far beyond the hand-picked cases."""
import os
import random

import pytest
import torch

from tests.parity import rel_rms
from tests.test_kernels_gpu import gemm_ref, sdpa_ref

pytestmark = pytest.mark.gpu
DEV = "cuda"
# a longer hunt on a GPU box: FINO_FUZZ_N=400 FINO_FUZZ_SEED=7 python -m pytest tests/test_fuzz_gpu.py -m gpu -q
FUZZ_N = int(os.environ.get("FINO_FUZZ_N", 48))
FUZZ_SEED = int(os.environ.get("FINO_FUZZ_SEED", 0))


def _attention_cases(n, seed):
    rng = random.Random(seed)
    out = []
    for _ in range(n):
        dh = rng.choice([64, 128])
        heads = rng.choice([1, 2, 3, 5, 8, 9, 12, 16, 24])
        b = rng.choice([1, 1, 2])
        lq = rng.choice([rng.randint(1, 300), rng.randint(300, 2600), 256 * rng.randint(1, 9), 256 * rng.randint(1, 9) + 1])
        lk = rng.choice([lq, rng.randint(1, 200), rng.randint(200, 3000), 64 * rng.randint(1, 40)])
        if b * heads * dh * (lq + 2 * lk) > 60e6:          # keep the fp32 reference small
            lk = min(lk, 1500)
            lq = min(lq, 1500)
        out.append((b, heads, dh, lq, lk))
    return out


@pytest.mark.parametrize("case", _attention_cases(FUZZ_N, 1234 + FUZZ_SEED), ids=lambda c: "b%d_h%d_d%d_q%d_k%d" % c)
def test_attention_random_shapes(case):
    from frameino_amd import ops
    b, heads, dh, lq, lk = case
    g = torch.Generator(device=DEV).manual_seed(hash(case) & 0xffff)
    q = torch.randn(b, lq, heads * dh, device=DEV, generator=g).bfloat16()
    k = torch.randn(b, lk, heads * dh, device=DEV, generator=g).bfloat16()
    v = torch.randn(b, lk, heads * dh, device=DEV, generator=g).bfloat16()
    o = ops.attention(q, k, v, heads)
    ref = sdpa_ref(q, k, v, heads)
    assert torch.isfinite(o.float()).all()
    assert rel_rms(o, ref) < 2.0 ** -6, rel_rms(o, ref)
    assert (o.float() - ref).abs().max().item() < 0.06


def _gemm_cases(n, seed):
    rng = random.Random(seed)
    out = []
    for _ in range(n):
        m = rng.choice([rng.randint(1, 64), rng.randint(64, 1200), 256 * rng.randint(1, 6), 256 * rng.randint(1, 6) + rng.randint(1, 255)])
        nn = 8 * rng.choice([rng.randint(1, 40), rng.randint(40, 400), 32 * rng.randint(1, 12)])
        k = rng.choice([64 * rng.randint(1, 24), 64 * rng.randint(1, 24), 8 * rng.randint(1, 100)])   # aligned twice as often
        epi = rng.randint(0, 4)
        out.append((m, nn, k, epi))
    return out


@pytest.mark.parametrize("case", _gemm_cases(FUZZ_N, 4321 + FUZZ_SEED), ids=lambda c: "m%d_n%d_k%d_e%d" % c)
def test_gemm_random_shapes(case):
    from frameino_amd import ops
    m, n, k, epi = case
    g = torch.Generator(device=DEV).manual_seed(hash(case) & 0xffff)
    a = torch.randn(m, k, device=DEV, generator=g).bfloat16()
    w = (torch.randn(n, k, device=DEV, generator=g) * 0.05).bfloat16()
    bias = torch.randn(n, device=DEV, generator=g).bfloat16() if epi != 2 else None
    res = torch.randn(m, n, device=DEV, generator=g).bfloat16() if epi >= 2 else None
    gate = torch.randn(3, n, device=DEV, generator=g) if epi >= 3 else None
    sel = torch.randint(0, 3, (m,), device=DEV, generator=g).to(torch.int32) if epi >= 3 else None
    out = ops.gemm(a, w, bias, epi, res, gate, sel)
    ref = gemm_ref(a, w, bias, epi, res, gate, sel)
    assert rel_rms(out, ref.float()) < 2.0 ** -7, rel_rms(out, ref.float())


@pytest.mark.parametrize("case", _attention_cases(FUZZ_N, 777 + FUZZ_SEED), ids=lambda c: "b%d_h%d_d%d_q%d_k%d" % c)
def test_attention_four_wave_folded_random_shapes(case):
    """the 4-wave kernel (FINO_TUNE_ATTN_KERNEL = 2) with the softmax scale folded into q -- CogVideoX's default
    attention path -- over the same kind of sweep: each output against SDPA of the q it was given"""
    from frameino_amd import _lib, ops
    b, heads, dh, lq, lk = case
    g = torch.Generator(device=DEV).manual_seed(hash(case) & 0xffff)
    c = dh ** -0.5 * ops.LOG2E
    qs = (torch.randn(b, lq, heads * dh, device=DEV, generator=g) * c).bfloat16()
    k = torch.randn(b, lk, heads * dh, device=DEV, generator=g).bfloat16()
    v = torch.randn(b, lk, heads * dh, device=DEV, generator=g).bfloat16()
    lib = _lib.lib()
    lib.fino_tune_set(4, 2)
    try:
        o = ops.attention(qs, k, v, heads, scale=ops.SCALE_FOLDED)
    finally:
        lib.fino_tune_set(4, 0)
    ref = sdpa_ref((qs.float() / c), k, v, heads)
    assert torch.isfinite(o.float()).all()
    assert rel_rms(o, ref) < 2.0 ** -6, rel_rms(o, ref)
    assert (o.float() - ref).abs().max().item() < 0.06


def _split_cases(n, seed):
    rng = random.Random(seed)
    out = []
    for _ in range(n):
        dh = rng.choice([64, 128])
        heads = rng.choice([1, 2, 3, 6, 12, 24])
        b = rng.choice([1, 2])
        lq = rng.choice([rng.randint(1, 300), rng.randint(300, 1600), 256 * rng.randint(1, 6)])
        nparts = rng.choice([1, 2, 3])
        lens = tuple(rng.choice([rng.randint(1, 70), rng.randint(64, 900), 64 * rng.randint(1, 12)]) for _ in range(nparts))
        out.append((b, heads, dh, lq, lens, rng.choice([0, 2, 4])))
    return out


@pytest.mark.parametrize("case", _split_cases(FUZZ_N, 99 + FUZZ_SEED), ids=lambda c: "b%d_h%d_d%d_q%d_k%s_t%d" % (
    c[0], c[1], c[2], c[3], "+".join(map(str, c[4])), c[5]))
def test_attention_partials_random_key_splits(case):
    """what the token-sharded forward does (own K/V chunk, then the gathered chunks before / after it): 1-3 partials over
    random disjoint key ranges, merged, against fp32 SDPA over all keys -- the policy, the 4-wave kernel and the LDS-DMA-staged ping-pong kernel everywhere"""
    from frameino_amd import _lib, ops
    b, heads, dh, lq, lens, tune = case
    g = torch.Generator(device=DEV).manual_seed(hash(case) & 0xffff)
    d = heads * dh
    lk = sum(lens)
    q = torch.randn(b, lq, d, device=DEV, generator=g).bfloat16()
    kv = torch.randn(b, lk, 2 * d, device=DEV, generator=g).bfloat16()
    k, v = kv[:, :, :d], kv[:, :, d:]
    lib = _lib.lib()
    lib.fino_tune_set(4, tune)
    try:
        parts, a = [], 0
        for n in lens:
            parts.append(ops.attention_partial(q, k[:, a:a + n], v[:, a:a + n], heads))
            a += n
        merged = ops.attention_merge(parts, b, lq, heads, dh, q.dtype)
    finally:
        lib.fino_tune_set(4, 0)
    ref = sdpa_ref(q, k, v, heads)
    assert torch.isfinite(merged.float()).all()
    assert rel_rms(merged, ref) < 2.0 ** -6, rel_rms(merged, ref)
    assert (merged.float() - ref).abs().max().item() < 0.06


def _walk_cases(n, seed):
    rng = random.Random(seed)
    out = []
    for _ in range(n):
        heads = rng.choice([1, 2, 3, 5, 8, 12, 24])
        b = rng.choice([1, 2, 3])
        lq = rng.choice([rng.randint(1, 300), rng.randint(300, 2600), 256 * rng.randint(1, 9), 256 * rng.randint(1, 9) + 1])
        lk = rng.choice([rng.randint(65, 200), rng.randint(200, 1024), 64 * rng.randint(2, 16), 512])
        grid = rng.choice([0, 1, 2, 3, 5, 8, 13, 64])
        out.append((b, heads, lq, lk, grid))
    return out


@pytest.mark.parametrize("case", _walk_cases(FUZZ_N, 4321 + FUZZ_SEED), ids=lambda c: "b%d_h%d_q%d_k%d_g%d" % c)
def test_walking_attention_kernel_random_shapes_and_runs(case):
    """attn_ppw_kernel (head_dim 128, whole q-blocks, 2 .. 16 key tiles) with random workgroup counts, so that a workgroup's
    run of q-blocks starts, ends and crosses heads and batches anywhere: against fp32 SDPA, and bit-equal to the policy's
    kernels on the same inputs"""
    from frameino_amd import _lib, ops
    b, heads, lq, lk, grid = case
    dh = 128
    g = torch.Generator(device=DEV).manual_seed(hash(case) & 0xffff)
    q = torch.randn(b, lq, heads * dh, device=DEV, generator=g).bfloat16()
    k = torch.randn(b, lk, heads * dh, device=DEV, generator=g).bfloat16()
    v = torch.randn(b, lk, heads * dh, device=DEV, generator=g).bfloat16()
    lib = _lib.lib()
    split = ops.SPLIT_ATTENTION_TAIL
    try:
        ops.SPLIT_ATTENTION_TAIL = False
        lib.fino_tune_set(4, 1)
        want = ops.attention(q, k, v, heads)
        lib.fino_tune_set(4, 6)
        lib.fino_tune_set(6, grid)
        o = ops.attention(q, k, v, heads)
    finally:
        lib.fino_tune_set(4, 0)
        lib.fino_tune_set(6, 0)
        ops.SPLIT_ATTENTION_TAIL = split
    ref = sdpa_ref(q, k, v, heads)
    assert torch.isfinite(o.float()).all()
    assert rel_rms(o, ref) < 2.0 ** -6, rel_rms(o, ref)
    assert torch.equal(o, want), (o.float() - want.float()).abs().max().item()
