"""fino_attn_fwd_tail (csrc/fino_attention.hip, attn_ppw_kernel<T, true>): cross-attention over key sequences whose tail is ONE
row repeated -- the zero-padded prompt of pipelines/pipeline_wan_i2v_motion_FrameINO.py:235-238 seen through attn2's K / V
projections (architecture/transformer_wan.py:108): every padding token yields the same K and V row, so
    softmax(q.[K; k x M]^T) [V; v x M] = softmax(q.[K; k]^T + [0; ln M]) [V; v].
Checked against the library's own attention on the EXPANDED sequences (what the reference computes) and against fp32 SDPA."""
import pytest
import torch

from tests.parity import record, rel_rms

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _sdpa(q, k, v, heads):
    b, lq, hd = q.shape
    dh = hd // heads
    qh, kh, vh = (t.float().view(b, -1, heads, dh).transpose(1, 2) for t in (q, k, v))
    p = torch.softmax(qh @ kh.transpose(2, 3) * dh ** -0.5, dim=-1)
    return (p @ vh).transpose(1, 2).reshape(b, lq, hd)


def _case(b, heads, lq, n_real, total, lc, seed, dtype=torch.bfloat16, junk=True):
    """expanded k / v [b, total, d] whose rows >= n_real[i] all equal one row, and the compact form [b, lc, d] (+ lk_b, mult)"""
    d = heads * 128
    g = torch.Generator(device=DEV).manual_seed(seed)
    q = torch.randn(b, lq, d, device=DEV, generator=g).to(dtype)
    k = torch.randn(b, total, d, device=DEV, generator=g).to(dtype)
    v = torch.randn(b, total, d, device=DEV, generator=g).to(dtype)
    for i, n in enumerate(n_real):
        k[i, n:] = k[i, n].clone()
        v[i, n:] = v[i, n].clone()
    kc, vc = k[:, :lc].clone(), v[:, :lc].clone()
    if junk:                                   # rows past lk_b are never looked at (finite values)
        for i, n in enumerate(n_real):
            kc[i, n + 1:] = 37.0
            vc[i, n + 1:] = -91.0
    return q, k, v, kc, vc, [n + 1 for n in n_real], [total - n for n in n_real]


@pytest.mark.parametrize("b,heads,lq,n_real,total,lc", [
    (2, 24, 3000, (64, 8), 512, 128),          # the bench's prompt lengths, one ragged q-block at the end
    (2, 4, 1111, (127, 0), 512, 128),          # the last key closes a tile; a prompt of padding only
    (1, 6, 700, (100,), 512, 192),             # three key tiles, the second half of them masked
    (3, 2, 513, (5, 190, 64), 256, 192),
    (2, 24, 12320, (64, 8), 512, 128),         # the bench shape itself
])
def test_tail_attention_equals_attention_on_the_expanded_keys(b, heads, lq, n_real, total, lc):
    from frameino_amd import ops
    assert ops.attention_tail_supported(b, heads, lq, lc, 128)
    q, k, v, kc, vc, lk_b, mult = _case(b, heads, lq, n_real, total, lc, seed=lq + total)
    full = ops.attention(q, k, v, heads)
    tail = ops.attention_tail(q, kc, vc, heads, lk_b, mult)
    assert torch.isfinite(tail.float()).all()
    rows = slice(None) if lq <= 3000 else torch.randint(0, lq, (512,), device=DEV)
    ref = _sdpa(q[:, rows], k, v, heads)
    r_tail, r_full, r_between = rel_rms(tail[:, rows], ref), rel_rms(full[:, rows], ref), rel_rms(tail, full)
    record(f"attention_tail[b{b}-h{heads}-lq{lq}-real{'/'.join(map(str, n_real))}-of{total}]",
           f"rel_rms vs fp32 SDPA on the expanded keys (plain kernel on them: {r_full:.5f}; tail vs plain: {r_between:.5f})",
           r_tail, 2 ** -7.5)
    assert r_tail < 2 ** -7.5 and r_tail < 1.3 * r_full + 1e-4 and r_between < 2 ** -7.5, (r_tail, r_full, r_between)


def test_tail_attention_fp16():
    from frameino_amd import ops
    b, heads, lq, n_real, total, lc = 2, 4, 1000, (64, 8), 512, 128
    q, k, v, kc, vc, lk_b, mult = _case(b, heads, lq, n_real, total, lc, seed=4, dtype=torch.float16)
    tail = ops.attention_tail(q, kc, vc, heads, lk_b, mult)
    r = rel_rms(tail, _sdpa(q, k, v, heads))
    record("attention_tail[fp16]", "rel_rms vs fp32 SDPA on the expanded keys", r, 2 ** -10)
    assert r < 2 ** -10, r


def test_tail_attention_with_multiplicity_one_is_attention_on_the_first_keys():
    """tail_mult = 1: nothing but per-batch key counts -- the same sums over the same keys"""
    from frameino_amd import ops
    b, heads, lq = 2, 4, 900
    q, k, v, kc, vc, lk_b, _ = _case(b, heads, lq, (70, 20), 128, 128, seed=3)
    out = ops.attention_tail(q, kc, vc, heads, lk_b, [1.0, 1.0])
    for i in range(b):
        ref = ops.attention(q[i:i + 1], kc[i:i + 1, :lk_b[i]], vc[i:i + 1, :lk_b[i]], heads)
        assert rel_rms(out[i:i + 1], ref) < 1e-3, i


def test_tail_attention_prompt_of_padding_only_returns_the_padding_value_row():
    from frameino_amd import ops
    heads, lq = 2, 300
    q, k, v, kc, vc, lk_b, mult = _case(1, heads, lq, (0,), 512, 128, seed=9)
    out = ops.attention_tail(q, kc, vc, heads, lk_b, mult)
    assert torch.equal(out, vc[:, :1].expand(1, lq, heads * 128))


def test_tail_attention_refuses_what_the_walking_kernel_cannot_run():
    from frameino_amd import ops
    q = torch.zeros(1, 64, 128, device=DEV, dtype=torch.bfloat16)
    kv = torch.zeros(1, 64, 128, device=DEV, dtype=torch.bfloat16)
    assert not ops.attention_tail_supported(1, 1, 64, 64, 128) and not ops.attention_tail_supported(1, 2, 64, 128, 64)
    with pytest.raises(RuntimeError):
        ops.attention_tail(q, kv, kv, 1, [3], [10.0])


@pytest.mark.parametrize("b,heads,lq,n_real,total,lk_alloc", [
    (1, 24, 3000, (64,), 512, 128), (1, 24, 2500, (8,), 512, 128), (2, 4, 1111, (127, 0), 512, 128),
    (1, 6, 700, (30,), 512, 64), (3, 2, 513, (5, 100, 64), 256, 128),
])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_attention_probabilities_of_a_short_key_sequence(b, heads, lq, n_real, total, lk_alloc, dtype):
    """fino_attn_probs: P = softmax(q.K^T) per head with the padding run as one key == the probabilities of fp32 softmax over the
    expanded keys, the run's mass summed; rows sum to one; columns past a sample's key count are zeros."""
    from frameino_amd import ops
    q, k, v, kc, vc, lk_b, mult = _case(b, heads, lq, n_real, total, lk_alloc, seed=7 * lq + total, dtype=dtype)
    kp = -(-max(lk_b) // 8) * 8
    assert ops.attention_probs_supported(b, heads, lq, lk_alloc, 128)
    p = ops.attention_probs(q, kc, heads, lk_b, mult, kp).float().view(b, lq, heads, kp)
    qh = q.float().view(b, lq, heads, 128).transpose(1, 2)
    kh = k.float().view(b, total, heads, 128).transpose(1, 2)
    full = torch.softmax(qh @ kh.transpose(2, 3) * 128 ** -0.5, dim=-1).transpose(1, 2)          # [b, lq, heads, total]
    worst = 0.0
    for i, n in enumerate(n_real):
        ref = torch.cat([full[i, :, :, :n], full[i, :, :, n:].sum(-1, keepdim=True)], dim=-1)      # the run's mass on key n
        got = p[i, :, :, :n + 1]
        worst = max(worst, (got - ref).abs().max().item())
        assert not p[i, :, :, n + 1:].any()
        assert (p[i].sum(-1) - 1).abs().max().item() < 2e-2
    record(f"attention_probs[b{b}-h{heads}-lq{lq}-real{'/'.join(map(str, n_real))}-{str(dtype)[6:]}]",
           "max abs error of a probability vs fp32 softmax over the expanded keys", worst, 6e-3)
    assert worst < 6e-3, worst


def test_probabilities_with_q_normalised_on_load_equal_those_of_the_normalised_q():
    """fino_row_rrms + fino_attn_probs(q_rrms, q_weight): RMS-normalising q inside the kernel (statistic from its own pass) gives the
    bits of normalising q first with fino_rmsnorm_rope"""
    from frameino_amd import ops
    b, heads, lq = 2, 6, 1500
    d = heads * 128
    g = torch.Generator(device=DEV).manual_seed(21)
    q = (torch.randn(b * lq, d, device=DEV, generator=g) * 3).bfloat16()
    w = (1 + 0.1 * torch.randn(d, device=DEV, generator=g)).bfloat16()
    k = torch.randn(b, 128, d, device=DEV, generator=g).bfloat16()
    lk_b, mult = [65, 9], [448.0, 504.0]
    rr = ops.row_rrms(q, 1e-6)
    ref_rr = torch.rsqrt(q.float().pow(2).mean(-1) + 1e-6)
    assert (rr / ref_rr - 1).abs().max().item() < 1e-5
    p_fused = ops.attention_probs(q.view(b, lq, d), k, heads, lk_b, mult, 72, q_rrms=rr.view(b, lq), q_weight=w)
    qn = q.clone()
    ops.rmsnorm_rope_(qn, w, 1e-6)
    p_ref = ops.attention_probs(qn.view(b, lq, d), k, heads, lk_b, mult, 72)
    assert torch.equal(p_fused, p_ref)


@pytest.mark.parametrize("grid", [1, 3, 7])
def test_tail_attention_runs_that_walk_across_samples_with_different_key_counts(grid):
    """FINO_TUNE_ATTN_WALK_GRID small: one workgroup walks a run of q-blocks that crosses heads AND samples, i.e. the per-sample key
    count and logit offset change inside a run (and the K / V streams, three tiles ahead, are already in the next sample)"""
    from frameino_amd import _lib, ops
    b, heads, lq, n_real, total, lc = 3, 2, 700, (5, 190, 64), 256, 192
    q, k, v, kc, vc, lk_b, mult = _case(b, heads, lq, n_real, total, lc, seed=grid)
    ref = ops.attention_tail(q, kc, vc, heads, lk_b, mult)
    _lib.lib().fino_tune_set(6, grid)
    try:
        out = ops.attention_tail(q, kc, vc, heads, lk_b, mult)
    finally:
        _lib.lib().fino_tune_set(6, 0)
    assert torch.equal(out, ref)
    assert rel_rms(out, _sdpa(q, k, v, heads)) < 2 ** -7.5


def test_tail_and_probabilities_random_shapes():
    """random (samples, heads, rows, prompt lengths, allocated rows): the tail kernel against fp32 SDPA over the expanded keys, the
    probabilities against fp32 softmax, P.V of the probabilities against the tail kernel's output"""
    import os
    import random
    from frameino_amd import ops
    rng = random.Random(int(os.environ.get("FINO_FUZZ_SEED", "20240")))
    for case in range(int(os.environ.get("FINO_FUZZ_CASES", "16"))):
        b = rng.randint(1, 4)
        heads = rng.choice([1, 2, 3, 6])
        lq = rng.randint(1, 1400)
        lc = rng.choice([128, 128, 192])
        total = rng.choice([256, 512])
        n_real = tuple(rng.randint(0, min(lc, total) - 2) for _ in range(b))
        q, k, v, kc, vc, lk_b, mult = _case(b, heads, lq, n_real, total, lc, seed=1000 + case)
        ref = _sdpa(q, k, v, heads)
        tail = ops.attention_tail(q, kc, vc, heads, lk_b, mult)
        r = rel_rms(tail, ref)
        assert r < 2 ** -7.5, (case, b, heads, lq, lc, total, n_real, r)
        if lc <= 128:
            kp = -(-max(lk_b) // 8) * 8
            p = ops.attention_probs(q, kc, heads, lk_b, mult, kp).float().view(b, lq, heads, kp)
            vh = vc[:, :kp].float().view(b, kp, heads, 128)
            o = torch.einsum("blhk,bkhd->blhd", p, vh).reshape(b, lq, heads * 128)
            assert rel_rms(o, ref) < 2 ** -7, (case, "probs", rel_rms(o, ref))
