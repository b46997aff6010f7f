"""Host-side pieces of the product path that need no GPU: tables and caches."""
import pytest
import torch

from frameino_amd.transformer_wan import WanTransformer3DModel, wan_rope_tables


def test_product_rope_tables_equal_reference_tables_at_the_real_geometry(golden):
    """`wan_rope_tables` (compact [L, 64] cos / sin) vs the rows recorded from the reference's WanRotaryPosEmbed at
    (14, 22, 40), head_dim 128 (tests/golden/wan_rope.npz): the processor reads cos[..., 0::2] and sin[..., 1::2]
    (architecture/transformer_wan.py:82-83), which is exactly what the compact table stores.  Bit-exact."""
    _, _, a = golden("wan_rope")
    f, h, w, d = a["shape"].tolist()
    cos, sin = wan_rope_tables(d, 1024, f, h, w)
    assert cos.shape == (f * h * w, d // 2) and cos.dtype == torch.float32
    rows = a["rows"].long()
    assert torch.equal(cos[rows], a["cos"][:, 0::2])
    assert torch.equal(sin[rows], a["sin"][:, 1::2])
    # the reference's table repeats every value pairwise (repeat_interleave_real): nothing is lost by the compaction
    assert torch.equal(a["cos"][:, 0::2], a["cos"][:, 1::2]) and torch.equal(a["sin"][:, 0::2], a["sin"][:, 1::2])


def _tiny():
    return WanTransformer3DModel(num_attention_heads=2, attention_head_dim=24, in_channels=8, out_channels=4, text_dim=16,
                                 freq_dim=32, ffn_dim=64, num_layers=2, rope_max_seq_len=64)


def test_derived_state_is_dropped_when_parameters_move_or_change():
    """`_packed` / `_text_cache` / `_fp8` are detached copies -- `.to()`, `load_state_dict` and
    `load_reference_state_dict` must invalidate them."""
    m = _tiny()
    m._pack()
    m._text_cache["cond"] = (torch.zeros(1), 0, None)
    assert m._packed is not None
    m.to(torch.float64)
    assert m._packed is None and not m._text_cache
    m._pack()
    m.load_state_dict(m.state_dict())
    assert m._packed is None
    m._pack()
    m.load_reference_state_dict({k: v.clone() for k, v in m.state_dict().items()})
    assert m._packed is None


def test_mxfp8_requantise_is_deferred_to_the_next_forward():
    """with MXFP8 on, `.to()` / `load_state_dict` must not run the (GPU) quantiser on the spot -- a move to the
    host would raise half way through; the flag is remembered and the next forward re-quantises."""
    from frameino_amd.cogvideox_transformer_3d import CogVideoXTransformer3DModel
    cog = CogVideoXTransformer3DModel(num_attention_heads=2, attention_head_dim=16, in_channels=6, out_channels=2,
                                      time_embed_dim=16, text_embed_dim=8, num_layers=1, sample_width=8, sample_height=8,
                                      sample_frames=9, max_text_seq_length=5, use_rotary_positional_embeddings=True,
                                      use_learned_positional_embeddings=True)
    for m in (_tiny(), cog):
        m._fp8 = {(0, "qkv"): ("bytes", "scales")}          # as left by enable_mxfp8_linears() on a GPU
        m.to(torch.float64)                                   # no quantiser call (it would fail: no GPU here)
        assert not m._fp8 and m._fp8_pending
        m.load_state_dict(m.state_dict())
        assert m._fp8_pending
        m.enable_mxfp8_linears(False)
        assert not m._fp8_pending


def test_shared_prefix_needs_identical_modulation_rows():
    """`x.expand(2, ...)` with two DIFFERENT timesteps is a legal call; the branch-invariant-prefix
    shortcut must not serve element 0's layer-0 branch to element 1."""
    from tests import cpu_ops
    torch.manual_seed(5)
    m = _tiny().float().eval()
    with torch.no_grad():
        for p_ in m.parameters():
            p_.copy_(torch.randn(p_.shape) * (0.5 if p_.ndim == 1 else p_.shape[-1] ** -0.5))
    m.reset_caches()
    m.ops = cpu_ops
    x = torch.randn(1, 8, 2, 4, 4)
    txt = torch.randn(2, 6, 16)
    L = 2 * 2 * 2

    def run(ts, dedup, **kw):
        m.dedup_shared_prefix = dedup
        return m(x.expand(2, -1, -1, -1, -1), ts, txt, return_dict=False, **kw)[0]

    for ts in (torch.tensor([900.0, 100.0]),                                  # per-sample scalar timesteps
               torch.stack([torch.full((L,), 900.0), torch.full((L,), 100.0)])):     # per-sample per-token timesteps
        on, off = run(ts, True), run(ts, False)
        assert torch.equal(on, off)
        assert (on[0] - on[1]).abs().max() > 1e-3                             # the elements really differ
    # the cases the shortcut is for stay bit-identical to the general path
    per_tok = torch.full((1, L), 700.0)
    per_tok[0, :4] = 0.0
    for ts, kw in ((torch.tensor([500.0]), {}), (per_tok, {}),
                   (None, {"timestep_rows": (torch.tensor([0.0, 700.0]),
                                             (per_tok[0] > 0).to(torch.int32))})):
        assert torch.equal(run(ts, True, **kw), run(ts, False, **kw))


def test_swapping_a_processor_after_the_first_forward_is_seen():
    from frameino_amd.attention_processor import MI355WanAttnProcessor

    class Mine(MI355WanAttnProcessor):
        pass

    m = _tiny()
    m._pack()
    assert m._default_processors()
    m.blocks[1].attn2.set_processor(Mine())
    assert not m._default_processors()           # evaluated per call, not frozen at pack time


def test_bench_watchdog_prints_the_measured_line_and_exits_zero_on_a_stall():
    """bench.py (N>1): once the first plan's JSON line exists, a stalled second phase must not lose it."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, time; sys.path.insert(0, %r); import bench; d = bench.Watchdog(0); "
            "d.arm('probe', 0.5, fallback='{\"metric\": \"x\"}'); time.sleep(30); print('NOT REACHED')" % root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == '{"metric": "x"}' and "stalled" in r.stderr
    code = ("import sys, time; sys.path.insert(0, %r); import bench; d = bench.Watchdog(1); "
            "d.arm('timed run', 0.5); time.sleep(30)" % root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and r.stdout.strip() == ""


def test_step_graph_policy_and_schedule():
    """frameino_amd/graph_step.py: which loops are captured (the static rule behind the RCCL capture probes) and the
    schedule -- step 0 eager, step 1 captured, the rest replays -- with the capture machinery replaced by a counter."""
    from types import SimpleNamespace
    from frameino_amd import graph_step as G
    assert G.groups_capturable(None)
    shard = SimpleNamespace(active=True)
    plan = lambda **kw: SimpleNamespace(**{**dict(interleave=False, exchange="kv", shard=shard, world=1, token_group=None,
                                                  cfg_group=None, token_group_b=None), **kw})      # noqa: E731
    assert G.groups_capturable(plan())                                    # split plan, K|V all-gather, one rank
    # round 5: the interleaved plan's K|V all-gathers are issued on the step's own stream (TokenShard.issue_stream) while the
    # branches' kernels run on side streams -- that pattern captures (profiles/r05_rccl_capture_probe.txt)
    assert G.groups_capturable(plan(interleave=True))
    # round 6: the heads all-to-all captures as a SYNCHRONOUS collective on the step's own stream (TokenShard._issue)
    assert G.groups_capturable(plan(exchange="heads")) and G.groups_capturable(plan(interleave=True, exchange="heads"))
    assert not G.groups_capturable(plan(world=4))                         # real links: only on request
    assert G.groups_capturable(plan(world=4), explicit=True)
    assert G.groups_capturable(plan(world=4, interleave=True), explicit=True)
    assert G.groups_capturable(plan(world=4, interleave=True, exchange="heads"), explicit=True)
    assert not G.groups_capturable(plan(world=4, interleave=True, exchange="heads"))
    # schedule: eager when not capturable / fewer than three steps / mode False; an impossible explicit request raises
    calls = []
    sg = G.StepGraph(lambda: calls.append("s"), None, False, 10)
    for _ in range(4):
        sg.step()
    assert calls == ["s"] * 4 and sg.graph is None
    assert not G.StepGraph(lambda: None, None, True, 2).enabled and G.StepGraph(lambda: None, None, True, 3).enabled
    assert not G.StepGraph(lambda: None, False, True, 50).enabled
    with pytest.raises(RuntimeError, match="cannot be captured"):
        G.StepGraph(lambda: None, True, False, 50)


def test_text_padding_fold_and_reassociated_out_projection_are_exact_algebra_in_fp32():
    """The host logic of DESIGN.md 4.6 / 4.7 without a GPU: `_text_tail` (the zero run found per sample, rows kept, key counts,
    multiplicities) and `_text_out_weights` (W2 = [W_o,h V_h^T]_h from the block-diagonal V) driven through torch stand-ins of
    fino_attn_fwd_tail / fino_attn_probs / fino_row_rrms in fp32: the folded forward and the folded + re-associated forward equal
    the forward over all text rows up to fp32 rounding -- the identities are exact, only bf16 rounding separates them on the GPU."""
    import math
    import types
    from tests import cpu_ops
    torch.manual_seed(11)
    m = _tiny().float().eval()
    with torch.no_grad():
        for p_ in m.parameters():
            p_.copy_(torch.randn(p_.shape) * (0.5 if p_.ndim == 1 else p_.shape[-1] ** -0.5))
    m.reset_caches()
    ops = types.SimpleNamespace(**{k: getattr(cpu_ops, k) for k in dir(cpu_ops) if not k.startswith("_")})

    def logits(q, k, heads, lk_b, mult):
        b, lq, hd = q.shape
        dh = hd // heads
        s = (q.reshape(b, lq, heads, dh).transpose(1, 2).float() @ k.reshape(b, -1, heads, dh).transpose(1, 2).float().transpose(2, 3)) * dh ** -0.5
        for i in range(b):
            s[i, :, :, lk_b[i] - 1] += math.log(mult[i])
            s[i, :, :, lk_b[i]:] = -math.inf
        return s

    def attention_tail(q, k, v, heads, lk_b, tail_mult, out=None, scale=None):
        b, lq, hd = q.shape
        p = torch.softmax(logits(q, k, heads, lk_b, tail_mult), dim=-1)
        o = (p @ v.reshape(b, -1, heads, hd // heads).transpose(1, 2).float()).transpose(1, 2).reshape(b, lq, hd).to(q.dtype)
        return o if out is None else out.copy_(o)

    def attention_probs(q, k, heads, lk_b, tail_mult, kp, out=None, scale=None, q_rrms=None, q_weight=None):
        b, lq, hd = q.shape
        if q_rrms is not None:
            q = q * q_rrms.reshape(b, lq, 1) * q_weight
        p = torch.softmax(logits(q, k, heads, lk_b, tail_mult), dim=-1)[..., :kp]          # [b, heads, lq, kp]
        p = p.transpose(1, 2).reshape(b, lq, heads * kp).to(q.dtype)
        return p if out is None else out.copy_(p)

    ops.attention_tail, ops.attention_probs = attention_tail, attention_probs
    ops.attention_tail_supported = ops.attention_probs_supported = lambda *a: True
    ops.row_rrms = lambda x, eps, out=None: out.copy_(torch.rsqrt(x.float().pow(2).mean(-1) + eps)) if out is not None \
        else torch.rsqrt(x.float().pow(2).mean(-1) + eps)
    m.ops = ops
    x = torch.randn(1, 8, 2, 4, 4).expand(2, -1, -1, -1, -1).contiguous()
    txt = torch.randn(2, 256, 16)
    txt[0, 5:] = 0
    txt[1, 3:] = 0
    ts = torch.tensor([700.0, 700.0])

    def run(fold, reassoc):
        m.dedup_text_padding, m.reassociate_text_out = fold, reassoc
        m.reset_caches()
        y = m(x, ts, txt, return_dict=False)[0]
        return y, next(iter(m._text_cache.values()))[2]

    plain, t0 = run(False, False)
    folded, t1 = run(True, False)
    both, t2 = run(True, True)
    assert t0.tail is None and t0.lt == 256
    assert t1.tail == ([6, 4], [251.0, 253.0]) and t1.lt == 128 and t1.w2 is None
    assert t2.w2 is not None and t2.kp == [8, 8] and t2.w2[0][0].shape == (48, 2 * 8)
    assert (folded - plain).abs().max().item() < 1e-4 * plain.abs().max().item()
    assert (both - plain).abs().max().item() < 1e-4 * plain.abs().max().item()
    # a prompt without a zero run is left alone
    m.dedup_text_padding, m.reassociate_text_out = True, True
    m.reset_caches()
    m(x, ts, torch.randn(2, 256, 16), return_dict=False)
    assert next(iter(m._text_cache.values()))[2].tail is None
