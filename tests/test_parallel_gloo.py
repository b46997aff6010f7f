"""N>1 path of the denoising loop on CPU (gloo, world_size 2): the sharding / collective logic of
frameino_amd/parallel.py -- CFG-branch parallelism and token shards with the K|V all-gather -- must reproduce the
single-process result.  Kernels are replaced by tests/cpu_ops.py here (no GPU in this container); the HIP kernels
themselves are covered by the -m gpu tests."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import cpu_ops
from tests.conftest import load_golden
from tests.parity import model_cfg


def _build(cfg, dit_sd):
    from frameino_amd.pipeline_wan_i2v_motion_frameino import WanImageToVideoPipeline
    from frameino_amd.schedulers import FlowMatchEulerDiscreteScheduler
    from frameino_amd.transformer_wan import WanTransformer3DModel
    m = WanTransformer3DModel(**model_cfg(cfg))
    m.load_reference_state_dict(dit_sd, dtype=torch.float32)
    m.ops = cpu_ops
    return WanImageToVideoPipeline(scheduler=FlowMatchEulerDiscreteScheduler(shift=5.0), transformer=m.eval(),
                                   expand_timesteps=True)


def _run(pipe, a):
    return pipe.denoise(a["latents0"], a["condition"], a["traj_latents"], a["id_latent"], a["mask"],
                        a["prompt_embeds"], a["negative_embeds"], float(a["guidance"]), int(a["steps"]))


def _four_head_case(heads=4):
    """a seeded random 4-head (8-head) model (the golden model has 2 heads: 4 / 8 token shards cannot trade them) and the
    golden run's inputs; the oracle here is the same model in one process"""
    cfg, sd, a = load_golden("wan_pipe_tiny")
    cfg = dict(cfg, num_attention_heads=heads)
    from frameino_amd.pipeline_wan_i2v_motion_frameino import WanImageToVideoPipeline
    from frameino_amd.schedulers import FlowMatchEulerDiscreteScheduler
    from frameino_amd.transformer_wan import WanTransformer3DModel
    torch.manual_seed(11)
    m = WanTransformer3DModel(**model_cfg(cfg)).float()
    with torch.no_grad():
        for p_ in m.parameters():
            p_.copy_(torch.randn(p_.shape) * (0.5 if p_.ndim == 1 else p_.shape[-1] ** -0.5))
    m.reset_caches()
    m.ops = cpu_ops
    g = torch.Generator().manual_seed(3)
    a = dict(a, prompt_embeds=torch.randn(1, 12, cfg["text_dim"], generator=g),
             negative_embeds=torch.randn(1, 12, cfg["text_dim"], generator=g))
    pipe = WanImageToVideoPipeline(scheduler=FlowMatchEulerDiscreteScheduler(shift=5.0), transformer=m.eval(),
                                   expand_timesteps=True)
    return pipe, a


def _worker(rank, world, port, cfg_parallel, q, mode="split", exchange="kv", four_heads=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from frameino_amd.parallel import shard_pipeline
        if four_heads:
            pipe, a = _four_head_case(4 if four_heads is True else int(four_heads))
        else:
            cfg, sd, a = load_golden("wan_pipe_tiny")
            pipe = _build(cfg, {k[4:]: v for k, v in sd.items() if k.startswith("dit.")})
        plan = shard_pipeline(pipe, rank, world, cfg_parallel=cfg_parallel, mode=mode, exchange=exchange)
        out = _run(pipe, a)
        q.put((rank, plan.desc, out))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("cfg_parallel,desc", [(True, "cfg2xtoken1"), (False, "cfg1xtoken2"),
                                               ("interleave", "token2x2branches-interleaved")])
def test_two_rank_plans_match_single_process(cfg_parallel, desc):
    cfg, sd, a = load_golden("wan_pipe_tiny")
    single = _run(_build(cfg, {k[4:]: v for k, v in sd.items() if k.startswith("dit.")}), a)
    # the CPU stand-in itself reproduces the reference pipeline's recorded run (fp32)
    torch.testing.assert_close(single, a["out_latents"], atol=2e-4, rtol=2e-4)

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    mode = "interleave" if cfg_parallel == "interleave" else "split"
    procs = [ctx.Process(target=_worker, args=(r, 2, port, cfg_parallel is True, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, d, out in outs:
        assert d == desc
        torch.testing.assert_close(out, single, atol=1e-4, rtol=1e-4)   # fp32; the local-first attention merges
        # (O, m, l) partials of two or three key ranges: another summation order than the single softmax pass


@pytest.mark.parametrize("mode,desc", [("split", "cfg2xtoken2"), ("interleave", "token4x2branches-interleaved")])
def test_four_rank_plans_match_single_process(mode, desc):
    """BASELINE config 3's structure (4 ranks): cfg x token = 2 x 2, and both branches interleaved on 4 token shards."""
    cfg, sd, a = load_golden("wan_pipe_tiny")
    single = _run(_build(cfg, {k[4:]: v for k, v in sd.items() if k.startswith("dit.")}), a)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 4, port, True, q, mode)) for r in range(4)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=600) for _ in range(4)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, d, out in outs:
        assert d == desc
        torch.testing.assert_close(out, single, atol=1e-4, rtol=1e-4)   # fp32; the local-first attention merges
        # (O, m, l) partials of two or three key ranges: another summation order than the single softmax pass


def _spawn(world, args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port) + tuple(args[:1]) + (q,) + tuple(args[1:])) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return outs


@pytest.mark.parametrize("world,cfg_parallel,mode,desc", [
    (2, False, "split", "cfg1xtoken2-heads"), (2, True, "interleave", "token2x2branches-interleaved-heads"),
    (4, True, "split", "cfg2xtoken2-heads")])
def test_heads_exchange_matches_single_process(world, cfg_parallel, mode, desc):
    """the all-to-all exchange (token shards <-> head shards around the self-attention) on the golden 2-head model:
    2 token shards, alone, with both branches interleaved, and under the CFG split of 4 ranks"""
    cfg, sd, a = load_golden("wan_pipe_tiny")
    single = _run(_build(cfg, {k[4:]: v for k, v in sd.items() if k.startswith("dit.")}), a)
    for rank, d, out in _spawn(world, (cfg_parallel, mode, "heads")):
        assert d == desc
        torch.testing.assert_close(out, single, atol=2e-5, rtol=2e-5)     # one softmax pass per head, as on one GPU


@pytest.mark.parametrize("exchange,desc", [("heads", "token4x2branches-interleaved-heads"),
                                           ("kv", "token4x2branches-interleaved")])
def test_four_token_shards_of_a_four_head_model(exchange, desc):
    """4 token shards trading 4 heads (one head per rank), both CFG branches interleaved; the K|V all-gather on the same
    model beside it"""
    pipe, a = _four_head_case()
    single = _run(pipe, a)
    assert torch.isfinite(single).all() and single.std() > 0.1
    for rank, d, out in _spawn(4, (True, "interleave", exchange, True)):
        assert d == desc
        torch.testing.assert_close(out, single, atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("mode,exchange,heads,desc", [
    ("split", "kv", 0, "cfg2xtoken4"), ("interleave", "kv", 0, "token8x2branches-interleaved"),
    ("split", "heads", 8, "cfg2xtoken4-heads"), ("interleave", "heads", 8, "token8x2branches-interleaved-heads")])
def test_eight_rank_plans_match_single_process(mode, exchange, heads, desc):
    """The machine the north star names is 8 ranks (BASELINE config 4 is 8-way): the plans `make_plan(rank, 8)` builds -- cfg 2 x
    token 4 (2 + 4 communicators... every rank creates all of them in the same order) and both branches interleaved on 8 token
    shards (two communicators over all 8 ranks) -- as 8 PROCESSES over gloo, every rank's latents against the single process.
    24 tokens: 6 / 3 rows per shard.  The heads all-to-all runs on a seeded 8-head model (4 shards x 2 heads, 8 shards x 1 head)."""
    if heads:
        pipe, a = _four_head_case(heads)
    else:
        cfg, sd, a = load_golden("wan_pipe_tiny")
        pipe = _build(cfg, {k[4:]: v for k, v in sd.items() if k.startswith("dit.")})
    single = _run(pipe, a)
    assert torch.isfinite(single).all()
    outs = _spawn(8, (True, mode, exchange, heads or False))
    assert sorted(r for r, _, _ in outs) == list(range(8))
    for rank, d, out in outs:
        assert d == desc, d
        torch.testing.assert_close(out, single, atol=1e-4, rtol=1e-4)
    for rank, d, out in outs[1:]:
        assert torch.equal(out, outs[0][2])           # every rank ends the loop on the same latents, bit for bit


def test_two_token_shards_trade_two_head_groups():
    """4 heads on 2 token shards: each rank's 2 heads travel as two groups (TokenShard.head_groups), i.e. two all-to-alls
    in flight around two attention launches"""
    from frameino_amd.parallel import TokenShard
    assert TokenShard(0, 2).head_ranges(2) == [(0, 1), (1, 2)] and TokenShard(0, 8).head_ranges(3) == [(0, 1), (1, 3)]
    assert TokenShard(0, 2).head_ranges(1) == [(0, 1)] and TokenShard(0, 2).head_ranges(12) == [(0, 6), (6, 12)]
    pipe, a = _four_head_case()
    single = _run(pipe, a)
    for rank, d, out in _spawn(2, (False, "split", "heads", True)):
        assert d == "cfg1xtoken2-heads"
        torch.testing.assert_close(out, single, atol=1e-4, rtol=1e-4)


def test_heads_exchange_refuses_indivisible_heads():
    from frameino_amd.parallel import ParallelPlan, shard_pipeline
    cfg, sd, _ = load_golden("wan_pipe_tiny")
    pipe = _build(cfg, {k[4:]: v for k, v in sd.items() if k.startswith("dit.")})
    plan = ParallelPlan(0, 3, 1, 3, None, None, exchange="heads")
    with pytest.raises(ValueError, match="divisible"):
        shard_pipeline(pipe, 0, 3, plan=plan)


def test_token_shard_rows_cover_sequence_with_padding():
    from frameino_amd.parallel import TokenShard
    for L, ways in ((12320, 8), (12320, 4), (25088, 8), (101, 4), (72, 2)):
        seen = []
        for r in range(ways):
            lo, n, lpad = TokenShard(r, ways).rows(L)
            assert lpad * ways >= L and n <= lpad
            seen += list(range(lo, lo + n))
        assert seen == list(range(L))
