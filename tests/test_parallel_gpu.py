"""N>1 path with the REAL kernels: two processes sharing one MI355X, the collectives of frameino_amd/parallel.py carried
by gloo through host memory (RCCL needs one GPU per rank; the driver's 8-GPU node runs that) -- and, separately, the
same code forced through REAL RCCL communicators of one rank (backend "nccl", world_size 1: the call sequence, the
async all_gather_into_tensor issued from side streams, two communicators used concurrently on two streams).  Checks that the
token-sharded forward (Lq = L/2 queries against the gathered K|V, padded shard buffers, attention tail split at those
shapes) and the CFG-parallel step reproduce the single-process HIP result."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.conftest import load_golden
from tests.parity import hip_wan_model, rel_rms

pytestmark = pytest.mark.gpu


def _pipe(dev, dtype=torch.bfloat16):
    from frameino_amd.pipeline_wan_i2v_motion_frameino import WanImageToVideoPipeline
    from frameino_amd.schedulers import FlowMatchEulerDiscreteScheduler
    cfg, sd, a = load_golden("wan_pipe_tiny")
    m = hip_wan_model(cfg, {k[4:]: v for k, v in sd.items() if k.startswith("dit.")}, dev, dtype=dtype)
    return WanImageToVideoPipeline(scheduler=FlowMatchEulerDiscreteScheduler(shift=5.0), transformer=m,
                                   expand_timesteps=True), a


def _run(pipe, a, dev):
    d = lambda k: a[k].to(dev)          # noqa: E731
    return pipe.denoise(d("latents0"), d("condition"), d("traj_latents"), d("id_latent"), d("mask"),
                        d("prompt_embeds"), d("negative_embeds"), float(a["guidance"]), int(a["steps"]))


def _worker(rank, world, port, cfg_parallel, q, mode="split", overlap_local=True, exchange="kv", dtype=torch.bfloat16):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from frameino_amd import parallel
        from frameino_amd.parallel import shard_pipeline
        parallel.TokenShard.overlap_local = overlap_local
        pipe, a = _pipe("cuda:0", dtype)
        plan = shard_pipeline(pipe, rank, world, cfg_parallel=cfg_parallel, mode=mode, exchange=exchange)
        out = _run(pipe, a, "cuda:0")
        q.put((rank, plan.desc, out.cpu()))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("cfg_parallel,desc,overlap_local",
                         [(True, "cfg2xtoken1", True), (False, "cfg1xtoken2", False), (False, "cfg1xtoken2", True),
                          ("interleave", "token2x2branches-interleaved", False),
                          ("interleave", "token2x2branches-interleaved", True)])
def test_two_ranks_on_one_gpu_match_single_process(cfg_parallel, desc, overlap_local):
    pipe, a = _pipe("cuda:0")
    pipe.batch_cfg = False                                     # two batch-1 forwards, as each rank group runs them
    single = _run(pipe, a, "cuda:0").cpu()
    del pipe
    torch.cuda.empty_cache()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    mode = "interleave" if cfg_parallel == "interleave" else "split"
    procs = [ctx.Process(target=_worker, args=(r, 2, port, cfg_parallel is True, q, mode, overlap_local))
             for r in range(2)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=600) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    golden = a["out_latents"]
    e_single = rel_rms(single, golden)
    for rank, d, out in outs:
        assert d == desc
        if cfg_parallel is True or not overlap_local:
            # one attention launch over the gathered keys: same kernels on the same rows as the unsharded forward
            assert rel_rms(out, single) < 5e-3, (rank, rel_rms(out, single))
        else:
            # local-first: the softmax is taken per key range against that range's running max and merged, so P is
            # rounded to bf16 at other points than in the single pass.  Each attention is as close to fp32 as the single
            # pass (tests/test_kernels_gpu.py); through 4 sampler steps of the random tiny model the two bf16 runs
            # drift apart like any two bf16 runs do, so the bar is the fp32 oracle's golden, as for the unsharded loop
            e = rel_rms(out, golden)
            assert e < 5e-2 and e < 2.0 * e_single + 5e-3, (rank, e, e_single)
    assert torch.equal(outs[0][2], outs[1][2])                # every rank holds the same latents


@pytest.mark.parametrize("mode,desc", [("split", "cfg1xtoken2-heads"), ("interleave", "token2x2branches-interleaved-heads")])
def test_two_ranks_on_one_gpu_heads_exchange(mode, desc):
    """the all-to-all exchange with the real kernels: 2 token shards trading the tiny model's 2 heads (gloo, staged
    through host memory): one attention launch per head over the whole sequence, i.e. the single-GPU arithmetic"""
    pipe, a = _pipe("cuda:0")
    pipe.batch_cfg = False
    single = _run(pipe, a, "cuda:0").cpu()
    del pipe
    torch.cuda.empty_cache()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, False, q, mode, True, "heads")) for r in range(2)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=600) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, d, out in outs:
        assert d == desc
        assert rel_rms(out, single) < 5e-3, (rank, rel_rms(out, single))
    assert torch.equal(outs[0][2], outs[1][2])


def _nccl_worker(q, port, mode, exchange="kv"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        from frameino_amd.parallel import shard_pipeline
        assert dist.get_backend() == "nccl"
        pipe, a = _pipe("cuda:0")
        pipe.batch_cfg = False
        pipe.use_hip_graph = False
        single = _run(pipe, a, "cuda:0").cpu()
        plan = shard_pipeline(pipe, 0, 1, cfg_parallel=False, mode=mode, allow_single=True, exchange=exchange)
        assert plan.shard.force and pipe.transformer.parallel is plan.shard
        assert dist.get_backend(plan.shard.group) == "nccl"
        out = _run(pipe, a, "cuda:0")
        # a second pass: buffers and communicators are reused, nothing stale
        out2 = _run(pipe, a, "cuda:0")
        # the SHARDED step captured into a hipGraph and replayed -- for the call pattern this image's runtime captures (the
        # split plan's K|V all-gather on the step's own stream; frameino_amd/graph_step.py has the probe results).  True =
        # a loop that cannot be captured is an error; None (the default) = automatic
        from frameino_amd.graph_step import groups_capturable
        pipe.use_hip_graph = True
        if groups_capturable(plan, explicit=True):
            out_g = _run(pipe, a, "cuda:0").cpu()
        else:
            try:
                _run(pipe, a, "cuda:0")
                out_g = "captured a plan that graph_step says cannot be"
            except RuntimeError as ex:
                out_g = f"refused: {ex}"
        pipe.use_hip_graph = None
        out_auto = _run(pipe, a, "cuda:0")
        torch.cuda.synchronize()
        q.put((plan.desc, single, out.cpu(), out2.cpu(), out_g, out_auto.cpu()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,desc,exchange", [("split", "cfg1xtoken1", "kv"), ("interleave", "token1x2branches-interleaved", "kv"),
                                                ("split", "cfg1xtoken1-heads", "heads"),
                                                ("interleave", "token1x2branches-interleaved-heads", "heads")])
def test_sharded_path_through_rccl_single_rank(mode, desc, exchange):
    """backend nccl (= RCCL), world_size 1: TokenShard.all_gather_kv -> all_gather_into_tensor(async_op=True) issued
    on the branch's stream, work.wait() before the attention launch, all_gather_out, and -- interleave -- two
    communicators driven alternately from two HIP streams; `exchange="heads"`: the two all_to_all_single calls around
    the self-attention instead.  One GPU cannot show bandwidth; it does prove the call sequence, the stream semantics
    and that the result is the unsharded one."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_worker, args=(q, _free_port(), mode, exchange))
    p.start()
    try:
        d, single, out, out2, out_g, out_auto = q.get(timeout=600)
    finally:
        p.join(timeout=120)
        if p.is_alive():
            p.kill()
    assert p.exitcode == 0
    assert d == desc
    assert torch.isfinite(out).all() and torch.equal(out, out2)
    # graph replay of the sharded step (steps 1 .. n-1 replayed; step 0 eager) == the eager sharded loop, bit for bit --
    # where the runtime captures the plan's collectives; elsewhere the explicit request is refused and the default loop is eager
    # (interleave: the two branches' kernels on two side streams, both communicators' collectives issued on the step's own
    # stream -- TokenShard.issue_stream -- which is what makes the capture possible: round 5.  heads, round 6: inside a capture the
    # all-to-all is issued SYNCHRONOUSLY on that stream -- the asynchronous form segfaults in hipStreamEndCapture -- and the graph
    # dies before the process group does: the worker's destroy_process_group() returning is part of this test)
    assert torch.is_tensor(out_g) and torch.equal(out, out_g), out_g
    assert torch.equal(out, out_auto)
    # separate K|V and Q projections instead of the fused QKV GEMM: same per-element arithmetic
    assert rel_rms(out, single) < 5e-3, rel_rms(out, single)


class _Loopback:
    """the wire of P token shards simulated inside ONE process: every simulated rank runs its forward in its own thread on
    the same (default) stream and meets the others at the exchanges"""

    def __init__(self, ways):
        import threading
        self.ways, self.slots, self.barrier = ways, [None] * ways, threading.Barrier(ways)


def _loopback_shard(board, rank, exchange):
    from frameino_amd.parallel import TokenShard

    class Shard(TokenShard):
        def _meet(self, t):
            board.slots[self.rank] = t
            board.barrier.wait()            # every rank has ENQUEUED what produces its tensor (one stream: ordered)
            got = list(board.slots)
            return got

        def _all_gather(self, key, t, async_op):
            out = self._get(key, (self.ways * t.shape[0],) + tuple(t.shape[1:]), t.dtype, t.device)
            for j, tj in enumerate(self._meet(t)):
                out[j * t.shape[0]:(j + 1) * t.shape[0]].copy_(tj)
            board.barrier.wait()            # nobody overwrites its buffer before all copies are enqueued
            return out, None

        def all_to_all(self, key, send, async_op=False):
            recv = self._get(key, tuple(send.shape), send.dtype, send.device)
            for j, sj in enumerate(self._meet(send)):
                recv[j].copy_(sj[self.rank])
            board.barrier.wait()
            return recv, None

    return Shard(rank, board.ways, exchange=exchange)


@pytest.mark.parametrize("ways,exchange,lat_hw", [(4, "heads", (44, 80)), (8, "heads", (44, 80)), (4, "kv", (44, 80)),
                                                  (8, "kv", (44, 80)), (2, "kv", (44, 80)),
                                                  (8, "heads", (64, 112)), (8, "kv", (64, 112))])
def test_full_size_token_shards_in_one_process(ways, exchange, lat_hw):
    _run_full_size_token_shards(ways, exchange, lat_hw, 512)


@pytest.mark.parametrize("ways,exchange", [(4, "kv"), (8, "heads")])
def test_full_size_token_shards_with_a_zero_padded_prompt(ways, exchange):
    """the same with a 64-token prompt zero-padded to 512 rows: every rank folds the padding run into one key and runs the text
    cross-attention's out-projection re-associated on its shard's rows (frameino_amd/transformer_wan.py: _text_tail,
    _text_out_weights) -- as the unsharded forward does"""
    _run_full_size_token_shards(ways, exchange, (44, 80), 64)


def _run_full_size_token_shards(ways, exchange, lat_hw, prompt_tokens):
    """Wan2.2-5B width, L = 12320, 2 layers: the forwards of all P simulated ranks (threads, loopback exchanges) against
    the unsharded forward -- the real kernels on the real shard shapes (1540 / 3080 / 6160 tokens per rank; 3 / 6 heads
    per rank after the heads exchange), which no multi-process test on one GPU reaches.  (64, 112): BASELINE config 4's
    clip, 1024x1792, L = 25088 on 8 shards."""
    import threading
    import sys as _sys
    _sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import build_model
    from frameino_amd.configs import WAN22_5B_CFG
    cfg = dict(WAN22_5B_CFG, num_layers=2)
    dev = torch.device("cuda")
    m = build_model(cfg, dev)
    g = torch.Generator(device=dev).manual_seed(5)
    lh, lw = lat_hw
    x = torch.randn(1, 96, 14, lh, lw, device=dev, generator=g).bfloat16()
    pe = torch.randn(1, 512, cfg["text_dim"], device=dev, generator=g).bfloat16()
    pe[:, prompt_tokens:] = 0
    L = 14 * (lh // 2) * (lw // 2)
    sel = (torch.arange(L, device=dev) >= (lh // 2) * (lw // 2)).to(torch.int32)
    t_rows = torch.tensor([0.0, 700.0], device=dev)
    with torch.no_grad():
        ref = m(x, None, pe, return_dict=False, timestep_rows=(t_rows, sel))[0].float()
        board = _Loopback(ways)
        gens, outs, errs = [], [None] * ways, []
        for r in range(ways):                      # bind every rank's workspace / shard under its own cache context
            with m.cache_context(f"rank{r}"):
                gen = m.forward_steps(x, None, pe, None, False, None, (t_rows, sel), shard=_loopback_shard(board, r, exchange))
                next(gen)
            gens.append(gen)

        def drive(r):
            try:
                with torch.no_grad():
                    while True:
                        next(gens[r])
            except StopIteration as done:
                outs[r] = done.value[0]
            except Exception as ex:      # noqa: BLE001
                errs.append((r, repr(ex)))
                board.barrier.abort()

        ths = [threading.Thread(target=drive, args=(r,)) for r in range(ways)]
        for t in ths:
            t.start()
        for t in ths:
            t.join(timeout=300)
        torch.cuda.synchronize()
    assert not errs, errs
    assert ref.std().item() > 1e-3
    for r in range(ways):
        assert outs[r] is not None and torch.isfinite(outs[r].float()).all()
        e = rel_rms(outs[r], ref)
        assert e < 5e-3, (r, e)
    assert all(torch.equal(outs[0], outs[r]) for r in range(1, ways))
    if prompt_tokens < 512:
        folded = [v[2] for v in m._text_cache.values()]
        assert folded and all(t.tail is not None and t.w2 is not None for t in folded)


_CAPTURE_STRESS = r"""
import os, socket, sys
import torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from frameino_amd.graph_step import drain_collectives
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
ga, gb = dist.new_group([0]), dist.new_group([0])
x = torch.randn(4096, 256, device=dev); y1, y2 = torch.empty_like(x), torch.empty_like(x)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def step(hold=0.0):              # the interleaved plan's call pattern: kernels on two side streams, collectives issued on main
    import time
    main = torch.cuda.current_stream()
    s1.wait_stream(main); s2.wait_stream(main)
    with torch.cuda.stream(s1): a1 = x * 2
    with torch.cuda.stream(s2): a2 = x * 3
    main.wait_stream(s1); w1 = dist.all_gather_into_tensor(y1, a1, group=ga, async_op=True)
    main.wait_stream(s2); w2 = dist.all_gather_into_tensor(y2, a2, group=gb, async_op=True)
    time.sleep(hold)             # (a real step spends 5 - 15 ms of host time here: the window a watchdog sweep must not fall into)
    with torch.cuda.stream(s1): w1.wait(); r1 = y1 + a1
    with torch.cuda.stream(s2): w2.wait(); r2 = y2 + a2
    main.wait_stream(s1); main.wait_stream(s2)
    return r1 + r2

for it in range(int(sys.argv[2])):
    ref = step()                                   # eager: its collectives go on the watchdogs' lists
    drain_collectives()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        out = step(0.04)
    g.replay(); torch.cuda.synchronize()
    assert torch.equal(out, ref), it
print("STRESS OK", flush=True)
dist.destroy_process_group()
"""


def test_capture_right_behind_eager_collectives_does_not_abort():
    """graph_step.drain_collectives: c10d's watchdog must hold no eager collective when a capture pulls the communicator's stream
    in (its hipEventQuery then fails with hipErrorCapturedEvent and the watchdog aborts the process: 6 of 30 runs of the
    forced-shard rehearsal before the drain existed, profiles/r05z_capture_watchdog_abort.txt).  Twelve eager-step / capture cycles
    in a child process (an abort must not take pytest down)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", _CAPTURE_STRESS, root, "12"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "STRESS OK" in p.stdout, (p.returncode, p.stderr[-2500:])


@pytest.mark.parametrize("mode,exchange,desc", [("split", "kv", "cfg1xtoken2"), ("interleave", "kv", "token2x2branches-interleaved"),
                                                ("interleave", "heads", "token2x2branches-interleaved-heads")])
def test_two_ranks_on_one_gpu_in_fp16(mode, exchange, desc):
    """round 6: the token-sharded loop with the DiT in fp16 (the dtype the reference app loads it in, app.py:156): send / receive buffers,
    partials and the exchanges in fp16; one attention launch over the gathered keys (overlap_local off) = the unsharded arithmetic"""
    pipe, a = _pipe("cuda:0", torch.float16)
    pipe.batch_cfg = False
    single = _run(pipe, a, "cuda:0").cpu()
    del pipe
    torch.cuda.empty_cache()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, False, q, mode, False, exchange, torch.float16)) for r in range(2)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=600) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, d, out in outs:
        assert d == desc
        assert rel_rms(out, single) < 2e-3, (rank, rel_rms(out, single))          # fp16: 8x tighter than the bf16 bound
    assert torch.equal(outs[0][2], outs[1][2])
    assert rel_rms(single, a["out_latents"]) < 1e-2                                 # and the fp16 loop itself sits on the reference's run


# ---- round 6: the Wan VAE decode sharded over the ranks (parallel.sharded_vae_decode) ----
def _vae_worker(rank, world, port, q, backend):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if backend == "nccl":
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from frameino_amd.autoencoder_kl_wan import AutoencoderKLWan
        from frameino_amd.configs import WAN22_VAE_CFG
        from frameino_amd.parallel import sharded_vae_decode
        vae = AutoencoderKLWan(**WAN22_VAE_CFG).random_init_(seed=7, device="cuda:0")
        z = torch.randn(1, 48, 2, 12, 20, device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(8))
        whole = vae.decode(z, return_dict=False)[0]
        out = sharded_vae_decode(vae, z, rank, world)
        # ... and the encode of a 5-frame 192 x 320 video: 24 rows of activation behind down_blocks.1, slabs of 12 / 8 + halo 4
        from frameino_amd.parallel import sharded_vae_encode
        xv = torch.rand(1, 3, 5, 192, 320, device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(9)) * 2 - 1
        enc_same = torch.equal(sharded_vae_encode(vae, xv, rank, world).latent_dist.parameters, vae.encode(xv).latent_dist.parameters)
        q.put((rank, bool(torch.equal(out, whole)) and bool(enc_same), tuple(out.shape)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,backend", [(2, "gloo"), (3, "gloo"), (1, "nccl")], ids=["2 ranks gloo", "3 ranks gloo (ragged slabs)",
                                                                                         "RCCL communicator of one rank"])
def test_sharded_vae_decode_and_encode_on_every_rank_equal_the_whole_ones(world, backend):
    """every rank decodes its slab of the frame, one all-gather of the video rows: all ranks end with the whole clip, bit-equal to
    `vae.decode` (full-width Wan2.2 VAE, 192 x 320 frame: 48 rows of decoder-tail input / slabs of 24 and 16 + halo 10)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_vae_worker, args=(r, world, port, q, backend)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(r for r, _, _ in outs) == list(range(world))
    for rank, same, shape in outs:
        assert same and shape == (1, 3, 5, 192, 320), (rank, same, shape)
