#!/usr/bin/env python3
"""Headline benchmark: denoise-steps/s of the FrameINO Wan2.2-TI2V-5B pipeline, 49 frames at 704x1280
("720p": 720 is not a legal Wan2.2-5B height), bf16, synthetic latents, random-init weights.

One "step" = what the reference loop does per iteration (pipelines/pipeline_wan_i2v_motion_FrameINO.py:809-908):
model-input assembly, cond + uncond DiT forward (L = 14 x 22 x 40 = 12320 tokens, one ID frame), CFG, Euler update.

    python bench.py [--gpus N] [--steps K] [--warmup W]          # N > 1: starts its own N ranks (launch_ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                   # ... or is started as one of them

Prints ONE JSON line (rank 0).  N > 1 shards the SAME clip (strong scaling): CFG branches and/or token shards with a
K/V all-gather over RCCL per attention layer (frameino_amd/parallel.py).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

WORKLOADS = {
    # name: (latent frames generated, latent h, latent w)
    "wan2.2-5b-49f-704x1280": (13, 44, 80),
    "wan2.2-5b-49f-1024x1792": (13, 64, 112),
    "wan2.2-5b-81f-704x1280": (21, 44, 80),          # the app's default clip length (app.py:543): L = 19360
    "tiny": (3, 8, 12),
}


def wan_flops_per_forward(L, cfg, text_len=512, cross_keys=None, text_proj=True):
    """algorithmic FLOPs of one DiT forward (2.M.N.K), time-MLP on 2 rows.  For the FLOPs actually ISSUED per step:
    `cross_keys` = the keys the text cross-attention really attends to (the zero-padded tail of a prompt folded into one key:
    WanTransformer3DModel.dedup_text_padding), `text_proj=False` = without the text embedder and the layers' text K / V
    projections (step-invariant: computed once per prompt, outside the steps)."""
    d = cfg["num_attention_heads"] * cfg["attention_head_dim"]
    f, nl = cfg["ffn_dim"], cfg["num_layers"]
    kin = cfg["in_channels"] * 4
    ck = text_len if cross_keys is None else cross_keys
    per_layer = 8 * L * d * d + 4 * L * L * d + (4 * L * d * d + (4 * text_len * d * d if text_proj else 0)) + 4 * L * ck * d + 4 * L * d * f
    return nl * per_layer + 2 * L * kin * d + 2 * L * d * cfg["out_channels"] * 4 + \
        (2 * text_len * (cfg["text_dim"] * d + d * d) if text_proj else 0)


def wan_flops_shared_prefix(L, cfg):
    """what the CFG-batched forward computes ONCE for both branches (WanTransformer3DModel.dedup_shared_prefix): the patch
    embedding and layer 0's self-attention branch (QKV, SDPA, out projection)"""
    d = cfg["num_attention_heads"] * cfg["attention_head_dim"]
    return 2 * L * cfg["in_channels"] * 4 * d + 8 * L * d * d + 4 * L * L * d


def wan_flops_dead_rows(L, dead, cfg, cross_keys, reassoc_k):
    """what one forward does NOT compute for `dead` token rows in the LAST block and the output head when the caller reads only
    the other rows (WanTransformer3DModel.forward(live_rows=): the ID frame's and the re-imposed first frame's tokens): their
    self-attention queries, out-projection, text branch (q projection, attention over `cross_keys` keys, out-projection -- K =
    `reassoc_k` when it ran re-associated), FFN and head.  Their q | k | v projection still runs (k | v are needed, the GEMM is one)."""
    d, f = cfg["num_attention_heads"] * cfg["attention_head_dim"], cfg["ffn_dim"]
    out2 = 2 * (reassoc_k if reassoc_k else d) * d + (0 if reassoc_k else 2 * cross_keys * d)
    per_row = 4 * L * d + 2 * d * d + 2 * d * d + 2 * cross_keys * d + out2 + 4 * d * f + 2 * d * cfg["out_channels"] * 4
    return dead * per_row


def build_model(cfg, device, seed=0, dtype=torch.bfloat16):
    """Random-init Wan2.2-5B (no checkpoints offline): N(0, 0.02^2) weights generated on the device."""
    from frameino_amd.random_init import random_wan_model
    return random_wan_model(cfg, device, seed, dtype=dtype)


def cpu_baseline(cfg, L, budget_s=28.0):
    """The oracle (CPU restatement, `kind: port`) timed on this box's host cores on a bounded sample: ONE full-size
    WanTransformerBlock in fp32 at a token count sized to the budget, extrapolated to steps/s =
    1 / (2 forwards x layers x t_block(L)).  Reported baseline only.  The thread count is FOUND, not assumed: a sweep over
    {16, 32, 64, 128, all cores} (those the box has; it stops once a setting is 1.8x slower than the best so far), the first
    (cold) repetition of every setting discarded, the best setting reported with its thread count (256
    threads on a 3080-row block from a cold start gave 108 GFLOP/s)."""
    from oracle import wan_dit as W
    ncpu = os.cpu_count() or 1
    one = dict(cfg, num_layers=1)
    sd = W.wan_random_state_dict(one, seed=1, dtype=torch.float32)
    d = cfg["num_attention_heads"] * cfg["attention_head_dim"]
    Ls = min(L, 3080)                      # quarter of the sequence: ~0.7 TFLOP, seconds on host cores
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, Ls, d, generator=g)
    txt = torch.randn(1, 512, d, generator=g)
    temb = torch.randn(1, 6, d, generator=g) * 0.1
    rot = W.wan_rope(cfg["attention_head_dim"], 1024, Ls // 880 if Ls >= 880 else 1, 22 if Ls >= 880 else 1,
                     40 if Ls >= 880 else Ls)
    if rot[0].shape[2] != Ls:
        rot = (rot[0][:, :, :1].expand(1, 1, Ls, -1).contiguous(), rot[1][:, :, :1].expand(1, 1, Ls, -1).contiguous())
    f = cfg["ffn_dim"]
    fl = lambda n: 8 * n * d * d + 4 * n * n * d + 4 * n * d * d + 4 * 512 * d * d + 4 * n * 512 * d + 4 * n * d * f  # noqa
    settings = sorted({t for t in (16, 32, 64, 128, ncpu) if t <= ncpu} or {ncpu})
    t_start = time.time()
    sweep, best = {}, None
    for threads in settings:
        if best is not None and time.time() - t_start > budget_s:
            break                                                    # (the budget bounds the sweep; the best so far stands)
        torch.set_num_threads(threads)
        t0 = time.time()
        W.wan_block(sd, "blocks.0", one, x, txt, temb, rot)          # cold repetition: discarded
        cold = time.time() - t0
        reps, t0 = 0, time.time()
        while True:
            W.wan_block(sd, "blocks.0", one, x, txt, temb, rot)
            reps += 1
            if reps >= 3 or time.time() - t0 > 4.0 or time.time() - t_start > budget_s:
                break
        t_blk = (time.time() - t0) / reps
        sweep[str(threads)] = {"s_per_block": round(t_blk, 3), "gflops": round(fl(Ls) / t_blk / 1e9), "reps": reps,
                               "cold_s": round(cold, 3)}
        if best is None or t_blk < best[1]:
            best = (threads, t_blk, reps)
        elif t_blk > 1.8 * best[1]:
            break                                                    # past the knee: more threads only fight over the caches
    threads, t_blk, reps = best
    torch.set_num_threads(threads)
    # per-block FLOPs at Ls and at L -> scale the measured time by the FLOP ratio (attention is quadratic)
    t_full = t_blk * fl(L) / fl(Ls)
    steps_s = 1.0 / (2 * cfg["num_layers"] * t_full)
    return {"value": steps_s, "unit": "denoise-steps/s", "cores": threads, "kind": "port", "host_cpus": ncpu,
            "thread_sweep": sweep,
            "sample": f"oracle WanTransformerBlock fp32, D={d} F={f}, L={Ls}: best of a thread sweep {settings} = {threads} threads "
                      f"({reps} warm reps, {t_blk:.2f}s each, {fl(Ls) / t_blk / 1e9:.0f} GFLOP/s; the cold repetition of every "
                      f"setting discarded), extrapolated by FLOPs to L={L} x {cfg['num_layers']} layers x 2 forwards"}


class Watchdog:
    """Ends the process when a phase of the N>1 run stalls (a hung collective would otherwise lose the whole record).
    `fallback` is a JSON line already measured: rank 0 prints it before leaving, so the driver still gets a valid
    line.  Never re-exec: the process has touched the GPU; os._exit tears the context down."""

    ABORT_KEY = "fino_bench_abort"

    def __init__(self, rank):
        import threading
        self.rank, self.phase, self.deadline, self.fallback, self.ok = rank, "init", None, None, False
        self.store = None              # the process group's store once it exists: carries a peer's "I am leaving"
        self._lock = threading.Lock()
        t = threading.Thread(target=self._run, daemon=True)
        t.start()

    def arm(self, phase, seconds, fallback=None, ok=False):
        """ok: the result is already out, a stall in this phase (teardown) is not a failure"""
        with self._lock:
            self.phase, self.deadline, self.fallback, self.ok = phase, time.monotonic() + seconds, fallback, ok

    def disarm(self):
        with self._lock:
            self.deadline = None

    def abort(self, why):
        """a rank that cannot go on (exception in a probe) tells its peers before it leaves: their watchdogs see the key
        within half a second instead of sitting in the next collective until their own deadline"""
        try:
            if self.store is not None:
                self.store.set(self.ABORT_KEY, f"rank {self.rank}: {why}")
        except Exception:      # noqa: BLE001  (rank 0 = the store's host may be gone already)
            pass

    def _peer_left(self):
        if self.store is None:
            return False
        try:
            return bool(self.store.check([self.ABORT_KEY]))
        except Exception:      # noqa: BLE001  the store's host (rank 0) has left
            return True

    def _run(self):
        while True:
            time.sleep(0.5)
            with self._lock:
                armed = self.deadline is not None
                late = armed and time.monotonic() >= self.deadline
                phase, fb, ok = self.phase, self.fallback, self.ok
            if not late and not (armed and fb is not None and self._peer_left()):
                continue
            print(f"[bench] rank {self.rank}: phase '{phase}' {'stalled' if late else 'left by a peer'} -- leaving"
                  f"{' with the line already measured' if fb else ''}", file=sys.stderr, flush=True)
            if fb is not None and self.rank == 0:
                print(fb, flush=True)
            os._exit(0 if (fb is not None or ok) else 3)


def launch_ranks(n, argv):
    """`python bench.py --gpus N` (N > 1) outside a launcher: start the N ranks OURSELVES, as a child process --
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py
    <same args>` -- relay rank 0's JSON line and return the child's exit status.  The parent has not touched the GPU
    (`import torch` does not initialise HIP) and never exec's: a process that holds a GPU context must not be replaced."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC only on this pool (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    print(f"[bench] --gpus {n} without a launcher: starting {' '.join(cmd[1:8])} ... as a child", file=sys.stderr, flush=True)
    import signal
    # the ranks run in their own session: a SIGTERM / SIGINT that reaches this parent (a harness timeout) is forwarded to the whole
    # group, and whatever way this function is left the group is ended -- no orphaned ranks holding the GPUs and their
    # RCCL communicators
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True, start_new_session=True)

    def end_group(sig=signal.SIGTERM):
        try:
            os.killpg(child.pid, sig)
        except (ProcessLookupError, PermissionError):
            pass

    def forward(signum, _frame):
        end_group(signum)
        try:
            child.wait(timeout=20)
        except subprocess.TimeoutExpired:
            end_group(signal.SIGKILL)
        raise SystemExit(128 + signum)

    old_handlers = {sg: signal.signal(sg, forward) for sg in (signal.SIGTERM, signal.SIGINT)}
    line = None
    try:
        for out in child.stdout:                  # stderr is inherited; stdout is filtered down to THE line
            if out.startswith('{"metric"'):
                line = out.strip()
            else:
                sys.stderr.write(out)
        rc = child.wait()
    finally:
        if child.poll() is None:                  # left by an exception: the ranks must not outlive the launcher
            end_group()
            try:
                child.wait(timeout=20)
            except subprocess.TimeoutExpired:
                end_group(signal.SIGKILL)
        else:
            end_group(signal.SIGKILL)             # stragglers of a finished launcher (no-op when the group is gone)
        for sg, h in old_handlers.items():
            signal.signal(sg, h)
    if line is not None:
        print(line, flush=True)
    if rc == 0 and line is None:
        print("[bench] the ranks exited 0 without a result line", file=sys.stderr, flush=True)
        return 4
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="wan2.2-5b-49f-704x1280", choices=sorted(WORKLOADS))
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured hipGraph")
    ap.add_argument("--cfg-streams", action="store_true", help="CFG branches on two concurrent streams (A/B)")
    ap.add_argument("--plan", choices=["auto", "split", "interleave"], default="auto",
                    help="N>1: cfg x token split, both CFG branches interleaved on token shards, or (auto, >= 4 GPUs) "
                         "measure split, then probe interleave under a watchdog and keep the faster")
    ap.add_argument("--exchange", choices=["auto", "kv", "heads"], default="auto",
                    help="N>1 self-attention exchange of the token shards: K|V all-gather, heads all-to-all, or (auto) "
                         "both probed")
    ap.add_argument("--mxfp8", action="store_true",
                    help="NOT the headline: large linears on the MXFP8 path (BASELINE config 5 style), attention in bf16")
    ap.add_argument("--fp8-attention", action="store_true",
                    help="NOT the headline (with --mxfp8): self-attention with fp8 (e4m3) operands too (fino_attn_fwd_fp8)")
    ap.add_argument("--logit-scale", type=float, default=1.0,
                    help="NOT the headline: scale of the attention-probe q (peaky logits make the rescale branch fire)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-vae", action="store_true", help="skip the once-per-clip VAE encode/decode timing")
    ap.add_argument("--no-clip", action="store_true", help="skip the one real 50-step pipeline call (sec_per_clip_measured)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary numbers (hipGraph replay, UniPC, config 4, config 5, attention probes)")
    ap.add_argument("--layers", type=int, default=None, help="debug only: fewer layers (result marked invalid)")
    ap.add_argument("--graph-probe", action="store_true",
                    help="N>1, opt-in: after the eager measurements also replay the best plan's step from a hipGraph (only "
                         "the split plan with the K|V all-gather is capturable on this image's runtime -- "
                         "frameino_amd/graph_step.py; a crash inside a capture cannot be caught, so never by default)")
    ap.add_argument("--no-graph-probe", action="store_true",
                    help="N>1: skip the hipGraph replay of the best plan that runs after the line is printed")
    ap.add_argument("--graph-probe-after-line", action="store_true",
                    help="more than one rank: after the line is printed, also replay the best plan's step from a hipGraph and "
                         "report to stderr.  The default with ONE rank (--force-shard rehearsal); with more ranks it has to be "
                         "asked for: a capture over real links has never run here, a fault inside it cannot be caught and a "
                         "hang would sit out --stall-s -- neither may cost a scaling run its exit status or its time")
    ap.add_argument("--force-shard", action="store_true",
                    help="rehearsal: take the N>1 code path (process group, sharded forward, collectives) with whatever "
                         "--gpus says, 1 included: one rank drives real RCCL communicators of size 1.  With --plan "
                         "interleave / --exchange heads the other plans' call sequences run the same way")
    ap.add_argument("--stall-s", type=float, default=240.0, help="N>1: seconds a phase may take before the watchdog acts")
    a = ap.parse_args()

    if (a.gpus > 1 or a.force_shard) and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(a.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} does not match WORLD_SIZE={world}: the line's n_gpus would be a guess")
    backend = os.environ.get("FINO_DIST_BACKEND", "nccl")      # "gloo": rehearsal of the N>1 flow on fewer GPUs than ranks
    local = local % max(torch.cuda.device_count(), 1) if backend != "nccl" else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    multi = world > 1 or a.force_shard
    dog = Watchdog(rank) if multi else None
    if multi:
        dog.arm("init_process_group", a.stall_s)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        dog.disarm()
        try:
            dog.store = dist.distributed_c10d._get_default_store()
        except Exception:      # noqa: BLE001
            dog.store = None

    from frameino_amd import _lib, ops
    _lib.load()                                   # no fallback: fail here if the HIP library is missing
    from frameino_amd.configs import WAN22_5B_CFG
    from frameino_amd.pipeline_wan_i2v_motion_frameino import WanImageToVideoPipeline
    from frameino_amd.schedulers import FlowMatchEulerDiscreteScheduler

    cfg = dict(WAN22_5B_CFG)
    if a.workload == "tiny":
        # (8 heads: the 8-rank rehearsal can then trade heads on 4 and on 8 token shards)
        cfg.update(num_attention_heads=8, num_layers=2, ffn_dim=512, text_dim=64, in_channels=8, out_channels=4)
    if a.layers:
        cfg["num_layers"] = a.layers
    fg, lh, lw = WORKLOADS[a.workload]
    nid = 1
    C = cfg["out_channels"]
    L = (fg + nid) * (lh // 2) * (lw // 2)
    model = build_model(cfg, dev)
    if a.mxfp8:
        model.enable_mxfp8_linears()
    if a.fp8_attention:
        model.enable_fp8_attention()
    pipe = WanImageToVideoPipeline(scheduler=FlowMatchEulerDiscreteScheduler(shift=5.0), transformer=model,
                                   expand_timesteps=True)
    plans = {}
    if multi:
        from frameino_amd.parallel import make_plan, shard_pipeline
        dog.arm("communicators", a.stall_s)
        # every rank creates every communicator, in the same order
        base = {}
        if a.plan in ("auto", "split"):
            base["split"] = make_plan(rank, world, True, "split", allow_single=a.force_shard)
        if a.plan == "interleave" or (a.plan == "auto" and world >= 4):
            base["interleave"] = make_plan(rank, world, mode="interleave", allow_single=a.force_shard)
        # each plan with the K|V all-gather and with the heads all-to-all (same communicators); the all-gather form of the
        # first plan is measured first -- its line stands whatever happens in the probes that follow
        nheads = cfg["num_attention_heads"]
        for name, pl in base.items():
            heads_ok = (pl.token_ways > 1 or a.force_shard) and nheads % pl.token_ways == 0
            if a.exchange in ("auto", "kv") or not heads_ok:
                plans[name] = pl
            if a.exchange in ("auto", "heads") and heads_ok:
                plans[name + "-heads"] = pl.with_exchange("heads")
            # the K|V gather in two head groups (attention of group 0 under the gather of group 1): only a MODELLED wire says it
            # wins (split plan, >= 4 token shards), so it is a probe on the real links, not a default
            if a.exchange == "auto" and name == "split" and pl.token_ways >= 4 and nheads % 2 == 0:
                plans[name + "-kvg2"] = pl.with_kv_groups(2)
        shard_pipeline(pipe, rank, world, plan=next(iter(plans.values())))
        dog.disarm()
    pipe.use_hip_graph = a.graph
    pipe.cfg_streams = a.cfg_streams

    def make_inputs(fg_, lh_, lw_, text_dim):
        g = torch.Generator().manual_seed(1234)           # CPU generator, then copy
        lat = torch.randn(1, C, fg_, lh_, lw_, generator=g).to(dev)
        cond = torch.randn(1, C, 1, lh_, lw_, generator=g).to(dev)
        traj = torch.randn(1, C, fg_ + nid, lh_, lw_, generator=g).to(dev)
        traj[:, :, fg_:] = 0
        idl = torch.randn(1, C, nid, lh_, lw_, generator=g).to(dev)
        mask = torch.ones(1, 1, fg_, lh_, lw_, device=dev)
        mask[:, :, 0] = 0
        pe = torch.randn(1, 512, text_dim, generator=g)
        pe[:, 64:] = 0                                     # zero padding past the prompt length (:235-238)
        ne = torch.randn(1, 512, text_dim, generator=g)
        ne[:, 8:] = 0
        return lat, cond, traj, idl, mask, pe.to(dev).bfloat16(), ne.to(dev).bfloat16()

    lat, cond, traj, idl, mask, pe, ne = make_inputs(fg, lh, lw, cfg["text_dim"])
    total = a.warmup + a.steps
    pipe.scheduler.set_timesteps(max(total, 2), device=dev)
    st = pipe.make_state(lat, cond, traj, idl, mask, pe, ne, 5.0)
    ts, dts = pipe.scheduler.timesteps.to(dev).float(), pipe.scheduler.dts.to(dev)
    lat0 = st.lat.clone()

    def sync():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if not multi:
            return x
        t = torch.tensor([x], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.item()

    def timed_run(warmup, steps, use_graph=False, timer_names=()):
        """`warmup` untimed steps, then exactly `steps` steps between barrier + synchronize pairs; MAX over ranks."""
        st.lat.copy_(lat0)
        graph = None

        def one_step(i):
            nonlocal graph
            i = min(i, ts.numel() - 1)
            st.t_rows[1:2].copy_(ts[i:i + 1])
            st.dt.copy_(dts[i:i + 1])
            if use_graph:
                if graph is None:
                    snap = st.lat.clone()
                    pipe._step(st)                       # eager pass fills every lazy cache before the capture
                    st.lat.copy_(snap)
                    graph = torch.cuda.CUDAGraph()
                    # (a process group's watchdog thread polls events: see frameino_amd/graph_step.py::capture_error_mode and
                    #  ::drain_collectives -- no eager collective may be on a watchdog's list when the capture opens)
                    from frameino_amd.graph_step import drain_collectives
                    drain_collectives()
                    with torch.cuda.graph(graph, capture_error_mode="thread_local" if multi else "global"):
                        pipe._step(st)
                graph.replay()
            else:
                pipe._step(st)

        with torch.no_grad():
            for i in range(warmup):
                one_step(i)
            sync()
            timer = ops.KernelTimer(set(timer_names))
            t0 = time.perf_counter()
            with timer:
                for i in range(warmup, warmup + steps):
                    one_step(i)
            sync()
            elapsed = time.perf_counter() - t0
        return max_over_ranks(elapsed), timer

    def kv_gather_us(plan):
        """the K|V all-gather of one layer-call on this plan's communicator, alone on the wire (HIP events, rank 0)"""
        sh = plan.shard
        if sh.ways <= 1:
            return None
        _, n_, lpad = sh.rows(L)
        buf = sh.kv_local(lpad, 2 * model.inner_dim, torch.bfloat16, dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for rep in range(6):
            if rep == 1:
                torch.cuda.synchronize()
                e0.record()
            _, work = sh.all_gather_kv(buf)
            if work is not None:
                work.wait()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 5 * 1e3

    def heads_a2a_us(plan):
        """the two all-to-all calls of one layer-call of the heads exchange (q|k|v out, o back), alone on the wire"""
        sh = plan.shard
        if sh.ways <= 1:
            return None
        _, n_, lpad = sh.rows(L)
        dp = model.inner_dim // sh.ways
        s1 = sh.a2a_buffer("qkv_send", (sh.ways, lpad, 3, dp), torch.bfloat16, dev)
        s2 = sh.a2a_buffer("o_send", (sh.ways, lpad, dp), torch.bfloat16, dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for rep in range(6):
            if rep == 1:
                torch.cuda.synchronize()
                e0.record()
            sh.all_to_all("qkv_recv", s1)            # (the forward sends the same bytes in TokenShard.head_groups pieces)
            sh.all_to_all("o_recv", s2)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 5 * 1e3

    def wire_us(plan):
        return heads_a2a_us(plan) if plan.exchange == "heads" else kv_gather_us(plan)

    base_cfg = {"workload": a.workload, "tokens": L, "layers": cfg["num_layers"], "guidance": 5.0, "id_frames": nid}

    issued = {}

    def issued_flops(shared):
        """FLOPs one step really ISSUES, from the model's per-prompt cache as it stands right after the headline run (later
        secondary runs reset it): called once, then frozen."""
        fold = getattr(model, "dedup_text_padding", False)      # noqa: F841  (the cache entries say what the fold did)
        # ... read from the model's per-prompt cache, not assumed: the keys each sample really attends to, and whether
        # its out-projection ran re-associated as P.(V W_o^T) with K = heads x kp instead of D
        d_ = cfg["num_attention_heads"] * cfg["attention_head_dim"]
        flops_step, text_route = 0.0, []
        entries = [v[2] for v in getattr(model, "_text_cache", {}).values()]
        samples = []
        for ent in entries:
            nb_ = ent.kv[0].shape[0] // ent.lt if ent.kv else 1
            for i in range(nb_):
                keys = ent.tail[0][i] if ent.tail is not None else ent.lt
                kp = ent.kp[i] if (ent.w2 is not None) else None
                samples.append((keys, kp))
        if len(samples) != 2:                       # (no forward has run yet, or an unexpected cache layout: the padded counts)
            samples = [(512, None), (512, None)]
        for keys, kp in samples:
            f_ = wan_flops_per_forward(L, cfg, cross_keys=keys, text_proj=False)
            if kp is not None:                      # out-projection 2 L D D -> 2 L (heads kp) D, the P.V product is gone
                f_ += cfg["num_layers"] * (2 * L * cfg["num_attention_heads"] * kp * d_ - 2 * L * d_ * d_ - 2 * L * keys * d_)
            # rows of the last block whose output the loop discards (ID frame, re-imposed first frame): keys / values only
            live_ = getattr(st, "live_rows", None) if (not multi and getattr(model, "skip_dead_rows", False)) else None
            if live_ is not None:
                f_ -= wan_flops_dead_rows(L, L - (live_[1] - live_[0]), cfg, keys,
                                          None if kp is None else cfg["num_attention_heads"] * kp)
            flops_step += f_
            text_route.append({"keys": int(keys), "reassociated_out_projection_k": None if kp is None else int(cfg["num_attention_heads"] * kp)})
        flops_step -= wan_flops_shared_prefix(L, cfg) if shared else 0
        issued.update(flops=flops_step, route=text_route,
                      dead_rows_last_block=0 if (getattr(st, "live_rows", None) is None or multi or not getattr(model, "skip_dead_rows", False))
                      else L - (st.live_rows[1] - st.live_rows[0]))

    def result_line(elapsed, parallelism, extra_cfg, roofline=None, cpu=None, use_graph=False):
        ms_step = elapsed / a.steps * 1e3
        # FLOPs actually issued: on one GPU the two CFG branches are one batch-2 forward whose branch-invariant prefix
        # runs once (the N > 1 plans run two whole batch-1 forwards)
        shared = not multi and getattr(model, "dedup_shared_prefix", False) and not a.cfg_streams
        # ... the text K / V come from the per-prompt cache, and the zero-padded tails of the two prompts (64 and 8 tokens of 512:
        # make_inputs) are one key each (token shards included)
        if "flops" not in issued:
            issued_flops(shared)
        flops_step = issued["flops"]
        extra_cfg = dict(extra_cfg, text_branch_as_run=issued["route"], dead_rows_last_block=issued.get("dead_rows_last_block", 0))
        out = {
            "metric": "denoise-steps/sec (Wan2.2-5B FrameINO, 49f 704x1280, cond+uncond DiT forward + CFG + Euler)",
            "value": a.steps / elapsed, "unit": "denoise-steps/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None,
            "dtype": ("bf16" if not (a.mxfp8 or a.fp8_attention) else
                      ("mxfp8 linears (e4m3 + e8m0/32)" if a.mxfp8 else "bf16 linears") +
                      (" + fp8 (e4m3) attention operands" if a.fp8_attention else " + bf16 attention") + " -- not the headline"),
            "data": "synthetic",
            "config": dict(base_cfg, hip_graph=bool(use_graph), parallelism=parallelism,
                           sec_per_50_step_clip_denoise_only=50 * ms_step / 1e3,
                           model_tflops_per_s=flops_step / (ms_step * 1e-3) / 1e12,
                           model_flops_per_step_issued=flops_step,
                           model_flops_per_step_algorithmic=2 * wan_flops_per_forward(L, cfg), **extra_cfg),
        }
        if a.layers:
            out["config"]["INVALID_reduced_layers"] = a.layers
        if roofline is not None:
            out["roofline"] = roofline
        if cpu is not None:
            out["cpu_baseline"] = cpu
        return json.dumps(out)

    # ================================================================ N > 1: strong scaling of the same clip
    if multi:
        names = list(plans)
        first = plans[names[0]]
        shard_pipeline(pipe, rank, world, plan=first)
        dog.arm(f"timed run ({first.desc})", a.stall_s + 2.0 * total)
        elapsed, _ = timed_run(a.warmup, a.steps)
        assert torch.isfinite(st.lat).all(), "non-finite latents"
        issued_flops(False)
        probe = {first.desc: elapsed / a.steps * 1e3}
        gather_us = {first.desc: wire_us(first)}

        # how many ranks the communicator really reached: a sum of ones over the default group (RCCL when backend = nccl)
        seen_t = torch.ones(1, device=dev, dtype=torch.float64)
        dist.all_reduce(seen_t)
        ranks_seen = int(seen_t.item())

        clip_cfg = {}

        def line_for(el, plan, graphed=False):
            return result_line(el, plan.desc, dict({"rccl_ranks": world, "ranks_seen": ranks_seen, "backend": backend,
                                                    "plan_probe_ms_per_step": probe,
                                                    "exchange_us_per_layer_call_alone_on_the_wire": gather_us,
                                                    "local_first_attention": plan.shard.local_first() and plan.exchange == "kv",
                                                    "attention_exchange": plan.exchange if plan.token_ways > 1 else None}, **clip_cfg),
                               use_graph=graphed)

        best = (elapsed, first, False)
        line = line_for(*best)
        graph_probe = backend == "nccl" and a.graph_probe             # (a gloo exchange is staged through the host)
        if len(names) > 1 or graph_probe:
            # The first plan's line exists; from here on a stall costs nothing: the watchdog prints that line and leaves.
            # An exception in a probe (a collective the node's RCCL refuses, an out-of-memory) is treated like a stall: the
            # line already measured goes out and the process leaves -- the other ranks' watchdogs do the same.
            try:
                cand = None
                for nm in names[1:]:
                    other = plans[nm]
                    dog.arm(f"probe ({other.desc})", a.stall_s, fallback=line)
                    shard_pipeline(pipe, rank, world, plan=other)
                    el_p, _ = timed_run(1, 2)
                    probe[other.desc] = el_p / 2 * 1e3
                    gather_us[other.desc] = wire_us(other)
                    if rank == 0:
                        print(f"[bench] {first.desc}: {probe[first.desc]:.1f} ms/step; {other.desc}: "
                              f"{probe[other.desc]:.1f} ms/step (probe)", file=sys.stderr, flush=True)
                    if cand is None or probe[other.desc] < probe[cand.desc]:
                        cand = other
                    line = line_for(*best)                               # carries the probe times so far
                if cand is not None and probe[cand.desc] < 0.98 * probe[first.desc]:
                    dog.arm(f"timed run ({cand.desc})", a.stall_s + 2.0 * total, fallback=line)
                    shard_pipeline(pipe, rank, world, plan=cand)
                    el2, _ = timed_run(a.warmup, a.steps)
                    if el2 < elapsed and bool(torch.isfinite(st.lat).all()):
                        best = (el2, cand, False)
                line = line_for(*best)                                   # carries every plan's probe time
                from frameino_amd.graph_step import groups_capturable
                if graph_probe and groups_capturable(best[1], explicit=True):
                    # the best plan's step captured into a hipGraph (RCCL collectives and side streams inside the capture)
                    # and replayed: what pipe.denoise() does by default.  Probed like another plan -- a stall or an error
                    # here costs nothing, the eager line is already there
                    bplan = best[1]
                    key = bplan.desc + "+hipgraph"
                    dog.arm(f"probe ({key})", a.stall_s, fallback=line)
                    shard_pipeline(pipe, rank, world, plan=bplan)
                    el_g, _ = timed_run(1, 2, use_graph=True)
                    probe[key] = el_g / 2 * 1e3
                    if rank == 0:
                        print(f"[bench] {bplan.desc}: eager {best[0] / a.steps * 1e3:.1f} ms/step; hipGraph replay "
                              f"{probe[key]:.1f} ms/step (probe)", file=sys.stderr, flush=True)
                    line = line_for(*best)
                    if probe[key] < 0.98 * best[0] / a.steps * 1e3:
                        dog.arm(f"timed run ({key})", a.stall_s + 2.0 * total, fallback=line)
                        el3, _ = timed_run(max(a.warmup, 1), a.steps, use_graph=True)
                        if el3 < best[0] and bool(torch.isfinite(st.lat).all()):
                            best = (el3, bplan, True)
                    line = line_for(*best)
            except Exception as ex:      # noqa: BLE001
                print(f"[bench] rank {rank}: {type(ex).__name__} while probing another plan: {ex} -- leaving with the "
                      f"line already measured", file=sys.stderr, flush=True)
                dog.abort(f"{type(ex).__name__} in a probe")      # peers leave at once (each with the same line on rank 0)
                if rank == 0:
                    print(line, flush=True)
                    time.sleep(1.5)                                # keep the store up for the peers' next poll
                os._exit(0)
        # ---- sec/clip, the metric's second half, on the ranks: ONE real `pipe(...)` call under the best plan -- VAE encodes of the
        # conditions (every rank: they are 0.25 s and nothing waits on a broadcast), the sharded loop, the VAE decode in `world`
        # slabs + one all-gather (round 6), post-processing.  The steps/s line above exists already: a stall or an error here costs
        # nothing but these keys.
        if not a.no_clip and not a.layers and not a.mxfp8 and not a.fp8_attention:
            try:
                dog.arm("measured clip", a.stall_s + 120.0, fallback=line)
                vae = bench_vae(a.workload, dev)
                clip = measured_clip(model, vae, cfg, dev, fg, lh, lw, steps=4 if a.workload == "tiny" else 50, plan=best[1],
                                     rank=rank, world=world)
                t_clip = max_over_ranks(clip["sec_per_clip_measured"] if clip["sec_per_clip_measured"] else -1.0)
                clip["sec_per_clip_measured"] = t_clip if t_clip > 0 else None
                clip["vae_decode_slabs"] = world
                clip_cfg.update(clip)
                line = line_for(*best)
                del vae
            except Exception as ex:      # noqa: BLE001
                print(f"[bench] rank {rank}: the measured clip failed: {type(ex).__name__}: {ex} -- the line goes out without it",
                      file=sys.stderr, flush=True)
                dog.abort(f"{type(ex).__name__} in the measured clip")
                if rank == 0:
                    print(line, flush=True)
                    time.sleep(1.5)
                os._exit(0)
        dog.disarm()
        if rank == 0:
            print(line, flush=True)
        if backend == "nccl" and not a.graph_probe and not a.no_graph_probe and (world == 1 or a.graph_probe_after_line):
            # Round 5: the best plan's step replayed from a hipGraph AFTER the line is out (default with one rank, opt-in with
            # more: --graph-probe-after-line).  The call patterns were probed through RCCL communicators of ONE rank
            # (frameino_amd/graph_step.py); with more ranks a capture has never run here, and a fault inside a capture cannot
            # be caught -- it cannot cost the line (the result goes to stderr), but it would cost the exit status.
            try:
                from frameino_amd.graph_step import groups_capturable
                bplan = best[1]
                if groups_capturable(bplan, explicit=True):
                    dog.arm(f"post-line probe ({bplan.desc}+hipgraph)", a.stall_s, ok=True)
                    shard_pipeline(pipe, rank, world, plan=bplan)
                    el_g, _ = timed_run(1, 3, use_graph=True)
                    if rank == 0:
                        print(f"[bench] after the line: {bplan.desc} replayed from a hipGraph {el_g / 3 * 1e3:.1f} ms/step "
                              f"(eager {best[0] / a.steps * 1e3:.1f}); finite={bool(torch.isfinite(st.lat).all())}",
                              file=sys.stderr, flush=True)
                    dog.disarm()
            except Exception as ex:      # noqa: BLE001
                print(f"[bench] rank {rank}: post-line hipGraph probe failed: {type(ex).__name__}: {ex}", file=sys.stderr,
                      flush=True)
        dog.arm("shutdown", 30.0, ok=True)           # the line is out: a hang in teardown is not a failed run
        dist.destroy_process_group()
        dog.disarm()
        return

    # ================================================================ N = 1: the headline + its evidence
    elapsed, timer = timed_run(a.warmup, a.steps, use_graph=a.graph, timer_names=() if a.graph else ("attn_self",))
    assert torch.isfinite(st.lat).all(), "non-finite latents"
    issued_flops(getattr(model, "dedup_shared_prefix", False) and not a.cfg_streams)
    ms_step = elapsed / a.steps * 1e3
    extra = {}
    heads, dh = cfg["num_attention_heads"], cfg["attention_head_dim"]

    secondary = {}
    gemm_classes = None
    if not a.no_secondary and a.workload == "wan2.2-5b-49f-704x1280" and not a.layers:
        # (a) the same step replayed from a captured hipGraph (north_star: the sampler loop is graph-captured)
        if not a.graph:
            el_g, _ = timed_run(1, 5, use_graph=True)
            extra["graph_ms_per_step"] = el_g / 5 * 1e3
        # (b) UniPC (the scheduler the released Wan2.2 folder ships) instead of Euler: corrector+predictor+CFG kernel
        extra["unipc_ms_per_step"] = unipc_ms_per_step(pipe, model, (lat, cond, traj, idl, mask, pe, ne), dev)
        # (b') the block GEMMs by epilogue class, HIP events around every launch of two more steps (their own run: the headline's
        # timed region carries events around the dominant kernel only) -> roofline.secondary
        _, kt_g = timed_run(1, 2, timer_names=("gemm_epi0", "gemm_epi1", "gemm_epi2", "gemm_epi3"))
        gemm_classes = gemm_class_rooflines(kt_g)
        # (c) attention on peaky logits (q x 4: the deferred-rescale branch fires on most tiles) next to the N(0,1) case
        secondary["attention_probe"] = attention_probe(ops, dev, L, heads, dh, a.logit_scale)
        st.lat.copy_(lat0)
        # (c'') the headline step with the DiT in fp16 -- the dtype the reference's canonical caller loads it in (app.py:156) and the
        # one the north star states its tolerance for; same seed, same inputs, same eager loop, right behind the headline run
        try:
            m16 = build_model(cfg, dev, dtype=torch.float16)
            pipe16 = WanImageToVideoPipeline(scheduler=FlowMatchEulerDiscreteScheduler(shift=5.0), transformer=m16,
                                             expand_timesteps=True)
            pipe16.use_hip_graph = False
            pipe16.scheduler.set_timesteps(max(total, 5), device=dev)
            r16 = other_workload_ms_per_step(pipe16, make_inputs, cfg, dev, a.workload, steps=3)
            extra["fp16_ms_per_step"] = r16["ms_per_step"]
            extra["fp16_over_bf16"] = r16["ms_per_step"] / ms_step
            del pipe16, m16
        except Exception as ex:      # noqa: BLE001   (a secondary measurement never costs the line)
            extra["fp16_ms_per_step"] = None
            extra["fp16_note"] = f"failed: {type(ex).__name__}: {ex}"
        torch.cuda.empty_cache()

    # ---- once-per-clip stages (outside the timed region): Wan VAE encode of the conditions + decode ----
    vae_times = None
    if a.workload != "tiny" and not a.no_vae:
        from frameino_amd.autoencoder_kl_wan import AutoencoderKLWan
        from frameino_amd.configs import WAN22_VAE_CFG
        vae = AutoencoderKLWan(**WAN22_VAE_CFG).random_init_(seed=2, device=dev)
        vid = torch.rand(1, 3, 1 + 4 * (fg - 1), lh * 16, lw * 16, device=dev) * 2 - 1
        with torch.no_grad():
            for rep in range(2):                          # first pass warms the kernels
                torch.cuda.synchronize(); t1 = time.perf_counter()
                vae.encode(vid).latent_dist.mode()        # trajectory video (49 frames)
                vae.encode(vid[:, :, :1]); vae.encode(vid[:, :, :1])   # first frame + ID frame
                torch.cuda.synchronize(); t2 = time.perf_counter()
                vae.decode(st.lat[None], return_dict=False)
                torch.cuda.synchronize(); t3 = time.perf_counter()
        vae_times = (t2 - t1, t3 - t2)
        del vid
        enc_s, dec_s = vae_times
        extra.update({"vae_encode_conditions_s": enc_s, "vae_decode_s": dec_s,
                      "sec_per_clip_50_steps": enc_s + 50 * ms_step / 1e3 + dec_s})
        # ---- the clip as ONE real call of the drop-in pipeline (what app.py:705-726 does), timed by the wall clock ----
        if not a.no_clip and not a.layers and not a.mxfp8 and not a.fp8_attention and a.workload.startswith("wan2.2-5b-49f"):
            try:
                extra.update(measured_clip(model, vae, cfg, dev, fg, lh, lw))
                if extra.get("sec_per_clip_measured"):
                    extra["sec_per_clip_measured_over_composed"] = extra["sec_per_clip_measured"] / extra["sec_per_clip_50_steps"]
            except Exception as ex:      # noqa: BLE001   (the headline line must not be lost to a secondary measurement)
                extra["sec_per_clip_measured"] = None
                extra["sec_per_clip_measured_note"] = f"failed: {type(ex).__name__}: {ex}"
            st.lat.copy_(lat0)
        extra["peak_device_memory_gib"] = torch.cuda.max_memory_allocated() / 2 ** 30
        # ---- the same VAE computing like the fp32 the reference app runs it in (app.py:157): split-bf16 products, opt-in
        # (`vae.set_compute_dtype(torch.float32)`); bf16 convolutions stay the default and are what the lines above time ----
        if not a.no_clip and not a.layers and a.workload.startswith("wan2.2-5b-49f"):
            try:
                vae.set_compute_dtype(torch.float32)
                vid = torch.rand(1, 3, 1 + 4 * (fg - 1), lh * 16, lw * 16, device=dev) * 2 - 1
                with torch.no_grad():
                    for rep in range(2):                      # first pass packs the weight planes and warms the kernels
                        torch.cuda.synchronize(); t1 = time.perf_counter()
                        vae.encode(vid).latent_dist.mode()
                        vae.encode(vid[:, :, :1]); vae.encode(vid[:, :, :1])
                        torch.cuda.synchronize(); t2 = time.perf_counter()
                        vae.decode(st.lat[None], return_dict=False)
                        torch.cuda.synchronize(); t3 = time.perf_counter()
                extra.update({"vae_encode_conditions_fp32_s": t2 - t1, "vae_decode_fp32_s": t3 - t2,
                              "sec_per_clip_50_steps_fp32_vae": (t2 - t1) + 50 * ms_step / 1e3 + (t3 - t2),
                              "vae_fp32_what": "vae.set_compute_dtype(torch.float32): fp32 activations, every convolution a "
                                               "split-bf16 product (3 bf16 planes per operand, 6 MFMA terms, fp32 accumulate): "
                                               "fp32-faithful (tests/test_wan_vae_gpu.py), opt-in"})
                del vid
                # ---- ... and the whole clip in the APP'S precision mix as ONE call (app.py:156-157: the DiT in fp16, the VAE in fp32 --
                # here its fp32-compute mode): what a user of the reference's app gets from the drop-in, timed by the wall clock ----
                if not a.no_secondary and not a.mxfp8 and not a.fp8_attention:
                    m16 = build_model(cfg, dev, dtype=torch.float16)
                    clip16 = measured_clip(m16, vae, cfg, dev, fg, lh, lw)
                    extra["sec_per_clip_measured_app_mix"] = clip16["sec_per_clip_measured"]
                    extra["sec_per_clip_measured_app_mix_stages"] = clip16["sec_per_clip_measured_stages"]
                    extra["sec_per_clip_measured_app_mix_what"] = ("the same pipe(...) call with the DiT in fp16 and the VAE computing like "
                                                                   "fp32 (set_compute_dtype(torch.float32)): app.py:156-157's precision mix")
                    del m16
                    torch.cuda.empty_cache()
            except Exception as ex:      # noqa: BLE001
                extra["vae_decode_fp32_s"] = None
                extra["vae_fp32_what"] = f"failed: {type(ex).__name__}: {ex}"
        del vae
        torch.cuda.empty_cache()

    if not a.no_secondary and a.workload == "wan2.2-5b-49f-704x1280" and not a.layers and not a.mxfp8 and not a.fp8_attention:
        # (c') the power-capped step's sensitivity to OPERAND BITS: the same step on all-zero weights (a floor: nothing
        # toggles) and on heavy-tailed weights, next to the N(0, 0.02^2) of the headline -- bounds how far the number can move
        # on a real checkpoint
        secondary["operand_sensitivity"] = operand_sensitivity(pipe, model, make_inputs, cfg, dev, a.workload, ms_step)
        # (d) BASELINE config 4's per-GPU-independent part: the same model at 1024x1792 (L = 25088), whole on one GPU
        secondary["config4_wan_1024x1792_L25088"] = other_workload_ms_per_step(
            pipe, make_inputs, cfg, dev, "wan2.2-5b-49f-1024x1792")
        # the app's default clip (81 frames 704x1280, app.py:543)
        secondary["app_default_81f_704x1280_L19360"] = other_workload_ms_per_step(
            pipe, make_inputs, cfg, dev, "wan2.2-5b-81f-704x1280")
        # the headline workload again with MXFP8 linears (e4m3 + e8m0 per 32; attention stays bf16): not the headline
        model.enable_mxfp8_linears()
        secondary["wan_704x1280_mxfp8_linears"] = other_workload_ms_per_step(pipe, make_inputs, cfg, dev, a.workload)
        # ... and with fp8 (e4m3) attention operands on top (fino_attn_fwd_fp8 at head_dim 128): the whole step on the fp8 MFMA path
        model.enable_fp8_attention()
        secondary["wan_704x1280_mxfp8_linears_fp8_attention"] = other_workload_ms_per_step(pipe, make_inputs, cfg, dev, a.workload)
        model.enable_fp8_attention(False)
        model.enable_mxfp8_linears(False)
        # (e) BASELINE config 5: CogVideoX-5B FrameINO 49f 480x720, bf16 and MXFP8 linears
        del pipe, st
        model.reset_caches()
        torch.cuda.empty_cache()
        secondary.update(config5_ms_per_step(dev))
    if secondary:
        extra["secondary"] = secondary
        # the same numbers as SCALAR config keys: the driver's record of the line keeps scalars only, and
        # the second backbone / config 4 / the fp8 legs are what a reader of BENCH_rNN.json looks for
        flat = {"config4_ms_per_step": "config4_wan_1024x1792_L25088", "app_81f_ms_per_step": "app_default_81f_704x1280_L19360",
                "wan_mxfp8_ms_per_step": "wan_704x1280_mxfp8_linears", "wan_fp8_ms_per_step": "wan_704x1280_mxfp8_linears_fp8_attention",
                "config5_bf16_ms_per_step": "config5_cogvideox5b_480x720_bf16", "config5_fp16_ms_per_step": "config5_cogvideox5b_480x720_fp16",
                "config5_mxfp8_ms_per_step": "config5_cogvideox5b_480x720_mxfp8_linears",
                "config5_fp8_ms_per_step": "config5_cogvideox5b_480x720_mxfp8_linears_fp8_attention"}
        for k, src in flat.items():
            if isinstance(secondary.get(src), dict) and "ms_per_step" in secondary[src]:
                extra[k] = secondary[src]["ms_per_step"]

    roofline = None
    ks = timer.summary().get("attn_self") if not a.graph else None
    if ks:
        traffic, traffic_src = profiled_traffic("attn_ppd_kernel<BF16, 128, 0>")
        # algorithmic FLOPs of the timed launches (4.Lq.Lk.H.Dh per batch element) / their summed duration
        total_fl = timer.flops["attn_self"]
        ach = total_fl / (ks["total_ms"] * 1e-3) / 1e12
        # Q, K, V read + O written, bf16, summed over the launches that were timed (29 of a forward's 30 cover both CFG
        # branches, layer 0's covers one: 1.967 batch elements per launch on average) / the number of launches
        alg_bytes = timer.bytes["attn_self"] / ks["launches"]
        batch = alg_bytes / (4 * L * heads * dh * 2)
        roofline = {"bound": "mfma", "kernel": "attn_ppd_kernel<BF16,128,0> (3D self-attention, head_dim 128)",
                    "achieved": ach, "peak": 2500.0, "unit": "TFLOP/s", "frac": ach / 2500.0,
                    # what the matrix pipe ALONE sustains on THIS box on gaussian operands before the board's power cap
                    # takes the clock down: measured in this run (fino_diag_mfma_peak, 32x32x16 bf16, ~0.25 s launches)
                    **measured_mfma_peak(dev, ach),
                    "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
                    "algorithmic_bytes": alg_bytes,
                    "traffic_over_algorithmic": None if traffic is None else traffic / alg_bytes,
                    "traffic_note": "K/V of one head (6.3 MB) exceed an XCD's 4 MiB L2 and are partly re-fetched; the "
                                    "kernel is MFMA-bound (traffic / duration = 0.4 TB/s of 8)",
                    "launches": ks["launches"], "avg_us": ks["avg_us"],
                    "flops_per_launch": total_fl / ks["launches"], "batch_per_launch": round(batch, 4),
                    "launch_mix": "per batch-2 forward 29 launches cover both CFG branches; layer 0's (identical for "
                                  "the branches) covers one: achieved = summed algorithmic FLOPs / summed durations"}
    if roofline is not None and gemm_classes:
        roofline["secondary"] = gemm_classes
    cpu = None if a.no_cpu_baseline else cpu_baseline(cfg, L)
    if cpu is not None and a.workload != "tiny":
        try:
            cpu.update(cpu_config1())
        except Exception as ex:      # noqa: BLE001
            cpu["config1_s"] = None
            cpu["config1_sample"] = f"failed: {type(ex).__name__}: {ex}"
    print(result_line(elapsed, "single", extra, roofline, cpu, use_graph=a.graph), flush=True)


GEMM_CLASSES = {                # epilogue id of gemm_pp_kernel<T, EPI, ...> -> what the Wan block uses it for
    0: "q|k|v projection + cross-attention q projection (bias epilogue; + patch embedding, output head)",
    1: "FFN up + GELU-tanh",
    2: "text cross-attention out-projection, re-associated P.(V W_o^T) (K = heads x keys; residual epilogue)",
    3: "gated-residual pair: self-attention out-projection + FFN down",
}


def gemm_class_rooflines(kt):
    """roofline.secondary: the block GEMMs (60 % of the step) by epilogue class.  achieved = summed algorithmic FLOPs (2.M.N.K) /
    summed launch durations from HIP events in this run; traffic = HBM-side bytes per launch of the same kernels from the committed
    PMC summary (FETCH_SIZE x 2 + WRITE_SIZE, KiB -- MI355X_MICROARCH.md), averaged over every dispatch of the epilogue class;
    algorithmic bytes = A + W read, C written (+ residual read), averaged over the same launch mix."""
    out = []
    summ = kt.summary()
    for epi, what in GEMM_CLASSES.items():
        ks = summ.get(f"gemm_epi{epi}")
        if not ks:
            continue
        fl, by = kt.flops[f"gemm_epi{epi}"], kt.bytes[f"gemm_epi{epi}"]
        ach = fl / (ks["total_ms"] * 1e-3) / 1e12
        traffic, src = profiled_traffic(f"gemm_pp_kernel<BF16, {epi}, ", combine=True)
        alg = by / ks["launches"]
        out.append({"kernel": f"gemm_pp_kernel<BF16,{epi},...>", "what": what, "bound": "mfma", "achieved": ach, "peak": 2500.0,
                    "unit": "TFLOP/s", "frac": ach / 2500.0, "launches": ks["launches"], "avg_us": ks["avg_us"],
                    "ms_per_step": ks["total_ms"] / 2, "algorithmic_bytes": alg, "traffic": traffic,
                    "traffic_over_algorithmic": None if traffic is None else traffic / alg, "traffic_source": src})
    return out


def bench_vae(workload, dev):
    """the Wan2.2 VAE with seeded random weights (no checkpoints offline); `tiny`: a 4-channel VAE of the same structure, so that the
    one-GPU rehearsals of the N > 1 flow run the whole `pipe(...)` call too"""
    from frameino_amd.autoencoder_kl_wan import AutoencoderKLWan
    from frameino_amd.configs import WAN22_VAE_CFG
    if workload == "tiny":
        return AutoencoderKLWan(base_dim=32, decoder_base_dim=32, z_dim=4, dim_mult=[1, 2, 4, 4], num_res_blocks=1,
                                temperal_downsample=[False, True, True], is_residual=True, in_channels=12, out_channels=12,
                                patch_size=2, scale_factor_temporal=4, scale_factor_spatial=16, latents_mean=[0.0] * 4,
                                latents_std=[1.0] * 4).random_init_(seed=2, device=dev)
    return AutoencoderKLWan(**WAN22_VAE_CFG).random_init_(seed=2, device=dev)


def measured_clip(model, vae, cfg, dev, fg, lh, lw, steps=50, plan=None, rank=0, world=1):
    """ONE real call of the drop-in pipeline at the headline workload, wall clock around `pipe(...)` (the reference's caller:
    app.py:705-726): PIL canvas + trajectory video + identity reference + prompt embeddings -> host preprocessing, the three
    VAE encodes, 50 denoise steps with the sampler the released Wan2.2 folder ships (UniPC) replayed from the captured hipGraph
    (the pipeline's default loop), VAE decode, `output_type="np"` (clamp, permute, device -> host).  The text encoder is not part
    of the call (prompt_embeds given: UMT5-XXL is a third-party model in front of the path, SURVEY 8d "text-encode excluded").
    Returns `sec_per_clip_measured` and its stage split (synchronize + wall-clock pairs around each stage: they add up)."""
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    from frameino_amd.conditions import prepare_traj_tensor
    from frameino_amd.pipeline_wan_i2v_motion_frameino import WanImageToVideoPipeline
    from frameino_amd.schedulers import UniPCMultistepScheduler
    from run_wan_frameino import synthetic_conditions
    H, W, F = lh * 16, lw * 16, 1 + 4 * (fg - 1)
    pipe = WanImageToVideoPipeline(vae=vae, scheduler=UniPCMultistepScheduler(flow_shift=5.0), transformer=model,
                                   expand_timesteps=True)
    if plan is not None:
        # N > 1 (round 6): the same call on every rank of the plan -- token-sharded denoise loop, and the VAE decode sharded too
        # (every rank decodes a slab of the frame, one all-gather of video rows: frameino_amd/parallel.py::sharded_vae_decode)
        from frameino_amd.parallel import shard_pipeline
        shard_pipeline(pipe, rank, world, plan=plan)
    canvas, tracks, id_tensor, _ = synthetic_conditions(F, H, W, dev)
    traj = prepare_traj_tensor(tracks, H, W, 6, W, H, device=dev)
    g = torch.Generator().manual_seed(1234)
    pe = torch.randn(1, 512, cfg["text_dim"], generator=g)
    pe[:, 64:] = 0
    ne = torch.randn(1, 512, cfg["text_dim"], generator=g)
    ne[:, 8:] = 0
    times = {}

    def wrap(obj, name, label):
        fn = getattr(obj, name)

        def timed(*args, **kwargs):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = fn(*args, **kwargs)
            torch.cuda.synchronize()
            times[label] = times.get(label, 0.0) + time.perf_counter() - t0
            return out
        setattr(obj, name, timed)

    wrap(pipe.video_processor, "preprocess", "preprocess_s")
    wrap(pipe, "_prepare_conditions", "vae_encode_conditions_s")
    wrap(pipe, "denoise", "denoise_s")
    wrap(pipe.vae, "decode", "vae_decode_s")
    wrap(pipe.video_processor, "postprocess_video", "postprocess_to_host_s")
    try:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        frames = pipe(image=canvas, traj_tensor=traj, ID_tensor=id_tensor, height=H, width=W, num_frames=F,
                      num_inference_steps=steps, guidance_scale=5.0, generator=torch.Generator().manual_seed(1234),
                      prompt_embeds=pe.to(dev), negative_prompt_embeds=ne.to(dev), output_type="np").frames[0]
        torch.cuda.synchronize()
        total = time.perf_counter() - t0
    finally:
        del pipe.vae.decode                 # the wrappers are instance attributes: the VAE object goes back as it came
    ok = tuple(frames.shape) == (F, H, W, 3) and bool((frames == frames).all())
    return {"sec_per_clip_measured": total if ok else None,
            "sec_per_clip_measured_stages": dict({k: round(v, 4) for k, v in times.items()},
                                                 other_s=round(total - sum(times.values()), 4),
                                                 denoise_ms_per_step=times.get("denoise_s", 0.0) / steps * 1e3),
            "sec_per_clip_measured_what": f"one pipe(...) call, {F} frames {H}x{W}, {steps} steps, UniPC, "
                                          + ("default hipGraph replay (step 0 eager, step 1 captured)" if plan is None else
                                             f"plan {plan.desc} (eager loop), VAE decode in {world} slabs + one all-gather")
                                          + f", prompt_embeds given, 3 VAE encodes + decode in the "
                                          f"VAE's compute dtype, output_type='np' on the host; frames {tuple(frames.shape)} finite={ok}"}


def measured_mfma_peak(dev, achieved_tflops):
    """`power_capped_peak`: dense bf16 MFMA rate of the whole chip with ONLY the matrix pipe working (every wave issues
    independent v_mfma_f32_32x32x16_bf16 from registers, 2 waves per SIMD, gaussian operand bits), over launches long
    enough (~0.25 s each, 3 of them after one warm launch) for the power manager to settle.  A measurement of this box
    in this run -- never a constant from another one; None if the diagnostic fails."""
    import ctypes
    from frameino_amd import _lib
    try:
        lib = _lib.lib()
        scratch = torch.zeros(64 + 2 * 256 * 4, device=dev)
        g = torch.Generator(device=dev).manual_seed(0)
        ops_view = scratch[64:].view(torch.bfloat16)
        ops_view.copy_(torch.randn(ops_view.shape, device=dev, generator=g).bfloat16())
        fl = ctypes.c_double()
        stream = torch.cuda.current_stream().cuda_stream

        def run(iters):
            _lib.check(lib.fino_diag_mfma_peak(0, 2, iters, scratch.data_ptr(), ctypes.byref(fl), stream),
                       "fino_diag_mfma_peak")
        run(100)
        run(400000)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            run(400000)
        e1.record()
        torch.cuda.synchronize()
        tf = fl.value / (e0.elapsed_time(e1) / 3 * 1e-3) / 1e12
        return {"power_capped_peak": tf, "frac_of_power_capped_peak": achieved_tflops / tf,
                "power_capped_peak_source": "measured in this run: fino_diag_mfma_peak, 32x32x16 bf16, gaussian operands, "
                                            "2 waves/SIMD, 3 launches of 4e5 loop iterations x 16 MFMAs = 6.4e6 MFMAs per "
                                            "wave (~0.25 s each)"}
    except Exception as ex:      # noqa: BLE001
        return {"power_capped_peak": None, "frac_of_power_capped_peak": None,
                "power_capped_peak_source": f"not measured ({type(ex).__name__}: {ex})"}


def cpu_config1(budget_s=12.0):
    """BASELINE config 1 (CogVideoX-I2V-5B stage-1 pipeline, 13 frames 256x256, 10 steps, fp32 on
    the host) through the oracle.  All of it is ~2.5e14 FLOP -- an hour on host cores -- so the timed SAMPLE is one
    denoise step (B = 2 forward + CFG + DDIM) of the full-width model with 2 of its 42 layers; `config1_s` extrapolates
    it by layers and steps and says so."""
    from oracle import cog_pipeline as CP
    from frameino_amd.cogvideox_transformer_3d import CogVideoXTransformer3DModel
    from frameino_amd.configs import COGVIDEOX_5B_FRAMEINO_CFG as COG5B
    from frameino_amd.pipeline_cogvideox_i2v_motion import CogVideoXImageToVideoPipeline
    threads = min(os.cpu_count() or 1, 64)           # L = 1250 rows: more threads than that only fight over the caches
    torch.set_num_threads(threads)
    cfg = dict(COG5B, use_FrameIn=False, num_layers=2)
    g = torch.Generator().manual_seed(1)
    m = CogVideoXTransformer3DModel(**cfg)                 # the mirror only as a container of reference-keyed parameters
    with torch.no_grad():
        for name, p_ in m.named_parameters():
            p_.copy_(torch.randn(p_.shape, generator=g) * (0.02 if p_.ndim > 1 else 0.1) +
                     (1.0 if name.endswith("norm.weight") or "norm_q.weight" in name or "norm_k.weight" in name else 0.0))
        m.patch_embed.pos_embedding.copy_(torch.randn(m.patch_embed.pos_embedding.shape, generator=g) * 0.1)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    rot = CogVideoXImageToVideoPipeline(transformer=m, scheduler=None)._prepare_rotary_positional_embeddings(256, 256, 4, "cpu")
    del m
    F_, C_, h, w = 4, 16, 32, 32
    lat = torch.randn(1, F_, C_, h, w, generator=g)
    img = torch.cat([torch.randn(1, 1, C_, h, w, generator=g), torch.zeros(1, F_ - 1, C_, h, w)], 1)
    trj = torch.randn(1, F_, C_, h, w, generator=g)
    pe, ne = torch.randn(1, 226, 4096, generator=g), torch.randn(1, 226, 4096, generator=g)
    t0 = time.time()
    nst = 1
    CP.cog_denoise_loop(sd, cfg, lat, img, trj, None, pe, ne, rot, 6.0, nst)
    t_step2 = (time.time() - t0) / nst
    layers = COG5B["num_layers"]
    return {"config1_s": t_step2 * layers / 2 * 10,
            "config1_sample": f"oracle stage-1 CogVideoX pipeline, fp32, 13 f 256x256 (L = 226 + 1024, B = 2), full width, "
                              f"{nst} step with 2 of {layers} layers on {threads} threads: {t_step2:.2f} s; extrapolated "
                              f"x {layers // 2} (layers) x 10 (steps)"}


def unipc_ms_per_step(pipe_euler, model, inputs, dev, steps=3):
    """The same step with the UniPC multistep update (corrector + predictor + CFG in one kernel, two extra fp32 history
    buffers) instead of Euler -- what examples/run_wan_frameino.py runs with the released scheduler config."""
    from frameino_amd.pipeline_wan_i2v_motion_frameino import WanImageToVideoPipeline
    from frameino_amd.schedulers import UniPCMultistepScheduler
    pipe = WanImageToVideoPipeline(scheduler=UniPCMultistepScheduler(flow_shift=5.0), transformer=model,
                                   expand_timesteps=True)
    pipe.scheduler.set_timesteps(steps + 2, device=dev)
    st = pipe.make_state(*inputs, 5.0)
    coefs = pipe.scheduler.coefs.to(dev).clone()
    coefs[:, 0] = 5.0
    ts = pipe.scheduler.timesteps.to(dev).float()
    with torch.no_grad():
        for i in range(steps + 1):
            if i == 1:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            st.t_rows[1:2].copy_(ts[i:i + 1])
            st.coef.copy_(coefs[i])
            pipe._step(st)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def operand_sensitivity(pipe, model, make_inputs, cfg, dev, workload, headline_ms):
    """ms/step of the headline workload with other WEIGHT statistics (same shapes, same kernels, same FLOPs; 1 warm + 2 timed
    steps each): under the board's power cap the clock follows how many operand bits toggle, so the step time is a function
    of the data.  `zero_weights`: every large matrix 0 (the W operand of every block GEMM is zeros and what they feed
    collapses to biases: a floor no checkpoint reaches).  `heavy_tailed_weights`: the N(0, 0.02^2) weights with a log-normal (sigma 1) gain per output
    channel and 0.1 % of the elements x 16 -- outlier channels and elements as trained transformers have them.  The
    parameters are restored afterwards (bit-exact: saved copies)."""
    out = {"n(0,0.02^2)_headline": {"ms_per_step": headline_ms}}
    big = [p_ for p_ in model.parameters() if p_.ndim == 2 and p_.numel() >= 1 << 20]
    saved = [p_.detach().clone() for p_ in big]
    g = torch.Generator(device=dev).manual_seed(99)

    def run(tag):
        model.reset_caches()
        r = other_workload_ms_per_step(pipe, make_inputs, cfg, dev, workload)
        out[tag] = {"ms_per_step": r["ms_per_step"], "vs_headline": r["ms_per_step"] / headline_ms}

    try:
        with torch.no_grad():
            for p_ in big:
                p_.zero_()
            run("zero_weights")
            for p_, sv in zip(big, saved):
                gain = torch.exp(torch.randn(p_.shape[0], 1, device=dev, generator=g))
                spike = 1.0 + 15.0 * (torch.rand(p_.shape, device=dev, generator=g) < 1e-3)
                p_.copy_((sv.float() * gain * spike).to(p_.dtype))
                del gain, spike
            run("heavy_tailed_weights")
    finally:
        with torch.no_grad():
            for p_, sv in zip(big, saved):
                p_.copy_(sv)
        del saved
        model.reset_caches()
        torch.cuda.empty_cache()
    return out


def attention_probe(ops, dev, L, heads, dh, logit_scale):
    """The dominant kernel alone at the bench shape (batch 2), on N(0,1) q/k (flat softmax rows: the deferred-rescale
    branch almost never fires) and on peaky logits (q x 4, or --logit-scale: row maxima jump tile to tile)."""
    d = heads * dh
    out = {}
    g = torch.Generator(device=dev).manual_seed(7)
    kv = torch.randn(2, L, 2 * d, device=dev, generator=g).bfloat16()
    q0 = torch.randn(2, L, d, device=dev, generator=g)
    for name, sc in (("q_x1", 1.0), ("q_x4_peaky", 4.0 if logit_scale == 1.0 else logit_scale)):
        q = (q0 * sc).bfloat16()
        o = torch.empty_like(q)
        for _ in range(2):
            ops.attention(q, kv[:, :, :d], kv[:, :, d:], heads, out=o)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.attention(q, kv[:, :, :d], kv[:, :, d:], heads, out=o)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        out[name] = {"avg_us": us, "tflops": 4.0 * 2 * L * L * d / (us * 1e-6) / 1e12, "logit_scale": sc}
    return out


def other_workload_ms_per_step(pipe, make_inputs, cfg, dev, workload, steps=2):
    """Another clip size through the same pipeline on ONE GPU, 1 warm + `steps` timed steps: BASELINE config 4's clip
    (49 f 1024x1792 + ID frame, L = 25088: what the 8 ranks' shards add up to before the wire) and the app's default
    81-frame clip (L = 19360)."""
    fg, lh, lw = WORKLOADS[workload]
    inputs = make_inputs(fg, lh, lw, cfg["text_dim"])
    st = pipe.make_state(*inputs, 5.0)
    ts, dts = pipe.scheduler.timesteps.to(dev).float(), pipe.scheduler.dts.to(dev)
    with torch.no_grad():
        for i in range(steps + 1):
            if i == 1:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            st.t_rows[1:2].copy_(ts[i:i + 1])
            st.dt.copy_(dts[i:i + 1])
            pipe._step(st)
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    L = (fg + 1) * (lh // 2) * (lw // 2)
    return {"ms_per_step": ms, "tokens": L, "steps": steps,
            "model_tflops_per_s": 2 * wan_flops_per_forward(L, cfg) / (ms * 1e-3) / 1e12}


def config5_ms_per_step(dev, steps=2):
    """BASELINE config 5: CogVideoX-5B FrameINO, 49 f 480x720 -> model input [2, 14, 48, 60, 90], L = 226 + 18900, 42
    layers, 48 heads x 64; one step = the B=2 forward + guidance + v-prediction DDIM update
    (pipelines/pipeline_cogvideox_i2v_motion_FrameINO.py:848-944).  bf16, then MXFP8 linears (+ bf16 attention), then MXFP8
    linears + fp8 (e4m3) attention operands: the "fp8 MFMA path" end to end."""
    from frameino_amd.configs import COGVIDEOX_5B_FRAMEINO_CFG as COG5B
    from frameino_amd.pipeline_cogvideox_i2v_motion_frameino import CogVideoXImageToVideoPipeline
    from frameino_amd.random_init import random_cog_model
    from frameino_amd.schedulers import CogVideoXDDIMScheduler
    m = random_cog_model(dict(COG5B), dev)
    pipe = CogVideoXImageToVideoPipeline(transformer=m, scheduler=CogVideoXDDIMScheduler())
    g = torch.Generator(device=dev).manual_seed(1)
    F_, C_, h, w = 13, 16, 60, 90
    lat = torch.randn(1, F_, C_, h, w, device=dev, generator=g)
    img = torch.cat([torch.randn(1, 1, C_, h, w, device=dev, generator=g), torch.zeros(1, F_ - 1, C_, h, w, device=dev)], 1)
    trj = torch.randn(1, F_, C_, h, w, device=dev, generator=g)
    idl = torch.randn(1, 1, C_, h, w, device=dev, generator=g)
    pe = torch.randn(1, 226, 4096, device=dev, generator=g)
    ne = torch.randn(1, 226, 4096, device=dev, generator=g)
    L, d, nl = 226 + 14 * 30 * 45, 3072, COG5B["num_layers"]
    flops = 2 * nl * (8 * L * d * d + 4 * L * L * d + 16 * L * d * d)          # B=2: proj + SDPA + FFN (4x)
    out = {}
    for key, fp8, fp8_attn in (("config5_cogvideox5b_480x720_bf16", False, False),
                               ("config5_cogvideox5b_480x720_mxfp8_linears", True, False),
                               ("config5_cogvideox5b_480x720_mxfp8_linears_fp8_attention", True, True)):
        if fp8:
            m.enable_mxfp8_linears()
        m.enable_fp8_attention(fp8_attn)
        seen = []

        def cb(p, i, t, kw):
            torch.cuda.synchronize()
            seen.append(time.perf_counter())
            return {}

        res = pipe.denoise(lat, img, trj, idl, pe, ne, 6.0, steps + 1, callback_on_step_end=cb)
        assert torch.isfinite(res.float()).all()
        ms = (seen[-1] - seen[0]) / steps * 1e3
        out[key] = {"ms_per_step": ms, "denoise_steps_per_s": 1e3 / ms, "tokens": L, "steps": steps,
                    "model_tflops_per_s": flops / (ms * 1e-3) / 1e12}
    # the same backbone all-fp16, as test_code/run_cogvideox_FrameIn_mass_evaluation.py:92 loads it
    try:
        del pipe, m
        torch.cuda.empty_cache()
        m = random_cog_model(dict(COG5B), dev, dtype=torch.float16)
        pipe = CogVideoXImageToVideoPipeline(transformer=m, scheduler=CogVideoXDDIMScheduler())
        seen = []
        res = pipe.denoise(lat, img, trj, idl, pe, ne, 6.0, steps + 1, callback_on_step_end=cb)
        assert torch.isfinite(res.float()).all()
        ms = (seen[-1] - seen[0]) / steps * 1e3
        out["config5_cogvideox5b_480x720_fp16"] = {"ms_per_step": ms, "denoise_steps_per_s": 1e3 / ms, "tokens": L, "steps": steps,
                                                   "model_tflops_per_s": flops / (ms * 1e-3) / 1e12}
    except Exception as ex:      # noqa: BLE001
        out["config5_cogvideox5b_480x720_fp16"] = {"failed": f"{type(ex).__name__}: {ex}"}
    return out


def profiled_traffic(kernel_substr, combine=False):
    """HBM bytes per launch of the dominant kernel from the committed PMC summary of this same command
    (tools/profile_bench.sh: separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes; counters are in KiB and
    gfx950's FETCH_SIZE under-counts by 2x -- MI355X_MICROARCH.md, HBM section).  PMC passes cannot run inside the
    timed bench, so this is the last profiled value, or None when no summary is in the tree."""
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    for f in sorted(glob.glob(os.path.join(here, "profiles", "r*_summary.json")), reverse=True):
        try:
            pmc = json.load(open(f)).get("pmc_avg_per_dispatch", {})
        except (OSError, ValueError):
            continue
        tot, disp = 0.0, 0
        for name, c in pmc.items():
            if kernel_substr in name and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                if not combine:
                    return (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0, "profiles/" + os.path.basename(f)
                n_ = c.get("dispatches", 1)           # combine: dispatch-weighted mean over every kernel that matches
                tot += (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 * n_
                disp += n_
        if disp:
            return tot / disp, "profiles/" + os.path.basename(f)
    return None, None


if __name__ == "__main__":
    main()
