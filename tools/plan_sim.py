#!/usr/bin/env python3
"""Compute-side cost of the multi-GPU plans on ONE GPU: the per-rank work of an N-GPU run with the collectives replaced
by local copies (a fake TokenShard), full Wan2.2-5B size.  split(N): one CFG branch, 1/(N/2) of the tokens;
interleave(N): both branches, 1/N of the tokens each, advanced alternately on two streams.  Add the K|V all-gather
time of the node by hand -- this measures what the GPU has to do, not the wire; or model the wire's
DURATION with FINO_PLAN_SIM_WIRE_GBPS (FakeShard) to see which plans hide it."""
import os
import sys
import time
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import build_model  # noqa: E402
from frameino_amd.parallel import TokenShard  # noqa: E402


class _WireWork:
    """stands for c10d's work handle: wait() makes the CURRENT stream wait for the modelled wire"""

    def __init__(self, ev):
        self.ev = ev

    def wait(self):
        torch.cuda.current_stream().wait_event(self.ev)


class FakeShard(TokenShard):
    """rank 0 of `ways`; the all-gather copies the local block into its slot (the other slots keep old data).

    FINO_PLAN_SIM_WIRE_GBPS=<GB/s into one rank, all links together> (round 5): the exchange then also TAKES TIME -- a spin kernel
    of bytes_received / rate on a stream of its own (one per communicator, as c10d keeps one), after which the copy runs; the
    forward's work.wait() waits for it.  One GPU cannot show what xGMI delivers; this shows which plans HIDE an exchange of a given
    duration behind the other branch's kernels and which stand and wait for it (1 WG spinning: it takes no CU from the GEMMs)."""

    wire_gbps = float(os.environ.get("FINO_PLAN_SIM_WIRE_GBPS", "0") or 0)
    if os.environ.get("FINO_PLAN_SIM_KV_GROUPS"):          # A/B: head groups of the K|V all-gather (TokenShard.kv_head_groups)
        kv_head_groups = int(os.environ["FINO_PLAN_SIM_KV_GROUPS"])
    _spin_per_us = None

    def _wire(self, nbytes, cur_tensor, copy_fn, async_op):
        if not self.wire_gbps:
            copy_fn()
            return None
        if FakeShard._spin_per_us is None:                 # calibrate torch.cuda._sleep's unit once
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda._sleep(1000000)
            e0.record(); torch.cuda._sleep(20000000); e1.record()
            torch.cuda.synchronize()
            FakeShard._spin_per_us = 20000000 / (e0.elapsed_time(e1) * 1e3)
        if getattr(self, "_wire_stream", None) is None:
            self._wire_stream = torch.cuda.Stream()
        cur = torch.cuda.current_stream()
        iss = self.issue_stream                       # as TokenShard._issue: the step's own stream issues, the branch's stream waits
        if iss is not None and iss != cur:            # (a stream forked from a SIDE stream inside a capture segfaults in
            iss.wait_stream(cur)                      # hipStreamEndCapture on this image -- the same runtime limit as c10d's)
            cur = iss
        self._wire_stream.wait_stream(cur)
        with torch.cuda.stream(self._wire_stream):
            torch.cuda._sleep(int(nbytes / (self.wire_gbps * 1e3) * FakeShard._spin_per_us))      # bytes / (GB/s) = ns -> us
            copy_fn()
            ev = torch.cuda.Event()
            ev.record()
        work = _WireWork(ev)
        if not async_op:
            work.wait()
            return None
        return work

    def _all_gather(self, key, t, async_op):
        out = self._get(key, (self.ways * t.shape[0],) + tuple(t.shape[1:]), t.dtype, t.device)
        work = self._wire((self.ways - 1) * t.numel() * t.element_size(), t, lambda: out[:t.shape[0]].copy_(t), async_op)
        return out, work

    def all_to_all(self, key, send, async_op=False):   # heads exchange: every slice "arrives" as a copy of what was sent
        recv = self._get(key, tuple(send.shape), send.dtype, send.device)
        work = self._wire((self.ways - 1) * send[0].numel() * send.element_size(), send, lambda: recv.copy_(send), async_op)
        return recv, work


def main():
    from frameino_amd.pipeline_wan_i2v_motion_frameino import WanImageToVideoPipeline
    from frameino_amd.schedulers import FlowMatchEulerDiscreteScheduler
    from frameino_amd.configs import WAN22_5B_CFG
    dev = torch.device("cuda")
    cfg = dict(WAN22_5B_CFG)
    model = build_model(cfg, dev)
    pipe = WanImageToVideoPipeline(scheduler=FlowMatchEulerDiscreteScheduler(shift=5.0), transformer=model,
                                   expand_timesteps=True)
    g = torch.Generator().manual_seed(1234)
    C, fg, lh, lw = 48, 13, 44, 80
    lat = torch.randn(1, C, fg, lh, lw, generator=g).to(dev)
    cond = torch.randn(1, C, 1, lh, lw, generator=g).to(dev)
    traj = torch.randn(1, C, fg + 1, lh, lw, generator=g).to(dev)
    idl = torch.randn(1, C, 1, lh, lw, generator=g).to(dev)
    mask = torch.ones(1, 1, fg, lh, lw, device=dev)
    mask[:, :, 0] = 0
    pe = torch.randn(1, 512, cfg["text_dim"], generator=g)
    ne = torch.randn(1, 512, cfg["text_dim"], generator=g)
    if not os.environ.get("FINO_PLAN_SIM_UNPADDED"):
        # round 5: the prompts of bench.py's headline workload -- 64 / 8 tokens zero-padded to 512 rows, as the reference pads
        # every prompt (pipeline_wan_i2v_motion_FrameINO.py:235-238) -- so that a simulated rank runs the text branch the bench
        # step runs (padding run folded, out-projection re-associated).  Rounds 2-4 simulated un-padded 512-token prompts
        # (FINO_PLAN_SIM_UNPADDED=1): their N = 1 line is ~2 % above the bench's for that reason.
        pe[:, 64:] = 0
        ne[:, 8:] = 0
    pe, ne = pe.to(dev).bfloat16(), ne.to(dev).bfloat16()
    pipe.scheduler.set_timesteps(8, device=dev)
    st = pipe.make_state(lat, cond, traj, idl, mask, pe, ne, 5.0)
    st.t_rows[1:2] = 700.0
    st.dt[0] = -0.01

    host_ms = [0.0]
    graph_txt = [""]
    want_graph = not os.environ.get("FINO_PLAN_SIM_NO_GRAPH")

    def timed(n=2):
        """ms per step; host_ms[0]: the host's share (time to ENQUEUE a step: if it is close to the step time, the rank is
        launch-bound, not GPU-bound).  Round 4: the same step captured into a hipGraph and replayed (what pipe.denoise()
        does by default): graph_txt[0] = its ms per step and host enqueue."""
        with torch.no_grad():
            pipe._step(st)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                pipe._step(st)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            eager = (time.perf_counter() - t0) / n * 1e3
            host_ms[0] = (t1 - t0) / n * 1e3
            graph_txt[0] = ""
            if want_graph:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    pipe._step(st)
                g.replay()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    g.replay()
                t1 = time.perf_counter()
                torch.cuda.synchronize()
                graph_txt[0] = (f"; hipGraph replay {(time.perf_counter() - t0) / n * 1e3:.1f} ms/step "
                                f"(host enqueue {(t1 - t0) / n * 1e3:.2f} ms)")
                del g
        return eager

    from frameino_amd import _lib
    if os.environ.get("FINO_PLAN_SIM_TILE_M"):      # A/B: 8 = 256-row GEMM tiles only (round 2), 2..7 = that height everywhere
        _lib.lib().fino_tune_set(3, int(os.environ["FINO_PLAN_SIM_TILE_M"]))
    forced_tile = bool(os.environ.get("FINO_PLAN_SIM_TILE_M"))

    def set_tiling(interleaved):          # what ParallelPlan does for a real plan (unless the A/B knob above is set):
        if not forced_tile and interleaved:          # a per-call tile height carried by the shards, no process state
            for sh_ in pipe.parallel.shards:
                sh_.gemm_tile_m = 8

    only = sys.argv[1] if len(sys.argv) > 1 else None     # e.g. "interleave:8" or "split-heads:4": that plan alone (profiling)
    if only:
        kind, ways = only.split(":")
        ways = int(ways)
        ex = "heads" if kind.endswith("heads") else "kv"
        if kind.startswith("interleave"):
            pipe.parallel = SimpleNamespace(interleave=True, cfg_ways=1, token_ways=ways,
                                            shards=(FakeShard(0, ways, exchange=ex), FakeShard(0, ways, exchange=ex)))
            set_tiling(True)
            for sh in pipe.parallel.shards:
                sh.head_groups = 1
                sh.fused_qkv = not os.environ.get("FINO_PLAN_SIM_NO_FUSED_QKV")      # as ParallelPlan sets it
                sh.kv_head_groups = int(os.environ.get("FINO_PLAN_SIM_KV_GROUPS", "1"))
            model.parallel = pipe.parallel.shards[0]
        else:
            pipe.parallel = SimpleNamespace(interleave=False, cfg_ways=2, cfg_idx=0, token_ways=ways,
                                            exchange_cfg=lambda mine: (mine, mine))
            set_tiling(False)
            model.parallel = FakeShard(0, ways, exchange=ex) if ways > 1 else None
        print(f"{only}: {timed(3):.1f} ms/step of GPU work per rank (host enqueue {host_ms[0]:.1f} ms){graph_txt[0]}")
        return
    print(f"N=1 (batch-2 forward): {timed():.1f} ms/step (host enqueue {host_ms[0]:.1f} ms){graph_txt[0]}")
    for ways in (1, 2, 4):          # split plans: one branch per rank, token_ways = N/2
        n_gpus = 2 * ways
        sh = FakeShard(0, ways)
        pipe.parallel = SimpleNamespace(interleave=False, cfg_ways=2, cfg_idx=0, token_ways=ways,
                                        exchange_cfg=lambda mine: (mine, mine))
        set_tiling(False)
        model.parallel = sh if ways > 1 else None
        print(f"split      N={n_gpus}: cfg2 x token{ways}: {timed():.1f} ms/step of GPU work per rank (host enqueue {host_ms[0]:.1f} ms){graph_txt[0]}")
    for ways in (2, 4, 8):          # interleaved plans: both branches per rank, token_ways = N
        pipe.parallel = SimpleNamespace(interleave=True, cfg_ways=1, token_ways=ways,
                                        shards=(FakeShard(0, ways), FakeShard(0, ways)))
        set_tiling(True)
        for sh in pipe.parallel.shards:
            sh.fused_qkv = True                 # as ParallelPlan sets it for the interleaved plan
            sh.kv_head_groups = int(os.environ.get("FINO_PLAN_SIM_KV_GROUPS", "1"))
        model.parallel = pipe.parallel.shards[0]
        print(f"interleave N={ways}: 2 branches x token{ways}: {timed():.1f} ms/step of GPU work per rank (host enqueue {host_ms[0]:.1f} ms){graph_txt[0]}")
    # the same plans with the heads exchange (all-to-all instead of the K|V all-gather; attention over H/ways heads x all tokens)
    for ways in (2, 4):
        sh = FakeShard(0, ways, exchange="heads")
        pipe.parallel = SimpleNamespace(interleave=False, cfg_ways=2, cfg_idx=0, token_ways=ways,
                                        exchange_cfg=lambda mine: (mine, mine))
        set_tiling(False)
        model.parallel = sh
        print(f"split-heads      N={2 * ways}: cfg2 x token{ways}: {timed():.1f} ms/step of GPU work per rank (host enqueue {host_ms[0]:.1f} ms){graph_txt[0]}")
    for ways in (2, 4, 8):
        pipe.parallel = SimpleNamespace(interleave=True, cfg_ways=1, token_ways=ways,
                                        shards=(FakeShard(0, ways, exchange="heads"), FakeShard(0, ways, exchange="heads")))
        set_tiling(True)
        for sh in pipe.parallel.shards:
            sh.head_groups = 1                  # as ParallelPlan sets it for the interleaved plan
        model.parallel = pipe.parallel.shards[0]
        print(f"interleave-heads N={ways}: 2 branches x token{ways}: {timed():.1f} ms/step of GPU work per rank (host enqueue {host_ms[0]:.1f} ms){graph_txt[0]}")


if __name__ == "__main__":
    main()
