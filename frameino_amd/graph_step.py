"""hipGraph replay of a sampler loop's per-step program (BASELINE.json north_star: "the 50-step sampler loop is
hipGraph-captured").

Both pipelines express one denoise step as `_step(st)` on static device buffers: no host sync, no data-dependent shape,
timestep / dt / coefficient rows refreshed by device-to-device copies between steps.  `StepGraph` runs such a program

    step 0      eagerly -- it is a REAL step of the loop, and it fills every lazy cache the program reads (text K/V of the
                30 cross-attention layers, workspaces, RoPE tables, side streams, send / receive buffers of the exchanges);
    step 1      is captured (`torch.cuda.CUDAGraph` = hipGraph: capture enqueues nothing) and then replayed;
    step 2 ...  are replays.

So capture costs one host enqueue pass, no wasted step.  Token-sharded steps: what this image (torch 2.10, HIP 7.0 runtime,
RCCL 2.26) captures was probed pattern by pattern on an MI355X (tools/debug/rccl_capture_probe.py,
profiles/r04_rccl_capture_probe.txt, one rank):

    all_gather_into_tensor issued on the CAPTURING stream (async_op or not)     captured, replayed, bit-equal
    the same collective on a side stream forked from the capturing stream      SIGSEGV inside hipStreamEndCapture
    all_to_all_single with async_op=True on the capturing stream               SIGSEGV inside hipStreamEndCapture
    all_to_all_single, blocking                                                 replays; the process group then hangs in teardown

Round 5 (profiles/r05_rccl_capture_probe.txt) -- what decides is the stream that is CURRENT when the collective is issued, not
where the kernels around it run:

    kernels on two side streams, no collective                                                  captured, replayed, bit-equal
    kernels on two side streams, all_gather_into_tensor(async_op=True) issued with the
      capturing stream current (it waits for the producer's event first), work.wait() on
      the side stream -- one communicator or two                                                captured, replayed, bit-equal
    the same with work.wait() on the capturing stream and the side stream forked again          captured, replayed, bit-equal

A segfault cannot be caught, so the rule is static: the K|V all-gather captures when it is issued on the step's own stream --
the split plan does that by construction, the interleaved plan (two side streams, two communicators) since round 5 through
`TokenShard.issue_stream`; tests/test_parallel_gpu.py replays both bit-equal through real RCCL communicators.

Round 6 (profiles/r06_rccl_capture_probe.txt) -- the heads all-to-all, every c10d entry point that moves its bytes:

    all_to_all_single / all_to_all (tensor lists) / unequal splits / batch_isend_irecv,
      async_op=True, capturing stream current (with the watchdog drained, thread_local mode)    SIGSEGV inside hipStreamEndCapture
    all_to_all_single / all_to_all (lists), SYNCHRONOUS, capturing stream current                 captured, replayed, bit-equal, also
                                                                                                on new operand data; the process
                                                                                                group's teardown hangs IF the graph
                                                                                                is still alive -- with the graph
                                                                                                destroyed first it returns
    all_gather_into_tensor of the send blocks + a local pick (ways x the bytes)                 captured, replayed, bit-equal

So a captured step issues the all-to-all synchronously on the step's own stream (`TokenShard._issue(sync_in_capture=True)`; the
eager step keeps the asynchronous form and its overlap), and `StepGraph.close()` drops the graph when the loop ends -- before
anyone can destroy the process group.  With more than one rank a capture has never run on real links here: it is
taken only when asked for (`use_hip_graph = True`).  A gloo exchange staged through host memory (tests) and a user callback
between steps cannot be captured either.

`mode`: None = automatic (graph when capturable; a failed capture falls back to the eager loop with a warning -- the same
HIP kernels either way), True = required (a failed capture raises), False = eager.
"""
import warnings

import torch


def groups_capturable(plan, explicit=False):
    """May the step of this parallel plan be captured?  (module docstring: the call patterns this image's runtime captures.)
    `explicit`: the caller asked for the graph (`use_hip_graph = True`) -- needed with more than one rank."""
    if plan is None:
        return True
    # (the interleaved plan's collectives are issued on the step's own stream -- TokenShard.issue_stream -- while the branches'
    # kernels run on two side streams: that pattern captures; round 5 probe.  Round 6: the heads all-to-all captures too, as a
    # SYNCHRONOUS collective on that stream -- TokenShard._issue(sync_in_capture=True) -- provided the graph is destroyed before
    # the process group: StepGraph.close)
    if plan.world > 1 and not explicit:
        return False                       # never run on real xGMI links here: opt-in
    import torch.distributed as dist
    groups = [plan.token_group, plan.cfg_group, getattr(plan, "token_group_b", None)]
    try:
        return all(g is None or dist.get_backend(g) == "nccl" for g in groups)      # gloo: staged through the host
    except (RuntimeError, ValueError):
        return False


def capture_error_mode(plan):
    """`capture_error_mode` of torch.cuda.graph for a step of this plan.  With a process group alive, c10d's watchdog THREAD polls
    the events of the eager collectives that ran just before the capture (hipEventQuery); under the default "global" mode any
    such call from another thread while a capture is open fails, the watchdog takes that for a communicator error and aborts the
    process (seen once in ~10 runs of the forced-shard rehearsal: SIGABRT in ProcessGroupNCCL::Watchdog::run).  "thread_local"
    confines the check to the capturing thread, which makes no such call."""
    return "thread_local" if plan is not None else "global"


def _process_groups():
    """Every process group this process holds (the default one and the plan's sub-groups)."""
    import torch.distributed as dist
    try:
        return list(dist.distributed_c10d._world.pg_map.keys())
    except AttributeError:
        return [dist.group.WORLD]


def drain_collectives(wait_s=None):
    """Call before a capture whenever a process group is alive: returns once c10d's watchdog threads hold no eager work.

    Why (round 5, `tools/debug/loop_bench.sh`: 6 aborts in 30 runs of the forced-shard rehearsal): a watchdog sweeps its list of
    issued collectives every 100 ms and `hipEventQuery`s each one's end event.  An eager step's collectives are still on that
    list when the next step is captured right behind it, and as soon as the capture pulls the communicator's stream in, HIP
    answers a query of an event recorded on that stream with `hipErrorCapturedEvent` ("operation not permitted on an event last
    recorded in a capturing stream") -- the watchdog rethrows and the process dies with SIGABRT, whatever the capture mode.
    CUDA builds of torch wait for the pending work themselves when a capture begins; the ROCm build of torch 2.10 does not.

    Round 6: the wait is a CONDITION, not a sleep.  `ProcessGroup._wait_for_pending_works()` (c10d's
    `ProcessGroupNCCL::waitForPendingWorks`, the loop CUDA builds run at `capture_begin`) returns when the backend's
    `workMetaList_` -- exactly the list the watchdog sweeps -- and its completed-work list are empty.  It is called on every
    process group of the process after a device synchronize (every issued collective has then completed on the GPU, so the
    watchdog retires each one on its next sweep, however late that sweep is scheduled).  Round 5's fixed sleep of three sweep
    periods remains only as the fallback for a torch build without that binding (`FINO_CAPTURE_DRAIN_S`, default 0.3 s per
    process group), and `wait_s` adds a sleep on top for A/B rehearsals."""
    import os
    import time
    try:
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return
    except (ImportError, RuntimeError):
        return
    torch.cuda.synchronize()
    blind = 0
    for pg in _process_groups():
        wait = getattr(pg, "_wait_for_pending_works", None)
        if wait is None:
            blind += 1
            continue
        try:
            wait()
        except RuntimeError:             # a backend that does not implement it: fall back to the sleep for this group
            blind += 1
    if blind:
        time.sleep(blind * float(os.environ.get("FINO_CAPTURE_DRAIN_S", "0.3")))
    if wait_s:
        time.sleep(wait_s)


class StepGraph:
    def __init__(self, step_fn, mode, capturable, total_steps, error_mode="global"):
        self.step_fn, self.mode, self.error_mode = step_fn, mode, error_mode
        # fewer than three steps: one eager + one capture pass and nothing left to replay
        self.enabled = mode is not False and bool(capturable) and (total_steps >= 3 or mode is True)
        if mode is True and not capturable:
            raise RuntimeError("use_hip_graph=True, but this loop cannot be captured (a callback between steps, a gloo "
                               "exchange staged through the host, or tensors that are not on a GPU)")
        self.graph, self.ran = None, 0

    def step(self):
        if not self.enabled or self.ran == 0:
            self.step_fn()
        else:
            if self.graph is None:
                g = torch.cuda.CUDAGraph()
                drain_collectives()              # (a no-op without a process group)
                try:
                    with torch.cuda.graph(g, capture_error_mode=self.error_mode):
                        self.step_fn()
                except RuntimeError as ex:
                    if self.mode is True:
                        raise
                    warnings.warn(f"hipGraph capture of the denoise step failed ({ex}); the loop continues eagerly on the "
                                  f"same kernels", RuntimeWarning, stacklevel=2)
                    self.enabled = False
                    torch.cuda.synchronize()
                    self.step_fn()
                    self.ran += 1
                    return
                self.graph = g
            self.graph.replay()
        self.ran += 1

    def close(self):
        """Drop the captured graph (call when the loop ends).  A graph that holds a captured all-to-all must die BEFORE the process
        group does: `destroy_process_group()` never returns while such a graph is alive (profiles/r06_rccl_capture_probe.txt)."""
        if self.graph is not None:
            self.graph = None
            torch.cuda.synchronize()
