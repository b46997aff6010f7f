#!/usr/bin/env python3
"""Which RCCL call patterns survive hipGraph capture on this image (torch 2.10 + RCCL 2.26, one rank)?  One pattern per
child process; faulthandler prints the Python stack of a crash.
    python tools/debug/rccl_capture_probe.py            # all patterns
    python tools/debug/rccl_capture_probe.py only <pattern> ...   # those patterns
    python tools/debug/rccl_capture_probe.py child <pattern>"""
import faulthandler
import os
import subprocess
import sys

PATTERNS = ["allgather_main", "allgather_side", "allgather_two_comms_two_sides", "alltoall_main", "alltoall_side",
            "allgather_main_sync", "alltoall_main_sync",
            # round 5: kernels on two side streams, every collective ISSUED with the capturing stream current
            "sides_compute_only", "allgather_main_issue_side_wait", "allgather_main_issue_main_wait",
            "allgather_main_issue_side_wait_two_comms",
            # round 5, second pass: SYNCHRONOUS collectives (torch >= 2.7 runs them on the current stream, no communicator-stream
            # fork) on a side stream forked from the capturing one
            "alltoall_side_sync", "allgather_side_sync", "alltoall_two_sides_sync",
            # round 6: the heads exchange's bytes through OTHER c10d entry points (all_to_all_single with equal splits is RCCL's
            # ncclAllToAll on ROCm builds; these go through grouped ncclSend / ncclRecv, or through the all-gather that captures)
            "alltoall_list_main", "alltoall_list_main_sync", "alltoall_unequal_main", "isend_irecv_main",
            "alltoall_list_side_wait", "allgather_select_main"]
# PROBE_SAFE=1: capture the way frameino_amd/graph_step.py does since round 5 / 6 -- drain_collectives() first (c10d's watchdog
# holds no eager work), capture_error_mode="thread_local"
SAFE = os.environ.get("PROBE_SAFE") == "1"


def child(pattern):
    import socket
    import torch
    import torch.distributed as dist
    faulthandler.enable()
    faulthandler.dump_traceback_later(40, exit=True)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    ga, gb = dist.new_group([0]), dist.new_group([0])
    x = torch.randn(1024, 256, device=dev)
    y1, y2 = torch.empty_like(x), torch.empty_like(x)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def body():
        main = torch.cuda.current_stream()
        if pattern == "allgather_main":
            w = dist.all_gather_into_tensor(y1, x, group=ga, async_op=True)
            z = x * 2
            w.wait()
            return y1 + z
        if pattern == "allgather_main_sync":
            dist.all_gather_into_tensor(y1, x, group=ga)
            return y1 + 1
        if pattern == "alltoall_main":
            w = dist.all_to_all_single(y1, x, group=ga, async_op=True)
            z = x * 2
            w.wait()
            return y1 + z
        if pattern == "alltoall_main_sync":
            dist.all_to_all_single(y1, x, group=ga)
            return y1 + 1
        if pattern in ("alltoall_list_main", "alltoall_list_main_sync", "alltoall_list_side_wait"):
            # dist.all_to_all on tensor LISTS: ProcessGroupNCCL::alltoall -> torch::cuda::nccl::all2all = ncclGroupStart,
            # one ncclSend + ncclRecv per peer, ncclGroupEnd
            ins, outs = list(x.view(1, -1, 256).unbind(0)), list(y1.view(1, -1, 256).unbind(0))
            if pattern == "alltoall_list_main_sync":
                dist.all_to_all(outs, ins, group=ga)
                return y1 + 1
            w = dist.all_to_all(outs, ins, group=ga, async_op=True)
            z = x * 2
            if pattern == "alltoall_list_side_wait":
                s1.wait_stream(main)
                with torch.cuda.stream(s1):
                    w.wait()
                    r = y1 + z
                main.wait_stream(s1)
                return r
            w.wait()
            return y1 + z
        if pattern == "alltoall_unequal_main":
            # explicit split sizes: all2all_single_unequal_split (grouped send / recv) instead of ncclAllToAll
            w = dist.all_to_all_single(y1, x, output_split_sizes=[x.shape[0]], input_split_sizes=[x.shape[0]], group=ga,
                                       async_op=True)
            z = x * 2
            w.wait()
            return y1 + z
        if pattern == "isend_irecv_main":
            ops_ = [dist.P2POp(dist.isend, x, 0, group=ga), dist.P2POp(dist.irecv, y1, 0, group=ga)]
            reqs = dist.batch_isend_irecv(ops_)
            z = x * 2
            for r_ in reqs:
                r_.wait()
            return y1 + z
        if pattern == "allgather_select_main":
            # the exchange as an all-gather of every rank's whole send buffer [ways, rows, width] + a local pick of "my" slice of
            # each: `ways` times the bytes, but the collective that is known to capture
            big = torch.empty((1,) + tuple(x.shape), device=x.device)
            w = dist.all_gather_into_tensor(big, x[None], group=ga, async_op=True)
            z = x * 2
            w.wait()
            y1.copy_(big[0])
            return y1 + z
        if pattern in ("allgather_side", "alltoall_side"):
            s1.wait_stream(main)
            with torch.cuda.stream(s1):
                f = dist.all_gather_into_tensor if pattern == "allgather_side" else dist.all_to_all_single
                w = f(y1, x, group=ga, async_op=True)
                z = x * 2
                w.wait()
                r = y1 + z
            main.wait_stream(s1)
            return r
        if pattern in ("alltoall_side_sync", "allgather_side_sync"):
            s1.wait_stream(main)
            with torch.cuda.stream(s1):
                f = dist.all_gather_into_tensor if pattern == "allgather_side_sync" else dist.all_to_all_single
                f(y1, x, group=ga)
            z = x * 2                            # main goes on while the collective runs on s1
            main.wait_stream(s1)
            return y1 + z
        if pattern == "alltoall_two_sides_sync":
            s1.wait_stream(main)
            s2.wait_stream(main)
            with torch.cuda.stream(s1):
                dist.all_to_all_single(y1, x, group=ga)
                r1 = y1 * 2
            with torch.cuda.stream(s2):
                dist.all_to_all_single(y2, x, group=gb)
                r2 = y2 * 3
            main.wait_stream(s1)
            main.wait_stream(s2)
            return r1 + r2
        if pattern == "allgather_two_comms_two_sides":
            s1.wait_stream(main)
            s2.wait_stream(main)
            with torch.cuda.stream(s1):
                w1 = dist.all_gather_into_tensor(y1, x, group=ga, async_op=True)
            with torch.cuda.stream(s2):
                w2 = dist.all_gather_into_tensor(y2, x, group=gb, async_op=True)
            with torch.cuda.stream(s1):
                w1.wait()
                r1 = y1 * 2
            with torch.cuda.stream(s2):
                w2.wait()
                r2 = y2 * 3
            main.wait_stream(s1)
            main.wait_stream(s2)
            return r1 + r2
        if pattern == "sides_compute_only":
            s1.wait_stream(main)
            s2.wait_stream(main)
            with torch.cuda.stream(s1):
                r1 = x * 2
            with torch.cuda.stream(s2):
                r2 = x * 3
            main.wait_stream(s1)
            main.wait_stream(s2)
            return r1 + r2
        if pattern in ("allgather_main_issue_side_wait", "allgather_main_issue_main_wait",
                       "allgather_main_issue_side_wait_two_comms"):
            # the interleaved plan re-expressed: the branches' kernels run on s1 / s2, the capturing stream carries only
            # event waits and the collectives (c10d forks its communicator stream from the CAPTURING stream, the pattern
            # that captures); a branch's stream then waits for its own collective
            gb_ = gb if pattern.endswith("two_comms") else ga
            s1.wait_stream(main)
            s2.wait_stream(main)
            with torch.cuda.stream(s1):
                a1 = x * 2                      # "S_A": produces what the collective sends
            with torch.cuda.stream(s2):
                a2 = x * 3
            main.wait_stream(s1)
            w1 = dist.all_gather_into_tensor(y1, a1, group=ga, async_op=True)
            main.wait_stream(s2)
            w2 = dist.all_gather_into_tensor(y2, a2, group=gb_, async_op=True)
            if pattern == "allgather_main_issue_main_wait":
                w1.wait()
                s1.wait_stream(main)
                with torch.cuda.stream(s1):
                    r1 = y1 + a1
                w2.wait()
                s2.wait_stream(main)
                with torch.cuda.stream(s2):
                    r2 = y2 + a2
            else:
                with torch.cuda.stream(s1):
                    w1.wait()
                    r1 = y1 + a1
                with torch.cuda.stream(s2):
                    w2.wait()
                    r2 = y2 + a2
            main.wait_stream(s1)
            main.wait_stream(s2)
            return r1 + r2
        raise ValueError(pattern)

    ref = body()
    body()
    torch.cuda.synchronize()
    print(f"[{pattern}] eager ok", flush=True)
    g = torch.cuda.CUDAGraph()
    if SAFE:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
        from frameino_amd.graph_step import drain_collectives
        drain_collectives()
    with torch.cuda.graph(g, capture_error_mode="thread_local" if SAFE else "global"):
        out = body()
    print(f"[{pattern}] captured", flush=True)
    g.replay()
    torch.cuda.synchronize()
    print(f"[{pattern}] replayed equal={bool(torch.equal(out, ref))}", flush=True)
    x.mul_(-0.5)                                    # new input through the same graph: a replay that re-reads its operands
    ref2 = body()
    g.replay()
    torch.cuda.synchronize()
    print(f"[{pattern}] replayed on new data equal={bool(torch.equal(out, ref2))}", flush=True)
    faulthandler.cancel_dump_traceback_later()
    if os.environ.get("PROBE_DEL_GRAPH") == "1":
        # does the teardown hang of a captured synchronous all-to-all go away when the graph dies BEFORE the process group?
        import gc
        del g, out
        gc.collect()
        torch.cuda.synchronize()
        print(f"[{pattern}] graph deleted", flush=True)
    if os.environ.get("PROBE_ABORT_PG") == "1":
        # ... or when the communicators are aborted instead of destroyed in order?
        for pg in (ga, gb, dist.group.WORLD):
            try:
                pg._get_backend(dev).abort()
            except Exception as ex:      # noqa: BLE001
                print(f"[{pattern}] abort: {type(ex).__name__}: {ex}", flush=True)
        print(f"[{pattern}] aborted", flush=True)
    faulthandler.dump_traceback_later(25, exit=True)
    dist.destroy_process_group()
    print(f"[{pattern}] DONE", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(sys.argv[2])
        sys.exit(0)
    pats = sys.argv[2:] if len(sys.argv) > 2 and sys.argv[1] == "only" else PATTERNS
    for pat in pats:
        print(f"===== {pat}", flush=True)
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "child", pat], timeout=80, capture_output=True,
                               text=True)
            err = "\n".join(ln for ln in p.stderr.splitlines() if "amdgpu.ids" not in ln and "socket.cpp" not in ln)
            print(p.stdout[-800:], err[-3500:], f"rc={p.returncode}", flush=True)
        except subprocess.TimeoutExpired as ex:
            print("TIMEOUT", (ex.stdout or b"")[-800:], (ex.stderr or b"")[-3000:], flush=True)
