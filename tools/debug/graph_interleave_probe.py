#!/usr/bin/env python3
"""Where does hipGraph capture of the INTERLEAVED sharded step (two RCCL communicators driven from two side streams) get
stuck?  One experiment per child process (each under its own timeout, faulthandler dumps every thread's stack before the
timeout fires):  python tools/debug/graph_interleave_probe.py            # runs all
                 python tools/debug/graph_interleave_probe.py child <mode> <plan> <exchange>"""
import faulthandler
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child(mode, plan_mode, exchange):
    import socket
    import torch
    import torch.distributed as dist
    faulthandler.dump_traceback_later(50, exit=True)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from frameino_amd.parallel import shard_pipeline
    from tests.test_parallel_gpu import _pipe
    pipe, a = _pipe("cuda:0")
    pipe.batch_cfg = False
    shard_pipeline(pipe, 0, 1, cfg_parallel=False, mode=plan_mode, allow_single=True, exchange=exchange)
    d = lambda k: a[k].to(dev)          # noqa: E731
    pipe.scheduler.set_timesteps(4, device=dev)
    st = pipe.make_state(d("latents0"), d("condition"), d("traj_latents"), d("id_latent"), d("mask"), d("prompt_embeds"),
                         d("negative_embeds"), 5.0)
    st.t_rows[1:2] = 700.0
    st.dt[0] = -0.01
    with torch.no_grad():
        pipe._step(st)
        pipe._step(st)
        torch.cuda.synchronize()
        print(f"[{mode} {plan_mode} {exchange}] eager ok", flush=True)
        if mode == "sync-first":
            time.sleep(2.0)                      # let the process group's watchdog retire the eager works first
            mode = "global"
        g = torch.cuda.CUDAGraph()
        t0 = time.time()
        with torch.cuda.graph(g, capture_error_mode=mode):
            pipe._step(st)
        print(f"[{mode} {plan_mode} {exchange}] captured in {time.time() - t0:.2f}s", flush=True)
        ref = st.lat.clone()
        g.replay()
        torch.cuda.synchronize()
        print(f"[{mode} {plan_mode} {exchange}] replayed, finite={bool(torch.isfinite(st.lat).all())} "
              f"moved={bool((st.lat != ref).any())}", flush=True)
    faulthandler.cancel_dump_traceback_later()
    dist.destroy_process_group()
    print(f"[{mode} {plan_mode} {exchange}] DONE", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(*sys.argv[2:5])
        sys.exit(0)
    for mode, plan_mode, exchange in (("global", "split", "kv"), ("global", "interleave", "kv"),
                                      ("sync-first", "interleave", "kv"), ("thread_local", "interleave", "kv"),
                                      ("relaxed", "interleave", "kv"), ("thread_local", "interleave", "heads"),
                                      ("thread_local", "split", "heads")):
        print(f"===== {mode} {plan_mode} {exchange}", flush=True)
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "child", mode, plan_mode, exchange],
                               timeout=90, capture_output=True, text=True)
            print(p.stdout[-1500:], p.stderr[-6000:], f"rc={p.returncode}", flush=True)
        except subprocess.TimeoutExpired as ex:
            print("TIMEOUT", (ex.stdout or b"")[-1500:], (ex.stderr or b"")[-6000:], flush=True)
