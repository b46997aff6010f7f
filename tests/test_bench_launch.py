"""`python bench.py --gpus N` must start by itself: the driver's multi-GPU leg may run the bare command.

CPU: the parent starts its ranks as a child `torch.distributed.run` and hands the child's failure on (there is no GPU here,
so the ranks die at `torch.cuda.set_device`) -- no re-exec, no silent success.
GPU: the bare command with 2 gloo ranks on the one GPU prints a line with n_gpus = 2, and the forced-shard rehearsal drives
the N > 1 code path through a real RCCL communicator of one rank.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(args, env_extra=None, timeout=900):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env,
                       timeout=timeout, cwd=ROOT)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    return p, lines


def test_bare_command_launches_ranks_and_propagates_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-box check (on a GPU box the ranks succeed: see the gpu tests)")
    p, lines = run_bench(["--gpus", "2", "--workload", "tiny", "--steps", "1", "--warmup", "0"])
    assert "starting -m torch.distributed.run" in p.stderr, p.stderr[-2000:]
    assert "--nproc-per-node=2" in p.stderr
    assert p.returncode != 0 and not lines, (p.returncode, lines)


@pytest.mark.gpu
def test_bare_command_two_gloo_ranks_on_one_gpu():
    p, lines = run_bench(["--gpus", "2", "--workload", "tiny", "--steps", "2", "--warmup", "1"],
                         {"FINO_DIST_BACKEND": "gloo"})
    assert p.returncode == 0 and len(lines) == 1, (p.returncode, p.stderr[-3000:])
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["value"] > 0
    assert out["config"]["ranks_seen"] == 2 and out["config"]["backend"] == "gloo"


@pytest.mark.gpu
@pytest.mark.parametrize("plan,exchange", [("split", "kv"), ("interleave", "kv"), ("interleave", "heads"), ("split", "heads")])
def test_forced_shard_path_through_rccl_of_one_rank(plan, exchange):
    p, lines = run_bench(["--gpus", "1", "--force-shard", "--plan", plan, "--exchange", exchange, "--workload", "tiny",
                          "--steps", "2", "--warmup", "1"])
    assert p.returncode == 0 and len(lines) == 1, (p.returncode, p.stderr[-3000:])
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["config"]["ranks_seen"] == 1 and out["config"]["backend"] == "nccl"
    assert out["config"]["parallelism"] != "single"
