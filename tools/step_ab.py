#!/usr/bin/env python3
"""Same-box A/B of whole denoise steps (the bench workload) under switches that do not change results:
GEMM raster group height (r01's fixed 4 vs the per-shape default) and the branch-invariant prefix computed once.
Interleaved rounds in one process, median ms/step.  GPU box only."""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import build_model  # noqa: E402
from frameino_amd import _lib  # noqa: E402
from frameino_amd.configs import WAN22_5B_CFG  # noqa: E402
from frameino_amd.pipeline_wan_i2v_motion_frameino import WanImageToVideoPipeline  # noqa: E402
from frameino_amd.schedulers import FlowMatchEulerDiscreteScheduler  # noqa: E402

dev = torch.device("cuda")
cfg = dict(WAN22_5B_CFG)
model = build_model(cfg, dev)
pipe = WanImageToVideoPipeline(scheduler=FlowMatchEulerDiscreteScheduler(shift=5.0), transformer=model, expand_timesteps=True)
g = torch.Generator().manual_seed(1234)
C, fg, lh, lw = 48, 13, 44, 80
lat = torch.randn(1, C, fg, lh, lw, generator=g).to(dev)
cond = torch.randn(1, C, 1, lh, lw, generator=g).to(dev)
traj = torch.randn(1, C, fg + 1, lh, lw, generator=g).to(dev)
idl = torch.randn(1, C, 1, lh, lw, generator=g).to(dev)
mask = torch.ones(1, 1, fg, lh, lw, device=dev); mask[:, :, 0] = 0
pe = torch.randn(1, 512, 4096, generator=g).to(dev).bfloat16()
ne = torch.randn(1, 512, 4096, generator=g).to(dev).bfloat16()
pipe.scheduler.set_timesteps(50, device=dev)
st = pipe.make_state(lat, cond, traj, idl, mask, pe, ne, 5.0)
st.t_rows[1:2].copy_(pipe.scheduler.timesteps[10:11].float())
st.dt.copy_(pipe.scheduler.dts[10:11])
lib = _lib.lib()
settings = {"r01 (group_m 4, no dedup)": (4, False), "group_m per shape": (0, False), "+ shared prefix once": (0, True)}
res = {k: [] for k in settings}


def run(gm, dedup, steps):
    lib.fino_tune_set(0, gm)
    model.dedup_shared_prefix = dedup
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad():
        for _ in range(steps):
            pipe._step(st)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for k, (gm, dd) in settings.items():
    run(gm, dd, 2)
for rnd in range(5):
    for k, (gm, dd) in settings.items():
        res[k].append(run(gm, dd, 3))
lib.fino_tune_set(0, 0)
for k, v in res.items():
    print(f"{k:32s} median {statistics.median(v):7.2f} ms/step  (min {min(v):.2f}, max {max(v):.2f})")
