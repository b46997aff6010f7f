#!/usr/bin/env python3
"""Samples rocm-smi power / clocks while the bench workload loops: is the denoise step power-capped?
(diagnostic; GPU box only)"""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
samples = []
stop = False


def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp", "--json"], capture_output=True, text=True, timeout=10).stdout
            samples.append((time.time(), out))
        except Exception as ex:      # noqa: BLE001
            samples.append((time.time(), repr(ex)))
        time.sleep(0.2)


import torch  # noqa: E402
from bench import build_model  # noqa: E402
from frameino_amd.configs import WAN22_5B_CFG  # noqa: E402
from frameino_amd.pipeline_wan_i2v_motion_frameino import WanImageToVideoPipeline  # noqa: E402
from frameino_amd.schedulers import FlowMatchEulerDiscreteScheduler  # noqa: E402
dev = torch.device("cuda")
model = build_model(dict(WAN22_5B_CFG), dev)
pipe = WanImageToVideoPipeline(scheduler=FlowMatchEulerDiscreteScheduler(shift=5.0), transformer=model, expand_timesteps=True)
g = torch.Generator().manual_seed(1234)
C, fg, lh, lw = 48, 13, 44, 80
lat = torch.randn(1, C, fg, lh, lw, generator=g).to(dev); cond = torch.randn(1, C, 1, lh, lw, generator=g).to(dev)
traj = torch.randn(1, C, fg + 1, lh, lw, generator=g).to(dev); idl = torch.randn(1, C, 1, lh, lw, generator=g).to(dev)
mask = torch.ones(1, 1, fg, lh, lw, device=dev); mask[:, :, 0] = 0
pe = torch.randn(1, 512, 4096, generator=g).to(dev).bfloat16(); ne = torch.randn(1, 512, 4096, generator=g).to(dev).bfloat16()
pipe.scheduler.set_timesteps(50, device=dev)
st = pipe.make_state(lat, cond, traj, idl, mask, pe, ne, 5.0)
st.t_rows[1:2].copy_(pipe.scheduler.timesteps[10:11].float()); st.dt.copy_(pipe.scheduler.dts[10:11])
with torch.no_grad():
    pipe._step(st)
torch.cuda.synchronize()
th = threading.Thread(target=sampler); th.start()
time.sleep(1.0)
t0 = time.time()
with torch.no_grad():
    for _ in range(20):
        pipe._step(st)
torch.cuda.synchronize()
t1 = time.time()
time.sleep(1.0)
stop = True; th.join()
print(f"20 steps in {t1 - t0:.2f} s")
import json  # noqa: E402
for ts, out in samples:
    tag = "busy" if t0 <= ts <= t1 else "idle"
    try:
        d = json.loads(out)
        c = d[sorted(d)[0]]
        keep = {k: v for k, v in c.items() if any(w in k.lower() for w in ("power", "sclk", "mclk", "temperature (sensor junction)", "fclk"))}
        print(f"{ts - t0:6.2f}s {tag}: {keep}")
    except Exception:      # noqa: BLE001
        print(f"{ts - t0:6.2f}s {tag}: {out[:200]}")
