import statistics
import sys
import time
sys.path.insert(0, ".")
import torch
from bench import build_model
from frameino_amd import _lib
from frameino_amd.configs import WAN22_5B_CFG
from frameino_amd.pipeline_wan_i2v_motion_frameino import WanImageToVideoPipeline
from frameino_amd.schedulers import FlowMatchEulerDiscreteScheduler
dev = torch.device("cuda")
cfg = dict(WAN22_5B_CFG)
model = build_model(cfg, dev)
pipe = WanImageToVideoPipeline(scheduler=FlowMatchEulerDiscreteScheduler(shift=5.0), transformer=model, expand_timesteps=True)
g = torch.Generator().manual_seed(1234)
C, fg, lh, lw = 48, 13, 44, 80
lat = torch.randn(1, C, fg, lh, lw, generator=g).to(dev); cond = torch.randn(1, C, 1, lh, lw, generator=g).to(dev)
traj = torch.randn(1, C, fg + 1, lh, lw, generator=g).to(dev); idl = torch.randn(1, C, 1, lh, lw, generator=g).to(dev)
mask = torch.ones(1, 1, fg, lh, lw, device=dev); mask[:, :, 0] = 0
pe = torch.randn(1, 512, 4096, generator=g); ne = torch.randn(1, 512, 4096, generator=g); pe[:, 64:] = 0; ne[:, 8:] = 0
pipe.scheduler.set_timesteps(50, device=dev)
st = pipe.make_state(lat, cond, traj, idl, mask, pe.to(dev), ne.to(dev), 5.0)
st.t_rows[1:2].copy_(pipe.scheduler.timesteps[10:11].float()); st.dt.copy_(pipe.scheduler.dts[10:11])
lib = _lib.lib()
res = {0: [], 2: [], 1: [], 3: []}
with torch.no_grad():
    for k in (0, 2, 1, 3): lib.fino_tune_set(1, k); pipe._step(st); pipe._step(st)
    for rnd in range(5):
        for k in (2, 0, 1, 3):
            lib.fino_tune_set(1, k)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(4): pipe._step(st)
            torch.cuda.synchronize(); res[k].append((time.perf_counter() - t0) / 4 * 1e3)
lib.fino_tune_set(1, 0)
for k, nm in ((2, "long-K GEMM rows first to last (rounds 1-5)"), (0, "long-K GEMM rows last to first (default)"),
              (1, "EVERY GEMM's rows last to first"), (3, "long-K and wide-N (>= 8192) GEMMs last to first")):
    print(f"{nm}: median {statistics.median(res[k]):.2f} ms/step ({' '.join(f'{v:.1f}' for v in res[k])})")
