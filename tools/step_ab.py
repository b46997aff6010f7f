#!/usr/bin/env python3
"""Same-box A/B of whole denoise steps (the bench workload) under switches that do not change results:
GEMM raster group height (r01's fixed 4 vs the per-shape default), the branch-invariant prefix computed once, the 4-wave
attention kernel and the softmax scale folded into q.
Interleaved rounds in one process, median ms/step.  GPU box only."""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import build_model  # noqa: E402
from frameino_amd import _lib, ops  # noqa: E402
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
pe = torch.randn(1, 512, 4096, generator=g)
ne = torch.randn(1, 512, 4096, generator=g)
pe[:, 64:] = 0          # the bench's zero-padded prompts (round 5; rounds 2 - 4 ran un-padded ones here)
ne[:, 8:] = 0
pe, ne = pe.to(dev).bfloat16(), ne.to(dev).bfloat16()
pipe.scheduler.set_timesteps(50, device=dev)
st = pipe.make_state(lat, cond, traj, idl, mask, pe, ne, 5.0)
st.t_rows[1:2].copy_(pipe.scheduler.timesteps[10:11].float())
st.dt.copy_(pipe.scheduler.dts[10:11])
lib = _lib.lib()
# (GEMM group_m, shared prefix once, attention kernel tune [1 = 8-wave, 0 = policy: 4-wave at this shape], folded scale)
settings = {"r01 (group_m 4, no dedup, 8-wave attention)": (4, False, 1, False),
            "group_m per shape": (0, False, 1, False), "+ shared prefix once": (0, True, 1, False),
            "+ 4-wave attention kernel": (0, True, 0, False), "+ softmax scale folded into q (MFMA fold)": (0, True, 0, True)}
if "--tiles" in sys.argv:      # round 3: GEMM tile heights (FINO_TUNE_GEMM_TILE_M: 8 = 256-row tiles only, 0 = planned), default kernels otherwise
    settings = {"256-row GEMM tiles only": (0, True, 1, False, 8), "planned GEMM tile heights": (0, True, 1, False, 0)}
if "--cross" in sys.argv:      # round 4: text cross-attention on the free-running kernel vs the walking ping-pong kernel
    settings = {"cross-attention: free-running kernel (tune 7, the round-3 choice)": (0, True, 7, False, 0),
                "cross-attention: walking ping-pong kernel (the policy)": (0, True, 0, False, 0)}
if "--ppd" in sys.argv:        # round 4: self-attention on the register-staged ping-pong kernel
    # vs the LDS-DMA-staged one; the text cross-attention stays on the free-running kernel in both
    settings = {"self-attention: register-staged ping-pong kernel (tune 5, the round-3 policy)": (0, True, 5, False, 0),
                "self-attention: LDS-DMA-staged ping-pong kernel (the policy)": (0, True, 0, False, 0)}
res = {k: [] for k in settings}


attn = {}
attn_c = {}


def run(gm, dedup, attn_k, fold, steps, tile_m=0):
    lib.fino_tune_set(3, tile_m)
    lib.fino_tune_set(0, gm)
    lib.fino_tune_set(4, attn_k)
    model.fold_softmax_scale = fold
    model.dedup_shared_prefix = dedup
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad(), ops.KernelTimer({"attn_self", "attn_cross"}) as kt:
        for _ in range(steps):
            pipe._step(st)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    key = (gm, dedup, attn_k, fold, tile_m)[:len(next(iter(settings.values())))]
    attn.setdefault(key, []).append(kt.summary()["attn_self"]["total_ms"] / steps)
    attn_c.setdefault(key, []).append(kt.summary()["attn_cross"]["total_ms"] / steps)
    return ms


for k, a in settings.items():
    run(*a[:4], 2, *a[4:])
for rnd in range(5):
    for k, a in settings.items():
        res[k].append(run(*a[:4], 3, *a[4:]))
lib.fino_tune_set(0, 0)
lib.fino_tune_set(4, 0)
lib.fino_tune_set(3, 0)
for k, v in res.items():
    a_ms = statistics.median(attn[settings[k]][1:])
    c_ms = statistics.median(attn_c[settings[k]][1:])
    print(f"    self-attention launches {a_ms:7.2f} ms/step, text cross-attention launches {c_ms:6.2f}, everything else {statistics.median(v) - a_ms - c_ms:7.2f}")
    print(f"{k:44s} median {statistics.median(v):7.2f} ms/step  (min {min(v):.2f}, max {max(v):.2f})")
