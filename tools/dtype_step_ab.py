#!/usr/bin/env python3
"""Same-box A/B of the bench's denoise step with the DiT in bf16 and in fp16 (round 6: fp16 is the dtype the reference app loads
the model in, app.py:156).  Interleaved rounds in one process; per round the step time and the summed HIP-event time of the
self-attention, the text cross-attention and the four GEMM epilogue classes; optionally (--peak) the matrix pipe's own rate on
bf16 and fp16 operands of the same N(0, 1) values (fino_diag_mfma_peak kinds 0 / 3) under the same power cap.  GPU box only."""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import build_model  # noqa: E402
from frameino_amd import ops  # noqa: E402
from frameino_amd.configs import WAN22_5B_CFG  # noqa: E402
from frameino_amd.pipeline_wan_i2v_motion_frameino import WanImageToVideoPipeline  # noqa: E402
from frameino_amd.schedulers import FlowMatchEulerDiscreteScheduler  # noqa: E402

dev = torch.device("cuda")
cfg = dict(WAN22_5B_CFG)
NAMES = ("attn_self", "attn_cross", "gemm_epi0", "gemm_epi1", "gemm_epi2", "gemm_epi3")


def make(dtype):
    model = build_model(cfg, dev, dtype=dtype)
    pipe = WanImageToVideoPipeline(scheduler=FlowMatchEulerDiscreteScheduler(shift=5.0), transformer=model, expand_timesteps=True)
    g = torch.Generator().manual_seed(1234)
    C, fg, lh, lw = 48, 13, 44, 80
    lat = torch.randn(1, C, fg, lh, lw, generator=g).to(dev)
    cond = torch.randn(1, C, 1, lh, lw, generator=g).to(dev)
    traj = torch.randn(1, C, fg + 1, lh, lw, generator=g).to(dev)
    idl = torch.randn(1, C, 1, lh, lw, generator=g).to(dev)
    mask = torch.ones(1, 1, fg, lh, lw, device=dev)
    mask[:, :, 0] = 0
    pe = torch.randn(1, 512, 4096, generator=g)
    ne = torch.randn(1, 512, 4096, generator=g)
    pe[:, 64:] = 0
    ne[:, 8:] = 0
    pipe.scheduler.set_timesteps(50, device=dev)
    st = pipe.make_state(lat, cond, traj, idl, mask, pe.to(dev), ne.to(dev), 5.0)
    st.t_rows[1:2].copy_(pipe.scheduler.timesteps[10:11].float())
    st.dt.copy_(pipe.scheduler.dts[10:11])
    return pipe, st


def run(pipe, st, steps, timed):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad(), ops.KernelTimer(set(NAMES) if timed else set()) as kt:
        for _ in range(steps):
            pipe._step(st)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    return ms, ({k: v["total_ms"] / steps for k, v in kt.summary().items()} if timed else {})


sides = {"bf16": make(torch.bfloat16), "fp16": make(torch.float16)}
for p, s in sides.values():
    run(p, s, 2, False)
res = {k: [] for k in sides}
parts = {k: [] for k in sides}
for rnd in range(4):
    for k, (p, s) in sides.items():
        res[k].append(run(p, s, 3, False)[0])             # the step as the bench times it (no events inside)
        parts[k].append(run(p, s, 2, True)[1])
for k in sides:
    print(f"{k}: step {statistics.median(res[k]):7.2f} ms   (rounds: {' '.join(f'{x:.1f}' for x in res[k])})")
print(f"fp16 / bf16 = {statistics.median(res['fp16']) / statistics.median(res['bf16']):.4f}")
print(f"{'class (HIP events, ms per step)':34s} {'bf16':>9s} {'fp16':>9s}  fp16/bf16")
tot = {"bf16": 0.0, "fp16": 0.0}
for n in NAMES:
    a = statistics.median(x.get(n, 0.0) for x in parts["bf16"])
    b = statistics.median(x.get(n, 0.0) for x in parts["fp16"])
    tot["bf16"] += a
    tot["fp16"] += b
    print(f"{n:34s} {a:9.2f} {b:9.2f}  {b / max(a, 1e-9):.4f}")
print(f"{'sum of the timed classes':34s} {tot['bf16']:9.2f} {tot['fp16']:9.2f}  {tot['fp16'] / tot['bf16']:.4f}")
if "--peak" in sys.argv:
    import ctypes
    from frameino_amd import _lib
    lib = _lib.lib()
    stream = torch.cuda.current_stream().cuda_stream
    for rnd in range(2):
        for kind, name, dt in ((0, "bf16", torch.bfloat16), (3, "fp16", torch.float16)):
            scratch = torch.zeros(64 + 2 * 256 * 4, device=dev)
            view = scratch[64:].view(dt)
            view.copy_(torch.randn(view.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(0)).to(dt))
            fl = ctypes.c_double()
            _lib.check(lib.fino_diag_mfma_peak(kind, 2, 400000, scratch.data_ptr(), ctypes.byref(fl), stream), "fino_diag_mfma_peak")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                _lib.check(lib.fino_diag_mfma_peak(kind, 2, 400000, scratch.data_ptr(), ctypes.byref(fl), stream), "fino_diag_mfma_peak")
            e1.record()
            torch.cuda.synchronize()
            print(f"matrix pipe alone, 32x32x16 {name}, N(0,1) operands, 2 waves/SIMD: {fl.value / (e0.elapsed_time(e1) / 3 * 1e-3) / 1e12:7.1f} TFLOP/s")
