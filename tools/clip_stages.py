#!/usr/bin/env python3
"""Where the seconds of ONE pipeline call go: every stage of `WanImageToVideoPipeline.__call__` wrapped in a
synchronize + wall-clock pair (so the numbers add up to the call; the wrappers' syncs cost < 1 ms in total).

    python tools/clip_stages.py [--steps 50] [--scheduler unipc] [--repeat 2]

Stages: encode_prompt | preprocess (host LANCZOS resize) | prepare (noise + 3 VAE encodes) | denoise (the loop) |
decode (VAE) | postprocess (clamp, permute, device -> host, numpy)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--scheduler", choices=["euler", "unipc"], default="unipc")
    ap.add_argument("--repeat", type=int, default=2)
    a = ap.parse_args()
    from bench import build_model
    from frameino_amd import _lib
    from frameino_amd.autoencoder_kl_wan import AutoencoderKLWan
    from frameino_amd.conditions import prepare_traj_tensor
    from frameino_amd.configs import WAN22_5B_CFG, WAN22_VAE_CFG
    from frameino_amd.pipeline_wan_i2v_motion_frameino import WanImageToVideoPipeline
    from frameino_amd.schedulers import FlowMatchEulerDiscreteScheduler, UniPCMultistepScheduler
    from run_wan_frameino import synthetic_conditions
    _lib.load()
    dev = torch.device("cuda")
    cfg = dict(WAN22_5B_CFG)
    vae = AutoencoderKLWan(**WAN22_VAE_CFG).random_init_(seed=2, device=dev)
    tr = build_model(cfg, dev)
    sched = UniPCMultistepScheduler(flow_shift=5.0) if a.scheduler == "unipc" else FlowMatchEulerDiscreteScheduler(shift=5.0)
    pipe = WanImageToVideoPipeline(vae=vae, scheduler=sched, transformer=tr, expand_timesteps=True)
    H, W, F = 704, 1280, 49
    canvas, tracks, id_tensor, _ = synthetic_conditions(F, H, W, dev)
    traj = prepare_traj_tensor(tracks, H, W, 6, W, H, device=dev)
    g = torch.Generator().manual_seed(0)
    pe = torch.randn(1, 512, cfg["text_dim"], generator=g)
    pe[:, 32:] = 0
    kw = dict(prompt_embeds=pe.to(dev), negative_prompt_embeds=torch.zeros(1, 512, cfg["text_dim"], device=dev))

    times = {}

    def wrap(obj, name, label):
        fn = getattr(obj, name)

        def timed(*args, **kwargs):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = fn(*args, **kwargs)
            torch.cuda.synchronize()
            times[label] = times.get(label, 0.0) + time.perf_counter() - t0
            return out
        setattr(obj, name, timed)

    wrap(pipe, "encode_prompt", "encode_prompt")
    wrap(pipe.video_processor, "preprocess", "preprocess")
    wrap(pipe, "_prepare_conditions", "prepare")
    wrap(pipe, "denoise", "denoise")
    wrap(pipe.vae, "decode", "decode")
    wrap(pipe.video_processor, "postprocess_video", "postprocess")
    for rep in range(a.repeat):
        times.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        frames = pipe(image=canvas, traj_tensor=traj, ID_tensor=id_tensor, height=H, width=W, num_frames=F,
                      num_inference_steps=a.steps, guidance_scale=5.0, generator=torch.Generator().manual_seed(1234),
                      **kw).frames[0]
        torch.cuda.synchronize()
        total = time.perf_counter() - t0
        parts = "  ".join(f"{k} {v:.3f}" for k, v in times.items())
        print(f"call {rep}: {total:.3f} s = {parts}  other {total - sum(times.values()):.3f}   "
              f"(denoise / step {times['denoise'] / a.steps * 1e3:.1f} ms)  frames {frames.shape} {frames.dtype}", flush=True)


if __name__ == "__main__":
    main()
