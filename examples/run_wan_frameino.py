#!/usr/bin/env python3
"""End-to-end FrameINO clip on one MI355X, the way app.py:inference drives the reference pipeline (app.py:560-760):
canvas image + point tracks + identity reference + prompt -> 49 frames.

With `--ckpt <folder>` (a diffusers-format Wan2.2-TI2V-5B FrameINO checkpoint: transformer/, vae/, and -- for text
prompts -- text_encoder/ + tokenizer/ loadable by transformers) it runs the released weights.  Without it the script
runs the same code on random-init weights of the same architecture and synthetic conditions (there is no network
here), which is what `--smoke` (tiny shapes, seconds) checks in the test suite.

    python examples/run_wan_frameino.py --steps 50 --out clip.npy
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def synthetic_conditions(frames, height, width, seed=0):
    """Stand-ins for the app's inputs (SURVEY 8c harness rows): a canvas that is black outside the first-frame region,
    two objects moving on straight lines (the second one entering the frame late = 'frame in'), one identity image."""
    import PIL.Image
    rng = np.random.default_rng(seed)
    canvas = np.zeros((height, width, 3), dtype=np.uint8)
    y0, y1, x0, x1 = height // 8, height - height // 8, width // 6, width - width // 6
    canvas[y0:y1, x0:x1] = rng.integers(0, 255, (y1 - y0, x1 - x0, 3), dtype=np.uint8)
    tracks = []
    for f in range(frames):
        a = f / max(frames - 1, 1)
        obj0 = [(int(x0 + 40 + a * (x1 - x0 - 80)) + dx, int(height * 0.5) + dy) for dx in (0, 12) for dy in (0, 12)]
        obj1 = [(int(-60 + a * (width * 0.6)), int(height * 0.3))]           # starts outside the canvas
        tracks.append([obj0, obj1])
    ident = rng.integers(0, 255, (height, width, 3), dtype=np.uint8)
    return PIL.Image.fromarray(canvas), tracks, ident


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ckpt", default=None)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--frames", type=int, default=49)
    ap.add_argument("--height", type=int, default=704)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--guidance", type=float, default=5.0)
    ap.add_argument("--prompt", default="A corgi runs into the frame from the left.")
    ap.add_argument("--scheduler", choices=["euler", "unipc"], default="unipc")
    ap.add_argument("--smoke", action="store_true", help="tiny random model + tiny VAE, 64x96, 5 frames")
    ap.add_argument("--repeat", type=int, default=1, help="generate the clip this many times (the first call is cold)")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()

    from frameino_amd import _lib
    from frameino_amd.autoencoder_kl_wan import AutoencoderKLWan
    from frameino_amd.conditions import prepare_traj_tensor
    from frameino_amd.pipeline_wan_i2v_motion_frameino import WanImageToVideoPipeline
    from frameino_amd.schedulers import FlowMatchEulerDiscreteScheduler, UniPCMultistepScheduler
    _lib.load()
    dev = torch.device("cuda")
    sched = UniPCMultistepScheduler(flow_shift=5.0) if a.scheduler == "unipc" else FlowMatchEulerDiscreteScheduler(shift=5.0)

    tokenizer = text_encoder = None
    if a.ckpt:
        from frameino_amd.loading import load_wan_transformer, load_wan_vae
        transformer = load_wan_transformer(os.path.join(a.ckpt, "transformer"), torch.bfloat16, dev)
        vae = load_wan_vae(os.path.join(a.ckpt, "vae"), torch.bfloat16, dev)
        if os.path.isdir(os.path.join(a.ckpt, "text_encoder")):
            from transformers import AutoTokenizer, UMT5EncoderModel
            tokenizer = AutoTokenizer.from_pretrained(os.path.join(a.ckpt, "tokenizer"))
            text_encoder = UMT5EncoderModel.from_pretrained(os.path.join(a.ckpt, "text_encoder"),
                                                            torch_dtype=torch.bfloat16).to(dev)
        text_dim = transformer.config.text_dim
    else:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from bench import build_model
        if a.smoke:
            a.height, a.width, a.frames, a.steps = 64, 96, 5, min(a.steps, 3)
            cfg = dict(patch_size=(1, 2, 2), num_attention_heads=2, attention_head_dim=128, in_channels=32,
                       out_channels=16, text_dim=64, freq_dim=256, ffn_dim=512, num_layers=2, cross_attn_norm=True,
                       qk_norm="rms_norm_across_heads", eps=1e-6, rope_max_seq_len=1024)
            vae = AutoencoderKLWan(base_dim=32, decoder_base_dim=32, z_dim=16, dim_mult=[1, 2, 4, 4], num_res_blocks=1,
                                   temperal_downsample=[False, True, True], is_residual=True, in_channels=12,
                                   out_channels=12, patch_size=2, scale_factor_temporal=4, scale_factor_spatial=16,
                                   latents_mean=[0.0] * 16, latents_std=[1.0] * 16).random_init_(seed=2, device=dev)
        else:
            from frameino_amd.configs import WAN22_5B_CFG, WAN22_VAE_CFG
            cfg = dict(WAN22_5B_CFG)
            vae = AutoencoderKLWan(**WAN22_VAE_CFG).random_init_(seed=2, device=dev)
        transformer = build_model(cfg, dev)
        text_dim = cfg["text_dim"]

    pipe = WanImageToVideoPipeline(tokenizer=tokenizer, text_encoder=text_encoder, vae=vae, scheduler=sched,
                                   transformer=transformer, expand_timesteps=True)
    canvas, tracks, ident = synthetic_conditions(a.frames, a.height, a.width)
    t0 = time.perf_counter()
    traj = prepare_traj_tensor(tracks, a.height, a.width, 6, a.width, a.height, device=dev)        # [F, 3, H, W]
    id_tensor = (torch.from_numpy(ident).to(dev).float() / 255.0 * 2.0 - 1.0).permute(2, 0, 1)[None, :, None]
    kw = {}
    if text_encoder is None:                                   # no text encoder offline: synthetic prompt embeddings
        g = torch.Generator().manual_seed(0)
        pe = torch.randn(1, 512, text_dim, generator=g)
        pe[:, 32:] = 0
        kw = dict(prompt_embeds=pe.to(dev), negative_prompt_embeds=torch.zeros(1, 512, text_dim, device=dev))
    else:
        kw = dict(prompt=a.prompt, negative_prompt="")
    torch.cuda.synchronize()
    tc = time.perf_counter()
    for rep in range(a.repeat):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        frames = pipe(image=canvas, traj_tensor=traj, ID_tensor=id_tensor, height=a.height, width=a.width,
                      num_frames=a.frames, num_inference_steps=a.steps, guidance_scale=a.guidance,
                      generator=torch.Generator().manual_seed(1234), **kw).frames[0]
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        assert frames.shape == (a.frames, a.height, a.width, 3) and np.isfinite(frames).all()
        cond_s = f"conditions {tc - t0:.2f} s, " if rep == 0 else ""
        print(f"{cond_s}clip ({a.frames} frames {a.height}x{a.width}, {a.steps} steps, "
              f"{a.scheduler}) {t2 - t1:.2f} s{' (cold)' if rep == 0 and a.repeat > 1 else ''}, "
              f"frames in [{frames.min():.3f}, {frames.max():.3f}], peak device memory "
              f"{torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
    if a.out:
        np.save(a.out, frames)


if __name__ == "__main__":
    main()
