#!/usr/bin/env python3
"""End-to-end CogVideoX FrameINO clip on one MI355X, the way the reference's mass-evaluation scripts drive the pipeline
(test_code/run_cogvideox_FrameIn_mass_evaluation.py:92-108 the loader lines, :203-213 the call): first frame + trajectory
video + identity reference (+ prompt) -> 49 frames 480x720 as PIL images, `guidance_scale=6, use_dynamic_cfg=True,
num_inference_steps=50`.  `--frame-out` is the FrameOut evaluation's variant (no identity reference:
run_cogvideox_FrameOut_mass_evaluation.py passes a black placeholder).

With `--ckpt <folder>` (diffusers-format CogVideoX-5b-I2V + FrameINO transformer: transformer/, vae/, scheduler/, and --
for text prompts -- text_encoder/ + tokenizer/) it runs the released weights.  Without it the script runs the same code on
random-init weights of the same architecture and synthetic conditions (there is no network here); `--smoke` is the tiny
version the test suite runs.  `--mxfp8 --fp8-attention` is BASELINE config 5's "fp8 MFMA path" (opt-in, reduced precision).

    python examples/run_cogvideox_frameino.py --steps 50 --repeat 2
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def synthetic_conditions(frames, height, width, device, frame_out, seed=0):
    """Stand-ins for what VideoDataset_Motion_FrameINO hands the evaluation script, built with the device-side builders of
    frameino_amd.conditions: a first frame (the user's photo area-resampled into the black extended canvas), per-frame
    point tracks (one object entering late = 'frame in'), the trajectory video painted from them, an identity reference
    zero-padded to the canvas ([3, H, W] for CogVideoX; the FrameOut evaluation has none)."""
    import PIL.Image
    from frameino_amd.conditions import prepare_id_tensor, prepare_traj_tensor, resize_area_pad, tracks_from_trajectories
    rng = np.random.default_rng(seed)
    pad_h, pad_w = (height // 8) // 16 * 16, (width // 6) // 16 * 16
    first = rng.integers(0, 255, (360, 640, 3), dtype=np.uint8)
    # (the app's build_canvas insists on multiples of 32 -- a Wan constraint; 480 x 720 is the CogVideoX evaluation preset)
    canvas = resize_area_pad(first, (height - 2 * pad_h, width - 2 * pad_w), (height, width), (pad_h, pad_w), 0, device)
    uh, uw = 480, 720
    clicks = [[[(uw * 0.25, uh * 0.5), (uw * 0.75, uh * 0.5)]], [[(-20.0, uh * 0.3), (uw * 0.55, uh * 0.3)]]]
    tracks = tracks_from_trajectories(clicks, frames, height, width, uh, uw)
    traj = prepare_traj_tensor(tracks, height, width, 6, width, height, device=device)             # [F, 3, H, W] in [-1, 1]
    ident = None if frame_out else rng.integers(0, 255, (300, 200, 3), dtype=np.uint8)
    id_tensor = prepare_id_tensor(ident, height, width, "CogVideoX", device)                       # [3, H, W]
    return PIL.Image.fromarray(canvas.cpu().numpy()), traj, id_tensor, (pad_h, pad_w, pad_h, pad_w)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ckpt", default=None)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--frames", type=int, default=49)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=720)
    ap.add_argument("--guidance", type=float, default=6.0)
    ap.add_argument("--prompt", default="A corgi runs into the frame from the left.")
    ap.add_argument("--scheduler", choices=["ddim", "dpm"], default="dpm",
                    help="the released CogVideoX-5b-I2V folder ships CogVideoXDPMScheduler; the FrameINO training validation uses DDIM")
    ap.add_argument("--frame-out", action="store_true", help="the FrameOut evaluation: no identity reference")
    ap.add_argument("--mxfp8", action="store_true", help="MXFP8 linears (reduced precision, opt-in)")
    ap.add_argument("--fp8-attention", action="store_true", help="fp8 (e4m3) attention operands (reduced precision, opt-in)")
    ap.add_argument("--dtype", choices=["fp16", "bf16"], default="fp16",
                    help="dtype of transformer, VAE and text encoder.  Default fp16 = what the evaluation script loads all three in "
                         "(test_code/run_cogvideox_FrameIn_mass_evaluation.py:92-94,106)")
    ap.add_argument("--smoke", action="store_true", help="tiny random model + tiny VAE, 64x96, 9 frames")
    ap.add_argument("--no-tiling", action="store_true", help="random-weight run: leave the VAE's tiling off (the evaluation "
                                                              "script switches it on)")
    ap.add_argument("--repeat", type=int, default=1, help="generate the clip this many times (the first call is cold)")
    ap.add_argument("--out", default=None, help="write the cropped uint8 frames [F, h, w, 3] as .npy")
    a = ap.parse_args()

    from frameino_amd import _lib
    from frameino_amd.autoencoder_kl_cogvideox import AutoencoderKLCogVideoX
    from frameino_amd.pipeline_cogvideox_i2v_motion_frameino import CogVideoXImageToVideoPipeline
    from frameino_amd.schedulers import CogVideoXDDIMScheduler, CogVideoXDPMScheduler
    _lib.load()
    dev = torch.device("cuda")
    dt = torch.float16 if a.dtype == "fp16" else torch.bfloat16
    sched = CogVideoXDPMScheduler() if a.scheduler == "dpm" else CogVideoXDDIMScheduler()
    tokenizer = text_encoder = None
    if a.ckpt:
        # the loader lines of the evaluation script with only the imports changed (tests/test_loading_cpu.py replays them)
        from frameino_amd.cogvideox_transformer_3d import CogVideoXTransformer3DModel
        transformer = CogVideoXTransformer3DModel.from_pretrained(os.path.join(a.ckpt, "transformer"), torch_dtype=dt)
        vae = AutoencoderKLCogVideoX.from_pretrained(a.ckpt, subfolder="vae", torch_dtype=dt)
        vae.enable_slicing()
        vae.enable_tiling()
        if os.path.isdir(os.path.join(a.ckpt, "text_encoder")):
            from transformers import AutoTokenizer, T5EncoderModel
            tokenizer = AutoTokenizer.from_pretrained(os.path.join(a.ckpt, "tokenizer"))
            text_encoder = T5EncoderModel.from_pretrained(os.path.join(a.ckpt, "text_encoder"), torch_dtype=dt).to(dev)
        pipe = CogVideoXImageToVideoPipeline.from_pretrained(a.ckpt, text_encoder=text_encoder, tokenizer=tokenizer,
                                                             transformer=transformer, vae=vae, torch_dtype=dt)
        pipe.to("cuda")
        text_dim = transformer.config.text_embed_dim
    else:
        from frameino_amd.configs import COGVIDEOX_5B_FRAMEINO_CFG
        from frameino_amd.random_init import random_cog_model
        cfg = dict(COGVIDEOX_5B_FRAMEINO_CFG)
        vae_kw = {}
        if a.smoke:
            a.height, a.width, a.frames, a.steps = 64, 96, 9, min(a.steps, 3)
            cfg.update(num_attention_heads=2, num_layers=2, text_embed_dim=64, time_embed_dim=64,
                       sample_height=a.height // 8, sample_width=a.width // 8, sample_frames=a.frames)
            vae_kw = dict(block_out_channels=(32, 64, 64, 128), layers_per_block=1, norm_num_groups=8)
        transformer = random_cog_model(cfg, dev, dtype=dt)
        vae = AutoencoderKLCogVideoX(**vae_kw).random_init_(seed=2, device=dev, dtype=dt)
        if not a.no_tiling:
            # as the evaluation script does (test_code/run_cogvideox_FrameIn_mass_evaluation.py:95-96): at 480 x 720 the tiles
            # (240 x 360, overlapping) are active in every encode and in the decode
            vae.enable_slicing()
            vae.enable_tiling()
        text_dim = cfg["text_embed_dim"]
        pipe = CogVideoXImageToVideoPipeline(vae=vae, transformer=transformer, scheduler=sched)
    if a.mxfp8:
        transformer.enable_mxfp8_linears()
    if a.fp8_attention:
        transformer.enable_fp8_attention()

    t0 = time.perf_counter()
    image, traj, id_tensor, pads = synthetic_conditions(a.frames, a.height, a.width, dev, a.frame_out)
    if text_encoder is None:                                   # no text encoder offline: synthetic prompt embeddings
        g = torch.Generator().manual_seed(0)
        kw = dict(prompt_embeds=torch.randn(1, 226, text_dim, generator=g).to(dev),
                  negative_prompt_embeds=torch.zeros(1, 226, text_dim, device=dev))
    else:
        kw = dict(prompt=a.prompt)
    torch.cuda.synchronize()
    tc = time.perf_counter()
    for rep in range(a.repeat):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        video = pipe(image=image, traj_tensor=traj, ID_tensor=id_tensor, height=a.height, width=a.width, num_frames=a.frames,
                     guidance_scale=a.guidance, use_dynamic_cfg=True, num_inference_steps=a.steps,
                     generator=torch.Generator().manual_seed(1234), **kw).frames[0]            # list of PIL images
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        frames = np.stack([np.asarray(f) for f in video])
        assert frames.shape == (a.frames, a.height, a.width, 3) and frames.dtype == np.uint8
        cond_s = f"conditions {tc - t0:.2f} s, " if rep == 0 else ""
        mode = ("mxfp8 linears" if a.mxfp8 else f"{a.dtype} linears") + (" + fp8 attention" if a.fp8_attention else "")
        print(f"{cond_s}clip ({a.frames} frames {a.height}x{a.width}, {a.steps} steps, {a.scheduler}, {mode}) {t2 - t1:.2f} s"
              f"{' (cold)' if rep == 0 and a.repeat > 1 else ''}, frames in [{frames.min()}, {frames.max()}], peak device memory "
              f"{torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
    top, left, bottom, right = pads
    region = frames[:, top:a.height - bottom, left:a.width - right]        # :219-224 of the evaluation script: the un-extended region
    print(f"cropped region {tuple(region.shape)} uint8")
    if a.out:
        np.save(a.out, region)


if __name__ == "__main__":
    main()
