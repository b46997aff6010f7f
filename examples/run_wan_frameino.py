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


def synthetic_conditions(frames, height, width, device, seed=0):
    """Stand-ins for the app's inputs (SURVEY 8c harness rows), built the way app.py builds the real ones (:270-350,
    :582-620, :634-695) with the device-side builders of frameino_amd.conditions: a first frame area-resampled into the
    black unbounded canvas, clicked trajectories (one object entering the frame late = 'frame in') resampled by arc
    length into per-frame tracks, an identity reference scaled and zero-padded to the canvas."""
    import PIL.Image
    from frameino_amd.conditions import build_inference_canvas, prepare_id_tensor, tracks_from_trajectories
    rng = np.random.default_rng(seed)
    pad_h, pad_w = (height // 8) // 16 * 16, (width // 6) // 16 * 16
    first = rng.integers(0, 255, (360, 640, 3), dtype=np.uint8)                  # the user's photo, any size
    canvas = build_inference_canvas(first, height - 2 * pad_h, width - 2 * pad_w, pad_h, pad_w, pad_h, pad_w, device)
    uh, uw = 480, 720                                                            # the UI board the user clicks on
    clicks = [[[(uw * 0.25, uh * 0.5), (uw * 0.75, uh * 0.5)], [(uw * 0.27, uh * 0.52), (uw * 0.77, uh * 0.52)]],
              [[(-20.0, uh * 0.3), (uw * 0.55, uh * 0.3)]]]                      # second object starts outside
    tracks = tracks_from_trajectories(clicks, frames, height, width, uh, uw)
    ident = rng.integers(0, 255, (300, 200, 3), dtype=np.uint8)                  # already-masked reference, portrait
    id_tensor = prepare_id_tensor(ident, height, width, "Wan", device)           # [1, 3, 1, H, W] in [-1, 1]
    return PIL.Image.fromarray(canvas.cpu().numpy()), tracks, id_tensor, (pad_h, pad_w, pad_h, pad_w)


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
    ap.add_argument("--dtype", choices=["fp16", "bf16"], default="fp16",
                    help="the DiT's dtype.  Default fp16 = what the reference app loads it in (app.py:156: "
                         "`WanTransformer3DModel.from_pretrained(..., torch_dtype=torch.float16)`, fp32 islands kept); bf16 is what "
                         "bench.py's headline times (the two run within 1 %% of each other)")
    ap.add_argument("--vae-fp32", type=int, nargs="?", const=3, default=0, metavar="PLANES",
                    help="run the VAE like the reference app does (app.py:157 loads it in fp32): fp32-compute mode on split-bf16 "
                         "products, 3 (default) or 2 bf16 planes per fp32 operand -- `vae.set_compute_dtype(torch.float32)`")
    a = ap.parse_args()

    from frameino_amd import _lib
    from frameino_amd.autoencoder_kl_wan import AutoencoderKLWan
    from frameino_amd.conditions import prepare_traj_tensor
    from frameino_amd.pipeline_wan_i2v_motion_frameino import WanImageToVideoPipeline
    from frameino_amd.schedulers import FlowMatchEulerDiscreteScheduler, UniPCMultistepScheduler
    _lib.load()
    dev = torch.device("cuda")
    dit_dtype = torch.float16 if a.dtype == "fp16" else torch.bfloat16
    sched = UniPCMultistepScheduler(flow_shift=5.0) if a.scheduler == "unipc" else FlowMatchEulerDiscreteScheduler(shift=5.0)

    tokenizer = text_encoder = None
    if a.ckpt:
        from frameino_amd.autoencoder_kl_wan import AutoencoderKLWan as _V
        from frameino_amd.configs import WAN22_5B_CFG, WAN22_VAE_CFG
        from frameino_amd.loading import config_report, load_scheduler, load_wan_transformer, load_wan_vae, read_config
        from frameino_amd.transformer_wan import WanTransformer3DModel as _T
        # the hyper-parameters this package assumes offline were written from memory: say, key by
        # key, where the checkpoint disagrees -- the checkpoint wins, a surprise here is worth knowing before the clip
        for sub, assumed, cls in (("transformer", WAN22_5B_CFG, _T), ("vae", WAN22_VAE_CFG, _V)):
            for line in config_report(read_config(os.path.join(a.ckpt, sub)), assumed, cls, f"{sub}/config.json"):
                print("[config]", line)
        transformer = load_wan_transformer(os.path.join(a.ckpt, "transformer"), dit_dtype, dev)
        # (app.py:157 loads the VAE in fp32: with --vae-fp32 the interface dtype is fp32 too, not only the arithmetic)
        vae = load_wan_vae(os.path.join(a.ckpt, "vae"), torch.float32 if a.vae_fp32 else torch.bfloat16, dev)
        if os.path.isfile(os.path.join(a.ckpt, "scheduler", "scheduler_config.json")):
            sched = load_scheduler(os.path.join(a.ckpt, "scheduler"))
            print("[config] scheduler from the checkpoint:", type(sched).__name__, dict(sched.config))
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
        transformer = build_model(cfg, dev, dtype=dit_dtype)
        text_dim = cfg["text_dim"]

    if a.vae_fp32:
        vae.set_compute_dtype(torch.float32, planes=a.vae_fp32)
    pipe = WanImageToVideoPipeline(tokenizer=tokenizer, text_encoder=text_encoder, vae=vae, scheduler=sched,
                                   transformer=transformer, expand_timesteps=True)
    t0 = time.perf_counter()
    canvas, tracks, id_tensor, pads = synthetic_conditions(a.frames, a.height, a.width, dev)
    traj = prepare_traj_tensor(tracks, a.height, a.width, 6, a.width, a.height, device=dev)        # [F, 3, H, W]
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
    from frameino_amd.conditions import crop_unpadded
    region = crop_unpadded(frames, *pads)                      # app.py:741-748: the original (un-extended) region
    print(f"cropped region {tuple(region.shape)} uint8")
    if a.out:
        np.save(a.out, frames)


if __name__ == "__main__":
    main()
