"""Oracle restatement of the CogVideoX FrameINO denoise loop (pipelines/pipeline_cogvideox_i2v_motion_FrameINO.py
:848-944) with a v-prediction DDIM step (diffusers CogVideoXDDIMScheduler, third-party, restated: unpinned).
Test infrastructure."""
import math

import numpy as np
import torch

from .cog_dit import cog_forward


def ddim_tables(num_inference_steps, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.0120,
                snr_shift_scale=1.0):
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float64) ** 2
    ac = torch.cumprod(1.0 - betas, dim=0)
    ac = ac / (snr_shift_scale + (1 - snr_shift_scale) * ac)
    s = ac.sqrt()
    a0, at = s[0].clone(), s[-1].clone()
    ac = ((s - at) * (a0 / (a0 - at))) ** 2                        # zero-terminal-SNR rescale
    ts = np.round(np.arange(num_train_timesteps, 0, -num_train_timesteps / num_inference_steps)).astype(np.int64) - 1
    return ac, ts


def cog_denoise_loop(sd, cfg, latents, image_latents, traj_latents, id_latent, prompt_embeds, negative_embeds,
                     rotary, guidance, steps, dynamic_cfg=False, dpm_generator=None, use_dpm=False):
    """`use_dpm`: the CogVideoXDPMScheduler branch of the loop (:915-926) through oracle.schedulers.CogDPMOracle (its
    noise comes from `dpm_generator`, a CPU generator, drawn in the latent dtype like diffusers' randn_tensor)."""
    ac, ts = ddim_tables(steps)
    dpm = None
    if use_dpm:
        from .schedulers import CogDPMOracle
        dpm = CogDPMOracle()
        dpm.set_timesteps(steps)
        old = None
    nlf = latents.shape[1]
    prompt = torch.cat([negative_embeds, prompt_embeds], dim=0)                    # :768
    lat = latents.clone()
    tl = ts.tolist()
    for i, t in enumerate(tl):
        x = torch.cat([lat] * 2)
        img, trj = torch.cat([image_latents] * 2), torch.cat([traj_latents] * 2)
        if id_latent is not None:                                                  # stage 1 (no ID frame): skip
            lid = torch.cat([id_latent] * 2)
            pad = torch.zeros_like(lid)
            x = torch.cat([x, lid], dim=1)                                         # :868
            img, trj = torch.cat([img, pad], dim=1), torch.cat([trj, pad], dim=1)
        x = torch.cat([x, img, trj], dim=2)                                        # :880
        pred = cog_forward(sd, cfg, x, prompt, torch.full((2,), float(t), device=x.device), rotary).float()[:, :nlf]
        g = guidance
        if dynamic_cfg:
            g = 1 + guidance * ((1 - math.cos(math.pi * ((steps - t) / steps) ** 5.0)) / 2)
        u, c = pred.chunk(2)
        v = u + g * (c - u)
        if dpm is not None:
            lat, old = dpm.step(v, old, t, tl[i - 1] if i > 0 else None, lat, generator=dpm_generator)
            lat = lat.to(latents.dtype)                                            # :927
            continue
        prev = t - 1000 // steps
        a_t = ac[t]
        a_p = ac[prev] if prev >= 0 else torch.tensor(1.0, dtype=torch.float64)
        x0 = (a_t ** 0.5) * lat - ((1 - a_t) ** 0.5) * v
        ca = ((1 - a_p) / (1 - a_t)) ** 0.5
        cb = a_p ** 0.5 - a_t ** 0.5 * ca
        lat = (ca * lat + cb * x0).to(latents.dtype)
    return lat


def cog_conditions(vae_sd, vae_cfg, image, traj, id_tensor, num_frames, dtype=torch.float32, generator=None,
                   add_id_noise=False):
    """The three condition encodes of the FrameINO CogVideoX pipeline, through oracle/cog_vae.py:
    `prepare_latents` (pipelines/pipeline_cogvideox_i2v_motion_FrameINO.py:350-423: first frame -> posterior sample ->
    [B, F, C, h, w] x scaling factor, zero frames appended), the trajectory video (:809-817) and the identity reference
    (:820-822 -> train_code/train_cogvideox_motion_FrameINO.py:515-546).  image [1, 3, H, W] and traj [F, 3, H, W],
    id_tensor [3, H, W] in [-1, 1].  Posterior samples draw like diffusers' DiagonalGaussianDistribution.sample: the
    first-frame one from `generator`, the other two from the global RNG.  Pinned by tests/golden/cog_pipe_tiny.npz
    (recorded from the reference pipeline's own __call__)."""
    from . import cog_vae as V
    sf = vae_cfg["scaling_factor"]

    def sample(moments, gen=None):
        mean, logvar = torch.chunk(moments, 2, dim=1)
        std = torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0))
        return mean + std * torch.randn(mean.shape, generator=gen, dtype=moments.dtype)

    nlf = (num_frames - 1) // vae_cfg["temporal_compression_ratio"] + 1
    img = image.unsqueeze(2)                                                           # :387
    il = sample(V.encode_moments(vae_sd, vae_cfg, img), generator).to(dtype).permute(0, 2, 1, 3, 4)
    il = (1 / sf if vae_cfg.get("invert_scale_latents") else sf) * il                  # :393-398
    pad = torch.zeros((il.shape[0], nlf - 1) + tuple(il.shape[2:]), dtype=dtype)
    image_latents = torch.cat([il, pad], dim=1)                                        # :400-409
    tv = traj[None].permute(0, 2, 1, 3, 4)                                             # :809-811
    tl = (sample(V.encode_moments(vae_sd, vae_cfg, tv)) * sf).permute(0, 2, 1, 3, 4).contiguous().float().to(dtype)
    x = id_tensor.unsqueeze(0).unsqueeze(2)                                            # train_code :519
    if add_id_noise:                                                                   # :523-526
        sigma = torch.exp(torch.normal(mean=-3.0, std=0.5, size=(1,))).to(x.dtype)
        x = x + torch.randn_like(x) * sigma[:, None, None, None, None]
    idl = (sample(V.encode_moments(vae_sd, vae_cfg, x)) * sf).squeeze(2).contiguous().float()      # :529-543
    return image_latents, tl, idl.unsqueeze(1).to(dtype)                               # :822
