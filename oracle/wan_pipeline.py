"""Oracle restatement of the Wan FrameINO denoising loop.  Test infrastructure.

Follows /root/reference/pipelines/pipeline_wan_i2v_motion_FrameINO.py:809-913 (expand_timesteps
= Wan2.2 path, single transformer, one or more ID frames appended on the frame axis).
"""
import torch

from .wan_dit import wan_forward


def wan_denoise_loop(sd, cfg, scheduler, latents, condition, traj_latents, id_latent, first_frame_mask,
                     prompt_embeds, negative_embeds, guidance_scale, num_steps, model_dtype=torch.float32,
                     forward=None):
    """Returns final latents (after first-frame re-imposition, :912-913).
    `forward(x, timestep, text)` defaults to the oracle DiT; tests may pass another callable."""
    if forward is None:
        def forward(x, t, e):
            return wan_forward(sd, cfg, x, t, e)
    scheduler.set_timesteps(num_steps)
    n_gen = latents.shape[2]
    lh, lw = latents.shape[3], latents.shape[4]
    pe = prompt_embeds.to(model_dtype)
    ne = negative_embeds.to(model_dtype) if negative_embeds is not None else None
    for t in scheduler.timesteps:
        x = ((1 - first_frame_mask) * condition + first_frame_mask * latents).to(model_dtype)       # :829-830
        if id_latent is not None:
            pad = torch.ones(1, 1, id_latent.shape[2], lh, lw, dtype=model_dtype)
            mask_adj = torch.cat([first_frame_mask, pad], dim=2)                                    # :833-837
        else:
            mask_adj = first_frame_mask
        timestep = (mask_adj[0][0][:, ::2, ::2] * t).flatten().unsqueeze(0).expand(latents.shape[0], -1)  # :842-843
        if id_latent is not None:
            x = torch.cat([x, id_latent], dim=2)                                                    # :854
        x = torch.cat([x, traj_latents], dim=1).to(model_dtype)                                     # :858
        noise = forward(x, timestep, pe)
        if guidance_scale > 1:
            unc = forward(x, timestep, ne)
            noise = unc + guidance_scale * (noise - unc)                                            # :882
        noise = noise[:, :, :n_gen]                                                                 # :886
        latents = scheduler.step(noise, latents)                                                    # :891
    return (1 - first_frame_mask) * condition + first_frame_mask * latents                          # :913
