#!/usr/bin/env python3
"""Build container only (needs /root/reference): what the REFERENCE's `enable_tiling()` does on the Wan2.2-style VAE
(patch_size=2, is_residual=True -- the VAE of FrameINO's Wan path) and on a Wan2.1-style one (no patchify, plain blocks).
Output kept in profiles/r04_ref_vae_tiling_probe.txt; frameino_amd/autoencoder_kl_wan.py::enable_tiling cites it."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "diffusers_stub")); sys.path.insert(0, "/root/reference"); os.chdir("/root/reference")
import torch
torch.set_grad_enabled(False)
from architecture.autoencoder_kl_wan import AutoencoderKLWan
V22 = dict(base_dim=8, decoder_base_dim=16, z_dim=4, dim_mult=[1, 2, 4, 4], num_res_blocks=2, attn_scales=[],
                temperal_downsample=[False, True, True], dropout=0.0, latents_mean=[0.1, -0.2, 0.3, 0.05],
                latents_std=[1.1, 0.9, 1.3, 0.7], is_residual=True, in_channels=12, out_channels=12, patch_size=2,
                scale_factor_temporal=4, scale_factor_spatial=16)
V21 = dict(V22, is_residual=False, in_channels=3, out_channels=3, patch_size=None, scale_factor_spatial=8, decoder_base_dim=8)
for name, cfg in (("2.2", V22), ("2.1", V21)):
    torch.manual_seed(0)
    vae = AutoencoderKLWan(**cfg).eval()
    vae.enable_tiling(tile_sample_min_height=32, tile_sample_min_width=32, tile_sample_stride_height=24, tile_sample_stride_width=24)
    x = torch.rand(1, 3, 5, 64, 96) * 2 - 1
    try:
        e = vae.encode(x).latent_dist.parameters
        print(name, "tiled encode ->", tuple(e.shape))
    except Exception as ex:
        print(name, "tiled encode FAILS:", type(ex).__name__, str(ex)[:150])
    sf = cfg["scale_factor_spatial"]
    z = torch.randn(1, 4, 2, 64 // sf * (2 if name == "2.2" else 1), 96 // sf * (2 if name=="2.2" else 1))
    try:
        d = vae.decode(z, return_dict=False)[0]
        print(name, "tiled decode", tuple(z.shape), "->", tuple(d.shape), float(d.abs().max()))
        vae.disable_tiling()
        d0 = vae.decode(z, return_dict=False)[0]
        print(name, "untiled decode ->", tuple(d0.shape))
    except Exception as ex:
        print(name, "tiled decode FAILS:", type(ex).__name__, str(ex)[:150])
