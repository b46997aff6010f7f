#!/usr/bin/env python3
"""Full-size Wan2.2 VAE decode / encode timing on one MI355X (random weights, synthetic latents)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frameino_amd import ops
from frameino_amd.autoencoder_kl_wan import AutoencoderKLWan

WAN22_VAE = dict(base_dim=160, decoder_base_dim=256, z_dim=48, dim_mult=[1, 2, 4, 4], num_res_blocks=2,
                 temperal_downsample=[False, True, True], is_residual=True, in_channels=12, out_channels=12,
                 patch_size=2, scale_factor_temporal=4, scale_factor_spatial=16)
ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=13)
ap.add_argument("--h", type=int, default=44)
ap.add_argument("--w", type=int, default=80)
ap.add_argument("--encode", action="store_true")
ap.add_argument("--cog", action="store_true", help="CogVideoX VAE (49 f 480x720: latent 13 x 60 x 90) instead of the Wan one")
ap.add_argument("--fp32", type=int, default=0, metavar="PLANES",
                help="Wan VAE in the fp32-compute mode (set_compute_dtype(torch.float32, planes=2|3): split-bf16 products)")
ap.add_argument("--slabs", type=int, default=0, metavar="N",
                help="Wan decode: time what ONE rank of an N-rank sharded decode does (slab 0 .. N-1 each, "
                     "AutoencoderKLWan.decode_slab: replicated head of the decoder + the tail on a slab with its halo)")
ap.add_argument("--tiling", action="store_true", help="--cog: diffusers' tiled encode / decode (enable_tiling())")
a = ap.parse_args()
dev = torch.device("cuda")
if a.cog:
    from frameino_amd.autoencoder_kl_cogvideox import AutoencoderKLCogVideoX
    vae = AutoencoderKLCogVideoX().random_init_(seed=0, device=dev)
    if a.tiling:
        vae.enable_tiling()
    h, w = (60, 90) if (a.h, a.w) == (44, 80) else (a.h, a.w)
    z = torch.randn(1, 16, a.frames, h, w, device=dev)
    for it in range(2):
        torch.cuda.reset_peak_memory_stats(); torch.cuda.synchronize(); t0 = time.time()
        with ops.KernelTimer({"conv3d"}) as kt:
            out = vae.decode(z).sample
        torch.cuda.synchronize(); dt = time.time() - t0
        s = kt.summary().get("conv3d", {})
        print(f"cog decode {tuple(z.shape)} -> {tuple(out.shape)}: {dt:.3f} s; conv launches {s.get('launches')} total "
              f"{s.get('total_ms', 0):.1f} ms, {kt.flops.get('conv3d', 0) / max(s.get('total_ms', 1), 1e-9) / 1e9:.1f} TFLOP/s "
              f"(padded FLOPs {kt.flops.get('conv3d', 0):.3e}); peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
    assert torch.isfinite(out.float()).all()
    vid = torch.rand(1, 3, 1 + 4 * (a.frames - 1), h * 8, w * 8, device=dev) * 2 - 1
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        m = vae.encode(vid).latent_dist.mode()
        torch.cuda.synchronize()
        print(f"cog encode {tuple(vid.shape)} -> {tuple(m.shape)}: {time.time() - t0:.3f} s", flush=True)
    sys.exit(0)
vae = AutoencoderKLWan(**WAN22_VAE).random_init_(seed=0, device=dev)
if a.fp32:
    vae.set_compute_dtype(torch.float32, planes=a.fp32)
z = torch.randn(1, 48, a.frames, a.h, a.w, device=dev)
for it in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    with ops.KernelTimer({"conv3d"}) as kt:
        out = vae.decode(z, return_dict=False)[0]
    torch.cuda.synchronize(); dt = time.time() - t0
    s = kt.summary().get("conv3d", {})
    print(f"decode {tuple(z.shape)} -> {tuple(out.shape)}: {dt:.3f} s; conv launches {s.get('launches')} "
          f"total {s.get('total_ms', 0):.1f} ms, {kt.flops.get('conv3d', 0) / max(s.get('total_ms', 1), 1e-9) / 1e9:.1f} TFLOP/s "
          f"(padded FLOPs {kt.flops.get('conv3d', 0):.3e}); peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
assert torch.isfinite(out).all()
if a.slabs:
    for i in range(a.slabs):
        for it in range(2):
            torch.cuda.synchronize(); t0 = time.time()
            part, geo = vae.decode_slab(z, i, a.slabs)
            torch.cuda.synchronize(); dt = time.time() - t0
        print(f"decode_slab {i} of {a.slabs}: rows [{geo[0]}, {geo[1]}) of {geo[3]}: {dt:.3f} s", flush=True)
        assert torch.equal(part, out[:, :, :, geo[0]:geo[1]])
if a.encode:
    vid = torch.rand(1, 3, 1 + 4 * (a.frames - 1), a.h * 16, a.w * 16, device=dev) * 2 - 1
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        m = vae.encode(vid).latent_dist.mode()
        torch.cuda.synchronize()
        print(f"encode {tuple(vid.shape)} -> {tuple(m.shape)}: {time.time() - t0:.3f} s", flush=True)
