#!/usr/bin/env python3
"""Second backbone at BASELINE config 5's shape, bf16 or (--mxfp8) MXFP8 linears + bf16 attention: CogVideoX-5B FrameINO,
49 frames 480x720 -> model input [2, 14, 48, 60, 90] (CFG-batched, one ID frame), L = 226 + 18900 joint tokens,
42 layers, 48 heads x 64.  Random-init weights on the device, synthetic latents; one step = the B=2 forward + guidance +
v-prediction DDIM update (pipelines/pipeline_cogvideox_i2v_motion_FrameINO.py:848-944).  GPU box only."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from frameino_amd.configs import COGVIDEOX_5B_FRAMEINO_CFG as COG5B  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--mxfp8", action="store_true", help="QKV / out / FFN linears on the MXFP8 path (config 5)")
    ap.add_argument("--fp8-attention", action="store_true", help="fp8 (e4m3) attention operands (fino_attn_fwd_fp8)")
    ap.add_argument("--w4", action="store_true", help="4-wave head_dim-64 attention kernel with the softmax scale folded into q")
    a = ap.parse_args()
    from frameino_amd.pipeline_cogvideox_i2v_motion_frameino import CogVideoXImageToVideoPipeline
    from frameino_amd.schedulers import CogVideoXDDIMScheduler
    cfg = dict(COG5B)
    if a.layers:
        cfg["num_layers"] = a.layers
    dev = torch.device("cuda")
    from frameino_amd.random_init import random_cog_model
    m = random_cog_model(cfg, dev)
    g = torch.Generator(device=dev).manual_seed(1)
    if a.mxfp8:
        m.enable_mxfp8_linears()
    m.enable_fp8_attention(a.fp8_attention)
    if a.w4:
        from frameino_amd import _lib
        _lib.lib().fino_tune_set(4, 2)
        m.fold_softmax_scale = True
    pipe = CogVideoXImageToVideoPipeline(transformer=m, scheduler=CogVideoXDDIMScheduler())
    F, C, h, w = 13, 16, 60, 90
    lat = torch.randn(1, F, C, h, w, device=dev, generator=g)
    img = torch.cat([torch.randn(1, 1, C, h, w, device=dev, generator=g), torch.zeros(1, F - 1, C, h, w, device=dev)], 1)
    trj = torch.randn(1, F, C, h, w, device=dev, generator=g)
    idl = torch.randn(1, 1, C, h, w, device=dev, generator=g)
    pe = torch.randn(1, 226, 4096, device=dev, generator=g)
    ne = torch.randn(1, 226, 4096, device=dev, generator=g)
    total = a.warmup + a.steps
    seen = []

    def cb(p, i, t, kw):
        torch.cuda.synchronize()
        seen.append(time.perf_counter())
        return {}

    pipe.denoise(lat, img, trj, idl, pe, ne, 6.0, total, callback_on_step_end=cb)
    dt = (seen[-1] - seen[a.warmup - 1]) / a.steps if a.warmup else (seen[-1] - seen[0]) / (a.steps - 1)
    L, d, nl = 226 + 14 * 30 * 45, 3072, cfg["num_layers"]
    flops = 2 * nl * (8 * L * d * d + 4 * L * L * d + 16 * L * d * d)          # B=2: proj + SDPA + FFN (4x)
    print(f"CogVideoX-5B FrameINO 49f 480x720 {('mxfp8 linears + ' + ('fp8-operand' if a.fp8_attention else 'bf16') + ' attention') if a.mxfp8 else ('bf16 linears + fp8-operand attention' if a.fp8_attention else 'bf16')}{' + 4-wave folded attention' if a.w4 else ''}: {dt * 1e3:.1f} ms/step, {1 / dt:.3f} denoise-steps/s, "
          f"{flops / dt / 1e12:.0f} TFLOP/s model ({nl} layers, L={L})")


if __name__ == "__main__":
    main()
