#!/usr/bin/env python3
"""Headline benchmark: denoise-steps/s of the FrameINO Wan2.2-TI2V-5B pipeline, 49 frames at 704x1280
("720p": 720 is not a legal Wan2.2-5B height), bf16, synthetic latents, random-init weights.

One "step" = what the reference loop does per iteration (pipelines/pipeline_wan_i2v_motion_FrameINO.py:809-908):
model-input assembly, cond + uncond DiT forward (L = 14 x 22 x 40 = 12320 tokens, one ID frame), CFG, Euler update.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0).  N > 1 shards the SAME clip (strong scaling): CFG branches and/or token shards with a
K/V all-gather over RCCL per attention layer (frameino_amd/parallel.py).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

WORKLOADS = {
    # name: (latent frames generated, latent h, latent w)
    "wan2.2-5b-49f-704x1280": (13, 44, 80),
    "wan2.2-5b-49f-1024x1792": (13, 64, 112),
    "tiny": (3, 8, 12),
}


def wan_flops_per_forward(L, cfg, text_len=512):
    """algorithmic FLOPs of one DiT forward (2.M.N.K), time-MLP on 2 rows."""
    d = cfg["num_attention_heads"] * cfg["attention_head_dim"]
    f, nl = cfg["ffn_dim"], cfg["num_layers"]
    kin = cfg["in_channels"] * 4
    per_layer = 8 * L * d * d + 4 * L * L * d + (4 * L * d * d + 4 * text_len * d * d) + 4 * L * text_len * d + 4 * L * d * f
    return nl * per_layer + 2 * L * kin * d + 2 * L * d * cfg["out_channels"] * 4 + 2 * text_len * (cfg["text_dim"] * d + d * d)


def build_model(cfg, device, seed=0):
    """Random-init Wan2.2-5B (no checkpoints offline): N(0, 0.02^2) weights generated on the device."""
    from frameino_amd.transformer_wan import WanTransformer3DModel
    torch.manual_seed(seed)
    with torch.device("meta"):
        m = WanTransformer3DModel(**cfg)
    m = m.to_empty(device=device)
    g = torch.Generator(device=device).manual_seed(seed)
    keep = WanTransformer3DModel._keep_in_fp32_modules
    with torch.no_grad():
        for name, p in m.named_parameters():
            if name.endswith("norm_q.weight") or name.endswith("norm_k.weight") or name.endswith("norm2.weight"):
                t = 1.0 + 0.05 * torch.randn(p.shape, generator=g, device=device)
            elif "scale_shift_table" in name:
                t = torch.randn(p.shape, generator=g, device=device) / p.shape[-1] ** 0.5
            else:
                t = 0.02 * torch.randn(p.shape, generator=g, device=device)
            p.data = t.to(torch.float32 if any(k in name for k in keep) else torch.bfloat16)
    return m.eval()


def cpu_baseline(cfg, L, budget_s=25.0):
    """The oracle (CPU restatement, `kind: port`) timed on this box's host cores on a bounded sample: ONE full-size
    WanTransformerBlock in fp32 at a token count sized to the budget, extrapolated to steps/s =
    1 / (2 forwards x layers x t_block(L)).  Reported baseline only."""
    from oracle import wan_dit as W
    threads = os.cpu_count() or 1
    torch.set_num_threads(threads)
    one = dict(cfg, num_layers=1)
    sd = W.wan_random_state_dict(one, seed=1, dtype=torch.float32)
    d = cfg["num_attention_heads"] * cfg["attention_head_dim"]
    Ls = min(L, 3080)                      # quarter of the sequence: ~0.7 TFLOP, seconds on host cores
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, Ls, d, generator=g)
    txt = torch.randn(1, 512, d, generator=g)
    temb = torch.randn(1, 6, d, generator=g) * 0.1
    rot = W.wan_rope(cfg["attention_head_dim"], 1024, Ls // 880 if Ls >= 880 else 1, 22 if Ls >= 880 else 1,
                     40 if Ls >= 880 else Ls)
    if rot[0].shape[2] != Ls:
        rot = (rot[0][:, :, :1].expand(1, 1, Ls, -1).contiguous(), rot[1][:, :, :1].expand(1, 1, Ls, -1).contiguous())
    t0 = time.time()
    reps = 0
    while True:
        W.wan_block(sd, "blocks.0", one, x, txt, temb, rot)
        reps += 1
        if time.time() - t0 > budget_s * 0.5 or reps >= 3:
            break
    t_blk = (time.time() - t0) / reps
    # per-block FLOPs at Ls and at L -> scale the measured time by the FLOP ratio (attention is quadratic)
    f = cfg["ffn_dim"]
    fl = lambda n: 8 * n * d * d + 4 * n * n * d + 4 * n * d * d + 4 * 512 * d * d + 4 * n * 512 * d + 4 * n * d * f  # noqa
    t_full = t_blk * fl(L) / fl(Ls)
    steps_s = 1.0 / (2 * cfg["num_layers"] * t_full)
    return {"value": steps_s, "unit": "denoise-steps/s", "cores": threads, "kind": "port",
            "sample": f"oracle WanTransformerBlock fp32, D={d} F={f}, L={Ls} ({reps} reps, {t_blk:.2f}s each, "
                      f"{fl(Ls) / t_blk / 1e9:.0f} GFLOP/s), extrapolated by FLOPs to L={L} x {cfg['num_layers']} "
                      f"layers x 2 forwards"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="wan2.2-5b-49f-704x1280", choices=sorted(WORKLOADS))
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured hipGraph")
    ap.add_argument("--cfg-streams", action="store_true", help="CFG branches on two concurrent streams (A/B)")
    ap.add_argument("--plan", choices=["auto", "split", "interleave"], default="auto",
                    help="N>1: cfg x token split, both CFG branches interleaved on token shards, or probe both (>= 4 GPUs)")
    ap.add_argument("--mxfp8", action="store_true",
                    help="NOT the headline: large linears on the MXFP8 path (BASELINE config 5 style), attention in bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-vae", action="store_true", help="skip the once-per-clip VAE encode/decode timing")
    ap.add_argument("--layers", type=int, default=None, help="debug only: fewer layers (result marked invalid)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    backend = os.environ.get("FINO_DIST_BACKEND", "nccl")      # "gloo": rehearsal of the N>1 flow on fewer GPUs than ranks
    local = local % max(torch.cuda.device_count(), 1) if backend != "nccl" else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from frameino_amd import _lib, ops
    _lib.load()                                   # no fallback: fail here if the HIP library is missing
    from frameino_amd.configs import WAN22_5B_CFG
    from frameino_amd.pipeline_wan_i2v_motion_frameino import WanImageToVideoPipeline
    from frameino_amd.schedulers import FlowMatchEulerDiscreteScheduler

    cfg = dict(WAN22_5B_CFG)
    if a.workload == "tiny":
        cfg.update(num_attention_heads=2, num_layers=2, ffn_dim=512, text_dim=64, in_channels=8, out_channels=4)
    if a.layers:
        cfg["num_layers"] = a.layers
    fg, lh, lw = WORKLOADS[a.workload]
    nid = 1
    C = cfg["out_channels"]
    L = (fg + nid) * (lh // 2) * (lw // 2)
    model = build_model(cfg, dev)
    if a.mxfp8:
        model.enable_mxfp8_linears()
    pipe = WanImageToVideoPipeline(scheduler=FlowMatchEulerDiscreteScheduler(shift=5.0), transformer=model,
                                   expand_timesteps=True)
    plans = {}
    if world > 1:
        from frameino_amd.parallel import make_plan, shard_pipeline
        # every rank creates every communicator, in the same order
        if a.plan in ("auto", "split"):
            plans["split"] = make_plan(rank, world, True, "split")
        if a.plan == "interleave" or (a.plan == "auto" and world >= 4):
            plans["interleave"] = make_plan(rank, world, mode="interleave")
        shard_pipeline(pipe, rank, world, plan=next(iter(plans.values())))
    pipe.use_hip_graph = a.graph
    pipe.cfg_streams = a.cfg_streams

    g = torch.Generator().manual_seed(1234)           # CPU generator, then copy
    lat = torch.randn(1, C, fg, lh, lw, generator=g).to(dev)
    cond = torch.randn(1, C, 1, lh, lw, generator=g).to(dev)
    traj = torch.randn(1, C, fg + nid, lh, lw, generator=g).to(dev)
    traj[:, :, fg:] = 0
    idl = torch.randn(1, C, nid, lh, lw, generator=g).to(dev)
    mask = torch.ones(1, 1, fg, lh, lw, device=dev)
    mask[:, :, 0] = 0
    pe = torch.randn(1, 512, cfg["text_dim"], generator=g)
    pe[:, 64:] = 0                                     # zero padding past the prompt length (:235-238)
    ne = torch.randn(1, 512, cfg["text_dim"], generator=g)
    ne[:, 8:] = 0
    pe, ne = pe.to(dev).bfloat16(), ne.to(dev).bfloat16()

    total = a.warmup + a.steps
    pipe.scheduler.set_timesteps(max(total, 2), device=dev)
    st = pipe.make_state(lat, cond, traj, idl, mask, pe, ne, 5.0)
    ts, dts = pipe.scheduler.timesteps.to(dev).float(), pipe.scheduler.dts.to(dev)

    graph = None

    def one_step(i):
        nonlocal graph
        st.t_rows[1:2].copy_(ts[i:i + 1])
        st.dt.copy_(dts[i:i + 1])
        if a.graph:
            if graph is None:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    pipe._step(st)
            graph.replay()
        else:
            pipe._step(st)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if len(plans) > 1:
        # probe: one warm + two timed steps per plan on this node, MAX over ranks, keep the faster plan.  (With one
        # CFG branch per rank nothing can overlap the K|V all-gather; with both branches per rank it can hide, at
        # the price of smaller GEMMs -- which wins depends on the node's xGMI, so it is measured, not assumed.)
        snap = st.lat.clone()
        best = None
        probe_ms = {}
        with torch.no_grad():
            for name, plan in plans.items():
                shard_pipeline(pipe, rank, world, plan=plan)
                st.t_rows[1:2].copy_(ts[0:1])
                st.dt.copy_(dts[0:1])
                pipe._step(st)
                sync()
                t0 = time.perf_counter()
                pipe._step(st)
                pipe._step(st)
                sync()
                tt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                probe_ms[plan.desc] = tt.item() / 2 * 1e3
                if rank == 0:
                    print(f"[bench] plan {plan.desc}: {tt.item() / 2 * 1e3:.1f} ms/step (probe)", file=sys.stderr, flush=True)
                if best is None or tt.item() < best[0]:
                    best = (tt.item(), plan)
                st.lat.copy_(snap)
        shard_pipeline(pipe, rank, world, plan=best[1])
        pipe.plan_probe_ms = probe_ms
    with torch.no_grad():
        for i in range(a.warmup):
            if a.graph and i == 0:
                snap = st.lat.clone()
                pipe._step(st)
                st.lat.copy_(snap)
            one_step(i)
        sync()
        timer = ops.KernelTimer({"attn_self"} if not a.graph else set())
        t0 = time.perf_counter()
        with timer:
            for i in range(a.warmup, total):
                one_step(i)
        sync()
        elapsed = time.perf_counter() - t0
    if world > 1:                                     # MAX over ranks, before rank 0 goes off to time the VAE
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    # ---- once-per-clip stages (rank 0, outside the timed region): Wan VAE encode of the conditions + decode ----
    vae_times = None
    if rank == 0 and a.workload != "tiny" and not a.no_vae:
        from frameino_amd.autoencoder_kl_wan import AutoencoderKLWan
        from frameino_amd.configs import WAN22_VAE_CFG
        vae = AutoencoderKLWan(**WAN22_VAE_CFG).random_init_(seed=2, device=dev)
        vid = torch.rand(1, 3, 1 + 4 * (fg - 1), lh * 16, lw * 16, device=dev) * 2 - 1
        with torch.no_grad():
            for rep in range(2):                          # first pass warms the kernels
                torch.cuda.synchronize(); t1 = time.perf_counter()
                vae.encode(vid).latent_dist.mode()        # trajectory video (49 frames)
                vae.encode(vid[:, :, :1]); vae.encode(vid[:, :, :1])   # first frame + ID frame
                torch.cuda.synchronize(); t2 = time.perf_counter()
                vae.decode(st.lat[None], return_dict=False)
                torch.cuda.synchronize(); t3 = time.perf_counter()
        vae_times = (t2 - t1, t3 - t2)
        del vae, vid
    assert torch.isfinite(st.lat).all(), "non-finite latents"

    if rank == 0:
        ms_step = elapsed / a.steps * 1e3
        flops_step = 2 * wan_flops_per_forward(L, cfg)
        out = {
            "metric": "denoise-steps/sec (Wan2.2-5B FrameINO, 49f 704x1280, cond+uncond DiT forward + CFG + Euler)",
            "value": a.steps / elapsed, "unit": "denoise-steps/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None,
            "dtype": "bf16" if not a.mxfp8 else "mxfp8 linears (e4m3 + e8m0/32) + bf16 attention -- not the headline",
            "data": "synthetic",
            "config": {"workload": a.workload, "tokens": L, "layers": cfg["num_layers"], "guidance": 5.0,
                       "id_frames": nid, "hip_graph": bool(a.graph),
                       "parallelism": getattr(pipe, "parallel_desc", "single"),
                       "sec_per_50_step_clip_denoise_only": 50 * ms_step / 1e3,
                       "model_tflops_per_s": flops_step / (ms_step * 1e-3) / 1e12},
        }
        if getattr(pipe, "plan_probe_ms", None):
            out["config"]["plan_probe_ms_per_step"] = pipe.plan_probe_ms
        if vae_times is not None:
            enc_s, dec_s = vae_times
            out["config"].update({"vae_encode_conditions_s": enc_s, "vae_decode_s": dec_s,
                                  "sec_per_clip_50_steps": enc_s + 50 * ms_step / 1e3 + dec_s})
        if a.layers:
            out["config"]["INVALID_reduced_layers"] = a.layers
        ks = timer.summary().get("attn_self") if not a.graph else None
        heads, dh = cfg["num_attention_heads"], cfg["attention_head_dim"]
        if ks:
            traffic, traffic_src = profiled_traffic("attn_pp_kernel<BF16, 128, 0>")
            # algorithmic FLOPs of the timed launches (4.Lq.Lk.H.Dh per batch element) / their summed duration
            total_fl = timer.flops["attn_self"]
            ach = total_fl / (ks["total_ms"] * 1e-3) / 1e12
            out["roofline"] = {"bound": "mfma", "kernel": "attn_pp_kernel<BF16,128,0> (3D self-attention)",
                               "achieved": ach, "peak": 2500.0, "unit": "TFLOP/s", "frac": ach / 2500.0,
                               "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
                               "launches": ks["launches"], "avg_us": ks["avg_us"],
                               "flops_per_launch": total_fl / ks["launches"],
                               "batch_per_launch": 2 if (pipe.batch_cfg and world == 1) else 1}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, L)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def profiled_traffic(kernel_substr):
    """HBM bytes per launch of the dominant kernel from the committed PMC summary of this same command
    (tools/profile_bench.sh: separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes; counters are in KiB and
    gfx950's FETCH_SIZE under-counts by 2x -- MI355X_MICROARCH.md, HBM section).  PMC passes cannot run inside the
    timed bench, so this is the last profiled value, or None when no summary is in the tree."""
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    for f in sorted(glob.glob(os.path.join(here, "profiles", "r*_summary.json")), reverse=True):
        try:
            pmc = json.load(open(f)).get("pmc_avg_per_dispatch", {})
        except (OSError, ValueError):
            continue
        for name, c in pmc.items():
            if kernel_substr in name and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                return (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0, "profiles/" + os.path.basename(f)
    return None, None


if __name__ == "__main__":
    main()
