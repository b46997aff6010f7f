#!/usr/bin/env python3
"""Energy ledger of the headline denoise step: joules per step by kernel class.

The step runs at the board's power cap, so what moves it is joules, not matrix-pipe duty cycle.  For
every kernel class of the Wan2.2-5B step at the bench shapes (batch-2 forward: M = 24640 rows, L = 12320) this tool loops the
launch for ~2.5 s, samples the socket power (rocm-smi, 10 Hz) and reports

    us per launch | board W | J per launch = W x t | dynamic J = (W - idle W) x t | launches per step | J per step
    and, for the matrix kernels, pJ per algorithmic FLOP beside the pure-MFMA loop's (fino_diag_mfma_peak on gaussian operands)

then times the whole step the same way, so that the classes can be added up against it.  One GPU, ~1.5 min.

    python tools/energy_ledger.py [seconds per class] > profiles/rNN_energy_ledger.txt"""
import ctypes
import json
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from frameino_amd import _lib, ops  # noqa: E402

SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5
dev = torch.device("cuda")
lib = _lib.load()
g = torch.Generator(device=dev).manual_seed(0)


def smi_loop(samples, stop):
    while not stop[0]:
        try:
            dd = json.loads(subprocess.run(["rocm-smi", "--showpower", "--json"], capture_output=True, text=True, timeout=10).stdout)
            c = dd[sorted(dd)[0]]
            samples.append((time.time(), float(c.get("Current Socket Graphics Package Power (W)", 0))))
        except Exception:      # noqa: BLE001
            pass
        time.sleep(0.1)


def measure(fn, secs=SECS, batch=10):
    """-> (us per call, mean board W over the loop without its first 0.8 s)"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    samples, stop = [], [False]
    th = threading.Thread(target=smi_loop, args=(samples, stop))
    th.start()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.time()
    cnt = 0
    s.record()
    while time.time() - t0 < secs:
        for _ in range(batch):
            fn()
        cnt += batch
        torch.cuda.synchronize()
    e.record()
    torch.cuda.synchronize()
    t1 = time.time()
    stop[0] = True
    th.join()
    busy = [p for (ts, p) in samples if t0 + 0.8 <= ts <= t1]
    return s.elapsed_time(e) / cnt * 1e3, sum(busy) / max(len(busy), 1)


def idle_power(secs=2.0):
    torch.cuda.synchronize()
    samples, stop = [], [False]
    th = threading.Thread(target=smi_loop, args=(samples, stop))
    th.start()
    time.sleep(secs)
    stop[0] = True
    th.join()
    return sum(p for _, p in samples) / max(len(samples), 1)


def main():
    from frameino_amd.configs import WAN22_5B_CFG
    L, D, F, H, dh = 12320, 3072, 14336, 24, 128
    M = 2 * L
    bf = torch.bfloat16
    rn = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).to(bf)      # noqa: E731
    p_idle = idle_power()
    print(f"# energy ledger of the Wan2.2-5B denoise step (49 f 704x1280, batch-2 forward, L = {L}); idle board power {p_idle:.0f} W; "
          f"{SECS:.1f} s per class")
    rows = []

    # ---- pure MFMA reference: pJ per FLOP of the matrix pipe alone on gaussian operands ----
    scratch = torch.zeros(64 + 2 * 256 * 4, device=dev)
    ov = scratch[64:].view(bf)
    ov.copy_(torch.randn(ov.shape, device=dev, generator=g).to(bf))
    fl = ctypes.c_double()
    stream = torch.cuda.current_stream().cuda_stream
    mf = lambda: _lib.check(lib.fino_diag_mfma_peak(0, 2, 40000, scratch.data_ptr(), ctypes.byref(fl), stream), "mfma")      # noqa: E731
    us, w = measure(mf, batch=2)
    pj_mfma = (w - p_idle) * us * 1e-6 / fl.value * 1e12
    print(f"pure MFMA loop (32x32x16 bf16, gaussian operands): {fl.value / us / 1e6:6.0f} TFLOP/s at {w:5.0f} W = "
          f"{w * us * 1e-6 / fl.value * 1e12:.3f} pJ/FLOP total, {pj_mfma:.3f} pJ/FLOP dynamic")

    x = rn(M, D)
    nrm = torch.empty_like(x)
    mod = torch.randn(2, 6, D, device=dev, generator=g) * 0.1
    sel = (torch.arange(M, device=dev) % L >= 880).to(torch.int32)
    wqkv, bqkv = rn(3 * D, D, sc=0.02), rn(3 * D)
    qkv = torch.empty(M, 3 * D, device=dev, dtype=bf)
    wo, bo = rn(D, D, sc=0.02), rn(D)
    w1, b1 = rn(F, D, sc=0.02), rn(F)
    w2, b2 = rn(D, F, sc=0.02), rn(D)
    ff = torch.empty(M, F, device=dev, dtype=bf)
    att = rn(M, D)
    nw = torch.ones(D, device=dev, dtype=bf)
    from frameino_amd.transformer_wan import wan_rope_tables
    cos1, sin1 = wan_rope_tables(dh, 1024, 14, 22, 40)
    cos, sin = cos1.repeat(2, 1).contiguous().to(dev), sin1.repeat(2, 1).contiguous().to(dev)
    ln_w, ln_b = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    kp_c, kp_u = 72, 16
    pr_c, pr_u = rn(L, H * kp_c, sc=0.1), rn(L, H * kp_u, sc=0.1)
    w2c, w2u = rn(D, H * kp_c, sc=0.02), rn(D, H * kp_u, sc=0.02)
    ktxt = rn(1, 128, D)
    rr = torch.empty(M, dtype=torch.float32, device=dev)
    ops.gemm(x, wqkv, bqkv, out=qkv)
    q3 = qkv.view(2, L, 3 * D)

    def add(name, fn, per_step, flops=0.0, nbytes=0.0, batch=10):
        us_, w_ = measure(fn, batch=batch)
        rows.append(dict(name=name, us=us_, w=w_, per_step=per_step, flops=flops, bytes=nbytes))
        j = w_ * us_ * 1e-6
        jd = (w_ - p_idle) * us_ * 1e-6
        extra = ""
        if flops:
            extra = f"  {flops / us_ / 1e6:6.0f} TFLOP/s  {jd / flops * 1e12:.3f} pJ/FLOP dynamic ({jd / flops * 1e12 / pj_mfma:.2f}x the bare MFMA loop)"
        elif nbytes:
            extra = f"  {nbytes / us_ / 1e6:6.2f} TB/s  {jd / nbytes * 1e12:.1f} pJ/byte dynamic"
        print(f"{name:46s} {us_:8.1f} us  {w_:5.0f} W  {j:6.3f} J ({jd:6.3f} dynamic) x {per_step:3d} = {j * per_step:7.2f} J "
              f"({jd * per_step:7.2f} dynamic){extra}", flush=True)

    add("self-attention (attn_ppd, B=2, 12320^2, 24x128)", lambda: ops.attention(q3[:, :, :D], q3[:, :, D:2 * D], q3[:, :, 2 * D:], H, out=att.view(2, L, D)),
        30, 4.0 * 2 * L * L * D, batch=4)
    add("q|k|v projection (N=9216, bias)", lambda: ops.gemm(x, wqkv, bqkv, out=qkv), 30, 2.0 * M * 3 * D * D)
    add("out-projection (N=K=3072, gated residual)", lambda: ops.gemm(att, wo, bo, ops.EPI_GATED_RESIDUAL, x, mod[:, 2], sel, out=nrm), 30,
        2.0 * M * D * D)
    add("cross-attention q projection (N=K=3072, bias)", lambda: ops.gemm(x, wo, bo, out=nrm), 30, 2.0 * M * D * D)
    add("FFN up + GELU (N=14336)", lambda: ops.gemm(x, w1, b1, ops.EPI_GELU_TANH, out=ff), 30, 2.0 * M * F * D)
    add("FFN down (K=14336, gated residual)", lambda: ops.gemm(ff, w2, b2, ops.EPI_GATED_RESIDUAL, x, mod[:, 5], sel, out=nrm), 30, 2.0 * M * D * F)
    add("text out-projection P.(V W_o^T), K=1728 (cond)", lambda: ops.gemm(pr_c, w2c, bo, ops.EPI_RESIDUAL, residual=x[:L], out=nrm[:L]), 30,
        2.0 * L * D * H * kp_c)
    add("text out-projection P.(V W_o^T), K=384 (uncond)", lambda: ops.gemm(pr_u, w2u, bo, ops.EPI_RESIDUAL, residual=x[:L], out=nrm[:L]), 30,
        2.0 * L * D * H * kp_u)
    add("adaLN modulate (LN + scale/shift)", lambda: ops.adaln_modulate(x, mod[:, 0], mod[:, 1], sel, 1e-6, out=nrm), 60, nbytes=2.0 * M * D * 2)
    add("LayerNorm (norm2, affine)", lambda: ops.layernorm(x, ln_w, ln_b, 1e-6, out=nrm), 30, nbytes=2.0 * M * D * 2)
    add("q|k RMSNorm + RoPE (in place, one launch)", lambda: ops.qkv_rmsnorm_rope_(qkv, D, nw, 1e-6, nw, 1e-6, cos, sin, dh), 30, nbytes=4.0 * M * D * 2)
    add("row rrms (norm_q statistic)", lambda: ops.row_rrms(x, 1e-6, out=rr), 30, nbytes=1.0 * M * D * 2)
    add("text probabilities (65 keys, cond)", lambda: ops.attention_probs(x[:L].view(1, L, D), ktxt[:, :kp_c], H, [65], [448.0], kp_c,
                                                                           out=pr_c.view(1, L, -1), q_rrms=rr[:L].view(1, L), q_weight=nw), 30,
        nbytes=L * D * 2 + L * H * kp_c * 2.0)
    add("text probabilities (9 keys, uncond)", lambda: ops.attention_probs(x[:L].view(1, L, D), ktxt[:, :kp_u], H, [9], [504.0], kp_u,
                                                                           out=pr_u.view(1, L, -1), q_rrms=rr[:L].view(1, L), q_weight=nw), 30,
        nbytes=L * D * 2 + L * H * kp_u * 2.0)
    tot = sum(r["w"] * r["us"] * 1e-6 * r["per_step"] for r in rows)
    tot_d = sum((r["w"] - p_idle) * r["us"] * 1e-6 * r["per_step"] for r in rows)
    tot_ms = sum(r["us"] * r["per_step"] for r in rows) / 1e3
    fl_all = sum(r["flops"] * r["per_step"] for r in rows)
    print(f"{'sum of the classes':46s} {tot_ms:8.1f} ms per step  {tot:7.1f} J ({tot_d:7.1f} dynamic); their algorithmic FLOPs x the bare MFMA "
          f"loop's {pj_mfma:.3f} pJ/FLOP = {fl_all * pj_mfma * 1e-12:6.1f} J dynamic: the rest, {tot_d - fl_all * pj_mfma * 1e-12:6.1f} J, is not MFMA")
    for r in sorted(rows, key=lambda r_: -((r_["w"] - p_idle) * r_["us"] * 1e-6 - r_["flops"] * pj_mfma * 1e-12) * r_["per_step"]):
        non = ((r["w"] - p_idle) * r["us"] * 1e-6 - r["flops"] * pj_mfma * 1e-12) * r["per_step"]
        print(f"    non-MFMA dynamic joules per step: {non:7.2f}  {r['name']}")

    # ---- the whole step, the same way ----
    del x, nrm, qkv, ff, att, w1, w2, wqkv
    torch.cuda.empty_cache()
    from bench import build_model
    from frameino_amd.pipeline_wan_i2v_motion_frameino import WanImageToVideoPipeline
    from frameino_amd.schedulers import FlowMatchEulerDiscreteScheduler
    cfg = dict(WAN22_5B_CFG)
    model = build_model(cfg, dev)
    pipe = WanImageToVideoPipeline(scheduler=FlowMatchEulerDiscreteScheduler(shift=5.0), transformer=model, expand_timesteps=True)
    gc = torch.Generator().manual_seed(1234)
    C, fg, lh, lw = 48, 13, 44, 80
    lat = torch.randn(1, C, fg, lh, lw, generator=gc).to(dev)
    cond = torch.randn(1, C, 1, lh, lw, generator=gc).to(dev)
    traj = torch.randn(1, C, fg + 1, lh, lw, generator=gc).to(dev)
    idl = torch.randn(1, C, 1, lh, lw, generator=gc).to(dev)
    mask = torch.ones(1, 1, fg, lh, lw, device=dev)
    mask[:, :, 0] = 0
    pe = torch.randn(1, 512, cfg["text_dim"], generator=gc)
    ne = torch.randn(1, 512, cfg["text_dim"], generator=gc)
    pe[:, 64:] = 0
    ne[:, 8:] = 0
    pipe.scheduler.set_timesteps(8, device=dev)
    st = pipe.make_state(lat, cond, traj, idl, mask, pe.to(dev).to(bf), ne.to(dev).to(bf), 5.0)
    st.t_rows[1:2] = 700.0
    st.dt[0] = -0.001
    with torch.no_grad():
        us, w = measure(lambda: pipe._step(st), secs=max(SECS, 4.0), batch=2)
    print(f"{'the whole denoise step (eager)':46s} {us / 1e3:8.1f} ms  {w:5.0f} W  {w * us * 1e-6:7.1f} J ({(w - p_idle) * us * 1e-6:7.1f} dynamic)")


if __name__ == "__main__":
    main()
