#!/usr/bin/env python3
"""The block GEMMs' energy by phase: joules per launch of the product kernel and of its drop-one builds.

The denoise step is power-capped, so what moves it is joules; the ledger (tools/energy_ledger.py) says the GEMM classes
cost 0.82 - 0.92 pJ per FLOP against 0.57 for the bare MFMA loop -- 58 J per step that are not matrix instructions.  This tool
splits that remainder.  For ONE library build (FINO_LIB_PATH; the wrong-result builds come from tools/debug/mkvar.sh --experiments and
need FINO_ALLOW_EXPERIMENT=1) it loops each block GEMM of the bench shape (M = 24640) for SECS seconds under a 10 Hz rocm-smi power
sampler and prints one machine-readable line per GEMM:

    <lib> | <gemm> | us | W | J | dynamic J = (W - idle W) x t

`--raster`: the product kernel under every raster group height (fino_tune_set(FINO_TUNE_GEMM_GROUP_M, G)) -- the L2-fill traffic of a
launch moves 2.2x between G = 1 and G = 4 at nearly equal time (profiles/r02_gemm_raster.md has the PMC bytes): what a GB of
fabric traffic costs in joules.  `tools/debug/gemm_energy_dropone.sh` runs it over the builds and prints the table."""
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

SECS = float(os.environ.get("FINO_ENERGY_SECS", "2.5"))
dev = torch.device("cuda")
lib = _lib.load()
g = torch.Generator(device=dev).manual_seed(0)
NAME = os.path.basename(_lib.LIB_PATH).replace("libframeino_", "").replace(".so", "")


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
    M, D, F, L = 24640, 3072, 14336, 12320
    bf = torch.bfloat16
    rn = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).to(bf)      # noqa: E731
    p_idle = idle_power()
    print(f"# {NAME}: idle {p_idle:.0f} W, {SECS:.1f} s per GEMM", flush=True)
    if NAME == "hip":
        scratch = torch.zeros(64 + 2 * 256 * 4, device=dev)
        ov = scratch[64:].view(bf)
        ov.copy_(torch.randn(ov.shape, device=dev, generator=g).to(bf))
        fl = ctypes.c_double()
        stream = torch.cuda.current_stream().cuda_stream
        us, w = measure(lambda: _lib.check(lib.fino_diag_mfma_peak(0, 2, 40000, scratch.data_ptr(), ctypes.byref(fl), stream), "mfma"), batch=2)
        print(f"{NAME} | bare MFMA loop | {us:.1f} | {w:.0f} | {w * us * 1e-6:.4f} | {(w - p_idle) * us * 1e-6:.4f} | "
              f"pJ/FLOP dynamic {(w - p_idle) * us * 1e-6 / fl.value * 1e12:.4f}", flush=True)
    x, att = rn(M, D), rn(M, D)
    nrm = torch.empty_like(x)
    mod = torch.randn(2, 6, D, device=dev, generator=g) * 0.1
    sel = (torch.arange(M, device=dev) % L >= 880).to(torch.int32)
    wqkv, bqkv = rn(3 * D, D, sc=0.02), rn(3 * D)
    qkv = torch.empty(M, 3 * D, device=dev, dtype=bf)
    wo, bo = rn(D, D, sc=0.02), rn(D)
    w1, b1 = rn(F, D, sc=0.02), rn(F)
    w2, b2 = rn(D, F, sc=0.02), rn(D)
    # the FFN-down operand: what the PRODUCT build's FFN-up would leave -- gelu(N(0, 1.1^2)) -- made by torch, so that every build
    # (a wrong-result one would leave zeros or garbage here, and zero operands draw less power) multiplies the same bits
    ff = torch.empty(M, F, device=dev, dtype=bf)
    for r0 in range(0, M, 4096):
        ff[r0:r0 + 4096] = torch.nn.functional.gelu(torch.randn(min(4096, M - r0), F, device=dev, generator=g) * 1.1, approximate="tanh").to(bf)
    ff_out = torch.empty(M, F, device=dev, dtype=bf)
    gemms = [("qkv (N=9216, bias)", lambda: ops.gemm(x, wqkv, bqkv, out=qkv), 2.0 * M * 3 * D * D),
             ("out-proj (gated residual)", lambda: ops.gemm(att, wo, bo, ops.EPI_GATED_RESIDUAL, x, mod[:, 2], sel, out=nrm), 2.0 * M * D * D),
             ("ffn-up + GELU (N=14336)", lambda: ops.gemm(x, w1, b1, ops.EPI_GELU_TANH, out=ff_out), 2.0 * M * F * D),
             ("ffn-down (K=14336, gated residual)", lambda: ops.gemm(ff, w2, b2, ops.EPI_GATED_RESIDUAL, x, mod[:, 5], sel, out=nrm), 2.0 * M * D * F)]
    groups = [0] if "--raster" not in sys.argv else [1, 2, 4, 8, 16]
    for gm in groups:
        lib.fino_tune_set(0, gm)
        for name, fn, flops in gemms:
            us, w = measure(fn)
            tag = NAME if gm == 0 else f"{NAME} G={gm}"
            print(f"{tag} | {name} | {us:.1f} | {w:.0f} | {w * us * 1e-6:.4f} | {(w - p_idle) * us * 1e-6:.4f} | "
                  f"{flops / us / 1e6:.0f} TFLOP/s, {(w - p_idle) * us * 1e-6 / flops * 1e12:.4f} pJ/FLOP dynamic", flush=True)
    lib.fino_tune_set(0, 0)


if __name__ == "__main__":
    main()
