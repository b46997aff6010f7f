#!/usr/bin/env python3
"""Energy per launch (rocm-smi socket power x time) of the FFN-up GEMM (M = 24640, N = 14336, K = 3072, GELU epilogue) and the
gated-residual out-projection (N = K = 3072) under raster group heights 2 .. 16 (FINO_TUNE_GEMM_GROUP_M): VERDICT r2 item 7
asked whether the group height that wins on time also wins on joules under the power cap.  One setting loops ~3 s."""
import json, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frameino_amd import _lib, ops
lib = _lib.lib()
g = torch.Generator(device="cuda").manual_seed(0)
M = 24640
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0


def smi_loop(samples, stop):
    while not stop[0]:
        try:
            dd = json.loads(subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10).stdout)
            c = dd[sorted(dd)[0]]
            samples.append((time.time(), float(c.get("Current Socket Graphics Package Power (W)", 0)), c.get("sclk clock speed:", "")))
        except Exception:      # noqa: BLE001
            pass
        time.sleep(0.1)


for name, n, k, epi in (("ffn_up + GELU", 14336, 3072, 1), ("out-proj gated residual", 3072, 3072, 3)):
    A = torch.randn(M, k, device="cuda", generator=g).bfloat16()
    W = (torch.randn(n, k, device="cuda", generator=g) * 0.02).bfloat16()
    b = torch.randn(n, device="cuda", generator=g).bfloat16()
    res = torch.randn(M, n, device="cuda", generator=g).bfloat16() if epi == 3 else None
    gate = torch.randn(2, n, device="cuda", generator=g) if epi == 3 else None
    sel = (torch.arange(M, device="cuda") % 2).to(torch.int32) if epi == 3 else None
    out = torch.empty(M, n, device="cuda", dtype=torch.bfloat16)
    f = lambda: ops.gemm(A, W, b, epi, res, gate, sel, out=out)
    for gm in (0, 2, 4, 8, 16):
        lib.fino_tune_set(0, gm)
        f(); f(); torch.cuda.synchronize()
        samples, stop = [], [False]
        th = threading.Thread(target=smi_loop, args=(samples, stop)); th.start()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.time(); cnt = 0
        s.record()
        while time.time() - t0 < secs:
            for _ in range(20): f()
            cnt += 20
            torch.cuda.synchronize()
        e.record(); torch.cuda.synchronize()
        t1 = time.time()
        stop[0] = True; th.join()
        us = s.elapsed_time(e) / cnt * 1e3
        busy = [p for (ts, p, c) in samples if t0 + 0.8 <= ts <= t1]
        w = sum(busy) / max(len(busy), 1)
        print(f"{name:26s} group_m {gm if gm else 'default':>7}: {us:8.1f} us  {2.0 * M * n * k / us / 1e6:5.0f} TFLOP/s  {w:5.0f} W  "
              f"{w * us * 1e-6:6.3f} J per launch  ({len(busy)} power samples)", flush=True)
    lib.fino_tune_set(0, 0)
