#!/usr/bin/env python3
"""What the board sustains with only the matrix pipe working (fino_diag_mfma_peak): dense bf16 MFMA TFLOP/s for short and
long launches, 1 and 2 waves per SIMD, operands all zero or gaussian noise, with rocm-smi power / clock samples.
The power drawn depends on how many operand bits toggle; under the board's cap so does the clock: the long-launch
gaussian figure is the ceiling the MFMA-bound kernels are compared with in DESIGN.md section 4.1."""
import ctypes
import json
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from frameino_amd import _lib  # noqa: E402

lib = _lib.lib()
scratch = torch.zeros(64 + 2 * 256 * 4 + 2 * 256 * 8, device="cuda")   # 256 B + A and B operands of 256 lanes (8 bf16 each) + fp8 ones (32 B each)
fl = ctypes.c_double()
samples, stop = [], False


def smi():
    while not stop:
        try:
            d = json.loads(subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True,
                                          text=True, timeout=10).stdout)
            c = d[sorted(d)[0]]
            samples.append((time.time(), float(c.get("Current Socket Graphics Package Power (W)", 0)),
                            c.get("sclk clock speed:", "")))
        except Exception:      # noqa: BLE001
            pass
        time.sleep(0.1)


def run(kind, wps, iters):
    _lib.check(lib.fino_diag_mfma_peak(kind, wps, iters, scratch.data_ptr(), ctypes.byref(fl), None), "fino_diag_mfma_peak")


th = threading.Thread(target=smi)
th.start()
g = torch.Generator(device="cuda").manual_seed(0)
operands = scratch[64:64 + 2 * 256 * 4].view(torch.bfloat16)
operands8 = scratch[64 + 2 * 256 * 4:].view(torch.uint8)
for data in ("zeros", "gaussian"):
    operands.copy_(torch.zeros_like(operands) if data == "zeros"
                   else torch.randn(operands.shape, device="cuda", generator=g).bfloat16())
    # e4m3 bytes: random sign and mantissa, exponent field 5 .. 8 of 15 (|x| in [0.25, 4)): what a block-scaled activation looks like
    r = torch.randint(0, 256, operands8.shape, device="cuda", generator=g)
    e = torch.randint(5, 9, operands8.shape, device="cuda", generator=g)
    operands8.copy_(torch.zeros_like(operands8) if data == "zeros" else ((r & 0x87) | (e << 3)).to(torch.uint8))
    print(f"--- operands: {data}")
    for kind, nm in ((0, "32x32x16"), (1, "16x16x32"), (2, "fp8 32x32x64 (block-scaled, scales 1.0)")):
        for wps in (1, 2):
            for iters, label in ((2000, "short (~1 ms)"), (400000, "long (~0.3 s)")):
                run(kind, wps, 100)
                torch.cuda.synchronize()
                reps = 20 if iters < 10000 else 3
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0 = time.time()
                s.record()
                for _ in range(reps):
                    run(kind, wps, iters)
                e.record()
                torch.cuda.synchronize()
                t1 = time.time()
                ms = s.elapsed_time(e) / reps
                pw = [p for (ts, p, c) in samples if t0 + 0.15 <= ts <= t1]
                ck = [c for (ts, p, c) in samples if t0 + 0.15 <= ts <= t1]
                print(f"{nm} {wps} wave/SIMD {label:15s}: {fl.value / ms / 1e9:7.0f} TFLOP/s  ({ms:8.2f} ms)  "
                      f"power {max(pw) if pw else float('nan'):6.0f} W max  sclk {ck[-1] if ck else '?'}", flush=True)
                time.sleep(0.5)
stop = True
th.join()
