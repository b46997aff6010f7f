#!/usr/bin/env python3
"""Power and shader clock (rocm-smi) while one attention shape loops for a few seconds: tells a power-capped kernel
(sclk well under 2.4 GHz at ~1.3 kW) from an issue-bound one.  usage: attn_power.py head_dim L heads tune(0|1|2) [seconds]"""
import json, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frameino_amd import _lib, ops
hd, L, heads, tune = (int(x) for x in sys.argv[1:5])
secs = float(sys.argv[5]) if len(sys.argv) > 5 else 4.0
lib = _lib.lib()
g = torch.Generator(device="cuda").manual_seed(0)
d = heads * hd
qkv = torch.randn(2, L, 3 * d, device="cuda", generator=g).bfloat16()
fold = tune == 2
q = (qkv[:, :, :d].float() * (hd ** -0.5 * ops.LOG2E)).bfloat16() if fold else qkv[:, :, :d]
k, v = qkv[:, :, d:2 * d], qkv[:, :, 2 * d:]
out = torch.empty(2, L, d, device="cuda", dtype=torch.bfloat16)
lib.fino_tune_set(4, tune)
sc = ops.SCALE_FOLDED if fold else None
for _ in range(3): ops.attention(q, k, v, heads, out=out, scale=sc)
torch.cuda.synchronize()
samples, stop = [], False


def smi():
    while not stop:
        try:
            dd = json.loads(subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10).stdout)
            c = dd[sorted(dd)[0]]
            samples.append((time.time(), float(c.get("Current Socket Graphics Package Power (W)", 0)), c.get("sclk clock speed:", "")))
        except Exception:      # noqa: BLE001
            pass
        time.sleep(0.1)


th = threading.Thread(target=smi); th.start()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.time(); n = 0
s.record()
while time.time() - t0 < secs:
    for _ in range(10): ops.attention(q, k, v, heads, out=out, scale=sc)
    n += 10
    torch.cuda.synchronize()
e.record(); torch.cuda.synchronize()
t1 = time.time()
stop = True; th.join()
us = s.elapsed_time(e) / n * 1e3
busy = [(p, c) for (ts, p, c) in samples if t0 + 1.0 <= ts <= t1]
print(f"head_dim {hd} L {L} heads {heads} tune {tune}: {us:8.1f} us  {4.0 * 2 * heads * L * L * hd / us / 1e6:5.0f} TFLOP/s   "
      f"power {sum(p for p, _ in busy) / max(len(busy), 1):5.0f} W avg  sclk samples {[c for _, c in busy][::4]}")
