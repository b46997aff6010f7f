#!/usr/bin/env python3
"""Where a workgroup of the text cross-attention spends its time (stamp build: make -C frameino_amd/csrc stamp): s_memtime at
kernel entry, Q loaded, first tiles staged, loop start, loop end, output stored -- workgroup 40, per wave, in shader cycles."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["FINO_LIB_PATH"] = os.path.join(ROOT, "frameino_amd/lib/libframeino_stamp.so")
import torch
from frameino_amd import ops
b, lq, lk, heads, dh = 2, 12320, int(sys.argv[1]) if len(sys.argv) > 1 else 512, 24, 128
d = heads * dh
q = torch.randn(b, lq, d, device="cuda").bfloat16()
kv = torch.randn(b, lk, 2 * d, device="cuda").bfloat16()
for _ in range(3): ops.attention(q, kv[:, :, :d], kv[:, :, d:], heads)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 128)()
lib = ctypes.CDLL(os.environ["FINO_LIB_PATH"]); lib.fino_attn_debug_read(buf)
names = ["Q loaded", "K/V tiles 0, 1 staged", "S(0) + max", "tile loop", "normalise + store"]
for wv in range(8):
    t = [buf[64 + wv * 8 + i] for i in range(6)]
    print(f"wave {wv}: " + "  ".join(f"{n} {t[i + 1] - t[i]:6d}" for i, n in enumerate(names)) + f"  | total {t[5] - t[0]:6d} cycles")
