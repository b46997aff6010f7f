#!/usr/bin/env python3
"""Where a wave of the fp8-operand attention kernel spends a key tile (experiment build -DF8_X_STAMP, loaded through
FINO_LIB_PATH): s_memtime stamps (100 MHz constant clock on gfx950 -> printed in ns) around the segments of both phases,
summed over the tiles of workgroup 40, per wave."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from frameino_amd import ops, _lib
b, heads, L = 2, 48, 19126
d = heads * 64
g = torch.Generator(device="cuda").manual_seed(0)
q = torch.randn(b, L, d, device="cuda", generator=g).bfloat16()
kv = torch.randn(b, L, 2 * d, device="cuda", generator=g).bfloat16()
o = torch.empty_like(q)
for _ in range(3):
    ops.attention_fp8(q, kv[:, :, :d], kv[:, :, d:], heads, out=o)
torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 64)()
assert raw.fino_attn_f8_debug_read(buf) == 0
names = ["prefetch reads + staging write (vmcnt wait)", "rescale vote + exp2 + pack + global loads", "lgkm wait + barrier",
         "matrix phase work", "lgkm wait + barrier"]
for w in range(8):
    nt = buf[w * 8 + 5]
    per = [buf[w * 8 + i] / max(nt, 1) for i in range(5)]
    print(f"wave {w}: tiles {nt}; per tile, s_memtime ticks: " + " | ".join(f"{x:6.1f}" for x in per) + f" | sum {sum(per):6.1f}")
print("segments: " + " | ".join(names))
