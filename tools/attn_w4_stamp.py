"""Diagnostic (not a benchmark): per-segment s_memtime stamps of the 4-wave attention tile loop
(`make -C frameino_amd/csrc variant NAME=stamp VFLAGS=-DFINO_ATTN_STAMP`)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FINO_LIB_PATH", os.path.join(ROOT, "frameino_amd/lib/libframeino_stamp.so"))
import torch
from frameino_amd import _lib, ops
L, D, H = (int(x) for x in (sys.argv[1:4] + ["12288", "3072", "24"][len(sys.argv) - 1:]))
qkv = torch.randn(2, L, 3 * D, device="cuda").bfloat16()
_lib.lib().fino_tune_set(4, 2)
fold = (D // H == 64) or os.environ.get("FINO_STAMP_FOLD") == "1"          # head_dim 64 exists with the folded scale only
q = (qkv[:, :, :D].float() * ((D // H) ** -0.5 * ops.LOG2E)).bfloat16() if fold else qkv[:, :, :D]
for _ in range(3): ops.attention(q, qkv[:, :, D:2 * D], qkv[:, :, 2 * D:], H, scale=ops.SCALE_FOLDED if fold else None)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
lib = ctypes.CDLL(os.environ["FINO_LIB_PATH"]); lib.fino_attn_w4_debug_read(buf)
for wv in range(4):
    v = [buf[wv * 8 + i] for i in range(8)]
    nt = max(v[5], 1)
    print(f"wave {wv}: per tile: vmcnt wait {v[0]/nt:6.0f}  barrier {v[1]/nt:6.0f}  dma issue {v[2]/nt:6.0f}  phase1 (QK || exp) {v[3]/nt:6.0f}  "
          f"phase2 (PV || max, exp) {v[4]/nt:6.0f}  total {sum(v[:5])/nt:6.0f}  (s_memtime ticks; nt {nt})")
