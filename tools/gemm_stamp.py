"""Diagnostic (not a benchmark): per-segment s_memtime stamps of the GEMM main loop from a FINO_GEMM_STAMP build."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["FINO_LIB_PATH"] = os.path.join(ROOT, "frameino_amd/lib/libframeino_stamp.so")
import torch
from frameino_amd import ops
L, D = 12320, 3072
a = torch.randn(L, D, device="cuda").bfloat16(); w = (torch.randn(3 * D, D, device="cuda") * 0.02).bfloat16(); b = torch.randn(3 * D, device="cuda").bfloat16()
for _ in range(3): ops.gemm(a, w, b)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
lib = ctypes.CDLL(os.environ["FINO_LIB_PATH"]); lib.fino_gemm_debug_read(buf)
for wv in range(8):
    v = [buf[wv * 8 + i] for i in range(5)]
    nk = max(v[4], 1); tot = sum(v[:4])
    print(f"wave {wv}: per K-tile cycles: block0 {v[0]/nk:7.0f}  vmcnt-wait {v[1]/nk:6.0f}  barrier {v[2]/nk:6.0f}  block1+dma {v[3]/nk:7.0f}  total {tot/nk:7.0f}")
