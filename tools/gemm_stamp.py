"""Diagnostic (not a benchmark): per-segment s_memtime stamps of the GEMM main loop from a FINO_GEMM_STAMP build."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["FINO_LIB_PATH"] = os.path.join(ROOT, "frameino_amd/lib/libframeino_stamp.so")
import torch
from frameino_amd import ops
L = 12320
N, K, EPI = (int(x) for x in (sys.argv[1:4] + ["9216", "3072", "0"][len(sys.argv) - 1:]))
a = torch.randn(L, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.02).bfloat16(); b = torch.randn(N, device="cuda").bfloat16()
print(f"L={L} N={N} K={K} epilogue={EPI}")
for _ in range(3): ops.gemm(a, w, b, EPI)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
lib = ctypes.CDLL(os.environ["FINO_LIB_PATH"]); lib.fino_gemm_debug_read(buf)
for wv in range(8):
    v = [buf[wv * 8 + i] for i in range(5)]
    nk = max(v[4], 1); tot = sum(v[:4])
    if os.environ.get("FINO_GEMM_PP", "1") != "0":
        print(f"wave {wv}: per K-tile cycles: LOAD+dma {v[0]/nk:7.0f}  barrier {v[1]/nk:6.0f}  COMPUTE+wait {v[2]/nk:7.0f}  barrier {v[3]/nk:6.0f}  total {tot/nk:7.0f}")
    else:
        print(f"wave {wv}: per K-tile cycles: block0 {v[0]/nk:7.0f}  vmcnt-wait {v[1]/nk:6.0f}  barrier {v[2]/nk:6.0f}  block1+dma {v[3]/nk:7.0f}  total {tot/nk:7.0f}")
