"""Diagnostic (not a benchmark): per-segment s_memtime stamps of the GEMM main loop from a FINO_GEMM_STAMP build."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["FINO_LIB_PATH"] = os.path.join(ROOT, "frameino_amd/lib/libframeino_stamp.so")
import torch
from frameino_amd import ops
L = int(os.environ.get("FINO_STAMP_ROWS", 12320))
N, K, EPI = (int(x) for x in (sys.argv[1:4] + ["9216", "3072", "0"][len(sys.argv) - 1:]))
a = torch.randn(L, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.02).bfloat16(); b = torch.randn(N, device="cuda").bfloat16()
res = torch.randn(L, N, device="cuda").bfloat16() if EPI >= 2 else None
gate = torch.randn(2, N, device="cuda") if EPI >= 3 else None
sel = (torch.arange(L, device="cuda") >= 880).to(torch.int32) if EPI >= 3 else None
print(f"L={L} N={N} K={K} epilogue={EPI}")
for _ in range(3): ops.gemm(a, w, b, EPI, res, gate, sel)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
lib = ctypes.CDLL(os.environ["FINO_LIB_PATH"]); lib.fino_gemm_debug_read(buf)
for wv in range(8):
    v = [buf[wv * 8 + i] for i in range(5)]
    nk = max(v[4], 1); tot = sum(v[:4])
    if os.environ.get("FINO_GEMM_PP", "1") != "0":
        e = [buf[wv * 8 + i] for i in (5, 6, 7)]
        print(f"wave {wv}: per K-tile cycles: LOAD+dma {v[0]/nk:7.0f}  barrier {v[1]/nk:6.0f}  COMPUTE+wait {v[2]/nk:7.0f}  barrier {v[3]/nk:6.0f}  total {tot/nk:7.0f}"
              f" | tile: prologue + re-sync {e[0] - tot:7.0f}  loop {tot:8.0f}  epilogue to staged {e[1]:7.0f}  epilogue store loop {e[2]:7.0f} cycles")
    else:
        print(f"wave {wv}: per K-tile cycles: block0 {v[0]/nk:7.0f}  vmcnt-wait {v[1]/nk:6.0f}  barrier {v[2]/nk:6.0f}  block1+dma {v[3]/nk:7.0f}  total {tot/nk:7.0f}")
