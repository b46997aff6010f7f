"""Diagnostic (not a benchmark): per-segment s_memtime stamps of the attention tile loop (FINO_ATTN_STAMP build)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FINO_LIB_PATH", os.path.join(ROOT, "frameino_amd/lib/libframeino_stamp.so"))
import torch
from frameino_amd import ops
L, D, H = (int(x) for x in (sys.argv[1:4] + ["12320", "3072", "24"][len(sys.argv) - 1:]))
print(f"L={L} D={D} heads={H} head_dim={D // H}")
qkv = torch.randn(1, L, 3 * D, device="cuda").bfloat16()
KERNEL = int(os.environ.get("FINO_STAMP_KERNEL", "0"))      # 4: attn_ppd_kernel (its stamps: softmax | matrix = S part + P.V part)
if KERNEL:
    from frameino_amd import _lib
    _lib.lib().fino_tune_set(4, KERNEL)
for _ in range(3): ops.attention(qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:], H)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
lib = ctypes.CDLL(os.environ["FINO_LIB_PATH"]); lib.fino_attn_debug_read(buf)
for wv in range(8):
    v = [buf[wv * 8 + i] for i in range(8)]
    nt = max(v[4], 1)
    if KERNEL == 4:
        print(f"wave {wv}: per tile cycles: softmax {v[0]/nt:6.0f} (to its last arithmetic {v[6]/nt:5.0f})  barrier {v[1]/nt:6.0f}  matrix {v[2]/nt:6.0f} = S {v[5]/nt:5.0f} + P.V {(v[2]-v[5])/nt:5.0f}  barrier {v[3]/nt:6.0f}  total {sum(v[:4])/nt:6.0f}")
        continue
    print(f"wave {wv}: per tile cycles: softmax {v[0]/nt:6.0f}  barrier {v[1]/nt:6.0f}  matrix {v[2]/nt:6.0f}  barrier {v[3]/nt:6.0f}  total {sum(v[:4])/nt:6.0f} | softmax = staging {v[5]/nt:5.0f} + max/rescale {v[6]/nt:5.0f} + exp {v[7]/nt:5.0f} + sum/pack {(v[0]-v[5]-v[6]-v[7])/nt:5.0f}")
