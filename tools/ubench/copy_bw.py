"""Diagnostic (not a benchmark): what a plain device copy of the adaLN kernel's traffic (151 MB in, 151 MB out) reaches
on this box, next to the library's LayerNorm + modulate kernel on the same rows -- the ceiling for equal read and write
streams that section 4 of DESIGN.md prices the elementwise kernels against."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from frameino_amd import ops

def timed(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

L, D = 24640, 3072
x = torch.randn(L, D, device="cuda").bfloat16()
y = torch.empty_like(x)
nbytes = x.numel() * 2
us = timed(lambda: y.copy_(x));            print(f"torch copy        {us:7.1f} us  {2 * nbytes / us / 1e6:6.2f} TB/s (read + write)")
us = timed(lambda: y.fill_(1.0));          print(f"torch fill        {us:7.1f} us  {nbytes / us / 1e6:6.2f} TB/s (write only)")
us = timed(lambda: x.sum(dtype=torch.float32)); print(f"torch sum         {us:7.1f} us  {nbytes / us / 1e6:6.2f} TB/s (read only)")
xi = x.view(torch.int32); yi = y.view(torch.int32)
us = timed(lambda: torch.add(xi, 1, out=yi)); print(f"torch int32 add   {us:7.1f} us  {2 * nbytes / us / 1e6:6.2f} TB/s (read + write)")
tab = torch.randn(2, 2 * D, device="cuda")
shift, scale = tab[:, :D], tab[:, D:]
sel = (torch.arange(L, device="cuda") >= L // 2).to(torch.int32)
try:
    us = timed(lambda: ops.adaln_modulate(x, shift, scale, sel, 1e-6, y))
    print(f"adaln_modulate    {us:7.1f} us  {2 * nbytes / us / 1e6:6.2f} TB/s (read + write)")
except Exception as e:  # the diagnostic keeps going if the wrapper's signature moved
    print("adaln_modulate: ", repr(e))

# where the adaLN kernel's time goes: the same rows without tables, with one table row for every token, in place
def rate(us): return f"{us:7.1f} us  {2 * nbytes / us / 1e6:6.2f} TB/s"
print("layernorm, no affine (no tables)      ", rate(timed(lambda: ops.layernorm(x, None, None, 1e-6, y))))
print("adaln, sel=None (one table row)       ", rate(timed(lambda: ops.adaln_modulate(x, shift[:1], scale[:1], None, 1e-6, y))))
print("adaln, two table rows by sel          ", rate(timed(lambda: ops.adaln_modulate(x, shift, scale, sel, 1e-6, y))))
x2 = x.clone()
print("adaln, in place                       ", rate(timed(lambda: ops.adaln_modulate(x2, shift, scale, sel, 1e-6, x2))))
print("torch int32 add in place              ", rate(timed(lambda: xi.add_(1))))
x = torch.randn(L, D, device="cuda").bfloat16()
cos = torch.randn(L, 64, device="cuda"); sin = torch.randn(L, 64, device="cuda"); wq = torch.randn(D, device="cuda").bfloat16()
try:
    print("rmsnorm_rope in place (one segment)   ", rate(timed(lambda: ops.rmsnorm_rope_(x, wq, 1e-6, cos, sin, 128))))
except Exception as e:
    print("rmsnorm_rope:", repr(e))
