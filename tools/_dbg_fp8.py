import torch, sys
sys.path.insert(0, '.')
from frameino_amd import ops
torch.manual_seed(0)
import os
from frameino_amd import _lib
_lib.lib().fino_tune_set(5, int(os.environ.get('FINO_FP8_KERNEL','0')))
for (b,h,lq,lk) in [(1,2,256,256),(1,2,256,64),(1,2,256,128),(1,1,64,512),(2,4,1000,1000)]:
    q=torch.randn(b,lq,h*64,device='cuda',dtype=torch.bfloat16); k=torch.randn(b,lk,h*64,device='cuda',dtype=torch.bfloat16); v=torch.randn(b,lk,h*64,device='cuda',dtype=torch.bfloat16)
    o=ops.attention_fp8(q,k,v,h); r=ops.attention(q,k,v,h)
    torch.cuda.synchronize()
    bad=~torch.isfinite(o.float())
    e=((o.float()-r.float()).pow(2).mean().sqrt()/r.float().pow(2).mean().sqrt()).item()
    print((b,h,lq,lk),"nan",bad.sum().item(),"relrms",e)
    if bad.any():
        idx=bad.nonzero(); print(idx[:5].tolist(), idx[-5:].tolist(), "rows with nan", bad.any(-1).sum().item(), "cols", bad.any(1).sum().item())
