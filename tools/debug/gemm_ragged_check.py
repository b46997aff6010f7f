import sys, torch
sys.path.insert(0, ".")
from frameino_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
for M in (24640, 3080, 1540, 10560, 300, 8, 65, 129, 257):
    for (n, k, epi) in ((3072, 3072, 3), (9216, 3072, 0), (14336, 3072, 1), (3072, 14336, 3), (1728, 3072, 2)):
        if M < 1000 and n > 3072: continue
        A = torch.randn(M, k, device="cuda", generator=g).bfloat16()
        W = (torch.randn(n, k, device="cuda", generator=g) * 0.02).bfloat16()
        b = torch.randn(n, device="cuda", generator=g).bfloat16()
        res = torch.randn(M, n, device="cuda", generator=g).bfloat16() if epi >= 2 else None
        gate = torch.randn(2, n, device="cuda", generator=g) if epi == 3 else None
        sel = (torch.arange(M, device="cuda") % 2).to(torch.int32) if epi == 3 else None
        outs = []
        for tm in (8, 4, 0):
            outs.append(ops.gemm(A, W, b, epi, res, gate, sel, tile_m=tm))
        ref = torch.nn.functional.linear(A.float(), W.float(), b.float())
        ok = all(torch.equal(outs[0], o) for o in outs[1:])
        print(M, n, k, epi, "tile heights bit-equal:", ok)
        assert ok
print("OK")
