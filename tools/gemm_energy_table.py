#!/usr/bin/env python3
"""tools/gemm_energy_dropone.py's raw lines (stdin or files) -> the drop-one table in joules: per block GEMM the product build's
dynamic joules per launch, the bare-MFMA share (algorithmic FLOPs x the bare loop's pJ/FLOP of the same run), and for every wrong-result
build what its removal saves: J(product) - J(variant).  The shares are of the product's dynamic joules."""
import re
import sys

rows, bare = {}, []
for ln in (l for f in (sys.argv[1:] or ["-"]) for l in (sys.stdin if f == "-" else open(f))):
    if ln.startswith("#") or "|" not in ln:
        continue
    p = [x.strip() for x in ln.split("|")]
    lib, name, us, w, j, jd = p[0], p[1], float(p[2]), float(p[3]), float(p[4]), float(p[5])
    if name.startswith("bare MFMA"):
        bare.append(float(re.search(r"dynamic ([0-9.]+)", p[6]).group(1)))
        continue
    rows.setdefault(name, {}).setdefault(lib, []).append((us, w, jd))
pj = sum(bare) / len(bare)
FLOPS = {"qkv": 2.0 * 24640 * 9216 * 3072, "out-proj": 2.0 * 24640 * 3072 * 3072, "ffn-up": 2.0 * 24640 * 14336 * 3072,
         "ffn-down": 2.0 * 24640 * 3072 * 14336}
print(f"bare MFMA loop (32x32x16 bf16, gaussian operands, same runs): {pj:.4f} pJ/FLOP dynamic")
LABEL = {"gnomfma": "the kernel WITHOUT its matrix instructions (what is left costs)", "gnodma": "no operand staging (LDS-DMA + all L2 / fabric traffic)",
         "gnoread": "no LDS fragment reads", "gnobar": "no barriers (2 per K-tile)", "gnoepi": "no epilogue at all",
         "gnostore": "  of which: residual loads + gate + global stores", "gnogelu": "  of which: GELU arithmetic"}
for name, libs in rows.items():
    prod = [v for k, vs in libs.items() if k.startswith("hip") and "G=" not in k for v in vs]
    us0, w0, j0 = (sum(x[i] for x in prod) / len(prod) for i in range(3))
    fl = FLOPS[name.split()[0]]
    print(f"\n{name}: product {us0:7.1f} us at {w0:4.0f} W = {j0:.3f} J dynamic per launch ({j0 / fl * 1e12:.3f} pJ/FLOP); "
          f"matrix instructions alone {fl * pj * 1e-12:.3f} J = {fl * pj * 1e-12 / j0 * 100:4.1f} %")
    for k in ("gnomfma", "gnodma", "gnoread", "gnobar", "gnoepi", "gnostore", "gnogelu"):
        if k in libs:
            us, w, j = libs[k][0]
            what = "is left" if k == "gnomfma" else "saved"
            val = j if k == "gnomfma" else j0 - j
            print(f"    {LABEL[k]:66s} {us:7.1f} us {w:4.0f} W  {j:.3f} J -> {val:+.3f} J {what} ({val / j0 * 100:+5.1f} %)")
    gs = sorted((int(k.split("G=")[1]), v[0]) for k, v in libs.items() if "G=" in k)
    if gs:
        print("    raster group height G (same kernel, same bits; L2-fill traffic falls 2.2x from G = 1 to G = 4, profiles/r02_gemm_raster.md): "
              + "  ".join(f"G={g_}: {v[0]:.0f} us {v[2]:.3f} J" for g_, v in gs))
