#!/bin/bash
# GPU box: the block GEMMs' joules per launch for the product build, its raster group heights and its drop-one builds.
# Build the variants first (here or on the build box: the .so files travel with gpurun):
#   for v in "gnodma -DGP_X_NODMA" "gnoread -DGP_X_NOREAD=1" "gnostore -DGP_X_NOSTORE" "gnomfma -DGP_X_NOMFMA" "gnoepi -DGP_X_NOEPI" \
#            "gnobar -DGP_X_NOBAR" "gnogelu -DGP_X_NOGELU"; do set -- $v; tools/debug/mkvar.sh --experiments $1 fino_gemm.hip "-DFINO_EXPERIMENT $2"; done
# then: tools/debug/gemm_energy_dropone.sh > profiles/rNN_gemm_energy_dropone_raw.txt ; python tools/gemm_energy_table.py < that file
export FINO_ALLOW_EXPERIMENT=1
python tools/gemm_energy_dropone.py 2>&1 | grep -v amdgpu.ids
python tools/gemm_energy_dropone.py --raster 2>&1 | grep -v amdgpu.ids
for v in gnomfma gnodma gnoread gnobar gnoepi gnostore gnogelu; do
  [ -f frameino_amd/lib/libframeino_$v.so ] || continue
  FINO_LIB_PATH=$PWD/frameino_amd/lib/libframeino_$v.so timeout 300 python tools/gemm_energy_dropone.py 2>&1 | grep -v amdgpu.ids
done
python tools/gemm_energy_dropone.py 2>&1 | grep -v amdgpu.ids | sed "s/^hip /hip(again) /"
