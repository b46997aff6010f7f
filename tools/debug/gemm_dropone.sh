#!/bin/bash
# GPU box: drop-one builds of gemm_pp_kernel (wrong results by design): build them first, here or there, with
#   tools/debug/mkvar.sh --experiments <name> fino_gemm.hip "-DFINO_EXPERIMENT -DGP_X_NODMA | -DGP_X_NOREAD=1|2 | -DGP_X_NOSTORE"
# (the switches live in tools/debug/experiments.patch, not in the product sources), then: tools/debug/gemm_dropone.sh <name>...
export FINO_ALLOW_EXPERIMENT=1
for v in "$@"; do
  FINO_LIB_PATH=$PWD/frameino_amd/lib/libframeino_$v.so timeout 300 python tools/gemm_shapes.py 2>&1 | grep -v amdgpu.ids
done
