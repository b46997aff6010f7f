#!/bin/bash
# GPU box: drop-one builds of gemm_pp_kernel (libframeino_<name>.so made with -DFINO_EXPERIMENT -DGP_X_...; wrong results)
for v in "$@"; do
  FINO_LIB_PATH=$PWD/frameino_amd/lib/libframeino_$v.so timeout 300 python tools/gemm_shapes.py 2>&1 | grep -v amdgpu.ids
done
