#!/bin/bash
# GPU box: A/B of attn_ppd_kernel builds (libframeino_<name>.so made with -DPD_...=x) against attn_pp_kernel in the same process
for v in "$@"; do
  echo "== $v"
  FINO_LIB_PATH=$PWD/frameino_amd/lib/libframeino_$v.so timeout 300 python tools/attn_kernel_ab.py 1 4 2>&1 | grep -v amdgpu.ids | head -${ROWS:-2}
  FINO_LIB_PATH=$PWD/frameino_amd/lib/libframeino_$v.so timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "dma_staged" 2>&1 | tail -1
done
