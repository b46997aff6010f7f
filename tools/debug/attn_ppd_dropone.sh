#!/bin/bash
# GPU box: drop-one builds of attn_ppd_kernel against the product (wrong results by design).  Build with
#   tools/debug/mkvar.sh --experiments xnoread fino_attention.hip "-DFINO_EXPERIMENT -DPD_X_NOREAD"   (PD_X_NOEXP, PD_X_NODMA, PD_X_AOP=1|2, PW_X_NOSTORE)
# (the switches live in tools/debug/experiments.patch, not in the product sources).
export FINO_ALLOW_EXPERIMENT=1
for v in ${@:-hip xnoread xnoexp xnodma xall hip}; do
  echo "== $v"
  FINO_LIB_PATH=$PWD/frameino_amd/lib/libframeino_$v.so timeout 300 python tools/attn_kernel_ab.py 1 4 2>&1 | grep -v amdgpu.ids | head -1
done
