for v in ${@:-hip xnoread xnoexp xnodma xall hip}; do
  echo "== $v"
  FINO_LIB_PATH=$PWD/frameino_amd/lib/libframeino_$v.so timeout 300 python tools/attn_kernel_ab.py 1 4 2>&1 | grep -v amdgpu.ids | head -1
done
