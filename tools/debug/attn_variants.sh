for v in hip noprek prio1 prio2 maxs maxsp1; do
  echo "== $v"
  FINO_LIB_PATH=$PWD/frameino_amd/lib/libframeino_$v.so timeout 300 python tools/attn_kernel_ab.py 1 4 2>&1 | grep -v amdgpu.ids | head -2
done
echo "== correctness of default build"
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "dma_staged" 2>&1 | tail -2
for v in prio1 maxs maxsp1; do FINO_LIB_PATH=$PWD/frameino_amd/lib/libframeino_$v.so timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "dma_staged" 2>&1 | tail -1; done
