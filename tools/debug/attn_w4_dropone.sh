#!/bin/bash
# Drop-one builds of the 4-wave folded-scale attention kernel (attn_w4_kernel: the head_dim-64 default, CogVideoX) against the
# product: WRONG results by design, only the time counts (which phase bounds the bf16 head_dim-64 attention).
#   tools/debug/attn_w4_dropone.sh build      # here or on the GPU box (hipcc cross-compiles): ~15 s per variant
#   tools/debug/attn_w4_dropone.sh run        # GPU box: B = 2, 48 heads x 64, L = 19126 (tools/attn_w4_variant_time.py)
# The switches live in tools/debug/experiments.patch (mkvar.sh --experiments).
set -e
cd "$(dirname "$0")/../.."
VARS="NOEXP NOMAX NOPACK PACKCONST PACKPERM NOBAR NODMA NOLGKM"
if [ "$1" = build ]; then
  make -s -C frameino_amd/csrc
  for v in $VARS; do tools/debug/mkvar.sh --experiments w4x_$v fino_attention_w4.hip "-DFINO_EXPERIMENT -DW4_X_$v"; done
  tools/debug/mkvar.sh --experiments w4x_NOEXP_NOPACK fino_attention_w4.hip "-DFINO_EXPERIMENT -DW4_X_NOEXP -DW4_X_NOPACK"
  tools/debug/mkvar.sh --experiments w4x_NOEXP_NOPACK_NOMAX fino_attention_w4.hip "-DFINO_EXPERIMENT -DW4_X_NOEXP -DW4_X_NOPACK -DW4_X_NOMAX"
  tools/debug/mkvar.sh --experiments w4x_ALL fino_attention_w4.hip "-DFINO_EXPERIMENT -DW4_X_NOEXP -DW4_X_NOPACK -DW4_X_NOMAX -DW4_X_NOBAR -DW4_X_NODMA"
else
  export FINO_ALLOW_EXPERIMENT=1
  echo "# 4-wave attention kernel, head_dim 64, B = 2, 48 heads, L = 19126: product vs drop-one builds (wrong results by design), same box"
  python3 tools/attn_w4_variant_time.py 2>&1 | grep TFLOP
  for v in $VARS NOEXP_NOPACK NOEXP_NOPACK_NOMAX ALL; do
    FINO_LIB_PATH=$PWD/frameino_amd/lib/libframeino_w4x_$v.so python3 tools/attn_w4_variant_time.py 2>&1 | grep TFLOP
  done
  python3 tools/attn_w4_variant_time.py 2>&1 | grep TFLOP
fi
