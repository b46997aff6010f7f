#!/bin/bash
# Timing experiments on the fp8-operand attention kernel (WRONG results by construction: each build drops one ingredient of
# the loop; -DFINO_EXPERIMENT, loaded through FINO_LIB_PATH only).  `build` here (no GPU needed), `run` on the GPU box.
set -e
cd "$(dirname "$0")/../frameino_amd/csrc"
VARS="STAMP NOEXP NOPACK NOMAX NOSTAGE NOPV NOBAR NOOPENER"
RIM=${RIM:-0}   # F8_READS_IN_MATRIX of the variant builds
if [ "$1" = build ]; then
  make -s
  cd ../..
  # the F8_X_* switches live in tools/debug/experiments.patch: mkvar.sh --experiments builds from a patched scratch copy of csrc/
  for v in $VARS; do
    tools/debug/mkvar.sh --experiments f8x_$v fino_attention_fp8.hip "-DFINO_EXPERIMENT -DF8_X_$v -DF8_READS_IN_MATRIX=$RIM"
  done
else
  cd ../..
  echo "product:"; FINO_FP8_ONLY=1 python3 tools/attn_fp8_bench.py | grep TFLOP
  FINO_ALLOW_EXPERIMENT=1 FINO_LIB_PATH=frameino_amd/lib/libframeino_f8x_STAMP.so python3 tools/attn_fp8_stamp.py
  for v in ${VARS#STAMP }; do
    echo "$v:"; FINO_ALLOW_EXPERIMENT=1 FINO_FP8_ONLY=1 FINO_LIB_PATH=frameino_amd/lib/libframeino_f8x_$v.so python3 tools/attn_fp8_bench.py | grep TFLOP
  done
fi
