#!/bin/bash
# Drop-one builds of the free-running fp8 attention kernel (attn_fp8_fr_kernel; WRONG results by construction, -DFINO_EXPERIMENT,
# loaded through FINO_LIB_PATH only): `build` here (no GPU needed), `run` on the GPU box.  Each build drops one ingredient of the
# key-tile loop; tools/attn_fp8_bench.py times both P modes of each.
set -e
VARS="NOMAX NOLACC NOCVT NOFOLD NOBAR NOREAD NODMA"
cd "$(dirname "$0")/../.."
if [ "$1" = build ]; then
  make -s -C frameino_amd/csrc
  # (sequential: --experiments re-creates the patched scratch copy of csrc/ for every build)
  for v in $VARS; do tools/debug/mkvar.sh --experiments frx_$v fino_attention_fp8.hip "-DFINO_EXPERIMENT -DFR_X_$v"; done
else
  echo "product:"; FINO_FP8_ONLY=1 python3 tools/attn_fp8_bench.py | grep TFLOP
  for v in $VARS; do
    echo "$v:"; FINO_ALLOW_EXPERIMENT=1 FINO_FP8_ONLY=1 FINO_LIB_PATH=frameino_amd/lib/libframeino_frx_$v.so python3 tools/attn_fp8_bench.py | grep TFLOP
  done
fi
