#!/bin/bash
# Build a variant library that differs from the product build in ONE object (seconds instead of a full `make variant`):
#   tools/debug/mkvar.sh [--experiments] <name> <source.hip> "<extra flags>"   ->  frameino_amd/lib/libframeino_<name>.so
# e.g.  tools/debug/mkvar.sh gorder0 fino_gemm.hip "-DGP_ORDER=0"                                  (a same-bits A/B knob)
#       tools/debug/mkvar.sh --experiments xnodma fino_attention.hip "-DFINO_EXPERIMENT -DPD_X_NODMA"
# --experiments: the wrong-result timing switches (PD_X_* / PW_X_* / F8_X_* / FR_X_* / GP_X_* / W4_X_* / FINO_GEMM_DESYNC_EXP) are
# not in the product sources; they live in tools/debug/experiments.patch, which is applied to a scratch copy of csrc/ first
# (tools/debug/strip_experiments.py --apply).  Such a library reports a negative fino_version() and loads only with
# FINO_ALLOW_EXPERIMENT=1.
# Needs an up-to-date `make -C frameino_amd/csrc` (the other objects come from csrc/build/).  Load it with FINO_LIB_PATH.
set -e
here="$(cd "$(dirname "$0")" && pwd)"
csrc="$here/../../frameino_amd/csrc"
srcdir="$csrc"
if [ "$1" = "--experiments" ]; then
    shift
    srcdir="$here/../../gpurun_out/csrc_exp"   # scratch, outside the package (two levels under the root: the sources include ../../include/frameino_hip.h)
    mkdir -p "$(dirname "$srcdir")"
    python3 "$here/strip_experiments.py" --apply "$srcdir" > /dev/null
fi
name=$1; src=$2; flags=$3
extra=""
[ "$src" = fino_elementwise.hip ] && extra="-ffp-contract=off"
[ "$src" = fino_attention_w4.hip ] && extra="-fno-honor-nans"
mkdir -p "$csrc/build/$name"
( cd "$srcdir" && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $extra $flags -x hip -c $src -o "$csrc/build/$name/$src.o" )
objs="$csrc/build/$name/$src.o $(ls $csrc/build/*.o | grep -v "build/$src.o")"
if echo "$flags" | grep -q FINO_EXPERIMENT; then      # the version export must say "experiment" too
    ( cd "$srcdir" && hipcc -O3 -std=c++17 -fPIC -DFINO_EXPERIMENT -c fino_api.cpp -o "$csrc/build/$name/fino_api.cpp.o" )
    objs="$csrc/build/$name/$src.o $csrc/build/$name/fino_api.cpp.o $(ls $csrc/build/*.o | grep -v "build/$src.o" | grep -v "build/fino_api.cpp.o")"
fi
hipcc --offload-arch=gfx950 -shared -fPIC -o "$csrc/../lib/libframeino_$name.so" $objs
echo "frameino_amd/lib/libframeino_$name.so"
