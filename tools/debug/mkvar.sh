#!/bin/bash
# Build a variant library that differs from the product build in ONE object (seconds instead of a full `make variant`):
#   tools/debug/mkvar.sh <name> <source.hip> "<extra flags>"   ->  frameino_amd/lib/libframeino_<name>.so
# e.g.  tools/debug/mkvar.sh xnodma fino_attention.hip "-DFINO_EXPERIMENT -DPD_X_NODMA"
#       tools/debug/mkvar.sh gorder0 fino_gemm.hip "-DGP_ORDER=0"
# Needs an up-to-date `make -C frameino_amd/csrc` (the other objects come from csrc/build/).  Load it with FINO_LIB_PATH.
set -e
cd "$(dirname "$0")/../../frameino_amd/csrc"
name=$1; src=$2; flags=$3
extra=""
[ "$src" = fino_elementwise.hip ] && extra="-ffp-contract=off"
[ "$src" = fino_attention_w4.hip ] && extra="-fno-honor-nans"
mkdir -p build/$name
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $extra $flags -x hip -c $src -o build/$name/$src.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libframeino_$name.so build/$name/$src.o $(ls build/*.o | grep -v "build/$src.o")
echo "frameino_amd/lib/libframeino_$name.so"
