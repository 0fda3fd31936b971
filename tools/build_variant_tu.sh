#!/bin/bash
# Build an experimental variant of the kernel library with ONE translation unit recompiled with extra flags:
#   tools/build_variant_tu.sh <tag> <file.hip> <extra hipcc flags>   -> nfdp_hip/libnfdp_hip_<tag>.so (select with NFDP_LIB_PATH=...)
# The other objects are re-used from build/ (run __graft_entry__.build() first).
set -e
TAG=$1; TU=$2; shift 2
PKG=$(dirname "$0")/../normalizing-flow-with-diffusion-prior-model_amd
mkdir -p $PKG/build_$TAG
F="-DNFDP_DIAG -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result"
BASE=$(basename $TU .hip)
hipcc -x hip $F "$@" -c $PKG/csrc/$TU -o $PKG/build_$TAG/$BASE.o
OBJS=$(ls $PKG/build/*.o | grep -v "/$BASE.o")
hipcc -shared -fPIC --offload-arch=gfx950 -o $PKG/nfdp_hip/libnfdp_hip_$TAG.so $OBJS $PKG/build_$TAG/$BASE.o
echo $PKG/nfdp_hip/libnfdp_hip_$TAG.so
