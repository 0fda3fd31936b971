#!/bin/bash
# Build an experimental variant of the kernel library next to the production one:
#   tools/build_variant.sh <tag> <extra hipcc flags for the two fused-kernel translation units>
# -> nfdp_hip/libnfdp_hip_<tag>.so (select it with NFDP_LIB_PATH=...).  The other objects are re-used from build/.
set -e
TAG=$1; shift
PKG=$(dirname "$0")/../normalizing-flow-with-diffusion-prior-model_amd
mkdir -p $PKG/build_$TAG
F="-DNFDP_DIAG -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result"
hipcc -x hip $F -mllvm -amdgpu-mfma-vgpr-form "$@" -c $PKG/csrc/coupling_fused_v.hip -o $PKG/build_$TAG/coupling_fused_v.o &
hipcc -x hip $F "$@" -c $PKG/csrc/coupling_fused_a.hip -o $PKG/build_$TAG/coupling_fused_a.o &
wait
OBJS=$(ls $PKG/build/*.o | grep -v coupling_fused_)
hipcc -shared -fPIC --offload-arch=gfx950 -o $PKG/nfdp_hip/libnfdp_hip_$TAG.so $OBJS $PKG/build_$TAG/coupling_fused_v.o $PKG/build_$TAG/coupling_fused_a.o
echo $PKG/nfdp_hip/libnfdp_hip_$TAG.so
