#!/bin/bash
# usage: tools/ab_build.sh <tag> [extra hipcc flags, e.g. -DSDM_X=1]  ->  build/ab/libsdm_<tag>.so
# A/B variants of the library for one and the same GPU run (box-to-box spread is a few percent: variants are only
# comparable inside one gpurun call); load one with SDM_LIB_PATH=build/ab/libsdm_<tag>.so.
set -e
tag=$1; shift
cd "$(dirname "$0")/../semantic_dsp_map_amd/csrc"
make -s -j8
mkdir -p ../../build/ab
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-unused-result -Wno-unused-value"
for f in kernels moves map; do /opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS "$@" -c $f.hip -o ../../build/ab/${f}_$tag.o & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/ab/libsdm_$tag.so primitives.o ../../build/ab/kernels_$tag.o ../../build/ab/moves_$tag.o ../../build/ab/map_$tag.o objects.o -L/opt/rocm/lib -lrocrand -lrccl -Wl,-rpath,/opt/rocm/lib
rm -f ../../build/ab/*_$tag.o
echo built build/ab/libsdm_$tag.so
