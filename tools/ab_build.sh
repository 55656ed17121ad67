#!/bin/bash
# Build a variant of libqmvt.so with extra -D flags into gpurun_out/ab/<tag>/ and time it.
# usage: bash tools/ab_build.sh <tag> "<-Dflags>" [n_vcf]
set -e
TAG=$1; FLAGS=$2; NV=${3:-256}
D=$PWD/gpurun_out/ab/$TAG; mkdir -p $D
S=$PWD/quasimodo_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FLAGS -c -o $D/k.o $S/qmvt_kernels.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FLAGS -x hip -c -o $D/a.o $S/qmvt_api.cpp
g++ -O3 -std=c++17 -fPIC -c -o $D/h.o $S/qmvt_host.cpp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libqmvt.so $D/k.o $D/a.o $D/h.o -lz
echo -n "$TAG [$FLAGS]: "; QM_LIBQMVT=$D/libqmvt.so python3 tools/run_once.py $NV 6 2>&1 | grep -v amdgpu.ids
