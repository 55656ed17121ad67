#!/bin/bash
# Build a variant of libqmvt.so with extra -D flags and time the shuffled (radix-sort) path on 256 VCFs.
# usage: bash tools/ab_sort.sh <tag> "<-Dflags>"
set -e
TAG=$1; FLAGS=$2
D=$PWD/gpurun_out/ab/$TAG; mkdir -p $D
S=$PWD/quasimodo_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FLAGS -c -o $D/k.o $S/qmvt_kernels.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FLAGS -x hip -c -o $D/a.o $S/qmvt_api.cpp
g++ -O3 -std=c++17 -fPIC -pthread -c -o $D/h.o $S/qmvt_host.cpp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libqmvt.so $D/k.o $D/a.o $D/h.o
echo -n "$TAG [$FLAGS]: "
QM_LIBQMVT=$D/libqmvt.so python3 bench.py --shuffled --vcfs 256 --steps 4 --warmup 1 --cpu-sample 0 --shell-sample 0 --alleles-vcfs 0 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('%.2f ms/step  %.3e rec/s' % (d['ms_per_step'], d['value']))"
