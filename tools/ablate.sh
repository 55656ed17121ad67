#!/bin/bash
# time k_classify with phases switched off (QM_ABLATE bits: 1 join, 2 histogram, 4 masks, 8 R-path dedupe).
# The switch only exists in a debug build of the kernels (-DQM_ABLATE_SUPPORT), made here; results are WRONG when it is on.
set -e
D=$PWD/gpurun_out/ab/ablate; mkdir -p $D
S=$PWD/quasimodo_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DQM_ABLATE_SUPPORT -c -o $D/k.o $S/qmvt_kernels.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libqmvt.so $D/k.o $S/qmvt_api.o $S/qmvt_host.o $S/qmvt_pipeline.o -lz
for ab in ${ABLATES:-0 1 2 4 3 7 15}; do
  echo -n "ablate=$ab: "; QM_LIBQMVT=$D/libqmvt.so QM_ABLATE=$ab python3 tools/run_once.py ${NV:-256} 6 2>&1 | grep -v amdgpu.ids
done
