#!/bin/bash
# phase clocks of k_join_ext (a -DHB_PROFILE -DXJ_PROFILE build, QM_HB_PROFILE=1): ticks per workgroup and phase on stderr
S=$GRAFT_REPO_ROOT/quasimodo_amd/csrc
D=/tmp/ab/profx; mkdir -p $D
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DHB_PROFILE -DXJ_PROFILE $1 -c -o $D/k.o $S/qmvt_kernels.hip || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libqmvt.so $D/k.o $S/qmvt_api.o $S/qmvt_host.o $S/qmvt_pipeline.o -lz
QM_LIBQMVT=$D/libqmvt.so QM_HB_PROFILE=1 python3 $GRAFT_REPO_ROOT/tools/shuffled_ext.py 64 2>&1 | grep -v amdgpu.ids | grep -E "dj profile|shuffled=True" | tail -3
