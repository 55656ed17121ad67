#!/bin/bash
# same-box A/B of build variants on the SORTED main workload (1 000 VCFs x 1 M): bash tools/ab_main.sh "<tag>=<flags>" ...
S=$GRAFT_REPO_ROOT/quasimodo_amd/csrc
cd /tmp
for spec in "$@"; do
  TAG=${spec%%=*}; FLAGS=${spec#*=}
  D=/tmp/abm/$TAG; mkdir -p $D
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FLAGS -c -o $D/k.o $S/qmvt_kernels.hip 2>$D/build.err || { echo "build failed: $TAG"; head -5 $D/build.err; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FLAGS -x hip -c -o $D/a.o $S/qmvt_api.cpp 2>>$D/build.err || echo "api build failed: $TAG"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libqmvt.so $D/k.o $D/a.o $S/qmvt_host.o $S/qmvt_pipeline.o -lz
done
for rep in 1 2; do for spec in "$@"; do
  TAG=${spec%%=*}
  echo -n "$TAG: "; QM_LIBQMVT=/tmp/abm/$TAG/libqmvt.so python3 $GRAFT_REPO_ROOT/tools/run_once.py ${NV:-1000} 8 ${RARGS:-} 2>&1 | grep classify
done; done
