#!/bin/bash
# k_classify build variants against the allocation lottery: every variant on REPS fresh batches (tools/col_skew_probe.py, five
# separate column allocations).  bash tools/ab_classify.sh "<tag>=<flags>" ...
S=$GRAFT_REPO_ROOT/quasimodo_amd/csrc
cd /tmp
for spec in "$@"; do
  TAG=${spec%%=*}; FLAGS=${spec#*=}
  D=/tmp/abk/$TAG; mkdir -p $D
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FLAGS -c -o $D/k.o $S/qmvt_kernels.hip 2>$D/build.err || { echo "build failed: $TAG"; head -5 $D/build.err; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FLAGS -x hip -c -o $D/a.o $S/qmvt_api.cpp 2>>$D/build.err || echo "api build failed: $TAG"   # (the layout constants are shared)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libqmvt.so $D/k.o $D/a.o $S/qmvt_host.o $S/qmvt_pipeline.o -lz
done
for round in 1 2; do
for spec in "$@"; do
  TAG=${spec%%=*}
  echo "== $TAG (round $round)"
  QM_LIBQMVT=/tmp/abk/$TAG/libqmvt.so SKEWS="${SKEWS:-none}" REPS=${REPS:-4} python3 $GRAFT_REPO_ROOT/tools/col_skew_probe.py 2>&1 | grep skew
done
done
