#!/bin/bash
# same-box A/B of kernel build variants on the shuffled workload, scatter and join one after the other (QM_BUCKET_PARTS=1)
# under the kernel trace: bash tools/ab_join.sh "<tag>=<flags>" ...   -> average k_join_direct / k_bucket_scatter time per launch
S=$GRAFT_REPO_ROOT/quasimodo_amd/csrc
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  TAG=${spec%%=*}; FLAGS=${spec#*=}
  D=/tmp/ab/$TAG; mkdir -p $D
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FLAGS -c -o $D/k.o $S/qmvt_kernels.hip 2>$D/build.err || { echo "build failed: $TAG"; head -5 $D/build.err; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FLAGS -x hip -c -o $D/a.o $S/qmvt_api.cpp 2>>$D/build.err || echo "api build failed: $TAG"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libqmvt.so $D/k.o $D/a.o $S/qmvt_host.o $S/qmvt_pipeline.o -lz
done
for spec in "$@"; do
  TAG=${spec%%=*}
  rm -rf /tmp/prof_$TAG
  QM_LIBQMVT=/tmp/ab/$TAG/libqmvt.so QM_BUCKET_PARTS=${PARTS:-1} rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python3 $GRAFT_REPO_ROOT/tools/join_ab.py $ABARGS > /tmp/prof_$TAG.log 2>&1
  f=$(find /tmp/prof_$TAG -name '*kernel_stats.csv' | head -1)
  echo "== $TAG: $(grep join= /tmp/prof_$TAG.log)"
  grep -E "k_join_direct|k_bucket_scatter|k_classify_hash|k_compact" "$f" | cut -d, -f1-4,6,7 | sed 's/qm:://g'
done
