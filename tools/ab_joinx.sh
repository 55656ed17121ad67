#!/bin/bash
# same-box A/B of kernel build variants on shuffled allele-extended VCFs under the kernel trace: bash tools/ab_joinx.sh "<tag>=<flags>" ...
S=$GRAFT_REPO_ROOT/quasimodo_amd/csrc
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  TAG=${spec%%=*}; FLAGS=${spec#*=}
  D=/tmp/abx/$TAG; mkdir -p $D
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FLAGS -c -o $D/k.o $S/qmvt_kernels.hip 2>$D/build.err || { echo "build failed: $TAG"; head -5 $D/build.err; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FLAGS -x hip -c -o $D/a.o $S/qmvt_api.cpp 2>>$D/build.err || echo "api build failed: $TAG"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libqmvt.so $D/k.o $D/a.o $S/qmvt_host.o $S/qmvt_pipeline.o -lz
done
for spec in "$@"; do
  TAG=${spec%%=*}
  rm -rf /tmp/profx_$TAG
  QM_LIBQMVT=/tmp/abx/$TAG/libqmvt.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profx_$TAG -- python3 $GRAFT_REPO_ROOT/tools/shuffled_ext.py ${NV:-256} > /tmp/profx_$TAG.log 2>&1
  f=$(find /tmp/profx_$TAG -name '*kernel_stats.csv' | head -1)
  echo "== $TAG: $(grep -E 'shuffled=True|equal' /tmp/profx_$TAG.log | cut -c1-60 | tr '\n' ' ')"
  grep -E "k_join_direct|k_join_ext|k_bucket_scatter" "$f" | cut -d, -f1-4 | sed 's/qm:://g'
done
