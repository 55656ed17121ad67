#!/bin/bash
# k_compact build variants on PHYSICALLY CONTIGUOUS batches (QM_ALLOC_CONTIG: the deterministic setting -- DESIGN 4.3) and on plain
# ones: bash tools/ab_contig.sh "<tag>=<flags>" ...     (timing builds may switch stores off: results are wrong then)
S=$GRAFT_REPO_ROOT/quasimodo_amd/csrc
cd /tmp
for spec in "$@"; do
  TAG=${spec%%=*}; FLAGS=${spec#*=}
  D=/tmp/abc/$TAG; mkdir -p $D
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FLAGS -c -o $D/k.o $S/qmvt_kernels.hip 2>$D/build.err || { echo "build failed: $TAG"; head -5 $D/build.err; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libqmvt.so $D/k.o $S/qmvt_api.o $S/qmvt_host.o $S/qmvt_pipeline.o -lz
done
for spec in "$@"; do
  TAG=${spec%%=*}
  for mode in 64 0; do
    echo "== $TAG contig=$mode"
    QM_LIBQMVT=/tmp/abc/$TAG/libqmvt.so QM_ALLOC_CONTIG=$mode REPS=${REPS:-2} python3 $GRAFT_REPO_ROOT/tools/compact_map_probe.py 2>&1 | grep batch | sed 's/classify [0-9.]*//g'
  done
done
