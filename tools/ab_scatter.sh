#!/bin/bash
# kernel times of the shuffled step for build variants (rocprofv3 kernel trace of 256 shuffled VCFs x 1 M): bash tools/ab_scatter.sh "<tag>=<flags>" ...
S=$GRAFT_REPO_ROOT/quasimodo_amd/csrc
export TMPDIR=/tmp
for spec in "$@"; do
  TAG=${spec%%=*}; FLAGS=${spec#*=}
  D=/tmp/abs/$TAG; mkdir -p $D
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FLAGS -c -o $D/k.o $S/qmvt_kernels.hip 2>$D/build.err || { echo "build failed: $TAG"; head -5 $D/build.err; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libqmvt.so $D/k.o $S/qmvt_api.o $S/qmvt_host.o $S/qmvt_pipeline.o -lz
done
for spec in "$@"; do
  TAG=${spec%%=*}
  rm -rf /tmp/abs/$TAG/tr
  (cd /tmp && QM_LIBQMVT=/tmp/abs/$TAG/libqmvt.so timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abs/$TAG/tr -- python3 $GRAFT_REPO_ROOT/tools/run_once.py 256 4 1 > /tmp/abs/$TAG/log.txt 2>&1)
  echo "== $TAG: $(grep classify /tmp/abs/$TAG/log.txt | sed 's/.*wall/wall/')"
  python3 - /tmp/abs/$TAG/tr <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if any(k in n for k in ("k_bucket_scatter", "k_join_direct", "k_compact", "k_tile_counts", "k_finalize")):
            print("   %-40s calls %4s avg %9.1f us" % (n.split("(")[0][-40:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
