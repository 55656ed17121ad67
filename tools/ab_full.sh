#!/bin/bash
# same-box A/B of build variants whose flags touch the host side too (kernels AND api compiled with them), shuffled workload,
# per-kernel times from rocprofv3: bash tools/ab_full.sh "<tag>=<flags>" ...
S=$PWD/quasimodo_amd/csrc
ROOT=$PWD
export TMPDIR=/tmp
for spec in "$@"; do
  TAG=${spec%%=*}; FLAGS=${spec#*=}
  D=$ROOT/gpurun_out/ab/$TAG; mkdir -p $D
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FLAGS -c -o $D/k.o $S/qmvt_kernels.hip 2>/dev/null || echo "build failed: $TAG"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FLAGS -c -o $D/a.o $S/qmvt_api.cpp 2>/dev/null || echo "build failed (api): $TAG"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libqmvt.so $D/k.o $D/a.o $S/qmvt_host.o $S/qmvt_pipeline.o -lz
done
for spec in "$@"; do
  TAG=${spec%%=*}
  D=$ROOT/gpurun_out/ab/$TAG
  rm -rf $D/prof
  (cd /tmp && QM_LIBQMVT=$D/libqmvt.so rocprofv3 --kernel-trace --stats --output-format csv -d $D/prof -o p -- python3 $ROOT/tools/run_once.py ${NV:-256} 6 1 > $D/run.log 2>&1)
  echo "== $TAG"
  python3 - $D/prof <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if any(k in n for k in ("hash", "bucket_scatter")):
        print("  %-40s calls %3s avg %9.1f us" % (n.split("(")[0][:40], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
