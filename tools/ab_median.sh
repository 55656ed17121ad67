#!/bin/bash
# same-box A/B of build variants on the SORTED main workload (1 000 VCFs x 1 M), interleaved ROUNDS times, medians reported:
#   ROUNDS=6 bash tools/ab_median.sh "<tag>=<flags>" ...
# (the allocation noise of this part is +-3 % from process to process -- DESIGN 5 -- so single draws cannot rank variants that
# differ by a few per cent)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
GRAFT_REPO_ROOT=$ROOT
S=$ROOT/quasimodo_amd/csrc
cd /tmp
for spec in "$@"; do
  TAG=${spec%%=*}; FLAGS=${spec#*=}
  D=/tmp/abm/$TAG; mkdir -p $D
  KSRC=$S/qmvt_kernels.hip
  if [ "${FLAGS:0:1}" = "@" ]; then KSRC=$GRAFT_REPO_ROOT/${FLAGS%% *}; KSRC=${KSRC/@/}; FLAGS="${FLAGS#* } -I$S"; [ "$FLAGS" = "${spec#*=} -I$S" ] && FLAGS="-I$S"; fi   # "<tag>=@<kernel source> <flags>": another kernels file (e.g. last round's)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FLAGS -c -o $D/k.o $KSRC 2>$D/build.err || { echo "build failed: $TAG"; head -5 $D/build.err; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FLAGS -x hip -c -o $D/a.o $S/qmvt_api.cpp 2>>$D/build.err || echo "api build failed: $TAG"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libqmvt.so $D/k.o $D/a.o $S/qmvt_host.o $S/qmvt_pipeline.o -lz
done
: > /tmp/abm/all.log
for rep in $(seq 1 ${ROUNDS:-5}); do for spec in "$@"; do
  TAG=${spec%%=*}
  echo -n "$TAG: " >> /tmp/abm/all.log; QM_LIBQMVT=/tmp/abm/$TAG/libqmvt.so python3 $GRAFT_REPO_ROOT/tools/run_once.py ${NV:-1000} ${RUNS:-8} ${RARGS:-} 2>&1 | grep classify >> /tmp/abm/all.log
done; done
python3 - <<'PY'
import re, collections, statistics
d = collections.defaultdict(list)
for line in open("/tmp/abm/all.log"):
    m = re.match(r"(\S+): .*classify ([\d.]+) ms .*finalize ([\d.]+) compact ([\d.]+) total ([\d.]+) wall ([\d.]+)", line)
    if m: d[m.group(1)].append(tuple(float(x) for x in m.groups()[1:]))
for tag, v in d.items():
    med = [statistics.median(x[i] for x in v) for i in range(5)]
    mn = [min(x[i] for x in v) for i in range(5)]
    print("%-12s n=%d  median: classify %.3f compact %.3f wall %.3f   min: classify %.3f compact %.3f wall %.3f" %
          (tag, len(v), med[0], med[2], med[4], mn[0], mn[2], mn[4]))
PY
