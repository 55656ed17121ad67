#!/bin/bash
# A/B of a kernels build on ONE box: the same shuffled batches through the default library and through tools/probe/ab/libqmvt_<tag>.so,
# interleaved; step times and the trace's per-kernel averages.   usage: bash tools/probe/ab/run.sh <tag> [reps]
# (the other build: `git archive <rev> quasimodo_amd/csrc include | tar -x -C /tmp/x`, `make` in its csrc with include/ two levels up, copy libqmvt.so to
# tools/probe/ab/libqmvt_<tag>.so -- git-ignored, travels with gpurun)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
cd $ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=$1; REPS=${2:-3}
for r in $(seq $REPS); do
  for tag in "" $TAG; do
    if [ -n "$tag" ]; then export QM_LIBQMVT=$ROOT/tools/probe/ab/libqmvt_$tag.so; else unset QM_LIBQMVT; fi
    for a in "16 10000000 50000000 1000000" "80 2000000 50000000 1000000" "256 1000000 5000000 100000"; do
      echo "${tag:-default} $(python3 tools/join_ab.py $a | tail -1)"
    done
  done
done
for tag in "" $TAG; do
  if [ -n "$tag" ]; then export QM_LIBQMVT=$ROOT/tools/probe/ab/libqmvt_$tag.so; else unset QM_LIBQMVT; fi
  for a in "16 10000000 50000000 1000000" "256 1000000 5000000 100000"; do
    d=/tmp/ab_${tag:-default}_${a%% *}
    (cd /tmp && timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $ROOT/tools/join_ab.py $a > /dev/null 2>&1)
    python3 - $d "${tag:-default} $a" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "join" in r["Name"] or "scatter" in r["Name"]:
        print(sys.argv[2], r["Name"][:60], r["Calls"], "%.1f us" % (float(r["AverageNs"]) / 1e3))
PY
  done
done
