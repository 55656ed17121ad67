#!/bin/bash
# the allele-extended shuffled shapes through the default library and tools/probe/ab/libqmvt_<tag>.so, interleaved on one box
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
cd $ROOT; export TMPDIR=/tmp
TAG=$1; REPS=${2:-2}
for r in $(seq $REPS); do
  for tag in "" $TAG; do
    if [ -n "$tag" ]; then export QM_LIBQMVT=$ROOT/tools/probe/ab/libqmvt_$tag.so; else unset QM_LIBQMVT; fi
    for a in "256" "64 2000000 10000000 200000"; do
      echo "${tag:-default} [$a] $(python3 tools/shuffled_ext.py $a | grep -E "shuffled=True|equal" | cut -c1-60 | tr '\n' ' ')"
    done
  done
done
for tag in "" $TAG; do
  if [ -n "$tag" ]; then export QM_LIBQMVT=$ROOT/tools/probe/ab/libqmvt_$tag.so; else unset QM_LIBQMVT; fi
  d=/tmp/abx_${tag:-default}
  (cd /tmp && timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $ROOT/tools/shuffled_ext.py 256 > /dev/null 2>&1)
  python3 - $d "${tag:-default}" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "join" in r["Name"] or "scatter" in r["Name"]:
        print(sys.argv[2], r["Name"][:60], r["Calls"], "%.1f us" % (float(r["AverageNs"]) / 1e3))
PY
done
