#!/bin/bash
# k_compact's duration against what it fetched from HBM, dispatch by dispatch, over several processes (the kernel has a fast and a
# slow mode that changes with where a batch landed in memory): rocprofv3 --pmc <counters> --kernel-trace, one process per round.
# usage: bash tools/pmc_compact_modes.sh <tag> [rounds] [counters...]
export TMPDIR=/tmp
ROOT=$PWD
TAG=${1:-x}; ROUNDS=${2:-6}; shift; shift
CNT=${@:-FETCH_SIZE}
OUT=$ROOT/gpurun_out/pmc_modes_$TAG; mkdir -p $OUT
for r in $(seq 1 $ROUNDS); do
  (cd /tmp && timeout -k 10 150 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT/r$r -- python3 $ROOT/tools/run_once.py 1000 3 > $OUT/log$r.txt 2>&1) || echo "round $r failed (see $OUT/log$r.txt)"
done
python3 - $OUT <<'PY'
import csv, glob, collections, sys, os
for d in sorted(glob.glob(sys.argv[1] + "/r*")):
    dur = {}
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_compact" in row["Kernel_Name"]:
                dur[row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
    cnt = collections.defaultdict(dict)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_compact" in row["Kernel_Name"]:
                cnt[row["Dispatch_Id"]][row["Counter_Name"]] = float(row["Counter_Value"])
    for k in sorted(dur, key=int):
        print(os.path.basename(d), "dispatch", k, "%.3f ms" % dur[k], {c: "%.4g" % v for c, v in cnt.get(k, {}).items()})
PY
