#!/bin/bash
# the kernels of the LAST step of a command, with start offsets, durations and the gaps in front of them (rocprofv3 --kernel-trace):
#   bash tools/step_timeline.sh <tag> <first kernel of a step, substring> <command ...>
TAG=$1; FIRST=$2; shift; shift
case "$(basename -- "$1")" in env|bash|sh|taskset|numactl) echo "$0: give the program itself (python3 ..., ./bench ...): a launcher in front of it is an exec hop behind the profiler's preloaded library, which this pool forbids; export variables before calling this script" >&2; exit 2;; esac
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/tl_$TAG; rm -rf $OUT; mkdir -p $OUT
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -- "$@" > $OUT/cmd.log 2>&1)
f=$(ls $OUT/*/*kernel_trace.csv | head -1)
python3 - $f "$FIRST" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]]
i0 = first[-1]
t0 = int(rows[i0]["Start_Timestamp"]); prev_end = t0
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  +%7.1f gap  %8.1f us  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r["Kernel_Name"].split("(")[0][-50:]))
    prev_end = max(prev_end, e)
print("step span %.1f us" % ((prev_end - t0) / 1e3))
PY
