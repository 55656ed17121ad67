#!/bin/bash
# per-kernel average times of a command under rocprofv3 --kernel-trace: bash tools/trace_kernels.sh <tag> <command ...>
TAG=$1; shift
case "$(basename -- "$1")" in env|bash|sh|taskset|numactl) echo "$0: give the program itself (python3 ..., ./bench ...): a launcher in front of it is an exec hop behind the profiler's preloaded library, which this pool forbids; export variables before calling this script" >&2; exit 2;; esac
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/tk_$TAG; rm -rf $OUT; mkdir -p $OUT
ROOT=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- "$@" > $OUT/cmd.log 2>&1)
f=$(ls $OUT/*/*kernel_stats.csv | head -1)
python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print("%-60s calls %4s avg %9.1f us" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
