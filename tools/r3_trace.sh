#!/bin/bash
# kernel trace of the shuffled path (direct join); summary -> gpurun_out/r3_trace_<tag>.csv
cd /tmp && export TMPDIR=/tmp
TAG=${1:-direct}
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python3 $R/tools/join_ab.py > $R/gpurun_out/r3_trace_$TAG.log 2>&1
f=$(find /tmp/prof_$TAG -name '*kernel_stats.csv' | head -1)
cp "$f" $R/gpurun_out/r3_trace_$TAG.csv
cut -d, -f1-8 $R/gpurun_out/r3_trace_$TAG.csv | head -20
