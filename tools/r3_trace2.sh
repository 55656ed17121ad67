#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_t2
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t2 -- python3 $R/tools/join_ab.py 16 10000000 50000000 1000000 > $R/gpurun_out/r3_trace2.log 2>&1
f=$(find /tmp/prof_t2 -name '*kernel_stats.csv' | head -1)
cut -d, -f1-4 "$f" | sed 's/qm:://g' | cut -c1-120 | head -24
grep join= $R/gpurun_out/r3_trace2.log
