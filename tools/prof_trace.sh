#!/bin/bash
# rocprofv3 kernel trace + stats of the bench command (no PMC in this run, as the pool requires).
# usage: bash tools/prof_trace.sh <tag> [bench args...]
set -e
TAG=${1:-x}; shift || true
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/trace_$TAG
mkdir -p $OUT
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/bench.py --cpu-sample 0 --shell-sample 0 --shuffled-vcfs 0 --shuffled3-vcfs 0 --shuffled-alleles-vcfs 0 --shuffled4-vcfs 0 --alleles-vcfs 0 --alloc-reps 0 "$@" > $OUT/bench.json 2> $OUT/bench.err) || true
cat $OUT/bench.json
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'echo "== {}"; cat {}'
