#!/bin/bash
# rocprofv3 kernel trace + stats of the bench command (no PMC in this run, as the pool requires).
# usage: bash tools/prof_trace.sh <tag> [bench args...]
set -e
TAG=${1:-x}; shift || true
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/trace_$TAG
mkdir -p $OUT
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/bench.py --cpu-sample 0 --shell-sample 0 --shuffled-vcfs 0 --shuffled3-vcfs 0 --shuffled-alleles-vcfs 0 --shuffled4-vcfs 0 --alleles-vcfs 0 --multicontig-vcfs 0 --alloc-reps 0 --detail "" "$@" > $OUT/bench.json 2> $OUT/bench.err) || true
cat $OUT/bench.json
# per-kernel statistics of the TIMED launches (the tracer's own *_kernel_stats.csv averages the warm-up launches in)
python3 $ROOT/tools/trace_stats.py $OUT $(python3 -c "import json,sys; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print(d['steps'], d['warmup'])") > $OUT/kernel_stats_timed.csv
echo "== $OUT/kernel_stats_timed.csv"; cat $OUT/kernel_stats_timed.csv
