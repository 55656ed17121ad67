#!/bin/bash
# round 6, VERDICT 5 next #1(a): BASELINE configs[3] and configs[4] per-GPU shards at N = 1 as bench lines, a kernel trace of
# the configs[4] shard, and PMC passes of k_classify<false,true> beside <false,false>'s (1 000 VCFs x 1 M each).
# usage (on the GPU box): bash tools/r6_configs.sh <tag>      outputs under gpurun_out/
TAG=${1:-r06a}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python3 bench.py --config 3 > gpurun_out/${TAG}_bench_config3.json 2> gpurun_out/${TAG}_bench_config3.err || { echo config3 failed; tail -5 gpurun_out/${TAG}_bench_config3.err; exit 1; }
tail -c 400 gpurun_out/${TAG}_bench_config3.json; echo
python3 bench.py --config 4 > gpurun_out/${TAG}_bench_config4.json 2> gpurun_out/${TAG}_bench_config4.err || { echo config4 failed; tail -5 gpurun_out/${TAG}_bench_config4.err; exit 1; }
tail -c 400 gpurun_out/${TAG}_bench_config4.json; echo
bash tools/prof_trace.sh ${TAG}c4 --config 4 --steps 5 --warmup 2 > gpurun_out/${TAG}_trace_config4.log 2>&1
bash tools/prof_trace.sh ${TAG}c3 --config 3 --steps 5 --warmup 2 > gpurun_out/${TAG}_trace_config3.log 2>&1
# PMC: the allele-extended instantiation on configs[4]'s record shape, and the default one beside it
RARGS="0 30" bash tools/prof_pmc.sh ${TAG}x 1000 > gpurun_out/${TAG}_pmc_alleles.log 2>&1
bash tools/prof_pmc.sh ${TAG} 1000 > gpurun_out/${TAG}_pmc_default.log 2>&1
tail -8 gpurun_out/${TAG}_pmc_alleles.log
echo done
