#!/bin/bash
# after `gpurun -- bash tools/final_profiles.sh <tag>`: copy what is judged from gpurun_out/ into profiles/   (bash tools/copy_profiles.sh r03f r03)
TAG=${1:-r05f}; R=${2:-r05}
st() { ls -t $(find gpurun_out/trace_$1 -name '*kernel_stats.csv') | head -1; }   # the newest run of that tag
# the bench traces: statistics of the TIMED launches only (tools/trace_stats.py; the tracer's own file averages the warm-up launches in)
cp gpurun_out/trace_$TAG/kernel_stats_timed.csv profiles/${R}_kernel_stats.csv
cp gpurun_out/trace_${TAG}shuf/kernel_stats_timed.csv profiles/${R}_kernel_stats_shuffled.csv
cp "$(st ${TAG}shuf3)" profiles/${R}_kernel_stats_shuffled_config3.csv
cp "$(st ${TAG}alle)" profiles/${R}_kernel_stats_alleles.csv
cp "$(st ${TAG}shufx)" profiles/${R}_kernel_stats_shuffled_alleles.csv
cp "$(st ${TAG}shufx4)" profiles/${R}_kernel_stats_shuffled_config4.csv
cp gpurun_out/${R}_pmc_per_launch.json profiles/${R}_pmc_per_launch.json
cp gpurun_out/traffic.json profiles/traffic.json
cp gpurun_out/${TAG}_pmc_shuffled.json profiles/${R}_pmc_shuffled.json
cp gpurun_out/${TAG}_pmc_shuffled_alleles.json profiles/${R}_pmc_shuffled_alleles.json
cp gpurun_out/${TAG}_shuffled_ext.log profiles/${R}_shuffled_ext.log
cp gpurun_out/${TAG}_e2e.log profiles/${R}_e2e_files.log
tail -1 gpurun_out/${TAG}_bench.json > profiles/${R}_bench_1gpu.json
grep -h '^{' gpurun_out/${TAG}_trace.log | tail -1 > profiles/${R}_bench_under_rocprof.json
grep -h '^{' gpurun_out/${TAG}_trace_shuf.log | tail -1 > profiles/${R}_bench_under_rocprof_shuffled.json
cp gpurun_out/${TAG}_pmc_compact.log profiles/${R}_pmc_compact.log
