#!/bin/bash
# after `gpurun -- bash tools/final_profiles.sh <tag> 1|2`: copy what is judged from gpurun_out/ into profiles/   (bash tools/copy_profiles.sh r06f r06)
TAG=${1:-r06f}; R=${2:-r06}
st() { ls -t $(find gpurun_out/trace_$1 -name '*kernel_stats.csv') | head -1; }   # the newest run of that tag
c() { [ -e "$1" ] && cp "$1" "$2" || echo "missing: $1"; }
# the bench traces: statistics of the TIMED launches only (tools/trace_stats.py; the tracer's own file averages the warm-up launches in)
c gpurun_out/trace_$TAG/kernel_stats_timed.csv profiles/${R}_kernel_stats.csv
c gpurun_out/trace_${TAG}shuf/kernel_stats_timed.csv profiles/${R}_kernel_stats_shuffled.csv
c gpurun_out/trace_${TAG}c3/kernel_stats_timed.csv profiles/${R}_kernel_stats_config3.csv
c gpurun_out/trace_${TAG}c4/kernel_stats_timed.csv profiles/${R}_kernel_stats_config4.csv
c "$(st ${TAG}shuf3)" profiles/${R}_kernel_stats_shuffled_config3.csv
c "$(st ${TAG}alle)" profiles/${R}_kernel_stats_alleles.csv
c "$(st ${TAG}mc)" profiles/${R}_kernel_stats_multicontig.csv
c "$(st ${TAG}shufx)" profiles/${R}_kernel_stats_shuffled_alleles.csv
c "$(st ${TAG}shufx4)" profiles/${R}_kernel_stats_shuffled_config4.csv
c gpurun_out/${R}_pmc_per_launch.json profiles/${R}_pmc_per_launch.json
c gpurun_out/${R}_pmc_per_launch_alleles.json profiles/${R}_pmc_per_launch_alleles.json
c gpurun_out/traffic.json profiles/traffic.json
c gpurun_out/${TAG}_pmc_shuffled.json profiles/${R}_pmc_shuffled.json
c gpurun_out/${TAG}_pmc_shuffled_alleles.json profiles/${R}_pmc_shuffled_alleles.json
c gpurun_out/${TAG}_pmc_shuffled_config3.json profiles/${R}_pmc_shuffled_config3.json
c gpurun_out/${TAG}_shuffled_ext.log profiles/${R}_shuffled_ext.log
c gpurun_out/${TAG}_e2e.log profiles/${R}_e2e_files.log
c gpurun_out/${TAG}_pmc_compact.log profiles/${R}_pmc_compact.log
for k in "" _config3 _config4; do
  c gpurun_out/${TAG}_bench${k}_detail.json profiles/${R}_bench${k:-_1gpu}.json      # the nested record (every side figure with its notes)
  [ -e gpurun_out/${TAG}_bench${k}.json ] && tail -1 gpurun_out/${TAG}_bench${k}.json > profiles/${R}_bench${k:-_1gpu}_line.json   # the flat line as printed
done
grep -h '^{' gpurun_out/trace_$TAG/bench.json | tail -1 > profiles/${R}_bench_under_rocprof.json
grep -h '^{' gpurun_out/trace_${TAG}shuf/bench.json | tail -1 > profiles/${R}_bench_under_rocprof_shuffled.json
