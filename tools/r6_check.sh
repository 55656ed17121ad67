#!/bin/bash
# round 6: the GPU suite, the bench line (config 2, flat), the configs[4] shard and PMC of the allele-extended k_classify in one call
# usage (on the GPU box): bash tools/r6_check.sh <tag> [steps: t=tests b=bench 3=config3 4=config4 p=pmc-alleles d=pmc-default]
TAG=${1:-r06b}; STEPS=${2:-tb4p}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
has() { case "$STEPS" in *$1*) return 0;; esac; return 1; }
if has t; then
  timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_pytest.log 2>&1 || { tail -30 gpurun_out/${TAG}_pytest.log; exit 1; }
  tail -2 gpurun_out/${TAG}_pytest.log
fi
if has b; then
  python3 bench.py --detail gpurun_out/${TAG}_bench_detail.json > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || { tail -5 gpurun_out/${TAG}_bench.err; exit 1; }
  wc -c gpurun_out/${TAG}_bench.json; cat gpurun_out/${TAG}_bench.json
fi
for c in 3 4; do
  if has $c; then
    python3 bench.py --config $c --detail gpurun_out/${TAG}_bench_config${c}_detail.json > gpurun_out/${TAG}_bench_config$c.json 2> gpurun_out/${TAG}_bench_config$c.err || { tail -5 gpurun_out/${TAG}_bench_config$c.err; exit 1; }
    cat gpurun_out/${TAG}_bench_config$c.json
  fi
done
if has p; then RARGS="0 30" bash tools/prof_pmc.sh ${TAG}x 1000 > gpurun_out/${TAG}_pmc_alleles.log 2>&1; grep "classify" gpurun_out/${TAG}_pmc_alleles.log; fi
if has d; then bash tools/prof_pmc.sh ${TAG} 1000 > gpurun_out/${TAG}_pmc_default.log 2>&1; grep "classify" gpurun_out/${TAG}_pmc_default.log; fi
echo done
