#!/bin/bash
# same-box A/B of ENVIRONMENT variants (no rebuild) on the SHUFFLED workload (256 VCFs x 1 M, bucket path), interleaved:
#   ROUNDS=4 bash tools/ab_env_shuf.sh "<tag>=<VAR=value ...>" ...
cd /tmp
: > /tmp/abenvs.log
for rep in $(seq 1 ${ROUNDS:-4}); do for spec in "$@"; do
  TAG=${spec%%=*}; ENVS=${spec#*=}
  echo -n "$TAG: " >> /tmp/abenvs.log
  ( export $ENVS; python3 $GRAFT_REPO_ROOT/tools/run_once.py ${NV:-256} ${RUNS:-6} 1 ${PCT:-0} 2>&1 | grep -A1 classify | tr '\n' ' ' >> /tmp/abenvs.log; echo >> /tmp/abenvs.log )
done; done
cat /tmp/abenvs.log
