#!/bin/bash
# same-box A/B of ENVIRONMENT variants on the SHUFFLED workload run FIRST-SEEN (QM_MEMO=0: optimistic pass, flags read back, bucket
# path), interleaved:   ROUNDS=4 bash tools/ab_env_unseen.sh "<tag>=<VAR=value ...>" ...
cd /tmp
: > /tmp/abenvu.log
for rep in $(seq 1 ${ROUNDS:-4}); do for spec in "$@"; do
  TAG=${spec%%=*}; ENVS=${spec#*=}
  echo -n "$TAG: " >> /tmp/abenvu.log
  ( export QM_MEMO=0 $ENVS; python3 $GRAFT_REPO_ROOT/tools/run_once.py ${NV:-256} ${RUNS:-6} 1 ${PCT:-0} 2>&1 | grep -o "wall [0-9.]*" >> /tmp/abenvu.log )
done; done
cat /tmp/abenvu.log
