#!/bin/bash
# same-box A/B of ENVIRONMENT variants (no rebuild) on the sorted main workload, interleaved ROUNDS times, medians reported:
#   ROUNDS=8 bash tools/ab_env.sh "<tag>=<VAR=value VAR2=value ...>" ...
cd /tmp
: > /tmp/abenv.log
for rep in $(seq 1 ${ROUNDS:-8}); do for spec in "$@"; do
  TAG=${spec%%=*}; ENVS=${spec#*=}
  echo -n "$TAG: " >> /tmp/abenv.log
  ( export $ENVS; python3 $GRAFT_REPO_ROOT/tools/run_once.py ${NV:-1000} ${RUNS:-8} ${RARGS:-} 2>&1 | grep classify >> /tmp/abenv.log )
done; done
cat /tmp/abenv.log
python3 - <<'PY'
import re, collections, statistics
d = collections.defaultdict(list)
for line in open("/tmp/abenv.log"):
    m = re.match(r"(\S+): .*classify ([\d.]+) ms .*finalize ([\d.]+) compact ([\d.]+) total ([\d.]+) wall ([\d.]+)", line)
    if m: d[m.group(1)].append(tuple(float(x) for x in m.groups()[1:]))
for tag, v in d.items():
    med = [statistics.median(x[i] for x in v) for i in range(5)]
    mn = [min(x[i] for x in v) for i in range(5)]
    mx = [max(x[i] for x in v) for i in range(5)]
    print("%-12s n=%d  median: classify %.3f compact %.3f wall %.3f   min: compact %.3f wall %.3f  max: compact %.3f wall %.3f" %
          (tag, len(v), med[0], med[2], med[4], mn[2], mn[4], mx[2], mx[4]))
PY
