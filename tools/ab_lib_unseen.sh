#!/bin/bash
# same-box A/B of two LIBRARIES on the shuffled workload, first-seen (QM_MEMO=0) and with the batch's memory, interleaved:
#   ROUNDS=4 bash tools/ab_lib_unseen.sh <tag>=<path to libqmvt.so> ...
cd /tmp
: > /tmp/ablib.log
for rep in $(seq 1 ${ROUNDS:-4}); do for spec in "$@"; do
  TAG=${spec%%=*}; LIB=${spec#*=}
  for memo in 0 1; do
    echo -n "$TAG memo=$memo: " >> /tmp/ablib.log
    ( export QM_MEMO=$memo QM_LIBQMVT=$LIB QM_SKIP_BUILD_CHECK=1; python3 $GRAFT_REPO_ROOT/tools/run_once.py ${NV:-256} ${RUNS:-6} 1 ${PCT:-0} 2>&1 | grep -o "wall [0-9.]*" >> /tmp/ablib.log )
  done
done; done
python3 - <<'PY'
import re, collections, statistics
d = collections.defaultdict(list)
for line in open("/tmp/ablib.log"):
    m = re.match(r"(\S+ memo=\d): wall ([\d.]+)", line)
    if m: d[m.group(1)].append(float(m.group(2)))
for k, v in d.items():
    print("%-24s n=%d median %.3f ms  min %.3f  max %.3f" % (k, len(v), statistics.median(v), min(v), max(v)))
PY
