#!/bin/bash
# A/B of a kernels build: the same shuffled batch through the default library and through tools/probe/ab/libqmvt_<tag>.so
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
cd $ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for tag in "" $*; do
  if [ -n "$tag" ]; then export QM_LIBQMVT=$ROOT/tools/probe/ab/libqmvt_$tag.so; fi
  echo "== ${tag:-default}"
  python3 tools/join_ab.py 16 10000000 50000000 1000000 || exit 1
  python3 tools/join_ab.py 256 1000000 5000000 100000 || exit 1
  (cd /tmp && timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_${tag:-default} -- python3 $ROOT/tools/join_ab.py 16 10000000 50000000 1000000 > /dev/null 2>&1; f=$(find /tmp/ab_${tag:-default} -name "*kernel_stats.csv" | head -1); cut -d, -f1-4 $f | head -6)
done
