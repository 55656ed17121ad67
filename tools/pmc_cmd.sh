#!/bin/bash
# PMC passes of an arbitrary command (one rocprofv3 run per pass; nothing but --pmc in a run, as the pool requires), summed per kernel:
#   bash tools/pmc_cmd.sh <tag> "<counters pass 1>" ["<counters pass 2>" ...] -- python3 tools/join_ab.py 16 10000000 50000000 1000000
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
PASSES=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do PASSES+=("$1"); shift; done
shift
case "$(basename -- "$1")" in env|bash|sh|taskset|numactl) echo "$0: give the program itself after --" >&2; exit 2;; esac
OUT=$ROOT/gpurun_out/pmc_cmd_$TAG; mkdir -p $OUT
i=0
for PASS in "${PASSES[@]}"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $PASS --output-format csv -d $OUT/p$i -- "$@" > $OUT/log$i.txt 2>&1) || true
done
python3 - $OUT <<'PY'
import csv, glob, collections, sys, json
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("qm::") and "synth" not in k and "bw_probe" not in k:
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
out = {k: dict({c: v / cnt[k][c] for c, v in sorted(d.items())}, dispatches=max(cnt[k].values())) for k, d in sorted(agg.items())}
print(json.dumps(out, indent=1))
PY
