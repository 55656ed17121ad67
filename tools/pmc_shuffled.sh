#!/bin/bash
# PMC counters of the bucket-path kernels on the shuffled workload: bash tools/pmc_shuffled.sh <tag> "<counters pass 1>" ["<counters pass 2>" ...]
# (one rocprofv3 run per pass; nothing but --pmc in a run, as the pool requires)   PCT=30: allele-extended VCFs (two entry streams)
export TMPDIR=/tmp
ROOT=$PWD
TAG=$1; shift
OUT=$ROOT/gpurun_out/pmc_shuf_$TAG${PCT:+_pct$PCT}; mkdir -p $OUT
i=0
for PASS in "$@"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $PASS --output-format csv -d $OUT/p$i -- python3 $ROOT/tools/run_once.py ${NV:-256} 2 1 ${PCT:-0} > $OUT/log$i.txt 2>&1) || true
done
python3 - $OUT <<'PY'
import csv, glob, collections, sys, json
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if any(x in k for x in ("hash", "bucket", "sort_", "tile_counts", "join_direct", "join_lean", "join_ext", "k_part", "k_compact")):
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
out = {k: {c: v / cnt[k][c] for c, v in sorted(d.items())} for k, d in agg.items()}
print(json.dumps(out, indent=1))
PY
