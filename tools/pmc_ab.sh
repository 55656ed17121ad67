#!/bin/bash
# dynamic instruction counts of two kernel builds (gpurun_out/ab/<tag>/libqmvt.so from tools/ab_r2.sh): bash tools/pmc_ab.sh tagA tagB
export TMPDIR=/tmp
ROOT=$PWD
for TAG in "$@"; do
  OUT=$ROOT/gpurun_out/pmc_ab_$TAG; mkdir -p $OUT
  (cd /tmp && QM_LIBQMVT=$ROOT/gpurun_out/ab/$TAG/libqmvt.so rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY --output-format csv -d $OUT -- python3 $ROOT/tools/run_once.py 1000 2 > $OUT/log.txt 2>&1) || true
  python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(float); cnt=collections.Counter()
for f in glob.glob("$OUT/**/*counter_collection.csv",recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_classify" in row["Kernel_Name"]:
            agg[row["Counter_Name"]]+=float(row["Counter_Value"]); cnt[row["Counter_Name"]]+=1
print("$TAG", {k:"%.4g"%(v/cnt[k]) for k,v in sorted(agg.items())})
PY
done
