#!/bin/bash
# derived PMC metrics of k_classify, default and allele-extended, one rocprofv3 run per pass: bash tools/pmc_classify.sh <tag>
export TMPDIR=/tmp
ROOT=$PWD
TAG=${1:-x}
for MODE in "" "0 30"; do
  NAME=$([ -z "$MODE" ] && echo default || echo alleles)
  OUT=$ROOT/gpurun_out/pmc_cls_${TAG}_$NAME; mkdir -p $OUT
  i=0
  for PASS in "VALUBusy SALUBusy VALUUtilization LdsUtil MemUnitStalled" "MemUnitBusy LdsLatency LDSBankConflict" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
    i=$((i+1))
    (cd /tmp && rocprofv3 --pmc $PASS --output-format csv -d $OUT/p$i -- python3 $ROOT/tools/run_once.py 1000 2 $MODE > $OUT/log$i.txt 2>&1) || true
  done
  python3 - $OUT $NAME <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_classify<" in row["Kernel_Name"]:
            agg[row["Counter_Name"]] += float(row["Counter_Value"]); cnt[row["Counter_Name"]] += 1
print(sys.argv[2], {k: float("%.4g" % (v / cnt[k])) for k, v in sorted(agg.items())})
PY
done
