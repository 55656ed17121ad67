#!/bin/bash
# PMC passes for k_compact: bash tools/pmc_compact.sh <tag> [extra env]   (separate runs, counters only)
export TMPDIR=/tmp
ROOT=$PWD
TAG=${1:-x}
OUT=$ROOT/gpurun_out/pmc_compact_$TAG; mkdir -p $OUT
i=0
for PASS in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_sum TCC_WRITE_sum TCC_WRITEBACK_sum" "VALUBusy SALUBusy LdsUtil MemUnitStalled WriteUnitStalled" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $PASS --output-format csv -d $OUT/p$i -- python3 $ROOT/tools/run_once.py 1000 2 > $OUT/log$i.txt 2>&1) || true
done
python3 - $OUT <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_compact" in row["Kernel_Name"]:
            agg[row["Counter_Name"]] += float(row["Counter_Value"]); cnt[row["Counter_Name"]] += 1
print({k: float("%.5g" % (v / cnt[k])) for k, v in sorted(agg.items())})
PY
