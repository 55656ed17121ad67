#!/bin/bash
# PMC passes for k_classify (separate from any trace run, as the pool requires).
# usage: bash tools/prof_pmc.sh <tag> [n_vcf]      RARGS="0 30": run_once.py's further arguments (shuffled, indel_pct) -- the allele-extended instantiation
set -e
TAG=${1:-x}; NV=${2:-256}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
run() { # name counters...
  local name=$1; shift
  (cd /tmp && rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $OLDPWD/tools/run_once.py $NV 2 ${RARGS:-} > $OUT/$name.log 2>&1) || true
}
OLDPWD=$PWD
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE
run grbm GRBM_GUI_ACTIVE
python3 - <<PY
import csv,glob,collections,os
out="$OUT"
for name in ("sq1","sq2","tcc1","tcc2","grbm"):
    files=glob.glob(os.path.join(out,name,"**","*counter_collection.csv"),recursive=True)
    if not files: print(name,"no csv"); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for f in files:
        for row in csv.DictReader(open(f)):
            k=row.get("Kernel_Name","?").split("(")[0][-40:]
            agg[k][row["Counter_Name"]]+=float(row["Counter_Value"]); 
            cnt[(k,row["Counter_Name"])]+=1
    for k,d in agg.items():
        if "classify" in k or "compact" in k or "finalize" in k:
            print(name,k,{c:"%.4g"%(v/cnt[(k,c)]) for c,v in d.items()},"dispatches",max(cnt[(k,c)] for c in d))
PY
