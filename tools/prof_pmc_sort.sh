#!/bin/bash
# PMC passes for the radix-sort kernels (shuffled run).  usage: bash tools/prof_pmc_sort.sh <tag> [n_vcf]
set -e
TAG=${1:-x}; NV=${2:-64}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmcs_$TAG
mkdir -p $OUT
OLDPWD=$PWD
run() { local name=$1; shift
  (cd /tmp && rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $OLDPWD/tools/run_once.py $NV 2 1 > $OUT/$name.log 2>&1) || true; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE
python3 - <<PY
import csv,glob,collections,os
out="$OUT"
for name in ("sq1","sq2","tcc1","tcc2"):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for f in glob.glob(os.path.join(out,name,"**","*counter_collection.csv"),recursive=True):
        for row in csv.DictReader(open(f)):
            k=row.get("Kernel_Name","?").split("(")[0][-32:]
            agg[k][row["Counter_Name"]]+=float(row["Counter_Value"]); cnt[(k,row["Counter_Name"])]+=1
    for k,d in agg.items():
        if "sort" in k: print(name,k,{c:"%.4g"%(v/cnt[(k,c)]) for c,v in d.items()})
PY
