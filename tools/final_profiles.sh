#!/bin/bash
# every profile of a round in one GPU call: bash tools/final_profiles.sh <tag>   (outputs under gpurun_out/, copied to profiles/ by hand)
TAG=${1:-r02f}
ROOT=$PWD
export TMPDIR=/tmp
bash tools/prof_trace.sh $TAG --steps 5 --warmup 2 > gpurun_out/${TAG}_trace.log 2>&1
bash tools/prof_pmc.sh $TAG 1000 > gpurun_out/${TAG}_pmc.log 2>&1
bash tools/prof_trace.sh ${TAG}shuf --shuffled --vcfs 256 --steps 4 --warmup 1 > gpurun_out/${TAG}_trace_shuf.log 2>&1
mkdir -p gpurun_out/trace_${TAG}alle
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/trace_${TAG}alle -- python3 $ROOT/tools/run_once.py 256 6 0 30 > $ROOT/gpurun_out/${TAG}_trace_alle.log 2>&1)
bash tools/pmc_shuffled.sh $TAG "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "VALUBusy SALUBusy VALUUtilization LdsUtil MemUnitStalled" "FETCH_SIZE" "WRITE_SIZE" > gpurun_out/${TAG}_pmc_shuffled.json 2> gpurun_out/${TAG}_pmc_shuffled.err
python3 tools/e2e_files_bench.py 16 > gpurun_out/${TAG}_e2e.log 2>&1
python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -c 600 gpurun_out/${TAG}_bench.json
