#!/bin/bash
# every profile of a round in two GPU calls (20 minutes each at most): bash tools/final_profiles.sh <tag> 1 ; ... <tag> 2
# (outputs under gpurun_out/, copied to profiles/ by tools/copy_profiles.sh)
TAG=${1:-r06f}; PART=${2:-1}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
R=${TAG%%f*}
if [ "$PART" = "1" ]; then
bash tools/prof_trace.sh $TAG --steps 5 --warmup 2 > gpurun_out/${TAG}_trace.log 2>&1
bash tools/prof_pmc.sh $TAG 1000 > gpurun_out/${TAG}_pmc.log 2>&1
python3 tools/pmc_to_profiles.py gpurun_out/pmc_$TAG 1000 $R > gpurun_out/${TAG}_traffic.log 2>&1   # profiles/<round>_pmc_per_launch.json, profiles/traffic.json (with the device code's id)
RARGS="0 30" bash tools/prof_pmc.sh ${TAG}x 1000 > gpurun_out/${TAG}_pmc_alleles.log 2>&1
python3 tools/pmc_to_profiles.py gpurun_out/pmc_${TAG}x 1000 $R alleles >> gpurun_out/${TAG}_traffic.log 2>&1
cp profiles/traffic.json profiles/${R}_pmc_per_launch.json profiles/${R}_pmc_per_launch_alleles.json gpurun_out/ 2>/dev/null
bash tools/prof_trace.sh ${TAG}shuf --shuffled --vcfs 256 --steps 4 --warmup 1 > gpurun_out/${TAG}_trace_shuf.log 2>&1
bash tools/prof_trace.sh ${TAG}c3 --config 3 --steps 5 --warmup 2 > gpurun_out/${TAG}_trace_config3.log 2>&1
bash tools/prof_trace.sh ${TAG}c4 --config 4 --steps 5 --warmup 2 > gpurun_out/${TAG}_trace_config4.log 2>&1
mkdir -p gpurun_out/trace_${TAG}shuf3 gpurun_out/trace_${TAG}alle gpurun_out/trace_${TAG}mc
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/trace_${TAG}shuf3 -- python3 $ROOT/tools/join_ab.py 16 10000000 50000000 1000000 > $ROOT/gpurun_out/${TAG}_trace_shuf3.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/trace_${TAG}alle -- python3 $ROOT/tools/run_once.py 256 6 0 30 > $ROOT/gpurun_out/${TAG}_trace_alle.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/trace_${TAG}mc -- python3 $ROOT/tools/run_once.py 256 6 24 > $ROOT/gpurun_out/${TAG}_trace_mc.log 2>&1)
echo part 1 done
exit 0
fi
bash tools/pmc_shuffled.sh $TAG "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" > gpurun_out/${TAG}_pmc_shuffled.json 2> gpurun_out/${TAG}_pmc_shuffled.err
PCT=30 bash tools/pmc_shuffled.sh $TAG "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "FETCH_SIZE" "WRITE_SIZE" > gpurun_out/${TAG}_pmc_shuffled_alleles.json 2> gpurun_out/${TAG}_pmc_shuffled_alleles.err
bash tools/pmc_cmd.sh ${TAG}shuf3 "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE" -- python3 $ROOT/tools/join_ab.py 16 10000000 50000000 1000000 > gpurun_out/${TAG}_pmc_shuffled_config3.json 2> gpurun_out/${TAG}_pmc_shuffled_config3.err
mkdir -p gpurun_out/trace_${TAG}shufx gpurun_out/trace_${TAG}shufx4
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/trace_${TAG}shufx -- python3 $ROOT/tools/shuffled_ext.py 256 > $ROOT/gpurun_out/${TAG}_trace_shufx.log 2>&1)
python3 tools/shuffled_ext.py 256 > gpurun_out/${TAG}_shuffled_ext.log 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/trace_${TAG}shufx4 -- python3 $ROOT/tools/shuffled_ext.py 64 2000000 10000000 200000 > $ROOT/gpurun_out/${TAG}_trace_shufx4.log 2>&1)
python3 tools/shuffled_ext.py 64 2000000 10000000 200000 >> gpurun_out/${TAG}_shuffled_ext.log 2>&1
python3 tools/e2e_files_bench.py 16 > gpurun_out/${TAG}_e2e.log 2>&1
bash tools/pmc_compact.sh $TAG > gpurun_out/${TAG}_pmc_compact.log 2>&1
python3 bench.py --detail gpurun_out/${TAG}_bench_detail.json > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python3 bench.py --config 3 --detail gpurun_out/${TAG}_bench_config3_detail.json > gpurun_out/${TAG}_bench_config3.json 2> gpurun_out/${TAG}_bench_config3.err
python3 bench.py --config 4 --detail gpurun_out/${TAG}_bench_config4_detail.json > gpurun_out/${TAG}_bench_config4.json 2> gpurun_out/${TAG}_bench_config4.err
cat gpurun_out/${TAG}_bench.json gpurun_out/${TAG}_bench_config3.json gpurun_out/${TAG}_bench_config4.json
