#!/bin/bash
# a long randomised parity hunt over every variant of the unsorted path: bash tools/fuzz_campaign.sh [rounds per variant]
N=${1:-600}
mkdir -p gpurun_out
OUT=gpurun_out/fuzz_campaign.log; : > $OUT
run() { echo "== $*" >> $OUT; env "$@" python3 tools/gpu_fuzz.py $N $SEED 2>&1 | tail -1 >> $OUT; }
SEED=101 run QM_X=0
SEED=102 run QM_BUCKET2=2
SEED=103 run QM_JOIN=hash
SEED=104 run QM_SORT_PATH=radix
SEED=105 run QM_BUCKET_EXT=0
SEED=106 run QM_MEMO=0
SEED=108 run QM_PIPE_CHUNKS=3
SEED=109 run QM_BUCKETX=2
SEED=110 run QM_BUCKETX=3
SEED=112 run QM_BUCKET2=2 QM_MEMO=0
# finish without its round trips (flags event, queued chunk tails) against the old waits; several chunks per finish
SEED=117 run QM_SPECULATE=0
SEED=118 run QM_FLAGS_WAIT=stream
SEED=119 run QM_SORT_CHUNK_RECORDS=60000
SEED=120 run QM_SORT_CHUNK_RECORDS=60000 QM_MEMO=0
SEED=121 run QM_NO_MIRRORS=1
# the sizes where the paths hand over to one another, with the library's own routing (VCFs of 20 000 ... 4 M records on 3 ... 120 Mb)
echo "== tools/gpu_fuzz_big.py" >> $OUT; python3 tools/gpu_fuzz_big.py 80 301 2>&1 | tail -1 >> $OUT
cat $OUT
