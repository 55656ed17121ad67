#!/bin/bash
# a long randomised parity hunt over every variant of the unsorted path: bash tools/fuzz_campaign.sh [rounds per variant]
N=${1:-600}
OUT=gpurun_out/fuzz_campaign.log; : > $OUT
run() { echo "== $*" >> $OUT; env "$@" python3 tools/gpu_fuzz.py $N $SEED 2>&1 | tail -1 >> $OUT; }
SEED=101 run QM_X=0
SEED=102 run QM_BUCKET2=2
SEED=103 run QM_JOIN=hash
SEED=104 run QM_SORT_PATH=radix
SEED=105 run QM_BUCKET_EXT=0
SEED=106 run QM_MEMO=0
SEED=107 run QM_BUCKET_PARTS=4
SEED=108 run QM_PIPE_CHUNKS=3 QM_PIPE_MIN_SPANS=1
SEED=109 run QM_BUCKETX=2
# round 5: round 3's join and the hashed one behind the new scatter (zero entries, the scatter's histogram off for them), the join without the look at the highest bucket
SEED=110 run QM_JOIN=direct
SEED=111 run QM_JOIN=direct QM_BUCKET2=2
SEED=112 run QM_BUCKET2=2 QM_MEMO=0
SEED=113 run QM_NO_TIGHT_NBK=1
SEED=115 run QM_TIGHT_NBK_ALL=1 QM_BUCKET2=2
SEED=116 run QM_TIGHT_NBK_ALL=1 QM_BUCKETX=2
# round 5, second half: finish without its round trips (flags event, queued chunk tails, lazy k_finalize) against the old waits; several chunks per finish
SEED=117 run QM_SPECULATE=0
SEED=118 run QM_FLAGS_WAIT=stream QM_NO_LAZY_FINALIZE=1
SEED=119 run QM_SORT_CHUNK_RECORDS=60000
SEED=120 run QM_SORT_CHUNK_RECORDS=60000 QM_MEMO=0
SEED=121 run QM_NO_MIRRORS=1
SEED=114 run QM_COL_SLAB=1280                                   # the columns as pieces of one allocation
cat $OUT
