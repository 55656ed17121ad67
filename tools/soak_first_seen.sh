mkdir -p gpurun_out
OUT=gpurun_out/soak_first_seen.log; : > $OUT
run() { echo "== $*" >> $OUT; env "$@" python3 tools/gpu_fuzz.py $N $SEED 2>&1 | tail -1 >> $OUT; }
N=1200
SEED=201 run QM_X=0
SEED=202 run QM_MEMO=0
SEED=203 run QM_SORT_CHUNK_RECORDS=60000
SEED=204 run QM_SORT_CHUNK_RECORDS=60000 QM_MEMO=0
SEED=205 run QM_SORT_CHUNK_RECORDS=90000 QM_BUCKET2=2
SEED=206 run QM_BUCKETX=2 QM_MEMO=0
SEED=207 run QM_PIPE_CHUNKS=3 QM_MEMO=0
SEED=208 run QM_SORT_CHUNK_RECORDS=60000 QM_JOIN=hash
cat $OUT
