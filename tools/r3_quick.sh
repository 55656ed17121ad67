#!/bin/bash
# quick parity (fuzz) + timing of the shuffled path + phase clocks
timeout -k 10 300 python3 tools/gpu_fuzz.py ${FUZZ:-60} 4242 2>&1 | tail -1
for i in 1 2; do timeout -k 10 120 python3 tools/join_ab.py 2>&1 | grep join=; done
QM_BUCKET_PARTS=1 timeout -k 10 120 python3 tools/join_ab.py 2>&1 | grep join=
bash tools/prof_join.sh 2>&1 | grep "dj profile" | tail -1
