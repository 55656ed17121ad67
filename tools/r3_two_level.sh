#!/bin/bash
# two-level bucket path: parity (fuzz with every unsorted VCF forced through it, the 10 M-record tests), then timing on configs[3]'s shape
QM_BUCKET2=2 timeout -k 10 300 python3 tools/gpu_fuzz.py 150 777 2>&1 | tail -2
python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ten_million or synthetic_batch or maximum_sizes" 2>&1 | tail -3
for i in 1 2; do timeout -k 10 300 python3 tools/join_ab.py 16 10000000 50000000 1000000 2>&1 | grep join=; done
QM_BUCKET2=0 timeout -k 10 300 python3 tools/join_ab.py 16 10000000 50000000 1000000 2>&1 | grep join=
