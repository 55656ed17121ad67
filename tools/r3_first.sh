#!/bin/bash
# round 3, first GPU visit: parity of the new join (tests + fuzz), then same-box A/B against the hashed join
set -o pipefail
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "unsorted or shuffled or synthetic or small or stale or fuzz" > gpurun_out/r3_tests.log 2>&1; echo "pytest rc $?" | tee -a gpurun_out/r3_tests.log
tail -3 gpurun_out/r3_tests.log
timeout -k 10 300 python3 tools/gpu_fuzz.py 150 31337 > gpurun_out/r3_fuzz.log 2>&1; echo "fuzz rc $?"; tail -2 gpurun_out/r3_fuzz.log
for i in 1 2; do
  QM_JOIN=hash timeout -k 10 120 python3 tools/join_ab.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r3_join_ab.log
  timeout -k 10 120 python3 tools/join_ab.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r3_join_ab.log
done
