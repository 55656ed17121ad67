#!/bin/bash
# same-box A/B of the pipelined run (k_compact of one VCF range beside k_classify of the next): QM_PIPE_CHUNKS = 1, 2, 4, 8
# usage: bash tools/ab_pipe.sh [n_vcf]
NV=${1:-1000}
for r in 1 2; do
for c in 1 2 4 8; do
  echo -n "chunks=$c: "; QM_PIPE_CHUNKS=$c python3 tools/run_once.py $NV 8 2>&1 | grep -v amdgpu.ids
done
done
