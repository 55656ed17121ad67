#!/bin/bash
# time k_classify with phases switched off (QM_ABLATE bits: 1 join, 2 histogram, 4 masks, 8 R-path dedupe)
mkdir -p gpurun_out
for ab in ${ABLATES:-0 1 2 4 3 7 15}; do
  echo -n "ablate=$ab: "; QM_ABLATE=$ab python3 tools/run_once.py ${NV:-256} 6 2>&1 | grep -v amdgpu.ids
done
