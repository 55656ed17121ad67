#!/bin/bash
# same-box, alternating A/B of qm_batch_run with k_finalize split over two streams (QM_FINALIZE_SPLIT=1, the default) and in one piece
for i in 1 2 3 4; do for m in 1 0; do
  echo -n "split $m: "; QM_FINALIZE_SPLIT=$m python3 bench.py --cpu-sample 0 --shell-sample 0 --alleles-vcfs 0 --shuffled-vcfs 0 --steps 40 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4))"
done; done
