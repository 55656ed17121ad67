#!/bin/bash
for parts in 1 2 4 8; do for pad in 0 24576; do
  echo -n "parts=$parts pad=$pad: "; QM_BUCKET_PARTS=$parts QM_DJ_PAD=$pad timeout -k 10 120 python3 tools/join_ab.py 2>&1 | grep join= | cut -d: -f2
done; done
