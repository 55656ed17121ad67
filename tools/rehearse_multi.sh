#!/bin/bash
# 2-rank rehearsal of bench.py's N > 1 path on a ONE-GPU box (both ranks on device 0, gloo for the collective since RCCL refuses two
# ranks on one GPU), for the three presets at reduced VCF counts; every rank checks its first / last VCF against the oracle.
export QM_BENCH_SAME_DEVICE=1 QM_BENCH_BACKEND=gloo
for spec in "2 60" "3 12" "4 30"; do
  set -- $spec
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29500 + $1)) \
    bench.py --gpus 2 --config $1 --vcfs $2 --steps 3 --warmup 1 2>/dev/null | tail -1
done
