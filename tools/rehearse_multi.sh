#!/bin/bash
# 2-rank rehearsal of bench.py's N > 1 path on a ONE-GPU box, as typed (`python3 bench.py --gpus 2`: the bench starts its own
# ranks); both ranks on device 0, gloo for the collective since RCCL refuses two ranks on one GPU; the three presets at
# reduced VCF counts; every rank checks its first / last VCF against the oracle.  Then once under torch.distributed.run.
export QM_BENCH_SAME_DEVICE=1 QM_BENCH_BACKEND=gloo
for spec in "2 60" "3 12" "4 30"; do
  set -- $spec
  python3 bench.py --gpus 2 --config $1 --vcfs $2 --steps 3 --warmup 1 2>/dev/null | tail -1
done
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
  bench.py --gpus 2 --config 2 --vcfs 60 --steps 3 --warmup 1 2>/dev/null | tail -1
# and the collective over RCCL itself at the only world size one card allows (QM_BENCH_FORCE_PG=1: process group "nccl" on the
# device, the all-reduce of the engine's buffer inside every step)
unset QM_BENCH_SAME_DEVICE QM_BENCH_BACKEND
QM_BENCH_FORCE_PG=1 python3 bench.py --steps 10 --warmup 3 --cpu-sample 0 --shell-sample 0 --shuffled-vcfs 0 --shuffled3-vcfs 0 --shuffled-alleles-vcfs 0 --shuffled4-vcfs 0 --alleles-vcfs 0 --alloc-reps 0 2>/dev/null | tail -1
