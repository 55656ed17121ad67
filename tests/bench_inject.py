"""Injected steps for the tests of bench.py's own launcher (QM_BENCH_INJECT=bench_inject:<fn>): what a rank does instead of
the GPU body.  No GPU, no torch."""
import os
import time


def step(args, rank, world):
    assert int(os.environ["LOCAL_RANK"]) == rank and os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
    return {"metric": "injected", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "rank": rank}


def fail_on_rank1(args, rank, world):
    if rank == 1:
        raise RuntimeError("rank 1 breaks on purpose")
    time.sleep(120)      # the others would wait in a collective: the launcher must stop them
    return {}


def hang(args, rank, world):
    time.sleep(120)
    return {}
