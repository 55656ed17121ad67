"""N > 1 path on CPU: world_size-2 gloo.  VCFs shard over ranks (LPT), each rank's confusion
counters are summed with the path's single all-reduce, and the result equals the single-process
total.  The per-rank classification here is the ORACLE (there is no GPU in this test); what is
under test is the sharding and the collective bench.py / a multi-GPU host use."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from conftest import random_columns, random_truth
    from oracle import qm_oracle as O
    from quasimodo_amd.sharding import allreduce_counters, lpt_shards
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(2024)                       # same inputs on every rank
    truths = [random_truth(rng, 500, 20000), random_truth(rng, 200, 20000)]
    sizes = [3000, 10, 1200, 800, 2500, 1, 900]
    cols = [random_columns(rng, n, 20000, truths[i % 2]) for i, n in enumerate(sizes)]
    shards = lpt_shards(sizes, world)
    local = np.zeros((2, 3, 256), np.int64)
    for v in shards[rank]:
        _, roc, _ = O.classify_columns(*cols[v], *truths[v % 2])
        local[v % 2] += roc.astype(np.int64)
    t = torch.from_numpy(local.copy())
    allreduce_counters(t)
    total = np.zeros((2, 3, 256), np.int64)
    for v in range(len(sizes)):
        _, roc, _ = O.classify_columns(*cols[v], *truths[v % 2])
        total[v % 2] += roc.astype(np.int64)
    q.put((rank, bool(np.array_equal(t.numpy(), total)), [len(s) for s in shards]))
    dist.barrier()
    dist.destroy_process_group()


def test_vcf_sharding_allreduce_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert sum(res[0][2]) == 7
