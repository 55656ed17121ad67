"""Multi-GPU partitioning of the path: the caller x sample VCFs are
independent units, so they shard over ranks with no data-path collective; the only exchange is
one all-reduce of the per-truth-set confusion counters.  One process per GPU (torch.distributed,
backend "nccl" = RCCL over xGMI on MI355X, "gloo" in the CPU tests)."""
import numpy as np


def lpt_shards(n_records, world):
    """Longest-processing-time assignment of VCFs to ranks by record count.
    Returns a list (per rank) of VCF indices, each in ascending order."""
    n_records = np.asarray(n_records, np.int64)
    order = np.argsort(-n_records, kind="stable")
    load = np.zeros(world, np.int64)
    shards = [[] for _ in range(world)]
    for v in order:
        r = int(np.argmin(load))
        shards[r].append(int(v))
        load[r] += n_records[v]
    return [sorted(s) for s in shards]


def allreduce_counters(counters, group=None):
    """In-place sum over ranks of the [n_truth][3][n_bins] counter tensor (int64).  This is the
    path's single collective.  It runs whenever a process group exists, a group of one rank included (the sum
    of one term: the same call on the same fabric library, which is what a one-GPU box can test of it)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(counters, op=dist.ReduceOp.SUM, group=group)
    return counters
