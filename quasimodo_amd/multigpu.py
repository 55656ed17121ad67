"""The path on several GPUs of one node.

The caller x sample VCFs are independent units (the reference runs one process per VCF,
rules/extract_TP.smk:17-20), so they shard over ranks with no data-path collective: one process
per GPU, VCFs dealt by longest-processing-time-first on their size (`sharding.lpt_shards`), every
rank runs the ordinary batch path (`extract_many`) on its share and writes its own files.  The
only exchange is ONE all-reduce of the per-truth-set confusion counters [n_truth][3][n_bins]
(RCCL over xGMI: torch.distributed backend "nccl"; "gloo" in the CPU tests) plus a gather of the
per-VCF rows to rank 0 for the tables.

The parent never touches a GPU: it writes the job list to a spec file and starts one child per
rank (`python -m quasimodo_amd.multigpu <spec> <rank>`); children rendezvous on 127.0.0.1.
"""
import importlib
import os
import pickle
import socket
import subprocess
import sys
import tempfile

import numpy as np

from .sharding import allreduce_counters, lpt_shards


def truth_key(job):
    return (os.path.abspath(job.snp_file), job.mode)


def job_weights(jobs):
    """LPT weight of a VCF: its size in bytes (proportional to its records; known without scanning,
    and the same on every rank)."""
    return [os.path.getsize(j.vcf_file) for j in jobs]


def default_classify(jobs, device, n_bins=256, alleles=None, strict=None):
    """One rank's share through the ordinary batch path on GPU `device`."""
    from .engine import Engine
    from .extract import extract_many
    if not jobs:
        return []
    with Engine(device) as eng:
        extract_many(jobs, engine=eng, strict=strict, n_bins=n_bins, alleles=alleles)
    return [j.stats for j in jobs]


def _resolve(name):
    if not name:
        return default_classify
    mod, fn = name.split(":")
    return getattr(importlib.import_module(mod), fn)


def run_rank(jobs, rank, world, backend="nccl", classify=None, n_bins=256, alleles=None, strict=None, same_device=False):
    """What one rank does.  Returns, on rank 0, {"stats": [per job], "counters": int64 [n_truth][3][n_bins],
    "truth_keys": [...], "shards": [[job index, ...] per rank]}; None on the other ranks.
    torch.distributed must be initialised by the caller (world > 1)."""
    import torch
    import torch.distributed as dist
    from .extract import is_pure_strain
    classify = classify or default_classify
    shards = lpt_shards(job_weights(jobs), world)
    mine = shards[rank]
    device = 0 if same_device else rank
    local = classify([jobs[i] for i in mine], device, n_bins=n_bins, alleles=alleles, strict=strict)
    if len(local) != len(mine):
        raise RuntimeError("classify returned %d rows for %d jobs" % (len(local), len(mine)))
    # the confusion counters of every truth set, summed over this rank's VCFs, in ONE tensor
    keys = sorted({truth_key(j) for j in jobs if not is_pure_strain(j.vcf_file)})
    kidx = {k: i for i, k in enumerate(keys)}
    cnt = np.zeros((max(len(keys), 1), 3, n_bins), np.int64)
    for i, st in zip(mine, local):
        if st.get("roc") is not None:
            cnt[kidx[truth_key(jobs[i])]] += np.asarray(st["roc"]).astype(np.int64)
    t = torch.from_numpy(cnt)
    if backend == "nccl":
        t = t.to(torch.device("cuda", device))
    allreduce_counters(t)                      # the path's single collective
    rows = list(zip(mine, local))
    if world > 1:
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(rows, gathered, dst=0)
    else:
        gathered = [rows]
    if rank != 0:
        return None
    stats = [None] * len(jobs)
    for part in gathered:
        for i, st in part:
            stats[i] = st
    return {"stats": stats, "counters": t.cpu().numpy(), "truth_keys": keys, "shards": shards}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def extract_many_sharded(jobs, gpus, backend="nccl", classify=None, n_bins=256, alleles=None, strict=None, same_device=False,
                         timeout=None):
    """`extract_many` over `gpus` GPUs of this node: one child process per GPU, started before anything
    touches a GPU.  `classify` ("module:function", optional) replaces the per-rank batch path (tests inject
    a CPU stand-in; the product default is the HIP engine).  Returns (jobs with .stats filled, result dict of rank 0)."""
    from .extract import Job, _paths
    if gpus < 1:
        raise ValueError("gpus must be >= 1")
    for j in jobs:
        _paths(j)
    with tempfile.TemporaryDirectory(prefix="qmvt_mgpu_") as tmp:
        spec = {"jobs": [dict(vcf_file=j.vcf_file, snp_file=j.snp_file, mode=j.mode, outdir=j.outdir, caller=j.caller) for j in jobs],
                "world": gpus, "backend": backend, "classify": classify, "n_bins": n_bins, "alleles": alleles, "strict": strict,
                "same_device": same_device, "result": os.path.join(tmp, "result.pkl")}
        sp = os.path.join(tmp, "spec.pkl")
        with open(sp, "wb") as fh:
            pickle.dump(spec, fh)
        env = dict(os.environ)
        env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(gpus))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
        procs = []
        for r in range(gpus):
            e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
            with open(os.path.join(tmp, "rank%d.log" % r), "wb") as lf:
                procs.append(subprocess.Popen([sys.executable, "-m", "quasimodo_amd.multigpu", sp, str(r)], env=e,
                                              stdout=lf, stderr=subprocess.STDOUT))
        # a rank that dies leaves its peers waiting in the collective: watch all of them, stop the rest when one fails
        import time
        logs = [os.path.join(tmp, "rank%d.log" % r) for r in range(gpus)]
        t0, bad = time.monotonic(), []
        while True:
            codes = [p.poll() for p in procs]
            bad = [r for r, c in enumerate(codes) if c not in (None, 0)]
            if bad or all(c == 0 for c in codes):
                break
            if timeout is not None and time.monotonic() - t0 > timeout:
                bad = [r for r, c in enumerate(codes) if c is None]
                break
            time.sleep(0.05)
        for p in procs:
            if p.poll() is None:
                p.kill()
            p.wait()
        if bad:
            tails = []
            for r in bad:
                try:
                    with open(logs[r], "rb") as fh:
                        tails.append("--- rank %d ---\n%s" % (r, fh.read()[-3000:].decode("utf-8", "replace")))
                except OSError:
                    pass
            raise RuntimeError("rank(s) %s failed or timed out:\n%s" % (bad, "\n".join(tails)))
        with open(spec["result"], "rb") as fh:
            res = pickle.load(fh)
    for j, st in zip(jobs, res["stats"]):
        j.stats = st
        if st.get("pure_strain"):
            j.tp_out = ""
    return jobs, res


def _main(argv):
    sp, rank = argv[0], int(argv[1])
    with open(sp, "rb") as fh:
        spec = pickle.load(fh)
    import torch
    import torch.distributed as dist
    from .extract import Job
    world, backend = spec["world"], spec["backend"]
    jobs = [Job(**d) for d in spec["jobs"]]
    from .extract import _paths
    for j in jobs:
        _paths(j)
    if backend == "nccl":
        dev = torch.device("cuda", 0 if spec["same_device"] else rank)
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        res = run_rank(jobs, rank, world, backend=backend, classify=_resolve(spec["classify"]), n_bins=spec["n_bins"],
                       alleles=spec["alleles"], strict=spec["strict"], same_device=spec["same_device"])
        if rank == 0:
            with open(spec["result"] + ".tmp", "wb") as fh:
                pickle.dump(res, fh)
            os.replace(spec["result"] + ".tmp", spec["result"])
        dist.barrier()
    finally:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(_main(sys.argv[1:]))
