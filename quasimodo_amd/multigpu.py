"""The path on several GPUs of one node.

The caller x sample VCFs are independent units (the reference runs one process per VCF,
rules/extract_TP.smk:17-20), so they shard over ranks with no data-path collective: one process
per GPU, VCFs dealt by longest-processing-time-first on their size (`sharding.lpt_shards`) -- whole
GROUPS of VCFs where something needs them together: the FP overlap of rules/compare_FP.smk:5-8 compares
the callers of ONE sample, so the workflows deal by sample and every rank computes the overlap of its own
samples with no exchange -- every rank runs the ordinary batch path (`extract_many` -> qm_extract_files_ex)
on its share and writes its own files.  The only exchange between the ranks is ONE all-reduce of the per-truth-file
confusion counters [n_truth][3][n_bins], on the device buffer the engine filled (RCCL over xGMI: torch.distributed
backend "nccl"; "gloo" in the CPU tests) -- nothing else: no gather, no barrier.  The per-VCF rows never travel between
ranks: every rank leaves them in a result file of its own, the way rank 0 always handed its result to the parent.

The parent never touches a GPU: it writes the job list to a spec file, starts one child per rank
(`python -m quasimodo_amd.multigpu <spec>`, quasimodo_amd.launch.spawn_ranks; children rendezvous on 127.0.0.1) and, when
all of them have left with status 0, puts the ranks' rows together.
"""
import datetime
import importlib
import os
import pickle
import sys
import tempfile

import numpy as np

from .launch import DEFAULT_TIMEOUT, RankFailure, spawn_ranks
from .sharding import allreduce_counters, lpt_shards


def truth_key(job):
    return (os.path.abspath(job.snp_file), job.mode)


def job_weights(jobs):
    """LPT weight of a VCF: its size in bytes (proportional to its records; known without scanning,
    and the same on every rank)."""
    return [os.path.getsize(j.vcf_file) for j in jobs]


def plan_shards(jobs, world, groups=None):
    """Which job goes to which rank: LPT on the file sizes, over whole groups when `groups` (a list of lists of job
    indices that must share a rank; jobs in no group are groups of their own) is given.  Returns a list (per rank) of
    job indices in ascending order -- the same on every rank."""
    w = job_weights(jobs)
    if not groups:
        return lpt_shards(w, world)
    seen = set()
    units = []
    for g in groups:
        g = sorted(set(int(i) for i in g))
        if seen & set(g):
            raise ValueError("a job is in two groups")
        seen |= set(g)
        if g:
            units.append(g)
    units += [[i] for i in range(len(jobs)) if i not in seen]
    units.sort(key=lambda g: g[0])
    shards_u = lpt_shards([sum(w[i] for i in g) for g in units], world)
    return [sorted(i for u in su for i in units[u]) for su in shards_u]


def truth_layout(jobs):
    """The rows of the all-reduced counters: one per distinct truth file of the WHOLE run, in sorted order -- every rank
    derives the same layout from the same job list.  Returns (keys, slot per job; -1 for pure-strain samples)."""
    from .extract import is_pure_strain
    keys = sorted({truth_key(j) for j in jobs if not is_pure_strain(j.vcf_file)})
    kidx = {k: i for i, k in enumerate(keys)}
    return keys, [-1 if is_pure_strain(j.vcf_file) else kidx[truth_key(j)] for j in jobs]


def default_body(jobs, indices, device, opts):
    """One rank's share through the ordinary batch path on GPU `device` (the product's per-rank body).
    opts: n_bins, alleles, strict, slots (row of every job of `jobs` in the counters), n_slots, backend, post ("module:function",
    optional: post(engine, jobs, indices, post_args) runs while the engine is alive and returns something picklable for rank 0),
    post_args.  Returns {"stats": [...], "counters": device tensor [n_slots][3][n_bins] int64 filled by the engine, "extra": ...}."""
    import torch
    from .engine import Engine
    from .extract import extract_many, is_pure_strain
    n_bins, n_slots = opts["n_bins"], max(opts["n_slots"], 1)
    on_gpu = opts.get("backend", "nccl") == "nccl" or opts.get("device_counters", True)
    dev = torch.device("cuda", device)
    counters = torch.zeros((n_slots, 3, n_bins), dtype=torch.int64, device=dev) if on_gpu else None
    extra = None
    need_engine = bool(jobs) and (not all(is_pure_strain(j.vcf_file) for j in jobs) or opts.get("post"))
    eng = Engine(device) if need_engine else None
    try:
        if jobs:
            extract_many(jobs, engine=eng, strict=opts.get("strict"), n_bins=n_bins, alleles=opts.get("alleles"),
                         truth_slots=opts["slots"] if counters is not None else None, n_slots=n_slots,
                         global_dev=counters.data_ptr() if counters is not None and eng is not None else None)
        if opts.get("post"):
            extra = _resolve(opts["post"])(eng, jobs, indices, opts.get("post_args"))
    finally:
        if eng is not None:
            eng.close()
    if counters is not None:
        torch.cuda.synchronize(dev)
    return {"stats": [j.stats for j in jobs], "counters": counters, "extra": extra, "paths": extract_many.last_paths if jobs else None}


def _resolve(name):
    if not name:
        return default_body
    if callable(name):
        return name
    mod, fn = name.split(":")
    return getattr(importlib.import_module(mod), fn)


def run_rank(jobs, rank, world, backend="nccl", body=None, n_bins=256, alleles=None, strict=None, same_device=False,
             groups=None, post=None, post_args=None):
    """What one rank does: its share of the jobs, then the all-reduce.  Returns {"rows": [(job index, stats), ...] of ITS jobs,
    "extra": the post hook's result} and, on rank 0, also "counters" (int64 [n_truth][3][n_bins] summed over ALL ranks),
    "truth_keys" and "shards" ([[job index, ...] per rank]); `merge_ranks` puts the ranks' values together.
    torch.distributed must be initialised by the caller: the all-reduce runs whenever a process group exists."""
    import torch
    body = _resolve(body)
    shards = plan_shards(jobs, world, groups)
    mine = shards[rank]
    device = 0 if same_device else rank
    keys, slot = truth_layout(jobs)
    opts = dict(n_bins=n_bins, alleles=alleles, strict=strict, slots=[slot[i] for i in mine], n_slots=len(keys), backend=backend,
                post=post, post_args=post_args, rank=rank, world=world)
    res = body([jobs[i] for i in mine], list(mine), device, opts)
    local = res["stats"]
    if len(local) != len(mine):
        raise RuntimeError("the rank body returned %d rows for %d jobs" % (len(local), len(mine)))
    t = res.get("counters")
    if t is None:
        # a body without device counters (the CPU stand-in of the tests): the rows summed on the host
        cnt = np.zeros((max(len(keys), 1), 3, n_bins), np.int64)
        for i, st in zip(mine, local):
            if st.get("roc") is not None:
                cnt[slot[i]] += np.asarray(st["roc"]).astype(np.int64)
        t = torch.from_numpy(cnt)
        if backend == "nccl":
            t = t.to(torch.device("cuda", device))
    elif backend != "nccl" and t.is_cuda:
        t = t.cpu()                            # rehearsals on one card: gloo sums host tensors
    ops0 = _collectives_so_far()
    allreduce_counters(t)                      # the path's single collective
    if t.is_cuda:
        # RCCL only ENQUEUES the sum on the device.  Every rank waits for its own copy here -- a local wait, not a collective --
        # so that no rank tears its communicator down (destroy_process_group at exit) while a peer's sum is still in flight.
        torch.cuda.synchronize(t.device)
    out = {"rows": list(zip(mine, local)), "extra": res.get("extra"), "collective": _collective_record(ops0), "paths": res.get("paths")}
    if rank == 0:
        out.update(counters=t.cpu().numpy(), truth_keys=keys, shards=shards)
    return out


def _collectives_so_far():
    """How many collectives the default process group has issued (its sequence number), or None without a group / the counter."""
    try:
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return None
        return int(dist.distributed_c10d._get_default_group()._get_sequence_number_for_group())
    except Exception:
        return None


def _collective_record(ops_before):
    """What the run's one exchange was: backend, world size and how far the process group's collective counter moved over it
    (1: the all-reduce and nothing else) -- the evidence a one-GPU test has that the sum really went through RCCL."""
    try:
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return None
        now = _collectives_so_far()
        return {"backend": str(dist.get_backend()), "world": int(dist.get_world_size()), "ops_before": ops_before, "ops_after": now}
    except Exception:
        return None


def merge_ranks(n_jobs, parts):
    """The ranks' results (run_rank's return values, rank order) as ONE: {"stats": [per job], "counters", "truth_keys",
    "shards", "extras": [per rank]}."""
    stats = [None] * n_jobs
    for part in parts:
        for i, st in part["rows"]:
            stats[i] = st
    missing = [i for i, st in enumerate(stats) if st is None]
    if missing:
        raise RuntimeError("no rank returned a row for job(s) %s" % missing[:8])
    r0 = parts[0]
    paths = None
    for part in parts:   # where the VCFs found out of order went, summed over the ranks
        if part.get("paths"):
            paths = {k: (paths or {}).get(k, 0) + v for k, v in part["paths"].items()}
    return {"paths": paths, "stats": stats, "counters": r0["counters"], "truth_keys": r0["truth_keys"], "shards": r0["shards"],
            "extras": [part.get("extra") for part in parts], "collectives": [part.get("collective") for part in parts]}


def extract_many_sharded(jobs, gpus, backend="nccl", body=None, n_bins=256, alleles=None, strict=None, same_device=False,
                         timeout=None, groups=None, post=None, post_args=None, classify=None):
    """`extract_many` over `gpus` GPUs of this node: one child process per GPU, started before anything
    touches a GPU.  `body` ("module:function", optional) replaces the per-rank body (tests inject a CPU stand-in; the
    product default is the HIP engine, `default_body`).  groups / post / post_args: see plan_shards / default_body.
    timeout: seconds for the whole run (default QM_RANK_TIMEOUT, 3600): a rendezvous or a collective that hangs with every
    rank alive must not hang the caller.  Returns (jobs with .stats filled, result dict of rank 0)."""
    from .extract import _paths
    if gpus < 1:
        raise ValueError("gpus must be >= 1")
    body = body or classify   # (the old name of the hook)
    for j in jobs:
        _paths(j)
    timeout = DEFAULT_TIMEOUT if timeout is None else timeout
    with tempfile.TemporaryDirectory(prefix="qmvt_mgpu_") as tmp:
        spec = {"jobs": [dict(vcf_file=j.vcf_file, snp_file=j.snp_file, mode=j.mode, outdir=j.outdir, caller=j.caller) for j in jobs],
                "world": gpus, "backend": backend, "body": body, "n_bins": n_bins, "alleles": alleles, "strict": strict,
                "same_device": same_device, "result": os.path.join(tmp, "result.pkl"), "groups": groups, "post": post,
                "post_args": post_args, "timeout": timeout}
        sp = os.path.join(tmp, "spec.pkl")
        with open(sp, "wb") as fh:
            pickle.dump(spec, fh)
        cmd = [sys.executable, "-m", "quasimodo_amd.multigpu", sp]
        try:
            spawn_ranks(cmd, gpus, timeout=timeout)   # (starts the ranks once more when their port was taken under them)
        except RankFailure as e:
            raise RuntimeError(str(e)) from None
        parts = []
        for r in range(gpus):   # every rank has left with status 0: its file is complete (written under another name, then renamed)
            with open("%s.%d" % (spec["result"], r), "rb") as fh:
                parts.append(pickle.load(fh))
        res = merge_ranks(len(jobs), parts)
    for j, st in zip(jobs, res["stats"]):
        j.stats = st
        if st.get("pure_strain"):
            j.tp_out = ""
    return jobs, res


def _main(argv):
    sp = argv[0]
    rank = int(os.environ["RANK"]) if len(argv) < 2 else int(argv[1])
    with open(sp, "rb") as fh:
        spec = pickle.load(fh)
    import torch
    import torch.distributed as dist
    from .extract import Job, _paths
    world, backend = spec["world"], spec["backend"]
    jobs = [Job(**d) for d in spec["jobs"]]
    for j in jobs:
        _paths(j)
    tmo = datetime.timedelta(seconds=max(60.0, float(spec.get("timeout") or DEFAULT_TIMEOUT)))
    if backend == "nccl":
        dev = torch.device("cuda", 0 if spec["same_device"] else rank)
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=tmo)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world, timeout=tmo)
    try:
        res = run_rank(jobs, rank, world, backend=backend, body=spec["body"], n_bins=spec["n_bins"],
                       alleles=spec["alleles"], strict=spec["strict"], same_device=spec["same_device"], groups=spec.get("groups"),
                       post=spec.get("post"), post_args=spec.get("post_args"))
        if res.get("collective") is not None:
            res["collective"]["ops_at_exit"] = _collectives_so_far()   # nothing follows the all-reduce: still 1
        mine = "%s.%d" % (spec["result"], rank)   # the rows go to the parent, not to another rank
        with open(mine + ".tmp", "wb") as fh:
            pickle.dump(res, fh)
        os.replace(mine + ".tmp", mine)
    finally:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(_main(sys.argv[1:]))
