#!/usr/bin/env python3
"""bench.py -- variant TP/FP classifications/sec on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (config.workload): BASELINE.json configs[2] -- per GPU 1 000 synthetic VCFs x
1 000 000 SNP records on a 5 Mb reference, 100 000-key truth set, 256-threshold ROC sweep,
generated on the device (DESIGN.md "Synthetic generator") and resident in HBM before the
timed region.  A step = one pass of the whole path over that batch: classify (LDS-staged
merge-join, ballot masks, ROC histograms) -> finalize -> TP/FP index compaction, plus, for
N > 1, the one RCCL all-reduce of the per-truth-set confusion counters.  VCFs shard over
ranks with no data-path collective (weak scaling: per-GPU work fixed).

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (k_classify), timed
with HIP events on its own stream inside the timed region; `cpu_baseline` is the oracle
(a C restatement of the reference's awk/fgrep semantics) timed on this box's host cores
on a bounded sample of the same VCFs -- a reported baseline, not the target.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s measured copy)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--vcfs", type=int, default=int(os.environ.get("QM_BENCH_VCFS", "1000")), help="VCFs per GPU")
    ap.add_argument("--records", type=int, default=1_000_000, help="records per VCF")
    ap.add_argument("--genome", type=int, default=5_000_000)
    ap.add_argument("--truth", type=int, default=100_000)
    ap.add_argument("--bins", type=int, default=256)
    ap.add_argument("--cpu-sample", type=int, default=int(os.environ.get("QM_BENCH_CPU_VCFS", "100")),
                    help="VCFs timed through the CPU oracle on rank 0 (N=1 only); 0 disables")
    ap.add_argument("--shell-sample", type=int, default=int(os.environ.get("QM_BENCH_SHELL_VCFS", "3")),
                    help="VCFs timed through the reference's own mechanism (awk + fgrep pipeline) on rank 0 (N=1 only); 0 disables")
    ap.add_argument("--shuffled", action="store_true", help="records permuted: every VCF takes the radix-sort path")
    ap.add_argument("--shuffled-vcfs", type=int, default=int(os.environ.get("QM_BENCH_SHUFFLED_VCFS", "128")),
                    help="also time the shuffled variant (radix-sort path) on this many VCFs at N=1; 0 disables")
    ap.add_argument("--alleles-vcfs", type=int, default=int(os.environ.get("QM_BENCH_ALLELES_VCFS", "128")),
                    help="also time the allele-extended variant (config 5 shape: 30 %% indels) on this many VCFs at N=1; 0 disables")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import quasimodo_amd as q

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    if os.environ.get("QM_BENCH_SAME_DEVICE") == "1":   # rehearsal of the N > 1 code path on a one-GPU box
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("QM_BENCH_BACKEND", "nccl")   # "nccl" is RCCL on ROCm; gloo only for the rehearsal
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    # The HIP library is built in-tree and travels with the snapshot; if it is missing (fresh checkout)
    # local rank 0 compiles it with hipcc and the others wait for it.  No other implementation exists.
    if not os.path.exists(q.library_path()):
        if int(os.environ.get("LOCAL_RANK", "0")) == 0:
            q.build_library()
        else:
            for _ in range(600):
                if os.path.exists(q.library_path()):
                    break
                time.sleep(0.5)
            time.sleep(1.0)
    eng = q.Engine(local_rank)
    tseed = 3
    tid = eng.truth_synth(args.genome, args.truth, tseed)
    t_unique = eng.truth_size(tid)
    n_vcf = args.vcfs
    batch = eng.batch([args.records] * n_vcf, [tid] * n_vcf, n_bins=args.bins)
    # VCF v of rank r is global VCF r * n_vcf + v: seed 3000 + that
    batch.synth(args.genome, args.truth, tseed, 3000 + rank * n_vcf, shuffled=args.shuffled)
    # One explicit (non-default) stream carries the engine's kernels AND the collective, so the all-reduce is
    # ordered after the counters it sums.  (A NULL handle would make the engine use its own private stream.)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    glob = torch.zeros((eng.n_truth, 3, args.bins), dtype=torch.int64, device=dev)
    assert stream.cuda_stream != 0

    def step():
        batch.run(stream=stream.cuda_stream, global_dev=glob.data_ptr())
        if args.shuffled:
            batch.finish(stream=stream.cuda_stream)      # radix-sort path completes here
        if world > 1:
            dist.all_reduce(glob, op=dist.ReduceOp.SUM)   # the path's only collective

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    batch.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    tm = batch.timings()
    batch.set_timing(False)
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # ---- correctness guards on the timed data (cheap, outside the timed region) ----
    batch.finish(stream=stream.cuda_stream)
    scal = batch.scalars()
    roc = batch.roc()
    assert int(scal[:, 6].sum()) == n_vcf * args.records
    assert np.array_equal(roc[:, 0, 20].astype(np.int64), scal[:, 1]) and np.array_equal(roc[:, 1, 20].astype(np.int64), scal[:, 2])
    local_sum = roc.sum(axis=0).astype(np.int64)
    got = glob.cpu().numpy()[tid]
    if world == 1:
        assert np.array_equal(got, local_sum), "per-truth counters != sum of the per-VCF ROC rows"
    else:
        ref = torch.from_numpy(local_sum).to(dev)
        dist.all_reduce(ref, op=dist.ReduceOp.SUM)
        if not np.array_equal(got, ref.cpu().numpy()):
            raise AssertionError("all-reduced counters != sum over ranks of the per-VCF ROC rows: rank %d got %d, local %d, summed %d"
                                 % (rank, int(got.sum()), int(local_sum.sum()), int(ref.sum().item())))

    total_records = float(n_vcf) * args.records * world
    value = total_records * args.steps / dt
    # algorithmic bytes of one k_classify launch: 17 B per record + 12 B per truth key per VCF
    alg_bytes = n_vcf * (17.0 * args.records + 12.0 * t_unique)
    k1_s = tm["classify_ms"] * 1e-3
    roof_kernel = "k_classify"
    if args.shuffled:
        # the optimistic k_classify pass stops early on shuffled VCFs; the work is the radix-sort path, which is
        # overhead in SURVEY 8d's accounting: the same algorithmic bytes over the whole step
        k1_s = dt / args.steps
        roof_kernel = "whole step (optimistic pass + radix-sort path + packed k_classify)"
    achieved = alg_bytes / k1_s / 1e9 if k1_s > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")   # HBM bytes per launch from a separate --pmc pass
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("vcfs") == n_vcf and tj.get("records") == args.records and not args.shuffled:
                traffic = tj.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    out = {
        "metric": "variant TP/FP classifications/sec across all caller x sample VCFs",
        "value": value,
        "unit": "classifications/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int32",
        "data": "synthetic (generated on device; seeds 3000+v, truth seed 3)",
        "config": {"workload": "%s: %d VCFs x %d SNPs per GPU, %d bp reference, %d truth keys, %d-threshold ROC%s"
                               % (workload_name(args), n_vcf, args.records, args.genome, t_unique, args.bins, ", records shuffled (radix-sort path)" if args.shuffled else ", position sorted"),
                   "vcfs_per_gpu": n_vcf, "records_per_vcf": args.records, "parallelism": "vcf-shard x%d" % world,
                   "collective": "1 all-reduce of [%d x 3 x %d] int64 per step" % (eng.n_truth, args.bins) if world > 1 else "none"},
        "roofline": {"bound": "hbm", "kernel": roof_kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": k1_s * 1e3},
        "kernels_ms": tm,
        "device_bytes": batch.device_bytes,
    }

    if rank == 0:
        # the second denominator SURVEY 8d asks for: what a plain device-to-device copy moves on this very GPU
        # (bytes read + bytes written per second), measured after the timed region
        try:
            cp = measured_copy_GBps(dev)
            out["roofline"]["measured_copy_GBps"] = cp
            out["roofline"]["frac_of_measured_copy"] = achieved / cp if cp > 0 else None
        except Exception as e:   # never fatal
            out["roofline"]["measured_copy_GBps"] = None
            out["roofline"]["measured_copy_error"] = str(e)[:120]
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        out["cpu_baseline"] = cpu_baseline(batch, args, min(args.cpu_sample, n_vcf), args.genome, args.truth, tseed)
    if rank == 0 and world == 1 and args.shell_sample > 0:
        try:
            out["cpu_baseline_shell"] = shell_baseline(batch, args, min(args.shell_sample, n_vcf), tseed)
        except Exception as e:   # awk / GNU grep missing on the box: a reported extra, never fatal
            out["cpu_baseline_shell"] = {"error": str(e)[:200]}
    if rank == 0 and world == 1 and not args.shuffled and args.shuffled_vcfs > 0:
        out["shuffled_variant"] = shuffled_variant(eng, tid, args, min(args.shuffled_vcfs, n_vcf), tseed, roc)
    if rank == 0 and world == 1 and not args.shuffled and args.alleles_vcfs > 0:
        out["alleles_variant"] = alleles_variant(eng, args, min(args.alleles_vcfs, n_vcf))
    if rank == 0:
        print(json.dumps(out))
    batch.close()
    eng.close()
    if world > 1:
        dist.destroy_process_group()


def measured_copy_GBps(dev, nbytes=2 << 30, reps=5):
    """device-to-device copy of 2 GiB: (bytes read + bytes written) / second"""
    import torch
    a = torch.empty(nbytes // 4, dtype=torch.int32, device=dev)
    b = torch.empty_like(a)
    a.fill_(1)
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    del a, b
    return 2.0 * nbytes / (ms * 1e-3) / 1e9


def workload_name(args):
    """which BASELINE.json configuration the arguments describe (the default is configs[2])"""
    if (args.records, args.genome, args.truth) == (1_000_000, 5_000_000, 100_000):
        return "BASELINE configs[2]"
    if (args.records, args.genome, args.truth) == (10_000_000, 50_000_000, 1_000_000):
        return "BASELINE configs[3] shape (per-GPU shard)"
    return "custom shape"


def shuffled_variant(eng, tid, args, nv, tseed, sorted_roc):
    """Config 3's second variant: the same VCFs with their records permuted, so every VCF takes the
    optimistic pass, is found out of order and goes through the batched radix-sort path.  A side
    measurement on a subset; its counters must equal those of the sorted VCFs."""
    import numpy as np
    import torch
    b = eng.batch([args.records] * nv, [tid] * nv, n_bins=args.bins)
    b.synth(args.genome, args.truth, tseed, 3000, shuffled=True)
    for _ in range(1):
        b.run()
        b.finish()
    torch.cuda.synchronize()
    steps = 3
    t0 = time.perf_counter()
    for _ in range(steps):
        b.run()
        b.finish()
    dt = time.perf_counter() - t0
    ok = bool(np.array_equal(b.roc(), sorted_roc[:nv])) and int(b.scalars()[:, 5].sum()) == 0
    b.close()
    return {"value": nv * float(args.records) * steps / dt, "unit": "classifications/s", "vcfs": nv, "steps": steps,
            "ms_per_step": dt / steps * 1e3, "roc_equals_sorted_variant": ok,
            "note": "records permuted: optimistic pass (stops early) + batched LSD radix sort (8-bit digits, XCD-aware tile order) + packed k_classify + TP bits scattered back"}


def alleles_variant(eng, args, nv):
    """BASELINE configs[4]'s shape on one GPU: mixed SNP + indel records (30 %) with variable-length
    alleles, matched exactly in the allele-extended mode (a build-defined widening of the reference's
    single-base filter).  A side measurement on a subset; VCF 0 is checked against the oracle."""
    import numpy as np
    from oracle import qm_oracle as O
    from oracle.synth import synth_truth_keys
    pct, tseeds = 30, (5, 6, 7)
    tids = [eng.truth_synth(args.genome, args.truth, ts, indel_pct=pct) for ts in tseeds]   # three truth sets: VCF v uses v mod 3
    tid, tseed = tids[0], tseeds[0]
    b = eng.batch([args.records] * nv, [tids[v % 3] for v in range(nv)], n_bins=args.bins, alleles=True)
    b.synth(args.genome, args.truth, None, 5000, indel_pct=pct)
    b.run(); b.finish()
    b.set_timing(True)
    steps = 5
    t0 = time.perf_counter()
    for _ in range(steps):
        b.run()
    b.finish()
    dt = time.perf_counter() - t0
    tm = b.timings()
    cols = b.columns(0)
    cls, roc, sc = O.classify_columns(*cols, *synth_truth_keys(args.genome, args.truth, tseed, pct), n_bins=args.bins, ext=True)
    ok = bool(np.array_equal(b.cls(0), cls) and np.array_equal(b.roc()[0], roc))
    t_ext = eng.truth_size(tid, alleles=True)
    b.close()
    alg = nv * (17.0 * args.records + 12.0 * t_ext)
    return {"value": nv * float(args.records) * steps / dt, "unit": "classifications/s", "vcfs": nv, "steps": steps,
            "ms_per_step": dt / steps * 1e3, "classify_ms": tm["classify_ms"], "classify_GBps": alg / tm["classify_ms"] / 1e6,
            "indel_pct": pct, "equals_oracle_on_vcf0": ok,
            "note": "k_classify<false,true>: allele codes staged in LDS beside the keys, (key, ref, alt) equality"}


def shell_baseline(batch, args, n_sample, tseed):
    """The reference's mechanism on this box's host: the same five shell commands per VCF that
    program/extract_TP_FP_SNPs.py:24-57 issues (awk filter, grep header, fgrep -wf / -wvf with
    process substitution), authored here (the reference file itself does not travel), on text
    renderings of the first n_sample VCFs of the batch, one VCF at a time as rules/extract_TP.smk:17
    serialises them.  Line counts are checked against the GPU's."""
    import shutil
    import subprocess
    import tempfile
    import numpy as np
    from oracle.synth import synth_truth_keys
    for tool in ("awk", "grep", "bash"):
        if not shutil.which(tool):
            raise RuntimeError("%s not found" % tool)
    awkv = subprocess.run("awk -W version 2>&1 | head -1 || awk --version | head -1", shell=True, capture_output=True, text=True).stdout.strip()
    bases = np.array([b"A", b"C", b"G", b"T"])
    tp, tr, ta = synth_truth_keys(args.genome, args.truth, tseed)
    scal = batch.scalars()
    hdr = b"##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n"

    def render(pos, ref, alt, qual):
        cols = [np.full(len(pos), b"chrS"), pos.astype("S"), np.full(len(pos), b"."), bases[ref], bases[alt], qual.astype(np.int64).astype("S"),
                np.full(len(pos), b"PASS"), np.full(len(pos), b"DP=30")]
        line = cols[0]
        for c in cols[1:]:
            line = np.char.add(np.char.add(line, b"\t"), c)
        return hdr + b"\n".join(line.tolist()) + b"\n"

    with tempfile.TemporaryDirectory() as w:
        truth = os.path.join(w, "truth.vcf")
        open(truth, "wb").write(render(tp, tr, ta, np.full(len(tp), 30.0)))
        paths = []
        for v in range(n_sample):
            pos, ref, alt, qual, _ = batch.columns(v)
            p = os.path.join(w, "S-1-10.R%d.c.vcf" % v)
            open(p, "wb").write(render(pos, ref, alt, qual))
            paths.append(p)
        flt = r'''awk -F"\t" '$4~/^[ACGT]$/&&$5~/^[ACGT]$/&&($6>=20||$6==".")' %s'''
        gs = r'''awk -F"\t" '$4~/^[ACGT]$/&&$5~/^[ACGT]$/{print $2, ".", $4, $5}' OFS="\t" %s''' % truth
        t0 = time.perf_counter()
        for p in paths:
            f = flt % p
            subprocess.run(["bash", "-c", '(grep -E "^#" %s;%s) > %s.filtered' % (p, f, p)], check=True)
            a = subprocess.Popen(["bash", "-c", '(grep -E "^#" %s;grep -F -wf <(%s) <(%s)) > %s.tp' % (p, gs, f, p)])
            b = subprocess.Popen(["bash", "-c", '(grep -E "^#" %s;grep -F -wvf <(%s) <(%s)) > %s.fp' % (p, gs, f, p)])
            a.wait(); b.wait()
        dt = time.perf_counter() - t0
        mb = os.path.getsize(paths[0]) / 1e6
        for v, p in enumerate(paths):
            nl = lambda q: sum(1 for ln in open(q, "rb") if not ln.startswith(b"#"))
            assert nl(p + ".filtered") == scal[v][0] and nl(p + ".tp") == scal[v][1] and nl(p + ".fp") == scal[v][2], \
                "shell pipeline and GPU disagree on VCF %d" % v
    n = float(n_sample) * args.records
    return {"value": n / dt, "unit": "classifications/s", "cores": 3, "kind": "reference-mechanism",
            "sample": "first %d VCFs as text (%.0f MB each), awk filter + fgrep -wf/-wvf as extract_TP_FP_SNPs.py:24-57, VCFs serial, "
                      "<= 3 concurrent pipelines per VCF, %.1f s; line counts equal the GPU's" % (n_sample, mb, dt),
            "awk": awkv}


def cpu_baseline(batch, args, n_sample, L, T, tseed):
    """The oracle (single thread, C) on the first n_sample VCFs of this very batch; its
    answers are also checked against the GPU's.  Reported, never used by the product."""
    import numpy as np
    from oracle import qm_oracle as O
    from oracle.synth import synth_truth_keys
    truth = synth_truth_keys(L, T, tseed)
    cols = [batch.columns(v) for v in range(n_sample)]
    roc = batch.roc()
    t0 = time.perf_counter()
    res = [O.classify_columns(*c, *truth, n_bins=args.bins) for c in cols]
    dt = time.perf_counter() - t0
    for v, (cls, oroc, sc) in enumerate(res):
        assert np.array_equal(oroc, roc[v]), "GPU ROC row %d differs from the oracle" % v
    assert np.array_equal(batch.cls(0), res[0][0]), "GPU class bits differ from the oracle"
    n = float(sum(len(c[0]) for c in cols))
    out = {"value": n / dt, "unit": "classifications/s", "cores": 1, "kind": "port",
           "sample": "first %d VCFs of the batch (%d records), oracle/qm_oracle.c classify_columns, 1 thread, %.1f s"
                     % (n_sample, int(n), dt)}
    # the same sample with one oracle call in flight per host core (ctypes drops the GIL during the call)
    from concurrent.futures import ThreadPoolExecutor
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, len(cols)))   # one VCF per thread: more threads than VCFs would idle
    with ThreadPoolExecutor(cores) as ex:
        t0 = time.perf_counter()
        res2 = list(ex.map(lambda c: O.classify_columns(*c, *truth, n_bins=args.bins), cols))
        dt2 = time.perf_counter() - t0
    assert all(np.array_equal(a[1], b[1]) for a, b in zip(res, res2))
    out["all_cores"] = {"value": n / dt2, "cores": cores, "seconds": dt2}
    return out


if __name__ == "__main__":
    main()
