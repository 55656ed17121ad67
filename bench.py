#!/usr/bin/env python3
"""bench.py -- variant TP/FP classifications/sec on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config {2,3,4}]      (N > 1: starts one process per GPU itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W [--config {2,3,4}]

Workloads (config.workload names the one that ran; all synthetic, generated on the device, resident in HBM
before the timed region, DESIGN.md "Synthetic generator"):
  --config 2 (default)  BASELINE.json configs[2]: per GPU 1 000 VCFs x 1 000 000 SNP records, 5 Mb reference,
                        100 000-key truth set, 256-threshold ROC sweep
  --config 3            configs[3], one GPU's shard of the 8-GPU run: 1 250 VCFs x 10 000 000 records, 50 Mb
                        reference, 1 000 000 truth keys (267 GB resident)
  --config 4            configs[4], one GPU's shard: 6 250 VCFs x 2 000 000 mixed SNP + indel records (30 % with
                        variable-length alleles), three truth sets (VCF v against set v mod 3), allele-extended mode
`--vcfs` overrides the VCFs per GPU (rehearsals on small boxes).

A step = one pass of the whole path over the batch: classify (LDS-staged merge-join, class masks, ROC
histograms) -> finalize -> TP/FP index compaction -> qm_batch_finish (reads the per-VCF flags: unsorted VCFs
would be redone through the radix sort here), plus, for N > 1, the one RCCL all-reduce of the per-truth-set
confusion counters.  VCFs shard over ranks with no data-path collective (weak scaling: per-GPU work fixed).

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (k_classify), timed with HIP events on its own
stream inside the timed region; `roofline.step_frac` is SURVEY 8d's formula over the whole step of the timed batch, and
`roofline.step_frac_median` / `_min` / `_max` the same over that batch and `--alloc-reps` re-creations of it (the step moves
by several per cent with where a batch lands in physical memory: the median is the figure of record).  `cpu_baseline` is
the oracle (a C restatement of the reference's awk/fgrep semantics) timed on this box's host cores on a bounded
sample of the same VCFs -- a reported baseline, not the target.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s measured copy)

PRESETS = {
    2: dict(vcfs=1000, records=1_000_000, genome=5_000_000, truth=100_000, truth_seeds=(3,), seed=3000, indel_pct=0,
            name="BASELINE configs[2]"),
    3: dict(vcfs=1250, records=10_000_000, genome=50_000_000, truth=1_000_000, truth_seeds=(4,), seed=4000, indel_pct=0,
            name="BASELINE configs[3] (one GPU's shard of 10 000 VCFs x 10 M over 8 GPUs)"),
    4: dict(vcfs=6250, records=2_000_000, genome=10_000_000, truth=200_000, truth_seeds=(5, 6, 7), seed=5000, indel_pct=30,
            name="BASELINE configs[4] (one GPU's shard of 50 000 VCFs x 2 M over 8 GPUs; mixed SNP + indel, allele-extended mode)"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=int(os.environ.get("QM_BENCH_CONFIG", "2")), choices=sorted(PRESETS),
                    help="which BASELINE.json configuration (index into its `configs`)")
    ap.add_argument("--vcfs", type=int, default=int(os.environ.get("QM_BENCH_VCFS", "0")), help="VCFs per GPU (0 = the preset's)")
    ap.add_argument("--records", type=int, default=0, help="records per VCF (0 = the preset's)")
    ap.add_argument("--genome", type=int, default=0)
    ap.add_argument("--truth", type=int, default=0)
    ap.add_argument("--bins", type=int, default=256)
    ap.add_argument("--cpu-sample", type=int, default=int(os.environ.get("QM_BENCH_CPU_RECORDS", "100000000")),
                    help="records timed through the CPU oracle on rank 0 (N=1 only); 0 disables")
    ap.add_argument("--shell-sample", type=int, default=int(os.environ.get("QM_BENCH_SHELL_VCFS", "3")),
                    help="VCFs timed through the reference's own mechanism (awk + fgrep pipeline) on rank 0 (N=1, config 2 only); 0 disables")
    ap.add_argument("--shuffled", action="store_true", help="records permuted: every VCF takes the path of the unsorted ones (buckets, radix sort behind them)")
    ap.add_argument("--shuffled-vcfs", type=int, default=int(os.environ.get("QM_BENCH_SHUFFLED_VCFS", "256")),
                    help="also time the shuffled variant (bucket path) on this many VCFs at N=1, config 2; 0 disables")
    ap.add_argument("--shuffled3-vcfs", type=int, default=int(os.environ.get("QM_BENCH_SHUFFLED3_VCFS", "16")),
                    help="also time shuffled VCFs of configs[3]'s shape (10 M records on a 50 Mb reference: the two-level bucket path) on this many VCFs at N=1, config 2; 0 disables")
    ap.add_argument("--alleles-vcfs", type=int, default=int(os.environ.get("QM_BENCH_ALLELES_VCFS", "1000")),
                    help="also time the allele-extended variant (config 4's record shape) on this many VCFs at N=1, config 2; 0 disables")
    ap.add_argument("--shuffled-alleles-vcfs", type=int, default=int(os.environ.get("QM_BENCH_SHUFFLED_ALLELES_VCFS", "256")),
                    help="also time shuffled allele-extended VCFs (config 4's record shape: two entry streams, k_join_lean + k_join_ext) on this many VCFs at N=1, config 2; 0 disables")
    ap.add_argument("--shuffled4-vcfs", type=int, default=int(os.environ.get("QM_BENCH_SHUFFLED4_VCFS", "48")),
                    help="also time shuffled VCFs of configs[4]'s shape (2 M records on a 10 Mb reference, 30 % variable-length alleles, three truth sets: partitions of buckets) on this many VCFs at N=1, config 2; 0 disables")
    ap.add_argument("--multicontig-vcfs", type=int, default=int(os.environ.get("QM_BENCH_MULTICONTIG_VCFS", "256")),
                    help="also time VCFs sorted per contig (24 ascending runs one behind the other; CHROM is never compared) on this many VCFs at N=1, config 2; 0 disables")
    ap.add_argument("--alloc-reps", type=int, default=int(os.environ.get("QM_BENCH_ALLOC_REPS", "3")),
                    help="re-create the timed batch this many times after the timed region and report k_classify's time for each "
                         "(roofline.alloc_spread: the kernel moves by several per cent with where a batch lands in memory); N=1, config 2; 0 disables")
    ap.add_argument("--detail", default=os.environ.get("QM_BENCH_DETAIL", os.path.join(ROOT, "gpurun_out", "bench_detail.json")),
                    help="where the full record goes (every side measurement with its notes, nested); the ONE line on stdout is the flat summary of it")
    ap.add_argument("--check-vcfs", type=int, default=2, help="VCFs of this rank's batch checked against the oracle after the timed region (N > 1 and configs 3 / 4)")
    args = ap.parse_args()
    P = dict(PRESETS[args.config])
    custom = False
    for k in ("vcfs", "records", "genome", "truth"):
        v = getattr(args, k)
        if v:
            custom = custom or (k != "vcfs" and v != P[k])
            P[k] = v
    alleles = P["indel_pct"] > 0

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as typed: this process has touched no GPU (nothing of torch or HIP is imported yet) and
        # becomes the launcher -- one fresh child per GPU with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rank 0's JSON line
        # relayed, everybody stopped and a non-zero exit if one rank fails.  Under torch.distributed.run WORLD_SIZE is set
        # and every process is a rank.
        import importlib.util
        spec = importlib.util.spec_from_file_location("qm_launch", os.path.join(ROOT, "quasimodo_amd", "launch.py"))
        launch = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(launch)
        sys.exit(launch.main_relay([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus,
                                   timeout=float(os.environ.get("QM_BENCH_TIMEOUT", "3000"))))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if os.environ.get("QM_BENCH_INJECT"):
        # tests of the launcher: the step is injected (module:function -> the dict rank 0 prints); no GPU is touched
        import importlib
        mod, fn = os.environ["QM_BENCH_INJECT"].split(":")
        out = getattr(importlib.import_module(mod), fn)(args, rank, world)
        if rank == 0:
            print(json.dumps(out))
        return

    # ONE line on stdout: whatever a library prints there on the way (RCCL's version banner when a communicator is made) goes to
    # stderr -- file descriptor 1 is pointed at 2 until rank 0 prints its line
    sys.stdout.flush()
    fd_out = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist
    import quasimodo_amd as q

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    if os.environ.get("QM_BENCH_SAME_DEVICE") == "1":   # rehearsal of the N > 1 code path on a one-GPU box
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # QM_BENCH_FORCE_PG=1: the process group and the collective also at N = 1 (a one-GPU box can at least show that RCCL
    # initialises on this device and all-reduces the engine's buffer on the bench's stream)
    force_pg = world == 1 and os.environ.get("QM_BENCH_FORCE_PG") == "1"
    if force_pg:
        os.environ.setdefault("MASTER_PORT", "29533")
    if world > 1 or force_pg:
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
    tseeds = P["truth_seeds"]
    tids = [eng.truth_synth(P["genome"], P["truth"], ts, indel_pct=P["indel_pct"]) for ts in tseeds]
    t_unique = eng.truth_size(tids[0], alleles=alleles)
    n_vcf = P["vcfs"]
    # VCF v of rank r is global VCF r * n_vcf + v: seed (preset seed) + that, truth set (global v) mod 3 in config 4
    g0 = rank * n_vcf
    batch = eng.batch([P["records"]] * n_vcf, [tids[(g0 + v) % len(tids)] for v in range(n_vcf)], n_bins=args.bins, alleles=alleles)
    batch.synth(P["genome"], P["truth"], None if len(tseeds) > 1 else tseeds[0], P["seed"] + g0, shuffled=args.shuffled, indel_pct=P["indel_pct"])
    # One explicit (non-default) stream carries the engine's kernels AND the collective, so the all-reduce is
    # ordered after the counters it sums.  (A NULL handle would make the engine use its own private stream.)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    glob = torch.zeros((batch.n_truth, 3, args.bins), dtype=torch.int64, device=dev)
    assert stream.cuda_stream != 0

    def step():
        batch.run(stream=stream.cuda_stream, global_dev=glob.data_ptr())
        batch.finish(stream=stream.cuda_stream)          # per-VCF flags read back; the radix-sort path of unsorted VCFs completes here
        if world > 1 or force_pg:
            dist.all_reduce(glob, op=dist.ReduceOp.SUM)   # the path's only collective

    def fence():
        if world > 1 or force_pg:
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
    per_rank = None
    if world > 1:
        # every rank's own figures, so that a scaling curve can tell an unbalanced or slow rank from the fabric: its wall clock over
        # the K steps (each ends in the all-reduce, so the ranks' walls agree unless one is late to the closing barrier) and its
        # own kernels' time per step by HIP events (no collective in it)
        mine = torch.tensor([dt / args.steps * 1e3, tm["total_ms"], tm["classify_ms"], tm["compact_ms"]], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rows = np.array([e.cpu().numpy() for e in every])
        per_rank = {k: {"min": float(rows[:, i].min()), "max": float(rows[:, i].max()), "all": [float(x) for x in rows[:, i]]}
                    for i, k in enumerate(("ms_per_step", "kernels_ms", "classify_ms", "compact_ms"))}
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # ---- correctness guards on the timed data (cheap, outside the timed region) ----
    scal = batch.scalars()
    roc = batch.roc()
    assert int(scal[:, 6].sum()) == n_vcf * P["records"]
    assert np.array_equal(roc[:, 0, 20].astype(np.int64), scal[:, 1]) and np.array_equal(roc[:, 1, 20].astype(np.int64), scal[:, 2])
    local_sum = np.zeros((batch.n_truth, 3, args.bins), np.int64)
    for w, t_ in enumerate(tids):
        local_sum[t_] = roc[(w - g0) % len(tids)::len(tids)].sum(axis=0).astype(np.int64)
    got = glob.cpu().numpy()
    if world == 1:
        assert np.array_equal(got, local_sum), "per-truth counters != sum of the per-VCF ROC rows"
    else:
        ref = torch.from_numpy(local_sum).to(dev)
        dist.all_reduce(ref, op=dist.ReduceOp.SUM)
        if not np.array_equal(got, ref.cpu().numpy()):
            raise AssertionError("all-reduced counters != sum over ranks of the per-VCF ROC rows: rank %d got %d, local %d, summed %d"
                                 % (rank, int(got.sum()), int(local_sum.sum()), int(ref.sum().item())))
    checked = 0
    if (world > 1 or args.config != 2) and args.check_vcfs > 0:
        checked = oracle_check(batch, P, tseeds, g0, alleles, args.bins, min(args.check_vcfs, n_vcf), roc, scal)

    total_records = float(n_vcf) * P["records"] * world
    value = total_records * args.steps / dt
    # algorithmic bytes of one k_classify launch: 17 B per record + 12 B per truth key per VCF
    alg_bytes = n_vcf * (17.0 * P["records"] + 12.0 * t_unique)
    k1_s = tm["classify_ms"] * 1e-3
    step_s = dt / args.steps
    roof_kernel = "k_classify"
    if args.shuffled:
        # the optimistic k_classify pass stops early on shuffled VCFs; the work is the radix-sort path, which is
        # overhead in SURVEY 8d's accounting: the same algorithmic bytes over the whole step
        k1_s = step_s
        roof_kernel = "whole step (optimistic pass + radix-sort path + packed k_classify)"
    achieved = alg_bytes / k1_s / 1e9 if k1_s > 0 else 0.0
    # HBM bytes per launch come from a SEPARATE --pmc pass (rocprofv3 cannot count inside this run): profiles/traffic.json, written
    # by tools/pmc_to_profiles.py together with the id of the device code it profiled.  The figure is quoted only for the very
    # same device code and workload; otherwise `traffic` is null and `traffic_note` says why.
    traffic, traffic_build, traffic_note = None, None, "no profiles/traffic.json"
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            traffic_build = tj.get("kernels_build")
            if args.shuffled or alleles or tj.get("vcfs") != n_vcf or tj.get("records") != P["records"]:
                traffic_note = "the PMC pass measured another workload (%s VCFs x %s records, sorted, single-base)" % (tj.get("vcfs"), tj.get("records"))
            elif traffic_build != q.kernel_source_id():
                traffic_note = "stale: the PMC pass ran on device code %s, this is %s" % (traffic_build, q.kernel_source_id())
            else:
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_note = "rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE of k_classify, separate passes on this device code (tools/prof_pmc.sh)"
        except Exception as e:
            traffic, traffic_note = None, "unreadable profiles/traffic.json: %s" % str(e)[:80]

    workload = "%s%s: %d VCFs x %d %s per GPU, %d bp reference, %d truth keys%s, %d-threshold ROC%s" % (
        P["name"], " (custom shape)" if custom else "", n_vcf, P["records"], "mixed SNP + indel records" if alleles else "SNPs",
        P["genome"], t_unique, " x %d truth sets" % len(tids) if len(tids) > 1 else "", args.bins,
        ", records shuffled (radix-sort path)" if args.shuffled else ", position sorted")
    # the line's own copy stays under the 120 characters the driver's record keeps of a string
    mb = lambda x: "%g M" % (x / 1e6) if x % 100_000 == 0 else str(x)
    workload_short = "BASELINE configs[%d]%s%s: %d VCFs x %s %s/GPU, %sb ref, %d truth keys%s, %d-bin ROC, %s" % (
        args.config, " shard" if args.config != 2 else "", " (custom)" if custom else "", n_vcf, mb(P["records"]), "SNP+indel" if alleles else "SNPs", mb(P["genome"]), t_unique,
        " x%d" % len(tids) if len(tids) > 1 else "", args.bins, "shuffled" if args.shuffled else "sorted")
    out = {
        "metric": "variant TP/FP classifications/sec across all caller x sample VCFs",
        "value": value,
        "unit": "classifications/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": step_s * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int32",
        "data": "synthetic (generated on device; VCF seeds %d+v, truth seed%s %s)" % (P["seed"], "s" if len(tseeds) > 1 else "", ",".join(map(str, tseeds))),
        "config": {"workload": workload_short, "workload_long": workload, "baseline_config_index": args.config,
                   "vcfs_per_gpu": n_vcf, "records_per_vcf": P["records"], "parallelism": "vcf-shard x%d" % world,
                   "collective": ("1 all-reduce of [%d x 3 x %d] int64 per step" % (batch.n_truth, args.bins)
                                  + (" (QM_BENCH_FORCE_PG: a process group of one rank)" if force_pg else "")) if world > 1 or force_pg else "none",
                   # VCFs of this rank's batch compared with the oracle after the timed region: class bits, ROC row, scalars and both
                   # index lists (oracle_check), plus what the cpu_baseline leg verifies while it times the oracle (filled in below)
                   "vcfs_checked_against_oracle_per_rank": checked},
        "roofline": {"bound": "hbm", "kernel": roof_kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_build": traffic_build, "traffic_note": traffic_note,
                     "kernels_build": q.kernel_source_id(),
                     "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": k1_s * 1e3,
                     # SURVEY 8d's own formula over the WHOLE step: 18.2 B x classifications/s / peak
                     "step_frac": alg_bytes / step_s / 1e9 / HBM_PEAK_GBS},
        "kernels_ms": tm,
        "per_rank": per_rank,     # N > 1: min / max / all over the ranks of ms_per_step and of each rank's own kernel time
        "device_bytes": batch.device_bytes,
    }

    if rank == 0:
        # the second denominator SURVEY 8d asks for: what this very GPU streams, measured with the library's own
        # 16-byte-per-lane kernels after the timed region (read only / copy = read + written / write only)
        try:
            bw = eng.bw_probe(4 << 30, 5)
            out["roofline"].update(measured_read_GBps=bw["read_GBps"], measured_copy_GBps=bw["copy_GBps"], measured_write_GBps=bw["write_GBps"],
                                   frac_of_measured_read=achieved / bw["read_GBps"] if bw["read_GBps"] > 0 else None)
        except Exception as e:   # never fatal
            out["roofline"]["measured_error"] = str(e)[:120]
    side = rank == 0 and world == 1
    if side and args.config == 2 and not custom and not args.shuffled and args.alloc_reps > 0:
        # The same workload in fresh allocations, outside the timed region: within one allocation the step is steady to a per
        # cent, from one allocation to the next it moves by several (profiles/r03_alloc_pmc.log: the same requests, 5-9 % more
        # memory latency) -- the headline above is ONE draw of this spread, so the line also carries the WHOLE step (wall clock
        # over back-to-back steps, as the timed region measures it) of every allocation and the median of them: the figure of record.
        spread = [round(tm["classify_ms"], 4)]
        steps_ms = [round(step_s * 1e3, 4)]
        for _ in range(args.alloc_reps):
            b2 = eng.batch([P["records"]] * n_vcf, [tids[0]] * n_vcf, n_bins=args.bins)
            b2.synth(P["genome"], P["truth"], tseeds[0], P["seed"])
            for _ in range(2):
                b2.run(); b2.finish()
            steps_ms.append(round(_timed_steps(b2, max(5, min(args.steps, 10))) * 1e3, 4))
            b2.set_timing(True)
            for _ in range(5):
                b2.run()
            b2.finish()
            spread.append(round(b2.timings()["classify_ms"], 4))
            b2.close()
        frac_of = lambda ms: alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        med = float(np.median(steps_ms))
        out["roofline"]["alloc_spread"] = {"k_classify_ms": spread, "min": min(spread), "max": max(spread), "step_ms": steps_ms,
                                           "note": "first entry = the timed batch; the others = the same batch created again (k_classify: 5 runs each; "
                                                   "step: wall clock over back-to-back steps)"}
        out["roofline"].update(step_frac_median=frac_of(med), step_frac_min=frac_of(max(steps_ms)), step_frac_max=frac_of(min(steps_ms)),
                               step_ms_median=med, value_at_median_allocation=total_records / (med * 1e-3))
    if side and args.cpu_sample > 0:
        out["cpu_baseline"] = cpu_baseline(batch, P, tseeds, alleles, args.bins, max(1, min(n_vcf, args.cpu_sample // P["records"])))
        ck = out["cpu_baseline"].pop("checked")
        out["config"]["vcfs_checked_against_oracle_per_rank"] = max(checked, ck["roc_rows"])
        out["config"]["oracle_checks"] = dict(ck, full_vcfs=checked,
                                              note="roc_rows: VCFs whose [3][n_bins] ROC row equals the oracle's; class_bits_vcfs: VCFs whose per-record "
                                                   "class bits equal the oracle's; full_vcfs: class bits + ROC row + scalars + both index lists (oracle_check)")
    if side and args.config == 2 and not custom and args.shell_sample > 0:
        try:
            out["cpu_baseline_shell"] = shell_baseline(batch, P, min(args.shell_sample, n_vcf), tseeds[0])
        except Exception as e:   # awk / GNU grep missing on the box: a reported extra, never fatal
            out["cpu_baseline_shell"] = {"error": str(e)[:200]}
    if side and args.config == 2 and not custom and not args.shuffled and args.shuffled_vcfs > 0:
        out["shuffled_variant"] = shuffled_variant(eng, tids[0], P, args.bins, min(args.shuffled_vcfs, n_vcf), tseeds[0], roc)
    if side and args.config == 2 and not custom and not args.shuffled and args.shuffled3_vcfs > 0:
        out["shuffled_config3_variant"] = shuffled_config3_variant(eng, args.bins, args.shuffled3_vcfs)
    if side and args.config == 2 and not custom and not args.shuffled and args.alleles_vcfs > 0:
        out["alleles_variant"] = alleles_variant(eng, P, args.bins, min(args.alleles_vcfs, n_vcf))
    if side and args.config == 2 and not custom and not args.shuffled and args.shuffled_alleles_vcfs > 0:
        out["shuffled_alleles_variant"] = shuffled_alleles_variant(eng, P, args.bins, min(args.shuffled_alleles_vcfs, n_vcf))
    if side and args.config == 2 and not custom and not args.shuffled and args.shuffled4_vcfs > 0:
        out["shuffled_config4_variant"] = shuffled_config4_variant(eng, args.bins, args.shuffled4_vcfs)
    if side and args.config == 2 and not custom and not args.shuffled and args.multicontig_vcfs > 0:
        out["multicontig_variant"] = multicontig_variant(eng, tids[0], P, args.bins, min(args.multicontig_vcfs, n_vcf), tseeds[0], roc, scal)
    if rank == 0:
        write_detail(out, args.detail)
        sys.stdout.flush()
        os.dup2(fd_out, 1)
        print(json.dumps(compact_line(out)), flush=True)
        os.dup2(2, 1)
    batch.close()
    eng.close()
    if world > 1 or force_pg:
        dist.destroy_process_group()


LINE_LIMIT = 2000   # the driver keeps the last 2 000 characters of the output and only flat scalars of `roofline` / `cpu_baseline` / `config`


def _sig(x, n=5):
    """floats to n significant digits (a line of 60 figures has no room for 17 each)"""
    if isinstance(x, float) and x == x and x not in (float("inf"), float("-inf")):
        return float("%.*g" % (n, x))
    return x


def write_detail(out, path):
    """the full record (nested, with every note) beside the line: profiles/rNN_bench_*.json are copies of this file"""
    if not path:
        return
    try:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path + ".tmp", "w") as f:
            json.dump(out, f)
        os.replace(path + ".tmp", path)
    except OSError as e:   # a read-only tree: the line on stdout is what counts
        print("bench.py: detail file not written: %s" % e, file=sys.stderr)


def compact_line(out):
    """The ONE line rank 0 prints: the contract's keys, and every figure of record as a FLAT scalar of `roofline` or `cpu_baseline`
    (round 5's line was 9 KB of nested side measurements: the driver's record kept its last 2 000 characters and dropped the nested
    objects, so the shuffled, all-cores and shell figures never reached it).  At most LINE_LIMIT characters: the keys at the END of
    the priority lists below are dropped first if a line ever grows beyond that."""
    g = lambda d, *ks: (g(d.get(ks[0]), *ks[1:]) if len(ks) > 1 else d.get(ks[0])) if isinstance(d, dict) else None
    cfg, rf, cb = out["config"], out["roofline"], out.get("cpu_baseline") or {}
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    line["value"], line["ms_per_step"] = _sig(out["value"], 7), _sig(out["ms_per_step"], 6)
    line["config"] = {k: cfg[k] for k in ("workload", "baseline_config_index", "vcfs_per_gpu", "records_per_vcf", "parallelism", "collective",
                                          "vcfs_checked_against_oracle_per_rank") if k in cfg}
    # (key in the line, value), in the order they are given up when the line is too long (last first)
    sh, s3, al, sa, s4, mc = (out.get(k) or {} for k in ("shuffled_variant", "shuffled_config3_variant", "alleles_variant", "shuffled_alleles_variant",
                                                         "shuffled_config4_variant", "multicontig_variant"))
    alg_per_rec = rf["algorithmic_bytes_per_launch"] / max(1.0, float(cfg["vcfs_per_gpu"]) * cfg["records_per_vcf"])
    sfrac = lambda v: None if not v else v * alg_per_rec / 1e9 / HBM_PEAK_GBS     # SURVEY 8d's formula on a side workload's rate (same bytes per record)
    asp = rf.get("alloc_spread") or {}
    r = [("bound", rf["bound"]), ("kernel", rf["kernel"]), ("achieved", rf["achieved"]), ("peak", rf["peak"]), ("unit", rf["unit"]), ("frac", rf["frac"]),
         ("traffic", rf["traffic"]), ("kernel_ms", rf["kernel_ms"]), ("step_frac", rf["step_frac"]), ("algorithmic_bytes_per_launch", rf["algorithmic_bytes_per_launch"]),
         ("kernels_build", rf["kernels_build"]), ("step_frac_median", rf.get("step_frac_median")), ("step_frac_min", rf.get("step_frac_min")),
         ("step_frac_max", rf.get("step_frac_max")), ("alloc_step_ms_min", min(asp["step_ms"]) if asp else None), ("alloc_step_ms_max", max(asp["step_ms"]) if asp else None),
         ("shuffled_value_first_seen", sh.get("value")), ("shuffled_step_frac", sfrac(sh.get("value"))),
         ("variants_equal_sorted", None if not (sh or s3 or sa or s4 or mc) else all(v.get("roc_equals_sorted_variant", v.get("equals_sorted_variant")) is True
                                                                                  for v in (sh, s3, sa, s4, mc) if v)),
         ("alleles_step_frac", g(al, "step_frac")), ("alleles_classify_frac", g(al, "classify_frac")),
         ("shuffled3_value_first_seen", s3.get("value")), ("shuffled3_two_level", g(s3, "paths", "bucket_two_level")),
         ("multicontig_value", mc.get("value")), ("multicontig_path", mc.get("path")),
         ("shuffled4_value_first_seen", s4.get("value")), ("shuffled_alleles_value_first_seen", sa.get("value")),
         ("measured_read_GBps", rf.get("measured_read_GBps")), ("measured_copy_GBps", rf.get("measured_copy_GBps")),
         ("shuffled_value_repeated", sh.get("value_repeated_run")), ("traffic_build", rf.get("traffic_build") if rf.get("traffic_build") != rf["kernels_build"] else None),
         ("compact_ms", g(out, "kernels_ms", "compact_ms")), ("finalize_ms", g(out, "kernels_ms", "finalize_ms"))]
    shell = out.get("cpu_baseline_shell") or {}
    c = [("value", cb.get("value")), ("unit", cb.get("unit")), ("cores", cb.get("cores")), ("kind", cb.get("kind")), ("sample", cb.get("sample")),
         ("all_cores_value", g(cb, "all_cores", "value")), ("all_cores_n", g(cb, "all_cores", "cores")),
         ("shell_value", shell.get("value")), ("shell_all_cores_value", g(shell, "all_cores", "value")), ("shell_all_cores_jobs", g(shell, "all_cores", "jobs")),
         ("awk", shell.get("awk")), ("cpu_quota", g(shell, "all_cores", "cpu_quota")), ("shell_error", shell.get("error"))]
    keep_r, keep_c = [kv for kv in r if kv[1] is not None or kv[0] == "traffic"], [kv for kv in c if kv[1] is not None]

    def build():
        line["roofline"] = {k: _sig(v) for k, v in keep_r}
        if cb:
            line["cpu_baseline"] = {k: _sig(v) for k, v in keep_c}
        if out.get("per_rank"):
            line["per_rank_ms_min_max"] = [_sig(out["per_rank"]["ms_per_step"]["min"]), _sig(out["per_rank"]["ms_per_step"]["max"])]
        return json.dumps(line)
    fixed_r, fixed_c = 11, 5          # the contract's own keys are never dropped
    while len(build()) > LINE_LIMIT and (len(keep_r) > fixed_r or len(keep_c) > fixed_c):
        if len(keep_r) - fixed_r >= len(keep_c) - fixed_c:
            keep_r.pop()
        else:
            keep_c.pop()
    return line


def truth_of(P, tseeds, gv):
    """the (cached) truth keys VCF `gv` (global index) was generated against"""
    from oracle.synth import synth_truth_keys
    ts = tseeds[gv % len(tseeds)]
    key = (P["genome"], P["truth"], ts, P["indel_pct"])
    if key not in truth_of.cache:
        truth_of.cache[key] = synth_truth_keys(P["genome"], P["truth"], ts, P["indel_pct"])
    return truth_of.cache[key]


truth_of.cache = {}


def oracle_check(batch, P, tseeds, g0, alleles, bins, n_check, roc, scal):
    """first / last VCFs of this rank's batch against the oracle (checker only, after the timed region)"""
    import numpy as np
    from oracle import qm_oracle as O
    nv = batch.n_vcf
    picks = sorted({0, nv - 1} if n_check >= 2 else {0})
    for v in picks:
        cols = batch.columns(v)
        cls, oroc, sc = O.classify_columns(*cols, *truth_of(P, tseeds, g0 + v), n_bins=bins, ext=alleles)
        assert np.array_equal(batch.cls(v), cls), "class bits of VCF %d differ from the oracle" % v
        assert np.array_equal(roc[v], oroc), "ROC row of VCF %d differs from the oracle" % v
        assert [int(x) for x in scal[v][:5]] == [sc[k] for k in ("n_pass", "tp_lines", "fp_lines", "TP_R", "FP_R")]
        idx = batch.idx(v)
        n = len(cols[0])
        assert np.array_equal(idx[:sc["tp_lines"]], np.nonzero(cls == 3)[0]) and np.array_equal(idx[n - sc["fp_lines"]:], np.nonzero(cls == 1)[0])
    return len(picks)


def _timed_steps(b, steps):
    import torch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        b.run()
        b.finish()
    return (time.perf_counter() - t0) / steps


def _timed_with_and_without_memory(b, steps):
    """A batch remembers which VCFs a finish found out of order while its columns stay the same: `ms_per_step` is
    the repeated run, `ms_per_step_unseen` the same run with that memory switched off (QM_MEMO=0: optimistic pass over every
    VCF, flags read back, then the bucket path -- what a batch pays when it is run for the first time).  The variants report
    the FIRST-SEEN figure as their `value` (`value_repeated_run` beside it)."""
    dt = _timed_steps(b, steps)
    old = os.environ.get("QM_MEMO")
    os.environ["QM_MEMO"] = "0"
    try:
        b.run(); b.finish()
        dt0 = _timed_steps(b, steps)
    finally:
        if old is None:
            del os.environ["QM_MEMO"]
        else:
            os.environ["QM_MEMO"] = old
    b.run(); b.finish()
    return dt, dt0


def _timed_fresh_columns(b, steps, rewrite):
    """First-seen without any knob: the columns are written again before every step (`rewrite(i)`, untimed) -- what a caller does who
    streams new VCFs through one batch object; the write itself clears the batch's memory of what it found (include/qmvt.h).  Every step is
    timed on its own, from an idle device to the end of qm_batch_finish."""
    import torch
    tot = 0.0
    for i in range(steps):
        rewrite(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        b.run()
        b.finish()
        tot += time.perf_counter() - t0
    return tot / steps


def shuffled_variant(eng, tid, P, bins, nv, tseed, sorted_roc):
    """Config 3's second variant: the same VCFs with their records permuted, so every VCF takes the
    optimistic pass, is found out of order and goes through the batched radix-sort path.  A side
    measurement on a subset; its counters must equal those of the sorted VCFs."""
    import numpy as np
    import torch
    b = eng.batch([P["records"]] * nv, [tid] * nv, n_bins=bins)
    b.synth(P["genome"], P["truth"], tseed, P["seed"], shuffled=True)
    for _ in range(1):
        b.run()
        b.finish()
    steps = 3
    dt, dt0 = _timed_with_and_without_memory(b, steps)
    # the same shape with OTHER records every step (other seeds), then the first records again for the comparison below
    dtf = _timed_fresh_columns(b, steps, lambda i: b.synth(P["genome"], P["truth"], tseed, P["seed"] + 104729 * (i + 1), shuffled=True))
    b.synth(P["genome"], P["truth"], tseed, P["seed"], shuffled=True)
    b.run()
    b.finish()
    ok = bool(np.array_equal(b.roc(), sorted_roc[:nv])) and int(b.scalars()[:, 5].sum()) == 0
    paths = b.path_stats()
    b.close()
    return {"value": nv * float(P["records"]) / dt0, "value_repeated_run": nv * float(P["records"]) / dt, "unit": "classifications/s", "vcfs": nv, "steps": steps,
            "ms_per_step": dt * 1e3, "ms_per_step_unseen": dt0 * 1e3, "ms_per_step_fresh_columns": dtf * 1e3, "value_fresh_columns": nv * float(P["records"]) / dtf,
            "roc_equals_sorted_variant": ok, "paths": paths,
            "note": "records permuted: optimistic pass (stops early) + bucket path (one scatter pass into 256 position buckets per VCF, k_join_lean: two bits per "
                    "POSITION of the bucket in LDS, no sort and no hashing inside a bucket, TP bits straight into the input-order mask); `value` = first-seen (QM_MEMO=0 on the same "
                    "records), `value_fresh_columns` = the columns written again with other records before every step, no knob; `paths` says where the VCFs went"}


def multicontig_variant(eng, tid, P, bins, nv, tseed, sorted_roc, sorted_scal, contigs=24):
    """The same VCFs sorted PER CONTIG: 24 ascending runs one behind the other, POS restarting with every CHROM (the reference never
    compares CHROM -- extract_TP_FP_SNPs.py:47 --, so the runs share one position space and a key may repeat across them).  What
    `vareval` on a multi-contig genome hands over.  The records are the sorted VCF's, so every counter must equal the sorted run's."""
    import numpy as np
    b = eng.batch([P["records"]] * nv, [tid] * nv, n_bins=bins)
    b.synth(P["genome"], P["truth"], tseed, P["seed"], shuffled=contigs)
    b.run(); b.finish()
    steps = 3
    dt, dt0 = _timed_with_and_without_memory(b, steps)
    ok = bool(np.array_equal(b.roc(), sorted_roc[:nv]) and np.array_equal(b.scalars()[:, :5], sorted_scal[:nv, :5]))
    paths = b.path_stats()
    b.close()
    took = [k for k, v in paths.items() if isinstance(v, int) and v > 0 and k != "finishes"] if isinstance(paths, dict) else []
    return {"value": nv * float(P["records"]) / dt0, "value_repeated_run": nv * float(P["records"]) / dt, "unit": "classifications/s", "vcfs": nv, "contigs": contigs,
            "steps": steps, "ms_per_step": dt * 1e3, "ms_per_step_unseen": dt0 * 1e3, "equals_sorted_variant": ok, "paths": paths, "path": ",".join(took)[:40],
            "note": "every VCF = %d ascending runs (per-contig sorted); `value` = first-seen" % contigs}


def shuffled_config3_variant(eng, bins, nv):
    """configs[3]'s VCF shape (10 M records, 50 Mb reference, 10^6 truth keys) with the records permuted: too large for 256
    buckets of 8 192 records, so the VCFs take the two-level bucket path (a first scatter into partitions of 2^27 keys sized
    by a counting pass, then the one-level path per partition).  A side measurement; the counters must equal those of the
    same VCFs in position order."""
    import numpy as np
    import torch
    P3 = PRESETS[3]
    tid = eng.truth_synth(P3["genome"], P3["truth"], P3["truth_seeds"][0])
    rocs = {}
    for shuffled in (False, True):
        b = eng.batch([P3["records"]] * nv, [tid] * nv, n_bins=bins)
        b.synth(P3["genome"], P3["truth"], P3["truth_seeds"][0], P3["seed"], shuffled=shuffled)
        b.run(); b.finish()
        if shuffled:
            steps = 3
            dt, dt0 = _timed_with_and_without_memory(b, steps)
            paths = b.path_stats()
        rocs[shuffled] = (b.roc(), b.scalars()[:, :5].copy())
        b.close()
    eng.truth_release(tid)
    ok = bool(np.array_equal(rocs[True][0], rocs[False][0]) and np.array_equal(rocs[True][1], rocs[False][1]))
    return {"value": nv * float(P3["records"]) / dt0, "value_repeated_run": nv * float(P3["records"]) / dt, "unit": "classifications/s", "vcfs": nv, "records_per_vcf": P3["records"], "steps": steps,
            "ms_per_step": dt * 1e3, "ms_per_step_unseen": dt0 * 1e3, "equals_sorted_variant": ok, "paths": paths,
            "note": "10 M-record VCFs permuted: two partitions of 256 WIDE buckets (2^17 positions, up to 32 768 records) filled by ONE pass of the 512-digit "
                    "k_bucket_scatter over the columns, k_join_lean<.., BIG> per bucket (`paths.bucket_partitions`; through round 5: two levels, `bucket_two_level`)"}


def alleles_variant(eng, P, bins, nv):
    """BASELINE configs[4]'s record shape on config 2's sizes: mixed SNP + indel records (30 %) with variable-length
    alleles, matched exactly in the allele-extended mode (a build-defined widening of the reference's
    single-base filter).  A side measurement on a subset; VCF 0 is checked against the oracle."""
    import numpy as np
    from oracle import qm_oracle as O
    from oracle.synth import synth_truth_keys
    pct, tseeds = 30, (5, 6, 7)
    tids = [eng.truth_synth(P["genome"], P["truth"], ts, indel_pct=pct) for ts in tseeds]   # three truth sets: VCF v uses v mod 3
    tid, tseed = tids[0], tseeds[0]
    b = eng.batch([P["records"]] * nv, [tids[v % 3] for v in range(nv)], n_bins=bins, alleles=True)
    b.synth(P["genome"], P["truth"], None, 5000, indel_pct=pct)
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
    cls, roc, sc = O.classify_columns(*cols, *synth_truth_keys(P["genome"], P["truth"], tseed, pct), n_bins=bins, ext=True)
    ok = bool(np.array_equal(b.cls(0), cls) and np.array_equal(b.roc()[0], roc))
    t_ext = eng.truth_size(tid, alleles=True)
    b.close()
    for t in tids:
        eng.truth_release(t)
    alg = nv * (17.0 * P["records"] + 12.0 * t_ext)
    return {"value": nv * float(P["records"]) * steps / dt, "unit": "classifications/s", "vcfs": nv, "steps": steps,
            "ms_per_step": dt / steps * 1e3, "classify_ms": tm["classify_ms"], "classify_GBps": alg / tm["classify_ms"] / 1e6,
            "classify_frac": alg / tm["classify_ms"] / 1e6 / HBM_PEAK_GBS, "step_frac": alg / (dt / steps) / 1e9 / HBM_PEAK_GBS,
            "indel_pct": pct, "equals_oracle_on_vcf0": ok,
            "note": "k_classify<false,true>: allele codes staged in LDS beside the keys, (key, ref, alt) equality"}


def shuffled_alleles_variant(eng, P, bins, nv):
    """BASELINE configs[4]'s record shape (30 % of the records with variable-length alleles) with the records permuted: the bucket
    path with two entry streams -- single-base records through k_join_lean, the others (16-byte entries with their allele
    codes) through k_join_ext, exact on (position, REF, ALT).  A side measurement; the counters must equal the sorted run's."""
    import numpy as np
    import torch
    pct, tseed = 30, 5
    tid = eng.truth_synth(P["genome"], P["truth"], tseed, indel_pct=pct)
    rows = {}
    for shuffled in (False, True):
        b = eng.batch([P["records"]] * nv, [tid] * nv, n_bins=bins, alleles=True)
        b.synth(P["genome"], P["truth"], tseed, 5000, shuffled=shuffled, indel_pct=pct)
        b.run(); b.finish()
        if shuffled:
            steps = 3
            dt, dt0 = _timed_with_and_without_memory(b, steps)
            paths = b.path_stats()
        rows[shuffled] = (b.roc(), b.scalars()[:, :5].copy())
        b.close()
    eng.truth_release(tid)
    ok = bool(np.array_equal(rows[True][0], rows[False][0]) and np.array_equal(rows[True][1], rows[False][1]))
    return {"value": nv * float(P["records"]) / dt0, "value_repeated_run": nv * float(P["records"]) / dt, "unit": "classifications/s", "vcfs": nv, "steps": steps, "ms_per_step": dt * 1e3, "ms_per_step_unseen": dt0 * 1e3,
            "indel_pct": pct, "equals_sorted_variant": ok, "paths": paths,
            "note": "allele-extended VCFs permuted: one scatter into two entry streams per bucket, k_join_lean (single-base records) + k_join_ext (the others)"}


def shuffled_config4_variant(eng, bins, nv):
    """configs[4]'s VCF shape (2 M records on a 10 Mb reference, 30 % of them with variable-length alleles, VCF v against truth
    set v mod 3) with the records permuted: too many records and too wide a key range for 256 buckets, so every partition of 2^27
    keys is a segment of the one-level scatter that reads its VCF's columns and keeps its own key range; two entry streams and two
    joins per bucket.  A side measurement; the counters must equal those of the same VCFs in position order."""
    import numpy as np
    P4 = PRESETS[4]
    tids = [eng.truth_synth(P4["genome"], P4["truth"], ts, indel_pct=P4["indel_pct"]) for ts in P4["truth_seeds"]]
    rows = {}
    for shuffled in (False, True):
        b = eng.batch([P4["records"]] * nv, [tids[v % len(tids)] for v in range(nv)], n_bins=bins, alleles=True)
        b.synth(P4["genome"], P4["truth"], None, P4["seed"], shuffled=shuffled, indel_pct=P4["indel_pct"])
        b.run(); b.finish()
        if shuffled:
            steps = 3
            dt, dt0 = _timed_with_and_without_memory(b, steps)
            paths = b.path_stats()
        rows[shuffled] = (b.roc(), b.scalars()[:, :5].copy())
        b.close()
    for t in tids:
        eng.truth_release(t)
    ok = bool(np.array_equal(rows[True][0], rows[False][0]) and np.array_equal(rows[True][1], rows[False][1]))
    return {"value": nv * float(P4["records"]) / dt0, "value_repeated_run": nv * float(P4["records"]) / dt, "unit": "classifications/s", "vcfs": nv, "records_per_vcf": P4["records"], "steps": steps,
            "ms_per_step": dt * 1e3, "ms_per_step_unseen": dt0 * 1e3, "indel_pct": P4["indel_pct"], "equals_sorted_variant": ok, "paths": paths,
            "note": "2 M-record allele-extended VCFs on a 10 Mb reference permuted: two partitions of 2^27 keys per VCF, each a segment of the one-level "
                    "scatter reading the VCF's columns (SortSeg.part); k_join_lean + k_join_ext per bucket; `paths` counts them as bucket_partitions"}


def shell_baseline(batch, P, n_sample, tseed):
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
    tp, tr, ta = synth_truth_keys(P["genome"], P["truth"], tseed)
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
        nl = lambda q: sum(1 for ln in open(q, "rb") if not ln.startswith(b"#"))
        for v, p in enumerate(paths):
            assert nl(p + ".filtered") == scal[v][0] and nl(p + ".tp") == scal[v][1] and nl(p + ".fp") == scal[v][2], \
                "shell pipeline and GPU disagree on VCF %d" % v
        # SURVEY 8d's second form: the same five commands with one VCF per host core.  The texts are rendered once: job k reads
        # the text of VCF k mod n_sample (the reads are the same work whichever VCF) and writes outputs of its own.  One job is
        # ~9 processes (3 bash, 3 grep header, 5 awk / grep stages), and the box allows 1 024 of ours at once: jobs <= 48.
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        quota = None
        try:
            q_, per_ = open("/sys/fs/cgroup/cpu.max").read().split()
            quota = None if q_ == "max" else float(q_) / float(per_)
        except Exception:
            pass
        jobs = max(1, min(cores, 48, int(os.environ.get("QM_BENCH_SHELL_JOBS", "48"))))

        def one(k):
            p = paths[k % n_sample]
            o = os.path.join(w, "job%d" % k)
            f = flt % p
            subprocess.run(["bash", "-c", '(grep -E "^#" %s;%s) > %s.filtered' % (p, f, o)], check=True)
            a = subprocess.Popen(["bash", "-c", '(grep -E "^#" %s;grep -F -wf <(%s) <(%s)) > %s.tp' % (p, gs, f, o)])
            b = subprocess.Popen(["bash", "-c", '(grep -E "^#" %s;grep -F -wvf <(%s) <(%s)) > %s.fp' % (p, gs, f, o)])
            a.wait(); b.wait()
            return o

        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(jobs) as ex:
            t0 = time.perf_counter()
            outs = list(ex.map(one, range(jobs)))
            dtp = time.perf_counter() - t0
        for k, o in enumerate(outs):
            v = k % n_sample
            assert nl(o + ".filtered") == scal[v][0] and nl(o + ".tp") == scal[v][1] and nl(o + ".fp") == scal[v][2], \
                "parallel shell pipeline and GPU disagree on job %d" % k
    n = float(n_sample) * P["records"]
    return {"value": n / dt, "unit": "classifications/s", "cores": 3, "kind": "reference-mechanism",
            "sample": "first %d VCFs as text (%.0f MB each), awk filter + fgrep -wf/-wvf as extract_TP_FP_SNPs.py:24-57, VCFs serial, "
                      "<= 3 concurrent pipelines per VCF, %.1f s; line counts equal the GPU's" % (n_sample, mb, dt),
            "all_cores": {"value": float(jobs) * P["records"] / dtp, "unit": "classifications/s", "jobs": jobs, "cores": cores, "cpu_quota": quota,
                          "seconds": dtp,
                          "sample": "%d VCF jobs at once, one per host core (capped at 48: a job is ~9 processes), each the same five commands on the "
                                    "text of VCF k mod %d with outputs of its own; line counts of every job equal the GPU's" % (jobs, n_sample)},
            "awk": awkv}


def cpu_baseline(batch, P, tseeds, alleles, bins, n_sample):
    """The oracle (single thread, C) on the first n_sample VCFs of this very batch; its
    answers are also checked against the GPU's.  Reported, never used by the product."""
    import numpy as np
    from oracle import qm_oracle as O
    truths = [truth_of(P, tseeds, v) for v in range(n_sample)]
    cols = [batch.columns(v) for v in range(n_sample)]
    roc = batch.roc()
    t0 = time.perf_counter()
    res = [O.classify_columns(*c, *t, n_bins=bins, ext=alleles) for c, t in zip(cols, truths)]
    dt = time.perf_counter() - t0
    for v, (cls, oroc, sc) in enumerate(res):
        assert np.array_equal(oroc, roc[v]), "GPU ROC row %d differs from the oracle" % v
    assert np.array_equal(batch.cls(0), res[0][0]), "GPU class bits differ from the oracle"
    n = float(sum(len(c[0]) for c in cols))
    out = {"value": n / dt, "unit": "classifications/s", "cores": 1, "kind": "port",
           "sample": "first %d VCFs (%.3g records), oracle/qm_oracle.c, 1 thread, %.1f s" % (n_sample, n, dt),
           "checked": {"roc_rows": len(res), "class_bits_vcfs": 1}}
    # the same sample with one oracle call in flight per host core (ctypes drops the GIL during the call)
    from concurrent.futures import ThreadPoolExecutor
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, len(cols)))   # one VCF per thread: more threads than VCFs would idle
    with ThreadPoolExecutor(cores) as ex:
        t0 = time.perf_counter()
        res2 = list(ex.map(lambda ct: O.classify_columns(*ct[0], *ct[1], n_bins=bins, ext=alleles), list(zip(cols, truths))))
        dt2 = time.perf_counter() - t0
    assert all(np.array_equal(a[1], b[1]) for a, b in zip(res, res2))
    out["all_cores"] = {"value": n / dt2, "cores": cores, "seconds": dt2}
    return out


if __name__ == "__main__":
    main()
