"""Driver-run checks of the shapes BASELINE.json names beyond configs[2]: one GPU's FULL shard of the 8-GPU
configurations (configs[3]: 1 250 VCFs x 10 M records; configs[4]: 6 250 VCFs x 2 M mixed SNP + indel records, three
truth sets) -- 1.25e10 records, 267 GB resident of the 288 GB -- and the product's multi-GPU entry point with two ranks
on this one GPU.  First / middle / last VCF against the oracle, size-independent invariants over all of them."""
import os
import warnings

import numpy as np
import pytest

from conftest import ROOT, case_id, golden_cases, read_case

pytestmark = pytest.mark.gpu


def _full_shard(engine, oracle, nv, L, T, N, seeds, vseed, pct):
    from oracle.synth import synth_truth_keys
    ext = pct > 0
    tids = [engine.truth_synth(L, T, ts, indel_pct=pct) for ts in seeds]
    b = engine.batch([N] * nv, [tids[v % len(tids)] for v in range(nv)], alleles=ext)
    try:
        assert b.device_bytes > 250e9
        b.synth(L, T, None if len(seeds) > 1 else seeds[0], vseed, indel_pct=pct)
        b.set_timing(True)
        for _ in range(2):
            b.run()
        b.finish()
        tm = b.timings()
        roc, scal = b.roc(), b.scalars()
        assert (scal[:, 6] == N).all() and (scal[:, 5] == 1).all()
        assert np.array_equal(roc[:, 0, 20].astype(np.int64), scal[:, 1]) and np.array_equal(roc[:, 1, 20].astype(np.int64), scal[:, 2])
        assert (scal[:, 1] + scal[:, 2] == scal[:, 0]).all() and (scal[:, 3] <= scal[:, 7]).all()
        glob = b.global_counts()
        for w, t_ in enumerate(tids):      # what the all-reduce carries: per truth set, the sum of its VCFs' rows
            assert np.array_equal(glob[t_], roc[w::len(tids)].sum(axis=0))
        truths = [synth_truth_keys(L, T, ts, pct) for ts in seeds]
        for v in (0, nv // 2, nv - 1):     # the last one lies far beyond the 2^32nd record of the batch
            cols = b.columns(v)
            cls, oroc, sc = oracle.classify_columns(*cols, *truths[v % len(tids)], ext=ext)
            assert np.array_equal(b.cls(v), cls) and np.array_equal(roc[v], oroc), v
            assert [int(x) for x in scal[v][:5]] == [sc[k] for k in ("n_pass", "tp_lines", "fp_lines", "TP_R", "FP_R")], v
            idx = b.idx(v)
            assert np.array_equal(idx[:sc["tp_lines"]], np.nonzero(cls == 3)[0]) and np.array_equal(idx[N - sc["fp_lines"]:], np.nonzero(cls == 1)[0])
        rate = nv * float(N) / (tm["total_ms"] * 1e-3)
        print("full shard %d x %d: %.1f ms per pass, %.3g classifications/s, k_classify %.0f GB/s algorithmic" %
              (nv, N, tm["total_ms"], rate, nv * (17.0 * N + 12.0 * T) / tm["classify_ms"] / 1e6))
        if rate < 1e11:   # rates are judged by bench.py (--config 3 / 4 time these very shards): a slow box must not turn a parity run red
            warnings.warn("full shard %d x %d at %.3g classifications/s: below 1e11" % (nv, N, rate))
    finally:
        b.close()
        for t in tids:
            engine.truth_release(t)


def test_configs3_full_per_gpu_shard(engine, oracle):
    """BASELINE configs[3]: 10 000 VCFs x 10 M over 8 GPUs = 1 250 VCFs per GPU"""
    _full_shard(engine, oracle, 1250, 50_000_000, 1_000_000, 10_000_000, (4,), 4000, 0)


def test_configs4_full_per_gpu_shard(engine, oracle):
    """BASELINE configs[4]: 50 000 VCFs x 2 M mixed SNP + indel over 8 GPUs = 6 250 VCFs per GPU, VCF v against truth set v mod 3"""
    _full_shard(engine, oracle, 6250, 10_000_000, 200_000, 2_000_000, (5, 6, 7), 5000, 30)


def test_product_multi_gpu_entry_point_two_ranks_on_this_gpu(tmp_path, qmlib):
    """quasimodo_amd.multigpu.extract_many_sharded with the HIP engine on both ranks (same device, gloo for the collective
    since RCCL refuses two ranks on one GPU): LPT shards, files by both ranks, one all-reduce, rows gathered."""
    from quasimodo_amd.extract import Job
    from quasimodo_amd.multigpu import extract_many_sharded, truth_key
    cases = [c for c in golden_cases() if c["family"] in ("hcmv", "quirks", "custom")]
    jobs, exps = [], []
    for e in cases:
        vcf, truth, exp = read_case(e)
        root = tmp_path / e["family"] / e["mode"]
        vp = root / e["vcf"][len("input/"):]
        tp = root / e["truth"][len("input/"):]
        vp.parent.mkdir(parents=True, exist_ok=True)
        tp.parent.mkdir(parents=True, exist_ok=True)
        vp.write_bytes(vcf)
        tp.write_bytes(truth)
        jobs.append(Job(str(vp), str(tp), e["mode"], str(root / e["outdir"]), e["caller"]))
        exps.append((e, exp))
    jobs, res = extract_many_sharded(jobs, 2, backend="gloo", same_device=True, strict=True, timeout=600)
    for job, (e, exp) in zip(jobs, exps):
        assert open(job.filtered_out, "rb").read() == exp["filtered"], case_id(e)
        assert open(job.fp_out, "rb").read() == exp["fp"], case_id(e)
        if not e["pure"]:
            assert open(job.tp_out, "rb").read() == exp["tp"], case_id(e)
    assert all(len(s) > 0 for s in res["shards"])
    keys = res["truth_keys"]
    want = np.zeros((len(keys), 3, 256), np.int64)
    for j in jobs:
        if j.stats.get("roc") is not None:
            want[keys.index(truth_key(j))] += np.asarray(j.stats["roc"]).astype(np.int64)
    assert np.array_equal(res["counters"], want) and want.sum() > 0


def test_product_multi_gpu_entry_point_one_rank_over_rccl(tmp_path, qmlib):
    """The same entry point with the backend the product uses, "nccl" (= RCCL on ROCm), at the only world size one card allows:
    the child initialises the process group on its device and all-reduces the DEVICE buffer the engine filled -- the call is
    issued at every world size (sharding.allreduce_counters) -- and exchanges nothing else: the rows reach the parent through
    the rank's own result file.  What two ranks add to this is RCCL's transport between GPUs, which no one-GPU box can show."""
    from quasimodo_amd.extract import Job
    from quasimodo_amd.multigpu import extract_many_sharded, truth_key
    cases = [c for c in golden_cases() if c["family"] == "hcmv"][:12]
    jobs, exps = [], []
    for e in cases:
        vcf, truth, exp = read_case(e)
        root = tmp_path / e["family"] / e["mode"]
        vp = root / e["vcf"][len("input/"):]
        tp = root / e["truth"][len("input/"):]
        vp.parent.mkdir(parents=True, exist_ok=True)
        tp.parent.mkdir(parents=True, exist_ok=True)
        vp.write_bytes(vcf)
        tp.write_bytes(truth)
        jobs.append(Job(str(vp), str(tp), e["mode"], str(root / e["outdir"]), e["caller"]))
        exps.append((e, exp))
    jobs, res = extract_many_sharded(jobs, 1, backend="nccl", strict=True, timeout=600)
    for job, (e, exp) in zip(jobs, exps):
        assert open(job.fp_out, "rb").read() == exp["fp"], case_id(e)
        if not e["pure"]:
            assert open(job.tp_out, "rb").read() == exp["tp"], case_id(e)
    keys = res["truth_keys"]
    want = np.zeros((max(len(keys), 1), 3, 256), np.int64)
    for j in jobs:
        if j.stats.get("roc") is not None:
            want[keys.index(truth_key(j))] += np.asarray(j.stats["roc"]).astype(np.int64)
    assert np.array_equal(res["counters"], want) and want.sum() > 0
    # the product's ONE collective really went through RCCL: the process group of the child is "nccl", and its collective
    # counter moved by exactly one over the all-reduce and stood at one when the rank left (no gather, no barrier)
    (col,) = res["collectives"]
    assert col["backend"] == "nccl" and col["world"] == 1
    assert col["ops_after"] - col["ops_before"] == 1 and col["ops_at_exit"] == col["ops_after"] == 1, col


def test_workflows_on_two_ranks_of_this_gpu_write_what_one_gpu_writes(tmp_path, qmlib, engine):
    """run_hcmv_variantcall / run_vareval with gpus=2 and the HIP engine on both ranks (one card, gloo for the collective
    since RCCL refuses two ranks on one GPU): the VCFs dealt by sample, the FP overlap by the rank that holds the sample, the
    counters all-reduced from the device buffer qm_extract_files_ex filled -- every output file and the three tables byte for
    byte what the one-GPU workflow writes."""
    import filecmp
    from quasimodo_amd import workflow
    from test_tables_workflow import _build_bundle
    data = tmp_path / "data" / "snp"
    _build_bundle(str(data))
    o1, o2 = tmp_path / "one", tmp_path / "two"
    jobs1 = workflow.run_hcmv_variantcall(str(data), str(o1), engine=engine)
    jobs2 = workflow.run_hcmv_variantcall(str(data), str(o2), gpus=2, _backend="gloo", _same_device=True)
    res = workflow.run_hcmv_variantcall.last_result
    assert len(jobs1) == len(jobs2) == 60 and all(len(s) > 0 for s in res["shards"])
    n = 0
    for root, _, files in os.walk(o1 / "results"):
        for f in files:
            a = os.path.join(root, f)
            b = os.path.join(str(o2 / "results"), os.path.relpath(a, str(o1 / "results")))
            assert os.path.exists(b) and filecmp.cmp(a, b, shallow=False), a
            n += 1
    assert n > 60 * 5
    for t in ("caller_performance.tsv", "snpcaller_fp_snp_compare.txt"):
        assert (o2 / "results" / "final_tables" / t).read_bytes() == (o1 / "results" / "final_tables" / t).read_bytes()
    # the all-reduced counters = the sum of the rows of ALL VCFs, per truth file
    from quasimodo_amd.multigpu import truth_key
    keys = res["truth_keys"]
    want = np.zeros((len(keys), 3, 256), np.int64)
    for j in jobs2:
        if j.stats.get("roc") is not None:
            want[keys.index(truth_key(j))] += np.asarray(j.stats["roc"]).astype(np.int64)
    assert np.array_equal(res["counters"], want) and want.sum() > 0
    cs = [e for e in golden_cases() if e["family"] == "custom"]
    vcfs = []
    for e in cs:
        vcf, truth, _ = read_case(e)
        p = tmp_path / os.path.basename(e["vcf"])
        p.write_bytes(vcf)
        vcfs.append(str(p))
    snps = tmp_path / "g1_g2.maskrepeat.snps"
    snps.write_bytes(truth)
    workflow.run_vareval(vcfs, str(snps), str(tmp_path / "v1"), engine=engine)
    workflow.run_vareval(vcfs, str(snps), str(tmp_path / "v2"), gpus=2, _backend="gloo", _same_device=True)
    assert (tmp_path / "v2" / "results" / "final_tables" / "snpcall_benchmark.txt").read_bytes() == \
           (tmp_path / "v1" / "results" / "final_tables" / "snpcall_benchmark.txt").read_bytes()


def test_bench_line_contract_on_a_small_batch(qmlib, tmp_path):
    """`python bench.py` as the driver runs it (N = 1), on a reduced batch: ONE JSON line of at most 2 000 characters (what the driver's
    record keeps) with the contract's keys and every figure of record as a FLAT scalar of `roofline` / `cpu_baseline` (the driver's
    parser drops nested objects: VERDICT round 5) -- the whole step over the timed batch and its re-creations, the first-seen rates of
    the shuffled shapes, the allele-extended and per-contig-sorted side workloads, the oracle on one and on all cores, the reference's
    five shell commands per VCF serial and one job per core.  The nested record with the notes goes to a side file.  The numbers of a
    40-VCF batch mean nothing; the shape of the line is the driver's contract."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, QM_BENCH_SHELL_JOBS="4")
    detail = tmp_path / "detail.json"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--vcfs", "40", "--steps", "3", "--warmup", "1", "--cpu-sample", "3000000",
                        "--shell-sample", "1", "--shuffled-vcfs", "40", "--shuffled3-vcfs", "2", "--alleles-vcfs", "40", "--shuffled-alleles-vcfs", "40",
                        "--shuffled4-vcfs", "6", "--multicontig-vcfs", "40", "--alloc-reps", "2", "--detail", str(detail)], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    assert len(lines[0]) <= 2000, len(lines[0])
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["unit"] == "classifications/s" and d["vs_baseline"] is None
    assert abs(d["value"] - 40e6 / (d["ms_per_step"] * 1e-3)) < 1e-4 * d["value"]
    assert d["config"]["vcfs_per_gpu"] == 40 and len(d["config"]["workload"]) <= 120 and d["config"]["collective"] == "none"
    assert d["config"]["vcfs_checked_against_oracle_per_rank"] == 3
    r = d["roofline"]
    assert all(not isinstance(v, (dict, list)) for v in r.values()) and all(not isinstance(v, (dict, list)) for v in d["cpu_baseline"].values())
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert r["algorithmic_bytes_per_launch"] == 40 * (17e6 + 12 * 1e5) and r["kernel_ms"] > 0 and len(r["kernels_build"]) == 16
    assert r["step_frac_min"] <= r["step_frac_median"] <= r["step_frac_max"] and r["alloc_step_ms_min"] <= r["alloc_step_ms_max"]
    assert r["traffic"] is None          # the PMC figure is quoted for the full-size batch only
    for k in ("shuffled_value_first_seen", "shuffled_step_frac", "shuffled3_value_first_seen", "alleles_step_frac", "alleles_classify_frac", "multicontig_value"):
        assert r[k] > 0, k
    assert r["variants_equal_sorted"] is True
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["all_cores_value"] > 0 and c["all_cores_n"] >= 1
    assert c["shell_value"] > 0 and c["shell_all_cores_value"] > 0 and c["shell_all_cores_jobs"] == 4 and "awk" in c
    # the nested record beside the line
    D = json.loads(detail.read_text())
    assert abs(D["value"] - d["value"]) < 1e-4 * d["value"] and D["config"]["oracle_checks"]["roc_rows"] == 3
    assert len(D["roofline"]["alloc_spread"]["step_ms"]) == 3 and "another workload" in D["roofline"]["traffic_note"]
    assert D["cpu_baseline_shell"]["kind"] == "reference-mechanism" and D["cpu_baseline_shell"]["all_cores"]["jobs"] == 4
    for k in ("shuffled_variant", "shuffled_config3_variant", "shuffled_alleles_variant", "shuffled_config4_variant"):
        v = D[k]
        assert v["ms_per_step_unseen"] >= 0 and v["value"] > 0 and v["value_repeated_run"] > 0
        assert abs(v["value"] - v["vcfs"] * v.get("records_per_vcf", 1_000_000) / (v["ms_per_step_unseen"] * 1e-3)) < 1e-6 * v["value"]
        assert v.get("roc_equals_sorted_variant", v.get("equals_sorted_variant")) is True
    sv = D["shuffled_variant"]   # first-seen a third way: the columns written again with other records before every step (and the first ones back for the comparison above)
    assert sv["value_fresh_columns"] > 0 and abs(sv["value_fresh_columns"] - sv["vcfs"] * 1_000_000 / (sv["ms_per_step_fresh_columns"] * 1e-3)) < 1e-6 * sv["value_fresh_columns"]
    assert abs(r["shuffled_value_first_seen"] - sv["value"]) < 1e-4 * sv["value"]
    assert D["alleles_variant"]["equals_oracle_on_vcf0"] is True
    mc = D["multicontig_variant"]
    assert mc["equals_sorted_variant"] is True and mc["contigs"] == 24


def test_the_two_bucket_joins_agree_behind_one_scatter(qmlib):
    """k_join_lean (default) and the hashed join (QM_JOIN=hash) behind the same scatter, on the same 24 shuffled VCFs of configs[2]'s
    shape: one digest over ROC rows, scalars, an index list and a VCF's class bits.  (The choice is read once per process, so every
    variant is a process of its own; with and without the batch's memory.)  Round 3's third join, one bit per key, left the library in
    round 6 (tools/probe/k_join_direct.hip.txt)."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = {}
    for name, extra in (("lean", {}), ("hash", {"QM_JOIN": "hash"}), ("lean, first-seen", {"QM_MEMO": "0"})):
        p = subprocess.run([sys.executable, os.path.join(root, "tools", "join_ab.py"), "24", "1000000"], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, **extra))
        assert p.returncode == 0, (name, p.stderr[-1500:])
        m = re.search(r"digest ([0-9a-f]+)", p.stdout)
        assert m and "'radix': 0" in p.stdout and "'radix_after_overflow': 0" in p.stdout, (name, p.stdout[-500:])
        got[name] = m.group(1)
    assert len(set(got.values())) == 1, got
