"""N > 1 path on CPU: world_size-2 gloo through the PRODUCT's multi-GPU entry point
(quasimodo_amd.multigpu.extract_many_sharded): the parent starts one process per rank, the ranks deal the VCFs by
LPT, each classifies its share and writes its files, the confusion counters of every truth set go through the
path's single all-reduce and the per-VCF rows are gathered on rank 0.  There is no GPU here, so the per-rank
classification is injected (tests/sharded_cpu_classify.py: product host code + the oracle as the device); the
sharding, the collective, the gather and the process handling under test are the product's own."""
import os

import numpy as np
import pytest

from conftest import ROOT, case_id, golden_cases, read_case

CASES = [c for c in golden_cases() if c["family"] in ("hcmv", "quirks", "custom", "edge")]


def _jobs(tmp_path):
    from quasimodo_amd.extract import Job
    jobs, exps = [], []
    for e in CASES:
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
    return jobs, exps


def test_sharded_extract_gloo_world2(tmp_path, monkeypatch, oracle, qmlib):
    from quasimodo_amd.multigpu import extract_many_sharded, job_weights, truth_key
    from quasimodo_amd.sharding import lpt_shards
    monkeypatch.setenv("PYTHONPATH", os.path.join(ROOT, "tests") + os.pathsep + os.environ.get("PYTHONPATH", ""))
    jobs, exps = _jobs(tmp_path)
    jobs, res = extract_many_sharded(jobs, 2, backend="gloo", classify="sharded_cpu_classify:classify", timeout=300)
    # every rank wrote its own files: the reference's bytes, whichever rank a VCF went to
    for job, (e, exp) in zip(jobs, exps):
        assert open(job.filtered_out, "rb").read() == exp["filtered"], case_id(e)
        assert open(job.fp_out, "rb").read() == exp["fp"], case_id(e)
        if not e["pure"]:
            assert open(job.tp_out, "rb").read() == exp["tp"], case_id(e)
    # LPT on the file sizes, both ranks busy, rows gathered back in job order
    shards = lpt_shards(job_weights(jobs), 2)
    assert res["shards"] == shards and all(len(s) > 0 for s in shards)
    w = np.array(job_weights(jobs))
    loads = [int(w[s].sum()) for s in shards]
    assert abs(loads[0] - loads[1]) <= int(w.max())          # LPT: the loads differ by at most one job
    for r, s in enumerate(shards):
        for i in s:
            assert jobs[i].stats["rank"] == r and jobs[i].stats["device"] == r
    # the single all-reduce: per truth set, the sum over ALL VCFs of their ROC rows
    keys = res["truth_keys"]
    want = np.zeros((len(keys), 3, 256), np.int64)
    for j in jobs:
        if j.stats["roc"] is not None:
            want[keys.index(truth_key(j))] += j.stats["roc"].astype(np.int64)
    assert np.array_equal(res["counters"], want) and want.sum() > 0
    assert len(keys) == len({truth_key(j) for j, (e, _) in zip(jobs, exps) if not e["pure"]})


def test_sharded_extract_reports_a_failing_rank(tmp_path, monkeypatch, qmlib):
    from quasimodo_amd.extract import Job
    from quasimodo_amd.multigpu import extract_many_sharded
    monkeypatch.setenv("PYTHONPATH", os.path.join(ROOT, "tests") + os.pathsep + os.environ.get("PYTHONPATH", ""))
    (tmp_path / "a.vcf").write_bytes(b"#h\nc\t1\t.\tA\tG\t50\n")
    (tmp_path / "b.vcf").write_bytes(b"#h\nc\t1\t.\tA\tG\t50\n")
    jobs = [Job(str(tmp_path / "a.vcf"), str(tmp_path / "missing.vcf"), "hcmv"), Job(str(tmp_path / "b.vcf"), str(tmp_path / "missing.vcf"), "hcmv")]
    with pytest.raises(RuntimeError) as ei:
        extract_many_sharded(jobs, 2, backend="gloo", classify="sharded_cpu_classify:classify", timeout=300)
    assert "rank" in str(ei.value) and "missing.vcf" in str(ei.value)


def test_lpt_shards_properties():
    from quasimodo_amd.sharding import lpt_shards
    rng = np.random.default_rng(5)
    for world in (1, 2, 3, 8):
        n = rng.integers(1, 10**7, size=50)
        sh = lpt_shards(n, world)
        assert sorted(i for s in sh for i in s) == list(range(50)) and all(s == sorted(s) for s in sh)
        loads = [int(n[s].sum()) for s in sh]
        assert max(loads) - min(loads) <= int(n.max())
