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
    jobs, res = extract_many_sharded(jobs, 2, backend="gloo", body="sharded_cpu_classify:body", timeout=300)
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
    # ONE collective per rank and run: the all-reduce.  The rows came through the ranks' result files (no gather, no barrier)
    assert len(res["collectives"]) == 2
    for col in res["collectives"]:
        assert col["backend"] == "gloo" and col["world"] == 2
        assert col["ops_after"] - col["ops_before"] == 1 and col["ops_at_exit"] == 1, col


def test_sharded_extract_reports_a_failing_rank(tmp_path, monkeypatch, qmlib):
    from quasimodo_amd.extract import Job
    from quasimodo_amd.multigpu import extract_many_sharded
    monkeypatch.setenv("PYTHONPATH", os.path.join(ROOT, "tests") + os.pathsep + os.environ.get("PYTHONPATH", ""))
    (tmp_path / "a.vcf").write_bytes(b"#h\nc\t1\t.\tA\tG\t50\n")
    (tmp_path / "b.vcf").write_bytes(b"#h\nc\t1\t.\tA\tG\t50\n")
    jobs = [Job(str(tmp_path / "a.vcf"), str(tmp_path / "missing.vcf"), "hcmv"), Job(str(tmp_path / "b.vcf"), str(tmp_path / "missing.vcf"), "hcmv")]
    with pytest.raises(RuntimeError) as ei:
        extract_many_sharded(jobs, 2, backend="gloo", body="sharded_cpu_classify:body", timeout=300)
    assert "rank" in str(ei.value) and "missing.vcf" in str(ei.value)


def _bundle(root):
    from test_tables_workflow import _build_bundle
    _build_bundle(str(root))


def test_workflows_on_two_ranks_write_the_tables_of_one(tmp_path, monkeypatch, oracle, qmlib):
    """run_hcmv_variantcall / run_vareval with gpus=2 (gloo, the per-rank body injected): VCFs dealt by SAMPLE, so the four
    compared callers of a sample meet on one rank and its FP overlap is computed there; the three tables
    (caller_performance.tsv, snpcaller_fp_snp_compare.txt, snpcall_benchmark.txt) are byte-identical to the one-rank run."""
    from quasimodo_amd import workflow
    monkeypatch.setenv("PYTHONPATH", os.path.join(ROOT, "tests") + os.pathsep + os.environ.get("PYTHONPATH", ""))
    data = tmp_path / "data" / "snp"
    _bundle(data)
    outs = {}
    for world in (1, 2):
        out = tmp_path / ("out%d" % world)
        jobs = workflow.run_hcmv_variantcall(str(data), str(out), gpus=world, _body="sharded_cpu_classify:body", _backend="gloo")
        assert len(jobs) == 60
        res = workflow.run_hcmv_variantcall.last_result
        outs[world] = {n: (out / "results" / "final_tables" / n).read_bytes() for n in ("caller_performance.tsv", "snpcaller_fp_snp_compare.txt")}
        # whole samples per rank, every rank busy, every sample's overlap from the rank that holds it
        for r, sh in enumerate(res["shards"]):
            assert len(sh) > 0 or world == 1
            smp = {os.path.basename(jobs[i].vcf_file).split(".")[0] for i in sh}
            for other in res["shards"][r + 1:]:
                assert not smp & {os.path.basename(jobs[i].vcf_file).split(".")[0] for i in other}
        assert sorted(s for e in res["extras"] for s in e["overlap"]) == sorted(s for s in workflow.SAMPLE_REF if not s.endswith(("-1-0", "-0-1")))
    assert outs[1] == outs[2] and len(outs[1]["caller_performance.tsv"].splitlines()) == 61
    assert len(outs[1]["snpcaller_fp_snp_compare.txt"].splitlines()) == 1 + 6 * 15
    # vareval
    cs = [e for e in golden_cases() if e["family"] == "custom"]
    vcfs = []
    for e in cs:
        vcf, truth, _ = read_case(e)
        p = tmp_path / os.path.basename(e["vcf"])
        p.write_bytes(vcf)
        vcfs.append(str(p))
    snps = tmp_path / "g1_g2.maskrepeat.snps"
    snps.write_bytes(truth)
    t = {}
    for world in (1, 2):
        workflow.run_vareval(vcfs, str(snps), str(tmp_path / ("v%d" % world)), gpus=world, _body="sharded_cpu_classify:body", _backend="gloo")
        t[world] = (tmp_path / ("v%d" % world) / "results" / "final_tables" / "snpcall_benchmark.txt").read_bytes()
    assert t[1] == t[2] and len(t[1].splitlines()) == 1 + len(cs)


def test_plan_shards_keeps_groups_together(tmp_path):
    from quasimodo_amd.extract import Job
    from quasimodo_amd.multigpu import plan_shards
    jobs = []
    for i, n in enumerate([10, 2000, 30, 400, 5, 60, 7000, 80]):
        p = tmp_path / ("v%d.vcf" % i)
        p.write_bytes(b"x" * n)
        jobs.append(Job(str(p), "t"))
    sh = plan_shards(jobs, 3, groups=[[0, 1], [2, 3, 4], [6]])
    assert sorted(i for s in sh for i in s) == list(range(8))
    where = {i: r for r, s in enumerate(sh) for i in s}
    assert where[0] == where[1] and where[2] == where[3] == where[4]
    assert plan_shards(jobs, 3) == __import__("quasimodo_amd.sharding", fromlist=["x"]).lpt_shards([10, 2000, 30, 400, 5, 60, 7000, 80], 3)
    with pytest.raises(ValueError):
        plan_shards(jobs, 2, groups=[[0, 1], [1, 2]])


def test_lpt_shards_properties():
    from quasimodo_amd.sharding import lpt_shards
    rng = np.random.default_rng(5)
    for world in (1, 2, 3, 8):
        n = rng.integers(1, 10**7, size=50)
        sh = lpt_shards(n, world)
        assert sorted(i for s in sh for i in s) == list(range(50)) and all(s == sorted(s) for s in sh)
        loads = [int(n[s].sum()) for s in sh]
        assert max(loads) - min(loads) <= int(n.max())


def test_the_bundle_is_unpacked_when_data_snp_is_absent(tmp_path, monkeypatch, oracle, qmlib):
    """rules/load_config.smk:28-31: `data/snp` missing, `data/snp.tar.gz` there -> unpacked beside it, then the run as ever.
    The tables of the run from the tarball are byte-identical to those of the run from the directory (per-rank body injected:
    product host code, the oracle as the device)."""
    import tarfile
    from quasimodo_amd import workflow
    monkeypatch.setenv("PYTHONPATH", os.path.join(ROOT, "tests") + os.pathsep + os.environ.get("PYTHONPATH", ""))
    explicit = tmp_path / "explicit" / "data" / "snp"
    _bundle(explicit)
    packed = tmp_path / "packed" / "data"
    packed.mkdir(parents=True)
    with tarfile.open(packed / "snp.tar.gz", "w:gz") as tf:
        tf.add(str(explicit), arcname="snp")
    t = {}
    for name, data in (("explicit", explicit), ("packed", packed / "snp")):
        out = tmp_path / ("out_" + name)
        jobs = workflow.run_hcmv_variantcall(str(data), str(out), gpus=1, _body="sharded_cpu_classify:body", _backend="gloo")
        assert len(jobs) == 60
        t[name] = {n: (out / "results" / "final_tables" / n).read_bytes() for n in ("caller_performance.tsv", "snpcaller_fp_snp_compare.txt")}
    assert t["explicit"] == t["packed"] and (packed / "snp" / "vcf" / "clc").is_dir()
    # neither the directory nor the tarball; a tarball that would write outside its directory
    with pytest.raises(workflow.WorkflowError):
        workflow.ensure_bundle(str(tmp_path / "nowhere" / "snp"))
    evil = tmp_path / "evil"
    evil.mkdir()
    (tmp_path / "x.txt").write_text("x")
    with tarfile.open(evil / "snp.tar.gz", "w:gz") as tf:
        tf.add(str(tmp_path / "x.txt"), arcname="../escaped.txt")
    with pytest.raises(Exception):
        workflow.ensure_bundle(str(evil / "snp"))
    assert not (tmp_path / "escaped.txt").exists()


def test_the_bundle_unpacks_on_a_python_without_extraction_filters(tmp_path, monkeypatch, qmlib):
    """The fallback of workflow.ensure_bundle (tarfile.extractall without `filter=`): an archive made with `tar -C data .` carries a
    '.' member that resolves to the directory itself, and links inside the tree are harmless -- the reference's `tar -xzvf`
    (rules/load_config.smk:30) takes both; a member or a link that leaves the directory is refused."""
    import tarfile
    from quasimodo_amd import workflow
    real = tarfile.TarFile.extractall

    def old_python(self, path=".", members=None, **kw):
        if "filter" in kw:
            raise TypeError("extractall() got an unexpected keyword argument 'filter'")
        return real(self, path, members)

    monkeypatch.setattr(tarfile.TarFile, "extractall", old_python)
    src = tmp_path / "src"
    (src / "snp" / "vcf" / "clc").mkdir(parents=True)
    (src / "snp" / "vcf" / "clc" / "TA-1-10.AD169.clc.vcf").write_text("#x\n")
    os.symlink("clc", src / "snp" / "vcf" / "alias")                      # a link that stays inside
    data = tmp_path / "data"
    data.mkdir()
    with tarfile.open(data / "snp.tar.gz", "w:gz") as tf:
        tf.add(str(src), arcname=".")                                     # members '.', './snp', './snp/vcf', ...
    assert workflow.ensure_bundle(str(data / "snp")) == str(data / "snp")
    assert (data / "snp" / "vcf" / "clc" / "TA-1-10.AD169.clc.vcf").read_text() == "#x\n" and os.path.islink(data / "snp" / "vcf" / "alias")
    for kind in ("member", "link"):
        evil = tmp_path / ("evil_" + kind)
        (evil / "s" / "snp").mkdir(parents=True)
        (evil / "data").mkdir()
        (tmp_path / "x.txt").write_text("x")
        with tarfile.open(evil / "data" / "snp.tar.gz", "w:gz") as tf:
            tf.add(str(evil / "s" / "snp"), arcname="snp")
            if kind == "member":
                tf.add(str(tmp_path / "x.txt"), arcname="../escaped.txt")
            else:
                ti = tarfile.TarInfo("snp/out")
                ti.type = tarfile.SYMTYPE
                ti.linkname = "../../.."
                tf.addfile(ti)
        with pytest.raises(workflow.WorkflowError, match="outside"):
            workflow.ensure_bundle(str(evil / "data" / "snp"))
        assert not (evil / "escaped.txt").exists()


def test_vareval_from_the_config_file_alone(tmp_path, monkeypatch, oracle, qmlib):
    """run_benchmark.py:153-166 + rules/load_config_custom.smk:3 + eval_variant_custom.smk:3-34: what the command line leaves out
    comes from config/customize_data.yaml (paths relative to the workflow's directory; command-line paths relative to the
    caller's).  A YAML-only run writes the table of the run with everything on the command line."""
    import yaml
    from quasimodo_amd import workflow
    monkeypatch.setenv("PYTHONPATH", os.path.join(ROOT, "tests") + os.pathsep + os.environ.get("PYTHONPATH", ""))
    wd = tmp_path / "wd"
    cd = tmp_path / "elsewhere"
    (wd / "in").mkdir(parents=True)
    cd.mkdir()
    cs = [e for e in golden_cases() if e["family"] == "custom"]
    names = []
    for e in cs:
        vcf, truth, _ = read_case(e)
        (wd / "in" / os.path.basename(e["vcf"])).write_bytes(vcf)
        names.append("in/" + os.path.basename(e["vcf"]))
    cfg = {"outpath": "out_yaml/", "vcfs": ", ".join(names), "refs": "ref/g1.fa,ref/g2.fa", "labels": None, "novenn": False, "threads": 6}
    (wd / "customize_data.yaml").write_text(yaml.safe_dump(cfg))
    st = workflow.vareval_settings(config=workflow.load_yaml(str(wd / "customize_data.yaml")), cd=str(cd), wd=str(wd))
    assert st["vcfs"] == [str(wd / n) for n in names] and st["outpath"] == str(wd / "out_yaml") and st["labels"] is None
    assert st["refs"] == [str(wd / "ref/g1.fa"), str(wd / "ref/g2.fa")]
    # the command line wins, relative to the caller's directory; labels from the file when the command line has none
    st2 = workflow.vareval_settings(vcfs="a.vcf, b.vcf", outpath="o", config=dict(cfg, labels="x,y"), cd=str(cd), wd=str(wd))
    assert st2["vcfs"] == [str(cd / "a.vcf"), str(cd / "b.vcf")] and st2["outpath"] == str(cd / "o") and st2["labels"] == ["x", "y"]
    assert workflow.vareval_settings(vcfs="a.vcf", outpath="o", labels="k", config=dict(cfg, labels="x"), cd=str(cd), wd=str(wd))["labels"] == ["k"]
    # the reference's two complaints (eval_variant_custom.smk:14-17, 27-28)
    with pytest.raises(workflow.PathNotGiven, match="reference genome files or output directory"):
        workflow.vareval_settings(vcfs="a.vcf", config={"outpath": None}, cd=str(cd), wd=str(wd))
    with pytest.raises(workflow.PathNotGiven, match="VCF files from SNP calling"):
        workflow.vareval_settings(outpath="o", config={"vcfs": None}, cd=str(cd), wd=str(wd))
    # the two runs
    snps_rel = os.path.join("results", "snp", "nucmer", "g1_g2.maskrepeat.snps")
    t = {}
    for name, s in (("yaml", st), ("explicit", workflow.vareval_settings(vcfs=",".join(str(wd / n) for n in names), refs="g1.fa,g2.fa",
                                                                       outpath=str(tmp_path / "out_explicit"), config={}, cd=str(cd), wd=str(wd)))):
        snps = os.path.join(s["outpath"], snps_rel)
        os.makedirs(os.path.dirname(snps))
        open(snps, "wb").write(truth)
        workflow.run_vareval(s["vcfs"], snps, s["outpath"], labels=s["labels"], gpus=1, _body="sharded_cpu_classify:body", _backend="gloo")
        t[name] = open(os.path.join(s["outpath"], "results", "final_tables", "snpcall_benchmark.txt"), "rb").read()
    assert t["yaml"] == t["explicit"] and len(t["yaml"].splitlines()) == 1 + len(cs)
