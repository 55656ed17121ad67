"""The re-authored rule files -- rules/extract_TP.smk, rules/compare_FP.smk, eval_variant_custom.smk (level 2 of INTEGRATION.md).

No Snakemake exists in the build image, so the files are read by tests/smk_harness.py: sections evaluated, `run:` bodies
compiled and executed with `input` / `output` / `params` bound as Snakemake binds them.  Checked:
  * the DECLARED outputs are the reference's strings (rules/extract_TP.smk:6-11, rules/compare_FP.smk:10-11,
    eval_variant_custom.smk:63-64,82-83 of the reference -- quoted here, the reference is not read at test time);
  * on CPU, the bodies on the golden families with the device stood in for by the oracle (the host logic under test is the
    product's: job building, path derivation, the declared-files check, table and PDF writers);
  * on the GPU (`-m gpu`), the same bodies on the HIP engine: the files are the reference's bytes, the tables equal the
    workflow's own (quasimodo_amd.workflow, which run_benchmark.py uses).
"""
import os
import sys
import types

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, golden_cases, read_case

sys.path.insert(0, os.path.join(ROOT, "tests"))
import smk_harness  # noqa: E402

SAMPLE_REF = {"TM-0-1": "Merlin", "TM-1-1": "Merlin", "TM-1-10": "Merlin", "TM-1-50": "Merlin", "TM-1-0": "TB40E",      # rules/load_config.smk:20-23
              "TA-1-0": "TB40E", "TA-1-1": "AD169", "TA-1-10": "AD169", "TA-1-50": "AD169", "TA-0-1": "AD169"}
CALLERS = ["lofreq", "varscan", "clc", "bcftools", "freebayes", "gatk"]                                                  # eval_variantcall.smk:10


def _hcmv_tree(root):
    """the golden hcmv family where eval_variantcall.smk's cp_vcf / cp_genome_diff put their copies (:62-93)"""
    snp_dir = os.path.join(root, "results", "snp")
    call = os.path.join(snp_dir, "callers")
    fam = os.path.join(GOLDEN, "hcmv", "input")
    for sub in os.listdir(fam):
        dst = os.path.join(snp_dir, "nucmer") if sub == "nucmer" else os.path.join(call, sub)
        os.makedirs(dst, exist_ok=True)
        for f in os.listdir(os.path.join(fam, sub)):
            with open(os.path.join(fam, sub, f), "rb") as a, open(os.path.join(dst, f), "wb") as b:
                b.write(a.read())
    samples = sorted({os.path.basename(e["vcf"]).split(".")[0] for e in golden_cases() if e["family"] == "hcmv"})
    ns = dict(snpcall_dir=call, snp_dir=snp_dir, results_dir=os.path.join(root, "results"), snpcallers=list(CALLERS), threads=4,
              sample_list=samples, sample_refname_dict=dict(SAMPLE_REF), genome_diff_list=["TM", "TA"],
              sample_ref=["%s.%s" % (s, SAMPLE_REF[s]) for s in samples])
    return ns


def _cpu_extract_many(jobs, engine=None, gpus=None, **kw):
    """quasimodo_amd.extract.extract_many with the oracle standing in for the device (tests/sharded_cpu_classify.py)"""
    from sharded_cpu_classify import classify
    for j, st in zip(jobs, classify(jobs, 0)):
        j.stats = st
    return jobs


class CpuEngine:
    """What the rule bodies ask of quasimodo_amd.engine.Engine, answered by the oracle / plain set arithmetic (CPU tests only)."""

    def __init__(self, device=0):
        self._truth = {}

    def close(self):
        pass

    def truth_load(self, pos, ref, alt):
        self._truth[len(self._truth)] = (np.asarray(pos), np.asarray(ref), np.asarray(alt))
        return len(self._truth) - 1

    def truth_release(self, tid):
        self._truth.pop(tid)

    def classify_batch(self, columns, truth_ids, n_bins=256, alleles=False):
        from oracle import qm_oracle as O
        out = []
        for c, t in zip(columns, truth_ids):
            cls, roc, sc = O.classify_columns(*c, *self._truth[t], n_bins=n_bins)
            out.append({"cls": cls, "roc": roc, "scalars": sc})
        return out, None

    def fp_overlap(self, key_sets):
        sets = [set(zip(np.asarray(p).tolist(), np.asarray(r).tolist(), np.asarray(a).tolist())) for p, r, a in key_sets]
        reg = np.zeros(1 << len(sets), np.int64)
        for k in set().union(*sets) if sets else ():
            reg[sum(1 << i for i, s in enumerate(sets) if k in s)] += 1
        return reg


def test_declared_outputs_are_the_references_strings(tmp_path, qmlib):
    ns = _hcmv_tree(str(tmp_path))
    call, res = ns["snpcall_dir"], ns["results_dir"]
    _, r = smk_harness.load(os.path.join(ROOT, "rules", "extract_TP.smk"), ns)
    pairs = [(c, sr.split(".", 1)) for c in CALLERS for sr in ns["sample_ref"]]
    # rules/extract_TP.smk:6-11 of the reference
    ref_filtered = call + "/{snpcaller}/{sample}.{ref}.{snpcaller}.filtered.vcf"
    ref_fp = call + "/{snpcaller}/fp/{sample}.{ref}.{snpcaller}.fp.vcf"
    assert list(r["extractTP"]["output"].filtered) == [ref_filtered.format(snpcaller=c, sample=s, ref=f) for c, (s, f) in pairs]
    assert list(r["extractTP"]["output"].fp) == [ref_fp.format(snpcaller=c, sample=s, ref=f) for c, (s, f) in pairs]
    assert r["extractTP"]["params"].data == "hcmv" and r["extractTP"]["threads"] == 4           # :13,17
    assert sorted(r["extractTP"]["input"].genome_diff) == sorted(ns["snp_dir"] + "/nucmer/%s.maskrepeat.variants.vcf" % m for m in ("TM", "TA"))   # :4
    _, r = smk_harness.load(os.path.join(ROOT, "rules", "compare_FP.smk"), ns)
    assert r["compareFP"]["output"].fp_compare_figure == res + "/final_figures/snpcaller_fp_snp_compare.pdf"    # rules/compare_FP.smk:10
    assert r["compareFP"]["output"].fp_compare_table == res + "/final_tables/snpcaller_fp_snp_compare.txt"      # :11 (commented out there)
    assert r["compareFP"]["params"].fp_compared_snpcallers == ["lofreq", "clc", "varscan", "freebayes"]        # :1
    mixed = [s for s in ns["sample_list"] if not s.endswith(("-1-0", "-0-1"))]
    assert r["compareFP"]["params"].mix_sample == mixed                                                        # :15-16
    assert len(r["compareFP"]["input"].fp) == len(CALLERS) * len(mixed)                                        # :5-8: every caller, mixed samples
    out = tmp_path / "custom"
    cfg = dict(vcfs="a/x.vcf, b/y.vcf", refs="g/r1.fa,g/r2.fa", outpath=str(out), threads=2, labels=None, novenn=True)
    cwd = os.getcwd()
    os.chdir(str(tmp_path))
    try:
        ns2, r = smk_harness.load(os.path.join(ROOT, "eval_variant_custom.smk"), {"config": cfg})
    finally:
        os.chdir(cwd)
    sc = str(out) + "/results/snp/callers"
    assert list(r["extract_TP"]["output"].filtered) == [sc + "/x.filtered.vcf", sc + "/y.filtered.vcf"]       # eval_variant_custom.smk:63
    assert list(r["extract_TP"]["output"].fp) == [sc + "/fp/x.fp.vcf", sc + "/fp/y.fp.vcf"]                   # :64
    assert r["extract_TP"]["params"].data == "custom" and r["extract_TP"]["params"].outdir == sc               # :66-67
    assert r["snp_benchmark"]["output"].snp_benchmark_table == str(out) + "/results/final_tables/snpcall_benchmark.txt"     # :82
    assert r["snp_benchmark"]["output"].snp_benchmark_figure == str(out) + "/results/final_figures/snpcall_benchmark.pdf"   # :83
    # the genome-difference table enters as a path, where the reference's `rule gdiff` puts it (:40); that rule and the target rule
    # are not in the file (they were a lift of out-of-path reference text)
    assert ns2["genome_diff_snps"] == str(out) + "/results/snp/nucmer/r1_r2.maskrepeat.snps"
    assert r["extract_TP"]["input"].genome_diff == ns2["genome_diff_snps"] == r["snp_benchmark"]["input"].genome_diff
    assert sorted(r) == ["extract_TP", "snp_benchmark"]


def _check_hcmv_files(ns):
    cases = {os.path.basename(e["vcf"]): e for e in golden_cases() if e["family"] == "hcmv"}
    n = 0
    for c in CALLERS:
        for sr in ns["sample_ref"]:
            e = cases["%s.%s.vcf" % (sr, c)]
            _, _, exp = read_case(e)
            d = os.path.join(ns["snpcall_dir"], c)
            assert open(os.path.join(d, "%s.%s.filtered.vcf" % (sr, c)), "rb").read() == exp["filtered"]
            assert open(os.path.join(d, "fp", "%s.%s.fp.vcf" % (sr, c)), "rb").read() == exp["fp"]
            if not e["pure"]:
                assert open(os.path.join(d, "tp", "%s.%s.tp.vcf" % (sr, c)), "rb").read() == exp["tp"]
            n += 1
    return n


def _check_fp_compare(ns, oracle):
    table = open(os.path.join(ns["results_dir"], "final_tables", "snpcaller_fp_snp_compare.txt")).read().splitlines()
    cmp_callers = ["lofreq", "clc", "varscan", "freebayes"]
    mixed = [s for s in ns["sample_list"] if not s.endswith(("-1-0", "-0-1"))]
    assert table[0] == "sample\tcallers\tcount" and len(table) == 1 + 15 * len(mixed)
    got = {}
    for ln in table[1:]:
        s, names, cnt = ln.split("\t")
        got.setdefault(s, []).append(int(cnt))
    for s in mixed:
        texts = [open(os.path.join(ns["snpcall_dir"], c, "fp", "%s.%s.%s.fp.vcf" % (s, SAMPLE_REF[s], c)), "rb").read() for c in cmp_callers]
        assert got[s] == [int(x) for x in oracle.fp_overlap_text(texts)[1:]], s      # snpcaller_fp_compare.R:36-47, restated by the oracle
    pdf = open(os.path.join(ns["results_dir"], "final_figures", "snpcaller_fp_snp_compare.pdf"), "rb").read()
    assert pdf.startswith(b"%PDF-1.4") and pdf.rstrip().endswith(b"%%EOF") and pdf.count(b"/Type /Page ") == len(mixed)
    assert ("sample %s" % mixed[0]).encode() in pdf and b"LoFreq & CLC & VarScan2 & FreeBayes" in pdf


def test_hcmv_rule_bodies_on_the_golden_family_cpu(tmp_path, monkeypatch, oracle, qmlib):
    """rules/extract_TP.smk and rules/compare_FP.smk executed as Snakemake would (bodies compiled from the files), the device
    stood in for by the oracle"""
    import quasimodo_amd.engine
    import quasimodo_amd.rules
    monkeypatch.setattr(quasimodo_amd.rules, "extract_many", _cpu_extract_many)
    monkeypatch.setattr(quasimodo_amd.engine, "Engine", CpuEngine)
    ns = _hcmv_tree(str(tmp_path))
    ns1, r = smk_harness.load(os.path.join(ROOT, "rules", "extract_TP.smk"), ns)
    smk_harness.run_rule(ns1, r["extractTP"])
    assert _check_hcmv_files(ns) == 60
    ns2, r2 = smk_harness.load(os.path.join(ROOT, "rules", "compare_FP.smk"), ns)
    assert all(os.path.exists(p) for p in r2["compareFP"]["input"].fp)          # what extractTP left is what compareFP asks for
    smk_harness.run_rule(ns2, r2["compareFP"])
    _check_fp_compare(ns, oracle)
    # a rule that declares other files than the path writes is refused before anything runs
    from quasimodo_amd.rules import RuleError, extract_tp_hcmv
    bad = types.SimpleNamespace(filtered=list(r["extractTP"]["output"].filtered)[:-1], fp=list(r["extractTP"]["output"].fp))
    with pytest.raises(RuleError, match="declares other filtered files"):
        extract_tp_hcmv(r["extractTP"]["input"], bad, r["extractTP"]["params"])


def _custom_tree(tmp_path):
    cs = [e for e in golden_cases() if e["family"] == "custom"]
    vcfs = []
    for e in cs:
        vcf, truth, _ = read_case(e)
        p = tmp_path / "in" / os.path.basename(e["vcf"])
        p.parent.mkdir(exist_ok=True)
        p.write_bytes(vcf)
        vcfs.append(str(p))
    out = tmp_path / "o"
    cfg = dict(vcfs=",".join(vcfs), refs="%s,%s" % (tmp_path / "g1.fa", tmp_path / "g2.fa"), outpath=str(out), threads=1, labels=None, novenn=None)
    return cs, cfg, truth, out


def _run_custom_rules(tmp_path, cs, cfg, truth, out):
    ns, r = smk_harness.load(os.path.join(ROOT, "eval_variant_custom.smk"), {"config": cfg})
    snps = ns["genome_diff_snps"]                         # the genome-difference rule is nucmer's: its output is laid down by hand
    os.makedirs(os.path.dirname(snps), exist_ok=True)
    open(snps, "wb").write(truth)
    smk_harness.run_rule(ns, r["extract_TP"])
    for e in cs:
        _, _, exp = read_case(e)
        d = os.path.join(str(out), "results", "snp", "callers")
        assert open(os.path.join(d, e["caller"] + ".filtered.vcf"), "rb").read() == exp["filtered"]
        assert open(os.path.join(d, "fp", e["caller"] + ".fp.vcf"), "rb").read() == exp["fp"]
        assert open(os.path.join(d, "tp", e["caller"] + ".tp.vcf"), "rb").read() == exp["tp"]
    smk_harness.run_rule(ns, r["snp_benchmark"])
    return r


def _check_custom_table(cs, out, oracle, truth):
    lines = open(os.path.join(str(out), "results", "final_tables", "snpcall_benchmark.txt")).read().splitlines()
    assert lines[0].split("\t") == ["caller", "genomediff", "calleridentify", "TP", "FP", "precision", "recall", "f1"] and len(lines) == 1 + len(cs)
    for e, ln in zip(cs, lines[1:]):
        _, _, exp = read_case(e)
        rc = oracle.count_text(exp["filtered"], truth, custom=True)          # custom_snp_benchmark.R:23-27,47-67, restated by the oracle
        assert ln.split("\t")[:5] == [e["caller"], str(rc["genomediff"]), str(rc["calleridentify"]), str(rc["TP"]), str(rc["FP"])]
    for name in ("snpcall_benchmark.pdf", "snpcall_venn.pdf"):               # the declared figure, and the Venn file R writes unless novenn
        pdf = open(os.path.join(str(out), "results", "final_figures", name), "rb").read()
        assert pdf.startswith(b"%PDF-1.4") and cs[0]["caller"].encode() in pdf
    return lines


def test_custom_rule_bodies_on_the_golden_family_cpu(tmp_path, monkeypatch, oracle, qmlib):
    import quasimodo_amd.engine
    import quasimodo_amd.rules
    monkeypatch.setattr(quasimodo_amd.rules, "extract_many", _cpu_extract_many)
    monkeypatch.setattr(quasimodo_amd.engine, "Engine", CpuEngine)
    cs, cfg, truth, out = _custom_tree(tmp_path)
    _run_custom_rules(tmp_path, cs, cfg, truth, out)
    _check_custom_table(cs, out, oracle, truth)


def test_text_pdf_is_a_pdf(tmp_path):
    """written from the format description: header, objects, a cross-reference table whose offsets point at the objects, trailer"""
    import re
    from quasimodo_amd.pdftext import write_text_pdf
    p = tmp_path / "t.pdf"
    n = write_text_pdf(str(p), [["a (b) \\ c", "x"], ["line %d" % i for i in range(100)]], title="t")
    data = p.read_bytes()
    assert n == 3 and data.count(b"/Type /Page ") == 3            # the 100-line page is split at 65 lines
    assert data.count(b"text-only stand-in for the figure") == 3  # every page says what the file is
    xref = int(re.search(rb"startxref\n(\d+)\n%%EOF", data).group(1))
    assert data[xref:xref + 4] == b"xref"
    count = int(data[xref:].split(b"\n")[1].split()[1])
    offs = [int(x[:10]) for x in data[xref:].split(b"\n")[3:2 + count]]
    for k, o in enumerate(offs):
        assert data[o:].startswith(b"%d 0 obj" % (k + 1))
    assert b"(a \\(b\\) \\\\ c) Tj" in data
    for m in re.finditer(rb"<< /Length (\d+) >>\nstream\n", data):
        assert data[m.end() + int(m.group(1)):].startswith(b"\nendstream")
    assert write_text_pdf(str(p), [["same"]]) == 1 and p.read_bytes() == (write_text_pdf(str(p), [["same"]]), p.read_bytes())[1]   # deterministic


@pytest.mark.gpu
def test_rule_bodies_on_the_engine(engine, oracle, tmp_path):
    """the same rule files on the HIP engine: reference bytes, and the tables quasimodo_amd.workflow writes for run_benchmark.py"""
    from quasimodo_amd import workflow
    ns = _hcmv_tree(str(tmp_path / "h"))
    ns1, r = smk_harness.load(os.path.join(ROOT, "rules", "extract_TP.smk"), ns)
    smk_harness.run_rule(ns1, r["extractTP"])
    assert _check_hcmv_files(ns) == 60
    ns2, r2 = smk_harness.load(os.path.join(ROOT, "rules", "compare_FP.smk"), ns)
    smk_harness.run_rule(ns2, r2["compareFP"])
    _check_fp_compare(ns, oracle)
    # ... byte for byte the overlap table of the workflow (run_benchmark.py hcmv -e variantcall)
    from test_tables_workflow import _build_bundle
    data = tmp_path / "data" / "snp"
    _build_bundle(str(data))
    workflow.run_hcmv_variantcall(str(data), str(tmp_path / "w"), engine=engine)
    a = open(os.path.join(ns["results_dir"], "final_tables", "snpcaller_fp_snp_compare.txt"), "rb").read()
    assert a == open(tmp_path / "w" / "results" / "final_tables" / "snpcaller_fp_snp_compare.txt", "rb").read()
    cs, cfg, truth, out = _custom_tree(tmp_path)
    _run_custom_rules(tmp_path, cs, cfg, truth, out)
    lines = _check_custom_table(cs, out, oracle, truth)
    snps = tmp_path / "g1_g2.maskrepeat.snps"
    snps.write_bytes(truth)
    workflow.run_vareval([str(tmp_path / "in" / os.path.basename(e["vcf"])) for e in cs], str(snps), str(tmp_path / "wv"), engine=engine)
    assert lines == open(tmp_path / "wv" / "results" / "final_tables" / "snpcall_benchmark.txt").read().splitlines()
