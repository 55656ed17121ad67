"""Table writers (A6/A6c/A7 outputs) -- CPU -- and the Snakemake-free workflows -- GPU."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_cases, read_case


def test_r_round3_and_formatting(qmlib):
    from quasimodo_amd.tables import performance_row, r_round3, r_str
    assert r_round3(0.12345) == 0.123 and r_round3(0.9996) == 1.0 and r_round3(2 / 3) == 0.667
    assert r_str(0.5) == "0.5" and r_str(1.0) == "1" and r_str(None) == "NA" and r_str(float("nan")) == "NaN"
    # mixed sample, hand-derived: 6 kept lines, TP 2, FP 2, truth rows 4
    row = performance_row({"n_pass": 6, "TP_R": 2, "FP_R": 2, "genomediff": 4, "pure_strain": False})
    assert row == (4, 6, 2, 2, 0.333, 0.5, 0.4)
    # pure strain (caller_performance_compare.R:121-128) and empty VCF (:101-108)
    assert performance_row({"n_pass": 7, "pure_strain": True}) == (0, 7, 0, 7, 0.0, None, None)
    assert performance_row({"n_pass": 0, "TP_R": 0, "FP_R": 0, "genomediff": 9, "pure_strain": False}) == (9, 0, 0, 0, None, None, None)
    # no true positive at all: F1 = 0/0 -> NaN in R
    r = performance_row({"n_pass": 3, "TP_R": 0, "FP_R": 3, "genomediff": 5, "pure_strain": False})
    assert r[4] == 0.0 and r[5] == 0.0 and np.isnan(r[6])


def test_table_files(qmlib, tmp_path):
    from quasimodo_amd.tables import write_caller_performance, write_fp_overlap, write_snpcall_benchmark
    st = {"n_pass": 6, "TP_R": 2, "FP_R": 2, "genomediff": 4, "pure_strain": False}
    p = tmp_path / "caller_performance.tsv"
    write_caller_performance(str(p), [("lofreq", "TA-1-10", st), ("varscan", "TA-1-0", {"n_pass": 3, "pure_strain": True})])
    assert p.read_text().splitlines() == [
        "caller\tmixture\tgenomediff\tcalleridentify\tTP\tFP\tPrecision\tRecall\tF1",
        "LoFreq\tTA-1-10\t4\t6\t2\t2\t0.333\t0.5\t0.4",
        "VarScan2\tTA-1-0\t0\t3\t0\t3\t0\tNA\tNA"]
    p2 = tmp_path / "snpcall_benchmark.txt"
    write_snpcall_benchmark(str(p2), [("lab", st)])
    assert p2.read_text().splitlines()[0] == "caller\tgenomediff\tcalleridentify\tTP\tFP\tprecision\trecall\tf1"
    p3 = tmp_path / "fp.txt"
    write_fp_overlap(str(p3), {"TA-1-1": [0, 3, 2, 1]}, ["lofreq", "clc"])
    assert p3.read_text().splitlines()[1:] == ["TA-1-1\tLoFreq\t3", "TA-1-1\tCLC\t2", "TA-1-1\tLoFreq&CLC\t1"]


def test_weighted_roc_layout(qmlib, tmp_path):
    import gzip
    from quasimodo_amd.tables import write_weighted_roc
    roc = np.zeros((3, 256), np.uint64)
    roc[0, :31] = 4; roc[1, :31] = 6; roc[2, :31] = 3        # 4 TP lines, 6 FP lines, 3 distinct truth keys, all with QUAL 30
    roc[0, :11] = 5; roc[1, :11] = 9; roc[2, :11] = 4        # below 11: one more of each
    p = tmp_path / "weighted_roc.tsv.gz"
    write_weighted_roc(str(p), roc, 8)
    rows = [ln.split("\t") for ln in gzip.open(p, "rt").read().splitlines() if not ln.startswith("#")]
    assert rows[0] == ["30", "3", "6", "4", "5", "0.4000", "0.3750", "0.3871"]
    assert rows[-1][:5] == ["0", "4", "9", "5", "4"] and len(rows) == 31


def test_lines_r_read_table_would_not_split_at_tabs_are_counted_and_written_as_r_would(qmlib, tmp_path, monkeypatch, capsys):
    """A6 / A6c are restated from the R text (unpinned: no R here).  R reads <x>.filtered.vcf and the truth file with
    read.table(sep = "\\t", comment.char = "#") and the default quote = "\\"'" (scripts/caller_performance_compare.R:29-39):
    a '#' in front of the last column cuts the line, a quote swallows tabs and newlines -- a file R cannot parse then counts as
    EMPTY (tryCatch, :37-40,101-108) and every other file is still counted.  Hand-derived fixture: the tokenizer counts the
    KEPT lines holding ' or ", or a '#' in front of the last column (headers and lines the A2 filter drops do not matter to R:
    comment / not in the file); the strict table writers write such a file's row as R's empty vector and warn, strict =
    "refuse" raises before anything is written, QM_LENIENT=1 writes the tab-split counts."""
    from quasimodo_amd.tables import RTableError, check_r_readable, r_hostile_rows, write_caller_performance, write_snpcall_benchmark
    from quasimodo_amd.vcfio import scan_vcf
    text = (b'##INFO=<ID=DP,Number=1,Type=Integer,Description="Raw depth">\n'      # header: a comment to R, quotes and all
            b"#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n"
            b"c\t100\t.\tA\tG\t50\tPASS\tDP=9\n"                             # plain kept line
            b'c\t200\t.\tA\tG\t50\tPASS\tDP=9;NOTE="x"\n'                    # kept, balanced quotes in the last column: R drops the quotes only
            b"c\t300\trs#7\tC\tT\t60\tPASS\tDP=1\n"                          # kept, '#' in column 3: R sees 3 fields -> the whole file fails
            b"c\t400\t.\tC\tT\t70\tit's\tDP=1\n"                             # kept, a lone apostrophe: swallows what follows
            b"c\t450\t.\tC\tT\t70\tPASS\tDP=1;X=#5\n"                        # kept, '#' inside the LAST column: R loses the tail of a column nobody reads
            b"c\t500\t.\tC\tT\t7\tq'#\"\tDP=1\n"                            # QUAL 7: not kept -> not in the filtered file
            b"c\t600\t.\tCA\tT\t90\t'\tDP=1\n")                              # not single-base: not kept
    sv = scan_vcf(text)
    assert int((sv.flags & 1).sum()) == 5 and sv.n_r_hostile == 3 and sv.first_r_hostile_line == 4
    assert scan_vcf(b"c\t1\t.\tA\tG\t50\tPASS\tDP=9 %&$! x\n").n_r_hostile == 0   # the coarse vector test (bytes 0x20..0x27) is not the answer
    assert scan_vcf(b"c#1\t1\t.\tA\tG\t50\tPASS\tDP=9\n").n_r_hostile == 1 and scan_vcf(b"c\t1\t.\tA\tG\t50\tPASS\tDP=9#\n").n_r_hostile == 0
    long = b"c\t1\t.\tA\tG\t50\tPASS\t" + b"D" * 70 + b"'\n"                      # beyond the first 32-byte step of the line index
    assert scan_vcf(long * 3).n_r_hostile == 3
    truth = tmp_path / "t.vcf"
    truth.write_bytes(b"##x=\"y\"\nc\t100\t.\tA\tG\t30\tPASS\tDP=30;ORIG=a'b\nc\t200\t.\tA\tG\t30\tPASS\tDP=30\nc\t300\t.\tA\tG\t30\tPASS\tDP=30;K=#1\n"
                      b"c\t400\tid#\tA\tG\t30\tPASS\tDP=30\n")
    assert r_hostile_rows(str(truth)) == 2
    ok = {"n_pass": 6, "TP_R": 2, "FP_R": 2, "genomediff": 4, "pure_strain": False, "r_hostile": 0}
    bad = dict(ok, r_hostile=3)
    badt = dict(ok, truth_r_hostile=1)
    monkeypatch.delenv("QM_LENIENT", raising=False)
    assert check_r_readable([("a", ok)]) == []
    for st in (bad, badt):
        with pytest.raises(RTableError, match="read.table"):
            write_caller_performance(str(tmp_path / "cp.tsv"), [("lofreq", "TA-1-10", ok), ("varscan", "TA-1-10", st)], strict="refuse")
        with pytest.raises(RTableError):
            write_snpcall_benchmark(str(tmp_path / "sb.txt"), [("lab", st)], strict="refuse")
    assert not (tmp_path / "cp.tsv").exists() and not (tmp_path / "sb.txt").exists()      # refused before anything is written
    # the default: the reference's per-file tryCatch -- the unreadable file is an empty vector, every other row is written
    write_caller_performance(str(tmp_path / "cp.tsv"), [("lofreq", "TA-1-10", ok), ("varscan", "TA-1-10", bad), ("clc", "TA-1-10", badt)])
    assert "varscan/TA-1-10" in capsys.readouterr().err
    assert (tmp_path / "cp.tsv").read_text().splitlines()[1:] == ["LoFreq\tTA-1-10\t4\t6\t2\t2\t0.333\t0.5\t0.4",
                                                                   "VarScan2\tTA-1-10\t4\t0\t0\t0\tNA\tNA\tNA",        # :101-108
                                                                   "CLC\tTA-1-10\t0\t6\t0\t4\t0\tNaN\tNaN"]            # an empty truth vector: :84-99
    assert write_snpcall_benchmark(str(tmp_path / "sb.txt"), [("lab", bad), ("lab2", ok)]) == [("lab", 3, 0)]      # what was substituted is reported
    assert (tmp_path / "sb.txt").read_text().splitlines()[1:] == ["lab\t4\t0\t0\t0\tNA\tNA\tNA", "lab2\t4\t6\t2\t2\t0.333\t0.5\t0.4"]
    # the custom script reads its truth table without a tryCatch (custom_snp_benchmark.R:23-24): R stops, and so does the strict writer
    (tmp_path / "sb2.txt").unlink(missing_ok=True)
    with pytest.raises(RTableError):
        write_snpcall_benchmark(str(tmp_path / "sb2.txt"), [("lab", badt), ("lab2", ok)])
    assert not (tmp_path / "sb2.txt").exists()
    monkeypatch.setenv("QM_LENIENT", "1")
    write_caller_performance(str(tmp_path / "cp.tsv"), [("lofreq", "TA-1-10", bad)])
    assert (tmp_path / "cp.tsv").read_text().splitlines()[1] == "LoFreq\tTA-1-10\t4\t6\t2\t2\t0.333\t0.5\t0.4"
    assert check_r_readable([("x", bad)]) == [("x", 3, 0)]


def test_lpt_shards(qmlib):
    from quasimodo_amd.sharding import lpt_shards
    n = [10, 1, 1, 1, 9, 8, 2, 2]
    sh = lpt_shards(n, 3)
    assert sorted(v for s in sh for v in s) == list(range(8))
    loads = [sum(n[v] for v in s) for s in sh]
    assert max(loads) - min(loads) <= 2
    assert lpt_shards([5, 5], 4) == [[0], [1], [], []]


def _build_bundle(root):
    """Lay the golden hcmv family out like the unpacked data/snp bundle (rules/load_config.smk:28-36)."""
    fam = os.path.join(GOLDEN, "hcmv", "input")
    for caller in os.listdir(fam):
        if caller == "nucmer":
            dst = os.path.join(root, "nucmer")
        else:
            dst = os.path.join(root, "vcf", caller)
        os.makedirs(dst, exist_ok=True)
        for f in os.listdir(os.path.join(fam, caller)):
            with open(os.path.join(fam, caller, f), "rb") as a, open(os.path.join(dst, f), "wb") as b:
                b.write(a.read())


@pytest.mark.gpu
def test_hcmv_variantcall_workflow(engine, oracle, tmp_path):
    """BASELINE config 2: 10 samples x 6 callers through ONE batch; files, table and FP overlap."""
    from quasimodo_amd import workflow
    data = tmp_path / "data" / "snp"
    _build_bundle(str(data))
    out = tmp_path / "out"
    jobs = workflow.run_hcmv_variantcall(str(data), str(out), engine=engine)
    assert len(jobs) == 60
    results = out / "results"
    cases = {os.path.basename(e["vcf"]): e for e in golden_cases() if e["family"] == "hcmv"}
    rows = {}
    for line in (results / "final_tables" / "caller_performance.tsv").read_text().splitlines()[1:]:
        c = line.split("\t")
        rows[(c[0], c[1])] = c
    from quasimodo_amd.tables import CALLER_MAP
    for job in jobs:
        e = cases[os.path.basename(job.vcf_file)]
        vcf, truth, exp = read_case(e)
        assert open(job.filtered_out, "rb").read() == exp["filtered"]
        assert open(job.fp_out, "rb").read() == exp["fp"]
        if not e["pure"]:
            assert open(job.tp_out, "rb").read() == exp["tp"]
        sample, ref, caller = os.path.basename(job.vcf_file).split(".")[:3]
        row = rows[(CALLER_MAP[caller], sample)]
        rc = oracle.count_text(exp["filtered"], truth)
        if e["pure"]:
            assert row[2:6] == ["0", str(rc["calleridentify"]), "0", str(rc["calleridentify"])] and row[6:] == ["0", "NA", "NA"]
        else:
            assert row[2:6] == [str(rc["genomediff"]), str(rc["calleridentify"]), str(rc["TP"]), str(rc["FP"])]
    # the exact-match ROC in RTG's file layout: its score-20 row is the reference's tp/fp split
    import gzip
    e = cases["TA-1-10.AD169.lofreq.vcf"]
    _, _, exp = read_case(e)
    assert not (results / "snp" / "rtg").exists()       # that directory belongs to the reference's rtg rules
    rocf = results / "snp" / "qmvt_roc" / "lofreq" / "TA-1-10.AD169.xsnp" / "exact_roc.tsv.gz"
    r20 = [ln.split("\t") for ln in gzip.open(rocf, "rt").read().splitlines() if ln.startswith("20\t")][0]
    nd = lambda b: sum(1 for ln in b.split(b"\n") if ln and not ln.startswith(b"#"))
    assert int(r20[3]) == nd(exp["tp"]) and int(r20[2]) == nd(exp["fp"])
    # the xindel sweep (allele-extended mode on the rules' own xindel files): its score-20 row against set arithmetic
    # on the two texts -- whole-string (pos, ref, alt) keys, ID '.', QUAL >= 20
    xrocf = results / "snp" / "qmvt_roc" / "lofreq" / "TA-1-10.AD169.xindel" / "exact_roc.tsv.gz"
    xr = [ln.split("\t") for ln in gzip.open(xrocf, "rt").read().splitlines() if not ln.startswith("#")]
    acgt = lambda s: len(s) > 0 and set(s) <= set(b"ACGT")
    rowsof = lambda p: [ln.split(b"\t") for ln in open(p, "rb").read().split(b"\n") if ln and not ln.startswith(b"#")]
    tkeys = {(f[1], f[3], f[4]) for f in rowsof(results / "snp" / "nucmer" / "TA.maskrepeat.xindel.vcf") if acgt(f[3]) and acgt(f[4])}
    crow = [f for f in rowsof(results / "snp" / "callers" / "lofreq" / "TA-1-10.AD169.lofreq.xindel.vcf")
            if acgt(f[3]) and acgt(f[4]) and oracle.awk_ge(f[5] if len(f) > 5 else b"", 20)]
    tp20 = sum(1 for f in crow if (f[1], f[3], f[4]) in tkeys and f[2] == b".")
    got20 = [r for r in xr if r[0] == "20"]
    if crow:
        assert got20 and int(got20[0][3]) == tp20 and int(got20[0][2]) == len(crow) - tp20
    assert len(tkeys) > 0 and "#total baseline variants: %d" % len(tkeys) in gzip.open(xrocf, "rt").read()
    # extract_snp / extract_indel / extract_nucmer_* outputs sit where rules/vis_eval_vcf.smk:25-86 put them
    cdir = results / "snp" / "callers" / "lofreq"
    xs = (cdir / "TA-1-10.AD169.lofreq.xsnp.vcf").read_bytes().split(b"\n")
    assert all(ln.startswith(b"#") or (len(ln.split(b"\t")[3]) == 1 and len(ln.split(b"\t")[4]) == 1) for ln in xs if ln)
    assert (cdir / "TA-1-10.AD169.lofreq.xindel.vcf").exists()
    assert (results / "snp" / "nucmer" / "TA.maskrepeat.xsnp.vcf").exists() and (results / "snp" / "nucmer" / "TA.maskrepeat.xindel.vcf").exists()
    # ... each with its bgzip (BGZF: gzip members with a 'BC' field, closed by the EOF member; zcat gives the plain file)
    eof = bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])
    for plain in (cdir / "TA-1-10.AD169.lofreq.xsnp.vcf", cdir / "TA-1-10.AD169.lofreq.xindel.vcf",
                  results / "snp" / "nucmer" / "TA.maskrepeat.xsnp.vcf", results / "snp" / "nucmer" / "TM.maskrepeat.xindel.vcf"):
        gz = open(str(plain) + ".gz", "rb").read()
        assert gz[:4] == b"\x1f\x8b\x08\x04" and gz[12:16] == b"BC\x02\x00" and gz.endswith(eof)
        assert gzip.decompress(gz) == plain.read_bytes()
        tbi = gzip.decompress(open(str(plain) + ".gz.tbi", "rb").read())      # ... and its `tabix -p vcf` (vis_eval_vcf.smk:37,52,68,83)
        assert tbi[:4] == b"TBI\x01" and tbi[8:12] == (2).to_bytes(4, "little")
    # FP overlap regions against the oracle's restatement of snpcaller_fp_compare.R
    table = (results / "final_tables" / "snpcaller_fp_snp_compare.txt").read_text().splitlines()[1:]
    got = {}
    for line in table:
        s, names, cnt = line.split("\t")
        got[(s, names)] = int(cnt)
    callers = workflow.FP_COMPARED
    for sample in ("TA-1-10", "TM-1-1"):
        texts = [open(str(results / "snp" / "callers" / c / "fp" / ("%s.%s.%s.fp.vcf" % (sample, workflow.SAMPLE_REF[sample], c))), "rb").read()
                 for c in callers]
        reg = oracle.fp_overlap_text(texts)
        for m in range(1, 16):
            names = "&".join(CALLER_MAP[callers[i]] for i in range(4) if m >> i & 1)
            assert got[(sample, names)] == reg[m], (sample, names)


@pytest.mark.gpu
def test_vareval_workflow(engine, oracle, tmp_path):
    from quasimodo_amd import workflow
    cs = [e for e in golden_cases() if e["family"] == "custom"]
    vcfs = []
    for e in cs:
        vcf, truth, exp = read_case(e)
        p = tmp_path / os.path.basename(e["vcf"])
        p.write_bytes(vcf)
        vcfs.append(str(p))
    snps = tmp_path / "g1_g2.maskrepeat.snps"
    snps.write_bytes(truth)
    jobs = workflow.run_vareval(vcfs, str(snps), str(tmp_path / "o"), engine=engine)
    lines = (tmp_path / "o" / "results" / "final_tables" / "snpcall_benchmark.txt").read_text().splitlines()
    assert len(lines) == 1 + len(cs)
    for e, job, line in zip(cs, jobs, lines[1:]):
        _, truth, exp = read_case(e)
        assert open(job.filtered_out, "rb").read() == exp["filtered"]
        assert open(job.tp_out, "rb").read() == exp["tp"] and open(job.fp_out, "rb").read() == exp["fp"]
        rc = oracle.count_text(exp["filtered"], truth, custom=True)
        assert line.split("\t")[1:5] == [str(rc["genomediff"]), str(rc["calleridentify"]), str(rc["TP"]), str(rc["FP"])]
        assert job.filtered_out.endswith("results/snp/callers/%s.filtered.vcf" % e["caller"])


@pytest.mark.gpu
def test_run_benchmark_cli_vareval_from_a_config_file_hcmv_from_the_tarball(tmp_path):
    """The reference's command lines on the HIP engine: `vareval` with nothing but a config file (run_benchmark.py:153-166,
    rules/load_config_custom.smk:3), `hcmv -e variantcall` on a data directory that exists only as snp.tar.gz
    (rules/load_config.smk:28-31); --json says where VCFs that were out of order went."""
    import json
    import subprocess
    import sys
    import tarfile
    import yaml
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cs = [e for e in golden_cases() if e["family"] == "custom"]
    names = []
    for e in cs:
        vcf, truth, exp = read_case(e)
        (tmp_path / os.path.basename(e["vcf"])).write_bytes(vcf)
        names.append(str(tmp_path / os.path.basename(e["vcf"])))
    out = tmp_path / "o"
    snps = out / "results" / "snp" / "nucmer" / "g1_g2.maskrepeat.snps"
    snps.parent.mkdir(parents=True)
    snps.write_bytes(truth)
    cfg = tmp_path / "customize_data.yaml"
    cfg.write_text(yaml.safe_dump({"outpath": str(out), "vcfs": ",".join(names), "refs": "x/g1.fa,x/g2.fa", "labels": None}))
    js = tmp_path / "run.json"
    r = subprocess.run([sys.executable, os.path.join(root, "run_benchmark.py"), "vareval", "--config", str(cfg), "--json", str(js)],
                       cwd=str(tmp_path), capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = (out / "results" / "final_tables" / "snpcall_benchmark.txt").read_text().splitlines()
    assert len(lines) == 1 + len(cs)
    doc = json.load(open(js))
    assert doc["command"] == "vareval" and len(doc["rows"]) == len(cs) and doc["unsorted_paths"]["radix_after_overflow"] == 0
    for e, row in zip(cs, doc["rows"]):
        _, _, exp = read_case(e)
        assert open(row["fp"], "rb").read() == exp["fp"] and open(row["tp"], "rb").read() == exp["tp"]
    # without a config file and without arguments: the reference's complaint, a non-zero exit
    r = subprocess.run([sys.executable, os.path.join(root, "run_benchmark.py"), "vareval", "-o", "o2"], cwd=str(tmp_path), capture_output=True, text=True)
    assert r.returncode != 0 and "VCF files from SNP calling are not specified" in r.stdout
    # hcmv from the tarball
    data = tmp_path / "data"
    _build_bundle(str(tmp_path / "stage" / "snp"))
    data.mkdir()
    with tarfile.open(data / "snp.tar.gz", "w:gz") as tf:
        tf.add(str(tmp_path / "stage" / "snp"), arcname="snp")
    r = subprocess.run([sys.executable, os.path.join(root, "run_benchmark.py"), "hcmv", "-e", "variantcall", "--data", str(data / "snp"), "-o", str(tmp_path / "h"),
                        "--json", str(tmp_path / "h.json")], cwd=str(tmp_path), capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert len((tmp_path / "h" / "results" / "final_tables" / "caller_performance.tsv").read_text().splitlines()) == 61
    assert len(json.load(open(tmp_path / "h.json"))["rows"]) == 60


def test_module_cli_split_and_argument_errors(tmp_path):
    """`python -m quasimodo_amd` -- the parts that need no GPU"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    probe = os.path.join(root, "tests", "golden", "split", "input", "probe.vcf")
    out = tmp_path / "x.vcf"
    env = dict(os.environ, QM_AWK_FLAVOUR="mawk-literal")
    r = subprocess.run([sys.executable, "-m", "quasimodo_amd", "split", probe, str(out), "xsnp"], cwd=root, env=env, capture_output=True, text=True)
    assert r.returncode == 0 and out.read_bytes() == open(os.path.join(root, "tests", "golden", "split", "expected", "probe.xsnp.vcf"), "rb").read()
    r = subprocess.run([sys.executable, "-m", "quasimodo_amd", "extract", "a.vcf"], cwd=root, capture_output=True, text=True)
    assert r.returncode == 2 and "--truth" in r.stderr


@pytest.mark.gpu
def test_module_cli_extract(tmp_path):
    """one GPU batch from the command line; files and counts equal the golden ones"""
    import json
    import shutil
    import subprocess
    import sys
    from conftest import golden_cases, read_case
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cases = [c for c in golden_cases() if c["family"] == "hcmv" and not c["pure"]][:4]
    g = os.path.join(root, "tests", "golden", "hcmv")
    vcfs = []
    for c in cases:
        d = tmp_path / c["caller"]
        (d / "fp").mkdir(parents=True, exist_ok=True)
        dst = d / os.path.basename(c["vcf"])
        shutil.copyfile(os.path.join(g, c["vcf"]), dst)
        vcfs.append(str(dst))
    same_truth = [c for c in cases if c["truth"] == cases[0]["truth"]]
    sel = [v for v, c in zip(vcfs, cases) if c in same_truth]
    js = tmp_path / "rows.json"
    r = subprocess.run([sys.executable, "-m", "quasimodo_amd", "extract", "--truth", os.path.join(g, cases[0]["truth"]), "--json", str(js)] + sel,
                       cwd=root, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    doc = json.load(open(js))
    rows = doc["rows"]
    # where VCFs that were out of order went (qm_path_stats_total): every counter there, nothing on the radix sort after an overflow
    assert set(doc["unsorted_paths"]) >= {"unsorted", "bucket_direct", "radix", "radix_after_overflow"} and doc["unsorted_paths"]["radix_after_overflow"] == 0
    assert doc["unsorted_paths"]["unsorted"] == sum(1 for row in rows if not row["sorted"])
    assert len(rows) == len(sel) and r.stdout.count("\n") == len(sel) + 1
    for row, c in zip(rows, same_truth):
        _, _, exp = read_case(c)
        for k in ("filtered", "tp", "fp"):
            assert open(row[k], "rb").read() == exp[k], (c["vcf"], k)
        nd = lambda b: sum(1 for ln in b.split(b"\n") if ln and not ln.startswith(b"#"))
        assert (row["n_pass"], row["tp_lines"], row["fp_lines"]) == (nd(exp["filtered"]), nd(exp["tp"]), nd(exp["fp"]))
