"""Allele-extended mode, text side.  Golden = the reference's shell mechanism with the allele regex
widened to `^[ACGT]+$` (tests/golden/make_alleles_golden.py), run with this image's awk / grep.
CPU tests: tokenizer + dictionary + writers around the ORACLE's extended restatement.
GPU test: the product path (extract_many(alleles=True)) byte for byte."""
import json
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(__file__), "golden")
MAN = json.load(open(os.path.join(G, "alleles", "manifest.json")))
CASES = MAN["cases"]


def _read(rel):
    with open(os.path.join(G, rel), "rb") as fh:
        return fh.read()


@pytest.mark.parametrize("c", CASES, ids=lambda c: c["name"])
def test_tokenizer_and_writers_reproduce_the_widened_pipeline(qmlib, oracle, tmp_path, c):
    from quasimodo_amd import vcfio
    d = vcfio.AlleleDict()
    sv = vcfio.scan_vcf(_read(c["vcf"]), alleles=d)
    tk = vcfio.scan_truth(_read(c["truth"]), alleles=d)
    assert sv.n_host == 0 and sv.n_refused == 0 and tk.n_refused == 0
    cls, roc, sc = oracle.classify_columns(*sv.columns, tk.pos, tk.ref, tk.alt, ext=True)
    for sel, k in ((0, "filtered"), (1, "tp"), (2, "fp")):
        out = tmp_path / (k + ".vcf")
        sv.write(str(out), cls, sel)
        assert out.read_bytes() == _read(c["expected"][k]), k
    assert (sc["n_pass"], sc["tp_lines"], sc["fp_lines"]) == tuple(c["lines"][k] for k in ("filtered", "tp", "fp"))
    # single-base rows are tokenised exactly as without the mode
    sv0 = vcfio.scan_vcf(_read(c["vcf"]))
    snp = (sv0.ref < 4) & (sv0.alt < 4)
    for a, b in zip(sv.columns, sv0.columns):
        assert np.array_equal(a[snp], b[snp])
    assert not (sv0.flags[~snp] & 1).any()


def test_allele_codes_and_dictionary(qmlib):
    from quasimodo_amd import vcfio
    d = vcfio.AlleleDict()
    assert [d.code(b) for b in (b"A", b"C", b"G", b"T")] == [0, 1, 2, 3]
    assert d.code(b"AC") == (2 << 26) | 0b0100 and d.code(b"CA") == (2 << 26) | 0b0001
    assert d.code(b"ACGTACGTACGTA") >> 26 == 13 and len(d) == 0
    long1, long2 = b"ACGTACGTACGTAC", b"ACGTACGTACGTAG"
    c1, c2 = d.code(long1), d.code(long2)
    assert c1 == 0x40000000 and c2 == 0x40000001 and d.code(long1) == c1 and len(d) == 2
    for a in (b"", b"N", b"AN", b"a", b"A,C", b"<DEL>", b".", b"*", b"AC GT"):
        assert d.code(a) == -1
    for a in (b"A", b"TG", b"ACGTACGTACGTA", long1, long2):
        assert d.spell(d.code(a)) == a
    with pytest.raises(ValueError):
        d.spell(-1)
    with pytest.raises(ValueError):
        d.spell(0x40000005)
    # concurrent interning from the tokenizer's threads gives one id per string
    import threading
    words = [("ACGTACGTACGTAC" + "ACGT"[i % 4] * (1 + i % 7)).encode() for i in range(64)]
    got = [[None] * 64 for _ in range(8)]

    def work(t):
        for i, w in enumerate(words):
            got[t][i] = d.code(w)
    th = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert all(g == got[0] for g in got) and len(set(got[0])) == len(set(words))


def test_alleles_mode_refuses_show_snps_truth(qmlib):
    from quasimodo_amd import vcfio
    with pytest.raises(ValueError):
        vcfio.scan_truth(b"1\tA\tG\t1\n", custom=True, alleles=vcfio.AlleleDict())


@pytest.mark.gpu
def test_extract_many_alleles_bytes(engine, tmp_path):
    """product path end to end: tokenizer -> GPU (k_classify<false,true>) -> writers"""
    from quasimodo_amd.extract import Job, extract_many
    jobs = []
    for c in CASES:
        d = tmp_path / c["name"]
        (d / "fp").mkdir(parents=True)
        base = "XX-1-10.R." + c["name"]
        (d / (base + ".vcf")).write_bytes(_read(c["vcf"]))
        (d / "truth.vcf").write_bytes(_read(c["truth"]))
        jobs.append(Job(str(d / (base + ".vcf")), str(d / "truth.vcf"), "hcmv"))
    extract_many(jobs, engine=engine, alleles=True)
    for j, c in zip(jobs, CASES):
        for k, p in (("filtered", j.filtered_out), ("tp", j.tp_out), ("fp", j.fp_out)):
            assert open(p, "rb").read() == _read(c["expected"][k]), (c["name"], k)
        assert (j.stats["n_pass"], j.stats["tp_lines"], j.stats["fp_lines"]) == tuple(c["lines"][k] for k in ("filtered", "tp", "fp"))
    # without the mode the same inputs give the reference's single-base answer: no indel line is kept
    extract_many(jobs, engine=engine, alleles=False)
    for j in jobs:
        for ln in open(j.filtered_out, "rb"):
            if not ln.startswith(b"#"):
                f = ln.split(b"\t")
                assert len(f[3]) == 1 and len(f[4]) == 1
