"""Host text side of libqmvt.so (tokenizer / packer / writers) against the oracle and the
golden bytes.  CPU only: no kernel is launched here; where a classification is needed to
exercise the writers the ORACLE provides it (checker role only)."""
import os

import numpy as np
import pytest

from conftest import case_id, golden_cases, read_case

CASES = golden_cases()


def host_classify(oracle, sv, truth, custom=False, alleles=False):
    """What extract_many does between the scan and the writers, with the ORACLE's column-level restatement standing
    in for the device (checker role only): patterns -> host path (decisions into the flags) -> classification of the
    columns -> exchange of the text keys for R's unique-key counts."""
    from quasimodo_amd import scan_truth
    from quasimodo_amd.vcfio import Patterns
    tk = scan_truth(truth, custom=custom)
    pt = Patterns(truth, custom=custom, alleles=alleles)
    ex = sv.hostpath(pt) if (sv.n_host or sv.n_nokey_kept or pt.needs_full_hostpath) else None
    pt.close()
    cls, roc, sc = oracle.classify_columns(*sv.columns, tk.pos, tk.ref, tk.alt)
    if ex is not None:
        sc["FP_R"] += ex["fp_r"] - ex["device_nokey_keys"]
        sc["TP_R"] += ex["tp_r"]
    return tk, cls, roc, sc


def _data_lines(text):
    lines = text.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    return [ln for ln in lines if not ln.startswith(b"#")]


@pytest.mark.parametrize("e", CASES, ids=case_id)
def test_tokenizer_flags_match_awk_filter(qmlib, oracle, e):
    from quasimodo_amd import scan_vcf
    vcf, _, exp = read_case(e)
    sv = scan_vcf(vcf)
    lines = _data_lines(vcf)
    assert sv.n_records == len(lines)
    want = np.array([oracle.caller_filter(ln) for ln in lines], dtype=bool)
    assert np.array_equal((sv.flags & 1).astype(bool), want)
    # ID == "." bit
    ids = np.array([(ln.split(b"\t") + [b"", b"", b""])[2] == b"." for ln in lines], dtype=bool)
    assert np.array_equal(((sv.flags >> 1) & 1).astype(bool), ids)


@pytest.mark.parametrize("e", CASES, ids=case_id)
def test_filtered_writer_bytes(qmlib, tmp_path, e):
    from quasimodo_amd import scan_vcf
    vcf, _, exp = read_case(e)
    sv = scan_vcf(vcf)
    out = tmp_path / "f.vcf"
    sv.write(str(out), (sv.flags & 1).astype(np.uint8), 0)
    assert out.read_bytes() == exp["filtered"]


@pytest.mark.parametrize("e", [c for c in CASES if not c["pure"]], ids=case_id)
def test_pack_classify_write_roundtrip(qmlib, oracle, tmp_path, e):
    """text -> columns (product tokenizer) + host path for the lines the columns cannot describe -> classes (oracle
    as checker) -> files == reference bytes.  Pins the packing (POS/allele codes, effective QUAL, truth keys) and
    the product's own `fgrep -w` (the `quirks` family: SURVEY Q10) to the golden vectors on CPU."""
    from quasimodo_amd import scan_vcf
    vcf, truth, exp = read_case(e)
    sv = scan_vcf(vcf)
    assert sv.n_refused == 0 and (sv.n_host == 0 or e["family"] in ("quirks", "utf8"))   # (utf8: valid UTF-8 single-base lines are decided from their text)
    tk, cls, roc, sc = host_classify(oracle, sv, truth, custom=e["mode"] == "custom")
    assert tk.n_refused == 0
    for sel, kind in ((0, "filtered"), (1, "tp"), (2, "fp")):
        out = tmp_path / (kind + ".vcf")
        sv.write(str(out), cls, sel)
        assert out.read_bytes() == exp[kind], kind
    # effective QUAL invariant: floor(qual) >= 20 <=> awk kept the record (given single-base alleles)
    snp = (sv.ref < 4) & (sv.alt < 4)
    assert np.array_equal(snp & (np.floor(sv.qual) >= 20), (sv.flags & 1).astype(bool))
    assert int(roc[0, 20]) == sc["tp_lines"] and int(roc[1, 20]) == sc["fp_lines"]
    # R-path counts (A6) from columns == R restatement on the text
    rc = oracle.count_text(exp["filtered"], truth, custom=e["mode"] == "custom")
    assert rc["calleridentify"] == sc["n_pass"]
    assert rc["TP"] == sc["TP_R"] and rc["FP"] == sc["FP_R"]
    assert rc["genomediff"] == tk.genomediff


def test_noncanonical_lines_are_flagged(qmlib):
    from quasimodo_amd import scan_vcf
    e = [c for c in CASES if c["family"] == "quirks" and c["mode"] == "hcmv"][0]
    vcf, _, _ = read_case(e)
    sv = scan_vcf(vcf)
    lines = vcf.split(b"\n")
    flagged = [lines[i] for i in range(sv.n_lines) if sv.line_kind[i] == 2]
    # POS spelled non-canonically (2) + a "\t.\t" behind the ALT column that a pattern could sit on (3; not the line
    # whose field after it reads "CT": no pattern ends inside a word)
    assert len(flagged) == 5 and sv.n_host == 5 and sv.n_refused == 0 and sv.n_nokey_kept == 2
    assert any(b"\t1-1000\t" in ln for ln in flagged) and any(b"\t01000\t" in ln for ln in flagged)
    assert sum(b"\t30\t.\tA\tC" in ln for ln in flagged) == 2 and any(b"\t77\t.\tG\tT\tz" in ln for ln in flagged)


@pytest.mark.gpu
def test_strict_mode_refuses_only_locale_dependent_lines(qmlib, tmp_path):
    """extract_many refuses (before any kernel runs) when a kept line holds a NUL or an INVALID UTF-8 sequence -- the one kind of
    input left whose reference answer the engine does not reproduce (grep says "binary file matches" under the locale Python
    exports); QM_LENIENT / strict=False classifies such lines by their columns.  Valid UTF-8 is text (round 6: tests/golden/utf8/)."""
    import quasimodo_amd as q
    from quasimodo_amd.extract import Job
    d = tmp_path / "q"
    d.mkdir()
    (d / "QK-1-10.R.q.vcf").write_bytes(b"#h\nc\t5\t.\tA\tG\t50\tPASS\tname=\xe9\n")          # Latin-1 e-acute: not UTF-8
    (tmp_path / "t.vcf").write_bytes(b"c\t5\t.\tA\tG\n")
    with pytest.raises(q.QmvtError) as ei:
        q.extract_many([Job(str(d / "QK-1-10.R.q.vcf"), str(tmp_path / "t.vcf"), "hcmv")], strict=True)
    assert ei.value.code == -8 and "line 2" in str(ei.value)
    job = q.extract_many([Job(str(d / "QK-1-10.R.q.vcf"), str(tmp_path / "t.vcf"), "hcmv")], strict=False)[0]
    assert job.stats["tp_lines"] == 1 and open(job.tp_out, "rb").read().count(b"\n") == 2
    (d / "QK-1-10.R.q.vcf").write_bytes(b"#h\nc\t5\t.\tA\tG\t50\tPASS\tname=\xc3\xa9\n")      # the same letter in UTF-8: accepted, a TP line
    job = q.extract_many([Job(str(d / "QK-1-10.R.q.vcf"), str(tmp_path / "t.vcf"), "hcmv")], strict=True)[0]
    assert job.stats["tp_lines"] == 1 and open(job.tp_out, "rb").read() == b"#h\nc\t5\t.\tA\tG\t50\tPASS\tname=\xc3\xa9\n"


def test_non_utf8_kept_line_flagged(qmlib, oracle):
    from quasimodo_amd import scan_vcf
    vcf = b"c\t5\t.\tA\tG\t50\tPASS\tname=\xe9\nc\t6\t.\tA\tG\t5\tPASS\tname=\xe9\n#c\t7\t.\tA\tG\t50\t\xc3\n"
    sv = scan_vcf(vcf)
    assert list(sv.line_kind) == [5, 0, 6] and sv.n_refused == 2 and sv.first_refused_line == 1   # only KEPT lines matter
    with pytest.raises(ValueError):
        oracle.extract_text(vcf, b"", False, False)
    for bad in (b"c\t5\t.\tA\tG\t50\tPASS\tx\x00y\n", b"c\t5\t.\tA\tG\t50\tPASS\t\xc0\xaf\n", b"c\t5\t.\tA\tG\t50\tPASS\t\xed\xa0\x80\n",
                b"c\t5\t.\tA\tG\t50\tPASS\t\xf4\x90\x80\x80\n", b"c\t5\t.\tA\tG\t50\tPASS\t\xe2\x82\n"):   # NUL, overlong, surrogate, > U+10FFFF, truncated
        assert scan_vcf(bad).n_refused == 1, bad
        with pytest.raises(ValueError):
            oracle.extract_text(bad, b"", False, False)
    ok = "c\t5\t.\tA\tG\t50\tPASS\tname=\u00e9 \u65e5\u672c \U0001d7d1\n".encode("utf-8")          # valid: kept, decided on the host path from its text
    sv = scan_vcf(ok)
    assert sv.n_refused == 0 and list(sv.line_kind) == [2] and sv.n_host == 1
    oracle.extract_text(ok, b"", False, False)


def test_pure_strain_paths_and_copy(qmlib, tmp_path):
    """A5p: fp is a copy of filtered, no tp/, truth never opened (extract_TP_FP_SNPs.py:33-36)."""
    import quasimodo_amd as q
    e = [c for c in CASES if c["family"] == "edge" and c["vcf"].endswith("ED-1-0.R.e.vcf")][0]
    vcf, _, exp = read_case(e)
    d = tmp_path / "e"
    d.mkdir()
    p = d / "ED-1-0.R.e.vcf"
    p.write_bytes(vcf)
    job = q.extract_tp_fp_snp(str(p), str(tmp_path / "does-not-exist.vcf"))
    assert open(job.filtered_out, "rb").read() == exp["filtered"]
    assert open(job.fp_out, "rb").read() == exp["fp"]
    assert job.filtered_out == str(d / "ED-1-0.R.e.filtered.vcf")
    assert job.fp_out == str(d / "fp" / "ED-1-0.R.e.fp.vcf")
    assert not (d / "tp").exists()
    assert job.stats["pure_strain"] and job.stats["fp_lines"] == job.stats["n_pass"]


def test_custom_mode_paths(qmlib):
    from quasimodo_amd.extract import Job, _paths
    j = Job("/x/in/a.vcf", "/x/t.snps", "custom", "/out/callers", "lab")
    _paths(j)
    assert (j.filtered_out, j.fp_out, j.tp_out) == ("/out/callers/lab.filtered.vcf", "/out/callers/fp/lab.fp.vcf",
                                                     "/out/callers/tp/lab.tp.vcf")
    h = Job("/x/lofreq/TA-1-10.AD169.lofreq.vcf", "/x/t.vcf", "hcmv", "ignored", "ignored")
    _paths(h)
    assert h.filtered_out == "/x/lofreq/TA-1-10.AD169.lofreq.filtered.vcf"
    assert h.fp_out == "/x/lofreq/fp/TA-1-10.AD169.lofreq.fp.vcf"
    assert h.tp_out == "/x/lofreq/tp/TA-1-10.AD169.lofreq.tp.vcf"


def test_effective_qual_rounds_down(qmlib):
    from quasimodo_amd import scan_vcf
    sv = scan_vcf(b"c\t1\t.\tA\tG\t19.99999999\nc\t2\t.\tA\tG\t20.00000001\nc\t3\t.\tA\tG\t.\nc\t4\t.\tA\tG\tPASS\n"
                  b"c\t5\t.\tA\tG\t1,2\nc\t6\t.\tA\tG\t16777217.5\n")
    assert np.floor(sv.qual[0]) == 19 and np.floor(sv.qual[1]) == 20
    assert np.isposinf(sv.qual[2]) and np.isposinf(sv.qual[3]) and np.isneginf(sv.qual[4])
    assert sv.qual[5] <= 16777217.5
    assert list(sv.flags & 1) == [0, 1, 1, 1, 0, 1]


def test_truth_scan_modes(qmlib):
    from quasimodo_amd import scan_truth
    t = b"##x\n#CHROM\tPOS\tID\tREF\tALT\nc\t10\t.\tA\tG\nc\t11\t.\tA\tG,T\nc\t12\t.\tAC\tA\nc\t013\t.\tA\tC\nc\t14\t.\tc\tt\n"
    tk = scan_truth(t)
    assert list(tk.pos) == [10] and tk.genomediff == 2 and tk.n_never == 1 and tk.n_refused == 0 and tk.n_comment == 0
    tk = scan_truth(t + b"#c\t15\t.\tA\tG\n")        # awk makes a pattern of a '#' row, R does not read it
    assert list(tk.pos) == [10] and tk.genomediff == 2 and tk.n_comment == 1
    s = b"10\tA\tG\tx\n11\tN\tG\tx\n12\t.\tG\tx\n\n13\tC\tT\n"
    tk = scan_truth(s, custom=True)
    assert list(tk.pos) == [10, 13] and list(tk.ref) == [0, 1] and list(tk.alt) == [2, 3]
    assert tk.genomediff == 3 and tk.n_never == 2     # 'N' row and the empty line's pattern


def _random_case(rng):
    """Hostile VCF / truth texts, ASCII and valid UTF-8 (same generator family as make_golden.py --fuzz)."""
    fields = ["A", "C", "G", "T", ".", "N", "a", "AC", "A,C", "", "PASS", "20", "19", "30", "5", "1e2", "0x14", "rs1", "-",
              "5_", "x.y", " 20", "20 ", "2e-400", "7\r", "100", "10", "1", "19.9999999", "20.0000001", "1e400", "+20", "nan",
              # (round 6) letters, a digit of another script, a four-byte digit, an arrow, a no-break space: alone and abutting what a pattern ends or begins with
              "\u00e9", "\u00e95", "5\u00e9", "\u21925", "5\u2192", "\u06633", "\U0001d7d17", "\u00a02", "A\u00e9", "C\u2192", "\u65e5\u672c", "\u00e9A"]
    npos = int(rng.integers(1, 40))

    def line(ncol_max=10):
        nc = int(rng.integers(1, ncol_max + 1))
        cols = []
        for c in range(nc):
            u = rng.random()
            if c == 1 and u < 0.985:
                cols.append(str(int(rng.integers(1, npos + 1))))
            elif c == 2 and u < 0.7:
                cols.append(".")
            elif c in (3, 4) and u < 0.85:
                cols.append("ACGT"[int(rng.integers(0, 4))])
            elif c == 5 and u < 0.6:
                cols.append(str(int(rng.integers(0, 60))))
            else:
                cols.append(fields[int(rng.integers(0, len(fields)))])
        return "\t".join(cols)

    v = [("##" + line()) if rng.random() < 0.1 else line() for _ in range(int(rng.integers(0, 80)))]
    vtxt = "\n".join(v) + ("\n" if rng.random() < 0.8 else "")
    custom = bool(rng.random() < 0.4)
    t = []
    for _ in range(int(rng.integers(0, 30))):
        if custom:
            al = ["A", "C", "G", "T", ".", "N"] + (["AC", "", "a"] if rng.random() < 0.15 else [])   # sometimes patterns no key can stand for
            t.append("\t".join([str(int(rng.integers(1, npos + 1))), al[int(rng.integers(0, len(al)))], al[int(rng.integers(0, len(al)))]] + ["x"] * 9))
        else:
            t.append(line(8))
        if rng.random() < 0.08:
            t[-1] = "#" + t[-1]          # awk makes a pattern of a '#' row all the same; R does not read it
    ttxt = "\n".join(t) + ("\n" if t else "")
    return vtxt.encode("utf-8"), ttxt.encode("utf-8"), custom


@pytest.mark.parametrize("seed", range(8))
def test_columns_equal_text_semantics_on_random_inputs(qmlib, oracle, tmp_path, seed):
    """Property: for every input of ASCII and valid UTF-8, packing to columns, deciding what the columns cannot describe on the host
    path, classifying the columns (oracle as checker) and writing back gives exactly the text-level result, which
    is pinned to the reference by the golden vectors and the live fuzz of make_golden.py.  The generator makes
    POS spellings like "20 " and "x.y", pattern-shaped runs of later fields, '#' lines that pass the filter,
    '#' truth rows and truth rows with multi-character alleles: nothing is skipped."""
    from quasimodo_amd import scan_vcf
    rng = np.random.default_rng(9000 + seed)
    n_host = 0
    for _ in range(60):
        vcf, truth, custom = _random_case(rng)
        sv = scan_vcf(vcf)
        assert sv.n_refused == 0
        f, tp, fp, st = oracle.extract_text(vcf, truth, custom=custom, pure_strain=False)
        out = tmp_path / "o.vcf"
        sv.write(str(out), (sv.flags & 1).astype(np.uint8), 0)
        assert out.read_bytes() == f
        n_host += sv.n_host
        tk, cls, roc, sc = host_classify(oracle, sv, truth, custom=custom)
        sv.write(str(out), cls, 1)
        assert out.read_bytes() == tp
        sv.write(str(out), cls, 2)
        assert out.read_bytes() == fp
        hk, hk_tp = sv.header_kept
        assert sc["tp_lines"] + hk_tp == st["tp_lines"] and sc["fp_lines"] + hk - hk_tp == st["fp_lines"]
        assert int(roc[0, 20]) == sc["tp_lines"] and int(roc[1, 20]) == sc["fp_lines"]
        rc = oracle.count_text(f, truth, custom=custom)
        assert rc["calleridentify"] == sc["n_pass"] and rc["TP"] == sc["TP_R"] and rc["FP"] == sc["FP_R"]
    assert n_host > 20


# ---- SNP / indel splitters (rules/vis_eval_vcf.smk:25-86) ------------------------------------
def _split_cases():
    import json
    g = os.path.join(os.path.dirname(__file__), "golden")
    return [(g, c) for c in json.load(open(os.path.join(g, "split", "manifest.json")))["cases"]]


@pytest.mark.parametrize("g,c", _split_cases(), ids=lambda x: x["expected"].split("/")[-1] if isinstance(x, dict) else "")
def test_splitter_equals_awk_output(qmlib, tmp_path, g, c):
    """golden = this image's awk (mawk 1.3.4: `{2,}` literal) run with the rule's program"""
    from quasimodo_amd import vcfio
    out = tmp_path / "o.vcf"
    n = vcfio.split_variants(os.path.join(g, c["input"]), str(out), c["kind"], flavour=vcfio.AWK_MAWK_LITERAL)
    exp = open(os.path.join(g, c["expected"]), "rb").read()
    assert out.read_bytes() == exp
    assert n == exp.count(b"\n")


def test_bgzf_writer(qmlib, tmp_path):
    """the rules' second output (`bgzip -c`, rules/vis_eval_vcf.smk:36,51,67,82): BGZF members of at most 64 KiB with a
    'BC' extra field carrying their size, the fixed EOF member last; gzip reads the concatenation"""
    import gzip
    import struct
    from quasimodo_amd import vcfio
    eof = bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])
    rng = np.random.default_rng(1)
    for name, data in (("empty", b""), ("small", b"##x\n#CHROM\nc\t1\t.\tA\tG\n"), ("edge", b"A" * 0xff00), ("edge1", b"C" * (0xff00 + 1)),
                       ("noise", rng.integers(0, 256, 200000, dtype=np.uint8).tobytes()), ("text", b"chr1\t12345\t.\tA\tAC\t99\tPASS\tDP=10\n" * 20000)):
        src = tmp_path / (name + ".vcf")
        src.write_bytes(data)
        gz = open(vcfio.bgzip(str(src)), "rb").read()
        assert gzip.decompress(gz) == data and gz.endswith(eof)
        off = total = 0
        while off < len(gz):                                   # walk the members by their BSIZE fields
            assert gz[off:off + 4] == b"\x1f\x8b\x08\x04" and gz[off + 10:off + 16] == b"\x06\x00BC\x02\x00"
            bsize = struct.unpack_from("<H", gz, off + 16)[0] + 1
            isize = struct.unpack_from("<I", gz, off + bsize - 4)[0]
            assert isize <= 0x10000 and bsize <= 0x10000
            total += isize
            off += bsize
        assert off == len(gz) and total == len(data)
    g = os.path.join(os.path.dirname(__file__), "golden", "split", "input", "probe.vcf")
    out = tmp_path / "p.xsnp.vcf"
    vcfio.split_variants(g, str(out), "xsnp", bgz=True)
    assert gzip.decompress(open(str(out) + ".gz", "rb").read()) == out.read_bytes()


def test_splitter_posix_interval_flavour(qmlib, tmp_path):
    """`{2,}` as an interval: REF or ALT *starting* with two bases (the pattern has no `$`), '#' lines
    always, and twice when they satisfy the pattern themselves.  Hand-derived from the awk program."""
    from quasimodo_amd import vcfio
    g = os.path.join(os.path.dirname(__file__), "golden", "split", "input", "probe.vcf")
    out = tmp_path / "o.vcf"
    vcfio.split_variants(g, str(out), "xindel", flavour=vcfio.AWK_POSIX)
    got = out.read_bytes().split(b"\n")
    assert got[-1] == b""
    pos = [ln.split(b"\t")[1] for ln in got[:-1] if not ln.startswith(b"#")]
    assert pos == [b"12", b"13", b"18"]          # AC>G, A>GT, AC,G>T ; G,T / A{2,} / c{2,}TT do not start with two bases
    assert [ln.split(b"\t")[0] for ln in got[:-1] if ln.startswith(b"#")] == [b"##fileformat=VCFv4.2", b"#CHROM", b"#mid", b"#mid2"]
    # default flavour comes from QM_AWK_FLAVOUR and is POSIX
    out2 = tmp_path / "o2.vcf"
    vcfio.split_variants(g, str(out2), "xindel")
    assert out2.read_bytes() == out.read_bytes()
    with pytest.raises(ValueError):
        vcfio.split_variants(g, str(out2), "snp")


def test_many_sample_columns_are_tokenised_without_a_field_limit(qmlib, oracle):
    """a 200-sample VCF line is canonical; the same line with a pattern-shaped run of fields (digits, '.', Y, Z)
    far to the right is the reference's fgrep hazard and is flagged"""
    from quasimodo_amd import vcfio
    samples = "\t".join("0/1:%d" % (10 + i) for i in range(200))
    ok = "c\t100\t.\tA\tG\t50\tPASS\tDP=9\tGT:DP\t" + samples
    hazard = "c\t200\t.\tA\tG\t50\tPASS\tDP=9\tGT:DP\t" + samples + "\t77\t.\tC\tT\tend"
    text = ("##x\n#CHROM\n" + ok + "\n" + hazard + "\n").encode()
    sv = vcfio.scan_vcf(text)
    assert sv.n_records == 2 and sv.line_kind.tolist() == [1, 1, 0, 2] and sv.n_host == 1 and sv.n_refused == 0
    # and the reference mechanism agrees that the hazard is real: truth 77 . C T matches line 2 by its tail
    f, tp, fp, st = oracle.extract_text(text, b"c\t77\t.\tC\tT\t30\tPASS\tX\n")
    assert st["tp_lines"] == 1 and tp.count(b"\t200\t") == 1


def test_vector_scan_equals_the_line_by_line_rules(qmlib):
    """The tokenizer counts lines / header lines with population counts over compare masks and indexes a line's tabs 16 or 32
    bytes at a time (SURVEY f-1): the same answers as the plain rules -- a last line without a newline counts, an empty line
    is a data line, '#' only counts at a line start -- with line starts, tabs, NUL and non-ASCII bytes on every offset of a
    vector block, and more tabs in a line than the index keeps."""
    import ctypes as C
    from quasimodo_amd import vcfio
    rng = np.random.default_rng(23)
    L = qmlib
    L.qm_vcf_count_lines.restype = C.c_int64
    alphabet = np.frombuffer(b"\n\n\t\t#AC.1 \xc3\x00", np.uint8)
    for trial in range(300):
        n = int(rng.integers(0, 200))
        raw = alphabet[rng.integers(0, len(alphabet) - (2 if trial % 3 else 0), n)].tobytes()
        lines = raw.split(b"\n")
        if raw.endswith(b"\n") or not raw:
            lines = lines[:-1]
        assert L.qm_vcf_count_lines(raw, len(raw)) == len(lines), raw
        sv = vcfio.scan_vcf(raw)
        assert sv.n_lines == len(lines)
        assert sv.n_records == sum(1 for ln in lines if not ln.startswith(b"#")), raw
        off = np.cumsum([0] + [len(ln) + 1 for ln in lines])[:len(lines)]
        assert list(sv.line_off[:len(lines)]) == list(off), raw
    # a data line with 100 sample columns (more tabs than the index keeps) and a pattern-shaped window far behind the ID column
    many = b"chr\t7\t.\tA\tG\t50\tPASS\tDP=1\tGT" + b"\t0/1" * 60 + b"\t.\tA\tG\n"
    few = b"chr\t7\t.\tA\tG\t50\tPASS\tDP=1\tGT\t.\tA\tG;x\n"
    sv = vcfio.scan_vcf(many + few + b"chr\t8\t.\tA\tG\t50\tPASS\tDP=1\n")
    assert list(sv.line_kind[:3]) == [2, 2, 0]      # QM_LINE_DATA_HOST twice: the text decides, on the host path


def test_bgzf_blocks_are_what_htslib_indexes(qmlib, tmp_path):
    """qm_bgzf_write (the *.vcf.gz the rules declare, rules/vis_eval_vcf.smk:29,36): tabix / htslib address a BGZF file by
    virtual offsets = (start of a block in the file) << 16 | (offset inside its inflated data), so every block must be a
    complete gzip member of at most 64 KiB whose 'BC' extra field holds its size - 1, inflating to at most 64 KiB, and the
    file must end with the 28-byte EOF member.  Walked block by block here, as bgzf_read_block does."""
    import struct
    import zlib
    rng = np.random.default_rng(3)
    lines = [b"##fileformat=VCFv4.2", b"#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO"]
    for p in range(1, 20001):
        lines.append(b"chr1\t%d\t.\t%s\t%s\t%d\tPASS\tDP=%d" % (p * 7, b"ACGT"[p % 4:p % 4 + 1], b"CGTA"[p % 4:p % 4 + 1], int(rng.integers(0, 300)), int(rng.integers(1, 10**6))))
    data = b"\n".join(lines) + b"\n"
    assert len(data) > 5 * 0xff00
    out = tmp_path / "x.vcf.gz"
    assert qmlib.qm_bgzf_write(str(out).encode(), data, len(data), -1) == 0
    raw = out.read_bytes()
    off, got, nblocks, voffsets = 0, [], 0, []
    while off < len(raw):
        magic, cm, flg, _, _, _, xlen = struct.unpack_from("<HBBIBBH", raw, off)
        assert (magic, cm, flg) == (0x8b1f, 8, 4) and xlen == 6
        si1, si2, slen, bsize = struct.unpack_from("<BBHH", raw, off + 12)
        assert (si1, si2, slen) == (66, 67, 2)
        size = bsize + 1
        assert 28 <= size <= 0x10000 and off + size <= len(raw)
        body = raw[off + 18:off + size - 8]
        crc, isize = struct.unpack_from("<II", raw, off + size - 8)
        inflated = zlib.decompress(body, -15)
        assert len(inflated) == isize <= 0x10000 and zlib.crc32(inflated) == crc
        voffsets.append((off << 16, isize))
        got.append(inflated)
        off += size
        nblocks += 1
    assert b"".join(got) == data and got[-1] == b"" and nblocks == (len(data) + 0xfeff) // 0xff00 + 1
    # the virtual offset of any byte: block start << 16 | offset in block -- what a .tbi written by `tabix -p vcf` holds
    pos = data.index(b"chr1\t70007\t")
    blk, inner = divmod(pos, 0xff00)
    v = voffsets[blk][0] | inner
    start = v >> 16
    bsz = struct.unpack_from("<H", raw, start + 16)[0] + 1
    assert zlib.decompress(raw[start + 18:start + bsz - 8], -15)[v & 0xffff:].startswith(b"chr1\t70007\t")


class _Bgzf:
    """A BGZF file as htslib addresses it: members located by their BC field, a virtual offset = member start << 16 | offset
    in its inflated bytes."""

    def __init__(self, raw):
        import struct
        import zlib
        self.blocks = {}          # member start -> (inflated bytes, start of the next member)
        off = 0
        while off < len(raw):
            assert raw[off:off + 4] == b"\x1f\x8b\x08\x04"
            size = struct.unpack_from("<H", raw, off + 16)[0] + 1
            self.blocks[off] = (zlib.decompress(raw[off + 18:off + size - 8], -15), off + size)
            off += size

    def read_lines(self, vbeg, vend):
        """the whole lines between two virtual offsets, with the virtual offset each starts at"""
        coff, inner = vbeg >> 16, vbeg & 0xffff
        out, cur, cur_v = [], b"", vbeg
        while coff in self.blocks and (coff << 16 | inner) < vend:
            data, nxt = self.blocks[coff]
            if inner >= len(data):
                coff, inner = nxt, 0
                continue
            stop = len(data) if (vend >> 16) != coff else min(len(data), vend & 0xffff)
            nl = data.find(b"\n", inner, stop)
            if not cur:
                cur_v = coff << 16 | inner
            if nl < 0:
                cur += data[inner:stop]
                inner = stop
                if stop < len(data):
                    break
            else:
                out.append((cur_v, cur + data[inner:nl + 1]))
                cur = b""
                inner = nl + 1
        assert not cur, "a chunk ends inside a line"
        return out


def _reg2bin(beg, end):       # SAM specification, section 5.3 (min_shift 14, depth 5)
    end -= 1
    if beg >> 14 == end >> 14: return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17: return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20: return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23: return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26: return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def _reg2bins(beg, end):
    end -= 1
    bins = [0]
    for shift, first in ((26, 1), (23, 9), (20, 73), (17, 585), (14, 4681)):
        bins += list(range(first + (beg >> shift), first + (end >> shift) + 1))
    return bins


def _parse_tbi(raw):
    import gzip
    import struct
    b = gzip.decompress(raw)
    assert b[:4] == b"TBI\x01"
    n_ref, fmt, cs, cb, ce, meta, skip, l_nm = struct.unpack_from("<8i", b, 4)
    assert (fmt, cs, cb, ce, meta, skip) == (2, 1, 2, 0, ord("#"), 0)
    names = b[36:36 + l_nm].split(b"\0")[:-1]
    assert len(names) == n_ref
    off = 36 + l_nm
    refs = []
    for _ in range(n_ref):
        n_bin, = struct.unpack_from("<i", b, off); off += 4
        bins = {}
        for _ in range(n_bin):
            bn, n_chunk = struct.unpack_from("<Ii", b, off); off += 8
            assert bn not in bins
            bins[bn] = [struct.unpack_from("<QQ", b, off + 16 * k) for k in range(n_chunk)]
            off += 16 * n_chunk
        n_intv, = struct.unpack_from("<i", b, off); off += 4
        lin = list(struct.unpack_from("<%dQ" % n_intv, b, off)); off += 8 * n_intv
        refs.append((bins, lin))
    assert off == len(b) or (off + 8 == len(b) and struct.unpack_from("<Q", b, off)[0] == 0)
    return names, refs


def test_tabix_index_beside_the_bgzf_output(qmlib, tmp_path):
    """qm_bgzf_write_tbi = `bgzip -c` + `tabix -p vcf` (rules/vis_eval_vcf.smk:36-37): the .tbi is read back by a reader written
    from the format description (not from the writer) -- header, names, bins, chunks, linear index, pseudo-bin --, every chunk
    must hold whole records of its bin only, every record must be in exactly one chunk, and region queries answered the way
    tabix answers them (bins of the region, linear index as the lower bound, records filtered by overlap) must return what a
    scan of the text returns.  Three sequences, several BGZF members, records with long REF alleles and INFO END= reaching
    over bin and window borders; a VCF out of order is refused and leaves no file."""
    import gzip
    rng = np.random.default_rng(11)
    lines = [b"##fileformat=VCFv4.2", b"#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO"]
    recs = []      # (chrom, beg, end, line)
    for chrom, n, span in ((b"chrA", 9000, 3_000_000), (b"ctg-2", 40, 200_000_000), (b"z", 3000, 70_000)):
        pos = np.sort(rng.integers(1, span, n))
        for p in pos:
            kind = rng.integers(0, 20)
            ref, info = b"ACGT"[int(p) % 4:int(p) % 4 + 1], b"DP=%d" % int(rng.integers(1, 99))
            end = int(p) - 1 + 1
            if kind == 0:
                ref = b"ACGT" * int(rng.integers(2, 6000)); end = int(p) - 1 + len(ref)
            elif kind == 1:
                e = int(p) + int(rng.integers(0, 100_000)); info = b"SVTYPE=DEL;END=%d;X=1" % e; end = e if e > int(p) - 1 else end
            elif kind == 2:
                e = int(p) + int(rng.integers(20_000, 40_000)); info = b"END=%d" % e; end = e
            elif kind == 3:
                info = b"BEND=5;AEND=7"           # not an END key
            line = b"%s\t%d\t.\t%s\tG\t50\tPASS\t%s\n" % (chrom, int(p), ref, info)
            lines.append(line[:-1])
            recs.append((chrom, int(p) - 1, end, line))
    data = b"\n".join(lines) + b"\n"
    assert len(data) > 6 * 0xff00
    out = tmp_path / "x.vcf.gz"
    assert qmlib.qm_bgzf_write_tbi(str(out).encode(), data, len(data), -1) == 0
    raw = out.read_bytes()
    assert gzip.decompress(raw) == data
    plain = tmp_path / "y.vcf.gz"
    assert qmlib.qm_bgzf_write(str(plain).encode(), data, len(data), -1) == 0 and plain.read_bytes() == raw   # the same .gz as without the index
    names, refs = _parse_tbi((tmp_path / "x.vcf.gz.tbi").read_bytes())
    assert names == [b"chrA", b"ctg-2", b"z"]
    bg = _Bgzf(raw)
    first_v = {}
    for tid, (bins, lin) in enumerate(refs):
        mine = [r for r in recs if r[0] == names[tid]]
        by_line = {r[3]: r for r in mine}
        assert len(by_line) == len(mine)
        seen = []
        for bn, chunks in bins.items():
            if bn == 37450:
                assert len(chunks) == 2 and chunks[1] == (len(mine), 0)
                got = bg.read_lines(*chunks[0])
                assert [l for _, l in got] == [r[3] for r in mine]
                continue
            for cb, ce in chunks:
                assert cb < ce
                for v, l in bg.read_lines(cb, ce):
                    r = by_line[l]
                    assert _reg2bin(r[1], r[2]) == bn
                    seen.append(l)
                    first_v[l] = v
        assert sorted(seen) == sorted(r[3] for r in mine)            # every record in exactly one chunk
        # the linear index: the first record overlapping each 16 kb window, holes filled from behind
        want = {}
        for r in mine:
            for w in range(r[1] >> 14, ((r[2] - 1) >> 14) + 1):
                want.setdefault(w, first_v[r[3]])
        assert len(lin) == max(want) + 1
        for w in range(len(lin) - 1, -1, -1):
            assert lin[w] == (want[w] if w in want else lin[w + 1])
        # region queries the way tabix answers them
        span = max(r[2] for r in mine)
        for _ in range(60):
            qb = int(rng.integers(0, span)); qe = qb + int(rng.integers(1, 1 << int(rng.integers(1, 22))))
            lo = lin[min(qb >> 14, len(lin) - 1)] if (qb >> 14) < len(lin) else lin[-1]
            hits = []
            for bn in _reg2bins(qb, qe):
                for cb, ce in bins.get(bn, []):
                    if ce > lo:
                        for v, l in bg.read_lines(cb, ce):
                            r = by_line[l]
                            if r[1] < qe and r[2] > qb:
                                hits.append((v, l))
            assert [l for _, l in sorted(set(hits))] == [r[3] for r in mine if r[1] < qe and r[2] > qb]
    # out of order: refused like tabix refuses it, and nothing is left behind
    for bad in (b"c1\t10\t.\tA\tC\t9\t.\t.\nc1\t9\t.\tA\tC\t9\t.\t.\n", b"c1\t1\t.\tA\tC\t9\t.\t.\nc2\t1\t.\tA\tC\t9\t.\t.\nc1\t2\t.\tA\tC\t9\t.\t.\n"):
        o2 = tmp_path / "bad.vcf.gz"
        assert qmlib.qm_bgzf_write_tbi(str(o2).encode(), bad, len(bad), -1) == -10
        assert not o2.exists() and not (tmp_path / "bad.vcf.gz.tbi").exists()
    assert sorted(p.name for p in tmp_path.iterdir()) == ["x.vcf.gz", "x.vcf.gz.tbi", "y.vcf.gz"]
    # the Python entry points
    from quasimodo_amd.vcfio import bgzip
    from quasimodo_amd._lib import QmvtError
    src = tmp_path / "s.vcf"
    src.write_bytes(data)
    bgzip(str(src), tbi=True)
    assert (tmp_path / "s.vcf.gz.tbi").read_bytes() == (tmp_path / "x.vcf.gz.tbi").read_bytes()
    src.write_bytes(b"c1\t10\t.\tA\tC\t9\t.\t.\nc1\t9\t.\tA\tC\t9\t.\t.\n")
    with pytest.raises(QmvtError):
        bgzip(str(src), tbi=True)
    bgzip(str(src), tbi="if-sorted")
    assert (tmp_path / "s.vcf.gz").exists() and not (tmp_path / "s.vcf.gz.tbi").exists()
