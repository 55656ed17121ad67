"""The oracle against the reference's own outputs (tests/golden/, made by running the
reference script) and against independent restatements.  CPU only."""
import numpy as np
import pytest

from conftest import case_id, golden_cases, random_columns, random_truth, read_case

CASES = golden_cases()


@pytest.mark.parametrize("e", CASES, ids=case_id)
def test_oracle_matches_reference_bytes(oracle, e):
    vcf, truth, exp = read_case(e)
    f, tp, fp, st = oracle.extract_text(vcf, truth, custom=e["mode"] == "custom", pure_strain=e["pure"])
    assert f == exp["filtered"]
    assert tp == exp["tp"]
    assert fp == exp["fp"]
    # line accounting
    assert st["kept"] == st["tp_lines"] + st["fp_lines"] or e["pure"]


def test_golden_covers_the_quirk_list():
    fams = {e["family"] for e in CASES}
    assert {"quirks", "quirks_canon", "hcmv", "config1", "custom", "edge"} <= fams
    assert sum(1 for e in CASES if e["family"] == "hcmv") == 60          # 10 samples x 6 callers
    assert sum(1 for e in CASES if e["pure"]) >= 24


# mawk 1.3.4 answers observed in the build container (tests/golden/PROVENANCE.md)
AWK_PROBES = {
    b"20": 1, b"20.0": 1, b"2e1": 1, b"+20": 1, b" 20": 1, b"020": 1, b"20.": 1, b"19.999": 0, b"-1": 0, b".5": 0,
    b"19.9999999999999999999": 1, b"PASS": 1, b"20abc": 1, b"nan": 1, b"inf": 1, b"30\r": 1, b"5\r": 1, b"10\r": 0,
    b"1,2": 0, b"": 0, b"0x14": 1, b"0x13": 0, b"1e400": 0, b"+inf": 0, b"-inf": 0, b"1e-400": 0, b"2e-400": 1,
    b"20 ": 1, b"5 ": 0, b"20\x0b": 1, b"\x0b20": 0, b"5\x0c": 1, b"3e": 1, b"e1": 1, b"+.5e2": 1, b"+": 0, b"..": 0,
    b"1.": 0, b"1.e1": 0, b"0x": 0, b"0X14": 1, b"0x1p5": 1, b"infinity": 1, b"NAN": 1, b"2_0": 1, b"9": 0,
    b"100": 1, b"1e2": 1, b"1d2": 0, b"077": 1, b"0b1": 0, b"2e1.": 1, b"0x14.": 1, b"2e400": 1, b"3e-310": 1,
    b"1e-310": 0, b"4.9e-324": 1, b"2.4e-324": 0, b"1.7976931348623158e308": 1, b"1.7976931348623159e308": 0,
    b"00x14": 0, b"+5": 0, b"-0": 0, b"- 30": 0, b"0e400": 0, b"0x1p1024": 0, b"0x1p1023": 1, b"\xff": 1,
}


@pytest.mark.parametrize("field,want", sorted(AWK_PROBES.items()))
def test_awk_ge20_probe(oracle, field, want):
    assert oracle.awk_ge(field, 20) == bool(want)


def test_dot_qual_kept_by_second_clause(oracle):
    assert not oracle.awk_ge(b".", 20)
    assert oracle.caller_filter(b"c\t1\t.\tA\tG\t.\tPASS\tX")
    assert not oracle.caller_filter(b"c\t1\t.\tA\tG")          # Q8: fewer than 6 columns
    assert not oracle.caller_filter(b"c\t1\t.\ta\tg\t99")      # Q4: lower case


def test_r_counts_hand_derived(oracle):
    """caller_performance_compare.R:84-99 on a case small enough to do by hand."""
    hdr = b"#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n"
    truth = hdr + b"c\t10\t.\tA\tG\t30\tPASS\tX\nc\t20\t.\tC\tT\t30\tPASS\tX\nc\t20\t.\tC\tT\t30\tPASS\tX\n" \
                  b"c\t30\t.\tG\tA,T\t30\tPASS\tX\nc\t40\t.\tGA\tG\t30\tPASS\tX\nc\t50\t.\tT\tA\t30\tPASS\tX\n"
    filtered = hdr + b"c\t10\t.\tA\tG\t50\tPASS\tX\nc\t10\t.\tA\tG\t60\tPASS\tX\nc\t20\trs9\tC\tT\t50\tPASS\tX\n" \
                     b"c\t21\t.\tC\tT\t50\tPASS\tX\nc\t21\t.\tC\tT\t50\tPASS\tX\nc\t22\t.\tC\tA\t50\tPASS\tX\n"
    c = oracle.count_text(filtered, truth)
    # truth vector: 10-A-G, 20-C-T, 20-C-T, 50-T-A  -> genomediff 4 (not de-duplicated), 3 unique
    assert c == {"genomediff": 4, "calleridentify": 6, "TP": 2, "FP": 2, "FN": 1, "truth_unique": 3}


def test_r_counts_custom_truth(oracle):
    snps = b"10\tA\tG\t1\t1\t1\t9\t9\t1\t1\tr\tq\n20\tN\tT\t1\t1\t1\t9\t9\t1\t1\tr\tq\n30\t.\tT\t1\t1\t1\t9\t9\t1\t1\tr\tq\n"
    filtered = b"c\t10\t.\tA\tG\t50\nc\t11\t.\tA\tG\t50\n"
    c = oracle.count_text(filtered, snps, custom=True)
    assert c["genomediff"] == 2 and c["TP"] == 1 and c["FP"] == 1 and c["FN"] == 1


def test_fp_overlap_hand_derived(oracle):
    mk = lambda keys: b"#h\n" + b"".join(b"c\t%d\t.\t%s\t%s\t50\n" % (p, r, a) for p, r, a in keys)
    a = mk([(1, b"A", b"G"), (2, b"A", b"G"), (3, b"A", b"G"), (3, b"A", b"G")])
    b = mk([(2, b"A", b"G"), (3, b"A", b"G"), (4, b"A", b"G")])
    c = mk([(3, b"A", b"G"), (3, b"A", b"C"), (5, b"T", b"C")])
    reg = oracle.fp_overlap_text([a, b, c])
    assert reg == [0, 1, 1, 1, 2, 0, 0, 1]   # masks: 1:{1} 2:{4} 3:{2} 4:{3AC,5} 7:{3AG}


def _python_restatement(pos, ref, alt, qual, flags, truth, n_bins):
    """Independent set-based restatement of the column semantics."""
    tset = {(int(p), int(r), int(a)) for p, r, a in zip(*truth) if 0 <= r < 4 and 0 <= a < 4}
    n = len(pos)
    cls = np.zeros(n, np.uint8)
    hist = np.zeros((3, n_bins), np.int64)
    tmax, tr_keys, fp_keys = {}, set(), set()
    for i in range(n):
        k = (int(pos[i]), int(ref[i]), int(alt[i]))
        snp = 0 <= k[1] < 4 and 0 <= k[2] < 4
        hit = snp and k in tset and not (flags[i] & 4)
        iddot = bool(flags[i] & 2)
        passed = bool(flags[i] & 1) and snp
        tpline = (hit and iddot) or (snp and bool(flags[i] & 8))   # bit3: the host path found the line selected by fgrep
        if passed:
            cls[i] = 3 if tpline else 1
            (tr_keys if hit else fp_keys).add(k + (bool(flags[i] & 4),))   # keyless records are keys of their own
        q = float(qual[i])
        b = -1 if (q != q or q < 0) else int(min(np.floor(q), n_bins - 1))
        if snp and b >= 0:
            if tpline:
                hist[0, b] += 1
                if hit and iddot:
                    tmax[k] = max(tmax.get(k, -1), b)
            else:
                hist[1, b] += 1
    for b in tmax.values():
        hist[2, b] += 1
    roc = np.flip(np.cumsum(np.flip(hist, 1), 1), 1)
    return cls, roc, len(tr_keys), len(fp_keys)


@pytest.mark.parametrize("seed", range(6))
def test_oracle_columns_vs_python_sets(oracle, seed):
    rng = np.random.default_rng(seed)
    truth = random_truth(rng, 200, 500)
    cols = random_columns(rng, 1500, 500, truth, sorted_=bool(seed % 2))
    cls, roc, sc = oracle.classify_columns(*cols, *truth, n_bins=256)
    pcls, proc, ptr, pfr = _python_restatement(*cols, truth, 256)
    assert np.array_equal(cls, pcls)
    assert np.array_equal(roc.astype(np.int64), proc)
    assert sc["TP_R"] == ptr and sc["FP_R"] == pfr
    assert sc["n_pass"] == int((pcls & 1).sum()) and sc["tp_lines"] == int((pcls == 3).sum())
    # pinned consistency point: the ROC at t = 20 is the tp/fp line split
    assert int(roc[0, 20]) == sc["tp_lines"] and int(roc[1, 20]) == sc["fp_lines"]
