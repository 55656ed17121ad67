"""R-free writers of the path's tables.

  final_tables/caller_performance.tsv   scripts/caller_performance_compare.R:57-143
  final_tables/snpcall_benchmark.txt    scripts/custom_snp_benchmark.R:30-95
  final_tables/snpcaller_fp_snp_compare.txt   region sizes of the Venn diagrams drawn by
                                        scripts/snpcaller_fp_compare.R (the PDF itself is out of scope)

The counts come from the engine (Job.stats); only the three ratios are computed here, with R's
round(x, 3) (R 3.5.1: x * 10^3 in long double, nearbyint, / 10^3).  Raw counts are written next to
the rounded columns so that a tie-rounding difference between R versions is visible."""
import math
import os

import numpy as np


class RTableError(RuntimeError):
    """An input that R's read.table would not read the way the engine's counts assume."""


def r_hostile_rows(path):
    """How many non-comment rows of a truth file hold ' or ", or a '#' in front of their last column (see check_r_readable): R reads
    the truth files with the same read.table call (caller_performance_compare.R:29-39 through make_snp_vector,
    custom_snp_benchmark.R:23-24)."""
    n = 0
    with open(path, "rb") as fh:
        for ln in fh:
            if ln[:1] == b"#":
                continue
            ln = ln.rstrip(b"\n")
            h, last_tab = ln.find(b"#"), ln.rfind(b"\t")
            if b"'" in ln or b'"' in ln or (h >= 0 and not (last_tab >= 0 and h > last_tab)):
                n += 1
    return n


def check_r_readable(rows, strict=None):
    """A6 / A6c are RESTATED from the R text (no R in this image: parity unpinned), and the restatement reads a file as lines
    split at tabs.  R's read.table -- comment.char = "#", the default quote = "\"'", eight colClasses recycled over the
    columns of the first five lines (scripts/caller_performance_compare.R:29-39, custom_snp_benchmark.R:23-24,45-48,
    snpcaller_fp_compare.R:36-39) -- does not: a '#' in front of the last column cuts the line there (fewer fields than the
    first lines had: read.table stops, the tryCatch turns the WHOLE FILE into an empty vector, :37-40,101-108; a '#' inside the
    last column only shortens it, which no count reads), a ' or " opens a string that swallows tabs and newlines up to the next
    one (balanced inside one field: the quotes are dropped, also from POS; otherwise fields shift or the file fails the same
    way).  The engine counts what the reference's shell pipeline kept (those bytes are pinned); it does not guess what R makes
    of such a line.  Returns the offending rows [(name, kept lines, truth rows)]; what the writers do with them:
      strict (default: QM_LENIENT unset): such a file gets the row R's tryCatch writes for a file read.table gave up on -- an
        EMPTY vector (a VCF: calleridentify 0, NA ratios; a truth file: genomediff 0, everything the caller kept is FP) -- every
        other row is written as usual, and a warning on stderr names the files (the reference does the same per file and goes
        on; ADVICE round 4: one quoted INFO value must not cost a run its whole table);
      strict = "refuse": RTableError before anything is written;
      lenient (QM_LENIENT=1): the tab-split counts.
    rows: iterable of (name, stats)."""
    if strict is None:
        strict = os.environ.get("QM_LENIENT", "0") in ("", "0")
    bad = [(name, int(st.get("r_hostile") or 0), int(st.get("truth_r_hostile") or 0)) for name, st in rows
           if (st.get("r_hostile") or 0) or (st.get("truth_r_hostile") or 0)]
    if bad and strict:
        msg = ("R's read.table (comment.char = \"#\", quote = \"\\\"'\") would not read these files as tab-split lines: "
               + "; ".join("%s (%d kept line(s), %d truth row(s) with ' or \", or '#' in front of the last column)" % b for b in bad[:5])
               + (" ... and %d more" % (len(bad) - 5) if len(bad) > 5 else ""))
        if strict == "refuse":
            raise RTableError(msg + ".  QM_LENIENT=1 writes the tab-split counts.")
        import sys
        sys.stderr.write("quasimodo_amd.tables: " + msg + " -- their rows are written as R's tryCatch writes a file it cannot read "
                         "(empty); QM_LENIENT=1 writes the tab-split counts instead.\n")
    return bad


def _as_r_reads(stats, strict):
    """The stats of one row as the strict writers count them: a VCF / truth file R cannot read is an empty vector."""
    if not strict:
        return stats
    st = dict(stats)
    if st.get("truth_r_hostile"):
        st.update(genomediff=0, FP_R=int(st.get("TP_R", 0)) + int(st.get("FP_R", 0)), TP_R=0)
    if st.get("r_hostile"):
        st.update(n_pass=0, TP_R=0, FP_R=0)
    return st


CALLER_MAP = {"bcftools": "BCFtools", "clc": "CLC", "freebayes": "FreeBayes", "gatk": "GATK", "lofreq": "LoFreq",
              "varscan": "VarScan2"}  # caller_performance_compare.R:24-27


def r_div(a, b):
    """R's a / b on doubles: x/0 is Inf, 0/0 is NaN."""
    a, b = float(a), float(b)
    if b == 0.0:
        return float("nan") if a == 0.0 else math.copysign(float("inf"), a)
    return a / b


def r_round3(x):
    if x is None or math.isnan(x) or math.isinf(x):
        return x
    return float(np.rint(np.longdouble(x) * np.longdouble(1000.0)) / np.longdouble(1000.0))


def r_str(x):
    """write.table(quote = FALSE) formatting of one cell."""
    if x is None:
        return "NA"
    if isinstance(x, str):
        return x
    if isinstance(x, (int, np.integer)):
        return str(int(x))
    if math.isnan(x):
        return "NaN"
    if math.isinf(x):
        return "Inf" if x > 0 else "-Inf"
    return "%.15g" % x


def performance_row(stats):
    """One VCF -> (genomediff, calleridentify, TP, FP, precision, recall, f1) as R computes them
    (caller_performance_compare.R:84-128).  None = NA."""
    n = int(stats["n_pass"])
    if stats.get("pure_strain"):
        return 0, n, 0, n, 0.0, None, None                       # :121-128
    gd = int(stats["genomediff"])
    if n == 0:
        return gd, 0, 0, 0, None, None, None                      # :101-108
    tp, fp = int(stats["TP_R"]), int(stats["FP_R"])
    p = r_round3(r_div(tp, n))
    r = r_round3(r_div(tp, gd))
    f1 = r_round3(r_div(2 * (p * r), p + r))
    return gd, n, tp, fp, p, r, f1


def write_caller_performance(path, rows, strict=None):
    """rows: iterable of (caller_lower, sample, stats).  Returns the (name, kept lines, truth rows) of the rows that were written as
    R's empty vector (strict mode; a caller can fail its CI on a non-empty answer)."""
    rows = list(rows)
    if strict is None:
        strict = os.environ.get("QM_LENIENT", "0") in ("", "0")
    bad = check_r_readable([("%s/%s" % (c, smp), st) for c, smp, st in rows], strict)   # (both reads of this script sit in a tryCatch: make_snp_vector, :29-55)
    with open(path, "w") as fh:
        fh.write("\t".join(["caller", "mixture", "genomediff", "calleridentify", "TP", "FP", "Precision", "Recall", "F1"]) + "\n")
        for caller, sample, stats in rows:
            vals = performance_row(_as_r_reads(stats, strict))
            fh.write("\t".join([CALLER_MAP.get(caller, caller), sample] + [r_str(v) for v in vals]) + "\n")
    return bad if strict else []


def write_snpcall_benchmark(path, rows, strict=None):
    """rows: iterable of (label, stats).  custom_snp_benchmark.R:30-95.
    The custom script reads the genome-difference table with NO tryCatch (:23-24): a truth file R cannot read as tab-split lines
    stops the script -- so the strict writer refuses it too (RTableError, nothing written) instead of writing a table the
    reference can never produce; only a CALLER file R gives up on (its read has a tryCatch, :45-48) gets the
    empty-vector row.  Returns the (name, kept lines, truth rows) of the rows written as R's empty vector."""
    rows = list(rows)
    if strict is None:
        strict = os.environ.get("QM_LENIENT", "0") in ("", "0")
    if strict and any(st.get("truth_r_hostile") for _, st in rows):
        check_r_readable([(n, st) for n, st in rows if st.get("truth_r_hostile")], "refuse")
    bad = check_r_readable(rows, strict)
    with open(path, "w") as fh:
        fh.write("\t".join(["caller", "genomediff", "calleridentify", "TP", "FP", "precision", "recall", "f1"]) + "\n")
        for label, stats in rows:
            st = dict(_as_r_reads(stats, strict))
            st["pure_strain"] = False     # the custom script has no pure-strain branch
            vals = performance_row(st)
            fh.write("\t".join([label] + [r_str(v) for v in vals]) + "\n")
    return bad if strict else []


def write_fp_overlap(path, per_sample, callers):
    """per_sample: {sample: region counts indexed by membership mask (bit i = callers[i])}."""
    n = len(callers)
    with open(path, "w") as fh:
        fh.write("\t".join(["sample", "callers", "count"]) + "\n")
        for sample in sorted(per_sample):
            reg = per_sample[sample]
            for m in range(1, 1 << n):
                names = "&".join(CALLER_MAP.get(callers[i], callers[i]) for i in range(n) if m >> i & 1)
                fh.write("%s\t%s\t%d\n" % (sample, names, int(reg[m])))


def write_weighted_roc(path, roc, truth_unique):
    """The engine's exact-match ROC sweep in the column layout the reference's plotting code reads from
    `rtg vcfeval` (scripts/caller_performance_compare.R:329-340: score, TP_baseline, FP, TP_call, FN,
    Precision, Recall, F1; '#' comment lines; gzip when the path ends in .gz).  The VALUES are a
    build-defined quantity -- exact (pos, ref, alt) matching, not RTG's haplotype-aware matching -- and
    equal the reference's own tp/fp split at score 20 only."""
    import gzip
    n_bins = roc.shape[1]
    lines = ["#Version qmvt exact-match ROC (not rtg vcfeval), baseline = distinct single-base truth keys",
             "#total baseline variants: %d" % int(truth_unique),
             "#score field: QUAL",
             "#score\ttrue_positives_baseline\tfalse_positives\ttrue_positives_call\tfalse_negatives\tprecision\tsensitivity\tf_measure"]
    for t in range(n_bins - 1, -1, -1):
        tp_call, fp, tp_base = int(roc[0, t]), int(roc[1, t]), int(roc[2, t])
        if tp_call + fp == 0:
            continue
        fn = int(truth_unique) - tp_base
        prec = tp_call / (tp_call + fp)
        sens = tp_base / truth_unique if truth_unique else 0.0
        f1 = 2 * prec * sens / (prec + sens) if prec + sens > 0 else 0.0
        lines.append("%d\t%d\t%d\t%d\t%d\t%.4f\t%.4f\t%.4f" % (t, tp_base, fp, tp_call, fn, prec, sens, f1))
    data = ("\n".join(lines) + "\n").encode()
    if str(path).endswith(".gz"):
        # no name and no time stamp in the header: the same numbers give the same bytes, whoever writes them and when
        with open(path, "wb") as raw, gzip.GzipFile(filename="", mode="wb", fileobj=raw, mtime=0) as fh:
            fh.write(data)
    else:
        with open(path, "wb") as fh:
            fh.write(data)
