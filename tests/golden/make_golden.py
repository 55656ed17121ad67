#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE.

Run in the build container only (needs /root/reference, bash, awk = mawk 1.3.4,
GNU grep 3.7):

    python3 tests/golden/make_golden.py            # regenerate everything
    python3 tests/golden/make_golden.py --fuzz 300 # also fuzz oracle vs reference (nothing stored)

What is committed is DATA: generated inputs and the bytes the reference script
(program/extract_TP_FP_SNPs.py, run unmodified from /root/reference) wrote for
them.  No reference source is copied.  The bundled HCMV VCFs (data/snp.tar.gz)
are absent from the reference mount, so the "hcmv" family is HCMV-*shaped*:
real contig names/lengths and REF bases from ref/*.fa, bundle file naming,
mummer2vcf truth dialect (program/mummer2vcf.py:83-84,310,349-350).
"""
import argparse
import json
import os
import random
import shutil
import subprocess
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
SCRIPT = os.path.join(REF, "program", "extract_TP_FP_SNPs.py")

SAMPLE_REF = {  # rules/load_config.smk:20-23
    "TM-0-1": "Merlin", "TM-1-1": "Merlin", "TM-1-10": "Merlin", "TM-1-50": "Merlin", "TM-1-0": "TB40E",
    "TA-1-0": "TB40E", "TA-1-1": "AD169", "TA-1-10": "AD169", "TA-1-50": "AD169", "TA-0-1": "AD169"}
CALLERS = ["lofreq", "varscan", "clc", "bcftools", "freebayes", "gatk"]  # eval_variantcall.smk:10
FASTA = {"Merlin": "Merlin.BAC.fa", "TB40E": "TB40E.GFP.fa", "AD169": "AD169.BAC.fa"}
BASES = "ACGT"


def read_fasta(name):
    path = os.path.join(REF, "ref", FASTA[name])
    contig, seq = None, []
    with open(path) as fh:
        for ln in fh:
            if ln.startswith(">"):
                contig = ln[1:].split()[0]
            else:
                seq.append(ln.strip())
    return contig, "".join(seq).upper()


# ----------------------------------------------------------------------------
# input generators
# ----------------------------------------------------------------------------
def truth_vcf(rng, contig, seq, qry_name, n_snv, n_indel, n_multi):
    """mummer2vcf.py dialect; returns (text, list of (pos, ref, alt) single-base SNVs)."""
    L = len(seq)
    positions = sorted(rng.sample(range(2, L - 40), n_snv + n_indel + n_multi))
    kinds = ["snv"] * n_snv + ["indel"] * n_indel + ["multi"] * n_multi
    rng.shuffle(kinds)
    hdr = ["##fileformat=VCFv4.2", "##fileDate=20190801", "##source=mummer2vcf.py",
           "##reference=/data/ref/%s" % contig,
           "##contig=<ID=%s,length=%d>" % (contig, L),
           '##INFO=<ID=DP,Number=1,Type=Integer,Description="Total depth of quality bases">',
           '##INFO=<ID=REF1,Number=1,Type=String,Description="The name of the 1st reference sequence">',
           '##INFO=<ID=REF2,Number=1,Type=String,Description="The name of the 2nd reference sequence">',
           '##INFO=<ID=ORIG,Number=1,Type=String,Description="The original position of variant at 2nd reference sequence">',
           '##INFO=<ID=TYPE,Number=1,Type=String,Description="Indicates that the variant is an INDEL or SNV.">',
           "\t".join(["#CHROM", "POS", "ID", "REF", "ALT", "QUAL", "FILTER", "INFO"])]
    rows, snvs = [], []
    for p, kind in zip(positions, kinds):
        r = seq[p - 1]
        if r not in BASES:
            r = "A"
        orig = "%s:%d" % (qry_name, p + rng.randint(-3000, 3000))
        if kind == "snv":
            a = rng.choice([b for b in BASES if b != r])
            ref, alt, typ = r, a, "SNV"
            snvs.append((p, r, a))
        elif kind == "multi":
            al = rng.sample([b for b in BASES if b != r], 2)
            ref, alt, typ = r, ",".join(al), "SNV"
        else:
            k = rng.randint(1, 6)
            if rng.random() < 0.5:
                ref, alt = seq[p - 1:p + k], r
            else:
                ref, alt = r, r + "".join(rng.choice(BASES) for _ in range(k))
            typ = "INDEL"
        info = "DP=30;REF1=%s;REF2=%s;ORIG=%s;TYPE=%s" % (contig, qry_name, orig, typ)
        rows.append("\t".join([contig, str(p), ".", ref, alt, "30", "PASS", info]))
    return "\n".join(hdr + rows) + "\n", snvs


def qual_str(rng, caller):
    u = rng.random()
    if caller == "lofreq":
        return str(int(round(10 ** rng.uniform(0.5, 4.7))))
    if caller == "varscan":  # varscan2vcf.py:73  (int+int)/2 under python3
        return str((rng.randint(5, 60) + rng.randint(5, 60)) / 2)
    if caller == "bcftools":
        return "%.3f" % rng.uniform(3, 228) if u < 0.8 else "%.6g" % (10 ** rng.uniform(-1, 2.4))
    if caller == "freebayes":
        if u < 0.15:
            return "%.6g" % (10 ** rng.uniform(-15, 0))      # 4.41e-15 style
        if u < 0.2:
            return "0"
        return "%.6g" % (10 ** rng.uniform(0, 4.5))
    if caller == "clc":
        return "%.1f" % rng.uniform(10, 200)
    return "%.2f" % (10 ** rng.uniform(1, 4))               # gatk


def caller_header(caller, contig, L, sample):
    h = ["##fileformat=VCFv4.%d" % (0 if caller == "lofreq" else 1 if caller == "varscan" else 2)]
    if caller == "lofreq":
        h += ["##fileDate=20190801", '##source=lofreq call --call-indels -f ref.fa -o out.vcf in.bam',
              "##reference=ref.fa",
              '##INFO=<ID=DP,Number=1,Type=Integer,Description="Raw Depth">',
              '##INFO=<ID=AF,Number=1,Type=Float,Description="Allele Frequency">',
              '##INFO=<ID=SB,Number=1,Type=Integer,Description="Phred-scaled strand bias at this position">',
              '##INFO=<ID=DP4,Number=4,Type=Integer,Description="Counts for ref-forward bases, ref-reverse, alt-forward and alt-reverse bases">',
              '##INFO=<ID=INDEL,Number=0,Type=Flag,Description="Indicates that the variant is an INDEL.">',
              '##FILTER=<ID=min_dp_10,Description="Minimum Coverage 10">']
    elif caller == "varscan":
        h += ["##source=VarScan2",
              '##INFO=<ID=DP,Number=1,Type=Integer,Description="Total depth of quality bases">',
              '##INFO=<ID=PV,Number=1,Type=Float,Description="Significance of variant read count vs. expected baseline error">',
              '##INFO=<ID=AF,Number=1,Type=Float,Description="Allele Frequency">',
              '##FILTER=<ID=str10,Description="Depth over 8, PV below 0.001, QUAL over 20, MQUAL over 20">']
    else:
        h += ["##source=%s" % caller, "##contig=<ID=%s,length=%d>" % (contig, L),
              '##INFO=<ID=DP,Number=1,Type=Integer,Description="Raw read depth">',
              '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">']
    cols = ["#CHROM", "POS", "ID", "REF", "ALT", "QUAL", "FILTER", "INFO"]
    if caller in ("bcftools", "freebayes", "gatk"):
        cols += ["FORMAT", sample]
    h.append("\t".join(cols))
    return h


def caller_row(rng, caller, contig, p, ident, ref, alt, qual):
    dp = rng.randint(20, 3000)
    if caller == "lofreq":
        info = "DP=%d;AF=%.6f;SB=%d;DP4=%d,%d,%d,%d" % (dp, rng.random(), rng.randint(0, 90), *[rng.randint(0, 900) for _ in range(4)])
        cols = [contig, str(p), ident, ref, alt, qual, "PASS", info]
    elif caller == "varscan":
        info = "DP=%d;PV=%.4g;AF=%.5f;DP4=%d,%d,%d,%d" % (dp, rng.random() * 1e-3, rng.random(), *[rng.randint(0, 900) for _ in range(4)])
        cols = [contig, str(p), ident, ref, alt, qual, "PASS", info]
    elif caller == "clc":
        cols = [contig, str(p), ident, ref, alt, qual, ".", "DP=%d;AF=%.2f" % (dp, 100 * rng.random())]
    else:
        info = "DP=%d;MQ=%d" % (dp, rng.randint(20, 60))
        cols = [contig, str(p), ident, ref, alt, qual, "." if caller != "gatk" else "PASS", info,
                "GT:PL" if caller == "bcftools" else "GT:DP:AD", "1:255,0" if caller == "bcftools" else "0/1:%d:%d,%d" % (dp, dp // 2, dp // 2)]
    return "\t".join(cols)


def caller_vcf(rng, caller, sample, contig, seq, snvs, n_records, frac_truth=0.8):
    """One caller x sample VCF, position sorted, with the quirks sprinkled in."""
    L = len(seq)
    recs = []  # (pos, line)
    take = [s for s in snvs if rng.random() < frac_truth]
    n_rand = max(0, n_records - len(take))
    for (p, r, a) in take:
        ident = "." if rng.random() > 0.01 else "rs%d" % rng.randint(1, 99999)
        recs.append((p, caller_row(rng, caller, contig, p, ident, r, a, qual_str(rng, caller))))
    for _ in range(n_rand):
        p = rng.randint(2, L - 40)
        r = seq[p - 1] if seq[p - 1] in BASES else "A"
        u = rng.random()
        if u < 0.80:
            a = rng.choice([b for b in BASES if b != r])
        elif u < 0.88:   # indel
            k = rng.randint(1, 4)
            if rng.random() < 0.5:
                r, a = seq[p - 1:p + k], r
            else:
                a = r + "".join(rng.choice(BASES) for _ in range(k))
        elif u < 0.92:   # multi-allelic
            a = ",".join(rng.sample([b for b in BASES if b != r], 2))
        elif u < 0.95:   # lower-case (Q4)
            r, a = r.lower(), rng.choice("acgt")
        elif u < 0.97:   # MNP
            r, a = seq[p - 1:p + 1], "".join(rng.choice(BASES) for _ in range(2))
        else:
            a = "N"
        q = qual_str(rng, caller) if rng.random() > 0.03 else "."
        ident = "." if rng.random() > 0.01 else "rs%d" % rng.randint(1, 99999)
        recs.append((p, caller_row(rng, caller, contig, p, ident, r, a, q)))
    for _ in range(max(1, len(recs) // 100)):  # duplicate lines (Q6)
        recs.append(rng.choice(recs))
    recs.sort(key=lambda x: x[0])
    return "\n".join(caller_header(caller, contig, L, sample) + [r[1] for r in recs]) + "\n"


QUAL_PROBES = ["20", "20.0", "2e1", "+20", " 20", "020", "20.", "19.999", "-1", ".5", "19.9999999999999999999",
               "PASS", "20abc", "nan", "inf", "30\r", "5\r", "10\r", "1,2", "", ".", "0x14", "0x13", "1e400", "+inf",
               "-inf", "1e-400", "2e-400", "20 ", "5 ", "20\x0b", "5\x0c", "3e", "e1", "+.5e2", "+", "..", "1.", "1.e1",
               "0x", "0X14", "0x1p5", "infinity", "NAN", "2_0", "9", "100", "1e2", "1d2", "077", "0b1",
               "2e1.", "0x14.", "2e400", "3e-310", "4.9e-324", "1.7976931348623158e308", "1.7976931348623159e308",
               "0x1f", "00x14", "19.99999999999999999999e0", "99999999999999999999999999999999", "+5", "-0", "- 30"]


NONCANON_MARKS = ("\t01000\t", "\t1-1000\t", "chrT\t999\t", "chrT\t998\t", "chrT\t996\t")


def quirks_case(canonical_only=False):
    """One line per quirk of SURVEY.md section 8a (Q1-Q11) plus the QUAL strnum probes.
    canonical_only drops the lines whose fgrep match hangs on non-canonical field
    alignment (Q10) or POS spelling -- the engine refuses those in strict mode."""
    truth = ["##fileformat=VCFv4.2", "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO",
             "chrT\t1000\t.\tA\tG\t30\tPASS\tDP=30;TYPE=SNV",
             "chrT\t1100\t.\tT\tC\t30\tPASS\tDP=30;TYPE=SNV",
             "chrT\t1200\t.\tG\tA,T\t30\tPASS\tDP=30;TYPE=SNV",      # multi-allelic truth (Q5)
             "chrT\t1300\t.\tGA\tG\t30\tPASS\tDP=30;TYPE=INDEL",
             "chrT\t1400\t.\tC\tT\t30\tPASS\tDP=30;TYPE=SNV",
             "chrT\t1400\t.\tC\tT\t30\tPASS\tDP=30;TYPE=SNV",       # duplicate truth row
             "chrT\t1500\t.\tc\tt\t30\tPASS\tDP=30;TYPE=SNV",       # lower-case truth: no pattern
             "chrT\t100\t.\tA\tC\t30\tPASS\tX",
             "chrT\t30\t.\tA\tC\t30\tPASS\tX",                       # pos "30" also appears as a QUAL below
             "chrT\t77\t.\tG\tT"]                                     # 5-column truth row
    v = ["##fileformat=VCFv4.2", "##source=quirks", "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO",
         "chrT\t1000\t.\tA\tG\t50\tPASS\tDP=9",            # plain TP
         "other\t1000\t.\tA\tG\t50\tPASS\tDP=9",           # Q1: CHROM not compared
         "chrT\t1100\trs1\tT\tC\t50\tPASS\tDP=9",          # Q2: ID != '.' -> FP
         "chrT\t1200\t.\tG\tA\t50\tPASS\tDP=9",            # Q5: multi-allelic truth never matches
         "chrT\t1200\t.\tG\tA,T\t50\tPASS\tDP=9",          # multi-allelic caller row dropped
         "chrT\t1300\t.\tGA\tG\t50\tPASS\tDP=9",           # indel dropped
         "chrT\t1400\t.\tC\tT\t50\tPASS\tDP=9",
         "chrT\t1400\t.\tC\tT\t50\tPASS\tDP=9",            # Q6: duplicate line, both emitted
         "chrT\t1400\t.\tC\tT\t19\tPASS\tDP=9",            # same key, fails QUAL
         "chrT\t1500\t.\tc\tt\t50\tPASS\tDP=9",            # Q4: lower-case dropped
         "#mid-file comment line",                          # Q7: header anywhere
         "chrT\t1600\t.\tA\tG",                             # Q8: <6 columns dropped
         "chrT\t1700\t.\tA\tG\t.",                          # 6 columns, QUAL '.'
         "chrT\t1000\t.\tA\tG\t50",                         # 6 columns TP, line ends after QUAL
         "chrT\t01000\t.\tA\tG\t50\tPASS\tDP=9",           # leading zero pos: different string
         "chrT\t11000\t.\tA\tG\t50\tPASS\tDP=9",           # truth pos is a suffix but inside a word
         "chrT\t1-1000\t.\tA\tG\t50\tPASS\tDP=9",          # Q10: suffix after a non-word char matches
         "chrT\t1000\t.\tA\tG\t50\tPASS\tDP=9\r",          # CRLF line kept verbatim
         "chrT\t999\t.\tA\tC\t30\t.\tA\tC\tmore",           # Q10: pattern '30\t.\tA\tC' at fields 6..9
         "chrT\t998\t.\tA\tC\t30\t.\tA\tC,T",               # Q10: ALT-like field followed by ','
         "chrT\t997\t.\tA\tC\t30\t.\tA\tCT",                # no: 'CT' is one word
         "chrT\t996\t.\tG\tT\t77\t.\tG\tT\tz",              # pattern from the 5-column truth row
         "\t100\t.\tA\tC\t40\tPASS\tX",                     # empty CHROM
         "chrT\t100\t.\tA\tC\t40\tPASS\tX\t",               # trailing tab
         "",                                                # empty line
         ]
    for i, q in enumerate(QUAL_PROBES):
        v.append("chrQ\t%d\t.\tA\tG\t%s\tPASS\tprobe%d" % (5000 + i, q, i))
    if canonical_only:
        v = [ln for ln in v if not any(m in ln for m in NONCANON_MARKS)]
    vt = ("\n".join(v)).encode("latin1")   # NO trailing newline (Q7)
    return vt, ("\n".join(truth) + "\n").encode()


def utf8_case():
    """Valid UTF-8 in kept lines, headers and truth rows (round 6).  The reference's grep runs under the locale CPython exports
    (PEP 538: LC_CTYPE=C.UTF-8 here), so `fgrep -w` asks iswalnum() about the CHARACTER next to a match: a letter, a digit of
    another script or '_' abutting a pattern is a word character (no match there), an arrow or a no-break space is not."""
    hdr = "##fileformat=VCFv4.2\n##source=\u00fcber-caller \u65e5\u672c\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n"
    truth = hdr + "chrT\t1000\t.\tA\tG\t30\tPASS\tX\nchrT\t30\t.\tA\tC\t30\tPASS\tX\n" \
                  "chrT\t1100\t.\tT\tC\t30\tPASS\tNOTE=\u00e9\nchrT\t1200\t.\tG\tT\t30\tPASS\t\u2192\n"
    v = ["chrT\t1000\t.\tA\tG\t50\tPASS\tNOTE=\u00e9",           # TP, non-ASCII INFO
         "chrT\t1001\t.\tA\tG\t50\tPASS\tNOTE=\u65e5\u672c",    # FP, non-ASCII INFO
         "chr\u00dc\t1100\t.\tT\tC\t77\tPASS\tDP=1",             # TP, non-ASCII CHROM; the truth row's INFO is non-ASCII too
         "chrT\t1200\t\u00e9\tG\tT\t77\tPASS\tDP=1",             # ID is a non-ASCII string, not '.': FP
         "chrT\t995\t.\tA\tC\t\u00e930\t.\tA\tC\tz",           # a letter abuts the pattern '30 . A C' at the later fields: word
         "chrT\t994\t.\tA\tC\t\u219230\t.\tA\tC\tz",           # an arrow abuts it: not a word character -> TP
         "chrT\t993\t.\tA\tC\t30\t.\tA\tC\u00e9",               # a letter behind Z
         "chrT\t992\t.\tA\tC\t30\t.\tA\tC\u2192",               # an arrow behind Z -> TP
         "chrT\t991\t.\tA\tC\t\u066330\t.\tA\tC\tz",           # an Arabic-Indic digit abuts: alnum
         "chrT\t990\t.\tA\tC\t_30\t.\tA\tC\tz",                 # underscore
         "chrT\t989\t.\tA\tC\t\u00a030\t.\tA\tC\tz",           # no-break space -> TP
         "chrT\t988\t.\tA\tC\t\U0001d7d130\t.\tA\tC\tz",       # a four-byte character (mathematical bold digit three): alnum
         "chrT\t1000\t.\tA\tG\t\u00e9\tPASS\tX",                 # QUAL is a non-ASCII string: awk compares it bytewise with "20" -> kept
         "chrT\t1000\t.\tA\tG\t19\tPASS\tNOTE=\u00e9",           # fails QUAL
         "#comment \u00e9",
         "chrT\t1002\t.\tA\tG\t50\tPASS\t\u00e9\t.\tA\tG"]    # a letter field, then '. A G': no digits before the dot
    snps = "1000\tA\tG\t1\t1\t1\t9\t9\t1\t1\tr\u00e9f\tr2\n30\tA\tC\t1\t1\t1\t9\t9\t1\t1\tr1\tr2\n1100\tT\tC\t1\t1\t1\t9\t9\t1\t1\tr1\tr2\n"
    return (hdr + "\n".join(v) + "\n").encode("utf-8"), truth.encode("utf-8"), snps.encode("utf-8")


def snps_tsv(rng, snvs, ref1, ref2):
    """show-snps -CTHIlr dialect (eval_variant_custom.smk:51): 12 tab columns, SNPs only."""
    rows = []
    for (p, r, a) in snvs:
        rows.append("\t".join([str(p), r, a, str(p + 7), str(rng.randint(1, 99)), str(p), "235169", "237928",
                               "1", "1", ref1, ref2]))
    # rows whose patterns can never match a filtered record
    rows.insert(3, "\t".join(["4242", "N", "A", "4249", "5", "4242", "235169", "237928", "1", "1", ref1, ref2]))
    rows.insert(5, "\t".join(["4343", "a", "g", "4350", "5", "4343", "235169", "237928", "1", "1", ref1, ref2]))
    rows.insert(7, "\t".join(["4444", ".", "G", "4451", "5", "4444", "235169", "237928", "1", "1", ref1, ref2]))
    rows.insert(9, "\t".join(["4545", "G", ".", "4552", "5", "4545", "235169", "237928", "1", "1", ref1, ref2]))
    return "\n".join(rows) + "\n"


# ----------------------------------------------------------------------------
# running the reference
# ----------------------------------------------------------------------------
def run_reference(vcf_path, truth_path, mode, outdir, caller, expect_tp):
    """Runs the unmodified reference script; returns when every output is complete.
    fp/ must pre-exist (Snakemake makes it); the tp writer is not
    awaited by the script (extract_TP_FP_SNPs.py:55-57) so poll for it."""
    subprocess.check_call([sys.executable, SCRIPT, vcf_path, truth_path, mode, outdir, caller])
    if mode == "hcmv":
        d = os.path.dirname(vcf_path); base = os.path.basename(vcf_path)[:-4]
        filtered = vcf_path[:-4] + ".filtered.vcf"
    else:
        d = outdir; base = caller
        filtered = os.path.join(outdir, caller + ".filtered.vcf")
    fp = os.path.join(d, "fp", base + ".fp.vcf")
    tp = os.path.join(d, "tp", base + ".tp.vcf")
    if expect_tp:
        def nl(p):
            with open(p, "rb") as fh:
                return fh.read().count(b"\n")
        nh = sum(1 for ln in open(vcf_path, "rb").read().split(b"\n") if ln.startswith(b"#"))
        want = nl(filtered) - nl(fp) + nh
        t0 = time.time()
        while True:
            if os.path.exists(tp) and nl(tp) == want:
                time.sleep(0.05)
                if nl(tp) == want:
                    break
            if time.time() - t0 > 30:
                raise RuntimeError("tp writer did not finish: %s" % tp)
            time.sleep(0.02)
    return filtered, (tp if expect_tp else None), fp


def is_pure(vcf_path):
    return os.path.basename(vcf_path).split(".")[0].endswith(("-1-0", "-0-1"))


def store_case(family, manifest, work_root, rel_vcf, rel_truth, mode, rel_outdir, caller):
    """Run one case inside work_root and copy inputs + outputs under tests/golden/<family>/."""
    vcf = os.path.join(work_root, rel_vcf)
    truth = os.path.join(work_root, rel_truth)
    outdir = os.path.join(work_root, rel_outdir)
    os.makedirs(os.path.join(os.path.dirname(vcf) if mode == "hcmv" else outdir, "fp"), exist_ok=True)
    pure = is_pure(vcf)
    filtered, tp, fp = run_reference(vcf, truth, mode, outdir, caller, not pure)
    fam = os.path.join(HERE, family)
    entry = {"family": family, "mode": mode, "vcf": os.path.join("input", rel_vcf), "truth": os.path.join("input", rel_truth),
             "outdir": rel_outdir, "caller": caller, "pure": pure, "expected": {}}
    for kind, path in (("filtered", filtered), ("tp", tp), ("fp", fp)):
        if path is None:
            entry["expected"][kind] = None
            continue
        rel = os.path.relpath(path, work_root)
        dst = os.path.join(fam, "expected", rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copyfile(path, dst)
        entry["expected"][kind] = os.path.join("expected", rel)
    for rel in (rel_vcf, rel_truth):
        dst = os.path.join(fam, "input", rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        if not os.path.exists(dst):
            shutil.copyfile(os.path.join(work_root, rel), dst)
    manifest.append(entry)


def write(path, data):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "wb") as fh:
        fh.write(data if isinstance(data, bytes) else data.encode())


def gen_all():
    manifest = []
    for fam in ("quirks", "quirks_canon", "hcmv", "config1", "custom", "edge", "utf8"):
        shutil.rmtree(os.path.join(HERE, fam), ignore_errors=True)
    refs = {k: read_fasta(k) for k in FASTA}

    # ---- quirks -----------------------------------------------------------
    with tempfile.TemporaryDirectory() as w:
        v, t = quirks_case()
        write(os.path.join(w, "q/QK-1-10.R.q.vcf"), v)
        write(os.path.join(w, "nucmer/QK.maskrepeat.variants.vcf"), t)
        store_case("quirks", manifest, w, "q/QK-1-10.R.q.vcf", "nucmer/QK.maskrepeat.variants.vcf", "hcmv", "q", "q")
        # same VCF through custom mode against a .snps style truth
        snps = "1000\tA\tG\t1\t1\t1\t9\t9\t1\t1\tr1\tr2\n1100\tT\tC\t1\t1\t1\t9\t9\t1\t1\tr1\tr2\n" \
               "1400\tC\tT\t1\t1\t1\t9\t9\t1\t1\tr1\tr2\n30\tA\tC\t1\t1\t1\t9\t9\t1\t1\tr1\tr2\n" \
               "1500\tc\tt\t1\t1\t1\t9\t9\t1\t1\tr1\tr2\n5001\t.\tG\t1\t1\t1\t9\t9\t1\t1\tr1\tr2\n\n"
        write(os.path.join(w, "nucmer/r1_r2.maskrepeat.snps"), snps)
        write(os.path.join(w, "in/quirks.vcf"), v)
        store_case("quirks", manifest, w, "in/quirks.vcf", "nucmer/r1_r2.maskrepeat.snps", "custom", "callers", "quirks")
    with tempfile.TemporaryDirectory() as w:
        v, t = quirks_case(canonical_only=True)
        write(os.path.join(w, "q/QC-1-10.R.q.vcf"), v)
        write(os.path.join(w, "nucmer/QC.maskrepeat.variants.vcf"), t)
        store_case("quirks_canon", manifest, w, "q/QC-1-10.R.q.vcf", "nucmer/QC.maskrepeat.variants.vcf", "hcmv", "q", "q")
        write(os.path.join(w, "nucmer/r1_r2.maskrepeat.snps"), snps)
        write(os.path.join(w, "in/quirks_canon.vcf"), v)
        store_case("quirks_canon", manifest, w, "in/quirks_canon.vcf", "nucmer/r1_r2.maskrepeat.snps", "custom", "callers", "quirks_canon")

    # ---- config 1: TA-1-10 LoFreq vs TA truth ------------------
    with tempfile.TemporaryDirectory() as w:
        rng = random.Random(1)
        contig, seq = refs["AD169"]
        ttxt, snvs = truth_vcf(rng, contig, seq, refs["TB40E"][0], 3000, 300, 30)
        write(os.path.join(w, "nucmer/TA.maskrepeat.variants.vcf"), ttxt)
        write(os.path.join(w, "lofreq/TA-1-10.AD169.lofreq.vcf"), caller_vcf(rng, "lofreq", "TA-1-10", contig, seq, snvs, 2000 + 2400))
        store_case("config1", manifest, w, "lofreq/TA-1-10.AD169.lofreq.vcf", "nucmer/TA.maskrepeat.variants.vcf", "hcmv", "lofreq", "lofreq")

    # ---- config 2: 10 samples x 6 callers (small N so the repo stays small) --
    with tempfile.TemporaryDirectory() as w:
        rng = random.Random(100)
        truths = {}
        for mix, refname in (("TM", "Merlin"), ("TA", "AD169")):
            contig, seq = refs[refname]
            ttxt, snvs = truth_vcf(rng, contig, seq, refs["TB40E"][0], 160, 16, 4)
            truths[mix] = snvs
            write(os.path.join(w, "nucmer/%s.maskrepeat.variants.vcf" % mix), ttxt)
        i = 0
        for sample, refname in SAMPLE_REF.items():
            for caller in CALLERS:
                rng = random.Random(100 + i); i += 1
                contig, seq = refs[refname]
                mix = sample[:2]
                pure = sample.endswith(("-1-0", "-0-1"))
                # a pure-strain sample mapped to TB40E shares no coordinates with the mix truth
                snvs = truths[mix] if refname != "TB40E" else []
                rel = "%s/%s.%s.%s.vcf" % (caller, sample, refname, caller)
                write(os.path.join(w, rel), caller_vcf(rng, caller, sample, contig, seq, snvs, 60 if pure else 150,
                                                       frac_truth=0.05 if pure else 0.8))
                store_case("hcmv", manifest, w, rel, "nucmer/%s.maskrepeat.variants.vcf" % mix, "hcmv", caller, caller)

    # ---- custom mode (vareval) ---------------------------------------------
    with tempfile.TemporaryDirectory() as w:
        rng = random.Random(7)
        contig, seq = refs["Merlin"]
        _, snvs = truth_vcf(rng, contig, seq, refs["TB40E"][0], 300, 0, 0)
        write(os.path.join(w, "nucmer/Merlin.BAC_TB40E.GFP.maskrepeat.snps"), snps_tsv(rng, snvs, contig, refs["TB40E"][0]))
        for k, caller in enumerate(("lofreq", "varscan", "bcftools")):
            rng = random.Random(70 + k)
            rel = "in/TM-1-1.Merlin.%s.vcf" % caller
            write(os.path.join(w, rel), caller_vcf(rng, caller, "TM-1-1", contig, seq, snvs, 250))
            store_case("custom", manifest, w, rel, "nucmer/Merlin.BAC_TB40E.GFP.maskrepeat.snps", "custom", "callers",
                       "TM-1-1.Merlin.%s" % caller)

    # ---- valid UTF-8 (the reference's grep runs under C.UTF-8: PEP 538) -------
    with tempfile.TemporaryDirectory() as w:
        v, t, snps = utf8_case()
        write(os.path.join(w, "u/U8-1-10.R.u.vcf"), v)
        write(os.path.join(w, "nucmer/U8.maskrepeat.variants.vcf"), t)
        store_case("utf8", manifest, w, "u/U8-1-10.R.u.vcf", "nucmer/U8.maskrepeat.variants.vcf", "hcmv", "u", "u")
        write(os.path.join(w, "nucmer/r1_r2.maskrepeat.snps"), snps)
        write(os.path.join(w, "in/utf8.vcf"), v)
        store_case("utf8", manifest, w, "in/utf8.vcf", "nucmer/r1_r2.maskrepeat.snps", "custom", "callers", "utf8")
        write(os.path.join(w, "u/U8-1-0.R.u.vcf"), v)          # pure strain: filtered copied to fp
        store_case("utf8", manifest, w, "u/U8-1-0.R.u.vcf", "nucmer/U8.maskrepeat.variants.vcf", "hcmv", "u", "u")

    # ---- edge cases ----------------------------------------------------------
    with tempfile.TemporaryDirectory() as w:
        hdr = "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n"
        t_ok = hdr + "c\t10\t.\tA\tG\t30\tPASS\tX\nc\t20\t.\tC\tT\t30\tPASS\tX\n"
        write(os.path.join(w, "nucmer/ED.maskrepeat.variants.vcf"), t_ok)
        write(os.path.join(w, "nucmer/EE.maskrepeat.variants.vcf"), hdr)          # truth with no rows
        write(os.path.join(w, "nucmer/EZ.maskrepeat.variants.vcf"), "")           # zero-byte truth
        cases = {
            "e/ED-1-1.R.e.vcf": (hdr, "ED"),                                         # header only
            "e/ED-1-2.R.e.vcf": ("", "ED"),                                          # zero-byte VCF
            "e/ED-1-3.R.e.vcf": ("c\t10\t.\tA\tG\t30\tPASS\tX", "ED"),              # no header, no final newline
            "e/ED-1-4.R.e.vcf": (hdr + "c\t10\t.\tA\tG\t30\tPASS\tX\r\nc\t20\t.\tC\tT\t99\tPASS\tX\r\n", "ED"),   # CRLF
            "e/EE-1-1.R.e.vcf": (hdr + "c\t10\t.\tA\tG\t30\tPASS\tX\nc\t11\t.\tA\tG\t3\tPASS\tX\n", "EE"),        # empty truth: all FP
            "e/EZ-1-1.R.e.vcf": (hdr + "c\t10\t.\tA\tG\t30\tPASS\tX\n", "EZ"),
            "e/ED-1-0.R.e.vcf": (hdr + "c\t10\t.\tA\tG\t30\tPASS\tX\nc\t20\t.\tC\tT\t5\tPASS\tX\n", "ED"),        # pure strain
            "e/ED-0-1.R.e.vcf": (hdr + "c\t10\t.\tA\tG\t30\tPASS\tX\n", "ED"),                                     # pure strain
            "e/ED-1-5.R.e.vcf": (hdr + "c\t20\t.\tC\tT\t99\tPASS\tX\nc\t10\t.\tA\tG\t30\tPASS\tX\nc\t20\t.\tC\tT\t21\tPASS\tX\n", "ED"),  # unsorted
            "e/ED-1-6.R.e.vcf": (hdr + "".join("c\t%d\t.\tA\tG\t30\tPASS\tX\n" % 10 for _ in range(70)), "ED"),  # long same-pos run
        }
        for rel, (txt, mix) in cases.items():
            write(os.path.join(w, rel), txt)
            store_case("edge", manifest, w, rel, "nucmer/%s.maskrepeat.variants.vcf" % mix, "hcmv", "e", "e")

    with open(os.path.join(HERE, "manifest.json"), "w") as fh:
        json.dump(manifest, fh, indent=1, sort_keys=True)
    awkv = subprocess.run("awk -W version 2>&1 | head -1", shell=True, capture_output=True, text=True).stdout.strip()
    grepv = subprocess.run("grep --version | head -1", shell=True, capture_output=True, text=True).stdout.strip()
    bashv = subprocess.run("bash --version | head -1", shell=True, capture_output=True, text=True).stdout.strip()
    with open(os.path.join(HERE, "PROVENANCE.md"), "w") as fh:
        fh.write("# Golden fixture provenance\n\n"
                 "Generated by `tests/golden/make_golden.py`; expected outputs written by the reference script\n"
                 "`program/extract_TP_FP_SNPs.py` (hzi-bifo/Quasimodo v0.4.2) run unmodified in the build container.\n\n"
                 "* awk: %s\n* grep: %s\n* bash: %s\n* python: %s\n* locale: POSIX (CPython exports LC_CTYPE=C.UTF-8 to the reference's children: PEP 538)\n* cases: %d\n\n"
                 "mawk's numeric-string rule as observed here (probe list `QUAL_PROBES` in make_golden.py, outputs in\n"
                 "`quirks/expected/`): strip blanks; last char digit or '.', first char digit/+/-/.; glibc strtod must\n"
                 "consume the field; any ERANGE (overflow, underflow, subnormal) makes the field a plain string, which\n"
                 "is then compared bytewise (unsigned) with \"20\".  gawk differs on hex and overflow spellings.\n\n"
                 "Locale: the reference runs its children under whatever locale Python exports (CPython >= 3.7 coerces\n"
                 "POSIX to LC_CTYPE=C.UTF-8, PEP 538).  GNU grep then suppresses selected lines holding bytes that are\n"
                 "not valid UTF-8 ('binary file matches') and classifies non-ASCII letters as word characters, so the\n"
                 "reference's output for non-ASCII data lines is locale dependent.  Round 6: the `utf8` family holds VALID\n"
                 "UTF-8 in kept lines, headers and truth rows, written by the reference under that exported locale; the engine\n"
                 "and the oracle ask the same glibc question (iswalnum_l on C.UTF-8) about the character next to a match.\n"
                 "Kept data lines holding a NUL or an invalid UTF-8 sequence are still rejected instead of guessed about.\n"
                 % (awkv, grepv, bashv, sys.version.split()[0], len(manifest)))
    print("wrote %d cases" % len(manifest))


# ----------------------------------------------------------------------------
# fuzz: oracle vs the reference on random hostile inputs (nothing stored)
# ----------------------------------------------------------------------------
def fuzz(n_rounds, seed):
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    from oracle import qm_oracle as O
    rng = random.Random(seed)
    alphabet_field = ["A", "C", "G", "T", ".", "N", "a", "AC", "A,C", "", "PASS", "20", "19", "30", "5", "1e2", "0x14",
                      "rs1", "-", "1-5", "5_", "x.y", " 20", "20 ", "2e-400", "7\r", "100", "10", "1",
                      # valid UTF-8 (round 6): letters, a digit of another script, a four-byte digit, an arrow, a no-break space -- alone and
                      # abutting the digits / bases a pattern could end or begin with
                      "\u00e9", "\u00e95", "5\u00e9", "\u21925", "5\u2192", "\u06633", "\U0001d7d17", "\u00a02", "A\u00e9", "C\u2192", "\u65e5\u672c", "\u00e9A"]
    bad = 0
    for r in range(n_rounds):
        with tempfile.TemporaryDirectory() as w:
            npos = rng.randint(1, 30)
            def rnd_line(ncol_max=10):
                nc = rng.randint(1, ncol_max)
                cols = []
                for c in range(nc):
                    if c == 1 and rng.random() < 0.7:
                        cols.append(str(rng.randint(1, npos)))
                    elif c == 2 and rng.random() < 0.7:
                        cols.append(".")
                    elif c in (3, 4) and rng.random() < 0.8:
                        cols.append(rng.choice("ACGT"))
                    else:
                        cols.append(rng.choice(alphabet_field))
                return "\t".join(cols)
            vlines = []
            for _ in range(rng.randint(0, 60)):
                vlines.append("#" + rnd_line() if rng.random() < 0.1 else rnd_line())
            vtxt = "\n".join(vlines) + ("\n" if rng.random() < 0.8 else "")
            custom = rng.random() < 0.4
            tl = []
            for _ in range(rng.randint(0, 25)):
                if custom:
                    cols = [str(rng.randint(1, npos)) if rng.random() < 0.8 else rng.choice(alphabet_field),
                            rng.choice("ACGT.N") if rng.random() < 0.9 else rng.choice(alphabet_field),
                            rng.choice("ACGT.N") if rng.random() < 0.9 else rng.choice(alphabet_field)] + ["x"] * rng.choice([0, 9])
                    tl.append("\t".join(cols))
                else:
                    tl.append(rnd_line(8))
            ttxt = "\n".join(tl) + ("\n" if tl and rng.random() < 0.9 else "")
            name = rng.choice(["S-1-10", "S-1-1", "S-1-0", "S-0-1", "X"])
            vcf = os.path.join(w, "c", name + ".R.c.vcf")
            write(vcf, vtxt.encode("utf-8")); truth = os.path.join(w, "truth.x"); write(truth, ttxt.encode("utf-8"))
            os.makedirs(os.path.join(w, "c", "fp")); os.makedirs(os.path.join(w, "out", "fp"))
            mode = "custom" if custom else "hcmv"
            pure = is_pure(vcf)
            filtered, tp, fp = run_reference(vcf, truth, mode, os.path.join(w, "out"), "lab", not pure)
            of, ot, op, _ = O.extract_text(open(vcf, "rb").read(), open(truth, "rb").read(), custom=custom, pure_strain=pure)
            rf = open(filtered, "rb").read(); rp = open(fp, "rb").read(); rt = open(tp, "rb").read() if tp else None
            if (of, ot, op) != (rf, rt, rp):
                bad += 1
                keep = os.path.join(tempfile.gettempdir(), "qm_fuzz_fail_%d_%d" % (seed, r))
                shutil.copytree(w, keep, dirs_exist_ok=True)
                print("MISMATCH round", r, "kept at", keep, "filtered", of == rf, "tp", ot == rt, "fp", op == rp)
    print("fuzz rounds=%d mismatches=%d" % (n_rounds, bad))
    return bad


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--fuzz", type=int, default=0)
    ap.add_argument("--seed", type=int, default=12345)
    ap.add_argument("--no-gen", action="store_true")
    a = ap.parse_args()
    if not os.path.exists(SCRIPT):
        sys.exit("reference not mounted at %s: fixtures can only be regenerated in the build container" % REF)
    if not a.no_gen:
        gen_all()
    if a.fuzz:
        sys.exit(1 if fuzz(a.fuzz, a.seed) else 0)
