#!/usr/bin/env python3
"""Golden vectors for the allele-extended mode (QM_BATCH_ALLELES, BASELINE.json configs[4]).

The reference drops every non-single-base record, so it cannot produce these.  What this script
pins is the *mechanism*: the same five shell commands program/extract_TP_FP_SNPs.py:24-57 issues
(awk filter, grep header, awk truth patterns, fgrep -wf / -wvf), authored here, with the one change
that defines the mode -- `^[ACGT]$` widened to `^[ACGT]+$` in both awk programs -- run with this
image's mawk 1.3.4 / GNU grep 3.7 on committed inputs.  Outputs go to tests/golden/alleles/expected/.
Re-run: python tests/golden/make_alleles_golden.py"""
import json
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
FILTER = r'''awk -F"\t" '$4~/^[ACGT]+$/&&$5~/^[ACGT]+$/&&($6>=20||$6==".")' %s'''
PATTERNS = r'''awk -F"\t" '$4~/^[ACGT]+$/&&$5~/^[ACGT]+$/{print $2, ".", $4, $5}' OFS="\t" %s'''

LONG1 = "ACGTTGCAACGTAGCTAGCTAGGATC"          # 26 bases: dictionary alleles
LONG2 = "ACGTTGCAACGTAGCTAGCTAGGATG"          # differs in the last base
T_PROBE = "\n".join([
    "##fileformat=VCFv4.2",
    "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO",
    "c\t100\t.\tA\tG\t30\tPASS\tTYPE=SNV",
    "c\t100\t.\tA\tAC\t30\tPASS\tTYPE=INDEL",
    "c\t100\t.\tAT\tA\t30\tPASS\tTYPE=INDEL",
    "c\t250\t.\tGCGGAGA\tG\t30\tPASS\tTYPE=INDEL",
    "c\t300\t.\tA\t%s\t30\tPASS\tTYPE=INDEL" % LONG1,
    "c\t300\t.\t%s\tA\t30\tPASS\tTYPE=INDEL" % LONG1,
    "c\t410\t.\tA\tG,T\t30\tPASS\tTYPE=SNV",
    "c\t420\t.\tac\ta\t30\tPASS\tTYPE=INDEL",
    "c\t430\t.\tAN\tA\t30\tPASS\tTYPE=INDEL",
    "c\t500\t.\tACGTACGTACGTA\tA\t30\tPASS\tTYPE=INDEL",       # 13 bases: the longest inline allele
    "c\t500\t.\tACGTACGTACGTAC\tA\t30\tPASS\tTYPE=INDEL",      # 14: the shortest dictionary allele
    "c\t600\t.\tC\tCA\t30\tPASS\tTYPE=INDEL",
    "c\t600\t.\tC\tCA\t30\tPASS\tTYPE=INDEL",
]) + "\n"
V_PROBE = "\n".join([
    "##fileformat=VCFv4.2",
    "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO",
    "c\t100\t.\tA\tG\t50\tPASS\tDP=9",
    "c\t100\t.\tA\tAC\t50\tPASS\tDP=9",
    "c\t100\t.\tA\tACG\t50\tPASS\tDP=9",          # truth allele is a prefix
    "c\t100\t.\tAT\tA\t19\tPASS\tDP=9",           # fails QUAL
    "c\t100\t.\tAT\tA\t.\tPASS\tDP=9",
    "c\t100\trs9\tAT\tA\t60\tPASS\tDP=9",         # ID != '.'
    "c\t100\t.\tA\tAC\t70\tPASS\tDP=9",           # repeated line
    "c\t100\t.\tTA\tA\t70\tPASS\tDP=9",
    "d\t250\t.\tGCGGAGA\tG\t30\tPASS\tDP=9",      # other contig: CHROM is not compared
    "c\t250\t.\tGCGGAGA\tGA\t30\tPASS\tDP=9",
    "c\t250\t.\tCGGAGA\tG\t30\tPASS\tDP=9",
    "c\t300\t.\tA\t%s\t30\tPASS\tDP=9" % LONG1,
    "c\t300\t.\tA\t%s\t30\tPASS\tDP=9" % LONG2,
    "c\t300\t.\t%s\tA\t30\tPASS\tDP=9" % LONG1,
    "c\t300\t.\t%s\tA\t30\tPASS\tDP=9" % LONG1[:-1],
    "c\t410\t.\tA\tG\t30\tPASS\tDP=9",
    "c\t410\t.\tA\tG,T\t30\tPASS\tDP=9",
    "c\t420\t.\tac\ta\t30\tPASS\tDP=9",
    "c\t420\t.\tAC\tA\t30\tPASS\tDP=9",
    "c\t430\t.\tAN\tA\t30\tPASS\tDP=9",
    "c\t500\t.\tACGTACGTACGTA\tA\t30\tPASS\tDP=9",
    "c\t500\t.\tACGTACGTACGTAC\tA\t30\tPASS\tDP=9",
    "c\t500\t.\tACGTACGTACGTACG\tA\t30\tPASS\tDP=9",
    "c\t600\t.\tC\tCA\t20\tPASS\tDP=9",
    "c\t600\t.\tC\tCA\t20.0\tPASS\tDP=9",
    "c\t600\t.\tCA\tC\t20\tPASS\tDP=9",
    "c\t700\t.\tG\t<DEL>\t99\tPASS\tDP=9",
    "c\t700\t.\tG\t*\t99\tPASS\tDP=9",
    "c\t700\t.\tGT\t.\t99\tPASS\tDP=9",
    "c\t800\t.\tT\tTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTT\t99\tPASS\tDP=9",
]) + "\n"


def mixed_vcf(truth_path, seed=5):
    """A caller VCF over the config-1 truth: 80 % of its SNV and indel rows (some with another ALT,
    a non-'.' ID, a failing QUAL or twice), plus indels of its own; position sorted."""
    import random
    rng = random.Random(seed)
    rows = []
    for ln in open(truth_path):
        if ln.startswith("#"):
            continue
        c = ln.rstrip("\n").split("\t")
        if "," in c[4] or rng.random() > 0.8:
            continue
        pos, ref, alt = int(c[1]), c[3], c[4]
        r = rng.random()
        if r < 0.05:
            alt = alt + rng.choice("ACGT")
        elif r < 0.08 and len(ref) > 1:
            ref = ref[:-1]
        ident = "rs%d" % pos if rng.random() < 0.03 else "."
        qual = rng.choice(["19", "20", "3070", ".", "19.99", "255.7", "1e3"]) if rng.random() < 0.2 else str(rng.randint(20, 5000))
        rows.append((pos, ident, ref, alt, qual))
        if rng.random() < 0.03:
            rows.append((pos, ".", ref, alt, str(rng.randint(0, 40))))
    for _ in range(150):
        pos = rng.randint(1, 235000)
        ref = "".join(rng.choice("ACGT") for _ in range(rng.choice([1, 1, 2, 3, 5, 14, 20])))
        alt = ref[0] + "".join(rng.choice("ACGT") for _ in range(rng.choice([0, 1, 2, 4, 13, 15])))
        if alt != ref:
            rows.append((pos, ".", ref, alt, str(rng.randint(0, 300))))
    rows.sort(key=lambda r: r[0])
    head = "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n"
    return head + "".join("pHB5\t%d\t%s\t%s\t%s\t%s\tPASS\tDP=100\n" % r for r in rows)


def run_case(vcf, truth, outdir, name):
    os.makedirs(outdir, exist_ok=True)
    f = FILTER % vcf
    gs = PATTERNS % truth
    outs = {k: os.path.join(outdir, "%s.%s.vcf" % (name, k)) for k in ("filtered", "tp", "fp")}
    subprocess.run(["bash", "-c", '(grep -E "^#" %s;%s) > %s' % (vcf, f, outs["filtered"])], check=False)
    subprocess.run(["bash", "-c", '(grep -E "^#" %s;grep -F -wf <(%s) <(%s)) > %s' % (vcf, gs, f, outs["tp"])], check=False)
    subprocess.run(["bash", "-c", '(grep -E "^#" %s;grep -F -wvf <(%s) <(%s)) > %s' % (vcf, gs, f, outs["fp"])], check=False)
    return outs


def main():
    root = os.path.join(HERE, "alleles")
    os.makedirs(os.path.join(root, "input"), exist_ok=True)
    for nm, txt in (("probe.vcf", V_PROBE), ("probe.truth.vcf", T_PROBE)):
        with open(os.path.join(root, "input", nm), "w", newline="") as fh:
            fh.write(txt)
    with open(os.path.join(root, "input", "TA-1-10.AD169.mixed.vcf"), "w", newline="") as fh:
        fh.write(mixed_vcf(os.path.join(HERE, "config1/input/nucmer/TA.maskrepeat.variants.vcf")))
    cases = [
        ("mixed", "alleles/input/TA-1-10.AD169.mixed.vcf", "config1/input/nucmer/TA.maskrepeat.variants.vcf"),
        ("probe", "alleles/input/probe.vcf", "alleles/input/probe.truth.vcf"),
        ("config1", "config1/input/lofreq/TA-1-10.AD169.lofreq.vcf", "config1/input/nucmer/TA.maskrepeat.variants.vcf"),
        ("hcmv_varscan", "hcmv/input/varscan/TM-1-10.Merlin.varscan.vcf", "hcmv/input/nucmer/TM.maskrepeat.variants.vcf"),
    ]
    man = {"tools": subprocess.run("awk -W version 2>&1 | head -1; grep --version | head -1", shell=True, capture_output=True,
                                   text=True).stdout.strip().splitlines(), "cases": []}
    for name, vcf, truth in cases:
        if not os.path.exists(os.path.join(HERE, vcf)):
            raise SystemExit("missing input %s" % vcf)
        outs = run_case(os.path.join(HERE, vcf), os.path.join(HERE, truth), os.path.join(root, "expected"), name)
        nd = lambda p: sum(1 for ln in open(p, "rb") if not ln.startswith(b"#"))
        man["cases"].append({"name": name, "vcf": vcf, "truth": truth,
                             "expected": {k: os.path.relpath(v, HERE) for k, v in outs.items()},
                             "lines": {k: nd(v) for k, v in outs.items()}})
        print(name, man["cases"][-1]["lines"])
    with open(os.path.join(root, "manifest.json"), "w") as fh:
        json.dump(man, fh, indent=1, sort_keys=True)


def fuzz(rounds, seed):
    """Random VCF / truth pairs: the widened shell pipeline against tokenizer + oracle (extended) + writers."""
    import random
    import sys
    import tempfile
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import qm_oracle as O
    from quasimodo_amd import vcfio
    rng = random.Random(seed)
    alleles = ["A", "C", "G", "T", "AC", "CA", "ACG", "AAC", "GT", "TTTTTTTTTTTTT", "TTTTTTTTTTTTTT", "TTTTTTTTTTTTTTA",
               "ACGTACGTACGTACGTACGT", "a", "N", "AN", "A,C", "<DEL>", ".", "*", ""]
    quals = ["19", "20", "20.0", "19.999", ".", "300", "1e2", "PASS", "", "2e1", "0"]
    bad = 0
    with tempfile.TemporaryDirectory() as w:
        for it in range(rounds):
            span = rng.choice([30, 300, 5000])
            trows = ["#h"] + ["c\t%d\t%s\t%s\t%s\t30\tPASS\tX" % (rng.randint(1, span), rng.choice([".", ".", "rs1"]), rng.choice(alleles), rng.choice(alleles))
                              for _ in range(rng.randint(0, 60))]
            rows = []
            for _ in range(rng.randint(0, 120)):
                if trows[1:] and rng.random() < 0.4:
                    c = rng.choice(trows[1:]).split("\t")
                    pos, ref, alt = c[1], c[3], c[4]
                else:
                    pos, ref, alt = str(rng.randint(1, span)), rng.choice(alleles), rng.choice(alleles)
                rows.append((int(pos), "%s\t%s\t%s\t%s\t%s\t%s\tPASS\tDP=3" % (rng.choice("cd"), pos, rng.choice([".", ".", ".", "id"]), ref, alt, rng.choice(quals))))
            rows.sort(key=lambda r: r[0])
            vtxt = "##x\n#CHROM\n" + "".join(r[1] + "\n" for r in rows)
            vp, tp = os.path.join(w, "XX-1-10.R.c.vcf"), os.path.join(w, "t.vcf")
            open(vp, "w", newline="").write(vtxt)
            open(tp, "w", newline="").write("\n".join(trows) + "\n")
            outs = run_case(vp, tp, os.path.join(w, "o"), "f")
            d = vcfio.AlleleDict()
            sv = vcfio.scan_vcf(vtxt.encode(), alleles=d)
            tk = vcfio.scan_truth(open(tp, "rb").read(), alleles=d)
            cls, _, _ = O.classify_columns(*sv.columns, tk.pos, tk.ref, tk.alt, ext=True)
            for sel, k in ((0, "filtered"), (1, "tp"), (2, "fp")):
                o = os.path.join(w, "mine.vcf")
                sv.write(o, cls, sel)
                if open(o, "rb").read() != open(outs[k], "rb").read():
                    bad += 1
                    if bad <= 3:
                        print("MISMATCH round %d %s" % (it, k))
                        print(open(vp).read()); print(open(tp).read())
    print("fuzz: %d rounds, %d mismatches" % (rounds, bad))
    return bad


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 1 and sys.argv[1] == "--fuzz":
        raise SystemExit(1 if fuzz(int(sys.argv[2]) if len(sys.argv) > 2 else 200, int(sys.argv[3]) if len(sys.argv) > 3 else 1) else 0)
    main()
