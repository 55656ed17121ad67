#!/usr/bin/env python3
"""Golden vectors for the SNP / indel splitters (rules/vis_eval_vcf.smk:25-86).

The rules are one-line awk programs; this script runs THIS image's awk (mawk 1.3.4 20200120,
which reads `{2,}` as literal text) with the rule's program text on committed inputs and stores
the outputs under tests/golden/split/expected/.  Inputs: two existing golden inputs plus a probe
file written here.  Re-run: python tests/golden/make_split_golden.py"""
import json
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
PROGRAMS = {
    "xsnp": r'''/^#.*/{print}$4~/^[actgACTG]$/&&$5~/^[actgACTG]$/''',
    "xindel": r'''/^#.*/{print}$4~/^[actgACTG]{2,}/||$5~/^[actgACTG]{2,}/''',
}
PROBE = "\n".join([
    "##fileformat=VCFv4.2",
    "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO",
    "c\t10\t.\tA\tG\t30\tPASS\tDP=1",
    "c\t11\t.\ta\tg\t30\tPASS\tDP=1",
    "c\t12\t.\tAC\tG\t30\tPASS\tDP=1",
    "c\t13\t.\tA\tGT\t30\tPASS\tDP=1",
    "c\t14\t.\tA\tG,T\t30\tPASS\tDP=1",
    "c\t15\t.\tA{2,}\tG\t30\tPASS\tDP=1",
    "c\t16\t.\tA\tc{2,}TT\t30\tPASS\tDP=1",
    "c\t17\t.\tN\tG\t30\tPASS\tDP=1",
    "c\t18\t.\tAC,G\tT\t30\tPASS\tDP=1",
    "c\t19\t.\tA\tG",
    "c\t20\t.\tA",
    "",
    "#mid\t1\t.\tA\tG\tx",
    "#mid2\t1\t.\tT{2,}\tG\tx",
    "c\t21\t.\tA\tG\r",
    "c\t22\t.\tT\tC\t.\t.\t.\r",
    "c\t23\t.\tT\tC\t1",          # no final newline
])


def main():
    out = os.path.join(HERE, "split")
    os.makedirs(os.path.join(out, "input"), exist_ok=True)
    os.makedirs(os.path.join(out, "expected"), exist_ok=True)
    with open(os.path.join(out, "input", "probe.vcf"), "w", newline="") as fh:
        fh.write(PROBE)
    inputs = {
        "probe": "split/input/probe.vcf",
        "lofreq": "config1/input/lofreq/TA-1-10.AD169.lofreq.vcf",
        "truth": "config1/input/nucmer/TA.maskrepeat.variants.vcf",
        "quirks": "quirks/input/q/QK-1-10.R.q.vcf",
    }
    awkv = subprocess.run("awk -W version 2>&1 | head -1", shell=True, capture_output=True, text=True).stdout.strip()
    manifest = {"awk": awkv, "flavour": "mawk-literal", "cases": []}
    for name, rel in inputs.items():
        for kind, prog in PROGRAMS.items():
            res = subprocess.run(["awk", "-F\t", prog, os.path.join(HERE, rel)], capture_output=True, check=True)
            erel = "split/expected/%s.%s.vcf" % (name, kind)
            with open(os.path.join(HERE, erel), "wb") as fh:
                fh.write(res.stdout)
            manifest["cases"].append({"input": rel, "kind": kind, "expected": erel})
    with open(os.path.join(out, "manifest.json"), "w") as fh:
        json.dump(manifest, fh, indent=1, sort_keys=True)
    print("wrote %d cases with %s" % (len(manifest["cases"]), awkv))


if __name__ == "__main__":
    main()
