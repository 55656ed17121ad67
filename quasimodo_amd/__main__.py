"""Batch command line over the engine: classify many caller VCFs against one truth file in ONE GPU batch.

    python -m quasimodo_amd extract --truth TM.maskrepeat.variants.vcf a.lofreq.vcf b.varscan.vcf ...
    python -m quasimodo_amd extract --truth r1_r2.maskrepeat.snps --custom OUTDIR --labels a,b a.vcf b.vcf
    python -m quasimodo_amd split in.vcf out.xsnp.vcf xsnp

`extract` writes the same three files per VCF as the reference worker (program/extract_TP_FP_SNPs.py:19-22,39-41;
custom :71-72,91) and prints one TSV row per VCF: the counts R derives from them
(scripts/caller_performance_compare.R:84-99).  `--alleles` switches the allele-extended mode on (include/qmvt.h)."""
import argparse
import json
import os
import sys


def _extract(a):
    from .extract import Job, extract_many
    vcfs = a.vcf
    if a.custom is not None:
        labels = a.labels.split(",") if a.labels else [os.path.splitext(os.path.basename(v))[0] for v in vcfs]
        if len(labels) != len(vcfs):
            sys.exit("--labels needs one label per VCF")
        os.makedirs(os.path.join(a.custom, "fp"), exist_ok=True)
        jobs = [Job(v, a.truth, "custom", a.custom, lab) for v, lab in zip(vcfs, labels)]
    else:
        jobs = [Job(v, a.truth, "hcmv") for v in vcfs]
    if a.gpus and a.gpus > 1:
        from .multigpu import extract_many_sharded
        jobs, res = extract_many_sharded(jobs, int(a.gpus), alleles=True if a.alleles else None)
        paths = res.get("paths")
    else:
        extract_many(jobs, alleles=True if a.alleles else None)
        paths = extract_many.last_paths
    cols = ("n_records", "n_pass", "tp_lines", "fp_lines", "TP_R", "FP_R", "genomediff", "truth_unique", "pure_strain", "sorted")
    print("\t".join(("vcf",) + cols))
    out = []
    for j in jobs:
        # ("sorted" is absent from a pure-strain row -- nothing was classified --; every other column must be there)
        row = {k: bool(j.stats[k]) if k == "pure_strain" else int(j.stats.get(k, 1)) if k == "sorted" else int(j.stats[k]) for k in cols}
        print("\t".join([j.vcf_file] + [str(row[k]) for k in cols]))
        out.append(dict(vcf=j.vcf_file, filtered=j.filtered_out, tp=j.tp_out, fp=j.fp_out, **row))
    if a.json:
        # unsorted_paths: where the VCFs that were not in position order went (include/qmvt.h QM_PATH_*): the bucket paths, or the
        # several times slower radix sort behind them ("radix", "radix_after_overflow")
        with open(a.json, "w") as fh:
            json.dump({"rows": out, "unsorted_paths": paths}, fh, indent=1)
    return 0


def _split(a):
    from .vcfio import split_variants
    n = split_variants(a.vcf, a.out, a.kind)
    print("%d lines -> %s" % (n, a.out))
    return 0


def main(argv=None):
    p = argparse.ArgumentParser(prog="python -m quasimodo_amd", description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = p.add_subparsers(dest="cmd", required=True)
    e = sub.add_parser("extract", help="filtered / tp / fp VCFs and counts for many VCFs in one GPU batch")
    e.add_argument("--truth", required=True, help="truth VCF (mummer2vcf) or, with --custom, the show-snps table")
    e.add_argument("--custom", metavar="OUTDIR", default=None, help="custom (vareval) mode: outputs under OUTDIR, truth is a .snps table")
    e.add_argument("--labels", default=None, help="comma-separated output labels for --custom")
    e.add_argument("--alleles", action="store_true", help="allele-extended mode: indels / MNPs matched exactly (off by default)")
    e.add_argument("--gpus", type=int, default=1, help="deal the VCFs to this many GPUs of the node: one process per GPU, one all-reduce "
                                                       "of the confusion counters (RCCL); the per-VCF rows reach the parent through the ranks' result files")
    e.add_argument("--json", default=None, help="also write {rows, unsorted_paths: where VCFs that were out of order went} as JSON")
    e.add_argument("vcf", nargs="+")
    e.set_defaults(fn=_extract)
    s = sub.add_parser("split", help="the extract_snp / extract_indel rules (rules/vis_eval_vcf.smk:25-86)")
    s.add_argument("vcf"); s.add_argument("out"); s.add_argument("kind", choices=["xsnp", "xindel"])
    s.set_defaults(fn=_split)
    a = p.parse_args(argv)
    return a.fn(a)


if __name__ == "__main__":
    sys.exit(main())
