#!/usr/bin/env python
"""Drop-in for Quasimodo's program/extract_TP_FP_SNPs.py (same argv, same output
paths), backed by the MI355X engine instead of awk/fgrep.

    python program/extract_TP_FP_SNPs.py <vcf> <snps> {hcmv,custom} <outdir> <caller>

Exit status is non-zero on any error (the reference always exits 0)."""
import argparse
import os
import sys
from argparse import RawTextHelpFormatter

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main(argv=None):
    usage = r'''
    extract_TP_FP_SNPs.py --- extract the TP and FP SNPs from VCF input and output them as VCF files.

    Usage:
    python extract_TP_FP_SNPs.py <VCF input file> <SNPs file (genome differences)> <hcmv|custom> <outdir> <caller>
    '''
    parser = argparse.ArgumentParser(description=usage, formatter_class=RawTextHelpFormatter)
    parser.add_argument("vcffile", type=str, help="the input VCF file")
    parser.add_argument("snpfile", type=str, help="the input VCF file of genome differences from MUMmer")
    parser.add_argument("data", type=str, choices=["hcmv", "custom"], help="the source of input data")
    parser.add_argument("outdir", type=str, help="the output dir")
    parser.add_argument("caller", type=str, help="the label of the caller")
    args = parser.parse_args(argv)
    from quasimodo_amd import extract_tp_fp_custom_snp, extract_tp_fp_snp
    try:
        if args.data == "hcmv":
            extract_tp_fp_snp(args.vcffile, args.snpfile)
        else:
            extract_tp_fp_custom_snp(args.vcffile, args.snpfile, args.outdir, args.caller)
    except Exception as e:  # loud, unlike the reference
        sys.stderr.write("extract_TP_FP_SNPs.py: %s\n" % e)
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
