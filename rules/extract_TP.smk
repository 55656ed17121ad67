# rules/extract_TP.smk -- drop-in for the reference's file of the same name (level 2 of INTEGRATION.md).
#
# The reference's rule runs `python program/extract_TP_FP_SNPs.py {input.vcf} {input.genome_diff} hcmv ...` once per
# {snpcaller} x {sample}, serialised by `threads: threads` (rules/extract_TP.smk:1-21 of the reference).  This rule declares the
# SAME files -- <snpcall_dir>/{snpcaller}/{sample}.{ref}.{snpcaller}.filtered.vcf and .../fp/{sample}.{ref}.{snpcaller}.fp.vcf,
# the patterns below are the reference's strings (:6-11) -- for every {snpcaller} x {sample_ref} of the run at once and hands all
# of them to ONE engine batch (quasimodo_amd.rules.extract_tp_hcmv -> qm_extract_files: one context, one upload, one launch
# sequence; the per-VCF worker pays the HIP start-up 60 times).  tp/{sample}.{ref}.{snpcaller}.tp.vcf is written beside them as
# the reference's worker does (undeclared there too, :8-9; scripts/mutation_context_profile.R reads it).
#
# Names it expects from the including workflow, as the reference's eval_variantcall.smk / rules/load_config.smk define them:
# snpcall_dir, snp_dir, snpcallers, sample_ref (["{sample}.{ref}", ...]), genome_diff_list (["TM", "TA"]), threads.
# QM_GPUS=<n> deals the VCFs to n GPUs of the node (one process each, one all-reduce; quasimodo_amd.multigpu).

EXTRACT_TP_FILTERED = snpcall_dir + "/{snpcaller}/{sample}.{ref}.{snpcaller}.filtered.vcf"
EXTRACT_TP_FP = snpcall_dir + "/{snpcaller}/fp/{sample}.{ref}.{snpcaller}.fp.vcf"


def _sample_and_ref(sample_ref_names):
    # "TA-1-10.AD169" -> ("TA-1-10", "AD169")
    pairs = [sr.split(".", 1) for sr in sample_ref_names]
    return [p[0] for p in pairs], [p[1] for p in pairs]


def _expand_per_caller(pattern):
    samples, refs = _sample_and_ref(sample_ref)
    return [pattern.format(snpcaller=c, sample=s, ref=r) for c in snpcallers for s, r in zip(samples, refs)]


rule extractTP:
    input:
        vcf = _expand_per_caller(snpcall_dir + "/{snpcaller}/{sample}.{ref}.{snpcaller}.vcf"),
        genome_diff = expand(snp_dir + "/nucmer/{mix}.maskrepeat.variants.vcf", mix=genome_diff_list)
    output:
        filtered = _expand_per_caller(EXTRACT_TP_FILTERED),
        fp = _expand_per_caller(EXTRACT_TP_FP)
    params:
        data = "hcmv"
    threads: threads
    run:
        import os
        from quasimodo_amd.rules import extract_tp_hcmv
        extract_tp_hcmv(input, output, params, threads=threads, gpus=int(os.environ.get("QM_GPUS", "1")) if os.environ.get("QM_GPUS") else None)
