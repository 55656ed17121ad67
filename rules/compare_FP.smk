# rules/compare_FP.smk -- drop-in for the reference's file of the same name (level 2 of INTEGRATION.md).
#
# The reference's rule hands the fp.vcf files of every caller and mixed sample to scripts/snpcaller_fp_compare.R, which builds
# the sets of pos-ref-alt keys per caller and draws one Venn diagram per sample into final_figures/snpcaller_fp_snp_compare.pdf
# (rules/compare_FP.smk:3-19 of the reference; its table output is commented out, :11).  The SET ARITHMETIC is the path's
# (qm_fp_overlap on the device); the drawing is R's and out of scope.  This rule declares the reference's figure under the
# reference's string (:10) -- it holds the region sizes as text pages, so that the rule leaves what it declares -- and enables
# the reference's own table line (:11) for the same numbers as a TSV.
#
# Names it expects from the including workflow (eval_variantcall.smk / rules/load_config.smk of the reference):
# snpcall_dir, results_dir, snpcallers, sample_list, sample_refname_dict.

fp_compared_snpcallers = ["lofreq", "clc", "varscan", "freebayes"]

# the samples that are mixtures of two strains (the pure ones, *-1-0 / *-0-1, have no truth set and take no part)
mixed_samples = [s for s in sample_list if not s.endswith(("-1-0", "-0-1"))]

rule compareFP:
    input:
        fp = expand(snpcall_dir + "/{snpcaller}/fp/{sample_ref}.{snpcaller}.fp.vcf", snpcaller=snpcallers,
                    sample_ref=["%s.%s" % (s, sample_refname_dict[s]) for s in mixed_samples])
    output:
        fp_compare_figure = results_dir + "/final_figures/snpcaller_fp_snp_compare.pdf",
        fp_compare_table = results_dir + "/final_tables/snpcaller_fp_snp_compare.txt"
    params:
        mix_sample = mixed_samples,
        fp_compared_snpcallers = fp_compared_snpcallers
    run:
        from quasimodo_amd.rules import compare_fp
        compare_fp(input, output, params)
