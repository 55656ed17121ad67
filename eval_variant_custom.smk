# eval_variant_custom.smk -- the two rules of the vareval workflow (run_benchmark.py vareval) that lie on the path, re-authored.
#
# This file holds ONLY `rule extract_TP` and `rule snp_benchmark` and the names they need.  The reference's workflow of the same
# name also carries the target rule and the genome-difference rule (nucmer + show-snps, upstream of the path: not rebuilt); a
# maintainer keeps those two in their own file and replaces its two path rules with
#     include: "<this file>"
# behind them (INTEGRATION.md, level 2).  The table the genome-difference rule writes enters here as a PATH --
# `genome_diff_snps`, the name below is where the reference's rule puts it -- not as a reference to another rule's output.
#   * the settings come through quasimodo_amd.workflow.vareval_settings (the precedence of rules/load_config_custom.smk and of
#     the reference workflow's first lines, restated; same two error messages);
#   * `rule extract_TP` declares the reference's outputs (its :63-64) for EVERY caller at once and hands them to one engine batch;
#   * `rule snp_benchmark` declares the reference's outputs (its :82-83): the table from the engine's counts over the filtered
#     VCFs (as R reads them), the figure as a text page of the same table (drawing is R's, out of scope).
import os
from quasimodo_amd.workflow import vareval_settings

configfile: "config/customize_data.yaml"

_cfg = vareval_settings(config=config, cd=os.getcwd(), wd=os.getcwd())
if _cfg["refs"] is None:
    raise RuntimeError("The reference genome files or output directory are not specified.")
refs = _cfg["refs"]
vcfs = _cfg["vcfs"]
project_dir = _cfg["outpath"]
threads = config.get("threads", 1)
results_dir = os.path.join(project_dir, "results")
snp_dir = results_dir + "/snp"
snpcall_dir = snp_dir + "/callers"
g1_name, g2_name = (os.path.splitext(os.path.basename(ref))[0] for ref in refs)
gdiff_name = g1_name + "_" + g2_name
novenn = config.get("novenn")
callers = _cfg["labels"] if _cfg["labels"] else [os.path.splitext(os.path.basename(vcf))[0] for vcf in vcfs]
caller_vcf_dict = dict(zip(callers, vcfs))

genome_diff_snps = snp_dir + "/nucmer/" + gdiff_name + ".maskrepeat.snps"    # show-snps -CTHIlr of the two references: 12 tab-separated columns


rule extract_TP:
    input:
        vcf = [caller_vcf_dict[c] for c in callers],
        genome_diff = genome_diff_snps
    output:
        filtered = expand(snpcall_dir + "/{snpcaller}.filtered.vcf", snpcaller=callers),
        fp = expand(snpcall_dir + "/fp/{snpcaller}.fp.vcf", snpcaller=callers)
    params:
        data = "custom",
        outdir = snpcall_dir
    threads: threads
    run:
        from quasimodo_amd.rules import extract_tp_custom
        extract_tp_custom(input, output, params, threads=threads, callers=callers)

rule snp_benchmark:
    input:
        vcfs = expand(snpcall_dir + "/{snpcaller}.filtered.vcf",
                      snpcaller=callers),
        genome_diff = genome_diff_snps
    output:
        snp_benchmark_table = results_dir + "/final_tables/snpcall_benchmark.txt",
        snp_benchmark_figure = results_dir + "/final_figures/snpcall_benchmark.pdf"
    params:
        callers = callers,
        novenn = novenn,
        snp_venn_figure = results_dir + "/final_figures/snpcall_venn.pdf"
    run:
        from quasimodo_amd.rules import snp_benchmark
        snp_benchmark(input, output, params)
