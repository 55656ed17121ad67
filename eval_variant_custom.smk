# eval_variant_custom.smk -- the vareval workflow (run_benchmark.py vareval) with the path's two rules re-authored.
#
# The reference's workflow of the same name: load_config_custom.smk (config/customize_data.yaml: refs, outpath, threads),
# vcfs / labels / novenn from the config, `rule gdiff` (nucmer + show-snps -> <g1>_<g2>.maskrepeat.snps), `rule extract_TP` (one
# worker process per caller VCF, :58-74) and `rule snp_benchmark` (scripts/custom_snp_benchmark.R, :76-92).  Here:
#   * the settings come through quasimodo_amd.workflow.vareval_settings (the same precedence, restated from
#     rules/load_config_custom.smk:1-19 and :3-34 of the reference's workflow; same two error messages);
#   * `rule gdiff` is the reference's tool invocation (nucmer / delta-filter / show-snps are upstream of the path: not rebuilt);
#   * `rule extract_TP` declares the reference's outputs (:63-64) for EVERY caller at once and hands them to one engine batch;
#   * `rule snp_benchmark` declares the reference's outputs (:82-83): the table from the engine's counts over the filtered VCFs
#     (as R reads them), the figure as a text page of the same table (drawing is R's, out of scope).
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


rule all:
    input:
        snp_benchmark_figure = results_dir + "/final_figures/snpcall_benchmark.pdf",
        snp_benchmark_table = results_dir + "/final_tables/snpcall_benchmark.txt"

# The first given ref should be the ref used to generate the VCFs
rule gdiff:
    input:
        refs
    output:
        delta = snp_dir + "/nucmer/" + gdiff_name + ".delta",
        snps = snp_dir + "/nucmer/" + gdiff_name + ".maskrepeat.snps"
    params:
        genome_diff_prefix = snp_dir + "/nucmer/" + gdiff_name
    shell:
        """
        nucmer --prefix={params.genome_diff_prefix} {input}
        show-snps -CTHIlr <(delta-filter -r -q {output.delta}) > {output.snps}
        """

rule extract_TP:
    input:
        vcf = [caller_vcf_dict[c] for c in callers],
        genome_diff = rules.gdiff.output.snps
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
        genome_diff = rules.gdiff.output.snps
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
