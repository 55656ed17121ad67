#!/usr/bin/env python3
"""run_benchmark.py -- the reference's CLI surface (run_benchmark.py:62-192) for the commands
that sit on the accelerated path.

  hcmv    -e variantcall   TP/FP extraction + caller_performance.tsv + FP overlap counts on the bundled VCFs
  vareval                  the same for user VCFs against a precomputed genome difference

Everything that is not this path (read mapping, variant calling, nucmer, assembly evaluation,
figures) is out of scope and reported as such; `asmeval` and `-e assembly` exit with status 2."""
import os
import sys

import click

wd = os.path.dirname(os.path.realpath(__file__))
sys.path.insert(0, wd)
VERSION = "0.4.2"
cd = os.getcwd()


def print_version(ctx, param, value):
    if not value or ctx.resilient_parsing:
        return
    click.echo("Version {}".format(VERSION))
    ctx.exit()


@click.group()
@click.option("--version", is_flag=True, callback=print_version, expose_value=False, is_eager=True, help="Print the version.")
def cli():
    pass


def common_options(f):
    for opt in reversed([
        click.option("-d", "--dryrun", is_flag=True, default=False, show_default=True, help="Print the details without run the pipeline."),
        click.option("-t", "--threads", type=int, default=2, show_default=True, help="The number of threads to use."),
        click.option("-c", "--conda_prefix", type=click.Path(exists=True), default=None, help="Accepted for compatibility; unused."),
        click.option("-o", "--outpath", type=click.Path(), default=None, help="The directory where to put the results."),
        click.option("--gpus", type=int, default=1, show_default=True, help="GPUs of this node to shard the VCFs over (one process each)."),
        click.option("--json", "json_out", type=click.Path(), default=None,
                     help="Also write the run's statistics as JSON: one row per VCF (counts, whether it was in position order) and "
                          "where the VCFs that were not went (bucket paths / radix sort)."),
    ]):
        f = opt(f)
    return f


def _write_json(path, command, jobs, workflow_fn):
    """--json: a side file; the declared outputs are untouched (metrics / logging)."""
    import json
    from quasimodo_amd.extract import extract_many
    res = getattr(workflow_fn, "last_result", None)
    paths = res.get("paths") if res else extract_many.last_paths
    rows = []
    for j in jobs or []:
        st = j.stats or {}
        row = {k: int(st[k]) for k in ("n_records", "n_pass", "tp_lines", "fp_lines", "TP_R", "FP_R", "genomediff", "truth_unique", "sorted") if k in st}
        rows.append(dict(vcf=j.vcf_file, filtered=j.filtered_out, tp=j.tp_out, fp=j.fp_out, pure_strain=bool(st.get("pure_strain")), **row))
    with open(os.path.join(cd, path), "w") as fh:
        json.dump({"command": command, "rows": rows, "unsorted_paths": paths,
                   "phases_seconds": None if res else extract_many.last_phases}, fh, indent=1)


def _fail(e):
    from datetime import datetime
    print("ERROR")
    print("{}\t{}\n".format(datetime.now().isoformat(" ", timespec="minutes"), e))
    raise RuntimeError(e)


@cli.command(help="Benchmarking for HCMV dataset")
@common_options
@click.option("-e", "--evaluation", required=True, type=click.Choice(["all", "variantcall", "assembly"]), help="The evaluation to run.")
@click.option("-s", "--slow", is_flag=True, default=False, show_default=True, help="Run the evaluation based on reads (not supported by this build).")
@click.option("--data", type=click.Path(), default=None, help="Unpacked bundle directory (default: <repo>/data/snp).")
def hcmv(evaluation, dryrun=False, conda_prefix=None, slow=False, outpath=None, threads=2, data=None, gpus=1, json_out=None):
    if slow:
        click.echo("--slow (reads -> VCF) is outside the accelerated path; not supported", err=True)
        sys.exit(2)
    if evaluation == "assembly":
        click.echo("assembly evaluation is outside the accelerated path; not supported", err=True)
        sys.exit(2)
    from quasimodo_amd import workflow
    if outpath:
        out = os.path.join(cd, outpath)
    else:   # rules/load_config.smk:5,17: config/config.yaml, relative to the workflow's directory
        cfg_file = os.path.join(wd, "config", "config.yaml")
        cfg = workflow.load_yaml(cfg_file) if os.path.exists(cfg_file) else {}
        out = os.path.join(wd, str(cfg.get("outpath") or "../revision_output_1"))
    try:
        # data/snp is unpacked from data/snp.tar.gz when it is not there yet (rules/load_config.smk:28-31)
        workflow.run_hcmv_variantcall.last_result = None
        jobs = workflow.run_hcmv_variantcall(data or os.path.join(wd, "data", "snp"), out, dryrun=dryrun, gpus=gpus if gpus > 1 else None)
        if json_out and not dryrun:
            _write_json(json_out, "hcmv", jobs, workflow.run_hcmv_variantcall)
    except Exception as e:
        _fail(e)
    if evaluation == "all":
        click.echo("assembly evaluation skipped: outside the accelerated path", err=True)


@cli.command(help="Variants benchmark for customized dataset")
@common_options
@click.option("-v", "--vcfs", type=str, help="Comma-separated list of VCF files.")
@click.option("-l", "--labels", help="Comma-separated list of labels of VCF.", default=None)
@click.option("-r", "--refs", type=str, help="Comma-separated list of reference genome files (used to name the genome difference).")
@click.option("--novenn", is_flag=True, help="Accepted for compatibility; no figure is drawn.")
@click.option("--snps", type=click.Path(), default=None,
              help="show-snps -CTHIlr table of the two references; default <outpath>/results/snp/nucmer/<g1>_<g2>.maskrepeat.snps")
@click.option("--config", type=click.Path(exists=True), default=None,
              help="YAML with vcfs / refs / outpath / labels for what the command line leaves out (default: config/customize_data.yaml).")
def vareval(dryrun=False, conda_prefix=None, vcfs=None, labels=None, refs=None, novenn=False, outpath=None, threads=2, snps=None, gpus=1,
            config=None, json_out=None):
    from quasimodo_amd import workflow
    try:
        # what the command line leaves out comes from config/customize_data.yaml (run_benchmark.py:153-166,
        # rules/load_config_custom.smk:3, eval_variant_custom.smk:3-34)
        cfg_file = config or os.path.join(wd, "config", "customize_data.yaml")
        cfg = workflow.load_yaml(cfg_file) if os.path.exists(cfg_file) else {}
        st = workflow.vareval_settings(vcfs, refs, outpath, labels, cfg, cd=cd, wd=wd)
        out = st["outpath"]
        if snps is None:
            if not st["refs"] or len(st["refs"]) != 2:
                raise workflow.PathNotGiven("The reference genome files or output directory are not specified.")
            g = [os.path.splitext(os.path.basename(r))[0] for r in st["refs"]]
            snps = os.path.join(out, "results", "snp", "nucmer", "%s_%s.maskrepeat.snps" % (g[0], g[1]))   # eval_variant_custom.smk:14-17,40
        else:
            snps = os.path.join(cd, snps)
        workflow.run_vareval.last_result = None
        jobs = workflow.run_vareval(st["vcfs"], snps, out, labels=st["labels"], dryrun=dryrun, gpus=gpus if gpus > 1 else None)
        if json_out and not dryrun:
            _write_json(json_out, "vareval", jobs, workflow.run_vareval)
    except Exception as e:
        _fail(e)


@cli.command(help="Assembly benchmark for customized dataset (not on the accelerated path)")
@common_options
@click.option("-s", "--scaffolds", type=str)
@click.option("-r", "--refs", type=str)
def asmeval(**kwargs):
    click.echo("asmeval is outside the accelerated path; not supported", err=True)
    sys.exit(2)


if __name__ == "__main__":
    cli()
