"""The bodies of the re-authored Snakemake rules (rules/extract_TP.smk, rules/compare_FP.smk, eval_variant_custom.smk of THIS
repository): what a `run:` block calls with the rule's own `input` / `output` / `params` objects.

The reference runs one process per VCF (rules/extract_TP.smk:17-20: `python program/extract_TP_FP_SNPs.py ...` per
{snpcaller} x {sample}); a batch rule hands every VCF of the rule to ONE engine batch instead (qm_extract_files: one context,
one upload, one launch sequence) and declares the same files as its outputs.  Nothing here needs Snakemake: `input` / `output` /
`params` are read through attribute access only (`input.vcf`, `output.filtered`, ...), so the tests drive the same functions
with plain namespaces, and every function checks that the files it was asked for are the files the path writes
(extract_TP_FP_SNPs.py:19-22,39-41 derives its output names from the VCF's name, not from the rule).
"""
import os

from .extract import Job, _paths, extract_many, is_pure_strain


class RuleError(RuntimeError):
    pass


def _as_list(x):
    if x is None:
        return []
    if isinstance(x, (str, bytes, os.PathLike)):
        return [os.fspath(x)]
    return [os.fspath(p) for p in x]


def _get(obj, name, default=None):
    if obj is None:
        return default
    if isinstance(obj, dict):
        return obj.get(name, default)
    return getattr(obj, name, default)


def _same_files(kind, declared, derived):
    a, b = sorted(os.path.abspath(p) for p in declared), sorted(os.path.abspath(p) for p in derived)
    if a != b:
        only_a = [p for p in a if p not in b][:3]
        only_b = [p for p in b if p not in a][:3]
        raise RuleError("the rule declares other %s files than the path writes (declared only: %s; written only: %s)" % (kind, only_a, only_b))


def _truth_for(vcf, genome_diffs):
    """rules/extract_TP.smk:4: the truth VCF of a sample is <its first two letters>.maskrepeat.variants.vcf."""
    mix = os.path.basename(vcf)[:2]
    hit = [g for g in genome_diffs if os.path.basename(g).split(".")[0] == mix]
    if len(hit) != 1:
        raise RuleError("%s: %d of the rule's genome_diff inputs are named %s.*" % (vcf, len(hit), mix))
    return hit[0]


def extract_tp_hcmv(input, output, params=None, threads=1, gpus=None, engine=None):
    """rule extractTP (rules/extract_TP.smk:1-21) for EVERY {snpcaller} x {sample} of the run at once.
    input.vcf: the caller VCFs <snpcall_dir>/{snpcaller}/{sample}.{ref}.{snpcaller}.vcf; input.genome_diff: the truth VCFs
    <snp_dir>/nucmer/{mix}.maskrepeat.variants.vcf; output.filtered / output.fp: the declared files (:6-11); params.data must be
    "hcmv" (:13).  Returns the jobs (stats filled).  `threads` is the rule's (the engine takes the whole batch whatever it says)."""
    data = _get(params, "data", "hcmv")
    if data != "hcmv":
        raise RuleError("rules/extract_TP.smk is the hcmv rule (params.data = %r)" % (data,))
    vcfs, truths = _as_list(_get(input, "vcf")), _as_list(_get(input, "genome_diff"))
    jobs = [Job(v, _truth_for(v, truths), "hcmv") for v in vcfs]   # (pure-strain samples too: the rule names their mix's file, the worker never reads it)
    for j in jobs:
        _paths(j)
    _same_files("filtered", _as_list(_get(output, "filtered")), [j.filtered_out for j in jobs])
    _same_files("fp", _as_list(_get(output, "fp")), [j.fp_out for j in jobs])
    extract_many(jobs, engine=engine, gpus=gpus)
    return jobs


def extract_tp_custom(input, output, params=None, wildcards=None, threads=1, gpus=None, engine=None, callers=None):
    """rule extract_TP of eval_variant_custom.smk (:58-74).  The reference's rule is per {snpcaller}; the batch form takes
    every caller at once: input.vcf a list (the order of `callers`, the labels) -- or one path with wildcards.snpcaller, the
    reference's own shape.  input.genome_diff: the show-snps table (:61); output.filtered / output.fp (:63-64);
    params.outdir (:67), params.data = "custom" (:66)."""
    data = _get(params, "data", "custom")
    if data != "custom":
        raise RuleError("eval_variant_custom.smk's extract_TP is the custom rule (params.data = %r)" % (data,))
    outdir = _get(params, "outdir")
    if not outdir:
        raise RuleError("params.outdir is missing")
    vcfs = _as_list(_get(input, "vcf"))
    snps = _as_list(_get(input, "genome_diff"))
    if len(snps) != 1:
        raise RuleError("one genome_diff (.snps table) per run, got %d" % len(snps))
    labels = list(callers) if callers is not None else [_get(wildcards, "snpcaller")] if _get(wildcards, "snpcaller") else None
    if labels is None or len(labels) != len(vcfs):
        raise RuleError("one label per VCF (callers = [...], or wildcards.snpcaller for the per-caller form)")
    jobs = [Job(v, snps[0], "custom", outdir, lab) for v, lab in zip(vcfs, labels)]
    for j in jobs:
        _paths(j)
    _same_files("filtered", _as_list(_get(output, "filtered")), [j.filtered_out for j in jobs])
    _same_files("fp", _as_list(_get(output, "fp")), [j.fp_out for j in jobs])
    os.makedirs(os.path.join(outdir, "fp"), exist_ok=True)
    extract_many(jobs, engine=engine, gpus=gpus)
    return jobs


def _parse_fp_name(path):
    """<...>/{snpcaller}/fp/{sample}.{ref}.{snpcaller}.fp.vcf (rules/compare_FP.smk:5) -> (sample, caller)."""
    parts = os.path.basename(path).split(".")
    if len(parts) < 5 or parts[-2:] != ["fp", "vcf"]:
        raise RuleError("%s is not named {sample}.{ref}.{snpcaller}.fp.vcf" % path)
    return parts[0], parts[2]


def compare_fp(input, output, params=None, engine=None):
    """rule compareFP (rules/compare_FP.smk:3-19): the false positives of the compared callers (params.fp_compared_snpcallers,
    :1,17) of every mixed sample (params.mix_sample, :15-16), region by region of their Venn diagram
    (scripts/snpcaller_fp_compare.R:21-65: sets of pos-ref-alt keys of single-base rows).  The COUNTS are the path's
    (qm_fp_overlap on the device); the drawing is R's and out of scope: output.fp_compare_table (the reference's own commented
    line :11) takes the counts as a table, output.fp_compare_figure (:10) the same counts as text pages of a PDF, so that the
    rule leaves every output it declares."""
    from .engine import Engine
    from .pdftext import write_text_pdf
    from .tables import CALLER_MAP, write_fp_overlap
    from .workflow import fp_overlap_tables
    callers = list(_get(params, "fp_compared_snpcallers") or [])
    mixed = [s for s in (_get(params, "mix_sample") or []) if not s.endswith(("-1-0", "-0-1"))]
    files = {}
    for p in _as_list(_get(input, "fp")):
        smp, c = _parse_fp_name(p)
        files.setdefault(smp, {})[c] = p
    missing = [(s, c) for s in mixed for c in callers if c not in files.get(s, {})]
    if missing:
        raise RuleError("no fp.vcf among the inputs for %s" % ", ".join("%s/%s" % m for m in missing[:5]))
    own = engine is None
    if own:
        engine = Engine(int(os.environ.get("QM_DEVICE", "0")))
    try:
        reg = fp_overlap_tables(engine, {s: {c: files[s][c] for c in callers} for s in mixed}, callers)
    finally:
        if own:
            engine.close()
    table = _get(output, "fp_compare_table")
    figure = _get(output, "fp_compare_figure")
    if table:
        os.makedirs(os.path.dirname(os.fspath(table)) or ".", exist_ok=True)
        write_fp_overlap(os.fspath(table), reg, callers)
    if figure:
        os.makedirs(os.path.dirname(os.fspath(figure)) or ".", exist_ok=True)
        pages = []
        n = len(callers)
        for s in sorted(reg):
            lines = ["False-positive SNPs shared between callers -- sample %s" % s,
                     "(region sizes of the Venn diagram scripts/snpcaller_fp_compare.R draws; counts by quasimodo_amd, no drawing)", "",
                     "%-52s %10s" % ("callers (exactly these)", "SNPs")]
            for m in range(1, 1 << n):
                names = " & ".join(CALLER_MAP.get(callers[i], callers[i]) for i in range(n) if m >> i & 1)
                lines.append("%-52s %10d" % (names, int(reg[s][m])))
            pages.append(lines)
        write_text_pdf(os.fspath(figure), pages, title="snpcaller_fp_snp_compare")
    return reg


def snp_benchmark(input, output, params=None, engine=None):
    """rule snp_benchmark of eval_variant_custom.smk (:76-92; scripts/custom_snp_benchmark.R:23-95): every caller's
    <label>.filtered.vcf (input.vcfs, the order of params.callers) against the show-snps table (input.genome_diff) ->
    output.snp_benchmark_table (:82).  Like the R script it reads the FILES (whoever wrote them): rows with single-base REF / ALT
    are the caller's keys, TP / FP are set sizes on distinct keys.  output.snp_benchmark_figure (:83, a bar chart in R) takes the
    table as a text page of a PDF; params.snp_venn_figure (:88, written by R unless params.novenn) likewise the per-caller key
    counts -- drawing is out of scope, the declared files exist."""
    from .engine import Engine
    from .pdftext import write_text_pdf
    from .tables import r_hostile_rows, write_snpcall_benchmark
    from .vcfio import scan_truth, scan_vcf
    vcfs = _as_list(_get(input, "vcfs"))
    snps = _as_list(_get(input, "genome_diff"))
    callers = list(_get(params, "callers") or [])
    if len(callers) != len(vcfs):
        raise RuleError("params.callers and input.vcfs differ in length")
    if len(snps) != 1:
        raise RuleError("one genome_diff (.snps table) per run, got %d" % len(snps))
    if not os.path.exists(snps[0]) or os.path.getsize(snps[0]) == 0:
        raise RuleError("No difference between two genomes!")          # custom_snp_benchmark.R:19-21
    with open(snps[0], "rb") as fh:
        tk = scan_truth(fh.read(), custom=True)
    truth_hostile = r_hostile_rows(snps[0])
    scans = []
    for v in vcfs:
        with open(v, "rb") as fh:
            scans.append(scan_vcf(fh.read()))
    own = engine is None
    if own:
        engine = Engine(int(os.environ.get("QM_DEVICE", "0")))
    try:
        tid = engine.truth_load(tk.pos, tk.ref, tk.alt)
        try:
            res, _ = engine.classify_batch([sv.columns for sv in scans], [tid] * len(scans))
        finally:
            engine.truth_release(tid)
    finally:
        if own:
            engine.close()
    rows = []
    for lab, sv, r in zip(callers, scans, res):
        st = dict(r["scalars"])
        st.update(genomediff=tk.genomediff, pure_strain=False, r_hostile=sv.n_r_hostile, truth_r_hostile=truth_hostile)
        rows.append((lab, st))
    table = _get(output, "snp_benchmark_table")
    os.makedirs(os.path.dirname(os.fspath(table)) or ".", exist_ok=True)
    write_snpcall_benchmark(os.fspath(table), rows)
    figure = _get(output, "snp_benchmark_figure")
    if figure:
        os.makedirs(os.path.dirname(os.fspath(figure)) or ".", exist_ok=True)
        with open(os.fspath(table)) as fh:
            body = [ln.rstrip("\n").expandtabs(16) for ln in fh]
        write_text_pdf(os.fspath(figure), [["SNP calling benchmark (the table scripts/custom_snp_benchmark.R plots; no drawing)", ""] + body],
                       title="snpcall_benchmark")
    venn = _get(params, "snp_venn_figure")
    if venn and not _get(params, "novenn"):
        os.makedirs(os.path.dirname(os.fspath(venn)) or ".", exist_ok=True)
        lines = ["Caller SNP sets against the genome difference (what scripts/custom_snp_benchmark.R draws as a Venn diagram; counts only)", "",
                 "%-24s %12s %12s %12s" % ("caller", "distinct keys", "in truth", "not in truth")]
        for lab, st in rows:
            lines.append("%-24s %12d %12d %12d" % (lab, int(st["TP_R"]) + int(st["FP_R"]), int(st["TP_R"]), int(st["FP_R"])))
        lines.append("%-24s %12d" % ("Genome", int(tk.genomediff)))
        write_text_pdf(os.fspath(venn), [lines], title="snpcall_venn")
    return rows
