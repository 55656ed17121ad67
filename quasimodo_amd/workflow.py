"""The path's slice of the two Snakemake workflows, without Snakemake.

  hcmv -e variantcall (bundled-VCF mode): eval_variantcall.smk cp_vcf / cp_genome_diff (:62-93) ->
      rules/extract_TP.smk -> rules/vis_eval_vcf.smk snp_evaluate (table only) -> rules/compare_FP.smk
  vareval: eval_variant_custom.smk extract_TP (:58-74) -> snp_benchmark (table only, :76-92)

Everything upstream (reads -> VCF, nucmer) and every figure is out of scope:
the inputs must already exist.  Every mixed-sample VCF of a run goes through ONE engine batch."""
import glob
import os
import shutil

from .engine import Engine
from .extract import Job, extract_many, is_pure_strain
from .tables import r_hostile_rows, write_caller_performance, write_fp_overlap, write_snpcall_benchmark, write_weighted_roc
from .vcfio import scan_vcf

SAMPLE_REF = {  # rules/load_config.smk:20-23
    "TM-0-1": "Merlin", "TM-1-1": "Merlin", "TM-1-10": "Merlin", "TM-1-50": "Merlin", "TM-1-0": "TB40E",
    "TA-1-0": "TB40E", "TA-1-1": "AD169", "TA-1-10": "AD169", "TA-1-50": "AD169", "TA-0-1": "AD169"}
SNPCALLERS = ["lofreq", "varscan", "clc", "bcftools", "freebayes", "gatk"]   # eval_variantcall.smk:10
FP_COMPARED = ["lofreq", "clc", "varscan", "freebayes"]                       # rules/compare_FP.smk:1


class WorkflowError(RuntimeError):
    pass


def ensure_bundle(data_dir):
    """rules/load_config.smk:28-31: when data/snp is absent, the bundle data/snp.tar.gz is unpacked beside it (the reference
    shells `tar -xzvf data/snp.tar.gz -C data/`).  Returns data_dir; raises when neither exists."""
    data_dir = data_dir.rstrip("/")
    if os.path.isdir(data_dir):
        return data_dir
    tarball = data_dir + ".tar.gz"
    if not os.path.exists(tarball):
        raise WorkflowError("neither %s nor %s exists" % (data_dir, tarball))
    import tarfile
    parent = os.path.dirname(data_dir) or "."
    with tarfile.open(tarball, "r:gz") as tf:
        try:
            tf.extractall(parent, filter="data")   # (members that would leave `parent` are refused)
        except TypeError:                           # a Python without extraction filters: the same refusals by hand
            root = os.path.realpath(parent)
            inside = lambda p: p == root or p.startswith(root + os.sep)     # ('.' and './' members of `tar -C dir .` resolve to the root itself)
            for m in tf.getmembers():
                dest = os.path.realpath(os.path.join(parent, m.name))
                if not inside(dest):
                    raise WorkflowError("%s: member %s would be written outside %s" % (tarball, m.name, parent))
                if m.issym() or m.islnk():          # a link is fine as long as what it points at stays inside
                    target = m.linkname if m.islnk() else os.path.join(os.path.dirname(m.name), m.linkname)
                    if os.path.isabs(m.linkname) or not inside(os.path.realpath(os.path.join(parent, target))):
                        raise WorkflowError("%s: link %s points outside %s" % (tarball, m.name, parent))
                elif not (m.isfile() or m.isdir()):
                    raise WorkflowError("%s: member %s is neither a file, a directory nor a link" % (tarball, m.name))
            tf.extractall(parent)
    if not os.path.isdir(data_dir):
        raise WorkflowError("%s does not hold a directory %s" % (tarball, os.path.basename(data_dir)))
    return data_dir


def load_yaml(path):
    import yaml
    with open(path) as fh:
        return yaml.safe_load(fh) or {}


class PathNotGiven(WorkflowError):
    """eval_variant_custom.smk:6-17 / :24-28 (same messages)."""


def vareval_settings(vcfs=None, refs=None, outpath=None, labels=None, config=None, cd=None, wd=None):
    """What `run_benchmark.py vareval` runs on: the command line where it says something, config/customize_data.yaml where it
    does not (run_benchmark.py:153-166: only vcfs / refs / outpath travel from the command line, made absolute against the
    caller's directory; rules/load_config_custom.smk:3 + eval_variant_custom.smk:3-17,24-34 take the rest, and whatever the
    command line left out, from the config file, relative to the workflow's directory).
    One deliberate difference: the reference DROPS `-l/--labels` given on the command line (run_benchmark.py:157-165,
    `else: continue`) and always uses the config file's; here a command-line value wins.
    vcfs / refs / labels: comma-separated strings or None; config: the YAML as a dict.
    Returns {"vcfs": [...], "refs": [...] or None, "outpath": str, "labels": [...] or None}."""
    config = config or {}
    cd = cd or os.getcwd()
    wd = wd or cd

    def items(val, base):
        return [os.path.join(base, x.strip()) for x in val.split(",")]

    def pick(name, cli):
        if cli is not None:
            return cli, cd
        v = config.get(name)
        return (v, wd) if isinstance(v, str) else (None, wd)

    r, rbase = pick("refs", refs)
    o, obase = pick("outpath", outpath)
    if o is None:
        raise PathNotGiven("The reference genome files or output directory are not specified.")
    v, vbase = pick("vcfs", vcfs)
    if v is None:
        raise PathNotGiven("The VCF files from SNP calling are not specified.")
    lab = labels if labels is not None else config.get("labels")
    return {"vcfs": items(v, vbase), "refs": items(r, rbase) if r is not None else None, "outpath": os.path.join(obase, o.rstrip("/")),
            "labels": [x for x in lab.split(",")] if isinstance(lab, str) and lab else None}


def _fp_keys(path):
    """snpcaller_fp_compare.R:36-47: pos/ref/alt of the data rows with single-base alleles."""
    with open(path, "rb") as fh:
        sv = scan_vcf(fh.read())
    keep = (sv.ref < 4) & (sv.alt < 4) & ((sv.flags & 4) == 0)
    return sv.pos[keep], sv.ref[keep], sv.alt[keep]


def _flag_truth_rows(jobs):
    """stats["truth_r_hostile"]: rows of the job's truth file that R's read.table would not read as tab-split text
    (tables.check_r_readable); pure-strain samples never read theirs."""
    seen = {}
    for j in jobs:
        if j.stats is None or j.stats.get("pure_strain"):
            continue
        if j.snp_file not in seen:
            seen[j.snp_file] = r_hostile_rows(j.snp_file)
        j.stats["truth_r_hostile"] = seen[j.snp_file]


def fp_overlap_tables(engine, fp_files_by_sample, callers):
    out = {}
    for sample, files in fp_files_by_sample.items():
        sets = [_fp_keys(files[c]) for c in callers]
        out[sample] = engine.fp_overlap(sets)
    return out


def hcmv_rank_post(engine, jobs, indices, args):
    """What a rank of the multi-GPU workflow does with its own VCFs once they are extracted, while its engine is alive:
    the FP overlap of its samples (rules/compare_FP.smk:5-8 compares the callers of ONE sample, and the workflow dealt the
    VCFs by sample: no exchange) and the xindel sweeps of its VCFs.  Returns {"overlap": {sample: [region sizes]}}."""
    meta = [args["meta"][i] for i in indices]
    cmp_callers = args["cmp_callers"]
    out = {}
    if len(cmp_callers) >= 2:
        by_sample = {}
        for (c, s), j in zip(meta, jobs):
            if c in cmp_callers and not s.endswith(("-1-0", "-0-1")):
                by_sample.setdefault(s, {})[c] = j.fp_out
        for s, files in by_sample.items():
            if len(files) != len(cmp_callers):
                raise WorkflowError("sample %s: the compared callers are not all on one rank" % s)
            out[s] = [int(x) for x in fp_overlap_tables(engine, {s: files}, cmp_callers)[s]]
    if args.get("indel_roc", True):
        indel_roc(engine, [(c, s, j) for (c, s), j in zip(meta, jobs) if not j.stats.get("pure_strain")], args["snp_dir"])
    return {"overlap": out}


def run_hcmv_variantcall(data_dir, outpath, callers=None, engine=None, dryrun=False, gpus=None, _body=None, _backend="nccl",
                         _same_device=False):
    """data_dir: the unpacked bundle (data/snp): vcf/{caller}/{sample}.{ref}.{caller}.vcf and
    nucmer/{TM,TA}.maskrepeat.variants.vcf (rules/load_config.smk:28-36); when it is absent and <data_dir>.tar.gz exists,
    that is unpacked first (:28-31).
    gpus > 1: one process per GPU (quasimodo_amd.multigpu); the VCFs are dealt by SAMPLE (longest first), so the four
    compared callers of a sample meet on one rank and the FP overlap needs no exchange; every rank writes its own files,
    the confusion counters go through the one all-reduce, the rows come to this process for the three tables."""
    callers = list(callers or SNPCALLERS)
    data_dir = ensure_bundle(data_dir)
    results = os.path.join(outpath.rstrip("/"), "results")
    snp_dir = os.path.join(results, "snp")
    call_dir = os.path.join(snp_dir, "callers")
    samples = sorted(s for s in (os.path.basename(p).split(".")[0] for p in glob.glob(os.path.join(data_dir, "vcf", "clc", "*.clc.vcf")))
                     if s in SAMPLE_REF)
    if not samples:
        raise WorkflowError("no bundled VCFs under %s/vcf/clc" % data_dir)
    plan = []
    for s in samples:
        for c in callers:
            src = os.path.join(data_dir, "vcf", c, "%s.%s.%s.vcf" % (s, SAMPLE_REF[s], c))
            if not os.path.exists(src):
                raise WorkflowError("missing input %s" % src)
            plan.append((s, c, src))
    if dryrun:
        for s, c, src in plan:
            print("extractTP\t%s\t%s" % (c, src))
        return None
    os.makedirs(os.path.join(snp_dir, "nucmer"), exist_ok=True)
    for mix in ("TM", "TA"):                                           # cp_genome_diff
        src = os.path.join(data_dir, "nucmer", "%s.maskrepeat.variants.vcf" % mix)
        if os.path.exists(src):
            shutil.copyfile(src, os.path.join(snp_dir, "nucmer", os.path.basename(src)))
    jobs, meta = [], []
    for s, c, src in plan:                                              # cp_vcf
        d = os.path.join(call_dir, c)
        os.makedirs(os.path.join(d, "fp"), exist_ok=True)
        dst = os.path.join(d, os.path.basename(src))
        shutil.copyfile(src, dst)
        jobs.append(Job(dst, os.path.join(snp_dir, "nucmer", "%s.maskrepeat.variants.vcf" % s[:2]), "hcmv", d, c))
        meta.append((c, s))
    from .vcfio import split_variants
    for kind in ("xsnp", "xindel"):                                      # extract_snp / extract_indel / extract_nucmer_*:
        for j in jobs:                                                   # both declared outputs, *.vcf and its bgzip
            split_variants(j.vcf_file, j.vcf_file[:-4] + ".%s.vcf" % kind, kind, bgz=True, tbi="if-sorted")
        for mix in ("TM", "TA"):
            t = os.path.join(snp_dir, "nucmer", "%s.maskrepeat.variants.vcf" % mix)
            if os.path.exists(t):
                split_variants(t, os.path.join(snp_dir, "nucmer", "%s.maskrepeat.%s.vcf" % (mix, kind)), kind, bgz=True, tbi="if-sorted")
    mixed = [s for s in samples if not s.endswith(("-1-0", "-0-1"))]
    cmp_callers = [c for c in FP_COMPARED if c in callers]
    tables = os.path.join(results, "final_tables")
    if gpus is not None and (int(gpus) > 1 or _body):
        if engine is not None:
            raise ValueError("gpus > 1 starts one process (and one engine) per GPU: do not pass an engine")
        from .multigpu import extract_many_sharded
        groups = [[i for i, (c, s) in enumerate(meta) if s == smp] for smp in samples]
        jobs, res = extract_many_sharded(jobs, int(gpus), backend=_backend, body=_body, same_device=_same_device, groups=groups,
                                         post="quasimodo_amd.workflow:hcmv_rank_post",
                                         post_args=dict(meta=meta, cmp_callers=cmp_callers, snp_dir=snp_dir))
        os.makedirs(tables, exist_ok=True)
        _flag_truth_rows(jobs)
        write_caller_performance(os.path.join(tables, "caller_performance.tsv"), [(c, s, j.stats) for (c, s), j in zip(meta, jobs)])
        _write_snp_rocs(meta, jobs, snp_dir)
        if mixed and len(cmp_callers) >= 2:
            reg = {}
            for e in res["extras"]:
                reg.update((e or {}).get("overlap", {}))
            missing = [s for s in mixed if s not in reg]
            if missing:
                raise WorkflowError("no FP overlap came back for %s" % ", ".join(missing))
            write_fp_overlap(os.path.join(tables, "snpcaller_fp_snp_compare.txt"), {s: reg[s] for s in mixed}, cmp_callers)
        run_hcmv_variantcall.last_result = res
        return jobs
    own = engine is None
    if own:
        engine = Engine(int(os.environ.get("QM_DEVICE", "0")))
    try:
        extract_many(jobs, engine=engine)                                # extractTP, one batch
        os.makedirs(os.path.join(results, "final_tables"), exist_ok=True)
        _flag_truth_rows(jobs)
        write_caller_performance(os.path.join(results, "final_tables", "caller_performance.tsv"),
                                 [(c, s, j.stats) for (c, s), j in zip(meta, jobs)])
        _write_snp_rocs(meta, jobs, snp_dir)
        indel_roc(engine, [(c, smp, j) for (c, smp), j in zip(meta, jobs) if not j.stats.get("pure_strain")], snp_dir)
        if mixed and len(cmp_callers) >= 2:                              # compareFP (counts only)
            files = {s: {c: j.fp_out for (c, ss), j in zip(meta, jobs) if ss == s and c in cmp_callers} for s in mixed}
            reg = fp_overlap_tables(engine, files, cmp_callers)
            write_fp_overlap(os.path.join(results, "final_tables", "snpcaller_fp_snp_compare.txt"), reg, cmp_callers)
    finally:
        if own:
            engine.close()
    return jobs


run_hcmv_variantcall.last_result = None


def _write_snp_rocs(meta, jobs, snp_dir):
    """The engine's exact-match ROC sweeps, in the column layout of RTG's weighted_roc.tsv.gz but under a directory of
    their own: results/snp/rtg/ belongs to the reference's rtg rules (rules/vis_eval_vcf.smk:5-121), whose
    haplotype-aware numbers these are not."""
    for (c, smp), j in zip(meta, jobs):
        if not j.stats.get("pure_strain") and j.stats.get("roc") is not None:
            d = os.path.join(snp_dir, "qmvt_roc", c, "%s.%s.xsnp" % (smp, SAMPLE_REF[smp]))
            os.makedirs(d, exist_ok=True)
            write_weighted_roc(os.path.join(d, "exact_roc.tsv.gz"), j.stats["roc"], j.stats["truth_unique"])


def indel_roc(engine, items, snp_dir, n_bins=256):
    """The xindel counterpart of the exact-match ROC: the caller's *.xindel.vcf against the truth's *.xindel.vcf
    (extract_indel / extract_nucmer_indel, rules/vis_eval_vcf.smk:40-86) in the allele-extended mode -- REF and ALT
    matched as whole [ACGT]+ strings -- one batch for all VCFs.  Build-defined (the reference leaves indels to rtg
    vcfeval, rule rtg_indel); rows whose alleles are not [ACGT]+ (multi-allelic, lower case) take no part."""
    from .vcfio import AlleleDict, scan_truth, scan_vcf
    if not items:
        return
    adict = AlleleDict()
    tids, cols, keep = {}, [], []
    try:
        for c, smp, j in items:
            tfile = os.path.join(snp_dir, "nucmer", "%s.maskrepeat.xindel.vcf" % smp[:2])
            vfile = j.vcf_file[:-4] + ".xindel.vcf"
            if not (os.path.exists(tfile) and os.path.exists(vfile)):
                continue
            if tfile not in tids:
                with open(tfile, "rb") as fh:
                    tk = scan_truth(fh.read(), alleles=adict)
                tids[tfile] = engine.truth_load(tk.pos, tk.ref, tk.alt)
            with open(vfile, "rb") as fh:
                sv = scan_vcf(fh.read(), alleles=adict)
            cols.append(sv.columns)
            keep.append((c, smp, tids[tfile]))
        if not cols:
            return
        res, _ = engine.classify_batch(cols, [k[2] for k in keep], n_bins=n_bins, alleles=True)
        for (c, smp, tid), r in zip(keep, res):
            d = os.path.join(snp_dir, "qmvt_roc", c, "%s.%s.xindel" % (smp, SAMPLE_REF[smp]))
            os.makedirs(d, exist_ok=True)
            write_weighted_roc(os.path.join(d, "exact_roc.tsv.gz"), r["roc"], r["scalars"]["truth_unique"])
    finally:
        for t in tids.values():
            engine.truth_release(t)
        adict.close()


def run_vareval(vcfs, snps_file, outpath, labels=None, engine=None, dryrun=False, gpus=None, _body=None, _backend="nccl", _same_device=False):
    """eval_variant_custom.smk with the genome difference (show-snps -CTHIlr TSV) already computed.
    gpus > 1: the VCFs are dealt to that many GPUs (one process each); the rows come back for the table."""
    results = os.path.join(outpath.rstrip("/"), "results")
    call_dir = os.path.join(results, "snp", "callers")
    labels = list(labels) if labels else [os.path.splitext(os.path.basename(v))[0] for v in vcfs]
    if len(labels) != len(vcfs):
        raise WorkflowError("labels and vcfs differ in length")
    if dryrun:
        for lab, v in zip(labels, vcfs):
            print("extract_TP\t%s\t%s" % (lab, v))
        return None
    if not os.path.exists(snps_file) or os.path.getsize(snps_file) == 0:
        raise WorkflowError("No difference between two genomes!")       # custom_snp_benchmark.R:19-21
    os.makedirs(os.path.join(call_dir, "fp"), exist_ok=True)
    jobs = [Job(v, snps_file, "custom", call_dir, lab) for lab, v in zip(labels, vcfs)]
    if gpus is not None and (int(gpus) > 1 or _body):
        if engine is not None:
            raise ValueError("gpus > 1 starts one process (and one engine) per GPU: do not pass an engine")
        from .multigpu import extract_many_sharded
        jobs, res = extract_many_sharded(jobs, int(gpus), backend=_backend, body=_body, same_device=_same_device)
        run_vareval.last_result = res
    else:
        extract_many(jobs, engine=engine)
    os.makedirs(os.path.join(results, "final_tables"), exist_ok=True)
    _flag_truth_rows(jobs)
    write_snpcall_benchmark(os.path.join(results, "final_tables", "snpcall_benchmark.txt"),
                            [(lab, j.stats) for lab, j in zip(labels, jobs)])
    return jobs


run_vareval.last_result = None
