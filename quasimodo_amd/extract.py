"""Host mirror of the reference worker program/extract_TP_FP_SNPs.py.

Same function names and argument meaning as the reference
(extract_tp_fp_snp :12, extract_tp_fp_custom_snp :60); the shell pipeline is
replaced by one text scan on the host + the HIP engine.  `extract_many` is the
batch form used by the rule bodies: every mixed-sample VCF of a run is
classified in ONE engine batch (the qm_batch_* entry points; files are read, scanned and written by a thread pool).

Differences from the reference, all deliberate:
  * errors raise (the CLI exits non-zero) instead of being ignored;
  * outputs are written atomically and tp/ is complete before returning
    (the reference does not wait for its tp writer, :55-57);
  * fp/ and tp/ are created when missing (the reference relies on Snakemake for fp/).
"""
import os
from dataclasses import dataclass, field

import numpy as np

from ._lib import SCALAR_NAMES, QmvtError
from .engine import Engine
from .vcfio import scan_vcf


def is_pure_strain(vcf_file):
    """extract_TP_FP_SNPs.py:33 -- sample name ends in -1-0 / -0-1: no truth comparison."""
    return os.path.basename(vcf_file).split(".")[0].endswith(("-1-0", "-0-1"))


@dataclass
class Job:
    vcf_file: str
    snp_file: str
    mode: str = "hcmv"        # "hcmv" | "custom"
    outdir: str = ""
    caller: str = ""
    # filled by extract_many
    filtered_out: str = ""
    tp_out: str = ""
    fp_out: str = ""
    stats: dict = field(default_factory=dict)


def _paths(job):
    """Output paths exactly as the reference derives them (:19-22,39-41 hcmv; :71-72,91 custom)."""
    if job.mode == "hcmv":
        dirname = os.path.dirname(job.vcf_file)
        base = os.path.basename(job.vcf_file)[:-4]
        job.filtered_out = job.vcf_file[:-4] + ".filtered.vcf"
        job.fp_out = os.path.join(dirname, "fp", base + ".fp.vcf")
        job.tp_out = os.path.join(dirname, "tp", base + ".tp.vcf")
    elif job.mode == "custom":
        job.filtered_out = os.path.join(job.outdir, job.caller + ".filtered.vcf")
        job.fp_out = os.path.join(job.outdir, "fp", job.caller + ".fp.vcf")
        job.tp_out = os.path.join(job.outdir, "tp", job.caller + ".tp.vcf")
    else:
        raise ValueError("data must be 'hcmv' or 'custom', got %r" % job.mode)


def _strict_default():
    return os.environ.get("QM_LENIENT", "0") in ("", "0")


def _io_threads():
    try:
        n = int(os.environ.get("QM_IO_THREADS", "0"))
    except ValueError:
        n = 0
    return n if n > 0 else min(16, os.cpu_count() or 1)


def _alleles_default():
    return os.environ.get("QM_ALLELES", "0") not in ("", "0")


def extract_many(jobs, engine=None, strict=None, n_bins=256, alleles=None, gpus=None, truth_slots=None, n_slots=0, global_dev=None):
    """Classify and write filtered / tp / fp VCFs for a list of Job.  Returns the jobs
    with .stats filled (line counts, R-path counts, ROC rows).
    gpus > 1: the VCFs are dealt to that many GPUs of this node, one process each (quasimodo_amd.multigpu).
    alleles=True (or QM_ALLELES=1): the allele-extended mode -- every record whose REF and ALT are
    [ACGT]+ takes part, not only single bases (a build-defined widening of the reference's filter,
    include/qmvt.h; hcmv mode only).
    truth_slots / n_slots / global_dev: this call is one rank's share of a multi-GPU run (Engine.extract_files)."""
    strict = _strict_default() if strict is None else strict
    alleles = _alleles_default() if alleles is None else bool(alleles)
    if gpus is not None and int(gpus) > 1:
        if engine is not None:
            raise ValueError("gpus > 1 starts one process (and one engine) per GPU: do not pass an engine")
        from .multigpu import extract_many_sharded
        return extract_many_sharded(jobs, int(gpus), n_bins=n_bins, alleles=alleles, strict=strict)[0]
    if alleles and any(j.mode != "hcmv" for j in jobs):
        raise ValueError("the allele-extended mode needs VCF truth sets (hcmv mode)")
    own = engine is None
    for job in jobs:
        _paths(job)
    if not jobs:
        return jobs
    pure = [is_pure_strain(j.vcf_file) for j in jobs]
    if engine is None:
        # a context is needed even for a batch of pure-strain samples only when something is to be classified
        engine = Engine(int(os.environ.get("QM_DEVICE", "0"))) if not all(pure) else None
    try:
        for job, p in zip(jobs, pure):
            os.makedirs(os.path.dirname(job.fp_out) or ".", exist_ok=True)
            os.makedirs(os.path.dirname(job.filtered_out) or ".", exist_ok=True)
            if not p:
                os.makedirs(os.path.dirname(job.tp_out) or ".", exist_ok=True)
        fj = [dict(vcf=j.vcf_file, truth=None if p else j.snp_file, mode=j.mode, pure=p, filtered=j.filtered_out,
                   tp=None if p else j.tp_out, fp=j.fp_out) for j, p in zip(jobs, pure)]
        extract_many.last_paths = None
        if engine is None:
            rows = _pure_only(fj, strict)
        else:
            before = engine.path_stats_total()
            rows, phases = engine.extract_files(fj, n_bins=n_bins, alleles=alleles, strict=strict, truth_slots=truth_slots, n_slots=n_slots,
                                                global_dev=global_dev)
            extract_many.last_phases = phases
            # where the VCFs found out of order went (bucket paths / radix sort: a silent fall onto the slow path shows here)
            extract_many.last_paths = {k: v - before[k] for k, v in engine.path_stats_total().items()}
    finally:
        if own and engine is not None:
            engine.close()
    for job, p, r in zip(jobs, pure, rows):
        job.stats = r
        job.stats.update(pure_strain=p)
        if p:
            job.tp_out = ""
            job.stats["roc"] = None
    return jobs


extract_many.last_phases = None
extract_many.last_paths = None


def _pure_only(file_jobs, strict):
    """Pure-strain samples need no device (extract_TP_FP_SNPs.py:33-36: fp is a copy of filtered): when a call holds
    nothing else, no context is created and the host side of the library does the work."""
    rows = []
    for j in file_jobs:
        with open(j["vcf"], "rb") as fh:
            sv = scan_vcf(fh.read())
        if sv.n_refused and strict:
            raise QmvtError(-8, "%s line %d: a kept line holds a NUL or bytes that are not valid UTF-8 -- the reference's answer for it depends on the "
                                "locale Python exports to grep; set QM_LENIENT=1 to classify it by its columns" % (j["vcf"], sv.first_refused_line))
        cls = (sv.flags & 1).astype(np.uint8)
        sv.write(j["filtered"], cls, 0)
        sv.write(j["fp"], cls, 0)
        npass = int(cls.sum())
        r = dict(zip(SCALAR_NAMES, (npass, 0, npass, 0, 0, 1, sv.n_records, 0)))
        r.update(n_lines=sv.n_lines, n_refused=sv.n_refused, genomediff=0, header_kept=sv.header_kept, host_decided=0, roc=None,
                 r_hostile=sv.n_r_hostile)
        rows.append(r)
    return rows


def extract_tp_fp_snp(vcf_file, snp_file, engine=None, strict=None):
    """Extract the TP and FP SNPs of one caller VCF (hcmv mode).
    @param vcf_file: caller VCF.  @param snp_file: truth VCF written by mummer2vcf.py.
    Outputs next to the input: <x>.filtered.vcf, fp/<x>.fp.vcf, tp/<x>.tp.vcf (:19-22,39-41)."""
    return extract_many([Job(vcf_file, snp_file, "hcmv")], engine=engine, strict=strict)[0]


def extract_tp_fp_custom_snp(vcf_file, snp_file, outdir, caller, engine=None, strict=None):
    """Custom (vareval) mode: truth is the show-snps TSV; outputs under outdir (:71-72,91)."""
    return extract_many([Job(vcf_file, snp_file, "custom", outdir, caller)], engine=engine, strict=strict)[0]
