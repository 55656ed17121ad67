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
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass, field

import numpy as np

from ._lib import SCALAR_NAMES, QmvtError
from .engine import Engine
from .vcfio import AlleleDict, Patterns, scan_truth, scan_vcf


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


def extract_many(jobs, engine=None, strict=None, n_bins=256, alleles=None, gpus=None):
    """Classify and write filtered / tp / fp VCFs for a list of Job.  Returns the jobs
    with .stats filled (line counts, R-path counts, ROC rows).
    gpus > 1: the VCFs are dealt to that many GPUs of this node, one process each (quasimodo_amd.multigpu).
    alleles=True (or QM_ALLELES=1): the allele-extended mode -- every record whose REF and ALT are
    [ACGT]+ takes part, not only single bases (a build-defined widening of the reference's filter,
    include/qmvt.h; hcmv mode only)."""
    strict = _strict_default() if strict is None else strict
    alleles = _alleles_default() if alleles is None else bool(alleles)
    if gpus is not None and int(gpus) > 1:
        if engine is not None:
            raise ValueError("gpus > 1 starts one process (and one engine) per GPU: do not pass an engine")
        from .multigpu import extract_many_sharded
        return extract_many_sharded(jobs, int(gpus), n_bins=n_bins, alleles=alleles, strict=strict)[0]
    if alleles and any(j.mode != "hcmv" for j in jobs):
        raise ValueError("the allele-extended mode needs VCF truth sets (hcmv mode)")
    adict = AlleleDict() if alleles else None
    own = engine is None
    scanned, mixed = [], []
    for job in jobs:
        _paths(job)

    def _scan(job):
        with open(job.vcf_file, "rb") as fh:
            return scan_vcf(fh.read(), alleles=adict)

    # files are read, tokenized and (below) written by a small pool: the library drops the GIL, and
    # one VCF's scan is itself multi-threaded only when the file is large
    pool = ThreadPoolExecutor(max(1, min(len(jobs), _io_threads())))
    try:
        all_scanned = list(pool.map(_scan, jobs))
    except BaseException:
        pool.shutdown()
        raise
    for j, job in enumerate(jobs):
        sv = all_scanned[j]
        if sv.n_refused and strict:
            pool.shutdown()
            raise QmvtError(-8, "%s line %d: a kept line holds NUL or non-ASCII bytes -- the reference's answer for it depends on "
                                "the locale Python exports to grep; set QM_LENIENT=1 to classify it by its columns"
                            % (job.vcf_file, sv.first_refused_line))
        scanned.append(sv)
        if not is_pure_strain(job.vcf_file):
            mixed.append(j)
    results, r_exchange = {}, {}
    if mixed:
        if engine is None:
            engine = Engine(int(os.environ.get("QM_DEVICE", "0")))
        truth_ids, patterns = {}, {}
        try:
            truth_info = {}
            for j in mixed:
                key = (os.path.abspath(jobs[j].snp_file), jobs[j].mode)
                if key not in truth_ids:
                    with open(jobs[j].snp_file, "rb") as fh:
                        ttext = fh.read()
                    tk = scan_truth(ttext, custom=jobs[j].mode == "custom", alleles=adict)
                    if tk.n_refused and strict:
                        raise QmvtError(-8, "%s: %d truth rows hold NUL or non-ASCII bytes" % (jobs[j].snp_file, tk.n_refused))
                    truth_ids[key] = engine.truth_load(tk.pos, tk.ref, tk.alt)
                    truth_info[key] = tk
                    patterns[key] = Patterns(ttext, custom=jobs[j].mode == "custom", alleles=alleles)
            # lines whose fgrep answer the columns cannot give are decided on the host, from the text of
            # the patterns, BEFORE the upload: the decision travels in the flags column and the device counts and
            # lists these lines like all others
            def _host(j):
                key = (os.path.abspath(jobs[j].snp_file), jobs[j].mode)
                pt = patterns[key]
                if scanned[j].n_host or scanned[j].n_nokey_kept or pt.needs_full_hostpath:
                    return scanned[j].hostpath(pt)
                return None
            for j, ex in zip(mixed, pool.map(_host, mixed)):
                if ex is not None:
                    r_exchange[j] = ex
            cols = [scanned[j].columns for j in mixed]
            tids = [truth_ids[(os.path.abspath(jobs[j].snp_file), jobs[j].mode)] for j in mixed]
            res, _ = engine.classify_batch(cols, tids, n_bins=n_bins, alleles=alleles)
            for j, r in zip(mixed, res):
                r["genomediff"] = truth_info[(os.path.abspath(jobs[j].snp_file), jobs[j].mode)].genomediff
                ex = r_exchange.get(j)
                if ex is not None:
                    # R keys a line by the TEXT of POS / REF / ALT; for lines without a comparable key the device
                    # counted distinct (carried pos, ref, alt) instead: swap those for the text keys
                    r["scalars"]["FP_R"] += ex["fp_r"] - ex["device_nokey_keys"]
                    r["scalars"]["TP_R"] += ex["tp_r"]
                results[j] = r
        except BaseException:
            pool.shutdown()
            raise
        finally:
            for pt in patterns.values():
                pt.close()
            if own:
                engine.close()
            else:
                for tid in truth_ids.values():   # a shared engine does not keep this call's truth sets
                    engine.truth_release(tid)
    writes = []
    for j, job in enumerate(jobs):
        sv = scanned[j]
        os.makedirs(os.path.dirname(job.fp_out) or ".", exist_ok=True)
        if j in results:
            r = results[j]
            cls = r["cls"]
            os.makedirs(os.path.dirname(job.tp_out) or ".", exist_ok=True)
            writes += [pool.submit(sv.write, job.filtered_out, cls, 0), pool.submit(sv.write, job.tp_out, cls, 1),
                       pool.submit(sv.write, job.fp_out, cls, 2)]
            job.stats = dict(r["scalars"])
            job.stats.update(pure_strain=False, genomediff=r["genomediff"], roc=r["roc"], header_kept=sv.header_kept)
        else:  # pure strain: fp is a copy of filtered, truth never read (:33-36)
            cls = (sv.flags & 1).astype(np.uint8)
            writes += [pool.submit(sv.write, job.filtered_out, cls, 0), pool.submit(sv.write, job.fp_out, cls, 0)]
            job.tp_out = ""
            npass = int(cls.sum())
            job.stats = dict(zip(SCALAR_NAMES, (npass, 0, npass, 0, 0, 1, sv.n_records, 0)))
            job.stats.update(pure_strain=True, genomediff=0, roc=None, header_kept=sv.header_kept)
    try:
        for w in writes:
            w.result()          # the first writer error surfaces here
    finally:
        pool.shutdown()
    return jobs


def extract_tp_fp_snp(vcf_file, snp_file, engine=None, strict=None):
    """Extract the TP and FP SNPs of one caller VCF (hcmv mode).
    @param vcf_file: caller VCF.  @param snp_file: truth VCF written by mummer2vcf.py.
    Outputs next to the input: <x>.filtered.vcf, fp/<x>.fp.vcf, tp/<x>.tp.vcf (:19-22,39-41)."""
    return extract_many([Job(vcf_file, snp_file, "hcmv")], engine=engine, strict=strict)[0]


def extract_tp_fp_custom_snp(vcf_file, snp_file, outdir, caller, engine=None, strict=None):
    """Custom (vareval) mode: truth is the show-snps TSV; outputs under outdir (:71-72,91)."""
    return extract_many([Job(vcf_file, snp_file, "custom", outdir, caller)], engine=engine, strict=strict)[0]
