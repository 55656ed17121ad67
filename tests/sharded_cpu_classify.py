"""Per-rank classification for the world-2 gloo test of quasimodo_amd.multigpu: the product's own host side (scan,
host path, writers -- all CPU code of libqmvt.so) with the ORACLE's column-level restatement standing in for the
device (there is no GPU in that test).  Imported by the rank processes through the `body` hook of
extract_many_sharded; never by the product."""
import os

import numpy as np


def classify(jobs, device, n_bins=256, alleles=None, strict=None):
    from oracle import qm_oracle as O
    from quasimodo_amd import scan_truth, scan_vcf
    from quasimodo_amd.extract import is_pure_strain
    from quasimodo_amd.vcfio import Patterns
    out = []
    for j in jobs:
        with open(j.vcf_file, "rb") as fh:
            sv = scan_vcf(fh.read())
        os.makedirs(os.path.dirname(j.fp_out), exist_ok=True)
        if is_pure_strain(j.vcf_file):
            cls = (sv.flags & 1).astype(np.uint8)
            sv.write(j.filtered_out, cls, 0)
            sv.write(j.fp_out, cls, 0)
            st = dict(n_pass=int(cls.sum()), tp_lines=0, fp_lines=int(cls.sum()), TP_R=0, FP_R=0, pure_strain=True, roc=None)
        else:
            with open(j.snp_file, "rb") as fh:
                truth = fh.read()
            custom = j.mode == "custom"
            tk = scan_truth(truth, custom=custom)
            pt = Patterns(truth, custom=custom)
            ex = sv.hostpath(pt) if (sv.n_host or sv.n_nokey_kept or pt.needs_full_hostpath) else None
            pt.close()
            cls, roc, sc = O.classify_columns(*sv.columns, tk.pos, tk.ref, tk.alt, n_bins=n_bins)
            if ex is not None:
                sc["FP_R"] += ex["fp_r"] - ex["device_nokey_keys"]
                sc["TP_R"] += ex["tp_r"]
            os.makedirs(os.path.dirname(j.tp_out), exist_ok=True)
            for sel, path in ((0, j.filtered_out), (1, j.tp_out), (2, j.fp_out)):
                sv.write(path, cls, sel)
            st = dict(sc, pure_strain=False, roc=roc, genomediff=tk.genomediff)
        st["rank"] = int(os.environ.get("RANK", "0"))
        st["device"] = device
        out.append(st)
    return out



def body(jobs, indices, device, opts):
    """the per-rank body of quasimodo_amd.multigpu (default_body's interface) without a GPU: no device counters (run_rank sums
    the rows on the host then), the FP overlap of the rank's samples by the oracle's restatement of snpcaller_fp_compare.R"""
    stats = classify(jobs, device, n_bins=opts["n_bins"], alleles=opts.get("alleles"), strict=opts.get("strict"))
    for j, st in zip(jobs, stats):
        j.stats = st
    extra = None
    if opts.get("post"):
        from oracle import qm_oracle as O
        args = opts["post_args"]
        meta = [args["meta"][i] for i in indices]
        cmp_callers = args["cmp_callers"]
        by_sample = {}
        for (c, s), j in zip(meta, jobs):
            if c in cmp_callers and not s.endswith(("-1-0", "-0-1")):
                by_sample.setdefault(s, {})[c] = j.fp_out
        ov = {}
        for s, files in by_sample.items():
            assert len(files) == len(cmp_callers), "the compared callers of %s are not on one rank" % s
            ov[s] = [int(x) for x in O.fp_overlap_text([open(files[c], "rb").read() for c in cmp_callers])]
        extra = {"overlap": ov, "rank": opts["rank"], "samples": sorted({s for _, s in meta})}
    return {"stats": stats, "counters": None, "extra": extra}
