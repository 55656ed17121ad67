"""Per-rank classification for the world-2 gloo test of quasimodo_amd.multigpu: the product's own host side (scan,
host path, writers -- all CPU code of libqmvt.so) with the ORACLE's column-level restatement standing in for the
device (there is no GPU in that test).  Imported by the rank processes through the `classify` hook of
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
