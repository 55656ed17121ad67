#!/usr/bin/env python3
"""Full-size sanity check: run the bench workload and compare a few VCFs with the oracle.
usage: python3 tools/check_big.py [n_vcf] [shuffled]"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import quasimodo_amd as q
from oracle import qm_oracle as O
from oracle.synth import synth_truth_keys

nv = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
shuffled = len(sys.argv) > 2 and sys.argv[2] == "1"
L, T, N = 5_000_000, 100_000, 1_000_000
eng = q.Engine(0)
tid = eng.truth_synth(L, T, 3)
b = eng.batch([N] * nv, [tid] * nv)
b.synth(L, T, 3, 3000, shuffled=shuffled)
for rep in range(2):
    b.run()
    b.finish()
roc, scal = b.roc(), b.scalars()
bad = np.nonzero((roc[:, 0, 20].astype(np.int64) != scal[:, 1]) | (roc[:, 1, 20].astype(np.int64) != scal[:, 2]))[0]
print("VCFs with ROC(20) != line counts:", len(bad), bad[:20])
truth = synth_truth_keys(L, T, 3)
for v in sorted(set([0, 1, nv // 2, nv - 1] + bad[:3].tolist())):
    cols = b.columns(v)
    cls, oroc, sc = O.classify_columns(*cols, *truth)
    gcls = b.cls(v)
    print("vcf", v, "cls_equal", np.array_equal(gcls, cls), "roc_equal", np.array_equal(roc[v], oroc),
          "scal", scal[v][:5].tolist(), "oracle", [sc[k] for k in ("n_pass", "tp_lines", "fp_lines", "TP_R", "FP_R")])
    if not np.array_equal(roc[v], oroc):
        d = np.nonzero(roc[v] != oroc)
        print("   roc diff at", d[0][:10], d[1][:10], roc[v][d][:10], oroc[d][:10])
    if not np.array_equal(gcls, cls):
        w = np.nonzero(gcls != cls)[0]
        print("   cls diff count", len(w), "first", w[:10], gcls[w[:10]], cls[w[:10]])
