#!/usr/bin/env python3
"""which counters of shuffled allele-extended VCFs differ from the sorted run (synthetic shape, no adversarial input)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import quasimodo_amd as q
nv = 4
eng = q.Engine(0)
tid = eng.truth_synth(5_000_000, 100_000, 5, indel_pct=30)
res = {}
for shuffled in (False, True):
    b = eng.batch([1_000_000] * nv, [tid] * nv, alleles=True)
    b.synth(5_000_000, 100_000, 5, 5000, shuffled=shuffled, indel_pct=30)
    b.run(); b.finish()
    res[shuffled] = (b.roc().astype(np.int64), b.scalars().copy())
    if shuffled: print("paths", b.path_stats())
    b.close()
names = ("n_pass", "tp_lines", "fp_lines", "TP_R", "FP_R", "sorted", "n", "T")
for v in range(nv):
    a, s = res[False][1][v], res[True][1][v]
    print("vcf", v, {n: (int(x), int(y)) for n, x, y in zip(names, a, s) if x != y})
    for k, nm in enumerate(("TPhist", "FPhist", "U")):
        d = res[True][0][v][k] - res[False][0][v][k]
        nz = np.nonzero(d)[0]
        print("   ", nm, "bins differing:", len(nz), "first", [(int(i), int(d[i])) for i in nz[:6]])
