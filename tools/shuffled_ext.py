#!/usr/bin/env python3
"""shuffled allele-extended VCFs (config 5's record shape): python3 tools/shuffled_ext.py [n_vcf]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import quasimodo_amd as q
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 128
eng = q.Engine(0)
tid = eng.truth_synth(5_000_000, 100_000, 5, indel_pct=30)
res = {}
for shuffled in (False, True):
    b = eng.batch([1_000_000] * nv, [tid] * nv, alleles=True)
    b.synth(5_000_000, 100_000, 5, 5000, shuffled=shuffled, indel_pct=30)
    b.run(); b.finish()
    t0 = time.time()
    for _ in range(3):
        b.run(); b.finish()
    dt = (time.time() - t0) / 3
    res[shuffled] = (b.roc(), b.scalars()[:, :5].copy())
    print("shuffled=%s: %.3f ms per step, %.3e /s, paths %s" % (shuffled, dt * 1e3, nv * 1e6 / dt, b.path_stats()), flush=True)
    b.close()
print("equal:", np.array_equal(res[True][0], res[False][0]) and np.array_equal(res[True][1], res[False][1]))
