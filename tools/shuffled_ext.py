#!/usr/bin/env python3
"""shuffled allele-extended VCFs (config 5's record shape): python3 tools/shuffled_ext.py [n_vcf [records genome truth]]
(e.g. 64 2000000 10000000 200000: BASELINE configs[4]'s VCF shape)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import quasimodo_amd as q
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N, L, T = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (1_000_000, 5_000_000, 100_000)
eng = q.Engine(0)
tid = eng.truth_synth(L, T, 5, indel_pct=30)
res = {}
for shuffled in (False, True):
    b = eng.batch([N] * nv, [tid] * nv, alleles=True)
    b.synth(L, T, 5, 5000, shuffled=shuffled, indel_pct=30)
    b.run(); b.finish()
    t0 = time.time()
    for _ in range(3):
        b.run(); b.finish()
    dt = (time.time() - t0) / 3
    res[shuffled] = (b.roc(), b.scalars()[:, :5].copy())
    print("shuffled=%s: %.3f ms per step, %.3e /s, paths %s" % (shuffled, dt * 1e3, nv * float(N) / dt, b.path_stats()), flush=True)
    b.close()
print("equal:", np.array_equal(res[True][0], res[False][0]) and np.array_equal(res[True][1], res[False][1]))
