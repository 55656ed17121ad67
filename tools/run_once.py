#!/usr/bin/env python3
"""Minimal driver for profiling: build one synthetic batch, run the path a few times.
usage: python3 tools/run_once.py [n_vcf] [runs] [shuffled | number of runs] [indel_pct]
indel_pct > 0: allele-extended batch of config 5's shape (mixed SNP + indel)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import quasimodo_amd as q

nv = int(sys.argv[1]) if len(sys.argv) > 1 else 256
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 5
shuffled = int(sys.argv[3]) if len(sys.argv) > 3 else 0     # 1: records permuted; R >= 2: R ascending runs (a VCF sorted per contig)
pct = int(sys.argv[4]) if len(sys.argv) > 4 else 0
ext = pct != 0          # negative: allele-extended kernel on single-base data
pct = max(pct, 0)
eng = q.Engine(0)
tid = eng.truth_synth(5_000_000, 100_000, 3, indel_pct=pct)
b = eng.batch([1_000_000] * nv, [tid] * nv, alleles=ext)
b.synth(5_000_000, 100_000, 3, 3000, shuffled=shuffled, indel_pct=pct)
b.set_timing(True)
for _ in range(runs):
    b.run()
    b.finish()
t = b.timings()
byts = nv * (17e6 + 1.2e6)
# the step as the bench times it: wall clock over back-to-back steps, no timing events in the stream
import time
b.set_timing(False)
b.run(); b.finish()
t0 = time.perf_counter()
for _ in range(2 * runs):
    b.run()
    b.finish()
wall = (time.perf_counter() - t0) / (2 * runs) * 1e3
print("n_vcf=%d classify %.3f ms (%.0f GB/s algorithmic) finalize %.3f compact %.3f total %.3f wall %.3f" %
      (nv, t["classify_ms"], byts / t["classify_ms"] / 1e6, t["finalize_ms"], t["compact_ms"], t["total_ms"], wall))
sc = b.scalars()
print("  paths:", b.path_stats())
print("  sums: kept %d tp_lines %d fp_lines %d TP_R %d FP_R %d" % tuple(int(sum(int(r[k]) for r in sc)) for k in range(5)))
