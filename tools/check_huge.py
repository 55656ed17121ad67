#!/usr/bin/env python3
"""Batches beyond 2^32 records (BASELINE configs[3]: one GPU's shard is 1 250 VCFs x 10 M = 1.25e10 records):
global record indices need 64 bits everywhere.  Generates n_vcf x 10 M records on the device, runs the path and
checks the LAST VCFs (the ones past the 2^32nd record) against the oracle, and invariants over all of them.
usage: python3 tools/check_huge.py [n_vcf=600] [config5]
config5: BASELINE configs[4]'s shape instead -- 2 M-record VCFs, 30 % indels, three truth sets (VCF v uses v mod 3),
allele-extended mode; one GPU's shard of the 8-GPU run is 6 250 VCFs."""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import quasimodo_amd as q
from oracle import qm_oracle as O
from oracle.synth import synth_truth_keys

nv = int(sys.argv[1]) if len(sys.argv) > 1 else 600
c5 = len(sys.argv) > 2 and sys.argv[2] == "config5"
L, T, N = (10_000_000, 200_000, 2_000_000) if c5 else (50_000_000, 1_000_000, 10_000_000)
pct = 30 if c5 else 0
seeds = (5, 6, 7) if c5 else (4,)
eng = q.Engine(0)
tids = [eng.truth_synth(L, T, ts, indel_pct=pct) for ts in seeds]
tid = tids[0]
t0 = time.time()
b = eng.batch([N] * nv, [tids[v % len(tids)] for v in range(nv)], alleles=c5)
print("batch of %.2e records allocated: %.1f GB in HBM, %.1f s" % (nv * N, b.device_bytes / 1e9, time.time() - t0), flush=True)
b.synth(L, T, None, 5000 if c5 else 4000, indel_pct=pct)
b.set_timing(True)
for _ in range(3):
    b.run()
b.finish()
tm = b.timings()
print("classify %.2f ms (%.0f GB/s algorithmic), finalize %.2f, compact %.2f" %
      (tm["classify_ms"], nv * (17.0 * N + 12.0 * T) / tm["classify_ms"] / 1e6, tm["finalize_ms"], tm["compact_ms"]), flush=True)
roc, scal = b.roc(), b.scalars()
assert (scal[:, 6] == N).all() and (scal[:, 5] == 1).all()
assert np.array_equal(roc[:, 0, 20].astype(np.int64), scal[:, 1]) and np.array_equal(roc[:, 1, 20].astype(np.int64), scal[:, 2])
for w, t_ in enumerate(tids):
    assert np.array_equal(b.global_counts()[t_], roc[w::len(tids)].sum(axis=0))
truths = [synth_truth_keys(L, T, ts, pct) for ts in seeds]
for v in (0, nv // 2, nv - 1):
    cols = b.columns(v)
    cls, oroc, sc = O.classify_columns(*cols, *truths[v % len(tids)], ext=c5)
    ok = (np.array_equal(b.cls(v), cls) and np.array_equal(roc[v], oroc)
          and [int(x) for x in scal[v][:5]] == [sc[k] for k in ("n_pass", "tp_lines", "fp_lines", "TP_R", "FP_R")])
    idx = b.idx(v)
    ok = ok and np.array_equal(idx[:sc["tp_lines"]], np.nonzero(cls == 3)[0]) and np.array_equal(idx[N - sc["fp_lines"]:], np.nonzero(cls == 1)[0])
    print("vcf %d (first record %.3e): %s" % (v, v * float(N), "equals the oracle" if ok else "MISMATCH"), flush=True)
    assert ok
print("ok")
