#!/usr/bin/env python3
"""Batches beyond 2^32 records (BASELINE configs[3]: one GPU's shard is 1 250 VCFs x 10 M = 1.25e10 records):
global record indices need 64 bits everywhere.  Generates n_vcf x 10 M records on the device, runs the path and
checks the LAST VCFs (the ones past the 2^32nd record) against the oracle, and invariants over all of them.
usage: python3 tools/check_huge.py [n_vcf=600]"""
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
L, T, N = 50_000_000, 1_000_000, 10_000_000
eng = q.Engine(0)
tid = eng.truth_synth(L, T, 4)
t0 = time.time()
b = eng.batch([N] * nv, [tid] * nv)
print("batch of %.2e records allocated: %.1f GB in HBM, %.1f s" % (nv * N, b.device_bytes / 1e9, time.time() - t0), flush=True)
b.synth(L, T, 4, 4000)
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
assert np.array_equal(b.global_counts()[tid], roc.sum(axis=0))
truth = synth_truth_keys(L, T, 4)
for v in (0, nv // 2, nv - 1):
    cols = b.columns(v)
    cls, oroc, sc = O.classify_columns(*cols, *truth)
    ok = (np.array_equal(b.cls(v), cls) and np.array_equal(roc[v], oroc)
          and [int(x) for x in scal[v][:5]] == [sc[k] for k in ("n_pass", "tp_lines", "fp_lines", "TP_R", "FP_R")])
    idx = b.idx(v)
    ok = ok and np.array_equal(idx[:sc["tp_lines"]], np.nonzero(cls == 3)[0]) and np.array_equal(idx[N - sc["fp_lines"]:], np.nonzero(cls == 1)[0])
    print("vcf %d (first record %.3e): %s" % (v, v * float(N), "equals the oracle" if ok else "MISMATCH"), flush=True)
    assert ok
print("ok")
