#!/usr/bin/env python3
"""One shuffled configs[2]-shaped batch through the bucket path; prints the step time and a digest of the results.
usage: [QM_JOIN=hash] python3 tools/join_ab.py [n_vcf] [records] [genome] [truth]"""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import quasimodo_amd as q

nv = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
L = int(sys.argv[3]) if len(sys.argv) > 3 else 5_000_000
T = int(sys.argv[4]) if len(sys.argv) > 4 else 100_000
eng = q.Engine(0)
tid = eng.truth_synth(L, T, 3)
b = eng.batch([N] * nv, [tid] * nv)
b.synth(L, T, 3, 3000, shuffled=True)
b.run(); b.finish()
t0 = time.time()
steps = 5
for _ in range(steps):
    b.run(); b.finish()
dt = (time.time() - t0) / steps
h = hashlib.sha1()
for a in (b.roc(), b.scalars(), b.idx(0), b.cls(nv - 1)):
    h.update(np.ascontiguousarray(a).tobytes())
print("paths", b.path_stats()); print("join=%s %d x %d: %.3f ms per step, %.3e classifications/s, digest %s" % (os.environ.get("QM_JOIN", "lean"), nv, N, dt * 1e3, nv * N / dt, h.hexdigest()[:12]), flush=True)
