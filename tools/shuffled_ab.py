#!/usr/bin/env python3
"""bucket path vs radix-sort path on shuffled configs[2] VCFs: python3 tools/shuffled_ab.py [n_vcf] [records]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import quasimodo_amd as q

nv = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
eng = q.Engine(0)
tid = eng.truth_synth(5_000_000, 100_000, 3)
rocs = {}
for path in ("buckets", "radix", "buckets", "radix"):
    os.environ["QM_SORT_PATH"] = path
    b = eng.batch([N] * nv, [tid] * nv)
    b.synth(5_000_000, 100_000, 3, 3000, shuffled=True)
    b.run(); b.finish()
    t0 = time.time()
    for _ in range(5):
        b.run(); b.finish()
    dt = (time.time() - t0) / 5
    rocs[path] = (b.roc(), b.scalars(), b.idx(0), b.cls(nv - 1))
    print(path, "%.3f ms per step, %.3e classifications/s" % (dt * 1e3, nv * N / dt), flush=True)
    b.close()
a, r = rocs["buckets"], rocs["radix"]
print("equal:", all(np.array_equal(x, y) for x, y in zip(a, r)))
