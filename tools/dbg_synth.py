import os, sys
import numpy as np
sys.path.insert(0, ".")
import quasimodo_amd as q
from oracle.synth import synth_truth_keys
L, T, N = 5_000_000, 100_000, 1_000_000
eng = q.Engine(0)
tid = eng.truth_synth(L, T, 3)
for nv in (200, 1000):
    b = eng.batch([N] * nv, [tid] * nv)
    b.synth(L, T, 3, 3000)
    for v in (0, 100, 124, 125, 126, nv-1):
        pos, ref, alt, qual, flags = b.columns(v)
        print(nv, v, "pass frac %.4f" % (flags & 1).mean(), "qual>=20 frac %.4f" % (qual >= 20).mean(), "mismatch", int(((qual >= 20) != ((flags & 1) == 1)).sum()),
              "pos[:4]", pos[:4], "sorted", bool(np.all(np.diff(pos) > 0)), "flags uniq", np.unique(flags)[:6], "qual max", qual.max())
    b.close()
