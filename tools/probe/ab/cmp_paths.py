#!/usr/bin/env python3
"""The same shuffled batch through the default routing and through QM_BUCKETX=3 / 0, against the sorted run and the oracle.
usage: python3 tools/probe/ab/cmp_paths.py n_vcf records genome truth"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "..")
sys.path.insert(0, ROOT)
import numpy as np
import quasimodo_amd as q
from oracle import qm_oracle
from oracle.synth import synth_truth_keys

nv, N, L, T = (int(x) for x in sys.argv[1:5])
qm_oracle.build()
eng = q.Engine(0)
tid = eng.truth_synth(L, T, 4)
tk = synth_truth_keys(L, T, 4)
res = {}
for name, shuffled, env in (("sorted", False, None), ("default", True, None), ("X3", True, "3"), ("X0", True, "0")):
    if env is None:
        os.environ.pop("QM_BUCKETX", None)
    else:
        os.environ["QM_BUCKETX"] = env
    b = eng.batch([N] * nv, [tid] * nv)
    b.synth(L, T, 4, 4000, shuffled=shuffled)
    b.run(); b.finish()
    res[name] = (b.roc().copy(), np.array(b.scalars())[:, :5].copy(), [b.cls(v).copy() for v in range(nv)], b.path_stats())
    if name == "sorted":
        cols = b.columns(0)
        cls, oroc, sc = qm_oracle.classify_columns(*cols, *tk)
        print("oracle vs sorted: cls", np.array_equal(cls, res[name][2][0]), "roc", np.array_equal(oroc, res[name][0][0]))
    if name == "default":
        cols = b.columns(nv - 1)
        cls, oroc, sc = qm_oracle.classify_columns(*cols, *tk)
        print("oracle vs shuffled default: cls", np.array_equal(cls, res[name][2][nv - 1]), "roc", np.array_equal(oroc, res[name][0][nv - 1]),
              "scal", [int(x) for x in res[name][1][nv - 1]] == [sc[k] for k in ("n_pass", "tp_lines", "fp_lines", "TP_R", "FP_R")])
        ocls = cls
    if name in ("X3", "X0"):
        print("oracle vs shuffled", name, ": cls", np.array_equal(ocls, res[name][2][nv - 1]))
    b.close()
    print(name, {k: v for k, v in res[name][3].items() if v})
for name in ("default", "X3", "X0"):
    s, r = res["sorted"], res[name]
    print(name, "vs sorted: roc", np.array_equal(s[0], r[0]), "scalars", np.array_equal(s[1], r[1]))
    if name != "default":
        d = res["default"]
        print(name, "vs default: roc", np.array_equal(d[0], r[0]), "scalars", np.array_equal(d[1], r[1]), "cls", all(np.array_equal(a, c) for a, c in zip(d[2], r[2])))
