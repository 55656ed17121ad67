#!/usr/bin/env python3
"""k_classify reading a round-interleaved copy of the columns (probe builds: -DK1_IL_PROBE for the kernels and the api; QM_IL_PROBE read at
every launch) against the five arrays, on the same batch and allocation; REPS fresh batches.  The counts must not move.
usage: QM_LIBQMVT=<probe build> [REPS=6] python3 tools/il_probe.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import quasimodo_amd as q
eng = q.Engine(0)
tid = eng.truth_synth(5_000_000, 100_000, 3)
for rep in range(int(os.environ.get("REPS", "6"))):
    b = eng.batch([1_000_000] * 1000, [tid] * 1000)
    b.synth(5_000_000, 100_000, 3, 3000)
    for _ in range(6): b.run(); b.finish()
    out, sums = [], []
    for il in ("0", "1", "0", "1"):
        os.environ["QM_IL_PROBE"] = il
        b.run(); b.finish()
        b.set_timing(True)
        for _ in range(6): b.run(); b.finish()
        t = b.timings()
        out.append("%s: %.3f+%.3f" % ("interleaved" if il == "1" else "five arrays", t["classify_ms"], t["compact_ms"]))
        sc = b.scalars()
        sums.append(tuple(int(sum(int(r[k]) for r in sc)) for k in range(5)))
    os.environ["QM_IL_PROBE"] = "0"
    print("batch %d: classify+compact ms  %s   counts equal: %s" % (rep, "  ".join(out), len(set(sums)) == 1), flush=True)
    b.close()
