#!/usr/bin/env python3
"""k_classify against WHERE the five record columns lie relative to each other: the same 1 000 VCFs x 1 M batch created REPS times
per setting, the columns either five allocations (skew 'none') or one slab with column k shifted by k x skew bytes behind
2 MiB-aligned pitches (QM_COL_SLAB).  QM_ALLOC_CONTIG=64 in the environment: the slab physically contiguous.
usage: [REPS=3] [SKEWS="none 0 256 4096"] python3 tools/col_skew_probe.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import quasimodo_amd as q
eng = q.Engine(0)
tid = eng.truth_synth(5_000_000, 100_000, 3)
for skew in os.environ.get("SKEWS", "none 0 256 1024 4096 65536 1114112").split():
    if skew == "none": os.environ.pop("QM_COL_SLAB", None)
    else: os.environ["QM_COL_SLAB"] = skew
    out = []
    for rep in range(int(os.environ.get("REPS", "3"))):
        b = eng.batch([1_000_000] * 1000, [tid] * 1000)
        b.synth(5_000_000, 100_000, 3, 3000)
        for _ in range(6): b.run(); b.finish()      # the compaction settles on its form
        b.set_timing(True)
        for _ in range(6): b.run(); b.finish()
        t = b.timings()
        out.append("%.3f/%.3f" % (t["classify_ms"], t["compact_ms"]))
        b.close()
    print("skew %8s: classify/compact ms %s" % (skew, "  ".join(out)), flush=True)
