#!/usr/bin/env python3
"""Which phase of k_classify carries its dependence on where a batch landed?  The same 1 000 VCFs x 1 M batch created REPS times; on
every allocation the kernel's time with phases switched off (QM_ABLATE, read at every launch; needs a -DQM_ABLATE_SUPPORT build of
the kernels: QM_LIBQMVT).  usage: [REPS=6] [ABLATES="0 32 4 15"] python3 tools/alloc_ablate_probe.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import quasimodo_amd as q
eng = q.Engine(0)
tid = eng.truth_synth(5_000_000, 100_000, 3)
abl = os.environ.get("ABLATES", "0 32 4 15 0").split()
for rep in range(int(os.environ.get("REPS", "6"))):
    b = eng.batch([1_000_000] * 1000, [tid] * 1000)
    b.synth(5_000_000, 100_000, 3, 3000)
    for _ in range(6): b.run(); b.finish()
    out = []
    for a in abl:
        os.environ["QM_ABLATE"] = a
        b.run(); b.finish()
        b.set_timing(True)
        for _ in range(6): b.run(); b.finish()
        t = b.timings()
        out.append("%s: %.3f+%.3f" % (a, t["classify_ms"], t["compact_ms"]))
    os.environ["QM_ABLATE"] = "0"
    print("batch %d: ablate -> classify+compact ms  %s" % (rep, "  ".join(out)), flush=True)
    b.close()
