#!/usr/bin/env python3
"""k_compact's block -> tile mapping against the place a batch landed in memory: ONE process, the same 1 000 VCFs x 1 M batch
created REPS times; for every allocation the compaction's time (HIP events, 6 steps each) with 8 / 4 / 2 / 1 windows
(QM_K3_WINDOWS, read at every launch) and, marked c, with whole-chunk ownership (QM_K3_OWN=chunks).  QM_ALLOC_CONTIG=64: the same
on physically contiguous batches.  usage: [REPS=6] python3 tools/compact_map_probe.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import quasimodo_amd as q
eng = q.Engine(0)
tid = eng.truth_synth(5_000_000, 100_000, 3)
for rep in range(int(os.environ.get("REPS", "6"))):
    b = eng.batch([1_000_000] * 1000, [tid] * 1000)
    b.synth(5_000_000, 100_000, 3, 3000)
    b.set_timing(True)
    out = []
    for w, own in ((8, "entries"), (4, "entries"), (2, "entries"), (1, "entries"), (8, "entries"), (8, "chunks"), (1, "chunks")):
        os.environ["QM_K3_WINDOWS"] = str(w)
        os.environ["QM_K3_OWN"] = own        # entries: a wave stores its own entries; chunks: whole 1 KiB chunks, completed from the tiles behind
        b.run(); b.finish()
        b.set_timing(True)
        for _ in range(6): b.run(); b.finish()
        t = b.timings()
        out.append("w%d%s: compact %.3f classify %.3f" % (w, "c" if own == "chunks" else "", t["compact_ms"], t["classify_ms"]))
    print("batch %d: %s" % (rep, " | ".join(out)), flush=True)
    b.close()
