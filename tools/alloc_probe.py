#!/usr/bin/env python3
"""How much does the SAME step vary with where its batch landed in memory?  One process, the same 1 000 VCFs x 1 M batch created,
timed (40 steps) and destroyed several times: within one allocation the step is steady to a per cent (tools/thermal_probe.py:
8 s of it at 3.82 ms), between allocations k_classify moved between 2.71 and 3.27 ms on the boxes of this pool.
usage: [REPS=6] python3 tools/alloc_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import quasimodo_amd as q
eng = q.Engine(0)
tid = eng.truth_synth(5_000_000, 100_000, 3)
hold = []
for rep in range(int(os.environ.get("REPS", "6"))):
    if os.environ.get("PERTURB"):   # something of another size in front of every batch: the allocator hands out other places
        import torch
        hold.append(torch.empty((rep + 1) * int(os.environ["PERTURB"]) * (1 << 20), dtype=torch.uint8, device="cuda"))
    b = eng.batch([1_000_000] * 1000, [tid] * 1000)
    b.synth(5_000_000, 100_000, 3, 3000)
    b.set_timing(True)
    for _ in range(5): b.run(); b.finish()
    t0 = time.perf_counter()
    steps = int(os.environ.get("STEPS", "40"))
    for _ in range(steps): b.run(); b.finish()
    dt = (time.perf_counter() - t0) / steps * 1e3
    t = b.timings()
    print("batch %d: %.3f ms per step, classify %.3f compact %.3f" % (rep, dt, t["classify_ms"], t["compact_ms"]), flush=True)
    b.close()
