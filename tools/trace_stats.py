#!/usr/bin/env python3
"""Per-kernel statistics of a rocprofv3 --kernel-trace run of bench.py from the TIMED launches only (the
tracer's own *_kernel_stats.csv averages the warm-up launches in).  A kernel launched c times per step appears
c x (warmup + steps) times in the trace; its first c x warmup launches are dropped.  Kernels whose launch count is not a
multiple of warmup + steps (probes, checks behind the timed region) are listed as they are, marked 'all'.
usage: python3 tools/trace_stats.py <dir with *_kernel_trace.csv> <steps> <warmup> > profiles/rNN_kernel_stats.csv"""
import csv
import glob
import statistics
import sys

d, steps, warmup = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rows = {}
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.setdefault(r["Kernel_Name"], []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
out = []
for name, v in rows.items():
    v.sort()
    n = len(v)
    what = "all"
    if n % (steps + warmup) == 0 and warmup:
        c = n // (steps + warmup)
        v = v[c * warmup:]
        what = "timed"
    du = [x[1] for x in v]
    out.append((sum(du), name, len(du), what, statistics.mean(du), min(du), max(du), statistics.pstdev(du)))
tot = sum(o[0] for o in out) or 1
w = csv.writer(sys.stdout)
w.writerow(["Name", "Calls", "Launches", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
for t, name, n, what, mean, mn, mx, sd in sorted(out, reverse=True):
    w.writerow([name, n, what, t, "%.1f" % mean, "%.2f" % (100.0 * t / tot), mn, mx, "%.1f" % sd])
