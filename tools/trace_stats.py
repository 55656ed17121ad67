#!/usr/bin/env python3
"""Per-kernel statistics of a rocprofv3 --kernel-trace run of bench.py from the TIMED launches only (the
tracer's own *_kernel_stats.csv averages the warm-up launches in).  A kernel launched c times per step appears
c x (warmup + steps) times in the trace; its first c x warmup launches are dropped.  Kernels whose launch count is not a
multiple of warmup + steps (probes, checks behind the timed region) are listed as they are, marked 'all' -- and so is one whose
count happens to divide but whose launches lie outside the timed window (the window: from the first timed launch of the
kernel with the largest total to the end of the last launch of any per-step kernel; the column fills of the batch's creation
and the bandwidth probe's made seven launches in a 5+2 run).
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
for v in rows.values():
    v.sort()


def kept(v):
    n = len(v)
    if n % (steps + warmup) == 0 and warmup:
        return v[(n // (steps + warmup)) * warmup:]
    return None


lead = max((v for v in rows.values() if kept(v)), key=lambda v: sum(x[1] for x in kept(v)), default=None)
t0 = kept(lead)[0][0] if lead else 0
t1 = lead[-1][0] + 4 * max(x[1] for x in lead) if lead else 0   # the step's other kernels end within a few of its durations
for name, v in rows.items():
    what = "all"
    k = kept(v)
    if k and all(t0 <= x[0] <= t1 for x in k):
        v = k
        what = "timed"
    du = [x[1] for x in v]
    out.append((sum(du), name, len(du), what, statistics.mean(du), min(du), max(du), statistics.pstdev(du)))
tot = sum(o[0] for o in out) or 1
w = csv.writer(sys.stdout)
w.writerow(["Name", "Calls", "Launches", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
for t, name, n, what, mean, mn, mx, sd in sorted(out, reverse=True):
    w.writerow([name, n, what, t, "%.1f" % mean, "%.2f" % (100.0 * t / tot), mn, mx, "%.1f" % sd])
