#!/usr/bin/env python3
"""Does the step slow down under sustained load?  1 000 VCFs x 1 M, the step timed in blocks of 25 for a few seconds, with the
card's clocks and power from rocm-smi beside it (when the tool answers).  usage: python3 tools/thermal_probe.py [seconds]"""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import quasimodo_amd as q

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
eng = q.Engine(0)
tid = eng.truth_synth(5_000_000, 100_000, 3)
b = eng.batch([1_000_000] * 1000, [tid] * 1000)
b.synth(5_000_000, 100_000, 3, 3000)


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "-d", "0"], capture_output=True, text=True, timeout=5).stdout
        keep = [ln.strip() for ln in out.splitlines() if any(k in ln for k in ("sclk", "mclk", "Power", "junction", "Temperature (Sensor junction)"))]
        return " | ".join(x.split(":", 1)[-1].strip()[:60] for x in keep[:4])
    except Exception as e:
        return "rocm-smi: %s" % e


t_end = time.time() + secs
k = 0
print("idle:", smi())
while time.time() < t_end:
    t0 = time.perf_counter()
    for _ in range(25):
        b.run(); b.finish()
    dt = (time.perf_counter() - t0) / 25 * 1e3
    k += 25
    print("steps %4d..%4d: %.3f ms per step %s" % (k - 25, k, dt, ("   " + smi()) if k % 200 == 0 else ""))
