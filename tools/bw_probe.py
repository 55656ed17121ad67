#!/usr/bin/env python3
"""qm_bw_probe on this GPU: read-only / copy / write-only streaming rates (GB/s)."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import quasimodo_amd as q

eng = q.Engine(0)
out = {}
for gib in (1, 4):
    out["%dGiB" % gib] = eng.bw_probe(gib << 30, 6)
print(json.dumps(out))
