#!/usr/bin/env python3
"""Does plain streaming depend on where an allocation landed?  qm_bw_probe (read-only / copy / write-only GB/s over fresh
allocations of the given size) called REPS times in one process.  usage: [REPS=8] [GB=4] python3 tools/bw_lottery.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import quasimodo_amd as q
eng = q.Engine(0)
gb = int(os.environ.get("GB", "4"))
for rep in range(int(os.environ.get("REPS", "8"))):
    r, c, w = eng.bw_probe(gb << 30, 5).values()
    print("%2d GB, allocation %d: read %.0f  copy %.0f  write %.0f GB/s" % (gb, rep, r, c, w), flush=True)
