#!/usr/bin/env python3
"""Throughput of the host tokenizer (qm_vcf_scan) and writer on a LoFreq-like 1 M-line VCF,
by thread count (QM_HOST_THREADS)."""
import ctypes as C
import os
import resource
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from quasimodo_amd import _lib

L = _lib.lib()
n = 1_000_000
text = b"##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n" + b"".join(
    b"chr1\t%d\t.\tA\tG\t%d\tPASS\tDP=100;AF=0.012;SB=3;DP4=10,20,30,40\n" % (i * 5 + 1, i % 256) for i in range(n))
p = lambda a: a.ctypes.data_as(C.c_void_p)
cap = int(L.qm_vcf_count_lines(text, len(text))) + 1
arrs = [np.ones(cap + 1, np.int64), np.ones(cap, np.uint8), np.ones(cap, np.int32), np.ones(cap, np.int32), np.ones(cap, np.int32),
        np.ones(cap, np.float32), np.ones(cap, np.uint8)]
info = _lib.VcfCols()
print("host cores:", os.cpu_count(), "text MB: %.1f" % (len(text) / 1e6))
for nt in (1, 2, 4, 8, 16):
    os.environ["QM_HOST_THREADS"] = str(nt)
    best = 1e9
    r0 = resource.getrusage(resource.RUSAGE_SELF)
    for _ in range(5):
        t = time.time()
        L.qm_vcf_scan(text, len(text), cap, *[p(a) for a in arrs], C.byref(info))
        best = min(best, time.time() - t)
    r1 = resource.getrusage(resource.RUSAGE_SELF)
    print("threads %2d: %.4f s -> %6.0f MB/s, %.2e records/s (cpu %.3f s per call)" %
          (nt, best, len(text) / 1e6 / best, n / best, (r1.ru_utime - r0.ru_utime) / 5))
