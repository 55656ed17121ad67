#!/usr/bin/env python3
"""Latency of the real-size use case: the 60 HCMV-shaped golden VCFs (2-4 k records each) through
extract_many, phase by phase.  usage: python3 tools/hcmv_latency.py"""
import glob
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
t0 = time.time()
import quasimodo_amd as q
from quasimodo_amd import vcfio
from quasimodo_amd.extract import Job, extract_many
t_import = time.time() - t0
g = os.path.join(ROOT, "tests", "golden", "hcmv", "input")
w = tempfile.mkdtemp(prefix="qm_hcmv_")
jobs = []
for p in sorted(glob.glob(os.path.join(g, "*", "*.vcf"))):
    c = os.path.basename(os.path.dirname(p))
    if c == "nucmer":
        continue
    d = os.path.join(w, c)
    os.makedirs(os.path.join(d, "fp"), exist_ok=True)
    dst = os.path.join(d, os.path.basename(p))
    shutil.copyfile(p, dst)
    jobs.append(Job(dst, os.path.join(g, "nucmer", os.path.basename(p)[:2] + ".maskrepeat.variants.vcf"), "hcmv"))
t = time.time(); eng = q.Engine(0); t_init = time.time() - t
t = time.time(); extract_many(jobs, engine=eng); t_first = time.time() - t
ts = []
for _ in range(5):
    t = time.time(); extract_many(jobs, engine=eng); ts.append(time.time() - t)
n = sum(j.stats["n_records"] for j in jobs)
print("import %.3f s, qm_init %.3f s, first extract_many (%d VCFs, %d records) %.3f s, warm %.3f s (min of 5)" % (t_import, t_init, len(jobs), n, t_first, min(ts)))
# where the warm time goes
svs = [vcfio.scan_vcf(open(j.vcf_file, "rb").read()) for j in jobs if not j.stats["pure_strain"]]
tk = vcfio.scan_truth(open(jobs[0].snp_file, "rb").read())
tid = eng.truth_load(tk.pos, tk.ref, tk.alt)
t = time.time()
for _ in range(5):
    res, _ = eng.classify_batch([s.columns for s in svs], [tid] * len(svs))
print("classify_batch alone on the %d mixed VCFs: %.4f s per call" % (len(svs), (time.time() - t) / 5))
shutil.rmtree(w, ignore_errors=True)
