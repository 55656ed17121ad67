#!/usr/bin/env python3
"""End to end on real files: N LoFreq-like VCFs of 1 M lines each on disk -> filtered / tp / fp VCFs
on disk, through the product path (extract_many = read + tokenize + upload + GPU + download + write),
with a per-phase breakdown, and the one-shot host-buffer rate of qm_classify_batch (PCIe inclusive).
usage: python3 tools/e2e_files_bench.py [n_vcf] [workdir]"""
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import quasimodo_amd as q
from quasimodo_amd import vcfio
from quasimodo_amd.extract import Job, extract_many

nv = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N, L, T = 1_000_000, 5_000_000, 100_000
work = sys.argv[2] if len(sys.argv) > 2 else tempfile.mkdtemp(prefix="qm_e2e_")
os.makedirs(os.path.join(work, "c", "fp"), exist_ok=True)
rng = np.random.default_rng(1)
bases = np.array([b"A", b"C", b"G", b"T"])
tpos = np.sort(rng.choice(L, T, replace=False) + 1)
tref = rng.integers(0, 4, T)
talt = (tref + rng.integers(1, 4, T)) & 3
hdr = b"##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n"


def render(pos, ref, alt, qual, info):
    line = np.full(len(pos), b"chrS")
    for c in (pos.astype("S"), np.full(len(pos), b"."), bases[ref], bases[alt], qual.astype("S"), np.full(len(pos), b"PASS"),
              np.full(len(pos), info)):
        line = np.char.add(np.char.add(line, b"\t"), c)
    return hdr + b"\n".join(line.tolist()) + b"\n"


truth = os.path.join(work, "T.maskrepeat.variants.vcf")
open(truth, "wb").write(render(tpos, tref, talt, np.full(T, 30), b"DP=30;TYPE=SNV"))
t0 = time.time()
one = None
paths = []
for v in range(nv):
    p = os.path.join(work, "c", "S%d-1-10.R.c.vcf" % v)
    if one is None:      # one rendering, copied: the contents do not matter for throughput
        pos = np.sort(rng.choice(L, N, replace=False) + 1)
        hit = rng.random(N) < 0.08
        j = rng.integers(0, T, N)
        pos = np.sort(np.where(hit, tpos[j], pos))
        k = np.searchsorted(tpos, pos).clip(0, T - 1)
        on = tpos[k] == pos
        ref = np.where(on, tref[k], rng.integers(0, 4, N))
        alt = np.where(on & (rng.random(N) < 0.9), talt[k], (ref + rng.integers(1, 4, N)) & 3)
        one = render(pos, ref, alt, rng.integers(0, 256, N), b"DP=100;AF=0.012;SB=3;DP4=10,20,30,40")
    open(p, "wb").write(one)
    paths.append(p)
mb = len(one) / 1e6
print("inputs: %d VCFs x %d lines (%.1f MB each) written in %.1f s; host cores %d" % (nv, N, mb, time.time() - t0, os.cpu_count()))

eng = q.Engine(0)
# ---- phases, measured one by one -------------------------------------------------------------
t = time.time(); texts = [open(p, "rb").read() for p in paths]; t_read = time.time() - t
t = time.time(); sv = [vcfio.scan_vcf(x) for x in texts]; t_scan = time.time() - t
tk = vcfio.scan_truth(open(truth, "rb").read())
tid = eng.truth_load(tk.pos, tk.ref, tk.alt)
t = time.time(); res, _ = eng.classify_batch([s.columns for s in sv], [tid] * nv); t_gpu = time.time() - t
t = time.time()
for s, r, p in zip(sv, res, paths):
    s.write(p + ".f", r["cls"], 0); s.write(p + ".t", r["cls"], 1); s.write(p + ".p", r["cls"], 2)
t_write = time.time() - t
tot = t_read + t_scan + t_gpu + t_write
print("phases for %d VCFs: read %.2f s, tokenize %.2f s (%.0f MB/s), one-shot classify incl. H2D/D2H %.2f s (%.2e records/s), "
      "write 3 files %.2f s; sum %.2f s = %.2e records/s" % (nv, t_read, t_scan, nv * mb / t_scan, t_gpu, nv * N / t_gpu, t_write, tot, nv * N / tot))
kept = sum(r["scalars"]["n_pass"] for r in res); tp = sum(r["scalars"]["tp_lines"] for r in res)
print("kept %d, TP %d, FP %d" % (kept, tp, kept - tp))
# ---- the product entry point as a whole --------------------------------------------------------
t = time.time()
extract_many([Job(p, truth, "hcmv") for p in paths], engine=eng)
dt = time.time() - t
print("extract_many end to end: %.2f s for %d VCFs = %.2e records/s (%.0f MB/s of VCF text)" % (dt, nv, nv * N / dt, nv * mb / dt))
if len(sys.argv) <= 2:
    shutil.rmtree(work, ignore_errors=True)
