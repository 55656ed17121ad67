#!/usr/bin/env python3
"""Randomised parity hunt: many small batches of adversarial shape through the engine (C ABI) against the
oracle.  Shapes: runs of equal positions straddling rounds / tiles / spans, dense truth against sparse
VCFs (oversize slices), many truth entries per position, unsorted and per-contig-sorted VCFs mixed with sorted ones, n_bins < 256,
saturated QUALs, records without keys, non-'.' IDs -- in both the default and the allele-extended mode.
usage: python3 tools/gpu_fuzz.py [rounds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
os.environ.setdefault("QM_BUCKET_MIN", "0")   # the fuzzed VCFs are small: send the unsorted ones through the bucket path all the same
sys.path.insert(0, ROOT)
import quasimodo_amd as q
from oracle import qm_oracle as O

def run(rounds, seed, eng=None):
    """returns the number of mismatches"""
    rng = np.random.default_rng(seed)
    eng = eng or q.Engine(0)



    def alleles(n, ext, p_ext):
        a = rng.integers(0, 4, n).astype(np.int64)
        if ext:
            k = rng.random(n)
            small = (2 << 26) | rng.integers(0, 16, n)
            big = (rng.integers(2, 14, n) << 26) | rng.integers(0, 1 << 4, n)
            dic = 0x40000000 | rng.integers(0, 8, n)
            a = np.where(k < p_ext * 0.5, small, a)
            a = np.where((k >= p_ext * 0.5) & (k < p_ext * 0.8), big, a)
            a = np.where((k >= p_ext * 0.8) & (k < p_ext), dic, a)
        a = np.where(rng.random(n) < 0.03, rng.choice(np.array([-1, 4, 9, 0x07ffffff])), a)
        return a.astype(np.int32)


    def make_vcf(n, L, truth, ext, nb, sorted_, style):
        tpos, tref, talt = truth
        if style == 0:      # uniform
            pos = rng.integers(1, L + 1, n)
        elif style == 1:    # a few positions only: very long runs
            pos = rng.choice(rng.integers(1, L + 1, max(1, n // 3000 + 1)), n)
        elif style == 2:    # clustered
            c = rng.integers(1, L + 1, max(1, n // 50 + 1))
            pos = np.clip(rng.choice(c, n) + rng.integers(-3, 4, n), 1, L)
        else:               # on truth positions mostly
            pos = tpos[rng.integers(0, len(tpos), n)] if len(tpos) else rng.integers(1, L + 1, n)
        pos = pos.astype(np.int32)
        ref, alt = alleles(n, ext, 0.3), alleles(n, ext, 0.3)
        if len(tpos) and n:
            take = rng.random(n) < rng.choice([0.05, 0.3, 0.8])
            j = rng.integers(0, len(tpos), n)
            pos = np.where(take, tpos[j], pos); ref = np.where(take, tref[j], ref); alt = np.where(take, talt[j], alt)
        if n:
            d = rng.random(n) < 0.1
            src = rng.integers(0, n, n)
            pos, ref, alt = np.where(d, pos[src], pos), np.where(d, ref[src], ref), np.where(d, alt[src], alt)
        qual = rng.integers(0, 400, n).astype(np.float32)
        qual = np.where(rng.random(n) < 0.1, np.float32(np.inf), qual)
        qual = np.where(rng.random(n) < 0.05, np.float32(-np.inf), qual)
        qual = np.where(rng.random(n) < 0.05, np.float32(np.nan), qual)
        qual = np.where(rng.random(n) < 0.1, qual + np.float32(0.75), qual).astype(np.float32)
        ok = (lambda c: ((c >= 0) & (c < 4)) | (c >= 0x08000000)) if ext else (lambda c: (c >= 0) & (c < 4))
        passed = ok(ref) & ok(alt) & (np.floor(np.nan_to_num(qual, nan=-1.0)) >= 20)
        flags = passed.astype(np.uint8) | ((rng.random(n) > 0.07).astype(np.uint8) << 1) | ((rng.random(n) < 0.03).astype(np.uint8) << 2)
        if sorted_ and n:
            o = np.argsort(pos, kind="stable")
            if sorted_ > 1 and n > 1:      # sorted PER CONTIG: `sorted_` ascending runs of random lengths one behind the other (the runs path)
                cuts = np.sort(rng.choice(np.arange(1, n), size=min(sorted_ - 1, n - 1), replace=False))
                o = np.concatenate([np.sort(part) for part in np.split(rng.permutation(n), cuts)])
                o = np.concatenate([part[np.argsort(pos[part], kind="stable")] for part in np.split(o, cuts)])
            pos, ref, alt, qual, flags = pos[o], ref[o], alt[o], qual[o], flags[o]
        c = lambda a, dt: np.ascontiguousarray(a, dt)
        return c(pos, np.int32), c(ref, np.int32), c(alt, np.int32), c(qual, np.float32), c(flags, np.uint8)


    bad = 0
    t0 = time.time()
    for it in range(rounds):
        ext = bool(rng.integers(0, 2))
        nb = int(rng.choice([256, 256, 256, 1, 2, 21, 100, 255]))
        L = int(rng.choice([50, 2000, 100000, 5_000_000, (1 << 28) - 1]))
        T = int(rng.choice([0, 1, 40, 3000, 40000]))
        tpos = rng.integers(1, L + 1, T).astype(np.int32)
        if T > 4:
            tpos[T // 2:] = tpos[:T - T // 2]
        truth = (tpos, alleles(T, ext, 0.4), alleles(T, ext, 0.4))
        tid = eng.truth_load(*truth)
        nv = int(rng.integers(1, 7))
        cols = []
        for v in range(nv):
            n = int(rng.choice([0, 1, 63, 64, 255, 256, 257, 1023, 1024, 1025, 4096, 16383, 16384, 16385, 33000, 70000]))
            how = rng.random()             # in position order / shuffled / sorted per contig (2 ... 100 runs; beyond 64 the scatter takes them)
            order = 1 if how < 0.6 else 0 if how < 0.8 else int(rng.choice([2, 3, 7, 24, 63, 64, 65, 100]))
            cols.append(make_vcf(n, L, truth, ext, nb, order, int(rng.integers(0, 4))))
        try:
            res, glob = eng.classify_batch(cols, [tid] * nv, n_bins=nb, alleles=ext)
        except q.QmvtError as e:
            if e.code == -9 and ext:      # the documented de-duplication limit of the extended mode (long runs, many alleles)
                continue
            raise
        want_glob = np.zeros((3, nb), np.uint64)
        for v, (r, c) in enumerate(zip(res, cols)):
            cls, roc, sc = O.classify_columns(*c, *truth, n_bins=nb, ext=ext)
            ok = (np.array_equal(r["cls"], cls) and np.array_equal(r["roc"], roc)
                  and all(r["scalars"][k] == sc[k] for k in ("n_pass", "tp_lines", "fp_lines", "TP_R", "FP_R", "truth_unique", "sorted"))
                  and np.array_equal(r["tp_idx"], np.nonzero(cls == 3)[0]) and np.array_equal(r["fp_idx"], np.nonzero(cls == 1)[0]))
            want_glob += roc
            if not ok:
                bad += 1
                print("MISMATCH round %d vcf %d: ext=%s nb=%d L=%d T=%d n=%d scal=%s oracle=%s" % (it, v, ext, nb, L, T, len(c[0]), r["scalars"], sc))
                dc = np.nonzero(r["cls"] != cls)[0]
                print("  differs: cls %d places (first %s: got %s want %s), roc %s, tp_idx %s, fp_idx %s; batch sizes %s" % (
                    len(dc), dc[:8], r["cls"][dc[:8]], cls[dc[:8]], not np.array_equal(r["roc"], roc),
                    not np.array_equal(r["tp_idx"], np.nonzero(cls == 3)[0]), not np.array_equal(r["fp_idx"], np.nonzero(cls == 1)[0]), [len(x[0]) for x in cols]))
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True); np.savez_compressed(os.path.join(ROOT, "gpurun_out", "fuzz_fail_%d_%d_%d.npz" % (seed, it, v)), *c, *truth)
        if not np.array_equal(glob[tid], want_glob):
            bad += 1
            print("MISMATCH round %d: per-truth sums" % it)
    print("gpu fuzz: %d rounds, seed %d, %d mismatches, %.0f s" % (rounds, seed, bad, time.time() - t0))
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)
