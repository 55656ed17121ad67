#!/usr/bin/env python3
"""Randomised parity hunt at the sizes where the unsorted paths hand over to one another: VCFs of 20 000 ... 4 M records on
references of 3 ... 120 M positions against truth sets of 10^3 ... 3 x 10^6 keys, in position order, shuffled or sorted per
contig, default and allele-extended mode, with the library's own routing (nothing forced) -- one level with the bit-map or the
hashed join, a pair of narrow partitions, wide buckets, two levels, the radix sort and every fallback between them.  Every VCF
against the oracle, twice (the second run with the batch's memory of the first); prints which paths the rounds took.
usage: python3 tools/gpu_fuzz_big.py [rounds] [seed]"""
import collections
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import quasimodo_amd as q
from quasimodo_amd.engine import SCALAR_NAMES
from oracle import qm_oracle as O
from conftest import random_columns, random_truth


def run(rounds, seed):
    rng = np.random.default_rng(seed)
    dry = os.environ.get("QM_FUZZ_DRY") is not None   # no GPU: the rounds' shapes only (to find the round behind a fault without running it again)
    eng = None if dry else q.Engine(0)
    O.build()
    bad = 0
    paths = collections.Counter()
    t0 = time.time()
    for it in range(rounds):
        ext = rng.random() < float(os.environ.get("QM_FUZZ_EXT", "0.25"))   # share of the rounds in the allele-extended mode
        L = int(np.exp(rng.uniform(np.log(3e6), np.log(1.2e8 if not ext else 3.0e7))))
        many = rng.random() < 0.2      # a fifth of the rounds: up to twelve VCFs (smaller ones), several chunks' worth of work for the fallbacks
        nv = int(rng.integers(4, 13)) if many else int(rng.integers(1, 4))
        sizes = [int(np.exp(rng.uniform(np.log(2e4), np.log(1.5e6 if many else 4e6)))) for _ in range(nv)]
        nb = int(rng.choice([256, 256, 256, 100, 21]))
        T = int(np.exp(rng.uniform(np.log(1e3), np.log(3e6))))
        if rng.random() < 0.7:
            T = max(T, max(sizes) // 8)   # (random_columns puts 5 % of the records on truth POSITIONS: against a small truth set every bucket overflows and the round only tests the radix sort)
        if rng.random() < 0.8:
            T = max(1000, min(T, L // 50))   # (a bucket stages the truth keys of 3 % of its positions at most: denser truth sets also only test the radix sort)
        truth = random_truth(rng, T, L)
        if ext:   # a third of the truth entries with a two-base ALT (a packed allele code, include/qmvt.h)
            k = rng.random(T) < 0.3
            truth = (truth[0], truth[1], np.where(k, (2 << 26) | rng.integers(0, 16, T), truth[2]).astype(np.int32))
        tid = None if dry else eng.truth_load(*truth)
        # a second truth set (a subset of the first) for every other VCF in a third of the rounds
        truth2 = tuple(a[::3] for a in truth) if rng.random() < 0.33 else None
        tid2 = None if (dry or truth2 is None) else eng.truth_load(*truth2)
        cols, notes = [], []
        for v in range(nv):
            n = sizes[v]
            how = rng.random()
            ft = min(float(rng.choice([0.02, 0.1, 0.3])), 2.0 * T / n) if rng.random() < 0.8 else 0.3   # (mostly: no more than two records per truth key, or every bucket overflows)
            dup = float(rng.choice([0.0, 0.01, 0.05]))
            c = random_columns(rng, n, L, truth, frac_truth=ft, sorted_=False, dup_frac=dup, weird=not ext,
                               near_frac=min(0.05, 0.2 * T / n))   # (records on truth POSITIONS with other alleles: at most one per five truth keys, or the exact sets overflow)
            if ext:   # a fifth of the records not on a truth key get a two-base ALT; kept as the extended mode keeps them
                pos, ref, alt, qual, flags = c
                k = rng.random(n) < 0.2
                alt = np.where(k & (alt < 4), (2 << 26) | rng.integers(0, 16, n), alt).astype(np.int32)
                okc = lambda a: ((a >= 0) & (a < 4)) | (a >= 0x08000000)
                flags = ((flags & 0xfe) | (okc(ref) & okc(alt) & (np.floor(qual) >= 20)).astype(np.uint8)).astype(np.uint8)
                c = (pos, ref, alt, qual, flags)
            if how < 0.25:
                o = np.argsort(c[0], kind="stable")
                c = tuple(a[o] for a in c)
            if 0.25 <= how < 0.45:   # sorted per contig: a few ascending runs
                runs = int(rng.choice([2, 5, 24]))
                cuts = np.sort(rng.choice(np.arange(1, n), size=runs - 1, replace=False))
                o = np.concatenate([part[np.argsort(c[0][part], kind="stable")] for part in np.split(np.arange(n), cuts)])
                c = tuple(a[o] for a in c)
            crowded = rng.random() < 0.15
            notes.append("%s ft=%.3f dup=%.2f%s" % ("sorted" if how < 0.25 else "runs" if how < 0.45 else "shuffled", ft, dup, " CROWD" if crowded else ""))
            if crowded:  # a crowd on a few thousand positions: buckets overflow, the fallbacks run
                crowd = rng.random(n) < 0.3
                c = (np.where(crowd, L // 2 + rng.integers(0, 3000, n), c[0]).astype(np.int32),) + tuple(c[1:])
                if how < 0.25:
                    o = np.argsort(c[0], kind="stable")
                    c = tuple(a[o] for a in c)
            cols.append(tuple(np.ascontiguousarray(a) for a in c))
        if dry:
            print("round %d: ext=%s L=%d T=%d sizes=%s max_pos=%s sorted=%s" % (it, ext, L, T, [len(c[0]) for c in cols], [int(c[0].max()) for c in cols],
                                                                             [bool((np.diff(c[0]) >= 0).all()) for c in cols]), flush=True)
            continue
        try:
            tids = [tid2 if (tid2 is not None and v % 2) else tid for v in range(nv)]
            b = eng.batch([len(c[0]) for c in cols], tids, n_bins=nb, alleles=ext)
            for v, c in enumerate(cols):
                b.upload(v, *c)
            for rep in range(2):
                b.run(); b.finish()
                ps = b.path_stats()
                for k, x in ps.items():
                    if x and k not in ("bucket_chunks", "overflow_chunks", "radix_chunks"):
                        paths[k] += x
                for v, c in enumerate(cols):
                    cls, roc, sc = O.classify_columns(*c, *(truth2 if tids[v] == tid2 and tid2 is not None else truth), n_bins=nb, ext=ext)
                    s = dict(zip(SCALAR_NAMES, b.scalars()[v].tolist()))
                    reg = b.idx(v)
                    ok = (np.array_equal(b.cls(v), cls) and np.array_equal(b.roc()[v], roc)
                          and all(s[k] == sc[k] for k in ("n_pass", "tp_lines", "fp_lines", "TP_R", "FP_R", "truth_unique", "sorted"))
                          and np.array_equal(reg[:s["tp_lines"]], np.nonzero(cls == 3)[0]) and np.array_equal(reg[len(c[0]) - s["fp_lines"]:], np.nonzero(cls == 1)[0]))
                    if not ok:
                        bad += 1
                        print("MISMATCH round %d rep %d vcf %d: ext=%s L=%d T=%d n=%d paths=%s scal=%s oracle=%s" % (it, rep, v, ext, L, T, len(c[0]), ps, s, sc), flush=True)
            if os.environ.get("QM_FUZZ_VERBOSE"):
                print("round %d: ext=%s L=%d T=%d sizes=%s %s -> %s" % (it, ext, L, T, [len(c[0]) for c in cols], notes, {k: x for k, x in ps.items() if x}), flush=True)
            b.close()
        except q.QmvtError as e:
            if e.code == -9 and ext:      # the documented de-duplication limit of the extended mode
                paths["ext_limit"] += 1
            else:
                raise
        eng.truth_release(tid)
        if tid2 is not None:
            eng.truth_release(tid2)
        if (it + 1) % 10 == 0:
            print("  %d rounds, %d mismatches, %.0f s, paths %s" % (it + 1, bad, time.time() - t0, dict(paths)), flush=True)
    print("gpu fuzz (big): %d rounds, seed %d, %d mismatches, %.0f s; VCFs by path: %s" % (rounds, seed, bad, time.time() - t0, dict(paths)))
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)
