import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_cases():
    with open(os.path.join(GOLDEN, "manifest.json")) as fh:
        return json.load(fh)


def case_id(e):
    return "%s:%s:%s" % (e["family"], e["mode"], os.path.basename(e["vcf"]))


def read_case(e):
    fam = os.path.join(GOLDEN, e["family"])
    rd = lambda rel: open(os.path.join(fam, rel), "rb").read()
    exp = {k: (rd(p) if p else None) for k, p in e["expected"].items()}
    return rd(e["vcf"]), rd(e["truth"]), exp


@pytest.fixture(scope="session")
def oracle():
    from oracle import qm_oracle
    qm_oracle.build()
    return qm_oracle


@pytest.fixture(scope="session")
def qmlib():
    """libqmvt.so must be built (in-tree); building it needs hipcc, not a GPU."""
    import quasimodo_amd
    from quasimodo_amd import _lib
    if not os.path.exists(_lib.library_path()):
        quasimodo_amd.build_library()
    return _lib.lib()


@pytest.fixture(scope="session")
def engine(qmlib):
    import quasimodo_amd
    eng = quasimodo_amd.Engine(0)   # raises loudly when there is no HIP device
    yield eng
    eng.close()


def random_columns(rng, n, genome_len, truth, frac_truth=0.3, sorted_=True, dup_frac=0.05, weird=True, near_frac=0.05):
    """Seeded column-level test input: (pos, ref, alt, qual, flags) with hits, duplicates,
    same-position runs, non-SNP codes, non-'.' IDs and failing QUALs."""
    tpos, tref, talt = truth
    pos = rng.integers(1, genome_len + 1, size=n).astype(np.int32)
    ref = rng.integers(0, 4, size=n).astype(np.int32)
    alt = rng.integers(0, 4, size=n).astype(np.int32)
    if len(tpos) and n:
        take = rng.random(n) < frac_truth
        j = rng.integers(0, len(tpos), size=n)
        pos = np.where(take, tpos[j], pos).astype(np.int32)
        ref = np.where(take, tref[j], ref).astype(np.int32)
        alt = np.where(take, talt[j], alt).astype(np.int32)
        # same position as a truth key but another allele
        near = rng.random(n) < near_frac
        pos = np.where(near, tpos[j], pos).astype(np.int32)
    if n:
        d = rng.random(n) < dup_frac                 # duplicate of another record
        src = rng.integers(0, n, size=n)
        pos = np.where(d, pos[src], pos); ref = np.where(d, ref[src], ref); alt = np.where(d, alt[src], alt)
    if weird and n:
        ns = rng.random(n) < 0.08                   # not a single-base allele
        alt = np.where(ns, rng.integers(4, 9, size=n), alt).astype(np.int32)
        ns2 = rng.random(n) < 0.03
        ref = np.where(ns2, 4, ref).astype(np.int32)
    qual = rng.integers(0, 300, size=n).astype(np.float32)
    if weird and n:
        qual = np.where(rng.random(n) < 0.05, np.float32(np.inf), qual)
        qual = np.where(rng.random(n) < 0.03, np.float32(-np.inf), qual)
        qual = np.where(rng.random(n) < 0.05, qual + np.float32(0.5), qual).astype(np.float32)
    snp = (ref >= 0) & (ref < 4) & (alt >= 0) & (alt < 4)
    passed = snp & (np.floor(qual) >= 20)
    iddot = rng.random(n) > (0.05 if weird else 0.0)
    nokey = (rng.random(n) < 0.02) if weird else np.zeros(n, bool)
    flags = (passed.astype(np.uint8)) | (iddot.astype(np.uint8) << 1) | (nokey.astype(np.uint8) << 2)
    if weird and n:
        # decisions of the host path (include/qmvt.h): QM_F_TPLINE on a few records (with or without '.' ID, with or
        # without a key in the truth set), QM_F_IDDOT cleared on a few others
        tpl = np.random.default_rng(int(n) * 7919 + 13).random(n) < 0.03
        flags = flags | (tpl.astype(np.uint8) << 3)
    if sorted_ and n:
        o = np.argsort(pos, kind="stable")
        pos, ref, alt, qual, flags = pos[o], ref[o], alt[o], qual[o], flags[o]
    return (np.ascontiguousarray(pos, np.int32), np.ascontiguousarray(ref, np.int32), np.ascontiguousarray(alt, np.int32),
            np.ascontiguousarray(qual, np.float32), np.ascontiguousarray(flags, np.uint8))


def random_truth(rng, t, genome_len):
    tpos = rng.integers(1, genome_len + 1, size=t).astype(np.int32)
    tref = rng.integers(0, 4, size=t).astype(np.int32)
    talt = rng.integers(0, 4, size=t).astype(np.int32)
    return tpos, tref, talt
