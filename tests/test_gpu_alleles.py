"""Allele-extended mode (QM_BATCH_ALLELES; BASELINE.json configs[4]: mixed SNP + indel with
variable-length alleles) on the GPU against the oracle's extended restatement.  The mode is a
build-defined widening of the reference's single-base filter; the single-base subset must behave
exactly as without it."""
import numpy as np
import pytest

from conftest import random_columns, random_truth
from test_gpu_parity import check_vcf

pytestmark = pytest.mark.gpu


def inline_code(s):
    c = len(s) << 26
    for k, ch in enumerate(s):
        c |= "ACGT".index(ch) << (2 * k)
    return c


def random_alleles(rng, n, p_ext=0.3):
    """allele codes: single bases, inline 2..13, dictionary ids, and codes that take no part"""
    out = rng.integers(0, 4, size=n).astype(np.int64)
    kind = rng.random(n)
    ln = rng.integers(2, 14, size=n)
    bits = rng.integers(0, 1 << 26, size=n)
    inl = (ln << 26) | (bits & ((1 << (2 * ln)) - 1))
    small = (2 << 26) | rng.integers(0, 16, size=n)            # few distinct values: collisions and repeats
    dic = 0x40000000 | rng.integers(0, 50, size=n)
    out = np.where(kind < p_ext * 0.4, inl, out)
    out = np.where((kind >= p_ext * 0.4) & (kind < p_ext * 0.8), small, out)
    out = np.where((kind >= p_ext * 0.8) & (kind < p_ext), dic, out)
    out = np.where(kind > 0.97, rng.choice(np.array([-1, 4, 7, 0x07ffffff, -2147483648]), size=n), out)
    return out.astype(np.int32)


def ext_truth(rng, t, L):
    tpos = rng.integers(1, L + 1, size=t).astype(np.int32)
    tpos[t // 2:] = tpos[: t - t // 2]        # several entries per position
    return tpos, random_alleles(rng, t, 0.5), random_alleles(rng, t, 0.5)


def ext_columns(rng, n, L, truth, frac_truth=0.35):
    tpos, tref, talt = truth
    pos = rng.integers(1, L + 1, size=n).astype(np.int32)
    ref, alt = random_alleles(rng, n), random_alleles(rng, n)
    if n and len(tpos):
        take = rng.random(n) < frac_truth
        j = rng.integers(0, len(tpos), size=n)
        pos = np.where(take, tpos[j], pos)
        ref = np.where(take, tref[j], ref)
        alt = np.where(take, talt[j], alt)
        near = rng.random(n) < 0.1                               # truth position, other alleles
        pos = np.where(near, tpos[j], pos)
        d = rng.random(n) < 0.08                                 # repeated records
        src = rng.integers(0, n, size=n)
        pos, ref, alt = np.where(d, pos[src], pos), np.where(d, ref[src], ref), np.where(d, alt[src], alt)
    qual = rng.integers(0, 300, size=n).astype(np.float32)
    qual = np.where(rng.random(n) < 0.05, np.float32(np.inf), qual).astype(np.float32)
    ok = lambda c: ((c >= 0) & (c < 4)) | (c >= 0x08000000)
    passed = ok(ref) & ok(alt) & (np.floor(qual) >= 20)
    iddot = rng.random(n) > 0.05
    nokey = rng.random(n) < 0.02
    flags = passed.astype(np.uint8) | (iddot.astype(np.uint8) << 1) | (nokey.astype(np.uint8) << 2)
    o = np.argsort(pos, kind="stable")
    c = lambda a, dt: np.ascontiguousarray(a[o], dt)
    return c(pos, np.int32), c(ref, np.int32), c(alt, np.int32), c(qual, np.float32), c(flags, np.uint8)


def check_ext(oracle, res, cols, truth):
    class X:   # the same checker, oracle in extended mode
        @staticmethod
        def classify_columns(*a, **k):
            return oracle.classify_columns(*a, ext=True, **k)
    check_vcf(X, res, cols, truth, expect_sorted=True)


SIZES = [0, 1, 5, 64, 255, 256, 257, 1023, 1024, 1025, 4097, 16384, 16385, 50000]


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_alleles_mode_vs_oracle(engine, oracle, seed):
    rng = np.random.default_rng(500 + seed)
    L = [3000, 60000, 2000000][seed]             # dense (long runs, oversize slices) .. sparse
    truths = [ext_truth(rng, [2500, 4000, 300][seed], L), ext_truth(rng, 50, L)]
    tids = [engine.truth_load(*t) for t in truths]
    for t, tid in zip(truths, tids):
        ok = lambda c: ((c >= 0) & (c < 4)) | (c >= 0x08000000)
        v = ok(t[1]) & ok(t[2])
        assert engine.truth_size(tid, alleles=True) == len(set(zip(t[0][v].tolist(), t[1][v].tolist(), t[2][v].tolist())))
    cols = [ext_columns(rng, n, L, truths[i % 2]) for i, n in enumerate(SIZES)]
    res, glob = engine.classify_batch(cols, [tids[i % 2] for i in range(len(SIZES))], alleles=True)
    for i, (r, c) in enumerate(zip(res, cols)):
        check_ext(oracle, r, c, truths[i % 2])
    for w in range(2):
        want = sum((r["roc"] for i, r in enumerate(res) if i % 2 == w), np.zeros((3, 256), np.uint64))
        assert np.array_equal(glob[tids[w]], want)


def test_alleles_mode_is_the_default_on_single_base_data(engine, oracle):
    rng = np.random.default_rng(77)
    L = 80000
    truth = random_truth(rng, 5000, L)
    tid = engine.truth_load(*truth)
    cols = [random_columns(rng, n, L, truth, sorted_=True, weird=False) for n in (3000, 40000)]
    a, _ = engine.classify_batch(cols, [tid, tid])
    b, _ = engine.classify_batch(cols, [tid, tid], alleles=True)
    for x, y, c in zip(a, b, cols):
        assert np.array_equal(x["cls"], y["cls"]) and np.array_equal(x["roc"], y["roc"]) and x["scalars"] == y["scalars"]
        check_vcf(oracle, y, c, truth, expect_sorted=True)


def test_longer_alleles_take_no_part_without_the_mode(engine, oracle):
    """default batches keep the reference's filter: extended codes are 'not a single base'"""
    rng = np.random.default_rng(78)
    L = 50000
    truth = ext_truth(rng, 3000, L)
    tid = engine.truth_load(*truth)
    cols = ext_columns(rng, 20000, L, truth)
    snp = (cols[1] >= 0) & (cols[1] < 4) & (cols[2] >= 0) & (cols[2] < 4)
    cols = cols[:4] + (np.where(snp, cols[4], cols[4] & 0xFE).astype(np.uint8),)   # bit0 only on single-base rows, as the packer does
    res, _ = engine.classify_batch([cols], [tid])
    check_vcf(oracle, res[0], cols, truth, expect_sorted=True)


def test_config5_shape_generated_on_device(engine, oracle):
    """BASELINE configs[4] shape: mixed SNP + indel records (30 %), three truth sets, VCF v uses truth v mod 3"""
    from oracle.synth import synth_truth_keys
    L, T, N, pct = 2_000_000, 50_000, 200_000, 30
    seeds = [5, 6, 7]
    tids = [engine.truth_synth(L, T, s, indel_pct=pct) for s in seeds]
    truths = [synth_truth_keys(L, T, s, pct) for s in seeds]
    for t, tid in zip(truths, tids):
        assert engine.truth_size(tid, alleles=True) == T
        assert engine.truth_size(tid) == int(((t[1] < 4) & (t[2] < 4)).sum())
    n_vcf = 6                   # ONE batch, VCF v against truth set v mod 3
    b = engine.batch([N] * n_vcf, [tids[v % 3] for v in range(n_vcf)], alleles=True)
    b.synth(L, T, None, 5000, indel_pct=pct)
    b.run()
    b.finish()
    roc, scal, glob = b.roc(), b.scalars(), b.global_counts()
    for v in range(n_vcf):
        cols = b.columns(v)
        assert 0.2 < float(((cols[1] >= 4) | (cols[2] >= 4)).mean()) < 0.4
        cls, oroc, sc = oracle.classify_columns(*cols, *truths[v % 3], ext=True)
        assert np.array_equal(b.cls(v), cls) and np.array_equal(roc[v], oroc)
        assert [int(x) for x in scal[v][:5]] == [sc[x] for x in ("n_pass", "tp_lines", "fp_lines", "TP_R", "FP_R")]
        assert int(scal[v][7]) == T and sc["tp_lines"] > 0.05 * N
        idx = b.idx(v)
        assert np.array_equal(idx[:sc["tp_lines"]], np.nonzero(cls == 3)[0])
        assert np.array_equal(idx[N - sc["fp_lines"]:], np.nonzero(cls == 1)[0])
    for w in range(3):          # the per-truth confusion counters (what the all-reduce carries) are sums of their VCFs' rows
        assert np.array_equal(glob[tids[w]], roc[w::3].sum(axis=0))
    b.close()


def test_alleles_mode_unsorted_vcfs_small_and_large(engine, oracle):
    """unsorted VCFs of an allele-extended batch: the small ones on the radix sort, 70 001 records on the bucket path (two streams)"""
    rng = np.random.default_rng(11)
    L = 40000
    truth = ext_truth(rng, 3000, L)
    tid = engine.truth_load(*truth)
    cols = []
    for n in (1, 300, 5000, 70001):
        c = ext_columns(rng, n, L, truth)
        o = rng.permutation(n)
        cols.append(tuple(np.ascontiguousarray(a[o]) for a in c))
    cols.append(ext_columns(rng, 9000, L, truth))            # one sorted VCF in the same batch
    res, _ = engine.classify_batch(cols, [tid] * len(cols), alleles=True)
    for i, (r, c) in enumerate(zip(res, cols)):
        cls, roc, sc = oracle.classify_columns(*c, *truth, ext=True)
        assert np.array_equal(r["cls"], cls) and np.array_equal(r["roc"], roc)
        for k in ("n_pass", "tp_lines", "fp_lines", "TP_R", "FP_R", "truth_unique"):
            assert r["scalars"][k] == sc[k], (i, k)
        assert r["scalars"]["sorted"] == sc["sorted"]
        assert np.array_equal(r["tp_idx"], np.nonzero(cls == 3)[0]) and np.array_equal(r["fp_idx"], np.nonzero(cls == 1)[0])


def test_alleles_mode_gives_up_on_overlong_runs(engine):
    from quasimodo_amd import QmvtError
    tid = engine.truth_load(np.array([10], np.int32), np.array([0], np.int32), np.array([inline_code("AC")], np.int32))
    # 40 000 kept records at one position, all alleles different: the de-duplication walk gives up
    n = 40000
    pos = np.full(n, 77, np.int32)
    alt = ((13 << 26) | np.arange(n)).astype(np.int32)
    with pytest.raises(QmvtError) as e:
        engine.classify_batch([(pos, np.zeros(n, np.int32), alt, np.full(n, 50, np.float32), np.full(n, 3, np.uint8))], [tid], alleles=True)
    assert e.value.code == -9
    # the limit is reported from the radix-sort path too (found by tools/gpu_fuzz.py: it used to be swallowed there)
    o = np.random.default_rng(3).permutation(n)
    pos2 = np.where(np.arange(n) % 2 == 0, 77, 5).astype(np.int32)[o]
    with pytest.raises(QmvtError) as e:
        engine.classify_batch([(pos2, np.zeros(n, np.int32), alt[o], np.full(n, 50, np.float32), np.full(n, 3, np.uint8))], [tid], alleles=True)
    assert e.value.code == -9
    # the same run with a handful of alleles is fine (walks stay short)
    alt = ((13 << 26) | (np.arange(n) % 7)).astype(np.int32)
    res, _ = engine.classify_batch([(pos, np.zeros(n, np.int32), alt, np.full(n, 50, np.float32), np.full(n, 3, np.uint8))], [tid], alleles=True)
    assert res[0]["scalars"]["FP_R"] == 7 and res[0]["scalars"]["fp_lines"] == n


def test_alleles_mode_unsorted_vcfs_on_the_bucket_path(engine, oracle):
    """Allele-extended VCFs out of order take the bucket path with TWO entry streams: single-base records through k_join_lean,
    the others -- 16-byte entries with their allele codes -- through k_join_ext (exact on position, REF, ALT).  Synthetic config-5
    VCFs against the sorted run, and an adversarial VCF (multi-allelic positions: many different and repeated alleles on one
    position, on truth positions and off them, keyless and non-'.' records) against the oracle."""
    from oracle.synth import synth_truth_keys
    L, T, N, pct = 2_000_000, 50_000, 400_000, 30
    tid = engine.truth_synth(L, T, 5, indel_pct=pct)
    rows = {}
    for shuffled in (False, True):
        b = engine.batch([N] * 5, [tid] * 5, alleles=True)
        b.synth(L, T, 5, 5000, shuffled=shuffled, indel_pct=pct)
        b.run(); b.finish()
        rows[shuffled] = (b.roc(), b.scalars()[:, :5].copy())
        if shuffled:
            ps = b.path_stats()
            assert ps["unsorted"] == 5 and ps["bucket_direct"] == 5 and ps["radix"] == 0 and ps["radix_after_overflow"] == 0
            cols = b.columns(3)
            cls, oroc, sc = oracle.classify_columns(*cols, *synth_truth_keys(L, T, 5, pct), ext=True)
            assert np.array_equal(b.cls(3), cls) and np.array_equal(rows[True][0][3], oroc)
            idx = b.idx(3)
            assert np.array_equal(idx[:sc["tp_lines"]], np.nonzero(cls == 3)[0]) and np.array_equal(idx[N - sc["fp_lines"]:], np.nonzero(cls == 1)[0])
        b.close()
    assert np.array_equal(rows[True][0], rows[False][0]) and np.array_equal(rows[True][1], rows[False][1])
    engine.truth_release(tid)
    # adversarial: 60 000 records on 20 000 positions, a third of them multi-allelic with a handful of alleles each
    rng = np.random.default_rng(77)
    Lp = 300_000
    truth = ext_truth(rng, 6000, Lp)
    tid2 = engine.truth_load(*truth)
    c = ext_columns(rng, 60_000, Lp, truth)
    pos, ref, alt, qual, flags = (a.copy() for a in c)
    hot = rng.choice(pos, 300)
    m = rng.random(len(pos)) < 0.3
    pos[m] = rng.choice(hot, int(m.sum()))
    alt[m] = np.where(rng.random(int(m.sum())) < 0.7, ((3 << 26) | rng.integers(0, 6, int(m.sum()))), alt[m]).astype(np.int32)
    o = rng.permutation(len(pos))
    cols = tuple(np.ascontiguousarray(a[o]) for a in (pos, ref, alt, qual, flags))
    res, _ = engine.classify_batch([cols], [tid2], alleles=True)
    cls, roc, sc = oracle.classify_columns(*cols, *truth, ext=True)
    assert np.array_equal(res[0]["cls"], cls) and np.array_equal(res[0]["roc"], roc)
    for k in ("n_pass", "tp_lines", "fp_lines", "TP_R", "FP_R", "truth_unique"):
        assert res[0]["scalars"][k] == sc[k], k
    engine.truth_release(tid2)


def test_alleles_mode_bucket_limits_send_the_chunk_to_the_radix_sort(engine, oracle):
    """What does not fit a bucket of the second stream -- here more than 1 024 records on positions that several kept records
    outside the truth set claim (multi-allelic sites packed into one bucket's position range) -- flags the VCF, nothing of the
    chunk joins the per-truth sums, and the radix sort redoes it: the answers are the oracle's, the statistics say so."""
    rng = np.random.default_rng(123)
    Lp = 4_000_000                       # 123 buckets of 2^15 positions
    truth = ext_truth(rng, 3000, Lp)
    tid = engine.truth_load(*truth)
    n = 60_000
    pos, ref, alt, qual, flags = (a.copy() for a in ext_columns(rng, n, Lp, truth))
    # 4 000 records on 1 300 positions inside ONE bucket, three different long alleles each: every one of them is listed
    hot = 70_000 + rng.choice(20_000, 1300, replace=False)
    m = rng.choice(n, 4000, replace=False)
    pos[m] = rng.choice(hot, 4000).astype(np.int32)
    alt[m] = ((3 << 26) | rng.integers(0, 64, 4000)).astype(np.int32)
    ref[m] = 0
    flags[m] = 3
    qual[m] = 60
    o = rng.permutation(n)
    cols = tuple(np.ascontiguousarray(a[o]) for a in (pos, ref, alt, qual, flags))
    b = engine.batch([n], [tid], alleles=True)
    b.upload(0, *cols)
    b.run(); b.finish()
    ps = b.path_stats()
    assert ps["unsorted"] == 1 and ps["overflow_chunks"] == 1 and ps["radix_after_overflow"] == 1 and ps["bucket_direct"] == 0
    cls, roc, sc = oracle.classify_columns(*cols, *truth, ext=True)
    assert np.array_equal(b.cls(0), cls) and np.array_equal(b.roc()[0], roc)
    assert [int(x) for x in b.scalars()[0, :5]] == [sc[k] for k in ("n_pass", "tp_lines", "fp_lines", "TP_R", "FP_R")]
    assert np.array_equal(b.global_counts()[tid], roc.astype(np.uint64))
    b.run(); b.finish()                  # and again, now known to be out of order
    assert np.array_equal(b.cls(0), cls) and np.array_equal(b.roc()[0], roc) and np.array_equal(b.global_counts()[tid], roc.astype(np.uint64))
    b.close()
    engine.truth_release(tid)


def test_alleles_mode_unsorted_vcfs_of_configs4_shape_take_partitions_of_buckets(engine, oracle):
    """BASELINE configs[4]'s VCFs -- 2 M records on a 10 Mb reference, 30 % with variable-length alleles -- out of order: too many
    records and too wide a key range for 256 buckets, so every partition of 2^27 keys is a segment of the one-level scatter that
    reads its VCF's columns and keeps its own key range (SortSeg.part), two streams and two joins per bucket as before.  Against
    the sorted run (all VCFs) and the oracle (one VCF); no radix sort."""
    from oracle.synth import synth_truth_keys
    L, T, N, pct = 10_000_000, 200_000, 2_000_000, 30
    tids = [engine.truth_synth(L, T, s, indel_pct=pct) for s in (5, 6)]
    nv = 4
    rows = {}
    for shuffled in (False, True):
        b = engine.batch([N] * nv, [tids[v % 2] for v in range(nv)], alleles=True)
        b.synth(L, T, None, 5000, shuffled=shuffled, indel_pct=pct)
        for rep in range(2 if shuffled else 1):       # (the second run: the batch knows its VCFs are out of order)
            b.run(); b.finish()
            if shuffled:
                ps = b.path_stats()
                assert ps["unsorted"] == nv and ps["bucket_partitions"] == nv and ps["radix"] == 0 and ps["radix_after_overflow"] == 0, ps
        rows[shuffled] = (b.roc(), b.scalars()[:, :5].copy(), b.global_counts())
        if shuffled:
            cols = b.columns(1)
            cls, oroc, sc = oracle.classify_columns(*cols, *synth_truth_keys(L, T, 6, pct), ext=True)
            assert np.array_equal(b.cls(1), cls) and np.array_equal(rows[True][0][1], oroc)
            idx = b.idx(1)
            assert np.array_equal(idx[:sc["tp_lines"]], np.nonzero(cls == 3)[0]) and np.array_equal(idx[N - sc["fp_lines"]:], np.nonzero(cls == 1)[0])
        b.close()
    for k in range(3):
        assert np.array_equal(rows[True][k], rows[False][k]), k
    for t in tids:
        engine.truth_release(t)
