"""numpy restatement of the synthetic-workload generator (quasimodo_amd/csrc/qmvt_dev.h: mix64 / hash3 /
synth_truth / synth_record; DESIGN.md section 4.5 is the specification).  Test infrastructure: lets the
oracle classify device-generated VCFs without asking the engine for its keys, and lets anybody regenerate
the bench workload -- truth sets AND VCF columns -- from the specification alone (synth_vcf_columns)."""
import numpy as np

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def mix64(x):
    x = (x + np.uint64(0x9e3779b97f4a7c15)) & _M
    x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xbf58476d1ce4e5b9)) & _M
    x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94d049bb133111eb)) & _M
    return x ^ (x >> np.uint64(31))


def hash3(seed, a, b):
    s = mix64(np.uint64(seed) ^ np.uint64(0x51ed270b7f3c9a11))
    return mix64((s + a * np.uint64(0x9e3779b97f4a7c15) + b * np.uint64(0xc2b2ae3d27d4eb4f)) & _M)


SYNTH_POOL = 1 << 20


def synth_allele(h, minlen):
    """qmvt_dev.h synth_allele: length minlen + Geom(0.5) capped at 32; <= 13 bases inline
    (len << 26 | 2-bit bases), longer ones an id of the synthetic dictionary."""
    with np.errstate(over="ignore"):
        h = np.asarray(h, np.uint64)
        ln = np.full(h.shape, minlen, np.int64)
        g = h.copy()
        alive = np.ones(h.shape, bool)
        for _ in range(32):
            alive &= (ln < 32) & ((g & np.uint64(1)) == np.uint64(1))
            ln += alive
            g = np.where(alive, g >> np.uint64(1), g)
        bits = mix64(h ^ np.uint64(0xa11e1e5))
        sh = np.minimum(2 * ln, 62).astype(np.uint64)
        inline = (ln.astype(np.uint64) << np.uint64(26)) | (bits & ((np.uint64(1) << sh) - np.uint64(1)))
        pool = np.uint64(0x40000000) | (bits % np.uint64(SYNTH_POOL))
        out = np.where(ln == 1, bits & np.uint64(3), np.where(ln <= 13, inline, pool))
    return out.astype(np.int64).astype(np.int32)


def synth_truth_keys(L, T, tseed, indel_pct=0):
    with np.errstate(over="ignore"):
        j = np.arange(T, dtype=np.uint64)
        wt = np.uint64(L // T)
        h = hash3(tseed, j, np.uint64(1))
        p = j * wt + np.uint64(1) + h % wt
        r = hash3(3, p, np.uint64(0)) & np.uint64(3)
        a = (r + np.uint64(1) + (h >> np.uint64(32)) % np.uint64(3)) & np.uint64(3)
        r, a = r.astype(np.int32), a.astype(np.int32)
        if indel_pct > 0:
            ind = (hash3(tseed, j, np.uint64(2)) % np.uint64(100)).astype(np.int64) < indel_pct
            r = np.where(ind, synth_allele(hash3(tseed, j, np.uint64(3)), 1), r)
            a = np.where(ind, synth_allele(hash3(tseed, j, np.uint64(4)), 2), a)
    return p.astype(np.int32), r.astype(np.int32), a.astype(np.int32)


def allele_string(code):
    """Spelling of an allele code (include/qmvt.h).  Dictionary ids are spelled as the synthetic
    dictionary defines them (14..32 bases from the id) -- only meaningful for generated data."""
    code = int(code)
    if 0 <= code < 4:
        return "ACGT"[code]
    if 0x08000000 <= code < 0x40000000:
        ln = code >> 26
        return "".join("ACGT"[(code >> (2 * k)) & 3] for k in range(ln))
    if code >= 0x40000000:
        i = code & 0x3fffffff
        ln = 14 + i % 19
        with np.errstate(over="ignore"):
            w = [int(mix64(np.uint64(i * 2 + k))) for k in range(2)]
        return "".join("ACGT"[(w[k // 32] >> (2 * (k % 32))) & 3] for k in range(ln))
    raise ValueError("not an allele code: %d" % code)


def synth_perm(n, vcf_sizes=None):
    """qm_batch_synth's record permutation for shuffled VCFs: slot i of a VCF of n records holds generated record
    (i * a + 12345) mod n, a = the first odd number >= 2654435761 coprime to the size of EVERY VCF of the batch."""
    from math import gcd
    sizes = list(vcf_sizes) if vcf_sizes is not None else [n]
    a = 2654435761
    while any(gcd(a, int(m)) != 1 for m in sizes):
        a += 2
    i = np.arange(n, dtype=object)
    return np.array((i * a + 12345) % n, dtype=np.int64)


def synth_vcf_columns(L, N, T, tseed, seed, indel_pct=0, shuffled=False, vcf_sizes=None):
    """qmvt_dev.h synth_record for every record of one VCF: (pos, ref, alt, qual, flags) as the device generates them.
    Genome 1..L in N strata of width L / N, one record per stratum; the truth set has one entry per stratum of width
    L / T; a record takes the truth entry that falls into its stratum with probability 0.8, otherwise a random position
    of its stratum with a random other base; QUAL uniform integer 0..255; ID '.'; PASS iff QUAL >= 20.
    VCF v of a batch uses seed = (batch seed) + v."""
    with np.errstate(over="ignore"):
        u = np.uint64
        i = np.arange(N, dtype=np.uint64)
        w, wt = u(L // N), u(L // T)
        s0 = i * w + u(1)
        j = (s0 - u(1)) // wt
        tp, tr, ta = synth_truth_keys(L, T, tseed, indel_pct)
        jj = np.minimum(j, u(T - 1)).astype(np.int64)
        tpj = tp[jj].astype(np.uint64)
        take = (j < u(T)) & (tpj >= s0) & (tpj < s0 + w) & ((hash3(seed, i, u(10)) % u(10)) < u(8))
        pp = s0 + hash3(seed, i, u(11)) % w
        rr = hash3(3, pp, u(0)) & u(3)
        aa = (rr + u(1) + hash3(seed, i, u(12)) % u(3)) & u(3)
        rr, aa = rr.astype(np.int32), aa.astype(np.int32)
        if indel_pct > 0:
            ind = (hash3(seed, i, u(14)) % u(100)).astype(np.int64) < indel_pct
            rr = np.where(ind, synth_allele(hash3(seed, i, u(15)), 1), rr)
            aa = np.where(ind, synth_allele(hash3(seed, i, u(16)), 2), aa)
        pos = np.where(take, tpj, pp).astype(np.int32)
        ref = np.where(take, tr[jj], rr).astype(np.int32)
        alt = np.where(take, ta[jj], aa).astype(np.int32)
        qi = (hash3(seed, i, u(13)) & u(255)).astype(np.int64)
        qual = qi.astype(np.float32)
        flags = (2 | (qi >= 20)).astype(np.uint8)
    if shuffled is True or shuffled == 1:
        p = synth_perm(N, vcf_sizes)
        pos, ref, alt, qual, flags = pos[p], ref[p], alt[p], qual[p], flags[p]
    elif shuffled:
        p = synth_runs_perm(N, int(shuffled))
        pos, ref, alt, qual, flags = pos[p], ref[p], alt[p], qual[p], flags[p]
    return pos, ref, alt, qual, flags


def synth_runs_perm(n, runs):
    """qm_batch_synth with shuffled = R >= 2 (a VCF of R contigs): R ascending runs one behind the other, run c = the generated
    records c, c + R, c + 2 R, ...; returns the generated record every slot holds"""
    return np.concatenate([np.arange(c, n, runs, dtype=np.int64) for c in range(runs)])
