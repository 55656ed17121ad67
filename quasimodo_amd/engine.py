"""Engine / Batch: thin object layer over the C ABI (include/qmvt.h)."""
import ctypes as C
import weakref

import numpy as np

from . import _lib
from ._lib import SCALAR_NAMES, QmvtError, SynthCfg, check

# include/qmvt.h QM_PATH_*: where the VCFs a finish found out of order went
PATH_NAMES = ("unsorted", "bucket_direct", "bucket_hashed", "radix", "radix_after_overflow", "bucket_chunks", "overflow_chunks",
              "radix_chunks", "bucket_two_level", "bucket_partitions")


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


class Engine:
    """One context per process and GPU (qm_init).  Raises QmvtError(QM_E_NODEVICE)
    when no HIP device is usable -- there is no CPU path."""

    def __init__(self, device=0):
        self._L = _lib.lib()
        h = C.c_void_p()
        check(self._L.qm_init(int(device), C.byref(h)))
        self._h = h
        self.device = int(device)
        self._batches = weakref.WeakSet()   # a batch must not outlive its context (qm_batch_destroy uses it)

    def close(self):
        if getattr(self, "_h", None):
            for b in list(self._batches):
                b.close()
            self._L.qm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- truth sets -----------------------------------------------------------
    def truth_load(self, pos, ref, alt):
        pos, ref, alt = _c(pos, np.int32), _c(ref, np.int32), _c(alt, np.int32)
        tid = C.c_int(-1)
        check(self._L.qm_truth_load(self._h, _p(pos), _p(ref), _p(alt), pos.shape[0], C.byref(tid)), self._h)
        return tid.value

    def truth_synth(self, genome_len, truth_n, truth_seed, indel_pct=0):
        tid = C.c_int(-1)
        check(self._L.qm_truth_synth_ext(self._h, int(genome_len), int(truth_n), int(truth_seed), int(indel_pct), C.byref(tid)), self._h)
        return tid.value

    def truth_size(self, tid, alleles=False):
        """distinct single-base keys; alleles=True: distinct valid keys of any allele length"""
        n = C.c_int64()
        check((self._L.qm_truth_size_ext if alleles else self._L.qm_truth_size)(self._h, int(tid), C.byref(n)), self._h)
        return n.value

    def truth_release(self, tid):
        """free a truth set's HBM; its id may be reused by a later load"""
        check(self._L.qm_truth_release(self._h, int(tid)), self._h)

    @property
    def n_truth(self):
        return self._L.qm_truth_count(self._h)

    # -- one-shot ---------------------------------------------------------------
    def classify_batch(self, columns, truth_ids, n_bins=256, alleles=False):
        """columns: list of (pos, ref, alt, qual, flags) per VCF.  Returns (per-VCF result dicts,
        per-truth sums): what qm_classify_batch computes (include/qmvt.h), through the resident-batch
        entry points so that each VCF's arrays are uploaded from where they are (no concatenation).
        alleles=True: the allele-extended mode (QM_BATCH_ALLELES)."""
        n_vcf = len(columns)
        if n_vcf == 0:
            return [], np.zeros((max(self.n_truth, 1), 3, n_bins), np.uint64)
        sizes = [int(np.asarray(c[0]).shape[0]) for c in columns]
        b = Batch(self, sizes, truth_ids, n_bins, alleles)
        try:
            for v, c in enumerate(columns):
                b.upload(v, *c)
            b.run()
            b.finish()
            roc, scal, glob = b.roc(), b.scalars(), b.global_counts()
            out = []
            for v in range(n_vcf):
                s = dict(zip(SCALAR_NAMES, scal[v].tolist()))
                reg = b.idx(v)
                out.append({"cls": b.cls(v), "roc": roc[v].copy(), "scalars": s,
                            "tp_idx": reg[:s["tp_lines"]].copy(), "fp_idx": reg[sizes[v] - s["fp_lines"]:].copy()})
        finally:
            b.close()
        return out, glob

    def classify_batch_oneshot(self, columns, truth_ids, n_bins=256, alleles=False):
        """The same through the single C call qm_classify_batch(_ext) on concatenated host buffers."""
        n_vcf = len(columns)
        sizes = [int(np.asarray(c[0]).shape[0]) for c in columns]
        offs = np.zeros(n_vcf + 1, np.int64)
        offs[1:] = np.cumsum(sizes)
        N = int(offs[-1])
        cat = lambda k, dt: _c(np.concatenate([np.asarray(c[k], dtype=dt) for c in columns]) if n_vcf else np.zeros(0, dt), dt)
        pos, ref, alt = cat(0, np.int32), cat(1, np.int32), cat(2, np.int32)
        qual, flags = cat(3, np.float32), cat(4, np.uint8)
        tids = _c(truth_ids, np.int32)
        cls = np.zeros(max(N, 1), np.uint8)
        idx = np.zeros(max(N, 1), np.int32)
        roc = np.zeros((n_vcf, 3, n_bins), np.uint64)
        scal = np.zeros((n_vcf, _lib.QM_N_SCALARS), np.int64)
        glob = np.zeros((max(self.n_truth, 1), 3, n_bins), np.uint64)
        check(self._L.qm_classify_batch_ext(self._h, n_vcf, _p(offs), _p(pos), _p(ref), _p(alt), _p(qual), _p(flags), _p(tids),
                                            int(n_bins), _lib.QM_BATCH_ALLELES if alleles else 0, _p(cls), _p(roc), _p(scal),
                                            _p(idx), _p(glob)), self._h)
        out = []
        for v in range(n_vcf):
            a, b = int(offs[v]), int(offs[v + 1])
            s = dict(zip(SCALAR_NAMES, scal[v].tolist()))
            reg = idx[a:b]
            out.append({"cls": cls[a:b].copy(), "roc": roc[v].copy(), "scalars": s,
                        "tp_idx": reg[:s["tp_lines"]].copy(), "fp_idx": reg[(b - a) - s["fp_lines"]:].copy()})
        return out, glob

    def bench_synth(self, n_vcf, records, genome_len, truth_n, truth_seed=3, seed=3000, n_bins=256, steps=5, shuffled=False, indel_pct=0):
        """qm_bench_synth: one self-contained synthetic run inside the library."""
        cfg = SynthCfg(int(genome_len), int(seed), int(truth_seed), int(truth_n), int(bool(shuffled)), int(indel_pct))
        r = _lib.BenchResult()
        check(self._L.qm_bench_synth(self._h, C.byref(cfg), int(n_vcf), int(records), int(n_bins), int(steps), C.byref(r)), self._h)
        return {k: getattr(r, k) for k, _ in r._fields_ if k != "reserved"}

    def extract_files(self, file_jobs, n_bins=256, alleles=False, strict=True, truth_slots=None, n_slots=0, global_dev=None):
        """qm_extract_files(_ex): files in, files out, everything between in the library (host threads + ONE engine batch).
        file_jobs: list of dicts vcf / truth / mode ("hcmv" | "custom") / pure / filtered / tp / fp.
        truth_slots / n_slots / global_dev (device pointer, int): one rank of a multi-GPU run -- the per-truth-file sums of
        this call's VCFs land in rows truth_slots[j] of the caller's [n_slots][3][n_bins] uint64 device buffer (cleared first).
        Returns (list of per-VCF dicts: scalars by name + n_lines, genomediff, header_kept, host_decided, roc; phase seconds)."""
        import os
        n = len(file_jobs)
        arr = (_lib.FileJob * max(n, 1))()
        enc = lambda p: None if p is None else os.fsencode(p)
        for k, j in enumerate(file_jobs):
            arr[k] = _lib.FileJob(enc(j["vcf"]), enc(j.get("truth")), 1 if j.get("mode", "hcmv") == "custom" else 0, int(bool(j.get("pure"))),
                                  enc(j["filtered"]), enc(j.get("tp")), enc(j["fp"]))
        st = (_lib.FileStats * max(n, 1))()
        roc = np.zeros((max(n, 1), 3, n_bins), np.uint64)
        ph = (C.c_double * 8)()
        slots = None if truth_slots is None else _c(list(truth_slots) + [0] * (1 if n == 0 else 0), np.int32)
        check(self._L.qm_extract_files_ex(self._h, n, arr, int(n_bins), _lib.QM_BATCH_ALLELES if alleles else 0, int(bool(strict)), st, _p(roc), ph,
                                          _p(slots), int(n_slots), C.c_void_p(global_dev) if global_dev else None), self._h)
        rows = []
        for k in range(n):
            r = dict(zip(SCALAR_NAMES, list(st[k].scalars)))
            r.update(n_lines=st[k].n_lines, n_refused=st[k].n_refused, genomediff=st[k].genomediff,
                     header_kept=(st[k].header_kept, st[k].header_kept_tp), host_decided=st[k].host_decided, r_hostile=st[k].r_hostile,
                     roc=roc[k].copy())
            rows.append(r)
        phases = dict(zip(("map_count", "truth_beside", "batch_layout", "tokenise_upload", "engine", "masks_back", "write", "release"), list(ph)))
        return rows, phases

    def path_stats_total(self):
        """qm_path_stats_total: where the VCFs found out of order went, summed over every batch this context has finished
        (PATH_NAMES); take the difference around a call."""
        out = np.zeros(len(PATH_NAMES), np.int64)
        check(self._L.qm_path_stats_total(self._h, _p(out)), self._h)
        return dict(zip(PATH_NAMES, (int(x) for x in out)))

    def bw_probe(self, nbytes=4 << 30, reps=5):
        """qm_bw_probe: GB/s this GPU streams read-only, copying (read + written) and write-only"""
        out = (C.c_double * 3)()
        check(self._L.qm_bw_probe(self._h, int(nbytes), int(reps), out), self._h)
        return {"read_GBps": out[0], "copy_GBps": out[1], "write_GBps": out[2]}

    def fp_overlap(self, key_sets):
        """key_sets: list of (pos, ref, alt) arrays, one per caller.  Returns region
        counts indexed by membership mask (snpcaller_fp_compare.R:36-47)."""
        n = len(key_sets)
        offs = np.zeros(n + 1, np.int64)
        offs[1:] = np.cumsum([np.asarray(k[0]).shape[0] for k in key_sets])
        cat = lambda j: _c(np.concatenate([np.asarray(k[j], np.int32) for k in key_sets]) if n else np.zeros(0, np.int32), np.int32)
        pos, ref, alt = cat(0), cat(1), cat(2)
        reg = np.zeros(1 << n, np.int64)
        check(self._L.qm_fp_overlap(self._h, n, _p(offs), _p(pos), _p(ref), _p(alt), _p(reg)), self._h)
        return reg

    def batch(self, n_records, truth_ids, n_bins=256, alleles=False):
        return Batch(self, n_records, truth_ids, n_bins, alleles)


class Batch:
    """Resident batch: columns stay in HBM across runs (qm_batch_*)."""

    def __init__(self, engine, n_records, truth_ids, n_bins=256, alleles=False):
        self.engine = engine
        self._L = engine._L
        self.n_records = _c(n_records, np.int64)
        self.truth_ids = _c(truth_ids, np.int32)
        self.n_vcf = int(self.n_records.shape[0])
        self.n_bins = int(n_bins)
        h = C.c_void_p()
        self.alleles = bool(alleles)
        check(self._L.qm_batch_create_ext(engine._h, self.n_vcf, _p(self.n_records), _p(self.truth_ids), self.n_bins,
                                          _lib.QM_BATCH_ALLELES if alleles else 0, C.byref(h)), engine._h)
        self._h = h
        engine._batches.add(self)

    def close(self):
        if getattr(self, "_h", None):
            self._L.qm_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        return check(rc, self.engine._h)

    def upload(self, v, pos, ref, alt, qual, flags):
        a = (_c(pos, np.int32), _c(ref, np.int32), _c(alt, np.int32), _c(qual, np.float32), _c(flags, np.uint8))
        if any(x.shape[0] != int(self.n_records[v]) for x in a):
            raise ValueError("column length != n_records[%d]" % v)
        self._ck(self._L.qm_batch_upload(self._h, int(v), *[_p(x) for x in a]))

    def synth(self, genome_len, truth_n, truth_seed, seed, shuffled=False, indel_pct=0):
        """truth_seed=None: every VCF is generated against the synthetic truth set it was assigned (per-VCF truth sets)"""
        ts = 0xffffffffffffffff if truth_seed is None else int(truth_seed)
        cfg = SynthCfg(int(genome_len), int(seed), ts, int(truth_n), int(shuffled), int(indel_pct))   # True = 1: permuted; R >= 2: R ascending runs
        self._ck(self._L.qm_batch_synth(self._h, C.byref(cfg)))

    def set_timing(self, on=True):
        self._ck(self._L.qm_batch_set_timing(self._h, int(on)))

    def run(self, stream=None, global_dev=None):
        """Enqueue classify -> finalize -> compact.  stream: raw hipStream_t (int) or None;
        global_dev: device pointer (int) of a [n_truth][3][n_bins] uint64 buffer or None."""
        self._ck(self._L.qm_batch_run(self._h, C.c_void_p(stream) if stream else None,
                                      C.c_void_p(global_dev) if global_dev else None))

    def finish(self, stream=None):
        self._ck(self._L.qm_batch_finish(self._h, C.c_void_p(stream) if stream else None))

    def timings(self):
        ms = (C.c_float * 4)()
        self._ck(self._L.qm_batch_timings(self._h, ms))
        return {"classify_ms": ms[0], "finalize_ms": ms[1], "compact_ms": ms[2], "total_ms": ms[3]}

    def cls(self, v):
        out = np.zeros(max(int(self.n_records[v]), 1), np.uint8)
        self._ck(self._L.qm_batch_get_cls(self._h, int(v), _p(out)))
        return out[:int(self.n_records[v])]

    def idx(self, v):
        out = np.zeros(max(int(self.n_records[v]), 1), np.int32)
        self._ck(self._L.qm_batch_get_idx(self._h, int(v), _p(out)))
        return out[:int(self.n_records[v])]

    def roc(self):
        out = np.zeros((self.n_vcf, 3, self.n_bins), np.uint64)
        self._ck(self._L.qm_batch_get_roc(self._h, _p(out)))
        return out

    def scalars(self):
        out = np.zeros((self.n_vcf, _lib.QM_N_SCALARS), np.int64)
        self._ck(self._L.qm_batch_get_scalars(self._h, _p(out)))
        return out

    @property
    def n_truth(self):
        """rows of the per-truth sums: truth-set slots of the context when the batch was created"""
        return int(self._L.qm_batch_n_truth(self._h))

    def global_counts(self):
        out = np.zeros((self.n_truth, 3, self.n_bins), np.uint64)
        self._ck(self._L.qm_batch_get_global(self._h, _p(out)))
        return out

    def columns(self, v):
        n = int(self.n_records[v])
        pos, ref, alt = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
        qual, flags = np.zeros(n, np.float32), np.zeros(n, np.uint8)
        self._ck(self._L.qm_batch_get_columns(self._h, int(v), _p(pos), _p(ref), _p(alt), _p(qual), _p(flags)))
        return pos, ref, alt, qual, flags

    def path_stats(self):
        """qm_batch_path_stats: where the VCFs the last finish found out of order went"""
        out = np.zeros(len(PATH_NAMES), np.int64)
        self._ck(self._L.qm_batch_path_stats(self._h, _p(out)))
        return dict(zip(PATH_NAMES, (int(x) for x in out)))

    @property
    def device_bytes(self):
        return int(self._L.qm_batch_device_bytes(self._h))
