"""ctypes binding of the CPU oracle (oracle/qm_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (quasimodo_amd/) never
imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libqm_oracle.so")


class _Buf(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_uint8)), ("len", C.c_size_t), ("cap", C.c_size_t)]


def build(force=False):
    src = os.path.join(_HERE, "qm_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libqm_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.qmo_extract_text.restype = C.c_int
        L.qmo_extract_text.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int, C.c_int,
                                       C.POINTER(_Buf), C.POINTER(_Buf), C.POINTER(_Buf), C.POINTER(C.c_int64)]
        L.qmo_buf_free.argtypes = [C.POINTER(_Buf)]
        L.qmo_count_text.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_int64)]
        L.qmo_fp_overlap_text.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.POINTER(C.c_int64)]
        L.qmo_awk_ge_int.argtypes = [C.c_char_p, C.c_size_t, C.c_int]
        L.qmo_caller_filter.argtypes = [C.c_char_p, C.c_size_t]
        L.qmo_qual_bin.argtypes = [C.c_float, C.c_int]
        vp = C.c_void_p
        L.qmo_classify_columns.argtypes = [C.c_int64, vp, vp, vp, vp, vp, C.c_int64, vp, vp, vp, C.c_int, vp, vp, vp]
        L.qmo_classify_columns_ext.argtypes = L.qmo_classify_columns.argtypes
        _lib = L
    return _lib


def is_pure_strain(vcf_path):
    """extract_TP_FP_SNPs.py:33 -- sample name ends in -1-0 / -0-1."""
    return os.path.basename(vcf_path).split(".")[0].endswith(("-1-0", "-0-1"))


def extract_text(vcf: bytes, truth: bytes, custom=False, pure_strain=False):
    """Returns (filtered, tp_or_None, fp, stats) as the reference writes them."""
    L = lib()
    f, t, p = _Buf(), _Buf(), _Buf()
    st = (C.c_int64 * 6)()
    rc = L.qmo_extract_text(vcf, len(vcf), truth, len(truth), int(custom), int(pure_strain),
                            C.byref(f), C.byref(t), C.byref(p), st)
    out = []
    for b in (f, t, p):
        out.append(C.string_at(b.data, b.len) if b.len else b"")
        L.qmo_buf_free(C.byref(b))
    if rc == -2:
        raise ValueError("kept data line with NUL or non-ASCII bytes: reference behaviour is locale dependent")
    stats = dict(zip(("records", "header_lines", "kept", "tp_lines", "fp_lines", "patterns"), list(st)))
    return out[0], (None if pure_strain else out[1]), out[2], stats


def count_text(filtered: bytes, truth: bytes, custom=False):
    L = lib()
    o = (C.c_int64 * 6)()
    L.qmo_count_text(filtered, len(filtered), truth, len(truth), int(custom), o)
    return dict(zip(("genomediff", "calleridentify", "TP", "FP", "FN", "truth_unique"), list(o)))


def fp_overlap_text(texts):
    L = lib()
    n = len(texts)
    arr = (C.c_char_p * n)(*texts)
    lens = (C.c_size_t * n)(*[len(t) for t in texts])
    reg = (C.c_int64 * (1 << n))()
    rc = L.qmo_fp_overlap_text(n, arr, lens, reg)
    if rc:
        raise ValueError("qmo_fp_overlap_text rc=%d" % rc)
    return list(reg)


def awk_ge(field: bytes, thr=20):
    return bool(lib().qmo_awk_ge_int(field, len(field), thr))


def caller_filter(line: bytes):
    return bool(lib().qmo_caller_filter(line, len(line)))


def classify_columns(pos, ref, alt, qual, flags, tpos, tref, talt, n_bins=256, ext=False):
    """ext: allele-extended mode (any valid allele code takes part, not only 0..3)."""
    L = lib()
    pos = np.ascontiguousarray(pos, np.int32); ref = np.ascontiguousarray(ref, np.int32)
    alt = np.ascontiguousarray(alt, np.int32); qual = np.ascontiguousarray(qual, np.float32)
    flags = np.ascontiguousarray(flags, np.uint8)
    tpos = np.ascontiguousarray(tpos, np.int32); tref = np.ascontiguousarray(tref, np.int32)
    talt = np.ascontiguousarray(talt, np.int32)
    n = pos.shape[0]
    cls = np.zeros(n, np.uint8)
    roc = np.zeros((3, n_bins), np.uint64)
    sc = np.zeros(8, np.int64)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = (L.qmo_classify_columns_ext if ext else L.qmo_classify_columns)(n, p(pos), p(ref), p(alt), p(qual), p(flags), tpos.shape[0], p(tpos), p(tref), p(talt),
                                n_bins, p(cls), p(roc), p(sc))
    if rc:
        raise ValueError("qmo_classify_columns rc=%d" % rc)
    scal = dict(zip(("n_pass", "tp_lines", "fp_lines", "TP_R", "FP_R", "sorted", "truth_unique"), sc[:7].tolist()))
    return cls, roc, scal
