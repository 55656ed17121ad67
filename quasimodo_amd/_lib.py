"""ctypes loader for libqmvt.so (built in-tree by quasimodo_amd/csrc/Makefile)."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_SO = os.environ.get("QM_LIBQMVT") or os.path.join(_CSRC, "libqmvt.so")   # override only for A/B builds of the kernels

QM_N_SCALARS = 8
SCALAR_NAMES = ("n_pass", "tp_lines", "fp_lines", "TP_R", "FP_R", "sorted", "n_records", "truth_unique")
ERRORS = {-1: "QM_E_INVAL", -2: "QM_E_NODEVICE", -3: "QM_E_HIP", -4: "QM_E_NOMEM", -5: "QM_E_RANGE",
          -6: "QM_E_STATE", -7: "QM_E_IO", -8: "QM_E_NONCANON", -9: "QM_E_LIMIT", -10: "QM_E_UNSORTED", -11: "QM_E_COMM"}
QM_E_UNSORTED = -10
QM_BATCH_ALLELES = 1
QM_ABI_VERSION = 6

# every symbol include/qmvt.h declares
EXPORTS = (
    "qm_abi_version", "qm_kernels_id", "qm_build_id", "qm_init", "qm_destroy", "qm_last_error", "qm_truth_load", "qm_truth_size", "qm_truth_count",
    "qm_classify_batch", "qm_batch_create", "qm_batch_destroy", "qm_batch_upload", "qm_truth_synth", "qm_batch_synth",
    "qm_batch_run", "qm_batch_finish", "qm_batch_set_timing", "qm_batch_timings", "qm_batch_get_cls", "qm_batch_get_idx",
    "qm_batch_get_roc", "qm_batch_get_scalars", "qm_batch_get_global", "qm_batch_get_columns", "qm_batch_device_bytes",
    "qm_fp_overlap", "qm_vcf_count_lines", "qm_vcf_scan", "qm_truth_scan", "qm_vcf_write", "qm_vcf_split_write",
    "qm_truth_size_ext", "qm_truth_synth_ext", "qm_batch_create_ext", "qm_classify_batch_ext",
    "qm_dict_create", "qm_dict_destroy", "qm_dict_size", "qm_allele_code", "qm_allele_spell", "qm_vcf_scan_ext", "qm_truth_scan_ext",
    "qm_bench_synth", "qm_truth_release", "qm_batch_n_truth",
    "qm_bw_probe", "qm_bgzf_write", "qm_bgzf_write_tbi", "qm_extract_files", "qm_extract_files_ex", "qm_batch_global_device", "qm_batch_path_stats", "qm_path_stats_total", "qm_batch_upload_async", "qm_batch_get_masks", "qm_patterns_create", "qm_patterns_destroy", "qm_patterns_info", "qm_vcf_hostpath",
    "qm_mummer2vcf", "qm_free",
    "qm_comm_create", "qm_comm_make_id", "qm_comm_create_rank", "qm_allreduce_counters", "qm_comm_collectives", "qm_comm_destroy",
)


class BenchResult(C.Structure):
    _fields_ = [("records", C.c_int64), ("seconds_per_step", C.c_double), ("classifications_per_s", C.c_double),
                ("classify_ms", C.c_float), ("finalize_ms", C.c_float), ("compact_ms", C.c_float), ("reserved", C.c_int32),
                ("kept", C.c_int64), ("tp_lines", C.c_int64), ("fp_lines", C.c_int64), ("device_bytes", C.c_int64)]


class QmvtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("%s (%d): %s" % (ERRORS.get(code, "QM_E_?"), code, message))
        self.code = code


class SynthCfg(C.Structure):
    _fields_ = [("genome_len", C.c_int64), ("seed", C.c_uint64), ("truth_seed", C.c_uint64), ("truth_n", C.c_int64),
                ("shuffled", C.c_int32), ("indel_pct", C.c_int32)]


class FileJob(C.Structure):
    _fields_ = [("vcf_path", C.c_char_p), ("truth_path", C.c_char_p), ("mode", C.c_int32), ("pure", C.c_int32),
                ("filtered_out", C.c_char_p), ("tp_out", C.c_char_p), ("fp_out", C.c_char_p)]


class FileStats(C.Structure):
    _fields_ = [("scalars", C.c_int64 * QM_N_SCALARS), ("n_lines", C.c_int64), ("n_refused", C.c_int64), ("genomediff", C.c_int64),
                ("header_kept", C.c_int64), ("header_kept_tp", C.c_int64), ("host_decided", C.c_int64), ("r_hostile", C.c_int64)]


class VcfCols(C.Structure):
    _fields_ = [("n_lines", C.c_int64), ("n_data", C.c_int64), ("n_host", C.c_int64), ("n_refused", C.c_int64),
                ("first_refused_line", C.c_int64), ("n_nokey_kept", C.c_int64), ("n_r_hostile", C.c_int64), ("first_r_hostile_line", C.c_int64)]


def library_path():
    return _SO


_KSRC = ("qmvt_kernels.hip", "qmvt_dev.h")
_ASRC = _KSRC + ("qmvt_api.cpp", "qmvt_host.cpp", "qmvt_pipeline.cpp", os.path.join("..", "..", "include", "qmvt.h"), "Makefile")


def _sha16(files):
    import hashlib
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(_CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def source_kernels_id():
    """sha256 (first 16 hex digits) of the device code's SOURCES in the tree (the Makefile compiles the same figure in)."""
    return _sha16(_KSRC)


def source_build_id():
    """The same over every source of the library."""
    return _sha16(_ASRC)


def embedded_ids(path=None):
    """(kernels id, build id) compiled into a libqmvt.so, read from the file without loading it; (None, None) when the file
    or the marker is missing."""
    import re
    try:
        with open(path or _SO, "rb") as fh:
            m = re.search(rb"@\(#\)qmvt-ids kernels=([0-9a-z]+) build=([0-9a-z]+);", fh.read())
    except OSError:
        return None, None
    return (m.group(1).decode(), m.group(2).decode()) if m else (None, None)


def kernel_source_id():
    """The id of the device code of the LOADED library (qm_kernels_id: compiled in at build time): what a profile of the
    kernels belongs to.  profiles/traffic.json carries the id of the build its PMC passes ran on; bench.py quotes that
    traffic only for the same id.  "unknown" for builds that did not go through the Makefile (QM_LIBQMVT A/B builds)."""
    L = lib()
    return L.qm_kernels_id().decode()


def build_library(force=False):
    """Compile libqmvt.so for gfx950 with hipcc (cross-compiles without a GPU).  Rebuilt whenever the ids compiled into the
    binary differ from the sources in the tree (content, not time stamps: a stale binary that travelled with the tree is
    replaced), when it is missing, or on request."""
    default = os.path.join(_CSRC, "libqmvt.so")
    stale = force or not os.path.exists(default) or embedded_ids(default) != (source_kernels_id(), source_build_id())
    if stale:
        if _lib is not None and _SO == default:
            raise QmvtError(-6, "libqmvt.so is stale but already loaded in this process; rebuild before importing the engine")
        subprocess.check_call(["make", "-s", "-C", _CSRC, "libqmvt.so"])
        if embedded_ids(default) != (source_kernels_id(), source_build_id()):
            raise QmvtError(-7, "libqmvt.so does not carry the ids of the sources it was just built from")
    return _SO


_lib = None


def lib():
    """The loaded library.  Missing .so is a hard error: there is no Python/CPU substitute."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise QmvtError(-2, "libqmvt.so is not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "or `make -C quasimodo_amd/csrc`" % _SO)
    L = C.CDLL(_SO)
    vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int
    L.qm_abi_version.restype = i32
    L.qm_kernels_id.restype = C.c_char_p
    L.qm_build_id.restype = C.c_char_p
    L.qm_last_error.restype = C.c_char_p
    L.qm_last_error.argtypes = [vp]
    L.qm_init.argtypes = [i32, C.POINTER(vp)]
    L.qm_destroy.argtypes = [vp]
    L.qm_destroy.restype = None
    L.qm_truth_load.argtypes = [vp, vp, vp, vp, i64, C.POINTER(i32)]
    L.qm_truth_synth.argtypes = [vp, i64, i64, C.c_uint64, C.POINTER(i32)]
    L.qm_truth_size.argtypes = [vp, i32, C.POINTER(i64)]
    L.qm_truth_size_ext.argtypes = [vp, i32, C.POINTER(i64)]
    L.qm_truth_synth_ext.argtypes = [vp, i64, i64, C.c_uint64, i32, C.POINTER(i32)]
    L.qm_batch_create_ext.argtypes = [vp, i32, vp, vp, i32, C.c_uint, C.POINTER(vp)]
    L.qm_classify_batch_ext.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, C.c_uint, vp, vp, vp, vp, vp]
    L.qm_truth_count.argtypes = [vp]
    L.qm_truth_release.argtypes = [vp, i32]
    L.qm_batch_n_truth.argtypes = [vp]
    L.qm_classify_batch.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp]
    L.qm_batch_create.argtypes = [vp, i32, vp, vp, i32, C.POINTER(vp)]
    L.qm_batch_destroy.argtypes = [vp]
    L.qm_batch_destroy.restype = None
    L.qm_batch_upload.argtypes = [vp, i32, vp, vp, vp, vp, vp]
    L.qm_batch_synth.argtypes = [vp, C.POINTER(SynthCfg)]
    L.qm_bench_synth.argtypes = [vp, C.POINTER(SynthCfg), i32, i64, i32, i32, C.POINTER(BenchResult)]
    L.qm_batch_run.argtypes = [vp, vp, vp]
    L.qm_batch_finish.argtypes = [vp, vp]
    L.qm_batch_set_timing.argtypes = [vp, i32]
    L.qm_batch_timings.argtypes = [vp, C.POINTER(C.c_float)]
    L.qm_batch_get_cls.argtypes = [vp, i32, vp]
    L.qm_batch_get_idx.argtypes = [vp, i32, vp]
    L.qm_batch_get_roc.argtypes = [vp, vp]
    L.qm_batch_get_scalars.argtypes = [vp, vp]
    L.qm_batch_get_global.argtypes = [vp, vp]
    L.qm_batch_get_columns.argtypes = [vp, i32, vp, vp, vp, vp, vp]
    L.qm_batch_device_bytes.argtypes = [vp]
    L.qm_batch_device_bytes.restype = i64
    L.qm_fp_overlap.argtypes = [vp, i32, vp, vp, vp, vp, vp]
    L.qm_vcf_count_lines.argtypes = [C.c_char_p, C.c_size_t]
    L.qm_vcf_count_lines.restype = i64
    L.qm_vcf_scan.argtypes = [C.c_char_p, C.c_size_t, i64, vp, vp, vp, vp, vp, vp, vp, C.POINTER(VcfCols)]
    L.qm_truth_scan.argtypes = [C.c_char_p, C.c_size_t, i32, i64, vp, vp, vp, vp]
    L.qm_truth_scan.restype = i64
    L.qm_vcf_scan_ext.argtypes = [C.c_char_p, C.c_size_t, i64, vp, vp, vp, vp, vp, vp, vp, C.POINTER(VcfCols), vp]
    L.qm_truth_scan_ext.argtypes = [C.c_char_p, C.c_size_t, i32, i64, vp, vp, vp, vp, vp]
    L.qm_truth_scan_ext.restype = i64
    L.qm_dict_create.argtypes = []
    L.qm_dict_create.restype = vp
    L.qm_dict_destroy.argtypes = [vp]
    L.qm_dict_destroy.restype = None
    L.qm_dict_size.argtypes = [vp]
    L.qm_dict_size.restype = i64
    L.qm_allele_code.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.qm_allele_code.restype = i32
    L.qm_allele_spell.argtypes = [vp, i32, C.c_char_p, C.c_size_t]
    L.qm_allele_spell.restype = i64
    L.qm_extract_files.argtypes = [vp, i32, C.POINTER(FileJob), i32, C.c_uint, i32, C.POINTER(FileStats), vp, C.POINTER(C.c_double)]
    L.qm_extract_files_ex.argtypes = [vp, i32, C.POINTER(FileJob), i32, C.c_uint, i32, C.POINTER(FileStats), vp, C.POINTER(C.c_double), vp, i32, vp]
    L.qm_batch_global_device.argtypes = [vp, C.POINTER(vp)]
    L.qm_batch_path_stats.argtypes = [vp, vp]
    L.qm_path_stats_total.argtypes = [vp, vp]
    L.qm_mummer2vcf.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p, C.c_uint, C.c_char_p, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.qm_free.argtypes = [vp]
    L.qm_free.restype = None
    L.qm_comm_create.argtypes = [C.POINTER(vp), i32, C.POINTER(vp)]
    L.qm_comm_make_id.argtypes = [vp]
    L.qm_comm_create_rank.argtypes = [vp, i32, i32, vp, C.POINTER(vp)]
    L.qm_allreduce_counters.argtypes = [vp, vp, vp]
    L.qm_comm_collectives.argtypes = [vp, vp]
    L.qm_comm_collectives.restype = i64
    L.qm_comm_destroy.argtypes = [vp]
    L.qm_comm_destroy.restype = None
    L.qm_batch_upload_async.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp]
    L.qm_batch_get_masks.argtypes = [vp, i32, vp, vp]
    L.qm_bgzf_write.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, i32]
    L.qm_bgzf_write_tbi.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, i32]
    L.qm_bw_probe.argtypes = [vp, i64, i32, C.POINTER(C.c_double)]
    L.qm_patterns_create.argtypes = [C.c_char_p, C.c_size_t, i32, i32]
    L.qm_patterns_create.restype = vp
    L.qm_patterns_destroy.argtypes = [vp]
    L.qm_patterns_destroy.restype = None
    L.qm_patterns_info.argtypes = [vp, vp]
    L.qm_vcf_hostpath.argtypes = [vp, C.c_char_p, C.c_size_t, i64, vp, vp, vp, vp, vp, vp, vp]
    L.qm_vcf_write.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, i64, vp, vp, vp, i32]
    L.qm_vcf_split_write.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, i32, i32, C.POINTER(C.c_int64)]
    _lib = L
    return L


def check(rc, ctx=None):
    if rc < 0:
        raise QmvtError(rc, lib().qm_last_error(ctx).decode("utf-8", "replace"))
    return rc
