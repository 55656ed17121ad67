"""Host text side: VCF / truth tokenizing into packed columns and the output writers.
All heavy lifting is in libqmvt.so (quasimodo_amd/csrc/qmvt_host.cpp)."""
import ctypes as C
import os
import sys
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import QM_E_UNSORTED, QmvtError, VcfCols, check


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


@dataclass
class ScannedVcf:
    """One VCF text scanned once (replaces the three awk passes + grep of
    extract_TP_FP_SNPs.py:24-32,50,52)."""
    text: bytes
    n_lines: int
    line_off: np.ndarray   # int64 [n_lines + 1]
    line_kind: np.ndarray  # uint8 [n_lines]: QM_LINE_* of include/qmvt.h (0 data, 1 header, 2 data for the host path,
                           # 3 / 4 header line that passes the A2 filter (4: selected by fgrep), 5 / 6 refused: NUL or invalid UTF-8)
    pos: np.ndarray
    ref: np.ndarray
    alt: np.ndarray
    qual: np.ndarray
    flags: np.ndarray
    n_host: int            # lines whose fgrep answer needs the text of the patterns (hostpath)
    n_refused: int
    first_refused_line: int
    n_nokey_kept: int      # kept lines without a comparable key (POS not a canonical decimal)
    n_r_hostile: int = 0   # kept lines holding '#', ' or ": R's read.table does not read them as tab-split text (include/qmvt.h)
    first_r_hostile_line: int = 0

    @property
    def n_records(self):
        return int(self.pos.shape[0])

    @property
    def columns(self):
        return self.pos, self.ref, self.alt, self.qual, self.flags

    def hostpath(self, patterns):
        """decide the lines the columns cannot describe with the exact `fgrep -w` of the reference
        (extract_TP_FP_SNPs.py:50-53) and write the decisions into the flags column / line kinds.  Call before the
        columns are uploaded.  Returns the exchange for R's unique-key counts (qm_vcf_hostpath, include/qmvt.h)."""
        out = np.zeros(5, np.int64)
        rc = _lib.lib().qm_vcf_hostpath(patterns._h, self.text, len(self.text), self.n_lines, _p(self.line_off), _p(self.line_kind),
                                        _p(self.pos), _p(self.ref), _p(self.alt), _p(self.flags) if self.n_records else _p(np.zeros(1, np.uint8)),
                                        _p(out))
        if rc < 0:
            raise QmvtError(rc, "qm_vcf_hostpath failed")
        return dict(decided=int(out[0]), selected=int(out[1]), device_nokey_keys=int(out[2]), tp_r=int(out[3]), fp_r=int(out[4]))

    @property
    def header_kept(self):
        """'#' lines that pass the A2 filter: (all, those fgrep selects)"""
        return int(np.count_nonzero((self.line_kind == 3) | (self.line_kind == 4))), int(np.count_nonzero(self.line_kind == 4))

    def write(self, path, cls, select):
        """select: 0 = kept lines (filtered.vcf), 1 = TP lines, 2 = FP lines; header lines always first."""
        cls = np.ascontiguousarray(cls, np.uint8)
        if cls.shape[0] != self.n_records:
            raise ValueError("cls length %d != records %d" % (cls.shape[0], self.n_records))
        rc = _lib.lib().qm_vcf_write(os.fsencode(path), self.text, len(self.text), self.n_lines, _p(self.line_off),
                                     _p(self.line_kind), _p(cls) if self.n_records else None, int(select))
        if rc < 0:
            raise QmvtError(rc, "cannot write %s" % path)


class AlleleDict:
    """Interned long alleles of the allele-extended mode (qm_dict): one per engine run, shared by
    the truth sets and every VCF classified against them."""

    def __init__(self):
        self._L = _lib.lib()
        self._h = self._L.qm_dict_create()

    def close(self):
        if getattr(self, "_h", None):
            self._L.qm_dict_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return int(self._L.qm_dict_size(self._h))

    def code(self, allele: bytes) -> int:
        return int(self._L.qm_allele_code(self._h, allele, len(allele)))

    def spell(self, code: int) -> bytes:
        buf = C.create_string_buffer(1 << 16)
        n = int(self._L.qm_allele_spell(self._h, int(code), buf, len(buf)))
        if n < 0:
            raise ValueError("not an allele code: %d" % code)
        return buf.raw[:n]


def scan_vcf(text: bytes, alleles: "AlleleDict | None" = None) -> ScannedVcf:
    """alleles: an AlleleDict switches to allele-extended tokenising (any [ACGT]+ REF / ALT takes part)."""
    L = _lib.lib()
    nl = int(L.qm_vcf_count_lines(text, len(text)))
    cap = nl + 1
    line_off = np.empty(cap + 1, np.int64)
    kind = np.empty(cap, np.uint8)
    pos, ref, alt = np.empty(cap, np.int32), np.empty(cap, np.int32), np.empty(cap, np.int32)
    qual, flags = np.empty(cap, np.float32), np.empty(cap, np.uint8)
    info = VcfCols()
    rc = L.qm_vcf_scan_ext(text, len(text), cap, _p(line_off), _p(kind), _p(pos), _p(ref), _p(alt), _p(qual), _p(flags),
                           C.byref(info), alleles._h if alleles is not None else None)
    if rc < 0:
        raise QmvtError(rc, "qm_vcf_scan failed")
    n, d = int(info.n_lines), int(info.n_data)
    # views, not copies: the spare tail of each buffer is a line or two
    return ScannedVcf(text, n, line_off[:n + 1], kind[:n], pos[:d], ref[:d], alt[:d], qual[:d], flags[:d],
                      int(info.n_host), int(info.n_refused), int(info.first_refused_line), int(info.n_nokey_kept),
                      int(info.n_r_hostile), int(info.first_r_hostile_line))


class Patterns:
    """The pattern list the reference feeds to `fgrep -wf`, as text (qm_patterns): what the host path matches
    against.  custom=False: truth VCF (extract_TP_FP_SNPs.py:47); custom=True: show-snps table (:92)."""

    def __init__(self, truth_text: bytes, custom: bool = False, alleles: bool = False):
        self._L = _lib.lib()
        self._h = self._L.qm_patterns_create(truth_text, len(truth_text), int(bool(custom)), int(bool(alleles)))
        if not self._h:
            raise QmvtError(-1, "qm_patterns_create failed")
        info = np.zeros(4, np.int64)
        check(self._L.qm_patterns_info(self._h, _p(info)))
        self.n_patterns, self.n_exotic, self.n_comment_only, self.n_refused = (int(x) for x in info)

    @property
    def needs_full_hostpath(self):
        """patterns no column key can stand for: every line compared with this truth set is decided from the text"""
        return self.n_exotic > 0 or self.n_comment_only > 0

    def close(self):
        if getattr(self, "_h", None):
            self._L.qm_patterns_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


@dataclass
class TruthKeys:
    pos: np.ndarray
    ref: np.ndarray
    alt: np.ndarray
    genomediff: int   # rows R counts as `genomediff` (caller_performance_compare.R:90)
    n_never: int      # rows whose pattern can never match a line through its columns (they live in Patterns only)
    n_refused: int    # rows whose pattern holds a NUL or invalid UTF-8 (strict mode stops)
    n_comment: int    # '#' rows awk turns into patterns all the same (Patterns keeps them; R does not read them)


def scan_truth(text: bytes, custom: bool = False, alleles: "AlleleDict | None" = None) -> TruthKeys:
    """Truth text -> (pos, ref, alt) key columns.  custom=False: VCF written by
    mummer2vcf.py (extract_TP_FP_SNPs.py:47); custom=True: show-snps TSV (:92).
    alleles: allele-extended keys (VCF truth only)."""
    if alleles is not None and custom:
        raise ValueError("allele-extended truth sets come from VCF truth files (show-snps tables spell gaps as '.')")
    L = _lib.lib()
    cap = int(L.qm_vcf_count_lines(text, len(text))) + 1
    pos, ref, alt = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap, np.int32)
    counts = np.zeros(5, np.int64)
    n = int(L.qm_truth_scan_ext(text, len(text), int(bool(custom)), cap, _p(pos), _p(ref), _p(alt), _p(counts),
                                alleles._h if alleles is not None else None))
    if n < 0:
        raise QmvtError(n, "qm_truth_scan failed")
    return TruthKeys(pos[:n].copy(), ref[:n].copy(), alt[:n].copy(), int(counts[0]), int(counts[2]), int(counts[3]), int(counts[4]))


AWK_POSIX, AWK_MAWK_LITERAL = 0, 1


def awk_flavour_default() -> int:
    """`{2,}` in the xindel rule: POSIX interval unless QM_AWK_FLAVOUR=mawk-literal
    (mawk 1.3.4 20200120 reads the braces as text)."""
    v = os.environ.get("QM_AWK_FLAVOUR", "posix")
    if v not in ("posix", "mawk-literal"):
        raise ValueError("QM_AWK_FLAVOUR must be posix or mawk-literal, not %r" % v)
    return AWK_MAWK_LITERAL if v == "mawk-literal" else AWK_POSIX


def bgzip(src_path, dst_path=None, level=-1, tbi=False) -> str:
    """`bgzip -c src > dst` (rules/vis_eval_vcf.smk:36,51,67,82) without the tool: BGZF members + EOF block.
    tbi=True: also `tabix -p vcf dst` (:37,52,68,83) -> dst + ".tbi"; a VCF whose sequences do not come in blocks or whose
    positions step backwards raises QmvtError(QM_E_UNSORTED) and nothing is written -- tabix stops on such a file too.
    tbi="if-sorted": such a VCF gets its .gz and no index (a note goes to stderr)."""
    dst_path = dst_path or src_path + ".gz"
    with open(src_path, "rb") as fh:
        data = fh.read()
    L = _lib.lib()
    if tbi:
        rc = L.qm_bgzf_write_tbi(os.fsencode(dst_path), data, len(data), int(level))
        if rc == QM_E_UNSORTED and tbi == "if-sorted":
            sys.stderr.write("%s: not sorted by sequence block and position -- no .tbi (tabix refuses it too)\n" % dst_path)
            if os.path.exists(dst_path + ".tbi"):
                os.remove(dst_path + ".tbi")      # an index of an earlier run would describe other bytes
            rc = L.qm_bgzf_write(os.fsencode(dst_path), data, len(data), int(level))
        elif rc == QM_E_UNSORTED:
            raise QmvtError(rc, "%s: sequences not in blocks or positions stepping backwards: no tabix index can describe it" % src_path)
    else:
        rc = L.qm_bgzf_write(os.fsencode(dst_path), data, len(data), int(level))
    if rc < 0:
        raise QmvtError(rc, "cannot write %s" % dst_path)
    return dst_path


def split_variants(vcf_path, out_path, kind, flavour=None, bgz=False, tbi=False) -> int:
    """`extract_snp` / `extract_indel` / `extract_nucmer_*` (rules/vis_eval_vcf.smk:25-86) without
    awk: kind "xsnp" or "xindel"; returns the number of lines written.  bgz=True also writes the rules' second
    declared output, <out_path>.gz (BGZF, what `bgzip -c` makes of it); tbi (with bgz): and its tabix index, see `bgzip`."""
    if kind not in ("xsnp", "xindel"):
        raise ValueError("kind must be xsnp or xindel")
    with open(vcf_path, "rb") as fh:
        text = fh.read()
    n = C.c_int64(0)
    rc = _lib.lib().qm_vcf_split_write(os.fsencode(out_path), text, len(text), 0 if kind == "xsnp" else 1,
                                       awk_flavour_default() if flavour is None else int(flavour), C.byref(n))
    if rc < 0:
        raise QmvtError(rc, "cannot write %s" % out_path)
    if bgz:
        bgzip(out_path, tbi=tbi)
    return int(n.value)
