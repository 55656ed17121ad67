"""quasimodo_amd -- MI355X-native variant-truth comparison engine.

Drop-in for the TP/FP classification path of hzi-bifo/Quasimodo
(program/extract_TP_FP_SNPs.py + rules/extract_TP.smk + rules/compare_FP.smk):
a Python host over a ctypes C ABI (include/qmvt.h) into libqmvt.so, whose HIP
kernels run on gfx950.  There is no CPU fallback for the classification.
"""
from ._lib import QmvtError, build_library, kernel_source_id, library_path  # noqa: F401
from .engine import Batch, Engine  # noqa: F401
from .vcfio import ScannedVcf, TruthKeys, scan_truth, scan_vcf  # noqa: F401
from .extract import (  # noqa: F401
    extract_many,
    extract_tp_fp_custom_snp,
    extract_tp_fp_snp,
    is_pure_strain,
)

from . import tables, workflow  # noqa: F401,E402

__version__ = "0.1.0"
