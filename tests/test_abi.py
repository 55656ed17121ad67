"""The C-ABI library loads, exports every symbol include/qmvt.h declares, and fails
loudly (never silently falls back) when there is no HIP device.  CPU only."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "qmvt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(qm_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_exported(qmlib):
    from quasimodo_amd import _lib
    names = _declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(qmlib, n), "libqmvt.so does not export %s" % n
    assert sorted(_lib.EXPORTS) == names
    assert qmlib.qm_abi_version() == 1


def test_no_cpu_fallback_without_device(qmlib):
    """On a box without a GPU qm_init must fail with QM_E_NODEVICE; on a GPU box it succeeds."""
    import torch
    import quasimodo_amd as q
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu tests")
    with pytest.raises(q.QmvtError) as ei:
        q.Engine(0)
    assert ei.value.code == -2
    assert "no CPU fallback" in str(ei.value)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under quasimodo_amd/ or program/ may reference it."""
    bad = []
    for top in ("quasimodo_amd", "program", "run_benchmark.py"):
        p = os.path.join(ROOT, top)
        files = [p] if os.path.isfile(p) else [os.path.join(d, f) for d, _, fs in os.walk(p) for f in fs
                                               if f.endswith((".py", ".cpp", ".hip", ".h"))]
        for f in files:
            if not os.path.exists(f):
                continue
            src = open(f, errors="replace").read()
            if re.search(r"\boracle\b|qm_oracle|libqm_oracle", src):
                bad.append(f)
    assert not bad, bad


def test_null_arguments_are_rejected(qmlib):
    assert qmlib.qm_init(0, None) == -1
    assert qmlib.qm_truth_load(None, None, None, None, 0, None) == -1
    assert qmlib.qm_batch_run(None, None, None) == -1
    assert b"NULL" in qmlib.qm_last_error(None)
