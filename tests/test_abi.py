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
    assert qmlib.qm_abi_version() == 6


def test_the_library_says_which_sources_it_was_built_from(qmlib, tmp_path):
    """qm_kernels_id / qm_build_id = the sha256 ids of the sources in the tree, taken by the Makefile at build time: what
    `build()` compares (content, not time stamps) and what a PMC profile is keyed on (a stale binary could
    carry a fresh id).  The ids are also readable from the file without loading it; a file without them is never 'up to date'."""
    from quasimodo_amd import _lib
    qmlib.qm_kernels_id.restype = C.c_char_p
    qmlib.qm_build_id.restype = C.c_char_p
    assert qmlib.qm_kernels_id().decode() == _lib.source_kernels_id() == _lib.kernel_source_id()
    assert qmlib.qm_build_id().decode() == _lib.source_build_id()
    assert _lib.embedded_ids() == (_lib.source_kernels_id(), _lib.source_build_id())
    other = tmp_path / "libqmvt.so"
    other.write_bytes(b"\x7fELF nothing of ours")
    assert _lib.embedded_ids(str(other)) == (None, None) and _lib.embedded_ids(str(tmp_path / "missing.so")) == (None, None)


def test_caller_allocated_structs_have_the_headers_size_everywhere(qmlib):
    """qm_vcf_cols / qm_file_stats / qm_file_job / qm_bench_result are allocated by the CALLER and filled by the library: a binding
    that declares fewer fields than the header lets the library write past its allocation (round 4: the stub of INTEGRATION.md was
    two fields short of a grown qm_vcf_cols for a few hours -- a heap corruption that showed as a crash at interpreter exit).
    The header's field counts against the package's ctypes structures and against the stub in INTEGRATION.md."""
    from quasimodo_amd import _lib
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "qmvt.h")).read(), flags=re.S)

    def c_size(name):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), text, re.S).group(1)
        size = 0
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            m = re.match(r"(const char\*|int64_t|uint64_t|int32_t|uint32_t|int|double|float)\s+(.*)$", decl, re.S)
            assert m, decl
            width = {"const char*": 8, "int64_t": 8, "uint64_t": 8, "double": 8, "int32_t": 4, "uint32_t": 4, "int": 4, "float": 4}[m.group(1)]
            for var in m.group(2).split(","):
                arr = re.search(r"\[(\w+)\]", var)
                n = 1 if not arr else (int(arr.group(1)) if arr.group(1).isdigit() else {"QM_N_SCALARS": 8}[arr.group(1)])
                size += width * n
        return size

    assert C.sizeof(_lib.VcfCols) == c_size("qm_vcf_cols") == 64
    assert C.sizeof(_lib.FileStats) == c_size("qm_file_stats") == 120
    assert C.sizeof(_lib.SynthCfg) == c_size("qm_synth_cfg")
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    stub = md[md.index("## The binding itself"):]
    fields = re.search(r"class Info\(C\.Structure\): _fields_ = \[\(k, C\.c_int64\) for k in \((.*?)\)\]", stub, re.S).group(1)
    assert 8 * len(re.findall(r'"\w+"', fields)) == c_size("qm_vcf_cols")
    assert "qm_abi_version() == %d" % _lib.QM_ABI_VERSION in stub


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


def _build_c_example(tmp_path, name="qm_bench"):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / name)
    libdir = os.path.join(root, "quasimodo_amd", "csrc")
    subprocess.run(["gcc", "-O2", "-Wall", "-Werror", "-pthread", "-I" + os.path.join(root, "include"), "-o", exe,
                    os.path.join(root, "examples", name + ".c"), "-L" + libdir, "-lqmvt", "-Wl,-rpath," + libdir], check=True)
    return exe


def test_c_program_links_against_the_abi_and_fails_loudly_without_a_gpu(qmlib, tmp_path):
    """include/qmvt.h is usable from plain C; without a HIP device the engine refuses (no CPU path)"""
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by the gpu test")
    exe = _build_c_example(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "no CPU fallback" in r.stderr and r.stdout == ""


@pytest.mark.gpu
def test_c_program_runs_the_synthetic_workload(qmlib, tmp_path):
    import json
    import subprocess
    exe = _build_c_example(tmp_path)
    r = subprocess.run([exe, "8", "1000000", "3"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout)
    assert d["vcfs"] == 8 and d["kept"] == d["tp_lines"] + d["fp_lines"] and d["tp_lines"] > 0 and d["classifications_per_s"] > 1e9
    r2 = subprocess.run([exe, "4", "1000000", "2", "1"], capture_output=True, text=True)      # shuffled: the radix-sort path
    d2 = json.loads(r2.stdout)
    assert r2.returncode == 0 and d2["kept"] > 0


def test_multi_device_c_program_fails_loudly_without_a_gpu(qmlib, tmp_path):
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by the gpu test")
    exe = _build_c_example(tmp_path, "qm_multi")
    r = subprocess.run([exe, "0,0", "4", "1000", "1"], capture_output=True, text=True)
    assert r.returncode == 2 and "no CPU fallback" in r.stderr and r.stdout == ""


@pytest.mark.gpu
def test_multi_device_c_program_shards_sum_to_the_single_batch(qmlib, tmp_path):
    """examples/qm_multi.c: one host thread and one context per device (here the same card twice), contiguous VCF shards, the
    counters summed on the host -- identical to the same VCFs in one batch."""
    import json
    import subprocess
    exe = _build_c_example(tmp_path, "qm_multi")
    r = subprocess.run([exe, "0,0", "10", "1000000", "2", "1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout)
    assert d["devices"] == 2 and d["vcfs"] == 10 and d["equals_one_batch"] is True and d["exchange"] == "host-sum"   # RCCL refuses two ranks on one card
    assert d["kept"] == d["tp_lines"] + d["fp_lines"] and d["roc_tp_at_20"] == d["tp_lines"]
    # one device: the exchange is the library's own collective over RCCL (a communicator of one member), one per step
    r = subprocess.run([exe, "0", "6", "1000000", "3", "1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout.strip().splitlines()[-1])          # (RCCL prints its version banner on stdout when a communicator is made)
    assert d["devices"] == 1 and d["exchange"].startswith("rccl all-reduce") and d["equals_one_batch"] is True and d["roc_tp_at_20"] == d["tp_lines"]


def test_the_collective_entry_points_refuse_bad_arguments_without_touching_a_device(qmlib):
    """qm_comm_* / qm_allreduce_counters (include/qmvt.h: the path's one exchange behind the C ABI): argument checks come first --
    no RCCL, no GPU is needed to be told so"""
    import ctypes as C
    out = C.c_void_p()
    assert qmlib.qm_comm_create(None, 1, C.byref(out)) == -1 and out.value is None
    assert qmlib.qm_comm_create_rank(None, 0, 1, None, C.byref(out)) == -1
    assert qmlib.qm_allreduce_counters(None, None, None) == -6        # QM_E_STATE: nothing was run
    assert qmlib.qm_comm_collectives(None, None) == -1
    qmlib.qm_comm_destroy(None)


@pytest.mark.gpu
def test_the_library_collective_over_rccl_with_one_member(qmlib):
    """qm_comm_create over ONE context (all a one-GPU box allows: RCCL refuses two ranks on a card), a step = run + finish +
    qm_allreduce_counters: RCCL initialises on this device, exactly one collective per step is issued, and the all-reduced sums of
    a communicator of one member are the batch's own.  The same through qm_comm_make_id + qm_comm_create_rank (rank 0 of 1: the
    one-process-per-GPU form)."""
    import ctypes as C
    import numpy as np
    import quasimodo_amd as q
    with q.Engine(0) as eng:
        tid = eng.truth_synth(5_000_000, 100_000, 3)
        b = eng.batch([1_000_000] * 6, [tid] * 6)
        b.synth(5_000_000, 100_000, 3, 3000)
        b.run(); b.finish()
        before = b.global_counts().copy()
        assert before.sum() > 0
        for how in ("all", "rank"):
            comm = C.c_void_p()
            if how == "all":
                arr = (C.c_void_p * 1)(eng._h)
                q._lib.check(qmlib.qm_comm_create(arr, 1, C.byref(comm)), eng._h)
            else:
                cid = (C.c_char * 128)()
                q._lib.check(qmlib.qm_comm_make_id(cid), eng._h)
                q._lib.check(qmlib.qm_comm_create_rank(eng._h, 0, 1, cid, C.byref(comm)), eng._h)
            try:
                for step in range(3):
                    b.run(); b.finish()
                    q._lib.check(qmlib.qm_allreduce_counters(b._h, comm, None), eng._h)
                    assert qmlib.qm_comm_collectives(comm, eng._h) == step + 1
                    assert np.array_equal(b.global_counts(), before)
            finally:
                qmlib.qm_comm_destroy(comm)
        b.close()



def test_traffic_json_names_the_current_device_code():
    """bench.py quotes profiles/traffic.json (PMC bytes per k_classify launch) only for the device code it was measured on:
    a kernel edit after the last PMC pass would silently turn `roofline.traffic` into null in the driver's bench line."""
    import json
    import quasimodo_amd
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    t = json.load(open(os.path.join(root, "profiles", "traffic.json")))
    assert isinstance(t.get("kernels_build"), str) and len(t["kernels_build"]) == 16 and t["hbm_bytes_per_launch"] > 0
    if t["kernels_build"] != quasimodo_amd.kernel_source_id():   # (a warning, not a failure: kernels are edited between PMC passes)
        import warnings
        warnings.warn("profiles/traffic.json was measured on device code %s, the tree holds %s: bench.py will print roofline.traffic = null "
                      "until tools/final_profiles.sh has run on the GPU again" % (t["kernels_build"], quasimodo_amd.kernel_source_id()))


def test_every_knob_the_sources_read_is_named_in_integration_md():
    """INTEGRATION.md's Environment section is the list of QM_* variables: a knob added to the library, the package or the CLI
    without a line there is a knob nobody can find (round 5 found five).  bench.py's own QM_BENCH_* are the defaults of its
    flags and are named as a family."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    named = set(re.findall(r"QM_[A-Z0-9_]+", doc))
    assert "QM_BENCH_" in doc
    read = set()
    for f in glob.glob(os.path.join(root, "quasimodo_amd", "csrc", "*.[ch]*")):
        read |= set(re.findall(r'getenv\("(QM_[A-Z0-9_]+)"\)', open(f, errors="replace").read()))
    for f in glob.glob(os.path.join(root, "quasimodo_amd", "*.py")) + [os.path.join(root, n) for n in ("run_benchmark.py", "__graft_entry__.py")] + \
            glob.glob(os.path.join(root, "program", "*.py")):
        read |= set(re.findall(r'environ[^\n"]*"(QM_[A-Z0-9_]+)"', open(f).read()))
    missing = sorted(k for k in read if k not in named and not k.startswith("QM_BENCH_"))
    assert not missing, "not named in INTEGRATION.md: %s" % ", ".join(missing)
    # round 6 cut the library's knobs to those a user or a test needs: the count is part of the contract (<= 20)
    lib = set()
    for f in glob.glob(os.path.join(root, "quasimodo_amd", "csrc", "*.[ch]*")):
        lib |= set(re.findall(r'getenv\("(QM_[A-Z0-9_]+)"\)', open(f, errors="replace").read()))
    assert len(lib) <= 20, sorted(lib)
    # ... and the shipped kernels compile ONE way: no ablation / profiling conditionals, only #ifndef defaults of tuning constants
    ksrc = open(os.path.join(root, "quasimodo_amd", "csrc", "qmvt_kernels.hip")).read()
    conds = re.findall(r"^#\s*(?:if|ifdef|elif)\b.*$", ksrc, flags=re.M)
    assert not conds, conds
    assert ksrc.count("\n") <= 3600, ksrc.count("\n")   # (4 219 before the prune of round 6)
