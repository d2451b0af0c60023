"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/ksgpu.h declares, binds with the ctypes signatures, and refuses to run without a GPU."""
import ctypes
import os
import subprocess

import pytest

import slepc_amd as ks
from slepc_amd import _lib


def test_library_is_in_tree_and_loads():
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    L = _lib.lib()
    assert L.ks_error_string(0) == b"success"
    assert b"inner product" in L.ks_error_string(95)          # PETSC_ERR_USER_INPUT
    assert b"pivot" in L.ks_error_string(71)                  # PETSC_ERR_MAT_LU_ZRPVT


def test_every_header_symbol_is_exported():
    L = _lib.lib()
    names = _lib.header_symbols()
    assert len(names) >= 80
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_ctypes_signatures_cover_header():
    declared = set(_lib.header_symbols())
    bound = set(_lib._SIG) | set(_lib._STR_FUNCS)
    assert declared == bound, (declared - bound, bound - declared)


def test_no_torch_types_in_abi():
    import re
    raw = open(_lib.HEADER_PATH).read()
    assert 'extern "C"' in raw
    code = re.sub(r"/\*.*?\*/", "", raw, flags=re.S)            # declarations only
    assert "torch" not in code.lower() and "at::" not in code and "c10" not in code and "Tensor" not in code
    assert "#include <hip" not in code                             # plain pointers and sizes only


def test_gfx950_code_object_present():
    """The fat binary carries exactly one device code object and its target is gfx950."""
    import re
    blob = open(_lib.LIB_PATH, "rb").read()
    targets = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-f]+)", blob))
    assert targets == {b"gfx950"}, targets


def test_fails_loudly_without_gpu():
    # (no torch here: once its bundled HIP runtime is initialised beside the library's own in one process, occupancy queries of the library's kernels fail)
    try:
        ks.Context(0)
    except ks.KsError as e:
        assert e.rc == 97             # KS_ERR_GPU, no CPU fallback
        return
    pytest.skip("a GPU is present")


def test_product_does_not_import_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dirpath, _, files in os.walk(os.path.join(root, "slepc_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cuh", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "liboracle" not in txt, f


def _build_c_example(tmp_path, name="ex2_abi"):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / name)
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(root, "include"),
           os.path.join(root, "tests", "c_abi", name + ".c"), "-o", exe, "-L" + os.path.join(root, "slepc_amd"), "-l:libksgpu.so",
           "-Wl,-rpath," + os.path.join(root, "slepc_amd"), "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


@pytest.mark.parametrize("name", ["bv_test1_abi", "gs_slot_abi"])
def test_c_drivers_of_the_bv_slots_compile_as_strict_c99(tmp_path, name):
    """The C programs that drive the BV-level slots (tests/test_gpu_cabi.py runs them on the GPU) build here."""
    _build_c_example(tmp_path, name)


def test_header_is_plain_c_and_library_links_from_c(tmp_path):
    """include/ksgpu.h compiles as strict C99 (-pedantic -Werror) and a C program links against libksgpu.so; without a
    GPU the program stops at ks_ctx_create with the PETSc-numbered error and says there is no CPU fallback."""
    import subprocess
    exe = _build_c_example(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    if r.returncode == 0:
        pytest.skip("a GPU is present: the run itself is checked by tests/test_gpu_krylov.py::test_c_program_against_the_abi")
    assert r.returncode == 1 and "97" in r.stderr and "no CPU fallback" in r.stderr


_ONE_RUNTIME_CHILD = r"""
import json, sys
sys.path.insert(0, %r)
order = sys.argv[1]
if order == "torch_first":
    import torch
from slepc_amd import _lib
_lib.lib()
before = _lib.runtime_info()
if order == "library_first":
    import torch
after = _lib.runtime_info()
print(json.dumps({"before": before, "after": after, "torch_lib": torch.__file__}))
"""


@pytest.mark.parametrize("order", ["library_first", "torch_first"])
def test_one_hip_runtime_per_process_whatever_the_import_order(order):
    """PyTorch's wheel bundles a libamdhip64.so under the SONAME libksgpu.so needs. Imported AFTER the library it used to become a second
    HIP + HSA runtime in the process (round 3: 15 GPU tests failed that way); the binding now maps torch's copy first when nothing is
    mapped yet, so the process ends up with ONE runtime in either order and ks_runtime_info says which the library is bound to."""
    import json
    import sys
    r = subprocess.run([sys.executable, "-c", _ONE_RUNTIME_CHILD % os.path.dirname(os.path.dirname(_lib.LIB_PATH)), order], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    for when in ("before", "after"):
        info = d[when]
        assert len(info["hip_runtimes_mapped"]) == 1, info
        assert os.path.realpath(info["hip_runtime_path"]) == os.path.realpath(info["hip_runtimes_mapped"][0]), info
        assert info["occupancy_query_failures"] == 0
    torch_rt = os.path.join(os.path.dirname(d["torch_lib"]), "lib", "libamdhip64.so")
    if os.path.exists(torch_rt):          # a wheel with a bundled runtime: that is the one both end up on
        assert os.path.realpath(d["after"]["hip_runtime_path"]) == os.path.realpath(torch_rt), d


def test_context_creation_refuses_a_process_with_two_hip_runtimes():
    """The policy defeated on purpose (the system runtime mapped before the binding loads, torch imported afterwards): ks_ctx_create says so
    instead of running beside a second runtime. No GPU needed: the check comes before the first HIP call."""
    import sys
    system_rt = "/opt/rocm/lib/libamdhip64.so.7"
    if not os.path.exists(system_rt):
        pytest.skip("no system HIP runtime to map first")
    code = r"""
import ctypes, sys
sys.path.insert(0, %r)
ctypes.CDLL(%r, mode=ctypes.RTLD_GLOBAL)
import slepc_amd as ks
from slepc_amd import _lib
_lib.lib()
import torch
info = _lib.runtime_info()
if len(info["hip_runtimes_mapped"]) < 2:
    print("SKIP one runtime only"); sys.exit(0)
try:
    ks.Context(0)
    print("CREATED")
except ks.KsError as e:
    print("REFUSED", e.rc, e)
""" % (os.path.dirname(os.path.dirname(_lib.LIB_PATH)), system_rt)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    if last.startswith("SKIP"):
        pytest.skip("torch bundles no second runtime here")
    assert last.startswith("REFUSED 76") and "two HIP runtimes" in last, last


def test_runtime_info_needs_no_gpu_and_checks_its_buffer():
    """ks_runtime_info works without a context or a GPU (it never initialises the runtime beyond asking its version) and refuses a short buffer."""
    L = _lib.lib()
    info = _lib.runtime_info()
    assert set(info) >= {"hip_runtime_path", "hip_runtime_version", "hip_runtimes_mapped", "occupancy_query_failures", "occupancy_last_error"}
    assert os.path.exists(info["hip_runtime_path"]) and info["hip_runtime_version"] > 0
    small = ctypes.create_string_buffer(16)
    assert L.ks_runtime_info(small, 16) == 60                       # KS_ERR_ARG_SIZ
    assert L.ks_runtime_info(None, 0) == 85                          # KS_ERR_ARG_NULL
    assert ks.event_name("bv_dot_sweep") == "BVDotVec" and ks.event_name("gs_update_fused_dot") == "BVMultVec+BVDotVec" and ks.event_name("spmv_csr") == "BVMatMultVec"
