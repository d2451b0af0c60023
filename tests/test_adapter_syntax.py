"""adapters/slepc/hipks.c cannot be compiled here (no PETSc in the image), but it can be kept from rotting: a syntax
and type check against prototypes of exactly the PETSc / SLEPc names it uses (tests/petsc_stub/, each citing the reference location
its signature was read from), the real include/ksgpu.h and the real HIP runtime header."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ADAPTER = os.path.join(ROOT, "adapters", "slepc", "hipks.c")
STUB = os.path.join(ROOT, "tests", "petsc_stub", "include")
HIP_INC = "/opt/rocm/include"


def _check(path, extra=()):
    cmd = ["gcc", "-std=gnu99", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter", "-D__HIP_PLATFORM_AMD__",
           "-I" + HIP_INC, "-I" + STUB, "-I" + os.path.join(ROOT, "include"), *extra, path]
    return subprocess.run(cmd, capture_output=True, text=True)


@pytest.mark.skipif(not os.path.exists(os.path.join(HIP_INC, "hip", "hip_runtime_api.h")), reason="needs the ROCm headers")
def test_adapter_passes_the_syntax_and_type_check():
    r = _check(ADAPTER)
    assert r.returncode == 0, r.stderr[:4000]


@pytest.mark.skipif(not os.path.exists(os.path.join(HIP_INC, "hip", "hip_runtime_api.h")), reason="needs the ROCm headers")
@pytest.mark.parametrize("old,new", [
    ("KS(ks_bv_scalecolumn(ctx->kbv,(int)j,alpha));", "KS(ks_bv_scalecolumn(ctx->kbv,alpha,(int)j,0));"),              # a ks_* call with the wrong argument list
    ("bv->ops->scale            = BVScale_HIPKS;", "bv->ops->scale            = BVNorm_HIPKS;"),                         # a function of the wrong type in a slot
    ("KS(ks_bv_set_state(ctx->kbv,(uint64_t)state));", "KS(ks_bv_set_state(ctx->kbv,(uint64_t)bv->no_such_field));"),   # a field struct _p_BV does not have
])
def test_the_check_bites(tmp_path, old, new):
    """Negative controls: the same check rejects the adapter with one deliberate mistake in it."""
    src = open(ADAPTER).read()
    assert src.count(old) == 1, old
    d = tmp_path / "adapters" / "slepc"
    d.mkdir(parents=True)
    bad = d / "hipks.c"
    bad.write_text(src.replace(old, new))
    r = _check(str(bad))
    assert r.returncode != 0


def test_view_slots_are_the_references_own_functions():
    """The slots that only move views around are BVSVEC's own HIP functions (svec.h declares them SLEPC_INTERN), installed as they are:
    the adapter has no body of its own for them, and its private data begins with a BV_SVEC so that their casts hold."""
    src = open(ADAPTER).read()
    for slot, fn in [("getcolumn", "BVGetColumn_Svec_HIP"), ("restorecolumn", "BVRestoreColumn_Svec_HIP"), ("getmat", "BVGetMat_Svec_HIP"),
                     ("restoremat", "BVRestoreMat_Svec_HIP"), ("matmult", "BVMatMult_Svec_HIP")]:
        assert re.search(r"bv->ops->%s\s*=\s*%s;" % (slot, fn), src), slot
    for gone in ("BVGetColumn_HIPKS", "BVRestoreColumn_HIPKS", "BVGetMat_HIPKS", "BVRestoreMat_HIPKS", "BVMatMult_HIPKS"):
        assert gone not in src
    m = re.search(r"typedef struct \{\s*BV_SVEC\s+svec;", src)
    assert m, "BV_HIPKS must begin with a BV_SVEC"
    # every prototype of the stub carries a citation
    stub_dir = os.path.join(ROOT, "tests", "petsc_stub")
    n = 0
    for dp, _, files in os.walk(stub_dir):
        for f in files:
            if f.endswith(".h"):
                for ln in open(os.path.join(dp, f)):
                    if re.match(r"^PetscErrorCode \w+\(", ln) and "Restore" not in ln and "BV_" not in ln:
                        n += 1
                        assert "/*" in ln, ln
    assert n > 40


def test_every_ks_entry_point_the_adapter_calls_is_declared():
    hdr = open(os.path.join(ROOT, "include", "ksgpu.h")).read()
    used = set(re.findall(r"\b(ks_[a-z0-9_]+)\s*\(", open(ADAPTER).read()))
    declared = set(re.findall(r"\b(ks_[a-z0-9_]+)\s*\(", re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)))
    assert used and used <= declared, used - declared
