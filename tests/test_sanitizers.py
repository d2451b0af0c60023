"""The sanitizer run DESIGN.md claims, as a recipe that is run: the host-only sources of the product (ks_dense.cpp, ks_csr.cpp: `make -C slepc_amd/csrc asan`)
and the C oracle (`make -C oracle asan`) built with -fsanitize=address,undefined, and their CPU suites run on those builds in a child interpreter with the
sanitizer runtime preloaded. GPU sanitizers are not available on this pool: device code is outside this check."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_host_sources_and_oracle_under_asan_ubsan():
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not asan:
        pytest.skip("no libasan next to gcc")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "slepc_amd", "csrc"), "asan"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    env = dict(os.environ)
    env["LD_PRELOAD"] = asan + ((":" + ubsan) if ubsan else "")
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=0:exitcode=23"        # the interpreter's own allocations are not ours to judge
    env["UBSAN_OPTIONS"] = "halt_on_error=1:print_stacktrace=1"
    env["KS_HOST_HOOKS_LIB"] = os.path.join(ROOT, "slepc_amd", "libks_host_asan.so")
    env["ORACLE_LIB"] = "asan"
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", "tests/test_dense_host.py", "tests/test_csr_host.py",
                        "tests/test_oracle_golden.py", "-k", "not eps_ and not c5 and not nhep"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout[-1500:], r.stderr[-3000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr, tail
    assert r.returncode == 0, tail
    assert " passed" in r.stdout, tail
