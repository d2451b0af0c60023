"""One HIP runtime per process, on the GPU (runs last: the file name sorts behind the other GPU tests).

Round 3 lost 15 GPU tests to a second HIP runtime in the test process: the library bound to /opt/rocm's libamdhip64.so.7, torch's bundled copy mapped and
initialised beside it, and from some point on hipOccupancyMaxActiveBlocksPerMultiprocessor of the library's runtime answered hipErrorUnknown for every kernel
instantiation that fits one workgroup per CU (reproduced in round 4 with the counters below: profiles/r04_hip_runtime_probe.txt). The binding now maps
torch's copy first whenever torch is installed, so the process has ONE runtime in either import order; ks_ctx_create refuses a process that maps two; a
failed occupancy query is counted and reported, never silent."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_library_first_then_torch_runs_on_one_runtime_at_full_occupancy():
    """A fresh process: the library first (contexts, the kernels that failed in round 3 used for the first time), then torch imported and used, then more
    first-use kernels and a solve. One runtime mapped throughout, no failed occupancy query, results right."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "hip_runtime_probe.py"), "torch"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("[torch")]
    infos = [json.loads(ln.split(": ", 1)[1]) for ln in lines if "hip_runtime_path" in ln]
    assert len(infos) == 3, r.stdout
    for info in infos:
        assert len(info["hip_runtimes_mapped"]) == 1 and info["occupancy_query_failures"] == 0, info
    ops = [json.loads(ln.split("ops: ", 1)[1]) for ln in lines if "] ops: " in ln]
    assert ops and all(v.startswith("ok") for d in ops for v in d.values()), ops
    assert any("a solve after torch: 4 converged" in ln for ln in lines), r.stdout
    assert any("a second context after torch: created" in ln for ln in lines), r.stdout


@pytest.mark.gpu
def test_this_session_ran_on_one_runtime_without_a_failed_occupancy_query(ctx):
    from slepc_amd import _lib
    info = _lib.runtime_info()
    assert len(info["hip_runtimes_mapped"]) == 1, info
    assert info["occupancy_query_failures"] == 0, info
