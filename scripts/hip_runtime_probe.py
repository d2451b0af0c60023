"""Which HIP runtime libksgpu.so runs on, and what a second one in the process does to it (round 3's 15 failures, gpurun_out/r03w_alltests.log).

Legs (each a fresh child process; `python scripts/hip_runtime_probe.py` runs them all, `... LEG` runs one in this process):
  system   /opt/rocm's libamdhip64.so.7 alone (mapped before the binding is imported; torch never imported)
  torch    the binding's policy: torch's bundled runtime mapped first, library bound to it, `import torch` + a CUDA tensor afterwards
  both     /opt/rocm's runtime mapped first, a context created, then torch imported and initialised (a SECOND runtime), more first-use kernels,
           then a second ks_ctx_create (must be refused with the two-runtime message)
  both_torch_first   round 3's order (tests/test_abi.py of that round): the library mapped and bound to /opt/rocm's runtime but not yet used, torch imported
           and INITIALISED first, then the library's first context (ks_runtime_allow_multiple: the refusal is lifted for the diagnosis) and the kernels
Every leg runs the instantiations that failed in round 3 (61-column BVDot: k_panel_dot_direct<4,4>; 64 x 33 and 48 x 64 panels; a 64-column Gram-Schmidt
update) for the first time in its process and prints ks_runtime_info."""
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SYSTEM_RT = "/opt/rocm/lib/libamdhip64.so.7"


def first_use_ops(ks, ctx, tag):
    import numpy as np
    rng = np.random.default_rng(1)
    res = {}
    for (n, my, nx) in ((5000, 61, 61), (9000, 64, 33), (3000, 48, 64)):
        X = ks.BV(ctx, n, nx); Y = ks.BV(ctx, n, my)
        Xh = rng.standard_normal((n, nx)); Yh = rng.standard_normal((n, my))
        X.set_dense(Xh); Y.set_dense(Yh)
        M = np.zeros((my, nx), order="F")
        try:
            X.Dot(Y, M)
            res["dot_%dx%d" % (my, nx)] = "ok, max err %.1e" % np.abs(M - Yh.T @ Xh).max()
        except Exception as e:                     # noqa: BLE001 - the probe reports whatever comes
            res["dot_%dx%d" % (my, nx)] = "FAILED: %s" % e
        Q = rng.standard_normal((nx, nx))
        try:
            X.MultInPlace(Q, 0, nx)
            res["multinplace_%d" % nx] = "ok, max err %.1e" % np.abs(X.dense() - Xh @ Q).max()
        except Exception as e:                     # noqa: BLE001
            res["multinplace_%d" % nx] = "FAILED: %s" % e
    n, m = 20000, 64
    V = ks.BV(ctx, n, m)
    Vh = rng.standard_normal((n, m)); V.set_dense(Vh)
    try:
        for j in range(m):
            V.OrthonormalizeColumn(j)
        G = V.dense(); res["gs_64_columns"] = "ok, |V^T V - I| %.1e" % np.abs(G.T @ G - np.eye(m)).max()
    except Exception as e:                         # noqa: BLE001
        res["gs_64_columns"] = "FAILED: %s" % e
    print("[%s] ops: %s" % (tag, json.dumps(res)))
    return res


def leg(name):
    if name in ("system", "both", "both_torch_first"):
        ctypes.CDLL(SYSTEM_RT, mode=ctypes.RTLD_GLOBAL)          # defeats the binding's policy: something is mapped already
    import slepc_amd as ks
    from slepc_amd import _lib
    print("[%s] after loading the library: %s" % (name, json.dumps(_lib.runtime_info())))
    if name == "both_torch_first":
        import torch
        ok = torch.cuda.is_available()
        t = torch.arange(1024, device="cuda", dtype=torch.float64).sum().item() if ok else None
        print("[%s] torch %s imported and initialised BEFORE the library's first HIP call, cuda available %s, a reduction gives %s" % (name, torch.__version__, ok, t))
        print("[%s] after torch: %s" % (name, json.dumps(_lib.runtime_info())))
        try:
            ks.Context(0)
            print("[%s] first context: created (unexpected)" % name)
        except ks.KsError as e:
            print("[%s] first context: refused: %s" % (name, str(e)[:160]))
        _lib.lib().ks_runtime_allow_multiple(1)
        ctx = ks.Context(0)
        first_use_ops(ks, ctx, "both_torch_first, torch's runtime initialised first, the library's second")
        print("[%s] at the end: %s" % (name, json.dumps(_lib.runtime_info())))
        ctx.close()
        return
    ctx = ks.Context(0)
    first = name != "both"
    if first:
        first_use_ops(ks, ctx, name + ", before torch" if name == "torch" else name)
    if name in ("torch", "both"):
        import torch
        ok = torch.cuda.is_available()
        t = torch.arange(1024, device="cuda", dtype=torch.float64).sum().item() if ok else None
        print("[%s] torch %s imported, cuda available %s, a reduction on the device gives %s" % (name, torch.__version__, ok, t))
        print("[%s] after torch: %s" % (name, json.dumps(_lib.runtime_info())))
        if name == "both":
            first_use_ops(ks, ctx, "both, library's runtime initialised first, torch's second")
        else:
            A = ks.Mat.laplacian3d(ctx, 40, 40, 40)
            eps = ks.EPS(ctx); eps.SetOperators(A); eps.SetProblemType(ks.EPS_HEP); eps.SetDimensions(4, 20); eps.Solve()
            print("[%s] a solve after torch: %d converged in %d iterations" % (name, eps.GetConverged(), eps.GetIterationNumber()))
        try:
            c2 = ks.Context(0)
            print("[%s] a second context after torch: created" % name)
            c2.close()
        except ks.KsError as e:
            print("[%s] a second context after torch: refused: %s" % (name, e))
    print("[%s] at the end: %s" % (name, json.dumps(_lib.runtime_info())))
    ctx.close()


if __name__ == "__main__":
    if len(sys.argv) > 1:
        leg(sys.argv[1])
    else:
        for name in ("system", "torch", "both", "both_torch_first"):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), name], capture_output=True, text=True, timeout=600)
            print(r.stdout.strip())
            err = [ln for ln in r.stderr.splitlines() if ln.strip()]
            print("[%s] exit code %d%s" % (name, r.returncode, ("; stderr: " + " | ".join(err[-6:])) if err else ""))
            sys.stdout.flush()
