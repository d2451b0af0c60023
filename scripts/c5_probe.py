"""Probe of the config-5-shaped problem at several sizes/targets: prints convergence and throughput (not a test)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import slepc_amd as ks
import nhep_cases as nc

ctx = ks.Context(0)
out = open(sys.argv[1], "a") if len(sys.argv) > 1 else sys.stdout
def log(*a):
    print(*a, file=out); out.flush()
cases = [(100000, 36.0, 1500), (100000, 0.0, 600), (1000000, 36.0, 600), (1000000, 0.0, 400), (5000000, 0.0, 300)]
if os.environ.get("C5_ONLY_BIG"):
    cases = cases[-2:]
for n, sigma, cap in cases:
    t = time.time(); Ao, Bo = nc.config5_pencil_fast(n); tg = time.time() - t
    A = ks.Mat.from_csr(ctx, Ao.rowptr, Ao.col, Ao.val); B = ks.Mat.from_csr(ctx, Bo.rowptr, Bo.col, Bo.val)
    eps = ks.EPS(ctx)
    eps.SetOperators(A, B); eps.SetProblemType(ks.EPS_GNHEP); eps.SetDimensions(20, 60); eps.SetTarget(sigma)
    st = eps.GetST(); st.SetType("sinvert"); st.SetKSPType(os.environ.get("C5_KSP", "gmres"))
    eps.SetMaxSteps(cap)
    t = time.time(); eps.Solve(); dt = time.time() - t
    s = st.GetKSPStats(); es = eps.GetStats()
    log("n=%d sigma=%g gen %.1fs: reason %d nconv %d its %d steps %d  %.2fs -> %.1f steps/s, GMRES %.1f its/solve" % (
        n, sigma, tg, eps.GetConvergedReason(), eps.GetConverged(), eps.GetIterationNumber(), es["arnoldi_steps"], dt, es["arnoldi_steps"] / dt, s["iterations"] / max(1, s["solves"])))
    for i in range(min(3, eps.GetConverged())):
        log("   ", eps.GetEigenvalue(i), eps.ComputeError(i))
    del eps, A, B
