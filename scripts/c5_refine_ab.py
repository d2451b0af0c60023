"""Config 5 (n = 5e6, target 0, nev 20, m 60) with the inner GMRES's Gram-Schmidt refinement never (PETSc's default) / ifneeded, same matrices,
alternating; prints steps/s of a capped solve and the inner iteration count (not a test)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import slepc_amd as ks
import nhep_cases as nc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000000
ctx = ks.Context(0)
Ao, Bo = nc.config5_pencil_fast(n)
A = ks.Mat.from_csr(ctx, Ao.rowptr, Ao.col, Ao.val); B = ks.Mat.from_csr(ctx, Bo.rowptr, Bo.col, Bo.val)
for rnd in range(2):
    for refine in ("never", "ifneeded"):
        eps = ks.EPS(ctx)
        eps.SetOperators(A, B); eps.SetProblemType(ks.EPS_GNHEP); eps.SetDimensions(20, 60); eps.SetTarget(0.0)
        st = eps.GetST(); st.SetType("sinvert"); st.SetGMRESCGSRefinement(refine)
        eps.SetMaxSteps(300)
        ctx.synchronize()
        t = time.time(); eps.Solve(); ctx.synchronize(); dt = time.time() - t
        s = st.GetKSPStats(); es = eps.GetStats()
        print("refine %-9s %.1f steps/s (whole capped solve incl. first cycle), GMRES %.2f its/solve, nconv %d, lambda0 %r" % (
            refine, es["arnoldi_steps"] / dt, s["iterations"] / max(1, s["solves"]), eps.GetConverged(), eps.GetEigenvalue(0) if eps.GetConverged() else None), flush=True)
        del eps
