"""BASELINE config 3 solved to convergence (not the step-capped bench run): 216^3 Laplacian, nev = 10, m = 30, tol 1e-8;
prints restarts, steps, time, and the converged Ritz values against the analytic spectrum (ex19.c:19-45)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import slepc_amd as ks
from oracle import oracle as O

N = int(sys.argv[1]) if len(sys.argv) > 1 else 216
max_it = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
ctx = ks.Context(0)
A = ks.Mat.laplacian3d(ctx, N, N, N)
eps = ks.EPS(ctx)
eps.SetOperators(A); eps.SetProblemType(ks.EPS_HEP); eps.SetDimensions(10, 30); eps.SetTolerances(1e-8, max_it)
t = time.time(); eps.Solve(); ctx.synchronize(); dt = time.time() - t
st = eps.GetStats()
print("n=%d reason %d nconv %d its %d steps %d  %.2f s  %.1f steps/s" % (A.n, eps.GetConvergedReason(), eps.GetConverged(), eps.GetIterationNumber(), st["arnoldi_steps"], dt, st["arnoldi_steps"] / dt), flush=True)
s1 = 4.0 * np.sin(np.arange(1, N + 1) * np.pi / (2.0 * (N + 1))) ** 2
top = np.sort(s1)[::-1][:12]
exact = np.sort((top[:, None, None] + top[None, :, None] + top[None, None, :]).ravel())[::-1]
for i in range(eps.GetConverged()):
    lam = eps.GetEigenvalue(i)[0]
    print("  %.12f  rel. distance to the spectrum %.2e  residual %.2e" % (lam, np.min(np.abs(exact - lam)) / lam, eps.ComputeError(i)), flush=True)
