"""Config 1 (2-D Laplacian 100^2, nev 4, m 20) for 3000 steps: run under rocprofv3 --kernel-trace to see whether the GPU waits for the host's launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slepc_amd as ks
ctx = ks.Context(0)
A = ks.Mat.laplacian2d(ctx, 100)
eps = ks.EPS(ctx); eps.SetOperators(A); eps.SetProblemType(ks.EPS_HEP); eps.SetDimensions(4, 20); eps.SetTolerances(1e-300, 1 << 30); eps.SetMaxSteps(3000)
t0 = time.perf_counter(); eps.Solve(); dt = time.perf_counter() - t0
st = eps.GetStats()
print(st, "%.1f steps/s" % (st["arnoldi_steps"] / dt))
