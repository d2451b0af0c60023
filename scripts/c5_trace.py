"""Config 5 (n = 5e6 random nonsymmetric pencil, sinvert at 0, nev 20, m 60) for 150 steps: run under rocprofv3 --kernel-trace (gaps between launches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slepc_amd as ks
from slepc_amd.workloads import config5_pencil_arrays
ctx = ks.Context(0)
(ar, ac, av), (br, bc, bv) = config5_pencil_arrays(5_000_000)
A = ks.Mat.from_csr(ctx, ar, ac, av); B = ks.Mat.from_csr(ctx, br, bc, bv)
del ar, ac, av, br, bc, bv
eps = ks.EPS(ctx); eps.SetOperators(A, B); eps.SetProblemType(ks.EPS_GNHEP); eps.SetDimensions(20, 60); eps.SetTarget(0.0)
eps.GetST().SetType("sinvert"); eps.SetMaxSteps(150)
eps.Solve()
print(eps.GetStats(), A.layout())
