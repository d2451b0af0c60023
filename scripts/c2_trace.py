"""Config 2 (2-D Laplacian 1000^2, nev 4, m 20) for 600 steps: run under rocprofv3 --kernel-trace to look at the gaps between dependent launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slepc_amd as ks
ctx = ks.Context(0)
A = ks.Mat.laplacian2d(ctx, 1000)
eps = ks.EPS(ctx); eps.SetOperators(A); eps.SetProblemType(ks.EPS_HEP); eps.SetDimensions(4, 20); eps.SetMaxSteps(600)
eps.Solve()
print(eps.GetStats())
