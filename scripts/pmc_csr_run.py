"""10 products of the 216^3 Laplacian in the layout KSGPU_SPMV names (the workload of scripts/pmc_csr.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slepc_amd as ks
ctx = ks.Context(0)
A = ks.Mat.laplacian3d(ctx, 216, 216, 216)
V = ks.BV(ctx, A.n, 2); V.SetRandomColumn(0)
for _ in range(10):
    A.mult_dev(V.column_ptr(0), V.column_ptr(1))
ctx.synchronize()
