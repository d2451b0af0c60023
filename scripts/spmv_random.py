"""Run the SpMV of the config-5-shaped random matrix 10 times (for rocprofv3 --pmc runs); KSGPU_SPMV picks the layout."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import slepc_amd as ks
import nhep_cases as nc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000000
ctx = ks.Context(0)
Ao, _ = nc.config5_pencil_fast(n)
A = ks.Mat.from_csr(ctx, Ao.rowptr, Ao.col, Ao.val)
V = ks.BV(ctx, n, 2); V.set_column(0, np.random.default_rng(0).standard_normal(n))
for _ in range(10):
    A.mult_dev(V.column_ptr(0), V.column_ptr(1))
ctx.L.ks_ctx_synchronize(ctx.h)
print("layout", A.layout(), "nnz", A.nnz)
