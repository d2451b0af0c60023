"""27-point stencil (rows of 27 entries) on a 128^3 grid in the SELL-64, dictionary (constant coefficients) and
offset-dictionary (random coefficients) layouts: us per product."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import slepc_amd as ks

ctx = ks.Context(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
tri = lambda n: sp.diags([np.ones(n - 1), np.ones(n), np.ones(n - 1)], [-1, 0, 1])       # noqa: E731
P = sp.kron(tri(N), sp.kron(tri(N), tri(N))).tocsr(); P.sort_indices()
n, nnz = P.shape[0], P.nnz
print("27-point stencil %d^3: n=%d nnz=%d" % (N, n, nnz), flush=True)
const = np.full(nnz, -1.0); const[P.indices == np.repeat(np.arange(n), np.diff(P.indptr))] = 26.0
rand = np.random.default_rng(0).standard_normal(nnz)
for name, data, fmts in (("constant coefficients", const, ("sell", "dict")), ("random coefficients", rand, ("sell", "odict"))):
    for fmt in fmts:
        os.environ["KSGPU_SPMV"] = fmt
        A = ks.Mat.from_csr(ctx, P.indptr, P.indices, data)
        V = ks.BV(ctx, n, 2); V.SetRandomColumn(0)
        x, y = V.column_ptr(0), V.column_ptr(1)
        for _ in range(10):
            A.mult_dev(x, y)
        ctx.synchronize(); t = time.time()
        for _ in range(100):
            A.mult_dev(x, y)
        ctx.synchronize(); dt = (time.time() - t) / 100
        print("%s, %s: layout=%s  %.1f us/product  CSR-algorithmic %.2f TB/s" % (name, fmt, A.layout(), dt * 1e6, (12.0 * nnz + 20.0 * n) / dt / 1e12), flush=True)
        del A, V
