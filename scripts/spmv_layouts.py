"""SpMV time of the 216^3 Laplacian (bench matrix) in the SELL-64 and dictionary layouts: us per product, algorithmic
GB/s (CSR bytes) and the layout's own bytes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import slepc_amd as ks

ctx = ks.Context(0)
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 216
for fmt in ("sell", "dict"):
    os.environ["KSGPU_SPMV"] = fmt
    A = ks.Mat.laplacian3d(ctx, nx, nx, nx)
    V = ks.BV(ctx, A.n, 2)
    V.SetRandomColumn(0)
    x, y = V.column_ptr(0), V.column_ptr(1)
    for _ in range(20):
        A.mult_dev(x, y)
    ctx.synchronize(); t = time.time()
    reps = 200
    for _ in range(reps):
        A.mult_dev(x, y)
    ctx.synchronize(); dt = (time.time() - t) / reps
    nnz = 7 * A.n
    csr = 12.0 * nnz + 4.0 * (A.n + 1) + 16.0 * A.n
    own = {"sell": 12.0 * nnz * 1.0 + 4.0 * A.n + 16.0 * A.n, "dict": 32.0 * A.n}[fmt]
    print("%s: layout=%s  %.1f us/product  CSR-algorithmic %.2f TB/s  own bytes %.2f TB/s" % (fmt, A.layout(), dt * 1e6, csr / dt / 1e12, own / dt / 1e12), flush=True)
    del A, V

# variable coefficients on the same stencil: SELL-64 against the offset-dictionary layout
from oracle import oracle as O
Ao = O.laplacian3d(nx, nx, nx, omp=True)
val = np.random.default_rng(0).standard_normal(Ao.val.shape[0])
for fmt in ("sell", "odict"):
    os.environ["KSGPU_SPMV"] = fmt
    A = ks.Mat.from_csr(ctx, Ao.rowptr, Ao.col, val)
    V = ks.BV(ctx, A.n, 2); V.SetRandomColumn(0)
    x, y = V.column_ptr(0), V.column_ptr(1)
    for _ in range(20):
        A.mult_dev(x, y)
    ctx.synchronize(); t = time.time()
    for _ in range(200):
        A.mult_dev(x, y)
    ctx.synchronize(); dt = (time.time() - t) / 200
    nnz = Ao.val.shape[0]
    csr = 12.0 * nnz + 4.0 * (A.n + 1) + 16.0 * A.n
    own = {"sell": csr, "odict": 8.0 * nnz + 24.0 * A.n}[fmt]
    print("variable coefficients %s: layout=%s  %.1f us/product  CSR-algorithmic %.2f TB/s  own bytes %.2f TB/s" % (fmt, A.layout(), dt * 1e6, csr / dt / 1e12, own / dt / 1e12), flush=True)
    del A, V
