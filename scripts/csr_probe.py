"""The general-matrix CSR kernels on the bench matrix (216^3 7-point Laplacian) and on ragged random matrices: us per product and
CSR-algorithmic TB/s (SURVEY 8d bytes) for KSGPU_SPMV = csr (wave-per-64-rows row-block kernel; short rows: its LDS-DMA form), csrregs (the register-staged form also for short rows),
csrblock (workgroup per 256 rows), csrvec, sell."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import slepc_amd as ks

ctx = ks.Context(0)
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 216
fmts = sys.argv[2].split(",") if len(sys.argv) > 2 else ["csr", "csrblock", "sell", "csrvec"]


def timeit(A, reps=100):
    V = ks.BV(ctx, A.n, 2); V.SetRandomColumn(0)
    x, y = V.column_ptr(0), V.column_ptr(1)
    for _ in range(10):
        A.mult_dev(x, y)
    ctx.synchronize()
    ctx.prof_enable(True, classes=["spmv_csr"]); ctx.prof_reset()
    for _ in range(reps):
        A.mult_dev(x, y)
    ctx.synchronize()
    p = ctx.prof_get(); ctx.prof_enable(False)
    v = p["spmv_csr"]
    return 1e3 * v["ms"] / v["launches"]


ref = None
for fmt in fmts:
    os.environ["KSGPU_SPMV"] = fmt
    A = ks.Mat.laplacian3d(ctx, nx, nx, nx)
    us = timeit(A)
    csr = 12.0 * A.nnz + 4.0 * (A.n + 1) + 16.0 * A.n
    xs = np.random.default_rng(0).standard_normal(A.n)
    y = A.mult(xs)
    if ref is None:
        ref = y
    print("laplacian %d^3 %-8s layout=%-6s %7.1f us  %.2f TB/s  bits_equal_to_first=%s" % (nx, fmt, A.layout(), us, csr / us / 1e6, bool(np.array_equal(y, ref))), flush=True)
    A.destroy()

rng = np.random.default_rng(1)
for n, mean in ((2_000_000, 8), (1_000_000, 32), (500_000, 100)):
    lens = np.clip(rng.poisson(mean, n), 0, None); lens[::17] = 0
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    nnz = int(rowptr[-1])
    # columns near the diagonal (band of 64k): the x gathers stay in cache, the kernels' own streams are what is measured
    col = (np.repeat(np.arange(n), lens) + rng.integers(-32768, 32768, nnz)).clip(0, n - 1).astype(np.int32)
    val = rng.uniform(-1, 1, nnz)
    ref = None
    for fmt in [f for f in fmts if f != "sell"]:
        os.environ["KSGPU_SPMV"] = fmt
        A = ks.Mat.from_csr(ctx, rowptr, col, val)
        us = timeit(A, 50)
        csr = 12.0 * nnz + 4.0 * (n + 1) + 16.0 * n
        y = A.mult(np.ones(n))
        if ref is None:
            ref = y
        print("banded random n=%d mean %d %-8s layout=%-6s %7.1f us  %.2f TB/s  maxdiff %.1e" % (n, mean, fmt, A.layout(), us, csr / us / 1e6, np.abs(y - ref).max()), flush=True)
        A.destroy()
