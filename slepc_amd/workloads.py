"""Synthetic inputs of the BASELINE configurations that are not generated on the device (numpy only; shared by
bench.py, the probe scripts and the tests so that they all run the same matrices).

The Laplacians of configs 1-4 are built directly in device memory by ks_mat_create_laplacian2d/3d (ex2.c:44-51,
ex19.c:47-78). Config 5: random nonsymmetric CSR, row lengths Poisson(32) clipped to [1, 64],
uniformly random columns, values uniform(-1, 1), diagonal + 40; B = tridiagonal (1/6, 2/3, 1/6), the 1-D mass matrix.
"""
import numpy as np


def config5_pencil_arrays(n, mean_nnz=32, seed=42):
    """(A, B) as CSR triplets (rowptr int32, col int32, val float64). Vectorised for large n: columns are drawn WITH
    replacement (a repeated (i, j) stays as two CSR entries, which MatMult and MatGetDiagonal sum) and are not sorted
    inside a row; the first entry of every row is its diagonal."""
    rng = np.random.default_rng(seed)
    lens = np.clip(rng.poisson(mean_nnz, n), 1, 2 * mean_nnz).astype(np.int64)
    rowptr = np.concatenate([[0], np.cumsum(lens + 1)]).astype(np.int32)        # + the diagonal entry
    nnz = int(rowptr[-1])
    col = rng.integers(0, n, nnz, dtype=np.int32)
    val = rng.uniform(-1, 1, nnz)
    col[rowptr[:-1]] = np.arange(n, dtype=np.int32); val[rowptr[:-1]] = 40.0
    # B: tridiagonal (1/6, 2/3, 1/6)
    brow = np.full(n, 3, dtype=np.int64); brow[0] = brow[-1] = 2 if n > 1 else 1
    browptr = np.concatenate([[0], np.cumsum(brow)]).astype(np.int32)
    bcol = np.empty(int(browptr[-1]), dtype=np.int32); bval = np.empty(int(browptr[-1]))
    i = np.arange(n)
    lo = browptr[:-1].astype(np.int64)
    has_l = i > 0; has_r = i < n - 1
    bcol[lo[has_l]] = i[has_l] - 1; bval[lo[has_l]] = 1 / 6
    d = lo + has_l
    bcol[d] = i; bval[d] = 2 / 3
    bcol[(d + 1)[has_r]] = i[has_r] + 1; bval[(d + 1)[has_r]] = 1 / 6
    return (rowptr, col, val), (browptr, bcol, bval)
