"""Shared inputs for the non-symmetric (EPS_NHEP, Arnoldi) cases: the reference's ex5 / test9 set-ups and a
seeded random matrix whose dominant eigenvalues include complex-conjugate pairs."""
import numpy as np

from oracle import oracle as O

SQRT_EPS = np.sqrt(np.finfo(float).eps)


def my_eigen_sort(ar, ai, br, bi, origin=0.0):
    """MyEigenSort of src/eps/tests/test9.c:195-204: closest to the origin, ties broken towards the right."""
    da, db = np.hypot(ar - origin, ai), np.hypot(br - origin, bi)
    d = (db - da) / max(da, db)
    if d > SQRT_EPS:
        return 1
    if d < -SQRT_EPS:
        return -1
    return 1 if br >= 0 else -1          # PetscSign(PetscRealPart(br))


def right_of(target):
    """MyEigenSort of src/eps/tests/test11.c:150-176: closest to the target, but on its right side first."""
    def cmp(ar, ai, br, bi):
        aright, bright = target < ar, target < br
        if aright == bright:
            da, db = np.hypot(ar - target, ai), np.hypot(br - target, bi)
            return -1 if da < db else (1 if da > db else 0)
        return -1 if aright else 1
    return cmp


def test9_v0(n):
    v0 = np.zeros(n)
    v0[0] = -1.5; v0[1] = 2.1            # test9.c:126-130
    return v0


def random_nonsymmetric(n, nnz_row=8, seed=7):
    """Sparse matrix with i.i.d. N(0, 1/nnz_row) entries: the spectrum fills the unit disk (circular law), so every
    extreme part of it is made of clustered complex-conjugate pairs and Krylov-Schur needs tens of restarts."""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    rows = np.repeat(np.arange(n), nnz_row)
    cols = rng.integers(0, n, n * nnz_row)
    vals = rng.standard_normal(n * nnz_row) / np.sqrt(nnz_row)
    S = sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr(); S.sum_duplicates(); S.sort_indices()
    return O.CSR(n, S.indptr, S.indices, S.data)


def planted_pairs(n, nnz_row=6, seed=7, rot=0.9):
    """Diagonal in [0,1] + small random off-diagonals + three 2x2 rotation blocks (radii 2.5, 2.2, 1.9) and two real
    outliers (2.35, -2.05): well-separated exterior eigenvalues, converges in one or two restarts."""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    rows = np.repeat(np.arange(n), nnz_row)
    cols = rng.integers(0, n, n * nnz_row)
    vals = rng.uniform(-1, 1, n * nnz_row) * 0.05
    A = sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tolil()
    A.setdiag(rng.uniform(0, 1, n))
    for b, (rad, th) in enumerate([(2.5, 0.6), (2.2, 1.3), (1.9, 2.2)]):
        i = 2 * b
        A[i, i] = rad * np.cos(th); A[i + 1, i + 1] = rad * np.cos(th)
        A[i, i + 1] = rad * np.sin(th) * rot; A[i + 1, i] = -rad * np.sin(th) / rot
    A[6, 6] = 2.35; A[7, 7] = -2.05
    S = A.tocsr(); S.sum_duplicates(); S.sort_indices()
    return O.CSR(n, S.indptr, S.indices, S.data)


def config5_pencil(n, mean_nnz=32, seed=42):
    """BASELINE config 5 at reduced n: A = random nonsymmetric CSR, row lengths Poisson(mean) clipped to
    [1, 2*mean], columns uniform without replacement, values uniform(-1,1), diagonal += 40; B = tridiagonal
    (1/6, 2/3, 1/6), the 1-D mass matrix. Returns (A, B)."""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    lens = np.clip(rng.poisson(mean_nnz, n), 1, min(2 * mean_nnz, n))
    rowptr = np.concatenate([[0], np.cumsum(lens)])
    cols = np.concatenate([rng.choice(n, l, replace=False) for l in lens])
    vals = rng.uniform(-1, 1, rowptr[-1])
    A = sp.csr_matrix((vals, cols, rowptr), shape=(n, n)) + 40.0 * sp.identity(n, format="csr")
    A.sum_duplicates(); A.sort_indices()
    B = sp.diags([np.full(n - 1, 1 / 6), np.full(n, 2 / 3), np.full(n - 1, 1 / 6)], [-1, 0, 1], format="csr")
    B.sort_indices()
    return (O.CSR(n, A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data),
            O.CSR(n, B.indptr.astype(np.int32), B.indices.astype(np.int32), B.data))


def config5_pencil_fast(n, mean_nnz=32, seed=42):
    """Vectorised variant for large n (slepc_amd/workloads.py, the generator bench.py uses): columns drawn WITH
    replacement, not sorted inside a row; same value distribution, diagonal + 40, same B."""
    from slepc_amd.workloads import config5_pencil_arrays
    (ar, ac, av), (br, bc, bv) = config5_pencil_arrays(n, mean_nnz, seed)
    return O.CSR(n, ar, ac, av), O.CSR(n, br, bc, bv)


def brusselator(N, alpha=2.0, beta=5.45, delta1=0.008, delta2=0.004, L=0.51302):
    """The operator of eps/tutorials/ex9.c (MatMult_Brussel :191-226) assembled: [[tau1 T + (beta-1) I, alpha^2 I],
    [-beta I, tau2 T - alpha^2 I]] with T = tridiag(1,-2,1) of order N, tau_i = delta_i / (h L)^2, h = 1/(N+1)."""
    import scipy.sparse as sp
    from oracle import oracle as O
    h = 1.0 / (N + 1)
    tau1, tau2 = delta1 / (h * L) ** 2, delta2 / (h * L) ** 2
    T = sp.diags([np.ones(N - 1), -2.0 * np.ones(N), np.ones(N - 1)], [-1, 0, 1])
    I = sp.identity(N)
    S = sp.bmat([[tau1 * T + (beta - 1.0) * I, alpha * alpha * I], [-beta * I, tau2 * T - alpha * alpha * I]]).tocsr()
    S.sort_indices()
    return O.CSR(2 * N, S.indptr.astype(np.int32), S.indices.astype(np.int32), S.data.astype(np.float64))
