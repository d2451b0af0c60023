"""Host dense kernels of the projected non-symmetric problem (slepc_amd/csrc/ks_dense.cpp: Hessenberg reduction,
real Schur form, Schur reordering, eigenvectors) against LAPACK through the oracle and against defining
properties. CPU only: the C hooks are exported by libksgpu.so and do not touch the GPU."""
import ctypes as C
import os

import numpy as np
import pytest

import slepc_amd._lib as L
from oracle import oracle as O

P = C.POINTER(C.c_double)


def p(a):
    return a.ctypes.data_as(P)


@pytest.fixture(scope="module")
def lib():
    lib = C.CDLL(os.environ.get("KS_HOST_HOOKS_LIB") or L.LIB_PATH)      # tests/test_sanitizers.py points this at the ASan + UBSan build of the host sources
    lib.ksd_hess_reduce.argtypes = [C.c_int, C.c_int, P, C.c_int, P]
    lib.ksd_hess_reduce.restype = None
    lib.ksd_real_schur.argtypes = [C.c_int, C.c_int, P, C.c_int, P, P, P]
    lib.ksd_trexc_up.argtypes = [C.c_int, P, C.c_int, P, C.c_int, C.c_int]
    lib.ksd_trevc_one.argtypes = [C.c_int, P, C.c_int, C.c_int, P, P]
    lib.ksd_potrf_upper.argtypes = [C.c_int, P, C.c_int]
    lib.ksd_trtri_upper.argtypes = [C.c_int, P, C.c_int]
    lib.ksd_sym_eig.argtypes = [C.c_int, P, C.c_int, P]
    lib.ksd_tsqr_combine.argtypes = [C.c_int, P, C.c_int, P, C.c_int]
    lib.ksd_tsqr_combine.restype = None
    lib.ksd_lu_solve_trans.argtypes = [C.c_int, P, C.c_int, P]
    lib.ksd_qr_explicit.argtypes = [C.c_int, C.c_int, P, C.c_int, P, C.c_int, P, C.c_int]
    lib.ksd_qr_explicit.restype = None
    return lib


def schur(lib, A0, ilo=0):
    n = A0.shape[0]
    A = np.asfortranarray(A0.copy()); Q = np.asfortranarray(np.eye(n))
    lib.ksd_hess_reduce(n, ilo, p(A), n, p(Q))
    H = A.copy(order="F")
    wr = np.zeros(n); wi = np.zeros(n)
    assert lib.ksd_real_schur(n, ilo, p(A), n, p(wr), p(wi), p(Q)) == 0
    return H, A, Q, wr, wi


@pytest.mark.parametrize("n", [1, 2, 3, 5, 8, 17, 30, 64, 150])
@pytest.mark.parametrize("ilo", [0, 3])
def test_schur_form(lib, n, ilo):
    ilo = min(ilo, n - 1)
    rng = np.random.default_rng(100 * n + ilo)
    A0 = rng.standard_normal((n, n))
    A0[:, :ilo] = np.triu(A0)[:, :ilo]            # locked part is already triangular (DSSolve after a restart)
    H, T, Q, wr, wi = schur(lib, A0, ilo)
    tol = 50 * n * np.finfo(float).eps * max(1.0, np.abs(A0).max())
    assert np.all(np.tril(H, -2) == 0)
    assert np.all(np.tril(T, -2) == 0)
    assert np.abs(Q.T @ Q - np.eye(n)).max() < tol
    assert np.abs(Q @ T @ Q.T - A0).max() < tol
    for j in range(n - 1):                        # 2x2 blocks are standardised (dlanv2): equal diagonal, opposite signs
        if T[j + 1, j] != 0:
            assert T[j, j] == T[j + 1, j + 1] and T[j, j + 1] * T[j + 1, j] < 0
            assert wi[j] > 0 and wi[j + 1] == -wi[j]
    ev = wr + 1j * wi
    ev[:ilo] = np.diag(T)[:ilo]
    ref = np.linalg.eigvals(A0)
    assert np.abs(np.sort_complex(ev) - np.sort_complex(ref)).max() < 1e3 * tol


def test_schur_matches_lapack_eigenvalue_order(lib):
    """Same deflation order as LAPACK's dhseqr (n <= 75 runs dlahqr): the Schur form's diagonal blocks line up."""
    rng = np.random.default_rng(5)
    for n in (6, 12, 31):
        A0 = rng.standard_normal((n, n))
        _, T, _, wr, wi = schur(lib, A0)
        ds = O.DSNHEP(n + 1, O.WHICH["largest_magnitude"])
        ds.A[:n, :n] = A0
        ds.SetDimensions(n, 0, 0)
        er = np.zeros(n + 1); ei = np.zeros(n + 1)
        ds.Solve(er, ei)
        assert np.allclose(wr, er[:n], rtol=0, atol=1e-11) and np.allclose(wi, ei[:n], rtol=0, atol=1e-11)


@pytest.mark.parametrize("n", [4, 9, 20, 40, 140])
def test_reorder_and_eigenvectors(lib, n):
    rng = np.random.default_rng(n)
    A0 = rng.standard_normal((n, n))
    _, T, Q, wr, wi = schur(lib, A0)
    tol = 200 * n * np.finfo(float).eps * np.abs(A0).max()
    ev0 = np.sort_complex(wr + 1j * wi)
    for _ in range(12):
        ifst = int(rng.integers(0, n)); ilst = int(rng.integers(0, ifst + 1))
        def block_eig(j):                           # eigenvalue (imag >= 0) of the diagonal block starting at row j
            if j < n - 1 and T[j + 1, j] != 0:
                return T[j, j] + 1j * np.sqrt(abs(T[j, j + 1])) * np.sqrt(abs(T[j + 1, j]))
            return T[j, j] + 0j
        fs = ifst - 1 if (ifst > 0 and T[ifst, ifst - 1] != 0) else ifst
        ls = ilst - 1 if (ilst > 0 and T[ilst, ilst - 1] != 0) else ilst
        moved = block_eig(fs)
        assert lib.ksd_trexc_up(n, p(T), n, p(Q), ifst, ilst) == 0
        assert np.all(np.tril(T, -2) == 0)
        assert np.abs(Q @ T @ Q.T - A0).max() < tol
        assert abs(block_eig(ls) - moved) < 1e-9 * max(1, abs(moved))
    k = 0
    evs = []
    while k < n:
        xr = np.zeros(n); xi = np.zeros(n)
        pair = lib.ksd_trevc_one(n, p(T), n, k, p(xr), p(xi))
        lam = T[k, k] + (1j * np.sqrt(abs(T[k, k + 1])) * np.sqrt(abs(T[k + 1, k])) if pair else 0.0)
        x = xr + 1j * xi
        assert np.abs(T @ x - lam * x).max() < 1e-9 * np.abs(x).max() * max(1, np.abs(T).max())
        assert np.all(x[k + 1 + pair:] == 0)
        assert abs((np.abs(x.real) + np.abs(x.imag)).max() - 1.0) < 1e-14      # dtrevc normalisation
        evs += [lam, np.conj(lam)] if pair else [lam]
        k += 2 if pair else 1
    assert np.abs(np.sort_complex(np.array(evs)) - ev0).max() < 1e-9


def test_reorder_matches_lapack(lib):
    """DSSort_NHEP_Total driven by LAPACK dtrexc (oracle) and by the host kernels selects the same ordering."""
    rng = np.random.default_rng(11)
    n = 14
    A0 = rng.standard_normal((n, n))
    _, T, Q, wr, wi = schur(lib, A0)
    ds = O.DSNHEP(n + 1, O.WHICH["largest_real"])
    ds.A[:n, :n] = A0
    ds.SetDimensions(n, 0, 0)
    er = np.zeros(n + 1); ei = np.zeros(n + 1)
    ds.Solve(er, ei); ds.Sort(er, ei)
    # same loop with the host kernels
    i = 0
    cmpf = O.WHICH["largest_real"]

    def eig_from_T():
        j = 0
        while j < n:
            if j == n - 1 or T[j + 1, j] == 0:
                wr[j] = T[j, j]; wi[j] = 0; j += 1
            else:
                wr[j] = wr[j + 1] = T[j, j]; wi[j] = np.sqrt(abs(T[j + 1, j])) * np.sqrt(abs(T[j, j + 1])); wi[j + 1] = -wi[j]; j += 2
    while i < n - 1:
        re, im, pos = wr[i], wi[i], 0
        j = i + 2 if im != 0 else i + 1
        while j < n:
            if cmpf(re, im, wr[j], wi[j]) > 0:
                re, im, pos = wr[j], wi[j], j
            if wi[j] != 0:
                j += 1
            j += 1
        if pos:
            assert lib.ksd_trexc_up(n, p(T), n, p(Q), pos, i) == 0
            eig_from_T()
        if wi[i] != 0:
            i += 1
        i += 1
    assert np.allclose(wr, er[:n], atol=1e-10) and np.allclose(wi, ei[:n], atol=1e-10)
    assert np.all(np.diff(wr) <= 1e-12)


@pytest.mark.parametrize("n", [1, 2, 5, 17, 40, 64, 100])
def test_cholesky_inverse_symmetric_eig_tsqr_combine(lib, n):
    """The k x k kernels of the block orthogonalisations (bvlapack.c:136-341,456-478) against LAPACK."""
    import scipy.linalg.lapack as la
    rng = np.random.default_rng(n)
    X = rng.standard_normal((3 * n + 2, n)); G = X.T @ X
    eps = np.finfo(float).eps
    A = np.asfortranarray(G.copy())
    assert lib.ksd_potrf_upper(n, p(A), n) == 0
    c, info = la.dpotrf(G, lower=0)
    assert np.abs(np.triu(A) - np.triu(c)).max() <= 50 * n * eps * np.abs(c).max()
    Ri = np.asfortranarray(np.triu(A))
    assert lib.ksd_trtri_upper(n, p(Ri), n) == 0
    assert np.abs(np.triu(Ri) @ np.triu(A) - np.eye(n)).max() < 1e3 * n * eps * np.linalg.cond(np.triu(A))
    E = np.asfortranarray(G.copy()); w = np.zeros(n)
    assert lib.ksd_sym_eig(n, p(E), n, p(w)) == 0
    assert np.all(np.diff(w) >= 0)                                        # ascending, as dsyev
    assert np.abs(w - np.linalg.eigvalsh(G)).max() <= 200 * n * eps * w.max()
    assert np.abs(E.T @ E - np.eye(n)).max() <= 100 * n * eps and np.abs(E @ np.diag(w) @ E.T - G).max() <= 200 * n * eps * np.abs(G).max()
    R1 = np.asfortranarray(np.triu(rng.standard_normal((n, n)))); R2 = np.asfortranarray(np.triu(rng.standard_normal((n, n))))
    S = np.vstack([R1, R2])
    lib.ksd_tsqr_combine(n, p(R1), n, p(R2), n)
    assert np.all(np.tril(R1, -1) == 0)
    assert np.abs(R1.T @ R1 - S.T @ S).max() <= 100 * n * eps * np.abs(S.T @ S).max()
    Rref = np.linalg.qr(S, mode="r")
    assert np.abs(np.abs(R1) - np.abs(Rref)).max() <= 1e3 * n * eps * np.abs(Rref).max()      # equal up to row signs


def test_cholesky_reports_indefinite(lib):
    B = np.asfortranarray(np.array([[1.0, 2.0], [2.0, 1.0]]))
    assert lib.ksd_potrf_upper(2, p(B), 2) == 2
    Z = np.asfortranarray(np.array([[1.0, 2.0], [0.0, 0.0]]))
    assert lib.ksd_trtri_upper(2, p(Z), 2) == 2


@pytest.mark.parametrize("n", [1, 2, 7, 33, 64])
def test_lu_solve_transposed_matches_lapack(lib, n):
    """(A - tau I)^T g = beta e_n of the harmonic translation (dsnhep.c:496-502): dgetrf + dgetrs 'T'."""
    import scipy.linalg as sl
    rng = np.random.default_rng(100 + n)
    A0 = np.triu(rng.standard_normal((n, n)), -1) - 0.3 * np.eye(n)       # upper Hessenberg, as the projected matrix
    b0 = np.zeros(n); b0[-1] = 0.7
    A = np.asfortranarray(A0.copy()); b = b0.copy()
    assert lib.ksd_lu_solve_trans(n, p(A), n, p(b)) == 0
    ref = sl.lu_solve(sl.lu_factor(A0), b0, trans=1)
    assert np.abs(b - ref).max() <= 1e3 * n * np.finfo(float).eps * max(np.abs(ref).max(), 1.0) * np.linalg.cond(A0)
    assert np.abs(A0.T @ b - b0).max() <= 1e3 * n * np.finfo(float).eps * np.abs(A0).max() * max(np.abs(b).max(), 1.0)
    # pivoting: a zero leading entry must not break it; an exactly singular matrix is reported
    Z = np.asfortranarray(np.array([[0.0, 2.0], [3.0, 1.0]])); c = np.array([1.0, 1.0])
    assert lib.ksd_lu_solve_trans(2, p(Z), 2, p(c)) == 0
    assert np.allclose(np.array([[0.0, 2.0], [3.0, 1.0]]).T @ c, [1.0, 1.0])
    S = np.asfortranarray(np.array([[1.0, 2.0], [2.0, 4.0]])); d = np.array([1.0, 0.0])
    assert lib.ksd_lu_solve_trans(2, p(S), 2, p(d)) == 2


@pytest.mark.parametrize("M,n", [(1, 1), (5, 5), (12, 4), (64 * 7, 64), (300, 30), (96, 32)])
def test_explicit_householder_qr_of_the_tsqr_stack(lib, M, n):
    """ksd::qr_explicit (dgeqr2 + dorg2r): the combine step of the tall-skinny QR - the stack of the row blocks' triangular factors
    factored with its orthogonal factor formed explicitly (bvlapack.c:421-433 does this per tree level with geqrf / orgqr). Against
    numpy's LAPACK QR up to column signs, incl. a stack of triangles, a rank-deficient stack and a zero column."""
    rng = np.random.default_rng(M * 100 + n)
    eps = np.finfo(float).eps
    cases = [rng.standard_normal((M, n))]
    if M % n == 0 and M > n:
        cases.append(np.vstack([np.triu(rng.standard_normal((n, n))) for _ in range(M // n)]))       # what the TSQR hands over
    if n >= 3:
        D = rng.standard_normal((M, n)); D[:, 2] = 2.0 * D[:, 0] - D[:, 1]; cases.append(D)           # dependent column
        Z = rng.standard_normal((M, n)); Z[:, 1] = 0.0; cases.append(Z)
    for A0 in cases:
        A = np.asfortranarray(A0.copy()); R = np.zeros((n, n), order="F"); Q = np.zeros((M, n), order="F")
        lib.ksd_qr_explicit(M, n, p(A), M, p(R), n, p(Q), M)
        assert np.all(np.tril(R, -1) == 0)
        assert np.abs(Q.T @ Q - np.eye(n)).max() <= 50 * max(M, 8) * eps
        assert np.abs(Q @ R - A0).max() <= 50 * max(M, 8) * eps * max(np.abs(A0).max(), 1.0)
        if np.linalg.matrix_rank(A0) == n:
            Rref = np.linalg.qr(A0, mode="r")
            assert np.abs(np.abs(R) - np.abs(Rref)).max() <= 1e3 * max(M, 8) * eps * np.abs(Rref).max()
