"""Host CSR helpers of the assembly path (slepc_amd/csrc/ks_csr.cpp): P = A + alpha B, the MatDuplicate + MatAXPY(DIFFERENT_NONZERO_PATTERN)
/ MatShift of ST_MATMODE_COPY (src/sys/classes/st/interface/stsolve.c:611-626). CPU only: the hook is exported by libksgpu.so and does not
touch the GPU."""
import ctypes as C
import os

import numpy as np
import pytest
import scipy.sparse as sp

import slepc_amd._lib as L

IP = C.POINTER(C.c_int)
DP = C.POINTER(C.c_double)


@pytest.fixture(scope="module")
def lib():
    lib = C.CDLL(os.environ.get("KS_HOST_HOOKS_LIB") or L.LIB_PATH)      # tests/test_sanitizers.py points this at the ASan + UBSan build of the host sources
    lib.ksc_csr_axpy.argtypes = [C.c_int, C.c_int, IP, IP, DP, C.c_double, IP, IP, DP, IP, IP, DP, C.c_longlong]
    lib.ksc_csr_axpy.restype = C.c_longlong
    return lib


def _i(a):
    return a.ctypes.data_as(IP)


def _d(a):
    return a.ctypes.data_as(DP)


def axpy(lib, A, alpha, B, row_start=0):
    """A, B: (rowptr, col, val) int32/int32/float64 arrays; B None = identity."""
    n = len(A[0]) - 1
    rp = np.zeros(n + 1, dtype=np.int32)
    nb = (None, None, None) if B is None else (_i(B[0]), _i(B[1]), _d(B[2]))
    nnz = lib.ksc_csr_axpy(n, row_start, _i(A[0]), _i(A[1]), _d(A[2]), alpha, nb[0], nb[1], nb[2], _i(rp), None, None, 0)
    col = np.zeros(max(nnz, 1), dtype=np.int32); val = np.zeros(max(nnz, 1))
    assert lib.ksc_csr_axpy(n, row_start, _i(A[0]), _i(A[1]), _d(A[2]), alpha, nb[0], nb[1], nb[2], _i(rp), _i(col), _d(val), nnz) == nnz
    return rp, col[:nnz], val[:nnz]


def _arrays(S):
    S = S.tocsr(); S.sort_indices()
    return S.indptr.astype(np.int32), S.indices.astype(np.int32), S.data.astype(np.float64)


@pytest.mark.parametrize("n,alpha", [(1, 2.0), (7, -0.5), (500, -38.25), (150000, 1.0 / 3.0)])
def test_axpy_union_pattern_entry_by_entry(lib, n, alpha):
    """Sorted rows: the union pattern, sorted; p_ij = a_ij + (alpha b_ij) to the bit, alpha b_ij / a_ij alone where only one has the entry.
    n = 150000 goes through the threaded path."""
    rng = np.random.default_rng(n)
    A = sp.random(n, n, density=min(1.0, 6.0 / n), random_state=rng, format="csr") + sp.identity(n) * 3.0
    B = sp.random(n, n, density=min(1.0, 3.0 / n), random_state=rng, format="csr") + sp.diags([np.full(max(n - 1, 0), 1 / 6)], [1], shape=(n, n))
    a, b = _arrays(A), _arrays(B)
    rp, col, val = axpy(lib, a, alpha, b)
    Ad = {}; Bd = {}
    for (arr, d) in ((a, Ad), (b, Bd)):
        rows = np.repeat(np.arange(n), np.diff(arr[0]))
        d.update(zip(zip(rows.tolist(), arr[1].tolist()), arr[2].tolist()))
    rows = np.repeat(np.arange(n), np.diff(rp))
    assert len(col) == len(set(Ad) | set(Bd))
    for r in range(n):
        assert np.all(np.diff(col[rp[r]:rp[r + 1]]) > 0)
    if n <= 500:
        for r, c, v in zip(rows.tolist(), col.tolist(), val.tolist()):
            ea, eb = Ad.get((r, c)), Bd.get((r, c))
            want = ea + alpha * eb if (ea is not None and eb is not None) else (ea if eb is None else alpha * eb)
            assert v == want, (r, c)
    else:
        P = sp.csr_matrix((val, col, rp), shape=(n, n))
        ref = (A + alpha * B).tocsr()
        assert abs(P - ref).max() <= 4e-16 * abs(ref).max()


def test_shift_is_axpy_with_the_identity(lib):
    """B NULL: MatShift - alpha on the diagonal, a missing diagonal entry inserted in its place; row_start moves the diagonal."""
    rp = np.array([0, 2, 3, 3, 5], dtype=np.int32)
    col = np.array([0, 2, 3, 1, 2], dtype=np.int32)          # row 1 and row 2 have no diagonal entry, row 3 has (3, 1), (3, 2)
    val = np.array([1.0, 2.0, 3.0, 4.0, 5.0])
    r2, c2, v2 = axpy(lib, (rp, col, val), -0.25, None)
    assert r2.tolist() == [0, 2, 4, 5, 8]
    assert c2.tolist() == [0, 2, 1, 3, 2, 1, 2, 3]
    assert v2.tolist() == [0.75, 2.0, -0.25, 3.0, -0.25, 4.0, 5.0, -0.25]
    # the same rows as rows 10..13 of a larger matrix: the diagonal is global column 10 + r
    colg = np.array([10, 12, 13, 1, 2], dtype=np.int32)
    r3, c3, v3 = axpy(lib, (rp, colg, val), -0.25, None, row_start=10)
    assert c3.tolist() == [10, 12, 11, 13, 12, 1, 2, 13] and v3.tolist() == [0.75, 2.0, -0.25, 3.0, -0.25, 4.0, 5.0, -0.25]


def test_rows_with_repeated_or_unordered_columns_keep_their_entries(lib):
    """ks_mat_create_csr takes rows with repeated / unordered columns (each entry a term of the row's sum); such a row keeps A's entries as
    they stand, B's go to the first entry with their column or to the end."""
    rp = np.array([0, 4, 6], dtype=np.int32)
    col = np.array([2, 0, 2, 1, 0, 1], dtype=np.int32)        # row 0: column 2 twice, unordered; row 1 sorted
    val = np.array([1.0, 2.0, 3.0, 4.0, 5.0, 6.0])
    brp = np.array([0, 2, 4], dtype=np.int32)
    bcol = np.array([2, 3, 1, 2], dtype=np.int32)
    bval = np.array([10.0, 20.0, 30.0, 40.0])
    r2, c2, v2 = axpy(lib, (rp, col, val), 0.5, (brp, bcol, bval))
    assert r2.tolist() == [0, 5, 8]
    assert c2.tolist() == [2, 0, 2, 1, 3, 0, 1, 2]
    assert v2.tolist() == [6.0, 2.0, 3.0, 4.0, 10.0, 5.0, 21.0, 20.0]
    x = np.array([1.0, -2.0, 3.0, 0.5])
    P = np.zeros((2, 4)); Ad = np.zeros((2, 4)); Bd = np.zeros((2, 4))
    for r in range(2):
        for k in range(r2[r], r2[r + 1]):
            P[r, c2[k]] += v2[k]
        for k in range(rp[r], rp[r + 1]):
            Ad[r, col[k]] += val[k]
        for k in range(brp[r], brp[r + 1]):
            Bd[r, bcol[k]] += bval[k]
    assert np.array_equal(P @ x, (Ad + 0.5 * Bd) @ x)


def test_empty_matrix_and_empty_rows(lib):
    rp = np.zeros(1, dtype=np.int32); e = np.zeros(1, dtype=np.int32); ev = np.zeros(1)
    r2, c2, v2 = axpy(lib, (rp, e, ev), 2.0, (rp, e, ev))
    assert r2.tolist() == [0] and len(c2) == 0
    rp = np.zeros(4, dtype=np.int32)
    brp = np.array([0, 0, 1, 1], dtype=np.int32); bc = np.array([0], dtype=np.int32); bv = np.array([7.0])
    r2, c2, v2 = axpy(lib, (rp, e, ev), 2.0, (brp, bc, bv))
    assert r2.tolist() == [0, 0, 1, 1] and c2.tolist() == [0] and v2.tolist() == [14.0]


def test_transpose_by_counting_sort(lib):
    """MatTranspose on the host: sorted rows out, repeated entries stay separate (their sum is what a product sees)."""
    lib.ksc_csr_transpose.argtypes = [C.c_int, IP, IP, DP, IP, IP, DP]
    lib.ksc_csr_transpose.restype = None
    rng = np.random.default_rng(3)
    for n in (1, 6, 300):
        A = sp.random(n, n, density=min(1.0, 5.0 / n), random_state=rng, format="csr") + sp.identity(n) * 2.0
        a = _arrays(A)
        nnz = len(a[1])
        rpt = np.zeros(n + 1, dtype=np.int32); ct = np.zeros(max(nnz, 1), dtype=np.int32); vt = np.zeros(max(nnz, 1))
        lib.ksc_csr_transpose(n, _i(a[0]), _i(a[1]), _d(a[2]), _i(rpt), _i(ct), _d(vt))
        T = A.T.tocsr(); T.sort_indices()
        assert np.array_equal(rpt, T.indptr) and np.array_equal(ct[:nnz], T.indices) and np.array_equal(vt[:nnz], T.data)
    # unordered row with a repeated column
    rp = np.array([0, 3, 4], dtype=np.int32); col = np.array([1, 0, 1, 0], dtype=np.int32); val = np.array([1.0, 2.0, 3.0, 4.0])
    rpt = np.zeros(3, dtype=np.int32); ct = np.zeros(4, dtype=np.int32); vt = np.zeros(4)
    lib.ksc_csr_transpose(2, _i(rp), _i(col), _d(val), _i(rpt), _i(ct), _d(vt))
    assert rpt.tolist() == [0, 2, 4] and ct.tolist() == [0, 1, 0, 0] and vt.tolist() == [2.0, 4.0, 1.0, 3.0]
