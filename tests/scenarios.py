"""The reference's BV test programs (src/sys/classes/bv/tests/test*.c) restated ONCE as backend-neutral
scenarios, so that the CPU oracle (tests/test_oracle_golden.py) and the HIP path (tests/test_gpu_*.py)
run literally the same sequence of BV calls and are compared with the same golden .out values."""
import numpy as np

import golden_inputs as gi


class OracleBackend:
    name = "oracle"

    def __init__(self):
        from oracle import oracle as O
        self.O = O

    def bv(self, n, m, ld=0):
        return self.O.BV(n, m, ld)

    def fill(self, bv, X):
        for j in range(X.shape[1]):
            bv.set_column(j, X[:, j])

    def vecref(self, bv, j):          # what MultVec/DotVec take as the Vec argument
        return bv.array[: bv.n, j]

    def new_vec(self, bv, x):
        return np.array(x, dtype=np.float64)

    def vec_to_host(self, v, n):
        return np.array(v[:n])

    def csr(self, rowptr, col, val):
        return self.O.CSR(len(rowptr) - 1, rowptr, col, val)


class GpuBackend:
    name = "gpu"

    def __init__(self, ctx):
        import slepc_amd as ks
        self.ks, self.ctx = ks, ctx
        self._scratch = []

    def bv(self, n, m, ld=0):
        return self.ks.BV(self.ctx, n, m, ld)

    def fill(self, bv, X):
        bv.set_dense(X)

    def vecref(self, bv, j):
        return bv.column_ptr(j)

    def new_vec(self, bv, x):
        w = self.ks.BV(self.ctx, bv.n, 1)      # a stand-alone device Vec
        w.set_column(0, x)
        self._scratch.append(w)
        return w.column_ptr(0)

    def vec_to_host(self, v, n):
        for w in self._scratch:
            if w.column_ptr(0) == v:
                return w.column(0)
        raise KeyError

    def csr(self, rowptr, col, val):
        return self.ks.Mat.from_csr(self.ctx, rowptr, col, val)


def bv_test1(be, testlda=False):
    """test1.c: BVMult, BVMultVec, BVDot, BVDotVec, BVMultInPlace, BVScale, BVNormColumn, BVNorm."""
    n, k, l = 10, 5, 3
    X = be.bv(n, k); Y = be.bv(n, l)
    be.fill(X, gi.test1_X()); be.fill(Y, gi.test1_Y())
    Q = gi.test1_Q()
    if testlda:                       # MatDenseSetLDA(Q,k+2)
        Qp = np.zeros((k + 2, l), order="F"); Qp[:k, :] = Q; Q = Qp
    out = {}
    Y.Mult(2.0, 1.0, X, Q); out["Mult"] = Y.dense()
    z = np.array([2.0 * (-0.5) ** i for i in range(k)])
    X.MultVec(-1.0, 1.0, be.vecref(Y, 0), z); out["MultVec"] = Y.dense()
    M = np.zeros((l + (2 if testlda else 0), k), order="F")
    X.Dot(Y, M); out["Dot"] = M[:l, :].copy()
    out["DotVec"] = np.array(X.DotVec(be.vecref(Y, 0)))
    X.MultInPlace(Q, 1, l); X.Scale(2.0); out["MultInPlace"] = X.dense()
    out["NormColumn0"] = X.NormColumn(0)
    out["NormF"] = X.Norm()
    out["FirstRow"] = X.dense()[0, :]
    return out


def bv_test2(be, orthog_type=0, refine=0, n=20, k=8):
    """test2.c: BVOrthogonalizeColumn loop, orthogonality level, BVOrthogonalizeVec of ones."""
    X = be.bv(n, k)
    X.SetOrthogonalization(orthog_type, refine, 0.7071)
    be.fill(X, gi.test2_X(n, k))
    norms = []
    for j in range(k):
        _, norm, _ = X.OrthogonalizeColumn(j)
        norms.append(norm)
        X.ScaleColumn(j, 1.0 / norm)
    M = np.zeros((k, k), order="F")
    X.Dot(X, M)
    level = np.abs(M - np.eye(k)).sum(axis=0).max()       # MatShift(-1); MatNorm(NORM_1)
    e = be.new_vec(X, np.ones(n))
    _, norm_e, _ = X.OrthogonalizeVec(e)
    return {"level": level, "norm_ones": norm_e, "norms": np.array(norms), "Xo": X.dense()}


def bv_test4(be, n=18, kx=12, lx=3, ky=8, ly=2, trans=False):
    """test4.c: the same ops on active/leading column windows."""
    X = be.bv(n, kx + 4)            # BVResize(X,kx+4,TRUE) applied up front (columns kx+2.. stay zero)
    Y = be.bv(n, ky + 1)
    X.SetActiveColumns(lx, kx); Y.SetActiveColumns(ly, ky)
    Xh = np.zeros((n, kx + 4))
    Xh[:, : kx + 2] = gi.test1_X(n, kx + 2)
    be.fill(X, Xh)
    be.fill(Y, gi.test1_Y(n, ky + 1))
    Q = np.array([[2.0 if i < j else -0.5 for j in range(ky)] for i in range(kx)], order="F")
    out = {}
    Y.Mult(2.0, 0.5, X, Q); out["Mult"] = Y.dense()
    z = np.array([2.0 * (-0.5) ** i for i in range(kx - lx)])
    X.MultVec(-1.0, 1.0, be.vecref(Y, 0), z); out["MultVec"] = Y.dense()
    M = np.zeros((ky, kx), order="F")
    X.Dot(Y, M); out["Dot"] = M.copy()
    out["DotVec"] = np.array(X.DotVec(be.vecref(Y, 0)))
    if trans:
        X.MultInPlace(np.asfortranarray(Q.T), lx + 1, ky, trans=True)
    else:
        X.MultInPlace(Q, lx + 1, ky)
    X.Scale(2.0)
    out["X"] = X.dense()
    out["NormColumn"] = X.NormColumn(lx)
    out["NormF"] = X.Norm()
    return out


def bv_test13(be):
    """test13.c: the NULL-array (buffer Vec) path used by Arnoldi: BVDotColumn(X,2,NULL); BVMultColumn(X,-1,1,2,NULL)."""
    n, k = 10, 5
    X = be.bv(n, k)
    be.fill(X, gi.test1_X(n, k))
    X.DotColumn(2, None)
    X.MultColumn(-1.0, 1.0, 2, None)
    return {"NormF": X.Norm(), "X": X.dense()}


def bv_test8(be, n=20, k=8, refine=0):
    """test8.c: MGS, BVOrthogonalizeSomeColumn against the odd columns."""
    X = be.bv(n, k)
    X.SetOrthogonalization(1, refine, 0.7071)
    be.fill(X, gi.test2_X(n, k))
    for j in range(k - 1):
        _, norm, _ = X.OrthogonalizeColumn(j)
        X.ScaleColumn(j, 1.0 / norm)
    which = [1 if i % 2 else 0 for i in range(k)]
    X.OrthogonalizeSomeColumn(k - 1, which)
    z = np.array(X.DotColumn(k - 1))
    z[np.abs(z) < 5.0 * np.finfo(float).eps] = 0.0
    return {"z": z}


def bv_test7(be, n=30, k=6):
    """test7.c: BVMatMult versus the column loop of MatMult."""
    from oracle import oracle as O
    A1 = O.laplacian1d(n)
    A = be.csr(A1.rowptr, A1.col, A1.val)
    V = be.bv(n, k); W = be.bv(n, k)
    rng = np.random.default_rng(7)
    Vh = rng.standard_normal((n, k))
    be.fill(V, Vh)
    V.MatMult(A, W)
    ref = A1.to_scipy() @ Vh
    return {"err": np.abs(W.dense() - ref).max()}


def _test11_X(n, k):
    X = np.zeros((n, k))
    for j in range(k):
        for i in range(n // 2 + 1):
            if i + j < n:
                X[i + j, j] = (3.0 * i + j - 2) / (2 * (i + j + 1))
    return X


def _orthogonalize(bv, R, block):
    """one call signature for both backends (the oracle takes the method per call, the library per BV)"""
    if hasattr(bv, "SetOrthogBlock"):
        bv.SetOrthogBlock(block); bv.Orthogonalize(R)
    else:
        bv.Orthogonalize(R, block)


def lap1d_csr(be, n):
    """tridiag(-1, 2, -1): the inner-product matrix B of test3.c / test11.c -withb / test18.c"""
    rowptr, col, val = [0], [], []
    for i in range(n):
        for j, v in ((i - 1, -1.0), (i, 2.0), (i + 1, -1.0)):
            if 0 <= j < n:
                col.append(j); val.append(v)
        rowptr.append(len(col))
    return be.csr(np.array(rowptr, dtype=np.int32), np.array(col, dtype=np.int32), np.array(val))


def bv_test3(be, orthog_type=0, n=10, k=5):
    """test3.c: B-norm, BVOrthogonalizeColumn and BVDot with the inner product of tridiag(-1,2,-1)."""
    B = lap1d_csr(be, n)
    X = be.bv(n, k)
    X.SetOrthogonalization(orthog_type, 0, 0.7071)
    X.SetMatrix(B)
    X0 = np.zeros((n, k))
    for j in range(k):
        for i in range(4):
            if i + j < n:
                X0[i + j, j] = 3 * i + j - 2
    be.fill(X, X0)
    out = {"norm0": X.NormColumn(0)}
    for j in range(k):
        _, nrm, _ = X.OrthogonalizeColumn(j)
        X.ScaleColumn(j, 1.0 / nrm)
    M = np.zeros((k, k), order="F")
    X.Dot(X, M)
    out["level"] = np.abs(M - np.eye(k)).sum(axis=0).max()
    out["norm0_after"] = X.NormColumn(0)
    out["X"] = X.dense()
    return out


def bv_test11(be, block, n=20, l=2, k=8, resid=True, X0=None, withb=False):
    """test11.c: BVOrthogonalize of the leading columns, then of the active ones; levels of orthogonality
    ||M(l:k,l:k) - I||_F (MyMatNorm) and residuals ||X - Q R||_F as the program prints them. withb: the B-inner product
    of tridiag(-1,2,-1) (output/test11_4.out, test11_9.out)."""
    X0 = _test11_X(n, k) if X0 is None else X0
    X = be.bv(n, k); Y = be.bv(n, k)
    be.fill(X, X0); be.fill(Y, X0)
    if withb:
        B = lap1d_csr(be, n)
        Y.SetMatrix(B)
    M = np.zeros((k, k), order="F")
    R = np.zeros((k, k), order="F") if resid else None
    out = {}

    def level(a, b):
        Y.Dot(Y, M)
        D = M[a:b, a:b] - np.eye(b - a)
        return np.sqrt((D * D).sum())

    if l > 0:
        Y.SetActiveColumns(0, l); X.SetActiveColumns(0, l)
        _orthogonalize(Y, R, block)
        out["Q1"] = level(0, l)
        if resid:
            out["res1"] = np.linalg.norm(X0[:, :l] - Y.dense()[:, :l] @ R[:l, :l])
    Y.SetActiveColumns(l, k); X.SetActiveColumns(l, k)
    _orthogonalize(Y, R, block)
    if l > 0:
        out["Q2"] = level(l, k)
    Y.SetActiveColumns(0, k); X.SetActiveColumns(0, k)
    out["Q"] = level(0, k)
    if resid:
        out["res"] = np.linalg.norm(X0 - Y.dense() @ R)
        out["R"] = R.copy()
    out["Y"] = Y.dense()
    return out


def bv_test12(be, block="gs", n=20, k=8):
    """test12.c: BVOrthogonalize of a basis with two linearly dependent columns (GS)."""
    X0 = np.zeros((n, k))
    full = _test11_X(n, k)
    j = 0
    while j < k // 2:
        X0[:, j] = full[:, j]; j += 1
    X0[:, j] = X0[:, 0] + 0.5 * X0[:, 1]; j += 1
    while j < k - 1:
        X0[:, j] = full[:, j]; j += 1
    X0[:, j] = X0[:, 1] - 1.2 * X0[:, k // 2 + 1]
    X = be.bv(n, k); be.fill(X, X0)
    R = np.zeros((k, k), order="F")
    _orthogonalize(X, R, block)
    M = np.eye(k, order="F") if True else None
    M = np.asfortranarray(M)
    X.Dot(X, M)
    level = np.abs(M - np.eye(k)).sum(axis=0).max()                   # MatShift(-1); MatNorm(NORM_1)
    res = np.linalg.norm(X0 - X.dense() @ R)
    return {"level": level, "res": res, "R": R.copy()}


def graph_laplacian_2d(n, m):
    """eps/tests/test10.c:42-52: Laplacian of the n x m mesh graph (degree on the diagonal, -1 per edge), II = i*n + j."""
    import scipy.sparse as sp
    rows, cols, vals = [], [], []
    for II in range(n * m):
        i, j = divmod(II, n)
        w = 0.0
        for ok, JJ in ((i > 0, II - n), (i < m - 1, II + n), (j > 0, II - 1), (j < n - 1, II + 1)):
            if ok:
                rows.append(II); cols.append(JJ); vals.append(-1.0); w += 1.0
        rows.append(II); cols.append(II); vals.append(w)
    S = sp.csr_matrix((vals, (rows, cols)), shape=(n * m, n * m)); S.sort_indices()
    return S


def bv_test6(be, orthog_type=0, n=20, k=8, nc=2, refine=0):
    """test6.c: BVInsertConstraints with nc staircase vectors, then the BVOrthogonalizeColumn loop on k columns;
    level of orthogonality ||X'X - I||_1, plus what the program does not print: the columns against the constraints."""
    X = be.bv(n, k)
    X.SetOrthogonalization(orthog_type, refine, 0.7071)
    Cm = np.zeros((n, nc))
    for j in range(nc):
        Cm[: j + 1, j] = 1.0
    kept = X.InsertConstraints(Cm)
    X0 = np.zeros((n, k))
    for j in range(k):
        for i in range(n // 2 + 1):
            if i + j < n:
                X0[i + j, j] = (3.0 * i + j - 2) / (2 * (i + j + 1))
    be.fill(X, X0)
    norms = []
    for j in range(k):
        _, norm, _ = X.OrthogonalizeColumn(j)
        norms.append(norm)
        X.ScaleColumn(j, 1.0 / norm)
    M = np.zeros((k, k), order="F")
    X.Dot(X, M)
    Cq = np.array(X.constraints_dense())[:n]
    Q = X.dense()[:n]
    return {"kept": kept, "level": np.abs(M - np.eye(k)).sum(axis=0).max(), "norms": np.array(norms), "X": Q, "C": Cq,
            "cross": np.abs(Cq.T @ Q).max(), "clevel": np.abs(Cq.T @ Cq - np.eye(kept)).max(), "buffer": np.array(X.buffer() if callable(X.buffer) else X.buffer)}


def tridiag_csr(n, sub, diag, sup):
    """Tridiagonal Toeplitz matrix as scipy CSR (sorted)."""
    import scipy.sparse as sp
    S = sp.diags([np.full(n - 1, sub), np.full(n, diag), np.full(n - 1, sup)], [-1, 0, 1], format="csr"); S.sort_indices()
    return S


def laplacian2d_csr(n, m):
    """eps/tests/test28.c:40-48: 5-point Laplacian on an n x m grid, II = i*n + j."""
    import scipy.sparse as sp
    S = (sp.kron(sp.identity(m), tridiag_csr(n, -1.0, 4.0, -1.0)) + sp.kron(tridiag_csr(m, -1.0, 0.0, -1.0), sp.identity(n))).tocsr()
    S.sort_indices()
    return S


def test16_converged(re, im, res):
    """eps/tests/test16.c:20-25 MyConvergedAbsolute"""
    return res if re < 0.0 else 100.0 * res


def test32_pencil(n):
    """eps/tests/test32.c:44-56: A = 5-point Laplacian on the n x n grid, B = diag(2 / log(II + 2)) with B(0,1) = B(1,0) = 0.4."""
    import scipy.sparse as sp
    N = n * n
    A = laplacian2d_csr(n, n)
    B = sp.diags(2.0 / np.log(np.arange(N) + 2.0)).tolil()
    B[0, 1] = 0.4; B[1, 0] = 0.4
    B = B.tocsr(); B.sort_indices()
    return A, B


def folded_csr(Ao, target=0.0):
    """ex24.c MatMult_Fold (ex24.c:192-205) as an explicit matrix: (A - target I)^2, for the CPU side of the comparison."""
    import scipy.sparse as sp
    from oracle import oracle as O
    S = Ao.to_scipy() - target * sp.identity(Ao.n, format="csr")
    F = (S @ S).tocsr(); F.sort_indices()
    return O.CSR(Ao.n, F.indptr, F.indices, F.data)
