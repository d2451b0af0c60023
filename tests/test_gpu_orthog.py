"""BVOrthogonalize (GS / CHOL / TSQR / TSQRCHOL / SVQB), BVMatProject and BVNormalize on the GPU: the reference's
test11 / test12 programs against their golden outputs and the CPU oracle, then size-independent properties
(orthogonality, V0 = Q R, triangularity, untouched leading columns) at sizes where the panel kernels run multi-block."""
import numpy as np
import pytest

import golden_inputs as gi
import scenarios as sc
from oracle import oracle as O

pytestmark = pytest.mark.gpu
EPS = np.finfo(float).eps
BLOCKS = ["gs", "chol", "tsqr", "tsqrchol", "svqb"]


@pytest.fixture(scope="module")
def gpu(ctx):
    return sc.GpuBackend(ctx)


@pytest.fixture(scope="module")
def cpu():
    return sc.OracleBackend()


@pytest.mark.parametrize("block", BLOCKS)
def test_bv_test11_golden_and_oracle(gpu, cpu, block):
    txt = gi.read("bv/test11_6.out")
    assert "Level of orthogonality of Q < 100*eps" in txt and "Residual ||X-Q*R|| < 100*eps" in txt
    a, b = sc.bv_test11(gpu, block), sc.bv_test11(cpu, block)
    for key in ("Q1", "Q2", "Q", "res1", "res"):
        assert a[key] < 100 * EPS, (block, key, a[key])
    if block != "svqb":
        assert np.all(np.tril(a["R"], -1) == 0)
    if block in ("gs", "chol"):                       # unique factorisation with a positive diagonal: compare entrywise
        assert np.allclose(a["R"], b["R"], rtol=0, atol=1e-13) and np.allclose(a["Y"], b["Y"], rtol=0, atol=1e-13)
    if block in ("tsqr", "tsqrchol"):                 # Householder R: rows defined up to sign
        assert np.allclose(np.abs(a["R"]), np.abs(b["R"]), rtol=0, atol=1e-13)
    if block == "svqb":                               # eigenvector signs are free; the Gram factor R'R is not
        assert np.allclose(a["R"][:, 2:].T @ a["R"][:, 2:], b["R"][:, 2:].T @ b["R"][:, 2:], rtol=0, atol=1e-12)


def test_bv_test12_dependent_columns(gpu, cpu):
    txt = gi.read("bv/test12_1.out")
    assert "Level of orthogonality < 100*eps" in txt and "Residual ||X-QR|| < 100*eps" in txt
    a = sc.bv_test12(gpu)
    assert a["level"] < 100 * EPS and a["res"] < 100 * EPS


@pytest.mark.parametrize("block", BLOCKS)
@pytest.mark.parametrize("n,l,k", [(5000, 0, 7), (100003, 3, 20), (1000000, 5, 37), (70000, 0, 64)])
def test_orthogonalize_properties(ctx, block, n, l, k):
    import slepc_amd as ks
    rng = np.random.default_rng(n + k)
    X0 = rng.standard_normal((n, k))
    X0[:, :l] = np.linalg.qr(X0[:, :l])[0] if l else X0[:, :l]        # leading columns come orthonormal
    V = ks.BV(ctx, n, k)
    V.set_dense(X0)
    V.SetActiveColumns(l, k); V.SetOrthogBlock(block)
    R = np.zeros((k, k), order="F")
    V.Orthogonalize(R)
    Q = V.dense()
    assert np.array_equal(Q[:, :l], X0[:, :l])                          # leading columns untouched
    G = Q.T @ Q
    assert np.abs(G - np.eye(k)).max() < 200 * EPS * np.sqrt(k)
    Rfull = R.copy(); Rfull[:l, :l] = np.eye(l)                        # leading block of R is not referenced
    assert np.abs(X0[:, l:] - Q @ Rfull[:, l:]).max() < 1e3 * EPS * np.abs(X0).max() * np.sqrt(k)
    if block != "svqb":
        assert np.all(np.tril(R, -1)[:, l:] == 0)
    if block in ("gs", "chol"):
        assert np.all(np.diag(R)[l:] > 0)
    V.Orthogonalize(None)                                              # R is optional; an orthonormal basis stays one
    assert np.abs(V.dense().T @ V.dense() - np.eye(k)).max() < 200 * EPS * np.sqrt(k)


@pytest.mark.parametrize("block", ["tsqr", "tsqrchol", "chol"])
def test_ill_conditioned_basis(ctx, block):
    """cond(V) = 1e6: TSQR keeps Q orthonormal to working precision (accumulated reflectors); V inv(R) from one pass
    (TSQRCHOL) or from the Gram matrix (CHOL, cond squared) loses orthogonality as the theory says."""
    import slepc_amd as ks
    n, k = 20000, 12
    rng = np.random.default_rng(5)
    U = np.linalg.qr(rng.standard_normal((n, k)))[0]; W = np.linalg.qr(rng.standard_normal((k, k)))[0]
    X0 = U @ np.diag(np.logspace(0, -6, k)) @ W.T
    V = ks.BV(ctx, n, k); V.set_dense(X0); V.SetOrthogBlock(block)
    R = np.zeros((k, k), order="F")
    V.Orthogonalize(R)
    Q = V.dense()
    lvl = np.abs(Q.T @ Q - np.eye(k)).max()
    assert np.abs(X0 - Q @ R).max() < 1e-13
    assert lvl < {"tsqr": 1e-13, "tsqrchol": 1e-8, "chol": 1e-2}[block]


@pytest.mark.parametrize("n,k", [(20000, 12), (100001, 40), (3, 3), (700, 64)])
def test_tsqr_accumulates_its_reflectors(ctx, n, k):
    """TSQR forms Q from the accumulated Householder reflectors, as the reference does (geqrf / orgqr per block and tree level,
    bvlapack.c:380-451): orthonormal to working precision whatever the conditioning - cond(V) = 1e12 here, where Q = V inv(R), even
    repeated, is no longer orthogonal - with V = Q R to rounding, also for a window l > 0 and sizes that leave partial tiles and blocks."""
    import slepc_amd as ks
    rng = np.random.default_rng(n + k)
    U = np.linalg.qr(rng.standard_normal((n, k)))[0]; W = np.linalg.qr(rng.standard_normal((k, k)))[0]
    X0 = U @ np.diag(np.logspace(0, -12, k)) @ W.T
    V = ks.BV(ctx, n, k); V.set_dense(X0); V.SetOrthogBlock("tsqr")
    R = np.zeros((k, k), order="F")
    V.Orthogonalize(R)
    Q = V.dense()
    assert np.abs(Q.T @ Q - np.eye(k)).max() < 1e-13
    assert np.abs(X0 - Q @ R).max() < 1e-13 and np.abs(np.tril(R, -1)).max() == 0.0
    if k >= 6:
        l = 3                                                   # leading columns stay, the rest is orthogonalised against them first
        Y0 = rng.standard_normal((n, k)); Y0[:, :l] = np.linalg.qr(Y0[:, :l])[0]
        V.set_dense(Y0); V.SetActiveColumns(l, k)
        R2 = np.zeros((k, k), order="F")
        V.Orthogonalize(R2)
        Q2 = V.dense()
        assert np.array_equal(Q2[:, :l], Y0[:, :l])
        assert np.abs(Q2.T @ Q2 - np.eye(k)).max() < 1e-13
        assert np.abs(Y0[:, l:] - Q2 @ R2[:, l:]).max() < 1e-12


def test_breakdown_is_an_error_for_gs(ctx):
    import slepc_amd as ks
    V = ks.BV(ctx, 50, 3)
    X0 = np.zeros((50, 3)); X0[0, 0] = 1.0; X0[1, 2] = 1.0           # column 1 is exactly zero
    V.set_dense(X0)
    with pytest.raises(ks.KsError) as e:
        V.Orthogonalize(None)
    assert e.value.rc == 82


def test_matproject_and_normalize(ctx):
    import slepc_amd as ks
    Ao = O.laplacian2d(40)
    A = ks.Mat.from_csr(ctx, Ao.rowptr, Ao.col, Ao.val)
    n = Ao.n
    rng = np.random.default_rng(9)
    X0 = rng.standard_normal((n, 9)); Y0 = rng.standard_normal((n, 6))
    X = ks.BV(ctx, n, 9); Y = ks.BV(ctx, n, 6); X.set_dense(X0); Y.set_dense(Y0)
    X.SetActiveColumns(2, 8); Y.SetActiveColumns(1, 5)
    M = np.full((6, 9), 7.0, order="F")
    X.MatProject(A, Y, M)
    S = Ao.to_scipy()
    want = Y0[:, 1:5].T @ (S @ X0[:, 2:8])
    assert np.allclose(M[1:5, 2:8], want, rtol=1e-12, atol=1e-11)
    M2 = M.copy(); M2[1:5, 2:8] = 7.0
    assert np.all(M2 == 7.0)                                           # nothing outside the active block is written
    X.MatProject(None, Y, M)
    assert np.allclose(M[1:5, 2:8], Y0[:, 1:5].T @ X0[:, 2:8], rtol=1e-12, atol=1e-11)
    # BVNormalize: plain, and pairwise for complex-conjugate pairs
    X.SetActiveColumns(0, 9); X.Normalize()
    assert np.allclose(np.linalg.norm(X.dense(), axis=0), 1.0, rtol=1e-14)
    X.set_dense(X0); X.SetActiveColumns(1, 6)
    X.Normalize(np.array([0.0, 0.3, -0.3, 0.0, 0.0]))
    D = X.dense()
    assert abs(np.linalg.norm(D[:, 1]) - 1) < 1e-14 and abs(np.linalg.norm(D[:, 4]) - 1) < 1e-14 and abs(np.linalg.norm(D[:, 5]) - 1) < 1e-14
    assert abs(np.hypot(np.linalg.norm(D[:, 2]), np.linalg.norm(D[:, 3])) - 1) < 1e-14
    assert np.array_equal(D[:, 0], X0[:, 0]) and np.array_equal(D[:, 6:], X0[:, 6:])


def test_bv_test9_matproject_golden(ctx):
    """test9.c: H0 = Y' G X by BVMatProject equals H1 = Y' Z with Z = G X stored by BVMatMult ("||H0-H1|| < 10*eps");
    integer data, so the two must agree exactly, and with a host evaluation."""
    import slepc_amd as ks
    assert "||H0-H1|| < 10*eps" in gi.read("bv/test9_1.out")
    n, kx, lx, ky, ly = 20, 6, 3, 5, 2
    rows, cols, vals = [], [], []
    for i in range(n):                                   # non-symmetric Toeplitz G: -1 1 1 1 1 on diagonals -1..3
        for d, v in zip(range(-1, 4), [-1.0, 1.0, 1.0, 1.0, 1.0]):
            if 0 <= i + d < n:
                rows.append(i); cols.append(i + d); vals.append(v)
    import scipy.sparse as sp
    S = sp.csr_matrix((vals, (rows, cols)), shape=(n, n)); S.sort_indices()
    G = ks.Mat.from_csr(ctx, S.indptr, S.indices, S.data)
    X0 = np.zeros((n, kx + 2))
    for j in range(kx + 2):
        for i in range(4):
            if i + j < n:
                X0[i + j, j] = 3 * i + j - 2
    Y0 = np.tile((np.arange(ky + 1) + 1) / 4.0, (n, 1))
    X = ks.BV(ctx, n, kx + 2); Z = ks.BV(ctx, n, kx + 2); Y = ks.BV(ctx, n, ky + 1)
    X.set_dense(X0); Y.set_dense(Y0)
    X.SetActiveColumns(0, kx); Z.SetActiveColumns(0, kx)
    X.MatMult(G, Z)
    X.SetActiveColumns(lx, kx); Z.SetActiveColumns(lx, kx); Y.SetActiveColumns(ly, ky)
    H0 = np.zeros((ky, kx), order="F"); H1 = np.zeros((ky, kx), order="F")
    X.MatProject(G, Y, H0)
    Z.MatProject(None, Y, H1)
    assert np.array_equal(H0, H1)
    want = np.zeros((ky, kx)); want[ly:ky, lx:kx] = Y0[:, ly:ky].T @ (S @ X0[:, lx:kx])
    assert np.array_equal(H0, want)


def test_bv_test18_normalize_golden(ctx):
    """test18.c parts 1 and 3 (the B-norm part needs BVSetMatrix, which is not built): 15 columns of length 250."""
    import slepc_amd as ks
    txt = gi.read("bv/test18_1.out")
    assert "Deviation from normalized vectors < 100*eps" in txt and "Deviation from normalized conjugate vectors < 100*eps" in txt
    n, k, l = 250, 15, 3
    X0 = sc._test11_X(n, k)
    X = ks.BV(ctx, n, k); X.set_dense(X0); X.SetActiveColumns(l, k)
    X.Normalize()
    assert max(abs(X.NormColumn(j) - 1.0) for j in range(l, k)) < 100 * EPS
    assert np.array_equal(X.dense()[:, :l], X0[:, :l])
    # conjugate pairs: eigi = (r, -r) for columns (l, l+1), (l+2, l+3), ... as the test draws them
    Z = ks.BV(ctx, n, k); Z.set_dense(X0); Z.SetActiveColumns(l, k)
    eigi = np.zeros(k - l)
    rng = np.random.default_rng(1)
    for j in range(0, k - l - 1, 2):
        eigi[j] = rng.uniform(0.1, 1.0); eigi[j + 1] = -eigi[j]
    Z.Normalize(eigi)
    err = 0.0
    for j in range(l, k - 1, 2):
        err = max(err, abs(np.hypot(Z.NormColumn(j), Z.NormColumn(j + 1)) - 1.0))
    assert err < 100 * EPS
