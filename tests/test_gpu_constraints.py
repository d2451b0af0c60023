"""BV constraints (BVInsertConstraints / BVSetNumConstraints: columns -nc..-1 that every Gram-Schmidt sweep deflates)
and EPSSetDeflationSpace on the GPU: the reference's bv/test6 and eps/test10 programs against their golden outputs and
the CPU oracle, then the fused Lanczos/Arnoldi run with constraints at a size where the sweeps run multi-block."""
import numpy as np
import pytest

import golden_inputs as gi
import scenarios as sc
from oracle import oracle as O

pytestmark = pytest.mark.gpu
EPS = np.finfo(float).eps


@pytest.fixture(scope="module")
def gpu(ctx):
    return sc.GpuBackend(ctx)


@pytest.fixture(scope="module")
def cpu():
    return sc.OracleBackend()


@pytest.mark.parametrize("otype,refine", [(0, 0), (0, 1), (0, 2), (1, 0), (1, 2)])
def test_bv_test6_golden_and_oracle(gpu, cpu, otype, refine):
    txt = gi.read("bv/test6_1.out")
    assert "8 columns + 2 constraints, of length 20" in txt and "Level of orthogonality < 100*eps" in txt
    a, b = sc.bv_test6(gpu, otype, refine=refine), sc.bv_test6(cpu, otype, refine=refine)
    assert a["kept"] == 2 and a["level"] < 100 * EPS and a["cross"] < 100 * EPS and a["clevel"] < 100 * EPS
    assert np.allclose(a["C"], b["C"], rtol=0, atol=1e-15)
    assert np.allclose(a["X"], b["X"], rtol=0, atol=1e-13) and np.allclose(a["norms"], b["norms"], rtol=1e-13)
    # coefficient buffer: column j holds the nc + j coefficients and the norm (bvbasic.c:775-791), rows = nc + m
    assert a["buffer"].shape == b["buffer"].shape == (10, 8)
    for j in range(1, 8):
        assert np.allclose(a["buffer"][: 2 + j + 1, j], b["buffer"][: 2 + j + 1, j], rtol=0, atol=1e-13)


def test_insert_constraints_drops_dependent_vectors_and_errors(ctx):
    import slepc_amd as ks
    X = ks.BV(ctx, 12, 4)
    Cm = np.zeros((12, 3)); Cm[0, 0] = 2.0; Cm[0, 1] = -1.0; Cm[3, 2] = 1.0      # the second is a multiple of the first
    assert X.InsertConstraints(Cm) == 2 and X.nc == 2 and X.m == 4
    Cq = X.constraints_dense()
    assert np.allclose(np.abs(Cq[[0, 3], [0, 1]]), 1.0) and np.count_nonzero(Cq) == 2
    X.set_column(0, np.ones(12))
    _, nrm, lin = X.OrthogonalizeColumn(0)
    assert not lin and abs(nrm - np.sqrt(10.0)) < 1e-14
    with pytest.raises(ks.KsError) as e:
        X.InsertConstraints(Cm)                                          # "Constraints already present in this BV object"
    assert e.value.rc == 73
    with pytest.raises(ks.KsError) as e:
        X.Resize(9)                                                      # "Cannot resize a BV with constraints"
    assert e.value.rc == 73
    with pytest.raises(ks.KsError) as e:
        X.Orthogonalize(None)                                            # bvorthog.c:742
    assert e.value.rc == 56
    # a vector inside the constraint space is flagged linearly dependent
    v = np.zeros(12); v[0] = 3.0; v[3] = -2.0
    X.set_column(1, v)
    _, nrm, lin = X.OrthogonalizeColumn(1)
    assert lin
    X.SetNumConstraints(0)                                               # constraints discarded, regular columns keep their index
    assert X.m == 6 and X.nc == 0 and np.allclose(X.column(0), np.r_[0.0, 1, 1, 0, np.ones(8)])


def test_insert_vecs(ctx):
    """BVInsertVecs with orthogonalisation: columns s.. receive an orthonormal basis of span(W) against the leading ones."""
    import slepc_amd as ks
    n = 5000
    rng = np.random.default_rng(3)
    X = ks.BV(ctx, n, 8)
    Q0 = np.linalg.qr(rng.standard_normal((n, 2)))[0]
    X.set_column(0, Q0[:, 0]); X.set_column(1, Q0[:, 1])
    W = rng.standard_normal((n, 4))
    assert X.InsertVecs(2, W, orth=True) == 4
    Q = X.dense()[:, :6]
    assert np.abs(Q.T @ Q - np.eye(6)).max() < 50 * EPS
    assert np.linalg.matrix_rank(np.c_[Q, W, Q0]) == 6
    # a vector whose remainder is exactly zero is dropped and the next one takes its column (rounding noise left by an
    # inexact cancellation would instead be kept as a new direction - in the reference too: only nrm = 0 or a failed
    # eta test count as dependence, bvorthog.c:186)
    E = np.zeros((n, 3)); E[5, 0] = 1.0; E[5, 1] = 2.0; E[7, 2] = -4.0
    Z = ks.BV(ctx, n, 3)
    assert Z.InsertVecs(0, E, orth=True) == 2
    D = Z.dense()
    assert D[5, 0] == 1.0 and D[7, 1] == -1.0 and np.count_nonzero(D[:, :2]) == 2
    Y = ks.BV(ctx, n, 4)
    assert Y.InsertVecs(0, W, orth=False) == 4 and np.array_equal(Y.dense(), W)


@pytest.mark.parametrize("nc", [1, 3])
def test_lanczos_with_constraints_matches_oracle(ctx, nc):
    import slepc_amd as ks
    Ao = O.laplacian2d(60)
    n, m = Ao.n, 20
    rng = np.random.default_rng(nc)
    Cm = rng.standard_normal((n, nc))
    v0 = rng.standard_normal(n)
    A = ks.Mat.from_csr(ctx, Ao.rowptr, Ao.col, Ao.val)
    Vg = ks.BV(ctx, n, m + 1); Vo = O.BV(n, m + 1)
    out = []
    for V, Aop in ((Vg, A), (Vo, Ao)):
        assert V.InsertConstraints(Cm) == nc
        V.set_column(0, v0)
        nrm, lin = V.OrthonormalizeColumn(0)
        assert not lin
        T = np.zeros((m + 1, 3), order="F")
        mm, beta, brk = V.MatLanczos(Aop, T, 0, m)
        assert mm == m and not brk
        out.append((T[:m, :2].copy(), beta, np.array(V.dense())[:n], np.array(V.constraints_dense())[:n]))
    (Tg, bg, Xg, Cg), (To, bo, Xo, Co) = out
    assert np.allclose(Tg, To, rtol=1e-11, atol=1e-12) and abs(bg - bo) < 1e-11
    assert np.allclose(Xg, Xo, rtol=0, atol=1e-9)
    assert np.abs(Cg.T @ Xg).max() < 100 * EPS and np.abs(Xg.T @ Xg - np.eye(m + 1)).max() < 100 * EPS
    # the recurrence holds for the projected operator (I - C C') A
    P = np.eye(n) - Cg @ Cg.T
    S = Ao.to_scipy()
    Tm = np.diag(Tg[:, 0]) + np.diag(Tg[:-1, 1], 1) + np.diag(Tg[:-1, 1], -1)
    R = P @ (S @ Xg[:, :m]) - Xg[:, :m] @ Tm
    R[:, m - 1] -= bg * Xg[:, m]
    assert np.abs(R).max() < 1e-12


def test_arnoldi_with_constraints_large(ctx):
    """n = 1e6 (multi-block sweeps), 5 constraints + 25 columns: the whole fused run, then orthogonality against both
    sets and the Arnoldi relation on a sample of rows."""
    import slepc_amd as ks
    nx = 1000
    Ao = O.laplacian2d(nx)
    n, m, nc = Ao.n, 24, 5
    rng = np.random.default_rng(11)
    Cm = rng.standard_normal((n, nc))
    A = ks.Mat.from_csr(ctx, Ao.rowptr, Ao.col, Ao.val)
    V = ks.BV(ctx, n, m + 1)
    assert V.InsertConstraints(Cm) == nc
    V.set_column(0, rng.standard_normal(n))
    V.OrthonormalizeColumn(0)
    H = np.zeros((m + 1, m), order="F")
    mm, beta, brk = V.MatArnoldi(A, H, 0, m)
    assert mm == m and not brk and abs(H[m, m - 1] - beta) == 0.0
    X = V.dense(); Cq = V.constraints_dense()
    assert np.abs(Cq.T @ Cq - np.eye(nc)).max() < 100 * EPS
    assert np.abs(Cq.T @ X).max() < 100 * EPS and np.abs(X.T @ X - np.eye(m + 1)).max() < 100 * EPS
    AX = Ao.to_scipy() @ X[:, :m]
    R = AX - Cq @ (Cq.T @ AX) - X @ H
    assert np.abs(R).max() < 1e-12
    assert np.abs(np.tril(H[:m, :m], -2)).max() == 0.0


def test_fused_width_limit_counts_constraints(ctx):
    import slepc_amd as ks
    n = 300
    V = ks.BV(ctx, n, 60)
    rng = np.random.default_rng(0)
    assert V.InsertConstraints(rng.standard_normal((n, 6))) == 6       # 66 columns in all: host-driven sweeps take over
    V.set_dense(rng.standard_normal((n, 60)))
    for j in range(60):
        V.OrthonormalizeColumn(j)
    X = V.dense(); Cq = V.constraints_dense()
    assert np.abs(Cq.T @ X).max() < 100 * EPS and np.abs(X.T @ X - np.eye(60)).max() < 200 * EPS


def _solve(ctx, S, nev, defl, which="smallest_real", B=None, max_it=500, **kw):
    import slepc_amd as ks
    A = ks.Mat.from_csr(ctx, S.indptr, S.indices, S.data)
    eps = ks.EPS(ctx)
    eps.SetOperators(A, B); eps.SetProblemType(ks.EPS_GHEP if B is not None else ks.EPS_HEP)
    eps.SetWhichEigenpairs(which); eps.SetDimensions(nev, kw.get("ncv", 0)); eps.SetTolerances(kw.get("tol", 0.0), max_it)
    if defl is not None:
        eps.SetDeflationSpace(defl)
    return eps


def test_eps_test10_deflation_golden(ctx):
    """test10 -eps_nev 4 -m 11: the constant null vector of the 10x11 mesh-graph Laplacian deflated."""
    S = sc.graph_laplacian_2d(10, 11)
    n = S.shape[0]
    eps = _solve(ctx, S, 4, np.ones((n, 1)))
    eps.Solve()
    Ao = O.CSR(n, S.indptr, S.indices, S.data)
    r = O.eps_krylovschur_hep(Ao, 4, which="smallest_real", max_it=500, deflation=np.ones((n, 1)))
    lam = np.array([eps.GetEigenvalue(i)[0] for i in range(eps.GetConverged())])
    assert np.allclose(np.round(lam[:4], 5), gi.eigenvalues_line(gi.read("eps/eps_test10_1.out")), atol=1.5e-5)
    assert eps.GetConverged() == r.nconv and eps.GetIterationNumber() == r.its
    assert np.allclose(lam, r.eigr[r.perm][: r.nconv], rtol=1e-10)
    for i in range(4):
        x = eps.GetEigenvector(i)
        assert abs(x.sum()) < 1e-12 and eps.ComputeError(i) < 1e-7       # orthogonal to the deflated vector
    # "the deflation space should be set every time": the next solve sees the null vector again
    eps.Solve()
    assert abs(eps.GetEigenvalue(0)[0]) < 1e-10
    V = eps.GetBV()
    assert V.nc == 0


def test_deflation_peels_off_converged_pairs(ctx):
    """Solve, deflate the converged eigenvectors, solve again: the second solve returns the next eigenvalues. 2-D
    Laplacian on a 70 x 33 grid (simple eigenvalues), n = 2310."""
    nx, ny = 70, 33
    Ao = O.laplacian2d(nx, ny)
    S = Ao.to_scipy().tocsr(); S.sort_indices()
    ref = O.laplacian_eigenvalues([nx, ny])[::-1]
    eps = _solve(ctx, S, 3, None, which="largest_real")
    eps.Solve()
    k1 = eps.GetConverged()
    X1 = np.stack([eps.GetEigenvector(i) for i in range(k1)], axis=1)
    assert np.allclose([eps.GetEigenvalue(i)[0] for i in range(k1)], ref[:k1], rtol=1e-9)
    eps.SetDeflationSpace(X1 @ np.random.default_rng(2).standard_normal((k1, k1)))     # any basis of the space
    eps.Solve()
    k2 = eps.GetConverged()
    assert k2 >= 3
    assert np.allclose([eps.GetEigenvalue(i)[0] for i in range(k2)], ref[k1: k1 + k2], rtol=1e-9)
    X2 = np.stack([eps.GetEigenvector(i) for i in range(k2)], axis=1)
    assert np.abs(X1.T @ X2).max() < 1e-10


def test_ghep_deflation_in_the_b_inner_product(ctx):
    """GHEP: the constraints are orthonormalised in the B-inner product (host-driven sweeps), so deflating B-eigenvectors
    removes exactly those pairs."""
    import scipy.sparse as sp
    import scipy.linalg as sl
    import slepc_amd as ks
    Ao = O.laplacian2d(24, 17)
    n = Ao.n
    S = Ao.to_scipy().tocsr(); S.sort_indices()
    Bs = sp.diags([np.full(n - 1, 0.2), 1.0 + 0.5 * np.cos(np.arange(n)) ** 2, np.full(n - 1, 0.2)], [-1, 0, 1], format="csr"); Bs.sort_indices()
    lam, Z = sl.eigh(S.toarray(), Bs.toarray())
    B = ks.Mat.from_csr(ctx, Bs.indptr, Bs.indices, Bs.data)
    eps = _solve(ctx, S, 4, Z[:, -2:], which="largest_real", B=B)
    eps.GetST().SetKSP(rtol=1e-14)
    eps.Solve()
    assert eps.GetConverged() >= 4
    got = np.array([eps.GetEigenvalue(i)[0] for i in range(4)])
    assert np.allclose(got, lam[::-1][2:6], rtol=1e-8)


def test_deflation_space_wider_than_the_fused_kernels(ctx):
    """10 constraints + ncv + 1 = 57 columns exceed the 64 columns of the register-tiled kernels: the solve takes the
    host-driven Gram-Schmidt loop and still matches the oracle."""
    S = sc.graph_laplacian_2d(20, 20)
    Cm = np.random.default_rng(0).standard_normal((400, 10))
    eps = _solve(ctx, S, 4, Cm, ncv=56)
    eps.Solve()
    Ao = O.CSR(400, S.indptr, S.indices, S.data)
    r = O.eps_krylovschur_hep(Ao, 4, ncv=56, which="smallest_real", max_it=500, deflation=Cm)
    assert eps.GetConverged() == r.nconv and eps.GetIterationNumber() == r.its
    assert np.allclose([eps.GetEigenvalue(i)[0] for i in range(4)], r.eigr[r.perm][:4], rtol=1e-9, atol=1e-12)
