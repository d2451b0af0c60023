"""Bases wider than the 64 columns of the register-tiled / MFMA panel kernels: the BV operations block over 64-column
panels, Gram-Schmidt runs its device-resident slot program over 64-column chunks, TSQR goes panel by panel, and the solver
accepts ncv + 1 > 64."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
EPS = np.finfo(float).eps


@pytest.mark.parametrize("n,mx,my", [(5000, 100, 70), (40000, 150, 130)])
def test_wide_bv_panel_operations(ctx, n, mx, my):
    import slepc_amd as ks
    rng = np.random.default_rng(n)
    X0 = rng.standard_normal((n, mx)); Y0 = rng.standard_normal((n, my))
    X = ks.BV(ctx, n, mx); Y = ks.BV(ctx, n, my)
    X.set_dense(X0); Y.set_dense(Y0)
    # BVDot: M = Y' X on 64 x 64 blocks
    X.SetActiveColumns(3, mx - 2); Y.SetActiveColumns(1, my)
    M = np.full((my, mx), 7.0, order="F")
    X.Dot(Y, M)
    want = Y0[:, 1:my].T @ X0[:, 3: mx - 2]
    assert np.allclose(M[1:my, 3: mx - 2], want, rtol=1e-12, atol=1e-10 * np.sqrt(n))
    G = np.zeros((mx, mx), order="F")
    X.SetActiveColumns(0, mx); X.Dot(X, G)
    assert np.allclose(G, X0.T @ X0, rtol=1e-12, atol=1e-10 * np.sqrt(n)) and np.allclose(G, G.T, rtol=0, atol=1e-9)
    # BVMult: Y = beta Y + alpha X Q with more than 64 inner and outer columns
    Q = np.asfortranarray(rng.standard_normal((mx, my)))
    Y.SetActiveColumns(0, my)
    Y.Mult(0.5, -2.0, X, Q)
    Y1 = -2.0 * Y0 + 0.5 * X0 @ Q
    assert np.allclose(Y.dense(), Y1, rtol=1e-12, atol=1e-11 * mx)
    # BVMultInPlace over a wide window, both forms
    Q2 = np.asfortranarray(rng.standard_normal((mx, mx)))
    X.SetActiveColumns(2, mx - 1)
    X.MultInPlace(Q2, 5, mx - 4)
    X1 = X0.copy(); X1[:, 5: mx - 4] = X0[:, 2: mx - 1] @ Q2[2: mx - 1, 5: mx - 4]
    assert np.allclose(X.dense(), X1, rtol=1e-12, atol=1e-11 * mx)
    X.set_dense(X0)
    X.MultInPlace(Q2, 5, mx - 4, trans=True)
    X2 = X0.copy(); X2[:, 5: mx - 4] = X0[:, 2: mx - 1] @ Q2[5: mx - 4, 2: mx - 1].T
    assert np.allclose(X.dense(), X2, rtol=1e-12, atol=1e-11 * mx)
    # DotVec / MultVec across chunks
    X.set_dense(X0); X.SetActiveColumns(0, mx)
    v = ks.BV(ctx, n, 1); v0 = rng.standard_normal(n); v.set_column(0, v0)
    d = X.DotVec(v.column_ptr(0))
    assert np.allclose(d, X0.T @ v0, rtol=1e-12, atol=1e-10 * np.sqrt(n))
    q = rng.standard_normal(mx)
    X.MultVec(1.5, 0.25, v.column_ptr(0), q)
    assert np.allclose(v.column(0), 0.25 * v0 + 1.5 * X0 @ q, rtol=1e-12, atol=1e-11 * mx)


@pytest.mark.parametrize("block", ["gs", "chol", "svqb", "tsqr", "tsqrchol"])
def test_wide_block_orthogonalization(ctx, block):
    import slepc_amd as ks
    n, k = 20000, 100
    X0 = np.random.default_rng(7).standard_normal((n, k))
    V = ks.BV(ctx, n, k); V.set_dense(X0); V.SetOrthogBlock(block)
    R = np.zeros((k, k), order="F")
    V.Orthogonalize(R)
    Q = V.dense()
    assert np.abs(Q.T @ Q - np.eye(k)).max() < 500 * EPS * np.sqrt(k)
    assert np.abs(X0 - Q @ R).max() < 1e4 * EPS * np.abs(X0).max() * np.sqrt(k)
    if block != "svqb":
        assert np.abs(np.tril(R, -1)).max() == 0.0             # TSQR runs panel by panel beyond 64 columns: R is still upper triangular


@pytest.mark.parametrize("ptype,nev,ncv", [("hep", 40, 100), ("nhep", 40, 100), ("hep", 70, 200), ("nhep", 70, 200)])
def test_solver_with_a_wide_basis(ctx, ptype, nev, ncv):
    """nev = 40, ncv = 100 (101 columns) and nev = 70, ncv = 200: restart products are blocked, Gram-Schmidt is host-driven;
    counts and values as the oracle."""
    import slepc_amd as ks
    import nhep_cases as nc
    if ptype == "hep":
        Ao = O.laplacian2d(41, 23)
        r = O.eps_krylovschur_hep(Ao, nev, ncv=ncv)
    else:
        Ao = nc.planted_pairs(1200)
        r = O.eps_krylovschur_nhep(Ao, nev, ncv=ncv)
    A = ks.Mat.from_csr(ctx, Ao.rowptr, Ao.col, Ao.val)
    eps = ks.EPS(ctx)
    eps.SetOperators(A); eps.SetProblemType(ks.EPS_HEP if ptype == "hep" else ks.EPS_NHEP); eps.SetDimensions(nev, ncv)
    eps.Solve()
    assert eps.GetConverged() == r.nconv >= nev and eps.GetIterationNumber() == r.its
    st = eps.GetStats()
    assert st["arnoldi_steps"] == r.steps
    for i in range(nev):
        kr, ki = eps.GetEigenvalue(i)
        j = r.perm[i]
        w = np.hypot(r.eigr[j], r.eigi[j] if ptype == "nhep" else 0.0)
        assert abs(kr - r.eigr[j]) <= 1e-9 * w and (ptype == "hep" or abs(ki - r.eigi[j]) <= 1e-9 * w)
        assert eps.ComputeError(i) < 1e-7


@pytest.mark.parametrize("withb", [False, True])
def test_wide_lanczos_is_enqueued_without_host_waits(ctx, withb):
    """More than 64 previous columns: the Gram-Schmidt slot program runs over 64-column chunks (chunked dot sweeps reduced into a
    device array, one bookkeeping launch, gated chunked updates, scaling as its own launch), still enqueued as a whole. The
    columns below 64 of the same run take the register-tiled path. Tridiagonal, pass counts and orthonormal basis as the
    oracle; two host waits for a run of 100 steps, also in a B-inner product."""
    import slepc_amd as ks
    Ao = O.laplacian2d(60, 50)
    A = ks.Mat.from_csr(ctx, Ao.rowptr, Ao.col, Ao.val)
    m = 100
    Vg = ks.BV(ctx, Ao.n, m + 1); Vo = O.BV(Ao.n, m + 1)
    Bo = None
    if withb:
        L = O.laplacian1d(Ao.n)
        Bo = O.CSR(Ao.n, L.rowptr, L.col, np.where(L.col == np.repeat(np.arange(Ao.n), np.diff(L.rowptr)), 4.0, 1.0))
        B = ks.Mat.from_csr(ctx, Bo.rowptr, Bo.col, Bo.val)
        Vg.SetMatrix(B); Vo.SetMatrix(Bo)
    for V in (Vg, Vo):
        V.SetRandomColumn(0)
        _, nrm, _ = V.OrthogonalizeColumn(0); V.ScaleColumn(0, 1.0 / nrm)
    Tg = np.zeros((m + 1, 3), order="F"); To = np.zeros((m + 1, 3), order="F")
    p0g, p0o = Vg.gs_passes()[0], Vo.passes_total()
    s0 = ctx.sync_count()
    rg = Vg.MatLanczos(A, Tg, 0, m)
    waits = ctx.sync_count() - s0
    ro = Vo.MatLanczos(Ao, To, 0, m)
    assert rg[0] == ro[0] == m and not rg[2]
    assert Vg.gs_passes()[0] - p0g == Vo.passes_total() - p0o
    assert np.abs(Tg - To).max() < 1e-9 and abs(rg[1] - ro[1]) < 1e-9
    Vd = Vg.dense()
    G = Vd.T @ (Bo.to_scipy() @ Vd) if withb else Vd.T @ Vd
    assert np.abs(G - np.eye(m + 1)).max() < 1e-11
    assert waits <= 3, waits
