"""Spectral transformation (STSHIFT / STSINVERT) and the generalized non-symmetric problem of BASELINE config 5 on the
GPU versus the CPU oracle, whose linear solves are the reference's default (sparse LU).

The GPU path solves with GMRES + Jacobi (the reference's KSP for matrix mode "shell"); for parity with the LU-based
oracle the inner tolerance is tightened to 1e-13, which makes STApply agree to ~1e-12 and the Ritz values to 1e-10.
With the reference's default inner tolerance (1e-8) the results are checked through residuals instead."""
import numpy as np
import pytest

import golden_inputs as gi
import nhep_cases as nc
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _mat(ctx, Ao):
    import slepc_amd as ks
    return ks.Mat.from_csr(ctx, Ao.rowptr, Ao.col, Ao.val)


def test_get_diagonal_both_layouts(ctx, monkeypatch):
    import slepc_amd as ks
    Ao, Bo = nc.config5_pencil(1500)
    for fmt in ("csr", "sell"):
        monkeypatch.setenv("KSGPU_SPMV", fmt)
        assert np.array_equal(_mat(ctx, Ao).get_diagonal(), Ao.to_scipy().diagonal())
        assert np.array_equal(_mat(ctx, Bo).get_diagonal(), Bo.to_scipy().diagonal())
    L = ks.Mat.laplacian3d(ctx, 9, 8, 7)
    assert np.all(L.get_diagonal() == 6.0)
    # a row without a stored diagonal entry reports 0
    Z = ks.Mat.from_csr(ctx, [0, 1, 2], [1, 0], [3.0, 4.0])
    assert np.array_equal(Z.get_diagonal(), [0.0, 0.0])


def test_shell_matrix_drives_arnoldi_and_eps(ctx):
    """MATSHELL route (ex3.c): a callback that applies 2*A through the library's own SpMV gives 2x the eigenvalues."""
    import slepc_amd as ks
    Ao = O.laplacian2d(30)
    A = _mat(ctx, Ao)
    calls = []

    def mult(x, y):
        calls.append(1)
        A.mult_dev(x, y)
        ctx.L.ks_ctx_synchronize(ctx.h)

    S = ks.Mat.shell(ctx, Ao.n, mult)
    x = np.random.default_rng(0).standard_normal(Ao.n)
    assert np.allclose(S.mult(x), Ao.mult(x), rtol=0, atol=1e-13)
    out = []
    for M in (A, S):
        eps = ks.EPS(ctx)
        eps.SetOperators(M); eps.SetProblemType(ks.EPS_HEP); eps.SetDimensions(4, 20)
        eps.Solve()
        out.append(([eps.GetEigenvalue(i)[0] for i in range(4)], eps.GetIterationNumber(), eps.GetStats()))
    assert out[0][1] == out[1][1] and out[0][2] == out[1][2]                 # same restarts, steps and passes
    assert np.allclose(out[0][0], out[1][0], rtol=1e-13)
    assert len(calls) > out[1][2]["arnoldi_steps"]
    # an enqueue-only callback (no host synchronisation inside) keeps the whole run enqueued ahead: same numbers
    S2 = ks.Mat.shell(ctx, Ao.n, lambda x, y: A.mult_dev(x, y))
    S2.set_enqueue_only(True)
    eps = ks.EPS(ctx)
    eps.SetOperators(S2); eps.SetProblemType(ks.EPS_HEP); eps.SetDimensions(4, 20)
    eps.Solve()
    assert ([eps.GetEigenvalue(i)[0] for i in range(4)], eps.GetIterationNumber(), eps.GetStats()) == out[0]
    with pytest.raises(ks.KsError):
        A.set_enqueue_only(True)                                            # not a matrix-free operator


def test_eps_ex3_shell_golden(ctx):
    """ex3.c -eps_nev 4: the 72x72 2-D Laplacian applied by a user callback (matrix-free)."""
    import slepc_amd as ks
    Ao = O.laplacian2d(72)
    A = _mat(ctx, Ao)

    def mult(x, y):
        A.mult_dev(x, y)
        ctx.L.ks_ctx_synchronize(ctx.h)

    S = ks.Mat.shell(ctx, Ao.n, mult)
    eps = ks.EPS(ctx)
    eps.SetOperators(S); eps.SetProblemType(ks.EPS_HEP); eps.SetDimensions(4)
    eps.Solve()
    r = O.eps_krylovschur_hep(Ao, 4)
    lam = np.array([eps.GetEigenvalue(i)[0] for i in range(4)])
    assert eps.GetConverged() == r.nconv and eps.GetIterationNumber() == r.its
    assert np.allclose(np.round(lam, 5), gi.eigenvalues_line(gi.read("eps/ex3_1.out")), atol=1.5e-5)
    assert np.allclose(lam, r.eigr[r.perm][:4], rtol=1e-12)


def test_eps_ex24_spectrum_folding_golden(ctx):
    """ex24.c: the operator is a callback applying (A - target I)^2 with two products of the library's own SpMV."""
    import slepc_amd as ks
    import scenarios as sc
    Ao = O.laplacian2d(15)
    A = _mat(ctx, Ao)
    W = ks.BV(ctx, Ao.n, 1)

    def fold(x, y):                      # target = 0: y = A (A x)
        A.mult_dev(x, W.column_ptr(0))
        A.mult_dev(W.column_ptr(0), y)

    S = ks.Mat.shell(ctx, Ao.n, fold)
    S.set_enqueue_only(True)
    eps = ks.EPS(ctx)
    eps.SetOperators(S); eps.SetProblemType(ks.EPS_HEP); eps.SetWhichEigenpairs("smallest_real")
    eps.SetDimensions(1, 12); eps.SetTolerances(1e-5, 1000)
    eps.Solve()
    r = O.eps_krylovschur_hep(sc.folded_csr(Ao, 0.0), 1, ncv=12, max_it=1000, tol=1e-5, which="smallest_real")
    assert eps.GetConverged() >= 1 and eps.GetConvergedReason() > 0
    assert abs(eps.GetIterationNumber() - r.its) <= max(2, r.its // 10)      # two products vs the explicit square: rounding only
    x = eps.GetEigenvector(0)
    theta = float(x @ Ao.mult(x))
    assert abs(round(theta, 5) - gi.eigenvalues_after(gi.read("eps/ex24_1.out"), "required tolerance:")[0]) < 1.5e-5
    assert abs(eps.GetEigenvalue(0)[0] - r.eigr[r.perm][0]) < 1e-5 * abs(r.eigr[r.perm][0]) + 1e-9


@pytest.mark.parametrize("kind,withB,sigma", [("shift", False, 0.7), ("shift", True, 0.3), ("sinvert", False, 1.3), ("sinvert", True, 0.0), ("sinvert", True, 35.0)])
def test_st_apply_matches_oracle(ctx, kind, withB, sigma):
    import slepc_amd as ks
    Ao, Bo = nc.config5_pencil(3000)
    if not withB:
        Bo = None
    A = _mat(ctx, Ao); B = _mat(ctx, Bo) if withB else None
    st = ks.ST(ctx)
    st.SetType(kind); st.SetShift(sigma); st.SetMatrices(A, B); st.SetKSP(rtol=1e-14)
    st.SetUp()
    ost = O.ST(Ao, Bo, kind, sigma)
    x = np.random.default_rng(3).standard_normal(Ao.n)
    y = st.Apply(x); y0 = ost.apply(x)
    assert np.linalg.norm(y - y0) <= 1e-11 * np.linalg.norm(y0)
    stats = st.GetKSPStats()
    if kind == "sinvert" or withB:
        assert stats["solves"] == 1 and 0 < stats["iterations"] < 200
    else:
        assert stats["solves"] == 0
    for (re, im) in [(0.5, 0.0), (0.3, 0.2), (-2.0, -1.0)]:
        r, i = st.BackTransform(re, im)
        assert np.allclose((r[0], i[0]), ost.backtransform(re, im), rtol=1e-15, atol=0)


def test_gmres_restarts_and_reports_failure(ctx):
    """Small restart forces several GMRES cycles; an iteration cap makes the solve fail loudly (KSPSetErrorIfNotConverged)."""
    import slepc_amd as ks
    Ao, Bo = nc.config5_pencil(2000)
    A = _mat(ctx, Ao); B = _mat(ctx, Bo)
    st = ks.ST(ctx)
    st.SetType("sinvert"); st.SetShift(38.0); st.SetMatrices(A, B); st.SetKSP(rtol=1e-12, restart=5)
    x = np.random.default_rng(4).standard_normal(Ao.n)
    y = st.Apply(x)
    y0 = O.ST(Ao, Bo, "sinvert", 38.0).apply(x)
    assert st.GetKSPStats()["iterations"] > 5
    assert np.linalg.norm(y - y0) <= 1e-8 * np.linalg.norm(y0)
    st.SetKSP(max_it=3)
    with pytest.raises(ks.KsError) as e:
        st.Apply(x)
    assert e.value.rc == 91


def _solve_c5(ctx, Ao, Bo, nev, ncv, sigma, inner_rtol):
    import slepc_amd as ks
    A = _mat(ctx, Ao); B = _mat(ctx, Bo)
    eps = ks.EPS(ctx)
    eps.SetOperators(A, B); eps.SetProblemType(ks.EPS_GNHEP); eps.SetDimensions(nev, ncv)
    eps.SetTarget(sigma)
    st = eps.GetST(); st.SetType("sinvert")
    if inner_rtol:
        st.SetKSP(rtol=inner_rtol)
    eps.Solve()
    return eps, st


def test_config5_generalized_sinvert_vs_oracle(ctx):
    """BASELINE config 5 at n = 4000: random nonsymmetric A (32 nnz/row, diagonal + 40), tridiagonal B, shift-and-invert
    at sigma = 38 (inside the cluster of eigenvalues around 40*1.0), nev = 6, ncv = 24."""
    Ao, Bo = nc.config5_pencil(4000)
    sigma = 38.0
    eps, st = _solve_c5(ctx, Ao, Bo, 6, 24, sigma, 1e-14)
    r = O.eps_krylovschur_nhep(Ao, 6, ncv=24, which=O.which_target_magnitude(sigma), st=O.ST(Ao, Bo, "sinvert", sigma))
    assert eps.GetConverged() == r.nconv and eps.GetIterationNumber() == r.its
    assert eps.GetStats()["arnoldi_steps"] == r.steps == st.GetKSPStats()["solves"]
    Sa, Sb = Ao.to_scipy(), Bo.to_scipy()
    for i in range(r.nconv):
        kr, ki = eps.GetEigenvalue(i)
        j = r.perm[i]
        assert abs(kr - r.eigr[j]) <= 1e-10 * abs(complex(r.eigr[j], r.eigi[j]))
        assert abs(ki - r.eigi[j]) <= 1e-10 * abs(complex(r.eigr[j], r.eigi[j]))
        err = eps.ComputeError(i)
        assert abs(err - O.eps_compute_error_nhep(Ao, r, i, Bo)) < 1e-10
        _, _, xr, xi = eps.GetEigenpair(i)
        x = xr + 1j * xi; lam = complex(kr, ki)
        assert abs(np.linalg.norm(Sa @ x - lam * (Sb @ x)) / abs(lam) - err) < 1e-12      # the residual it reports is the true one
    lam = np.array([complex(*eps.GetEigenvalue(i)) for i in range(r.nconv)])
    assert np.all(np.diff(np.abs(lam - sigma)) >= -1e-9)                                 # closest to the target first
    k = 0
    while k < r.nconv:
        if lam[k].imag != 0:
            assert lam[k].imag > 0 and lam[k + 1] == np.conj(lam[k])                    # sign fix-up after the inversion
            k += 1
        k += 1


def test_config5_default_inner_tolerance(ctx):
    """With the reference's default KSP tolerance (1e-8) the eigenpairs still satisfy the outer tolerance up to the
    inner-solve error; eigenvalues agree with the LU-based oracle to ~1e-7."""
    Ao, Bo = nc.config5_pencil(3000, seed=43)
    sigma = 38.5
    eps, st = _solve_c5(ctx, Ao, Bo, 4, 20, sigma, 0.0)
    r = O.eps_krylovschur_nhep(Ao, 4, ncv=20, which=O.which_target_magnitude(sigma), st=O.ST(Ao, Bo, "sinvert", sigma))
    assert eps.GetConverged() >= 4 and eps.GetConvergedReason() > 0
    lam = np.array([complex(*eps.GetEigenvalue(i)) for i in range(4)])
    ref = (r.eigr + 1j * r.eigi)[r.perm][:4]
    assert np.allclose(lam, ref, rtol=1e-6)
    for i in range(4):
        assert eps.ComputeError(i) < 1e-5
    s = st.GetKSPStats()
    assert s["solves"] == eps.GetStats()["arnoldi_steps"] and s["iterations"] / s["solves"] < 400   # shift inside the spectrum: slow GMRES


def test_shift_with_hep_and_standard_sinvert(ctx):
    """STSHIFT sigma != 0 on a symmetric problem (Lanczos on A - sigma I) and sinvert on a standard problem."""
    import slepc_amd as ks
    Ao = O.laplacian2d(30)
    A = _mat(ctx, Ao)
    exact = np.sort(O.laplacian_eigenvalues([30, 30]))
    eps = ks.EPS(ctx)
    eps.SetOperators(A); eps.SetProblemType(ks.EPS_HEP); eps.SetDimensions(3, 20)
    st = eps.GetST(); st.SetType("shift"); st.SetShift(4.0)          # |lambda - 4| largest: both ends of the spectrum
    eps.Solve()
    r = O.eps_krylovschur_hep(Ao, 3, ncv=20)                          # unshifted reference for the largest ones
    lam = np.array([eps.GetEigenvalue(i)[0] for i in range(3)])
    for l in lam:
        assert np.min(np.abs(exact - l)) < 1e-9 and eps.ComputeError(0) < 1e-7
    assert np.all(np.diff(np.abs(lam)) <= 1e-12)                      # final sort on back-transformed values (largest magnitude)
    eps2 = ks.EPS(ctx)
    eps2.SetOperators(A); eps2.SetProblemType(ks.EPS_NHEP); eps2.SetDimensions(3, 16); eps2.SetTarget(-0.5)   # A + 0.5 I is definite: GMRES(30) converges
    st2 = eps2.GetST(); st2.SetType("sinvert"); st2.SetKSP(rtol=1e-13)
    eps2.Solve()
    lam2 = np.array([eps2.GetEigenvalue(i)[0] for i in range(3)])
    uniq = np.unique(np.round(exact, 12))                               # a single-vector Krylov method finds one copy of a multiple eigenvalue
    want = uniq[np.argsort(np.abs(uniq + 0.5))][:3]
    assert np.allclose(np.sort(lam2), np.sort(want), rtol=1e-9)
    assert st2.GetShift() == -0.5                                       # the shift defaults to the target


def test_sinvert_requires_target_which(ctx):
    import slepc_amd as ks
    Ao, Bo = nc.config5_pencil(500)
    eps = ks.EPS(ctx)
    eps.SetOperators(_mat(ctx, Ao), _mat(ctx, Bo)); eps.SetProblemType(ks.EPS_GNHEP)
    eps.SetWhichEigenpairs("largest_magnitude")
    eps.GetST().SetType("sinvert")
    with pytest.raises(ks.KsError) as e:
        eps.Solve()
    assert e.value.rc == 95


def test_config5_large_properties(ctx):
    """Config 5 shape at n = 10^5 (32 nnz/row, nev = 20, m = 60, generalized, sinvert), too large for the LU oracle:
    checked through size-independent properties - every returned pair satisfies A x = lambda B x to the tolerance, pairs
    are conjugate and adjacent, the order is by distance to the target, one linear solve per Arnoldi step. The target
    is 36, just outside the spectrum: with the target 0 of the config line the wanted eigenvalues are so clustered
    relative to their distance that Krylov-Schur needs far more restarts than a test should run (scripts/c5_probe.py
    measures the throughput of that set-up with a step cap)."""
    import time
    n, sigma = 100_000, 36.0
    Ao, Bo = nc.config5_pencil_fast(n)
    t0 = time.time()
    eps, st = _solve_c5(ctx, Ao, Bo, 20, 60, sigma, 0.0)
    dt = time.time() - t0
    nconv = eps.GetConverged()
    assert nconv >= 20 and eps.GetConvergedReason() == 1
    assert eps.GetDimensions() == (20, 60, 60)
    lam = np.array([complex(*eps.GetEigenvalue(i)) for i in range(nconv)])
    assert np.all(np.diff(np.abs(lam - sigma)) >= -1e-9)
    Sa, Sb = Ao.to_scipy(), Bo.to_scipy()
    for i in range(nconv):
        err = eps.ComputeError(i)
        assert err < 1e-7                                  # outer tol 1e-8 on the transformed problem, inner solves at 1e-8
        kr, ki, xr, xi = eps.GetEigenpair(i)
        x = xr + 1j * xi
        assert abs(np.linalg.norm(x) - 1) < 1e-12
        assert abs(np.linalg.norm(Sa @ x - complex(kr, ki) * (Sb @ x)) / abs(complex(kr, ki)) - err) < 1e-10
    k = 0
    while k < nconv:
        if lam[k].imag != 0:
            assert lam[k].imag > 0 and lam[k + 1] == np.conj(lam[k])
            k += 1
        k += 1
    s = st.GetKSPStats(); steps = eps.GetStats()["arnoldi_steps"]
    assert s["solves"] == steps
    print("config5 n=%d: %d steps, %d restarts, %.1f GMRES its/solve, %.2f s -> %.1f steps/s" % (n, steps, eps.GetIterationNumber(), s["iterations"] / s["solves"], dt, steps / dt))


def test_st_apply_on_binned_matrix(ctx):
    """The inner GMRES running on the binned SpMV layout (chosen automatically for this 12 MB wide-scatter matrix):
    the solve satisfies (A - sigma B) y = B x to the KSP tolerance, checked with host arithmetic."""
    import slepc_amd as ks
    Ao, Bo = nc.config5_pencil_fast(1_500_000, mean_nnz=12)
    A = _mat(ctx, Ao); B = _mat(ctx, Bo)
    assert A.layout() == "binned"
    st = ks.ST(ctx)
    st.SetType("sinvert"); st.SetShift(1.5); st.SetMatrices(A, B); st.SetKSP(rtol=1e-11)
    x = np.random.default_rng(6).standard_normal(Ao.n)
    y = st.Apply(x)
    Sa, Sb = Ao.to_scipy(), Bo.to_scipy()
    rhs = Sb @ x
    assert np.linalg.norm(Sa @ y - 1.5 * (Sb @ y) - rhs) <= 1e-9 * np.linalg.norm(rhs)
    assert 0 < st.GetKSPStats()["iterations"] < 60


def test_eps_test11_sinvert_indefinite_shift_golden(ctx):
    """test11 -eps_nev 4 -st_type sinvert: the shift 0.5 lies inside the spectrum of the Markov matrix, where GMRES(30)
    stagnates; with the restart length at n (full GMRES, a Krylov basis wider than the fused kernels) the inner solves
    converge and the golden values come out: 0.51928, 0.55740, 0.57028, 0.57143."""
    import slepc_amd as ks
    Ao = O.markov_matrix(15)
    eps = ks.EPS(ctx)
    eps.SetOperators(_mat(ctx, Ao)); eps.SetProblemType(ks.EPS_NHEP); eps.SetDimensions(4); eps.SetTolerances(1e-10, 0)
    eps.SetEigenvalueComparison(nc.right_of(0.5)); eps.SetInitialVector(np.ones(Ao.n))
    st = eps.GetST(); st.SetType("sinvert"); st.SetShift(0.5); st.SetKSP(rtol=1e-13, restart=Ao.n, max_it=10 * Ao.n)
    # full GMRES on an indefinite matrix to 1e-13: the unrefined Gram-Schmidt of the KSP's default loses the basis's orthogonality here (196 instead
    # of at most n = 120 iterations per solve, same eigenvalues): -st_ksp_gmres_cgs_refinement_type refine_ifneeded
    st.SetGMRESCGSRefinement("ifneeded")
    eps.Solve()
    r = O.eps_krylovschur_nhep(Ao, 4, tol=1e-10, which=nc.right_of(0.5), st=O.ST(Ao, None, "sinvert", 0.5), v0=np.ones(Ao.n))
    lam = np.array([eps.GetEigenvalue(i)[0] for i in range(4)])
    assert np.allclose(np.round(lam, 5), gi.eigenvalues_line(gi.read("eps/eps_test11_1.out")), atol=1.5e-5)
    assert eps.GetConverged() == r.nconv and eps.GetIterationNumber() == r.its
    assert np.allclose(lam, r.eigr[r.perm][:4], rtol=1e-9)
    assert st.GetKSPStats()["iterations"] / st.GetKSPStats()["solves"] <= Ao.n


def test_eps_test1_sinvert_target_22_golden(ctx):
    """test1_1_ks_sinvert: -st_type sinvert -eps_target 22 on the GHEP of test1 reprints test1_1.out; A - 22 B is
    indefinite, full GMRES (restart = n = 324) solves it."""
    import slepc_amd as ks
    from test_gpu_ghep import _test1_pencil
    Ao, Bo = _test1_pencil()
    eps = ks.EPS(ctx)
    eps.SetOperators(_mat(ctx, Ao), _mat(ctx, Bo)); eps.SetProblemType(ks.EPS_GHEP); eps.SetDimensions(4); eps.SetTolerances(0.0, 1500)
    eps.SetConvergenceTest("norm"); eps.SetTarget(22.0)
    st = eps.GetST(); st.SetType("sinvert"); st.SetKSP(rtol=1e-13, restart=Ao.n, max_it=10 * Ao.n)
    eps.Solve()
    lam = np.array([eps.GetEigenvalue(i)[0] for i in range(4)])
    r = O.eps_krylovschur_hep(Ao, 4, max_it=1500, which=O.which_target_magnitude(22.0), st=O.ST(Ao, Bo, "sinvert", 22.0), B=Bo, conv="norm")
    assert np.allclose(np.sort(np.round(lam, 5)), np.sort(gi.eigenvalues_line(gi.read("eps/eps_test1_1.out"))), atol=1.5e-5)
    assert np.allclose(lam, r.eigr[r.perm][:4], rtol=1e-8)
    for i in range(4):
        assert eps.ComputeError(i) < 1e-6


def test_eps_test1_cayley_golden(ctx):
    """test1_1_ks_cayley: -st_type cayley -eps_target 22 (GHEP): inner solves with A - 22 B (indefinite: full GMRES), the
    basis orthonormal in the A + 22 B inner product (a matrix-free operator), same values as test1_1.out and the oracle."""
    import slepc_amd as ks
    from test_gpu_ghep import _test1_pencil
    Ao, Bo = _test1_pencil()
    eps = ks.EPS(ctx)
    eps.SetOperators(_mat(ctx, Ao), _mat(ctx, Bo)); eps.SetProblemType(ks.EPS_GHEP); eps.SetDimensions(4); eps.SetTolerances(0.0, 1500)
    eps.SetConvergenceTest("norm"); eps.SetTarget(22.0)
    st = eps.GetST(); st.SetType("cayley"); st.SetKSP(rtol=1e-13, restart=Ao.n, max_it=10 * Ao.n)
    eps.Solve()
    assert st.GetShift() == 22.0 and st.CayleyGetAntishift() == 22.0       # both default to the target
    lam = np.array([eps.GetEigenvalue(i)[0] for i in range(eps.GetConverged())])
    r = O.eps_krylovschur_hep(Ao, 4, max_it=1500, which=O.which_target_magnitude(22.0), st=O.ST(Ao, Bo, "cayley", 22.0), B=Bo, conv="norm")
    assert np.allclose(np.round(lam[:4], 5), gi.eigenvalues_line(gi.read("eps/eps_test1_1.out")), atol=1.5e-5)
    assert eps.GetConverged() == r.nconv and eps.GetIterationNumber() == r.its
    assert np.allclose(lam, r.eigr[r.perm][: r.nconv], rtol=1e-8)
    for i in range(4):
        assert eps.ComputeError(i) < 1e-6


def test_cayley_standard_nonsymmetric_and_backtransform(ctx):
    """STCAYLEY on a standard non-symmetric problem with its own antishift, complex pairs through the Moebius map."""
    import slepc_amd as ks
    st = ks.ST(ctx); st.SetType("cayley"); st.SetShift(1.0); st.CayleySetAntishift(1.0)
    re, im = st.BackTransform(np.array([2.0, 3.0]), np.array([1.0, 0.0]))
    assert np.allclose(re, [2.0, 2.0]) and np.allclose(im, [-1.0, 0.0])  # (1 + theta)/(theta - 1)
    Ao = nc.planted_pairs(800)
    A = _mat(ctx, Ao)
    eps = ks.EPS(ctx)
    eps.SetOperators(A); eps.SetProblemType(ks.EPS_NHEP); eps.SetDimensions(4, 24); eps.SetTarget(-2.0)
    s2 = eps.GetST(); s2.SetType("cayley"); s2.CayleySetAntishift(5.0); s2.SetKSP(rtol=1e-13)
    eps.Solve()
    r = O.eps_krylovschur_nhep(Ao, 4, ncv=24, which=O.which_target_magnitude(-2.0), st=O.ST(Ao, None, "cayley", -2.0, nu=5.0))
    assert eps.GetConverged() == r.nconv and eps.GetIterationNumber() == r.its
    for i in range(4):
        kr, ki = eps.GetEigenvalue(i); j = r.perm[i]
        assert abs(kr - r.eigr[j]) < 1e-8 * np.hypot(r.eigr[j], r.eigi[j]) and abs(ki - r.eigi[j]) < 1e-8 * np.hypot(r.eigr[j], r.eigi[j])
        err = eps.ComputeError(i)                                       # convergence is tested on theta, not on lambda
        assert err < 1e-4 and abs(err - O.eps_compute_error_nhep(Ao, r, i)) < 1e-8
    with pytest.raises(ks.KsError) as e:
        s3 = ks.ST(ctx); s3.SetType("cayley"); s3.SetShift(2.0); s3.CayleySetAntishift(-2.0); s3.SetMatrices(A, None); s3.SetUp()
    assert e.value.rc == 95


@pytest.mark.parametrize("kind,withB,sigma", [("shift", True, 0.3), ("sinvert", False, 1.3), ("sinvert", True, 0.0), ("sinvert", True, 35.0), ("cayley", True, 35.0)])
def test_bicgstab_inner_solver_matches_oracle(ctx, kind, withB, sigma):
    """KSPBCGS as the ST's inner solver (left Jacobi): the operator application agrees with the LU-based oracle, the solver
    keeps 7 work vectors, and an iteration cap fails loudly."""
    import slepc_amd as ks
    Ao, Bo = nc.config5_pencil(3000)
    if not withB:
        Bo = None
    A = _mat(ctx, Ao); B = _mat(ctx, Bo) if withB else None
    st = ks.ST(ctx)
    st.SetType(kind); st.SetShift(sigma); st.SetMatrices(A, B); st.SetKSPType("bcgs"); st.SetKSP(rtol=1e-14)
    st.SetUp()
    x = np.random.default_rng(3).standard_normal(Ao.n)
    y = st.Apply(x); y0 = O.ST(Ao, Bo, kind, sigma).apply(x)
    assert np.linalg.norm(y - y0) <= 1e-10 * np.linalg.norm(y0)
    stats = st.GetKSPStats()
    assert stats["solves"] == 1 and 0 < stats["iterations"] < 200 and stats["last_rnorm"] < 1e-10
    st.SetKSP(max_it=2)
    with pytest.raises(ks.KsError) as e:
        st.Apply(x)
    assert e.value.rc == 91


def test_config5_with_bicgstab(ctx):
    """The config-5 problem (generalized, non-symmetric, shift-and-invert) with BiCGStab inner solves: same eigenvalues as
    with GMRES and as the LU-based oracle."""
    import slepc_amd as ks
    Ao, Bo = nc.config5_pencil(3000)
    sigma = 36.0
    lam = {}
    for ksp in ("gmres", "bcgs"):
        eps = ks.EPS(ctx)
        eps.SetOperators(_mat(ctx, Ao), _mat(ctx, Bo)); eps.SetProblemType(ks.EPS_GNHEP); eps.SetDimensions(6, 24); eps.SetTarget(sigma)
        st = eps.GetST(); st.SetType("sinvert"); st.SetKSPType(ksp); st.SetKSP(rtol=1e-13)
        eps.Solve()
        assert eps.GetConverged() >= 6
        lam[ksp] = np.array([complex(*eps.GetEigenvalue(i)) for i in range(6)])
    r = O.eps_krylovschur_nhep(Ao, 6, ncv=24, which=O.which_target_magnitude(sigma), st=O.ST(Ao, Bo, "sinvert", sigma))
    ref = np.array([complex(r.eigr[j], r.eigi[j]) for j in r.perm[:6]])
    assert np.allclose(lam["bcgs"], lam["gmres"], rtol=1e-9) and np.allclose(lam["bcgs"], ref, rtol=1e-9)


def test_config5_at_its_stated_size_step_capped(ctx):
    """BASELINE config 5 at its real n = 5 * 10^6 (1.65e8 nonzeros, binned layout, generalized shift-and-invert at the
    config's target 0, nev 20, m 60), capped at one full cycle plus a restart cycle. Too large for the LU oracle and, at
    target 0, far from converged after 90 steps, so the checks are size-independent properties: STApply satisfies
    (A - sigma B) y = B x to the inner tolerance on a random vector (host arithmetic), one linear solve per Arnoldi step at
    the iteration count the spectrum predicts (rho(D^-1 R) = 0.08: 8), the Arnoldi basis is orthonormal to rounding."""
    import slepc_amd as ks
    from slepc_amd.workloads import config5_pencil_arrays
    import scipy.sparse as sp
    n = 5_000_000
    (ar, ac, av), (br, bc, bv) = config5_pencil_arrays(n)
    A = ks.Mat.from_csr(ctx, ar, ac, av); B = ks.Mat.from_csr(ctx, br, bc, bv)
    assert A.layout() == "binned" and A.nnz == int(ar[-1])
    Sa = sp.csr_matrix((av, ac, ar), shape=(n, n)); Sb = sp.csr_matrix((bv, bc, br), shape=(n, n))
    st = ks.ST(ctx); st.SetType("sinvert"); st.SetShift(0.0); st.SetMatrices(A, B)
    x = np.random.default_rng(11).standard_normal(n)
    y = st.Apply(x)
    rhs = Sb @ x
    assert np.linalg.norm(Sa @ y - rhs) <= 1e-7 * np.linalg.norm(rhs)          # KSP rtol 1e-8 on the preconditioned residual
    del st, Sa, Sb
    eps = ks.EPS(ctx)
    eps.SetOperators(A, B); eps.SetProblemType(ks.EPS_GNHEP); eps.SetDimensions(20, 60); eps.SetTarget(0.0)
    s = eps.GetST(); s.SetType("sinvert")
    eps.SetMaxSteps(90)
    eps.Solve()
    stats = eps.GetStats(); k = s.GetKSPStats()
    assert stats["arnoldi_steps"] == 90 and stats["restarts"] == 2
    assert k["solves"] == 90 and 7.0 <= k["iterations"] / k["solves"] <= 9.0
    V = eps.GetBV()
    m = 31                                                                       # columns in use after the capped second cycle: check the leading block
    V.SetActiveColumns(0, m)
    M = np.zeros((m, m), order="F"); V.Dot(V, M)
    assert np.abs(M - np.eye(m)).max() < 1e-12


def test_config5_against_the_independent_dense_fixture(ctx):
    """The GPU's generalized shift-and-invert solve of the config-5-shaped pencil at n = 900 against the committed LAPACK (dense dggev) eigenvalues
    of the same pencil (tests/golden/c5/): 1e-9 relative, closest to the target first - a known answer that neither the oracle nor this library produced."""
    import json
    import os
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c5", "c5_n900_target36.json")))
    Ao, Bo = nc.config5_pencil(fx["n"])
    eps, st = _solve_c5(ctx, Ao, Bo, 6, 24, fx["target"], 1e-14)
    assert eps.GetConverged() >= 6
    got = np.array([complex(*eps.GetEigenvalue(i)) for i in range(6)])
    want = np.array([complex(*z) for z in fx["eigenvalues_by_distance_to_target"]])
    for k in range(6):
        assert np.abs(want[:8] - got[k]).min() <= 1e-9 * abs(got[k]), (k, got[k])
    assert np.all(np.diff(np.abs(got - fx["target"])) >= -1e-9)


# ---- ST_MATMODE_COPY: P = A - sigma B assembled (STSetMatMode; STMatMAXPY_Private stsolve.c:603-631) -------------------------------------
def _mat_keep(ctx, Ao):
    import slepc_amd as ks
    return ks.Mat.from_csr(ctx, Ao.rowptr, Ao.col, Ao.val, keep_csr=True)


@pytest.mark.parametrize("withB,sigma", [(True, 35.0), (False, 1.3), (True, 0.0)])
def test_matmode_copy_assembles_the_matrix_of_the_solves(ctx, withB, sigma):
    """MatAXPY on the device side of the C ABI equals A - sigma B entry by entry (product against scipy's sum, diagonal exactly the
    formula a_ii + (-sigma b_ii)); the solves of copy and shell mode agree with the LU oracle; copy mode takes one product of P per
    GMRES iteration where shell mode takes one of A and one of B."""
    import slepc_amd as ks
    Ao, Bo = nc.config5_pencil(3000)
    A = _mat_keep(ctx, Ao); B = _mat_keep(ctx, Bo) if withB else None
    P = A.axpy_new(-sigma, B)
    Sa = Ao.to_scipy(); Sb = Bo.to_scipy() if withB else __import__("scipy.sparse").sparse.identity(Ao.n, format="csr")
    x = np.random.default_rng(5).standard_normal(Ao.n)
    ref = (Sa - sigma * Sb) @ x
    assert np.linalg.norm(P.mult(x) - ref) <= 1e-14 * np.linalg.norm(ref)
    assert np.array_equal(P.get_diagonal(), Sa.diagonal() + (-sigma) * Sb.diagonal())
    assert P.nnz == (abs(Sa) + abs(Sb)).nnz
    ys = {}
    for mode in ("copy", "shell"):
        st = ks.ST(ctx)
        st.SetType("sinvert"); st.SetShift(sigma); st.SetMatrices(A, B); st.SetKSP(rtol=1e-14); st.SetMatMode(mode)
        assert st.GetMatMode() == mode
        ctx.prof_enable(True); ctx.prof_reset()
        ys[mode] = st.Apply(x)
        prof = ctx.prof_get()
        ctx.prof_enable(False)
        its = st.GetKSPStats()["iterations"]
        nprod = prof["spmv_csr"]["launches"]
        nB = 1 if withB else 0                              # M = B applied once before the solve
        if mode == "copy":
            assert its + nB <= nprod <= its + nB + 2, (nprod, its)      # the residual of the zero guess costs no product; one of P per iteration (+ one enqueued ahead)
        elif withB and sigma != 0.0:
            assert nprod >= 2 * its + nB
    y0 = O.ST(Ao, Bo if withB else None, "sinvert", sigma).apply(x)
    for mode in ys:
        assert np.linalg.norm(ys[mode] - y0) <= 1e-11 * np.linalg.norm(y0), mode


def test_matmode_copy_follows_the_shift_and_needs_the_csr_arrays(ctx):
    import slepc_amd as ks
    Ao, Bo = nc.config5_pencil(2000)
    A = _mat_keep(ctx, Ao); B = _mat_keep(ctx, Bo)
    x = np.random.default_rng(6).standard_normal(Ao.n)
    st = ks.ST(ctx)
    st.SetType("sinvert"); st.SetMatrices(A, B); st.SetKSP(rtol=1e-13); st.SetMatMode("copy")
    for sigma in (35.0, 36.5, 35.0):                         # STSetShift after STSetUp: P is assembled again (sinvert.c:121-141)
        st.SetShift(sigma)
        y = st.Apply(x); y0 = O.ST(Ao, Bo, "sinvert", sigma).apply(x)
        assert np.linalg.norm(y - y0) <= 1e-10 * np.linalg.norm(y0), sigma
    # cayley: P assembled, M = A + nu B term by term
    st.SetType("cayley"); st.SetShift(35.0); st.CayleySetAntishift(2.0)
    y = st.Apply(x); y0 = O.ST(Ao, Bo, "cayley", 35.0, nu=2.0).apply(x)
    assert np.linalg.norm(y - y0) <= 1e-10 * np.linalg.norm(y0)
    # back to shell mode: same operator
    st.SetType("sinvert"); st.SetMatMode("shell")
    y = st.Apply(x); y0 = O.ST(Ao, Bo, "sinvert", 35.0).apply(x)
    assert np.linalg.norm(y - y0) <= 1e-10 * np.linalg.norm(y0)
    # matrices created without KS_MAT_KEEP_CSR hold only the layout their product runs on
    A2 = _mat(ctx, Ao)
    st2 = ks.ST(ctx); st2.SetType("sinvert"); st2.SetShift(35.0); st2.SetMatrices(A2, B); st2.SetMatMode("copy")
    with pytest.raises(ks.KsError) as e:
        st2.SetUp()
    assert e.value.rc == 58                                   # PETSC_ERR_ORDER
    with pytest.raises(ks.KsError):
        st2.SetMatMode(1)                                     # ST_MATMODE_INPLACE is not built
    # MatLoad keeps the arrays it read
    Af = ks.Mat.load(ctx, gi.matrix_path("bfw62a.petsc")); Bf = ks.Mat.load(ctx, gi.matrix_path("bfw62b.petsc"))
    Pf = Af.axpy_new(-0.5, Bf)
    Sa = O.load_petsc_binary(gi.matrix_path("bfw62a.petsc")).to_scipy(); Sb = O.load_petsc_binary(gi.matrix_path("bfw62b.petsc")).to_scipy()
    xf = np.random.default_rng(7).standard_normal(Af.n)
    ref = (Sa - 0.5 * Sb) @ xf
    assert np.linalg.norm(Pf.mult(xf) - ref) <= 1e-14 * np.linalg.norm(ref)


def test_config5_in_copy_mode_matches_the_oracle(ctx):
    """The config-5 solve of test_config5_generalized_sinvert_vs_oracle with the matrix of the solves assembled (the reference's default
    matmode; its LU is the oracle's)."""
    import slepc_amd as ks
    Ao, Bo = nc.config5_pencil(4000)
    sigma = 38.0
    A = _mat_keep(ctx, Ao); B = _mat_keep(ctx, Bo)
    eps = ks.EPS(ctx)
    eps.SetOperators(A, B); eps.SetProblemType(ks.EPS_GNHEP); eps.SetDimensions(6, 24); eps.SetTarget(sigma)
    st = eps.GetST(); st.SetType("sinvert"); st.SetMatMode("copy"); st.SetKSP(rtol=1e-12)
    eps.Solve()
    r = O.eps_krylovschur_nhep(Ao, 6, ncv=24, which=O.which_target_magnitude(sigma), st=O.ST(Ao, Bo, "sinvert", sigma))
    assert eps.GetConverged() >= 6 and eps.GetIterationNumber() == r.its
    lam = np.array([complex(*eps.GetEigenvalue(i)) for i in range(6)])
    ref = (r.eigr + 1j * r.eigi)[r.perm][:6]
    assert np.allclose(lam, ref, rtol=1e-9, atol=0)


def test_gmres_refinement_types_give_the_same_solution(ctx):
    """KSPGMRESSetCGSRefinementType: never (the default, PETSc's) / ifneeded / always - same solution to the solver's tolerance, same
    iteration count on this well-conditioned preconditioned matrix; an unknown type is an error."""
    import slepc_amd as ks
    Ao, Bo = nc.config5_pencil(3000)
    A = _mat(ctx, Ao); B = _mat(ctx, Bo)
    x = np.random.default_rng(8).standard_normal(Ao.n)
    y0 = O.ST(Ao, Bo, "sinvert", 35.0).apply(x)
    its = {}
    for t in (None, "never", "ifneeded", "always"):
        st = ks.ST(ctx)
        st.SetType("sinvert"); st.SetShift(35.0); st.SetMatrices(A, B); st.SetKSP(rtol=1e-13)
        if t:
            st.SetGMRESCGSRefinement(t)
        y = st.Apply(x)
        assert np.linalg.norm(y - y0) <= 1e-10 * np.linalg.norm(y0), t
        its[t] = st.GetKSPStats()["iterations"]
    assert its[None] == its["never"] and abs(its["never"] - its["always"]) <= 1 and abs(its["ifneeded"] - its["always"]) <= 1
    with pytest.raises(ks.KsError):
        st.SetGMRESCGSRefinement(7)


# ---- PCBJACOBI on the ST's KSP (ks_st_set_pc): dense diagonal blocks of P, solved exactly -------------------------------------------------
def _line_pencil(nx, ny):
    """A = 2-D 5-point Laplacian with a convective term (non-symmetric), B = a diagonal mass matrix: P = A - sigma B couples a grid line
    (nx consecutive rows) strongly - the case block Jacobi with one block per line is made for."""
    import scipy.sparse as sp
    n = nx * ny
    T = sp.diags([np.full(nx - 1, -1.3), np.full(nx, 4.0), np.full(nx - 1, -0.7)], [-1, 0, 1])
    A = (sp.kron(sp.identity(ny), T) + sp.kron(sp.diags([np.full(ny - 1, -0.2), np.full(ny - 1, -0.2)], [-1, 1]), sp.identity(nx))).tocsr()
    B = sp.diags([1.0 + 0.1 * np.cos(np.arange(n))], [0]).tocsr()
    A.sort_indices(); B.sort_indices()
    return (O.CSR(n, A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)),
            O.CSR(n, B.indptr.astype(np.int32), B.indices.astype(np.int32), B.data.astype(np.float64)))


@pytest.mark.parametrize("ksp", ["gmres", "bcgs"])
def test_block_jacobi_preconditioner(ctx, ksp):
    """Solves with PCBJACOBI agree with the LU oracle; one block per grid line takes fewer iterations than point Jacobi; block sizes that do
    not divide n leave a short last block; both matrix modes give the same blocks."""
    import slepc_amd as ks
    nx, ny = 24, 30
    Ao, Bo = _line_pencil(nx, ny)
    A = _mat_keep(ctx, Ao); B = _mat_keep(ctx, Bo)
    sigma = -0.5
    x = np.random.default_rng(21).standard_normal(Ao.n)
    y0 = O.ST(Ao, Bo, "sinvert", sigma).apply(x)
    its = {}
    for label, pc, bs, mode in (("jacobi", "jacobi", 0, "shell"), ("line", "bjacobi", nx, "shell"), ("line_copy", "bjacobi", nx, "copy"),
                                ("bs7", "bjacobi", 7, "shell"), ("bs32", "bjacobi", 32, "copy"), ("bs2", "bjacobi", 2, "shell")):
        st = ks.ST(ctx)
        st.SetType("sinvert"); st.SetShift(sigma); st.SetMatrices(A, B); st.SetKSP(rtol=1e-12); st.SetKSPType(ksp); st.SetMatMode(mode); st.SetPC(pc, bs)
        y = st.Apply(x)
        assert np.linalg.norm(y - y0) <= 1e-9 * np.linalg.norm(y0), label
        its[label] = st.GetKSPStats()["iterations"]
    assert its["line"] < its["jacobi"] and its["line"] == its["line_copy"], its
    assert its["bs32"] <= its["bs7"] <= its["bs2"] <= its["jacobi"], its
    # shift with two matrices: P = B (diagonal here: every block inverse is exact, one iteration)
    st = ks.ST(ctx); st.SetType("shift"); st.SetShift(0.3); st.SetMatrices(A, B); st.SetKSP(rtol=1e-13); st.SetKSPType(ksp); st.SetPC("bjacobi", 8)
    y = st.Apply(x); y1 = O.ST(Ao, Bo, "shift", 0.3).apply(x)
    assert np.linalg.norm(y - y1) <= 1e-12 * np.linalg.norm(y1) and st.GetKSPStats()["iterations"] <= 2


def test_block_jacobi_errors_and_a_whole_solve(ctx):
    import slepc_amd as ks
    import scipy.sparse as sp
    Ao, Bo = _line_pencil(16, 20)
    A = _mat_keep(ctx, Ao); B = _mat_keep(ctx, Bo)
    st = ks.ST(ctx); st.SetType("sinvert"); st.SetShift(-0.5); st.SetMatrices(_mat(ctx, Ao), B); st.SetPC("bjacobi", 4)
    with pytest.raises(ks.KsError) as e:
        st.SetUp()
    assert e.value.rc == 58                                   # PETSC_ERR_ORDER: the blocks come from the kept CSR arrays
    for bad in (1, 33):
        with pytest.raises(ks.KsError) as e:
            st.SetPC("bjacobi", bad)
        assert e.value.rc == 63
    # a singular diagonal block: rows 0 and 1 of P equal inside the first 2 x 2 block
    S = sp.lil_matrix(Ao.to_scipy()); S[0, :] = 0.0; S[1, :] = 0.0; S[0, 0] = S[0, 1] = S[1, 0] = S[1, 1] = 1.0; S[0, 5] = 2.0; S[1, 6] = 3.0
    S = S.tocsr(); S.sort_indices()
    As = ks.Mat.from_csr(ctx, S.indptr.astype(np.int32), S.indices.astype(np.int32), S.data, keep_csr=True)
    st2 = ks.ST(ctx); st2.SetType("sinvert"); st2.SetShift(0.0); st2.SetMatrices(As); st2.SetPC("bjacobi", 2)
    with pytest.raises(ks.KsError) as e:
        st2.SetUp()
    assert e.value.rc == 71                                   # PETSC_ERR_MAT_LU_ZRPVT
    # the whole eigensolve through the block-preconditioned inner solves
    sigma = -0.5
    eps = ks.EPS(ctx); eps.SetOperators(A, B); eps.SetProblemType(ks.EPS_GNHEP); eps.SetDimensions(4, 16); eps.SetTarget(sigma)
    s3 = eps.GetST(); s3.SetType("sinvert"); s3.SetKSP(rtol=1e-12); s3.SetPC("bjacobi", 16)
    eps.Solve()
    r = O.eps_krylovschur_nhep(Ao, 4, ncv=16, which=O.which_target_magnitude(sigma), st=O.ST(Ao, Bo, "sinvert", sigma))
    assert eps.GetConverged() >= 4 and eps.GetIterationNumber() == r.its
    lam = np.array([complex(*eps.GetEigenvalue(i)) for i in range(4)])
    assert np.allclose(lam, (r.eigr + 1j * r.eigi)[r.perm][:4], rtol=1e-9, atol=0)
