"""More of the reference's EPS test programs run through the C ABI on the GPU, each against its golden output and the
CPU oracle: test16 (user convergence function), test20 (second solve with a larger subspace on the same solver),
test24 (exact eigenvectors as deflation space), test28 (second solve with a matrix of another size on the same solver)."""
import numpy as np
import pytest

import golden_inputs as gi
import scenarios as sc
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _csr(S):
    return O.CSR(S.shape[0], S.indptr.astype(np.int32), S.indices.astype(np.int32), S.data.astype(np.float64))


def _mat(ctx, S):
    import slepc_amd as ks
    return ks.Mat.from_csr(ctx, S.indptr, S.indices, S.data)


def _same_run(eps, r):
    assert eps.GetConverged() == r.nconv and eps.GetIterationNumber() == r.its and eps.GetConvergedReason() == r.reason
    st = eps.GetStats()
    assert st["arnoldi_steps"] == r.steps and st["gs_passes"] == r.passes
    lam = np.array([eps.GetEigenvalue(i)[0] for i in range(r.nconv)])
    assert np.allclose(lam, r.eigr[r.perm][: r.nconv], rtol=1e-10, atol=1e-13)
    return lam


def test_eps_test16_user_convergence_golden(ctx):
    import slepc_amd as ks
    S = sc.tridiag_csr(200, -1.0, -1e-3, -1.0)
    eps = ks.EPS(ctx)
    eps.SetOperators(_mat(ctx, S)); eps.SetProblemType(ks.EPS_HEP)
    eps.SetConvergenceTestFunction(sc.test16_converged)
    eps.SetDimensions(6, 24); eps.SetWhichEigenpairs("smallest_magnitude")
    eps.Solve()
    r = O.eps_krylovschur_hep(_csr(S), 6, ncv=24, which="smallest_magnitude", conv=sc.test16_converged)
    lam = _same_run(eps, r)
    assert np.allclose(np.round(lam[:6], 5), gi.eigenvalue_lines(gi.read("eps/eps_test16_1.out"))[0], atol=1.5e-5)


def test_eps_test20_changing_ncv_golden(ctx):
    import slepc_amd as ks
    S = sc.tridiag_csr(18, -1.0, 2.0, -1.0)
    tol = max(1000 * np.finfo(float).eps, 1e-9)
    ref = gi.eigenvalue_lines(gi.read("eps/eps_test20_1.out"))
    eps = ks.EPS(ctx)
    eps.SetOperators(_mat(ctx, S)); eps.SetProblemType(ks.EPS_HEP); eps.SetTolerances(tol, 1500); eps.SetWhichEigenpairs("smallest_real")
    eps.Solve()
    r = O.eps_krylovschur_hep(_csr(S), 1, tol=tol, max_it=1500, which="smallest_real")
    lam = _same_run(eps, r)
    assert abs(round(lam[0], 5) - ref[0][0]) < 1.5e-5
    nev, ncv, mpd = eps.GetDimensions()
    assert (nev, ncv) == (1, r.ncv)
    eps.SetDimensions(nev, ncv + 2)                                       # EPSSetDimensions(eps,nev,ncv+2,PETSC_DETERMINE)
    eps.Solve()
    r2 = O.eps_krylovschur_hep(_csr(S), 1, ncv=ncv + 2, tol=tol, max_it=1500, which="smallest_real")
    lam2 = _same_run(eps, r2)
    assert eps.GetDimensions()[1] == ncv + 2 and abs(round(lam2[0], 5) - ref[1][0]) < 1.5e-5


def test_eps_test24_exact_eigenvectors_deflated_golden(ctx):
    import slepc_amd as ks
    n = 30
    S = sc.tridiag_csr(n, -1.0, 2.0, -1.0)
    alpha, beta = np.pi / (n + 1), np.sqrt(2.0 / (n + 1))
    Cm = np.stack([np.sin(alpha * (np.arange(n) + 1) * (i + 1)) * beta for i in range(2)], axis=1)
    eps = ks.EPS(ctx)
    eps.SetOperators(_mat(ctx, S)); eps.SetProblemType(ks.EPS_HEP); eps.SetWhichEigenpairs("smallest_real")
    eps.SetConvergenceTest("abs"); eps.SetDimensions(4); eps.SetTolerances(1e-8, 1200)
    eps.SetDeflationSpace(Cm)
    eps.Solve()
    r = O.eps_krylovschur_hep(_csr(S), 4, tol=1e-8, max_it=1200, which="smallest_real", conv="abs", deflation=Cm)
    lam = _same_run(eps, r)
    assert np.allclose(np.round(lam[:4], 5), gi.eigenvalue_lines(gi.read("eps/eps_test24_1.out"))[0], atol=1.5e-5)
    X = np.stack([eps.GetEigenvector(i) for i in range(4)], axis=1)
    assert np.abs(Cm.T @ X).max() < 1e-12


def test_eps_test28_second_matrix_of_another_size_golden(ctx):
    """EPSSetOperators with a matrix of a different size resets the solver (epssetup.c:477: EPSReset)."""
    import slepc_amd as ks
    ref = gi.eigenvalue_lines(gi.read("eps/eps_test28_1.out"))
    eps = ks.EPS(ctx)
    for k, (n, m) in enumerate(((10, 11), (20, 22))):
        S = sc.laplacian2d_csr(n, m)
        eps.SetOperators(_mat(ctx, S))
        if k == 0:
            eps.SetProblemType(ks.EPS_HEP); eps.SetWhichEigenpairs("smallest_real"); eps.SetDimensions(3)
        eps.Solve()
        r = O.eps_krylovschur_hep(_csr(S), 3, which="smallest_real")
        lam = _same_run(eps, r)
        assert np.allclose(np.round(lam[:3], 5), ref[k], atol=1.5e-5)
        assert eps.GetEigenvector(0).shape == (n * m,) and eps.ComputeError(0) < 1e-7


def test_eps_test13_arbitrary_selection_golden(ctx):
    """test13: EPSSetArbitrarySelection on the second solve of the same solver object (the callback sees the Ritz vectors
    on the device; the Python mirror copies them to the host for the user's function)."""
    import slepc_amd as ks
    S = sc.tridiag_csr(30, -1.0, 0.0, -1.0)
    tol = 1000 * np.finfo(float).eps
    ref = gi.eigenvalue_lines(gi.read("eps/eps_test13_1.out"))
    eps = ks.EPS(ctx)
    eps.SetProblemType(ks.EPS_HEP); eps.SetTolerances(tol, 5000); eps.SetOperators(_mat(ctx, S)); eps.SetWhichEigenpairs("smallest_real")
    eps.Solve()
    r = O.eps_krylovschur_hep(_csr(S), 1, tol=tol, max_it=5000, which="smallest_real")
    lam = _same_run(eps, r)
    assert abs(round(lam[0], 5) - ref[0][0]) < 1.5e-5
    sx = eps.GetEigenvector(0)
    calls = []

    def pick(re, im, xr, xi):
        calls.append(re)
        assert not np.any(xi)
        return abs(xr @ sx), 0.0
    eps.SetArbitrarySelection(pick); eps.SetWhichEigenpairs("largest_magnitude")
    eps.Solve()
    so = np.array(r.V.column(r.perm[0]))
    r2 = O.eps_krylovschur_hep(_csr(S), 1, tol=tol, max_it=5000, which="largest_magnitude", arbitrary=lambda re, im, xr, xi: (abs(xr @ so), 0.0))
    lam2 = _same_run(eps, r2)
    assert abs(round(lam2[0], 5) - ref[1][0]) < 1.5e-5 and len(calls) > 0
    assert abs(abs(eps.GetEigenvector(0) @ sx) - 1.0) < 1e-6
    # the non-symmetric variant does not offer it
    eps.SetProblemType(ks.EPS_NHEP)
    with pytest.raises(ks.KsError) as e:
        eps.Solve()
    assert e.value.rc == 56
    eps.SetArbitrarySelection(None)
    eps.Solve()
    assert eps.GetConverged() >= 1


def test_eps_ex11_fiedler_restart_parameter_golden(ctx):
    """ex11 -eps_nev 4 -eps_krylovschur_restart .2 (EPSKrylovSchurSetRestart) with the deflation space of the example."""
    import slepc_amd as ks
    S = sc.graph_laplacian_2d(10, 10)
    eps = ks.EPS(ctx)
    eps.SetOperators(_mat(ctx, S)); eps.SetProblemType(ks.EPS_HEP); eps.SetWhichEigenpairs("smallest_real"); eps.SetDimensions(4)
    eps.KrylovSchurSetRestart(0.2); eps.SetDeflationSpace(np.ones((100, 1)))
    eps.Solve()
    r = O.eps_krylovschur_hep(_csr(S), 4, which="smallest_real", keep=0.2, deflation=np.ones((100, 1)))
    # the second copy of the double eigenvalue 0.09789 grows out of rounding noise: the restart in which it passes the test can
    # differ by one between two summation orders, the values cannot
    assert eps.GetConverged() >= 4 and abs(eps.GetIterationNumber() - r.its) <= 2
    lam = np.array([eps.GetEigenvalue(i)[0] for i in range(4)])
    assert np.allclose(lam, r.eigr[r.perm][:4], rtol=1e-7)
    assert np.allclose(np.round(lam[:4], 5), gi.eigenvalue_lines(gi.read("eps/ex11_1.out"))[0], atol=1.5e-5)
    with pytest.raises(ks.KsError):
        eps.KrylovSchurSetRestart(0.95)                                   # krylovschur.c:349: must be in [0.1, 0.9]


def test_no_device_memory_leak_over_many_solver_cycles():
    """scripts/leak_check.py: 40 create / solve / destroy cycles over the solver variants (deflation, wide basis, sinvert,
    Cayley, GHEP, block orthogonalisations) leave the free device memory where it was."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "leak_check.py")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "no leak" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_initial_space_of_device_vectors(ctx):
    """EPSSetInitialSpace: the first vector of the space is the start vector (epssolve.c:853); same run as SetInitialVector."""
    import slepc_amd as ks
    S = sc.laplacian2d_csr(13, 11)
    W = np.random.default_rng(5).standard_normal((S.shape[0], 3))
    out = []
    for how in ("vector", "space"):
        eps = ks.EPS(ctx)
        eps.SetOperators(_mat(ctx, S)); eps.SetProblemType(ks.EPS_HEP); eps.SetDimensions(3, 12)
        eps.SetInitialVector(W[:, 0]) if how == "vector" else eps.SetInitialSpace(W)
        eps.Solve()
        out.append((eps.GetIterationNumber(), [eps.GetEigenvalue(i)[0] for i in range(3)]))
    assert out[0] == out[1]
    r = O.eps_krylovschur_hep(_csr(S), 3, ncv=12, v0=W[:, 0])
    assert out[0][0] == r.its and np.allclose(out[0][1], r.eigr[r.perm][:3], rtol=1e-10)


@pytest.mark.parametrize("n", [1, 2, 3, 5, 8])
@pytest.mark.parametrize("ptype", ["hep", "nhep"])
def test_tiny_problems(ctx, n, ptype):
    """Problems smaller than the default subspace: ncv is clipped to n, the Krylov space is exhausted (breakdown path of
    BVMatLanczos / BV_OrthogonalizeColumn_Safe), all n eigenvalues come out."""
    import slepc_amd as ks
    rng = np.random.default_rng(n)
    M = rng.standard_normal((n, n))
    if ptype == "hep":
        M = M + M.T
    import scipy.sparse as sp
    S = sp.csr_matrix(M); S.sort_indices()
    eps = ks.EPS(ctx)
    eps.SetOperators(_mat(ctx, S)); eps.SetProblemType(ks.EPS_HEP if ptype == "hep" else ks.EPS_NHEP); eps.SetDimensions(1)
    eps.Solve()
    assert eps.GetConverged() >= 1 and eps.GetConvergedReason() > 0
    ev = np.linalg.eigvals(M)
    top = ev[np.argsort(-np.abs(ev))][0]
    kr, ki = eps.GetEigenvalue(0)
    assert abs(abs(complex(kr, ki)) - abs(top)) < 1e-9 * max(1.0, abs(top))
    assert eps.ComputeError(0) < 1e-7


def test_operator_of_another_size_drops_what_was_sized_by_the_old_one(ctx):
    """EPSSetOperators with a matrix of another size resets the solver (EPSReset): initial vector, deflation space and a
    user balancing matrix of the old size are forgotten instead of being read out of bounds."""
    import slepc_amd as ks
    S1, S2 = sc.laplacian2d_csr(9, 7), sc.laplacian2d_csr(15, 12)
    eps = ks.EPS(ctx)
    eps.SetOperators(_mat(ctx, S1)); eps.SetProblemType(ks.EPS_HEP); eps.SetDimensions(2)
    eps.SetInitialVector(np.ones(S1.shape[0])); eps.SetDeflationSpace(np.random.default_rng(0).standard_normal((S1.shape[0], 2)))
    eps.SetOperators(_mat(ctx, S2))
    eps.Solve()
    r = O.eps_krylovschur_hep(_csr(S2), 2)
    assert eps.GetIterationNumber() == r.its and np.allclose([eps.GetEigenvalue(i)[0] for i in range(2)], r.eigr[r.perm][:2], rtol=1e-10)
