"""Parity of the HIP BV kernels (through the C ABI) with the CPU oracle and the reference's golden outputs.
Tolerances: the test1/test4/test13 inputs are small integers/dyadic rationals, so results are EXACT (bit
for bit); everything else is double arithmetic with a different summation order -> relative 1e-13."""
import numpy as np
import pytest

import golden_inputs as gi
import scenarios as sc
from test_oracle_golden import check_test1

pytestmark = pytest.mark.gpu

RTOL = 1e-13


@pytest.fixture(scope="module")
def gpu(ctx):
    return sc.GpuBackend(ctx)


@pytest.fixture(scope="module")
def cpu():
    return sc.OracleBackend()


def test_bv_test1_golden(gpu):
    check_test1(sc.bv_test1(gpu), gi.read("bv/test1_1_bv_type-mat.out"))


def test_bv_test1_bitexact_vs_oracle(gpu, cpu):
    for lda in (False, True):
        a, b = sc.bv_test1(gpu, testlda=lda), sc.bv_test1(cpu, testlda=lda)
        for k in b:
            if k in ("NormColumn0", "NormF"):
                assert abs(a[k] - b[k]) <= 4 * np.finfo(float).eps * b[k], k
            else:
                assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k


@pytest.mark.parametrize("otype", [0, 1])
@pytest.mark.parametrize("refine", [0, 1, 2])
def test_bv_test2(gpu, cpu, otype, refine):
    txt = gi.read("bv/test2_1.out")
    a, b = sc.bv_test2(gpu, otype, refine), sc.bv_test2(cpu, otype, refine)
    if refine != 1:
        assert a["level"] < 100 * np.finfo(float).eps                 # "Level of orthogonality < 100*eps"
    assert abs(a["norm_ones"] - gi.value_after(txt, "after orthogonalizing against X:")) < 5e-6
    assert np.allclose(a["norms"], b["norms"], rtol=1e-12)
    assert np.allclose(a["Xo"], b["Xo"], rtol=0, atol=1e-12)


@pytest.mark.parametrize("trans", [False, True])
def test_bv_test4_active_columns(gpu, cpu, trans):
    txt = gi.read("bv/test4_1.out")
    a, b = sc.bv_test4(gpu, trans=trans), sc.bv_test4(cpu, trans=trans)
    assert abs(a["NormColumn"] - gi.value_after(txt, "2-Norm of X[3] =")) < 5e-5
    assert abs(a["NormF"] - gi.value_after(txt, "Frobenius Norm of X =")) < 5e-4
    for k in ("Mult", "MultVec", "Dot", "DotVec", "X"):
        assert np.array_equal(a[k], b[k]), k                         # dyadic inputs: exact


def test_bv_test13_buffer_path(gpu, cpu):
    txt = gi.read("bv/test13_1.out")
    a, b = sc.bv_test13(gpu), sc.bv_test13(cpu)
    assert abs(a["NormF"] - gi.value_after(txt, "Frobenius Norm or X =")) < 5e-4
    assert np.array_equal(a["X"], b["X"])


def test_bv_test8_mgs_some_columns(gpu):
    txt = gi.read("bv/test8_1.out")
    ref = np.array([r[0] for r in gi.numeric_blocks(txt)[-1]])
    z = sc.bv_test8(gpu)["z"]
    assert np.allclose(z, ref, atol=5e-7)
    assert np.all(z[1::2] == 0.0)


def test_bv_test7_matmult(gpu):
    assert sc.bv_test7(gpu)["err"] < 1e-14


# ---- random panels, ragged sizes, odd leading dimensions -----------------------------------------------
@pytest.mark.parametrize("n,m,ld", [(1, 3, 0), (2, 2, 0), (63, 5, 0), (257, 9, 0), (1000, 31, 0), (4099, 33, 4101), (5000, 61, 0), (777, 7, 779)])
def test_ops_random(ctx, gpu, cpu, n, m, ld):
    rng = np.random.default_rng(n * 131 + m)
    Xh = rng.standard_normal((n, m)); Yh = rng.standard_normal((n, m))
    res = []
    for be in (gpu, cpu):
        X = be.bv(n, m, ld); Y = be.bv(n, m, ld)
        be.fill(X, Xh); be.fill(Y, Yh)
        l, k = (1, m - 1) if m > 3 else (0, m)
        X.SetActiveColumns(l, k); Y.SetActiveColumns(l, k)
        out = {}
        q = rng.standard_normal(k - l) if be is gpu else res[0]["q"]
        out["q"] = q
        out["dotvec"] = np.array(X.DotVec(be.vecref(Y, 0)))
        X.MultVec(0.75, -0.5, be.vecref(Y, m - 1), q); out["multvec"] = Y.dense()[:, m - 1].copy()
        X.MultVec(2.0, 0.0, be.vecref(Y, m - 1), q); out["multvec_beta0"] = Y.dense()[:, m - 1].copy()
        M = np.zeros((m, m), order="F"); X.Dot(Y, M); out["dot"] = M.copy()
        Q = rng.standard_normal((m, m)) if be is gpu else res[0]["Q"]
        out["Q"] = Q
        Y.Mult(1.5, 0.25, X, np.asfortranarray(Q)); out["mult"] = Y.dense()
        Y.Mult(-1.0, 2.0, X, None); out["axpy"] = Y.dense()
        X.MultInPlace(np.asfortranarray(Q), l, k); out["mip"] = X.dense()
        X.ScaleColumn(0, 3.0); X.Scale(-0.5); out["scale"] = X.dense()
        out["norms"] = np.array([X.Norm(2), X.Norm(0), X.Norm(3), X.NormColumn(0, 1), X.NormColumn(m - 1, 0), X.NormColumn(m - 1, 3)])
        X.CopyColumn(0, m - 1); Y.SetActiveColumns(l, k); X.Copy(Y); out["copy"] = np.hstack([X.dense(), Y.dense()])
        res.append(out)
    a, b = res
    scale = max(1.0, np.abs(b["dot"]).max())
    for key in b:
        if key in ("q", "Q"):
            continue
        tol = RTOL * max(1.0, np.abs(b[key]).max()) * (n ** 0.5)
        assert np.allclose(a[key], b[key], rtol=0, atol=tol), (key, np.abs(np.asarray(a[key]) - np.asarray(b[key])).max())


def test_alpha_zero_scale_and_nan_safety(ctx, gpu):
    """BVScale with alpha=0 zero-fills (bvblas.c:271) even over NaN; BVMultVec with beta=0 does not read y."""
    n, m = 300, 4
    X = gpu.bv(n, m)
    Xh = np.full((n, m), np.nan); Xh[:, 1:] = 1.0
    gpu.fill(X, Xh)
    X.ScaleColumn(0, 0.0)
    assert np.all(X.column(0) == 0.0)
    X.set_column(0, np.full(n, np.nan))
    X.SetActiveColumns(1, 4)
    X.MultVec(1.0, 0.0, X.column_ptr(0), np.array([1.0, 1.0, 1.0]))
    assert np.all(X.column(0) == 3.0)


def test_argument_errors_mirror_reference(ctx, gpu):
    import slepc_amd as ks
    X = gpu.bv(10, 4)
    with pytest.raises(ks.KsError) as e:
        X.OrthogonalizeColumn(4)                      # "Index j=4 but BV only has 4 columns"
    assert e.value.rc == 63
    with pytest.raises(ks.KsError) as e:
        X.OrthogonalizeColumn(-1)
    assert e.value.rc == 63
    with pytest.raises(ks.KsError) as e:
        X.Mult(1.0, 1.0, X, np.eye(4, order="F"))     # "X and Y arguments must be different"
    assert e.value.rc == 62
    with pytest.raises(ks.KsError) as e:
        X.MultInPlace(np.eye(3, order="F"), 0, 4)     # Mat has 3 rows, should have at least 4
    assert e.value.rc == 60
    with pytest.raises(ks.KsError) as e:
        ks.BV(ctx, 10, 3, ld=8)                       # leading dimension smaller than n
    assert e.value.rc == 95
    with pytest.raises(ks.KsError) as e:
        X.Norm(ks.NORM_2)                             # "Requested norm not available" for a whole BV
    assert e.value.rc == 56
    with pytest.raises(ks.KsError) as e:
        X.SetActiveColumns(3, 2)
    assert e.value.rc == 63


def test_empty_active_window_is_noop(ctx, gpu):
    X = gpu.bv(50, 4); Y = gpu.bv(50, 4)
    gpu.fill(X, np.ones((50, 4))); gpu.fill(Y, np.ones((50, 4)))
    X.SetActiveColumns(2, 2)
    M = np.full((4, 4), 7.0, order="F")
    X.Dot(Y, M)                                        # bvglobal.c:100: returns without touching M
    assert np.all(M == 7.0)
    X.Scale(5.0)
    assert np.all(X.dense() == 1.0)


@pytest.mark.parametrize("mfma", [True, False])
@pytest.mark.parametrize("n,my,nx", [(1, 1, 1), (130, 5, 3), (4097, 17, 31), (20000, 31, 31), (9000, 64, 33), (3000, 48, 64)])
def test_panel_contractions_mfma_and_valu(ctx, gpu, cpu, debug, mfma, n, my, nx):
    """BVDot / BVMult / BVMultInPlace on the FP64 matrix cores (v_mfma_f64_16x16x4) and on the VALU fallback
    against the oracle; tile edges (1..64 columns, ragged row tiles)."""
    if not mfma:
        debug("no_mfma")                                     # test hook: the VALU kernels the matrix-core ones replace
    rng = np.random.default_rng(n + my + nx)
    Xh = rng.standard_normal((n, nx)); Yh = rng.standard_normal((n, my))
    Q = np.asfortranarray(rng.standard_normal((nx, my)))
    QQ = np.asfortranarray(rng.standard_normal((nx, nx)))
    res = []
    for be in (gpu, cpu):
        X = be.bv(n, nx); Y = be.bv(n, my)
        be.fill(X, Xh); be.fill(Y, Yh)
        out = {}
        M = np.zeros((my, nx), order="F"); X.Dot(Y, M); out["dot"] = M.copy()
        Y.Mult(0.5, -1.5, X, Q); out["mult"] = Y.dense()
        Y.Mult(2.0, 0.0, X, Q); out["mult_beta0"] = Y.dense()
        s, e = (1, nx - 1) if nx > 2 else (0, nx)
        X.MultInPlace(QQ, s, e); out["mip"] = X.dense()
        X.MultInPlace(QQ, 0, nx, trans=True); out["mip_t"] = X.dense()
        res.append(out)
    a, b = res
    for key in b:
        tol = 1e-13 * max(1.0, np.abs(b[key]).max()) * max(1.0, np.sqrt(n))
        assert np.allclose(a[key], b[key], rtol=0, atol=tol), (key, np.abs(a[key] - b[key]).max())


def test_panel_dot_exact_on_integers(ctx, gpu):
    """A = I check with an asymmetric operand: catches a transposed or mis-mapped MFMA accumulator layout."""
    n, k = 64, 33
    Xh = np.zeros((n, k)); Yh = np.zeros((n, k))
    for j in range(k):
        Yh[j, j] = 1.0
        Xh[:k, j] = np.arange(k) * 3 + 7 * j          # asymmetric integer block
    X = gpu.bv(n, k); Y = gpu.bv(n, k)
    gpu.fill(X, Xh); gpu.fill(Y, Yh)
    M = np.zeros((k, k), order="F"); X.Dot(Y, M)
    assert np.array_equal(M, Xh[:k, :k])


def test_resize_setrandom_insert_copy_vec(ctx):
    """BVResize (test4.c grows X by four columns with copy=TRUE), BVSetRandom on the active window, BVInsertVec / BVCopyVec."""
    import slepc_amd as ks
    n = 3001
    X0 = np.random.default_rng(3).standard_normal((n, 5))
    X = ks.BV(ctx, n, 5); X.set_dense(X0)
    X.SetOrthogonalization(ks.MGS, ks.REFINE_ALWAYS, 0.5)
    X.Resize(9, copy=True)
    D = X.dense()
    assert D.shape == (n, 9) and np.array_equal(D[:, :5], X0) and np.all(D[:, 5:] == 0.0)
    X.Resize(3, copy=True)
    assert np.array_equal(X.dense(), X0[:, :3])
    X.Resize(6, copy=False)
    assert X.dense().shape == (n, 6)
    # the resized basis still works with every kernel sized by m (buffer, records, coefficient staging)
    X.set_dense(np.random.default_rng(4).standard_normal((n, 6)))
    for j in range(6):
        nrm, _ = X.OrthonormalizeColumn(j)
    Q = X.dense()
    assert np.abs(Q.T @ Q - np.eye(6)).max() < 1e-13
    # BVSetRandom touches the active columns only and is reproducible
    X.SetActiveColumns(2, 5); X.SetRandom(77)
    D2 = X.dense()
    assert np.array_equal(D2[:, :2], Q[:, :2]) and np.array_equal(D2[:, 5], Q[:, 5])
    Y = ks.BV(ctx, n, 6); Y.SetActiveColumns(2, 5); Y.SetRandom(77)
    assert np.array_equal(Y.dense()[:, 2:5], D2[:, 2:5]) and np.all((D2[:, 2:5] >= 0) & (D2[:, 2:5] < 1))
    # BVInsertVec / BVCopyVec
    W = ks.BV(ctx, n, 2); w = np.arange(n, dtype=float); W.set_column(0, w)
    X.InsertVec(4, W.column_ptr(0)); assert np.array_equal(X.column(4), w)
    X.CopyVec(0, W.column_ptr(1)); assert np.array_equal(W.column(1), Q[:, 0])
    with pytest.raises(ks.KsError) as e:
        X.Resize(0)
    assert e.value.rc == 63


def test_bv_test10_split_reductions_golden(ctx):
    """test10.c: BVDotVec / BVDotColumn / BVNormVec / BVNormColumn through their Begin/End forms give exactly what the plain
    calls give (the program prints the 1-norm of the difference: 0). Here also: Ends interleaved after several Begins, the
    B-inner product, and the order check."""
    import slepc_amd as ks
    assert "BV split ops (5 columns of dimension 10)" in gi.read("bv/test10_1.out") and gi.read("bv/test10_1.out").strip().endswith("0.")
    n, k = 10, 5
    X0 = np.zeros((n, k))
    for j in range(k):
        for i in range(4):
            if i + j < n:
                X0[i + j, j] = 3 * i + j - 2
    X = ks.BV(ctx, n, k); X.set_dense(X0)
    v = ks.BV(ctx, n, 1); v.set_column(0, np.ones(n))
    vp = v.column_ptr(0)
    z = np.concatenate([X.DotVec(vp), X.DotColumn(2), [X.NormVec(vp), X.NormColumn(0), X.NormColumn(1)]])
    a = X.DotVecBegin(vp); b = X.DotColumnBegin(2)
    X.NormVecBegin(vp); X.NormColumnBegin(0); X.NormColumnBegin(1)
    zs = np.concatenate([X.DotVecEnd(vp, a), X.DotColumnEnd(2, b), [X.NormVecEnd(vp), X.NormColumnEnd(0), X.NormColumnEnd(1)]])
    assert np.abs(z - zs).sum() == 0.0
    assert np.allclose(z[:k], X0.T @ np.ones(n)) and np.allclose(z[k: k + 2], X0[:, :2].T @ X0[:, 2])
    # a second round on the same context, with the inner product of a matrix
    B = sc.lap1d_csr(sc.GpuBackend(ctx), n)
    X.SetMatrix(B)
    n0 = X.NormColumn(0)
    X.NormColumnBegin(0); a = X.DotVecBegin(vp)
    assert X.NormColumnEnd(0) == n0 and np.array_equal(X.DotVecEnd(vp, a), X.DotVec(vp))
    X.SetMatrix(None)
    # Ends must come in the order of the Begins
    a = X.DotVecBegin(vp); X.NormColumnBegin(0)
    with pytest.raises(ks.KsError) as e:
        X.NormColumnEnd(0)
    assert e.value.rc == 58
    X.DotVecEnd(vp, a); X.NormColumnEnd(0)
    with pytest.raises(ks.KsError):
        X.NormColumnEnd(0)                                      # nothing pending


def test_norm_is_overflow_and_underflow_safe(ctx):
    """BVNormColumn / BVNorm(Frobenius) of entries around 1e+-200: the sum of squares leaves the double range, the norm does not
    (the reference scales inside lange and combines ranks with hypot, bvlapack.c:20-32,60-75)."""
    import slepc_amd as ks
    n = 70001
    x = np.random.default_rng(2).standard_normal(n)
    X = ks.BV(ctx, n, 3)
    X.set_column(0, 1e200 * x); X.set_column(1, 1e-200 * x); X.set_column(2, np.zeros(n))
    ref = np.linalg.norm(x)
    assert abs(X.NormColumn(0) / 1e200 / ref - 1.0) < 1e-13
    assert abs(X.NormColumn(1) / 1e-200 / ref - 1.0) < 1e-13
    assert X.NormColumn(2) == 0.0
    X.SetActiveColumns(0, 1)
    assert abs(X.Norm(ks.NORM_FROBENIUS) / 1e200 / ref - 1.0) < 1e-13
    assert abs(X.Norm(ks.NORM_INFINITY) - 1e200 * np.abs(x).max()) <= 1e185


def test_debug_hooks_and_broadcast_counters(ctx):
    """ks_ctx_set_debug takes the six documented keys and nothing else; ks_comm_bcast_stats counts nothing on a single rank without the force_multi hook."""
    from slepc_amd import _lib
    for key in ctx.DEBUG_KEYS:
        ctx.set_debug(key, 0 if key != "halo_overlap" else 1)       # every hook at its default
    assert ctx.L.ks_ctx_set_debug(ctx.h, 99, 1) == 63               # KS_ERR_ARG_OUTOFRANGE
    assert "unknown debug key" in ctx.L.ks_last_error_message().decode()
    n0, s0 = ctx.bcast_stats(reset=True)
    assert ctx.bcast_stats() == (0, 0.0)
    _lib.check(ctx.L.ks_comm_bcast_stats(ctx.h, None, None, 0))     # NULL outputs are allowed
