"""The ops->gramschmidt slot (ks_bv_gramschmidt_pass) with pass chaining (ks_bv_set_state), driven the way its one caller drives
it - BVOrthogonalizeGS, bvorthog.c:145-217 - against the CPU oracle's BVOrthogonalizeColumn: identical pass counts and lindep
(integer control flow), coefficients and norms to rounding; chained and unchained runs agree; a changed state, another column or an
intervening sweep drops the chain."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu

ETA = 0.7071


class Caller:
    """BVOrthogonalizeColumn (bvorthog.c:315-339) + BVOrthogonalizeGS (:145-217) on the caller's side of the slot, with the object
    state the adapter announces (adapters/slepc/hipks.c HipksSync)."""

    def __init__(self, ctx, V, refine, announce=True):
        self.ctx, self.V, self.refine, self.announce = ctx, V, refine, announce
        self.state = 1
        self.m = V.m
        self.buf = V.buffer_ptr()

    def _pass(self, j, on=True, nr=True):
        if self.announce:
            self.V.SetState(self.state)
        return self.V.GramSchmidtPass(j, on, nr)

    def orthogonalize_column(self, j):
        if j > 0:
            self.ctx.memset(self.buf + 8 * j * self.m, 0, 8 * j)                 # BV_CleanCoefficients
        passes = 1
        if self.refine == 0:
            onrm, nrm = self._pass(j)
            while passes < 3 and nrm != 0.0 and abs(nrm) < ETA * abs(onrm):
                passes += 1
                onrm, nrm = self._pass(j)
            lindep = not (nrm != 0.0 and abs(nrm) >= ETA * abs(onrm))
        elif self.refine == 1:
            self._pass(j, False, False)
            nrm = self.V.NormColumn(j)
            lindep = nrm == 0.0
        else:
            self._pass(j, False, False)
            onrm, nrm = self._pass(j); passes = 2
            lindep = not (nrm != 0.0 and abs(nrm) >= ETA * abs(onrm))
        self.ctx.memcpy_h2d(self.buf + 8 * (j * self.m + j), np.array([0.0 if lindep else nrm]))   # BV_SetValue
        self.state += 1                                                          # bvorthog.c:338
        return nrm, lindep, passes

    def scale(self, j, alpha):
        self.V.ScaleColumn(j, alpha)
        self.state += 1                                                          # bvops.c:356


def _columns(n, m, seed):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, m))
    X[:, 3] = X[:, 0] + 2.0 ** -10 * X[:, 3]          # these three need their second pass; what is left of them is 1e-3 of the column, so its
    X[:, 5] = X[:, 1] - 2.0 * X[:, 2] + 2.0 ** -12 * X[:, 5]      # direction is known to ~1e-9 even by the oracle's plain summation at n = 7e5
    X[:, 6] = X[:, 4] + 2.0 ** -8 * X[:, 6]
    X[:, m - 2] = 2.0 * X[:, 1] - 3.0 * X[:, 4]       # dependent
    return X


@pytest.mark.parametrize("refine", [0, 1, 2])
@pytest.mark.parametrize("n", [2000, 300001, 700000])
def test_chained_slot_matches_the_oracle_and_the_unchained_slot(ctx, n, refine):
    """Sizes on both sides of the occupancy-based grids (the dots a chained pass reads were written by the update kernel's grid,
    not by a dot sweep's)."""
    import slepc_amd as ks
    m = 10
    X = _columns(n, m, n + refine)
    dep = m - 2              # the dependent column: what rounding leaves of it decides its pass count and flag - not comparable; zeroed on both sides
    runs = []
    for announce in (True, False):
        V = ks.BV(ctx, n, m)
        V.SetOrthogonalization(ks.CGS, refine)
        V.set_dense(X)
        c = Caller(ctx, V, refine, announce)
        res = []
        for j in range(m):
            nrm, lin, passes = c.orthogonalize_column(j)
            res.append((nrm, lin, passes))
            c.scale(j, 0.0 if (lin or nrm == 0.0 or j == dep) else 1.0 / nrm)
        runs.append((res, V.buffer(), V.dense(), V.GsChainStats()))
    Vo = O.BV(n, m)
    Vo.SetOrthogonalization(O.CGS, refine)
    for j in range(m):
        Vo.set_column(j, X[:, j])
    ref = []
    for j in range(m):
        _, nrm, lin = Vo.OrthogonalizeColumn(j)
        ref.append((nrm, lin, Vo.passes_last()))
        Vo.ScaleColumn(j, 0.0 if (lin or nrm == 0.0 or j == dep) else 1.0 / nrm)
    Bo = np.array(Vo.buffer)
    for res, B, Q, stats in runs:
        for j in range(m):
            if j == dep:
                assert res[j][0] < 1e-8 * np.sqrt(n) and ref[j][0] < 1e-8 * np.sqrt(n)      # rounding noise of columns of norm ~sqrt(n)
                continue
            assert res[j][1] == ref[j][1] and res[j][2] == ref[j][2], (j, res[j], ref[j])      # lindep and pass count: identical
            # columns 3, 5, 6 are what is left of 2^-10 / 2^-12 / 2^-8 perturbations, and the later columns' coefficients against them
            # inherit their accuracy
            tol = 1e-11 if j < 3 else 1e-7
            if not ref[j][1]:
                assert abs(res[j][0] - ref[j][0]) <= tol * max(1.0, abs(ref[j][0])), (j, res[j], ref[j])
            if not ref[j][1] and j > 0:
                assert np.allclose(B[:j, j], Bo[:j, j], rtol=tol, atol=tol * np.abs(Bo[:j, j]).max()), j
        keep = [j for j in range(m) if not ref[j][1] and j != dep]
        if refine != 1:
            G = Q[:, keep].T @ Q[:, keep]
            assert np.abs(G - np.eye(len(keep))).max() < 1e-12
    total = sum(r[2] for r in runs[0][0][1:])                # column 0: nothing to orthogonalize against (plain path, not counted)
    a, b = runs[0][3], runs[1][3]
    assert b["chained"] == 0 and b["fresh"] == sum(r[2] for r in runs[1][0][1:])
    assert a["chained"] + a["fresh"] == total
    if refine == 1:
        assert a["chained"] == 0
    elif refine == 2:
        assert a["chained"] == m - 1                         # the second of the two passes of every column
    else:
        # every pass after a column's first one, under an unchanged state (the dependent column may fall back to an explicit norm between its
        # passes: that sweep drops the chain)
        assert total - (m - 1) - 2 <= a["chained"] <= total - (m - 1) and a["chained"] >= 3
    # chained and unchained agree to rounding on everything
    for j in range(m):
        if not ref[j][1] and j != dep:
            assert abs(runs[0][0][j][0] - runs[1][0][j][0]) <= 1e-9 * max(1.0, abs(ref[j][0]))


def test_chain_is_dropped_by_a_new_state_another_column_or_a_sweep(ctx):
    import slepc_amd as ks
    n, m = 50000, 6
    rng = np.random.default_rng(11)
    X = rng.standard_normal((n, m))
    X[:, 3] = X[:, 0] + 2.0 ** -12 * X[:, 3]
    X[:, 4] = X[:, 1] + 2.0 ** -12 * X[:, 4]

    def fresh_bv():
        V = ks.BV(ctx, n, m)
        V.set_dense(X)
        for j in range(3):
            V.OrthonormalizeColumn(j)
        ctx.memset(V.buffer_ptr() + 8 * 3 * m, 0, 8 * 3)
        ctx.memset(V.buffer_ptr() + 8 * 4 * m, 0, 8 * 4)
        return V

    # reference values: the unchained second pass
    V = fresh_bv()
    V.GramSchmidtPass(3)
    on_ref, nr_ref = V.GramSchmidtPass(3)
    assert V.GsChainStats() == {"chained": 0, "fresh": 2}

    V = fresh_bv(); V.SetState(7)
    on1, nr1 = V.GramSchmidtPass(3)
    assert nr1 < ETA * on1
    V.SetState(7)
    on2, nr2 = V.GramSchmidtPass(3)
    assert V.GsChainStats() == {"chained": 1, "fresh": 1}
    assert abs(on2 - on_ref) <= 1e-13 * on_ref and abs(nr2 - nr_ref) <= 1e-12 * nr_ref

    # a new state: the column was rewritten through a pointer the library lent earlier
    V = fresh_bv(); p3 = V.column_ptr(3); V.SetState(7)
    V.GramSchmidtPass(3)
    w = rng.standard_normal(n)
    ctx.memcpy_h2d(p3, w)
    V.SetState(8)
    on2, _ = V.GramSchmidtPass(3)
    assert V.GsChainStats() == {"chained": 0, "fresh": 2}
    assert abs(on2 - np.linalg.norm(w)) <= 1e-13 * np.linalg.norm(w)

    # another column in between
    V = fresh_bv(); V.SetState(7)
    V.GramSchmidtPass(3)
    V.GramSchmidtPass(4)
    on2, nr2 = V.GramSchmidtPass(3)
    assert V.GsChainStats()["chained"] == 0
    assert abs(on2 - on_ref) <= 1e-13 * on_ref

    # a sweep of this BV in between (its partial sums replace the ones the pass left)
    V = fresh_bv(); V.SetState(7)
    V.GramSchmidtPass(3)
    V.NormColumn(1)
    on2, nr2 = V.GramSchmidtPass(3)
    assert V.GsChainStats()["chained"] == 0
    assert abs(on2 - on_ref) <= 1e-13 * on_ref and abs(nr2 - nr_ref) <= 1e-12 * nr_ref

    # a library call that writes the column
    V = fresh_bv(); V.SetState(7)
    V.GramSchmidtPass(3)
    V.ScaleColumn(3, 2.0)
    on2, _ = V.GramSchmidtPass(3)
    assert V.GsChainStats()["chained"] == 0
    assert abs(on2 - 2.0 * on_ref) <= 1e-13 * on_ref


def test_slot_pass_does_not_wait_for_the_stream(ctx):
    """The slot's scalars come back through the mailbox when the update kernel starts: work enqueued behind a long sweep is still
    running when the call returns (it would have been drained by a stream synchronisation)."""
    import time
    import slepc_amd as ks
    n, m = 4_000_000, 24
    V = ks.BV(ctx, n, m)
    V.SetRandom()
    for j in range(m - 1):
        V.OrthonormalizeColumn(j)
    ctx.memset(V.buffer_ptr() + 8 * (m - 1) * m, 0, 8 * (m - 1))
    ctx.synchronize()
    V.SetState(1)
    t0 = time.perf_counter()
    V.GramSchmidtPass(m - 1)
    t_call = time.perf_counter() - t0
    t0 = time.perf_counter()
    ctx.synchronize()
    t_rest = time.perf_counter() - t0
    # dot sweep (23 columns) and update sweep (23 columns + write) are about the same size: the call returns after the first,
    # the second is still in flight
    assert t_rest > 0.25 * t_call, (t_call, t_rest)
