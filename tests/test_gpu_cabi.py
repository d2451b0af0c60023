"""The BV-level slots of the drop-in boundary driven from C programs (tests/c_abi/*.c, strict C99 against include/ksgpu.h
only), checked against the reference's own golden output and against the CPU oracle."""
import re
import subprocess

import numpy as np
import pytest

import golden_inputs as gi
from oracle import oracle as O
from test_abi import _build_c_example

pytestmark = pytest.mark.gpu


def _run(tmp_path, name, *args):
    exe = _build_c_example(tmp_path, name)
    r = subprocess.run([exe, *args], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    return r.stdout


@pytest.mark.parametrize("args,golden", [((), "bv/test1_1_bv_type-svec.out"), (("-testlda",), "bv/test1_2_bv_type-svec.out")])
def test_bv_test1_in_c_matches_the_reference_output_file(tmp_path, args, golden):
    """src/sys/classes/bv/tests/test1.c in C99 on the ABI: every line of the reference's output file (the test's own
    filter, sed 's/-0[.]/0./g', applied to both sides)."""
    out = _run(tmp_path, "bv_test1_abi", *args)
    filt = lambda t: [re.sub(r"-0[.]", "0.", ln.rstrip()) for ln in t.strip().splitlines()]     # noqa: E731
    got, want = filt(out), filt(gi.read(golden))
    assert got == want, [(i, a, b) for i, (a, b) in enumerate(zip(got, want)) if a != b][:5]


def _gs_inputs():
    n, m = 2000, 9
    i = np.arange(n, dtype=np.int64)

    def entry(j):
        return (((i * 37 + j * 101 + ((i * i) % 13) * 7) % 17) - 8) * 0.0625
    X = np.stack([entry(j) for j in range(m)], axis=1)
    X[:, 4] = entry(0) + np.ldexp(entry(4), -30)
    X[:, 6] = 2.0 * entry(1) - 3.0 * entry(2)
    X[:, 8] = 0.0
    return X


@pytest.mark.parametrize("refine", [0, 1, 2])
@pytest.mark.parametrize("mgs", [False, True])
def test_gramschmidt_slot_replayed_from_c_matches_the_oracle(tmp_path, refine, mgs):
    """ops->gramschmidt as its caller uses it (BVOrthogonalizeGS, bvorthog.c:145-217, restated in gs_slot_abi.c): one
    ks_bv_gramschmidt_pass per pass, NULL norms where the reference passes NULL, refinement loop and lindep on the caller's
    side. Pass counts and lindep identical to the oracle on a column that needs refinement, a dependent one and a zero
    one; norms and coefficients to rounding."""
    out = _run(tmp_path, "gs_slot_abi", str(refine), *(["mgs"] if mgs else []))
    X = _gs_inputs()
    n, m = X.shape
    V = O.BV(n, m)
    V.SetOrthogonalization(O.MGS if mgs else O.CGS, refine)
    for j in range(m):
        V.set_column(j, X[:, j])
    cols = [ln.split() for ln in out.splitlines() if ln.startswith("column")]
    hs = [ln.split() for ln in out.splitlines() if ln.startswith("H[")]
    assert len(cols) == m and len(hs) == m
    for j in range(m):
        _, nrm, lin = V.OrthogonalizeColumn(j)
        passes = V.passes_last()
        if lin or nrm == 0.0 or j == 6:
            V.ScaleColumn(j, 0.0)
        else:
            V.ScaleColumn(j, 1.0 / nrm)
        c = cols[j]
        g = float(c[7])
        if j == 6:
            # what rounding leaves of a dependent column: only its size is comparable (one pass without refinement
            # leaves what the loss of orthogonality of column 4 lets through)
            assert g < (1e-3 if refine == 1 else 1e-12) and nrm < (1e-3 if refine == 1 else 1e-12)
            continue
        assert int(c[1]) == j and int(c[5]) == int(lin), (j, c, lin)
        assert int(c[3]) == passes, (j, c, passes)
        if j >= 4:
            assert abs(g - nrm) <= (1e-4 if refine == 1 else 1e-6) * max(nrm, 1.0)   # column 4 is what is left of a 2^-30 perturbation: known to ~1e-7 relative, and later columns inherit that
        else:
            assert abs(g - nrm) <= 1e-12 * max(nrm, 1.0), (j, g, nrm)
    B = np.array(V.buffer)
    for j in range(m):
        hg = np.array([float(t) for t in hs[j][1:]])
        assert hg.shape[0] == j + 1
        tol = 1e-12 if j < 4 else (1e-4 if refine == 1 else 1e-6)
        if j != 6:
            assert np.allclose(hg[:j], B[:j, j], rtol=tol, atol=tol * 10), (j, hg, B[: j + 1, j])
    # the vector form with host h / c: one pass against three orthonormal columns
    vec = [ln.split() for ln in out.splitlines() if ln.startswith("vector")][0]
    Q = np.stack([V.column(j) for j in range(3)], axis=1)
    w = X[:, 3] * 1.0
    # column 3 of X was overwritten in the BV by its orthonormalised form; the C program used the original entries
    i = np.arange(n, dtype=np.int64)
    w = (((i * 37 + 3 * 101 + ((i * i) % 13) * 7) % 17) - 8) * 0.0625
    h = Q.T @ w
    assert np.allclose([float(vec[6]), float(vec[7]), float(vec[8])], h, rtol=1e-12, atol=1e-12)
    assert abs(float(vec[2]) - np.linalg.norm(w)) < 1e-11
    assert abs(float(vec[4]) - np.linalg.norm(w - Q @ h)) < 1e-9
    # constraints through the state mirror (ks_bv_set_layout) and an adopted coefficient buffer (ks_bv_set_buffer)
    con = [ln.split() for ln in out.splitlines() if ln.startswith("constraints")][0]
    ent = lambda j: (((i * 37 + j * 101 + ((i * i) % 13) * 7) % 17) - 8) * 0.0625      # noqa: E731
    w5, w6 = ent(5), ent(6)
    sc = 1.0 / np.sqrt(n / 2)
    c0 = np.where(i % 2 == 0, sc, 0.0); c1 = np.where(i % 2 == 1, sc, 0.0)
    q0 = w5 - (c0 @ w5) * c0 - (c1 @ w5) * c1; q0 /= np.linalg.norm(q0)
    h = np.array([c0 @ w6, c1 @ w6, q0 @ w6])
    assert abs(float(con[2]) - np.linalg.norm(w6)) < 1e-11
    assert np.allclose([float(con[6]), float(con[7]), float(con[8])], h, rtol=0, atol=1e-11)
    assert abs(float(con[4]) - np.linalg.norm(w6 - h[0] * c0 - h[1] * c1 - h[2] * q0)) < 1e-9
    assert max(abs(float(con[10])), abs(float(con[11])), abs(float(con[12]))) < 1e-12
    # pass chaining (ks_bv_set_state): the caller announced its object state at every call, as the adapter's HipksSync does
    stats = [ln.split() for ln in out.splitlines() if ln.startswith("chainstats")][0]
    chained, fresh = int(stats[2]), int(stats[4])
    if mgs:
        assert chained == 0 and fresh == 0            # MGS goes through the primitive ops
    elif refine == 1:
        assert chained == 0                           # REFINE_NEVER: one pass per column, nothing to chain to
    else:
        assert chained >= 3, stats                    # every second pass of a column under an unchanged state
        assert chained + fresh == sum(int(c[3]) for c in cols[1:])       # column 0 has nothing to orthogonalize against (k = 0: plain path)
    if refine == 0 and not mgs:
        ch = {int(t[2]): t for t in (ln.split() for ln in out.splitlines() if ln.startswith("chain mode") and " onrm1 " in ln)}
        cn = {int(t[2]): float(t[4]) for t in (ln.split() for ln in out.splitlines() if ln.startswith("chain mode") and " column_norm " in ln)}
        f = lambda t, i: float(t[i])                  # noqa: E731
        a, b, c2 = ch[0], ch[1], ch[2]
        assert (int(a[12]), int(a[14])) == (1, 1)     # announced, unchanged state: second pass chained to the first one's dots
        assert (int(b[12]), int(b[14])) == (0, 2)     # no state announced: both passes take their own dots
        assert (int(c2[12]), int(c2[14])) == (0, 2)   # column rewritten + state bumped between the passes: own dots of the new content
        assert f(a, 4) == f(b, 4) and f(a, 6) == f(b, 6)                   # first passes are the same launches
        assert f(a, 6) < 0.7071 * f(a, 4)                                  # the case does need its second pass
        for i in (8, 10):
            assert abs(f(a, i) - f(b, i)) <= 1e-12 * abs(f(b, i)), (i, a, b)
        assert np.allclose([f(a, 18), f(a, 19), f(a, 20)], [f(b, 18), f(b, 19), f(b, 20)], rtol=1e-14, atol=1e-15)
        assert abs(cn[0] - cn[1]) <= 1e-13 * cn[1]
        assert abs(f(c2, 8) - f(c2, 16)) <= 1e-13 * f(c2, 16), c2           # onrm of the second pass = norm of what the column NOW holds
