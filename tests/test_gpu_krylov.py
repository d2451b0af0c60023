"""SpMV, fused Gram-Schmidt, Lanczos/Arnoldi and the Krylov-Schur driver on the GPU versus the CPU oracle,
the reference's golden eigenvalues and the analytic Laplacian spectra.

Tolerances (north star): Ritz values within 1e-10 relative of the CPU reference path; relative residuals
||Ax-kx||/|k| <= tol=1e-8 (EPSComputeError); permutation / pass-count / iteration-count work must be
IDENTICAL to the oracle (integer control flow)."""
import numpy as np
import pytest

import golden_inputs as gi
import scenarios as sc
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _mat(ctx, Ao):
    import slepc_amd as ks
    return ks.Mat.from_csr(ctx, Ao.rowptr, Ao.col, Ao.val)


# ---- SpMV ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(1, 1, 1), (2, 1, 1), (5, 4, 3), (16, 16, 16), (33, 17, 9), (64, 64, 20)])
def test_spmv_laplacian3d(ctx, shape):
    import slepc_amd as ks
    Ao = O.laplacian3d(*shape)
    x = np.random.default_rng(1).standard_normal(Ao.n)
    y0 = Ao.mult(x)
    for A in (_mat(ctx, Ao), ks.Mat.laplacian3d(ctx, *shape)):
        assert A.nnz == Ao.nnz and A.n == Ao.n
        y = A.mult(x)
        assert np.abs(y - y0).max() <= 8 * np.finfo(float).eps * np.abs(x).max() * 12


def test_spmv_laplacian2d_generator_matches_ex2(ctx):
    import slepc_amd as ks
    Ao = O.laplacian2d(37, 23)
    x = np.random.default_rng(2).standard_normal(Ao.n)
    assert np.abs(ks.Mat.laplacian2d(ctx, 37, 23).mult(x) - Ao.mult(x)).max() < 1e-13


@pytest.mark.parametrize("n,mean", [(1000, 1), (3000, 3), (5000, 12), (4000, 32), (2000, 70), (300, 200)])
def test_spmv_random_csr_ragged(ctx, n, mean):
    """Ragged rows incl. EMPTY rows and rows longer than a wavefront; every lanes-per-row variant."""
    import scipy.sparse as sp
    rng = np.random.default_rng(n + mean)
    lens = np.clip(rng.poisson(mean, n), 0, n)
    lens[rng.integers(0, n, n // 20)] = 0
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    col = np.concatenate([np.sort(rng.choice(n, l, replace=False)) for l in lens] + [np.empty(0, int)]).astype(np.int32)
    val = rng.uniform(-1, 1, rowptr[-1])
    Ao = O.CSR(n, rowptr, col, val)
    x = rng.standard_normal(n)
    y0 = Ao.mult(x)
    y = _mat(ctx, Ao).mult(x)
    assert np.allclose(y, y0, rtol=0, atol=1e-13 * max(1, mean))
    assert np.all(y[lens == 0] == 0.0)


def test_spmv_csr_row_block_kernel(ctx, monkeypatch):
    """The general-matrix kernel (256-row blocks streamed through LDS): rows longer than a 1024-entry chunk, rows whose length
    is a multiple of 32, empty rows, a last block that is not full; the same bits as the CSR-vector kernel's reference sum order
    is not promised, the oracle's entry-order fma chain is: compare with the oracle to rounding and with the SELL kernel
    bit for bit where SELL applies."""
    import slepc_amd as ks
    rng = np.random.default_rng(21)
    n = 5000
    lens = rng.integers(0, 40, n); lens[::9] = 0; lens[7] = 1500; lens[2048] = 2600; lens[100:164] = 32; lens[n - 1] = 64
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    col = np.concatenate([np.sort(rng.choice(n, l, replace=False)) for l in lens] + [np.empty(0, int)]).astype(np.int32)
    val = rng.uniform(-1, 1, rowptr[-1])
    Ao = O.CSR(n, rowptr, col, val)
    x = rng.standard_normal(n)
    y0 = Ao.mult(x)
    monkeypatch.setenv("KSGPU_SPMV", "csr")
    A = ks.Mat.from_csr(ctx, rowptr, col, val); assert A.layout() == "csr"
    y = A.mult(x)
    assert np.allclose(y, y0, rtol=0, atol=1e-12) and np.all(y[lens == 0] == 0.0)
    monkeypatch.setenv("KSGPU_SPMV", "csrvec")
    yv = ks.Mat.from_csr(ctx, rowptr, col, val).mult(x)
    assert np.allclose(yv, y0, rtol=0, atol=1e-12)
    # a stencil: bit-identical to the SELL kernel (same entry order, same fma chain)
    Lo = O.laplacian3d(40, 30, 20)
    xs = rng.standard_normal(Lo.n)
    monkeypatch.setenv("KSGPU_SPMV", "csr"); yc = ks.Mat.laplacian3d(ctx, 40, 30, 20).mult(xs)
    monkeypatch.setenv("KSGPU_SPMV", "sell"); ys = ks.Mat.laplacian3d(ctx, 40, 30, 20).mult(xs)
    assert np.array_equal(yc, ys)
    monkeypatch.setenv("KSGPU_SPMV", "csrregs"); yr = ks.Mat.laplacian3d(ctx, 40, 30, 20).mult(xs)
    assert np.array_equal(yr, ys)


@pytest.mark.parametrize("n,hi", [(4999, 11), (70001, 11), (30011, 25), (30011, 35)])
def test_spmv_csr_short_rows_lds_dma_form(ctx, monkeypatch, n, hi):
    """Short rows (at most 16 entries on average) take the LDS-DMA form of the wave kernel (round 4: the col / val streams go straight into a lane-linear
    LDS image, global_load_lds_dwordx4; chunks of 512 entries up to 8 per row, 768 up to 12, 1024 up to 16 - hi = 11 / 25 / 35 lands in each): empty rows, stretches of rows of 8 and of 16 entries (all lanes of a row-side read on one bank group: slower, not
    wrong), one row far longer than a 512-entry chunk, a last group of fewer than 64 rows, a row pointer run that does not start on a multiple of four. The
    register-staged form (KSGPU_SPMV=csrregs) runs the same rows: bit for bit the same sums (same entry order, same fma chain), both at the oracle's."""
    import slepc_amd as ks
    rng = np.random.default_rng(n)
    lens = rng.integers(0, hi, n); lens[::5] = 0; lens[200:328] = 8; lens[1000:1128] = 16; lens[3] = 1; lens[n // 2] = 3000; lens[n - 1] = 5
    mean = lens.sum() / n
    assert {11: mean <= 8, 25: 8 < mean <= 12, 35: 12 < mean <= 16}[hi], mean
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    col = np.concatenate([np.sort(rng.choice(n, l, replace=False)) for l in lens] + [np.empty(0, int)]).astype(np.int32)
    val = rng.uniform(-1, 1, rowptr[-1])
    Ao = O.CSR(n, rowptr, col, val)
    x = rng.standard_normal(n)
    y0 = Ao.mult(x)
    out = {}
    for fmt in ("csr", "csrregs", "csrvec"):
        monkeypatch.setenv("KSGPU_SPMV", fmt)
        A = ks.Mat.from_csr(ctx, rowptr, col, val); assert A.layout() == "csr"
        out[fmt] = A.mult(x)
        assert np.allclose(out[fmt], y0, rtol=0, atol=1e-12) and np.all(out[fmt][lens == 0] == 0.0), fmt
        A.destroy()
    assert np.array_equal(out["csr"], out["csrregs"])


@pytest.mark.parametrize("fmt", ["csr", "csrvec", "sell"])
def test_spmv_both_layouts(ctx, monkeypatch, fmt):
    """The CSR-vector kernel and the SELL-64 kernel (layout picked at assembly, KSGPU_SPMV forces one) agree with
    the oracle on a stencil matrix and on a ragged matrix with empty rows and NaN-free padding semantics."""
    import slepc_amd as ks
    monkeypatch.setenv("KSGPU_SPMV", fmt)
    Ao = O.laplacian3d(20, 13, 11)
    x = np.random.default_rng(3).standard_normal(Ao.n)
    assert np.abs(ks.Mat.laplacian3d(ctx, 20, 13, 11).mult(x) - Ao.mult(x)).max() < 1e-13
    rng = np.random.default_rng(4)
    n = 1000
    lens = rng.integers(0, 9, n); lens[::7] = 0
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    col = np.concatenate([np.sort(rng.choice(n, l, replace=False)) for l in lens] + [np.empty(0, int)]).astype(np.int32)
    val = rng.uniform(-1, 1, rowptr[-1])
    B = O.CSR(n, rowptr, col, val)
    xb = rng.standard_normal(n)
    xb[0] = np.nan                                   # column 0 is only referenced by a few rows: NaN must not leak via padding
    y = _mat(ctx, B).mult(xb); y0 = B.mult(xb)
    assert np.array_equal(np.isnan(y), np.isnan(y0))
    ok = ~np.isnan(y0)
    assert np.allclose(y[ok], y0[ok], atol=1e-13)
    assert np.all(y[lens == 0] == 0.0)


def test_spmv_dictionary_layout(ctx, monkeypatch):
    """Dictionary ELL (two bytes per entry: offset code, value code) is picked for matrices with few distinct values
    and offsets and rows of at most 16 entries; its result is BIT-IDENTICAL to the SELL layout's (same order, same
    fma chain). Ragged rows, empty rows, padding that must not touch NaN, 13-entry rows (32-byte form), and matrices
    that do not qualify."""
    import scipy.sparse as sp
    import slepc_amd as ks

    def both(make):
        monkeypatch.setenv("KSGPU_SPMV", "sell"); S = make(); assert S.layout() == "sell"
        monkeypatch.delenv("KSGPU_SPMV"); D = make()
        return S, D
    rng = np.random.default_rng(8)
    # the generators
    for make, n in ((lambda: ks.Mat.laplacian3d(ctx, 37, 23, 19), 37 * 23 * 19), (lambda: ks.Mat.laplacian2d(ctx, 300), 90000)):
        S, D = both(make)
        assert D.layout() == "dict"
        x = rng.standard_normal(n)
        assert np.array_equal(S.mult(x), D.mult(x))
        assert np.array_equal(S.get_diagonal(), D.get_diagonal()) and S.norm_inf() == D.norm_inf()
    # user CSR: mesh-graph Laplacian (values -1, 2, 3, 4), ragged with empty rows, NaN guarded by the padding mask
    G = sc.graph_laplacian_2d(50, 41)
    S, D = both(lambda: ks.Mat.from_csr(ctx, G.indptr, G.indices, G.data))
    assert D.layout() == "dict"
    x = rng.standard_normal(G.shape[0])
    y = D.mult(x)
    assert np.array_equal(S.mult(x), y) and np.abs(y - G @ x).max() < 1e-13
    n = 3000
    lens = rng.integers(0, 14, n); lens[::5] = 0
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    col = np.concatenate([np.sort(np.clip(r + rng.choice(np.arange(-40, 41), l, replace=False), 0, n - 1)) for r, l in enumerate(lens)] + [np.empty(0, int)]).astype(np.int32)
    # clipping can repeat a column inside a row: legal CSR for a product, and it keeps the offset set small
    val = rng.choice([0.5, -1.25, 3.0, -0.0, 1e-300], rowptr[-1])
    S, D = both(lambda: ks.Mat.from_csr(ctx, rowptr, col, val))
    assert D.layout() == "dict"                                            # 13-entry rows: the 32-byte form
    xb = rng.standard_normal(n); xb[n - 1] = np.nan
    y, y0 = D.mult(xb), S.mult(xb)
    assert np.array_equal(np.isnan(y), np.isnan(y0)) and np.array_equal(y[~np.isnan(y)], y0[~np.isnan(y0)])
    assert np.all(y[lens == 0] == 0.0) and np.signbit(y[lens == 0]).sum() == 0
    ref = O.CSR(n, rowptr, col, val).mult(xb); ok = ~np.isnan(ref)
    assert np.allclose(y[ok], ref[ok], rtol=0, atol=1e-13)
    # not dictionary matrices: too many distinct values / rows too long / too many offsets
    val2 = rng.standard_normal(rowptr[-1])
    assert ks.Mat.from_csr(ctx, rowptr, col, val2).layout() == "odict"     # values in full, offsets still coded
    R = sp.random(2000, 2000, density=20 / 2000, random_state=3, format="csr", data_rvs=lambda k: np.ones(k)); R.sort_indices()
    assert ks.Mat.from_csr(ctx, R.indptr, R.indices, R.data).layout() in ("sell", "csr")


def test_spmv_offset_dictionary_layout(ctx, monkeypatch):
    """Variable-coefficient stencils: too many distinct values for the value dictionary, still few column offsets: the index
    of an entry becomes one byte, the values stay doubles in SELL order. Bit-identical to the SELL result; 13-entry rows;
    NaN guarded by the padding code; a forced "odict" on a constant-coefficient matrix."""
    import slepc_amd as ks
    rng = np.random.default_rng(12)
    Ao = O.laplacian3d(31, 19, 23)
    val = rng.standard_normal(Ao.val.shape[0])

    def both(rowptr, col, v):
        monkeypatch.setenv("KSGPU_SPMV", "sell"); S = ks.Mat.from_csr(ctx, rowptr, col, v); assert S.layout() == "sell"
        monkeypatch.delenv("KSGPU_SPMV"); D = ks.Mat.from_csr(ctx, rowptr, col, v)
        return S, D
    S, D = both(Ao.rowptr, Ao.col, val)
    assert D.layout() == "odict"
    x = rng.standard_normal(Ao.n)
    y = D.mult(x)
    assert np.array_equal(S.mult(x), y)
    assert np.abs(y - O.CSR(Ao.n, Ao.rowptr, Ao.col, val).mult(x)).max() < 1e-12
    assert np.array_equal(S.get_diagonal(), D.get_diagonal()) and S.norm_inf() == D.norm_inf()
    n = 3000
    lens = rng.integers(0, 14, n); lens[::5] = 0
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    col = np.concatenate([np.sort(np.clip(r + rng.choice(np.arange(-40, 41), l, replace=False), 0, n - 1)) for r, l in enumerate(lens)] + [np.empty(0, int)]).astype(np.int32)
    v2 = rng.standard_normal(rowptr[-1])
    S, D = both(rowptr, col, v2)
    assert D.layout() == "odict"
    xb = rng.standard_normal(n); xb[n - 1] = np.nan
    y, y0 = D.mult(xb), S.mult(xb)
    assert np.array_equal(np.isnan(y), np.isnan(y0)) and np.array_equal(y[~np.isnan(y)], y0[~np.isnan(y0)])
    assert np.all(y[lens == 0] == 0.0)
    monkeypatch.setenv("KSGPU_SPMV", "odict")
    L = ks.Mat.laplacian3d(ctx, 20, 13, 11)
    monkeypatch.delenv("KSGPU_SPMV")
    assert L.layout() == "odict"
    xl = rng.standard_normal(20 * 13 * 11)
    assert np.array_equal(L.mult(xl), ks.Mat.laplacian3d(ctx, 20, 13, 11).mult(xl))          # against the value+offset dictionary


def test_spmv_dictionary_layouts_27_point_stencil(ctx, monkeypatch):
    """Rows of up to 32 entries (27-point stencil: the 64-byte / 32-byte forms), constant and variable coefficients."""
    import scipy.sparse as sp
    import slepc_amd as ks
    nx, ny, nz = 17, 13, 11
    def tri(n):
        return sp.diags([np.ones(n - 1), np.ones(n), np.ones(n - 1)], [-1, 0, 1])
    P = sp.kron(tri(nz), sp.kron(tri(ny), tri(nx))).tocsr(); P.sort_indices()          # the 27-point pattern
    assert np.diff(P.indptr).max() == 27
    rng = np.random.default_rng(27)
    x = rng.standard_normal(P.shape[0])
    for name, data in (("dict", np.where(P.data > 0, -1.0, 0.0) + 0.0), ("odict", rng.standard_normal(P.nnz))):
        if name == "dict":
            data = np.full(P.nnz, -1.0); data[P.indices == np.repeat(np.arange(P.shape[0]), np.diff(P.indptr))] = 26.0
        monkeypatch.setenv("KSGPU_SPMV", "sell"); S = ks.Mat.from_csr(ctx, P.indptr, P.indices, data)
        monkeypatch.delenv("KSGPU_SPMV"); D = ks.Mat.from_csr(ctx, P.indptr, P.indices, data)
        assert D.layout() == name and S.layout() in ("sell", "csr")
        y = D.mult(x)
        assert np.array_equal(S.mult(x), y)
        assert np.abs(y - sp.csr_matrix((data, P.indices, P.indptr), shape=P.shape) @ x).max() < 1e-11


def test_spmv_xcd_sliced_layout(ctx, monkeypatch):
    """The XCD-sliced layout (column ranges pinned to XCDs by blockIdx % 8, eight partial results added in fixed order):
    forced on small and ragged matrices incl. empty rows and duplicate entries, and chosen automatically for a
    wide-scatter matrix with a 32 MB vector; diagonal and infinity norm stay available after the CSR arrays are released."""
    import slepc_amd as ks
    import nhep_cases as nc
    rng = np.random.default_rng(8)
    monkeypatch.setenv("KSGPU_SPMV", "sliced")
    for n, mean in [(5000, 3), (20000, 20), (70001, 7)]:
        lens = np.clip(rng.poisson(mean, n), 0, n); lens[rng.integers(0, n, n // 20)] = 0
        rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        col = rng.integers(0, n, rowptr[-1]).astype(np.int32)          # unsorted, duplicates allowed
        val = rng.uniform(-1, 1, rowptr[-1])
        Ao = O.CSR(n, rowptr, col, val)
        A = _mat(ctx, Ao)
        assert A.layout() == "sliced"
        x = rng.standard_normal(n)
        y, y0 = A.mult(x), Ao.mult(x)
        assert np.allclose(y, y0, rtol=0, atol=1e-13 * max(1, mean)) and np.all(y[lens == 0] == 0.0)
        S = Ao.to_scipy()
        assert np.allclose(A.get_diagonal(), S.diagonal(), rtol=0, atol=1e-15)
        assert abs(A.norm_inf() - abs(S).sum(axis=1).max()) < 1e-12
        assert np.array_equal(A.mult(x), y)                              # fixed-order partial sums: run-to-run identical
    Ao, _ = nc.config5_pencil_fast(1_500_000, mean_nnz=12)               # x = 12 MB > 6 MB, entries far from the diagonal
    A = _mat(ctx, Ao)
    assert A.layout() == "sliced"                                          # (forced here; the automatic choice for such a matrix is the binned layout, below)
    x = rng.standard_normal(Ao.n)
    assert np.allclose(A.mult(x), Ao.mult(x), rtol=0, atol=1e-11)
    assert abs(A.get_diagonal()[12345] - Ao.to_scipy().diagonal()[12345]) < 1e-13


def test_spmv_binned_layout(ctx, monkeypatch):
    """The binned (two-phase) layout: gather from a piece of x in LDS into bin-major order, then a wave per bin of rows adds val * G into its
    rows in LDS. Forced on small and ragged matrices incl. empty rows, unsorted columns and duplicate entries (many short segments: the
    per-pair lookup path), and chosen automatically for a wide-scatter matrix with a 12 MB vector (long segments: the window path);
    run-to-run identical; diagonal and infinity norm stay available after the CSR arrays are released."""
    import slepc_amd as ks
    import nhep_cases as nc
    rng = np.random.default_rng(18)
    monkeypatch.setenv("KSGPU_SPMV", "binned")
    for n, mean in [(5000, 3), (20000, 20), (70001, 7), (300007, 40)]:
        lens = np.clip(rng.poisson(mean, n), 0, n); lens[rng.integers(0, n, n // 20)] = 0
        rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        col = rng.integers(0, n, rowptr[-1]).astype(np.int32)          # unsorted, duplicates allowed
        val = rng.uniform(-1, 1, rowptr[-1])
        Ao = O.CSR(n, rowptr, col, val)
        A = _mat(ctx, Ao)
        assert A.layout() == "binned"
        x = rng.standard_normal(n)
        y, y0 = A.mult(x), Ao.mult(x)
        assert np.allclose(y, y0, rtol=0, atol=1e-13 * max(1, mean)) and np.all(y[lens == 0] == 0.0)
        assert np.array_equal(A.mult(x), y)                              # a fixed order of additions per row: run-to-run identical
        xi = x.copy(); xi[::97] = np.inf                                  # an infinite entry of x reaches exactly the rows that use it (padding entries do not spread it)
        yi = A.mult(xi)
        touched = np.zeros(n, bool); rows = np.repeat(np.arange(n), lens); touched[rows[col % 97 == 0]] = True
        assert np.all(np.isfinite(yi[~touched])) and not np.any(np.isfinite(yi[touched]))
        S = Ao.to_scipy()                                                 # (scipy sorts and merges the shared index arrays in place: last)
        assert np.allclose(A.get_diagonal(), S.diagonal(), rtol=0, atol=1e-15)
        assert abs(A.norm_inf() - abs(S).sum(axis=1).max()) < 1e-12
    monkeypatch.delenv("KSGPU_SPMV")
    Ao, _ = nc.config5_pencil_fast(1_500_000, mean_nnz=12)               # x = 12 MB > 6 MB, entries far from the diagonal
    A = _mat(ctx, Ao)
    assert A.layout() == "binned"
    assert ks.Mat.laplacian3d(ctx, 160, 160, 160).layout() == "dict"       # banded: stays with the row-ordered layouts
    x = rng.standard_normal(Ao.n)
    assert np.allclose(A.mult(x), Ao.mult(x), rtol=0, atol=1e-11)
    assert abs(A.get_diagonal()[12345] - Ao.to_scipy().diagonal()[12345]) < 1e-13


def test_spmv_rejects_bad_input(ctx):
    import slepc_amd as ks
    with pytest.raises(ks.KsError) as e:
        ks.Mat.from_csr(ctx, [0, 1], [5], [1.0])          # column out of range
    assert e.value.rc == 63


# ---- fused Gram-Schmidt -----------------------------------------------------------------------------------
@pytest.mark.parametrize("refine", [0, 1, 2])
@pytest.mark.parametrize("n,m", [(7, 3), (600, 12), (5000, 31), (3001, 33)])
def test_fused_gs_matches_oracle(ctx, refine, n, m):
    import slepc_amd as ks
    rng = np.random.default_rng(n + m + refine)
    Xh = rng.standard_normal((n, m))
    Xh[:, m // 2] = Xh[:, 0] + 1e-9 * rng.standard_normal(n)       # forces refinement passes
    Vg = ks.BV(ctx, n, m); Vo = O.BV(n, m)
    Vg.SetOrthogonalization(ks.CGS, refine); Vo.SetOrthogonalization(O.CGS, refine)
    Vg.set_dense(Xh)
    for j in range(m):
        Vo.set_column(j, Xh[:, j])
    for j in range(m):
        ng, lg = Vg.OrthonormalizeColumn(j)
        no, lo = Vo.OrthonormalizeColumn(j)
        if j == 0:
            # BV_SetValue(bv,0,0,NULL,nrm): column 0 of the buffer is both column 0's coefficient column and the scratch column
            assert Vg.buffer()[0, 0] == ng and np.array(Vo.buffer)[0, 0] == no
        assert lg == lo
        assert Vg.gs_passes()[1] == Vo.passes_last(), (j, Vg.gs_passes(), Vo.passes_last())
        # the norm comes from beta^2 - sum(h^2): its ABSOLUTE accuracy is eps*||x_j||, whatever is left of x_j
        # column m//2 is what is left of a 1e-9 perturbation, i.e. known to ~1e-7 relative: later columns inherit that
        rtol = 1e-12 if j <= m // 2 else 1e-5
        assert abs(ng - no) <= 1e-13 * np.linalg.norm(Xh[:, j]) + rtol * abs(no), (j, ng, no)
    B = Vg.buffer(); Bo = np.array(Vo.buffer)
    for j in range(1, m):
        assert np.allclose(B[: j + 1, j], Bo[: j + 1, j], rtol=1e-5, atol=1e-5 if j > m // 2 else 1e-9), j
    Q = Vg.dense()
    if refine != 1:
        assert np.abs(Q.T @ Q - np.eye(m)).max() < 1e-13


def test_fused_gs_generic_agree(ctx, debug):
    """The host-driven literal restatement (test hook no_fused_gs, read at BV creation) and the fused kernels give the same result."""
    import slepc_amd as ks
    n, m = 2000, 10
    Xh = np.random.default_rng(0).standard_normal((n, m))
    outs = []
    for fused in (True, False):
        if not fused:
            debug("no_fused_gs")
        V = ks.BV(ctx, n, m); V.set_dense(Xh)
        nr = [V.OrthonormalizeColumn(j)[0] for j in range(m)]
        outs.append((np.array(nr), V.dense(), V.buffer()))
    assert np.allclose(outs[0][0], outs[1][0], rtol=1e-13)
    assert np.allclose(outs[0][1], outs[1][1], atol=1e-13)
    assert np.allclose(np.triu(outs[0][2])[:, 1:], np.triu(outs[1][2])[:, 1:], atol=1e-12)


def test_explicit_norm_fallback_and_zero_vector(ctx):
    """norm^2 estimate <= 0 -> explicit BVNormColumn (bvorthog.c:126); exactly dependent column -> lindep."""
    import slepc_amd as ks
    n, m = 1000, 4
    rng = np.random.default_rng(5)
    Xh = rng.standard_normal((n, m))
    Xh[:, 2] = 2.0 * Xh[:, 0] - 3.0 * Xh[:, 1]        # exactly in the span
    Xh[:, 3] = 0.0
    Vg = ks.BV(ctx, n, m); Vo = O.BV(n, m)
    Vg.set_dense(Xh)
    for j in range(m):
        Vo.set_column(j, Xh[:, j])
    for j in range(m):
        ng, lg = Vg.OrthonormalizeColumn(j)
        no, lo = Vo.OrthonormalizeColumn(j)
        if j == 2:
            # what is left of an exactly dependent column is rounding noise: only its size is comparable
            assert ng < 1e-12 and no < 1e-12
            continue
        assert lg == lo, j
        assert Vg.gs_passes()[1] == Vo.passes_last(), j
    assert lg and ng == 0.0                             # zero column: norm 0, lindep (nrm==0)


def test_invalid_inner_product_is_an_error(ctx):
    """BV_SafeSqrt (bvimpl.h:137) raises PETSC_ERR_USER_INPUT when v'v is NaN/negative."""
    import slepc_amd as ks
    V = ks.BV(ctx, 100, 3)
    V.set_column(0, np.full(100, np.nan))
    with pytest.raises(ks.KsError) as e:
        V.OrthogonalizeColumn(0)
    assert e.value.rc == 95


# ---- Lanczos / Arnoldi ---------------------------------------------------------------------------------------
def _start(V):
    V.SetRandomColumn(0)
    _, nrm, _ = V.OrthogonalizeColumn(0)
    V.ScaleColumn(0, 1.0 / nrm)


@pytest.mark.parametrize("grid,m", [((30, 30), 12), ((50, 40), 30), ((21, 9), 20)])
def test_lanczos_matches_oracle(ctx, grid, m):
    import slepc_amd as ks
    Ao = O.laplacian2d(*grid)
    Ag = _mat(ctx, Ao)
    Vo = O.BV(Ao.n, m + 1); Vg = ks.BV(ctx, Ao.n, m + 1)
    _start(Vo); _start(Vg)
    assert np.array_equal(Vo.column(0), Vg.column(0)) or np.allclose(Vo.column(0), Vg.column(0), rtol=1e-15)
    To = np.zeros((m + 1, 3), order="F"); Tg = np.zeros((m + 1, 3), order="F")
    ro = Vo.MatLanczos(Ao, To, 0, m); rg = Vg.MatLanczos(Ag, Tg, 0, m)
    assert ro[0] == rg[0] and ro[2] == rg[2]
    assert abs(ro[1] - rg[1]) < 1e-12
    assert np.abs(To - Tg).max() < 1e-12
    assert Vg.gs_passes()[0] == Vo.passes_total()
    Vd = Vg.dense()
    assert np.abs(Vd.T @ Vd - np.eye(m + 1)).max() < 1e-13
    # three-term relation A V - V T = beta v e^T holds to rounding
    Tm = np.diag(Tg[:m, 0]) + np.diag(Tg[: m - 1, 1], 1) + np.diag(Tg[: m - 1, 1], -1)
    R = Ao.to_scipy() @ Vd[:, :m] - Vd[:, :m] @ Tm
    R[:, m - 1] -= rg[1] * Vd[:, m]
    assert np.abs(R).max() < 1e-12


def test_lanczos_restart_from_k(ctx):
    """k>0: the first k columns are locked and untouched (bvkrylov.c:30-33)."""
    import slepc_amd as ks
    Ao = O.laplacian2d(25)
    Ag = _mat(ctx, Ao)
    m = 10
    Vg = ks.BV(ctx, Ao.n, m + 1); Vo = O.BV(Ao.n, m + 1)
    _start(Vg); _start(Vo)
    Tg = np.zeros((m + 1, 3), order="F"); To = np.zeros((m + 1, 3), order="F")
    Vg.MatLanczos(Ag, Tg, 0, 4); Vo.MatLanczos(Ao, To, 0, 4)
    before = Vg.dense()[:, :5].copy()
    Vg.MatLanczos(Ag, Tg, 4, m); Vo.MatLanczos(Ao, To, 4, m)
    assert np.array_equal(before, Vg.dense()[:, :5])
    assert np.abs(Tg - To).max() < 1e-12


def test_arnoldi_matches_oracle(ctx):
    import scipy.sparse as sp
    import slepc_amd as ks
    n, m = 400, 16
    S = (sp.random(n, n, density=0.03, random_state=11, format="csr") + sp.eye(n, format="csr") * 2).tocsr()
    S.sort_indices()
    Ao = O.CSR(n, S.indptr, S.indices, S.data)
    Ag = _mat(ctx, Ao)
    Vo = O.BV(n, m + 1); Vg = ks.BV(ctx, n, m + 1)
    _start(Vo); _start(Vg)
    Ho = np.zeros((m + 1, m + 1), order="F"); Hg = np.zeros((m + 1, m + 1), order="F")
    ro = Vo.MatArnoldi(Ao, Ho, 0, m); rg = Vg.MatArnoldi(Ag, Hg, 0, m)
    assert ro[0] == rg[0] and ro[2] == rg[2] and abs(ro[1] - rg[1]) < 1e-12
    assert np.abs(Ho - Hg).max() < 1e-11
    Vd = Vg.dense()
    assert np.abs(S @ Vd[:, :m] - Vd @ Hg[: m + 1, :m]).max() < 1e-12
    assert np.abs(np.tril(Hg[:m, :m], -2)).max() == 0.0


def test_breakdown_halts_enqueued_run(ctx):
    """Invariant subspace of dimension 3: the device-side halt must report m=3, breakdown, like the oracle."""
    import slepc_amd as ks
    n = 12
    Ao = O.CSR(n, np.arange(n + 1), np.arange(n), np.arange(1, n + 1, dtype=float))
    Ag = _mat(ctx, Ao)
    v = np.zeros(n); v[[1, 4, 7]] = [1.0, 2.0, -1.0]; v /= np.linalg.norm(v)
    Vg = ks.BV(ctx, n, 8); Vo = O.BV(n, 8)
    Vg.set_column(0, v); Vo.set_column(0, v)
    Tg = np.zeros((8, 3), order="F"); To = np.zeros((8, 3), order="F")
    rg = Vg.MatLanczos(Ag, Tg, 0, 6); ro = Vo.MatLanczos(Ao, To, 0, 6)
    assert rg[0] == ro[0] == 3 and rg[2] and ro[2]
    assert np.allclose(Tg[:3], To[:3], atol=1e-12)


@pytest.mark.parametrize("n", [300001, 700000])
def test_halted_column_mid_run_reduces_with_its_own_grid(ctx, n):
    """A column that needs more than the optimistic two-pass program halts the enqueued run; its completion program is
    enqueued AFTER the (gated-off) sweeps of the later columns, whose kernels have other column tiles and, for
    131k < n < 1.05M rows, other occupancy-based grids. The partials of the halted column must be reduced with the grid
    that wrote them. Invariant subspace of dimension 3 at such an n: m = 3 and breakdown, like the oracle."""
    import slepc_amd as ks
    Ao = O.CSR(n, np.arange(n + 1), np.arange(n), 1.0 + np.arange(n, dtype=float) / n)
    Ag = _mat(ctx, Ao)
    v = np.zeros(n); v[[1, n // 2, n - 7]] = [1.0, 2.0, -1.0]; v /= np.linalg.norm(v)
    m = 24
    Vg = ks.BV(ctx, n, m + 1); Vo = O.BV(n, m + 1)
    Vg.set_column(0, v); Vo.set_column(0, v)
    Tg = np.zeros((m + 1, 3), order="F"); To = np.zeros((m + 1, 3), order="F")
    rg = Vg.MatLanczos(Ag, Tg, 0, m); ro = Vo.MatLanczos(Ao, To, 0, m)
    assert rg[0] == ro[0] == 3 and rg[2] and ro[2]
    assert np.allclose(Tg[:3], To[:3], atol=1e-12)


@pytest.mark.parametrize("n", [300001, 700000])
def test_third_pass_mid_run_keeps_the_basis_orthonormal(ctx, n):
    """The same with a start vector that leaks 1e-20 into every other direction: column 3 is what rounding leaves of
    it, needs the third pass (host completion in the middle of an enqueued run) and the run then goes on. Whatever
    that column's direction, the basis stays orthonormal and the Lanczos relation holds."""
    import slepc_amd as ks
    d = 1.0 + np.arange(n, dtype=float) / n
    Ao = O.CSR(n, np.arange(n + 1), np.arange(n), d)
    Ag = _mat(ctx, Ao)
    v = np.full(n, 1e-20); v[[1, n // 2, n - 7]] = [1.0, 2.0, -1.0]; v /= np.linalg.norm(v)
    m = 10
    Vg = ks.BV(ctx, n, m + 1)
    Vg.set_column(0, v)
    Tg = np.zeros((m + 1, 3), order="F")
    rg = Vg.MatLanczos(Ag, Tg, 0, m)
    mm = rg[0]
    Vd = Vg.dense()[:, : mm + 1]
    assert np.abs(Vd[:, :mm].T @ Vd[:, :mm] - np.eye(mm)).max() < 1e-10
    if not rg[2]:
        assert mm == m and Vg.gs_passes()[0] > 2 * m        # at least one column took the third pass
        # full reorthogonalisation: the projected matrix is V^T A V, tridiagonal or not after the junk column
        H = Vd[:, :mm].T @ (d[:, None] * Vd[:, :mm])
        assert np.allclose(np.diag(H), Tg[:mm, 0], atol=1e-10)


def test_lanczos_argument_checks(ctx):
    import slepc_amd as ks
    Ao = O.laplacian1d(20); Ag = _mat(ctx, Ao)
    V = ks.BV(ctx, 20, 6)
    T = np.zeros((6, 3), order="F")
    with pytest.raises(ks.KsError) as e:
        V.MatLanczos(Ag, T, 3, 3)            # "Argument m should be at least equal to k+1"
    assert e.value.rc == 63
    with pytest.raises(ks.KsError) as e:
        V.MatLanczos(Ag, T, 0, 6)            # needs m+1 columns
    assert e.value.rc == 63


# ---- EPS Krylov-Schur ----------------------------------------------------------------------------------------
def _solve_gpu(ctx, A, nev, ncv=0, which="largest_magnitude", tol=0.0):
    import slepc_amd as ks
    eps = ks.EPS(ctx)
    eps.SetOperators(A); eps.SetProblemType(ks.EPS_HEP); eps.SetDimensions(nev, ncv); eps.SetWhichEigenpairs(which); eps.SetTolerances(tol)
    eps.Solve()
    return eps


def test_eps_ex2_golden_and_oracle(ctx):
    """BASELINE config 1 anchor: ex2 -n 72 -eps_nev 4 -eps_ncv 20."""
    import slepc_amd as ks
    Ao = O.laplacian2d(72)
    eps = _solve_gpu(ctx, ks.Mat.laplacian2d(ctx, 72), 4, 20)
    r = O.eps_krylovschur_hep(Ao, 4, ncv=20)
    lam = np.array([eps.GetEigenvalue(i)[0] for i in range(4)])
    ref = gi.eigenvalues_line(gi.read("eps/ex2_1.out")); alt = gi.eigenvalues_line(gi.read("eps/ex2_1_alt.out"))
    assert all(min(abs(round(l, 5) - a), abs(round(l, 5) - b)) < 1.5e-5 for l, a, b in zip(lam, ref, alt))
    assert np.allclose(lam, r.eigr[r.perm][:4], rtol=1e-10)
    assert eps.GetConverged() == r.nconv and eps.GetIterationNumber() == r.its           # identical control flow
    st = eps.GetStats()
    assert st["arnoldi_steps"] == r.steps and st["gs_passes"] == r.passes
    for i in range(4):
        assert eps.ComputeError(i) < 1e-8
        assert abs(eps.ComputeError(i) - O.eps_compute_error(Ao, r, i)) < 1e-10
        x = eps.GetEigenvector(i)
        assert abs(np.linalg.norm(x) - 1.0) < 1e-12


def test_eps_config1_2d_100(ctx):
    """BASELINE config 1: 2-D 5-pt Laplacian n=10000, nev=4, m=20."""
    import slepc_amd as ks
    Ao = O.laplacian2d(100)
    eps = _solve_gpu(ctx, ks.Mat.laplacian2d(ctx, 100), 4, 20)
    r = O.eps_krylovschur_hep(Ao, 4, ncv=20)
    lam = np.array([eps.GetEigenvalue(i)[0] for i in range(eps.GetConverged())])
    assert eps.GetConverged() >= 4
    assert np.allclose(lam[:4], r.eigr[r.perm][:4], rtol=1e-10)
    exact = O.laplacian_eigenvalues([100, 100])
    for l in lam[:4]:
        assert np.min(np.abs(exact - l)) / l < 1e-10


def test_eps_ex19_smallest_real(ctx):
    import slepc_amd as ks
    eps = _solve_gpu(ctx, ks.Mat.laplacian3d(ctx, 10, 10, 10), 8, 64 - 1, which="smallest_real")
    lam = np.array([eps.GetEigenvalue(i)[0] for i in range(8)])
    ref = gi.eigenvalues_line(gi.read("eps/ex19_1.out"))
    assert np.allclose(np.round(lam, 5), ref, atol=1.5e-5)
    assert np.allclose(lam, O.laplacian_eigenvalues([10, 10, 10])[:8], rtol=1e-10)
    assert np.all(np.diff(lam) >= 0)                         # final sort is exact


def test_eps_test4_1d(ctx):
    Ao = O.laplacian1d(30)
    eps = _solve_gpu(ctx, _mat(ctx, Ao), 4)
    lam = np.array([eps.GetEigenvalue(i)[0] for i in range(4)])
    assert np.allclose(np.round(lam, 5), gi.eigenvalues_line(gi.read("eps/eps_test4_1.out")), atol=1.5e-5)
    assert eps.GetDimensions() == (4, 19, 19)                # ncv = max(2nev, nev+15)


def test_eps_diagonal_test6(ctx):
    n = 30
    Ao = O.CSR(n, np.arange(n + 1), np.arange(n), np.arange(1, n + 1, dtype=float))
    eps = _solve_gpu(ctx, _mat(ctx, Ao), 4)
    lam = np.array([eps.GetEigenvalue(i)[0] for i in range(4)])
    assert np.allclose(lam, gi.eigenvalues_line(gi.read("eps/eps_test6_1.out")), atol=1e-9)


@pytest.mark.parametrize("which", ["largest_magnitude", "smallest_real", "largest_real", "smallest_magnitude"])
def test_eps_sorting_criteria_match_oracle(ctx, which):
    Ao = O.laplacian2d(20)
    eps = _solve_gpu(ctx, _mat(ctx, Ao), 5, which=which)
    r = O.eps_krylovschur_hep(Ao, 5, which=which)
    assert eps.GetConverged() == r.nconv and eps.GetIterationNumber() == r.its
    lam = np.array([eps.GetEigenvalue(i)[0] for i in range(r.nconv)])
    assert np.allclose(lam, r.eigr[r.perm], rtol=1e-10)


def test_eps_config2_1M(ctx):
    """BASELINE config 2: 2-D Laplacian n=1e6 on one MI355X, Ritz values vs analytic spectrum + residuals."""
    import slepc_amd as ks
    eps = ks.EPS(ctx)
    eps.SetOperators(ks.Mat.laplacian2d(ctx, 1000)); eps.SetProblemType(ks.EPS_HEP); eps.SetDimensions(4, 20); eps.SetTolerances(1e-8, 400)
    eps.Solve()
    exact = O.laplacian_eigenvalues([1000, 1000])
    nconv = eps.GetConverged()
    assert eps.GetConvergedReason() in (1, -1)
    for i in range(nconv):
        lam = eps.GetEigenvalue(i)[0]
        assert np.min(np.abs(exact - lam)) / lam < 1e-10
        assert eps.ComputeError(i) < 1e-8


def test_full_size_properties_config3(ctx):
    """BASELINE config 3 size (216^3 = 10 077 696 rows): size-independent properties of one Lanczos run:
    orthonormal basis, T symmetric-tridiagonal relation via Rayleigh quotients, Ritz values inside [0,12]."""
    import slepc_amd as ks
    N = 216
    A = ks.Mat.laplacian3d(ctx, N, N, N)
    assert A.n == 10077696 and A.nnz == 70263936
    m = 30
    V = ks.BV(ctx, A.n, m + 1)
    _start(V)
    T = np.zeros((m + 1, 3), order="F")
    mm, beta, brk = V.MatLanczos(A, T, 0, m)
    assert mm == m and not brk
    M = np.zeros((m + 1, m + 1), order="F")
    V.SetActiveColumns(0, m + 1)
    V.Dot(V, M)
    assert np.abs(M - np.eye(m + 1)).max() < 1e-12
    # alpha_j = v_j' A v_j and beta_j = v_{j+1}' A v_j, checked with an independent SpMV + dot on 3 columns
    W = ks.BV(ctx, A.n, 2)
    for j in (0, 13, 29):
        A.mult_dev(V.column_ptr(j), W.column_ptr(0))
        V.SetActiveColumns(j, j + 2)
        d = V.DotVec(W.column_ptr(0))
        assert abs(d[0] - T[j, 0]) < 1e-11 and abs(d[1] - T[j, 1]) < 1e-11
    Tm = np.diag(T[:m, 0]) + np.diag(T[: m - 1, 1], 1) + np.diag(T[: m - 1, 1], -1)
    th = np.linalg.eigvalsh(Tm)
    assert th.min() > 0 and th.max() < 12.0
    assert V.gs_passes()[0] >= m


def test_config3_first_lanczos_run_matches_the_cpu_oracle_at_full_size(ctx):
    """The first Lanczos run (30 steps, CGS with refinement) of BASELINE config 3 at its full size on the GPU and on the CPU
    oracle (OpenMP build, team sized to the cores the box grants): the tridiagonal coefficients agree to 1e-10 relative,
    the Gram-Schmidt pass counts are identical, sampled basis rows agree."""
    import slepc_amd as ks
    N, m = 216, 30
    O.lib(omp=True).orc_set_num_threads(O.usable_cores())
    Ao = O.laplacian3d(N, N, N, omp=True)
    Vo = O.BV(Ao.n, m + 1, omp=True)
    _start(Vo)
    To = np.zeros((m + 1, 3), order="F")
    mo, bo, brko = Vo.MatLanczos(Ao, To, 0, m)
    A = ks.Mat.laplacian3d(ctx, N, N, N)
    V = ks.BV(ctx, A.n, m + 1)
    _start(V)
    T = np.zeros((m + 1, 3), order="F")
    mm, beta, brk = V.MatLanczos(A, T, 0, m)
    assert (mm, brk) == (mo, brko) == (m, False)
    assert np.allclose(T[:m, :2], To[:m, :2], rtol=1e-10, atol=0) and abs(beta - bo) <= 1e-10 * abs(bo)
    assert V.gs_passes()[0] == Vo.passes_total()
    rows = np.random.default_rng(0).integers(0, A.n, 2000)
    for j in (0, 1, 15, 30):
        assert np.allclose(V.column(j)[rows], np.array(Vo.column(j))[rows], rtol=0, atol=1e-12)


def test_config3_solved_to_convergence(ctx):
    """BASELINE config 3 itself, solved to convergence (about 7 300 Arnoldi steps, 8 s): every converged Ritz value lies
    within 1e-10 (relative) of the analytic spectrum of the 216^3 Laplacian (ex19.c:19-45) - measured: 1e-13 - the triple
    eigenvalue 11.99874... is found three times, residuals are below the tolerance."""
    import slepc_amd as ks
    N = 216
    eps = ks.EPS(ctx)
    eps.SetOperators(ks.Mat.laplacian3d(ctx, N, N, N)); eps.SetProblemType(ks.EPS_HEP); eps.SetDimensions(10, 30); eps.SetTolerances(1e-8, 3000)
    eps.Solve()
    assert eps.GetConvergedReason() == ks.EPS_CONVERGED_TOL and eps.GetConverged() >= 10
    s1 = np.sort(4.0 * np.sin(np.arange(1, N + 1) * np.pi / (2.0 * (N + 1))) ** 2)[::-1][:12]
    exact = np.sort((s1[:, None, None] + s1[None, :, None] + s1[None, None, :]).ravel())[::-1]
    lam = np.array([eps.GetEigenvalue(i)[0] for i in range(eps.GetConverged())])
    assert np.all(np.diff(lam) <= 1e-9)                                    # largest magnitude first
    for i, l in enumerate(lam):
        assert np.min(np.abs(exact - l)) / l < 1e-10
        assert eps.ComputeError(i) < 2e-8
    assert np.allclose(lam[:4], exact[:4], rtol=1e-10)                       # the simple top eigenvalue and all three copies of the next


def test_eps_test2_repeated_solves_one_object(ctx):
    """test2.c: one EPS object solved three times with changing criteria (largest real, smallest real, then closest to the
    target 2.1 through shift-and-invert added between solves); golden output/test2_1.out."""
    import slepc_amd as ks
    from test_oracle_golden import _test2_sections
    ref = _test2_sections()
    Ao = O.laplacian1d(30)
    eps = ks.EPS(ctx)
    eps.SetOperators(_mat(ctx, Ao)); eps.SetProblemType(ks.EPS_HEP); eps.SetDimensions(4)
    eps.SetWhichEigenpairs("largest_real"); eps.Solve()
    assert np.allclose(np.round([eps.GetEigenvalue(i)[0] for i in range(4)], 5), ref[0], atol=1.5e-5)
    eps.SetWhichEigenpairs("smallest_real"); eps.Solve()
    assert np.allclose(np.round([eps.GetEigenvalue(i)[0] for i in range(4)], 5), ref[1], atol=1.5e-5)
    eps.SetWhichEigenpairs("target_magnitude"); eps.SetTarget(2.1)
    st = eps.GetST(); st.SetType("sinvert"); st.SetKSP(rtol=1e-13, restart=30)      # GMRES(30) is exact on this 30 x 30 problem
    eps.Solve()
    lam = [eps.GetEigenvalue(i)[0] for i in range(4)]
    assert np.allclose(np.round(lam, 5), ref[2], atol=1.5e-5)
    for i in range(4):
        assert eps.ComputeError(i) < 1e-7
    with pytest.raises(ks.KsError):                       # results belong to the last solve only
        eps.SetWhichEigenpairs("largest_real"); eps.GetEigenvalue(0)


def test_c_program_against_the_abi(tmp_path):
    """tests/c_abi/ex2_abi.c: the reference's ex2 written in C99 against include/ksgpu.h only; its output line matches
    output/ex2_1.out (or the _alt file)."""
    import subprocess
    from test_abi import _build_c_example
    exe = _build_c_example(tmp_path)
    r = subprocess.run([exe, "72"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lam = gi.eigenvalues_line(r.stdout)
    ref = gi.eigenvalues_line(gi.read("eps/ex2_1.out")); alt = gi.eigenvalues_line(gi.read("eps/ex2_1_alt.out"))
    assert len(lam) == 4 and all(min(abs(l - a), abs(l - b)) < 1.5e-5 for l, a, b in zip(lam, ref, alt))


@pytest.mark.parametrize("shape", [("2d", 300), ("3d", 40), ("2d_odd", 301)])
def test_spmv_fused_into_the_dot_sweep_gives_the_same_bits(ctx, debug, shape):
    """Small problems (basis resident in the Infinity Cache, dictionary layout) run y = A x inside the dot sweep that follows it
    (k_dot_spmv_dict): same entry order and fma chain for y, same tiles and grid for the dots - the Lanczos coefficients and the basis are
    bit for bit those of the separate launches (test hook no_spmv_dot), and both match the oracle."""
    import slepc_amd as ks
    kind, N = shape
    if kind.startswith("2d"):
        mk = lambda: ks.Mat.laplacian2d(ctx, N); Ao = O.laplacian2d(N)        # noqa: E731
    else:
        mk = lambda: ks.Mat.laplacian3d(ctx, N, N, N); Ao = O.laplacian3d(N, N, N)      # noqa: E731
    m = 14
    outs = []
    for fused in (True, False):
        if not fused:
            debug("no_spmv_dot")
        A = mk(); assert A.layout() == "dict"
        V = ks.BV(ctx, A.n, m + 1)
        V.SetRandomColumn(0)
        _, nrm, _ = V.OrthogonalizeColumn(0); V.ScaleColumn(0, 1.0 / nrm)
        T = np.zeros((m + 1, 3), order="F")
        ctx.prof_enable(True); ctx.prof_reset()
        r = V.MatLanczos(A, T, 0, m)
        ctx.synchronize()
        p = ctx.prof_get(); ctx.prof_enable(False)
        outs.append((T.copy(), V.dense(), r, p))
    assert outs[0][3].get("spmv_dot_fused", {}).get("launches", 0) == m and outs[0][3].get("spmv_csr", {}).get("launches", 0) == 0
    assert outs[1][3].get("spmv_dot_fused", {}).get("launches", 0) == 0 and outs[1][3].get("spmv_csr", {}).get("launches", 0) == m
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]) and outs[0][2] == outs[1][2]
    Vo = O.BV(Ao.n, m + 1); Vo.SetRandomColumn(0)
    _, nrm, _ = Vo.OrthogonalizeColumn(0); Vo.ScaleColumn(0, 1.0 / nrm)
    To = np.zeros((m + 1, 3), order="F")
    Vo.MatLanczos(Ao, To, 0, m)
    assert np.allclose(outs[0][0][:m, :2], To[:m, :2], rtol=1e-11, atol=1e-12)
