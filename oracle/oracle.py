"""
CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this module.
Nothing under slepc_amd/ imports it; the product path fails loudly without the HIP library.

What it is: a CPU restatement of the SLEPc 3.22.2 EPS Krylov-Schur path
  * heavy kernels (CSR SpMV, BV ops, Gram-Schmidt, BVMatLanczos/Arnoldi): oracle/ks_oracle.c via ctypes
  * host dense step (DS HEP: DSArrowTridiag + LAPACK steqr, sort, extra row, truncate) and the
    restart driver (EPSSolve_KrylovSchur_Default, EPSKrylovConvergence): numpy restatement here,
    calling the SAME LAPACK routines the reference calls (dsteqr/dlartg/drot through
    scipy.linalg.cython_lapack's C entry points).
Citations are relative to /root/reference.

Pinning: PETSc/SLEPc cannot be built here (no PETSc in the image), so the oracle
is pinned by the reference's own golden outputs (tests/golden/*.out, copied from
src/sys/classes/bv/tests/output and src/eps/*/output) and by the analytic Laplacian spectra
(src/eps/tutorials/ex19.c:19-45) -- see tests/test_oracle_golden.py.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

CGS, MGS = 0, 1
REFINE_IFNEEDED, REFINE_NEVER, REFINE_ALWAYS = 0, 1, 2
NORM_1, NORM_2, NORM_FROBENIUS, NORM_INFINITY = 0, 1, 2, 3

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def _build_if_needed():
    src = os.path.join(_HERE, "ks_oracle.c")
    for so in ("liboracle.so", "liboracle_omp.so"):
        p = os.path.join(_HERE, so)
        if not os.path.exists(p) or os.path.getmtime(p) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, so], stdout=subprocess.DEVNULL)


def _load(name):
    if name == "liboracle_asan.so":
        subprocess.check_call(["make", "-C", _HERE, "asan"], stdout=subprocess.DEVNULL)
    _build_if_needed()
    lib = C.CDLL(os.path.join(_HERE, name))
    vp = C.c_void_p
    sig = {
        "orc_bv_create": (vp, [C.c_int, C.c_int, C.c_int]),
        "orc_bv_destroy": (None, [vp]),
        "orc_bv_array": (_dp, [vp]), "orc_bv_buffer": (_dp, [vp]), "orc_bv_column": (_dp, [vp, C.c_int]),
        "orc_bv_ld": (C.c_int, [vp]),
        "orc_bv_set_active": (None, [vp, C.c_int, C.c_int]),
        "orc_bv_set_orthog": (None, [vp, C.c_int, C.c_int, C.c_double]),
        "orc_bv_set_matrix": (None, [vp, vp]),
        "orc_bv_passes_last": (C.c_int, [vp]), "orc_bv_passes_total": (C.c_long, [vp]),
        "orc_bv_mult": (C.c_int, [vp, C.c_double, C.c_double, vp, _dp, C.c_int]),
        "orc_bv_multvec": (C.c_int, [vp, C.c_double, C.c_double, _dp, _dp]),
        "orc_bv_multcolumn": (C.c_int, [vp, C.c_double, C.c_double, C.c_int, _dp]),
        "orc_bv_multinplace": (C.c_int, [vp, _dp, C.c_int, C.c_int, C.c_int, C.c_int]),
        "orc_bv_dot": (C.c_int, [vp, vp, _dp, C.c_int]),
        "orc_bv_dotvec": (C.c_int, [vp, _dp, _dp]),
        "orc_bv_dotcolumn": (C.c_int, [vp, C.c_int, _dp]),
        "orc_bv_scale": (C.c_int, [vp, C.c_int, C.c_double]),
        "orc_bv_norm": (C.c_int, [vp, C.c_int, C.c_int, _dp]),
        "orc_bv_copy": (C.c_int, [vp, vp]), "orc_bv_copycolumn": (C.c_int, [vp, C.c_int, C.c_int]),
        "orc_bv_setrandomcolumn": (C.c_int, [vp, C.c_int, C.c_uint64, C.c_int]),
        "orc_random_value": (C.c_double, [C.c_uint64, C.c_uint64, C.c_uint64]),
        "orc_csr_mult": (None, [C.c_int, _ip, _ip, _dp, _dp, _dp]),
        "orc_csr_wrap": (vp, [C.c_int, C.c_int, _ip, _ip, _dp]), "orc_csr_free": (None, [vp]),
        "orc_bv_matmultcolumn": (C.c_int, [vp, vp, C.c_int]),
        "orc_bv_matmult": (C.c_int, [vp, vp, vp]),
        "orc_bv_orthogonalizevec": (C.c_int, [vp, _dp, _dp, _dp, _ip]),
        "orc_bv_orthogonalizecolumn": (C.c_int, [vp, C.c_int, _dp, _dp, _ip]),
        "orc_bv_orthogonalizesomecolumn": (C.c_int, [vp, C.c_int, _ip, _dp, _dp, _ip]),
        "orc_bv_orthonormalizecolumn": (C.c_int, [vp, C.c_int, _dp, _ip]),
        "orc_bv_matarnoldi": (C.c_int, [vp, vp, _dp, C.c_int, C.c_int, _ip, _dp, _ip]),
        "orc_bv_matlanczos": (C.c_int, [vp, vp, _dp, C.c_int, C.c_int, _ip, _dp, _ip]),
        "orc_bv_insert_vecs": (C.c_int, [vp, C.c_int, _ip, _dp, C.c_int, C.c_int]),
        "orc_bv_insert_constraints": (C.c_int, [vp, _ip, _dp, C.c_int]),
        "orc_bv_set_num_constraints": (C.c_int, [vp, C.c_int]),
        "orc_bv_get_num_constraints": (C.c_int, [vp]),
        "orc_num_threads": (C.c_int, []),
        "orc_set_num_threads": (None, [C.c_int]),
        "orc_laplacian3d_nnz": (C.c_long, [C.c_int] * 5),
        "orc_laplacian3d_fill": (None, [C.c_int] * 5 + [_ip, _ip, _dp]),
        "orc_laplacian2d_nnz": (C.c_long, [C.c_int, C.c_int]),
        "orc_laplacian2d_fill": (None, [C.c_int, C.c_int, _ip, _ip, _dp]),
    }
    for k, (res, args) in sig.items():
        f = getattr(lib, k)
        f.restype = res
        f.argtypes = args
    return lib


_libs = {}


def lib(omp=False):
    name = "liboracle_omp.so" if omp else "liboracle.so"
    if not omp and os.environ.get("ORACLE_LIB") == "asan":       # tests/test_sanitizers.py: the single-thread oracle under ASan + UBSan (make -C oracle asan)
        name = "liboracle_asan.so"
    if name not in _libs:
        _libs[name] = _load(name)
    return _libs[name]


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _pi(a):
    return None if a is None else a.ctypes.data_as(_ip)


def usable_cores():
    """Host cores this process may really use: the smallest of the CPU count, the affinity mask and the cgroup CPU quota
    (cpu.max); an OpenMP team larger than the quota is throttled at its barriers."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            q, p = open(path).read().split()[:2]
            if q != "max":
                n = min(n, max(1, int(float(q) / float(p) + 0.5)))
        except (OSError, ValueError):
            pass
    return n


class OracleError(RuntimeError):
    pass


def _chk(ierr):
    if ierr:
        raise OracleError({1: "argument out of range / wrong", 2: "Invalid inner product (BV_SafeSqrt)"}.get(ierr, str(ierr)))


# ------------------------------------------------------------------------------------------------
# matrices


class CSR:
    """PETSc SeqAIJ-layout CSR (rowptr int32[n+1], col int32[nnz], val float64[nnz])."""

    def __init__(self, n, rowptr, col, val, ncols=None, omp=False):
        self.n = int(n)
        self.ncols = int(ncols if ncols is not None else n)
        self.rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
        self.col = np.ascontiguousarray(col, dtype=np.int32)
        self.val = np.ascontiguousarray(val, dtype=np.float64)
        self.nnz = int(self.rowptr[-1])
        self._lib = lib(omp)
        self._h = self._lib.orc_csr_wrap(self.n, self.ncols, _pi(self.rowptr), _pi(self.col), _p(self.val))

    def __del__(self):
        try:
            self._lib.orc_csr_free(self._h)
        except Exception:
            pass

    def mult(self, x, y=None):
        x = np.ascontiguousarray(x, dtype=np.float64)
        if y is None:
            y = np.empty(self.n)
        self._lib.orc_csr_mult(self.n, _pi(self.rowptr), _pi(self.col), _p(self.val), _p(x), _p(y))
        return y

    def to_scipy(self):
        import scipy.sparse as sp
        return sp.csr_matrix((self.val, self.col, self.rowptr), shape=(self.n, self.ncols))


def laplacian3d(nx, ny, nz, z0=0, nzl=None, omp=False):
    """ex19.c:47-78 FillMatrix: 7-pt, diag 6, off -1, natural ordering (x fastest). Rows of planes z0..z0+nzl."""
    L = lib(omp)
    nzl = nz if nzl is None else nzl
    n = nx * ny * nzl
    nnz = L.orc_laplacian3d_nnz(nx, ny, nz, z0, nzl)
    rowptr = np.empty(n + 1, np.int32)
    col = np.empty(nnz, np.int32)
    val = np.empty(nnz, np.float64)
    L.orc_laplacian3d_fill(nx, ny, nz, z0, nzl, _pi(rowptr), _pi(col), _p(val))
    return CSR(n, rowptr, col, val, ncols=nx * ny * nz, omp=omp)


def laplacian2d(n, m=None, omp=False):
    """ex2.c:44-51: 5-pt, diag 4, off -1, II=i*n+j."""
    L = lib(omp)
    m = n if m is None else m
    N = n * m
    nnz = L.orc_laplacian2d_nnz(n, m)
    rowptr = np.empty(N + 1, np.int32)
    col = np.empty(nnz, np.int32)
    val = np.empty(nnz, np.float64)
    L.orc_laplacian2d_fill(n, m, _pi(rowptr), _pi(col), _p(val))
    return CSR(N, rowptr, col, val, omp=omp)


def laplacian1d(n):
    """ex1.c / test4.c: tridiag(-1,2,-1)."""
    rowptr = [0]
    col = []
    val = []
    for i in range(n):
        if i > 0:
            col.append(i - 1); val.append(-1.0)
        col.append(i); val.append(2.0)
        if i < n - 1:
            col.append(i + 1); val.append(-1.0)
        rowptr.append(len(col))
    return CSR(n, rowptr, col, val)


def laplacian_eigenvalues(dims):
    """Analytic Dirichlet Laplacian spectrum 4*sum_d sin^2(i_d*pi/(2(N_d+1)))  (ex19.c:19-45)."""
    ev = np.zeros(1)
    for N in dims:
        s = 4.0 * np.sin(np.arange(1, N + 1) * np.pi / (2.0 * (N + 1))) ** 2
        ev = (ev[:, None] + s[None, :]).ravel()
    return np.sort(ev)


# ------------------------------------------------------------------------------------------------
# BV


class BV:
    """Mirror of the BV interface slice used by the path (names follow slepcbv.h without the BV prefix)."""

    def __init__(self, n, m, ld=0, omp=False):
        self._lib = lib(omp)
        self._h = self._lib.orc_bv_create(n, m, ld)
        self.n, self.m = n, m
        self.ld = self._lib.orc_bv_ld(self._h)
        self.l, self.k = 0, m
        self.nc = 0
        self._remap()

    def _remap(self):
        """numpy views of the storage: `array` = the regular columns (ld, m), `constraints` = columns -nc..-1,
        `buffer` = the (nc+m, m) coefficient buffer, all column-major."""
        m, nc, ld = self.m, self.nc, self.ld
        arr = self._lib.orc_bv_array(self._h)
        full = np.ctypeslib.as_array(arr, shape=((nc + m) * ld,)).reshape(nc + m, ld).T
        self.constraints = full[:, :nc]
        self.array = full[:, nc:]
        buf = self._lib.orc_bv_buffer(self._h)
        self.buffer = np.ctypeslib.as_array(buf, shape=((nc + m) * m,)).reshape(m, nc + m).T

    def constraints_dense(self):
        return np.array(self.constraints[: self.n, :])

    def InsertVecs(self, s, W, orth=True):
        """BVInsertVecs(V,s,&m,W,orth): returns the number of vectors kept."""
        Wf = np.asfortranarray(W, dtype=np.float64)
        m = np.array([Wf.shape[1]], np.int32)
        _chk(self._lib.orc_bv_insert_vecs(self._h, s, _pi(m), _p(Wf), Wf.shape[0], int(bool(orth))))
        return int(m[0])

    def InsertConstraints(self, Cmat):
        """BVInsertConstraints(V,&nc,C) with the vectors as the columns of Cmat: returns the number kept."""
        Cf = np.asfortranarray(Cmat, dtype=np.float64)
        nc = np.array([Cf.shape[1]], np.int32)
        _chk(self._lib.orc_bv_insert_constraints(self._h, _pi(nc), _p(Cf), Cf.shape[0]))
        self.nc = int(nc[0])
        self.l, self.k = 0, self.m
        self._remap()
        return self.nc

    def SetNumConstraints(self, nc):
        _chk(self._lib.orc_bv_set_num_constraints(self._h, nc))
        self.m = self.nc + self.m - nc
        self.nc = nc
        self.l, self.k = min(self.l, self.m), min(self.k, self.m)
        self._remap()

    def __del__(self):
        try:
            self._lib.orc_bv_destroy(self._h)
        except Exception:
            pass

    # -- layout
    def column(self, j):
        return self.array[: self.n, j]

    def set_column(self, j, x):
        self.array[: self.n, j] = x

    def dense(self):
        return np.array(self.array[: self.n, :])

    def SetActiveColumns(self, l, k):
        self.l, self.k = l, k
        self._lib.orc_bv_set_active(self._h, l, k)

    def SetOrthogonalization(self, type=CGS, refine=REFINE_IFNEEDED, eta=0.7071):
        self._lib.orc_bv_set_orthog(self._h, type, refine, eta)

    _B = None

    def SetMatrix(self, B):
        """BVSetMatrix(bv,B,PETSC_FALSE): B a CSR (kept alive here) or None."""
        self._B = B
        self._lib.orc_bv_set_matrix(self._h, B._h if B is not None else None)

    def SetRandomColumn(self, j, seed=0x12345678, row0=0):
        _chk(self._lib.orc_bv_setrandomcolumn(self._h, j, seed, row0))

    # -- ops
    def Mult(self, alpha, beta, X, Q=None):
        """Y(self) = beta*Y + alpha*X*Q."""
        if Q is None:
            _chk(self._lib.orc_bv_mult(self._h, alpha, beta, X._h, None, 0))
        else:
            Qf = np.asfortranarray(Q, dtype=np.float64)
            _chk(self._lib.orc_bv_mult(self._h, alpha, beta, X._h, _p(Qf), Qf.shape[0]))

    def MultVec(self, alpha, beta, y, q=None):
        q = None if q is None else np.ascontiguousarray(q, dtype=np.float64)
        _chk(self._lib.orc_bv_multvec(self._h, alpha, beta, _p(y), _p(q)))

    def MultColumn(self, alpha, beta, j, q=None):
        q = None if q is None else np.ascontiguousarray(q, dtype=np.float64)
        _chk(self._lib.orc_bv_multcolumn(self._h, alpha, beta, j, _p(q)))

    def MultInPlace(self, Q, s, e, trans=False):
        Qf = np.asfortranarray(Q, dtype=np.float64)
        _chk(self._lib.orc_bv_multinplace(self._h, _p(Qf), Qf.shape[0], s, e, int(trans)))

    def Dot(self, Y, M):
        """M = Y^H * X(self); M is a Fortran-ordered array with >= Y.k rows and >= X.k columns."""
        assert M.flags.f_contiguous
        _chk(self._lib.orc_bv_dot(self._h, Y._h, _p(M), M.shape[0]))

    def DotVec(self, y, m=None):
        y = np.ascontiguousarray(y, dtype=np.float64)
        out = np.zeros(self.k - self.l) if m is None else m
        _chk(self._lib.orc_bv_dotvec(self._h, _p(y), _p(out)))
        return out

    def DotColumn(self, j, q=True):
        """q=None -> result goes to the buffer scratch (BVDotColumn(X,j,NULL))."""
        if q is None:
            _chk(self._lib.orc_bv_dotcolumn(self._h, j, None))
            return None
        out = np.zeros(j - self.l)
        _chk(self._lib.orc_bv_dotcolumn(self._h, j, _p(out)))
        return out

    def Scale(self, alpha):
        _chk(self._lib.orc_bv_scale(self._h, -1, alpha))

    def ScaleColumn(self, j, alpha):
        _chk(self._lib.orc_bv_scale(self._h, j, alpha))

    def Norm(self, type=NORM_FROBENIUS):
        v = C.c_double()
        _chk(self._lib.orc_bv_norm(self._h, -1, type, C.byref(v)))
        return v.value

    def NormColumn(self, j, type=NORM_2):
        v = C.c_double()
        _chk(self._lib.orc_bv_norm(self._h, j, type, C.byref(v)))
        return v.value

    def Copy(self, W):
        _chk(self._lib.orc_bv_copy(self._h, W._h))

    def CopyColumn(self, j, i):
        _chk(self._lib.orc_bv_copycolumn(self._h, j, i))

    def MatMultColumn(self, A, j):
        _chk(self._lib.orc_bv_matmultcolumn(self._h, A._h, j))

    def MatMult(self, A, W):
        _chk(self._lib.orc_bv_matmult(self._h, A._h, W._h))

    def OrthogonalizeVec(self, v):
        nrm = C.c_double(); lin = C.c_int()
        H = np.zeros(self.k - self.l)
        _chk(self._lib.orc_bv_orthogonalizevec(self._h, _p(v), _p(H), C.byref(nrm), C.byref(lin)))
        return H, nrm.value, bool(lin.value)

    def OrthogonalizeColumn(self, j):
        nrm = C.c_double(); lin = C.c_int()
        H = np.zeros(max(j - self.l, 0) + 1)
        _chk(self._lib.orc_bv_orthogonalizecolumn(self._h, j, _p(H), C.byref(nrm), C.byref(lin)))
        return H[: j - self.l], nrm.value, bool(lin.value)

    def OrthogonalizeSomeColumn(self, j, which):
        nrm = C.c_double(); lin = C.c_int()
        w = np.ascontiguousarray(which, dtype=np.int32)
        _chk(self._lib.orc_bv_orthogonalizesomecolumn(self._h, j, _pi(w), None, C.byref(nrm), C.byref(lin)))
        return nrm.value, bool(lin.value)

    def OrthonormalizeColumn(self, j):
        nrm = C.c_double(); lin = C.c_int()
        _chk(self._lib.orc_bv_orthonormalizecolumn(self._h, j, C.byref(nrm), C.byref(lin)))
        return nrm.value, bool(lin.value)

    def passes_last(self):
        return self._lib.orc_bv_passes_last(self._h)

    def passes_total(self):
        return self._lib.orc_bv_passes_total(self._h)

    def MatLanczos(self, A, T, k, m):
        """T: Fortran array (ldt, >=2): column 0 = alpha, column 1 = beta (DS_MAT_T). Returns (m, beta, breakdown)."""
        assert T.flags.f_contiguous
        mm = C.c_int(m); beta = C.c_double(); brk = C.c_int()
        _chk(self._lib.orc_bv_matlanczos(self._h, A._h, _p(T), T.shape[0], k, C.byref(mm), C.byref(beta), C.byref(brk)))
        return mm.value, beta.value, bool(brk.value)

    def MatArnoldi(self, A, H, k, m):
        assert H.flags.f_contiguous
        mm = C.c_int(m); beta = C.c_double(); brk = C.c_int()
        _chk(self._lib.orc_bv_matarnoldi(self._h, A._h, _p(H), H.shape[0], k, C.byref(mm), C.byref(beta), C.byref(brk)))
        return mm.value, beta.value, bool(brk.value)

    def Orthogonalize(self, R=None, block="gs"):
        """BVOrthogonalize (bvorthog.c:729-767) with the block methods of bvorthog.c:510-690 and the LAPACK kernels of
        bvlapack.c:136-451 (potrf/trtri, syev, geqrf/orgqr through scipy.linalg.lapack; single process, so TSQR is one
        geqrf). R: Fortran-ordered (>=k, >=k) array or None; columns l..k-1 are written."""
        import scipy.linalg.lapack as la
        l, k, n = self.l, self.k, self.n
        A = self.array
        if k <= l:
            return
        if block == "gs":
            for j in range(l, k):
                self.SetActiveColumns(0, k)                     # V->l = -V->nc around BV_StoreCoefficients (:536-539)
                H, norm, _ = self.OrthogonalizeColumn(j)
                self.SetActiveColumns(l, k)
                if R is not None:
                    R[:j, j] = H[:j]; R[j, j] = norm
                if norm == 0.0:
                    raise RuntimeError("Breakdown in BVOrthogonalize due to a linearly dependent column")
                self.ScaleColumn(j, 1.0 / norm)
            return
        Bm = getattr(self, "_B", None)
        ip = (lambda Xm: Bm.to_scipy() @ Xm) if Bm is not None else (lambda Xm: Xm)     # BVDot with a matrix: Y^H (B X)
        if Bm is not None and block in ("tsqr", "tsqrchol"):
            raise RuntimeError("Orthogonalization method not available for non-standard inner product")
        Rb = np.zeros((k, k), order="F")
        if l:                                                   # BVOrthogonalize_BlockGS :492-505
            Rb[:l, l:k] = A[:n, :l].T @ ip(A[:n, l:k])
            A[:n, l:k] -= A[:n, :l] @ Rb[:l, l:k]
        V2 = A[:n, l:k]
        if block == "chol":
            G = V2.T @ ip(V2)
            c, info = la.dpotrf(G, lower=0)
            if info:                                            # bvlapack.c:177-185
                c, info = la.dpotrf(G + 50.0 * np.finfo(float).eps * np.eye(k - l), lower=0)
                assert info == 0
            c = np.triu(c)
            S, info = la.dtrtri(c, lower=0); assert info == 0
            A[:n, l:k] = V2 @ np.triu(S)
            Rb[l:k, l:k] = c; tri = True
        elif block == "svqb":
            G = V2.T @ ip(V2)
            D = 1.0 / np.sqrt(np.diag(G))
            w, U, info = la.dsyev(G * D[:, None] * D[None, :], lower=1); assert info == 0
            A[:n, l:k] = V2 @ (D[:, None] * U / np.sqrt(w)[None, :])
            Rb[l:k, l:k] = np.sqrt(w)[:, None] * U.T / D[None, :]; tri = False
        elif block in ("tsqr", "tsqrchol"):
            qr, tau, _, info = la.dgeqrf(np.asfortranarray(V2)); assert info == 0
            Rr = np.triu(qr[: k - l, :])
            if block == "tsqr":
                Q, _, info = la.dorgqr(qr[:, : k - l], tau); assert info == 0
                A[:n, l:k] = Q[:, : k - l]
            else:
                S, info = la.dtrtri(Rr, lower=0); assert info == 0
                A[:n, l:k] = V2 @ np.triu(S)
            Rb[l:k, l:k] = Rr; tri = True
        else:
            raise ValueError(block)
        if R is not None:                                       # BV_StoreCoeffsBlock_Default :576-596
            for j in range(l, k):
                rows = j + 1 if tri else k
                R[:rows, j] = Rb[:rows, j]

    def MatLanczosOp(self, op, T, k, m):
        """BVMatLanczos (bvkrylov.c:165-226) with the operator given as a callable."""
        brk = False; beta = 0.0
        for j in range(k, m):
            self.set_column(j + 1, op(np.array(self.column(j))))
            beta, brk = self.OrthonormalizeColumn(j + 1)
            if brk:
                m = j + 1
                break
        for j in range(k, m):
            T[j, 0] = self.buffer[self.nc + j, j + 1]; T[j, 1] = self.buffer[self.nc + j + 1, j + 1]
        return m, beta, brk

    def MatArnoldiOp(self, op, H, k, m):
        """BVMatArnoldi (bvkrylov.c:56-113) with the operator given as a callable y = op(x) (an ST operator)."""
        brk = False; beta = 0.0
        for j in range(k, m):
            self.set_column(j + 1, op(np.array(self.column(j))))
            beta, brk = self.OrthonormalizeColumn(j + 1)
            if brk:
                m = j + 1
                break
        for j in range(k, m - 1):
            H[: j + 2, j] = self.buffer[self.nc: self.nc + j + 2, j + 1]
        H[:m, m - 1] = self.buffer[self.nc: self.nc + m, m]
        if H.shape[0] > m:
            H[m, m - 1] = self.buffer[self.nc + m, m]
        return m, beta, brk


# ------------------------------------------------------------------------------------------------
# LAPACK entry points (the very routines the reference calls), taken from scipy's bundled LAPACK


def _capi(mod, name, restype, argtypes):
    cap = mod.__pyx_capi__[name]
    C.pythonapi.PyCapsule_GetName.restype = C.c_char_p
    C.pythonapi.PyCapsule_GetName.argtypes = [C.py_object]
    C.pythonapi.PyCapsule_GetPointer.restype = C.c_void_p
    C.pythonapi.PyCapsule_GetPointer.argtypes = [C.py_object, C.c_char_p]
    ptr = C.pythonapi.PyCapsule_GetPointer(cap, C.pythonapi.PyCapsule_GetName(cap))
    return C.CFUNCTYPE(restype, *argtypes)(ptr)


_lapack = {}


def _L(name):
    if not _lapack:
        import scipy.linalg.cython_lapack as CL
        import scipy.linalg.cython_blas as CB
        cp = C.c_char_p
        _lapack["dsteqr"] = _capi(CL, "dsteqr", None, [cp, _ip, _dp, _dp, _dp, _ip, _dp, _ip])
        _lapack["dlartg"] = _capi(CL, "dlartg", None, [_dp, _dp, _dp, _dp, _dp])
        _lapack["drot"] = _capi(CB, "drot", None, [_ip, _dp, _ip, _dp, _ip, _dp, _dp])
        _lapack["dgehrd"] = _capi(CL, "dgehrd", None, [_ip, _ip, _ip, _dp, _ip, _dp, _dp, _ip, _ip])
        _lapack["dorghr"] = _capi(CL, "dorghr", None, [_ip, _ip, _ip, _dp, _ip, _dp, _dp, _ip, _ip])
        _lapack["dhseqr"] = _capi(CL, "dhseqr", None, [cp, cp, _ip, _ip, _ip, _dp, _ip, _dp, _dp, _dp, _ip, _dp, _ip, _ip])
        _lapack["dtrexc"] = _capi(CL, "dtrexc", None, [cp, _ip, _dp, _ip, _dp, _ip, _ip, _ip, _dp, _ip])
        _lapack["dtrevc"] = _capi(CL, "dtrevc", None, [cp, cp, _ip, _ip, _dp, _ip, _dp, _ip, _dp, _ip, _ip, _ip, _dp, _ip])
    return _lapack[name]


def _i(v):
    return C.byref(C.c_int(v))


def lartg(f, g):
    cs = C.c_double(); sn = C.c_double(); r = C.c_double()
    _L("dlartg")(C.byref(C.c_double(f)), C.byref(C.c_double(g)), C.byref(cs), C.byref(sn), C.byref(r))
    return cs.value, sn.value, r.value


# comparison functions, src/sys/slepcsc.c:152-300 (real scalars: |a| via SlepcAbsEigenvalue = hypot(re,im))
def _cmp(a, b):
    return 1 if a < b else (-1 if a > b else 0)


WHICH = {
    "largest_magnitude": lambda ar, ai, br, bi: _cmp(np.hypot(ar, ai), np.hypot(br, bi)),
    "smallest_magnitude": lambda ar, ai, br, bi: -_cmp(np.hypot(ar, ai), np.hypot(br, bi)),
    "largest_real": lambda ar, ai, br, bi: _cmp(ar, br),
    "smallest_real": lambda ar, ai, br, bi: -_cmp(ar, br),
    "largest_imaginary": lambda ar, ai, br, bi: _cmp(abs(ai), abs(bi)),
    "smallest_imaginary": lambda ar, ai, br, bi: -_cmp(abs(ai), abs(bi)),
}


def which_target_magnitude(t):         # SlepcCompareTargetMagnitude slepcsc.c:233-247
    return lambda ar, ai, br, bi: -_cmp(np.hypot(ar - t, ai), np.hypot(br - t, bi))


def which_target_real(t):              # SlepcCompareTargetReal slepcsc.c:249-263
    return lambda ar, ai, br, bi: -_cmp(abs(ar - t), abs(br - t))

DS_STATE_RAW, DS_STATE_INTERMEDIATE, DS_STATE_CONDENSED, DS_STATE_TRUNCATED = 0, 1, 2, 3


class DSHEP:
    """DS type HEP, compact storage, extra row (krylovschur.c:160-168). T = [d | e] (dshep.c:26-48 picture)."""

    def __init__(self, ld, compare):
        self.ld = ld
        self.T = np.zeros((ld, 3), order="F")   # DS_MAT_T: col 0 diag, col 1 offdiag
        self.Q = np.zeros((ld, ld), order="F")
        self.perm = np.zeros(ld, dtype=np.int64)
        self.n = self.l = self.k = self.t = 0
        self.state = DS_STATE_RAW
        self.compare = compare

    @property
    def d(self):
        return self.T[:, 0]

    @property
    def e(self):
        return self.T[:, 1]

    def SetDimensions(self, n, l, k):      # dsops.c:130-165
        self.n = n; self.t = n; self.l = l; self.k = k

    def SetState(self, st):                # dsops.c:63-80
        self.state = st

    def _arrow_tridiag(self, n, d, e, Q):  # DSArrowTridiag dshep.c:221-262 (d,e,Q are views offset by l)
        if n <= 2:
            return
        drot = _L("drot")
        one = _i(1)
        ld = self.ld
        for j in range(n - 2):
            temp = e[j + 1]
            c, s, r = lartg(temp, e[j]); e[j + 1] = r
            s = -s
            temp = d[j + 1]
            e[j] = c * s * (temp - d[j])
            d[j + 1] = s * s * d[j] + c * c * temp
            d[j] = c * c * d[j] + s * s * temp
            j2 = j + 2
            self._rot(Q, j2, j, j + 1, c, s)
            for i in range(j - 1, -1, -1):
                off = -s * e[i]
                e[i] = c * e[i]
                temp = e[i + 1]
                c, s, r = lartg(temp, off); e[i + 1] = r
                s = -s
                temp = (d[i] - d[i + 1]) * s - 2.0 * c * e[i]
                p = s * temp
                d[i + 1] += p
                d[i] -= p
                e[i] = -e[i] - c * temp
                self._rot(Q, j2, i, i + 1, c, s)

    @staticmethod
    def _rot(Q, n, ix, iy, c, s):          # BLAS drot on the first n entries of columns ix, iy
        x = Q[:n, ix].copy(); y = Q[:n, iy].copy()
        Q[:n, ix] = c * x + s * y
        Q[:n, iy] = c * y - s * x

    def Solve(self, wr):                   # DSSolve dsops.c:723 -> DSSolve_HEP_QR dshep.c:383-426
        if self.state >= DS_STATE_CONDENSED:
            return
        n, l, ld = self.n, self.l, self.ld
        d, e = self.d, self.e
        n1 = n - l
        # DSIntermediate_HEP dshep.c:267-321 (compact branch)
        self.Q[:, :] = 0.0
        np.fill_diagonal(self.Q, 1.0)
        if self.state < DS_STATE_INTERMEDIATE:
            na = max(0, self.k - l + 1)
            self._arrow_tridiag(na, d[l:], e[l:], self.Q[l:, l:])
        wr[:l] = d[:l]
        # LAPACKsteqr("V", n1, d+l, e+l, Q+off, ld, rwork)
        Qsub = np.asfortranarray(self.Q[l:l + n1, l:l + n1])
        dd = np.ascontiguousarray(d[l:n]); ee = np.ascontiguousarray(e[l:n])
        work = np.zeros(max(1, 2 * n1))
        info = C.c_int(0)
        _L("dsteqr")(b"V", _i(n1), _p(dd), _p(ee), _p(Qsub), _i(Qsub.shape[0] if n1 else 1), _p(work), C.byref(info))
        if info.value:
            raise OracleError("steqr info=%d" % info.value)
        d[l:n] = dd
        self.Q[l:l + n1, l:l + n1] = Qsub
        wr[l:n] = d[l:n]
        e[: n - 1] = 0.0                    # compact: zero e[0..n-2], keep e[n-1] (extra row)
        self.state = DS_STATE_CONDENSED

    def Sort(self, wr, rr=None, ri=None):  # DSSort dsops.c:329-345 -> DSSort_HEP dshep.c:323-347
        """rr/ri: auxiliary values of an arbitrary selection; the order then comes from them (dshep.c:335-336)."""
        n, l = self.n, self.l
        d = self.d
        perm = self.perm
        perm[:n] = np.arange(n)
        key = d if rr is None else rr
        kim = (lambda i: 0.0) if ri is None else (lambda i: ri[i])
        # DSSortEigenvaluesReal_Private dspriv.c:224-243 / DSSortEigenvalues_Private :172-222 (insertion sort, n = ds->t)
        nn = self.t
        for i in range(l + 1, nn):
            re = key[perm[i]]; rim = kim(perm[i])
            j = i - 1
            result = self.compare(re, rim, key[perm[j]], kim(perm[j]))
            while result < 0 and j >= l:
                perm[j], perm[j + 1] = perm[j + 1], perm[j]
                j -= 1
                if j >= l:
                    result = self.compare(re, rim, key[perm[j]], kim(perm[j]))
        self.last_perm = perm[:n].copy()
        for i in range(l, n):
            wr[i] = d[perm[i]]
        # DSPermuteColumns_Private dspriv.c:248-270
        Q = self.Q
        for i in range(l, n):
            p = perm[i]
            if p != i:
                j = i + 1
                while perm[j] != i:
                    j += 1
                perm[j] = p; perm[i] = i
                tmp = Q[:n, p].copy(); Q[:n, p] = Q[:n, i]; Q[:n, i] = tmp
        d[l:n] = wr[l:n]

    def UpdateExtraRow(self):              # DSUpdateExtraRow_HEP dshep.c:349-381 (compact)
        n = self.n
        beta = self.e[n - 1]
        self.e[:n] = beta * self.Q[n - 1, :n]
        self.k = n

    def Vectors_resnorm(self, j):          # DSVectors_HEP dshep.c:137-175: rnorm = |Q(n-1,j)|
        return abs(self.Q[self.n - 1, j])

    def Truncate(self, n, trim):           # DSTruncate dsops.c + DSTruncate_HEP dshep.c:643-671 (compact)
        if trim:
            self.l = 0; self.k = 0; self.n = n; self.t = n
            self.state = DS_STATE_RAW
        else:
            self.k = n; self.t = self.n; self.n = n
            self.state = DS_STATE_TRUNCATED

    def Qmat(self):                        # DSGetMat(DS_MAT_Q): rows = t if truncated else n; cols = n (dsops.c:276-296)
        rows = self.t if self.state == DS_STATE_TRUNCATED else self.n
        return self.Q[:rows, : self.n]


def _norm_inf(M):
    """MatNorm(M,NORM_INFINITY)"""
    return float(abs(M.to_scipy()).sum(axis=1).max())


def _converged(conv, re, im, res, nrma=0.0, nrmb=1.0):
    """EPSConvergedRelative / Absolute / Norm (epsdefault.c:224-257)"""
    if callable(conv):                                       # EPS_CONV_USER: EPSSetConvergenceTestFunction
        return conv(re, im, res)
    w = np.hypot(re, im)
    if conv == "abs":
        return res
    if conv == "norm":
        return res / (nrma + w * nrmb)
    return res / w if w != 0.0 else np.finfo(float).max


class EPSResult:
    pass


def eps_krylovschur_hep(A, nev, ncv=None, mpd=None, tol=1e-8, max_it=None, which="largest_magnitude",
                        keep=0.5, seed=0x12345678, omp=False, v0=None, orthog=(CGS, REFINE_IFNEEDED, 0.7071),
                        max_steps=None, monitor=None, lock=True, st=None, B=None, conv="rel", deflation=None, trueres=False, stopping=None, arbitrary=None, purify=True):
    """EPSSolve for a symmetric problem with the default Krylov-Schur solver: standard (HEP), or generalized (GHEP,
    B given: the basis carries the B-inner product, EPS_SetInnerProduct epsimpl.h:280-292; the start vector goes
    through the operator, epssolve.c:860-868; the eigenvectors are purified and B-normalised,
    EPSComputeVectors_Hermitian epsdefault.c:27-49). st: an ST (operator, back-transformation).

    EPSSetUp_KrylovSchur krylovschur.c:93-194 (EPS_KS_SYMM), EPSSetDimensions_Default epssetup.c:654-678,
    EPSSolve_KrylovSchur_Default krylovschur.c:227-337, EPSKrylovConvergence epskrylov.c:207-295,
    EPSConvergedRelative epsdefault.c:224, EPSStoppingBasic epsdefault.c:290, EPSGetStartVector epssolve.c:841-873,
    final SlepcSortEigenvalues epssolve.c:178 / slepcsc.c:89-140.
    """
    n = A.n
    if ncv is None:
        if mpd is not None:
            ncv = min(n, nev + mpd)
        else:
            ncv = min(n, max(2 * nev, nev + 15)) if nev < 500 else min(n, nev + 500)
    if mpd is None:
        mpd = ncv
    assert ncv >= nev + 1 or (ncv == nev and ncv == n), "The value of ncv must be at least nev+1"
    assert ncv <= nev + mpd
    if max_it is None:
        max_it = max(100, 2 * n // ncv)
    compare = which if callable(which) else WHICH[which]

    if st is not None:
        def ds_compare(ar, ai, br, bi):
            ar, ai = st.backtransform(ar, ai); br, bi = st.backtransform(br, bi)
            return compare(ar, ai, br, bi)
    else:
        ds_compare = compare
    nrma = _norm_inf(A) if conv == "norm" else 0.0
    nrmb = (_norm_inf(B) if B is not None else 1.0) if conv == "norm" else 1.0
    V = BV(n, ncv + 1, omp=omp)
    V.SetOrthogonalization(*orthog)
    Bip = st.bilinear if (B is not None and st is not None and st.kind == "cayley") else B     # STGetBilinearForm
    if B is not None:
        V.SetMatrix(Bip)
    if deflation is not None:                                # EPSSetDeflationSpace -> BVInsertConstraints (epssetup.c:397-404)
        V.InsertConstraints(deflation)
    ds = DSHEP(ncv + 1, ds_compare)
    eigr = np.zeros(ncv + 1); errest = np.zeros(ncv + 1)

    # EPSGetStartVector(eps,0)
    def start_vector(i):
        if v0 is not None and i == 0:
            V.set_column(0, v0)
        else:
            V.SetRandomColumn(i, seed)
        if B is not None:                                   # force the vector into the range of OP (epssolve.c:860-868)
            V.set_column(i, st.apply(np.array(V.column(i))))
        _, norm, lindep = V.OrthogonalizeColumn(i)
        if not (lindep or norm == 0.0):
            V.ScaleColumn(i, 1.0 / norm)
        return lindep or norm == 0.0

    if start_vector(0):
        raise OracleError("Initial vector is zero or belongs to the deflation space")
    l = 0
    nconv = 0
    its = 0
    reason = 0
    steps = 0
    cycles = []
    while reason == 0:
        its += 1
        nv = min(nconv + mpd, ncv)
        if max_steps is not None and steps + (nv - (nconv + l)) > max_steps:
            nv = nconv + l + (max_steps - steps)
        ds.SetDimensions(nv, nconv, nconv + l)
        k0 = nconv + l
        nv_req = nv
        if st is None:
            nv, beta, breakdown = V.MatLanczos(A, ds.T, nconv + l, nv)
        else:
            nv, beta, breakdown = V.MatLanczosOp(st.apply, ds.T, nconv + l, nv)
        steps += nv - k0
        cycles.append((k0, nv, V.passes_total()))
        ds.SetDimensions(nv, nconv, nconv + l)
        ds.SetState(DS_STATE_RAW if l else DS_STATE_INTERMEDIATE)
        V.SetActiveColumns(nconv, nv)

        ds.Solve(eigr)
        if arbitrary is not None:                           # EPSGetArbitraryValues krylovschur.c:30-58
            rr = np.zeros(ncv + 1); ri = np.zeros(ncv + 1)
            Xall = np.array(V.dense())[:n, :nv]
            for i in range(ds.l, ds.n):
                re = eigr[i] if st is None else st.backtransform(eigr[i], 0.0)[0]
                x = Xall @ ds.Q[:nv, i]
                if B is not None and purify:                # purification of EPSComputeRitzVector
                    y = st.apply(x); x = y / np.sqrt(y @ Bip.mult(y))
                rr[i], ri[i] = arbitrary(re, 0.0, x, np.zeros(n))
            ds.Sort(eigr, rr, ri)
        else:
            ds.Sort(eigr)
        ds.UpdateExtraRow()

        # EPSKrylovConvergence(eps,FALSE,nconv,nv-nconv,beta,0.0,1.0,&k)
        marker = -1
        kk = nconv
        for kk in range(nconv, nv):
            re = eigr[kk]                                   # shift ST with sigma=0: back-transform is identity
            if st is not None and (st.kind == "shift" or conv == "norm"):
                re = st.backtransform(re, 0.0)[0]           # epskrylov.c:253
            resnorm = ds.Vectors_resnorm(kk) * beta * 1.0
            if trueres:                                     # epskrylov.c:245,256-264
                if st is not None and not (st.kind == "shift" or conv == "norm"):
                    re = st.backtransform(re, 0.0)[0]
                resnorm = _true_residual(A, B, V, nv, re, 0.0, ds.Q[:, kk], purify=st.apply if (B is not None and purify) else None, Bnorm=Bip)
            errest[kk] = _converged(conv, re, 0.0, resnorm, nrma, nrmb)
            if marker == -1 and errest[kk] >= tol:
                marker = kk
            if marker != -1:
                break
        else:
            kk = nv
        k = marker if marker != -1 else nv
        # EPSStoppingBasic
        if stopping is not None:                            # EPSSetStoppingTestFunction
            reason = stopping(its, max_it, k, nev)
        elif k >= nev:
            reason = 1      # EPS_CONVERGED_TOL (slepceps.h)
        elif its >= max_it:
            reason = -1     # EPS_DIVERGED_ITS
        nconv_mon = k                                       # krylovschur.c:289: the monitor sees the count before the non-locking reset
        if max_steps is not None and steps >= max_steps and reason == 0:
            reason = 2      # EPS_CONVERGED_USER (step cap of the bench harness)
        # update l
        if reason != 0 or breakdown or k == nv:
            l = 0
        else:
            l = max(1, int((nv - k) * keep))
        if not lock and l > 0:                 # non-locking variant (krylovschur.c:294)
            l += k; k = 0
        if reason == 0:
            if breakdown or k == nv:
                if k < nev:
                    if start_vector(k):
                        reason = -2  # EPS_DIVERGED_BREAKDOWN
            else:
                ds.Truncate(k + l, False)
        V.MultInPlace(ds.Qmat(), nconv, k + l)
        if reason == 0 and not breakdown:
            V.CopyColumn(nv, k + l)
        nconv = k
        if monitor:
            monitor(its, nconv_mon, eigr[:nv].copy(), errest[:nv].copy(), nv)
    ds.Truncate(nconv, True)

    V.SetActiveColumns(0, nconv)
    if st is not None:                                      # EPSComputeValues (epssolve.c:27-41)
        for i in range(nconv):
            eigr[i] = st.backtransform(eigr[i], 0.0)[0]
    if B is not None and purify:                            # EPSComputeVectors_Hermitian: purify, then B-normalise
        for i in range(nconv):
            V.set_column(i, st.apply(np.array(V.column(i))))
        for i in range(nconv):
            V.ScaleColumn(i, 1.0 / V.NormColumn(i))
    elif B is not None and st is not None and st.kind == "cayley":      # epsdefault.c:38-47: B-normalise under Cayley
        for i in range(nconv):
            x = np.array(V.column(i)); V.set_column(i, x / np.sqrt(x @ B.mult(x)))
    # EPSSolve epilogue: final sort of the converged values (SlepcSortEigenvalues slepcsc.c:89-140, all real)
    perm = list(range(nconv))
    for i in range(nconv - 1, -1, -1):
        re = eigr[perm[i]]
        j = i + 1
        while j < nconv:
            if compare(re, 0.0, eigr[perm[j]], 0.0) <= 0:
                break
            perm[j - 1], perm[j] = perm[j], perm[j - 1]
            j += 1
    res = EPSResult()
    res.nconv = nconv; res.its = its; res.reason = reason; res.steps = steps
    res.eigr = eigr[:nconv].copy(); res.perm = np.array(perm, dtype=np.int64)
    res.errest = errest[:nconv].copy()
    res.V = V; res.cycles = cycles; res.ncv = ncv
    res.passes = V.passes_total()
    return res


def _residual_norm(A, B, kr, ki, xr, xi=None):
    """EPSComputeResidualNorm_Private epssolve.c:666-718 in real arithmetic."""
    Bm = (lambda v: B.mult(v)) if B is not None else (lambda v: v)
    if ki == 0 or abs(ki) < abs(kr * np.finfo(float).eps):
        u = A.mult(xr)
        if abs(kr) > np.finfo(float).eps:
            u = u + (-kr) * Bm(xr)
        return float(np.linalg.norm(u))
    u = A.mult(xr) - kr * Bm(xr) + ki * Bm(xi)
    w = A.mult(xi) - kr * Bm(xi) - ki * Bm(xr)
    return float(np.hypot(np.linalg.norm(u), np.linalg.norm(w)))


def _true_residual(A, B, V, nv, re, im, Zr, Zi=None, purify=None, Bnorm=None, D=None):
    """EPSComputeRitzVector epsdefault.c:313-364 + the residual of epskrylov.c:256-264 (-eps_true_residual)."""
    X = np.array(V.dense())[: V.n, :nv]
    x = X @ Zr[:nv]
    if purify is not None:                                   # STApply, B-norm, scale (GHEP)
        y = purify(x)
        x = y / np.sqrt(y @ (Bnorm if Bnorm is not None else B).mult(y))
    y = X @ Zi[:nv] if Zi is not None else None
    if D is not None:                                        # fix and normalise when balancing is used (epsdefault.c:336-361)
        x = x / D
        y = y / D if y is not None else None
        nrm = np.hypot(np.linalg.norm(x), np.linalg.norm(y) if y is not None else 0.0)
        x = x / nrm
        y = y / nrm if y is not None else None
    return _residual_norm(A, B, re, im, x, y)


def eps_compute_error(A, res, i, relative=True, B=None):
    """EPSComputeError epssolve.c:742-815 with EPSComputeResidualNorm_Private :666-718 (real eigenvalue); for a GHEP
    (B given) the residual is A x - k B x and the relative error is also divided by ||x||_2 (:774-780)."""
    j = int(res.perm[i])
    kr = res.eigr[j]
    x = np.array(res.V.column(j))
    u = A.mult(x)
    if abs(kr) > np.finfo(float).eps:
        u = u + (-kr) * (B.mult(x) if B is not None else x)
    err = np.linalg.norm(u)
    if relative:
        err /= abs(kr) * (np.linalg.norm(x) if B is not None else 1.0)
    return err


# ================================================================================================
# Non-symmetric problems: DS NHEP + the Arnoldi variant of the Krylov-Schur driver
# ================================================================================================
def load_petsc_binary(path):
    """MatLoad of a PETSc binary viewer file: big-endian int32 {1211216, rows, cols, nnz}, row lengths, column indices,
    float64 values (the format of share/slepc/datafiles/matrices/*.petsc)."""
    raw = open(path, "rb").read()
    hdr = np.frombuffer(raw, dtype=">i4", count=4)
    assert hdr[0] == 1211216 and hdr[1] == hdr[2], hdr
    n, nnz = int(hdr[1]), int(hdr[3])
    lens = np.frombuffer(raw, dtype=">i4", count=n, offset=16).astype(np.int64)
    col = np.frombuffer(raw, dtype=">i4", count=nnz, offset=16 + 4 * n).astype(np.int32)
    val = np.frombuffer(raw, dtype=">f8", count=nnz, offset=16 + 4 * n + 4 * nnz).astype(np.float64)
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    return CSR(n, rowptr, col, val)


def markov_matrix(m):
    """MatMarkovModel (src/eps/tutorials/ex5.c:137-170): random walk on a triangular grid, N = m(m+1)/2."""
    import scipy.sparse as sp
    cst = 0.5 / (m - 1)
    rows, cols, vals = [], [], []
    ix = 0
    for i in range(1, m + 1):
        jmax = m - i + 1
        for j in range(1, jmax + 1):
            ix += 1
            if j != jmax:
                pd = cst * (i + j - 1)
                rows.append(ix - 1); cols.append(ix); vals.append(2 * pd if i == 1 else pd)                   # north
                rows.append(ix - 1); cols.append(ix + jmax - 1); vals.append(2 * pd if j == 1 else pd)        # east
            pu = 0.5 - cst * (i + j - 3)
            if j > 1:
                rows.append(ix - 1); cols.append(ix - 2); vals.append(pu)                                     # south
            if i > 1:
                rows.append(ix - 1); cols.append(ix - jmax - 2); vals.append(pu)                              # west
    N = m * (m + 1) // 2
    S = sp.csr_matrix((vals, (rows, cols)), shape=(N, N))
    S.sort_indices()
    return CSR(N, S.indptr, S.indices, S.data)


class DSNHEP:
    """DS type NHEP with extra row (krylovschur.c:153-159). A is ld x ld column-major; row n holds the extra row."""

    def __init__(self, ld, compare):
        self.ld = ld
        self.A = np.zeros((ld, ld), order="F")
        self.Q = np.zeros((ld, ld), order="F")
        self.X = np.zeros((ld, ld), order="F")
        self.n = self.l = self.k = self.t = 0
        self.state = DS_STATE_RAW
        self.compare = compare

    def SetDimensions(self, n, l, k):
        self.n = n; self.t = n; self.l = l; self.k = k

    def SetState(self, st):
        self.state = st

    def TranslateHarmonic(self, tau, beta, recover, g):
        """DSTranslateHarmonic_NHEP dsnhep.c:466-537. g: array of ld entries, kept by the caller between the two calls.
        Forward: g = (A - tau I)^{-T} (beta e_n), A(:,n-1) += beta g. Recover (after solve/sort, with ds.l = nconv and
        ds.k = number of kept vectors): the rank-one update is undone on the kept block. Returns gamma = sqrt(1+|g|^2)."""
        import scipy.linalg as sl
        n, A = self.n, self.A
        if not recover:
            g[:] = 0.0; g[n - 1] = beta
            Bm = np.array(A[:n, :n], order="F"); Bm[np.arange(n), np.arange(n)] -= tau
            lu = sl.lu_factor(Bm, check_finite=False)                       # LAPACKgetrf
            g[:n] = sl.lu_solve(lu, g[:n], trans=1, check_finite=False)     # LAPACKgetrs 'C'
            A[:n, n - 1] += g[:n] * beta
        else:
            Q = self.Q
            ncol = self.l + self.k
            ghat = -(Q[:n, :ncol].T @ g[:n])                                # BLASgemv 'C'
            for i in range(ncol):
                for j in range(self.l, ncol):
                    A[i, j] += ghat[i] * Q[n - 1, j] * beta
            g[:n] = g[:n] + Q[:n, :ncol] @ ghat
        gamma = float(np.hypot(1.0, np.linalg.norm(g[:n])))
        if recover:                                                         # ds->extrarow
            for j in range(self.l, self.l + self.k):
                A[n, j] *= gamma
        return gamma

    def _eig_from_T(self, wr, wi, j0, j1):
        """recover eigenvalues of the diagonal blocks j0..j1-1 of the quasi-triangular A (dsutil.c:65-79,160-170)"""
        A, n = self.A, self.n
        j = j0
        while j < j1:
            if j == n - 1 or A[j + 1, j] == 0.0:
                wr[j] = A[j, j]; wi[j] = 0.0
            else:
                wr[j] = A[j, j]; wr[j + 1] = A[j, j]
                wi[j] = np.sqrt(abs(A[j + 1, j])) * np.sqrt(abs(A[j, j + 1])); wi[j + 1] = -wi[j]
                j += 1
            j += 1

    def Solve(self, wr, wi):               # DSSolve_NHEP_Private dsutil.c:21-91
        if self.state >= DS_STATE_CONDENSED:
            return
        n, ld, l = self.n, self.ld, self.l
        A, Q = self.A, self.Q
        Q[:, :] = 0.0
        for i in range(n):
            Q[i, i] = 1.0
        if n == 1:
            wr[0] = A[0, 0]; wi[0] = 0.0
            self.state = DS_STATE_CONDENSED
            return
        ilo = l + 1
        lwork = 6 * ld
        work = np.zeros(lwork); tau = np.zeros(ld); info = C.c_int(0)
        if self.state < DS_STATE_INTERMEDIATE:
            _L("dgehrd")(_i(n), _i(ilo), _i(n), _p(A), _i(ld), _p(tau), _p(work), _i(lwork), C.byref(info))
            assert info.value == 0
            for j in range(n - 1):
                for i in range(j + 2, n):
                    Q[i, j] = A[i, j]; A[i, j] = 0.0
            _L("dorghr")(_i(n), _i(ilo), _i(n), _p(Q), _i(ld), _p(tau), _p(work), _i(lwork), C.byref(info))
            assert info.value == 0
        wrr = np.zeros(ld); wii = np.zeros(ld)
        _L("dhseqr")(b"S", b"V", _i(n), _i(ilo), _i(n), _p(A), _i(ld), _p(wrr), _p(wii), _p(Q), _i(ld), _p(work), _i(lwork), C.byref(info))
        assert info.value == 0, info.value
        wr[:n] = wrr[:n]; wi[:n] = wii[:n]
        self._eig_from_T(wr, wi, 0, l)
        self.state = DS_STATE_CONDENSED

    def Sort(self, wr, wi):                # DSSort_NHEP_Total dsutil.c:93-175
        n, ld, l = self.n, self.ld, self.l
        A, Q = self.A, self.Q
        work = np.zeros(ld); info = C.c_int(0)
        i = l
        while i < n - 1:
            re, im = wr[i], wi[i]
            pos = 0
            j = i + 2 if im != 0 else i + 1
            while j < n:
                if self.compare(re, im, wr[j], wi[j]) > 0:
                    re, im = wr[j], wi[j]; pos = j
                if wi[j] != 0:
                    j += 1
                j += 1
            if pos:
                ifst = C.c_int(pos + 1); ilst = C.c_int(i + 1)
                _L("dtrexc")(b"V", _i(n), _p(A), _i(ld), _p(Q), _i(ld), C.byref(ifst), C.byref(ilst), _p(work), C.byref(info))
                assert info.value == 0, info.value
                self._eig_from_T(wr, wi, i, n)
            if wi[i] != 0:
                i += 1
            i += 1

    def UpdateExtraRow(self):              # DSUpdateExtraRow_NHEP dsnhep.c:318-341
        n = self.n
        x = self.A[n, :n].copy()
        self.A[n, :n] = self.Q[:n, :n].T @ x
        self.k = n

    def Vectors(self, k):
        """DSVectors_NHEP_Eigen_Some dsnhep.c:101-167: k-th eigenvector of A (through Q), returns (newk, rnorm)."""
        n, ld = self.n, self.ld
        A = self.A
        iscomplex = (k < n - 1 and A[k + 1, k] != 0.0)
        mm = 2 if iscomplex else 1
        select = np.zeros(ld, dtype=np.int32); select[k] = 1
        if iscomplex:
            select[k + 1] = 1
        Y = np.zeros((ld, 2), order="F")
        work = np.zeros(3 * ld); mout = C.c_int(0); info = C.c_int(0)
        _L("dtrevc")(b"R", b"S", select.ctypes.data_as(_ip), _i(n), _p(A), _i(ld), _p(Y), _i(ld), _p(Y), _i(ld), _i(mm), C.byref(mout), _p(work), C.byref(info))
        assert info.value == 0 and mout.value == mm
        Z = self.Q[:n, :n] @ Y[:n, :mm]
        norm = np.linalg.norm(Z[:, 0])
        if iscomplex:
            norm = np.hypot(norm, np.linalg.norm(Z[:, 1]))
        Z /= norm
        self.X[:n, k:k + mm] = Z
        rnorm = np.hypot(Z[n - 1, 0], Z[n - 1, 1]) if iscomplex else abs(Z[n - 1, 0])
        return (k + 1 if iscomplex else k), rnorm

    def VectorsAll(self):
        """DSVectors_NHEP_Eigen_All dsnhep.c:169-232 (state CONDENSED or later: back-transform with Q)."""
        n = self.n
        k = 0
        while k < n:
            newk, _ = self.Vectors(k)
            k = newk + 1
        return self.X[:n, :n]

    def GetTruncateSize(self, l, n, k):    # DSGetTruncateSize_Default dsops.c:329-345
        if self.A[l + k, l + k - 1] != 0.0:
            k = k + 1 if l + k < n - 1 else k - 1
        return k

    def Truncate(self, n, trim):           # DSTruncate_NHEP dsnhep.c:385-415
        A, l = self.A, self.l
        if trim:
            A[self.n, l:self.n] = 0.0
            self.l = 0; self.k = 0; self.n = n; self.t = n
            self.state = DS_STATE_RAW
        else:
            if self.k == self.n:
                A[n, l:n] = A[self.n, l:n]
                A[self.n, l:self.n] = 0.0
            self.k = n; self.t = self.n; self.n = n
            self.state = DS_STATE_TRUNCATED

    def Qmat(self):
        rows = self.t if self.state == DS_STATE_TRUNCATED else self.n
        return self.Q[:rows, : self.n]


class ST:
    """Spectral transformation, types shift and sinvert (src/sys/classes/st/impls/shift/shift.c:16-97,
    sinvert/sinvert.c:16-77). Operator per STApply_Generic (stsolve.c:16-25): y = P^-1 M x with
        shift:   nmat=1  M = A - sigma I, P = none      nmat=2  M = A - sigma B, P = B
        sinvert: nmat=1  M = none,        P = A - sigma I   nmat=2  M = B,       P = A - sigma B
    The linear solves use the reference's default KSP, preonly + LU (stsles.c:54-56), here SuperLU through scipy."""

    def __init__(self, A, B=None, kind="shift", sigma=0.0, nu=None):
        """kind "cayley" (cayley.c:138-165): Op = (A - sigma B)^-1 (A + nu B), antishift nu = sigma unless given; its
        bilinear form for symmetric problems is A + nu B (STGetBilinearForm_Cayley, cayley.c:70-77)."""
        import scipy.sparse as sp
        import scipy.sparse.linalg as spl
        self.kind = kind; self.sigma = float(sigma); self.n = A.n
        Sa = A.to_scipy().tocsc()
        Sb = B.to_scipy().tocsc() if B is not None else None
        Ib = Sb if Sb is not None else sp.identity(A.n, format="csc")
        T = (Sa - self.sigma * Ib).tocsc() if self.sigma != 0.0 else Sa
        if kind == "shift":
            self.M = T.tocsr(); self.lu = spl.splu(Sb) if Sb is not None else None
        elif kind == "sinvert":
            self.M = Sb.tocsr() if Sb is not None else None; self.lu = spl.splu(T)
        elif kind == "cayley":
            self.nu = self.sigma if nu is None else float(nu)
            assert (self.nu != 0.0 or self.sigma != 0.0) and self.nu != -self.sigma
            self.M = (Sa + self.nu * Ib).tocsr(); self.lu = spl.splu(T)
            Mb = self.M.tocsr(); Mb.sort_indices()
            self.bilinear = CSR(A.n, Mb.indptr.astype(np.int32), Mb.indices.astype(np.int32), Mb.data.astype(np.float64))
        else:
            raise ValueError(kind)
        self.solves = 0

    def apply(self, x):
        y = self.M @ x if self.M is not None else x
        if self.lu is not None:
            y = self.lu.solve(np.ascontiguousarray(y)); self.solves += 1
        return y

    def backtransform(self, re, im):
        if self.kind == "shift":                       # shift.c:49-56
            return re + self.sigma, im
        if self.kind == "cayley":                      # cayley.c:79-107 (real scalars)
            if im == 0.0:
                return (self.nu + re * self.sigma) / (re - 1.0), 0.0
            # (nu + theta sigma)/(theta - 1); cayley.c:93-99 takes |theta - 1|^2 after overwriting theta's parts: not followed
            t = im * im + re * (re - 2.0) + 1.0
            return (self.sigma * (re * re + im * im - re) + self.nu * (re - 1.0)) / t, (-self.sigma * im - self.nu * im) / t
        if im == 0.0:                                  # sinvert.c:16-40 (real scalars)
            return 1.0 / re + self.sigma, 0.0
        t = re * re + im * im
        return re / t + self.sigma, -im / t


def eps_krylovschur_nhep(A, nev, ncv=None, mpd=None, tol=1e-8, max_it=None, which="largest_magnitude", keep=0.5,
                         seed=0x12345678, v0=None, max_steps=None, st=None, lock=True, conv="rel", B=None, harmonic=None,
                         trueres=False, stopping=None, monitor=None, balance_its=0, balance="oneside", balance_cutoff=1e-8):
    """EPSSolve_KrylovSchur_Default with the Arnoldi expansion (krylovschur.c:227-337, non-Hermitian branch),
    EPSKrylovConvergence for conjugate pairs (epskrylov.c:262-287), EPSComputeVectors_Schur (epsdefault.c:105-169).
    With st (an ST): the Krylov operator is st.apply, the DS sorts through the back-transformation
    (EPSSetUpSort_Default epssetup.c:222-240, SlepcSCCompare slepcsc.c:41-62), convergence is tested on the
    transformed eigenvalue except for STSHIFT (epskrylov.c:253), and EPSComputeValues maps the eigenvalues back
    (epssolve.c:27-41) before the conjugate-pair fix-up and the final sort (epssolve.c:160-178).
    harmonic: the target tau of EPSSetExtraction(EPS_HARMONIC) - the Krylov decomposition is translated before the
    projected solve and translated back before the restart (krylovschur.c:270-271,310-320); a symmetric problem takes
    this same path (variant EPS_KS_DEFAULT, krylovschur.c:139)."""
    n = A.n
    if ncv is None:
        ncv = min(n, nev + mpd) if mpd is not None else (min(n, max(2 * nev, nev + 15)) if nev < 500 else min(n, nev + 500))
    if mpd is None:
        mpd = ncv
    if max_it is None:
        max_it = max(100, 2 * n // ncv)
    compare = which if callable(which) else WHICH[which]
    nrma = _norm_inf(A) if conv == "norm" else 0.0
    nrmb = (_norm_inf(B) if B is not None else 1.0) if conv == "norm" else 1.0
    if st is not None:
        def ds_compare(ar, ai, br, bi):
            ar, ai = st.backtransform(ar, ai); br, bi = st.backtransform(br, bi)
            return compare(ar, ai, br, bi)
    else:
        ds_compare = compare
    V = BV(n, ncv + 1)
    ds = DSNHEP(ncv + 1, ds_compare)
    eigr = np.zeros(ncv + 1); eigi = np.zeros(ncv + 1); errest = np.zeros(ncv + 1)
    gh = np.zeros(ncv + 1)

    def start_vector(i):
        if v0 is not None and i == 0:
            V.set_column(0, v0)
        else:
            V.SetRandomColumn(i, seed)
        _, norm, lindep = V.OrthogonalizeColumn(i)
        if not (lindep or norm == 0.0):
            V.ScaleColumn(i, 1.0 / norm)
        return lindep or norm == 0.0

    # EPSSetBalance(ONESIDE / TWOSIDE): EPSBuildBalance_Krylov epsdefault.c:370-434, then the expansion runs on D Op D^-1 (stsolve.c:252-256)
    D = None
    base_op = st.apply if st is not None else (lambda x: A.mult(x))
    if balance_its:
        D = np.ones(n)
        scratch = BV(n, 5)
        if balance == "twoside":                     # STApplyHermitianTranspose: built for the operators without a solve (A itself, or a shift of it)
            assert st is None or (st.kind == "shift" and st.lu is None)
            Mt = (A.to_scipy() if st is None else st.M).T.tocsr()
        norma = 0.0
        for j in range(balance_its):
            scratch.SetRandomColumn(3, seed + 7919 * (j + 1))
            z = np.where(np.array(scratch.column(3)) < 0.5, -1.0, 1.0)
            p = base_op(z / D) * D
            if balance == "twoside":                 # epsdefault.c:402-421
                if j == 0:
                    norma = np.abs(p).max()
                r = (Mt @ (z * D)) / D
                sel = (np.abs(p) > balance_cutoff * norma) & (r != 0.0)
                D[sel] = D[sel] * np.sqrt(np.abs(r[sel] / p[sel]))
                continue
            nz = p != 0.0
            D[nz] = D[nz] / np.abs(p[nz])
        bal_op = lambda x: base_op(x / D) * D                # noqa: E731
    assert not start_vector(0)
    l = 0; nconv = 0; its = 0; reason = 0; steps = 0
    while reason == 0:
        its += 1
        nv = min(nconv + mpd, ncv)
        if max_steps is not None and steps + (nv - (nconv + l)) > max_steps:
            nv = nconv + l + (max_steps - steps)
        ds.SetDimensions(nv, nconv, nconv + l)
        k0 = nconv + l
        H = ds.A[: nv + 1, :nv]                       # DSGetMat(DS_MAT_A): (n+1) x n with the extra row
        Hs = np.asfortranarray(ds.A)                  # BVMatArnoldi writes through ld = ds.ld
        if D is not None:
            nv, beta, breakdown = V.MatArnoldiOp(bal_op, Hs, k0, nv)
        elif st is None:
            nv, beta, breakdown = V.MatArnoldi(A, Hs, k0, nv)
        else:
            nv, beta, breakdown = V.MatArnoldiOp(st.apply, Hs, k0, nv)
        ds.A[:, :] = Hs
        steps += nv - k0
        ds.SetDimensions(nv, nconv, nconv + l)
        ds.SetState(DS_STATE_RAW if l else DS_STATE_INTERMEDIATE)
        V.SetActiveColumns(nconv, nv)
        gamma = 1.0
        if harmonic is not None:
            gamma = ds.TranslateHarmonic(harmonic, beta, False, gh)
        ds.Solve(eigr, eigi)
        ds.Sort(eigr, eigi)
        ds.UpdateExtraRow()
        # EPSKrylovConvergence
        marker = -1
        k = nconv
        while k < nv:
            re, im = eigr[k], eigi[k]
            if st is not None and (st.kind == "shift" or conv == "norm"):
                re, im = st.backtransform(re, im)
            newk, resnorm = ds.Vectors(k)
            resnorm *= beta * gamma
            if trueres:                                     # epskrylov.c:245,256-264
                if st is not None and not (st.kind == "shift" or conv == "norm"):
                    re, im = st.backtransform(re, im)
                resnorm = _true_residual(A, B, V, nv, re, im, ds.X[:, k], ds.X[:, newk] if newk == k + 1 else None, D=D)
            errest[k] = _converged(conv, re, im, resnorm, nrma, nrmb)
            if marker == -1 and errest[k] >= tol:
                marker = k
            if newk == k + 1:
                errest[k + 1] = errest[k]; k += 1
            if marker != -1:
                break
            k += 1
        k = marker if marker != -1 else nv
        if stopping is not None:
            reason = stopping(its, max_it, k, nev)
        elif k >= nev:
            reason = 1      # EPS_CONVERGED_TOL
        elif its >= max_it:
            reason = -1
        nconv_mon = k
        if max_steps is not None and steps >= max_steps and reason == 0:
            reason = 2      # EPS_CONVERGED_USER (step cap of the bench harness)
        if reason != 0 or breakdown or k == nv:
            l = 0
        else:
            l = max(1, int((nv - k) * keep))
            l = ds.GetTruncateSize(k, nv, l)
        if not lock and l > 0:                 # non-locking variant (krylovschur.c:294)
            l += k; k = 0
        if reason == 0:
            if breakdown or k == nv:
                if k < nev and start_vector(k):
                    reason = -2
            else:
                if harmonic is not None:                   # undo the translation (krylovschur.c:310-320)
                    ds.SetDimensions(nv, k, l)
                    gamma = ds.TranslateHarmonic(0.0, beta, True, gh)
                    V.SetActiveColumns(0, nv)               # gamma u^ = u - U g~
                    V.MultColumn(-1.0, 1.0, nv, gh[:nv].copy())
                    V.ScaleColumn(nv, 1.0 / gamma)
                    V.SetActiveColumns(nconv, nv)
                    ds.SetDimensions(nv, k, nv)
                ds.Truncate(k + l, False)
        V.MultInPlace(ds.Qmat(), nconv, k + l)
        if reason == 0 and not breakdown:
            V.CopyColumn(nv, k + l)
        nconv = k
        if monitor:
            monitor(its, nconv_mon, eigr[:nv].copy(), eigi[:nv].copy(), errest[:nv].copy(), nv)
    ds.Truncate(nconv, True)
    # EPSComputeVectors_Schur: X = V * Z, Z = eigenvectors of the truncated T
    V.SetActiveColumns(0, nconv)
    ds.state = DS_STATE_RAW            # after trimming, the eigenvectors are those of T itself (no back-transform)
    Qsave = ds.Q.copy(); ds.Q[:, :] = np.eye(ds.ld)
    Z = np.asfortranarray(ds.VectorsAll().copy()) if nconv else np.zeros((0, 0), order="F")
    ds.Q[:, :] = Qsave
    if nconv:
        V.MultInPlace(Z, 0, nconv)
        if D is not None:              # epsdefault.c:130-139: x <- D \ x, BVNormalize with the pairs together
            i = 0
            while i < nconv:
                V.set_column(i, np.array(V.column(i)) / D)
                if eigi[i] != 0.0 and i + 1 < nconv:
                    V.set_column(i + 1, np.array(V.column(i + 1)) / D)
                    nrm = np.hypot(np.linalg.norm(V.column(i)), np.linalg.norm(V.column(i + 1)))
                    V.ScaleColumn(i, 1.0 / nrm); V.ScaleColumn(i + 1, 1.0 / nrm); i += 2
                else:
                    V.ScaleColumn(i, 1.0 / np.linalg.norm(V.column(i))); i += 1
    if st is not None:                 # EPSComputeValues -> EPSBackTransform_Default
        for i in range(nconv):
            eigr[i], eigi[i] = st.backtransform(eigr[i], eigi[i])
    # conjugate pairs: positive imaginary part first (epssolve.c:163-175); trexc orders them so, but the inversion
    # of sinvert flips the sign of the imaginary parts
    i = 0
    while i < nconv - 1:
        if eigi[i] != 0:
            if eigi[i] < 0:
                eigi[i] = -eigi[i]; eigi[i + 1] = -eigi[i + 1]
                V.ScaleColumn(i + 1, -1.0)
            i += 1
        i += 1
    # final sort keeping pairs together (slepcsc.c:89-140)
    perm = list(range(nconv))
    i = nconv - 1
    while i >= 0:
        re = eigr[perm[i]]; im = eigi[perm[i]]
        j = i + 1
        if im != 0:
            i -= 1
            im = eigi[perm[i]]
        while j < nconv:
            if compare(re, im, eigr[perm[j]], eigi[perm[j]]) <= 0:
                break
            if not im:
                if eigi[perm[j]] == 0.0:
                    perm[j - 1], perm[j] = perm[j], perm[j - 1]; j += 1
                else:
                    tmp = perm[j - 1]; perm[j - 1] = perm[j]; perm[j] = perm[j + 1]; perm[j + 1] = tmp; j += 2
            else:
                if eigi[perm[j]] == 0.0:
                    tmp = perm[j - 2]; perm[j - 2] = perm[j]; perm[j] = perm[j - 1]; perm[j - 1] = tmp; j += 1
                else:
                    perm[j - 2], perm[j] = perm[j], perm[j - 2]
                    perm[j - 1], perm[j + 1] = perm[j + 1], perm[j - 1]; j += 2
        i -= 1
    res = EPSResult()
    res.nconv = nconv; res.its = its; res.reason = reason; res.steps = steps
    res.eigr = eigr[:nconv].copy(); res.eigi = eigi[:nconv].copy(); res.perm = np.array(perm, dtype=np.int64)
    res.errest = errest[:nconv].copy(); res.V = V; res.passes = V.passes_total(); res.ncv = ncv
    return res


def eps_compute_error_nhep(A, res, i, B=None):
    """EPSComputeError relative, real-arithmetic pair form (epssolve.c:666-718): ||A x - k B x|| / |k|."""
    j = int(res.perm[i])
    kr, ki = res.eigr[j], res.eigi[j]
    S = A.to_scipy()
    Bm = (lambda v: B.to_scipy() @ v) if B is not None else (lambda v: v)
    if ki == 0 or abs(ki) < abs(kr * np.finfo(float).eps):
        x = np.array(res.V.column(j))
        u = S @ x - kr * Bm(x)
        nrm = np.linalg.norm(u)
    else:
        jr = j if ki > 0 else j - 1                 # BV_GetEigenvector bvimpl.h:423-446
        xr = np.array(res.V.column(jr)); xi = np.array(res.V.column(jr + 1))
        if ki < 0:
            xi = -xi
        u = S @ xr - kr * Bm(xr) + ki * Bm(xi)
        nr = np.linalg.norm(u)
        u = S @ xi - kr * Bm(xi) - ki * Bm(xr)
        nrm = np.hypot(nr, np.linalg.norm(u))
    return nrm / np.hypot(kr, ki)
