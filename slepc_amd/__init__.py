"""
slepc_amd -- MI355X-native hot path of SLEPc's EPS Krylov-Schur solver (CSR SpMV + BV kernels + fused
classical Gram-Schmidt + restart panel products), behind a C ABI (include/ksgpu.h, slepc_amd/libksgpu.so).

This package is the thin host-side mirror of the reference's interfaces for that path:
  Context  - device / HIP stream / communicator (PETSc's comm + default stream)
  Mat      - MatCreateSeqAIJWithArrays-style CSR operator, `mult` = MatMult          (PETSc AIJ)
  BV       - basis vectors, method names = the reference's BV functions without the prefix
             (slepcbv.h: BVMult, BVMultVec, BVDot, BVDotVec, BVOrthogonalizeColumn, BVMatLanczos ...)
  EPS      - EPSSetOperators / EPSSetDimensions / EPSSolve / EPSGetEigenpair / EPSComputeError

Device memory belongs to the library; host<->device staging uses numpy arrays. torch is only
plumbing (streams, torch.distributed bootstrap of the RCCL communicator) and is imported lazily.
The library is gfx950-only and has NO CPU fallback: creating a Context without an MI355X raises.
"""
import os
import ctypes as C
import numpy as np

from . import _lib
from ._lib import KsError

CGS, MGS = 0, 1
REFINE_IFNEEDED, REFINE_NEVER, REFINE_ALWAYS = 0, 1, 2
NORM_1, NORM_2, NORM_FROBENIUS, NORM_INFINITY = 0, 1, 2, 3
EPS_LARGEST_MAGNITUDE, EPS_SMALLEST_MAGNITUDE, EPS_LARGEST_REAL, EPS_SMALLEST_REAL = 1, 2, 3, 4
EPS_HEP, EPS_GHEP, EPS_NHEP, EPS_GNHEP = 1, 2, 3, 4
EPS_ERROR_ABSOLUTE, EPS_ERROR_RELATIVE, EPS_ERROR_BACKWARD = 0, 1, 2
EPS_CONVERGED_TOL, EPS_CONVERGED_USER, EPS_DIVERGED_ITS, EPS_DIVERGED_BREAKDOWN = 1, 2, -1, -2
WHICH = {"largest_magnitude": 1, "smallest_magnitude": 2, "largest_real": 3, "smallest_real": 4,
         "largest_imaginary": 5, "smallest_imaginary": 6, "target_magnitude": 7, "target_real": 8, "user": 11}
BLOCK = {"gs": 0, "chol": 1, "tsqr": 2, "tsqrchol": 3, "svqb": 4}
SHELL_MULT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)
ST_SHIFT, ST_SINVERT, ST_CAYLEY = 0, 1, 2
EIG_COMPARE_FN = C.CFUNCTYPE(C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.POINTER(C.c_int), C.c_void_p)
EPS_CONVERGED_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_double, C.c_double, C.c_double, C.POINTER(C.c_double), C.c_void_p)
EPS_STOPPING_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_void_p)
EPS_ARBITRARY_FN = C.CFUNCTYPE(C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)
EPS_MONITOR_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.c_void_p)

KCLASSES = ["spmv_csr", "bv_dot_sweep", "gs_bookkeeping", "gs_update_fused_dot", "gs_update", "bv_scale", "bv_multinplace",
            "bv_copy", "bv_mult", "bv_dot_panel", "bv_norm", "halo_exchange", "allreduce", "gated_noop", "other", "spmv_dot_fused"]
def event_name(name):
    """The reference's log event (bvfunc.c:69-86) under which -log_view shows the work of profiling class `name` (ks_prof_event_name)."""
    return _lib.lib().ks_prof_event_name(KCLASSES.index(name)).decode() if name in KCLASSES else "-"


# kernel symbol behind each (class, variant) as rocprofv3 --kernel-trace names it (16-byte-load forms)
def kernel_symbol(name, var):
    if name == "spmv_csr":
        return ("k_spmv_dict<W>" if var == 16 else "k_spmv_odict<W>" if var == 17 else "k_binned_gather + k_binned_reduce" if var == 18
                else "k_spmv_sell<4>" if var == 8 else "k_spmv_csr_wave_dma<8, 6, 8> | k_spmv_csr_wave<ROWSIDE, 8> | k_spmv_csr<G, 4, false, false>")
    if name == "bv_dot_sweep":
        return "k_dot_sweep<%d, 2>" % var
    if name == "spmv_dot_fused":
        return "k_dot_spmv_dict<%d, W>" % var
    if name in ("gs_update_fused_dot", "gs_update") or (name == "gated_noop" and var > 0):
        return "k_gs_update<%d, 2, false>" % var        # (the ops->gramschmidt slot's passes run k_gs_update<KT, 2, true>)
    if name == "gs_bookkeeping":
        return "k_gs_finish<true, true>"
    if name in ("bv_multinplace", "bv_mult"):
        return "k_panel_mult_direct<%d, NT, U>" % (var // 4) if var else "k_multvec<2>"       # variant = 4 * KS4 (ks_panel.hip ksp_mult_mfma)
    if name == "bv_dot_panel":
        return "k_panel_dot_direct<MT, NT, SAME, U>"
    return name


_dp = _lib.dp
_ip = _lib.ip


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _pi(a):
    return None if a is None else a.ctypes.data_as(_ip)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Context:
    """Device + stream + communicator. stream: a raw hipStream_t (int) or None for a library-owned stream."""

    def __init__(self, device=0, stream=None):
        self.L = _lib.lib()
        h = C.c_void_p()
        _lib.check(self.L.ks_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(h)))
        self.h = h
        self.device = device
        self._cb = None
        self.rank, self.size = 0, 1

    def close(self):
        if self.h:
            self.L.ks_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        _lib.check(self.L.ks_ctx_synchronize(self.h))

    DEBUG_KEYS = {"no_fused_gs": 0, "no_mfma": 1, "no_spmv_dot": 2, "force_multi": 3, "halo_overlap": 4, "oneshot_seq0": 5}

    def set_debug(self, key, value=1):
        """Test hooks (ks_ctx_set_debug): run the path a fast one replaces, or the multi-rank path on one rank."""
        _lib.check(self.L.ks_ctx_set_debug(self.h, self.DEBUG_KEYS[key], int(value)))

    def sync_count(self):
        v = C.c_longlong(); _lib.check(self.L.ks_ctx_sync_count(self.h, C.byref(v))); return v.value

    def device_info(self):
        arch = C.create_string_buffer(64); ncu = C.c_int(); mem = C.c_size_t()
        _lib.check(self.L.ks_ctx_device_info(self.h, arch, 64, C.byref(ncu), C.byref(mem)))
        return {"arch": arch.value.decode(), "num_cu": ncu.value, "mem_total": mem.value}

    # -- communicator
    def init_rccl(self, rank, size, unique_id):
        """Native RCCL provider; unique_id: 128 bytes produced by get_unique_id() on rank 0 and broadcast."""
        _lib.check(self.L.ks_comm_init_rccl(self.h, rank, size, bytes(unique_id)))
        self.rank, self.size = rank, size

    @staticmethod
    def get_unique_id():
        buf = C.create_string_buffer(_lib.KS_UNIQUE_ID_BYTES)
        _lib.check(_lib.lib().ks_comm_get_unique_id(buf))
        return buf.raw

    def set_comm_ops(self, rank, size, allreduce_sum, allgather_host, exchange):
        """Caller-supplied communicator (what an MPI adapter would install). Python callables:
             allreduce_sum(dev_ptr, count, stream) ; allgather_host(send_ptr, nbytes, recv_ptr) ;
             exchange(peers, dev_send, send_off, send_cnt, dev_recv, recv_off, recv_cnt, elem_bytes, stream)
           each returning 0 on success."""
        def guard(f):
            def g(*a):
                try:
                    return int(f(*a) or 0)
                except Exception:       # noqa: BLE001 - must not unwind through C
                    import traceback
                    traceback.print_exc()
                    return 1
            return g
        ar = guard(lambda user, buf, count, stream: allreduce_sum(buf, count, stream))
        ag = guard(lambda user, send, nbytes, recv: allgather_host(send, nbytes, recv))

        def ex(user, npeers, peers, dsend, soff, scnt, drecv, roff, rcnt, eb, stream):
            L = lambda p: [p[i] for i in range(npeers)]      # noqa: E731
            return exchange(L(peers), dsend, L(soff), L(scnt), drecv, L(roff), L(rcnt), eb, stream)
        self._ops = _lib.CommOps(_lib.ALLREDUCE_FN(ar), _lib.ALLGATHER_FN(ag), _lib.EXCHANGE_FN(guard(ex)))
        _lib.check(self.L.ks_comm_set_ops(self.h, rank, size, C.byref(self._ops), None))
        self.rank, self.size = rank, size

    def set_allreduce(self, kind):
        """kind: "oneshot" (peer-mapped mailboxes, one kernel per rank) or "provider" (the communicator's allreduce).
        Collective; returns what is active afterwards - "provider" if any rank could not map the mailboxes."""
        act = C.c_int()
        _lib.check(self.L.ks_comm_set_allreduce(self.h, {"provider": 0, "oneshot": 1}[kind], C.byref(act)))
        return "oneshot" if act.value == 1 else "provider"

    def allreduce_sum_dev(self, dev_ptr, count):
        """In-place sum over the ranks of `count` doubles at a device address, ordered on the context's stream."""
        _lib.check(self.L.ks_comm_allreduce_sum(self.h, C.c_void_p(dev_ptr), count))

    def comm_check(self):
        """Known-answer run of the installed communicator (collective): allreduce, allgather, ring exchange."""
        _lib.check(self.L.ks_comm_check(self.h))

    def bcast_stats(self, reset=False):
        """(calls, seconds): host wall time in the per-restart broadcast of rank 0's projected problem (ks_comm_bcast_stats)."""
        n = C.c_longlong(); s = C.c_double()
        _lib.check(self.L.ks_comm_bcast_stats(self.h, C.byref(n), C.byref(s), int(reset)))
        return n.value, s.value

    def memcpy_h2d(self, dev_ptr, host_array, stream=None):
        """stream: the `stream` argument a communicator callback received (None: the context's own)."""
        _lib.check(self.L.ks_ctx_memcpy_stream(self.h, C.c_void_p(dev_ptr), host_array.ctypes.data_as(C.c_void_p), host_array.nbytes, 0, C.c_void_p(stream)))

    def memset(self, dev_ptr, value, nbytes):
        """hipMemsetAsync on the context's stream: enqueues only."""
        _lib.check(self.L.ks_ctx_memset(self.h, C.c_void_p(dev_ptr), value, nbytes))

    def memcpy_d2h(self, host_array, dev_ptr, stream=None):
        _lib.check(self.L.ks_ctx_memcpy_stream(self.h, host_array.ctypes.data_as(C.c_void_p), C.c_void_p(dev_ptr), host_array.nbytes, 1, C.c_void_p(stream)))

    # -- profiling
    def prof_enable(self, on=True, classes=None):
        """classes: optional list of class names to time (fewer HIP events in a timed region); default all."""
        v = int(bool(on))
        if on and classes:
            v = 0
            for c in classes:
                v |= 1 << (KCLASSES.index(c) + 1)
        _lib.check(self.L.ks_prof_enable(self.h, v))

    def prof_reset(self):
        _lib.check(self.L.ks_prof_reset(self.h))

    def prof_get(self, by_variant=False):
        """{class: {launches, ms, alg_bytes, hbm_bytes}}; by_variant=True keys are (class, KT)."""
        out = {}
        for i, name in enumerate(KCLASSES):
            variants = list(range(65)) if by_variant else [-1]
            for var in variants:
                n = C.c_longlong(); ms = C.c_double(); b = C.c_double(); hb = C.c_double()
                _lib.check(self.L.ks_prof_get(self.h, i, var, C.byref(n), C.byref(ms), C.byref(b), C.byref(hb)))
                if n.value:
                    out[(name, var) if by_variant else name] = {"launches": n.value, "ms": ms.value, "alg_bytes": b.value, "hbm_bytes": hb.value}
        return out


class Mat:
    """The MatMult(AIJ) slot. CSR arrays follow PETSc SeqAIJ (i, j, a); col holds GLOBAL column indices."""

    def __init__(self, ctx, handle):
        self.ctx, self.h = ctx, handle
        n = C.c_int(); N = C.c_int(); nnz = C.c_longlong()
        _lib.check(ctx.L.ks_mat_get_sizes(self.h, C.byref(n), C.byref(N), C.byref(nnz)))
        self.n, self.N, self.nnz = n.value, N.value, nnz.value

    @classmethod
    def from_csr(cls, ctx, rowptr, col, val, row_start=0, n_global=None, keep_csr=False):
        """keep_csr: KS_MAT_KEEP_CSR - the matrix keeps the arrays it was created from (MatAXPY, ST_MATMODE_COPY)."""
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
        col = np.ascontiguousarray(col, dtype=np.int32)
        val = _f64(val)
        n = len(rowptr) - 1
        if n_global is None:
            n_global = n
        h = C.c_void_p()
        _lib.check(ctx.L.ks_mat_create_csr_flags(ctx.h, n, row_start, n_global, _pi(rowptr), _pi(col), _p(val), 1 if keep_csr else 0, C.byref(h)))
        return cls(ctx, h)

    def axpy_new(self, alpha, B=None, keep_csr=False):
        """MatDuplicate(self) + MatAXPY(P, alpha, B, DIFFERENT_NONZERO_PATTERN); B None: MatShift(P, alpha)."""
        h = C.c_void_p()
        _lib.check(self.ctx.L.ks_mat_create_axpy(self.h, alpha, None if B is None else B.h, 1 if keep_csr else 0, C.byref(h)))
        return Mat(self.ctx, h)

    @classmethod
    def laplacian3d(cls, ctx, nx, ny, nz, z0=0, nz_local=None):
        h = C.c_void_p()
        _lib.check(ctx.L.ks_mat_create_laplacian3d(ctx.h, nx, ny, nz, z0, nz if nz_local is None else nz_local, C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def laplacian2d(cls, ctx, n, m=None):
        h = C.c_void_p()
        _lib.check(ctx.L.ks_mat_create_laplacian2d(ctx.h, n, n if m is None else m, C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def load(cls, ctx, path):
        """MatLoad of a PETSc binary file."""
        h = C.c_void_p()
        _lib.check(ctx.L.ks_mat_load_petsc_binary(ctx.h, os.fsencode(path), C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def shell(cls, ctx, n, mult, row_start=0, n_global=None):
        """MatCreateShell + MATOP_MULT: mult(x_ptr, y_ptr) works on device pointers (n doubles each)."""
        def tramp(_user, x, y):
            try:
                mult(x, y)
                return 0
            except KsError as e:
                return e.rc
        cb = SHELL_MULT_FN(tramp)
        h = C.c_void_p()
        _lib.check(ctx.L.ks_mat_create_shell(ctx.h, n, row_start, n if n_global is None else n_global, C.cast(cb, C.c_void_p), None, C.byref(h)))
        m = cls(ctx, h)
        m._cb = cb                      # keep the trampoline alive as long as the matrix
        return m

    def shell_set_mult_transpose(self, mult_t):
        """MATOP_MULT_TRANSPOSE of a shell matrix: mult_t(x_ptr, y_ptr) = A^T x on device pointers."""
        def tramp(_user, x, y):
            try:
                mult_t(x, y)
                return 0
            except KsError as e:
                return e.rc
        self._cb_t = SHELL_MULT_FN(tramp)
        _lib.check(self.ctx.L.ks_mat_shell_set_mult_transpose(self.h, C.cast(self._cb_t, C.c_void_p)))

    def mult_transpose_dev(self, x_ptr, y_ptr):
        """MatMultTranspose on device pointers."""
        _lib.check(self.ctx.L.ks_mat_mult_transpose(self.h, C.c_void_p(x_ptr), C.c_void_p(y_ptr)))

    def mult_transpose(self, x):
        """y = A^T x with host vectors (test convenience, single rank)."""
        x = _f64(x)
        W = BV(self.ctx, len(x), 2)
        W.set_column(0, x)
        self.mult_transpose_dev(W.column_ptr(0), W.column_ptr(1))
        return W.column(1)

    def set_enqueue_only(self, flag=True):
        _lib.check(self.ctx.L.ks_mat_shell_set_enqueue_only(self.h, int(bool(flag))))

    def set_halo(self, kind):
        """kind: "peer" (boundary entries stored straight into the neighbours' ghost mailboxes) or "provider" (the communicator's
        exchange). Collective; returns what is active afterwards - "provider" if any rank could not map its neighbours."""
        act = C.c_int()
        _lib.check(self.ctx.L.ks_mat_set_halo(self.h, {"provider": 0, "peer": 1}[kind], C.byref(act)))
        return "peer" if act.value == 1 else "provider"

    def layout(self):
        v = C.c_int(); _lib.check(self.ctx.L.ks_mat_get_layout(self.h, C.byref(v)))
        return ["csr", "sell", "sliced", "shell", "dict", "odict", "binned"][v.value]

    def norm_inf(self):
        v = C.c_double(); _lib.check(self.ctx.L.ks_mat_norm_inf(self.h, C.byref(v))); return v.value

    def get_diagonal(self):
        """MatGetDiagonal of the local diagonal block, as a host array (test convenience)."""
        V = BV(self.ctx, self.n, 1)
        _lib.check(self.ctx.L.ks_mat_get_diagonal(self.h, C.c_void_p(V.column_ptr(0))))
        return V.column(0)

    def destroy(self):
        if self.h and self.ctx.h:      # a closed context already released the device; never touch it again
            self.ctx.L.ks_mat_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    def mult(self, x):
        """y = A x with host vectors (test convenience, single rank)."""
        x = _f64(x)
        y = np.empty(self.n)
        _lib.check(self.ctx.L.ks_mat_mult_host(self.h, _p(x), _p(y)))
        return y

    def mult_dev(self, x_ptr, y_ptr):
        _lib.check(self.ctx.L.ks_mat_mult(self.h, C.c_void_p(x_ptr), C.c_void_p(y_ptr)))

    def spmv_bytes(self):
        """Algorithmic bytes of one MatMult: 12*nnz + 4*(n+1) + 16*n."""
        return 12.0 * self.nnz + 4.0 * (self.n + 1) + 16.0 * self.n


class BV:
    """Basis vectors on the device; one m*ld column-major block (BVSVEC layout)."""

    def __init__(self, ctx, n, m, ld=0, N=None, row_start=0, _handle=None):
        self.ctx = ctx
        L = ctx.L
        if _handle is None:
            h = C.c_void_p()
            _lib.check(L.ks_bv_create(ctx.h, n, n if N is None else N, m, ld, C.byref(h)))
            self.h = h
            self._own = True
            if row_start:
                _lib.check(L.ks_bv_set_ownership_start(self.h, row_start))
        else:
            self.h = _handle
            self._own = False
        nn = C.c_int(); NN = C.c_int(); mm = C.c_int(); ll = C.c_int()
        _lib.check(L.ks_bv_get_sizes(self.h, C.byref(nn), C.byref(NN), C.byref(mm), C.byref(ll)))
        self.n, self.N, self.m, self.ld = nn.value, NN.value, mm.value, ll.value

    def destroy(self):
        if self.h and self._own and self.ctx.h:
            self.ctx.L.ks_bv_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    # -- layout / data movement
    @property
    def l(self):
        l = C.c_int(); k = C.c_int()
        _lib.check(self.ctx.L.ks_bv_get_active_columns(self.h, C.byref(l), C.byref(k)))
        return l.value

    @property
    def k(self):
        l = C.c_int(); k = C.c_int()
        _lib.check(self.ctx.L.ks_bv_get_active_columns(self.h, C.byref(l), C.byref(k)))
        return k.value

    def SetActiveColumns(self, l, k):
        _lib.check(self.ctx.L.ks_bv_set_active_columns(self.h, l, k))

    def SetOrthogonalization(self, type=CGS, refine=REFINE_IFNEEDED, eta=0.7071):
        _lib.check(self.ctx.L.ks_bv_set_orthogonalization(self.h, type, refine, eta))

    def column_ptr(self, j):
        p = C.c_void_p()
        _lib.check(self.ctx.L.ks_bv_get_column(self.h, j, C.byref(p)))
        return p.value

    def set_column(self, j, x):
        x = _f64(x)
        assert x.shape == (self.n,)
        _lib.check(self.ctx.L.ks_bv_set_column_host(self.h, j, _p(x)))

    def column(self, j):
        x = np.empty(self.n)
        _lib.check(self.ctx.L.ks_bv_get_column_host(self.h, j, _p(x)))
        return x

    def dense(self):
        return np.stack([self.column(j) for j in range(self.m)], axis=1)

    def set_dense(self, X):
        for j in range(X.shape[1]):
            self.set_column(j, X[:, j])

    def buffer(self):
        """(nc+m x m) coefficient buffer, column 0 = scratch, column j = H(:,j)  (BVGetBufferVec)."""
        rows = self.nc + self.m
        b = np.empty(rows * self.m)
        _lib.check(self.ctx.L.ks_bv_get_buffer_host(self.h, _p(b)))
        return b.reshape(self.m, rows).T.copy()

    @property
    def nc(self):
        v = C.c_int()
        _lib.check(self.ctx.L.ks_bv_get_num_constraints(self.h, C.byref(v)))
        return v.value

    def _upload(self, W):
        """Columns of a host matrix as device vectors (a scratch BV keeps them alive): (holder, pointer array)."""
        W = np.asarray(W, dtype=np.float64)
        tmp = BV(self.ctx, self.n, max(W.shape[1], 1), N=self.N)
        tmp.set_dense(W)
        ptrs = (C.c_void_p * W.shape[1])(*[tmp.column_ptr(j) for j in range(W.shape[1])])
        return tmp, ptrs

    def InsertVecs(self, s, W, orth=True):
        """BVInsertVecs(V,s,&m,W,orth) with the vectors given as the columns of a host matrix: returns the number kept."""
        tmp, ptrs = self._upload(W)
        m = C.c_int(len(ptrs))
        _lib.check(self.ctx.L.ks_bv_insert_vecs(self.h, s, C.byref(m), ptrs, int(bool(orth))))
        return m.value

    def InsertConstraints(self, Cmat):
        """BVInsertConstraints(V,&nc,C): destructive; returns the number of independent constraints kept."""
        tmp, ptrs = self._upload(Cmat)
        nc = C.c_int(len(ptrs))
        _lib.check(self.ctx.L.ks_bv_insert_constraints(self.h, C.byref(nc), ptrs))
        return nc.value

    def SetNumConstraints(self, nc):
        _lib.check(self.ctx.L.ks_bv_set_num_constraints(self.h, nc))
        mm = C.c_int()
        _lib.check(self.ctx.L.ks_bv_get_sizes(self.h, None, None, C.byref(mm), None))
        self.m = mm.value

    def constraints_dense(self):
        nc = self.nc
        return np.stack([self.column(j) for j in range(-nc, 0)], axis=1) if nc else np.zeros((self.n, 0))

    def SetRandomColumn(self, j, seed=0x12345678):
        _lib.check(self.ctx.L.ks_bv_set_random_column(self.h, j, seed))

    def SetRandom(self, seed=0x12345678):
        _lib.check(self.ctx.L.ks_bv_set_random(self.h, seed))

    def Resize(self, m, copy=True):
        _lib.check(self.ctx.L.ks_bv_resize(self.h, m, int(bool(copy))))
        self.m = m

    def InsertVec(self, j, w_ptr):
        _lib.check(self.ctx.L.ks_bv_insert_vec(self.h, j, C.c_void_p(w_ptr)))

    def CopyVec(self, j, w_ptr):
        _lib.check(self.ctx.L.ks_bv_copy_vec(self.h, j, C.c_void_p(w_ptr)))

    # -- ops
    def Mult(self, alpha, beta, X, Q=None):
        if Q is None:
            _lib.check(self.ctx.L.ks_bv_mult(self.h, alpha, beta, X.h, None, 0))
        else:
            Qf = np.asfortranarray(Q, dtype=np.float64)
            _lib.check(self.ctx.L.ks_bv_mult(self.h, alpha, beta, X.h, _p(Qf), Qf.shape[0]))

    def MultVec(self, alpha, beta, y_ptr, q=None):
        q = None if q is None else _f64(q)
        _lib.check(self.ctx.L.ks_bv_multvec(self.h, alpha, beta, C.c_void_p(y_ptr), _p(q)))

    def MultColumn(self, alpha, beta, j, q=None):
        q = None if q is None else _f64(q)
        _lib.check(self.ctx.L.ks_bv_multcolumn(self.h, alpha, beta, j, _p(q)))

    def MultInPlace(self, Q, s, e, trans=False):
        Qf = np.asfortranarray(Q, dtype=np.float64)
        f = self.ctx.L.ks_bv_multinplace_trans if trans else self.ctx.L.ks_bv_multinplace
        _lib.check(f(self.h, _p(Qf), Qf.shape[0], s, e))

    def Dot(self, Y, M):
        assert M.flags.f_contiguous and M.dtype == np.float64
        _lib.check(self.ctx.L.ks_bv_dot(self.h, Y.h, _p(M), M.shape[0]))

    def DotVec(self, y_ptr, to_buffer=False):
        if to_buffer:
            _lib.check(self.ctx.L.ks_bv_dotvec(self.h, C.c_void_p(y_ptr), None))
            return None
        out = np.zeros(max(self.k - self.l, 0))
        _lib.check(self.ctx.L.ks_bv_dotvec(self.h, C.c_void_p(y_ptr), _p(out)))
        return out

    def DotColumn(self, j, q=True):
        if q is None:
            _lib.check(self.ctx.L.ks_bv_dotcolumn(self.h, j, None))
            return None
        out = np.zeros(max(j - self.l, 0))
        _lib.check(self.ctx.L.ks_bv_dotcolumn(self.h, j, _p(out)))
        return out

    def Scale(self, alpha):
        _lib.check(self.ctx.L.ks_bv_scale(self.h, alpha))

    def ScaleColumn(self, j, alpha):
        _lib.check(self.ctx.L.ks_bv_scalecolumn(self.h, j, alpha))

    def Norm(self, type=NORM_FROBENIUS):
        v = C.c_double()
        _lib.check(self.ctx.L.ks_bv_norm(self.h, type, C.byref(v)))
        return v.value

    def NormVec(self, v_ptr, type=NORM_2):
        v = C.c_double()
        _lib.check(self.ctx.L.ks_bv_normvec(self.h, C.c_void_p(v_ptr), type, C.byref(v)))
        return v.value

    def NormColumn(self, j, type=NORM_2):
        v = C.c_double()
        _lib.check(self.ctx.L.ks_bv_normcolumn(self.h, j, type, C.byref(v)))
        return v.value

    def Copy(self, W):
        _lib.check(self.ctx.L.ks_bv_copy(self.h, W.h))

    def CopyColumn(self, j, i):
        _lib.check(self.ctx.L.ks_bv_copycolumn(self.h, j, i))

    def MatMult(self, A, W):
        _lib.check(self.ctx.L.ks_bv_matmult(self.h, A.h, W.h))

    def MatMultColumn(self, A, j):
        _lib.check(self.ctx.L.ks_bv_matmultcolumn(self.h, A.h, j))

    def OrthogonalizeColumn(self, j):
        nrm = C.c_double(); lin = C.c_int()
        H = np.zeros(max(j - self.l, 0) + 1)
        _lib.check(self.ctx.L.ks_bv_orthogonalizecolumn(self.h, j, _p(H), C.byref(nrm), C.byref(lin)))
        return H[: max(j - self.l, 0)], nrm.value, bool(lin.value)

    def GramSchmidtPass(self, j, want_onrm=True, want_nrm=True):
        """ops->gramschmidt: ONE pass on column j with the coefficients in the buffer; returns (onrm, nrm), None where not asked for."""
        o = C.c_double(); nr = C.c_double()
        _lib.check(self.ctx.L.ks_bv_gramschmidt_pass(self.h, j, None, None, None, None, C.byref(o) if want_onrm else None, C.byref(nr) if want_nrm else None))
        return (o.value if want_onrm else None), (nr.value if want_nrm else None)

    def SetState(self, state):
        """The caller's modification counter (PetscObjectStateGet on the BV): arms the pass chaining of the ops->gramschmidt slot."""
        _lib.check(self.ctx.L.ks_bv_set_state(self.h, state))

    def GsChainStats(self):
        a = C.c_longlong(); b = C.c_longlong()
        _lib.check(self.ctx.L.ks_bv_gs_chain_stats(self.h, C.byref(a), C.byref(b)))
        return {"chained": a.value, "fresh": b.value}

    def buffer_ptr(self):
        p = C.c_void_p(); _lib.check(self.ctx.L.ks_bv_get_buffer(self.h, C.byref(p))); return p.value

    def SetMatrix(self, B):
        """BVSetMatrix(bv,B,PETSC_FALSE); B a Mat (kept alive here) or None."""
        self._B = B
        _lib.check(self.ctx.L.ks_bv_set_matrix(self.h, None if B is None else B.h))

    def SetOrthogBlock(self, block):
        _lib.check(self.ctx.L.ks_bv_set_orthog_block(self.h, BLOCK.get(block, block)))

    def Orthogonalize(self, R=None):
        """BVOrthogonalize: R (Fortran-ordered, at least k x k) receives the triangular factor in its columns l..k-1."""
        if R is not None:
            assert R.flags.f_contiguous
        _lib.check(self.ctx.L.ks_bv_orthogonalize(self.h, _p(R) if R is not None else None, R.shape[0] if R is not None else 0))

    def MatProject(self, A, Y, M):
        assert M.flags.f_contiguous
        _lib.check(self.ctx.L.ks_bv_matproject(self.h, None if A is None else A.h, Y.h, _p(M), M.shape[0]))

    def Normalize(self, eigi=None):
        _lib.check(self.ctx.L.ks_bv_normalize(self.h, _p(_f64(eigi)) if eigi is not None else None))

    def OrthonormalizeColumn(self, j, replace=False):
        nrm = C.c_double(); lin = C.c_int()
        _lib.check(self.ctx.L.ks_bv_orthonormalizecolumn(self.h, j, int(replace), C.byref(nrm), C.byref(lin)))
        return nrm.value, bool(lin.value)

    def OrthogonalizeVec(self, v_ptr):
        nrm = C.c_double(); lin = C.c_int()
        H = np.zeros(max(self.k - self.l, 1))
        _lib.check(self.ctx.L.ks_bv_orthogonalizevec(self.h, C.c_void_p(v_ptr), _p(H), C.byref(nrm), C.byref(lin)))
        return H, nrm.value, bool(lin.value)

    def OrthogonalizeSomeColumn(self, j, which):
        nrm = C.c_double(); lin = C.c_int()
        w = np.ascontiguousarray(which, dtype=np.int32)
        _lib.check(self.ctx.L.ks_bv_orthogonalizesomecolumn(self.h, j, _pi(w), None, C.byref(nrm), C.byref(lin)))
        return nrm.value, bool(lin.value)

    def gs_passes(self):
        t = C.c_longlong(); l = C.c_int()
        _lib.check(self.ctx.L.ks_bv_gs_passes(self.h, C.byref(t), C.byref(l)))
        return t.value, l.value

    # -- split reductions: Begin queues, the first End reduces everything queued with one allreduce
    def DotVecBegin(self, v_ptr):
        out = np.zeros(max(self.k - self.l, 1))
        _lib.check(self.ctx.L.ks_bv_dotvec_begin(self.h, C.c_void_p(v_ptr), _p(out)))
        return out                                             # filled by DotVecEnd(v_ptr, out)

    def DotVecEnd(self, v_ptr, out):
        _lib.check(self.ctx.L.ks_bv_dotvec_end(self.h, C.c_void_p(v_ptr), _p(out)))
        return out[: self.k - self.l]

    def DotColumnBegin(self, j):
        out = np.zeros(max(j - self.l, 1))
        _lib.check(self.ctx.L.ks_bv_dotcolumn_begin(self.h, j, _p(out)))
        return out

    def DotColumnEnd(self, j, out):
        _lib.check(self.ctx.L.ks_bv_dotcolumn_end(self.h, j, _p(out)))
        return out[: j - self.l]

    def NormVecBegin(self, v_ptr, type=NORM_2):
        _lib.check(self.ctx.L.ks_bv_normvec_begin(self.h, C.c_void_p(v_ptr), type, _p(np.zeros(1))))

    def NormVecEnd(self, v_ptr, type=NORM_2):
        out = np.zeros(1); _lib.check(self.ctx.L.ks_bv_normvec_end(self.h, C.c_void_p(v_ptr), type, _p(out))); return float(out[0])

    def NormColumnBegin(self, j, type=NORM_2):
        _lib.check(self.ctx.L.ks_bv_normcolumn_begin(self.h, j, type, _p(np.zeros(1))))

    def NormColumnEnd(self, j, type=NORM_2):
        out = np.zeros(1); _lib.check(self.ctx.L.ks_bv_normcolumn_end(self.h, j, type, _p(out))); return float(out[0])

    def MatLanczos(self, A, T, k, m):
        assert T.flags.f_contiguous and T.dtype == np.float64 and T.shape[1] >= 2
        mm = C.c_int(m); beta = C.c_double(); brk = C.c_int()
        _lib.check(self.ctx.L.ks_bv_matlanczos(self.h, A.h, _p(T), T.shape[0], k, C.byref(mm), C.byref(beta), C.byref(brk)))
        return mm.value, beta.value, bool(brk.value)

    def MatArnoldi(self, A, H, k, m):
        assert H.flags.f_contiguous and H.dtype == np.float64
        mm = C.c_int(m); beta = C.c_double(); brk = C.c_int()
        _lib.check(self.ctx.L.ks_bv_matarnoldi(self.h, A.h, _p(H), H.shape[0], k, C.byref(mm), C.byref(beta), C.byref(brk)))
        return mm.value, beta.value, bool(brk.value)


class ST:
    """Spectral transformation (STSHIFT / STSINVERT, shell matrix mode, GMRES + Jacobi inner solves)."""

    def __init__(self, ctx, _handle=None):
        self.ctx = ctx
        self._owned = _handle is None
        if _handle is None:
            _handle = C.c_void_p(); _lib.check(ctx.L.ks_st_create(ctx.h, C.byref(_handle)))
        self.h = _handle

    def destroy(self):
        if self._owned and self.h and self.ctx.h:
            self.ctx.L.ks_st_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    def SetType(self, t):
        _lib.check(self.ctx.L.ks_st_set_type(self.h, {"shift": ST_SHIFT, "sinvert": ST_SINVERT, "cayley": ST_CAYLEY}.get(t, t)))

    def SetShift(self, sigma):
        _lib.check(self.ctx.L.ks_st_set_shift(self.h, sigma))

    def GetShift(self):
        v = C.c_double(); _lib.check(self.ctx.L.ks_st_get_shift(self.h, C.byref(v))); return v.value

    def SetMatrices(self, A, B=None):
        _lib.check(self.ctx.L.ks_st_set_matrices(self.h, A.h, None if B is None else B.h))
        self._A, self._B = A, B

    def CayleySetAntishift(self, nu):
        _lib.check(self.ctx.L.ks_st_cayley_set_antishift(self.h, nu))

    def CayleyGetAntishift(self):
        v = C.c_double(); _lib.check(self.ctx.L.ks_st_cayley_get_antishift(self.h, C.byref(v))); return v.value

    def SetKSP(self, rtol=0.0, max_it=0, restart=0):
        _lib.check(self.ctx.L.ks_st_set_ksp(self.h, rtol, max_it, restart))

    def SetGMRESCGSRefinement(self, t):
        """KSPGMRESSetCGSRefinementType on the ST's KSP: "never" (PETSc's default), "ifneeded" or "always"."""
        _lib.check(self.ctx.L.ks_st_set_gmres_cgs_refinement(self.h, {"ifneeded": 0, "never": 1, "always": 2}.get(t, t)))

    def SetPC(self, t, block_size=0):
        """PCSetType on the ST's KSP: "jacobi" (default) or "bjacobi" with blocks of block_size consecutive local rows, solved exactly."""
        _lib.check(self.ctx.L.ks_st_set_pc(self.h, {"jacobi": 0, "bjacobi": 1}.get(t, t), block_size))

    def SetMatMode(self, mode):
        """STSetMatMode: "shell" (default here) or "copy" (P = A - sigma B assembled; the matrices need keep_csr)."""
        _lib.check(self.ctx.L.ks_st_set_matmode(self.h, {"copy": 0, "shell": 2}.get(mode, mode)))

    def GetMatMode(self):
        v = C.c_int(); _lib.check(self.ctx.L.ks_st_get_matmode(self.h, C.byref(v))); return {0: "copy", 2: "shell"}[v.value]

    def SetKSPType(self, t):
        """KSPSetType on the ST's KSP: "gmres" (default) or "bcgs"."""
        _lib.check(self.ctx.L.ks_st_set_ksp_type(self.h, {"gmres": 0, "bcgs": 1}.get(t, t)))

    def SetUp(self):
        _lib.check(self.ctx.L.ks_st_setup(self.h))

    def Apply(self, x):
        """y = Op x with host vectors (test convenience, single rank)."""
        x = _f64(x)
        W = BV(self.ctx, len(x), 2)
        W.set_column(0, x)
        _lib.check(self.ctx.L.ks_st_apply(self.h, C.c_void_p(W.column_ptr(0)), C.c_void_p(W.column_ptr(1))))
        return W.column(1)

    def ApplyTranspose(self, x):
        """y = Op^T x with host vectors (STApplyHermitianTranspose, real scalars; test convenience, single rank)."""
        x = _f64(x)
        W = BV(self.ctx, len(x), 2)
        W.set_column(0, x)
        _lib.check(self.ctx.L.ks_st_apply_transpose(self.h, C.c_void_p(W.column_ptr(0)), C.c_void_p(W.column_ptr(1))))
        return W.column(1)

    def BackTransform(self, eigr, eigi):
        r = _f64(np.atleast_1d(eigr)).copy(); i = _f64(np.atleast_1d(eigi)).copy()
        _lib.check(self.ctx.L.ks_st_backtransform(self.h, len(r), _p(r), _p(i)))
        return r, i

    def GetKSPStats(self):
        s = C.c_longlong(); it = C.c_longlong(); r = C.c_double()
        _lib.check(self.ctx.L.ks_st_get_ksp_stats(self.h, C.byref(s), C.byref(it), C.byref(r)))
        return {"solves": s.value, "iterations": it.value, "last_rnorm": r.value}


class EPS:
    """EPSCreate/EPSSetOperators/EPSSolve... for the default Krylov-Schur solver (symmetric problems)."""

    def __init__(self, ctx):
        self.ctx = ctx
        h = C.c_void_p()
        _lib.check(ctx.L.ks_eps_create(ctx.h, C.byref(h)))
        self.h = h
        self._A = None

    def destroy(self):
        if self.h and self.ctx.h:
            self.ctx.L.ks_eps_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    def SetOperators(self, A, B=None):
        _lib.check(self.ctx.L.ks_eps_set_operators(self.h, A.h, None if B is None else B.h))
        self._A = A; self._B = B

    def GetTolerances(self):
        t = C.c_double(); m = C.c_int(); _lib.check(self.ctx.L.ks_eps_get_tolerances(self.h, C.byref(t), C.byref(m))); return t.value, m.value

    def GetWhichEigenpairs(self):
        v = C.c_int(); _lib.check(self.ctx.L.ks_eps_get_which_eigenpairs(self.h, C.byref(v))); return v.value

    def GetTarget(self):
        v = C.c_double(); _lib.check(self.ctx.L.ks_eps_get_target(self.h, C.byref(v))); return v.value

    def GetConvergenceTest(self):
        v = C.c_int(); _lib.check(self.ctx.L.ks_eps_get_convergence_test(self.h, C.byref(v))); return v.value

    def GetProblemType(self):
        """(type, is_generalized, is_hermitian, is_positive)"""
        v = [C.c_int() for _ in range(4)]
        _lib.check(self.ctx.L.ks_eps_get_problem_type(self.h, *[C.byref(x) for x in v]))
        return v[0].value, bool(v[1].value), bool(v[2].value), bool(v[3].value)

    def GetST(self):
        h = C.c_void_p(); _lib.check(self.ctx.L.ks_eps_get_st(self.h, C.byref(h)))
        return ST(self.ctx, _handle=h)

    def SetProblemType(self, t):
        _lib.check(self.ctx.L.ks_eps_set_problem_type(self.h, t))

    def SetDimensions(self, nev, ncv=0, mpd=0):
        _lib.check(self.ctx.L.ks_eps_set_dimensions(self.h, nev, ncv or 0, mpd or 0))

    def SetTolerances(self, tol=0.0, max_it=0):
        _lib.check(self.ctx.L.ks_eps_set_tolerances(self.h, tol or 0.0, max_it or 0))

    def SetWhichEigenpairs(self, which):
        _lib.check(self.ctx.L.ks_eps_set_which_eigenpairs(self.h, WHICH.get(which, which)))

    def SetTarget(self, target):
        _lib.check(self.ctx.L.ks_eps_set_target(self.h, target))

    def SetEigenvalueComparison(self, func):
        """func(ar, ai, br, bi) -> negative if a is preferred, positive if b is (SlepcEigenvalueComparisonFn)."""
        def tramp(ar, ai, br, bi, res, _ctx):
            res[0] = int(func(ar, ai, br, bi))
            return 0
        self._cmp_cb = EIG_COMPARE_FN(tramp)          # keep the trampoline alive as long as the solver
        _lib.check(self.ctx.L.ks_eps_set_eigenvalue_comparison(self.h, C.cast(self._cmp_cb, C.c_void_p), None))

    def SetConvergenceTest(self, conv):
        _lib.check(self.ctx.L.ks_eps_set_convergence_test(self.h, {"abs": 0, "rel": 1, "norm": 2}.get(conv, conv)))

    def KrylovSchurSetLocking(self, lock):
        _lib.check(self.ctx.L.ks_eps_set_krylovschur_locking(self.h, int(bool(lock))))

    def KrylovSchurSetRestart(self, keep):
        _lib.check(self.ctx.L.ks_eps_set_krylovschur_restart(self.h, keep))

    def SetRandomSeed(self, seed):
        _lib.check(self.ctx.L.ks_eps_set_random_seed(self.h, seed))

    def SetInitialVector(self, v):
        _lib.check(self.ctx.L.ks_eps_set_initial_vector(self.h, _p(_f64(v)) if v is not None else None))

    def SetExtraction(self, extr):
        """EPSSetExtraction: "ritz" or "harmonic" (target from SetTarget)."""
        _lib.check(self.ctx.L.ks_eps_set_extraction(self.h, {"ritz": 0, "harmonic": 1}.get(extr, extr)))

    def GetExtraction(self):
        v = C.c_int(); _lib.check(self.ctx.L.ks_eps_get_extraction(self.h, C.byref(v))); return v.value

    def SetConvergenceTestFunction(self, func):
        """func(eigr, eigi, res) -> error estimate (EPSSetConvergenceTestFunction); None restores the relative test."""
        if func is None:
            self._conv_cb = None
            _lib.check(self.ctx.L.ks_eps_set_convergence_test_function(self.h, None, None)); return

        def tramp(_eps, re, im, res, out, _ctx):
            try:
                out[0] = float(func(re, im, res)); return 0
            except Exception:       # noqa: BLE001 - must not unwind through C
                import traceback; traceback.print_exc(); return 76
        self._conv_cb = EPS_CONVERGED_FN(tramp)
        _lib.check(self.ctx.L.ks_eps_set_convergence_test_function(self.h, C.cast(self._conv_cb, C.c_void_p), None))

    def SetStoppingTestFunction(self, func):
        """func(its, max_it, nconv, nev) -> reason (0 = keep iterating); EPS.StoppingBasic is the default rule."""
        if func is None:
            self._stop_cb = None
            _lib.check(self.ctx.L.ks_eps_set_stopping_test_function(self.h, None, None)); return

        def tramp(_eps, its, max_it, nconv, nev, reason, _ctx):
            try:
                reason[0] = int(func(its, max_it, nconv, nev)); return 0
            except Exception:       # noqa: BLE001
                import traceback; traceback.print_exc(); return 76
        self._stop_cb = EPS_STOPPING_FN(tramp)
        _lib.check(self.ctx.L.ks_eps_set_stopping_test_function(self.h, C.cast(self._stop_cb, C.c_void_p), None))

    def StoppingBasic(self, its, max_it, nconv, nev):
        r = C.c_int()
        _lib.check(self.ctx.L.ks_eps_stopping_basic(self.h, its, max_it, nconv, nev, C.byref(r), None))
        return r.value

    def MonitorSet(self, func):
        """func(its, nconv, eigr, eigi, errest) with numpy copies of the first nest entries; None cancels."""
        if func is None:
            self._mon_cb = None
            _lib.check(self.ctx.L.ks_eps_monitor_set(self.h, None, None)); return

        def tramp(_eps, its, nconv, er, ei, ee, nest, _ctx):
            try:
                func(its, nconv, np.array(er[:nest]), np.array(ei[:nest]), np.array(ee[:nest])); return 0
            except Exception:       # noqa: BLE001
                import traceback; traceback.print_exc(); return 76
        self._mon_cb = EPS_MONITOR_FN(tramp)
        _lib.check(self.ctx.L.ks_eps_monitor_set(self.h, C.cast(self._mon_cb, C.c_void_p), None))

    def SetArbitrarySelection(self, func):
        """func(eigr, eigi, xr, xi) -> (rr, ri) with the Ritz vector as host arrays (copied from the device for the call);
        EPSSetArbitrarySelection. None disables."""
        if func is None:
            self._arb_cb = None
            _lib.check(self.ctx.L.ks_eps_set_arbitrary_selection(self.h, None, None)); return
        n = self._A.n

        def tramp(re, im, xr, xi, rr, ri, _ctx):
            try:
                hx = np.empty(n); hy = np.empty(n)
                self.ctx.memcpy_d2h(hx, xr); self.ctx.memcpy_d2h(hy, xi)
                a, b = func(re, im, hx, hy)
                rr[0] = float(a); ri[0] = float(b); return 0
            except Exception:       # noqa: BLE001
                import traceback; traceback.print_exc(); return 76
        self._arb_cb = EPS_ARBITRARY_FN(tramp)
        _lib.check(self.ctx.L.ks_eps_set_arbitrary_selection(self.h, C.cast(self._arb_cb, C.c_void_p), None))

    def SetBalance(self, bal="oneside", its=0, cutoff=0.0):
        """EPSSetBalance: "none" or "oneside" (non-symmetric problems)."""
        _lib.check(self.ctx.L.ks_eps_set_balance(self.h, {"none": 0, "oneside": 1, "twoside": 2, "user": 3}.get(bal, bal), its, cutoff))

    def SetPurify(self, flag=True):
        _lib.check(self.ctx.L.ks_eps_set_purify(self.h, int(bool(flag))))

    def SetTrackAll(self, flag=True):
        _lib.check(self.ctx.L.ks_eps_set_track_all(self.h, int(bool(flag))))

    def KrylovSchurGet(self):
        k = C.c_double(); l = C.c_int(); _lib.check(self.ctx.L.ks_eps_get_krylovschur(self.h, C.byref(k), C.byref(l))); return k.value, bool(l.value)

    def SetBalanceMatrix(self, D):
        """EPS_BALANCE_USER with the diagonal D (host array of the local rows)."""
        tmp = BV(self.ctx, self._A.n, 1); tmp.set_column(0, _f64(D))
        _lib.check(self.ctx.L.ks_eps_set_balance_matrix(self.h, C.c_void_p(tmp.column_ptr(0))))

    def SetTrueResidual(self, flag=True):
        _lib.check(self.ctx.L.ks_eps_set_true_residual(self.h, int(bool(flag))))

    def GetTrueResidual(self):
        v = C.c_int(); _lib.check(self.ctx.L.ks_eps_get_true_residual(self.h, C.byref(v))); return bool(v.value)

    def SetInitialSpace(self, Vmat):
        """EPSSetInitialSpace with the vectors as the columns of a host matrix (uploaded, passed as device vectors)."""
        Vmat = np.asarray(Vmat, dtype=np.float64)
        tmp = BV(self.ctx, Vmat.shape[0], max(Vmat.shape[1], 1))
        tmp.set_dense(Vmat)
        ptrs = (C.c_void_p * Vmat.shape[1])(*[tmp.column_ptr(j) for j in range(Vmat.shape[1])])
        _lib.check(self.ctx.L.ks_eps_set_initial_space(self.h, Vmat.shape[1], ptrs))

    def SetDeflationSpace(self, Cmat):
        """EPSSetDeflationSpace with the vectors given as the columns of a host matrix (local rows of this rank)."""
        Cmat = np.asarray(Cmat, dtype=np.float64)
        if Cmat.size == 0:
            _lib.check(self.ctx.L.ks_eps_set_deflation_space(self.h, 0, None)); return
        tmp = BV(self.ctx, Cmat.shape[0], Cmat.shape[1])
        tmp.set_dense(Cmat)
        ptrs = (C.c_void_p * Cmat.shape[1])(*[tmp.column_ptr(j) for j in range(Cmat.shape[1])])
        _lib.check(self.ctx.L.ks_eps_set_deflation_space(self.h, Cmat.shape[1], ptrs))

    def SetDSParallel(self, synchronized=True):
        """DSSetParallel on the solver's DS: broadcast rank 0's projected solve after every restart (default) or trust
        the redundant computation."""
        _lib.check(self.ctx.L.ks_eps_set_ds_parallel(self.h, 1 if synchronized else 0))

    def SetMaxSteps(self, steps):
        _lib.check(self.ctx.L.ks_eps_set_max_steps(self.h, steps))

    def Solve(self):
        _lib.check(self.ctx.L.ks_eps_solve(self.h))

    def GetConverged(self):
        v = C.c_int(); _lib.check(self.ctx.L.ks_eps_get_converged(self.h, C.byref(v))); return v.value

    def GetIterationNumber(self):
        v = C.c_int(); _lib.check(self.ctx.L.ks_eps_get_iteration_number(self.h, C.byref(v))); return v.value

    def GetConvergedReason(self):
        v = C.c_int(); _lib.check(self.ctx.L.ks_eps_get_converged_reason(self.h, C.byref(v))); return v.value

    def GetDimensions(self):
        a = C.c_int(); b = C.c_int(); c = C.c_int()
        _lib.check(self.ctx.L.ks_eps_get_dimensions(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def GetEigenvalue(self, i):
        r = C.c_double(); im = C.c_double()
        _lib.check(self.ctx.L.ks_eps_get_eigenvalue(self.h, i, C.byref(r), C.byref(im)))
        return r.value, im.value

    def GetEigenvector(self, i):
        x = np.empty(self._A.n)
        _lib.check(self.ctx.L.ks_eps_get_eigenvector_host(self.h, i, _p(x)))
        return x

    def GetEigenpair(self, i):
        """EPSGetEigenpair: (kr, ki, xr, xi) with the reference's conjugate-pair convention."""
        r = C.c_double(); im = C.c_double()
        xr = np.empty(self._A.n); xi = np.empty(self._A.n)
        _lib.check(self.ctx.L.ks_eps_get_eigenpair_host(self.h, i, C.byref(r), C.byref(im), _p(xr), _p(xi)))
        return r.value, im.value, xr, xi

    def GetErrorEstimate(self, i):
        v = C.c_double(); _lib.check(self.ctx.L.ks_eps_get_error_estimate(self.h, i, C.byref(v))); return v.value

    def ComputeError(self, i, type=EPS_ERROR_RELATIVE):
        v = C.c_double(); _lib.check(self.ctx.L.ks_eps_compute_error(self.h, i, type, C.byref(v))); return v.value

    def GetEigenpairDev(self, i, xr_ptr, xi_ptr=None):
        """EPSGetEigenpair into device vectors (raw pointers of n_local doubles); returns (eigr, eigi)."""
        kr = C.c_double(); ki = C.c_double()
        _lib.check(self.ctx.L.ks_eps_get_eigenpair(self.h, i, C.byref(kr), C.byref(ki), C.c_void_p(xr_ptr) if xr_ptr else None,
                                                   C.c_void_p(xi_ptr) if xi_ptr else None))
        return kr.value, ki.value

    def GetInvariantSubspace(self):
        """EPSGetInvariantSubspace as a host matrix (n_local x nconv)."""
        k = self.GetConverged()
        tmp = BV(self.ctx, self._A.n, max(k, 1))
        ptrs = (C.c_void_p * max(k, 1))(*[tmp.column_ptr(j) for j in range(max(k, 1))])
        _lib.check(self.ctx.L.ks_eps_get_invariant_subspace(self.h, ptrs))
        return tmp.dense()[:, :k]

    def GetBV(self):
        h = C.c_void_p(); _lib.check(self.ctx.L.ks_eps_get_bv(self.h, C.byref(h)))
        return BV(self.ctx, 0, 0, _handle=h)

    def GetStats(self):
        s = C.c_longlong(); p = C.c_longlong(); r = C.c_int()
        _lib.check(self.ctx.L.ks_eps_get_stats(self.h, C.byref(s), C.byref(p), C.byref(r)))
        return {"arnoldi_steps": s.value, "gs_passes": p.value, "restarts": r.value}
