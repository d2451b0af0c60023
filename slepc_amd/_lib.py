"""ctypes binding of libksgpu.so (include/ksgpu.h). No fallback: if the HIP library is missing the import fails."""
import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libksgpu.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "ksgpu.h")

KS_UNIQUE_ID_BYTES = 128

dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)
vp = C.c_void_p
llp = C.POINTER(C.c_longlong)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, vp, vp, C.c_int, vp)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, vp, vp, C.c_int, vp)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, vp, C.c_int, ip, vp, ip, ip, vp, ip, ip, C.c_int, vp)


class CommOps(C.Structure):
    _fields_ = [("allreduce_sum", ALLREDUCE_FN), ("allgather_host", ALLGATHER_FN), ("exchange", EXCHANGE_FN)]


class KsError(RuntimeError):
    def __init__(self, rc, msg):
        super().__init__("libksgpu error %d: %s" % (rc, msg))
        self.rc = rc


_SIG = {
    # context
    "ks_ctx_create": [C.c_int, vp, C.POINTER(vp)],
    "ks_ctx_destroy": [vp],
    "ks_ctx_synchronize": [vp],
    "ks_ctx_sync_count": [vp, llp],
    "ks_ctx_set_debug": [vp, C.c_int, C.c_longlong],
    "ks_ctx_device_info": [vp, C.c_char_p, C.c_int, ip, C.POINTER(C.c_size_t)],
    "ks_runtime_info": [C.c_char_p, C.c_int],
    "ks_runtime_allow_multiple": [C.c_int],
    "ks_comm_get_unique_id": [C.c_char_p],
    "ks_comm_init_rccl": [vp, C.c_int, C.c_int, C.c_char_p],
    "ks_comm_set_ops": [vp, C.c_int, C.c_int, C.POINTER(CommOps), vp],
    "ks_ctx_memcpy": [vp, vp, vp, C.c_size_t, C.c_int],
    "ks_ctx_memcpy_stream": [vp, vp, vp, C.c_size_t, C.c_int, vp],
    "ks_ctx_memset": [vp, vp, C.c_int, C.c_size_t],
    "ks_comm_rank_size": [vp, ip, ip],
    "ks_comm_check": [vp],
    "ks_comm_bcast_stats": [vp, llp, dp, C.c_int],
    "ks_comm_set_allreduce": [vp, C.c_int, ip],
    "ks_comm_get_allreduce": [vp, ip],
    "ks_comm_allreduce_sum": [vp, vp, C.c_int],
    # mat
    "ks_mat_create_csr": [vp, C.c_int, C.c_int, C.c_int, ip, ip, dp, C.POINTER(vp)],
    "ks_mat_create_csr_flags": [vp, C.c_int, C.c_int, C.c_int, ip, ip, dp, C.c_uint, C.POINTER(vp)],
    "ks_mat_create_axpy": [vp, C.c_double, vp, C.c_uint, C.POINTER(vp)],
    "ks_mat_set_halo": [vp, C.c_int, ip],
    "ks_mat_mult_transpose": [vp, vp, vp],
    "ks_mat_shell_set_mult_transpose": [vp, vp],
    "ks_st_apply_transpose": [vp, vp, vp],
    "ks_mat_get_halo": [vp, ip],
    "ks_mat_create_laplacian3d": [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)],
    "ks_mat_create_laplacian2d": [vp, C.c_int, C.c_int, C.POINTER(vp)],
    "ks_mat_destroy": [vp],
    "ks_mat_get_layout": [vp, ip],
    "ks_mat_load_petsc_binary": [vp, C.c_char_p, C.POINTER(vp)],
    "ks_mat_create_shell": [vp, C.c_int, C.c_int, C.c_int, vp, vp, C.POINTER(vp)],
    "ks_mat_shell_set_enqueue_only": [vp, C.c_int],
    "ks_mat_get_diagonal": [vp, vp],
    "ks_mat_norm_inf": [vp, dp],
    "ks_mat_get_sizes": [vp, ip, ip, llp],
    "ks_mat_mult": [vp, vp, vp],
    "ks_mat_mult_host": [vp, dp, dp],
    # bv
    "ks_bv_create": [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)],
    "ks_bv_destroy": [vp],
    "ks_bv_duplicate": [vp, C.POINTER(vp)],
    "ks_bv_get_sizes": [vp, ip, ip, ip, ip],
    "ks_bv_set_ownership_start": [vp, C.c_int],
    "ks_bv_set_active_columns": [vp, C.c_int, C.c_int],
    "ks_bv_get_active_columns": [vp, ip, ip],
    "ks_bv_set_orthogonalization": [vp, C.c_int, C.c_int, C.c_double],
    "ks_bv_get_array": [vp, C.POINTER(vp)],
    "ks_bv_get_column": [vp, C.c_int, C.POINTER(vp)],
    "ks_bv_get_buffer": [vp, C.POINTER(vp)],
    "ks_bv_set_buffer": [vp, vp],
    "ks_bv_set_layout": [vp, C.c_int, C.c_int],
    "ks_bv_set_column_host": [vp, C.c_int, dp],
    "ks_bv_get_column_host": [vp, C.c_int, dp],
    "ks_bv_get_buffer_host": [vp, dp],
    "ks_bv_set_random_column": [vp, C.c_int, C.c_uint64],
    "ks_bv_mult": [vp, C.c_double, C.c_double, vp, dp, C.c_int],
    "ks_bv_multvec": [vp, C.c_double, C.c_double, vp, dp],
    "ks_bv_multcolumn": [vp, C.c_double, C.c_double, C.c_int, dp],
    "ks_bv_multinplace": [vp, dp, C.c_int, C.c_int, C.c_int],
    "ks_bv_multinplace_trans": [vp, dp, C.c_int, C.c_int, C.c_int],
    "ks_bv_dot": [vp, vp, dp, C.c_int],
    "ks_bv_dotvec": [vp, vp, dp],
    "ks_bv_dotvec_local": [vp, vp, dp],
    "ks_bv_dotcolumn": [vp, C.c_int, dp],
    "ks_bv_dotvec_begin": [vp, vp, dp], "ks_bv_dotvec_end": [vp, vp, dp],
    "ks_bv_dotcolumn_begin": [vp, C.c_int, dp], "ks_bv_dotcolumn_end": [vp, C.c_int, dp],
    "ks_bv_normvec_begin": [vp, vp, C.c_int, dp], "ks_bv_normvec_end": [vp, vp, C.c_int, dp],
    "ks_bv_normcolumn_begin": [vp, C.c_int, C.c_int, dp], "ks_bv_normcolumn_end": [vp, C.c_int, C.c_int, dp],
    "ks_bv_scale": [vp, C.c_double],
    "ks_bv_scalecolumn": [vp, C.c_int, C.c_double],
    "ks_bv_norm": [vp, C.c_int, dp],
    "ks_bv_normcolumn": [vp, C.c_int, C.c_int, dp],
    "ks_bv_norm_local": [vp, C.c_int, C.c_int, dp],
    "ks_bv_normvec": [vp, vp, C.c_int, dp],
    "ks_bv_copy": [vp, vp],
    "ks_bv_copycolumn": [vp, C.c_int, C.c_int],
    "ks_bv_matmult": [vp, vp, vp],
    "ks_bv_matmultcolumn": [vp, vp, C.c_int],
    "ks_bv_gramschmidt_pass": [vp, C.c_int, vp, ip, dp, dp, dp, dp],
    "ks_bv_orthogonalizecolumn": [vp, C.c_int, dp, dp, ip],
    "ks_bv_orthonormalizecolumn": [vp, C.c_int, C.c_int, dp, ip],
    "ks_bv_orthogonalizevec": [vp, vp, dp, dp, ip],
    "ks_bv_orthogonalizesomecolumn": [vp, C.c_int, ip, dp, dp, ip],
    "ks_bv_gs_passes": [vp, llp, ip],
    "ks_bv_set_state": [vp, C.c_uint64],
    "ks_bv_gs_chain_stats": [vp, llp, llp],
    "ks_bv_resize": [vp, C.c_int, C.c_int],
    "ks_bv_set_random": [vp, C.c_uint64],
    "ks_bv_insert_vec": [vp, C.c_int, vp],
    "ks_bv_insert_vecs": [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.c_int],
    "ks_bv_insert_constraints": [vp, C.POINTER(C.c_int), C.POINTER(C.c_void_p)],
    "ks_bv_set_num_constraints": [vp, C.c_int],
    "ks_bv_get_num_constraints": [vp, C.POINTER(C.c_int)],
    "ks_bv_copy_vec": [vp, C.c_int, vp],
    "ks_bv_set_orthog_block": [vp, C.c_int],
    "ks_bv_set_matrix": [vp, vp],
    "ks_bv_get_matrix": [vp, C.POINTER(vp)],
    "ks_bv_orthogonalize": [vp, dp, C.c_int],
    "ks_bv_matproject": [vp, vp, vp, dp, C.c_int],
    "ks_bv_normalize": [vp, dp],
    "ks_bv_matarnoldi": [vp, vp, dp, C.c_int, C.c_int, ip, dp, ip],
    "ks_bv_matlanczos": [vp, vp, dp, C.c_int, C.c_int, ip, dp, ip],
    # eps
    "ks_eps_create": [vp, C.POINTER(vp)],
    "ks_eps_destroy": [vp],
    "ks_eps_set_operators": [vp, vp, vp],
    "ks_eps_set_problem_type": [vp, C.c_int],
    "ks_eps_set_dimensions": [vp, C.c_int, C.c_int, C.c_int],
    "ks_eps_set_tolerances": [vp, C.c_double, C.c_int],
    "ks_eps_set_which_eigenpairs": [vp, C.c_int],
    "ks_eps_set_target": [vp, C.c_double],
    "ks_eps_set_krylovschur_locking": [vp, C.c_int],
    "ks_eps_set_convergence_test": [vp, C.c_int],
    "ks_eps_set_eigenvalue_comparison": [vp, C.c_void_p, vp],
    "ks_eps_set_krylovschur_restart": [vp, C.c_double],
    "ks_eps_set_random_seed": [vp, C.c_uint64],
    "ks_eps_set_initial_vector": [vp, dp],
    "ks_eps_set_deflation_space": [vp, C.c_int, C.POINTER(C.c_void_p)],
    "ks_eps_set_initial_space": [vp, C.c_int, C.POINTER(C.c_void_p)],
    "ks_eps_set_max_steps": [vp, C.c_longlong],
    "ks_eps_set_ds_parallel": [vp, C.c_int],
    "ks_eps_get_ds_parallel": [vp, ip],
    "ks_eps_solve": [vp],
    "ks_eps_get_converged": [vp, ip],
    "ks_eps_get_iteration_number": [vp, ip],
    "ks_eps_get_converged_reason": [vp, ip],
    "ks_eps_get_dimensions": [vp, ip, ip, ip],
    "ks_eps_get_eigenvalue": [vp, C.c_int, dp, dp],
    "ks_eps_get_eigenvector_host": [vp, C.c_int, dp],
    "ks_eps_get_eigenpair_host": [vp, C.c_int, dp, dp, dp, dp],
    "ks_eps_get_eigenpair": [vp, C.c_int, dp, dp, vp, vp],
    "ks_eps_get_error_estimate": [vp, C.c_int, dp],
    "ks_eps_get_invariant_subspace": [vp, C.POINTER(C.c_void_p)],
    "ks_eps_compute_error": [vp, C.c_int, C.c_int, dp],
    "ks_eps_get_bv": [vp, C.POINTER(vp)],
    "ks_eps_get_stats": [vp, llp, llp, ip],
    "ks_eps_get_st": [vp, C.POINTER(vp)],
    "ks_eps_get_tolerances": [vp, dp, ip],
    "ks_eps_get_which_eigenpairs": [vp, ip],
    "ks_eps_get_target": [vp, dp],
    "ks_eps_get_convergence_test": [vp, ip],
    "ks_eps_set_extraction": [vp, C.c_int],
    "ks_eps_get_extraction": [vp, C.POINTER(C.c_int)],
    "ks_eps_set_true_residual": [vp, C.c_int],
    "ks_eps_set_balance": [vp, C.c_int, C.c_int, C.c_double],
    "ks_eps_set_balance_matrix": [vp, vp],
    "ks_eps_set_purify": [vp, C.c_int], "ks_eps_get_purify": [vp, C.POINTER(C.c_int)], "ks_eps_set_track_all": [vp, C.c_int],
    "ks_eps_get_krylovschur": [vp, C.POINTER(C.c_double), C.POINTER(C.c_int)],
    "ks_eps_get_balance": [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double)],
    "ks_eps_set_convergence_test_function": [vp, vp, vp],
    "ks_eps_set_stopping_test_function": [vp, vp, vp],
    "ks_eps_stopping_basic": [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), vp],
    "ks_eps_monitor_set": [vp, vp, vp],
    "ks_eps_set_arbitrary_selection": [vp, vp, vp],
    "ks_eps_get_true_residual": [vp, C.POINTER(C.c_int)],
    "ks_eps_get_operators": [vp, C.POINTER(vp), C.POINTER(vp)],
    "ks_eps_get_problem_type": [vp, ip, ip, ip, ip],
    "ks_st_create": [vp, C.POINTER(vp)],
    "ks_st_destroy": [vp],
    "ks_st_set_type": [vp, C.c_int],
    "ks_st_set_shift": [vp, C.c_double],
    "ks_st_get_shift": [vp, dp],
    "ks_st_cayley_set_antishift": [vp, C.c_double],
    "ks_st_cayley_get_antishift": [vp, C.POINTER(C.c_double)],
    "ks_st_set_matrices": [vp, vp, vp],
    "ks_st_set_ksp": [vp, C.c_double, C.c_int, C.c_int],
    "ks_st_set_ksp_type": [vp, C.c_int],
    "ks_st_set_matmode": [vp, C.c_int],
    "ks_st_set_pc": [vp, C.c_int, C.c_int],
    "ks_st_set_gmres_cgs_refinement": [vp, C.c_int],
    "ks_st_get_matmode": [vp, C.POINTER(C.c_int)],
    "ks_st_setup": [vp],
    "ks_st_apply": [vp, vp, vp],
    "ks_st_backtransform": [vp, C.c_int, dp, dp],
    "ks_st_get_ksp_stats": [vp, llp, llp, dp],
    # profiling
    "ks_prof_enable": [vp, C.c_int],
    "ks_prof_reset": [vp],
    "ks_prof_get": [vp, C.c_int, C.c_int, llp, dp, dp, dp],
}
_STR_FUNCS = ("ks_error_string", "ks_last_error_message", "ks_prof_class_name", "ks_prof_event_name")


def header_symbols():
    """Every function name declared in include/ksgpu.h."""
    txt = open(HEADER_PATH).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ks_[a-z0-9_]+)\s*\(", txt)) - {"ks_comm_ops"})


def mapped_hip_runtimes():
    """Distinct libamdhip64 files this process maps (the C side's view: ks_runtime_info)."""
    out = []
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                i = line.find("/")
                if i < 0:
                    continue
                path = line[i:].strip()
                if os.path.basename(path).startswith("libamdhip64.so") and path not in out:
                    out.append(path)
    except OSError:
        pass
    return out


def _torch_bundled_runtime():
    """Path of the libamdhip64.so a PyTorch wheel bundles, without importing torch; None if torch is absent or bundles none."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return None
    path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    return path if os.path.exists(path) else None


_preloaded_runtime = None


def bind_hip_runtime():
    """ONE HIP runtime per process, whatever the import order. libksgpu.so NEEDs the SONAME libamdhip64.so.7; its RUNPATH finds /opt/rocm's copy.
    PyTorch's wheel bundles its own copy under the FILE name libamdhip64.so with that same SONAME: imported before the library, torch's copy is the one
    the library binds to (SONAME match); imported after it, torch asks for the file name, which matches no loaded SONAME, and a second HIP + HSA runtime
    pair comes up in the process (round 3: later occupancy queries of the library failed). So when nothing is mapped yet and torch is installed with a
    bundled runtime, that copy is mapped first, deliberately: the library binds to it and a later `import torch` finds the same file already mapped."""
    global _preloaded_runtime
    if mapped_hip_runtimes():
        return
    if os.environ.get("SLEPC_AMD_HIP_RUNTIME") == "system":
        # harness option: leave the choice to the library's RUNPATH (/opt/rocm's runtime, what a SLEPc build without PyTorch runs on) - for running the
        # GPU suite on that runtime too. Only safe in a process that does not import torch afterwards (ks_ctx_create refuses two runtimes anyway).
        return
    path = _torch_bundled_runtime()
    if path is not None:
        C.CDLL(path, mode=C.RTLD_GLOBAL)
        _preloaded_runtime = path


def load():
    if not os.path.exists(LIB_PATH):
        raise ImportError("libksgpu.so not found at %s: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc, gfx950). There is no CPU fallback." % LIB_PATH)
    bind_hip_runtime()
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, args in _SIG.items():
        f = getattr(lib, name)
        f.argtypes = args
        f.restype = C.c_int
    lib.ks_error_string.argtypes = [C.c_int]
    lib.ks_error_string.restype = C.c_char_p
    lib.ks_last_error_message.argtypes = []
    lib.ks_last_error_message.restype = C.c_char_p
    lib.ks_prof_class_name.argtypes = [C.c_int]
    lib.ks_prof_class_name.restype = C.c_char_p
    lib.ks_prof_event_name.argtypes = [C.c_int]
    lib.ks_prof_event_name.restype = C.c_char_p
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = load()
    return _lib


def runtime_info():
    """ks_runtime_info as a dict: the HIP runtime the library is bound to, every libamdhip64 the process maps, failed occupancy queries so far."""
    import json
    buf = C.create_string_buffer(2048)
    check(lib().ks_runtime_info(buf, 2048))
    d = json.loads(buf.value.decode())
    d["preloaded_by_binding"] = _preloaded_runtime
    return d


def check(rc):
    if rc != 0:
        L = lib()
        detail = L.ks_last_error_message().decode()
        raise KsError(rc, "%s: %s" % (L.ks_error_string(rc).decode(), detail))
