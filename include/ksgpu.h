/*
 * ksgpu.h -- C ABI of libksgpu: MI355X (gfx950) kernels for the SLEPc EPS Krylov-Schur hot path.
 *
 * Drop-in boundary.  Each entry point replaces one slot of the reference's plugin interfaces on raw
 * pointers (no PETSc types, no torch types).  Citations are SLEPc 3.22.2 paths (file:line):
 *
 *   struct _BVOps (include/slepc/private/bvimpl.h:25-61)        ->  ks_bv_*  below
 *   MatMult(Mat,Vec,Vec) reached through the ST shell
 *       (src/sys/classes/st/interface/stsolve.c:16-25,244-259)    ->  ks_mat_mult
 *   BVMatArnoldi / BVMatLanczos (bv/interface/bvkrylov.c:56,165)  ->  ks_bv_matarnoldi / ks_bv_matlanczos
 *   EPSSetOperators/EPSSolve/EPSGetEigenpair/EPSComputeError
 *       (src/eps/interface/epssetup.c:450, epssolve.c:119,406,742) ->  ks_eps_*
 *
 * Conventions (mirror the reference):
 *   - every function returns an int error code, 0 = success; non-zero values reuse PETSc's
 *     PetscErrorCode numbers (KS_ERR_*), so an adapter can `return (PetscErrorCode)rc;`
 *   - scalars are real double (PetscScalar), indices 32-bit int (PetscInt default build)
 *   - BV storage is ONE device array of (nc+m)*ld doubles, column-major, like BVSVEC
 *     (src/sys/classes/bv/impls/svec/svec.c:397-565); every op acts on the active window
 *     [l,k) at array+(nc+l)*ld (svec.c:30,47,124); nc constraint columns sit in front (ks_bv_insert_constraints)
 *   - Q / M / H arguments are caller-owned HOST column-major arrays, replicated on all ranks
 *     (bvops.c:33-36); q / m coefficient arrays are host arrays, NULL means "use the BV's
 *     device-resident buffer Vec" exactly as in BVMultVec/BVDotVec (svec.c:46,123)
 *   - numerical conditions (lindep, breakdown) are flags, not errors (bvkrylov.c:92-97)
 *   - single-threaded per context, collective across ranks: every rank calls in the same order
 *   - all device work is enqueued on the context's HIP stream; functions that return a host
 *     value synchronise that stream, the others do not.
 */
#ifndef KSGPU_H
#define KSGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes (PETSc numbering) ---------------------------------------------------------- */
#define KS_SUCCESS              0
#define KS_ERR_MEM             55
#define KS_ERR_SUP             56
#define KS_ERR_ORDER           58
#define KS_ERR_ARG_SIZ         60
#define KS_ERR_ARG_IDN         61
#define KS_ERR_FILE_OPEN       65
#define KS_ERR_FILE_UNEXPECTED 79
#define KS_ERR_ARG_WRONG       62
#define KS_ERR_ARG_OUTOFRANGE  63
#define KS_ERR_MAT_LU_ZRPVT    71   /* zero pivot in a block of the block-Jacobi preconditioner */
#define KS_ERR_USER_INPUT      95   /* BV_SafeSqrt "Invalid inner product" bvimpl.h:137 (PETSC_ERR_USER_INPUT; 71, used until round 3, is PETSC_ERR_MAT_LU_ZRPVT) */
#define KS_ERR_ARG_WRONGSTATE  73
#define KS_ERR_ARG_INCOMP      75
#define KS_ERR_LIB             76   /* HIP / RCCL runtime failure (PetscCallHIP analogue) */
#define KS_ERR_PLIB            77
#define KS_ERR_CONV_FAILED     82
#define KS_ERR_ARG_NULL        85
#define KS_ERR_NOT_CONVERGED   91   /* inner linear solve (KSPSetErrorIfNotConverged, stsles.c:58) */
#define KS_ERR_GPU             97   /* no usable gfx950 device */

const char *ks_error_string(int rc);
const char *ks_last_error_message(void);     /* detail of the last failure on this thread */

typedef struct ks_st_s  *ks_st;    /* spectral transformation: the operator of the Krylov expansion */
typedef struct ks_ctx_s *ks_ctx;   /* device + stream + communicator + profiling state            */
typedef struct ks_mat_s *ks_mat;   /* sparse operator (CSR / AIJ), row block owned by this rank   */
typedef struct ks_bv_s  *ks_bv;    /* basis vectors                                               */
typedef struct ks_eps_s *ks_eps;   /* Krylov-Schur eigensolver driver                             */

/* enums mirror include/slepcbv.h, include/slepceps.h, PETSc NormType */
enum { KS_BV_ORTHOG_CGS = 0, KS_BV_ORTHOG_MGS = 1 };
enum { KS_BV_ORTHOG_REFINE_IFNEEDED = 0, KS_BV_ORTHOG_REFINE_NEVER = 1, KS_BV_ORTHOG_REFINE_ALWAYS = 2 };
/* BVOrthogBlockType, include/slepcbv.h */
enum { KS_BV_ORTHOG_BLOCK_GS = 0, KS_BV_ORTHOG_BLOCK_CHOL = 1, KS_BV_ORTHOG_BLOCK_TSQR = 2, KS_BV_ORTHOG_BLOCK_TSQRCHOL = 3, KS_BV_ORTHOG_BLOCK_SVQB = 4 };
enum { KS_NORM_1 = 0, KS_NORM_2 = 1, KS_NORM_FROBENIUS = 2, KS_NORM_INFINITY = 3 };
/* EPSWhich, include/slepceps.h:109-119 (same numbering; TARGET_IMAGINARY and ALL are not offered) */
enum { KS_EPS_LARGEST_MAGNITUDE = 1, KS_EPS_SMALLEST_MAGNITUDE = 2, KS_EPS_LARGEST_REAL = 3, KS_EPS_SMALLEST_REAL = 4,
       KS_EPS_LARGEST_IMAGINARY = 5, KS_EPS_SMALLEST_IMAGINARY = 6, KS_EPS_TARGET_MAGNITUDE = 7, KS_EPS_TARGET_REAL = 8,
       KS_EPS_WHICH_USER = 11 };
/* SlepcEigenvalueComparisonFn (include/slepcsc.h): *res < 0 if a is preferred to b, > 0 if b is preferred, 0 if equal */
typedef int (*ks_eig_compare_fn)(double ar, double ai, double br, double bi, int *res, void *ctx);
enum { KS_EPS_HEP = 1, KS_EPS_GHEP = 2, KS_EPS_NHEP = 3, KS_EPS_GNHEP = 4 };   /* EPSProblemType, slepceps.h */
enum { KS_ST_SHIFT = 0, KS_ST_SINVERT = 1, KS_ST_CAYLEY = 2 };   /* STType "shift", "sinvert", "cayley" */
enum { KS_EPS_ERROR_ABSOLUTE = 0, KS_EPS_ERROR_RELATIVE = 1, KS_EPS_ERROR_BACKWARD = 2 };   /* EPSErrorType */
enum { KS_EPS_RITZ = 0, KS_EPS_HARMONIC = 1, KS_EPS_HARMONIC_RELATIVE, KS_EPS_HARMONIC_RIGHT, KS_EPS_HARMONIC_LARGEST, KS_EPS_REFINED, KS_EPS_REFINED_HARMONIC };  /* EPSExtraction slepceps.h:94-100; Krylov-Schur offers the first two */
enum { KS_EPS_BALANCE_NONE = 0, KS_EPS_BALANCE_ONESIDE = 1, KS_EPS_BALANCE_TWOSIDE = 2, KS_EPS_BALANCE_USER = 3 };   /* EPSBalance slepceps.h; the two-sided form needs the transposed product (ks_mat_mult_transpose) */
enum { KS_EPS_CONV_ABS = 0, KS_EPS_CONV_REL = 1, KS_EPS_CONV_NORM = 2, KS_EPS_CONV_USER = 3 };   /* EPSConv slepceps.h:153-156 */
/* user callbacks of the solver (slepceps.h EPSConvergenceTestFn, EPSStoppingTestFn, EPSMonitorFn); non-zero return = error */
typedef int (*ks_eps_converged_fn)(ks_eps eps, double eigr, double eigi, double res, double *errest, void *ctx);
typedef int (*ks_eps_stopping_fn)(ks_eps eps, int its, int max_it, int nconv, int nev, int *reason, void *ctx);
/* SlepcArbitrarySelectionFn (slepcsc.h): (eigr, eigi, xr, xi, &rr, &ri, ctx); xr/xi are device vectors of n_local doubles */
typedef int (*ks_eps_arbitrary_fn)(double eigr, double eigi, const double *xr_dev, const double *xi_dev, double *rr, double *ri, void *ctx);
typedef int (*ks_eps_monitor_fn)(ks_eps eps, int its, int nconv, const double *eigr, const double *eigi, const double *errest, int nest, void *ctx);
enum { KS_EPS_CONVERGED_TOL = 1, KS_EPS_CONVERGED_USER = 2, KS_EPS_DIVERGED_ITS = -1, KS_EPS_DIVERGED_BREAKDOWN = -2,
       KS_EPS_DIVERGED_SYMMETRY_LOST = -3, KS_EPS_CONVERGED_ITERATING = 0 };

/* ---- context -------------------------------------------------------------------------------- */
/* stream: a hipStream_t owned by the caller (e.g. torch's current stream) or NULL -> own stream. */
int ks_ctx_create(int device, void *stream, ks_ctx *ctx);
int ks_ctx_destroy(ks_ctx ctx);
int ks_ctx_synchronize(ks_ctx ctx);
/* Test hooks (compiled with -DKSD_TEST_HOOKS, which the in-tree build sets): each makes a test run the path the fast one replaces, to compare bits,
 * or a multi-rank path on one rank. Set BEFORE the objects they affect are created: NO_FUSED_GS is read at ks_bv_create, FORCE_MULTI at
 * ks_comm_init_rccl / ks_comm_set_ops, ONESHOT_SEQ0 (first stamp of the one-shot allreduce) at ks_comm_set_allreduce; the others per call. */
enum { KS_DEBUG_NO_FUSED_GS = 0, KS_DEBUG_NO_MFMA = 1, KS_DEBUG_NO_SPMV_DOT = 2, KS_DEBUG_FORCE_MULTI = 3, KS_DEBUG_HALO_OVERLAP = 4, KS_DEBUG_ONESHOT_SEQ0 = 5 };
int ks_ctx_set_debug(ks_ctx ctx, int key, long long value);
int ks_ctx_sync_count(ks_ctx ctx, long long *count);   /* instrumentation: host waits on the context's stream made by the library so far */
int ks_ctx_device_info(ks_ctx ctx, char *arch, int arch_len, int *num_cu, size_t *mem_total);
/* Which HIP runtime the library's calls are bound to. A process can map TWO copies of libamdhip64 (PyTorch wheels bundle one under the file name
 * libamdhip64.so with the SONAME libamdhip64.so.7: loaded AFTER this library they become a second runtime, loaded BEFORE it the dynamic linker binds
 * this library to that copy). ks_ctx_create refuses a process that maps more than one; a process that gains a second one later is reported here and by
 * one warning on stderr at the first occupancy query that fails. Writes a JSON object: {"hip_runtime_path", "hip_runtime_version",
 * "hip_runtimes_mapped": [paths], "occupancy_query_failures", "occupancy_last_error"}. No context needed; never initialises the GPU. */
int ks_runtime_info(char *json, int len);
/* Diagnosis only: let ks_ctx_create proceed in a process that maps two HIP runtimes (scripts/hip_runtime_probe.py reproduces round 3's failures with it). */
int ks_runtime_allow_multiple(int allow);

/* Row-wise distribution (PetscLayout, bvbasic.c:129-134).  Reductions inside BV ops
   (bvblas.c:218,255; bvlapack.c:50) become an allreduce over all ranks; MatMult exchanges the
   boundary entries of x with the owning ranks (PETSc VecScatter inside MatMult_MPIAIJ).
   Two providers: native RCCL over xGMI (ks_comm_init_rccl) or three caller-supplied operations
   (ks_comm_set_ops; e.g. GPU-aware MPI: MPIU_Allreduce / MPI_Allgather / MPI_Isend+Irecv).      */
#define KS_UNIQUE_ID_BYTES 128
int ks_comm_get_unique_id(unsigned char id[KS_UNIQUE_ID_BYTES]);                       /* rank 0, then broadcast by the launcher */
int ks_comm_init_rccl(ks_ctx ctx, int rank, int size, const unsigned char id[KS_UNIQUE_ID_BYTES]);
typedef struct ks_comm_ops {
  /* in-place SUM of `count` doubles in DEVICE memory, ordered on `stream` */
  int (*allreduce_sum)(void *user, double *dev_buf, int count, void *stream);
  /* setup only: every rank contributes `bytes` HOST bytes; recv gets size*bytes, in rank order */
  int (*allgather_host)(void *user, const void *send, int bytes, void *recv);
  /* neighbour exchange on DEVICE buffers, ordered on `stream`: to/from peers[i], segments
     [off[i], off[i]+cnt[i]) in units of elem_bytes */
  int (*exchange)(void *user, int npeers, const int *peers, const void *dev_send, const int *send_off, const int *send_cnt,
                  void *dev_recv, const int *recv_off, const int *recv_cnt, int elem_bytes, void *stream);
} ks_comm_ops;
int ks_comm_set_ops(ks_ctx ctx, int rank, int size, const ks_comm_ops *ops, void *user);
int ks_comm_rank_size(ks_ctx ctx, int *rank, int *size);
/* Known-answer run of the installed communicator: an allreduce with a closed-form sum, two allgathers of one int per rank and
 * a ring exchange with ranks rank+1 / rank-1. Collective; returns KS_ERR_LIB with a message naming the first mismatch. A no-op
 * on a single rank without forced collectives. (The reference leans on MPI's own correctness here; a communicator handed in
 * through ks_comm_set_ops is foreign code, so the library offers the check an integrator runs once after installing it.) */
int ks_comm_check(ks_ctx ctx);
/* Host wall time spent so far in the broadcast of rank 0's projected problem (one per restart with KS_DS_PARALLEL_SYNCHRONIZED, the DSSynchronize analogue,
   dshep.c:673-713): number of broadcasts and their seconds (upload, ncclBroadcast or the provider's allgather, download, host wait); reset != 0 clears them. */
int ks_comm_bcast_stats(ks_ctx ctx, long long *calls, double *seconds, int reset);
/* Which allreduce the Gram-Schmidt passes use. KS_ALLREDUCE_PROVIDER: the communicator's own (ncclAllReduce with the
 * RCCL provider; what bvblas.c:255 MPIU_Allreduce is to the reference). KS_ALLREDUCE_ONESHOT: one kernel per rank writes the k+1
 * values into a mailbox of every rank (peer-mapped uncached device memory: hipIpc between processes, xGMI between GPUs) and adds
 * what arrived in rank order - identical bits on all ranks, no library call on the critical path; reductions longer than 128
 * doubles keep going through the provider. Collective; *active returns what was installed: the one-shot path is taken only if
 * every rank could map every mailbox, else all ranks stay on the provider. A rank that receives no packet within
 * KSGPU_ONESHOT_TIMEOUT_MS (default 2000) returns NaN and the next host wait fails with KS_ERR_LIB - never a hang. */
#define KS_ALLREDUCE_PROVIDER 0
#define KS_ALLREDUCE_ONESHOT 1
int ks_comm_set_allreduce(ks_ctx ctx, int kind, int *active);
int ks_comm_get_allreduce(ks_ctx ctx, int *active);
/* in-place SUM over the ranks of `count` doubles in device memory, ordered on the context's stream (bvblas.c:255 MPIU_Allreduce): the
 * reduction every library kernel uses, for callers that keep their own replicated device scalars (an adapter's VecDot) */
int ks_comm_allreduce_sum(ks_ctx ctx, double *dev_buf, int count);
/* device<->host copy on the context's stream (synchronous); kind: 0 = host->device, 1 = device->host */
int ks_ctx_memcpy(ks_ctx ctx, void *dst, const void *src, size_t bytes, int kind);
/* the same on a given stream (the `stream` argument a ks_comm_ops callback receives; NULL = the context's): what a provider that
   stages through the host uses to order itself after the work already enqueued there */
int ks_ctx_memcpy_stream(ks_ctx ctx, void *dst, const void *src, size_t bytes, int kind, void *stream);
/* fill device memory on the context's stream; enqueues only (what BV_CleanCoefficients_HIP's hipMemset, bvhip.hip.cpp:345-360, is on the
   stream PETSc works on) */
int ks_ctx_memset(ks_ctx ctx, void *dev, int value, size_t bytes);

/* ---- Mat: the MatMult(AIJ) slot ------------------------------------------------------------- */
/* CSR arrays as in PETSc SeqAIJ (i,j,a): rowptr[n_local+1], col[nnz] (GLOBAL column indices),
   val[nnz].  Host arrays are copied to the device.  row_start = first global row owned by this
   rank; n_global = global size.  With one rank: row_start=0, n_local=n_global.               */
int ks_mat_create_csr(ks_ctx ctx, int n_local, int row_start, int n_global,
                      const int *rowptr, const int *col, const double *val, ks_mat *A);
/* The same with options. KS_MAT_KEEP_CSR: the matrix keeps a host copy of the arrays it was created from, as a PETSc AIJ Mat keeps its
   own - what MatDuplicate / MatAXPY need later (ks_mat_create_axpy, ST_MATMODE_COPY); without it only the layout the product runs on
   survives the assembly. Matrices read by ks_mat_load_petsc_binary keep theirs. */
#define KS_MAT_KEEP_CSR 1u
int ks_mat_create_csr_flags(ks_ctx ctx, int n_local, int row_start, int n_global,
                            const int *rowptr, const int *col, const double *val, unsigned flags, ks_mat *A);
/* P = A + alpha B as a new matrix: MatDuplicate(A,MAT_COPY_VALUES,&P) + MatAXPY(P,alpha,B,DIFFERENT_NONZERO_PATTERN), B == NULL:
   MatShift(P,alpha) - the assembly of A - sigma B in ST_MATMODE_COPY (src/sys/classes/st/interface/stsolve.c:611-626). Entry by
   entry p_ij = a_ij + (alpha b_ij); rows with ascending columns come out ascending. A and B must hold their CSR arrays
   (KS_MAT_KEEP_CSR; KS_ERR_ORDER otherwise) and the same row block. flags as above, for P. */
int ks_mat_create_axpy(ks_mat A, double alpha, ks_mat B, unsigned flags, ks_mat *P);
/* Synthetic generators that build the SAME CSR arrays directly in device memory (bench inputs):
   3-D 7-pt Laplacian of ex19.c:47-78 (rows of z-planes [z0,z0+nz_local) of an nx*ny*nz grid) and
   2-D 5-pt Laplacian of ex2.c:44-51.                                                           */
int ks_mat_create_laplacian3d(ks_ctx ctx, int nx, int ny, int nz, int z0, int nz_local, ks_mat *A);
int ks_mat_create_laplacian2d(ks_ctx ctx, int n, int m, ks_mat *A);
/* MatLoad from a PETSc binary viewer file (the .petsc files under share/slepc/datafiles/matrices, as ex4.c and ex7.c read them); every
   rank takes the row block PETSC_DECIDE would give it */
int ks_mat_load_petsc_binary(ks_ctx ctx, const char *path, ks_mat *A);
/* MatCreateShell + MATOP_MULT (the matrix-free route of src/eps/tutorials/ex3.c): y = mult(user, x) on device
   pointers of n_local doubles; the callback must order its work after everything already enqueued on the context's
   stream (enqueue on that stream, or synchronise) and leave y complete in that order.          */
typedef int (*ks_shell_mult_fn)(void *user, const double *x_dev, double *y_dev);
int ks_mat_create_shell(ks_ctx ctx, int n_local, int row_start, int n_global, ks_shell_mult_fn mult, void *user, ks_mat *A);
int ks_mat_norm_inf(ks_mat A, double *val);                             /* MatNorm(A,NORM_INFINITY) */
int ks_mat_get_diagonal(ks_mat A, double *d_dev);                      /* MatGetDiagonal (local diagonal block) */
int ks_mat_shell_set_enqueue_only(ks_mat A, int flag);   /* the callback only enqueues work on the context's stream: Krylov runs are then enqueued ahead through it */
int ks_mat_destroy(ks_mat A);
int ks_mat_get_sizes(ks_mat A, int *n_local, int *n_global, long long *nnz_local);
/* device layout chosen at assembly for the local diagonal block (KSGPU_SPMV=csr|csrregs|csrblock|csrvec|sell|dict|odict|binned|sliced overrides the choice): dictionary forms for
   stencil-like matrices with few distinct values / offsets, SELL-64 for short regular rows, CSR row blocks for ragged ones, BINNED (two streaming phases, every
   random access in LDS) for matrices whose columns scatter over a vector much larger than an L2; SLICED is round 1's layout for those */
enum { KS_MAT_LAYOUT_CSR = 0, KS_MAT_LAYOUT_SELL = 1, KS_MAT_LAYOUT_SLICED = 2, KS_MAT_LAYOUT_SHELL = 3, KS_MAT_LAYOUT_DICT = 4, KS_MAT_LAYOUT_ODICT = 5, KS_MAT_LAYOUT_BINNED = 6 };
int ks_mat_get_layout(ks_mat A, int *layout);
/* MatMult: y = A x on device pointers (x, y: n_local doubles owned by this rank).
   Multi-rank: performs the halo exchange of x (PETSc VecScatter inside MatMult_MPIAIJ).        */
int ks_mat_mult(ks_mat A, const double *x_dev, double *y_dev);
/* MatMultTranspose: y = A^T x. An assembled matrix builds its transpose on first use (MatTranspose on the host from the kept CSR arrays:
   KS_MAT_KEEP_CSR, KS_ERR_ORDER otherwise; one rank - the transpose of a row-sharded matrix is a redistribution, KS_ERR_SUP) and multiplies
   with it like any other matrix; a shell matrix needs ks_mat_shell_set_mult_transpose (MATOP_MULT_TRANSPOSE, as ex9.c:123 sets it).
   Used by the two-sided balancing (EPSBuildBalance_Krylov epsdefault.c:402-411). */
int ks_mat_mult_transpose(ks_mat A, const double *x_dev, double *y_dev);
int ks_mat_shell_set_mult_transpose(ks_mat A, ks_shell_mult_fn mult_transpose);
int ks_mat_mult_host(ks_mat A, const double *x_host, double *y_host);  /* convenience for tests (single rank) */
/* How MatMult moves the boundary entries of x between ranks. KS_HALO_PROVIDER: packed into a send buffer and handed to the
 * communicator's exchange (grouped ncclSend / ncclRecv with the RCCL provider; PETSc's VecScatter is the reference's). KS_HALO_PEER: the pack
 * kernel stores every boundary entry straight into the ghost mailbox of the rank that needs it (device memory of that rank mapped here:
 * hipIpc between processes, xGMI between GPUs), stamped and acknowledged per product, and the receiver copies its mailbox into its ghost
 * array - no library call per product. Collective; *active returns what was installed: the peer path only if every rank could map all
 * its neighbours' mailboxes (at most 16 neighbours per rank). A rank that waits longer than KSGPU_ONESHOT_TIMEOUT_MS (default 2000) for a
 * neighbour fills its ghosts with NaN and the next host wait fails with KS_ERR_LIB - never a hang.
 * Taking the peer path down is collective too: ks_mat_set_halo(A, KS_HALO_PROVIDER) (or a new KS_HALO_PEER set-up) lets every rank finish, closes the
 * mapped mailboxes and only then frees its own. Call it before ks_mat_destroy on a matrix whose peer halo is active; a destroy without it waits (bounded)
 * for the neighbours' last acknowledgements before freeing the mailbox, but cannot wait for them to unmap it. */
#define KS_HALO_PROVIDER 0
#define KS_HALO_PEER 1
int ks_mat_set_halo(ks_mat A, int kind, int *active);
int ks_mat_get_halo(ks_mat A, int *active);

/* ---- BV: the _BVOps slots ------------------------------------------------------------------- */
int ks_bv_create(ks_ctx ctx, int n_local, int n_global, int m, int ld /*0: default*/, ks_bv *bv); /* ops->create, BV_SetDefaultLD bvimpl.h:471 */
int ks_bv_destroy(ks_bv bv);                                                        /* ops->destroy */
int ks_bv_duplicate(ks_bv bv, ks_bv *out);                                           /* ops->duplicate (storage only; no copy) */
int ks_bv_get_sizes(ks_bv bv, int *n_local, int *n_global, int *m, int *ld);
int ks_bv_set_ownership_start(ks_bv bv, int row_start);                             /* first global row of this rank (PetscLayout rstart); used by the reproducible random vectors */
int ks_bv_set_active_columns(ks_bv bv, int l, int k);                               /* BVSetActiveColumns bvbasic.c:421 */
int ks_bv_get_active_columns(ks_bv bv, int *l, int *k);
int ks_bv_set_orthogonalization(ks_bv bv, int type, int refine, double eta);        /* BVSetOrthogonalization; eta<=0 keeps 0.7071 */
int ks_bv_get_array(ks_bv bv, double **dev);                                        /* ops->getarray: device pointer of the m*ld block */
int ks_bv_get_column(ks_bv bv, int j, double **dev);                                /* ops->getcolumn: device pointer view of column j (j<0: constraint) */
int ks_bv_get_buffer(ks_bv bv, double **dev);                                       /* BVGetBufferVec bvbasic.c:775: (nc+m)*m device doubles */
int ks_bv_set_buffer(ks_bv bv, double *dev);                                        /* BVSetBufferVec bvbasic.c:720 on a raw device array of (nc+m)*m doubles (caller-owned; NULL: the library's own) */
int ks_bv_set_layout(ks_bv bv, int nc, int m);                                      /* mirror of the fields BVSetNumConstraints changes (bvbasic.c:291-296); moves no data */
int ks_bv_set_column_host(ks_bv bv, int j, const double *host);                     /* H2D of n_local doubles */
int ks_bv_get_column_host(ks_bv bv, int j, double *host);                           /* D2H, synchronises */
int ks_bv_get_buffer_host(ks_bv bv, double *host);                                  /* D2H of the (nc+m)*m coefficient buffer */
int ks_bv_resize(ks_bv bv, int m, int copy);                                    /* BVResize bvbasic.c:190 (ops->resize) */
int ks_bv_set_random(ks_bv bv, uint64_t seed);                                   /* BVSetRandom bvops.c:380: all active columns */
int ks_bv_insert_vec(ks_bv bv, int j, const double *w_dev);                      /* BVInsertVec bvops.c:568 */
int ks_bv_copy_vec(ks_bv bv, int j, double *w_dev);                              /* BVCopyVec bvops.c:484 */
int ks_bv_insert_vecs(ks_bv bv, int s, int *m, const double *const *W_dev, int orth); /* BVInsertVecs bvfunc.c:331: *m device vectors into columns s..; orth drops dependent ones, *m = kept */
int ks_bv_insert_constraints(ks_bv bv, int *nc, const double *const *C_dev);     /* BVInsertConstraints bvfunc.c:411: destructive; *nc = kept; constraints are columns -nc..-1 */
int ks_bv_set_num_constraints(ks_bv bv, int nc);                                 /* BVSetNumConstraints bvbasic.c:260 */
int ks_bv_get_num_constraints(ks_bv bv, int *nc);                                /* BVGetNumConstraints bvbasic.c:310 */
int ks_bv_set_random_column(ks_bv bv, int j, uint64_t seed);                        /* BVSetRandomColumn with -bv_reproducible_random semantics */

int ks_bv_mult(ks_bv Y, double alpha, double beta, ks_bv X, const double *Q, int ldq);          /* ops->mult; Q NULL -> Y=beta*Y+alpha*X */
int ks_bv_multvec(ks_bv X, double alpha, double beta, double *y_dev, const double *q);          /* ops->multvec; q NULL -> buffer */
int ks_bv_multcolumn(ks_bv X, double alpha, double beta, int j, const double *q);               /* BVMultColumn bvops.c:165 */
int ks_bv_multinplace(ks_bv V, const double *Q, int ldq, int s, int e);                          /* ops->multinplace */
int ks_bv_multinplace_trans(ks_bv V, const double *Q, int ldq, int s, int e);                    /* ops->multinplacetrans */
int ks_bv_dot(ks_bv X, ks_bv Y, double *M, int ldm);                                             /* ops->dot: M = Y^H X (+allreduce) */
int ks_bv_dotvec(ks_bv X, const double *y_dev, double *m);                                       /* ops->dotvec; m NULL -> buffer */
int ks_bv_dotvec_local(ks_bv X, const double *y_dev, double *m);                                 /* ops->dotvec_local (no reduction) */
int ks_bv_dotcolumn(ks_bv X, int j, double *q);                                                  /* BVDotColumn bvglobal.c:302 */
/* split reductions (bvglobal.c:207-300, 343-430, 573-660, 705-790): the Begin calls queue this rank's parts on the device, the first
   End reduces everything queued with ONE allreduce; Ends in the order of the Begins. Norms: 2-norm (or the B-norm) only. */
int ks_bv_dotvec_begin(ks_bv X, const double *y_dev, double *m);       int ks_bv_dotvec_end(ks_bv X, const double *y_dev, double *m);
int ks_bv_dotcolumn_begin(ks_bv X, int j, double *q);                  int ks_bv_dotcolumn_end(ks_bv X, int j, double *q);
int ks_bv_normvec_begin(ks_bv bv, const double *v_dev, int type, double *val);  int ks_bv_normvec_end(ks_bv bv, const double *v_dev, int type, double *val);
int ks_bv_normcolumn_begin(ks_bv bv, int j, int type, double *val);    int ks_bv_normcolumn_end(ks_bv bv, int j, int type, double *val);
int ks_bv_scale(ks_bv bv, double alpha);                                                         /* ops->scale(-1,alpha) */
int ks_bv_scalecolumn(ks_bv bv, int j, double alpha);                                            /* ops->scale(j,alpha) */
int ks_bv_norm(ks_bv bv, int type, double *val);                                                 /* ops->norm(-1,type) */
int ks_bv_normcolumn(ks_bv bv, int j, int type, double *val);                                    /* ops->norm(j,type) */
int ks_bv_norm_local(ks_bv bv, int j, int type, double *val);                                    /* ops->norm_local */
int ks_bv_normvec(ks_bv bv, const double *v_dev, int type, double *val);                         /* BVNormVec bvglobal.c:530: norm of a device vector, B-norm when a matrix is set */
int ks_bv_copy(ks_bv V, ks_bv W);                                                                /* ops->copy */
int ks_bv_copycolumn(ks_bv V, int j, int i);                                                     /* ops->copycolumn */
int ks_bv_matmult(ks_bv V, ks_mat A, ks_bv W);                                                   /* ops->matmult (column loop, svec.c:213) */
int ks_bv_matmultcolumn(ks_bv V, ks_mat A, int j);                                               /* BVMatMultColumn bvops.c:862 */

/* ops->gramschmidt (bvimpl.h:53): ONE Gram-Schmidt pass, the function BVOrthogonalizeGS1 dispatches to (bvorthog.c:134) in place
   of BVOrthogonalizeCGS1 (:91-132) / BVOrthogonalizeMGS1 (:52-85), chosen by the BV's orthogonalization type. The caller
   (BVOrthogonalizeGS :145-217) owns the refinement loop, lindep, BV_CleanCoefficients and BV_SetValue.
     v_dev NULL : column j against the constraints and columns 0..j-1;  v_dev given: that device vector against columns [-nc, j)
     which      : MGS only, may be NULL
     h, c       : HOST arrays of nc+m entries (bv->h, bv->c), or both NULL = column j and the scratch column of the BV's buffer;
                  the pass ADDS its coefficients to h (BV_AddCoefficients), c holds this pass's coefficients
     onrm, nrm  : norm before the pass / estimated norm after it (explicit when the estimate breaks down); either may be NULL
   Returns KS_ERR_USER_INPUT for an invalid inner product (BV_SafeSqrt).
   Column form, CGS, standard inner product: the pass is ONE dot sweep and ONE update launch; onrm / nrm reach the host while the update
   still runs (the call waits for the pass's bookkeeping, not for the stream). The other forms synchronise.                          */
int ks_bv_gramschmidt_pass(ks_bv bv, int j, double *v_dev, const int *which, double *h, double *c, double *onrm, double *nrm);
/* Pass chaining for the slot above. `state` is the caller's modification counter of the BV - PetscObjectStateGet((PetscObject)bv,&state):
   PETSc bumps it in BVRestoreColumn (when the Vec was written), BVScaleColumn, BVMultInPlace ... (bvbasic.c:1176, bvops.c:356,243), i.e. whenever the contents may have
   changed, and NOT between the passes BVOrthogonalizeGS makes on one column (bvorthog.c:176-202). Once a state has been announced, a pass whose
   bookkeeping predicts that the caller's refinement loop comes back (same policy, same eta: ks_bv_set_orthogonalization) leaves the dot
   products of the next pass behind, and the next ks_bv_gramschmidt_pass on the same column under the SAME state uses them instead of a dot
   sweep: 3 reads of the basis per CGS2 step instead of 4. Any other state, column or intervening sweep of this BV: the pass takes its own
   dots, as before. A caller that never calls this gets the unchained behaviour. The caller promises that the state changes whenever BV storage
   or the coefficient buffer is written by anyone but ks_bv_gramschmidt_pass itself. */
int ks_bv_set_state(ks_bv bv, uint64_t state);
/* instrumentation: passes of the slot that were chained to their predecessor's dots / that ran their own dot sweep */
int ks_bv_gs_chain_stats(ks_bv bv, long long *chained, long long *fresh);
/* The whole of BVOrthogonalizeColumn / BVOrthonormalizeColumn (not ops slots: the entry points a caller uses that drives this
   library directly): fused, device-resident classical Gram-Schmidt of column j against columns [0,j) with the reference's
   refinement policy; coefficients accumulate in the buffer column j.                              */
int ks_bv_orthogonalizecolumn(ks_bv bv, int j, double *H, double *norm, int *lindep);            /* BVOrthogonalizeColumn bvorthog.c:315 */
int ks_bv_orthonormalizecolumn(ks_bv bv, int j, int replace, double *norm, int *lindep);         /* BVOrthonormalizeColumn bvorthog.c:380 */
int ks_bv_orthogonalizevec(ks_bv bv, double *v_dev, double *H, double *norm, int *lindep);       /* BVOrthogonalizeVec bvorthog.c:247 */
/* BVOrthogonalize bvorthog.c:729: QR of the active columns, V0 = V R, leading columns untouched. R (host, column-major,
   ldr >= k) may be NULL; only its columns l..k-1 are written (upper triangular except for SVQB). The block method is the
   fourth argument of BVSetOrthogonalization.                                                                          */
int ks_bv_set_orthog_block(ks_bv bv, int block);
/* BVSetMatrix(bv,B,PETSC_FALSE) bvfunc.c:200: inner products, norms of columns and every orthogonalisation use y^H B x
   (B symmetric positive definite; borrowed; NULL restores the standard inner product) */
int ks_bv_set_matrix(ks_bv bv, ks_mat B);
int ks_bv_get_matrix(ks_bv bv, ks_mat *B);
int ks_bv_orthogonalize(ks_bv V, double *R, int ldr);
int ks_bv_matproject(ks_bv X, ks_mat A /* NULL: identity */, ks_bv Y, double *M, int ldm);   /* BVMatProject bvglobal.c:1014: M = Y^H A X */
int ks_bv_normalize(ks_bv V, const double *eigi /* may be NULL; entry 0 belongs to column l (what the ops->normalize slot receives after svec.c:196's offset) */);   /* BVNormalize bvglobal.c:855 */
int ks_bv_orthogonalizesomecolumn(ks_bv bv, int j, const int *which, double *H, double *norm, int *lindep); /* bvorthog.c:432 (MGS) */
int ks_bv_gs_passes(ks_bv bv, long long *passes_total, int *passes_last);                        /* instrumentation */

/* Krylov expansions (bvkrylov.c).  H: host ldh x >=m column-major; T: host, alpha = T[0..ldt),
   beta = T[ldt..2ldt) (DS_MAT_T layout).  *m may be reduced on breakdown.  The whole run of
   m-k steps is enqueued without host synchronisation; one D2H at the end (the reference's
   VecGetArrayRead(buf), bvkrylov.c:103,215).                                                   */
int ks_bv_matarnoldi(ks_bv V, ks_mat A, double *H, int ldh, int k, int *m, double *beta, int *breakdown);
int ks_bv_matlanczos(ks_bv V, ks_mat A, double *T, int ldt, int k, int *m, double *beta, int *breakdown);

/* ---- EPS: Krylov-Schur driver (host side; restates krylovschur.c:227-337) -------------------- */
int ks_eps_create(ks_ctx ctx, ks_eps *eps);
int ks_eps_destroy(ks_eps eps);
int ks_eps_set_operators(ks_eps eps, ks_mat A, ks_mat B /* NULL: standard problem */);   /* EPSSetOperators epssetup.c:450 */
int ks_eps_set_problem_type(ks_eps eps, int type);                         /* KS_EPS_HEP / KS_EPS_GHEP (Lanczos, B-inner product) | KS_EPS_NHEP / KS_EPS_GNHEP (Arnoldi) */
int ks_eps_set_dimensions(ks_eps eps, int nev, int ncv /*<=0: default*/, int mpd /*<=0: default*/);
int ks_eps_set_tolerances(ks_eps eps, double tol /*<=0: 1e-8*/, int max_it /*<=0: default*/);
int ks_eps_set_which_eigenpairs(ks_eps eps, int which);
int ks_eps_get_st(ks_eps eps, ks_st *st);                                    /* EPSGetST: borrowed, owned by the EPS */
int ks_eps_set_target(ks_eps eps, double target);                            /* EPSSetTarget epsopts.c:604 (sorting only: no spectral transformation) */
int ks_eps_set_eigenvalue_comparison(ks_eps eps, ks_eig_compare_fn fn, void *ctx);   /* EPSSetEigenvalueComparison epsopts.c:563 */
int ks_eps_set_krylovschur_restart(ks_eps eps, double keep);               /* EPSKrylovSchurSetRestart, default 0.5 */
int ks_eps_set_convergence_test(ks_eps eps, int conv);                    /* EPSSetConvergenceTest: KS_EPS_CONV_* (epsdefault.c:224-257) */
int ks_eps_set_krylovschur_locking(ks_eps eps, int lock);               /* EPSKrylovSchurSetLocking: 0 = non-locking variant (krylovschur.c:294) */
int ks_eps_set_random_seed(ks_eps eps, uint64_t seed);
int ks_eps_set_initial_vector(ks_eps eps, const double *v_host);           /* EPSSetInitialSpace with one vector */
int ks_eps_set_initial_space(ks_eps eps, int n, const double *const *v_dev);   /* EPSSetInitialSpace epssetup.c:590: device vectors; a Krylov solver uses the first */
int ks_eps_set_deflation_space(ks_eps eps, int n, const double *const *v_dev); /* EPSSetDeflationSpace epssetup.c:555: n device vectors, copied; used by the next solve only */
/* DSSetParallel on the solver's DS (dsbasic.c; krylovschur.c:281 DSSynchronize): with KS_DS_PARALLEL_SYNCHRONIZED, after every projected
   solve rank 0's projected matrix, vectors, eigenvalues and the outcome of the expansion (beta, length, breakdown) are broadcast, so the
   replicated control flow cannot diverge when a caller-supplied allreduce does not return identical bits on every rank. Default:
   synchronized (the reference's default is redundant). Cost per restart: 2 ld^2 + 2 ncv + 3 doubles from rank 0 - 16 KB at ncv = 30,
   66 KB at ncv = 64 - as one ncclBroadcast and one host wait with the RCCL provider; a caller-supplied provider has no broadcast
   slot and pays an allgather of that many bytes per rank. */
enum { KS_DS_PARALLEL_REDUNDANT = 0, KS_DS_PARALLEL_SYNCHRONIZED = 1 };
int ks_eps_set_ds_parallel(ks_eps eps, int pmode);
int ks_eps_get_ds_parallel(ks_eps eps, int *pmode);
int ks_eps_set_max_steps(ks_eps eps, long long max_steps);                 /* bench harness: stop after this many Arnoldi steps (0 = off) */
int ks_eps_solve(ks_eps eps);
int ks_eps_get_converged(ks_eps eps, int *nconv);
int ks_eps_get_iteration_number(ks_eps eps, int *its);
int ks_eps_get_converged_reason(ks_eps eps, int *reason);
int ks_eps_get_dimensions(ks_eps eps, int *nev, int *ncv, int *mpd);
int ks_eps_get_eigenvalue(ks_eps eps, int i, double *eigr, double *eigi);
int ks_eps_get_eigenvector_host(ks_eps eps, int i, double *xr_host);       /* n_local doubles */
/* EPSGetEigenpair epssolve.c:405: conjugate pairs come as (xr,xi) of the first and (xr,-xi) of the second member
   (BV_GetEigenvector bvimpl.h:423-446); any of the four outputs may be NULL */
int ks_eps_get_eigenpair_host(ks_eps eps, int i, double *eigr, double *eigi, double *xr_host, double *xi_host);
int ks_eps_get_eigenpair(ks_eps eps, int i, double *eigr, double *eigi, double *xr_dev, double *xi_dev);   /* the same into device vectors of n_local doubles */
int ks_eps_get_invariant_subspace(ks_eps eps, double *const *v_dev);        /* EPSGetInvariantSubspace epssolve.c:247: nconv device vectors; non-symmetric: before any eigenvector is asked for */
int ks_eps_get_error_estimate(ks_eps eps, int i, double *errest);
int ks_eps_compute_error(ks_eps eps, int i, int type, double *error);      /* EPSComputeError epssolve.c:742 */
/* getters of the settings (EPSGetTolerances, EPSGetWhichEigenpairs, EPSGetTarget, EPSGetConvergenceTest, EPSGetOperators,
   EPSGetProblemType with EPSIsGeneralized / IsHermitian / IsPositive); any output may be NULL */
int ks_eps_get_tolerances(ks_eps eps, double *tol, int *max_it);
int ks_eps_get_which_eigenpairs(ks_eps eps, int *which);
int ks_eps_get_target(ks_eps eps, double *target);
int ks_eps_get_convergence_test(ks_eps eps, int *conv);
int ks_eps_set_extraction(ks_eps eps, int extr);                          /* EPSSetExtraction epsopts.c:968: KS_EPS_RITZ | KS_EPS_HARMONIC (target = EPSSetTarget) */
int ks_eps_get_extraction(ks_eps eps, int *extr);
int ks_eps_set_convergence_test_function(ks_eps eps, ks_eps_converged_fn fn, void *ctx); /* EPSSetConvergenceTestFunction: selects KS_EPS_CONV_USER; NULL restores the relative test */
int ks_eps_set_stopping_test_function(ks_eps eps, ks_eps_stopping_fn fn, void *ctx);    /* EPSSetStoppingTestFunction (ex29.c); NULL = EPSStoppingBasic */
int ks_eps_stopping_basic(ks_eps eps, int its, int max_it, int nconv, int nev, int *reason, void *ctx); /* EPSStoppingBasic epsdefault.c:290 */
int ks_eps_set_arbitrary_selection(ks_eps eps, ks_eps_arbitrary_fn fn, void *ctx);      /* EPSSetArbitrarySelection epsopts.c:600 (symmetric variant; the DS sorts on rr/ri, krylovschur.c:275-279); NULL disables */
int ks_eps_monitor_set(ks_eps eps, ks_eps_monitor_fn fn, void *ctx);                    /* EPSMonitorSet (one slot; NULL cancels): called once per restart with the DS-ordered values, untransformed */
int ks_eps_set_purify(ks_eps eps, int purify);                              /* EPSSetPurify: 0 skips the purification of GHEP eigenvectors (test1_1_ks_nopurify) */
int ks_eps_get_purify(ks_eps eps, int *purify);
int ks_eps_set_track_all(ks_eps eps, int trackall);                          /* EPSSetTrackAll: error estimates of all Ritz pairs at every restart (for monitors) */
int ks_eps_get_krylovschur(ks_eps eps, double *keep, int *lock);             /* EPSKrylovSchurGetRestart / EPSKrylovSchurGetLocking */
int ks_eps_set_balance(ks_eps eps, int bal, int its, double cutoff);        /* EPSSetBalance epsopts.c:1050 (non-symmetric problems; EPSBuildBalance_Krylov epsdefault.c:370); its / cutoff 0 keep 5 / 1e-8 */
int ks_eps_set_balance_matrix(ks_eps eps, const double *D_dev);             /* STSetBalanceMatrix on the solver's ST: EPS_BALANCE_USER with this diagonal (device, n_local, copied) */
int ks_eps_get_balance(ks_eps eps, int *bal, int *its, double *cutoff);
int ks_eps_set_true_residual(ks_eps eps, int trueres);                    /* EPSSetTrueResidual: convergence on ||A x - k B x|| of the Ritz vector (epskrylov.c:256-264) */
int ks_eps_get_true_residual(ks_eps eps, int *trueres);
int ks_eps_get_operators(ks_eps eps, ks_mat *A, ks_mat *B);
int ks_eps_get_problem_type(ks_eps eps, int *type, int *generalized, int *hermitian, int *positive);
int ks_eps_get_bv(ks_eps eps, ks_bv *V);
int ks_eps_get_stats(ks_eps eps, long long *arnoldi_steps, long long *gs_passes, int *restarts);

/* ---- ST: spectral transformation (slepcst.h) ---------------------------------------------------
   STSHIFT and STSINVERT in matrix mode "shell" (A - sigma*B is applied, never assembled); the linear solves are
   GMRES(restart) + Jacobi, the KSP that mode defaults to (stsles.c:51-53), run on the BV kernels of this library.
   An EPS owns one ST (ks_eps_get_st = EPSGetST); ks_st_create makes a stand-alone one. */
int ks_st_create(ks_ctx ctx, ks_st *st);
int ks_st_destroy(ks_st st);
int ks_st_set_type(ks_st st, int type);                                   /* STSetType */
int ks_st_set_shift(ks_st st, double sigma);                              /* STSetShift */
int ks_st_get_shift(ks_st st, double *sigma);
int ks_st_cayley_set_antishift(ks_st st, double nu);                      /* STCayleySetAntishift cayley.c:236 (default: the shift) */
int ks_st_cayley_get_antishift(ks_st st, double *nu);
int ks_st_set_matrices(ks_st st, ks_mat A, ks_mat B /* may be NULL */);   /* STSetMatrices */
/* STSetMatMode (src/sys/classes/st/interface/stfunc.c): how the matrix of the linear solves, P = A - sigma B, exists.
   KS_ST_MATMODE_SHELL (the default here, with the KSP that mode defaults to, stsles.c:51-53): never assembled, applied as two products and
   an axpy (stshellmat.c). KS_ST_MATMODE_COPY (the reference's default mode): assembled once per shift by ks_mat_create_axpy (STMatMAXPY_Private
   stsolve.c:603-631) - one product per application, the Jacobi diagonal is MatGetDiagonal of it; A and B need KS_MAT_KEEP_CSR. The solver
   around it is the same (GMRES / BiCGStab + Jacobi). ST_MATMODE_INPLACE (A overwritten) is not built. The M of cayley / shift stays in the
   term-by-term form in either mode. */
#define KS_ST_MATMODE_COPY  0
#define KS_ST_MATMODE_SHELL 2
int ks_st_set_matmode(ks_st st, int mode);
int ks_st_get_matmode(ks_st st, int *mode);
enum { KS_KSP_GMRES = 0, KS_KSP_BCGS = 1 };
int ks_st_set_ksp_type(ks_st st, int type);                                /* KSPSetType on STGetKSP: GMRES (restarted, the shell mode's default) or BiCGStab; both with Jacobi on the left */
/* KSPGMRESSetCGSRefinementType on STGetKSP, with the BV constants: KS_BV_ORTHOG_REFINE_NEVER (PETSc's default for KSPGMRES: one pass of
   classical Gram-Schmidt, then the norm), _IFNEEDED (PETSc's test is the BV's, eta 0.7071) or _ALWAYS */
int ks_st_set_gmres_cgs_refinement(ks_st st, int refine);
/* PCSetType on KSPGetPC(STGetKSP): KS_PC_JACOBI (the default: PCJACOBI, what the shell matrix mode defaults to, stsles.c:51-53) or
   KS_PC_BJACOBI - PCBJACOBI (the reference's choice with a split preconditioner, stsles.c:46-48; -st_pc_type bjacobi in ex46.c:113,
   test34.c:113) with blocks of block_size consecutive local rows (-pc_bjacobi_local_blocks n_local/block_size) solved exactly
   (-sub_pc_type lu): the dense diagonal blocks of P = A - sigma B are inverted on the host at STSetUp and applied by one kernel.
   2 <= block_size <= 32; the matrices must hold their CSR arrays (KS_MAT_KEEP_CSR); a singular block is KS_ERR_MAT_LU_ZRPVT. */
#define KS_PC_JACOBI  0
#define KS_PC_BJACOBI 1
int ks_st_set_pc(ks_st st, int type, int block_size);
int ks_st_set_ksp(ks_st st, double rtol, int max_it, int restart);        /* KSPSetTolerances / KSPGMRESSetRestart on STGetKSP; 0 keeps */
int ks_st_setup(ks_st st);                                                /* STSetUp */
int ks_st_apply(ks_st st, const double *x_dev, double *y_dev);            /* STApply stsolve.c:44 */
/* STApplyHermitianTranspose (stsolve.c:153-162) for the transformations without a solve: shift with one matrix, y = (A - sigma I)^T x
   (MatMultTranspose of A); the ones with a solve would need it with the transposed matrix: KS_ERR_SUP */
int ks_st_apply_transpose(ks_st st, const double *x_dev, double *y_dev);
int ks_st_backtransform(ks_st st, int n, double *eigr, double *eigi);     /* STBackTransform stsolve.c:563 */
int ks_st_get_ksp_stats(ks_st st, long long *solves, long long *iterations, double *last_rnorm);

/* ---- profiling: HIP-event timing per kernel class, on the context's stream -------------------- */
/* A class is one __global__ kernel template; `variant` is its compile-time column tile KT (0 when
   the kernel has none), so (class, variant) names ONE kernel symbol as rocprofv3 reports it.
   Launches of the speculative Gram-Schmidt slots that gated themselves off on the device are
   re-filed under KS_K_NOOP once the host has read back the pass counts.                          */
enum { KS_K_SPMV = 0, KS_K_DOT, KS_K_GSFIN, KS_K_UPD_FUSED, KS_K_UPD, KS_K_SCALE, KS_K_MULTINPLACE, KS_K_COPY,
       KS_K_MULT, KS_K_BVDOT, KS_K_NORM, KS_K_HALO, KS_K_ALLREDUCE, KS_K_NOOP, KS_K_OTHER,
       KS_K_SPMVDOT /* MatMult fused into the dot sweep that follows it (cache-resident bases, dictionary layout) */, KS_K_COUNT };
#define KS_PROF_VARIANTS 65      /* variant = KT in 0..64 (every count up to 32 is compiled, then 40, 48, 56, 64) */
int ks_prof_enable(ks_ctx ctx, int on);   /* 0: off, 1: every class, else a mask with bit (class+1) set per class to time */
int ks_prof_reset(ks_ctx ctx);
/* launches, summed elapsed ms, summed algorithmic bytes (SURVEY.md 8d formulas) and summed compulsory
   HBM bytes (what the kernel as designed must move) of one class; variant<0 sums over all variants */
int ks_prof_get(ks_ctx ctx, int kclass, int variant, long long *launches, double *ms, double *alg_bytes, double *hbm_bytes);
const char *ks_prof_class_name(int kclass);
/* the reference's log event under which -log_view shows that class's work (bvfunc.c:69-86: BVMatMultVec, BVDotVec, BVMultVec, BVScale, BVMultInPlace,
   BVOrthogonalizeV ...; a fused launch names both events it does the work of, e.g. "BVMultVec+BVDotVec") */
const char *ks_prof_event_name(int kclass);

#ifdef __cplusplus
}
#endif
#endif /* KSGPU_H */
