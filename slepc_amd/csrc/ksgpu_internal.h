// Internal declarations of libksgpu (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <string>
#include "../../include/ksgpu.h"

// ---- error plumbing ---------------------------------------------------------------------------
void ks_set_error(const char *fmt, ...);
// resident workgroups per CU of a kernel symbol; a query that fails is COUNTED (ks_runtime_info), reported once on stderr with the HIP runtimes the
// process maps, and answered with `fallback` - never silently (round 3's failures: profiles/r04_hip_runtime_probe.txt)
int ks_occupancy(const void *kernel, int block_threads, size_t lds_bytes, int fallback);
#define KS_FAIL(rc, ...) do { ks_set_error(__VA_ARGS__); return (rc); } while (0)
#define KS_CHECK(cond, rc, ...) do { if (!(cond)) KS_FAIL(rc, __VA_ARGS__); } while (0)
#define KS_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    ks_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); return KS_ERR_LIB; } } while (0)
#define KS_CALL(call) do { int rc_ = (call); if (rc_) return rc_; } while (0)

// ---- profiling ----------------------------------------------------------------------------------
struct KsProfSlot { long long launches = 0; double ms = 0.0; double bytes = 0.0; double hbm = 0.0; };
struct KsProfPending { hipEvent_t e0, e1; int kclass; int variant; double bytes; double hbm; int tag_col; int tag_slot; int tag_k; long long tag_n; bool done; };

// Host mailbox of the ops->gramschmidt slot (one per context, pinned coherent host memory): the update kernel's bookkeeping writes the scalars
// the slot returns here and stamps them with the call's sequence number; the host polls the stamp instead of waiting for the stream.
struct KsGsMail { double onrm, nrm; int fused, err; unsigned long long seq; };

// ---- communicator -------------------------------------------------------------------------------
#define KS_ONESHOT_MAX_RANKS 16
#define KS_ONESHOT_MAX_COUNT 128
#define KS_HALO_MAX_PEERS 16
struct KsComm {
  int rank = 0, size = 1;
  bool force_collectives = false;   // ks_ctx_set_debug(KS_DEBUG_FORCE_MULTI): take the multi-rank code path even with one rank (tests)
  // native RCCL (resolved with dlopen so that a process that already holds librccl reuses it)
  void *rccl_lib = nullptr;
  void *nccl_comm = nullptr;
  long long bcast_calls = 0; double bcast_seconds = 0.0;   // host wall time inside the broadcast of the projected problem (ks_comm_bcast_stats)
  char *ag_dev = nullptr; size_t ag_len = 0;   // device staging of the host allgather (kept: the projected solve is broadcast through it at every restart)
  // active provider (RCCL fills these with its own implementations)
  ks_comm_ops ops = {nullptr, nullptr, nullptr};
  void *user = nullptr;
  // one-shot allreduce over peer-mapped mailboxes (ks_comm_set_allreduce): every rank owns a mailbox in uncached device
  // memory, mapped by all the others (hipIpc between processes, the plain pointer inside one process)
  struct {
    bool enabled = false;
    unsigned seq = 0;                                    // stamp of the call, the same on every rank; travels inside every packet (never 0: what an untouched mailbox holds)
    unsigned par = 0;                                    // slot parity: flips on EVERY call (not derived from the stamp, which skips 0 when it wraps)
    unsigned long long *mine = nullptr;                  // [2 parities][KS_ONESHOT_MAX_RANKS][2 * KS_ONESHOT_MAX_COUNT] packets
    unsigned long long *peer[KS_ONESHOT_MAX_RANKS] = {}; // every rank's mailbox as mapped here (peer[rank] == mine)
    bool opened[KS_ONESHOT_MAX_RANKS] = {};              // mapped with hipIpcOpenMemHandle
    int *err_host = nullptr, *err_dev = nullptr;         // pinned: the sequence number of the first call that timed out, 0 = none
    int *err_local = nullptr;                            // the same word in device memory (what later kernels of this rank look at)
    long long timeout_ticks = 0;                         // wall_clock64 ticks (100 MHz) a rank waits for a packet before it gives up
  } oneshot;
  // peer-mapped halo exchange (ks_halo.hip): the error word a pack / unpack kernel raises when it gives up waiting (pinned, its device
  // address, and the copy in device memory later kernels of this rank look at)
  int *halo_err_host = nullptr, *halo_err_dev = nullptr, *halo_err_local = nullptr;
};

struct ks_ctx_s {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // multi-rank MatMult: the halo (pack + neighbour exchange) runs on its own stream under the diagonal-block product;
  // ev_x orders it after the producer of x on the main stream, ev_halo lets the off-diagonal rows wait for the ghosts
  hipStream_t halo_stream = nullptr; hipEvent_t ev_x = nullptr, ev_halo = nullptr;
  bool halo_overlap = true;
  // test hooks (ks_ctx_set_debug; each makes a test run the path the fast one replaces, or a multi-rank path on one rank)
  struct { bool no_fused_gs = false, no_mfma = false, no_spmv_dot = false, force_multi = false; unsigned oneshot_seq0 = 0; } dbg;
  long long nsync = 0;              // host synchronisations of the context's stream made by the library (ks_ctx_sync_count)
  int num_cu = 256;
  char arch[64] = {0};
  size_t mem_total = 0;
  KsComm comm;
  // profiling
  bool prof_on = false;
  unsigned prof_mask = 0xffffffffu;   // classes that get HIP events when profiling is on
  KsProfSlot prof[KS_K_COUNT][KS_PROF_VARIANTS];
  std::vector<KsProfPending> pending;
  std::vector<hipEvent_t> event_pool;
  // split reductions (BVDotVecBegin/End ...): values queued by the Begin calls wait in `dev`; the first End performs ONE
  // allreduce over all of them (PetscSplitReduction's merged MPI_Allreduce) and the Ends hand the results out in order
  struct KsSplitEntry { int off, cnt, kind; };
  struct { double *dev = nullptr; int cap = 0, used = 0; std::vector<double> host; std::vector<KsSplitEntry> entries; size_t nread = 0; bool reduced = false; } split;
  // pinned host staging: [0, KS_PINNED_D2H_BYTES) results coming back (state, records, coefficient buffer), then two halves of
  // KS_PINNED_H2D_DOUBLES doubles for coefficient uploads (alternating; an event per half says when its last upload has left)
  double *h_pinned = nullptr; size_t h_pinned_len = 0;
  void *h_pinned_dev = nullptr;        // the same area as the device sees it (results written by a kernel instead of the copy engine); nullptr: not mapped
  hipEvent_t ev_h2d[2] = {nullptr, nullptr}; int h2d_next = 0;
  unsigned long long fetch_waited = 0;  // the newest stamp the host has seen: every launch enqueued before that results kernel has finished
  unsigned long long h2d_seq[2] = {0, 0}; bool h2d_busy[2] = {false, false};   // fetch_seq when a kernel was last enqueued to read that pinned half
  bool fetch_by_kernel = false;         // the batch enqueued by fetch_state_begin was written by the kernel (wait on its stamp) rather than by copies (wait on the event)
  unsigned long long fetch_seq = 0;     // stamp of the last batch of results a kernel wrote into the pinned area (its last word): the host polls it instead of waiting on the stream
  hipEvent_t ev_fetch = nullptr;        // marks the end of a batch of result copies that was enqueued ahead of further work (ks_gs.hip: fetch_state_begin / _end)
  KsGsMail *gs_mail = nullptr, *gs_mail_dev = nullptr;   // ops->gramschmidt slot: host mailbox and its device address (allocated on first use)
  unsigned long long gs_mail_seq = 0;
  hipEvent_t ev_mail = nullptr;         // behind the launch that writes the mailbox: what the host falls back on if the stamp does not show up
  long long nmailwait = 0;              // waits on the mailbox (instrumentation; they also count in nsync)
};

// every host wait on the context's stream goes through here, so that tests can assert that a call enqueues without waiting
// (a one-shot allreduce that gave up waiting shows here as a launch time-out, at the first wait after it)
static inline hipError_t ks_sync(ks_ctx ctx)
{
  ctx->nsync++;
  hipError_t e = hipStreamSynchronize(ctx->stream);
  const volatile int *os = ctx->comm.oneshot.err_host, *hh = ctx->comm.halo_err_host;
  return (e == hipSuccess && ((os && *os) || (hh && *hh))) ? hipErrorLaunchTimeOut : e;
}

int ks_prof_begin(ks_ctx ctx, int kclass, int variant, double alg_bytes, double hbm_bytes);   // records start event when profiling
int ks_prof_end(ks_ctx ctx, size_t index);
int ks_prof_flush(ks_ctx ctx);
struct KsProfScope {      // scopes may nest (an allreduce inside a bookkeeping slot): each remembers its own record
  ks_ctx ctx; bool on; size_t index = 0;
  KsProfScope(ks_ctx c, int kclass, double bytes, int variant = 0, double hbm = -1.0) : ctx(c), on(c->prof_on && ((c->prof_mask >> kclass) & 1u)) {
    if (on) { on = ks_prof_begin(c, kclass, variant, bytes, hbm < 0 ? bytes : hbm) == KS_SUCCESS; index = c->pending.empty() ? 0 : c->pending.size() - 1; }
  }
  ~KsProfScope() { if (on) ks_prof_end(ctx, index); }
  // tag the record of a speculative Gram-Schmidt slot so it can be re-filed once pass counts are known
  void tag(int col, int slot, int k, long long n) { if (on && index < ctx->pending.size()) { auto &p = ctx->pending[index]; p.tag_col = col; p.tag_slot = slot; p.tag_k = k; p.tag_n = n; } }
};
struct KsStepRec;
void ks_prof_resolve_gs(ks_ctx ctx, const KsStepRec *recs, int col0, int col1);   // re-file tagged records of columns [col0,col1]

int ks_allreduce_sum(ks_ctx ctx, double *dev_buf, int count);   // no-op when size==1
struct KsOneShotArgs { unsigned long long *peer[KS_ONESHOT_MAX_RANKS]; const unsigned long long *mine; int *err; int *err_local; long long timeout_ticks; unsigned seq, par; int me, size; };
bool ks_oneshot_next(ks_ctx ctx, int count, KsOneShotArgs *o);  // arguments of the next one-shot call, or false when the provider has to do it
int ks_oneshot_error(ks_ctx ctx);                               // KS_ERR_LIB once a one-shot allreduce has timed out (checked wherever the host has just waited)
void ks_oneshot_release(ks_ctx ctx);
int ks_comm_allgather_host(ks_ctx ctx, const void *send, int bytes, void *recv);
int ks_comm_exchange(ks_ctx ctx, int npeers, const int *peers, const void *dev_send, const int *send_off, const int *send_cnt,
                     void *dev_recv, const int *recv_off, const int *recv_cnt, int elem_bytes, hipStream_t stream = nullptr);   // nullptr: the context's stream
int ks_comm_bcast0_host(ks_ctx ctx, void *buf, int bytes);     // every rank leaves with rank 0's bytes (through the provider's host allgather)
int ks_ctx_halo_stream(ks_ctx ctx);                            // creates the halo stream and its events on first use
static inline bool ks_is_multi(ks_ctx ctx) { return ctx->comm.size > 1 || ctx->comm.force_collectives; }

// ---- Mat ----------------------------------------------------------------------------------------
struct ks_mat_s {
  ks_ctx ctx = nullptr;
  int n = 0;             // local rows
  int row_start = 0;     // first global row
  int n_global = 0;
  long long nnz = 0;     // local nonzeros (diag + offdiag blocks)
  // the CSR arrays the matrix was created from (global columns), kept on the host for KS_MAT_KEEP_CSR: what MatDuplicate / MatAXPY need (ks_mat_create_axpy)
  bool keep_csr = false; std::vector<int> k_rowptr, k_col; std::vector<double> k_val;
  ks_mat At = nullptr;                        // MatMultTranspose: the transpose, built on first use from the kept arrays (owned)
  // diagonal block (columns owned by this rank, LOCAL column indices)
  int *d_rowptr = nullptr; int *d_col = nullptr; double *d_val = nullptr; long long nnz_d = 0;
  int lanes_per_row = 8;
  bool force_csr_vector = false;   // KSGPU_SPMV=csrvec
  bool force_csr_regs = false, force_csr_block = false;    // KSGPU_SPMV=csrblock
  // sliced-ELL copy of the diagonal block (slice = 64 rows = one wavefront), chosen at assembly when the
  // padding it needs is small; val/col stored column-major inside a slice: entry j of row 64s+lane at (sp[s]+j)*64+lane
  bool use_sell = false;
  int nslices = 0; int *s_ptr = nullptr; int *s_len = nullptr; int *s_col = nullptr; double *s_val = nullptr; long long s_entries = 0;
  // dictionary ELL (few distinct values and few distinct column offsets, rows of at most 32 entries): 2 bytes per entry
  bool use_dict = false; int dict_w = 0; int dict_nval = 0, dict_noff = 0;
  unsigned short *dc_codes = nullptr; double *dc_val = nullptr; int *dc_off = nullptr;
  // offset-dictionary ELL: any values, few distinct column offsets: 1 byte per entry for the index, values in SELL order
  bool use_odict = false; unsigned char *dc_codes8 = nullptr; double *dc_vals = nullptr;
  bool have_cache = false;                    // diag_cache / norm_inf_cache hold MatGetDiagonal / the infinity norm (CSR arrays released)
  // XCD-sliced copy of the diagonal block for wide-scatter matrices (columns spread over a vector much larger than one
  // XCD's 4 MiB L2): the columns are cut into nslice = 8*P ranges; slice s is a CSR of its own (rows 0..n-1) and is
  // multiplied only by workgroups with blockIdx % 8 == s % 8, i.e. on one XCD, whose L2 then holds that range of x.
  // Every XCD writes a partial y; a second kernel adds the eight partials in fixed order.
  bool use_sliced = false;
  int nslice = 0, slice_cols = 0;
  int *sl_rowptr = nullptr; int *sl_col = nullptr; double *sl_val = nullptr; long long *sl_base = nullptr;   // [nslice][n+1], entries, device offsets [nslice+1]
  double *ypart = nullptr;                    // [8][n]
  double *diag_cache = nullptr; double norm_inf_cache = -1.0;   // kept because the CSR arrays are released after slicing
  // Binned ("propagation blocking") copy of the diagonal block for wide-scatter matrices, the successor of the XCD-sliced one: the product
  // runs in two streaming phases with every random access in LDS. Columns are cut into bn_ns slices of bn_cs, rows into bn_wb wave-bins of
  // bn_wr. Entries are stored twice over: the 16-bit slice-local column in SLICE-major order (slice, wave-bin, row), the value and the 16-bit
  // bin-local row in BIN-major order (wave-bin, slice, row); a (slice, wave-bin) segment is contiguous in both, holds a multiple of 8
  // entries (padding entries with value 0 where needed) and so starts on a 64-byte boundary of the 8-byte streams. Phase 1 (a workgroup per slice, its piece of x in LDS) writes G = x[col] in bin-major
  // order; phase 2 (a wave per wave-bin, its rows of y in LDS) streams G, val and row and adds val * G into its rows.
  bool use_binned = false;
  int bn_ns = 0, bn_cs = 0, bn_wb = 0, bn_wr = 0, bn_nwin = 0;
  long long bn_entries = 0;                   // entries incl. padding
  unsigned short *bn_col16 = nullptr, *bn_row16 = nullptr;
  double *bn_val = nullptr, *bn_g = nullptr;
  int *bn_off1 = nullptr;                     // [ns][wb + 1] start of segment (s, wb) inside slice s's stream
  int *bn_off2t = nullptr;                    // [ns][wb]     start of that segment in bin-major order
  int *bn_wseg = nullptr;                     // [ns][nwin]   segment in which the 1024-entry window of slice s begins
  long long *bn_sbase = nullptr;              // [ns + 1]     start of slice s in bn_col16
  int *bn_off2 = nullptr;                     // [wb][ns]     physical start of segment (wb, s): bin-major order GROUPED by phase-2 workgroup, [wb / 4][ns][wb % 4]
  int *bn_log2 = nullptr;                     // [wb][ns + 1] logical start of that segment inside its wave-bin (piece after piece); [ns] = entries of the wave-bin
  // off-diagonal block (columns owned by other ranks), compressed to ghost indices [0,nghost)
  int *o_rowptr = nullptr; int *o_col = nullptr; double *o_val = nullptr; long long nnz_o = 0;
  int nghost = 0;
  int *o_rows = nullptr; int n_orows = 0;     // rows that have off-diagonal entries (compressed row list)
  double *ghost = nullptr;                    // received halo values (nghost)
  // halo plan (size>1): peers, counts, send index lists
  std::vector<int> peers, send_cnt, recv_cnt, send_off, recv_off;
  int *send_idx = nullptr; int nsend = 0;     // local row indices to pack
  double *send_buf = nullptr;
  // peer-mapped halo (ks_mat_set_halo, ks_halo.hip): this rank's mailbox - [2 parities][nghost] doubles | flag[2][KS_HALO_MAX_PEERS] | ack[KS_HALO_MAX_PEERS],
  // uncached device memory - and the neighbours' mailboxes as mapped here; ridx[i] = this rank's index in peer i's own peer list
  struct { bool enabled = false; unsigned long long seq = 0; char *mine = nullptr; size_t bytes = 0; char *peer_base[KS_HALO_MAX_PEERS] = {}; bool opened[KS_HALO_MAX_PEERS] = {};
           int ridx[KS_HALO_MAX_PEERS] = {}, remote_off[KS_HALO_MAX_PEERS] = {}, remote_nghost[KS_HALO_MAX_PEERS] = {}; unsigned *tickets = nullptr; long long timeout_ticks = 0; } hp;
  // matrix-free operator (MATSHELL with MATOP_MULT): y = shell_mult(user, x); may synchronise the host
  int (*shell_mult)(void *user, const double *x_dev, double *y_dev) = nullptr;
  int (*shell_mult_t)(void *user, const double *x_dev, double *y_dev) = nullptr;      // MATOP_MULT_TRANSPOSE
  bool shell_nosync = false;                  // the callback only enqueues work on the context's stream: a Krylov run may be enqueued ahead through it
  void *shell_user = nullptr;
};
int ks_mat_get_diagonal_internal(ks_mat A, double *d_dev);
int ks_mat_mult_transpose_internal(ks_mat A, const double *x, double *y);
int ks_st_apply_transpose_internal(ks_st st, const double *x, double *y);
void ks_halo_release(ks_mat A);
int ks_halo_peer_exchange(ks_mat A, const double *x, hipStream_t hs);     // pack into the neighbours' mailboxes, unpack this rank's own (ks_halo.hip)

int ks_mat_norm_inf_local(ks_mat A, double *val);           // max row sum of |a_ij| over this rank's rows

// ---- ST: spectral transformation (ks_st.hip) ----------------------------------------------------
struct ks_st_s {
  ks_ctx ctx = nullptr;
  int type = KS_ST_SHIFT;
  double sigma = 0.0; bool sigma_set = false;
  double nu = 0.0; bool nu_set = false;      // STCAYLEY antishift (defaults to sigma, cayley.c:140)
  ks_mat bil = nullptr;                       // STCAYLEY: shell matrix A + nu B, the bilinear form of symmetric problems (cayley.c:70-77)
  ks_mat A = nullptr, B = nullptr;            // borrowed
  double rtol = 1e-8; int max_it = 10000, restart = 30;   // KSP: SLEPC_DEFAULT_TOL (stsles.c:407), PETSc defaults
  int ksp_type = KS_KSP_GMRES;                // KSPGMRES (the default of the shell matrix mode) or KSPBCGS
  int gmres_refine = KS_BV_ORTHOG_REFINE_NEVER;   // KSPGMRESSetCGSRefinementType: PETSc's default is classical Gram-Schmidt without refinement
  ks_bv Kb = nullptr;                         // BiCGStab work vectors (7 columns)
  ks_bv K = nullptr, W = nullptr;             // GMRES basis (restart+1 columns), work vectors (3 columns)
  double *dinv = nullptr;                     // Jacobi: 1/diag(P)
  int pc_type = KS_PC_JACOBI, pc_bs = 0;      // PCSetType on the KSP's PC: point Jacobi, or block Jacobi with blocks of pc_bs consecutive local rows
  double *binv = nullptr;                     // block Jacobi: row i holds the pc_bs coefficients of row i of its block's inverse (n x pc_bs, zero beyond a short last block)
  double *pcwork = nullptr;                   // block Jacobi: the vector the blocks are applied to (n)
  int matmode = KS_ST_MATMODE_SHELL;          // STSetMatMode: how P = A - sigma B exists (shell: applied term by term; copy: assembled, stsolve.c:603-631)
  ks_mat Pmat = nullptr;                      // ST_MATMODE_COPY: the assembled P (owned)
  ks_mat op = nullptr;                        // shell matrix whose MatMult is STApply
  int n = 0; bool ready = false;
  long long solves = 0, its = 0; double last_rnorm = 0.0;
};
bool ks_st_is_plain(ks_st st);                // shift with sigma = 0 on a standard problem: Op = A itself
int ks_st_setup_internal(ks_st st);
int ks_st_apply_internal(ks_st st, const double *x, double *y);
void ks_st_backtransform_internal(ks_st st, int n, double *eigr, double *eigi);
int ksk_lincomb(ks_ctx ctx, long long n, const double *s, double a, const double *u, double b, const double *v, double *out);   // out = s.*(a*u + b*v)   // diagonal of the local diagonal block (MatGetDiagonal)

// ---- BV -----------------------------------------------------------------------------------------
// Device-resident Gram-Schmidt state (one per BV).  Written by the 1-block bookkeeping kernel,
// read by the sweep kernels; lets a whole Krylov run be enqueued with no host round trip.
struct KsGsState {
  int active;         // 0 after breakdown/error: every later kernel of the run is a no-op
  int do_update;      // coefficients c are ready: the next update kernel must apply them
  int fuse_dot;       // that update must also produce the partial dots of the next pass (or v'.v')
  int scale_now;      // that update is the final one: multiply by alpha while writing
  int expl;           // estimated norm^2 <= 0: explicit norm needed after the update (bvorthog.c:126)
  int pending_scale;  // final scaling still to be applied by the stand-alone scale kernel
  int pass;           // passes done for the current column
  int err;            // sticky error (KS_ERR_USER_INPUT: invalid inner product)
  int lindep;         // result for the current column
  int more_;          // another pass follows the pending update (bvorthog.c:179 loop condition)
  int halt_col;       // >=0: column whose orthogonalization needs slots beyond the optimistic program (host completes it)
  int store_now;      // the pending update writes the vector back (final update, or an explicit norm follows); a fused pass that
                      // only feeds the next pass's dots keeps its result in registers (writes cost ~5 read-columns of HBM time)
  int store_prev;     // the previous update of this column stored: the coefficients applied so far are in memory
  int npend;          // passes whose coefficients are not in memory yet: the next update applies pend[0..npend) one after the other
  int pgrid;          // gridDim.x of the sweep that last wrote `partials` (their stride and count): written by that sweep, read by the
                      // reduction, so a completion program enqueued after later columns' (gated-off) sweeps still reduces with the right grid
  double onrm, nrm, alpha;
  long long passes_total;
};
struct KsStepRec { double nrm, onrm; int passes, lindep, expl, col; };

struct ks_bv_s {
  ks_ctx ctx = nullptr;
  int n = 0, N = 0, m = 0, l = 0, k = 0, nc = 0, ld = 0;
  int orthog_type = KS_BV_ORTHOG_CGS, orthog_ref = KS_BV_ORTHOG_REFINE_IFNEEDED, orthog_block = KS_BV_ORTHOG_BLOCK_GS;
  ks_mat matrix = nullptr;   // inner-product matrix B of BVSetMatrix (positive definite), borrowed; nullptr = standard
  double *Bx = nullptr;      // B*x of the vector an inner product is being taken with (BV_IPMatMult bvimpl.h:147-158)
  bool fetch_pending = false; size_t fetch_coefs = 0;     // result copies of the last enqueued column are on their way (gs_enqueue_column with early copies)
  // ops->gramschmidt slot, pass chaining (ks_bv_set_state): the last pass on column `col` left the next pass's dots in `partials`; they are used
  // only if the caller's state token is still `token_at` and no sweep of this BV has rewritten the partials since
  struct { bool armed = false, valid = false; unsigned long long token = 0, token_at = 0; int col = -1, pass_idx = 0; long long chained = 0, fresh = 0; } spec;
  double *pend = nullptr;    // [3][KS_PSTRIDE] coefficients of the passes since the vector was last written back (what the next update applies, pass by pass)
  double orthog_eta = 0.7071;
  bool fused_gs = true;      // the device-resident Gram-Schmidt program may be used (the no_fused_gs test hook is read when the BV is created)
  double deftol = 10 * 2.220446049250313e-16;
  double *array = nullptr;      // m*ld
  double *buffer = nullptr;     // (nc+m)*m ; column 0 = scratch c
  bool own_buffer = true;       // false: adopted from the caller (ks_bv_set_buffer: the device array of the reference's bv->buffer Vec)
  double *buffer_own = nullptr; // the library's own allocation while an adopted one is in use
  double *partials = nullptr;   // [KS_MAX_BLOCKS][KS_PSTRIDE] block partial sums: the CURRENT one of two buffers (an update kernel that carries the
                                // bookkeeping reads the current one in its prologue while its fused dots go to the other; the host swaps the pointers)
  double *partials_alt = nullptr, *partials_base = nullptr;
  double *coef = nullptr;       // device scratch for host-provided q / Q (max(m*m, ...))
  double *hc = nullptr;         // device h,c arrays for orthogonalizevec (2*(nc+m))
  double *cw = nullptr;         // [nc+m+1] wide bases (more than 64 previous columns): the pass's reduced dots, chunk after chunk
  double *cred = nullptr;       // [KS_PSTRIDE] multi-rank: the pass's dots summed over blocks and ranks, read by every workgroup of the update kernel
                                // (not the buffer's scratch column: workgroup 0 writes H(:,0) there while the others may still be reading)
  size_t coef_len = 0;
  KsGsState *gs = nullptr, *gs_alt = nullptr, *gs_base = nullptr;   // current / other device state (same ping-pong), allocation
  KsStepRec *recs = nullptr;    // m records (one per column)
  long long passes_total_host = 0; int passes_last_host = 0;
  int row_start = 0;            // first global row (reproducible random)
  int sweep_dir = 0;            // direction of the next row sweep over the basis: consecutive sweeps alternate (forward / backward), so each starts
                                // on the rows the previous one left in the Infinity Cache
  int last_grid = 1;            // grid size of the sweep LAUNCHED last (profiling byte counts, and reductions that directly follow their sweep); the
                                // Gram-Schmidt bookkeeping reads the grid from KsGsState::pgrid instead
  double *panel = nullptr; size_t panel_len = 0;   // block partials of the MFMA panel dot (grid x 64 x 64 max)
};

constexpr size_t KS_PINNED_D2H_BYTES = 65536;
constexpr size_t KS_PINNED_STAMP_OFF = KS_PINNED_D2H_BYTES - 64;      // the results' stamp (8 bytes); results may use the area below it
constexpr size_t KS_PINNED_H2D_DOUBLES = 4096;
constexpr int KS_MAX_COLS   = 64;     // max columns handled by the register-tiled sweeps (k+1 <= 64)
constexpr int KS_PSTRIDE    = 72;     // doubles per block in the partials array
constexpr int KS_MAX_BLOCKS = 4096;

static inline double *ks_bv_col(ks_bv bv, int j) { return bv->array + (size_t)(bv->nc + j) * bv->ld; }

// kernel launchers (ks_bv_kernels.hip / ks_gs.hip)
int ksk_dot(ks_bv bv, const double *A, int lda, int ncols, const double *y, bool gate);      // partials <- A(:,0:ncols)^T y
int ksk_reduce_partials(ks_bv bv, int ncols, double *out_dev);                                // out[i] = sum_b partials[b][i]
int ksk_multvec(ks_bv bv, const double *A, int lda, int ncols, double alpha, double beta, const double *q_dev, double *y, const KsGsState *gate = nullptr);
int ksk_scale(ks_ctx ctx, double *x, size_t n, double alpha);
int ksk_copy(ks_ctx ctx, const double *src, double *dst, size_t n);

int ks_mat_mult_internal(ks_mat A, const double *x, double *y, const double *rowscale = nullptr);   // rowscale: y = rowscale .* (A x) where the layout can fold it into its last pass (else the caller scales)
bool ks_mat_can_rowscale(ks_mat A);
int ks_mat_mult_dot_fused(ks_mat A, ks_bv bv, const double *x, int jy, bool gate, bool *done);   // y = A x inside the dot sweep of column jy (ks_spmv.hip); *done = false: not applicable                                 // the product can take a row scaling in the same launches
int ks_bv_orthonormalize_coefs(ks_bv bv, int j, double *H, double *norm, int *lindep);
bool ks_bv_orthonormalize_can_split(ks_bv bv);
int ks_bv_orthonormalize_enqueue(ks_bv bv, int j);
int ks_bv_orthonormalize_collect(ks_bv bv, int j, double *H, double *norm, int *lindep, int *late_completion);
int ksb_ipmatmult(ks_bv bv, const double *x, const double **z);   // z = x, or B*x (in bv->Bx) when a matrix is set
int ksb_norm_b(ks_bv bv, const double *x, double *val);            // sqrt(x' B x) with the BV_SafeSqrt check (BVNorm_Private)
int ksb_dot_range(ks_bv X, int xs, int xe, ks_bv Y, int ys, int ye, double *M, int ldm);          // M(ys:ye,xs:xe) = Y(:,ys:ye)^T X(:,xs:xe)
int ksb_mult_range(ks_bv Y, int ys, int ye, double alpha, double beta, ks_bv X, int xs, int xe, const double *Q, int ldq);
// MFMA f64 panel contractions (ks_panel.hip)
int ksp_dot_mfma(ks_bv bv, const double *Y, int ldy, int my, const double *X, int ldx, int nx, int n, double *M_dev);
int ksp_mult_mfma(ks_ctx ctx, int kclass, const double *A, int lda, int n, int kin, const double *Qdev, int qsk, int qsi, int nout,
                  double alpha, double beta, double *C, int ldc);
int ks_sweep_grid(ks_ctx ctx, int n, int vec);
int ks_sweep_grid_for(ks_ctx ctx, int n, int vec, const void *kernel, int force_per_cu);   // resident-blocks grid of one kernel symbol   // blocks of a row sweep (shared by every sweep kernel so partials line up)
