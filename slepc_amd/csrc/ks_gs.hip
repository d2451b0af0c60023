// Gram-Schmidt orthogonalization and the Krylov expansions (BVMatLanczos / BVMatArnoldi) for gfx950.
//
// Reference semantics: src/sys/classes/bv/interface/bvorthog.c (BVOrthogonalizeCGS1 :91-132,
// BVOrthogonalizeGS :145-217, BVOrthogonalizeColumn :315, BVOrthonormalizeColumn :380-427),
// coefficient helpers include/slepc/private/bvimpl.h:121-141,289-415, loops bvkrylov.c:56-113,165-226.
//
// MI355X design ("ops->gramschmidt" slot, bvimpl.h:53).  One classical Gram-Schmidt pass in the
// reference is gemv-C (h = V^T v, k+1 dots incl. v.v) -> allreduce -> ~8 tiny host-synchronous helper
// calls -> gemv-N (v -= V h); CGS with refinement repeats it.  Here:
//   * the pass bookkeeping (BV_SquareRoot, BV_SquareSum, Pythagoras norm estimate, the refinement test
//     |nrm| < eta*|onrm|, lindep, BV_AddCoefficients, BV_SetValue) runs in a 1-block kernel on device
//     state (KsGsState) - no host round trip inside a Krylov run;
//   * the update sweep of pass p ALSO produces the dot products of pass p+1 (they are row-local:
//     h'_i = sum_r V(r,i) v'(r) with v'(r) just computed from the same V(r,:) held in registers), so a
//     CGS2 step reads V three times instead of four;
//   * the final update applies the 1/nrm scaling of BVOrthonormalizeColumn while storing (nrm is known
//     from the coefficients before the update runs);
//   * later passes are launched unconditionally and gate themselves on the device state, so the
//     reference's data-dependent control flow (1-3 passes, explicit-norm fallback, breakdown) is
//     reproduced exactly; a breakdown turns the rest of the enqueued run into no-ops.
// Multi-rank: the reduce and bookkeeping halves are split around an allreduce of k+1 doubles
// (bvblas.c:255 MPIU_Allreduce) on the same stream.
#include "ks_sweeps.cuh"
#include "ks_oneshot.cuh"
#include <algorithm>

using namespace ksk;

namespace {

struct GsArgs {
  int k;           // number of previous columns: the nc constraints and the regular columns 0..col-1
  int col;         // column being orthogonalized (index of its record and of its buffer column)
  int slot;        // 1..4 position in the launch sequence of this column
  int refine;      // KS_BV_ORTHOG_REFINE_*
  int normalize;   // BVOrthonormalizeColumn: scale by 1/nrm
  int krylov;      // inside BVMatLanczos/Arnoldi: lindep halts the rest of the run
  int ldb;         // leading dimension of the coefficient buffer (nc+m)
  int spec_last;   // this is the last slot of the optimistic program: unfinished business halts the run for the host
  int bmat;        // B-inner product (BVSetMatrix): the dots of every pass are taken with B*v, recomputed by an SpMV before each slot, so an
                   // update never carries the next pass's dots and always writes the vector back
  int wide;        // more than 64 previous columns: dots and updates run over 64-column chunks, the reduced coefficients live in bv->cw, the
                   // update never carries dots and the 1/nrm scaling is a kernel of its own
  int gs1;         // 0: slot of the device-resident program; 1 / 2: ONE pass with the semantics of the ops->gramschmidt slot
                   // (BVOrthogonalizeCGS1), without (1) / with (2) the self dot product in c[k]
  int ncols_in;    // gs1: dots waiting in the partials (k, or k + 1 with the self dot; a chained pass always finds k + 1)
  int pass_idx;    // gs1: 1 for a pass that took its own dots, 2, 3 for passes chained to the one before (the caller's refinement loop)
  int spec_ok;     // gs1: the update may carry the dots of the pass the caller is expected to ask for next
  unsigned long long mail_seq;   // gs1: stamp of this call in the host mailbox
  KsGsMail *mail;  // gs1: pinned host record the bookkeeping's scalars go to (the host polls it while the update sweep runs)
  double eta, deftol;
};

// What thread 0 decides and the other lanes then carry out in parallel (the O(k) global-memory loops of BV_AddCoefficients and of
// the pending-coefficient copy took a third of the kernel when one lane ran them: every iteration a dependent global access).
struct BookPlan {
  int hmode;          // 0: leave H, 1: H = c (first pass: BV_CleanCoefficients + add), 2: H += c
  int pslot;          // row of `pend` that receives c, -1: none
  int set_hk;         // BV_SetValue: H[k] = hk
  int set_rec;        // the column's record is final: recs[col] = rec
  double hk;
  KsStepRec rec;
};
// The bookkeeping functions below are PURE: they read the reduced coefficients c and a private copy of the state and return the new
// state and a plan. Whoever runs them applies the side effects: the 1-block k_gs_finish, or - single rank, the common case - the
// prologue of the update kernel itself, where EVERY workgroup runs the same bookkeeping on the same inputs (same decisions) and
// workgroup 0 alone writes the state, the coefficients and the record back (one launch and one dependent-kernel boundary less per pass).

// Bookkeeping of ONE classical Gram-Schmidt pass as the reference's slot defines it (BVOrthogonalizeCGS1 bvorthog.c:91-132): the
// refinement loop, lindep and BV_CleanCoefficients / BV_SetValue stay with the caller. c[0..k) are the reduced dots against the
// previous columns, c[k] the self dot (a.gs1 == 2). Leaves |v| and the raw estimate |v|^2 - sum c_i^2 in the column's record.
__device__ void gs1_bookkeep(const GsArgs a, const double *c, KsGsState *st, BookPlan *plan)
{
  const int k = a.k;
  double beta = 0.0;
  plan->hmode = 0; plan->pslot = -1; plan->set_hk = 0; plan->set_rec = 0;
  st->do_update = 0; st->err = 0;
  if (a.gs1 == 2) {                                                     // BV_SquareRoot -> BV_SafeSqrt (bvimpl.h:121-141)
    const double vv = c[k];
    if (!(vv > -a.deftol)) { st->err = KS_ERR_USER_INPUT; return; }
    beta = vv < 0.0 ? 0.0 : sqrt(vv);
  }
  double sum = 0.0;
  for (int i = 0; i < k; i++) sum += c[i] * c[i];                      // BV_SquareSum bvimpl.h:347-360
  plan->hmode = 2; plan->pslot = 0;                                    // BV_AddCoefficients bvimpl.h:308-322 (the caller cleaned H)
  st->npend = 1; st->do_update = k > 0 ? 1 : 0; st->fuse_dot = 0; st->scale_now = 0; st->store_now = 1; st->store_prev = 1; st->alpha = 1.0;
  st->pending_scale = 0; st->more_ = 0; st->expl = 0;
  KsStepRec r; r.onrm = beta; r.nrm = beta * beta - sum; r.passes = 1; r.lindep = 0; r.expl = 0; r.col = a.col;
  plan->rec = r; plan->set_rec = 1;
  // Will the caller's refinement loop (BVOrthogonalizeGS bvorthog.c:176-202, same policy and eta: mirrored by the adapter) ask for another
  // pass? Then this update also leaves the dots of that pass behind (they are row-local, ks_gs.hip header). A wrong guess costs nothing
  // but the unused partial sums: the host only chains the next call to them when nothing touched the BV in between.
  // The test below is the caller's own: in the classical Gram-Schmidt loop every pass is called with &onrm (bvorthog.c:183: NULL only for MGS
  // and indefinite inner products, which do not come through this function), CGS1 returns *onorm = beta of THAT pass (bvorthog.c:117), and the
  // loop condition :179 compares the pass's nrm with it - so beta here is not a stand-in for an earlier pass's estimate, it is the operand.
  if (a.spec_ok && k > 0) {
    if (a.refine == KS_BV_ORTHOG_REFINE_IFNEEDED && a.gs1 == 2 && r.nrm > 0.0) {
      const double nrm = sqrt(r.nrm);
      if (a.pass_idx < 3 && nrm != 0.0 && fabs(nrm) < a.eta * fabs(beta)) st->fuse_dot = 1;        // bvorthog.c:179
    } else if (a.refine == KS_BV_ORTHOG_REFINE_ALWAYS && a.gs1 == 1 && a.pass_idx == 1) st->fuse_dot = 1;   // bvorthog.c:198-199
  }
}

// Bookkeeping for one slot.  Thread 0 only.  c[0..k] are the (globally reduced) dots of the current
// vector against columns 0..k-1 and itself.
__device__ void gs_bookkeep(const GsArgs a, const double *c, KsGsState *st, BookPlan *plan)
{
  const int k = a.k;
  int upd = 0, fuse = 0, scal = 0;
  plan->hmode = 0; plan->pslot = -1; plan->set_hk = 0; plan->set_rec = 0;
  bool process = true, finalize = false, after_update = false;
  double nrm = st->nrm, onrm = st->onrm;

  if (a.slot == 1) { st->pass = 0; st->expl = 0; st->pending_scale = 0; st->lindep = 0; st->do_update = 0; st->store_prev = 1; st->npend = 0; }   // nothing applied yet
  else {
    if (st->expl) {
      // explicit norm of the updated vector (BV_NormVecOrColumn, bvorthog.c:126 / :191): c[k] = v'.v'
      nrm = sqrt(c[k] < 0.0 ? 0.0 : c[k]);
      st->expl = 0;
      bool more;
      if (a.refine == KS_BV_ORTHOG_REFINE_IFNEEDED) more = (st->pass < 3 && nrm != 0.0 && fabs(nrm) < a.eta * fabs(onrm));
      else more = false;
      if (!more) { process = false; finalize = true; after_update = true; }
      st->more_ = more ? 1 : 0;
    } else if (!st->more_) process = false;
  }

  if (process) {
    st->pass++; st->passes_total++;
    // BV_SquareRoot -> BV_SafeSqrt (bvimpl.h:121-141)
    const double vv = c[k];
    if (!(vv > -a.deftol)) { st->err = KS_ERR_USER_INPUT; st->active = 0; st->do_update = 0; st->more_ = 0; return; }
    const double beta = vv < 0.0 ? 0.0 : sqrt(vv);
    // BV_SquareSum (bvimpl.h:347-360) and BV_AddCoefficients (bvimpl.h:308-322)
    double sum = 0.0;
    for (int i = 0; i < k; i++) sum += c[i] * c[i];
    plan->hmode = st->pass == 1 ? 1 : 2;                                 // BV_CleanCoefficients + add / BV_AddCoefficients, by the other lanes
    // coefficients the next update applies to the vector AS IT IS IN MEMORY: the passes since it was last written back,
    // kept apart so that the update can subtract them one pass after the other - adding c2 to c1 first would round the
    // correction away exactly when refinement is needed (|c2| ~ eps |c1|)
    if (st->store_prev) st->npend = 0;
    plan->pslot = st->npend; st->npend++;
    if (a.wide) plan->pslot = -1;                                      // chunked updates read the coefficients from bv->cw
    upd = 1;
    if (a.refine == KS_BV_ORTHOG_REFINE_NEVER) {
      // one pass, then explicit norm (bvorthog.c:189-195)
      st->expl = 1; fuse = 1; st->more_ = 0; onrm = beta;
    } else if (a.refine == KS_BV_ORTHOG_REFINE_ALWAYS && st->pass == 1) {
      st->more_ = 1; fuse = 1; onrm = beta;                              // bvorthog.c:197-198
    } else {
      onrm = beta;
      const double n2 = beta * beta - sum;                               // bvorthog.c:124-127
      if (n2 <= 0.0) { st->expl = 1; fuse = 1; st->more_ = 0; }
      else {
        nrm = sqrt(n2);
        bool more;
        if (a.refine == KS_BV_ORTHOG_REFINE_IFNEEDED) more = (st->pass < 3 && nrm != 0.0 && fabs(nrm) < a.eta * fabs(onrm));   // bvorthog.c:179
        else more = false;
        st->more_ = more ? 1 : 0;
        if (more) fuse = 1; else finalize = true;
      }
    }
  }

  if (finalize) {
    int lindep;
    if (a.refine == KS_BV_ORTHOG_REFINE_NEVER) lindep = (nrm == 0.0);                                       // bvorthog.c:193
    else lindep = !(nrm != 0.0 && fabs(nrm) >= a.eta * fabs(onrm));                                          // bvorthog.c:186,201
    plan->set_hk = 1; plan->hk = lindep ? 0.0 : nrm;                                                         // BV_SetValue bvorthog.c:209-214
    const double alpha = (nrm != 1.0 && nrm != 0.0) ? 1.0 / nrm : 1.0;                                       // bvorthog.c:417-419
    st->alpha = alpha; st->lindep = lindep;
    if (a.normalize && alpha != 1.0) { if (after_update || a.wide) st->pending_scale = 1; else scal = 1; }
    KsStepRec r; r.nrm = nrm; r.onrm = onrm; r.passes = st->pass; r.lindep = lindep; r.expl = after_update ? 1 : 0; r.col = a.col;
    plan->rec = r; plan->set_rec = 1;
    if (a.krylov && lindep) st->active = 0;          // bvkrylov.c:92-95: stop the expansion
    st->more_ = 0;
  }
  if (a.bmat || a.wide) fuse = 0;
  st->nrm = nrm; st->onrm = onrm;
  st->do_update = upd; st->fuse_dot = fuse; st->scale_now = scal;
  if (upd) {
    // write the vector back only when it is final, or when an explicit norm follows (its resolution slot only rescales);
    // a fused pass that merely feeds the next pass's dots leaves memory untouched and the next update applies c1+c2
    const int store = (!fuse || st->expl) ? 1 : 0;
    st->store_now = store; st->store_prev = store;
  }
  // Optimistic program: only the slots of the common case (two passes) are enqueued. If this column still needs
  // a pass or an explicit norm after them, stop every later kernel of the run and tell the host which column to
  // complete (the pending update itself still runs: it gates on do_update only).
  if (a.spec_last && (st->more_ || st->expl)) { st->active = 0; st->halt_col = a.col; }
}

constexpr int GF_BLOCK = 1024;    // 16 waves gather the block partials in parallel (256 threads: 10.5 instead of 7.7 us per launch)
// out[i] = sum over the blocks of partials[i][.], i < ncols, with the grid the producing sweep left in the state
__global__ __launch_bounds__(GF_BLOCK) void k_reduce_state(const double *__restrict__ partials, const KsGsState *__restrict__ st, int ncols, double *__restrict__ out)
{
  __shared__ double c_lds[KS_MAX_COLS + 8];
  reduce_partials_to_lds(partials, st->pgrid, ncols, c_lds);
  if ((int)threadIdx.x < ncols) out[threadIdx.x] = c_lds[threadIdx.x];
}

// the same with the one-shot allreduce behind it: the block sums go from LDS into every rank's mailbox, out receives the global sums
__global__ __launch_bounds__(GF_BLOCK) void k_reduce_oneshot(const double *__restrict__ partials, const KsGsState *__restrict__ st, int ncols, double *__restrict__ out, KsOneShotArgs o)
{
  __shared__ double c_lds[KS_MAX_COLS + 8];
  __shared__ unsigned sh[KS_ONESHOT_MAX_RANKS][2 * KS_ONESHOT_MAX_COUNT];
  __shared__ int failed;
  reduce_partials_to_lds(partials, st->pgrid, ncols, c_lds);
  __syncthreads();
  ks_oneshot_sum(c_lds, out, ncols, o, sh, &failed);
}

// the global-memory side of a plan: H(:,col) (entries nc+i, bvbasic.c:784-786), the pending coefficients, BV_SetValue; all lanes
__device__ __forceinline__ void apply_plan(const GsArgs &a, const BookPlan &plan, const double *c, double *__restrict__ buffer, double *__restrict__ pend)
{
  const int i = threadIdx.x;
  double *H = buffer + (size_t)a.col * a.ldb;
  if (i < a.k) {
    if (plan.hmode == 1) H[i] = c[i]; else if (plan.hmode == 2) H[i] += c[i];
    if (plan.pslot >= 0) pend[(size_t)plan.pslot * KS_PSTRIDE + i] = c[i];
  }
  if (i == 0 && plan.set_hk) H[a.k] = plan.hk;
}

// ---- wide bases: the same slot program over 64-column chunks ------------------------------------------------------------------
// k + 1 > 64 coefficients do not fit the register-tiled sweeps or the update kernel's prologue. The slot then runs as: dot sweeps over
// 64-column chunks, each reduced into bv->cw (k_reduce_only), [allreduce], this 1-block bookkeeping kernel on the k + 1 reduced
// coefficients (same pure bookkeeping function), gated k_multvec updates chunk by chunk with the coefficients read from bv->cw, and
// the scaling as a kernel of its own - all enqueued, nothing waits for the host.
__global__ __launch_bounds__(1024) void k_gs_finish_wide(const double *__restrict__ cw, GsArgs a, double *buffer, KsGsState *st, KsStepRec *recs)
{
  extern __shared__ double cw_lds[];
  __shared__ int go;
  __shared__ BookPlan plan;
  if (threadIdx.x == 0) {
    go = 1;
    if (!st->active || (a.slot > 1 && !st->expl && !st->more_)) { go = 0; st->do_update = 0; }
  }
  __syncthreads();
  if (!go) return;
  for (int i = threadIdx.x; i <= a.k; i += blockDim.x) cw_lds[i] = cw[i];
  __syncthreads();
  if (threadIdx.x == 0) {
    KsGsState s = *st;
    gs_bookkeep(a, cw_lds, &s, &plan);
    *st = s;
    if (plan.set_rec) recs[a.col] = plan.rec;
  }
  __syncthreads();
  double *H = buffer + (size_t)a.col * a.ldb;
  for (int i = threadIdx.x; i < a.k; i += blockDim.x) { if (plan.hmode == 1) H[i] = cw_lds[i]; else if (plan.hmode == 2) H[i] += cw_lds[i]; }
  if (threadIdx.x == 0 && plan.set_hk) H[a.k] = plan.hk;
}
__global__ void k_gs_clear_pending_scale(KsGsState *st) { st->pending_scale = 0; }

// REDUCE: sum block partials -> c (LDS, and global scratch = buffer column 0).  BOOK: run the bookkeeping.
template <bool REDUCE, bool BOOK>
__global__ __launch_bounds__(GF_BLOCK) void k_gs_finish(const double *__restrict__ partials, GsArgs a, double *buffer, double *pend, KsGsState *st, KsStepRec *recs)
{
  __shared__ double c_lds[KS_MAX_COLS + 8];
  const int ncols = a.k + (a.gs1 == 1 ? 0 : 1);
  // Slot gating is decided by ONE thread and broadcast through LDS: thread 0 rewrites the state later in
  // this kernel, so letting every wave read it would let a late wave take a different branch around the
  // barriers below.
  __shared__ int go;
  if (threadIdx.x == 0) {
    go = 1;
    if (BOOK && !a.gs1 && (!st->active || (a.slot > 1 && !st->expl && !st->more_))) { go = 0; st->do_update = 0; }   // halted run / column already final
  }
  __syncthreads();
  if (!go) return;
  if (REDUCE) {
    // the grid that wrote the partials (not the host's idea of the last launch: sweeps of later columns may have been
    // enqueued, and gated off, since)
    reduce_partials_to_lds(partials, st->pgrid, ncols, c_lds);
    if ((int)threadIdx.x < ncols) buffer[threadIdx.x] = c_lds[threadIdx.x];     // scratch c = buffer column 0
  } else {
    if ((int)threadIdx.x < ncols) c_lds[threadIdx.x] = buffer[threadIdx.x];
    __syncthreads();
  }
  if (BOOK) {
    __shared__ BookPlan plan;
    if (threadIdx.x == 0) {
      KsGsState s = *st;
      if (a.gs1) gs1_bookkeep(a, c_lds, &s, &plan); else gs_bookkeep(a, c_lds, &s, &plan);
      *st = s;
      if (plan.set_rec) recs[a.col] = plan.rec;
    }
    __syncthreads();
    apply_plan(a, plan, c_lds, buffer, pend);
  }
}

// The tile loop of the update sweep. NP > 0 / FUSE >= 0 fix the number of pending passes and the fused flag at compile
// time (coefficients preloaded, pass loop unrolled) for the two shapes every CGS2 step runs: the first pass (one pending
// pass, fused dots, nothing written) and the final pass (two pending passes, scaled store). NP = 0 / FUSE = -1: runtime.
template <int KT, int VEC, int NP, int FUSE, bool PLAIN>
__device__ __forceinline__ void upd_tiles(const double *V, long long ld, int n, int k, double *v, const double *__restrict__ cg, int npend_rt, bool fuse_rt,
                                          bool scal, bool store, double alpha, int rev, double (&acc)[KT + 1])
{
  const long long tile = (long long)SW_BLOCK * VEC;
  const long long ntiles = ((long long)n + tile - 1) / tile;
  const int npend = NP > 0 ? NP : npend_rt;
  const bool fuse = FUSE >= 0 ? (FUSE != 0) : fuse_rt;
  double cc[NP > 0 ? NP : 1][KT];
  if (NP > 0) {
#pragma unroll
    for (int p = 0; p < NP; p++)
#pragma unroll
      for (int i = 0; i < KT; i++) cc[p][i] = (i < k) ? -cg[(size_t)p * KS_PSTRIDE + i] : 0.0;
  }
  for (long long t0 = blockIdx.x; t0 < ntiles; t0 += gridDim.x) {
    const long long t = rev ? ntiles - 1 - t0 : t0;
    const long long r = t * tile + (long long)threadIdx.x * VEC;
    if (VEC == 2 && r + 1 < n) {
      double2 s = *reinterpret_cast<const double2 *>(v + r);
      double2 xv[KT];
#pragma unroll
      for (int i = 0; i < KT; i++) { const int ii = i < k ? i : (k > 0 ? k - 1 : 0); xv[i] = ldbasis2<PLAIN>(V + (long long)ii * ld + r); }
      // pass by pass, exactly as if each pass had stored its result
      if (NP > 0) {
#pragma unroll
        for (int p = 0; p < NP; p++)
#pragma unroll
          for (int i = 0; i < KT; i++) { s.x = fma(cc[p][i], xv[i].x, s.x); s.y = fma(cc[p][i], xv[i].y, s.y); }
      } else {
        for (int p = 0; p < npend; p++) {
          const double *cp = cg + (size_t)p * KS_PSTRIDE;
#pragma unroll
          for (int i = 0; i < KT; i++) { const double c = (i < k) ? -cp[i] : 0.0; s.x = fma(c, xv[i].x, s.x); s.y = fma(c, xv[i].y, s.y); }
        }
      }
      if (scal) { s.x *= alpha; s.y *= alpha; }
      if (store) {
        // written through (sc1): beside the k read streams a write-through store costs 20 us less per sweep than a write-back one on the boxes where the
        // stored column is expensive (profiles/r02_micro_update_write5.txt); the vector is next read by another kernel in any case
        const ks_d2v sv = {s.x, s.y};
        asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(reinterpret_cast<ks_d2v *>(v + r)), "v"(sv) : "memory");
      }
      if (fuse) {
#pragma unroll
        for (int i = 0; i < KT; i++) { acc[i] = fma(xv[i].x, s.x, acc[i]); acc[i] = fma(xv[i].y, s.y, acc[i]); }
        acc[KT] = fma(s.x, s.x, acc[KT]); acc[KT] = fma(s.y, s.y, acc[KT]);
      }
    } else {
      for (int q = 0; q < VEC; q++) {
        const long long rr = r + q;
        if (rr < n) {
          double s = v[rr];
          double xs[KT];
#pragma unroll
          for (int i = 0; i < KT; i++) { const int ii = i < k ? i : (k > 0 ? k - 1 : 0); xs[i] = V[(long long)ii * ld + rr]; }
          for (int p = 0; p < npend; p++) {
            const double *cp = cg + (size_t)p * KS_PSTRIDE;
#pragma unroll
            for (int i = 0; i < KT; i++) s = fma((i < k) ? -cp[i] : 0.0, xs[i], s);
          }
          if (scal) s *= alpha;
          if (store) v[rr] = s;
          if (fuse) {
#pragma unroll
            for (int i = 0; i < KT; i++) acc[i] = fma(xs[i], s, acc[i]);
            acc[KT] = fma(s, s, acc[KT]);
          }
        }
      }
    }
  }
}

// v <- v - V(:,0:k) c   [* alpha if final]   and, when st->fuse_dot, partials <- [V(:,0:k) v]^T v  (k+1 values)
// What the update kernel needs to carry the bookkeeping of its own slot (single rank): the partials of the sweep before it, the
// state to read and the state to write (ping-pong: workgroup 0 writes the new state while the others may still be reading the old
// one), and where the plan's global side goes. fold == nullptr: the bookkeeping ran in k_gs_finish and the state is read in place.
struct FoldArgs {
  GsArgs a;
  const double *partials_in;     // written by the previous sweep (its grid is in st_in->pgrid)
  const double *cred;            // multi-rank: the partials already summed over blocks and ranks (k+1 values), nullptr: reduce partials_in here
  const KsGsState *st_in;
  KsGsState *st_out;
  double *buffer, *pend;
  KsStepRec *recs;
};

// GS1: the launch is one pass of the ops->gramschmidt slot (its own instantiation, so that the device-resident program's kernel carries none of it)
template <int KT, int VEC, bool GS1>
__global__ __launch_bounds__(SW_BLOCK) void k_gs_update(const double *V, long long ld, int n, int k, double *v, const double *__restrict__ cg_global,
                                                        double *__restrict__ partials, const KsGsState *__restrict__ st, int *__restrict__ pgrid, int rev, int plain,
                                                        int folded, FoldArgs fa)
{
  __shared__ double c_lds[KS_MAX_COLS + 8];
  __shared__ double spend[3 * KS_PSTRIDE];           // the pending passes' coefficients as this launch applies them
  __shared__ KsGsState s_sh;
  __shared__ BookPlan plan_sh;
  __shared__ int go_sh;
  bool fuse, scal, store; int npend; double alpha;
  const double *cg = cg_global;
  if (folded) {
    // ---- prologue: every workgroup runs the slot's bookkeeping on the same inputs ----
    const GsArgs &a = fa.a;
    if (threadIdx.x == 0) {
      KsGsState s = *fa.st_in;
      int go = 1;
      if (!GS1 && (!s.active || (a.slot > 1 && !s.expl && !s.more_))) { go = 0; s.do_update = 0; }     // halted run / column already final
      s_sh = s; go_sh = go;
    }
    __syncthreads();
    if (!go_sh) {
      if (blockIdx.x == 0 && threadIdx.x == 0) *fa.st_out = s_sh;                             // the chain of states goes on through a launch that does nothing
      return;
    }
    const int ncols = GS1 ? a.ncols_in : a.k + 1;
    if (fa.cred) { if ((int)threadIdx.x < ncols) c_lds[threadIdx.x] = fa.cred[threadIdx.x]; __syncthreads(); }
    else reduce_partials_to_lds(fa.partials_in, s_sh.pgrid, ncols, c_lds);
    if (threadIdx.x == 0) {
      KsGsState s = s_sh;
      if (GS1) gs1_bookkeep(a, c_lds, &s, &plan_sh); else gs_bookkeep(a, c_lds, &s, &plan_sh);
      if (s.do_update && s.fuse_dot) s.pgrid = gridDim.x;                                     // the grid of the partials this launch is about to write
      s_sh = s;
    }
    __syncthreads();
    // coefficients of the passes this update applies: older ones from memory (written by earlier launches), this slot's from c
    for (int p = 0; p < s_sh.npend; p++)
      if ((int)threadIdx.x < a.k) spend[p * KS_PSTRIDE + threadIdx.x] = (p == plan_sh.pslot) ? c_lds[threadIdx.x] : fa.pend[(size_t)p * KS_PSTRIDE + threadIdx.x];
    if (blockIdx.x == 0) {
      apply_plan(a, plan_sh, c_lds, fa.buffer, fa.pend);
      if ((int)threadIdx.x < ncols && !(a.col == 0 && (int)threadIdx.x <= a.k)) fa.buffer[threadIdx.x] = c_lds[threadIdx.x];   // scratch c = buffer column 0 (column 0's own coefficients live there: apply_plan has just written them)
      if (threadIdx.x == 0) {
        *fa.st_out = s_sh; if (plan_sh.set_rec) fa.recs[a.col] = plan_sh.rec;
        if (GS1 && a.mail) {
          // the scalars the slot returns go straight to pinned host memory: the host has them when this sweep STARTS, not when it ends, and
          // enqueues the caller's next pass behind it. Payload, system-scope fence, then the stamp the host polls for.
          KsGsMail *m = a.mail;
          __hip_atomic_store(&m->onrm, plan_sh.rec.onrm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(&m->nrm, plan_sh.rec.nrm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(&m->fused, s_sh.do_update && s_sh.fuse_dot ? 1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(&m->err, s_sh.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __threadfence_system();
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __hip_atomic_store(&m->seq, a.mail_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
    __syncthreads();
    if (!s_sh.do_update) return;
    fuse = s_sh.fuse_dot != 0; scal = s_sh.scale_now != 0; store = s_sh.store_now != 0; npend = s_sh.npend; alpha = s_sh.alpha;
    cg = spend;
  } else {
    if (!st->do_update) return;
    fuse = st->fuse_dot != 0; scal = st->scale_now != 0; store = st->store_now != 0; npend = st->npend; alpha = st->alpha;
  }

  // ALL k column loads of a tile are issued back to back (k x 1 KiB in flight per wave), then the update; the fused form
  // keeps the row panel V(r,0:k) in registers and also accumulates the next pass's dots (and writes nothing unless an
  // explicit norm follows), the final form applies the pending passes and the 1/nrm scaling while storing.
  double acc[KT + 1];
#pragma unroll
  for (int i = 0; i <= KT; i++) acc[i] = 0.0;
  // Wide row panels (4 KT registers) leave no room for preloaded coefficients: past a width the specialised forms spill (k_gs_update<64,2> ran its final
  // pass at 2.2 TB/s, <56,2> at 3.0) and the generic form, which reads the coefficients from where the prologue put them (LDS) as it uses them, is the
  // faster one. Measured per width (profiles/r02_wide_probe*.txt): final pass specialised up to 48 columns, fused pass up to 40.
  constexpr bool SPEC_FUSED = KT <= 40, SPEC_FINAL = KT <= 48;
  if (GS1) {
    // a pass of the slot applies its own coefficients only and always stores; it carries the next pass's dots or it does not
    if (VEC == 2 && plain && SPEC_FUSED && fuse && npend == 1 && !scal) upd_tiles<KT, VEC, 1, 1, true>(V, ld, n, k, v, cg, npend, fuse, scal, store, alpha, rev, acc);
    else if (VEC == 2 && !plain && SPEC_FUSED && fuse && npend == 1 && !scal) upd_tiles<KT, VEC, 1, 1, false>(V, ld, n, k, v, cg, npend, fuse, scal, store, alpha, rev, acc);
    else if (VEC == 2 && plain && SPEC_FINAL && !fuse && npend == 1) upd_tiles<KT, VEC, 1, 0, true>(V, ld, n, k, v, cg, npend, fuse, scal, store, alpha, rev, acc);
    else if (VEC == 2 && !plain && SPEC_FINAL && !fuse && npend == 1) upd_tiles<KT, VEC, 1, 0, false>(V, ld, n, k, v, cg, npend, fuse, scal, store, alpha, rev, acc);
    else upd_tiles<KT, VEC, 0, -1, false>(V, ld, n, k, v, cg, npend, fuse, scal, store, alpha, rev, acc);
  } else {
  if (VEC == 2 && plain && SPEC_FUSED && fuse && npend == 1 && !scal) upd_tiles<KT, VEC, 1, 1, true>(V, ld, n, k, v, cg, npend, fuse, scal, store, alpha, rev, acc);     // basis resident in the Infinity Cache: plain loads (ks_sweeps.cuh)
  else if (VEC == 2 && plain && SPEC_FINAL && !fuse && npend == 2) upd_tiles<KT, VEC, 2, 0, true>(V, ld, n, k, v, cg, npend, fuse, scal, store, alpha, rev, acc);
  else if (VEC == 2 && !plain && SPEC_FUSED && fuse && npend == 1 && !scal) upd_tiles<KT, VEC, 1, 1, false>(V, ld, n, k, v, cg, npend, fuse, scal, store, alpha, rev, acc);
  else if (VEC == 2 && !plain && SPEC_FINAL && !fuse && npend == 2) upd_tiles<KT, VEC, 2, 0, false>(V, ld, n, k, v, cg, npend, fuse, scal, store, alpha, rev, acc);
  else upd_tiles<KT, VEC, 0, -1, false>(V, ld, n, k, v, cg, npend, fuse, scal, store, alpha, rev, acc);
  }
  if (!fuse) return;
  if (!folded && blockIdx.x == 0 && threadIdx.x == 0) *pgrid = gridDim.x;        // (folded: already part of the state workgroup 0 wrote)
  // block combine: partial index i<k <- acc[i]; index k <- acc[KT] (the self dot)
  __shared__ double red[SW_WAVES][KT + 1];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i <= KT; i++) { const double s = wave_sum(acc[i]); if (lane == 0) red[w][i] = s; }
  __syncthreads();
  if ((int)threadIdx.x <= k) {
    const int src = ((int)threadIdx.x == k) ? KT : (int)threadIdx.x;
    double s = red[0][src];
#pragma unroll
    for (int ww = 1; ww < SW_WAVES; ww++) s += red[ww][src];
    partials[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = s;
  }
}

__global__ void k_scale_if(double *__restrict__ x, int n, const KsGsState *__restrict__ st)
{
  if (!st->pending_scale) return;
  const double alpha = st->alpha;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) x[i] *= alpha;
}

__global__ void k_gs_begin_run(KsGsState *st) { st->active = 1; st->err = 0; st->do_update = 0; st->more_ = 0; st->expl = 0; st->pending_scale = 0; st->halt_col = -1; }
__global__ void k_gs_resume(KsGsState *st) { st->active = 1; st->halt_col = -1; }

bool aligned16(const void *p) { return (((uintptr_t)p) & 15) == 0; }

int sweep_grid(ks_ctx ctx, int n, int vec) { return ks_sweep_grid(ctx, n, vec); }

int launch_finish(ks_bv bv, const GsArgs &a)
{
  ks_ctx ctx = bv->ctx;
  const bool multi = ks_is_multi(ctx);
  KsProfScope ps(ctx, KS_K_GSFIN, 8.0 * bv->last_grid * (a.k + 1));
  ps.tag(a.col, a.slot, a.k, bv->n);
  if (!multi) hipLaunchKernelGGL((k_gs_finish<true, true>), dim3(1), dim3(GF_BLOCK), 0, ctx->stream, bv->partials, a, bv->buffer, bv->pend, bv->gs, bv->recs);
  else {
    hipLaunchKernelGGL((k_gs_finish<true, false>), dim3(1), dim3(GF_BLOCK), 0, ctx->stream, bv->partials, a, bv->buffer, bv->pend, bv->gs, bv->recs);
    KS_CALL(ks_allreduce_sum(ctx, bv->buffer, a.k + (a.gs1 == 1 ? 0 : 1)));
    hipLaunchKernelGGL((k_gs_finish<false, true>), dim3(1), dim3(GF_BLOCK), 0, ctx->stream, bv->partials, a, bv->buffer, bv->pend, bv->gs, bv->recs);
  }
  KS_HIP(hipGetLastError());
  return KS_SUCCESS;
}

// multi-rank first half of a slot: the 1-block reduction of the block partials into buffer[0..k], then the allreduce; the update
// kernel's prologue takes the bookkeeping from there
int launch_reduce_allreduce(ks_bv bv, const GsArgs &a)
{
  ks_ctx ctx = bv->ctx;
  const int ncols = a.gs1 ? a.ncols_in : a.k + 1;
  KsProfScope ps(ctx, KS_K_GSFIN, 8.0 * bv->last_grid * ncols);
  ps.tag(a.col, a.slot, a.k, bv->n);
  KsOneShotArgs o;
  KS_CALL(ks_oneshot_error(ctx));                // before a sequence number is taken: a call that does not send must not consume one
  if (ks_oneshot_next(ctx, ncols, &o)) {         // reduction and exchange in one launch
    hipLaunchKernelGGL(k_reduce_oneshot, dim3(1), dim3(GF_BLOCK), 0, ctx->stream, bv->partials, bv->gs, ncols, bv->cred, o);
    KS_HIP(hipGetLastError());
    return KS_SUCCESS;
  }
  hipLaunchKernelGGL(k_reduce_state, dim3(1), dim3(GF_BLOCK), 0, ctx->stream, bv->partials, bv->gs, ncols, bv->cred);
  KS_HIP(hipGetLastError());
  return ks_allreduce_sum(ctx, bv->cred, ncols);
}

// fold: the slot's GsArgs when the kernel carries its own bookkeeping, nullptr after a k_gs_finish launch
int launch_update(ks_bv bv, int col, double *v, int slot, const GsArgs *fold = nullptr)
{
  ks_ctx ctx = bv->ctx;
  const int k = bv->nc + col;
  const double *V = ks_bv_col(bv, -bv->nc);      // constraints first, then the regular columns
  const bool v2 = (bv->ld % 2 == 0) && aligned16(V) && aligned16(v);
  int grid = 1;
  const int kk = std::max(k, 1);
  // blocks per CU of the update sweep: measured on MI355X at n = 1e7, k = 16..30: 1 block (4 waves, k KiB in flight each)
  // per CU is fastest (fewer concurrent DRAM streams), as long as every block still gets many tiles
  const long long ntl = ((long long)bv->n + 511) / 512;
  const int upd_per_cu = ntl >= 16LL * ctx->num_cu ? 1 : (ntl >= 4LL * ctx->num_cu ? 2 : 0);
  (void)slot;
  const int plain = ks_basis_is_cache_resident((size_t)(bv->nc + bv->m), (size_t)bv->ld);
  const int rev = bv->sweep_dir; bv->sweep_dir ^= 1;   // every sweep over the basis runs opposite to the one before it (dot sweeps included): it starts on
                                                       // the tail the previous one left in the 256 MB Infinity Cache (about a tenth of a 2.4 GB basis)
  KsProfScope ps(ctx, KS_K_UPD_FUSED, 8.0 * bv->n * (k + 2), ks_kt_for(kk));
  ps.tag(col, slot, k, bv->n);
  FoldArgs fa; memset(&fa, 0, sizeof(fa));
  const int folded = fold ? 1 : 0;
  bv->spec.valid = false;                          // whatever dots an earlier pass of the ops->gramschmidt slot left are about to be replaced
  double *pout = bv->partials;                     // where fused dots go
  if (folded) { fa.a = *fold; fa.cred = ks_is_multi(ctx) ? bv->cred : nullptr; fa.partials_in = bv->partials; fa.st_in = bv->gs; fa.st_out = bv->gs_alt; fa.buffer = bv->buffer; fa.pend = bv->pend; fa.recs = bv->recs; pout = bv->partials_alt; }
#define LAUNCH_UPD_G(KT, G)                                                                                                                            \
  do {                                                                                                                                                 \
    if (v2) { grid = ks_sweep_grid_for(ctx, bv->n, 2, (const void *)k_gs_update<KT, 2, G>, upd_per_cu); bv->last_grid = grid;                                                        \
      hipLaunchKernelGGL((k_gs_update<KT, 2, G>), dim3(grid), dim3(SW_BLOCK), 0, ctx->stream, V, (long long)bv->ld, bv->n, k, v, bv->pend, pout, bv->gs, &bv->gs->pgrid, rev, plain, folded, fa); } \
    else { grid = ks_sweep_grid_for(ctx, bv->n, 1, (const void *)k_gs_update<KT, 1, G>, upd_per_cu); bv->last_grid = grid;                                                           \
      hipLaunchKernelGGL((k_gs_update<KT, 1, G>), dim3(grid), dim3(SW_BLOCK), 0, ctx->stream, V, (long long)bv->ld, bv->n, k, v, bv->pend, pout, bv->gs, &bv->gs->pgrid, rev, plain, folded, fa); }   \
  } while (0)
#define LAUNCH_UPD(KT) LAUNCH_UPD_G(KT, false)
#define LAUNCH_UPD1(KT) LAUNCH_UPD_G(KT, true)
  if (fold && fold->gs1) KS_KT_DISPATCH(kk, LAUNCH_UPD1);
  else KS_KT_DISPATCH(kk, LAUNCH_UPD);
#undef LAUNCH_UPD1
#undef LAUNCH_UPD_G
#undef LAUNCH_UPD
  KS_HIP(hipGetLastError());
  if (folded) { std::swap(bv->gs, bv->gs_alt); std::swap(bv->partials, bv->partials_alt); }      // what the next launch reads
  return KS_SUCCESS;
}

// Slot programs of one column (no host sync inside):
//   optimistic:  dot, [finish p, update p] for p = 1..2              (IFNEEDED / ALWAYS: the common CGS2 case)
//   completion:  [finish p, update p] for the remaining passes, the explicit-norm resolution slot and the
//                stand-alone scaling - enqueued by the host only for a column the device flagged (halt_col)
//   NEVER refinement always needs the resolution slot, so its whole program is enqueued at once.
int spec_slots(ks_bv bv) { return bv->orthog_ref == KS_BV_ORTHOG_REFINE_NEVER ? 1 : 2; }
int total_slots(ks_bv bv) { return bv->orthog_ref == KS_BV_ORTHOG_REFINE_IFNEEDED ? 3 : (bv->orthog_ref == KS_BV_ORTHOG_REFINE_ALWAYS ? 2 : 1); }

// h = V(:,-nc:j+1)^T z with z = v, or z = B v for a B-inner product (BVDotColumnInc bvorthog.c:32-47 with l = -nc: nc+j+1 dots
// including (v,z); BV_IPMatMult bvimpl.h:147-158 inside the dotvec slot, svec.c:117-120)
int enqueue_dots(ks_bv bv, int j, int krylov)
{
  const double *z = ks_bv_col(bv, j);
  KS_CALL(ksb_ipmatmult(bv, z, &z));
  return ksk_dot(bv, ks_bv_col(bv, -bv->nc), bv->ld, bv->nc + j + 1, z, krylov != 0);
}

// wide bases: the dots of a pass over 64-column chunks of [V(:,-nc:j), v] (the last column is the vector itself: the self dot),
// each chunk reduced into its part of bv->cw, then summed over the ranks
int enqueue_dots_wide(ks_bv bv, int j, int krylov)
{
  const double *z = ks_bv_col(bv, j);
  KS_CALL(ksb_ipmatmult(bv, z, &z));
  const int total = bv->nc + j + 1;
  for (int c0 = 0; c0 < total; c0 += KS_MAX_COLS) {
    const int nq = std::min(KS_MAX_COLS, total - c0);
    KS_CALL(ksk_dot(bv, ks_bv_col(bv, -bv->nc) + (size_t)c0 * bv->ld, bv->ld, nq, z, krylov != 0));
    KS_CALL(ksk_reduce_partials(bv, nq, bv->cw + c0));
  }
  return ks_allreduce_sum(bv->ctx, bv->cw, total);
}
int launch_finish_wide(ks_bv bv, const GsArgs &a)
{
  ks_ctx ctx = bv->ctx;
  hipLaunchKernelGGL(k_gs_finish_wide, dim3(1), dim3(1024), sizeof(double) * (a.k + 8), ctx->stream, bv->cw, a, bv->buffer, bv->gs, bv->recs);
  KS_HIP(hipGetLastError());
  return KS_SUCCESS;
}
// v -= V(:,-nc:j) c chunk by chunk, each launch gated on the bookkeeping's do_update
int enqueue_update_wide(ks_bv bv, int j)
{
  const int k = bv->nc + j;
  for (int c0 = 0; c0 < k; c0 += KS_MAX_COLS) {
    const int nq = std::min(KS_MAX_COLS, k - c0);
    KS_CALL(ksk_multvec(bv, ks_bv_col(bv, -bv->nc) + (size_t)c0 * bv->ld, bv->ld, nq, -1.0, 1.0, bv->cw + c0, ks_bv_col(bv, j), bv->gs));
  }
  return KS_SUCCESS;
}
int enqueue_scale_if(ks_bv bv, int j, bool clear)
{
  ks_ctx ctx = bv->ctx;
  KsProfScope ps(ctx, KS_K_SCALE, 0.0);
  ps.tag(j, 0, j, bv->n);
  const int grid = std::max(1, std::min((bv->n + 255) / 256, ctx->num_cu * 4));
  hipLaunchKernelGGL(k_scale_if, dim3(grid), dim3(256), 0, ctx->stream, ks_bv_col(bv, j), bv->n, bv->gs);
  if (clear) hipLaunchKernelGGL(k_gs_clear_pending_scale, dim3(1), dim3(1), 0, ctx->stream, bv->gs);
  KS_HIP(hipGetLastError());
  return KS_SUCCESS;
}

int enqueue_gs_slots(ks_bv bv, int j, int normalize, int krylov, int first, int last, bool halt_at_last, bool resolution_and_scale)
{
  ks_ctx ctx = bv->ctx;
  const bool bmat = bv->matrix != nullptr;
  const bool fold = true, multi = ks_is_multi(ctx);
  const bool wide = bv->nc + j + 1 > KS_MAX_COLS;
  GsArgs a; a.gs1 = 0; a.wide = wide ? 1 : 0; a.bmat = bmat ? 1 : 0; a.k = bv->nc + j; a.col = j; a.refine = bv->orthog_ref; a.normalize = normalize; a.krylov = krylov; a.ldb = bv->nc + bv->m; a.eta = bv->orthog_eta; a.deftol = bv->deftol;
  double *v = ks_bv_col(bv, j);
  for (int p = first; p <= last; p++) {
    a.slot = p; a.spec_last = (halt_at_last && p == last) ? 1 : 0;
    if (wide) {
      KS_CALL(enqueue_dots_wide(bv, j, krylov));
      KS_CALL(launch_finish_wide(bv, a));
      KS_CALL(enqueue_update_wide(bv, j));
      if (normalize) KS_CALL(enqueue_scale_if(bv, j, true));
      continue;
    }
    if (bmat) KS_CALL(enqueue_dots(bv, j, krylov));          // every pass takes its dots with B v afresh (a pass that turns out not to be needed gates itself off in the bookkeeping)
    if (multi) KS_CALL(launch_reduce_allreduce(bv, a));      // ranks: block partials -> buffer[0..k], summed over the ranks (bvblas.c:255)
    if (fold) KS_CALL(launch_update(bv, j, v, p, &a));       // the update kernel runs the slot's bookkeeping in its prologue
    else { KS_CALL(launch_finish(bv, a)); KS_CALL(launch_update(bv, j, v, p)); }
  }
  if (resolution_and_scale) {
    a.slot = last + 1; a.spec_last = 0;     // resolves an explicit-norm request of the last update
    if (wide) { KS_CALL(enqueue_dots_wide(bv, j, krylov)); KS_CALL(launch_finish_wide(bv, a)); }
    else {
      if (bmat) KS_CALL(enqueue_dots(bv, j, krylov));
      KS_CALL(launch_finish(bv, a));
    }
    if (normalize) KS_CALL(enqueue_scale_if(bv, j, wide));
  }
  return KS_SUCCESS;
}

// Optimistic program of column j (against columns 0..j-1).
int enqueue_fused_gs(ks_bv bv, int j, int normalize, int krylov, bool dots_done = false)
{
  KS_CHECK(bv->nc + j + 1 <= 8000, KS_ERR_SUP, "device-resident Gram-Schmidt supports at most 8000 columns");
  if (!dots_done && !bv->matrix && bv->nc + j + 1 <= KS_MAX_COLS) KS_CALL(enqueue_dots(bv, j, krylov));      // with a matrix, or more than 64 coefficients, every slot starts with its own dots
  const int ns = spec_slots(bv), nt = total_slots(bv);
  const bool whole = (ns >= nt) && bv->orthog_ref == KS_BV_ORTHOG_REFINE_NEVER;
  if (whole) return enqueue_gs_slots(bv, j, normalize, krylov, 1, nt, false, true);
  return enqueue_gs_slots(bv, j, normalize, krylov, 1, ns, true, false);
}

// Completion program of the column the device flagged: remaining passes + resolution + scaling.
int enqueue_gs_completion(ks_bv bv, int j, int normalize, int krylov)
{
  hipLaunchKernelGGL(k_gs_resume, dim3(1), dim3(1), 0, bv->ctx->stream, bv->gs);
  KS_HIP(hipGetLastError());
  const int ns = spec_slots(bv), nt = total_slots(bv);
  return enqueue_gs_slots(bv, j, normalize, krylov, ns + 1, nt, false, true);
}

int begin_run(ks_bv bv)
{
  hipLaunchKernelGGL(k_gs_begin_run, dim3(1), dim3(1), 0, bv->ctx->stream, bv->gs);
  KS_HIP(hipGetLastError());
  return KS_SUCCESS;
}

struct HostGs { KsGsState st; };

// state, records and coefficient buffer straight into the pinned host area (mapped into the device's address space): one small launch
// instead of three copy-engine transfers, which sat 16 us apart at the end of every Krylov run (profiles/r03_config2_kernel_trace_gaps.txt)
__global__ __launch_bounds__(256) void k_results_to_host(const KsGsState *__restrict__ st, const KsStepRec *__restrict__ recs, int nrec, const double *__restrict__ coef, size_t coef_len,
                                                         char *__restrict__ out, size_t off_rec, size_t off_coef, unsigned long long seq)
{
  const unsigned *a = reinterpret_cast<const unsigned *>(st); unsigned *o = reinterpret_cast<unsigned *>(out);
  for (size_t i = threadIdx.x; i < sizeof(KsGsState) / 4; i += blockDim.x) o[i] = a[i];
  const unsigned *r = reinterpret_cast<const unsigned *>(recs); unsigned *orr = reinterpret_cast<unsigned *>(out + off_rec);
  for (size_t i = threadIdx.x; i < (size_t)nrec * sizeof(KsStepRec) / 4; i += blockDim.x) orr[i] = r[i];
  double *oc = reinterpret_cast<double *>(out + off_coef);
  for (size_t i = threadIdx.x; i < coef_len; i += blockDim.x) oc[i] = coef[i];
  // the stamp goes last: the host polls it (results_wait) and reads the results as soon as it shows, a stream wait returned 15 - 20 us later
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(reinterpret_cast<unsigned long long *>(out + KS_PINNED_STAMP_OFF), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
static_assert(sizeof(KsGsState) % 4 == 0 && sizeof(KsStepRec) % 4 == 0, "word copies");
bool enqueue_results_to_host(ks_bv bv, int j0, size_t nrec, size_t coef_len, size_t off_rec, size_t off_coef)
{
  ks_ctx ctx = bv->ctx;
  if (!ctx->h_pinned_dev) return false;
  if (!ctx->ev_fetch && hipEventCreateWithFlags(&ctx->ev_fetch, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); ctx->ev_fetch = nullptr; return false; }
  hipLaunchKernelGGL(k_results_to_host, dim3(1), dim3(256), 0, ctx->stream, bv->gs, nrec ? bv->recs + j0 : bv->recs, (int)nrec, bv->buffer, coef_len, (char *)ctx->h_pinned_dev, off_rec, off_coef, ++ctx->fetch_seq);
  if (hipGetLastError() != hipSuccess) return false;
  return hipEventRecord(ctx->ev_fetch, ctx->stream) == hipSuccess;      // the wait's fallback: cannot fail to arrive
}
// Wait for the results a k_results_to_host launch wrote: its stamp in the pinned area, the event behind the launch as the fallback (everything a
// finished launch wrote to host memory is visible). The stream itself is not waited for: in-order, everything before that launch has finished.
int results_wait(ks_ctx ctx)
{
  const unsigned long long *stamp = reinterpret_cast<const unsigned long long *>((const char *)ctx->h_pinned + KS_PINNED_STAMP_OFF);
  ctx->nsync++;
  for (unsigned spins = 1;; spins++) {
    if (__atomic_load_n(stamp, __ATOMIC_ACQUIRE) == ctx->fetch_seq) break;
    if ((spins & 255u) == 0) {
      const hipError_t q = hipEventQuery(ctx->ev_fetch);
      if (q == hipSuccess) break;
      if (q != hipErrorNotReady) { ks_set_error("waiting for the results of a run: %s", hipGetErrorString(q)); return KS_ERR_LIB; }
    }
    __builtin_ia32_pause();
  }
  KS_CHECK(__atomic_load_n(stamp, __ATOMIC_ACQUIRE) == ctx->fetch_seq, KS_ERR_LIB, "the results of a run arrived without their stamp (%llu expected)", ctx->fetch_seq);
  ctx->fetch_waited = ctx->fetch_seq;
  return ks_oneshot_error(ctx);                  // a one-shot allreduce / halo exchange that gave up shows at every host wait, this one included (ksgpu.h)
}

// One host wait for everything the host wants to know after an enqueued run: the device state, the records of columns j0..j1 and
// (coef_out) the whole coefficient buffer, all copied into the context's pinned area by copies enqueued back to back. Between a
// Lanczos run and the restart this wait, the host's projected solve and the upload of Q are all the GPU idles for; copies into
// pageable memory, each with a wait of its own, made that 135 us per restart (config 2: 10 % of the time).
int fetch_state(ks_bv bv, KsGsState *st, KsStepRec *recs, int j0, int j1, double *coef_out = nullptr, size_t coef_len = 0)
{
  ks_ctx ctx = bv->ctx;
  const size_t nrec = (recs && j1 >= j0) ? (size_t)(j1 - j0 + 1) : 0;
  const size_t off_rec = (sizeof(KsGsState) + 63) / 64 * 64, off_coef = (off_rec + nrec * sizeof(KsStepRec) + 63) / 64 * 64;
  const size_t need = off_coef + coef_len * sizeof(double);
  char *pin = (char *)ctx->h_pinned;
  if (need <= KS_PINNED_STAMP_OFF) {
    if (enqueue_results_to_host(bv, j0, nrec, coef_len, off_rec, off_coef)) KS_CALL(results_wait(ctx));
    else {
      KS_HIP(hipMemcpyAsync(pin, bv->gs, sizeof(KsGsState), hipMemcpyDeviceToHost, ctx->stream));
      if (nrec) KS_HIP(hipMemcpyAsync(pin + off_rec, bv->recs + j0, sizeof(KsStepRec) * nrec, hipMemcpyDeviceToHost, ctx->stream));
      if (coef_len) KS_HIP(hipMemcpyAsync(pin + off_coef, bv->buffer, sizeof(double) * coef_len, hipMemcpyDeviceToHost, ctx->stream));
      KS_HIP(ks_sync(ctx));
    }
    memcpy(st, pin, sizeof(KsGsState));
    if (nrec) memcpy(recs, pin + off_rec, sizeof(KsStepRec) * nrec);
    if (coef_len) memcpy(coef_out, pin + off_coef, sizeof(double) * coef_len);
  } else {                                            // a basis too wide for the pinned area: pageable destinations
    KS_HIP(hipMemcpyAsync(st, bv->gs, sizeof(KsGsState), hipMemcpyDeviceToHost, ctx->stream));
    if (nrec) KS_HIP(hipMemcpyAsync(recs, bv->recs + j0, sizeof(KsStepRec) * nrec, hipMemcpyDeviceToHost, ctx->stream));
    if (coef_len) KS_HIP(hipMemcpyAsync(coef_out, bv->buffer, sizeof(double) * coef_len, hipMemcpyDeviceToHost, ctx->stream));
    KS_HIP(ks_sync(ctx));
  }
  if (st->err) KS_FAIL(st->err, "Invalid inner product (BV_SafeSqrt): negative v^H v");
  return KS_SUCCESS;
}

// The same in two halves, for a caller that enqueues more work between them: _begin puts the three copies into the pinned area and marks their end
// with an event, _end waits for that event only - the work enqueued after _begin keeps the device busy while the host reads the results. Returns
// false (nothing enqueued) when the results do not fit the pinned area: the caller then uses fetch_state.
bool fetch_state_begin(ks_bv bv, int j0, int j1, size_t coef_len, int *rc)
{
  ks_ctx ctx = bv->ctx;
  *rc = KS_SUCCESS;
  const size_t nrec = (j1 >= j0) ? (size_t)(j1 - j0 + 1) : 0;
  const size_t off_rec = (sizeof(KsGsState) + 63) / 64 * 64, off_coef = (off_rec + nrec * sizeof(KsStepRec) + 63) / 64 * 64;
  if (off_coef + coef_len * sizeof(double) > KS_PINNED_STAMP_OFF) return false;
  char *pin = (char *)ctx->h_pinned;
  auto chk = [&](hipError_t e) { if (e != hipSuccess && *rc == KS_SUCCESS) { ks_set_error("fetch_state_begin: %s", hipGetErrorString(e)); *rc = KS_ERR_LIB; } };
  ctx->fetch_by_kernel = enqueue_results_to_host(bv, j0, nrec, coef_len, off_rec, off_coef);      // (records the event behind its launch itself)
  if (!ctx->fetch_by_kernel) {
    if (!ctx->ev_fetch) chk(hipEventCreateWithFlags(&ctx->ev_fetch, hipEventDisableTiming));
    chk(hipMemcpyAsync(pin, bv->gs, sizeof(KsGsState), hipMemcpyDeviceToHost, ctx->stream));
    if (nrec) chk(hipMemcpyAsync(pin + off_rec, bv->recs + j0, sizeof(KsStepRec) * nrec, hipMemcpyDeviceToHost, ctx->stream));
    if (coef_len) chk(hipMemcpyAsync(pin + off_coef, bv->buffer, sizeof(double) * coef_len, hipMemcpyDeviceToHost, ctx->stream));
    if (*rc == KS_SUCCESS) chk(hipEventRecord(ctx->ev_fetch, ctx->stream));
  }
  return true;
}
int fetch_state_end(ks_bv bv, KsGsState *st, KsStepRec *recs, int j0, int j1, double *coef_out, size_t coef_len)
{
  ks_ctx ctx = bv->ctx;
  const size_t nrec = (recs && j1 >= j0) ? (size_t)(j1 - j0 + 1) : 0;
  const size_t off_rec = (sizeof(KsGsState) + 63) / 64 * 64, off_coef = (off_rec + nrec * sizeof(KsStepRec) + 63) / 64 * 64;
  const char *pin = (const char *)ctx->h_pinned;
  if (ctx->fetch_by_kernel) KS_CALL(results_wait(ctx));
  else {
    ctx->nsync++;
    KS_HIP(hipEventSynchronize(ctx->ev_fetch));
    KS_CALL(ks_oneshot_error(ctx));                // a one-shot allreduce that gave up shows at every host wait, this one included (ksgpu.h)
  }
  memcpy(st, pin, sizeof(KsGsState));
  if (nrec) memcpy(recs, pin + off_rec, sizeof(KsStepRec) * nrec);
  if (coef_len) memcpy(coef_out, pin + off_coef, sizeof(double) * coef_len);
  if (st->err) KS_FAIL(st->err, "Invalid inner product (BV_SafeSqrt): negative v^H v");
  return KS_SUCCESS;
}

// ---- host-driven generic Gram-Schmidt (MGS, vector argument, `which` selection) -----------------
// Literal restatement of bvorthog.c on top of the primitive ops; every step synchronises, as the
// reference's GPU backend does.  h/c are host arrays of nc+m entries when v is given, else the buffer.
struct GenericGs {
  ks_bv bv; std::vector<double> h, c; bool use_vec;
};

int generic_norm(ks_bv bv, int j, double *v, double *nrm)     // BV_NormVecOrColumn bvorthog.c:20-26
{
  if (!v) return ks_bv_normcolumn(bv, j, KS_NORM_2, nrm);
  if (bv->matrix) return ksb_norm_b(bv, v, nrm);                 // BVNormVec with a matrix (bvglobal.c:556-560)
  // VecNorm of an arbitrary device vector: dot with itself
  KS_CALL(ksk_dot(bv, v, bv->ld, 1, v, false));
  KS_CALL(ksk_reduce_partials(bv, 1, bv->coef));
  KS_CALL(ks_allreduce_sum(bv->ctx, bv->coef, 1));
  double s = 0.0;
  KS_HIP(hipMemcpyAsync(&s, bv->coef, sizeof(double), hipMemcpyDeviceToHost, bv->ctx->stream));
  KS_HIP(ks_sync(bv->ctx));
  *nrm = sqrt(s);
  return KS_SUCCESS;
}

int generic_mgs1(ks_bv bv, int j, double *v, const int *which, double *hh, double *cc, double *onrm, double *nrm)   // bvorthog.c:52-85
{
  double *w = v ? v : ks_bv_col(bv, j);
  if (onrm) KS_CALL(generic_norm(bv, j, v, onrm));
  for (int i = -bv->nc; i < j; i++) {
    if (which && i >= 0 && !which[i]) continue;
    double dot = 0.0;
    // VecDot(z, vi) with the global reduction; z = B*w when a matrix is set (bvorthog.c:68-72)
    const double *z = w;
    KS_CALL(ksb_ipmatmult(bv, w, &z));
    KS_CALL(ksk_dot(bv, ks_bv_col(bv, i), bv->ld, 1, z, false));
    KS_CALL(ksk_reduce_partials(bv, 1, bv->coef));
    KS_CALL(ks_allreduce_sum(bv->ctx, bv->coef, 1));
    KS_HIP(hipMemcpyAsync(&dot, bv->coef, sizeof(double), hipMemcpyDeviceToHost, bv->ctx->stream));
    KS_HIP(ks_sync(bv->ctx));
    cc[bv->nc + i] = dot;                                                   // BV_SetValue(bv,i,0,c,dot)
    // VecAXPY(w,-dot,vi)
    double mdot = dot;
    KS_HIP(hipMemcpyAsync(bv->coef + 8, &mdot, sizeof(double), hipMemcpyHostToDevice, bv->ctx->stream));
    KS_HIP(ks_sync(bv->ctx));
    KS_CALL(ksk_multvec(bv, ks_bv_col(bv, i), bv->ld, 1, -1.0, 1.0, bv->coef + 8, w));
  }
  if (nrm) KS_CALL(generic_norm(bv, j, v, nrm));
  for (int i = 0; i < bv->nc + j; i++) hh[i] += cc[i];                       // BV_AddCoefficients
  return KS_SUCCESS;
}

int generic_cgs1(ks_bv bv, int j, double *v, double *hh, double *cc, double *onorm, double *norm)   // bvorthog.c:91-132
{
  double beta = 0.0;
  const int ksave = bv->k;
  bv->k = j;
  int rc = KS_SUCCESS;
  do {
    if (onorm || norm) {
      if (!v) {
        bv->k = j + 1;                                                       // BVDotColumnInc
        if ((rc = ks_bv_dotvec(bv, ks_bv_col(bv, j), cc))) break;
        bv->k = j;
        const double vv = cc[bv->nc + j];
        if (!(vv > -bv->deftol)) { ks_set_error("Invalid inner product: %g", vv); rc = KS_ERR_USER_INPUT; break; }
        beta = vv < 0.0 ? 0.0 : sqrt(vv);
      } else {
        if ((rc = ks_bv_dotvec(bv, v, cc))) break;
        if ((rc = generic_norm(bv, j, v, &beta))) break;
      }
    } else {
      if ((rc = ks_bv_dotvec(bv, v ? v : ks_bv_col(bv, j), cc))) break;
    }
    if ((rc = ks_bv_multvec(bv, -1.0, 1.0, v ? v : ks_bv_col(bv, j), cc))) break;
    if (onorm) *onorm = beta;
    if (norm) {
      double sum = 0.0;
      for (int i = 0; i < bv->nc + j; i++) sum += cc[i] * cc[i];
      *norm = beta * beta - sum;
      if (*norm <= 0.0) { if ((rc = generic_norm(bv, j, v, norm))) break; }
      else *norm = sqrt(*norm);
    }
    for (int i = 0; i < bv->nc + j; i++) hh[i] += cc[i];
  } while (0);
  bv->k = ksave;
  return rc;
}

// BVOrthogonalizeGS bvorthog.c:145-217 (host-driven). hh: destination coefficients (nc+k+1 entries).
int generic_gs(ks_bv bv, int j, double *v, const int *which, double *hh, double *norm, int *lindep, int *passes)
{
  const int k = v ? bv->k : j;
  const bool mgs = bv->orthog_type == KS_BV_ORTHOG_MGS;
  std::vector<double> cc(bv->nc + bv->m + 1, 0.0);
  double onrm = 0.0, nrm = 0.0;
  const bool dolindep = lindep != nullptr;
  int np = 0;
  for (int i = 0; i < bv->nc + k; i++) hh[i] = 0.0;                          // BV_CleanCoefficients
  auto gs1 = [&](double *on, double *nr) -> int { np++; return mgs ? generic_mgs1(bv, k, v, which, hh, cc.data(), on, nr) : generic_cgs1(bv, k, v, hh, cc.data(), on, nr); };
  switch (bv->orthog_ref) {
    case KS_BV_ORTHOG_REFINE_IFNEEDED: {
      KS_CALL(gs1(&onrm, &nrm));
      int l = 1;
      while (l < 3 && nrm != 0.0 && fabs(nrm) < bv->orthog_eta * fabs(onrm)) {
        l++;
        if (mgs) onrm = nrm;
        KS_CALL(gs1(mgs ? nullptr : &onrm, &nrm));
      }
      if (dolindep) *lindep = !(nrm != 0.0 && fabs(nrm) >= bv->orthog_eta * fabs(onrm));
    } break;
    case KS_BV_ORTHOG_REFINE_NEVER:
      KS_CALL(gs1(nullptr, nullptr));
      if (norm || dolindep) KS_CALL(generic_norm(bv, k, v, &nrm));
      if (dolindep) *lindep = (nrm == 0.0);
      break;
    case KS_BV_ORTHOG_REFINE_ALWAYS:
      KS_CALL(gs1(nullptr, nullptr));
      KS_CALL(gs1(dolindep ? &onrm : nullptr, (norm || dolindep) ? &nrm : nullptr));
      if (dolindep) *lindep = !(nrm != 0.0 && fabs(nrm) >= bv->orthog_eta * fabs(onrm));
      break;
    default: KS_FAIL(KS_ERR_ARG_WRONG, "unknown refinement");
  }
  if (norm) { *norm = nrm; if (!v) hh[bv->nc + k] = (dolindep && *lindep) ? 0.0 : nrm; }
  if (passes) *passes = np;
  return KS_SUCCESS;
}

// write host coefficients hh (nc+j+1 entries incl. the norm slot) into buffer column j
int store_buffer_column(ks_bv bv, int j, const double *hh, int len)
{
  KS_HIP(hipMemcpyAsync(bv->buffer + (size_t)j * (bv->nc + bv->m), hh, sizeof(double) * len, hipMemcpyHostToDevice, bv->ctx->stream));
  KS_HIP(ks_sync(bv->ctx));
  return KS_SUCCESS;
}

bool use_fused(ks_bv bv) { return bv->orthog_type == KS_BV_ORTHOG_CGS && bv->nc + bv->m <= 8000 && bv->fused_gs; }

// The two halves of the fused orthogonalisation of column j: the enqueue (no host wait) and the collection of its results (one host wait; a
// column the device flagged is completed here, *late_completion says so: whatever was enqueued behind the first half saw an unfinished column).
// early_copies: the result copies are enqueued right behind the program (with the coefficient buffer when want_coefs), so that work the caller enqueues
// next does not stand between the program and its results; gs_collect_column then only waits for those copies
int gs_enqueue_column(ks_bv bv, int j, int normalize, bool early_copies = false, bool want_coefs = false)
{
  KS_CALL(begin_run(bv));
  KS_CALL(enqueue_fused_gs(bv, j, normalize, 0));
  bv->fetch_pending = false;
  if (early_copies) {
    const size_t ldb = (size_t)(bv->nc + bv->m), want = (want_coefs && j > bv->l) ? (size_t)(j + 1) * ldb : 0;
    int rc = KS_SUCCESS;
    if (fetch_state_begin(bv, j, j, want, &rc)) { bv->fetch_pending = true; bv->fetch_coefs = want; }
    KS_CALL(rc);
  }
  return KS_SUCCESS;
}
int gs_collect_column(ks_bv bv, int j, int normalize, double *H, double *norm, int *lindep, int *late_completion)
{
  ks_ctx ctx = bv->ctx;
  KsGsState st; KsStepRec rec;
  // the coefficients the caller asks for travel with the state (one host wait): the buffer up to and including column j
  const size_t ldb = (size_t)(bv->nc + bv->m), want = (H && j > bv->l) ? (size_t)(j + 1) * ldb : 0;
  bool batched = want && want * sizeof(double) + 4096 <= KS_PINNED_D2H_BYTES;
  std::vector<double> cb(batched ? want : 0);
  if (bv->fetch_pending) {
    bv->fetch_pending = false;
    batched = want && bv->fetch_coefs == want;
    cb.resize(batched ? want : 0);
    KS_CALL(fetch_state_end(bv, &st, &rec, j, j, batched ? cb.data() : nullptr, batched ? want : 0));
  } else KS_CALL(fetch_state(bv, &st, &rec, j, j, batched ? cb.data() : nullptr, batched ? want : 0));
  bool fresh = batched;
  if (late_completion) *late_completion = 0;
  if (st.halt_col == j) {                       // rare: third pass and/or explicit norm needed
    if (ctx->prof_on) { KsStepRec tmp = rec; tmp.passes = st.pass + 1; tmp.expl = 0; ks_prof_resolve_gs(ctx, &tmp, j, j); }
    KS_CALL(enqueue_gs_completion(bv, j, normalize, 0));
    KS_CALL(fetch_state(bv, &st, &rec, j, j));
    fresh = false;
    if (late_completion) *late_completion = 1;
  }
  ks_prof_resolve_gs(ctx, &rec, j, j);
  bv->passes_last_host = rec.passes; bv->passes_total_host += rec.passes;
  if (norm) *norm = rec.nrm;
  if (lindep) *lindep = rec.lindep;
  if (H && j > bv->l) {   // BV_StoreCoefficients bvimpl.h:403-415: entries l..j-1
    if (fresh) memcpy(H, cb.data() + (size_t)j * ldb + bv->nc + bv->l, sizeof(double) * (j - bv->l));
    else {
      KS_HIP(hipMemcpyAsync(H, bv->buffer + (size_t)j * ldb + bv->nc + bv->l, sizeof(double) * (j - bv->l), hipMemcpyDeviceToHost, ctx->stream));
      KS_HIP(ks_sync(ctx));
    }
  }
  return KS_SUCCESS;
}

// Orthogonalize column j; fused or generic. Returns norm/lindep on the host (synchronises).
int orthogonalize_column(ks_bv bv, int j, int normalize, double *H, double *norm, int *lindep)
{
  ks_ctx ctx = bv->ctx;
  KS_HIP(hipSetDevice(ctx->device));
  if (use_fused(bv)) {
    KS_CALL(gs_enqueue_column(bv, j, normalize));
    return gs_collect_column(bv, j, normalize, H, norm, lindep, nullptr);
  }
  std::vector<double> hh(bv->nc + bv->m + 1, 0.0);
  double nrm = 0.0; int lin = 0, np = 0;
  const int lsave = bv->l, ksave = bv->k;
  bv->l = -bv->nc;
  int rc = generic_gs(bv, j, nullptr, nullptr, hh.data(), &nrm, &lin, &np);
  bv->l = lsave; bv->k = ksave;
  if (rc) return rc;
  bv->passes_last_host = np; bv->passes_total_host += np;
  KS_CALL(store_buffer_column(bv, j, hh.data(), bv->nc + j + 1));
  if (normalize && nrm != 1.0 && nrm != 0.0) KS_CALL(ksk_scale(ctx, ks_bv_col(bv, j), bv->n, 1.0 / nrm));
  if (norm) *norm = nrm;
  if (lindep) *lindep = lin;
  if (H) for (int i = bv->l; i < j; i++) H[i - bv->l] = hh[bv->nc + i];
  return KS_SUCCESS;
}

} // namespace

// ---- public GS entry points ----------------------------------------------------------------------
extern "C" int ks_bv_orthogonalizecolumn(ks_bv bv, int j, double *H, double *norm, int *lindep)   // bvorthog.c:315-339
{
  KS_CHECK(bv, KS_ERR_ARG_NULL, "BV is NULL");
  KS_CHECK(j >= 0, KS_ERR_ARG_OUTOFRANGE, "Index j must be non-negative");
  KS_CHECK(j < bv->m, KS_ERR_ARG_OUTOFRANGE, "Index j=%d but BV only has %d columns", j, bv->m);
  return orthogonalize_column(bv, j, 0, H, norm, lindep);
}

// BVOrthonormalizeColumn that also hands back the coefficients (what a GMRES over a BV needs for its Hessenberg column): the scaling rides in the
// final update, coefficients, norm and flag come back in one host wait
int ks_bv_orthonormalize_coefs(ks_bv bv, int j, double *H, double *norm, int *lindep)
{
  return orthogonalize_column(bv, j, 1, H, norm, lindep);
}
// the same in two halves, for a caller that has work to enqueue behind the orthogonalisation before it looks at the result (the next operator
// application of a GMRES): available when the fused program runs (classical Gram-Schmidt)
bool ks_bv_orthonormalize_can_split(ks_bv bv) { return use_fused(bv); }
int ks_bv_orthonormalize_enqueue(ks_bv bv, int j) { KS_HIP(hipSetDevice(bv->ctx->device)); return gs_enqueue_column(bv, j, 1, true, true); }
int ks_bv_orthonormalize_collect(ks_bv bv, int j, double *H, double *norm, int *lindep, int *late_completion) { return gs_collect_column(bv, j, 1, H, norm, lindep, late_completion); }

extern "C" int ks_bv_orthonormalizecolumn(ks_bv bv, int j, int replace, double *norm, int *lindep)   // bvorthog.c:380-427
{
  KS_CHECK(bv, KS_ERR_ARG_NULL, "BV is NULL");
  KS_CHECK(j >= 0, KS_ERR_ARG_OUTOFRANGE, "Index j must be non-negative");
  KS_CHECK(j < bv->m, KS_ERR_ARG_OUTOFRANGE, "Index j=%d but BV only has %d columns", j, bv->m);
  double nrm = 0.0; int lin = 0;
  if (!replace) { KS_CALL(orthogonalize_column(bv, j, 1, nullptr, &nrm, &lin)); }
  else {
    // the replacement decision needs nrm/lindep before scaling: orthogonalize, decide, then scale
    KS_CALL(orthogonalize_column(bv, j, 0, nullptr, &nrm, &lin));
    for (int attempt = 0; attempt < 2 && (nrm == 0.0 || lin); attempt++) {
      KS_CALL(ks_bv_set_random_column(bv, j, 0x12345678ULL + 7919ULL * (attempt + 1)));
      KS_CALL(orthogonalize_column(bv, j, 0, nullptr, &nrm, &lin));
    }
    if (nrm != 1.0 && nrm != 0.0) KS_CALL(ksk_scale(bv->ctx, ks_bv_col(bv, j), bv->n, 1.0 / nrm));
  }
  if (norm) *norm = nrm;
  if (lindep) *lindep = lin;
  return KS_SUCCESS;
}

extern "C" int ks_bv_orthogonalizevec(ks_bv bv, double *v_dev, double *H, double *norm, int *lindep)   // bvorthog.c:247-269
{
  KS_CHECK(bv && v_dev, KS_ERR_ARG_NULL, "NULL argument");
  KS_HIP(hipSetDevice(bv->ctx->device));
  std::vector<double> hh(bv->nc + bv->m + 1, 0.0);
  const int lsave = bv->l, ksave = bv->k;
  bv->l = -bv->nc;
  int np = 0;
  int rc = generic_gs(bv, 0, v_dev, nullptr, hh.data(), norm, lindep, &np);
  bv->l = lsave; bv->k = ksave;
  if (rc) return rc;
  bv->passes_last_host = np; bv->passes_total_host += np;
  if (H) for (int i = bv->l; i < bv->k; i++) H[i - bv->l] = hh[bv->nc + i];
  return KS_SUCCESS;
}

extern "C" int ks_bv_orthogonalizesomecolumn(ks_bv bv, int j, const int *which, double *H, double *norm, int *lindep)   // bvorthog.c:432-470
{
  KS_CHECK(bv && which, KS_ERR_ARG_NULL, "NULL argument");
  KS_CHECK(j >= 0, KS_ERR_ARG_OUTOFRANGE, "Index j must be non-negative");
  KS_CHECK(j < bv->m, KS_ERR_ARG_OUTOFRANGE, "Index j=%d but BV only has %d columns", j, bv->m);
  KS_CHECK(bv->orthog_type == KS_BV_ORTHOG_MGS, KS_ERR_SUP, "Operation only available for MGS orthogonalization");
  KS_HIP(hipSetDevice(bv->ctx->device));
  std::vector<double> hh(bv->nc + bv->m + 1, 0.0);
  const int lsave = bv->l, ksave = bv->k;
  bv->l = -bv->nc;
  int np = 0; double nrm = 0.0;
  int rc = generic_gs(bv, j, nullptr, which, hh.data(), &nrm, lindep, &np);
  bv->l = lsave; bv->k = ksave;
  if (rc) return rc;
  KS_CALL(store_buffer_column(bv, j, hh.data(), bv->nc + j + 1));
  if (norm) *norm = nrm;
  if (H) for (int i = bv->l; i < j; i++) H[i - bv->l] = hh[bv->nc + i];
  return KS_SUCCESS;
}


// ---- ops->gramschmidt: ONE pass ------------------------------------------------------------------
// The slot BVOrthogonalizeGS1 dispatches to (bvorthog.c:134, bvimpl.h:53). The caller - BVOrthogonalizeGS, bvorthog.c:145-217 -
// owns the refinement loop: it cleans the coefficients, calls the slot once per pass (onrm / nrm may be NULL: REFINE_NEVER and the
// first REFINE_ALWAYS call), compares |nrm| with eta |onrm|, computes lindep and stores the norm next to the coefficients.
namespace {
// The pass as three dependent launches and a stream wait: dot sweep, 1-block bookkeeping, update. Kept for what the chained form below does not
// cover: column 0 without constraints (nothing to update), a B-inner product (the dots of every pass are taken with B v).
int gs1_plain_column(ks_bv bv, int j, double *onrm, double *nrm)
{
  ks_ctx ctx = bv->ctx;
  const bool need = onrm || nrm;
  const int k = bv->nc + j;
  if (k == 0 && !need) return KS_SUCCESS;
  GsArgs a; memset(&a, 0, sizeof(a));
  a.wide = 0; a.bmat = bv->matrix ? 1 : 0; a.gs1 = need ? 2 : 1; a.k = k; a.col = j; a.slot = 1; a.refine = bv->orthog_ref; a.normalize = 0; a.krylov = 0; a.ldb = bv->nc + bv->m;
  a.spec_last = 0; a.eta = bv->orthog_eta; a.deftol = bv->deftol; a.ncols_in = k + (need ? 1 : 0); a.pass_idx = 1;
  double *v = ks_bv_col(bv, j);
  const double *z = v;
  KS_CALL(ksb_ipmatmult(bv, v, &z));                                                             // B v for a B-inner product
  KS_CALL(ksk_dot(bv, ks_bv_col(bv, -bv->nc), bv->ld, k + (need ? 1 : 0), z, false));          // BVDotColumnInc / BVDotColumn
  KS_CALL(launch_finish(bv, a));
  if (k > 0) KS_CALL(launch_update(bv, j, v, 1));                                              // BVMultColumn(bv,-1,1,j,c)
  KsGsState st; KsStepRec rec;
  KS_CALL(fetch_state(bv, &st, &rec, j, j));
  ks_prof_resolve_gs(ctx, &rec, j, j);
  bv->passes_last_host = 1; bv->passes_total_host += 1;
  if (onrm) *onrm = rec.onrm;
  if (nrm) {
    if (rec.nrm <= 0.0) KS_CALL(ks_bv_normcolumn(bv, j, KS_NORM_2, nrm));                      // BV_NormVecOrColumn bvorthog.c:126
    else *nrm = sqrt(rec.nrm);
  }
  return KS_SUCCESS;
}

int gs_mailbox(ks_ctx ctx)
{
  if (ctx->gs_mail) return KS_SUCCESS;
  KS_HIP(hipHostMalloc((void **)&ctx->gs_mail, 256, hipHostMallocMapped | hipHostMallocCoherent));
  memset(ctx->gs_mail, 0, 256);
  KS_HIP(hipHostGetDevicePointer((void **)&ctx->gs_mail_dev, ctx->gs_mail, 0));
  KS_HIP(hipEventCreateWithFlags(&ctx->ev_mail, hipEventDisableTiming));
  return KS_SUCCESS;
}

// Wait for the stamp `seq` in the mailbox. The stamp is written by the update kernel's prologue, i.e. when that launch STARTS; the event behind
// the launch is the fallback that cannot fail to arrive (everything a finished launch wrote to host memory is visible).
int gs_mail_wait(ks_ctx ctx, unsigned long long seq, KsGsMail *out)
{
  volatile KsGsMail *m = ctx->gs_mail;
  ctx->nsync++; ctx->nmailwait++;
  bool got = false;
  for (unsigned spins = 1; !got; spins++) {
    if (__atomic_load_n(&ctx->gs_mail->seq, __ATOMIC_ACQUIRE) == seq) { got = true; break; }
    if ((spins & 255u) == 0) {
      const hipError_t q = hipEventQuery(ctx->ev_mail);
      if (q == hipSuccess) break;
      if (q != hipErrorNotReady) { ks_set_error("waiting for a Gram-Schmidt pass: %s", hipGetErrorString(q)); return KS_ERR_LIB; }
    }
    __builtin_ia32_pause();
  }
  if (!got && __atomic_load_n(&ctx->gs_mail->seq, __ATOMIC_ACQUIRE) != seq) KS_FAIL(KS_ERR_LIB, "the Gram-Schmidt pass finished without reporting (stamp %llu expected)", seq);
  out->onrm = m->onrm; out->nrm = m->nrm; out->fused = m->fused; out->err = m->err; out->seq = seq;
  return ks_oneshot_error(ctx);
}

// The pass as the caller's refinement loop sees it, at the cost of the device-resident program: ONE launch per pass after the first dot sweep
// and no stream wait. The update kernel runs the pass's bookkeeping in its prologue, hands onrm / nrm to the host through the mailbox as it
// starts, applies and STORES the pass (the slot's contract: the column in memory is v - V c when the call returns... is enqueued) and, when the
// mirrored refinement policy says the caller will come back, also accumulates the dots of the next pass from the row panel it holds. A next
// call on the same column with the caller's state token unchanged (ks_bv_set_state) is chained to those dots: no dot sweep, 3 reads of the
// basis per CGS2 step instead of 4.
int gs1_fused_column(ks_bv bv, int j, double *onrm, double *nrm)
{
  ks_ctx ctx = bv->ctx;
  const bool need = onrm || nrm;
  const int k = bv->nc + j;
  if (k == 0 || bv->matrix) return gs1_plain_column(bv, j, onrm, nrm);
  KS_CALL(gs_mailbox(ctx));
  const bool chained = bv->spec.armed && bv->spec.valid && bv->spec.col == j && bv->spec.token_at == bv->spec.token;
  GsArgs a; memset(&a, 0, sizeof(a));
  a.gs1 = need ? 2 : 1; a.k = k; a.col = j; a.slot = 1; a.refine = bv->orthog_ref; a.ldb = bv->nc + bv->m; a.eta = bv->orthog_eta; a.deftol = bv->deftol;
  a.ncols_in = chained ? k + 1 : k + (need ? 1 : 0);
  a.pass_idx = chained ? bv->spec.pass_idx + 1 : 1;
  a.spec_ok = bv->spec.armed ? 1 : 0;
  a.mail = ctx->gs_mail_dev; a.mail_seq = ++ctx->gs_mail_seq;
  double *v = ks_bv_col(bv, j);
  if (!chained) KS_CALL(ksk_dot(bv, ks_bv_col(bv, -bv->nc), bv->ld, a.ncols_in, v, false));    // BVDotColumnInc / BVDotColumn
  if (ks_is_multi(ctx)) KS_CALL(launch_reduce_allreduce(bv, a));                               // bvblas.c:255
  KS_CALL(launch_update(bv, j, v, 1, &a));                                                     // bookkeeping + BVMultColumn(bv,-1,1,j,c) [+ next dots]
  KS_HIP(hipEventRecord(ctx->ev_mail, ctx->stream));
  KsGsMail r;
  KS_CALL(gs_mail_wait(ctx, a.mail_seq, &r));
  if (chained) bv->spec.chained++; else bv->spec.fresh++;
  if (r.err) KS_FAIL(r.err, "Invalid inner product (BV_SafeSqrt): negative v^H v");
  bv->spec.valid = bv->spec.armed && r.fused; bv->spec.col = j; bv->spec.token_at = bv->spec.token; bv->spec.pass_idx = a.pass_idx;
  if (ctx->prof_on) { KsStepRec rec; rec.nrm = r.nrm; rec.onrm = r.onrm; rec.passes = 1; rec.lindep = 0; rec.expl = 0; rec.col = j; ks_prof_resolve_gs(ctx, &rec, j, j); }
  bv->passes_last_host = 1; bv->passes_total_host += 1;
  if (onrm) *onrm = r.onrm;
  if (nrm) {
    if (r.nrm <= 0.0) KS_CALL(ks_bv_normcolumn(bv, j, KS_NORM_2, nrm));                        // BV_NormVecOrColumn bvorthog.c:126 (its sweep drops the chain)
    else *nrm = sqrt(r.nrm);
  }
  return KS_SUCCESS;
}
} // namespace

extern "C" int ks_bv_gramschmidt_pass(ks_bv bv, int j, double *v_dev, const int *which, double *h, double *c, double *onrm, double *nrm)
{
  KS_CHECK(bv, KS_ERR_ARG_NULL, "BV is NULL");
  KS_CHECK((h == nullptr) == (c == nullptr), KS_ERR_ARG_WRONG, "h and c must both be host arrays or both be NULL (the BV's buffer)");
  KS_CHECK(v_dev || (j >= 0 && j < bv->m), KS_ERR_ARG_OUTOFRANGE, "Index j=%d but BV only has %d columns", j, bv->m);
  KS_CHECK(!v_dev || (j >= 0 && j <= bv->m), KS_ERR_ARG_OUTOFRANGE, "Argument j=%d (number of columns to orthogonalize against) out of range", j);
  KS_CHECK(!v_dev || h, KS_ERR_ARG_WRONG, "orthogonalizing a vector needs host arrays h and c (bv->h, bv->c in BVOrthogonalizeVec)");
  ks_ctx ctx = bv->ctx;
  KS_HIP(hipSetDevice(ctx->device));
  const bool mgs = bv->orthog_type == KS_BV_ORTHOG_MGS;
  if (!v_dev && !h && !mgs && use_fused(bv) && bv->nc + j + 1 <= KS_MAX_COLS) return gs1_fused_column(bv, j, onrm, nrm);
  // host-driven pass on the primitive ops (MGS, a vector argument, a B-inner product, a basis wider than the fused kernels)
  const int len = bv->nc + j;
  const int ldb = bv->nc + bv->m;
  std::vector<double> hh(ldb + 1, 0.0), cc(ldb + 1, 0.0);
  double *hp = h ? h : hh.data(), *cp = c ? c : cc.data();
  if (!h && len > 0) {                                   // coefficients of column j live in the device buffer
    KS_HIP(hipMemcpyAsync(hp, bv->buffer + (size_t)j * ldb, sizeof(double) * len, hipMemcpyDeviceToHost, ctx->stream));
    KS_HIP(ks_sync(ctx));
  }
  const int lsave = bv->l, ksave = bv->k;
  bv->l = -bv->nc;
  if (v_dev) bv->k = j;
  int rc = mgs ? generic_mgs1(bv, j, v_dev, which, hp, cp, onrm, nrm) : generic_cgs1(bv, j, v_dev, hp, cp, onrm, nrm);
  bv->l = lsave; bv->k = ksave;
  if (rc) return rc;
  bv->passes_last_host = 1; bv->passes_total_host += 1;
  if (!h && len > 0) {
    KS_HIP(hipMemcpyAsync(bv->buffer + (size_t)j * ldb, hp, sizeof(double) * len, hipMemcpyHostToDevice, ctx->stream));
    KS_HIP(hipMemcpyAsync(bv->buffer, cp, sizeof(double) * len, hipMemcpyHostToDevice, ctx->stream));      // scratch column "s" (bvbasic.c:757-769)
    KS_HIP(ks_sync(ctx));
  }
  return KS_SUCCESS;
}

extern "C" int ks_bv_set_state(ks_bv bv, uint64_t state)
{
  KS_CHECK(bv, KS_ERR_ARG_NULL, "BV is NULL");
  bv->spec.armed = true; bv->spec.token = state;
  return KS_SUCCESS;
}
extern "C" int ks_bv_gs_chain_stats(ks_bv bv, long long *chained, long long *fresh)
{
  KS_CHECK(bv, KS_ERR_ARG_NULL, "BV is NULL");
  if (chained) *chained = bv->spec.chained;
  if (fresh) *fresh = bv->spec.fresh;
  return KS_SUCCESS;
}

extern "C" int ks_bv_gs_passes(ks_bv bv, long long *total, int *last)
{
  KS_CHECK(bv, KS_ERR_ARG_NULL, "BV is NULL");
  if (total) *total = bv->passes_total_host;
  if (last) *last = bv->passes_last_host;
  return KS_SUCCESS;
}

// ---- Krylov expansions ---------------------------------------------------------------------------
// Common loop of BVMatArnoldi (bvkrylov.c:88-96) and BVMatLanczos (:198-206).
static int krylov_run(ks_bv V, ks_mat A, int k, int *m, double *beta, int *breakdown, std::vector<double> &buf)
{
  KS_CHECK(V && A && m, KS_ERR_ARG_NULL, "NULL argument");
  KS_CHECK(k >= 0 && k <= V->m, KS_ERR_ARG_OUTOFRANGE, "Argument k has wrong value %d, should be between 0 and %d", k, V->m);
  KS_CHECK(*m > 0 && *m <= V->m, KS_ERR_ARG_OUTOFRANGE, "Argument m has wrong value %d, should be between 1 and %d", *m, V->m);
  KS_CHECK(*m > k, KS_ERR_ARG_OUTOFRANGE, "Argument m should be at least equal to k+1");
  KS_CHECK(*m < V->m, KS_ERR_ARG_OUTOFRANGE, "BV needs m+1 columns to hold an m-step factorization (m=%d, columns=%d)", *m, V->m);
  KS_CHECK(A->n == V->n, KS_ERR_ARG_INCOMP, "Mismatching local row dimension A %d, V %d", A->n, V->n);
  ks_ctx ctx = V->ctx;
  KS_HIP(hipSetDevice(ctx->device));
  const int m0 = *m;
  int lin = 0;
  double nrm_last = 0.0;
  bool buf_fresh = false;                            // buf already holds the coefficient buffer as the run left it
  if (use_fused(V) && m0 < V->N && (!A->shell_mult || A->shell_nosync)) {      // a matrix-free operator may synchronise: take it one column at a time
    // The whole run is enqueued with the optimistic two-pass program per step; a device-side breakdown, or a
    // column that needs more than the optimistic program, turns the remaining steps into no-ops. In the second
    // case the host completes that one column and re-enqueues the rest of the run.
    int mm = m0, j0 = k;
    while (j0 < m0 && !lin) {
      KS_CALL(begin_run(V));
      for (int j = j0; j < m0; j++) {
        bool dots_done = false;                                                      // the product inside the first dot sweep where that pays (small problems)
        KS_CALL(ks_mat_mult_dot_fused(A, V, ks_bv_col(V, j), j + 1, true, &dots_done));
        if (!dots_done) KS_CALL(ks_mat_mult_internal(A, ks_bv_col(V, j), ks_bv_col(V, j + 1)));    // BVMatMultColumn (not gated: harmless after a halt)
        KS_CALL(enqueue_fused_gs(V, j + 1, 1, 1, dots_done));
      }
      KsGsState st; std::vector<KsStepRec> recs(m0 - j0);
      buf.resize((size_t)V->m * (V->nc + V->m));
      KS_CALL(fetch_state(V, &st, recs.data(), j0 + 1, m0, buf.data(), buf.size()));          // the coefficient buffer travels with the state (VecGetArrayRead(buf) bvkrylov.c:103,215)
      buf_fresh = true;
      const int hc = st.halt_col;                    // column awaiting completion, or -1
      if (ctx->prof_on) {
        // columns after a halt never ran (0 passes, everything gated off); the flagged column ran both optimistic
        // slots in the fused form
        std::vector<KsStepRec> rr(recs); bool halted = false;
        for (int j = j0; j < m0; j++) {
          KsStepRec &r = rr[j - j0];
          if (halted) { r.passes = 0; r.expl = 0; r.lindep = 0; }
          else if (j + 1 == hc) { r.passes = st.pass + 1; r.expl = 0; r.lindep = 0; halted = true; }
          else if (r.lindep) halted = true;
        }
        ks_prof_resolve_gs(ctx, rr.data(), j0 + 1, m0);
      }
      int next = m0;
      for (int j = j0; j < m0; j++) {
        if (j + 1 == hc) {
          // rare path: finish column hc on the device (remaining passes, explicit norm, scaling), then go on
          KsStepRec rec;
          buf_fresh = false;                         // the completion (and whatever is enqueued after it) writes the buffer again
          KS_CALL(enqueue_gs_completion(V, hc, 1, 1));
          KS_CALL(fetch_state(V, &st, &rec, hc, hc));
          ks_prof_resolve_gs(ctx, &rec, hc, hc);
          V->passes_last_host = rec.passes; V->passes_total_host += rec.passes;
          nrm_last = rec.nrm;
          if (rec.lindep) { lin = 1; mm = j + 1; }
          next = j + 1;
          break;
        }
        const KsStepRec &r = recs[j - j0];
        V->passes_last_host = r.passes; V->passes_total_host += r.passes;
        nrm_last = r.nrm;
        if (r.lindep) { lin = 1; mm = j + 1; break; }
      }
      j0 = next;
    }
    *m = mm;
  } else {
    for (int j = k; j < m0; j++) {
      KS_CALL(ks_mat_mult_internal(A, ks_bv_col(V, j), ks_bv_col(V, j + 1)));
      if (j == V->N - 1) {
        // BV_OrthogonalizeColumn_Safe bvimpl.h:452-465: no refinement, norm=0, lindep=TRUE
        const int ref = V->orthog_ref; V->orthog_ref = KS_BV_ORTHOG_REFINE_NEVER;
        int rc = orthogonalize_column(V, j + 1, 0, nullptr, nullptr, nullptr);
        V->orthog_ref = ref;
        if (rc) return rc;
        nrm_last = 0.0; lin = 1;
      } else KS_CALL(orthogonalize_column(V, j + 1, 1, nullptr, &nrm_last, &lin));
      if (lin) { *m = j + 1; break; }
    }
  }
  if (beta) *beta = nrm_last;
  if (breakdown) *breakdown = lin;
  if (!buf_fresh) {
    buf.resize((size_t)V->m * (V->nc + V->m));
    KS_HIP(hipMemcpyAsync(buf.data(), V->buffer, sizeof(double) * buf.size(), hipMemcpyDeviceToHost, ctx->stream));   // VecGetArrayRead(buf) bvkrylov.c:103,215
    KS_HIP(ks_sync(ctx));
  }
  return KS_SUCCESS;
}

extern "C" int ks_bv_matarnoldi(ks_bv V, ks_mat A, double *H, int ldh, int k, int *m, double *beta, int *breakdown)   // bvkrylov.c:56-113
{
  std::vector<double> a;
  if (H && m) KS_CHECK(ldh >= *m, KS_ERR_ARG_SIZ, "Matrix H has %d rows, should have at least %d", ldh, *m);
  KS_CALL(krylov_run(V, A, k, m, beta, breakdown, a));
  if (H) {
    const int nb = V->nc + V->m, mm = *m;
    for (int j = k; j < mm - 1; j++) memcpy(H + (size_t)j * ldh, a.data() + V->nc + (size_t)(j + 1) * nb, sizeof(double) * (j + 2));
    memcpy(H + (size_t)(mm - 1) * ldh, a.data() + V->nc + (size_t)mm * nb, sizeof(double) * mm);
    if (ldh > mm) H[mm + (size_t)(mm - 1) * ldh] = a[V->nc + mm + (size_t)mm * nb];
  }
  return KS_SUCCESS;
}

extern "C" int ks_bv_matlanczos(ks_bv V, ks_mat A, double *T, int ldt, int k, int *m, double *beta, int *breakdown)   // bvkrylov.c:165-226
{
  std::vector<double> a;
  if (T && m) KS_CHECK(ldt >= *m, KS_ERR_ARG_SIZ, "Matrix T has %d rows, should have at least %d", ldt, *m);
  KS_CALL(krylov_run(V, A, k, m, beta, breakdown, a));
  if (T) {
    const int nb = V->nc + V->m;
    double *alpha = T, *betat = T + ldt;
    for (int j = k; j < *m; j++) { alpha[j] = a[V->nc + j + (size_t)(j + 1) * nb]; betat[j] = a[V->nc + j + 1 + (size_t)(j + 1) * nb]; }
  }
  return KS_SUCCESS;
}
