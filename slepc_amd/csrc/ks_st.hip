// Spectral transformation: the operator the Krylov expansion multiplies by (the caller side of MatMult on the path).
//
// Restates STSHIFT and STSINVERT (src/sys/classes/st/impls/shift/shift.c:16-97, sinvert/sinvert.c:16-77) with
// STApply_Generic (src/sys/classes/st/interface/stsolve.c:16-25):  y = P^-1 M x,
//     shift:    nmat=1  M = A - sigma I, P none          nmat=2  M = A - sigma B, P = B
//     sinvert:  nmat=1  M none,          P = A - sigma I nmat=2  M = B,           P = A - sigma B
//     cayley:   nmat=1  M = A + nu I,    P = A - sigma I nmat=2  M = A + nu B,    P = A - sigma B   (cayley/cayley.c:138-165)
// in the reference's ST_MATMODE_SHELL form: A - sigma B is never assembled, it is applied as two SpMVs and an
// axpy (stshellmat.c), and its diagonal is diag(A) - sigma diag(B). The linear solves are the KSP that mode
// defaults to (stsles.c:51-53): GMRES(30) with a Jacobi preconditioner on the left, relative tolerance
// SLEPC_DEFAULT_TOL on the preconditioned residual (stsles.c:407), zero initial guess, failure to converge is an
// error (KSPSetErrorIfNotConverged, stsles.c:58). PETSc's KSP itself is external to the reference; this GMRES runs on
// the same device kernels as the outer iteration: the Krylov basis is a BV, its Gram-Schmidt is the fused CGS of
// ks_gs.hip, the update x += K y is BVMultVec.
#include "ksgpu_internal.h"
#include "ks_csr.h"
#include <algorithm>

namespace {

// out = s .* (a*u + b*v)   (s, v may be NULL: s = 1, v ignored)
__global__ void k_lincomb(long long n, const double *__restrict__ s, double a, const double *__restrict__ u, double b, const double *__restrict__ v, double *__restrict__ out)
{
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    double t = a * u[i];
    if (v) t += b * v[i];
    out[i] = s ? s[i] * t : t;
  }
}
// d = 1 / (a*da + b*db), db NULL = identity; a zero diagonal entry leaves the row unscaled (PCJacobi does the same)
__global__ void k_jacobi_setup(long long n, double a, const double *__restrict__ da, double b, const double *__restrict__ db, double *__restrict__ d)
{
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const double t = (da ? a * da[i] : 0.0) + b * (db ? db[i] : 1.0);
    d[i] = (t != 0.0) ? 1.0 / t : 1.0;
  }
}

// out = M^-1 in for the block-Jacobi M: row i of out is row i of its block's inverse times the block's piece of in (binv: n x bs, row-major)
__global__ void k_bjacobi_apply(long long n, int bs, const double *__restrict__ binv, const double *__restrict__ in, double *__restrict__ out)
{
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long b0 = i / bs * bs;
    const double *row = binv + i * bs;
    double acc = 0.0;
    for (int c = 0; c < bs && b0 + c < n; c++) acc = fma(row[c], in[b0 + c], acc);
    out[i] = acc;
  }
}

} // namespace
int ksk_lincomb(ks_ctx ctx, long long n, const double *s, double a, const double *u, double b, const double *v, double *out)
{
  if (n == 0) return KS_SUCCESS;
  const unsigned nb = (unsigned)std::min<long long>((n + 255) / 256, (long long)ctx->num_cu * 16);
  hipLaunchKernelGGL(k_lincomb, dim3(nb), dim3(256), 0, ctx->stream, n, s, a, u, b, v, out);
  KS_HIP(hipGetLastError());
  return KS_SUCCESS;
}
namespace {
inline int lincomb(ks_ctx ctx, long long n, const double *s, double a, const double *u, double b, const double *v, double *out) { return ksk_lincomb(ctx, n, s, a, u, b, v, out); }

// out = s .* (a*A*x + b*(B*x | x)); tmp is scratch of n doubles (used when both terms are present)
int linop_apply(ks_st st, double a, ks_mat A, double b, ks_mat B, bool identity_term, const double *s, const double *x, double *out, double *tmp)
{
  ks_ctx ctx = st->ctx;
  const long long n = st->n;
  if (A) {
    if (s && a == 1.0 && (b == 0.0 || (!B && !identity_term)) && ks_mat_can_rowscale(A)) return ks_mat_mult_internal(A, x, out, s);   // out = s .* (A x) in the product's own launches
    KS_CALL(ks_mat_mult_internal(A, x, out));
    if (b != 0.0) {                                         // a zero shift leaves A alone (MatMult_Shell, stshellmat.c:52: "if (ctx->alpha!=0.0)")
      if (B) { KS_CALL(ks_mat_mult_internal(B, x, tmp)); return lincomb(ctx, n, s, a, out, b, tmp, out); }
      if (identity_term) return lincomb(ctx, n, s, a, out, b, x, out);
    }
    if (s || a != 1.0) return lincomb(ctx, n, s, a, out, 0.0, nullptr, out);
    return KS_SUCCESS;
  }
  if (B) { KS_CALL(ks_mat_mult_internal(B, x, out)); if (s || b != 1.0) return lincomb(ctx, n, s, b, out, 0.0, nullptr, out); return KS_SUCCESS; }
  return lincomb(ctx, n, s, b, x, 0.0, nullptr, out);
}

// the matrix P of the table above, y = s .* P x
int apply_P(ks_st st, const double *s, const double *x, double *out, double *tmp)
{
  if (st->Pmat) {                                            // ST_MATMODE_COPY: the assembled A - sigma B, one product
    if (s && ks_mat_can_rowscale(st->Pmat)) return ks_mat_mult_internal(st->Pmat, x, out, s);
    KS_CALL(ks_mat_mult_internal(st->Pmat, x, out));
    return s ? lincomb(st->ctx, st->n, s, 1.0, out, 0.0, nullptr, out) : KS_SUCCESS;
  }
  if (st->type == KS_ST_SINVERT || st->type == KS_ST_CAYLEY) return linop_apply(st, 1.0, st->A, -st->sigma, st->B, !st->B, s, x, out, tmp);
  return linop_apply(st, 0.0, nullptr, 1.0, st->B, false, s, x, out, tmp);           // shift, nmat=2: P = B
}

int bjacobi_apply(ks_st st, const double *in, double *out)
{
  if (st->n == 0) return KS_SUCCESS;
  const unsigned nb = (unsigned)std::min<long long>(((long long)st->n + 255) / 256, (long long)st->ctx->num_cu * 16);
  hipLaunchKernelGGL(k_bjacobi_apply, dim3(nb), dim3(256), 0, st->ctx->stream, (long long)st->n, st->pc_bs, st->binv, in, out);
  KS_HIP(hipGetLastError());
  return KS_SUCCESS;
}
// out = M^-1 (a u + b v) with the left preconditioner M of the KSP: diag(P) (one fused kernel) or the diagonal blocks of P
int pc_lincomb(ks_st st, double a, const double *u, double b, const double *v, double *out)
{
  if (st->pc_type == KS_PC_JACOBI) return lincomb(st->ctx, st->n, st->dinv, a, u, b, v, out);
  KS_CALL(lincomb(st->ctx, st->n, nullptr, a, u, b, v, st->pcwork));
  return bjacobi_apply(st, st->pcwork, out);
}
// out = M^-1 P x
int apply_MP(ks_st st, const double *x, double *out, double *tmp)
{
  if (st->pc_type == KS_PC_JACOBI) return apply_P(st, st->dinv, x, out, tmp);
  KS_CALL(apply_P(st, nullptr, x, st->pcwork, tmp));
  return bjacobi_apply(st, st->pcwork, out);
}

// Left-preconditioned restarted GMRES for P y = rhs (KSPGMRES defaults: restart 30, classical Gram-Schmidt)
int gmres_solve(ks_st st, const double *rhs, double *y)
{
  ks_ctx ctx = st->ctx; ks_bv K = st->K;
  const int m = st->restart;
  const long long n = st->n;
  double *t1 = ks_bv_col(st->W, 1), *t2 = ks_bv_col(st->W, 2);
  std::vector<double> H((size_t)(m + 1) * m, 0.0), g(m + 1, 0.0), cs(m, 0.0), sn(m, 0.0), h(m + 1, 0.0), yc(m, 0.0);
  st->solves++;
  KS_HIP(hipMemsetAsync(y, 0, sizeof(double) * std::max<long long>(n, 1), ctx->stream));
  KS_CALL(pc_lincomb(st, 1.0, rhs, 0.0, nullptr, ks_bv_col(K, 0)));
  double beta = 0.0;
  const bool split = ks_bv_orthonormalize_can_split(K);
  bool k0_normalised = false, applied0 = false;
  if (split) {
    // norm and scaling of the first basis vector as one enqueued program (BVOrthonormalizeColumn of column 0), the first operator application behind it:
    // the host learns beta while the product runs
    int lin0 = 0, late0 = 0;
    KS_CALL(ks_bv_orthonormalize_enqueue(K, 0));
    KS_CALL(apply_MP(st, ks_bv_col(K, 0), ks_bv_col(K, 1), t1));
    KS_CALL(ks_bv_orthonormalize_collect(K, 0, nullptr, &beta, &lin0, &late0));
    k0_normalised = true; applied0 = !late0;
  } else KS_CALL(ks_bv_normcolumn(K, 0, KS_NORM_2, &beta));
  st->last_rnorm = beta;
  if (beta == 0.0) return KS_SUCCESS;
  const double tol = std::max(st->rtol * beta, 1e-50);
  int its = 0;
  for (;;) {
    if (!k0_normalised) KS_CALL(ks_bv_scalecolumn(K, 0, 1.0 / beta));
    k0_normalised = false;
    std::fill(g.begin(), g.end(), 0.0); g[0] = beta;
    int jj = 0; double res = beta, res_before = beta;
    bool applied = applied0;                        // K(:, j+1) = P K(:, j) already enqueued (speculatively, during the previous iteration)
    applied0 = false;
    for (int j = 0; j < m; j++) {
      if (!applied) KS_CALL(apply_MP(st, ks_bv_col(K, j), ks_bv_col(K, j + 1), t1));
      applied = false;
      double hn = 0.0; int lindep = 0;
      if (split) {
        // The next operator application goes in behind the orthogonalisation BEFORE the host waits for this column's coefficients, so the device
        // is not idle while the host rotates and tests (30 us per iteration in the trace, profiles/r02_config5_kernel_trace_gaps.txt) - but only
        // when the residual history says another iteration is coming: a product wasted on the last iteration costs more than the gaps of a solve.
        const double predicted = (j == 0) ? res : res * std::min(1.0, res / res_before);
        const bool spec = (j + 1 < m) && (its + 1 < st->max_it) && predicted > 3.0 * tol;
        int late = 0;
        KS_CALL(ks_bv_orthonormalize_enqueue(K, j + 1));
        if (spec) KS_CALL(apply_MP(st, ks_bv_col(K, j + 1), ks_bv_col(K, j + 2), t1));
        KS_CALL(ks_bv_orthonormalize_collect(K, j + 1, h.data(), &hn, &lindep, &late));
        applied = spec && !late;                    // a column completed late was multiplied unfinished: apply again
      } else KS_CALL(ks_bv_orthonormalize_coefs(K, j + 1, h.data(), &hn, &lindep));       // the 1/hn scaling rides in the final update; h, hn and the flag arrive in one host wait
      its++; st->its++;
      h[j + 1] = hn;
      for (int i = 0; i < j; i++) { const double t = cs[i] * h[i] + sn[i] * h[i + 1]; h[i + 1] = -sn[i] * h[i] + cs[i] * h[i + 1]; h[i] = t; }
      const double r = hypot(h[j], h[j + 1]);
      if (r == 0.0) { cs[j] = 1.0; sn[j] = 0.0; } else { cs[j] = h[j] / r; sn[j] = h[j + 1] / r; }
      h[j] = r; h[j + 1] = 0.0;
      g[j + 1] = -sn[j] * g[j]; g[j] = cs[j] * g[j];
      for (int i = 0; i <= j; i++) H[(size_t)i + (size_t)j * (m + 1)] = h[i];
      res_before = res; res = fabs(g[j + 1]); jj = j + 1;
      if (res <= tol || its >= st->max_it || lindep || hn == 0.0) break;
    }
    for (int i = jj - 1; i >= 0; i--) {                     // back substitution R yc = g
      double t = g[i];
      for (int c = i + 1; c < jj; c++) t -= H[(size_t)i + (size_t)c * (m + 1)] * yc[c];
      const double d = H[(size_t)i + (size_t)i * (m + 1)];
      yc[i] = (d != 0.0) ? t / d : 0.0;
    }
    KS_CALL(ks_bv_set_active_columns(K, 0, jj));
    KS_CALL(ks_bv_multvec(K, 1.0, 1.0, y, yc.data()));
    st->last_rnorm = res;
    if (res <= tol) break;
    KS_CHECK(its < st->max_it, KS_ERR_NOT_CONVERGED, "KSPSolve has not converged: GMRES reached %d iterations, preconditioned residual %g > %g", its, res, tol);
    // restart from the true preconditioned residual
    KS_CALL(apply_P(st, nullptr, y, t1, t2));
    KS_CALL(pc_lincomb(st, 1.0, rhs, -1.0, t1, ks_bv_col(K, 0)));
    KS_CALL(ks_bv_normcolumn(K, 0, KS_NORM_2, &beta));
    st->last_rnorm = beta;
    if (beta <= tol) break;
  }
  return KS_SUCCESS;
}

// Left-preconditioned BiCGStab for P y = rhs (KSPBCGS with PC_LEFT: the iteration runs on D^-1 P, the residual that is tested
// is the preconditioned one, as for the GMRES above). Two operator applications per iteration, no growing basis: the
// memory is 7 vectors whatever the iteration count. The dot products of one phase travel in one BVDotVec (one allreduce).
int bcgs_solve(ks_st st, const double *rhs, double *y)
{
  ks_ctx ctx = st->ctx; ks_bv K = st->Kb;
  const long long n = st->n;
  double *t1 = ks_bv_col(st->W, 1);
  double *r = ks_bv_col(K, 0), *rh = ks_bv_col(K, 1), *p = ks_bv_col(K, 2), *v = ks_bv_col(K, 3), *s = ks_bv_col(K, 4), *t = ks_bv_col(K, 5);
  st->solves++;
  KS_HIP(hipMemsetAsync(y, 0, sizeof(double) * std::max<long long>(n, 1), ctx->stream));
  KS_CALL(pc_lincomb(st, 1.0, rhs, 0.0, nullptr, r));                   // r = D^-1 b (zero initial guess)
  double beta0 = 0.0;
  KS_CALL(ks_bv_normcolumn(K, 0, KS_NORM_2, &beta0));
  st->last_rnorm = beta0;
  if (beta0 == 0.0) return KS_SUCCESS;
  const double tol = std::max(st->rtol * beta0, 1e-50);
  KS_CALL(ksk_copy(ctx, r, rh, n));
  KS_HIP(hipMemsetAsync(p, 0, sizeof(double) * n, ctx->stream));
  KS_HIP(hipMemsetAsync(v, 0, sizeof(double) * n, ctx->stream));
  double rho = 1.0, alpha = 1.0, omega = 1.0, d[2];
  for (int its = 0; ; its++) {
    KS_CHECK(its < st->max_it, KS_ERR_NOT_CONVERGED, "KSPSolve has not converged: BiCGStab reached %d iterations, preconditioned residual %g > %g", its, st->last_rnorm, tol);
    KS_CALL(ks_bv_set_active_columns(K, 0, 1));
    KS_CALL(ks_bv_dotvec(K, rh, d));                                                 // rho' = (rhat, r)
    const double rho_new = d[0];
    KS_CHECK(rho_new != 0.0 && omega != 0.0, KS_ERR_NOT_CONVERGED, "KSPSolve has not converged: BiCGStab breakdown (rho = %g, omega = %g)", rho_new, omega);
    const double bt = (rho_new / rho) * (alpha / omega);
    KS_CALL(lincomb(ctx, n, nullptr, 1.0, p, -omega, v, p));                         // p = r + beta (p - omega v)
    KS_CALL(lincomb(ctx, n, nullptr, bt, p, 1.0, r, p));
    KS_CALL(apply_MP(st, p, v, t1));                                                // v = D^-1 P p
    KS_CALL(ks_bv_set_active_columns(K, 3, 4));
    KS_CALL(ks_bv_dotvec(K, rh, d));                                                 // (rhat, v)
    KS_CHECK(d[0] != 0.0, KS_ERR_NOT_CONVERGED, "KSPSolve has not converged: BiCGStab breakdown ((rhat,v) = 0)");
    alpha = rho_new / d[0];
    KS_CALL(lincomb(ctx, n, nullptr, 1.0, r, -alpha, v, s));                         // s = r - alpha v
    KS_CALL(apply_MP(st, s, t, t1));                                                // t = D^-1 P s
    KS_CALL(ks_bv_set_active_columns(K, 4, 6));
    KS_CALL(ks_bv_dotvec(K, t, d));                                                  // (s,t), (t,t) in one reduction
    omega = d[1] != 0.0 ? d[0] / d[1] : 0.0;
    KS_CALL(lincomb(ctx, n, nullptr, 1.0, y, alpha, p, y));                          // y += alpha p + omega s
    KS_CALL(lincomb(ctx, n, nullptr, 1.0, y, omega, s, y));
    KS_CALL(lincomb(ctx, n, nullptr, 1.0, s, -omega, t, r));                         // r = s - omega t
    rho = rho_new;
    st->its++;
    double rn = 0.0;
    KS_CALL(ks_bv_normcolumn(K, 0, KS_NORM_2, &rn));
    st->last_rnorm = rn;
    if (rn <= tol) break;
  }
  return KS_SUCCESS;
}
int inner_solve(ks_st st, const double *rhs, double *y) { return st->ksp_type == KS_KSP_BCGS ? bcgs_solve(st, rhs, y) : gmres_solve(st, rhs, y); }

int st_shell_mult(void *user, const double *x, double *y) { return ks_st_apply_internal((ks_st)user, x, y); }
int st_shell_mult_transpose(void *user, const double *x, double *y) { return ks_st_apply_transpose_internal((ks_st)user, x, y); }
// y = (A + nu B) x, MatMult_Cayley cayley.c:21-44
int st_bilinear_mult(void *user, const double *x, double *y)
{
  ks_st st = (ks_st)user;
  return linop_apply(st, 1.0, st->A, st->nu, st->B, !st->B, nullptr, x, y, ks_bv_col(st->W, 2));
}

} // namespace

// Block Jacobi set-up (PCSetUp_BJacobi with LU sub-solves, PETSc): the dense diagonal blocks of P - pc_bs consecutive local rows each - from the
// CSR arrays the matrices keep (entry by entry a_ij + (-sigma b_ij), ks_csr.cpp, as the assembled P of ST_MATMODE_COPY has them), inverted on
// the host by Gauss-Jordan elimination with partial pivoting, uploaded row by row.
static int bjacobi_setup(ks_st st)
{
  ks_ctx ctx = st->ctx; ks_mat A = st->A, B = st->B;
  const int n = st->n, bs = st->pc_bs;
  const bool p_is_b = (st->type == KS_ST_SHIFT);                   // shift, nmat = 2: P = B
  ks_mat M0 = p_is_b ? B : A;
  KS_CHECK(M0 && M0->keep_csr && (p_is_b || !B || B->keep_csr), KS_ERR_ORDER, "the block-Jacobi preconditioner takes its blocks from the CSR arrays of the matrices: create them with KS_MAT_KEEP_CSR");
  std::vector<int> rp, col; std::vector<double> val;
  const int *prp; const int *pcol; const double *pval;
  if (p_is_b) { prp = B->k_rowptr.data(); pcol = B->k_col.data(); pval = B->k_val.data(); }
  else {
    bool fits = false;
    try { fits = ksc::csr_axpy(n, A->row_start, A->k_rowptr.data(), A->k_col.data(), A->k_val.data(), -st->sigma, B ? B->k_rowptr.data() : nullptr, B ? B->k_col.data() : nullptr, B ? B->k_val.data() : nullptr, rp, col, val); }
    catch (const std::exception &e) { KS_FAIL(KS_ERR_MEM, "block Jacobi set-up: %s", e.what()); }
    KS_CHECK(fits, KS_ERR_ARG_OUTOFRANGE, "A - sigma B exceeds 32-bit PetscInt indices");
    prp = rp.data(); pcol = col.data(); pval = val.data();
  }
  std::vector<double> inv;
  try { inv.assign((size_t)n * bs, 0.0); } catch (const std::exception &e) { KS_FAIL(KS_ERR_MEM, "block Jacobi set-up: %s", e.what()); }
  std::vector<double> Mb((size_t)bs * 2 * bs);
  const long long base = M0->row_start;
  for (int b0 = 0; b0 < n; b0 += bs) {
    const int bl = std::min(bs, n - b0), w = 2 * bl;
    std::fill(Mb.begin(), Mb.end(), 0.0);
    for (int r = 0; r < bl; r++) {                                 // [block | I]
      for (int p = prp[b0 + r]; p < prp[b0 + r + 1]; p++) { const long long c = (long long)pcol[p] - base - b0; if (c >= 0 && c < bl) Mb[(size_t)r * w + c] += pval[p]; }
      Mb[(size_t)r * w + bl + r] = 1.0;
    }
    for (int k = 0; k < bl; k++) {
      int piv = k; double big = fabs(Mb[(size_t)k * w + k]);
      for (int r = k + 1; r < bl; r++) if (fabs(Mb[(size_t)r * w + k]) > big) { big = fabs(Mb[(size_t)r * w + k]); piv = r; }
      KS_CHECK(big != 0.0, KS_ERR_MAT_LU_ZRPVT, "Zero pivot in the block of local rows %d..%d (column %d)", b0, b0 + bl - 1, b0 + k);
      if (piv != k) for (int c = 0; c < w; c++) std::swap(Mb[(size_t)k * w + c], Mb[(size_t)piv * w + c]);
      const double d = 1.0 / Mb[(size_t)k * w + k];
      for (int c = 0; c < w; c++) Mb[(size_t)k * w + c] *= d;
      for (int r = 0; r < bl; r++) if (r != k) { const double f = Mb[(size_t)r * w + k]; if (f != 0.0) for (int c = 0; c < w; c++) Mb[(size_t)r * w + c] -= f * Mb[(size_t)k * w + c]; }
    }
    for (int r = 0; r < bl; r++) for (int c = 0; c < bl; c++) inv[(size_t)(b0 + r) * bs + c] = Mb[(size_t)r * w + bl + c];
  }
  if (st->binv) { hipFree(st->binv); st->binv = nullptr; }
  if (st->pcwork) { hipFree(st->pcwork); st->pcwork = nullptr; }
  KS_HIP(hipMalloc(&st->binv, sizeof(double) * std::max<size_t>(inv.size(), 1)));
  KS_HIP(hipMalloc(&st->pcwork, sizeof(double) * std::max(n, 1)));
  KS_HIP(hipMemcpyAsync(st->binv, inv.data(), sizeof(double) * inv.size(), hipMemcpyHostToDevice, ctx->stream));
  KS_HIP(ks_sync(ctx));
  return KS_SUCCESS;
}

bool ks_st_is_plain(ks_st st) { return !st || (st->type == KS_ST_SHIFT && st->sigma == 0.0 && !st->B); }

int ks_st_setup_internal(ks_st st)
{
  KS_CHECK(st && st->A, KS_ERR_ORDER, "STSetMatrices must be called first");
  if (st->ready) return KS_SUCCESS;
  ks_ctx ctx = st->ctx; ks_mat A = st->A, B = st->B;
  KS_CHECK(!B || (B->n == A->n && B->n_global == A->n_global), KS_ERR_ARG_INCOMP, "Mismatching dimensions of A (%d) and B (%d)", A->n, B ? B->n : 0);
  KS_HIP(hipSetDevice(ctx->device));
  st->n = A->n;
  if (st->type == KS_ST_CAYLEY) {                                      // STComputeOperator_Cayley cayley.c:138-147
    if (!st->nu_set) st->nu = st->sigma;
    KS_CHECK(st->nu != 0.0 || st->sigma != 0.0, KS_ERR_USER_INPUT, "Values of shift and antishift cannot be zero simultaneously");
    KS_CHECK(st->nu != -st->sigma, KS_ERR_USER_INPUT, "It is not allowed to set the antishift equal to minus the shift (the target)");
  }
  if (st->Pmat) { ks_mat_destroy(st->Pmat); st->Pmat = nullptr; }           // the assembled P of an earlier shift / type / mode
  const bool need_solve = (st->type == KS_ST_SINVERT) || (st->type == KS_ST_CAYLEY) || (st->type == KS_ST_SHIFT && B);
  if (st->W) { int wn = 0; ks_bv_get_sizes(st->W, &wn, nullptr, nullptr, nullptr); if (wn != A->n) { ks_bv_destroy(st->W); ks_bv_destroy(st->K); ks_bv_destroy(st->Kb); st->W = st->K = st->Kb = nullptr; if (st->dinv) hipFree(st->dinv); st->dinv = nullptr; } }
  if (!st->W) KS_CALL(ks_bv_create(ctx, A->n, A->n_global, 3, 0, &st->W));
  if (need_solve) {
    if (st->K) { int km = 0; ks_bv_get_sizes(st->K, nullptr, nullptr, &km, nullptr); if (km != st->restart + 1) { ks_bv_destroy(st->K); st->K = nullptr; } }
    if (!st->K) { KS_CALL(ks_bv_create(ctx, A->n, A->n_global, st->restart + 1, 0, &st->K)); st->K->row_start = A->row_start; }
    KS_CALL(ks_bv_set_orthogonalization(st->K, KS_BV_ORTHOG_CGS, st->gmres_refine, 0.7071));      // KSPGMRESClassicalGramSchmidtOrthogonalization + refinement type
    if (st->ksp_type == KS_KSP_BCGS && !st->Kb) { KS_CALL(ks_bv_create(ctx, A->n, A->n_global, 7, 0, &st->Kb)); st->Kb->row_start = A->row_start; }
    if (!st->dinv) KS_HIP(hipMalloc(&st->dinv, sizeof(double) * std::max(A->n, 1)));
    // Jacobi: diag(P)
    double *da = ks_bv_col(st->W, 1), *db = ks_bv_col(st->W, 2);
    const unsigned nb = (unsigned)std::max<long long>(1, std::min<long long>(((long long)A->n + 255) / 256, (long long)ctx->num_cu * 16));
    if (st->matmode == KS_ST_MATMODE_COPY && (st->type == KS_ST_SINVERT || st->type == KS_ST_CAYLEY)) {
      // STMatMAXPY_Private, ST_MATMODE_COPY (stsolve.c:603-631): P = A - sigma B assembled (nmat = 1: MatShift); a zero shift takes A itself
      // there (:611-614) - here a copy of it, so that the ST owns what it destroys
      KS_CALL(ks_mat_create_axpy(A, -st->sigma, B, 0u, &st->Pmat));
      KS_CALL(ks_mat_get_diagonal_internal(st->Pmat, da));
      hipLaunchKernelGGL(k_jacobi_setup, dim3(nb), dim3(256), 0, ctx->stream, (long long)A->n, 1.0, da, 0.0, (const double *)nullptr, st->dinv);
    } else if (st->type == KS_ST_SINVERT || st->type == KS_ST_CAYLEY) {
      KS_CALL(ks_mat_get_diagonal_internal(A, da));
      if (B) KS_CALL(ks_mat_get_diagonal_internal(B, db));
      hipLaunchKernelGGL(k_jacobi_setup, dim3(nb), dim3(256), 0, ctx->stream, (long long)A->n, 1.0, da, -st->sigma, B ? db : nullptr, st->dinv);
    } else {
      KS_CALL(ks_mat_get_diagonal_internal(B, db));
      hipLaunchKernelGGL(k_jacobi_setup, dim3(nb), dim3(256), 0, ctx->stream, (long long)A->n, 0.0, (const double *)nullptr, 1.0, db, st->dinv);
    }
    KS_HIP(hipGetLastError());
    if (st->pc_type == KS_PC_BJACOBI) KS_CALL(bjacobi_setup(st));
  }
  if (!st->op) KS_CALL(ks_mat_create_shell(ctx, A->n, A->row_start, A->n_global, st_shell_mult, st, &st->op));
  st->op->n = A->n; st->op->row_start = A->row_start; st->op->n_global = A->n_global;
  st->op->shell_mult_t = st_shell_mult_transpose;
  st->op->shell_nosync = !need_solve;                         // a plain shift is two kernel launches: Krylov runs stay enqueued ahead
  if (st->type == KS_ST_CAYLEY) {
    if (!st->bil) KS_CALL(ks_mat_create_shell(ctx, A->n, A->row_start, A->n_global, st_bilinear_mult, st, &st->bil));
    st->bil->n = A->n; st->bil->row_start = A->row_start; st->bil->n_global = A->n_global;
  }
  st->ready = true;
  return KS_SUCCESS;
}

int ks_st_apply_internal(ks_st st, const double *x, double *y)     // STApply_Generic stsolve.c:16-25
{
  if (!st->ready) KS_CALL(ks_st_setup_internal(st));
  double *w = ks_bv_col(st->W, 0), *t1 = ks_bv_col(st->W, 1);
  if (st->type == KS_ST_SINVERT) {
    if (st->B) { KS_CALL(ks_mat_mult_internal(st->B, x, w)); return inner_solve(st, w, y); }
    return inner_solve(st, x, y);
  }
  if (st->type == KS_ST_CAYLEY) {                                      // y = (A - sigma B)^-1 (A + nu B) x
    KS_CALL(linop_apply(st, 1.0, st->A, st->nu, st->B, !st->B, nullptr, x, w, t1));
    return inner_solve(st, w, y);
  }
  // shift
  if (st->B) { KS_CALL(linop_apply(st, 1.0, st->A, -st->sigma, st->B, false, nullptr, x, w, t1)); return inner_solve(st, w, y); }
  return linop_apply(st, 1.0, st->A, -st->sigma, nullptr, st->sigma != 0.0, nullptr, x, y, t1);
}

void ks_st_backtransform_internal(ks_st st, int n, double *eigr, double *eigi)
{
  if (!st) return;
  for (int j = 0; j < n; j++) {
    if (st->type == KS_ST_SHIFT) eigr[j] += st->sigma;                                  // shift.c:49-56
    else if (st->type == KS_ST_CAYLEY) {                                                 // cayley.c:79-107
      if (eigi[j] == 0.0) eigr[j] = (st->nu + eigr[j] * st->sigma) / (eigr[j] - 1.0);
      else {
        // lambda = (nu + theta sigma) / (theta - 1) for theta = a + b i. Stated deviation: cayley.c:93-99 forms the denominator
        // |theta - 1|^2 = b^2 + a (a - 2) + 1 AFTER it has overwritten a and b with the numerator; here it is taken from theta.
        const double a = eigr[j], b = eigi[j];
        const double t = b * b + a * (a - 2.0) + 1.0;
        eigr[j] = (st->sigma * (a * a + b * b - a) + st->nu * (a - 1.0)) / t;
        eigi[j] = (-st->sigma * b - st->nu * b) / t;
      }
    }
    else if (eigi[j] == 0.0) eigr[j] = 1.0 / eigr[j] + st->sigma;                        // sinvert.c:16-40
    else { const double t = eigr[j] * eigr[j] + eigi[j] * eigi[j]; eigr[j] = eigr[j] / t + st->sigma; eigi[j] = -eigi[j] / t; }
  }
}

extern "C" int ks_st_create(ks_ctx ctx, ks_st *out)
{
  KS_CHECK(ctx && out, KS_ERR_ARG_NULL, "ctx/out is NULL");
  ks_st st = new ks_st_s(); st->ctx = ctx; *out = st;
  return KS_SUCCESS;
}
extern "C" int ks_st_destroy(ks_st st)
{
  if (!st) return KS_SUCCESS;
  ks_bv_destroy(st->K); ks_bv_destroy(st->W); ks_bv_destroy(st->Kb);
  if (st->dinv) hipFree(st->dinv);
  if (st->op) ks_mat_destroy(st->op);
  if (st->bil) ks_mat_destroy(st->bil);
  if (st->Pmat) ks_mat_destroy(st->Pmat);
  if (st->binv) hipFree(st->binv);
  if (st->pcwork) hipFree(st->pcwork);
  delete st;
  return KS_SUCCESS;
}
extern "C" int ks_st_set_type(ks_st st, int type)                    // STSetType
{
  KS_CHECK(st, KS_ERR_ARG_NULL, "ST is NULL");
  KS_CHECK(type == KS_ST_SHIFT || type == KS_ST_SINVERT || type == KS_ST_CAYLEY, KS_ERR_SUP, "only STSHIFT, STSINVERT and STCAYLEY are built");
  if (st->type != type) { st->type = type; st->ready = false; }
  return KS_SUCCESS;
}
extern "C" int ks_st_set_shift(ks_st st, double sigma)               // STSetShift stfunc.c
{
  KS_CHECK(st, KS_ERR_ARG_NULL, "ST is NULL");
  if (st->sigma != sigma || !st->sigma_set) { st->sigma = sigma; st->ready = false; }
  st->sigma_set = true;
  return KS_SUCCESS;
}
extern "C" int ks_st_cayley_set_antishift(ks_st st, double nu)       // STCayleySetAntishift cayley.c:236
{
  KS_CHECK(st, KS_ERR_ARG_NULL, "ST is NULL");
  if (st->nu != nu || !st->nu_set) { st->nu = nu; st->ready = false; }
  st->nu_set = true;
  return KS_SUCCESS;
}
extern "C" int ks_st_cayley_get_antishift(ks_st st, double *nu) { KS_CHECK(st && nu, KS_ERR_ARG_NULL, "NULL argument"); *nu = st->nu; return KS_SUCCESS; }
extern "C" int ks_st_get_shift(ks_st st, double *sigma) { KS_CHECK(st && sigma, KS_ERR_ARG_NULL, "NULL argument"); *sigma = st->sigma; return KS_SUCCESS; }
extern "C" int ks_st_set_matrices(ks_st st, ks_mat A, ks_mat B)      // STSetMatrices (n = 1 or 2)
{
  KS_CHECK(st && A, KS_ERR_ARG_NULL, "NULL argument");
  KS_CHECK(!A->shell_mult && (!B || !B->shell_mult), KS_ERR_SUP, "ST needs assembled matrices (their diagonals feed the Jacobi preconditioner)");
  st->A = A; st->B = B; st->ready = false;
  return KS_SUCCESS;
}
extern "C" int ks_st_set_matmode(ks_st st, int mode)                 // STSetMatMode
{
  KS_CHECK(st, KS_ERR_ARG_NULL, "ST is NULL");
  KS_CHECK(mode == KS_ST_MATMODE_COPY || mode == KS_ST_MATMODE_SHELL, KS_ERR_SUP, "only ST_MATMODE_COPY and ST_MATMODE_SHELL are built");
  if (st->matmode != mode) { st->matmode = mode; st->ready = false; }
  return KS_SUCCESS;
}
extern "C" int ks_st_get_matmode(ks_st st, int *mode) { KS_CHECK(st && mode, KS_ERR_ARG_NULL, "NULL argument"); *mode = st->matmode; return KS_SUCCESS; }
extern "C" int ks_st_set_ksp(ks_st st, double rtol, int max_it, int restart)   // KSPSetTolerances / KSPGMRESSetRestart on STGetKSP
{
  KS_CHECK(st, KS_ERR_ARG_NULL, "ST is NULL");
  if (rtol > 0.0) st->rtol = rtol;
  if (max_it > 0) st->max_it = max_it;
  if (restart > 0 && restart != st->restart) { st->restart = restart; st->ready = false; }   // a basis wider than 64 columns orthogonalises through the host-driven loop
  return KS_SUCCESS;
}
extern "C" int ks_st_set_ksp_type(ks_st st, int type)              // KSPSetType on STGetKSP: KSPGMRES (default) or KSPBCGS
{
  KS_CHECK(st, KS_ERR_ARG_NULL, "ST is NULL");
  KS_CHECK(type == KS_KSP_GMRES || type == KS_KSP_BCGS, KS_ERR_SUP, "only KSPGMRES and KSPBCGS are built");
  if (st->ksp_type != type) { st->ksp_type = type; st->ready = false; }
  return KS_SUCCESS;
}
extern "C" int ks_st_set_pc(ks_st st, int type, int block_size)       // PCSetType (+ PCBJacobiSetLocalBlocks, -sub_pc_type lu) on the KSP's PC
{
  KS_CHECK(st, KS_ERR_ARG_NULL, "ST is NULL");
  KS_CHECK(type == KS_PC_JACOBI || type == KS_PC_BJACOBI, KS_ERR_SUP, "only PCJACOBI and PCBJACOBI are built");
  KS_CHECK(type == KS_PC_JACOBI || (block_size >= 2 && block_size <= 32), KS_ERR_ARG_OUTOFRANGE, "block size %d (2..32)", block_size);
  if (type == KS_PC_JACOBI) block_size = 0;
  if (st->pc_type != type || st->pc_bs != block_size) { st->pc_type = type; st->pc_bs = block_size; st->ready = false; }
  return KS_SUCCESS;
}
extern "C" int ks_st_set_gmres_cgs_refinement(ks_st st, int refine)   // KSPGMRESSetCGSRefinementType on STGetKSP
{
  KS_CHECK(st, KS_ERR_ARG_NULL, "ST is NULL");
  KS_CHECK(refine == KS_BV_ORTHOG_REFINE_NEVER || refine == KS_BV_ORTHOG_REFINE_IFNEEDED || refine == KS_BV_ORTHOG_REFINE_ALWAYS, KS_ERR_ARG_OUTOFRANGE, "unknown refinement type %d", refine);
  if (st->gmres_refine != refine) { st->gmres_refine = refine; st->ready = false; }
  return KS_SUCCESS;
}
extern "C" int ks_st_setup(ks_st st) { KS_CHECK(st, KS_ERR_ARG_NULL, "ST is NULL"); return ks_st_setup_internal(st); }
extern "C" int ks_st_apply(ks_st st, const double *x_dev, double *y_dev)        // STApply stsolve.c:44
{
  KS_CHECK(st && x_dev && y_dev, KS_ERR_ARG_NULL, "NULL argument");
  KS_CHECK(x_dev != y_dev, KS_ERR_ARG_IDN, "x and y must be different vectors");
  KS_CHECK(st->A, KS_ERR_ORDER, "STSetMatrices must be called first");
  KS_HIP(hipSetDevice(st->ctx->device));
  return ks_st_apply_internal(st, x_dev, y_dev);
}
// STApplyHermitianTranspose_Generic (stsolve.c:153-162), real scalars: only the branch without a solve (st->M alone: shift with one matrix)
int ks_st_apply_transpose_internal(ks_st st, const double *x, double *y)
{
  KS_CALL(ks_st_setup_internal(st));
  KS_CHECK(st->type == KS_ST_SHIFT && !st->B, KS_ERR_SUP, "STApplyHermitianTranspose is built for STSHIFT with one matrix (the others need a solve with the transposed matrix)");
  KS_CALL(ks_mat_mult_transpose_internal(st->A, x, y));
  if (st->sigma != 0.0) KS_CALL(ksk_lincomb(st->ctx, st->n, nullptr, 1.0, y, -st->sigma, x, y));      // (A - sigma I)^T x
  return KS_SUCCESS;
}
extern "C" int ks_st_apply_transpose(ks_st st, const double *x_dev, double *y_dev)
{
  KS_CHECK(st && x_dev && y_dev, KS_ERR_ARG_NULL, "NULL argument");
  KS_CHECK(x_dev != y_dev, KS_ERR_ARG_IDN, "x and y must be different vectors");
  KS_CHECK(st->A, KS_ERR_ORDER, "STSetMatrices must be called first");
  KS_HIP(hipSetDevice(st->ctx->device));
  return ks_st_apply_transpose_internal(st, x_dev, y_dev);
}
extern "C" int ks_st_backtransform(ks_st st, int n, double *eigr, double *eigi) // STBackTransform stsolve.c:563
{
  KS_CHECK(st && (n == 0 || (eigr && eigi)), KS_ERR_ARG_NULL, "NULL argument");
  ks_st_backtransform_internal(st, n, eigr, eigi);
  return KS_SUCCESS;
}
extern "C" int ks_st_get_ksp_stats(ks_st st, long long *solves, long long *iterations, double *last_rnorm)
{
  KS_CHECK(st, KS_ERR_ARG_NULL, "ST is NULL");
  if (solves) *solves = st->solves; if (iterations) *iterations = st->its; if (last_rnorm) *last_rnorm = st->last_rnorm;
  return KS_SUCCESS;
}
