// Host side of the path: the Krylov-Schur restart driver and the dense projected problem (DS HEP).
//
// Restates, in plain C++ on host scalars (no device code here):
//   EPSSetUp_KrylovSchur            src/eps/impls/krylov/krylovschur/krylovschur.c:93-194
//   EPSSetDimensions_Default        src/eps/interface/epssetup.c:654-678
//   EPSSolve_KrylovSchur_Default    krylovschur.c:227-337
//   EPSKrylovConvergence            src/eps/impls/krylov/epskrylov.c:207-295
//   EPSConvergedRelative / EPSStoppingBasic   src/eps/interface/epsdefault.c:224,290
//   EPSGetStartVector               src/eps/interface/epssolve.c:841-873
//   EPSSolve epilogue + SlepcSortEigenvalues  epssolve.c:119-208, src/sys/slepcsc.c:89-140
//   EPSComputeError / EPSComputeResidualNorm_Private  epssolve.c:666-718,742-815
//   DS HEP (compact, extra row):    src/sys/classes/ds/impls/hep/dshep.c:137-175 (vectors), :221-262
//       (DSArrowTridiag), :267-321 (intermediate), :323-347 (sort), :349-381 (extra row), :383-426
//       (solve), :643-671 (truncate); sort kernels src/sys/classes/ds/interface/dspriv.c:224-270
// The reference calls LAPACK steqr / lartg / BLAS rot for the m x m (m <= 64) problem; LAPACK is not
// part of this image's C toolchain, so the tridiagonal eigenproblem is solved by the implicit QL/QR
// iteration written out below (same algorithm family as steqr; eigenvalues returned ascending as steqr
// does, so that the insertion sort of DSSort sees the same input order).
#include "ksgpu_internal.h"
#include "ks_dense.h"
#include <algorithm>
#include <limits>

namespace {

// ---- small dense kernels ---------------------------------------------------------------------------
// Givens rotation with LAPACK-3.10 dlartg conventions: c >= 0, r = sign(f)*hypot(f,g)
void lartg(double f, double g, double *c, double *s, double *r)
{
  if (g == 0.0) { *c = 1.0; *s = 0.0; *r = f; }
  else if (f == 0.0) { *c = 0.0; *s = (g < 0.0) ? -1.0 : 1.0; *r = fabs(g); }
  else { const double d = hypot(f, g); *c = fabs(f) / d; *r = copysign(d, f); *s = g / *r; }
}

// BLAS drot on the first n entries of two columns
void rot(int n, double *x, double *y, double c, double s)
{
  for (int i = 0; i < n; i++) { const double t = c * x[i] + s * y[i]; y[i] = c * y[i] - s * x[i]; x[i] = t; }
}

// Symmetric tridiagonal eigenproblem by implicit QL with Wilkinson shifts, accumulating the rotations
// into the columns of Z (Z <- Z * eigvecs), eigenvalues sorted ascending on exit (steqr 'V' contract).
// d[0..n), e[0..n-1) ; Z is ldz x n column-major with n rows used (nz rows updated).
int tridiag_ql(int n, double *d, double *e, double *Z, int ldz, int nz)
{
  if (n <= 1) return 0;
  std::vector<double> ee(n, 0.0);
  for (int i = 0; i < n - 1; i++) ee[i] = e[i];
  const double eps = std::numeric_limits<double>::epsilon();
  for (int l = 0; l < n; l++) {
    int iter = 0, mm;
    do {
      for (mm = l; mm < n - 1; mm++) {
        const double dd = fabs(d[mm]) + fabs(d[mm + 1]);
        if (fabs(ee[mm]) <= eps * dd) break;
      }
      if (mm != l) {
        if (iter++ == 60 * 4) return l + 1;
        double g = (d[l + 1] - d[l]) / (2.0 * ee[l]);
        double r = hypot(g, 1.0);
        g = d[mm] - d[l] + ee[l] / (g + copysign(r, g));
        double s = 1.0, c = 1.0, p = 0.0;
        int i;
        for (i = mm - 1; i >= l; i--) {
          double f = s * ee[i];
          const double b = c * ee[i];
          r = hypot(f, g);
          ee[i + 1] = r;
          if (r == 0.0) { d[i + 1] -= p; ee[mm] = 0.0; break; }
          s = f / r; c = g / r;
          g = d[i + 1] - p;
          r = (d[i] - g) * s + 2.0 * c * b;
          p = s * r;
          d[i + 1] = g + p;
          g = c * r - b;
          for (int k = 0; k < nz; k++) {
            double *zk = Z + k;
            f = zk[(size_t)(i + 1) * ldz];
            zk[(size_t)(i + 1) * ldz] = s * zk[(size_t)i * ldz] + c * f;
            zk[(size_t)i * ldz] = c * zk[(size_t)i * ldz] - s * f;
          }
        }
        if (r == 0.0 && i >= l) continue;
        d[l] -= p; ee[l] = g; ee[mm] = 0.0;
      }
    } while (mm != l);
  }
  // selection sort, ascending, swapping eigenvector columns (dsteqr epilogue)
  for (int ii = 1; ii < n; ii++) {
    const int i = ii - 1; int k = i; double p = d[i];
    for (int j = ii; j < n; j++) if (d[j] < p) { k = j; p = d[j]; }
    if (k != i) { d[k] = d[i]; d[i] = p; for (int r = 0; r < nz; r++) std::swap(Z[r + (size_t)i * ldz], Z[r + (size_t)k * ldz]); }
  }
  for (int i = 0; i < n - 1; i++) e[i] = 0.0;
  return 0;
}

// SlepcCompare* (src/sys/slepcsc.c:152-300), real scalars; EPS_WHICH_USER calls the function installed with
// ks_eps_set_eigenvalue_comparison (EPSSetEigenvalueComparison epsopts.c:563)
struct KsCompare {
  int which = 0;                      // 0 = not set: resolved at set-up (EPSSetWhichEigenpairs_Default epsdefault.c:209-219)
  double target = 0.0;
  ks_eig_compare_fn fn = nullptr; void *fn_ctx = nullptr;
  ks_st map = nullptr;                // SlepcMap_ST: compare the back-transformed values (SlepcSCCompare slepcsc.c:41-62)
};
int compare_eig(const KsCompare &cmp, double ar, double ai, double br, double bi)
{
  double a, b;
  if (cmp.map) { ks_st_backtransform_internal(cmp.map, 1, &ar, &ai); ks_st_backtransform_internal(cmp.map, 1, &br, &bi); }
  switch (cmp.which) {
    case KS_EPS_LARGEST_MAGNITUDE:  a = hypot(ar, ai); b = hypot(br, bi); return a < b ? 1 : (a > b ? -1 : 0);
    case KS_EPS_SMALLEST_MAGNITUDE: a = hypot(ar, ai); b = hypot(br, bi); return a > b ? 1 : (a < b ? -1 : 0);
    case KS_EPS_LARGEST_REAL:       return ar < br ? 1 : (ar > br ? -1 : 0);
    case KS_EPS_SMALLEST_REAL:      return ar > br ? 1 : (ar < br ? -1 : 0);
    case KS_EPS_LARGEST_IMAGINARY:  a = fabs(ai); b = fabs(bi); return a < b ? 1 : (a > b ? -1 : 0);
    case KS_EPS_SMALLEST_IMAGINARY: a = fabs(ai); b = fabs(bi); return a > b ? 1 : (a < b ? -1 : 0);
    case KS_EPS_TARGET_MAGNITUDE:   a = hypot(ar - cmp.target, ai); b = hypot(br - cmp.target, bi); return a > b ? 1 : (a < b ? -1 : 0);
    case KS_EPS_TARGET_REAL:        a = fabs(ar - cmp.target); b = fabs(br - cmp.target); return a > b ? 1 : (a < b ? -1 : 0);
    case KS_EPS_WHICH_USER: { int r = 0; cmp.fn(ar, ai, br, bi, &r, cmp.fn_ctx); return r; }
  }
  return 0;
}

enum { DS_RAW = 0, DS_INTERMEDIATE = 1, DS_CONDENSED = 2, DS_TRUNCATED = 3 };

// DS type HEP, compact storage with extra row (krylovschur.c:160-168)
struct DsHep {
  int ld = 0, n = 0, l = 0, k = 0, t = 0, state = DS_RAW; KsCompare which;
  std::vector<double> T, Q; std::vector<int> perm;
  void allocate(int ld_) { ld = ld_; T.assign((size_t)3 * ld, 0.0); Q.assign((size_t)ld * ld, 0.0); perm.assign(ld, 0); }
  double *d() { return T.data(); }
  double *e() { return T.data() + ld; }
  void set_dimensions(int n_, int l_, int k_) { n = n_; t = n_; l = l_; k = k_; }           // dsops.c:130-165

  void arrow_tridiag(int nn, double *dd, double *ee, double *QQ)                             // dshep.c:221-262
  {
    if (nn <= 2) return;
    for (int j = 0; j < nn - 2; j++) {
      double c, s, temp = ee[j + 1];
      lartg(temp, ee[j], &c, &s, &ee[j + 1]);
      s = -s;
      temp = dd[j + 1];
      ee[j] = c * s * (temp - dd[j]);
      dd[j + 1] = s * s * dd[j] + c * c * temp;
      dd[j] = c * c * dd[j] + s * s * temp;
      const int j2 = j + 2;
      rot(j2, QQ + (size_t)j * ld, QQ + (size_t)(j + 1) * ld, c, s);
      for (int i = j - 1; i >= 0; i--) {
        const double off = -s * ee[i];
        ee[i] = c * ee[i];
        temp = ee[i + 1];
        lartg(temp, off, &c, &s, &ee[i + 1]);
        s = -s;
        temp = (dd[i] - dd[i + 1]) * s - 2.0 * c * ee[i];
        const double p = s * temp;
        dd[i + 1] += p;
        dd[i] -= p;
        ee[i] = -ee[i] - c * temp;
        rot(j2, QQ + (size_t)i * ld, QQ + (size_t)(i + 1) * ld, c, s);
      }
    }
  }

  int solve(double *wr)                                                                     // dsops.c:723, dshep.c:383-426
  {
    if (state >= DS_CONDENSED) return 0;
    const int n1 = n - l; const size_t off = (size_t)l + (size_t)l * ld;
    std::fill(Q.begin(), Q.end(), 0.0);
    for (int i = 0; i < ld; i++) Q[(size_t)i + (size_t)i * ld] = 1.0;                       // DSSetIdentity
    if (state < DS_INTERMEDIATE) arrow_tridiag(std::max(0, k - l + 1), d() + l, e() + l, Q.data() + off);   // DSIntermediate_HEP
    for (int i = 0; i < l; i++) wr[i] = d()[i];
    int info = tridiag_ql(n1, d() + l, e() + l, Q.data() + off, ld, n1);
    if (info) return info;
    for (int i = l; i < n; i++) wr[i] = d()[i];
    for (int i = 0; i < n - 1; i++) e()[i] = 0.0;                                           // compact: zero e(0:n-2), keep e(n-1)
    state = DS_CONDENSED;
    return 0;
  }

  // rr/ri: auxiliary values of an arbitrary selection (DSSort with rr: the order comes from them, dshep.c:335-336)
  void sort(double *wr, const double *rr = nullptr, const double *ri = nullptr)             // dsops.c:329-345, dshep.c:323-347
  {
    for (int i = 0; i < n; i++) perm[i] = i;
    double *dd = d();
    const double *key = rr ? rr : dd;
    auto im = [&](int i) { return ri ? ri[i] : 0.0; };
    // DSSortEigenvaluesReal_Private dspriv.c:224-243 / DSSortEigenvalues_Private :172-222: insertion sort of the first t values from l
    for (int i = l + 1; i < t; i++) {
      const double re = key[perm[i]], rim = im(perm[i]);
      int j = i - 1;
      int result = compare_eig(which, re, rim, key[perm[j]], im(perm[j]));
      while (result < 0 && j >= l) {
        std::swap(perm[j], perm[j + 1]); j--;
        if (j >= l) result = compare_eig(which, re, rim, key[perm[j]], im(perm[j]));
      }
    }
    for (int i = l; i < n; i++) wr[i] = dd[perm[i]];
    // DSPermuteColumns_Private dspriv.c:248-270
    for (int i = l; i < n; i++) {
      const int p = perm[i];
      if (p != i) {
        int j = i + 1;
        while (perm[j] != i) j++;
        perm[j] = p; perm[i] = i;
        for (int r = 0; r < n; r++) std::swap(Q[(size_t)r + (size_t)p * ld], Q[(size_t)r + (size_t)i * ld]);
      }
    }
    for (int i = l; i < n; i++) dd[i] = wr[i];
  }

  void update_extra_row()                                                                   // dshep.c:349-381 (compact)
  {
    const double beta = e()[n - 1];
    for (int i = 0; i < n; i++) e()[i] = beta * Q[(size_t)(n - 1) + (size_t)i * ld];
    k = n;
  }
  double vectors_resnorm(int j) { return fabs(Q[(size_t)(n - 1) + (size_t)j * ld]); }       // dshep.c:152

  void truncate(int nn, bool trim)                                                          // dsops.c DSTruncate + dshep.c:643-671
  {
    if (trim) { l = 0; k = 0; n = nn; t = nn; state = DS_RAW; }
    else { k = nn; t = n; n = nn; state = DS_TRUNCATED; }
  }
};

// DS type NHEP with extra row (krylovschur.c:153-159): A is ld x ld column-major, row n holds the extra row.
// Restates DSSolve_NHEP_Private / DSSort_NHEP_Total (src/sys/classes/ds/impls/dsutil.c:21-175),
// DSVectors_NHEP_Eigen_Some (nhep/dsnhep.c:101-167), DSUpdateExtraRow_NHEP (:318-341), DSTruncate_NHEP (:385-415),
// DSGetTruncateSize_Default (interface/dsops.c:329-345) on top of the host kernels of ks_dense.cpp.
struct DsNhep {
  int ld = 0, n = 0, l = 0, k = 0, t = 0, state = DS_RAW; KsCompare which;
  std::vector<double> A, Q, X;
  void allocate(int ld_) { ld = ld_; A.assign((size_t)ld * ld, 0.0); Q.assign((size_t)ld * ld, 0.0); X.assign((size_t)ld * ld, 0.0); }
  double &a(int i, int j) { return A[(size_t)i + (size_t)j * ld]; }
  double &q(int i, int j) { return Q[(size_t)i + (size_t)j * ld]; }
  void set_dimensions(int n_, int l_, int k_) { n = n_; t = n_; l = l_; k = k_; }

  // DSTranslateHarmonic_NHEP dsnhep.c:466-537. g (ld entries) lives in the caller between the two calls. Forward:
  // g = (A - tau I)^{-T} (beta e_n) and A(:,n-1) += beta g. Recover (after solve and sort, with l = converged and
  // k = kept): the rank-one term is removed from the kept block and g is projected out of the kept Schur vectors.
  int translate_harmonic(double tau, double beta, bool recover, double *g, double *gamma_out)
  {
    if (!recover) {
      std::vector<double> W((size_t)n * n);
      for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) W[i + (size_t)j * n] = a(i, j) - (i == j ? tau : 0.0);
      std::fill(g, g + ld, 0.0); g[n - 1] = beta;
      if (ksd::lu_solve_trans(n, W.data(), n, g)) return 1;                                  // getrf + getrs 'C'
      for (int i = 0; i < n; i++) a(i, n - 1) += g[i] * beta;
    } else {
      const int ncol = l + k;
      std::vector<double> ghat(ncol);
      for (int j = 0; j < ncol; j++) { double s2 = 0.0; for (int i = 0; i < n; i++) s2 += q(i, j) * g[i]; ghat[j] = -s2; }   // gemv 'C', alpha = -1
      for (int i = 0; i < ncol; i++) for (int j = l; j < ncol; j++) a(i, j) += ghat[i] * q(n - 1, j) * beta;
      for (int j = 0; j < ncol; j++) { const double t2 = ghat[j]; if (t2 != 0.0) for (int i = 0; i < n; i++) g[i] += t2 * q(i, j); }   // gemv 'N'
    }
    double scale = 0.0, ssq = 1.0;                                                           // dnrm2
    for (int i = 0; i < n; i++) if (g[i] != 0.0) { const double ax = fabs(g[i]); if (scale < ax) { ssq = 1.0 + ssq * (scale / ax) * (scale / ax); scale = ax; } else ssq += (ax / scale) * (ax / scale); }
    const double gamma = hypot(1.0, scale * sqrt(ssq));                                      // SlepcAbs(1.0, nrm2)
    if (gamma_out) *gamma_out = gamma;
    if (recover) for (int j = l; j < l + k; j++) a(n, j) *= gamma;                           // extra row
    return 0;
  }

  void eig_from_T(double *wr, double *wi, int j0, int j1)                                    // dsutil.c:65-79,160-170
  {
    for (int j = j0; j < j1; j++) {
      if (j == n - 1 || a(j + 1, j) == 0.0) { wr[j] = a(j, j); wi[j] = 0.0; }
      else {
        wr[j] = a(j, j); wr[j + 1] = a(j, j);
        wi[j] = sqrt(fabs(a(j + 1, j))) * sqrt(fabs(a(j, j + 1))); wi[j + 1] = -wi[j];
        j++;
      }
    }
  }

  int solve(double *wr, double *wi)                                                          // dsutil.c:21-91
  {
    if (state >= DS_CONDENSED) return 0;
    std::fill(Q.begin(), Q.end(), 0.0);
    for (int i = 0; i < n; i++) q(i, i) = 1.0;
    if (n == 1) { wr[0] = a(0, 0); wi[0] = 0.0; state = DS_CONDENSED; return 0; }
    if (state < DS_INTERMEDIATE) ksd::hess_reduce(n, l, A.data(), ld, Q.data());             // gehrd + orghr
    const int info = ksd::real_schur(n, l, A.data(), ld, wr, wi, Q.data());                  // hseqr 'S','V'
    if (info) return info;
    eig_from_T(wr, wi, 0, l);
    state = DS_CONDENSED;
    return 0;
  }

  int sort(double *wr, double *wi)                                                           // dsutil.c:93-175
  {
    for (int i = l; i < n - 1; i++) {
      double re = wr[i], im = wi[i];
      int pos = 0;
      for (int j = (im != 0.0) ? i + 2 : i + 1; j < n; j++) {
        if (compare_eig(which, re, im, wr[j], wi[j]) > 0) { re = wr[j]; im = wi[j]; pos = j; }
        if (wi[j] != 0.0) j++;
      }
      if (pos) {
        if (ksd::trexc_up(n, A.data(), ld, Q.data(), pos, i)) return 1;                      // trexc 'V', ifst=pos+1, ilst=i+1
        eig_from_T(wr, wi, i, n);
      }
      if (wi[i] != 0.0) i++;
    }
    return 0;
  }

  void update_extra_row()                                                                    // dsnhep.c:318-341
  {
    std::vector<double> x(n);
    for (int j = 0; j < n; j++) x[j] = a(n, j);
    for (int j = 0; j < n; j++) { double s = 0.0; for (int i = 0; i < n; i++) s += q(i, j) * x[i]; a(n, j) = s; }
    k = n;
  }

  // k-th eigenvector of A back-transformed with Q (or not), normalised, into X(:,k[,k+1]); returns the index of the
  // last column written; rnorm = |last component| (dsnhep.c:101-167)
  int vectors(int kk, bool back, double *rnorm)
  {
    std::vector<double> xr_(n + 1), xi_(n + 1), zr_(n + 1), zi_(n + 1);
    double *xr = xr_.data(), *xi = xi_.data(), *zr = zr_.data(), *zi = zi_.data();
    const bool cplx = ksd::trevc_one(n, A.data(), ld, kk, xr, xi) != 0;
    for (int i = 0; i < n; i++) {
      if (back) { double sr = 0.0, si = 0.0; for (int j = 0; j < n; j++) { sr += q(i, j) * xr[j]; si += q(i, j) * xi[j]; } zr[i] = sr; zi[i] = si; }
      else { zr[i] = xr[i]; zi[i] = xi[i]; }
    }
    double nr = 0.0, ni = 0.0;
    for (int i = 0; i < n; i++) { nr = hypot(nr, zr[i]); ni = hypot(ni, zi[i]); }
    const double norm = cplx ? hypot(nr, ni) : nr;
    for (int i = 0; i < n; i++) { X[(size_t)i + (size_t)kk * ld] = zr[i] / norm; if (cplx) X[(size_t)i + (size_t)(kk + 1) * ld] = zi[i] / norm; }
    if (rnorm) *rnorm = cplx ? hypot(zr[n - 1] / norm, zi[n - 1] / norm) : fabs(zr[n - 1] / norm);
    return cplx ? kk + 1 : kk;
  }

  int get_truncate_size(int ll, int nn, int kk)                                              // dsops.c:329-345
  {
    if (a(ll + kk, ll + kk - 1) != 0.0) kk = (ll + kk < nn - 1) ? kk + 1 : kk - 1;
    return kk;
  }

  void truncate(int nn, bool trim)                                                           // dsnhep.c:385-415
  {
    if (trim) {
      for (int j = l; j < n; j++) a(n, j) = 0.0;
      l = 0; k = 0; n = nn; t = nn; state = DS_RAW;
    } else {
      if (k == n) { for (int j = l; j < nn; j++) a(nn, j) = a(n, j); for (int j = l; j < n; j++) a(n, j) = 0.0; }
      k = nn; t = n; n = nn; state = DS_TRUNCATED;
    }
  }
};

} // namespace

struct ks_eps_s {
  ks_ctx ctx = nullptr;
  ks_mat A = nullptr, B = nullptr;
  ks_mat op = nullptr;                 // operator of the expansion: A itself, or the ST's shell matrix
  // EPSSetBalance (non-symmetric problems): D from EPSBuildBalance_Krylov, the expansion then runs on D Op D^-1 (STApply with st->D, stsolve.c:252-256)
  int balance = KS_EPS_BALANCE_NONE, balance_its = 5; double balance_cutoff = 1e-8;
  double *D = nullptr, *wb = nullptr; int D_n = 0; ks_mat op_inner = nullptr, bal_op = nullptr; bool balanced = false;
  ks_st st = nullptr;                  // owned (EPSGetST)
  ks_bv V = nullptr, W = nullptr;      // basis (ncv+1 columns), work vectors (3 columns)
  int problem_type = 0;               // not set: EPSSetUp picks NHEP (one matrix) or GNHEP (two), epssetup.c:318-322
  int nev = 1, ncv = 0, mpd = 0, ncv_user = 0, mpd_user = 0;
  double tol = 1e-8; int max_it = 0, max_it_user = 0;
  KsCompare which;                     // user settings; cmp_ds / cmp_final are what a solve uses
  KsCompare cmp_ds, cmp_final;
  double keep = 0.5; bool lock = true;   // EPSKrylovSchurSetRestart / SetLocking
  uint64_t seed = 0x12345678ULL;
  std::vector<double> v0; bool have_v0 = false;
  ks_bv defl = nullptr; int nds = 0;       // deflation space handed over by EPSSetDeflationSpace, consumed by the next solve
  long long max_steps = 0;
  int ds_parallel = KS_DS_PARALLEL_SYNCHRONIZED;   // DSSetParallel: broadcast rank 0's projected solve after every restart (krylovschur.c:281)
  // results
  std::vector<double> eigr, eigi, errest; std::vector<int> perm;
  int nconv = 0, its = 0, reason = 0;
  long long steps = 0, passes = 0; int restarts = 0;
  bool solved = false, ghep = false;
  ks_eps_converged_fn conv_fn = nullptr; void *conv_ctx = nullptr;     // EPSSetConvergenceTestFunction (conv = KS_EPS_CONV_USER)
  ks_eps_stopping_fn stop_fn = nullptr; void *stop_ctx = nullptr;      // EPSSetStoppingTestFunction; NULL = EPSStoppingBasic
  ks_eps_monitor_fn mon_fn = nullptr; void *mon_ctx = nullptr;         // EPSMonitorSet (one monitor)
  ks_eps_arbitrary_fn arb_fn = nullptr; void *arb_ctx = nullptr;       // EPSSetArbitrarySelection
  bool problem_type_resolved_hermitian = false;                        // the last solve ran the symmetric (Lanczos) variant
  bool vectors_done = true;                                            // non-symmetric variant: V holds Schur vectors until the eigenvectors are first asked for (EPS_STATE_EIGENVECTORS)
  int cb_err = 0;                                                      // first non-zero return of a user callback
  bool purify = true, trackall = false;                               // EPSSetPurify, EPSSetTrackAll
  bool trueres = false;                                          // EPSSetTrueResidual
  int extraction = KS_EPS_RITZ;                                  // EPSSetExtraction: Ritz or harmonic (krylovschur.c:120)
  int conv = KS_EPS_CONV_REL; double nrma = 0.0, nrmb = 0.0;   // EPSSetConvergenceTest; ||A||_inf, ||B||_inf for CONV_NORM / ERROR_BACKWARD
  DsHep ds;
  DsNhep dsn;
};

extern "C" int ks_eps_create(ks_ctx ctx, ks_eps *out)
{
  KS_CHECK(ctx && out, KS_ERR_ARG_NULL, "ctx/out is NULL");
  ks_eps eps = new ks_eps_s(); eps->ctx = ctx; *out = eps;
  return KS_SUCCESS;
}

extern "C" int ks_eps_destroy(ks_eps eps)
{
  if (!eps) return KS_SUCCESS;
  ks_bv_destroy(eps->V); ks_bv_destroy(eps->W); ks_bv_destroy(eps->defl);
  ks_st_destroy(eps->st);
  if (eps->D) hipFree(eps->D); if (eps->wb) hipFree(eps->wb); if (eps->bal_op) ks_mat_destroy(eps->bal_op);
  delete eps;
  return KS_SUCCESS;
}

extern "C" int ks_eps_set_operators(ks_eps eps, ks_mat A, ks_mat B)   // epssetup.c:450
{
  KS_CHECK(eps && A, KS_ERR_ARG_NULL, "NULL argument");
  KS_CHECK(!B || (B->n == A->n && B->n_global == A->n_global), KS_ERR_ARG_INCOMP, "Mismatching dimensions of A (%d) and B (%d)", A->n, B ? B->n : 0);
  if (eps->A && eps->A->n != A->n) {                                   // EPSReset (epsbasic.c): everything sized by the old operator goes
    ks_bv_destroy(eps->V); ks_bv_destroy(eps->W); eps->V = eps->W = nullptr;
    ks_bv_destroy(eps->defl); eps->defl = nullptr; eps->nds = 0;
    eps->have_v0 = false; eps->v0.clear();
    if (eps->D) { hipFree(eps->D); eps->D = nullptr; } if (eps->wb) { hipFree(eps->wb); eps->wb = nullptr; } eps->D_n = 0;
    if (eps->balance == KS_EPS_BALANCE_USER) eps->balance = KS_EPS_BALANCE_NONE;
  }
  eps->A = A; eps->B = B; eps->solved = false; eps->nrma = eps->nrmb = 0.0;
  if (eps->st) KS_CALL(ks_st_set_matrices(eps->st, A, B));
  return KS_SUCCESS;
}
extern "C" int ks_eps_get_st(ks_eps eps, ks_st *st)                     // EPSGetST epsbasic.c
{
  KS_CHECK(eps && st, KS_ERR_ARG_NULL, "NULL argument");
  if (!eps->st) { KS_CALL(ks_st_create(eps->ctx, &eps->st)); if (eps->A) KS_CALL(ks_st_set_matrices(eps->st, eps->A, eps->B)); }
  *st = eps->st; eps->solved = false;
  return KS_SUCCESS;
}
extern "C" int ks_eps_set_problem_type(ks_eps eps, int type)
{
  KS_CHECK(eps, KS_ERR_ARG_NULL, "EPS is NULL");
  KS_CHECK(type == KS_EPS_HEP || type == KS_EPS_GHEP || type == KS_EPS_NHEP || type == KS_EPS_GNHEP, KS_ERR_SUP, "EPS_HEP and EPS_GHEP (Lanczos), EPS_NHEP and EPS_GNHEP (Arnoldi) are driven by this build; PGNHEP / GHIEP are not");
  eps->problem_type = type; eps->solved = false; return KS_SUCCESS;
}
extern "C" int ks_eps_set_dimensions(ks_eps eps, int nev, int ncv, int mpd)
{
  KS_CHECK(eps, KS_ERR_ARG_NULL, "EPS is NULL");
  KS_CHECK(nev >= 1, KS_ERR_ARG_OUTOFRANGE, "Illegal value of nev. Must be > 0");
  eps->nev = nev; eps->ncv_user = ncv > 0 ? ncv : 0; eps->mpd_user = mpd > 0 ? mpd : 0; eps->solved = false;
  return KS_SUCCESS;
}
extern "C" int ks_eps_set_tolerances(ks_eps eps, double tol, int max_it)
{
  KS_CHECK(eps, KS_ERR_ARG_NULL, "EPS is NULL");
  eps->tol = tol > 0.0 ? tol : 1e-8;                 // SLEPC_DEFAULT_TOL epssetup.c:378, slepcmath.h:25
  eps->max_it_user = max_it > 0 ? max_it : 0;
  return KS_SUCCESS;
}
extern "C" int ks_eps_set_which_eigenpairs(ks_eps eps, int which)
{
  KS_CHECK(eps, KS_ERR_ARG_NULL, "EPS is NULL");
  switch (which) {                                    // epsopts.c:478-510
    case KS_EPS_LARGEST_MAGNITUDE: case KS_EPS_SMALLEST_MAGNITUDE: case KS_EPS_LARGEST_REAL: case KS_EPS_SMALLEST_REAL:
    case KS_EPS_LARGEST_IMAGINARY: case KS_EPS_SMALLEST_IMAGINARY: case KS_EPS_TARGET_MAGNITUDE: case KS_EPS_TARGET_REAL:
    case KS_EPS_WHICH_USER: break;
    default: KS_FAIL(KS_ERR_ARG_OUTOFRANGE, "Invalid 'which' value");
  }
  eps->which.which = which; eps->solved = false; return KS_SUCCESS;
}
extern "C" int ks_eps_set_target(ks_eps eps, double target)                     // EPSSetTarget epsopts.c:604
{
  KS_CHECK(eps, KS_ERR_ARG_NULL, "EPS is NULL");
  eps->which.target = target; eps->solved = false;
  if (eps->st && !eps->st->sigma_set) { eps->st->sigma = target; eps->st->ready = false; }   // STSetDefaultShift (epsbasic.c:386)
  return KS_SUCCESS;
}
extern "C" int ks_eps_set_eigenvalue_comparison(ks_eps eps, ks_eig_compare_fn fn, void *fctx)   // epsopts.c:563
{
  KS_CHECK(eps, KS_ERR_ARG_NULL, "EPS is NULL");
  eps->which.fn = fn; eps->which.fn_ctx = fctx; eps->which.which = KS_EPS_WHICH_USER; eps->solved = false; return KS_SUCCESS;
}
extern "C" int ks_eps_set_krylovschur_restart(ks_eps eps, double keep)   // krylovschur.c:339-350
{
  KS_CHECK(eps, KS_ERR_ARG_NULL, "EPS is NULL");
  KS_CHECK(keep >= 0.1 && keep <= 0.9, KS_ERR_ARG_OUTOFRANGE, "The keep argument %g must be in the range [.1,.9]", keep);
  eps->keep = keep; return KS_SUCCESS;
}
extern "C" int ks_eps_set_convergence_test(ks_eps eps, int conv)            // EPSSetConvergenceTest epsopts.c
{
  KS_CHECK(eps, KS_ERR_ARG_NULL, "EPS is NULL");
  KS_CHECK(conv == KS_EPS_CONV_ABS || conv == KS_EPS_CONV_REL || conv == KS_EPS_CONV_NORM || conv == KS_EPS_CONV_USER, KS_ERR_ARG_OUTOFRANGE, "Invalid 'conv' value");
  KS_CHECK(conv != KS_EPS_CONV_USER || eps->conv_fn, KS_ERR_ORDER, "Must call EPSSetConvergenceTestFunction() first");
  eps->conv = conv; eps->solved = false; return KS_SUCCESS;
}
extern "C" int ks_eps_set_true_residual(ks_eps eps, int trueres)            // EPSSetTrueResidual epsopts.c
{
  KS_CHECK(eps, KS_ERR_ARG_NULL, "EPS is NULL");
  eps->trueres = trueres != 0; eps->solved = false; return KS_SUCCESS;
}
extern "C" int ks_eps_get_true_residual(ks_eps eps, int *trueres) { KS_CHECK(eps && trueres, KS_ERR_ARG_NULL, "NULL argument"); *trueres = eps->trueres ? 1 : 0; return KS_SUCCESS; }
extern "C" int ks_eps_set_purify(ks_eps eps, int purify)                    // EPSSetPurify epsopts.c (generalized symmetric problems)
{
  KS_CHECK(eps, KS_ERR_ARG_NULL, "EPS is NULL");
  eps->purify = purify != 0; eps->solved = false; return KS_SUCCESS;
}
extern "C" int ks_eps_get_purify(ks_eps eps, int *purify) { KS_CHECK(eps && purify, KS_ERR_ARG_NULL, "NULL argument"); *purify = eps->purify ? 1 : 0; return KS_SUCCESS; }
extern "C" int ks_eps_set_track_all(ks_eps eps, int trackall)                // EPSSetTrackAll epsopts.c: residual estimates of all Ritz pairs at every restart
{
  KS_CHECK(eps, KS_ERR_ARG_NULL, "EPS is NULL");
  eps->trackall = trackall != 0; return KS_SUCCESS;
}
extern "C" int ks_eps_get_krylovschur(ks_eps eps, double *keep, int *lock)   // EPSKrylovSchurGetRestart / GetLocking
{
  KS_CHECK(eps, KS_ERR_ARG_NULL, "EPS is NULL");
  if (keep) *keep = eps->keep; if (lock) *lock = eps->lock ? 1 : 0;
  return KS_SUCCESS;
}
extern "C" int ks_eps_set_balance(ks_eps eps, int bal, int its, double cutoff)   // EPSSetBalance epsopts.c:1050-1095 (its, cutoff: 0 keeps)
{
  KS_CHECK(eps, KS_ERR_ARG_NULL, "EPS is NULL");
  KS_CHECK(bal >= KS_EPS_BALANCE_NONE && bal <= KS_EPS_BALANCE_USER, KS_ERR_ARG_OUTOFRANGE, "Invalid value of argument 'bal'");
  KS_CHECK(bal != KS_EPS_BALANCE_USER || eps->D, KS_ERR_ORDER, "EPS_BALANCE_USER: hand the diagonal over first (STSetBalanceMatrix: ks_eps_set_balance_matrix)");
  KS_CHECK(its >= 0, KS_ERR_ARG_OUTOFRANGE, "Illegal value of its. Must be >= 0");
  KS_CHECK(cutoff >= 0.0, KS_ERR_ARG_OUTOFRANGE, "Illegal value of cutoff. Must be >= 0");
  eps->balance = bal; if (its) eps->balance_its = its; if (cutoff > 0.0) eps->balance_cutoff = cutoff;
  eps->solved = false;
  return KS_SUCCESS;
}
// EPS_BALANCE_USER: the diagonal of the balancing matrix comes from the caller (STSetBalanceMatrix stfunc.c on the solver's ST);
// n_local positive doubles on the device, copied. Selects KS_EPS_BALANCE_USER.
extern "C" int ks_eps_set_balance_matrix(ks_eps eps, const double *D_dev)
{
  KS_CHECK(eps && eps->A && D_dev, KS_ERR_ORDER, "set the operators first; D must not be NULL");
  const long long n = eps->A->n;
  KS_HIP(hipSetDevice(eps->ctx->device));
  if (eps->D_n != n || !eps->D) { if (eps->D) hipFree(eps->D); if (eps->wb) hipFree(eps->wb); eps->D = eps->wb = nullptr;
    KS_HIP(hipMalloc(&eps->D, sizeof(double) * std::max<long long>(n, 1))); KS_HIP(hipMalloc(&eps->wb, sizeof(double) * std::max<long long>(n, 1))); eps->D_n = (int)n; }
  KS_HIP(hipMemcpyAsync(eps->D, D_dev, sizeof(double) * n, hipMemcpyDeviceToDevice, eps->ctx->stream));
  KS_HIP(ks_sync(eps->ctx));
  eps->balance = KS_EPS_BALANCE_USER; eps->solved = false;
  return KS_SUCCESS;
}
extern "C" int ks_eps_get_balance(ks_eps eps, int *bal, int *its, double *cutoff)
{
  KS_CHECK(eps, KS_ERR_ARG_NULL, "EPS is NULL");
  if (bal) *bal = eps->balance; if (its) *its = eps->balance_its; if (cutoff) *cutoff = eps->balance_cutoff;
  return KS_SUCCESS;
}
extern "C" int ks_eps_set_extraction(ks_eps eps, int extr)                  // EPSSetExtraction epsopts.c:968-994; krylovschur.c:120 accepts these two
{
  KS_CHECK(eps, KS_ERR_ARG_NULL, "EPS is NULL");
  KS_CHECK(extr >= KS_EPS_RITZ && extr <= KS_EPS_REFINED_HARMONIC, KS_ERR_ARG_OUTOFRANGE, "Invalid extraction type");
  KS_CHECK(extr == KS_EPS_RITZ || extr == KS_EPS_HARMONIC, KS_ERR_SUP, "Unsupported extraction type");
  eps->extraction = extr; eps->solved = false; return KS_SUCCESS;
}
extern "C" int ks_eps_get_extraction(ks_eps eps, int *extr) { KS_CHECK(eps && extr, KS_ERR_ARG_NULL, "NULL argument"); *extr = eps->extraction; return KS_SUCCESS; }
static int matrix_norms(ks_eps eps)                                         // epssetup.c:345-358
{
  if (!eps->nrma) KS_CALL(ks_mat_norm_inf(eps->A, &eps->nrma));
  if (eps->B) { if (!eps->nrmb) KS_CALL(ks_mat_norm_inf(eps->B, &eps->nrmb)); } else eps->nrmb = 1.0;
  return KS_SUCCESS;
}
extern "C" int ks_eps_set_krylovschur_locking(ks_eps eps, int lock)     // EPSKrylovSchurSetLocking krylovschur.c:388-423
{
  KS_CHECK(eps, KS_ERR_ARG_NULL, "EPS is NULL");
  eps->lock = lock != 0; eps->solved = false; return KS_SUCCESS;
}
extern "C" int ks_eps_set_random_seed(ks_eps eps, uint64_t seed) { KS_CHECK(eps, KS_ERR_ARG_NULL, "EPS is NULL"); eps->seed = seed; return KS_SUCCESS; }
extern "C" int ks_eps_set_initial_vector(ks_eps eps, const double *v)
{
  KS_CHECK(eps && eps->A, KS_ERR_ORDER, "set the operators first");
  if (!v) { eps->have_v0 = false; return KS_SUCCESS; }
  eps->v0.assign(v, v + eps->A->n); eps->have_v0 = true;
  return KS_SUCCESS;
}
// EPSSetDeflationSpace epssetup.c:555-570: the vectors are copied now and become constraints of the basis at the next
// solve (BVInsertConstraints, epssetup.c:397-404), which also forgets them (epssolve.c:201-205): "the deflation space
// should be set every time". They need not be orthonormal; dependent ones are dropped.
extern "C" int ks_eps_set_deflation_space(ks_eps eps, int n, const double *const *v_dev)
{
  KS_CHECK(eps && eps->A, KS_ERR_ORDER, "set the operators first");
  KS_CHECK(n >= 0, KS_ERR_ARG_OUTOFRANGE, "Argument n cannot be negative");
  ks_bv_destroy(eps->defl); eps->defl = nullptr; eps->nds = 0;
  if (!n) return KS_SUCCESS;
  KS_CHECK(v_dev, KS_ERR_ARG_NULL, "NULL argument");
  KS_CALL(ks_bv_create(eps->ctx, eps->A->n, eps->A->n_global, n, 0, &eps->defl));
  for (int i = 0; i < n; i++) { KS_CHECK(v_dev[i], KS_ERR_ARG_NULL, "vector %d is NULL", i); KS_CALL(ks_bv_insert_vec(eps->defl, i, v_dev[i])); }
  eps->nds = n; eps->solved = false;
  return KS_SUCCESS;
}
// EPSSetInitialSpace epssetup.c:590-610 with device vectors. A Krylov solver starts from ONE vector: the first of the
// space (EPSGetStartVector epssolve.c:853 uses column 0 of the inserted, orthonormalised set), the others are not used.
extern "C" int ks_eps_set_initial_space(ks_eps eps, int n, const double *const *v_dev)
{
  KS_CHECK(eps && eps->A, KS_ERR_ORDER, "set the operators first");
  KS_CHECK(n >= 0, KS_ERR_ARG_OUTOFRANGE, "Argument n cannot be negative");
  if (!n) { eps->have_v0 = false; return KS_SUCCESS; }
  KS_CHECK(v_dev && v_dev[0], KS_ERR_ARG_NULL, "NULL argument");
  eps->v0.resize(std::max(eps->A->n, 1));
  KS_HIP(hipSetDevice(eps->ctx->device));
  KS_HIP(hipMemcpyAsync(eps->v0.data(), v_dev[0], sizeof(double) * eps->A->n, hipMemcpyDeviceToHost, eps->ctx->stream));
  KS_HIP(ks_sync(eps->ctx));
  eps->have_v0 = true; eps->solved = false;
  return KS_SUCCESS;
}
extern "C" int ks_eps_set_ds_parallel(ks_eps eps, int pmode)
{
  KS_CHECK(eps, KS_ERR_ARG_NULL, "EPS is NULL");
  KS_CHECK(pmode == KS_DS_PARALLEL_REDUNDANT || pmode == KS_DS_PARALLEL_SYNCHRONIZED, KS_ERR_ARG_OUTOFRANGE, "unknown DS parallel mode %d", pmode);
  eps->ds_parallel = pmode;
  return KS_SUCCESS;
}
extern "C" int ks_eps_get_ds_parallel(ks_eps eps, int *pmode) { KS_CHECK(eps && pmode, KS_ERR_ARG_NULL, "NULL argument"); *pmode = eps->ds_parallel; return KS_SUCCESS; }

// DSSynchronize (dsops.c:875 -> DSSynchronize_HEP dshep.c:673-713, DSSynchronize_NHEP dsnhep.c:379): every rank leaves with rank 0's
// projected matrix, vectors and eigenvalues. The scalars the convergence test and the restart sizes are computed from (beta of the
// expansion, its length, the breakdown flag) travel in the same message, so the integer control flow that follows is identical on
// all ranks whatever the allreduce provider returned.
static int ds_synchronize(ks_eps eps, std::vector<double> *M1, std::vector<double> *M2, double *beta, int *nv, int *breakdown)
{
  ks_ctx ctx = eps->ctx;
  if (!ks_is_multi(ctx) || eps->ds_parallel != KS_DS_PARALLEL_SYNCHRONIZED) return KS_SUCCESS;      // (the force_multi test hook issues it on one rank too)
  std::vector<double> pack;
  pack.reserve(M1->size() + M2->size() + 2 * eps->eigr.size() + 3);
  pack.insert(pack.end(), M1->begin(), M1->end());
  pack.insert(pack.end(), M2->begin(), M2->end());
  pack.insert(pack.end(), eps->eigr.begin(), eps->eigr.end());
  pack.insert(pack.end(), eps->eigi.begin(), eps->eigi.end());
  pack.push_back(*beta); pack.push_back((double)*nv); pack.push_back((double)*breakdown);
  KS_CALL(ks_comm_bcast0_host(ctx, pack.data(), (int)(pack.size() * sizeof(double))));
  size_t o = 0;
  std::copy(pack.begin() + o, pack.begin() + o + M1->size(), M1->begin()); o += M1->size();
  std::copy(pack.begin() + o, pack.begin() + o + M2->size(), M2->begin()); o += M2->size();
  std::copy(pack.begin() + o, pack.begin() + o + eps->eigr.size(), eps->eigr.begin()); o += eps->eigr.size();
  std::copy(pack.begin() + o, pack.begin() + o + eps->eigi.size(), eps->eigi.begin()); o += eps->eigi.size();
  *beta = pack[o]; *nv = (int)pack[o + 1]; *breakdown = (int)pack[o + 2];
  return KS_SUCCESS;
}

extern "C" int ks_eps_set_max_steps(ks_eps eps, long long s) { KS_CHECK(eps, KS_ERR_ARG_NULL, "EPS is NULL"); eps->max_steps = s > 0 ? s : 0; return KS_SUCCESS; }

// EPSStoppingBasic epsdefault.c:290-307; user functions may call it first, as ex29.c does
extern "C" int ks_eps_stopping_basic(ks_eps eps, int its, int max_it, int nconv, int nev, int *reason, void *ctx)
{
  (void)eps; (void)ctx;
  KS_CHECK(reason, KS_ERR_ARG_NULL, "NULL argument");
  *reason = KS_EPS_CONVERGED_ITERATING;
  if (nconv >= nev) *reason = KS_EPS_CONVERGED_TOL;
  else if (its >= max_it) *reason = KS_EPS_DIVERGED_ITS;
  return KS_SUCCESS;
}
extern "C" int ks_eps_set_stopping_test_function(ks_eps eps, ks_eps_stopping_fn fn, void *ctx)   // EPSSetStoppingTestFunction epsopts.c
{
  KS_CHECK(eps, KS_ERR_ARG_NULL, "EPS is NULL");
  eps->stop_fn = fn; eps->stop_ctx = ctx; return KS_SUCCESS;
}
extern "C" int ks_eps_set_convergence_test_function(ks_eps eps, ks_eps_converged_fn fn, void *ctx)   // EPSSetConvergenceTestFunction epsopts.c
{
  KS_CHECK(eps, KS_ERR_ARG_NULL, "EPS is NULL");
  if (fn) { eps->conv_fn = fn; eps->conv_ctx = ctx; eps->conv = KS_EPS_CONV_USER; }
  else { eps->conv_fn = nullptr; eps->conv_ctx = nullptr; if (eps->conv == KS_EPS_CONV_USER) eps->conv = KS_EPS_CONV_REL; }
  eps->solved = false; return KS_SUCCESS;
}
extern "C" int ks_eps_set_arbitrary_selection(ks_eps eps, ks_eps_arbitrary_fn fn, void *ctx)      // EPSSetArbitrarySelection epsopts.c:600-615
{
  KS_CHECK(eps, KS_ERR_ARG_NULL, "EPS is NULL");
  eps->arb_fn = fn; eps->arb_ctx = ctx; eps->solved = false; return KS_SUCCESS;
}
extern "C" int ks_eps_monitor_set(ks_eps eps, ks_eps_monitor_fn fn, void *ctx)                   // EPSMonitorSet epsmon.c (one slot); NULL = EPSMonitorCancel
{
  KS_CHECK(eps, KS_ERR_ARG_NULL, "EPS is NULL");
  eps->mon_fn = fn; eps->mon_ctx = ctx; return KS_SUCCESS;
}
// the stopping test of one restart: user function or the basic one, then the step cap of the bench harness
static int stopping_test(ks_eps eps, int k)
{
  int reason = KS_EPS_CONVERGED_ITERATING;
  if (eps->stop_fn) { const int rc = eps->stop_fn(eps, eps->its, eps->max_it, k, eps->nev, &reason, eps->stop_ctx); KS_CHECK(!rc, rc, "the user's stopping test returned %d", rc); }
  else ks_eps_stopping_basic(eps, eps->its, eps->max_it, k, eps->nev, &reason, nullptr);
  eps->reason = reason;
  if (eps->reason == KS_EPS_CONVERGED_ITERATING && eps->max_steps && eps->steps >= eps->max_steps) eps->reason = KS_EPS_CONVERGED_USER;
  return KS_SUCCESS;
}
static int monitor(ks_eps eps, int nconv, int nest)                                            // EPSMonitor epsmon.c:21-33
{
  if (!eps->mon_fn) return KS_SUCCESS;
  const int rc = eps->mon_fn(eps, eps->its, nconv, eps->eigr.data(), eps->eigi.data(), eps->errest.data(), nest, eps->mon_ctx);
  KS_CHECK(!rc, rc, "the user's monitor returned %d", rc);
  return KS_SUCCESS;
}

// EPSConvergedRelative / Absolute / Norm epsdefault.c:224-257
static double converged_estimate(ks_eps eps, double re, double im, double res)
{
  const double w = hypot(re, im);
  switch (eps->conv) {
    case KS_EPS_CONV_USER: { double e = 0.0; const int rc = eps->conv_fn(eps, re, im, res, &e, eps->conv_ctx); if (rc && !eps->cb_err) eps->cb_err = rc; return e; }
    case KS_EPS_CONV_ABS:  return res;
    case KS_EPS_CONV_NORM: return res / (eps->nrma + w * eps->nrmb);
    default:               return (w != 0.0) ? res / w : std::numeric_limits<double>::max();
  }
}

namespace {
__global__ void k_pw(long long n, const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ out, int mul)
{
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) out[i] = mul ? a[i] * b[i] : a[i] / b[i];
}
__global__ void k_sign_half(long long n, double *__restrict__ z)
{
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) z[i] = z[i] < 0.5 ? -1.0 : 1.0;
}
__global__ void k_bal_update(long long n, double *__restrict__ D, const double *__restrict__ p)
{
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) if (p[i] != 0.0) D[i] /= fabs(p[i]);
}
// two-sided form (epsdefault.c:419-421): D_i *= sqrt(|r_i / p_i|) where |p_i| > cutoff * norma and r_i != 0
__global__ void k_bal_update2(long long n, double *__restrict__ D, const double *__restrict__ p, const double *__restrict__ r, double thr)
{
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    if (fabs(p[i]) > thr && r[i] != 0.0) D[i] *= sqrt(fabs(r[i] / p[i]));
}
__global__ void k_fill(long long n, double *__restrict__ x, double v)
{
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) x[i] = v;
}
}
static int pointwise(ks_eps eps, const double *a, const double *b, double *out, bool mul)     // VecPointwiseMult / VecPointwiseDivide
{
  const long long n = eps->V->n;
  if (!n) return KS_SUCCESS;
  const unsigned nb = (unsigned)std::max<long long>(1, std::min<long long>((n + 255) / 256, (long long)eps->ctx->num_cu * 16));
  hipLaunchKernelGGL(k_pw, dim3(nb), dim3(256), 0, eps->ctx->stream, n, a, b, out, mul ? 1 : 0);
  KS_HIP(hipGetLastError());
  return KS_SUCCESS;
}
// the balanced operator D Op D^-1 as a matrix-free operator (STApply with st->D, stsolve.c:252-256)
static int balanced_mult(void *user, const double *x, double *y)
{
  ks_eps eps = (ks_eps)user;
  KS_CALL(pointwise(eps, x, eps->D, eps->wb, false));
  KS_CALL(ks_mat_mult_internal(eps->op_inner, eps->wb, y));
  return pointwise(eps, y, eps->D, y, true);
}
// EPSBuildBalance_Krylov epsdefault.c:370-434 over balance_its random +-1 vectors z. One-sided: D <- D ./ |p|, p = D Op D^-1 z. Two-sided: also
// r = D^-1 Op^T D z (STApplyHermitianTranspose: ks_mat_mult_transpose on the operator), D_i *= sqrt(|r_i / p_i|) where |p_i| exceeds cutoff times the
// infinity norm of the first p
static int build_balance(ks_eps eps)
{
  ks_ctx ctx = eps->ctx; const long long n = eps->V->n;
  const unsigned nb = (unsigned)std::max<long long>(1, std::min<long long>((n + 255) / 256, (long long)ctx->num_cu * 16));
  if (eps->D_n != n) { if (eps->D) hipFree(eps->D); if (eps->wb) hipFree(eps->wb); eps->D = eps->wb = nullptr;
    KS_HIP(hipMalloc(&eps->D, sizeof(double) * std::max<long long>(n, 1))); KS_HIP(hipMalloc(&eps->wb, sizeof(double) * std::max<long long>(n, 1))); eps->D_n = (int)n; }
  if (n) hipLaunchKernelGGL(k_fill, dim3(nb), dim3(256), 0, ctx->stream, n, eps->D, 1.0);
  double *z = ks_bv_col(eps->W, 3), *p = ks_bv_col(eps->W, 4);
  eps->W->row_start = eps->V->row_start;
  double norma = 0.0;
  for (int j = 0; j < eps->balance_its; j++) {
    KS_CALL(ks_bv_set_random_column(eps->W, 3, eps->seed + 7919ULL * (uint64_t)(j + 1)));           // a random vector of +-1's
    if (n) hipLaunchKernelGGL(k_sign_half, dim3(nb), dim3(256), 0, ctx->stream, n, z);
    KS_CALL(balanced_mult(eps, z, p));                                                               // p = D Op (D \ z)
    if (eps->balance == KS_EPS_BALANCE_TWOSIDE) {
      if (j == 0) KS_CALL(ks_bv_normcolumn(eps->W, 4, KS_NORM_INFINITY, &norma));                    // VecAbs + VecMax: the estimate of the matrix infinity norm
      KS_CALL(pointwise(eps, z, eps->D, z, true));                                                    // r = D \ (Op' (D z))
      KS_CALL(ks_mat_mult_transpose_internal(eps->op_inner, z, eps->wb));
      KS_CALL(pointwise(eps, eps->wb, eps->D, eps->wb, false));
      if (n) hipLaunchKernelGGL(k_bal_update2, dim3(nb), dim3(256), 0, ctx->stream, n, eps->D, p, eps->wb, eps->balance_cutoff * norma);
    } else if (n) hipLaunchKernelGGL(k_bal_update, dim3(nb), dim3(256), 0, ctx->stream, n, eps->D, p);
    KS_HIP(hipGetLastError());
  }
  return KS_SUCCESS;
}

// EPSComputeResidualNorm_Private epssolve.c:666-718 (STGetMatrix 0/1 = the user's A and B): || A x - k B x ||_2 for a
// real eigenvalue, hypot of the two real-arithmetic residuals for a pair (xi_sign * xi is the imaginary part).
// Work vectors: W columns 0..2.
static int residual_norm(ks_eps eps, double kr, double ki, const double *xr, const double *xi, double xi_sign, double *out)
{
  ks_bv W = eps->W; ks_ctx ctx = eps->ctx; ks_mat A = eps->A, B = eps->B;
  const long long n = W->n;
  double *u = ks_bv_col(W, 0);
  double nrm = 0.0;
  if (ki == 0.0 || fabs(ki) < fabs(kr * std::numeric_limits<double>::epsilon())) {
    KS_CALL(ks_mat_mult_internal(A, xr, u));                                        // u = A*x
    if (fabs(kr) > std::numeric_limits<double>::epsilon()) {
      const double *w = xr;
      if (B) { KS_CALL(ks_mat_mult_internal(B, xr, ks_bv_col(W, 2))); w = ks_bv_col(W, 2); }   // w = B*x
      KS_CALL(ksk_lincomb(ctx, n, nullptr, 1.0, u, -kr, w, u));                      // u = A*x - k*B*x
    }
    KS_CALL(ks_bv_normcolumn(W, 0, KS_NORM_2, &nrm));
  } else {
    const double sg = xi_sign;
    const double *v = xr, *w = xi;                                                   // v = B*xr, w = B*xi (before the sign)
    if (B) { KS_CALL(ks_mat_mult_internal(B, xr, ks_bv_col(W, 1))); KS_CALL(ks_mat_mult_internal(B, xi, ks_bv_col(W, 2))); v = ks_bv_col(W, 1); w = ks_bv_col(W, 2); }
    double nr = 0.0, ni = 0.0;
    KS_CALL(ks_mat_mult_internal(A, xr, u));                                        // u = A*xr - kr*B*xr + ki*B*xi
    KS_CALL(ksk_lincomb(ctx, n, nullptr, 1.0, u, -kr, v, u));
    KS_CALL(ksk_lincomb(ctx, n, nullptr, 1.0, u, ki * sg, w, u));
    KS_CALL(ks_bv_normcolumn(W, 0, KS_NORM_2, &nr));
    KS_CALL(ks_mat_mult_internal(A, xi, u));                                        // u = A*xi - kr*B*xi - ki*B*xr
    KS_CALL(ksk_lincomb(ctx, n, nullptr, sg, u, -kr * sg, w, u));
    KS_CALL(ksk_lincomb(ctx, n, nullptr, 1.0, u, -ki, v, u));
    KS_CALL(ks_bv_normcolumn(W, 0, KS_NORM_2, &ni));
    nrm = hypot(nr, ni);
  }
  *out = nrm;
  return KS_SUCCESS;
}

// EPSComputeRitzVector epsdefault.c:313-364 followed by the residual of epskrylov.c:256-264 (-eps_true_residual):
// x = V(:,0:nv) Zr [, y = V(:,0:nv) Zi], purified through the operator for a GHEP, into W columns 3 and 4.
static int ritz_vector(ks_eps eps, int nv, const double *Zr, const double *Zi);
static int true_residual(ks_eps eps, int nv, double re, double im, const double *Zr, const double *Zi, double *resnorm)
{
  KS_CALL(ritz_vector(eps, nv, Zr, Zi));
  return residual_norm(eps, re, im, ks_bv_col(eps->W, 3), Zi ? ks_bv_col(eps->W, 4) : nullptr, 1.0, resnorm);
}
// EPSComputeRitzVector epsdefault.c:313-364: x (W column 3) and, for a pair, y (W column 4)
static int ritz_vector(ks_eps eps, int nv, const double *Zr, const double *Zi)
{
  ks_bv V = eps->V, W = eps->W;
  int ls = 0, ksv = 0;
  KS_CALL(ks_bv_get_active_columns(V, &ls, &ksv));
  KS_CALL(ks_bv_set_active_columns(V, 0, nv));
  double *x = ks_bv_col(W, 3), *y = ks_bv_col(W, 4);
  KS_CALL(ks_bv_multvec(V, 1.0, 0.0, x, Zr));
  if (eps->ghep && eps->purify) {                                                    // eps->purify (epssetup.c:365-373)
    double norm = 0.0;
    KS_CALL(ks_mat_mult_internal(eps->op, x, y));
    KS_CALL(ksb_norm_b(V, y, &norm));
    KS_CALL(ksk_scale(eps->ctx, y, V->n, 1.0 / norm));
    KS_CALL(ksk_copy(eps->ctx, y, x, V->n));
  }
  if (Zi) KS_CALL(ks_bv_multvec(V, 1.0, 0.0, y, Zi));
  else KS_HIP(hipMemsetAsync(y, 0, sizeof(double) * V->n, eps->ctx->stream));   // VecSet(y,0.0) epsdefault.c:352
  KS_CALL(ks_bv_set_active_columns(V, ls, ksv));
  if (eps->balanced) {                                      // fix and normalise the eigenvector when balancing is used (epsdefault.c:336,349,355-361)
    KS_CALL(pointwise(eps, x, eps->D, x, false));
    if (Zi) KS_CALL(pointwise(eps, y, eps->D, y, false));
    double nx = 0.0, ny = 0.0;
    KS_CALL(ks_bv_normvec(W, x, KS_NORM_2, &nx));
    if (Zi) KS_CALL(ks_bv_normvec(W, y, KS_NORM_2, &ny));
    const double nrm = hypot(nx, ny);
    if (nrm != 0.0) { KS_CALL(ksk_scale(eps->ctx, x, V->n, 1.0 / nrm)); if (Zi) KS_CALL(ksk_scale(eps->ctx, y, V->n, 1.0 / nrm)); }
  }
  return KS_SUCCESS;
}

// EPSGetStartVector epssolve.c:841-873
static int start_vector(ks_eps eps, int i, bool *breakdown)
{
  if (i == 0 && eps->have_v0) KS_CALL(ks_bv_set_column_host(eps->V, 0, eps->v0.data()));
  else KS_CALL(ks_bv_set_random_column(eps->V, i, eps->seed));
  if (eps->ghep) {                                   // force the vector to be in the range of OP (epssolve.c:860-868)
    KS_CALL(ksk_copy(eps->ctx, ks_bv_col(eps->V, i), ks_bv_col(eps->W, 0), eps->V->n));
    KS_CALL(ks_mat_mult_internal(eps->op, ks_bv_col(eps->W, 0), ks_bv_col(eps->V, i)));
  }
  double norm = 0.0; int lindep = 0;
  KS_CALL(ks_bv_orthogonalizecolumn(eps->V, i, nullptr, &norm, &lindep));
  if (breakdown) *breakdown = lindep != 0;
  else if (lindep || norm == 0.0) {
    if (i == 0) KS_FAIL(KS_ERR_PLIB, "Initial vector is zero or belongs to the deflation space");
    KS_FAIL(KS_ERR_CONV_FAILED, "Unable to generate more start vectors");
  }
  KS_CALL(ks_bv_scalecolumn(eps->V, i, 1.0 / norm));
  return KS_SUCCESS;
}

// EPSComputeVectors_Schur epsdefault.c:105-169, run on first use (EPSComputeVectors epssolve.c:... state EPS_STATE_EIGENVECTORS):
// X = V*Z with Z the normalised eigenvectors of the trimmed quasi-triangular T. Until then V(:,0:nconv) is the orthonormal
// Schur basis that EPSGetInvariantSubspace hands out.
static int compute_vectors(ks_eps eps)
{
  if (eps->vectors_done) return KS_SUCCESS;
  ks_bv V = eps->V; DsNhep &ds = eps->dsn;
  const int nc = eps->nconv;
  KS_CALL(ks_bv_set_active_columns(V, 0, nc));
  if (nc) {
    for (int k = 0; k < nc; k++) k = ds.vectors(k, false, nullptr);
    KS_CALL(ks_bv_multinplace(V, ds.X.data(), ds.ld, 0, nc));
    if (eps->balanced) {                                    // epsdefault.c:130-139: x <- D \ x, then normalise (pairs together)
      for (int i = 0; i < nc; i++) KS_CALL(pointwise(eps, ks_bv_col(V, i), eps->D, ks_bv_col(V, i), false));
      KS_CALL(ks_bv_normalize(V, eps->eigi.data()));
    }
  }
  eps->vectors_done = true;
  return KS_SUCCESS;
}

// Non-Hermitian branch of EPSSolve_KrylovSchur_Default (krylovschur.c:227-337 with BVMatArnoldi), the conjugate-pair
// handling of EPSKrylovConvergence (epskrylov.c:262-287), EPSComputeVectors_Schur (epsdefault.c:105-169) and the
// pair-aware SlepcSortEigenvalues (slepcsc.c:89-140).
static int solve_nhep(ks_eps eps, long long passes0)
{
  ks_mat A = eps->op; ks_bv V = eps->V;
  const int nev = eps->nev, ncv = eps->ncv, mpd = eps->mpd;
  ks_st map = eps->cmp_ds.map;
  const bool isshift = !map || map->type == KS_ST_SHIFT;
  DsNhep &ds = eps->dsn;
  ds.allocate(ncv + 1); ds.which = eps->cmp_ds; ds.state = DS_RAW;
  const bool harmonic = eps->extraction == KS_EPS_HARMONIC;
  std::vector<double> g(harmonic ? ncv + 1 : 0);
  KS_CALL(start_vector(eps, 0, nullptr));
  int l = 0;
  while (eps->reason == KS_EPS_CONVERGED_ITERATING) {
    eps->its++;
    int nv = std::min(eps->nconv + mpd, ncv);
    if (eps->max_steps && eps->steps + (nv - (eps->nconv + l)) > eps->max_steps) nv = eps->nconv + l + (int)(eps->max_steps - eps->steps);
    ds.set_dimensions(nv, eps->nconv, eps->nconv + l);
    double beta = 0.0; int breakdown = 0;
    const int k0 = eps->nconv + l;
    KS_CALL(ks_bv_matarnoldi(V, A, ds.A.data(), ds.ld, k0, &nv, &beta, &breakdown));
    eps->steps += nv - k0;
    ds.set_dimensions(nv, eps->nconv, eps->nconv + l);
    ds.state = l ? DS_RAW : DS_INTERMEDIATE;
    KS_CALL(ks_bv_set_active_columns(V, eps->nconv, nv));

    // translation of the Krylov decomposition for harmonic extraction (krylovschur.c:270-271)
    double gamma = 1.0;
    if (harmonic) KS_CHECK(!ds.translate_harmonic(eps->which.target, beta, false, g.data(), &gamma), KS_ERR_LIB, "harmonic extraction: H - target*I is singular");

    int info = ds.solve(eps->eigr.data(), eps->eigi.data());
    KS_CHECK(info == 0, KS_ERR_LIB, "Hessenberg QR iteration failed to converge (info=%d)", info);
    info = ds.sort(eps->eigr.data(), eps->eigi.data());
    KS_CHECK(info == 0, KS_ERR_LIB, "reordering of the Schur form failed: blocks too close to swap");
    ds.update_extra_row();
    KS_CALL(ds_synchronize(eps, &ds.A, &ds.Q, &beta, &nv, &breakdown));      // krylovschur.c:281

    // EPSKrylovConvergence(eps,FALSE,nconv,nv-nconv,beta,0.0,1.0,&k)
    int marker = -1, k;
    for (k = eps->nconv; k < nv; k++) {
      double re = eps->eigr[k], im = eps->eigi[k];
      if ((isshift || eps->conv == KS_EPS_CONV_NORM) && map) ks_st_backtransform_internal(map, 1, &re, &im);          // epskrylov.c:253
      double resnorm = 0.0;
      const int newk = ds.vectors(k, true, &resnorm);
      if (eps->trueres) {                                      // epskrylov.c:256-264
        if (!((isshift || eps->conv == KS_EPS_CONV_NORM) && map) && map) ks_st_backtransform_internal(map, 1, &re, &im);
        KS_CALL(true_residual(eps, nv, re, im, ds.X.data() + (size_t)k * ds.ld, newk == k + 1 ? ds.X.data() + (size_t)newk * ds.ld : nullptr, &resnorm));
      } else
      resnorm *= beta * gamma;                                 // corrf: only in harmonic KS (epskrylov.c:265)
      eps->errest[k] = converged_estimate(eps, re, im, resnorm);
      if (marker == -1 && eps->errest[k] >= eps->tol) marker = k;
      if (newk == k + 1) { eps->errest[k + 1] = eps->errest[k]; k++; }
      if (marker != -1 && !eps->trackall) break;               // getall: estimates for every Ritz pair (epskrylov.c:240,280)
    }
    k = (marker != -1) ? marker : nv;
    KS_CHECK(!eps->cb_err, eps->cb_err, "the user's convergence test returned %d", eps->cb_err);
    KS_CALL(stopping_test(eps, k));
    const int nconv_mon = k;

    if (eps->reason != KS_EPS_CONVERGED_ITERATING || breakdown || k == nv) l = 0;
    else {
      l = std::max(1, (int)((nv - k) * eps->keep));
      l = ds.get_truncate_size(k, nv, l);                      // do not split a 2x2 block (krylovschur.c:300)
    }
    if (!eps->lock && l > 0) { l += k; k = 0; }                // non-locking variant (krylovschur.c:294)
    if (eps->reason == KS_EPS_CONVERGED_ITERATING) {
      if (breakdown || k == nv) {
        if (k < nev) {
          bool brk = false;
          KS_CALL(start_vector(eps, k, &brk));
          if (brk) eps->reason = KS_EPS_DIVERGED_BREAKDOWN;
        }
      } else {
        if (harmonic) {                                        // undo the translation (krylovschur.c:310-320): gamma u^ = u - U g~
          ds.set_dimensions(nv, k, l);
          ds.translate_harmonic(0.0, beta, true, g.data(), &gamma);
          KS_CALL(ks_bv_set_active_columns(V, 0, nv));
          KS_CALL(ks_bv_multcolumn(V, -1.0, 1.0, nv, g.data()));
          KS_CALL(ks_bv_scalecolumn(V, nv, 1.0 / gamma));
          KS_CALL(ks_bv_set_active_columns(V, eps->nconv, nv));
          ds.set_dimensions(nv, k, nv);
        }
        ds.truncate(k + l, false);
      }
    }
    KS_CALL(ks_bv_multinplace(V, ds.Q.data(), ds.ld, eps->nconv, k + l));
    if (eps->reason == KS_EPS_CONVERGED_ITERATING && !breakdown) KS_CALL(ks_bv_copycolumn(V, nv, k + l));
    eps->nconv = k;
    KS_CALL(monitor(eps, nconv_mon, nv));
    eps->restarts++;
  }
  ds.truncate(eps->nconv, true);

  // the eigenvectors are formed on first use (compute_vectors); V(:,0:nconv) stays the Schur basis until then
  const int nc = eps->nconv;
  KS_CALL(ks_bv_set_active_columns(V, 0, nc));
  eps->vectors_done = false;
  // EPSComputeValues (epssolve.c:27-41), then conjugate pairs with the positive imaginary part first (:160-175):
  // the inversion of sinvert flips the sign
  if (map) ks_st_backtransform_internal(map, nc, eps->eigr.data(), eps->eigi.data());
  for (int i = 0; i < nc - 1; i++) {
    if (eps->eigi[i] != 0.0) {
      if (eps->eigi[i] < 0.0) {                                 // "the next correction only works with eigenvectors" (epssolve.c:166-169)
        eps->eigi[i] = -eps->eigi[i]; eps->eigi[i + 1] = -eps->eigi[i + 1];
        KS_CALL(compute_vectors(eps));
        KS_CALL(ks_bv_scalecolumn(V, i + 1, -1.0));
      }
      i++;
    }
  }
  // SlepcSortEigenvalues keeping conjugate pairs together
  std::vector<int> &perm = eps->perm;
  const double *eigr = eps->eigr.data(), *eigi = eps->eigi.data();
  for (int i = 0; i <= ncv; i++) perm[i] = i;
  for (int i = nc - 1; i >= 0; i--) {
    const double re = eigr[perm[i]]; double im = eigi[perm[i]];
    int j = i + 1;
    if (im != 0.0) { i--; im = eigi[perm[i]]; }                // complex eigenvalue: positive imaginary part first
    while (j < nc) {
      if (compare_eig(eps->cmp_final, re, im, eigr[perm[j]], eigi[perm[j]]) <= 0) break;
      if (im == 0.0) {
        if (eigi[perm[j]] == 0.0) { std::swap(perm[j - 1], perm[j]); j++; }
        else { const int tmp = perm[j - 1]; perm[j - 1] = perm[j]; perm[j] = perm[j + 1]; perm[j + 1] = tmp; j += 2; }
      } else {
        if (eigi[perm[j]] == 0.0) { const int tmp = perm[j - 2]; perm[j - 2] = perm[j]; perm[j] = perm[j - 1]; perm[j - 1] = tmp; j++; }
        else { std::swap(perm[j - 2], perm[j]); std::swap(perm[j - 1], perm[j + 1]); j += 2; }
      }
    }
  }
  long long passes1 = 0; ks_bv_gs_passes(V, &passes1, nullptr);
  eps->passes = passes1 - passes0;
  KS_CALL(ks_bv_set_num_constraints(V, 0));                            // remove the deflation space (epssolve.c:201-205)
  eps->solved = true;
  return KS_SUCCESS;
}

extern "C" int ks_eps_solve(ks_eps eps)   // EPSSolve epssolve.c:119 -> EPSSolve_KrylovSchur_Default krylovschur.c:227
{
  KS_CHECK(eps && eps->A, KS_ERR_ORDER, "EPSSetOperators must be called first");
  ks_mat A = eps->A;
  const int n = A->n_global;
  // ---- EPSSetUp (epssetup.c:286-420) ----
  KS_CHECK(eps->which.which != KS_EPS_WHICH_USER || eps->which.fn, KS_ERR_ORDER, "Must call EPSSetEigenvalueComparison() first");   // epssetup.c:311
  int ptype = eps->problem_type;
  if (!ptype) ptype = eps->B ? KS_EPS_GNHEP : KS_EPS_NHEP;             // default problem type (epssetup.c:318-322)
  if (!eps->B && ptype == KS_EPS_GNHEP) ptype = KS_EPS_NHEP;          // "reverting to a standard eigenproblem" (epssetup.c:324-327)
  if (!eps->B && ptype == KS_EPS_GHEP) ptype = KS_EPS_HEP;
  KS_CHECK(!eps->B || ptype == KS_EPS_GNHEP || ptype == KS_EPS_GHEP, KS_ERR_ARG_INCOMP, "Inconsistent EPS state: the problem type does not match the number of matrices");
  const bool ghep = ptype == KS_EPS_GHEP;
  if (eps->conv == KS_EPS_CONV_NORM) KS_CALL(matrix_norms(eps));
  ks_st st = eps->st;
  const bool cayley = st && st->type == KS_ST_CAYLEY;
  const bool sinvert = (st && st->type == KS_ST_SINVERT) || cayley;   // the set-up rules below are those of EPSCheckSinvertCayley
  if (sinvert && !st->sigma_set) { if (st->sigma != eps->which.target) st->ready = false; st->sigma = eps->which.target; }   // the shift of sinvert defaults to the target (STSetDefaultShift epsbasic.c:386, sinvert.c:64); STSHIFT keeps 0
  KsCompare cmp = eps->which;
  if (!cmp.which) cmp.which = sinvert ? KS_EPS_TARGET_MAGNITUDE : KS_EPS_LARGEST_MAGNITUDE;   // epsdefault.c:209-219
  KS_CHECK(!sinvert || cmp.which == KS_EPS_TARGET_MAGNITUDE || cmp.which == KS_EPS_TARGET_REAL || cmp.which == KS_EPS_WHICH_USER, KS_ERR_USER_INPUT,
           "Shift-and-invert requires a target 'which' (see EPSSetWhichEigenpairs), for instance -st_type sinvert -eps_target 0 -eps_target_magnitude");   // epssetup.c:117-120
  eps->cmp_final = cmp;                                              // EPSSetUpSort_Basic: eps->sc, no map
  eps->cmp_ds = cmp;                                                 // EPSSetUpSort_Default: DS sc with map = SlepcMap_ST
  if (eps->B || !ks_st_is_plain(st)) {
    if (!st) { KS_CALL(ks_eps_get_st(eps, &st)); }
    KS_CALL(ks_st_set_matrices(st, A, eps->B)); st->ready = false;
    KS_CALL(ks_st_setup_internal(st));
    eps->op = st->op; eps->cmp_ds.map = st;
  } else eps->op = A;
  int nev = eps->nev, ncv = eps->ncv_user, mpd = eps->mpd_user;
  if (ncv) { KS_CHECK(ncv >= nev + 1 || (ncv == nev && ncv == n), KS_ERR_USER_INPUT, "The value of ncv must be at least nev+1"); }
  else if (mpd) ncv = std::min(n, nev + mpd);
  else { if (nev < 500) ncv = std::min(n, std::max(2 * nev, nev + 15)); else { mpd = 500; ncv = std::min(n, nev + mpd); } }
  if (!mpd) mpd = ncv;
  KS_CHECK(ncv <= nev + mpd, KS_ERR_USER_INPUT, "The value of ncv must not be larger than nev+mpd");
  // ncv + 1 > 64 columns: the basis is wider than the register-tiled fused kernels; Gram-Schmidt then runs its slot program over
  // 64-column chunks (ks_gs.hip, still enqueued as a whole) and the panel products are blocked (ks_bv.hip)
  eps->ncv = ncv; eps->mpd = mpd;
  eps->max_it = eps->max_it_user ? eps->max_it_user : std::max(100, 2 * n / ncv);
  if (eps->V) { int vm = 0; ks_bv_get_sizes(eps->V, nullptr, nullptr, &vm, nullptr); if (vm != ncv + 1 || eps->V->nc) { ks_bv_destroy(eps->V); eps->V = nullptr; } }
  if (!eps->V) { KS_CALL(ks_bv_create(eps->ctx, A->n, A->n_global, ncv + 1, 0, &eps->V)); eps->V->row_start = A->row_start; }   // EPSAllocateSolution(eps,1)
  if (!eps->W) { KS_CALL(ks_bv_create(eps->ctx, A->n, A->n_global, 5, 0, &eps->W)); }     // work vectors: u, B*xr, B*xi, Ritz vector x, y
  eps->eigr.assign(ncv + 1, 0.0); eps->eigi.assign(ncv + 1, 0.0); eps->errest.assign(ncv + 1, 0.0);
  eps->perm.resize(ncv + 1); for (int i = 0; i <= ncv; i++) eps->perm[i] = i;
  DsHep &ds = eps->ds;
  ds.allocate(ncv + 1); ds.which = eps->cmp_ds; ds.state = DS_RAW;
  eps->cb_err = 0;
  eps->nconv = 0; eps->its = 0; eps->reason = KS_EPS_CONVERGED_ITERATING; eps->steps = 0; eps->restarts = 0; eps->solved = false;
  long long passes0 = 0; ks_bv_gs_passes(eps->V, &passes0, nullptr);
  ks_bv V = eps->V;
  KS_CALL(ks_bv_set_active_columns(V, 0, ncv + 1));
  KS_CALL(ks_bv_set_matrix(V, ghep ? (cayley ? st->bil : eps->B) : nullptr));   // EPS_SetInnerProduct epsimpl.h:280-292: STGetBilinearForm = B, or A + nu B for STCAYLEY (cayley.c:70-77)
  eps->ghep = ghep;
  eps->problem_type_resolved_hermitian = (ptype == KS_EPS_HEP || ghep) && eps->extraction != KS_EPS_HARMONIC;
  eps->vectors_done = true;
  if (eps->nds) {                                                      // process the deflation space (epssetup.c:397-404)
    KS_CHECK(eps->defl && eps->defl->n == A->n, KS_ERR_ARG_INCOMP, "the deflation space was set for an operator of another size");
    std::vector<const double *> cp(eps->nds);
    for (int i = 0; i < eps->nds; i++) cp[i] = ks_bv_col(eps->defl, i);
    int kd = eps->nds;
    int rc = ks_bv_insert_constraints(V, &kd, cp.data());
    ks_bv_destroy(eps->defl); eps->defl = nullptr; eps->nds = 0;
    if (rc) return rc;
  }

  // balancing of non-symmetric problems (epssetup.c:383-391): build D, then expand with D Op D^-1
  eps->balanced = false;
  if ((ptype == KS_EPS_NHEP || ptype == KS_EPS_GNHEP) && eps->balance != KS_EPS_BALANCE_NONE) {
    eps->op_inner = eps->op;
    if (eps->balance == KS_EPS_BALANCE_ONESIDE || eps->balance == KS_EPS_BALANCE_TWOSIDE) KS_CALL(build_balance(eps));
    else KS_CHECK(eps->D && eps->D_n == A->n, KS_ERR_ORDER, "EPS_BALANCE_USER: the balancing matrix does not match the operator");
    if (!eps->bal_op) KS_CALL(ks_mat_create_shell(eps->ctx, A->n, A->row_start, A->n_global, balanced_mult, eps, &eps->bal_op));
    eps->bal_op->n = A->n; eps->bal_op->row_start = A->row_start; eps->bal_op->n_global = A->n_global;
    eps->bal_op->shell_nosync = !eps->op_inner->shell_mult;         // D A D^-1 on an assembled matrix is three kernel launches: keep the enqueued-ahead run
    eps->op = eps->bal_op; eps->balanced = true;
  }
  KS_CHECK(eps->extraction == KS_EPS_RITZ || !ghep, KS_ERR_SUP, "harmonic extraction with a B-inner product is not built");
  KS_CHECK(!eps->arb_fn || ((ptype == KS_EPS_HEP || ghep) && eps->extraction == KS_EPS_RITZ), KS_ERR_SUP, "arbitrary selection is built for the symmetric (Lanczos) variant only");
  if ((ptype != KS_EPS_HEP && !ghep) || eps->extraction == KS_EPS_HARMONIC) return solve_nhep(eps, passes0);   // variant EPS_KS_DEFAULT (krylovschur.c:133-151)
  const bool isshift = !st || st->type == KS_ST_SHIFT;

  // ---- EPSSolve_KrylovSchur_Default ----
  KS_CALL(start_vector(eps, 0, nullptr));
  int l = 0;
  while (eps->reason == KS_EPS_CONVERGED_ITERATING) {
    eps->its++;
    int nv = std::min(eps->nconv + mpd, ncv);
    if (eps->max_steps && eps->steps + (nv - (eps->nconv + l)) > eps->max_steps) nv = eps->nconv + l + (int)(eps->max_steps - eps->steps);
    ds.set_dimensions(nv, eps->nconv, eps->nconv + l);
    double beta = 0.0; int breakdown = 0;
    const int k0 = eps->nconv + l;
    KS_CALL(ks_bv_matlanczos(V, eps->op, ds.T.data(), ds.ld, k0, &nv, &beta, &breakdown));
    eps->steps += nv - k0;
    ds.set_dimensions(nv, eps->nconv, eps->nconv + l);
    ds.state = l ? DS_RAW : DS_INTERMEDIATE;
    KS_CALL(ks_bv_set_active_columns(V, eps->nconv, nv));

    // solve projected problem
    int info = ds.solve(eps->eigr.data());
    KS_CHECK(info == 0, KS_ERR_LIB, "tridiagonal QL iteration failed to converge (info=%d)", info);
    if (eps->arb_fn) {                                         // EPSGetArbitraryValues krylovschur.c:30-58, then DSSort on rr/ri
      std::vector<double> rr(ncv + 1, 0.0), ri(ncv + 1, 0.0);
      for (int i = ds.l; i < ds.n; i++) {
        double re = eps->eigr[i], im0 = 0.0;
        if (eps->cmp_ds.map) ks_st_backtransform_internal(eps->cmp_ds.map, 1, &re, &im0);
        KS_CALL(ritz_vector(eps, nv, ds.Q.data() + (size_t)i * ds.ld, nullptr));       // DSVectors(X,i) = Q(:,i) for DSHEP
        KS_HIP(ks_sync(eps->ctx));
        const int rc = eps->arb_fn(re, im0, ks_bv_col(eps->W, 3), ks_bv_col(eps->W, 4), &rr[i], &ri[i], eps->arb_ctx);
        KS_CHECK(!rc, rc, "the user's arbitrary selection function returned %d", rc);
      }
      ds.sort(eps->eigr.data(), rr.data(), ri.data());
    } else ds.sort(eps->eigr.data());
    ds.update_extra_row();
    KS_CALL(ds_synchronize(eps, &ds.T, &ds.Q, &beta, &nv, &breakdown));      // krylovschur.c:281

    // EPSKrylovConvergence(eps,FALSE,nconv,nv-nconv,beta,0.0,1.0,&k)
    int marker = -1, k;
    for (k = eps->nconv; k < nv; k++) {
      double re = eps->eigr[k], im0 = 0.0;
      if ((isshift || eps->conv == KS_EPS_CONV_NORM) && eps->cmp_ds.map) ks_st_backtransform_internal(eps->cmp_ds.map, 1, &re, &im0);   // epskrylov.c:253 (identity for sigma = 0)
      double resnorm = ds.vectors_resnorm(k) * beta * 1.0;
      if (eps->trueres) {                                      // epskrylov.c:256-264: X(:,k) = Q(:,k) for DSHEP (dshep.c:140-155)
        if (!((isshift || eps->conv == KS_EPS_CONV_NORM) && eps->cmp_ds.map) && eps->cmp_ds.map) ks_st_backtransform_internal(eps->cmp_ds.map, 1, &re, &im0);
        KS_CALL(true_residual(eps, nv, re, 0.0, ds.Q.data() + (size_t)k * ds.ld, nullptr, &resnorm));
      }
      eps->errest[k] = converged_estimate(eps, re, 0.0, resnorm);
      if (marker == -1 && eps->errest[k] >= eps->tol) marker = k;
      if (marker != -1 && !eps->trackall) break;               // getall: estimates for every Ritz pair (epskrylov.c:240,280)
    }
    if (marker != -1) k = marker;
    // EPSStoppingBasic
    KS_CHECK(!eps->cb_err, eps->cb_err, "the user's convergence test returned %d", eps->cb_err);
    KS_CALL(stopping_test(eps, k));
    const int nconv_mon = k;

    // update l
    if (eps->reason != KS_EPS_CONVERGED_ITERATING || breakdown || k == nv) l = 0;
    else l = std::max(1, (int)((nv - k) * eps->keep));
    if (!eps->lock && l > 0) { l += k; k = 0; }      // non-locking variant: reset no. of converged pairs (krylovschur.c:294)
    if (eps->reason == KS_EPS_CONVERGED_ITERATING) {
      if (breakdown || k == nv) {
        if (k < nev) {
          bool brk = false;
          KS_CALL(start_vector(eps, k, &brk));
          if (brk) eps->reason = KS_EPS_DIVERGED_BREAKDOWN;
        }
      } else ds.truncate(k + l, false);
    }
    // V(:,nconv:k+l) = V(:,nconv:nv) * Q(nconv:nv, nconv:k+l)      krylovschur.c:324-327
    KS_CALL(ks_bv_multinplace(V, ds.Q.data(), ds.ld, eps->nconv, k + l));
    if (eps->reason == KS_EPS_CONVERGED_ITERATING && !breakdown) KS_CALL(ks_bv_copycolumn(V, nv, k + l));
    eps->nconv = k;
    KS_CALL(monitor(eps, nconv_mon, nv));
    eps->restarts++;
  }
  ds.truncate(eps->nconv, true);

  // ---- EPSSolve epilogue ----
  KS_CALL(ks_bv_set_active_columns(V, 0, eps->nconv));
  // EPSComputeValues (epssolve.c:27-41): map the eigenvalues back through the ST
  const int nc = eps->nconv;
  if (eps->cmp_ds.map) ks_st_backtransform_internal(eps->cmp_ds.map, nc, eps->eigr.data(), eps->eigi.data());
  if (ghep && eps->purify) {
    // EPSComputeVectors_Hermitian epsdefault.c:27-49: purification x <- OP x (EPS_Purify epsimpl.h:297-312), then B-normalise
    for (int i = 0; i < nc; i++) {
      KS_CALL(ksk_copy(eps->ctx, ks_bv_col(V, i), ks_bv_col(eps->W, 0), V->n));
      KS_CALL(ks_mat_mult_internal(eps->op, ks_bv_col(eps->W, 0), ks_bv_col(V, i)));
    }
    KS_CALL(ks_bv_normalize(V, nullptr));
  } else if (ghep && cayley) {
    // without purification the Lanczos vectors are the eigenvectors; under the Cayley transformation they are orthonormal in
    // the A + nu B inner product and still have to be B-normalised (epsdefault.c:38-47)
    KS_CALL(ks_bv_set_matrix(V, eps->B));
    int rc = ks_bv_normalize(V, nullptr);
    KS_CALL(ks_bv_set_matrix(V, st->bil));
    if (rc) return rc;
  }
  // SlepcSortEigenvalues slepcsc.c:89-140 (all eigenvalues real here)
  for (int i = 0; i <= ncv; i++) eps->perm[i] = i;
  for (int i = nc - 1; i >= 0; i--) {
    const double re = eps->eigr[eps->perm[i]];
    int j = i + 1;
    while (j < nc) {
      if (compare_eig(eps->cmp_final, re, 0.0, eps->eigr[eps->perm[j]], 0.0) <= 0) break;
      std::swap(eps->perm[j - 1], eps->perm[j]); j++;
    }
  }
  long long passes1 = 0; ks_bv_gs_passes(V, &passes1, nullptr);
  eps->passes = passes1 - passes0;
  KS_CALL(ks_bv_set_num_constraints(V, 0));                            // remove the deflation space (epssolve.c:201-205)
  eps->solved = true;
  return KS_SUCCESS;
}

extern "C" int ks_eps_get_converged(ks_eps eps, int *nconv) { KS_CHECK(eps && nconv, KS_ERR_ARG_NULL, "NULL argument"); KS_CHECK(eps->solved, KS_ERR_ARG_WRONGSTATE, "Must call EPSSolve() first"); *nconv = eps->nconv; return KS_SUCCESS; }
extern "C" int ks_eps_get_iteration_number(ks_eps eps, int *its) { KS_CHECK(eps && its, KS_ERR_ARG_NULL, "NULL argument"); *its = eps->its; return KS_SUCCESS; }
extern "C" int ks_eps_get_converged_reason(ks_eps eps, int *reason) { KS_CHECK(eps && reason, KS_ERR_ARG_NULL, "NULL argument"); KS_CHECK(eps->solved, KS_ERR_ARG_WRONGSTATE, "Must call EPSSolve() first"); *reason = eps->reason; return KS_SUCCESS; }
extern "C" int ks_eps_get_dimensions(ks_eps eps, int *nev, int *ncv, int *mpd)
{
  KS_CHECK(eps, KS_ERR_ARG_NULL, "EPS is NULL");
  if (nev) *nev = eps->nev; if (ncv) *ncv = eps->ncv; if (mpd) *mpd = eps->mpd;
  return KS_SUCCESS;
}
extern "C" int ks_eps_get_eigenvalue(ks_eps eps, int i, double *eigr, double *eigi)   // EPSGetEigenvalue epssolve.c:478: through perm
{
  KS_CHECK(eps, KS_ERR_ARG_NULL, "EPS is NULL");
  KS_CHECK(eps->solved, KS_ERR_ARG_WRONGSTATE, "Must call EPSSolve() first");
  KS_CHECK(i >= 0, KS_ERR_ARG_OUTOFRANGE, "The index cannot be negative");
  KS_CHECK(i < eps->nconv, KS_ERR_ARG_OUTOFRANGE, "The index can be nconv-1 at most, see EPSGetConverged()");
  const int k = eps->perm[i];
  if (eigr) *eigr = eps->eigr[k];
  if (eigi) *eigi = eps->eigi[k];
  return KS_SUCCESS;
}
extern "C" int ks_eps_get_eigenvector_host(ks_eps eps, int i, double *xr)
{
  KS_CHECK(eps && xr, KS_ERR_ARG_NULL, "NULL argument");
  KS_CHECK(eps->solved, KS_ERR_ARG_WRONGSTATE, "Must call EPSSolve() first");
  KS_CHECK(i >= 0 && i < eps->nconv, KS_ERR_ARG_OUTOFRANGE, "The index can be nconv-1 at most, see EPSGetConverged()");
  KS_CALL(compute_vectors(eps));
  const int k = eps->perm[i];
  // EPSComputeVectors_Hermitian: V already holds the Ritz vectors; pairs: BV_GetEigenvector bvimpl.h:423-446
  return ks_bv_get_column_host(eps->V, eps->eigi[k] < 0.0 ? k - 1 : k, xr);
}
extern "C" int ks_eps_get_eigenpair_host(ks_eps eps, int i, double *eigr, double *eigi, double *xr, double *xi)   // EPSGetEigenpair epssolve.c:405
{
  KS_CHECK(eps, KS_ERR_ARG_NULL, "EPS is NULL");
  KS_CHECK(eps->solved, KS_ERR_ARG_WRONGSTATE, "Must call EPSSolve() first");
  KS_CHECK(i >= 0 && i < eps->nconv, KS_ERR_ARG_OUTOFRANGE, "The index can be nconv-1 at most, see EPSGetConverged()");
  KS_CALL(compute_vectors(eps));
  const int k = eps->perm[i], nloc = eps->V->n;
  const double im = eps->eigi[k];
  if (eigr) *eigr = eps->eigr[k];
  if (eigi) *eigi = im;
  if (im > 0.0) { if (xr) KS_CALL(ks_bv_get_column_host(eps->V, k, xr)); if (xi) KS_CALL(ks_bv_get_column_host(eps->V, k + 1, xi)); }
  else if (im < 0.0) {
    if (xr) KS_CALL(ks_bv_get_column_host(eps->V, k - 1, xr));
    if (xi) { KS_CALL(ks_bv_get_column_host(eps->V, k, xi)); for (int r = 0; r < nloc; r++) xi[r] = -xi[r]; }
  } else { if (xr) KS_CALL(ks_bv_get_column_host(eps->V, k, xr)); if (xi) for (int r = 0; r < nloc; r++) xi[r] = 0.0; }
  return KS_SUCCESS;
}
// the same with device vectors of n_local doubles (what EPSGetEigenpair fills when the Vecs live on the GPU)
extern "C" int ks_eps_get_eigenpair(ks_eps eps, int i, double *eigr, double *eigi, double *xr_dev, double *xi_dev)
{
  KS_CHECK(eps, KS_ERR_ARG_NULL, "EPS is NULL");
  KS_CHECK(eps->solved, KS_ERR_ARG_WRONGSTATE, "Must call EPSSolve() first");
  KS_CHECK(i >= 0 && i < eps->nconv, KS_ERR_ARG_OUTOFRANGE, "The index can be nconv-1 at most, see EPSGetConverged()");
  KS_CALL(compute_vectors(eps));
  const int k = eps->perm[i]; const size_t nloc = eps->V->n;
  const double im = eps->eigi[k];
  ks_ctx ctx = eps->ctx; ks_bv V = eps->V;
  KS_HIP(hipSetDevice(ctx->device));
  if (eigr) *eigr = eps->eigr[k];
  if (eigi) *eigi = im;
  const int kr = im < 0.0 ? k - 1 : k;
  if (xr_dev) KS_CALL(ksk_copy(ctx, ks_bv_col(V, kr), xr_dev, nloc));
  if (xi_dev) {
    if (im == 0.0) KS_HIP(hipMemsetAsync(xi_dev, 0, nloc * sizeof(double), ctx->stream));
    else { KS_CALL(ksk_copy(ctx, ks_bv_col(V, kr + 1), xi_dev, nloc)); if (im < 0.0) KS_CALL(ksk_scale(ctx, xi_dev, nloc, -1.0)); }
  }
  KS_HIP(ks_sync(ctx));
  return KS_SUCCESS;
}
extern "C" int ks_eps_get_error_estimate(ks_eps eps, int i, double *errest)
{
  KS_CHECK(eps && errest, KS_ERR_ARG_NULL, "NULL argument");
  KS_CHECK(eps->solved, KS_ERR_ARG_WRONGSTATE, "Must call EPSSolve() first");
  KS_CHECK(i >= 0 && i < eps->nconv, KS_ERR_ARG_OUTOFRANGE, "The index can be nconv-1 at most, see EPSGetConverged()");
  *errest = eps->errest[eps->perm[i]];
  return KS_SUCCESS;
}

extern "C" int ks_eps_compute_error(ks_eps eps, int i, int type, double *error)   // epssolve.c:742-815 with :666-718
{
  KS_CHECK(eps && error, KS_ERR_ARG_NULL, "NULL argument");
  KS_CHECK(eps->solved, KS_ERR_ARG_WRONGSTATE, "Must call EPSSolve() first");
  KS_CHECK(i >= 0 && i < eps->nconv, KS_ERR_ARG_OUTOFRANGE, "The index can be nconv-1 at most, see EPSGetConverged()");
  KS_CALL(compute_vectors(eps));
  const int j = eps->perm[i];
  const double kr = eps->eigr[j], ki = eps->eigi[j];
  ks_bv V = eps->V;
  double nrm = 0.0;
  if (ki == 0.0) KS_CALL(residual_norm(eps, kr, ki, ks_bv_col(V, j), nullptr, 1.0, &nrm));
  else {
    // complex pair in real arithmetic: xr = V(:,jr), xi = sg*V(:,jr+1) (BV_GetEigenvector bvimpl.h:423-446)
    const int jr = ki > 0.0 ? j : j - 1;
    KS_CALL(residual_norm(eps, kr, ki, ks_bv_col(V, jr), ks_bv_col(V, jr + 1), ki > 0.0 ? 1.0 : -1.0, &nrm));
  }
  double vecnorm = 1.0;
  if (eps->ghep) { ks_mat Bsave = V->matrix; V->matrix = nullptr; int rc = ks_bv_normcolumn(V, j, KS_NORM_2, &vecnorm); V->matrix = Bsave; if (rc) return rc; }   // epssolve.c:774: 2-norm of the eigenvector
  if (type == KS_EPS_ERROR_RELATIVE) nrm /= hypot(kr, ki) * vecnorm;
  else if (type == KS_EPS_ERROR_BACKWARD) { KS_CALL(matrix_norms(eps)); nrm /= (eps->nrma + hypot(kr, ki) * eps->nrmb) * vecnorm; }   // epssolve.c:782-800
  else KS_CHECK(type == KS_EPS_ERROR_ABSOLUTE, KS_ERR_ARG_OUTOFRANGE, "Invalid error type");
  *error = nrm;
  return KS_SUCCESS;
}

// ---- getters of the settings (EPSGetTolerances, EPSGetWhichEigenpairs, EPSGetTarget, EPSGetProblemType, EPSIs*, ...) ----
extern "C" int ks_eps_get_tolerances(ks_eps eps, double *tol, int *max_it)
{
  KS_CHECK(eps, KS_ERR_ARG_NULL, "EPS is NULL");
  if (tol) *tol = eps->tol;
  if (max_it) *max_it = eps->solved ? eps->max_it : eps->max_it_user;      // 0 before set-up when left to the default
  return KS_SUCCESS;
}
extern "C" int ks_eps_get_which_eigenpairs(ks_eps eps, int *which) { KS_CHECK(eps && which, KS_ERR_ARG_NULL, "NULL argument"); *which = eps->which.which; return KS_SUCCESS; }
extern "C" int ks_eps_get_target(ks_eps eps, double *target) { KS_CHECK(eps && target, KS_ERR_ARG_NULL, "NULL argument"); *target = eps->which.target; return KS_SUCCESS; }
extern "C" int ks_eps_get_convergence_test(ks_eps eps, int *conv) { KS_CHECK(eps && conv, KS_ERR_ARG_NULL, "NULL argument"); *conv = eps->conv; return KS_SUCCESS; }
extern "C" int ks_eps_get_operators(ks_eps eps, ks_mat *A, ks_mat *B) { KS_CHECK(eps, KS_ERR_ARG_NULL, "EPS is NULL"); if (A) *A = eps->A; if (B) *B = eps->B; return KS_SUCCESS; }
extern "C" int ks_eps_get_problem_type(ks_eps eps, int *type, int *generalized, int *hermitian, int *positive)   // EPSGetProblemType + EPSIsGeneralized/IsHermitian/IsPositive
{
  KS_CHECK(eps, KS_ERR_ARG_NULL, "EPS is NULL");
  const int t = eps->problem_type;
  if (type) *type = t;
  if (generalized) *generalized = (t == KS_EPS_GHEP || t == KS_EPS_GNHEP);
  if (hermitian) *hermitian = (t == KS_EPS_HEP || t == KS_EPS_GHEP);
  if (positive) *positive = (t == KS_EPS_GHEP);
  return KS_SUCCESS;
}
// EPSGetInvariantSubspace epssolve.c:247-280: an orthonormal basis of the converged invariant subspace into nconv device
// vectors. Non-symmetric problems: the Schur vectors, which only exist until the eigenvectors are first formed.
extern "C" int ks_eps_get_invariant_subspace(ks_eps eps, double *const *v_dev)
{
  KS_CHECK(eps && v_dev, KS_ERR_ARG_NULL, "NULL argument");
  KS_CHECK(eps->solved, KS_ERR_ARG_WRONGSTATE, "Must call EPSSolve() first");
  KS_CHECK(!eps->vectors_done || eps->problem_type_resolved_hermitian, KS_ERR_ARG_WRONGSTATE,
           "EPSGetInvariantSubspace must be called before EPSGetEigenpair,EPSGetEigenvector or EPSComputeError");
  KS_HIP(hipSetDevice(eps->ctx->device));
  for (int i = 0; i < eps->nconv; i++) KS_CHECK(v_dev[i], KS_ERR_ARG_NULL, "vector %d is NULL", i);
  if (eps->balanced && !eps->vectors_done && eps->nconv) {  // epssolve.c:351-362: Q <- orth(D \ Q)
    ks_bv T = nullptr;
    KS_CALL(ks_bv_create(eps->ctx, eps->V->n, eps->V->N, eps->nconv, 0, &T));
    int rc = KS_SUCCESS;
    for (int i = 0; i < eps->nconv && !rc; i++) rc = pointwise(eps, ks_bv_col(eps->V, i), eps->D, ks_bv_col(T, i), false);
    if (!rc) rc = ks_bv_orthogonalize(T, nullptr, 0);
    for (int i = 0; i < eps->nconv && !rc; i++) rc = ksk_copy(eps->ctx, ks_bv_col(T, i), v_dev[i], eps->V->n);
    ks_sync(eps->ctx);
    ks_bv_destroy(T);
    return rc;
  }
  for (int i = 0; i < eps->nconv; i++) KS_CALL(ksk_copy(eps->ctx, ks_bv_col(eps->V, i), v_dev[i], eps->V->n));
  KS_HIP(ks_sync(eps->ctx));
  return KS_SUCCESS;
}
extern "C" int ks_eps_get_bv(ks_eps eps, ks_bv *V) { KS_CHECK(eps && V, KS_ERR_ARG_NULL, "NULL argument"); *V = eps->V; return KS_SUCCESS; }
extern "C" int ks_eps_get_stats(ks_eps eps, long long *steps, long long *passes, int *restarts)
{
  KS_CHECK(eps, KS_ERR_ARG_NULL, "EPS is NULL");
  if (steps) *steps = eps->steps; if (passes) *passes = eps->passes; if (restarts) *restarts = eps->restarts;
  return KS_SUCCESS;
}
