// Host dense kernels for the projected non-symmetric problem (see ks_dense.h). Algorithms follow the LAPACK
// routines the reference calls (dgehd2/dorg2r, dlahqr, dlanv2, dtrexc/dlaexc/dlasy2, dtrevc/dlaln2), written for
// well-scaled m <= 64 matrices: the overflow-guard rescaling loops of LAPACK are omitted, small pivots are perturbed.
#include "ks_dense.h"
#include <cmath>
#include <cfloat>
#include <complex>
#include <algorithm>
#include <cstring>
#include <vector>

namespace ksd {

namespace {

inline double sgn(double a, double b) { return b >= 0.0 ? std::fabs(a) : -std::fabs(a); }    // Fortran SIGN(a,b)
#define AT(M, i, j) M[(size_t)(i) + (size_t)(j) * ld]

// x' = c x + s y ; y' = c y - s x  (BLAS drot)
void rot(int n, double *x, int incx, double *y, int incy, double c, double s)
{
  for (int i = 0; i < n; i++) { const double t = c * x[(size_t)i * incx] + s * y[(size_t)i * incy]; y[(size_t)i * incy] = c * y[(size_t)i * incy] - s * x[(size_t)i * incx]; x[(size_t)i * incx] = t; }
}

// dlartg: [c s; -s c] [f; g] = [r; 0]
void lartg(double f, double g, double &c, double &s, double &r)
{
  if (g == 0.0) { c = 1.0; s = 0.0; r = f; }
  else if (f == 0.0) { c = 0.0; s = (g < 0.0) ? -1.0 : 1.0; r = std::fabs(g); }
  else { const double d = std::hypot(f, g); c = std::fabs(f) / d; r = sgn(d, f); s = g / r; }
}

// dlarfg: H [alpha; x] = [beta; 0], H = I - tau [1; v][1; v]^T ; v overwrites x, beta overwrites alpha
void larfg(int n, double &alpha, double *x, int incx, double &tau)
{
  if (n <= 1) { tau = 0.0; return; }
  double xn = 0.0;
  for (int i = 0; i < n - 1; i++) xn = std::hypot(xn, x[(size_t)i * incx]);
  if (xn == 0.0) { tau = 0.0; return; }
  const double beta = -sgn(std::hypot(alpha, xn), alpha);
  tau = (beta - alpha) / beta;
  const double sc = 1.0 / (alpha - beta);
  for (int i = 0; i < n - 1; i++) x[(size_t)i * incx] *= sc;
  alpha = beta;
}

// dlarfx with a 3-vector: C <- H C (side 'L', C is 3 x ncols) or C <- C H (side 'R', C is nrows x 3)
void larfx3(char side, int cnt, const double *v, double tau, double *C, int ldc)
{
  if (tau == 0.0) return;
  if (side == 'L') {
    for (int j = 0; j < cnt; j++) {
      double *c = C + (size_t)j * ldc;
      const double s = v[0] * c[0] + v[1] * c[1] + v[2] * c[2];
      c[0] -= tau * s * v[0]; c[1] -= tau * s * v[1]; c[2] -= tau * s * v[2];
    }
  } else {
    for (int i = 0; i < cnt; i++) {
      double *c0 = C + i, *c1 = C + i + ldc, *c2 = C + i + 2 * (size_t)ldc;
      const double s = v[0] * *c0 + v[1] * *c1 + v[2] * *c2;
      *c0 -= tau * s * v[0]; *c1 -= tau * s * v[1]; *c2 -= tau * s * v[2];
    }
  }
}

// Solve TL*X - X*TR = B for X (n1 x n2, n1,n2 in {1,2}) by the Kronecker system with complete pivoting (dlasy2,
// isgn=-1, scale=1). Column-major 2x2 arrays with leading dimension 2 for X and B.
void sylv_small(int n1, int n2, const double *TL, int ldtl, const double *TR, int ldtr, const double *B, int ldb, double *X, double &xnorm)
{
  const int N = n1 * n2;
  double K[16] = {0}, rhs[4] = {0};
  // unknown index p = i + j*n1 for X(i,j)
  for (int j = 0; j < n2; j++)
    for (int i = 0; i < n1; i++) {
      const int p = i + j * n1;
      rhs[p] = B[i + j * ldb];
      for (int k = 0; k < n1; k++) K[p + (k + j * n1) * 4] += TL[i + k * ldtl];        // (TL X)(i,j) = sum_k TL(i,k) X(k,j)
      for (int k = 0; k < n2; k++) K[p + (i + k * n1) * 4] -= TR[k + j * ldtr];        // (X TR)(i,j) = sum_k X(i,k) TR(k,j)
    }
  double big = 0.0;
  for (int a = 0; a < N; a++) for (int b = 0; b < N; b++) big = std::max(big, std::fabs(K[a + b * 4]));
  const double smin = std::max(DBL_EPSILON * big, DBL_MIN / DBL_EPSILON);
  int cperm[4] = {0, 1, 2, 3};
  for (int s = 0; s < N; s++) {
    int pr = s, pc = s; double mx = -1.0;
    for (int a = s; a < N; a++) for (int b = s; b < N; b++) if (std::fabs(K[a + b * 4]) > mx) { mx = std::fabs(K[a + b * 4]); pr = a; pc = b; }
    if (pr != s) { for (int b = 0; b < N; b++) std::swap(K[s + b * 4], K[pr + b * 4]); std::swap(rhs[s], rhs[pr]); }
    if (pc != s) { for (int a = 0; a < N; a++) std::swap(K[a + s * 4], K[a + pc * 4]); std::swap(cperm[s], cperm[pc]); }
    if (std::fabs(K[s + s * 4]) < smin) K[s + s * 4] = smin;
    for (int a = s + 1; a < N; a++) {
      const double f = K[a + s * 4] / K[s + s * 4];
      for (int b = s; b < N; b++) K[a + b * 4] -= f * K[s + b * 4];
      rhs[a] -= f * rhs[s];
    }
  }
  double sol[4];
  for (int s = N - 1; s >= 0; s--) { double t = rhs[s]; for (int b = s + 1; b < N; b++) t -= K[s + b * 4] * sol[b]; sol[s] = t / K[s + s * 4]; }
  xnorm = 0.0;
  for (int s = 0; s < N; s++) { const int p = cperm[s]; X[(p % n1) + (p / n1) * 2] = sol[s]; }
  for (int i = 0; i < n1; i++) { double r = 0.0; for (int j = 0; j < n2; j++) r += std::fabs(X[i + j * 2]); xnorm = std::max(xnorm, r); }
}

// standardise the 2x2 block at (j,j) of T and carry the rotation through T and Q
void standardize_block(int n, double *T, int ld, double *Q, int j)
{
  double wr1, wi1, wr2, wi2, cs, sn;
  lanv2(AT(T, j, j), AT(T, j, j + 1), AT(T, j + 1, j), AT(T, j + 1, j + 1), wr1, wi1, wr2, wi2, cs, sn);
  if (j + 2 < n) rot(n - j - 2, &AT(T, j, j + 2), ld, &AT(T, j + 1, j + 2), ld, cs, sn);
  rot(j, &AT(T, 0, j), 1, &AT(T, 0, j + 1), 1, cs, sn);
  rot(n, &AT(Q, 0, j), 1, &AT(Q, 0, j + 1), 1, cs, sn);
}

// dlaexc: swap adjacent diagonal blocks T11 (n1 x n1 at j1) and T22 (n2 x n2). Returns 1 if rejected (too ill-conditioned).
int laexc(int n, double *T, int ld, double *Q, int j1, int n1, int n2)
{
  if (n == 0 || n1 == 0 || n2 == 0) return 0;
  if (j1 + n1 > n - 1) return 0;
  const int j2 = j1 + 1, j3 = j1 + 2, j4 = j1 + 3;
  if (n1 == 1 && n2 == 1) {
    const double t11 = AT(T, j1, j1), t22 = AT(T, j2, j2);
    double cs, sn, temp;
    lartg(AT(T, j1, j2), t22 - t11, cs, sn, temp);
    if (j3 <= n - 1) rot(n - j1 - 2, &AT(T, j1, j3), ld, &AT(T, j2, j3), ld, cs, sn);
    rot(j1, &AT(T, 0, j1), 1, &AT(T, 0, j2), 1, cs, sn);
    AT(T, j1, j1) = t22; AT(T, j2, j2) = t11;
    rot(n, &AT(Q, 0, j1), 1, &AT(Q, 0, j2), 1, cs, sn);
    return 0;
  }
  const int nd = n1 + n2;
  double D[16];
  for (int j = 0; j < nd; j++) for (int i = 0; i < nd; i++) D[i + j * 4] = AT(T, j1 + i, j1 + j);
  double dnorm = 0.0;
  for (int j = 0; j < nd; j++) for (int i = 0; i < nd; i++) dnorm = std::max(dnorm, std::fabs(D[i + j * 4]));
  const double eps = DBL_EPSILON, smlnum = DBL_MIN / eps, thresh = std::max(10.0 * eps * dnorm, smlnum);
  double X[4] = {0, 0, 0, 0}, xnorm;
  const double scale = 1.0;
  sylv_small(n1, n2, D, 4, D + n1 + n1 * 4, 4, D + n1 * 4, 4, X, xnorm);
  double u[3], u1[3], u2[3], tau, tau1, tau2;
  if (n1 == 1 && n2 == 2) {
    u[0] = scale; u[1] = X[0]; u[2] = X[0 + 1 * 2];
    larfg(3, u[2], u, 1, tau); u[2] = 1.0;
    const double t11 = AT(T, j1, j1);
    larfx3('L', 3, u, tau, D, 4); larfx3('R', 3, u, tau, D, 4);
    if (std::max(std::max(std::fabs(D[2 + 0 * 4]), std::fabs(D[2 + 1 * 4])), std::fabs(D[2 + 2 * 4] - t11)) > thresh) return 1;
    larfx3('L', n - j1, u, tau, &AT(T, j1, j1), ld);
    larfx3('R', j2 + 1, u, tau, &AT(T, 0, j1), ld);
    AT(T, j3, j1) = 0.0; AT(T, j3, j2) = 0.0; AT(T, j3, j3) = t11;
    larfx3('R', n, u, tau, &AT(Q, 0, j1), ld);
  } else if (n1 == 2 && n2 == 1) {
    u[0] = -X[0]; u[1] = -X[1]; u[2] = scale;
    larfg(3, u[0], u + 1, 1, tau); u[0] = 1.0;
    const double t33 = AT(T, j3, j3);
    larfx3('L', 3, u, tau, D, 4); larfx3('R', 3, u, tau, D, 4);
    if (std::max(std::max(std::fabs(D[1 + 0 * 4]), std::fabs(D[2 + 0 * 4])), std::fabs(D[0] - t33)) > thresh) return 1;
    larfx3('R', j3 + 1, u, tau, &AT(T, 0, j1), ld);
    larfx3('L', n - j1 - 1, u, tau, &AT(T, j1, j2), ld);
    AT(T, j1, j1) = t33; AT(T, j2, j1) = 0.0; AT(T, j3, j1) = 0.0;
    larfx3('R', n, u, tau, &AT(Q, 0, j1), ld);
  } else {   // n1 == 2 && n2 == 2
    u1[0] = -X[0]; u1[1] = -X[1]; u1[2] = scale;
    larfg(3, u1[0], u1 + 1, 1, tau1); u1[0] = 1.0;
    const double temp = -tau1 * (X[0 + 1 * 2] + u1[1] * X[1 + 1 * 2]);
    u2[0] = -temp * u1[1] - X[1 + 1 * 2]; u2[1] = -temp * u1[2]; u2[2] = scale;
    larfg(3, u2[0], u2 + 1, 1, tau2); u2[0] = 1.0;
    larfx3('L', 4, u1, tau1, D, 4); larfx3('R', 4, u1, tau1, D, 4);
    larfx3('L', 4, u2, tau2, D + 1, 4); larfx3('R', 4, u2, tau2, D + 4, 4);
    if (std::max(std::max(std::fabs(D[2 + 0 * 4]), std::fabs(D[2 + 1 * 4])), std::max(std::fabs(D[3 + 0 * 4]), std::fabs(D[3 + 1 * 4]))) > thresh) return 1;
    larfx3('L', n - j1, u1, tau1, &AT(T, j1, j1), ld); larfx3('R', j4 + 1, u1, tau1, &AT(T, 0, j1), ld);
    larfx3('L', n - j1, u2, tau2, &AT(T, j2, j1), ld); larfx3('R', j4 + 1, u2, tau2, &AT(T, 0, j2), ld);
    AT(T, j3, j1) = 0.0; AT(T, j3, j2) = 0.0; AT(T, j4, j1) = 0.0; AT(T, j4, j2) = 0.0;
    larfx3('R', n, u1, tau1, &AT(Q, 0, j1), ld); larfx3('R', n, u2, tau2, &AT(Q, 0, j2), ld);
  }
  if (n2 == 2) standardize_block(n, T, ld, Q, j1);
  if (n1 == 2) standardize_block(n, T, ld, Q, j1 + n2);
  return 0;
}

} // namespace

void lanv2(double &a, double &b, double &c, double &d, double &rt1r, double &rt1i, double &rt2r, double &rt2i, double &cs, double &sn)
{
  const double multpl = 4.0, eps = DBL_EPSILON;
  if (c == 0.0) { cs = 1.0; sn = 0.0; }
  else if (b == 0.0) { cs = 0.0; sn = 1.0; const double t = d; d = a; a = t; b = -c; c = 0.0; }
  else if ((a - d) == 0.0 && sgn(1.0, b) != sgn(1.0, c)) { cs = 1.0; sn = 0.0; }
  else {
    double temp = a - d, p = 0.5 * temp;
    const double bcmax = std::max(std::fabs(b), std::fabs(c)), bcmis = std::min(std::fabs(b), std::fabs(c)) * sgn(1.0, b) * sgn(1.0, c);
    const double scale = std::max(std::fabs(p), bcmax);
    double z = (p / scale) * p + (bcmax / scale) * bcmis;
    if (z >= multpl * eps) {            // real eigenvalues: make the block upper triangular
      z = p + sgn(std::sqrt(scale) * std::sqrt(z), p);
      a = d + z; d = d - (bcmax / z) * bcmis;
      const double tau = std::hypot(c, z);
      cs = z / tau; sn = c / tau; b = b - c; c = 0.0;
    } else {                            // complex (or nearly equal real) eigenvalues: make the diagonal equal
      const double sigma = b + c, tau = std::hypot(sigma, temp);
      cs = std::sqrt(0.5 * (1.0 + std::fabs(sigma) / tau));
      sn = -(p / (tau * cs)) * sgn(1.0, sigma);
      const double aa = a * cs + b * sn, bb = -a * sn + b * cs, cc = c * cs + d * sn, dd = -c * sn + d * cs;
      a = aa * cs + cc * sn; b = bb * cs + dd * sn; c = -aa * sn + cc * cs; d = -bb * sn + dd * cs;
      temp = 0.5 * (a + d); a = temp; d = temp;
      if (c != 0.0) {
        if (b != 0.0) {
          if (sgn(1.0, b) == sgn(1.0, c)) {        // real eigenvalues after all
            const double sab = std::sqrt(std::fabs(b)), sac = std::sqrt(std::fabs(c));
            p = sgn(sab * sac, c);
            const double tau2 = 1.0 / std::sqrt(std::fabs(b + c));
            a = temp + p; d = temp - p; b = b - c; c = 0.0;
            const double cs1 = sab * tau2, sn1 = sac * tau2;
            temp = cs * cs1 - sn * sn1; sn = cs * sn1 + sn * cs1; cs = temp;
          }
        } else { b = -c; c = 0.0; temp = cs; cs = -sn; sn = temp; }
      }
    }
  }
  rt1r = a; rt2r = d;
  if (c == 0.0) { rt1i = 0.0; rt2i = 0.0; }
  else { rt1i = std::sqrt(std::fabs(b)) * std::sqrt(std::fabs(c)); rt2i = -rt1i; }
}

void hess_reduce(int n, int ilo, double *A, int ld, double *Q)
{
  std::vector<double> v_(n + 1), w_(n + 1);
  double *v = v_.data(), *w = w_.data();
  for (int i = ilo; i < n - 2; i++) {
    const int nr = n - i - 1;                      // length of the reflector (rows i+1..n-1)
    double alpha = AT(A, i + 1, i), tau;
    for (int r = 1; r < nr; r++) v[r] = AT(A, i + 1 + r, i);
    larfg(nr, alpha, v + 1, 1, tau);
    v[0] = 1.0;
    AT(A, i + 1, i) = alpha;
    for (int r = 1; r < nr; r++) AT(A, i + 1 + r, i) = 0.0;
    if (tau == 0.0) continue;
    // A(0:n, i+1:n) <- A(0:n, i+1:n) H
    for (int r = 0; r < n; r++) { double s = 0.0; for (int c = 0; c < nr; c++) s += AT(A, r, i + 1 + c) * v[c]; w[r] = s; }
    for (int c = 0; c < nr; c++) for (int r = 0; r < n; r++) AT(A, r, i + 1 + c) -= tau * w[r] * v[c];
    // A(i+1:n, i+1:n) <- H A(i+1:n, i+1:n)
    for (int c = i + 1; c < n; c++) { double s = 0.0; for (int r = 0; r < nr; r++) s += v[r] * AT(A, i + 1 + r, c); for (int r = 0; r < nr; r++) AT(A, i + 1 + r, c) -= tau * s * v[r]; }
    // Q(:, i+1:n) <- Q(:, i+1:n) H
    for (int r = 0; r < n; r++) { double s = 0.0; for (int c = 0; c < nr; c++) s += AT(Q, r, i + 1 + c) * v[c]; w[r] = s; }
    for (int c = 0; c < nr; c++) for (int r = 0; r < n; r++) AT(Q, r, i + 1 + c) -= tau * w[r] * v[c];
  }
}

int real_schur(int n, int ilo, double *H, int ld, double *wr, double *wi, double *Z)
{
  const int ihi = n - 1;
  if (n == 0) return 0;
  for (int j = 0; j < ilo; j++) { /* eigenvalues of the leading (already triangular) part are set by the caller */ }
  if (ilo == ihi) { wr[ilo] = AT(H, ilo, ilo); wi[ilo] = 0.0; return 0; }
  for (int j = ilo; j <= ihi - 3; j++) { AT(H, j + 2, j) = 0.0; AT(H, j + 3, j) = 0.0; }
  if (ilo <= ihi - 2) AT(H, ihi, ihi - 2) = 0.0;
  const int nh = ihi - ilo + 1;
  const double safmin = DBL_MIN, ulp = DBL_EPSILON, smlnum = safmin * ((double)nh / ulp);
  const int i1 = 0, i2 = n - 1, itmax = 30 * std::max(10, nh), kexsh = 10;
  int kdefl = 0;
  int i = ihi;
  while (i >= ilo) {
    int l = ilo;
    bool converged = false;
    for (int its = 0; its <= itmax; its++) {
      int k;
      for (k = i; k > l; k--) {
        if (std::fabs(AT(H, k, k - 1)) <= smlnum) break;
        double tst = std::fabs(AT(H, k - 1, k - 1)) + std::fabs(AT(H, k, k));
        if (tst == 0.0) { if (k - 2 >= ilo) tst += std::fabs(AT(H, k - 1, k - 2)); if (k + 1 <= ihi) tst += std::fabs(AT(H, k + 1, k)); }
        if (std::fabs(AT(H, k, k - 1)) <= ulp * tst) {
          const double ab = std::max(std::fabs(AT(H, k, k - 1)), std::fabs(AT(H, k - 1, k))), ba = std::min(std::fabs(AT(H, k, k - 1)), std::fabs(AT(H, k - 1, k)));
          const double aa = std::max(std::fabs(AT(H, k, k)), std::fabs(AT(H, k - 1, k - 1) - AT(H, k, k))), bb = std::min(std::fabs(AT(H, k, k)), std::fabs(AT(H, k - 1, k - 1) - AT(H, k, k)));
          const double s = aa + ab;
          if (ba * (ab / s) <= std::max(smlnum, ulp * (bb * (aa / s)))) break;
        }
      }
      l = k;
      if (l > ilo) AT(H, l, l - 1) = 0.0;
      if (l >= i - 1) { converged = true; break; }
      kdefl++;
      double h11, h21, h12, h22;
      if (kdefl % (2 * kexsh) == 0) { const double s = std::fabs(AT(H, i, i - 1)) + std::fabs(AT(H, i - 1, i - 2)); h11 = 0.75 * s + AT(H, i, i); h12 = -0.4375 * s; h21 = s; h22 = h11; }
      else if (kdefl % kexsh == 0) { const double s = std::fabs(AT(H, l + 1, l)) + std::fabs(AT(H, l + 2, l + 1)); h11 = 0.75 * s + AT(H, l, l); h12 = -0.4375 * s; h21 = s; h22 = h11; }
      else { h11 = AT(H, i - 1, i - 1); h21 = AT(H, i, i - 1); h12 = AT(H, i - 1, i); h22 = AT(H, i, i); }
      double s = std::fabs(h11) + std::fabs(h12) + std::fabs(h21) + std::fabs(h22);
      double rt1r, rt1i, rt2r, rt2i;
      if (s == 0.0) { rt1r = rt1i = rt2r = rt2i = 0.0; }
      else {
        h11 /= s; h21 /= s; h12 /= s; h22 /= s;
        const double tr = (h11 + h22) / 2.0, det = (h11 - tr) * (h22 - tr) - h12 * h21, rtdisc = std::sqrt(std::fabs(det));
        if (det >= 0.0) { rt1r = tr * s; rt2r = rt1r; rt1i = rtdisc * s; rt2i = -rt1i; }
        else {
          rt1r = tr + rtdisc; rt2r = tr - rtdisc;
          if (std::fabs(rt1r - h22) <= std::fabs(rt2r - h22)) { rt1r *= s; rt2r = rt1r; } else { rt2r *= s; rt1r = rt2r; }
          rt1i = rt2i = 0.0;
        }
      }
      double v[3] = {0, 0, 0};
      int m;
      for (m = i - 2; m >= l; m--) {
        double h21s = std::fabs(AT(H, m + 1, m));
        s = std::fabs(AT(H, m, m) - rt2r) + std::fabs(rt2i) + h21s;
        h21s = AT(H, m + 1, m) / s;
        v[0] = h21s * AT(H, m, m + 1) + (AT(H, m, m) - rt1r) * ((AT(H, m, m) - rt2r) / s) - rt1i * (rt2i / s);
        v[1] = h21s * (AT(H, m, m) + AT(H, m + 1, m + 1) - rt1r - rt2r);
        v[2] = h21s * AT(H, m + 2, m + 1);
        s = std::fabs(v[0]) + std::fabs(v[1]) + std::fabs(v[2]);
        v[0] /= s; v[1] /= s; v[2] /= s;
        if (m == l) break;
        if (std::fabs(AT(H, m, m - 1)) * (std::fabs(v[1]) + std::fabs(v[2])) <=
            ulp * std::fabs(v[0]) * (std::fabs(AT(H, m - 1, m - 1)) + std::fabs(AT(H, m, m)) + std::fabs(AT(H, m + 1, m + 1)))) break;
      }
      for (int k2 = m; k2 <= i - 1; k2++) {
        const int nr = std::min(3, i - k2 + 1);
        if (k2 > m) { v[0] = AT(H, k2, k2 - 1); v[1] = AT(H, k2 + 1, k2 - 1); if (nr == 3) v[2] = AT(H, k2 + 2, k2 - 1); }
        double t1;
        larfg(nr, v[0], v + 1, 1, t1);
        if (k2 > m) { AT(H, k2, k2 - 1) = v[0]; AT(H, k2 + 1, k2 - 1) = 0.0; if (k2 < i - 1) AT(H, k2 + 2, k2 - 1) = 0.0; }
        else if (m > l) AT(H, k2, k2 - 1) *= (1.0 - t1);
        const double v2 = v[1], t2 = t1 * v2;
        if (nr == 3) {
          const double v3 = v[2], t3 = t1 * v3;
          for (int j = k2; j <= i2; j++) { const double sum = AT(H, k2, j) + v2 * AT(H, k2 + 1, j) + v3 * AT(H, k2 + 2, j); AT(H, k2, j) -= sum * t1; AT(H, k2 + 1, j) -= sum * t2; AT(H, k2 + 2, j) -= sum * t3; }
          for (int j = i1; j <= std::min(k2 + 3, i); j++) { const double sum = AT(H, j, k2) + v2 * AT(H, j, k2 + 1) + v3 * AT(H, j, k2 + 2); AT(H, j, k2) -= sum * t1; AT(H, j, k2 + 1) -= sum * t2; AT(H, j, k2 + 2) -= sum * t3; }
          for (int j = 0; j < n; j++) { const double sum = AT(Z, j, k2) + v2 * AT(Z, j, k2 + 1) + v3 * AT(Z, j, k2 + 2); AT(Z, j, k2) -= sum * t1; AT(Z, j, k2 + 1) -= sum * t2; AT(Z, j, k2 + 2) -= sum * t3; }
        } else if (nr == 2) {
          for (int j = k2; j <= i2; j++) { const double sum = AT(H, k2, j) + v2 * AT(H, k2 + 1, j); AT(H, k2, j) -= sum * t1; AT(H, k2 + 1, j) -= sum * t2; }
          for (int j = i1; j <= i; j++) { const double sum = AT(H, j, k2) + v2 * AT(H, j, k2 + 1); AT(H, j, k2) -= sum * t1; AT(H, j, k2 + 1) -= sum * t2; }
          for (int j = 0; j < n; j++) { const double sum = AT(Z, j, k2) + v2 * AT(Z, j, k2 + 1); AT(Z, j, k2) -= sum * t1; AT(Z, j, k2 + 1) -= sum * t2; }
        }
      }
    }
    if (!converged) return i + 1;
    if (l == i) { wr[i] = AT(H, i, i); wi[i] = 0.0; }
    else {   // l == i-1: a 2x2 block
      double cs, sn;
      lanv2(AT(H, i - 1, i - 1), AT(H, i - 1, i), AT(H, i, i - 1), AT(H, i, i), wr[i - 1], wi[i - 1], wr[i], wi[i], cs, sn);
      if (i2 > i) rot(i2 - i, &AT(H, i - 1, i + 1), ld, &AT(H, i, i + 1), ld, cs, sn);
      rot(i - i1 - 1, &AT(H, i1, i - 1), 1, &AT(H, i1, i), 1, cs, sn);
      rot(n, &AT(Z, 0, i - 1), 1, &AT(Z, 0, i), 1, cs, sn);
    }
    kdefl = 0;
    i = l - 1;
  }
  return 0;
}

int trexc_up(int n, double *T, int ld, double *Q, int ifst, int ilst)
{
  if (n <= 1) return 0;
  if (ifst > 0 && AT(T, ifst, ifst - 1) != 0.0) ifst--;
  int nbf = 1;
  if (ifst < n - 1 && AT(T, ifst + 1, ifst) != 0.0) nbf = 2;
  if (ilst > 0 && AT(T, ilst, ilst - 1) != 0.0) ilst--;
  if (ifst == ilst) return 0;
  if (ifst < ilst) return 2;                       // downward moves are not needed by DSSort_NHEP_Total
  int here = ifst;
  while (here > ilst) {
    if (nbf == 1 || nbf == 2) {
      int nbnext = 1;
      if (here >= 2 && AT(T, here - 1, here - 2) != 0.0) nbnext = 2;
      if (laexc(n, T, ld, Q, here - nbnext, nbnext, nbf)) return 1;
      here -= nbnext;
      if (nbf == 2 && AT(T, here + 1, here) == 0.0) nbf = 3;     // the moved 2x2 block split into two 1x1 blocks
    } else {
      int nbnext = 1;
      if (here >= 2 && AT(T, here - 1, here - 2) != 0.0) nbnext = 2;
      if (laexc(n, T, ld, Q, here - nbnext, nbnext, 1)) return 1;
      if (nbnext == 1) { if (laexc(n, T, ld, Q, here, nbnext, 1)) return 1; here -= 1; }
      else {
        if (AT(T, here, here - 1) == 0.0) nbnext = 1;
        if (nbnext == 2) { if (laexc(n, T, ld, Q, here - 1, 2, 1)) return 1; here -= 2; }
        else { if (laexc(n, T, ld, Q, here, 1, 1)) return 1; if (laexc(n, T, ld, Q, here - 1, 1, 1)) return 1; here -= 2; }
      }
    }
  }
  return 0;
}

int trevc_one(int n, const double *T, int ld, int k, double *xr, double *xi)
{
  typedef std::complex<double> cd;
  const bool pair = (k < n - 1 && AT(T, k + 1, k) != 0.0);
  const double wr = AT(T, k, k);
  const double wi = pair ? std::sqrt(std::fabs(AT(T, k, k + 1))) * std::sqrt(std::fabs(AT(T, k + 1, k))) : 0.0;
  const cd lam(wr, wi);
  double tnorm = 0.0;
  for (int j = 0; j < n; j++) for (int i = 0; i <= std::min(j + 1, n - 1); i++) tnorm = std::max(tnorm, std::fabs(AT(T, i, j)));
  const double smin = std::max(DBL_EPSILON * (std::fabs(wr) + std::fabs(wi)), std::max(DBL_EPSILON * tnorm * 1e-3, DBL_MIN / DBL_EPSILON));
  std::vector<std::complex<double>> x_(n + 1);
  std::complex<double> *x = x_.data();
  for (int j = 0; j < n; j++) x[j] = 0.0;
  int top;                                          // first row of the eigenvalue's block
  if (!pair) {
    x[k] = 1.0;
    for (int j = 0; j < k; j++) x[j] = -AT(T, j, k);
    top = k;
  } else {
    if (std::fabs(AT(T, k, k + 1)) >= std::fabs(AT(T, k + 1, k))) { x[k] = cd(1.0, 0.0); x[k + 1] = cd(0.0, wi / AT(T, k, k + 1)); }
    else { x[k] = cd(-wi / AT(T, k + 1, k), 0.0); x[k + 1] = cd(0.0, 1.0); }
    for (int j = 0; j < k; j++) x[j] = -(x[k].real() * AT(T, j, k)) - cd(0.0, x[k + 1].imag() * AT(T, j, k + 1));
    top = k;
  }
  // back substitution on (T(0:top,0:top) - lam I) x(0:top) = rhs
  int j = top - 1;
  while (j >= 0) {
    const bool blk2 = (j > 0 && AT(T, j, j - 1) != 0.0);
    if (!blk2) {
      cd piv = cd(AT(T, j, j), 0.0) - lam;
      if (std::abs(piv) < smin) piv = cd(smin, 0.0);
      x[j] = x[j] / piv;
      for (int i = 0; i < j; i++) x[i] -= x[j] * AT(T, i, j);
      j -= 1;
    } else {
      const cd a = cd(AT(T, j - 1, j - 1), 0.0) - lam, b = AT(T, j - 1, j), c = AT(T, j, j - 1), d = cd(AT(T, j, j), 0.0) - lam;
      cd det = a * d - b * c;
      if (std::abs(det) < smin * smin) det = cd(smin * smin, 0.0);
      const cd r1 = x[j - 1], r2 = x[j];
      x[j - 1] = (d * r1 - b * r2) / det;
      x[j] = (a * r2 - c * r1) / det;
      for (int i = 0; i < j - 1; i++) x[i] -= x[j - 1] * AT(T, i, j - 1) + x[j] * AT(T, i, j);
      j -= 2;
    }
  }
  // dtrevc scales so that the element of largest magnitude has magnitude 1 (|re|+|im| for pairs)
  double emax = 0.0;
  const int last = pair ? k + 1 : k;
  for (int i = 0; i <= last; i++) emax = std::max(emax, pair ? std::fabs(x[i].real()) + std::fabs(x[i].imag()) : std::fabs(x[i].real()));
  const double sc = emax > 0.0 ? 1.0 / emax : 1.0;
  for (int i = 0; i < n; i++) { xr[i] = (i <= last) ? x[i].real() * sc : 0.0; if (xi) xi[i] = (pair && i <= last) ? x[i].imag() * sc : 0.0; }
  return pair ? 1 : 0;
}

int potrf_upper(int n, double *A, int ld)
{
  for (int j = 0; j < n; j++) {
    double d = AT(A, j, j);
    for (int k = 0; k < j; k++) d -= AT(A, k, j) * AT(A, k, j);
    if (!(d > 0.0)) return j + 1;                       // also catches NaN
    d = std::sqrt(d); AT(A, j, j) = d;
    for (int c = j + 1; c < n; c++) {
      double t = AT(A, j, c);
      for (int k = 0; k < j; k++) t -= AT(A, k, j) * AT(A, k, c);
      AT(A, j, c) = t / d;
    }
  }
  return 0;
}

int trtri_upper(int n, double *A, int ld)
{
  for (int j = 0; j < n; j++) if (AT(A, j, j) == 0.0) return j + 1;
  for (int j = 0; j < n; j++) {                        // dtrti2: column j of the inverse from the leading j x j inverse
    AT(A, j, j) = 1.0 / AT(A, j, j);
    const double ajj = -AT(A, j, j);
    for (int i = 0; i < j; i++) {                      // x = inv(A(0:j,0:j)) * A(0:j,j) (upper triangular mat-vec, in place top-down)
      double t = 0.0;
      for (int k = i; k < j; k++) t += AT(A, i, k) * AT(A, k, j);
      AT(A, i, j) = t;
    }
    for (int i = 0; i < j; i++) AT(A, i, j) *= ajj;
  }
  return 0;
}

int sym_eig(int n, double *A, int ld, double *w)
{
  if (n <= 0) return 0;
  std::vector<double> S_((size_t)n * n), V_((size_t)n * n);
  double *S = S_.data(), *V = V_.data();
  for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) { S[i + j * n] = (i >= j) ? AT(A, i, j) : AT(A, j, i); V[i + j * n] = (i == j) ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0.0, dg = 0.0;
    for (int j = 0; j < n; j++) { dg += S[j + j * n] * S[j + j * n]; for (int i = j + 1; i < n; i++) off += S[i + j * n] * S[i + j * n]; }
    if (off <= 1e-34 * dg || off == 0.0) break;
    for (int p = 0; p < n - 1; p++)
      for (int q = p + 1; q < n; q++) {
        const double apq = S[p + q * n];
        if (apq == 0.0) continue;
        const double app = S[p + p * n], aqq = S[q + q * n];
        if (std::fabs(apq) < 1e-300) continue;
        const double theta = (aqq - app) / (2.0 * apq);
        const double t = sgn(1.0, theta) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s2 = t * c;
        for (int k = 0; k < n; k++) { const double skp = S[k + p * n], skq = S[k + q * n]; S[k + p * n] = c * skp - s2 * skq; S[k + q * n] = s2 * skp + c * skq; }
        for (int k = 0; k < n; k++) { const double spk = S[p + k * n], sqk = S[q + k * n]; S[p + k * n] = c * spk - s2 * sqk; S[q + k * n] = s2 * spk + c * sqk; }
        for (int k = 0; k < n; k++) { const double vkp = V[k + p * n], vkq = V[k + q * n]; V[k + p * n] = c * vkp - s2 * vkq; V[k + q * n] = s2 * vkp + c * vkq; }
      }
  }
  std::vector<int> idx_(n);
  int *idx = idx_.data();
  for (int i = 0; i < n; i++) idx[i] = i;
  std::sort(idx, idx + n, [&](int a, int b) { return S[a + a * n] < S[b + b * n]; });
  for (int j = 0; j < n; j++) { w[j] = S[idx[j] + idx[j] * n]; for (int i = 0; i < n; i++) AT(A, i, j) = V[i + idx[j] * n]; }
  return 0;
}

void qr_explicit(int M, int n, double *A, int ld, double *R, int ldr, double *Q, int ldq)
{
  std::vector<double> tau(n, 0.0);
  for (int j = 0; j < n; j++) {                                   // dgeqr2: H_j = I - tau v v^T, v(j) = 1, v(j+1:M) stored below the diagonal
    double *a = A + (size_t)j * ld;
    double xn2 = 0.0;
    for (int i = j + 1; i < M; i++) xn2 += a[i] * a[i];
    const double alpha = a[j];
    if (xn2 == 0.0) { tau[j] = 0.0; continue; }
    const double nrm = sqrt(alpha * alpha + xn2);
    const double beta = alpha >= 0.0 ? -nrm : nrm;
    tau[j] = (beta - alpha) / beta;
    const double sc = 1.0 / (alpha - beta);
    for (int i = j + 1; i < M; i++) a[i] *= sc;
    a[j] = beta;
    for (int c = j + 1; c < n; c++) {
      double *b = A + (size_t)c * ld;
      double w = b[j];
      for (int i = j + 1; i < M; i++) w += a[i] * b[i];
      w *= tau[j];
      b[j] -= w;
      for (int i = j + 1; i < M; i++) b[i] -= w * a[i];
    }
  }
  for (int c = 0; c < n; c++) for (int r = 0; r < n; r++) R[(size_t)r + (size_t)c * ldr] = r <= c ? A[(size_t)r + (size_t)c * ld] : 0.0;
  for (int c = 0; c < n; c++) { double *q = Q + (size_t)c * ldq; for (int i = 0; i < M; i++) q[i] = i == c ? 1.0 : 0.0; }     // dorg2r on [I; 0]
  for (int j = n - 1; j >= 0; j--) {
    if (tau[j] == 0.0) continue;
    const double *a = A + (size_t)j * ld;
    for (int c = 0; c < n; c++) {
      double *q = Q + (size_t)c * ldq;
      double w = q[j];
      for (int i = j + 1; i < M; i++) w += a[i] * q[i];
      w *= tau[j];
      q[j] -= w;
      for (int i = j + 1; i < M; i++) q[i] -= w * a[i];
    }
  }
}

void tsqr_combine(int n, double *R1, int ld1, double *R2, int ld2)
{
  // eliminate R2 row by row: row i of R2 is annihilated against rows i..n-1 of R1 (entries left of the diagonal are zero)
  for (int i = 0; i < n; i++)
    for (int j = i; j < n; j++) {
      double &a = R1[(size_t)j + (size_t)j * ld1], &b = R2[(size_t)i + (size_t)j * ld2];
      if (b == 0.0) continue;
      double c, s, r;
      lartg(a, b, c, s, r);
      a = r; b = 0.0;
      for (int k = j + 1; k < n; k++) {
        double &x = R1[(size_t)j + (size_t)k * ld1], &y = R2[(size_t)i + (size_t)k * ld2];
        const double t = c * x + s * y; y = c * y - s * x; x = t;
      }
    }
}

// P A = L U with row interchanges (unblocked right-looking dgetf2), then A^T x = b as U^T y = b, L^T z = y, x = P^T z
int lu_solve_trans(int n, double *A, int ld, double *b)
{
  std::vector<int> piv(n);
  int info = 0;
  for (int j = 0; j < n; j++) {
    int p = j; double mx = fabs(A[j + (size_t)j * ld]);
    for (int i = j + 1; i < n; i++) { const double v = fabs(A[i + (size_t)j * ld]); if (v > mx) { mx = v; p = i; } }
    piv[j] = p;
    if (mx == 0.0) { if (!info) info = j + 1; continue; }
    if (p != j) for (int c = 0; c < n; c++) std::swap(A[j + (size_t)c * ld], A[p + (size_t)c * ld]);
    const double d = 1.0 / A[j + (size_t)j * ld];
    for (int i = j + 1; i < n; i++) A[i + (size_t)j * ld] *= d;
    for (int c = j + 1; c < n; c++) {
      const double u = A[j + (size_t)c * ld];
      if (u != 0.0) for (int i = j + 1; i < n; i++) A[i + (size_t)c * ld] -= A[i + (size_t)j * ld] * u;
    }
  }
  if (info) return info;
  for (int i = 0; i < n; i++) {                       // U^T y = b: forward substitution over the columns of U
    double s = b[i];
    for (int r = 0; r < i; r++) s -= A[r + (size_t)i * ld] * b[r];
    b[i] = s / A[i + (size_t)i * ld];
  }
  for (int i = n - 1; i >= 0; i--) {                  // L^T z = y: unit lower triangle, backward
    double s = b[i];
    for (int r = i + 1; r < n; r++) s -= A[r + (size_t)i * ld] * b[r];
    b[i] = s;
  }
  for (int j = n - 1; j >= 0; j--) if (piv[j] != j) std::swap(b[j], b[piv[j]]);   // x = P^T z
  return 0;
}

} // namespace ksd

// ---- C hooks for the CPU unit tests (tests/test_dense_host.py builds this file alone with g++) ----------------
#ifdef KSD_TEST_HOOKS
extern "C" {
void ksd_hess_reduce(int n, int ilo, double *A, int ld, double *Q) { ksd::hess_reduce(n, ilo, A, ld, Q); }
int ksd_real_schur(int n, int ilo, double *A, int ld, double *wr, double *wi, double *Q) { return ksd::real_schur(n, ilo, A, ld, wr, wi, Q); }
int ksd_trexc_up(int n, double *T, int ld, double *Q, int ifst, int ilst) { return ksd::trexc_up(n, T, ld, Q, ifst, ilst); }
int ksd_trevc_one(int n, const double *T, int ld, int k, double *xr, double *xi) { return ksd::trevc_one(n, T, ld, k, xr, xi); }
int ksd_potrf_upper(int n, double *A, int ld) { return ksd::potrf_upper(n, A, ld); }
int ksd_trtri_upper(int n, double *A, int ld) { return ksd::trtri_upper(n, A, ld); }
int ksd_sym_eig(int n, double *A, int ld, double *w) { return ksd::sym_eig(n, A, ld, w); }
void ksd_tsqr_combine(int n, double *R1, int ld1, double *R2, int ld2) { ksd::tsqr_combine(n, R1, ld1, R2, ld2); }
void ksd_qr_explicit(int M, int n, double *A, int ld, double *R, int ldr, double *Q, int ldq) { ksd::qr_explicit(M, n, A, ld, R, ldr, Q, ldq); }
int ksd_lu_solve_trans(int n, double *A, int ld, double *b) { return ksd::lu_solve_trans(n, A, ld, b); }
}
#endif
