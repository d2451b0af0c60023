// Small dense kernels for the projected (m x m, m <= 64) non-symmetric eigenproblem: the LAPACK routines the
// reference's DS NHEP calls (gehrd/orghr, hseqr, trexc, trevc; src/sys/classes/ds/impls/dsutil.c:21-175,
// src/sys/classes/ds/impls/nhep/dsnhep.c:101-167) written out for the host, since LAPACK is not part of this
// image's C toolchain. Column-major, 0-based indices, leading dimension ld.
#pragma once

namespace ksd {

// A(ilo:n, ilo:n) -> upper Hessenberg by Householder similarity (dgehd2), Q <- accumulated reflectors (dorghr);
// rows/columns below ilo are assumed already upper triangular. Q must come in as the identity.
void hess_reduce(int n, int ilo, double *A, int ld, double *Q);

// Real Schur form of the upper Hessenberg A (active window ilo..n-1): A <- T quasi-triangular with standardised
// 2x2 blocks, Q <- Q*Z, eigenvalues in wr/wi (dhseqr 'S','V' contract, algorithm of dlahqr). Returns 0 or the
// 1-based index where the QR iteration failed.
int real_schur(int n, int ilo, double *A, int ld, double *wr, double *wi, double *Q);

// Standardise a real 2x2 block (dlanv2).
void lanv2(double &a, double &b, double &c, double &d, double &rt1r, double &rt1i, double &rt2r, double &rt2i, double &cs, double &sn);

// Move the diagonal block starting at row ifst up to row ilst (ilst <= ifst) by adjacent swaps, updating Q
// (dtrexc 'V' for the upward direction used by DSSort_NHEP_Total). Returns 0, or 1 if a swap was rejected.
int trexc_up(int n, double *T, int ld, double *Q, int ifst, int ilst);

// Right eigenvector of the quasi-triangular T for the block starting at column k (dtrevc 'R','S'): xr (and xi for a
// complex pair; the pair is (k,k+1)) of length n, not back-transformed. Returns 1 for a complex pair, 0 for real.
int trevc_one(int n, const double *T, int ld, int k, double *xr, double *xi);

// ---- symmetric / triangular kernels of the block orthogonalisations (bvlapack.c:136-341) ----
// Upper Cholesky factor of the symmetric positive definite A (upper triangle referenced), in place (dpotrf 'U').
// Returns 0, or the 1-based order of the leading minor that is not positive definite.
int potrf_upper(int n, double *A, int ld);
// Inverse of an upper triangular matrix, in place (dtrtri 'U','N'). Returns 0, or the 1-based index of a zero pivot.
int trtri_upper(int n, double *A, int ld);
// Eigendecomposition of a symmetric matrix (lower triangle referenced): A <- eigenvectors (columns), w ascending
// (dsyev 'V','L' contract; cyclic Jacobi rotations, which are as accurate as QR for these n <= 64 Gram matrices).
int sym_eig(int n, double *A, int ld, double *w);
// R factor of the QR factorisation of the stacked upper triangles [R1; R2] (both n x n, column-major, ld), by Givens
// rotations, into R1 (the reduction operator of the parallel TSQR, SlepcGivensPacked bvlapack.c:456-478).
void tsqr_combine(int n, double *R1, int ld1, double *R2, int ld2);
// Solve A^T x = b for a general n x n matrix by LU with partial pivoting (dgetrf + dgetrs 'T'): A is overwritten by its
// factors, b by x. Returns 0, or the 1-based index of an exactly zero pivot.
int lu_solve_trans(int n, double *A, int ld, double *b);

// Householder QR of the M x n matrix A (column-major, ld; M >= n) with the orthogonal factor formed explicitly (dgeqr2 + dorg2r):
// R (n x n, upper triangular, ldr) and Q (M x n, ldq) with A = Q R. The combine step of the tall-skinny QR: A is the stack of the
// row blocks' triangular factors, block b of Q is what block b's reflectors are applied to when the panel's Q is formed.
void qr_explicit(int M, int n, double *A, int ld, double *R, int ldr, double *Q, int ldq);

} // namespace ksd
