/* The reference's BV test1 (src/sys/classes/bv/tests/test1.c, "-bv_type svec -verbose") written in C99 against the C ABI
   alone, so that the BV-level slots (mult, multvec, dot, dotvec, multinplace, scale, norm, getcolumn, getarray) are driven
   from C. The output follows the reference's viewers; the test-suite diffs it with output/test1_1_bv_type-svec.out. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "ksgpu.h"

#define CHK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s failed: %d (%s) %s\n", #call, rc_, ks_error_string(rc_), ks_last_error_message()); return 1; } } while (0)

/* PETSc's "%g" for reals: an integer-valued number keeps a trailing point */
static void pg(double x, const char *end)
{
  char b[64];
  snprintf(b, sizeof b, "%g", x);
  if (!strpbrk(b, ".en")) strcat(b, ".");
  printf("%s%s", b, end);
}

static int view_bv(ks_bv V, const char *name, int n, int m)
{
  double *col = (double *)malloc(sizeof(double) * (size_t)n);
  int i, j;
  printf("BV Object: %s 1 MPI process\n  type: svec\n", name);
  for (j = 0; j < m; j++) {
    CHK(ks_bv_get_column_host(V, j, col));
    printf("Vec Object: 1 MPI process\n  type: seq\n");
    for (i = 0; i < n; i++) pg(col[i], "\n");
  }
  free(col);
  return 0;
}

static void view_mat(const char *name, const double *a, int rows, int cols, int lda)
{
  int i, j;
  printf("Mat Object: %s 1 MPI process\n  type: seqdense\n", name);
  for (i = 0; i < rows; i++) {
    for (j = 0; j < cols; j++) printf("%18.16e ", a[i + j * lda]);
    printf("\n");
  }
}

int main(int argc, char **argv)
{
  int n = 10, k = 5, l = 3, i, j, ldx = 0, lda, ldm, testlda = argc > 1 && !strcmp(argv[1], "-testlda");
  ks_ctx ctx; ks_bv X, Y;
  double *col, *q, *M, *z, *pX, *v, nrm, first;

  CHK(ks_ctx_create(0, NULL, &ctx));
  printf("Test BV with %d columns of dimension %d.\n", k, n);
  CHK(ks_bv_create(ctx, n, n, k, 0, &X));
  printf("BV Object: X 1 MPI process\n  type: svec\n  %d columns of global length %d\n"
         "  vector orthogonalization method: classical Gram-Schmidt\n  orthogonalization refinement: if needed (eta: 0.7071)\n"
         "  block orthogonalization method: GS\n  doing matmult as a single matrix-matrix product\n", k, n);

  /* Fill X entries (test1.c:60-70) */
  col = (double *)malloc(sizeof(double) * (size_t)n);
  for (j = 0; j < k; j++) {
    for (i = 0; i < n; i++) col[i] = 0.0;
    for (i = 0; i < 4; i++) if (i + j < n) col[i + j] = (double)(3 * i + j - 2);
    CHK(ks_bv_set_column_host(X, j, col));
  }
  if (view_bv(X, "X", n, k)) return 1;

  /* Y (test1.c:73-85) */
  CHK(ks_bv_create(ctx, n, n, l, 0, &Y));
  for (j = 0; j < l; j++) {
    for (i = 0; i < n; i++) col[i] = (double)(j + 1) / 4.0;
    CHK(ks_bv_set_column_host(Y, j, col));
  }
  if (view_bv(Y, "Y", n, l)) return 1;

  /* Q (test1.c:88-102) */
  lda = testlda ? k + 2 : k;
  q = (double *)calloc((size_t)lda * (size_t)l, sizeof(double));
  for (i = 0; i < k; i++) for (j = 0; j < l; j++) q[i + j * lda] = (i < j) ? 2.0 : -0.5;
  view_mat("Q", q, k, l, lda);

  /* BVMult (test1.c:105) */
  CHK(ks_bv_mult(Y, 2.0, 1.0, X, q, lda));
  printf("After BVMult - - - - - - - - -\n");
  if (view_bv(Y, "Y", n, l)) return 1;

  /* BVMultVec on column 0 of Y (test1.c:112-118) */
  CHK(ks_bv_get_column(Y, 0, &v));
  z = (double *)malloc(sizeof(double) * (size_t)k);
  z[0] = 2.0;
  for (i = 1; i < k; i++) z[i] = -0.5 * z[i - 1];
  CHK(ks_bv_multvec(X, -1.0, 1.0, v, z));
  printf("After BVMultVec - - - - - - -\n");
  if (view_bv(Y, "Y", n, l)) return 1;

  /* BVDot (test1.c:125-134) */
  ldm = testlda ? l + 2 : l;
  M = (double *)calloc((size_t)ldm * (size_t)k, sizeof(double));
  CHK(ks_bv_dot(X, Y, M, ldm));
  printf("After BVDot - - - - - - - - -\n");
  view_mat("M", M, l, k, ldm);

  /* BVDotVec (test1.c:141-144) */
  CHK(ks_bv_dotvec(X, v, z));
  printf("After BVDotVec - - - - - - -\nVec Object: z 1 MPI process\n  type: seq\n");
  for (i = 0; i < k; i++) pg(z[i], "\n");

  /* BVMultInPlace and BVScale (test1.c:155-156) */
  CHK(ks_bv_multinplace(X, q, lda, 1, l));
  CHK(ks_bv_scale(X, 2.0));
  printf("After BVMultInPlace - - - - -\n");
  if (view_bv(X, "X", n, k)) return 1;

  /* BVNorm (test1.c:163-166) */
  CHK(ks_bv_normcolumn(X, 0, KS_NORM_2, &nrm));
  printf("2-Norm of X[0] = %g\n", nrm);
  CHK(ks_bv_norm(X, KS_NORM_FROBENIUS, &nrm));
  printf("Frobenius Norm of X = %g\n", nrm);

  /* BVGetArrayRead (test1.c:171-177) */
  printf("First row of X =\n");
  CHK(ks_bv_get_sizes(X, NULL, NULL, NULL, &ldx));
  CHK(ks_bv_get_array(X, &pX));
  for (i = 0; i < k; i++) { CHK(ks_ctx_memcpy(ctx, &first, pX + (size_t)i * (size_t)ldx, sizeof(double), 1)); pg(first, " "); }
  printf("\n");

  free(col); free(q); free(M); free(z);
  CHK(ks_bv_destroy(X)); CHK(ks_bv_destroy(Y)); CHK(ks_ctx_destroy(ctx));
  return 0;
}
