/* The reference's ex2 (src/eps/tutorials/ex2.c: 2-D Laplacian, -n 72 -eps_nev 4 -eps_ncv 20) written against the C ABI
   alone: plain C99, no C++ and no Python in between. Prints the eigenvalues in the -terse format of the reference. */
#include <stdio.h>
#include <stdlib.h>
#include "ksgpu.h"

#define CHK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s failed: %d (%s) %s\n", #call, rc_, ks_error_string(rc_), ks_last_error_message()); return 1; } } while (0)

int main(int argc, char **argv)
{
  int n = argc > 1 ? atoi(argv[1]) : 72, nconv = 0, its = 0, i;
  ks_ctx ctx; ks_mat A; ks_eps eps;
  CHK(ks_ctx_create(0, NULL, &ctx));
  CHK(ks_mat_create_laplacian2d(ctx, n, n, &A));
  CHK(ks_eps_create(ctx, &eps));
  CHK(ks_eps_set_operators(eps, A, NULL));
  CHK(ks_eps_set_problem_type(eps, KS_EPS_HEP));
  CHK(ks_eps_set_dimensions(eps, 4, 20, 0));
  CHK(ks_eps_solve(eps));
  CHK(ks_eps_get_converged(eps, &nconv));
  CHK(ks_eps_get_iteration_number(eps, &its));
  printf("2-D Laplacian Eigenproblem, N=%d (%dx%d grid)\n Number of iterations of the method: %d\n", n * n, n, n, its);
  printf(" All requested eigenvalues computed up to the required tolerance:\n    ");
  for (i = 0; i < 4 && i < nconv; i++) {
    double kr, ki, err;
    CHK(ks_eps_get_eigenvalue(eps, i, &kr, &ki));
    CHK(ks_eps_compute_error(eps, i, KS_EPS_ERROR_RELATIVE, &err));
    if (err > 1e-8) { fprintf(stderr, "residual %g too large\n", err); return 2; }
    printf("%s%.5f", i ? ", " : " ", kr);
  }
  printf("\n");
  CHK(ks_eps_destroy(eps)); CHK(ks_mat_destroy(A)); CHK(ks_ctx_destroy(ctx));
  return 0;
}
