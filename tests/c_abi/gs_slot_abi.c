/* The ops->gramschmidt slot (include/slepc/private/bvimpl.h:53) driven from C the way its one caller drives it.
   caller_orthogonalize_column() below restates the CALLER's side - BVOrthogonalizeColumn (bvorthog.c:315-339) around
   BVOrthogonalizeGS (:145-217) with BV_CleanCoefficients / BV_SetValue on the coefficient buffer - and hands every pass to
   ks_bv_gramschmidt_pass exactly where the reference calls BVOrthogonalizeGS1 (:176,182,190,198,199): NULL onrm / nrm for
   REFINE_NEVER and the first REFINE_ALWAYS call, the |nrm| < eta |onrm| loop and lindep on the caller's side.
   Inputs are dyadic rationals (exact in C and in numpy): column 4 is column 0 plus a 2^-30 perturbation (needs refinement),
   column 6 = 2 x1 - 3 x2 (dependent), column 8 = 0. The test-suite compares the printed lines with the CPU oracle.
   usage: gs_slot_abi <refine 0|1|2> [mgs]                                                                                  */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "ksgpu.h"

#define CHK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s failed: %d (%s) %s\n", #call, rc_, ks_error_string(rc_), ks_last_error_message()); return 1; } } while (0)

enum { N = 2000, M = 9 };

static double entry(int i, int j) { return (double)(((i * 37 + j * 101 + ((i * i) % 13) * 7) % 17) - 8) * 0.0625; }

typedef struct { ks_ctx ctx; ks_bv bv; double *buffer; int nc, m; int refine, mgs; double eta; int passes; uint64_t state; } Caller;

/* what the adapter's HipksSync does at the head of every slot: mirror the caller's fields, among them the object state
   (PetscObjectStateGet): it stays put between the passes BVOrthogonalizeGS makes on a column and moves whenever a column changes */
#define PASS(c, j, onrm, nrm) (ks_bv_set_state((c)->bv, (c)->state) || ks_bv_gramschmidt_pass((c)->bv, (j), NULL, NULL, NULL, NULL, (onrm), (nrm)))

/* BV_CleanCoefficients(bv,j,NULL) bvimpl.h:289-301: zero column j of the buffer, entries 0..nc+j-1 */
static int clean_coefficients(Caller *c, int j)
{
  double zero[M + 1] = {0};
  if (c->nc + j > 0) CHK(ks_ctx_memcpy(c->ctx, c->buffer + (size_t)j * (size_t)(c->nc + c->m), zero, sizeof(double) * (size_t)(c->nc + j), 0));
  return 0;
}
/* BV_SetValue(bv,j,k,NULL,value) bvimpl.h:328-340 */
static int set_value(Caller *c, int j, int k, double value)
{
  CHK(ks_ctx_memcpy(c->ctx, c->buffer + (size_t)k * (size_t)(c->nc + c->m) + (size_t)(c->nc + j), &value, sizeof(double), 0));
  return 0;
}

static int caller_orthogonalize_column(Caller *c, int j, double *norm, int *lindep)
{
  double onrm = 0.0, nrm = 0.0;
  int l;
  if (clean_coefficients(c, j)) return 1;
  switch (c->refine) {
  case KS_BV_ORTHOG_REFINE_IFNEEDED:
    CHK(PASS(c, j, &onrm, &nrm)); c->passes++;
    l = 1;
    while (l < 3 && nrm != 0.0 && fabs(nrm) < c->eta * fabs(onrm)) {
      l++;
      if (c->mgs) onrm = nrm;                                                              /* bvorthog.c:181-182 */
      CHK(PASS(c, j, c->mgs ? NULL : &onrm, &nrm)); c->passes++;
    }
    *lindep = !(nrm != 0.0 && fabs(nrm) >= c->eta * fabs(onrm));
    break;
  case KS_BV_ORTHOG_REFINE_NEVER:
    CHK(PASS(c, j, NULL, NULL)); c->passes++;
    CHK(ks_bv_normcolumn(c->bv, j, KS_NORM_2, &nrm));
    *lindep = !(nrm != 0.0);
    break;
  default:
    CHK(PASS(c, j, NULL, NULL)); c->passes++;
    CHK(PASS(c, j, &onrm, &nrm)); c->passes++;
    *lindep = !(nrm != 0.0 && fabs(nrm) >= c->eta * fabs(onrm));
    break;
  }
  *norm = nrm;
  if (set_value(c, j, j, *lindep ? 0.0 : nrm)) return 1;
  c->state++;                                                  /* PetscObjectStateIncrease at the end of BVOrthogonalizeColumn (bvorthog.c:338) */
  return 0;
}

/* Pass chaining (ks_bv_set_state): column 3 = column 0 + a 2^-12 perturbation needs two passes. Three runs of the same two passes
   on fresh copies of the same four columns: with the state announced and unchanged (the second pass is chained to the dots the
   first one left), without any state (both passes take their own dots), and with the column rewritten between the passes behind
   the library's back - a copy through a pointer lent earlier - and the state bumped as PETSc would (the second pass must take
   its own dots of the NEW content). */
static int chain_case(ks_ctx ctx, int mode, double *col)
{
  enum { MC = 4 };
  ks_bv W; double *wbuf, *c3, zero[MC] = {0}, onrm1 = 0, nrm1 = 0, onrm2 = 0, nrm2 = 0, expect = 0, h[MC * MC];
  long long chained = 0, fresh = 0;
  uint64_t state = 100;
  int i, j;
  CHK(ks_bv_create(ctx, N, N, MC, 0, &W));
  CHK(ks_bv_set_orthogonalization(W, KS_BV_ORTHOG_CGS, KS_BV_ORTHOG_REFINE_IFNEEDED, 0.0));
  for (j = 0; j < MC; j++) {
    for (i = 0; i < N; i++) col[i] = j == 3 ? entry(i, 0) + ldexp(entry(i, 4), -12) : entry(i, j);
    CHK(ks_bv_set_column_host(W, j, col));
  }
  CHK(ks_bv_get_buffer(W, &wbuf));
  CHK(ks_bv_get_column(W, 3, &c3));                 /* lent before the passes, as a Vec obtained with BVGetColumn would be */
  for (j = 0; j < 3; j++) { double nr; int ld; CHK(ks_bv_orthonormalizecolumn(W, j, 0, &nr, &ld)); }
  CHK(ks_ctx_memcpy(ctx, wbuf + 3 * MC, zero, sizeof(double) * 3, 0));
  if (mode != 1) CHK(ks_bv_set_state(W, state));
  CHK(ks_bv_gramschmidt_pass(W, 3, NULL, NULL, NULL, NULL, &onrm1, &nrm1));
  if (mode == 2) {
    for (i = 0; i < N; i++) col[i] = entry(i, 7);
    CHK(ks_ctx_memcpy(ctx, c3, col, sizeof(double) * N, 0));
    state++;
    { double s = 0; for (i = 0; i < N; i++) s += col[i] * col[i]; expect = sqrt(s); }
  }
  if (mode != 1) CHK(ks_bv_set_state(W, state));
  CHK(ks_bv_gramschmidt_pass(W, 3, NULL, NULL, NULL, NULL, &onrm2, &nrm2));
  CHK(ks_bv_gs_chain_stats(W, &chained, &fresh));
  CHK(ks_bv_get_buffer_host(W, h));
  printf("chain mode %d onrm1 %.17g nrm1 %.17g onrm2 %.17g nrm2 %.17g chained %lld fresh %lld expect %.17g h %.17g %.17g %.17g\n", mode, onrm1, nrm1, onrm2, nrm2, chained, fresh, expect,
         h[3 * MC], h[3 * MC + 1], h[3 * MC + 2]);
  CHK(ks_bv_get_column_host(W, 3, col));
  { double s = 0; for (i = 0; i < N; i++) s += col[i] * col[i]; printf("chain mode %d column_norm %.17g\n", mode, sqrt(s)); }
  CHK(ks_bv_destroy(W));
  return 0;
}

int main(int argc, char **argv)
{
  Caller c;
  ks_bv bufbv;
  double *col, *x0, *x1, *x2, H[(M + 1) * (M + 1)];
  int i, j, refine = argc > 1 ? atoi(argv[1]) : 0, mgs = argc > 2 && !strcmp(argv[2], "mgs");

  CHK(ks_ctx_create(0, NULL, &c.ctx));
  CHK(ks_bv_create(c.ctx, N, N, M, 0, &c.bv));
  CHK(ks_bv_set_orthogonalization(c.bv, mgs ? KS_BV_ORTHOG_MGS : KS_BV_ORTHOG_CGS, refine, 0.0));
  /* as the adapter does with the device array of the reference's bv->buffer Vec (HipksSync): the library adopts a caller-owned
     coefficient buffer; here the storage of a helper BV stands in for that Vec */
  CHK(ks_bv_create(c.ctx, M * M, M * M, 1, 0, &bufbv));
  CHK(ks_bv_get_column(bufbv, 0, &c.buffer));
  CHK(ks_bv_set_buffer(c.bv, c.buffer));
  { double *chk = NULL; CHK(ks_bv_get_buffer(c.bv, &chk)); if (chk != c.buffer) { fprintf(stderr, "the adopted buffer is not in use\n"); return 3; } }
  c.nc = 0; c.m = M; c.refine = refine; c.mgs = mgs; c.eta = 0.7071; c.passes = 0; c.state = 1;

  col = (double *)malloc(sizeof(double) * N); x0 = (double *)malloc(sizeof(double) * N);
  x1 = (double *)malloc(sizeof(double) * N); x2 = (double *)malloc(sizeof(double) * N);
  for (i = 0; i < N; i++) { x0[i] = entry(i, 0); x1[i] = entry(i, 1); x2[i] = entry(i, 2); }
  for (j = 0; j < M; j++) {
    for (i = 0; i < N; i++) {
      if (j == 4) col[i] = x0[i] + ldexp(entry(i, 4), -30);
      else if (j == 6) col[i] = 2.0 * x1[i] - 3.0 * x2[i];
      else if (j == 8) col[i] = 0.0;
      else col[i] = entry(i, j);
    }
    CHK(ks_bv_set_column_host(c.bv, j, col));
  }

  for (j = 0; j < M; j++) {
    double norm = 0.0; int lindep = 0, before = c.passes;
    /* the caller sets the window to [-nc, j) around the call (bvorthog.c:327-333) */
    CHK(ks_bv_set_active_columns(c.bv, 0, M));
    if (caller_orthogonalize_column(&c, j, &norm, &lindep)) return 1;
    /* BVOrthonormalizeColumn (bvorthog.c:417-419); a dependent column - column 6 by construction, whatever the flag says of the
       rounding noise that is left of it - is zeroed so that both sides go on with the same basis */
    if (lindep || norm == 0.0 || j == 6) CHK(ks_bv_scalecolumn(c.bv, j, 0.0));
    else CHK(ks_bv_scalecolumn(c.bv, j, 1.0 / norm));
    c.state++;                                                 /* BVScaleColumn: PetscObjectStateIncrease (bvops.c:356) */
    printf("column %d passes %d lindep %d norm %.17g\n", j, c.passes - before, lindep, norm);
  }
  { long long chained = 0, fresh = 0; CHK(ks_bv_gs_chain_stats(c.bv, &chained, &fresh)); printf("chainstats chained %lld fresh %lld\n", chained, fresh); }
  CHK(ks_bv_get_buffer_host(c.bv, H));
  for (j = 0; j < M; j++) {
    printf("H[:,%d]", j);
    for (i = 0; i <= j; i++) printf(" %.17g", H[i + j * M]);
    printf("\n");
  }
  /* the vector form of the slot with host h / c (BVOrthogonalizeVec -> bv->h, bv->c, bvorthog.c:247-269): a copy of column 3's
     original content against the first three (now orthonormal) columns, one pass with both norms */
  {
    double h[M] = {0}, cc[M] = {0}, onrm = 0.0, nrm = 0.0, *w;
    ks_bv W;
    CHK(ks_bv_create(c.ctx, N, N, 1, 0, &W));
    for (i = 0; i < N; i++) col[i] = entry(i, 3);
    CHK(ks_bv_set_column_host(W, 0, col));
    CHK(ks_bv_get_column(W, 0, &w));
    CHK(ks_bv_set_active_columns(c.bv, 0, 3));
    CHK(ks_bv_gramschmidt_pass(c.bv, 3, w, NULL, h, cc, &onrm, &nrm));
    printf("vector onrm %.17g nrm %.17g h %.17g %.17g %.17g\n", onrm, nrm, h[0], h[1], h[2]);
    CHK(ks_bv_destroy(W));
  }
  /* constraints as the interface layer sets them up: storage of nc + m columns, the constraint vectors in front, then only the
     FIELDS change (BVSetNumConstraints bvbasic.c:291-296, mirrored with ks_bv_set_layout); the slot then orthogonalizes column 0
     against columns -nc..-1 and its coefficients go to rows 0..nc-1 of buffer column 0 (BV_AddCoefficients with nc+j entries) */
  {
    enum { NC = 2, MM = 3 };
    ks_bv W; double *wbuf, hc0[NC + MM], nrm = 0.0, onrm = 0.0, dots[NC + MM], *w0;
    double zero[NC + MM] = {0};
    CHK(ks_bv_create(c.ctx, N, N, NC + MM, 0, &W));
    for (j = 0; j < NC; j++) {                      /* two orthonormal constraint vectors: e_j-like blocks */
      double s = 1.0 / sqrt((double)(N / 2));
      for (i = 0; i < N; i++) col[i] = ((i % 2) == j) ? s : 0.0;
      CHK(ks_bv_set_column_host(W, j, col));
    }
    for (i = 0; i < N; i++) col[i] = entry(i, 5);
    CHK(ks_bv_set_column_host(W, NC, col));         /* becomes regular column 0 */
    for (i = 0; i < N; i++) col[i] = entry(i, 6);
    CHK(ks_bv_set_column_host(W, NC + 1, col));     /* regular column 1 */
    CHK(ks_bv_set_layout(W, NC, MM));
    CHK(ks_bv_get_buffer(W, &wbuf));
    /* column 0: one pass against the two constraints, then normalise (its own coefficient column is the scratch column: the
       reference's BV_AddCoefficients adds that column to itself for j = 0, so nothing is read from it here) */
    CHK(ks_ctx_memcpy(c.ctx, wbuf, zero, sizeof(double) * NC, 0));
    CHK(ks_bv_gramschmidt_pass(W, 0, NULL, NULL, NULL, NULL, &onrm, &nrm));
    CHK(ks_bv_scalecolumn(W, 0, 1.0 / nrm));
    /* column 1 against the constraints and column 0: BV_CleanCoefficients(bv,1,NULL) = nc+1 entries of buffer column 1 */
    CHK(ks_ctx_memcpy(c.ctx, wbuf + (NC + MM), zero, sizeof(double) * (NC + 1), 0));
    CHK(ks_bv_gramschmidt_pass(W, 1, NULL, NULL, NULL, NULL, &onrm, &nrm));
    CHK(ks_ctx_memcpy(c.ctx, hc0, wbuf + (NC + MM), sizeof(double) * (NC + 1), 1));
    /* what is left of column 1 is orthogonal to the constraints and to column 0: look at the same storage as plain columns */
    CHK(ks_bv_set_layout(W, 0, NC + MM));
    CHK(ks_bv_set_active_columns(W, 0, NC + 1));
    CHK(ks_bv_get_column(W, NC + 1, &w0));
    CHK(ks_bv_dotvec(W, w0, dots));
    printf("constraints onrm %.17g nrm %.17g h %.17g %.17g %.17g residual_dots %.3e %.3e %.3e\n", onrm, nrm, hc0[0], hc0[1], hc0[2], dots[0], dots[1], dots[2]);
    CHK(ks_bv_destroy(W));
  }
  if (!mgs && refine == 0) { for (i = 0; i < 3; i++) if (chain_case(c.ctx, i, col)) return 1; }
  free(col); free(x0); free(x1); free(x2);
  CHK(ks_bv_destroy(c.bv)); CHK(ks_bv_destroy(bufbv)); CHK(ks_ctx_destroy(c.ctx));
  return 0;
}
