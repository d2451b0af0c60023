/*
 * ks_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY; never shipped, never on the product path).
 *
 * A plain-C restatement of the arithmetic of the SLEPc 3.22.2 Krylov-Schur hot path:
 *   - PETSc MatMult(SeqAIJ) CSR SpMV               (PETSc is NOT in /root/reference: third-party,
 *                                                   restated as y_i = sum_p val[p]*x[col[p]] in row order)
 *   - BV kernels as the reference's CPU path does them (src/sys/classes/bv/interface/bvblas.c,
 *     bvlapack.c) with netlib-reference-BLAS loop order for gemv/gemm
 *   - Gram-Schmidt (CGS/MGS + refinement)            src/sys/classes/bv/interface/bvorthog.c
 *   - BVMatArnoldi / BVMatLanczos                    src/sys/classes/bv/interface/bvkrylov.c
 *   - coefficient-buffer helpers                     include/slepc/private/bvimpl.h:121-141,289-415
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 * Scalars are real double (PetscScalar = PetscReal = double), indices 32-bit int (PetscInt).
 *
 * OpenMP (-fopenmp) is used only to row-split the SpMV / gemv loops for the timed CPU baseline;
 * it is the equivalent of the reference's one-MPI-rank-per-core row decomposition. The
 * single-thread build performs the operations in exactly the loop order written here.
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <float.h>
#include <stdint.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_OK            0
#define ORC_ERR_ARG       1
#define ORC_ERR_INNERPROD 2   /* BV_SafeSqrt: "Invalid inner product" (bvimpl.h:137) */

enum { ORC_CGS = 0, ORC_MGS = 1 };                               /* BVOrthogType   (slepcbv.h) */
enum { ORC_REFINE_IFNEEDED = 0, ORC_REFINE_NEVER = 1, ORC_REFINE_ALWAYS = 2 }; /* BVOrthogRefineType */
enum { ORC_NORM_1 = 0, ORC_NORM_2 = 1, ORC_NORM_FROBENIUS = 2, ORC_NORM_INFINITY = 3 }; /* PETSc NormType */

/* CSR matrix (PETSc SeqAIJ layout: i = rowptr, j = col, a = val) */
typedef struct {
  int n, ncols; int nnz;
  const int *rowptr; const int *col; const double *val;
} orc_csr;

/* struct _p_BV (include/slepc/private/bvimpl.h:63-113), the fields the path uses */
typedef struct {
  int     n, N;          /* local/global rows (single rank in the oracle: n == N) */
  int     m;             /* number of columns */
  int     l, k;          /* leading / active columns */
  int     nc;            /* constraints: columns -nc..-1 of the storage (BVInsertConstraints bvfunc.c:411) */
  int     ld;            /* leading dimension */
  int     orthog_type, orthog_ref;
  double  orthog_eta;
  double  deftol;        /* 10*eps, bvfunc.c:184 */
  double *array;         /* (nc+m)*ld, column-major: BV_SVEC storage svec.c:397-565 */
  double *buffer;        /* (nc+m)*m: column 0 = scratch c, column j = H(:,j)  (bvbasic.c:775-791) */
  double *h, *c;         /* nc+m each: coefficients for BVOrthogonalizeVec (bvimpl.h:205-211) */
  double *work;  int lwork;
  int     passes_last;   /* instrumentation: number of GS passes of the last orthogonalization */
  long    passes_total;
  const orc_csr *matrix; /* inner-product matrix B of BVSetMatrix (positive definite; indef = FALSE), or NULL */
  double *Bx;            /* B*x of the vector last used in an inner product (BV_IPMatMult bvimpl.h:147-158) */
} orc_bv;

/* ------------------------------------------------------------------------------------------- */
/* BV lifecycle                                                                                 */

/* BV_SetDefaultLD (bvimpl.h:471-484): ld = n rounded so ld*8 is a multiple of max(PETSC_MEMALIGN,16) */
static int orc_default_ld(int n) { size_t bytes = ((size_t)n*sizeof(double)+15) & ~(size_t)15; return (int)(bytes/sizeof(double)); }

orc_bv *orc_bv_create(int n, int m, int ld)
{
  orc_bv *bv = (orc_bv*)calloc(1,sizeof(orc_bv));
  bv->n = bv->N = n; bv->m = m; bv->l = 0; bv->k = m; bv->nc = 0;     /* bvfunc.c:176-184 defaults */
  bv->ld = ld ? ld : orc_default_ld(n);
  bv->orthog_type = ORC_CGS; bv->orthog_ref = ORC_REFINE_IFNEEDED; bv->orthog_eta = 0.7071;
  bv->deftol = 10*DBL_EPSILON;
  bv->array  = (double*)calloc((size_t)m*bv->ld,sizeof(double));
  bv->buffer = (double*)calloc((size_t)m*m,sizeof(double));
  bv->h = (double*)calloc(m,sizeof(double)); bv->c = (double*)calloc(m,sizeof(double));
  return bv;
}
void orc_bv_destroy(orc_bv *bv) { if (!bv) return; free(bv->array); free(bv->buffer); free(bv->h); free(bv->c); free(bv->work); free(bv->Bx); free(bv); }
void orc_csr_mult(int n,const int *rowptr,const int *col,const double *val,const double *x,double *y);
/* BVSetMatrix bvfunc.c:200-250 (indef = PETSC_FALSE) */
void orc_bv_set_matrix(orc_bv *bv,const orc_csr *B) { bv->matrix=B; if (B && !bv->Bx) bv->Bx=(double*)calloc((size_t)bv->n+1,sizeof(double)); }
/* BV_IPMatMult bvimpl.h:147-158 (no caching by vector id: recomputed on every use) */
static const double *orc_ipmatmult(orc_bv *bv,const double *x)
{ if (!bv->matrix) return x; orc_csr_mult(bv->matrix->n,bv->matrix->rowptr,bv->matrix->col,bv->matrix->val,x,bv->Bx); return bv->Bx; }
double *orc_bv_array(orc_bv *bv)  { return bv->array; }
double *orc_bv_buffer(orc_bv *bv) { return bv->buffer; }
double *orc_bv_column(orc_bv *bv,int j) { return bv->array + (size_t)(bv->nc+j)*bv->ld; }   /* svec.c:292-303 */
int  orc_bv_ld(orc_bv *bv) { return bv->ld; }
void orc_bv_set_active(orc_bv *bv,int l,int k) { bv->l = l; bv->k = k; }                      /* bvbasic.c:421 */
void orc_bv_set_orthog(orc_bv *bv,int type,int ref,double eta) { bv->orthog_type=type; bv->orthog_ref=ref; bv->orthog_eta=eta; }
int  orc_bv_passes_last(orc_bv *bv) { return bv->passes_last; }
long orc_bv_passes_total(orc_bv *bv) { return bv->passes_total; }

static double *orc_work(orc_bv *bv,int len) { if (len>bv->lwork) { free(bv->work); bv->work=(double*)malloc((size_t)len*sizeof(double)); bv->lwork=len; } return bv->work; }

/* ------------------------------------------------------------------------------------------- */
/* BLAS-level kernels, netlib reference loop order                                              */

/* y := alpha*A*x + beta*y, A n x k (lda)  -- dgemv('N'); used by BVMultVec_BLAS_Private bvblas.c:56-67 */
static void orc_gemv_n(int n,int k,double alpha,const double *A,int lda,const double *x,double beta,double *y)
{
  int i,j;
  if (beta!=1.0) { if (beta==0.0) for (i=0;i<n;i++) y[i]=0.0; else for (i=0;i<n;i++) y[i]*=beta; }
  if (alpha==0.0) return;
#ifdef _OPENMP
  /* row-block split (one block per thread = the reference's one-MPI-rank-per-core row decomposition); inside a block the
     netlib column order, so the block of y stays in the core's cache while the k column segments stream through */
  #pragma omp parallel private(i,j)
  {
    int nt=omp_get_num_threads(), t=omp_get_thread_num();
    long lo=(long)n*t/nt, hi=(long)n*(t+1)/nt;
    for (j=0;j<k;j++) { double tt=alpha*x[j]; const double *a=A+(size_t)j*lda; for (i=(int)lo;i<(int)hi;i++) y[i]+=tt*a[i]; }
  }
#else
  for (j=0;j<k;j++) { double t=alpha*x[j]; const double *a=A+(size_t)j*lda; for (i=0;i<n;i++) y[i]+=t*a[i]; }
#endif
}

/* y := A'*x, A n x k (lda) -- dgemv('C') with alpha=1,beta=0; BVDotVec_BLAS_Private bvblas.c:240-261 */
static void orc_gemv_t(int n,int k,const double *A,int lda,const double *x,double *y)
{
  int j;
#ifdef _OPENMP
  /* row-block split with per-thread partial sums added in thread order (the MPI_Allreduce of bvblas.c:255) */
  int nt=omp_get_max_threads();
  double *part=(double*)calloc((size_t)nt*(k>0?k:1),sizeof(double));
  #pragma omp parallel private(j)
  {
    int t=omp_get_thread_num(), ntt=omp_get_num_threads(); long lo=(long)n*t/ntt, hi=(long)n*(t+1)/ntt; long i;
    for (j=0;j<k;j++) { const double *a=A+(size_t)j*lda; double s=0.0; for (i=lo;i<hi;i++) s+=a[i]*x[i]; part[(size_t)t*k+j]=s; }
  }
  for (j=0;j<k;j++) { double s=0.0; int t; for (t=0;t<nt;t++) s+=part[(size_t)t*k+j]; y[j]=s; }
  free(part);
#else
  for (j=0;j<k;j++) { const double *a=A+(size_t)j*lda; double t=0.0; int i; for (i=0;i<n;i++) t+=a[i]*x[i]; y[j]=t; }
#endif
}

/* C := alpha*A*B + beta*C  (dgemm 'N','N'); A m x k, B k x n, C m x n */
static void orc_gemm_nn(int m,int n,int k,double alpha,const double *A,int lda,const double *B,int ldb,double beta,double *C,int ldc)
{
  int i,j,l;
  for (j=0;j<n;j++) {
    double *c=C+(size_t)j*ldc;
    if (beta==0.0) for (i=0;i<m;i++) c[i]=0.0; else if (beta!=1.0) for (i=0;i<m;i++) c[i]*=beta;
    for (l=0;l<k;l++) { double t=alpha*B[l+(size_t)j*ldb]; const double *a=A+(size_t)l*lda; for (i=0;i<m;i++) c[i]+=t*a[i]; }
  }
}
/* C := A*B' (dgemm 'N','C', alpha=1, beta=0) */
static void orc_gemm_nt(int m,int n,int k,const double *A,int lda,const double *B,int ldb,double *C,int ldc)
{
  int i,j,l;
  for (j=0;j<n;j++) {
    double *c=C+(size_t)j*ldc;
    for (i=0;i<m;i++) c[i]=0.0;
    for (l=0;l<k;l++) { double t=B[j+(size_t)l*ldb]; const double *a=A+(size_t)l*lda; for (i=0;i<m;i++) c[i]+=t*a[i]; }
  }
}
/* C := A'*B (dgemm 'C','N', alpha=1, beta=0); A k x m, B k x n, C m x n */
static void orc_gemm_tn(int m,int n,int k,const double *A,int lda,const double *B,int ldb,double *C,int ldc)
{
  int i,j,l;
  for (j=0;j<n;j++) for (i=0;i<m;i++) {
    const double *a=A+(size_t)i*lda,*b=B+(size_t)j*ldb; double t=0.0;
    for (l=0;l<k;l++) t+=a[l]*b[l];
    C[i+(size_t)j*ldc]=t;
  }
}

/* dlassq-style scaled sum of squares, as LAPACK dlange('F') does (BVNorm_LAPACK_Private bvlapack.c:37-52) */
static void orc_lassq(int n,const double *x,double *scale,double *sumsq)
{
  int i;
  for (i=0;i<n;i++) {
    double a=fabs(x[i]);
    if (a>0.0 || isnan(a)) {
      if (*scale<a) { double r=*scale/a; *sumsq=1.0+*sumsq*r*r; *scale=a; }
      else { double r=a/(*scale); *sumsq+=r*r; }
    }
  }
}

/* ------------------------------------------------------------------------------------------- */
/* BV ops (interface semantics from bvops.c / bvglobal.c, storage arithmetic from svec.c)        */

/* BVMult: Y = beta*Y + alpha*X*Q   (bvops.c:49, BVMult_Svec svec.c:17-36, BVMult_BLAS_Private bvblas.c:24-49;
   Q==NULL -> BVAXPY_BLAS_Private bvblas.c:163-192) */
int orc_bv_mult(orc_bv *Y,double alpha,double beta,orc_bv *X,const double *Q,int ldq)
{
  double *py=Y->array+(size_t)(Y->nc+Y->l)*Y->ld; const double *px=X->array+(size_t)(X->nc+X->l)*X->ld;
  int i,j;
  if (X==Y || X->n!=Y->n) return ORC_ERR_ARG;
  if (Q) orc_gemm_nn(Y->n,Y->k-Y->l,X->k-X->l,alpha,px,X->ld,Q+(size_t)Y->l*ldq+X->l,ldq,beta,py,Y->ld);
  else {
    for (j=0;j<Y->k-Y->l;j++) for (i=0;i<Y->n;i++) {
      if (beta!=1.0) py[i+(size_t)j*Y->ld] = alpha*px[i+(size_t)j*X->ld] + beta*py[i+(size_t)j*Y->ld];
      else           py[i+(size_t)j*Y->ld] += alpha*px[i+(size_t)j*X->ld];
    }
  }
  return ORC_OK;
}

/* BVMultVec: y = beta*y + alpha*X(:,l:k)*q; q==NULL -> buffer scratch  (bvops.c:110, BVMultVec_Svec svec.c:38-52) */
int orc_bv_multvec(orc_bv *X,double alpha,double beta,double *y,const double *q)
{
  const double *qq = q ? q : X->buffer;
  orc_gemv_n(X->n,X->k-X->l,alpha,X->array+(size_t)(X->nc+X->l)*X->ld,X->ld,qq,beta,y);
  return ORC_OK;
}

/* BVMultColumn (bvops.c:165-198): temporarily k=j, y = column j */
int orc_bv_multcolumn(orc_bv *X,double alpha,double beta,int j,const double *q)
{
  int ksave=X->k,ierr; if (j<0 || j>=X->m) return ORC_ERR_ARG;
  X->k=j; ierr=orc_bv_multvec(X,alpha,beta,orc_bv_column(X,j),q); X->k=ksave; return ierr;
}

/* BVMultInPlace: V(:,s:e-1) = V(:,l:k-1)*Q(l:k-1,s:e-1)   (bvops.c:220, BVMultInPlace_Svec svec.c:54-70,
   BVMultInPlace_BLAS_Private bvblas.c:74-106: row blocks of 64 through a 64 x (e-s) workspace) */
int orc_bv_multinplace(orc_bv *V,const double *Q,int ldq,int s,int e,int trans)
{
  const int bs=64;
  int m=V->n,k=V->k-V->l,ss=s-V->l,ee=e-V->l,n=ee-ss,l,j,lda=V->ld;
  double *A=V->array+(size_t)(V->nc+V->l)*V->ld,*work; const double *B=Q+(size_t)V->l*ldq+V->l,*pb;
  if (s<V->l || s>V->m || e<V->l || e>V->m) return ORC_ERR_ARG;
  if (s>=e || !V->n) return ORC_OK;
  work=orc_work(V,bs*n);
  pb = trans ? B+ss : B+(size_t)ss*ldq;
  l=m%bs;
  if (l) {
    if (trans) orc_gemm_nt(l,n,k,A,lda,pb,ldq,work,l); else orc_gemm_nn(l,n,k,1.0,A,lda,pb,ldq,0.0,work,l);
    for (j=0;j<n;j++) memcpy(A+(size_t)(ss+j)*lda,work+(size_t)j*l,(size_t)l*sizeof(double));
  }
  /* the 64-row blocks are independent (each reads and writes its own rows): in the OpenMP build they are split over the team, one
     workspace per thread - the decomposition the reference gets from its MPI ranks, each running this loop on its own rows */
#ifdef _OPENMP
#pragma omp parallel private(j)
  {
    double *wk=(double*)malloc((size_t)bs*n*sizeof(double)+8);
    long lb,l0=l;
#pragma omp for schedule(static)
    for (lb=0;lb<(m-l0)/bs;lb++) {
      const int ll=(int)(l0+lb*bs);
      if (trans) orc_gemm_nt(bs,n,k,A+ll,lda,pb,ldq,wk,bs); else orc_gemm_nn(bs,n,k,1.0,A+ll,lda,pb,ldq,0.0,wk,bs);
      for (j=0;j<n;j++) memcpy(A+(size_t)(ss+j)*lda+ll,wk+(size_t)j*bs,(size_t)bs*sizeof(double));
    }
    free(wk);
  }
#else
  for (;l<m;l+=bs) {
    if (trans) orc_gemm_nt(bs,n,k,A+l,lda,pb,ldq,work,bs); else orc_gemm_nn(bs,n,k,1.0,A+l,lda,pb,ldq,0.0,work,bs);
    for (j=0;j<n;j++) memcpy(A+(size_t)(ss+j)*lda+l,work+(size_t)j*bs,(size_t)bs*sizeof(double));
  }
#endif
  return ORC_OK;
}

/* BVDot: M(Y->l:Y->k, X->l:X->k) = Y^H X   (bvglobal.c:86, BVDot_Svec svec.c:89-107, BVDot_BLAS_Private bvblas.c:199-233) */
int orc_bv_dot(orc_bv *X,orc_bv *Y,double *M,int ldm)
{
  if (X->n!=Y->n) return ORC_ERR_ARG;
  if (X->l==X->k || Y->l==Y->k) return ORC_OK;
  if (X->matrix) {                                       /* bvglobal.c:103-107: cached = B*X, M = Y^H cached */
    int j,nx=X->k-X->l; double *W=(double*)malloc((size_t)X->n*nx*sizeof(double)+8);
    for (j=0;j<nx;j++) orc_csr_mult(X->matrix->n,X->matrix->rowptr,X->matrix->col,X->matrix->val,X->array+(size_t)(X->nc+X->l+j)*X->ld,W+(size_t)j*X->n);
    orc_gemm_tn(Y->k-Y->l,nx,X->n,Y->array+(size_t)(Y->nc+Y->l)*Y->ld,Y->ld,W,X->n,M+(size_t)X->l*ldm+Y->l,ldm);
    free(W); return ORC_OK;
  }
  orc_gemm_tn(Y->k-Y->l,X->k-X->l,X->n,Y->array+(size_t)(Y->nc+Y->l)*Y->ld,Y->ld,X->array+(size_t)(X->nc+X->l)*X->ld,X->ld,M+(size_t)X->l*ldm+Y->l,ldm);
  return ORC_OK;
}

/* BVDotVec: m = X(:,l:k)^H y; m==NULL -> buffer scratch   (bvglobal.c:151, BVDotVec_Svec svec.c:109-129) */
int orc_bv_dotvec(orc_bv *X,const double *y,double *m)
{
  double *qq = m ? m : X->buffer;
  const double *z = orc_ipmatmult(X,y);                  /* svec.c:117-120: z = B*y when a matrix is set */
  orc_gemv_t(X->n,X->k-X->l,X->array+(size_t)(X->nc+X->l)*X->ld,X->ld,z,qq);
  return ORC_OK;
}

/* BVDotColumn (bvglobal.c:302-327): temporarily k=j, y = column j */
int orc_bv_dotcolumn(orc_bv *X,int j,double *q)
{
  int ksave=X->k,ierr; if (j<0 || j>=X->m) return ORC_ERR_ARG;
  X->k=j; ierr=orc_bv_dotvec(X,orc_bv_column(X,j),q); X->k=ksave; return ierr;
}

/* BVScale / BVScaleColumn (bvops.c:311,341; BVScale_Svec svec.c:150-162; BVScale_BLAS_Private bvblas.c:266-278) */
static void orc_scal(int n,double *A,double alpha)
{
  int i;
  if (alpha==0.0) memset(A,0,(size_t)n*sizeof(double));
  else if (alpha!=1.0) {
#ifdef _OPENMP
    #pragma omp parallel for schedule(static)
#endif
    for (i=0;i<n;i++) A[i]*=alpha;
  }
}
int orc_bv_scale(orc_bv *bv,int j,double alpha)
{
  if (alpha==1.0) return ORC_OK;                         /* bvops.c:318,351 */
  if (j<0) orc_scal((bv->k-bv->l)*bv->ld,bv->array+(size_t)(bv->nc+bv->l)*bv->ld,alpha);
  else { if (j>=bv->m) return ORC_ERR_ARG; orc_scal(bv->n,orc_bv_column(bv,j),alpha); }
  return ORC_OK;
}

/* BVNorm / BVNormColumn (bvglobal.c:498,662; BVNorm_Svec svec.c:164-176; BVNorm_LAPACK_Private bvlapack.c:37-83) */
static int orc_norm_vec_or_column(orc_bv *bv,int j,double *v,double *nrm);
int orc_bv_norm(orc_bv *bv,int j,int type,double *val)
{
  const double *A; int ncols,i,c,n=bv->n,lda=bv->ld;
  if (bv->matrix && j>=0) { if (j>=bv->m) return ORC_ERR_ARG; return orc_norm_vec_or_column(bv,j,NULL,val); }   /* bvglobal.c:683-687 */
  if (j<0) { A=bv->array+(size_t)(bv->nc+bv->l)*bv->ld; ncols=bv->k-bv->l; }
  else { if (j>=bv->m) return ORC_ERR_ARG; A=orc_bv_column(bv,j); ncols=1; }
  if (type==ORC_NORM_FROBENIUS || type==ORC_NORM_2) {
    double scale=0.0,sumsq=1.0;
    if (type==ORC_NORM_2 && j<0) return ORC_ERR_ARG;     /* bvglobal.c:506 "Requested norm not available" */
    for (c=0;c<ncols;c++) orc_lassq(n,A+(size_t)c*lda,&scale,&sumsq);
    *val=scale*sqrt(sumsq);
  } else if (type==ORC_NORM_1) {
    double mx=0.0; for (c=0;c<ncols;c++) { double s=0.0; for (i=0;i<n;i++) s+=fabs(A[i+(size_t)c*lda]); if (s>mx) mx=s; } *val=mx;
  } else if (type==ORC_NORM_INFINITY) {
    double mx=0.0; for (i=0;i<n;i++) { double s=0.0; for (c=0;c<ncols;c++) s+=fabs(A[i+(size_t)c*lda]); if (s>mx) mx=s; } *val=mx;
  } else return ORC_ERR_ARG;
  return ORC_OK;
}

/* BVCopy / BVCopyColumn (BVCopy_Svec svec.c:232-247, BVCopyColumn_Svec :249-259) */
int orc_bv_copy(orc_bv *V,orc_bv *W)
{
  int j; if (V->n!=W->n || V->k-V->l!=W->k-W->l) return ORC_ERR_ARG;
  for (j=0;j<V->k-V->l;j++) memcpy(W->array+(size_t)(W->nc+W->l+j)*W->ld,V->array+(size_t)(V->nc+V->l+j)*V->ld,(size_t)V->n*sizeof(double));
  return ORC_OK;
}
int orc_bv_copycolumn(orc_bv *V,int j,int i)
{
  if (j<0||j>=V->m||i<0||i>=V->m) return ORC_ERR_ARG;
  if (i!=j) memcpy(orc_bv_column(V,i),orc_bv_column(V,j),(size_t)V->n*sizeof(double));
  return ORC_OK;
}

/* Reproducible random column: value depends only on (seed, column, global row), mirroring
   -bv_reproducible_random (bvops.c:368-376: same vector irrespective of the number of processes).
   PETSc's PetscRandom (rander48) is external to /root/reference; this is a documented splitmix64
   stream, uniform in [0,1), shared verbatim with the GPU path. */
static inline uint64_t orc_splitmix64(uint64_t x) { x+=0x9E3779B97F4A7C15ULL; x=(x^(x>>30))*0xBF58476D1CE4E5B9ULL; x=(x^(x>>27))*0x94D049BB133111EBULL; return x^(x>>31); }
double orc_random_value(uint64_t seed,uint64_t col,uint64_t row) { uint64_t z=orc_splitmix64(seed ^ orc_splitmix64(col*0x100000001B3ULL + row + 0x12345678ULL*(col+1))); return (double)(z>>11)*(1.0/9007199254740992.0); }
int orc_bv_setrandomcolumn(orc_bv *bv,int j,uint64_t seed,int row0)
{
  int i; double *x; if (j<0||j>=bv->m) return ORC_ERR_ARG; x=orc_bv_column(bv,j);
  for (i=0;i<bv->n;i++) x[i]=orc_random_value(seed,(uint64_t)j,(uint64_t)(row0+i));
  return ORC_OK;
}

/* ------------------------------------------------------------------------------------------- */
/* CSR SpMV = PETSc MatMult_SeqAIJ (external; call sites bvops.c:879, stsolve.c:22)             */
void orc_csr_mult(int n,const int *rowptr,const int *col,const double *val,const double *x,double *y)
{
  int i;
#ifdef _OPENMP
  #pragma omp parallel for schedule(static)
#endif
  for (i=0;i<n;i++) { double s=0.0; int p; for (p=rowptr[i];p<rowptr[i+1];p++) s+=val[p]*x[col[p]]; y[i]=s; }
}

/* BVMatMultColumn (bvops.c:862-885): V(:,j+1) = A*V(:,j) */
int orc_bv_matmultcolumn(orc_bv *V,const orc_csr *A,int j)
{
  if (j<0 || j+1>=V->m || A->n!=V->n) return ORC_ERR_ARG;
  orc_csr_mult(A->n,A->rowptr,A->col,A->val,orc_bv_column(V,j),orc_bv_column(V,j+1));
  return ORC_OK;
}
/* BVMatMult, column-by-column variant (BVMatMult_Svec svec.c:213-222): W(:,l+j) = A V(:,l+j) */
int orc_bv_matmult(orc_bv *V,const orc_csr *A,orc_bv *W)
{
  int j; if (V->k-V->l!=W->k-W->l || A->n!=W->n) return ORC_ERR_ARG;
  for (j=0;j<V->k-V->l;j++) orc_csr_mult(A->n,A->rowptr,A->col,A->val,orc_bv_column(V,V->l+j),orc_bv_column(W,W->l+j));
  return ORC_OK;
}

/* ------------------------------------------------------------------------------------------- */
/* coefficient-buffer helpers (bvimpl.h). h==NULL -> buffer column j / scratch column 0          */

/* BV_SafeSqrt bvimpl.h:121-141 (definite inner product branch) */
static int orc_safe_sqrt(orc_bv *bv,double alpha,double *res)
{
  if (!(alpha>-bv->deftol)) return ORC_ERR_INNERPROD;
  *res = (alpha<0.0)? 0.0: sqrt(alpha);
  return ORC_OK;
}
static void orc_clean_coefficients(orc_bv *bv,int j,double *h)            /* bvimpl.h:289-302 */
{ double *hh = h ? h : bv->buffer+(size_t)j*(bv->nc+bv->m); int i; for (i=0;i<bv->nc+j;i++) hh[i]=0.0; }
static void orc_add_coefficients(orc_bv *bv,int j,double *h,double *c)    /* bvimpl.h:308-322 */
{ double *cc = h ? c : bv->buffer, *hh = h ? h : bv->buffer+(size_t)j*(bv->nc+bv->m); int i; for (i=0;i<bv->nc+j;i++) hh[i]+=cc[i]; }
static void orc_set_value(orc_bv *bv,int j,int k,double *h,double value)  /* bvimpl.h:328-341 */
{ double *hh = h ? h : bv->buffer+(size_t)k*(bv->nc+bv->m); hh[bv->nc+j]=value; }
static double orc_square_sum(orc_bv *bv,int j,double *h)                  /* bvimpl.h:347-360 */
{ double *hh = h ? h : bv->buffer, sum=0.0; int i; for (i=0;i<bv->nc+j;i++) sum+=hh[i]*hh[i]; return sum; }
static int orc_square_root(orc_bv *bv,int j,double *h,double *beta)       /* bvimpl.h:387-397 */
{ double *hh = h ? h : bv->buffer; return orc_safe_sqrt(bv,hh[bv->nc+j],beta); }

/* BVDotColumnInc (bvorthog.c:32-47): dotvec over columns 0..j inclusive, y = column j */
static int orc_dotcolumn_inc(orc_bv *X,int j,double *q)
{ int ksave=X->k,ierr; X->k=j+1; ierr=orc_bv_dotvec(X,orc_bv_column(X,j),q); X->k=ksave; return ierr; }

/* BV_NormVecOrColumn bvorthog.c:20-26 */
static int orc_safe_sqrt(orc_bv *bv,double alpha,double *res);
static int orc_norm_vec_or_column(orc_bv *bv,int j,double *v,double *nrm)
{
  if (bv->matrix) {                                      /* BVNorm_Private bvglobal.c:444-453: sqrt(z'*B*z) */
    const double *z = v ? v : bv->array+(size_t)(bv->nc+j)*bv->ld; const double *bz = orc_ipmatmult(bv,z); double p=0.0; int r;
    for (r=0;r<bv->n;r++) p+=bz[r]*z[r];
    return orc_safe_sqrt(bv,p,nrm);
  }
  if (v) { double scale=0.0,sumsq=1.0; orc_lassq(bv->n,v,&scale,&sumsq); *nrm=scale*sqrt(sumsq); return ORC_OK; }  /* VecNorm */
  return orc_bv_norm(bv,j,ORC_NORM_2,nrm);
}

/* BVOrthogonalizeMGS1 bvorthog.c:52-85 (no inner-product matrix, no signature) */
static int orc_mgs1(orc_bv *bv,int j,double *v,const int *which,double *h,double *c,double *onrm,double *nrm)
{
  int i,r; double *w = v ? v : orc_bv_column(bv,j);
  if (onrm) orc_norm_vec_or_column(bv,j,v,onrm);
  for (i=-bv->nc;i<j;i++) {
    const double *vi; double dot=0.0;
    if (which && i>=0 && !which[i]) continue;
    vi=orc_bv_column(bv,i);
    { const double *z = orc_ipmatmult(bv,w);               /* bvorthog.c:68-71: z = B*w */
      for (r=0;r<bv->n;r++) dot+=z[r]*vi[r]; }             /* VecDot(z,vi) */
    orc_set_value(bv,i,0,c,dot);                           /* BV_SetValue(bv,i,0,c,dot) */
    for (r=0;r<bv->n;r++) w[r]+=(-dot)*vi[r];              /* VecAXPY(w,-dot,vi) */
  }
  if (nrm) orc_norm_vec_or_column(bv,j,v,nrm);
  orc_add_coefficients(bv,j,h,c);
  return ORC_OK;
}

/* BVOrthogonalizeCGS1 bvorthog.c:91-132: one CGS pass with one global synchronization */
static int orc_cgs1(orc_bv *bv,int j,double *v,double *h,double *c,double *onorm,double *norm)
{
  double sum,beta=0.0; int ierr;
  bv->k=j;
  if (onorm || norm) {
    if (!v) { if ((ierr=orc_dotcolumn_inc(bv,j,c))) return ierr; if ((ierr=orc_square_root(bv,j,c,&beta))) return ierr; }
    else    { orc_bv_dotvec(bv,v,c); orc_norm_vec_or_column(bv,j,v,&beta); }
  } else {
    if (!v) orc_bv_dotcolumn(bv,j,c); else orc_bv_dotvec(bv,v,c);
  }
  if (!v) orc_bv_multcolumn(bv,-1.0,1.0,j,c); else orc_bv_multvec(bv,-1.0,1.0,v,c);
  if (onorm) *onorm=beta;
  if (norm) {
    sum=orc_square_sum(bv,j,c);
    *norm=beta*beta-sum;
    if (*norm<=0.0) { if ((ierr=orc_norm_vec_or_column(bv,j,v,norm))) return ierr; }
    else *norm=sqrt(*norm);
  }
  orc_add_coefficients(bv,j,h,c);
  return ORC_OK;
}

static int orc_gs1(orc_bv *bv,int mgs,int j,double *v,const int *which,double *h,double *c,double *onrm,double *nrm)
{ bv->passes_last++; bv->passes_total++; return mgs ? orc_mgs1(bv,j,v,which,h,c,onrm,nrm) : orc_cgs1(bv,j,v,h,c,onrm,nrm); }

/* BVOrthogonalizeGS bvorthog.c:145-217 */
static int orc_orthogonalize_gs(orc_bv *bv,int j,double *v,const int *which,double *norm,int *lindep)
{
  double *h,*c,onrm=0.0,nrm=0.0; int k,l,mgs,dolindep,ierr;
  if (v) { k=bv->k; h=bv->h; c=bv->c; } else { k=j; h=NULL; c=NULL; }
  mgs = (bv->orthog_type==ORC_MGS);
  dolindep = lindep?1:0;
  bv->passes_last=0;
  orc_clean_coefficients(bv,k,h);
  switch (bv->orthog_ref) {
  case ORC_REFINE_IFNEEDED:
    if ((ierr=orc_gs1(bv,mgs,k,v,which,h,c,&onrm,&nrm))) return ierr;
    l=1;
    while (l<3 && nrm && fabs(nrm)<bv->orthog_eta*fabs(onrm)) {
      l++;
      if (mgs) onrm=nrm;
      if ((ierr=orc_gs1(bv,mgs,k,v,which,h,c,mgs?NULL:&onrm,&nrm))) return ierr;
    }
    if (dolindep) *lindep = !(nrm && fabs(nrm)>=bv->orthog_eta*fabs(onrm));
    break;
  case ORC_REFINE_NEVER:
    if ((ierr=orc_gs1(bv,mgs,k,v,which,h,c,NULL,NULL))) return ierr;
    if (norm || dolindep) if ((ierr=orc_norm_vec_or_column(bv,k,v,&nrm))) return ierr;
    if (dolindep) *lindep = !nrm;
    break;
  case ORC_REFINE_ALWAYS:
    if ((ierr=orc_gs1(bv,mgs,k,v,which,h,c,NULL,NULL))) return ierr;
    if ((ierr=orc_gs1(bv,mgs,k,v,which,h,c,dolindep?&onrm:NULL,(norm||dolindep)?&nrm:NULL))) return ierr;
    if (dolindep) *lindep = !(nrm && fabs(nrm)>=bv->orthog_eta*fabs(onrm));
    break;
  default: return ORC_ERR_ARG;
  }
  if (norm) {
    *norm=nrm;
    if (!v) { if (dolindep && *lindep) orc_set_value(bv,k,k,h,0.0); else orc_set_value(bv,k,k,h,nrm); }
  }
  return ORC_OK;
}

/* BV_StoreCoefficients bvimpl.h:403-415 */
static void orc_store_coefficients(orc_bv *bv,int j,double *h,double *dest)
{ double *hh = h ? h : bv->buffer+(size_t)j*(bv->nc+bv->m); int i; for (i=bv->l;i<j;i++) dest[i-bv->l]=hh[bv->nc+i]; }

/* BVOrthogonalizeVec bvorthog.c:247-269 */
int orc_bv_orthogonalizevec(orc_bv *bv,double *v,double *H,double *norm,int *lindep)
{
  int ksave=bv->k,lsave=bv->l,ierr;
  bv->l=-bv->nc;
  ierr=orc_orthogonalize_gs(bv,0,v,NULL,norm,lindep);
  bv->k=ksave; bv->l=lsave;
  if (!ierr && H) orc_store_coefficients(bv,bv->k,bv->h,H);
  return ierr;
}

/* BVOrthogonalizeColumn bvorthog.c:315-339 */
int orc_bv_orthogonalizecolumn(orc_bv *bv,int j,double *H,double *norm,int *lindep)
{
  int ksave=bv->k,lsave=bv->l,ierr;
  if (j<0 || j>=bv->m) return ORC_ERR_ARG;
  bv->l=-bv->nc;
  ierr=orc_orthogonalize_gs(bv,j,NULL,NULL,norm,lindep);
  bv->k=ksave; bv->l=lsave;
  if (!ierr && H) orc_store_coefficients(bv,j,NULL,H);
  return ierr;
}

/* BVOrthogonalizeSomeColumn bvorthog.c:432-470 (MGS only) */
int orc_bv_orthogonalizesomecolumn(orc_bv *bv,int j,const int *which,double *H,double *norm,int *lindep)
{
  int ksave=bv->k,lsave=bv->l,ierr;
  if (j<0 || j>=bv->m || bv->orthog_type!=ORC_MGS) return ORC_ERR_ARG;
  bv->l=-bv->nc;
  ierr=orc_orthogonalize_gs(bv,j,NULL,which,norm,lindep);
  bv->k=ksave; bv->l=lsave;
  if (!ierr && H) orc_store_coefficients(bv,j,NULL,H);
  return ierr;
}

/* BVOrthonormalizeColumn bvorthog.c:380-427 (replace=FALSE path; the random replacement is the caller's job) */
int orc_bv_orthonormalizecolumn(orc_bv *bv,int j,double *norm,int *lindep)
{
  int ksave=bv->k,lsave=bv->l,ierr,lndep=0; double nrm=0.0;
  if (j<0 || j>=bv->m) return ORC_ERR_ARG;
  bv->l=-bv->nc;
  ierr=orc_orthogonalize_gs(bv,j,NULL,NULL,&nrm,&lndep);
  bv->k=ksave; bv->l=lsave;
  if (ierr) return ierr;
  if (nrm!=1.0 && nrm!=0.0) orc_scal(bv->n,orc_bv_column(bv,j),1.0/nrm);   /* ops->scale(j,1/nrm) bvorthog.c:417-422 */
  if (norm) *norm=nrm;
  if (lindep) *lindep=lndep;
  return ORC_OK;
}

/* BVInsertVecs bvfunc.c:331-375: copy W(:,i) into columns s.., orthonormalising and dropping dependent ones */
int orc_bv_insert_vecs(orc_bv *V,int s,int *m,const double *W,int ldw,int orth)
{
  int i,ndep=0,ierr;
  if (!*m) return ORC_OK;
  if (*m<0 || s<0 || s>=V->m || s+*m>V->m) return ORC_ERR_ARG;
  for (i=0;i<*m;i++) {
    double norm=0.0; int lindep=0;
    memcpy(orc_bv_column(V,s+i-ndep),W+(size_t)i*ldw,(size_t)V->n*sizeof(double));
    if (orth) {
      if ((ierr=orc_bv_orthogonalizecolumn(V,s+i-ndep,NULL,&norm,&lindep))) return ierr;
      if (norm==0.0 || lindep) ndep++;
      else orc_scal(V->n,orc_bv_column(V,s+i-ndep),1.0/norm);
    }
  }
  *m-=ndep;
  return ORC_OK;
}

/* BVInsertConstraints bvfunc.c:411-439: destructive; the storage grows to nc+m columns (BVResize without copy),
   the vectors are orthonormalised into the leading columns, which then become columns -nc..-1 */
int orc_bv_insert_constraints(orc_bv *V,int *nc,const double *C,int ldc)
{
  int msave=V->m,tot,ierr;
  if (!*nc) return ORC_OK;
  if (*nc<0 || V->nc) return ORC_ERR_ARG;
  tot=*nc+msave;
  free(V->array); free(V->buffer); free(V->h); free(V->c);
  V->array=(double*)calloc((size_t)tot*V->ld,sizeof(double));
  V->h=(double*)calloc(tot,sizeof(double)); V->c=(double*)calloc(tot,sizeof(double));
  V->buffer=(double*)calloc((size_t)tot*tot,sizeof(double));
  V->m=tot; V->l=0; V->k=tot;
  ierr=orc_bv_insert_vecs(V,0,nc,C,ldc,1);
  V->nc=*nc; V->m=msave; V->l=0; V->k=msave;
  free(V->buffer); V->buffer=(double*)calloc((size_t)(V->nc+V->m)*V->m,sizeof(double));   /* BVGetBufferVec bvbasic.c:775-791: (nc+m)*m */
  return ierr;
}

/* BVSetNumConstraints bvbasic.c:260-294 (only lessening is used: EPSSolve drops the deflation space, epssolve.c:201-205) */
int orc_bv_set_num_constraints(orc_bv *V,int nc)
{
  int total=V->nc+V->m,diff=nc-V->nc,i;
  if (nc<0 || total-nc<=0) return ORC_ERR_ARG;
  if (!diff) return ORC_OK;
  if (diff<0) for (i=0;i<V->m;i++) memcpy(orc_bv_column(V,i+diff),orc_bv_column(V,i),(size_t)V->n*sizeof(double));
  V->nc=nc; V->m=total-nc;
  if (V->l>V->m) V->l=V->m;
  if (V->k>V->m) V->k=V->m;
  free(V->buffer); V->buffer=(double*)calloc((size_t)total*V->m,sizeof(double));
  return ORC_OK;
}
int orc_bv_get_num_constraints(const orc_bv *V) { return V->nc; }

/* BV_OrthogonalizeColumn_Safe bvimpl.h:452-465 */
static int orc_orthogonalizecolumn_safe(orc_bv *bv,int j,double *norm,int *lindep)
{
  int ref=bv->orthog_ref,ierr;
  bv->orthog_ref=ORC_REFINE_NEVER;
  ierr=orc_bv_orthogonalizecolumn(bv,j,NULL,NULL,NULL);
  bv->orthog_ref=ref;
  if (norm) *norm=0.0;
  if (lindep) *lindep=1;
  return ierr;
}

/* ------------------------------------------------------------------------------------------- */
/* BVMatArnoldi bvkrylov.c:56-113.  H is ldh x >=m, column-major (MATSEQDENSE)                   */
int orc_bv_matarnoldi(orc_bv *V,const orc_csr *A,double *H,int ldh,int k,int *m,double *beta,int *breakdown)
{
  int j,lindep=0,ierr; const double *a=V->buffer; int nb=V->nc+V->m;
  if (k<0 || k>V->m || *m<=0 || *m>V->m || *m<=k) return ORC_ERR_ARG;
  for (j=k;j<*m;j++) {
    if ((ierr=orc_bv_matmultcolumn(V,A,j))) return ierr;
    if (j==V->N-1) ierr=orc_orthogonalizecolumn_safe(V,j+1,beta,&lindep);
    else ierr=orc_bv_orthonormalizecolumn(V,j+1,beta,&lindep);
    if (ierr) return ierr;
    if (lindep) { *m=j+1; break; }
  }
  if (breakdown) *breakdown=lindep;
  if (H) {
    for (j=k;j<*m-1;j++) memcpy(H+(size_t)j*ldh,a+V->nc+(size_t)(j+1)*nb,(size_t)(j+2)*sizeof(double));
    memcpy(H+(size_t)(*m-1)*ldh,a+V->nc+(size_t)(*m)*nb,(size_t)(*m)*sizeof(double));
    if (ldh>*m) H[(*m)+(size_t)(*m-1)*ldh]=a[V->nc+(*m)+(size_t)(*m)*nb];
  }
  return ORC_OK;
}

/* BVMatLanczos bvkrylov.c:165-226.  T stored as alpha[0..ldt) then beta[0..ldt) (DS_MAT_T layout) */
int orc_bv_matlanczos(orc_bv *V,const orc_csr *A,double *T,int ldt,int k,int *m,double *beta,int *breakdown)
{
  int j,lindep=0,ierr; const double *a=V->buffer; int nb=V->nc+V->m;
  if (k<0 || k>V->m || *m<=0 || *m>V->m || *m<=k) return ORC_ERR_ARG;
  for (j=k;j<*m;j++) {
    if ((ierr=orc_bv_matmultcolumn(V,A,j))) return ierr;
    if (j==V->N-1) ierr=orc_orthogonalizecolumn_safe(V,j+1,beta,&lindep);
    else ierr=orc_bv_orthonormalizecolumn(V,j+1,beta,&lindep);
    if (ierr) return ierr;
    if (lindep) { *m=j+1; break; }
  }
  if (breakdown) *breakdown=lindep;
  if (T) {
    double *alpha=T,*betat=T+ldt;
    for (j=k;j<*m;j++) { alpha[j]=a[V->nc+j+(size_t)(j+1)*nb]; betat[j]=a[V->nc+j+1+(size_t)(j+1)*nb]; }
  }
  return ORC_OK;
}

/* ------------------------------------------------------------------------------------------- */
/* helpers for tests / cpu baseline                                                             */
orc_csr *orc_csr_wrap(int n,int ncols,const int *rowptr,const int *col,const double *val)
{ orc_csr *A=(orc_csr*)calloc(1,sizeof(orc_csr)); A->n=n; A->ncols=ncols; A->nnz=rowptr[n]; A->rowptr=rowptr; A->col=col; A->val=val; return A; }
void orc_csr_free(orc_csr *A) { free(A); }
void orc_set_num_threads(int t)
{
#ifdef _OPENMP
  if (t > 0) omp_set_num_threads(t);
#else
  (void)t;
#endif
}
int orc_num_threads(void)
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* 3-D 7-point Laplacian, natural ordering x fastest, diag 6, off -1, Dirichlet (ex19.c:47-78).
   Rows z0*nx*ny .. (z0+nzl)*nx*ny of the global nx*ny*nz grid; global column indices. */
long orc_laplacian3d_nnz(int nx,int ny,int nz,int z0,int nzl)
{
  long nnz=0; int k; for (k=z0;k<z0+nzl;k++) { long plane=(long)nx*ny; long cnt=7*plane-2L*ny-2L*nx; if (k==0) cnt-=plane; if (k==nz-1) cnt-=plane; nnz+=cnt; } return nnz;
}
void orc_laplacian3d_fill(int nx,int ny,int nz,int z0,int nzl,int *rowptr,int *col,double *val)
{
  long r,nrows=(long)nx*ny*nzl; long p=0;
  /* row lengths, prefix sum, then the rows in parallel (each thread touches the pages of its own row block first) */
  for (r=0;r<nrows;r++) {
    long i=r%nx, j=(r/nx)%ny, k=z0+r/((long)nx*ny);
    rowptr[r]=(int)p;
    p += 1+(k>0)+(j>0)+(i>0)+(i<nx-1)+(j<ny-1)+(k<nz-1);
  }
  rowptr[nrows]=(int)p;
#ifdef _OPENMP
  #pragma omp parallel for schedule(static)
#endif
  for (r=0;r<nrows;r++) {
    long i=r%nx, j=(r/nx)%ny, k=z0+r/((long)nx*ny), g=((long)k*ny+j)*nx+i, q=rowptr[r];
    if (k>0)    { col[q]=(int)(g-(long)nx*ny); val[q++]=-1.0; }
    if (j>0)    { col[q]=(int)(g-nx); val[q++]=-1.0; }
    if (i>0)    { col[q]=(int)(g-1);  val[q++]=-1.0; }
    col[q]=(int)g; val[q++]=6.0;
    if (i<nx-1) { col[q]=(int)(g+1);  val[q++]=-1.0; }
    if (j<ny-1) { col[q]=(int)(g+nx); val[q++]=-1.0; }
    if (k<nz-1) { col[q]=(int)(g+(long)nx*ny); val[q++]=-1.0; }
  }
}
/* 2-D 5-point Laplacian (ex2.c:44-51): diag 4, off -1, row-major grid index II=i*n+j, insertion order
   (i-1), (i+1), (j-1), (j+1), diag -- PETSc sorts columns within a row, so sorted order is emitted. */
long orc_laplacian2d_nnz(int n,int m) { return 5L*n*m-2L*n-2L*m; }
void orc_laplacian2d_fill(int n,int m,int *rowptr,int *col,double *val)
{
  long p=0; int i,j; long r=0;
  for (i=0;i<m;i++) for (j=0;j<n;j++) {
    long II=(long)i*n+j;
    rowptr[r++]=(int)p;
    if (i>0)   { col[p]=(int)(II-n); val[p++]=-1.0; }
    if (j>0)   { col[p]=(int)(II-1); val[p++]=-1.0; }
    col[p]=(int)II; val[p++]=4.0;
    if (j<n-1) { col[p]=(int)(II+1); val[p++]=-1.0; }
    if (i<m-1) { col[p]=(int)(II+n); val[p++]=-1.0; }
  }
  rowptr[r]=(int)p;
}
