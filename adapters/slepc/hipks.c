/*
   hipks.c -- SLEPc-side binding of libksgpu (include/ksgpu.h): the BV type "hipks", the MATSHELL that puts ks_mat_mult
   behind MatMult, and the communicator provider on PETSc's MPI.

   Goes into a SLEPc source tree as src/sys/classes/bv/impls/hipks/hipks.c (SLEPc 3.22 / PETSc 3.22 configured with HIP);
   registration is one line in BVRegisterAll (src/sys/classes/bv/interface/bvregis.c:28-38):
       PetscCall(BVRegister(BVHIPKS,BVCreate_HIPKS));
   then -bv_type hipks -vec_type hip. Nothing else in SLEPc changes: EPS, DS and ST stay in charge and reach the kernels
   through the ops table below (struct _BVOps, include/slepc/private/bvimpl.h:25-61; the checklist is BVCreate_Svec,
   src/sys/classes/bv/impls/svec/svec.c:489-557, and its HIP twins in svec/svechip/svechip.hip.cpp).

   PETSc is not part of the image this repository is developed in, so this file has not been compiled there. What it calls is:
   every ks_* entry point is exercised from C and Python by tests/ (tests/c_abi/bv_test1_abi.c and gs_slot_abi.c drive the
   BV-level slots exactly as the functions below do); every PETSc / SLEPc call below is used with the argument lists the
   reference's own svec / svechip implementation uses.

   Storage. The library owns the (nc+m)*ld column-major device block (BVSVEC layout). ctx->svec.v is a HIP Vec created WITH that
   block (VecCreate{Seq,MPI}HIPWithArray), so BVGetArray(Read), BVGetColumn, BVGetMat and PETSc's host<->device coherence
   work as they do for BVSVEC on HIP vectors; every compute slot brackets its ks_* call with VecHIPGetArray/Restore on
   ctx->svec.v, which tells PETSc that the device copy is the valid one.
   Coefficient buffer. bv->buffer (BVGetBufferVec, bvbasic.c:775-791) is a VECSEQHIP; its device array is handed to the
   library with ks_bv_set_buffer, so BV_CleanCoefficients / BV_SetValue / BV_StoreCoefficients of the interface layer
   (their _HIP forms: bv->hip is set) and the library's kernels work on the same memory.
   Stream. The context is created on the legacy default stream, which PETSc's HIP back-end uses, so no extra ordering is
   needed between PETSc's kernels and the library's.
*/
#include <slepc/private/bvimpl.h>
#include <petscdevice_hip.h>
#include "../src/sys/classes/bv/impls/svec/svec.h"
#include <ksgpu.h>

#define BVHIPKS "hipks"

/* The slots that only move views around - getcolumn / restorecolumn, getmat / restoremat, matmult (a MatMult per column on Vec
   views) - need no kernel of ours: they are BVSVEC's own HIP functions, which svec.h declares SLEPC_INTERN for use inside
   libslepc and which look at nothing but the two members of BV_SVEC. The private data therefore BEGINS with a BV_SVEC, so that
   (BV_SVEC*)bv->data is valid, and BVCreate_HIPKS installs BVGetColumn_Svec_HIP, BVRestoreColumn_Svec_HIP, BVGetMat_Svec_HIP,
   BVRestoreMat_Svec_HIP and BVMatMult_Svec_HIP (svechip.hip.cpp:269,363,375,458,488) as they are. */
typedef struct {
  BV_SVEC     svec;         /* v: HIP Vec over the library's column block; mpi. MUST stay the first member */
  ks_ctx      kctx;
  ks_bv       kbv;
  PetscScalar *bufptr;      /* device array of bv->buffer currently adopted by the library (NULL: the library's own) */
  MPI_Comm    comm;
} BV_HIPKS;

#define KS(call) do { int rc_ = (call); PetscCheck(!rc_,PETSC_COMM_SELF,(PetscErrorCode)rc_,"libksgpu: %s",ks_last_error_message()); } while (0)

/* ---- communicator provider: the three operations of ks_comm_ops on PETSc's MPI ------------------------------------ */
/* GPU-aware MPI works on the device pointers directly; otherwise the (tiny) reduction buffers and the halo segments are
   staged through the host. All three order themselves after the work already enqueued on `stream`. */
static int HipksAllreduceSum(void *user,double *dev_buf,int count,void *stream)
{
  BV_HIPKS    *ctx = (BV_HIPKS*)user;
  PetscScalar tmp[128],*h = tmp;
  int         ierr = 0;

  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return KS_ERR_LIB;
  if (use_gpu_aware_mpi) return MPI_Allreduce(MPI_IN_PLACE,dev_buf,count,MPIU_SCALAR,MPIU_SUM,ctx->comm) == MPI_SUCCESS ? 0 : KS_ERR_LIB;
  if (count > 128 && !(h = (PetscScalar*)malloc(sizeof(PetscScalar)*(size_t)count))) return KS_ERR_MEM;
  if (hipMemcpy(h,dev_buf,sizeof(PetscScalar)*(size_t)count,hipMemcpyDeviceToHost) != hipSuccess) ierr = KS_ERR_LIB;
  if (!ierr && MPI_Allreduce(MPI_IN_PLACE,h,count,MPIU_SCALAR,MPIU_SUM,ctx->comm) != MPI_SUCCESS) ierr = KS_ERR_LIB;
  if (!ierr && hipMemcpy(dev_buf,h,sizeof(PetscScalar)*(size_t)count,hipMemcpyHostToDevice) != hipSuccess) ierr = KS_ERR_LIB;
  if (h != tmp) free(h);
  return ierr;
}

static int HipksAllgatherHost(void *user,const void *send,int bytes,void *recv)
{
  BV_HIPKS *ctx = (BV_HIPKS*)user;
  return MPI_Allgather((void*)send,bytes,MPI_BYTE,recv,bytes,MPI_BYTE,ctx->comm) == MPI_SUCCESS ? 0 : KS_ERR_LIB;
}

static int HipksExchange(void *user,int npeers,const int *peers,const void *dev_send,const int *send_off,const int *send_cnt,void *dev_recv,const int *recv_off,const int *recv_cnt,int elem_bytes,void *stream)
{
  BV_HIPKS    *ctx = (BV_HIPKS*)user;
  MPI_Request *req;
  char        *hs = NULL,*hr = NULL;
  const char  *sbase = (const char*)dev_send;
  char        *rbase = (char*)dev_recv;
  size_t      stot = 0,rtot = 0;
  int         i,ierr = 0;

  if (!npeers) return 0;
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return KS_ERR_LIB;
  if (!(req = (MPI_Request*)malloc(sizeof(MPI_Request)*2*(size_t)npeers))) return KS_ERR_MEM;
  if (!use_gpu_aware_mpi) {
    for (i=0;i<npeers;i++) { if ((size_t)(send_off[i]+send_cnt[i]) > stot) stot = (size_t)(send_off[i]+send_cnt[i]); if ((size_t)(recv_off[i]+recv_cnt[i]) > rtot) rtot = (size_t)(recv_off[i]+recv_cnt[i]); }
    hs = (char*)malloc(stot*(size_t)elem_bytes+1); hr = (char*)malloc(rtot*(size_t)elem_bytes+1);
    if (!hs || !hr) { free(hs); free(hr); free(req); return KS_ERR_MEM; }
    if (stot && hipMemcpy(hs,dev_send,stot*(size_t)elem_bytes,hipMemcpyDeviceToHost) != hipSuccess) ierr = KS_ERR_LIB;
    sbase = hs; rbase = hr;
  }
  for (i=0;i<npeers && !ierr;i++) {
    if (MPI_Irecv(rbase+(size_t)recv_off[i]*(size_t)elem_bytes,recv_cnt[i]*elem_bytes,MPI_BYTE,peers[i],7701,ctx->comm,&req[2*i]) != MPI_SUCCESS) ierr = KS_ERR_LIB;
    if (MPI_Isend((void*)(sbase+(size_t)send_off[i]*(size_t)elem_bytes),send_cnt[i]*elem_bytes,MPI_BYTE,peers[i],7701,ctx->comm,&req[2*i+1]) != MPI_SUCCESS) ierr = KS_ERR_LIB;
  }
  if (!ierr && MPI_Waitall(2*npeers,req,MPI_STATUSES_IGNORE) != MPI_SUCCESS) ierr = KS_ERR_LIB;
  if (!ierr && !use_gpu_aware_mpi && rtot && hipMemcpy(dev_recv,hr,rtot*(size_t)elem_bytes,hipMemcpyHostToDevice) != hipSuccess) ierr = KS_ERR_LIB;
  free(hs); free(hr); free(req);
  return ierr;
}

static const ks_comm_ops SlepcKsMpiOps = {HipksAllreduceSum,HipksAllgatherHost,HipksExchange};

/* ---- state mirror --------------------------------------------------------------------------------------------------- */
/* l, k, nc, m and the orthogonalization options live in struct _p_BV and change without any ops slot being called
   (BVSetActiveColumns bvbasic.c:421, BVSetNumConstraints :260, BVSetOrthogonalization); every slot starts by mirroring
   them. The buffer Vec is re-created lazily after BVResize (bvbasic.c:356,783): adopt its device array when it changes. */
static PetscErrorCode HipksSync(BV bv)
{
  BV_HIPKS         *ctx = (BV_HIPKS*)bv->data;
  PetscScalar      *d_buf = NULL,*p;
  PetscObjectState state;

  PetscFunctionBegin;
  /* the object state moves whenever the contents may have changed (BVRestoreColumn of a written Vec bvbasic.c:1176, BVScaleColumn
     bvops.c:356, ...) and stays put between the passes BVOrthogonalizeGS makes on one column (bvorthog.c:176-202): the token under
     which ks_bv_gramschmidt_pass chains a pass to the dot products its predecessor left (3 reads of the basis per CGS2 step, not 4) */
  PetscCall(PetscObjectStateGet((PetscObject)bv,&state));
  KS(ks_bv_set_state(ctx->kbv,(uint64_t)state));
  KS(ks_bv_set_layout(ctx->kbv,(int)bv->nc,(int)bv->m));
  KS(ks_bv_set_active_columns(ctx->kbv,(int)bv->l,(int)bv->k));
  KS(ks_bv_set_orthogonalization(ctx->kbv,(int)bv->orthog_type,(int)bv->orthog_ref,(double)bv->orthog_eta));
  KS(ks_bv_set_orthog_block(ctx->kbv,(int)bv->orthog_block));
  if (bv->buffer) {
    PetscCall(VecHIPGetArray(bv->buffer,&p));
    d_buf = p;                                                  /* the allocation stays where it is for the life of the Vec */
    PetscCall(VecHIPRestoreArray(bv->buffer,&p));
  }
  if (d_buf != ctx->bufptr) { KS(ks_bv_set_buffer(ctx->kbv,d_buf)); ctx->bufptr = d_buf; }
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* ---- slots (same order as struct _BVOps) ------------------------------------------------------------------------------ */
static PetscErrorCode BVMult_HIPKS(BV Y,PetscScalar alpha,PetscScalar beta,BV X,Mat Q)
{
  BV_HIPKS          *y = (BV_HIPKS*)Y->data,*x = (BV_HIPKS*)X->data;
  const PetscScalar *q = NULL,*d_px;
  PetscScalar       *d_py;
  PetscInt          ldq = 0;

  PetscFunctionBegin;
  if (!Y->n) PetscFunctionReturn(PETSC_SUCCESS);
  PetscCall(HipksSync(Y)); PetscCall(HipksSync(X));
  PetscCall(VecHIPGetArrayRead(x->svec.v,&d_px));
  PetscCall(VecHIPGetArray(y->svec.v,&d_py));
  if (Q) { PetscCall(MatDenseGetLDA(Q,&ldq)); PetscCall(MatDenseGetArrayRead(Q,&q)); }      /* host seqdense, replicated (bvops.c:33-36) */
  KS(ks_bv_mult(y->kbv,alpha,beta,x->kbv,q,(int)ldq));
  if (Q) PetscCall(MatDenseRestoreArrayRead(Q,&q));
  PetscCall(VecHIPRestoreArrayRead(x->svec.v,&d_px));
  PetscCall(VecHIPRestoreArray(y->svec.v,&d_py));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode BVMultVec_HIPKS(BV X,PetscScalar alpha,PetscScalar beta,Vec y,PetscScalar *q)
{
  BV_HIPKS          *x = (BV_HIPKS*)X->data;
  const PetscScalar *d_px;
  PetscScalar       *d_py,*d_q;

  PetscFunctionBegin;
  PetscCall(HipksSync(X));
  PetscCall(VecHIPGetArrayRead(x->svec.v,&d_px));
  if (beta==(PetscScalar)0.0) PetscCall(VecHIPGetArrayWrite(y,&d_py));
  else PetscCall(VecHIPGetArray(y,&d_py));
  if (!q) PetscCall(VecHIPGetArray(X->buffer,&d_q));            /* coefficients in the buffer's scratch column (svec.c:46) */
  KS(ks_bv_multvec(x->kbv,alpha,beta,d_py,q));
  if (!q) PetscCall(VecHIPRestoreArray(X->buffer,&d_q));
  PetscCall(VecHIPRestoreArrayRead(x->svec.v,&d_px));
  if (beta==(PetscScalar)0.0) PetscCall(VecHIPRestoreArrayWrite(y,&d_py));
  else PetscCall(VecHIPRestoreArray(y,&d_py));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode BVMultInPlace_HIPKS(BV V,Mat Q,PetscInt s,PetscInt e)
{
  BV_HIPKS          *ctx = (BV_HIPKS*)V->data;
  const PetscScalar *q;
  PetscScalar       *d_pv;
  PetscInt          ldq;

  PetscFunctionBegin;
  if (s>=e || !V->n) PetscFunctionReturn(PETSC_SUCCESS);
  PetscCall(HipksSync(V));
  PetscCall(MatDenseGetLDA(Q,&ldq));
  PetscCall(VecHIPGetArray(ctx->svec.v,&d_pv));
  PetscCall(MatDenseGetArrayRead(Q,&q));
  KS(ks_bv_multinplace(ctx->kbv,q,(int)ldq,(int)s,(int)e));
  PetscCall(MatDenseRestoreArrayRead(Q,&q));
  PetscCall(VecHIPRestoreArray(ctx->svec.v,&d_pv));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode BVMultInPlaceHermitianTranspose_HIPKS(BV V,Mat Q,PetscInt s,PetscInt e)
{
  BV_HIPKS          *ctx = (BV_HIPKS*)V->data;
  const PetscScalar *q;
  PetscScalar       *d_pv;
  PetscInt          ldq;

  PetscFunctionBegin;
  if (s>=e || !V->n) PetscFunctionReturn(PETSC_SUCCESS);
  PetscCall(HipksSync(V));
  PetscCall(MatDenseGetLDA(Q,&ldq));
  PetscCall(VecHIPGetArray(ctx->svec.v,&d_pv));
  PetscCall(MatDenseGetArrayRead(Q,&q));
  KS(ks_bv_multinplace_trans(ctx->kbv,q,(int)ldq,(int)s,(int)e));
  PetscCall(MatDenseRestoreArrayRead(Q,&q));
  PetscCall(VecHIPRestoreArray(ctx->svec.v,&d_pv));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode BVDot_HIPKS(BV X,BV Y,Mat M)
{
  BV_HIPKS          *x = (BV_HIPKS*)X->data,*y = (BV_HIPKS*)Y->data;
  const PetscScalar *d_px,*d_py;
  PetscScalar       *m;
  PetscInt          ldm;

  PetscFunctionBegin;
  PetscCall(HipksSync(X)); PetscCall(HipksSync(Y));
  PetscCall(MatDenseGetLDA(M,&ldm));
  PetscCall(VecHIPGetArrayRead(x->svec.v,&d_px));
  PetscCall(VecHIPGetArrayRead(y->svec.v,&d_py));
  PetscCall(MatDenseGetArray(M,&m));
  KS(ks_bv_dot(x->kbv,y->kbv,m,(int)ldm));                    /* includes the reduction over ranks (bvblas.c:218) */
  PetscCall(MatDenseRestoreArray(M,&m));
  PetscCall(VecHIPRestoreArrayRead(x->svec.v,&d_px));
  PetscCall(VecHIPRestoreArrayRead(y->svec.v,&d_py));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode HipksDotVec(BV X,Vec y,PetscScalar *q,PetscBool reduce)
{
  BV_HIPKS          *x = (BV_HIPKS*)X->data;
  const PetscScalar *d_px,*d_py;
  PetscScalar       *d_q;
  Vec               z = y;

  PetscFunctionBegin;
  PetscCall(HipksSync(X));
  if (PetscUnlikely(X->matrix)) {                               /* B-inner product: the slot applies B itself (svechip.hip.cpp:133-136) */
    PetscCall(BV_IPMatMult(X,y));
    z = X->Bx;
  }
  PetscCall(VecHIPGetArrayRead(x->svec.v,&d_px));
  PetscCall(VecHIPGetArrayRead(z,&d_py));
  if (!q) PetscCall(VecHIPGetArray(X->buffer,&d_q));             /* result to the buffer's scratch column (svec.c:123) */
  if (reduce) KS(ks_bv_dotvec(x->kbv,d_py,q));
  else KS(ks_bv_dotvec_local(x->kbv,d_py,q));
  if (!q) PetscCall(VecHIPRestoreArray(X->buffer,&d_q));
  PetscCall(VecHIPRestoreArrayRead(z,&d_py));
  PetscCall(VecHIPRestoreArrayRead(x->svec.v,&d_px));
  PetscFunctionReturn(PETSC_SUCCESS);
}
static PetscErrorCode BVDotVec_HIPKS(BV X,Vec y,PetscScalar *q) { return HipksDotVec(X,y,q,PETSC_TRUE); }
static PetscErrorCode BVDotVec_Local_HIPKS(BV X,Vec y,PetscScalar *m) { return HipksDotVec(X,y,m,PETSC_FALSE); }   /* feeds BVDotVecBegin/End's PetscSplitReduction */

static PetscErrorCode BVScale_HIPKS(BV bv,PetscInt j,PetscScalar alpha)
{
  BV_HIPKS    *ctx = (BV_HIPKS*)bv->data;
  PetscScalar *d_pv;

  PetscFunctionBegin;
  if (!bv->n) PetscFunctionReturn(PETSC_SUCCESS);
  PetscCall(HipksSync(bv));
  PetscCall(VecHIPGetArray(ctx->svec.v,&d_pv));
  if (PetscUnlikely(j<0)) KS(ks_bv_scale(ctx->kbv,alpha));
  else KS(ks_bv_scalecolumn(ctx->kbv,(int)j,alpha));
  PetscCall(VecHIPRestoreArray(ctx->svec.v,&d_pv));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode HipksNorm(BV bv,PetscInt j,NormType type,PetscReal *val,PetscBool reduce)
{
  BV_HIPKS          *ctx = (BV_HIPKS*)bv->data;
  const PetscScalar *d_pv;
  int               kt = type==NORM_1 ? KS_NORM_1 : (type==NORM_2 ? KS_NORM_2 : (type==NORM_FROBENIUS ? KS_NORM_FROBENIUS : KS_NORM_INFINITY));

  PetscFunctionBegin;
  PetscCall(HipksSync(bv));
  PetscCall(VecHIPGetArrayRead(ctx->svec.v,&d_pv));
  if (!reduce) KS(ks_bv_norm_local(ctx->kbv,(int)j,kt,val));
  else if (PetscUnlikely(j<0)) KS(ks_bv_norm(ctx->kbv,kt,val));
  else KS(ks_bv_normcolumn(ctx->kbv,(int)j,kt,val));
  PetscCall(VecHIPRestoreArrayRead(ctx->svec.v,&d_pv));
  PetscFunctionReturn(PETSC_SUCCESS);
}
static PetscErrorCode BVNorm_HIPKS(BV bv,PetscInt j,NormType type,PetscReal *val) { return HipksNorm(bv,j,type,val,PETSC_TRUE); }
static PetscErrorCode BVNorm_Local_HIPKS(BV bv,PetscInt j,NormType type,PetscReal *val) { return HipksNorm(bv,j,type,val,PETSC_FALSE); }

static PetscErrorCode BVNormalize_HIPKS(BV bv,PetscScalar *eigi)
{
  BV_HIPKS    *ctx = (BV_HIPKS*)bv->data;
  PetscScalar *d_pv;

  PetscFunctionBegin;
  PetscCall(HipksSync(bv));
  PetscCall(VecHIPGetArray(ctx->svec.v,&d_pv));
  KS(ks_bv_normalize(ctx->kbv,eigi? eigi+bv->l: NULL));        /* entry 0 belongs to column l, as BVNormalize_Svec passes it (svec.c:196) */
  PetscCall(VecHIPRestoreArray(ctx->svec.v,&d_pv));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode BVCopy_HIPKS(BV V,BV W)
{
  BV_HIPKS          *v = (BV_HIPKS*)V->data,*w = (BV_HIPKS*)W->data;
  const PetscScalar *d_pv;
  PetscScalar       *d_pw;

  PetscFunctionBegin;
  PetscCall(HipksSync(V)); PetscCall(HipksSync(W));
  PetscCall(VecHIPGetArrayRead(v->svec.v,&d_pv));
  PetscCall(VecHIPGetArray(w->svec.v,&d_pw));
  KS(ks_bv_copy(v->kbv,w->kbv));
  PetscCall(VecHIPRestoreArrayRead(v->svec.v,&d_pv));
  PetscCall(VecHIPRestoreArray(w->svec.v,&d_pw));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode BVCopyColumn_HIPKS(BV V,PetscInt j,PetscInt i)
{
  BV_HIPKS    *ctx = (BV_HIPKS*)V->data;
  PetscScalar *d_pv;

  PetscFunctionBegin;
  PetscCall(HipksSync(V));
  PetscCall(VecHIPGetArray(ctx->svec.v,&d_pv));
  KS(ks_bv_copycolumn(ctx->kbv,(int)j,(int)i));
  PetscCall(VecHIPRestoreArray(ctx->svec.v,&d_pv));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* (re)create the HIP Vec over the library's column block */
static PetscErrorCode HipksWrapStorage(BV bv,PetscInt m)
{
  BV_HIPKS    *ctx = (BV_HIPKS*)bv->data;
  PetscScalar *d_array;
  PetscInt    bs;
  char        str[50];

  PetscFunctionBegin;
  PetscCall(VecDestroy(&ctx->svec.v));
  KS(ks_bv_get_array(ctx->kbv,&d_array));
  PetscCall(PetscLayoutGetBlockSize(bv->map,&bs));
  if (ctx->svec.mpi) PetscCall(VecCreateMPIHIPWithArray(PetscObjectComm((PetscObject)bv),bs,m*bv->ld,PETSC_DECIDE,d_array,&ctx->svec.v));
  else PetscCall(VecCreateSeqHIPWithArray(PetscObjectComm((PetscObject)bv),bs,m*bv->ld,d_array,&ctx->svec.v));
  if (((PetscObject)bv)->name) {
    PetscCall(PetscSNPrintf(str,sizeof(str),"%s_0",((PetscObject)bv)->name));
    PetscCall(PetscObjectSetName((PetscObject)ctx->svec.v,str));
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* ops->resize(bv,m,copy): called by BVResize (bvbasic.c:340-360) BEFORE it updates bv->m; the nc constraint columns are
   part of the storage, so m here is the new total the interface asks for */
static PetscErrorCode BVResize_HIPKS(BV bv,PetscInt m,PetscBool copy)
{
  BV_HIPKS *ctx = (BV_HIPKS*)bv->data;

  PetscFunctionBegin;
  KS(ks_bv_set_layout(ctx->kbv,0,(int)(bv->nc+bv->m)));        /* address the block as plain columns while it is re-created */
  KS(ks_bv_set_buffer(ctx->kbv,NULL)); ctx->bufptr = NULL;     /* BVResize destroys bv->buffer right after this slot */
  KS(ks_bv_resize(ctx->kbv,(int)m,copy?1:0));
  PetscCall(HipksWrapStorage(bv,m));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* BVGetArray_Svec and its three companions are static in svec.c (:297-326): the same one-line forwards to the Vec */
static PetscErrorCode BVGetArray_HIPKS(BV bv,PetscScalar **a) { PetscFunctionBegin; PetscCall(VecGetArray(((BV_HIPKS*)bv->data)->svec.v,a)); PetscFunctionReturn(PETSC_SUCCESS); }
static PetscErrorCode BVRestoreArray_HIPKS(BV bv,PetscScalar **a) { PetscFunctionBegin; PetscCall(VecRestoreArray(((BV_HIPKS*)bv->data)->svec.v,a)); PetscFunctionReturn(PETSC_SUCCESS); }
static PetscErrorCode BVGetArrayRead_HIPKS(BV bv,const PetscScalar **a) { PetscFunctionBegin; PetscCall(VecGetArrayRead(((BV_HIPKS*)bv->data)->svec.v,a)); PetscFunctionReturn(PETSC_SUCCESS); }
static PetscErrorCode BVRestoreArrayRead_HIPKS(BV bv,const PetscScalar **a) { PetscFunctionBegin; PetscCall(VecRestoreArrayRead(((BV_HIPKS*)bv->data)->svec.v,a)); PetscFunctionReturn(PETSC_SUCCESS); }

/* ops->gramschmidt (bvimpl.h:53): ONE Gram-Schmidt pass; BVOrthogonalizeGS (bvorthog.c:145-217) keeps the refinement loop,
   lindep and the coefficient clean-up. onrm / nrm are NULL when the caller passes NULL (REFINE_NEVER, first call of
   REFINE_ALWAYS). Column form on the standard inner product with CGS: the fused kernels (one dot sweep, device-side
   bookkeeping, one update; coefficients added into column j of bv->buffer, which the library has adopted).
   Everything else - a Vec argument, MGS with `which`, a B-inner product, an indefinite one - is one pass written on this
   BV's primitive operations through the public interface, which handle B and the signature themselves. That pass IS the
   reference's BVOrthogonalizeMGS1 / BVOrthogonalizeCGS1 (bvorthog.c:52-85, 91-132) step for step: those two are `static` in
   bvorthog.c and the BVOrthogonalizeGS1 macro (:134) reaches them only when ops->gramschmidt is NULL, so a type that fills
   the slot has to bring the non-fused cases along; nothing here is new arithmetic. */
static PetscErrorCode HipksGramSchmidtGeneric(BV bv,PetscInt j,Vec v,PetscBool *which,PetscScalar *h,PetscScalar *c,PetscReal *onrm,PetscReal *nrm)
{
  PetscBool   mgs = (bv->orthog_type==BV_ORTHOG_MGS)? PETSC_TRUE: PETSC_FALSE;
  PetscReal   beta = 0.0,sum;
  PetscScalar dot;
  PetscInt    i,ksave = bv->k;
  Vec         w = v,vi;

  PetscFunctionBegin;
  if (mgs) {                                                    /* one modified Gram-Schmidt sweep */
    if (!v) PetscCall(BVGetColumn(bv,j,&w));
    if (onrm) PetscCall(BVNormVec(bv,w,NORM_2,onrm));
    for (i=-bv->nc;i<j;i++) {
      if (which && i>=0 && !which[i]) continue;
      PetscCall(BVGetColumn(bv,i,&vi));
      if (bv->matrix) { PetscCall(BV_IPMatMult(bv,w)); PetscCall(VecDot(bv->Bx,vi,&dot)); }
      else PetscCall(VecDot(w,vi,&dot));
      PetscCall(BV_SetValue(bv,i,0,c,dot));
      if (bv->indef) { const PetscScalar *omega; PetscCall(VecGetArrayRead(bv->omega,&omega)); dot /= PetscRealPart(omega[bv->nc+i]); PetscCall(VecRestoreArrayRead(bv->omega,&omega)); }
      PetscCall(VecAXPY(w,-dot,vi));
      PetscCall(BVRestoreColumn(bv,i,&vi));
    }
    if (nrm) PetscCall(BVNormVec(bv,w,NORM_2,nrm));
    if (!v) PetscCall(BVRestoreColumn(bv,j,&w));
    PetscCall(BV_AddCoefficients(bv,j,h,c));
    PetscFunctionReturn(PETSC_SUCCESS);
  }
  /* one classical Gram-Schmidt pass with a single reduction */
  bv->k = j;
  if (!v) {
    if (onrm || nrm) {                                          /* dots against columns [-nc,j) and the column itself */
      bv->k = j+1;
      PetscCall(BVGetColumn(bv,j,&w));
      PetscUseTypeMethod(bv,dotvec,w,c);
      PetscCall(BVRestoreColumn(bv,j,&w));
      bv->k = j;
      PetscCall(BV_SquareRoot(bv,j,c,&beta));
    } else PetscCall(BVDotColumn(bv,j,c));
  } else {
    PetscCall(BVDotVec(bv,v,c));
    if (onrm || nrm) PetscCall(BVNormVec(bv,v,NORM_2,&beta));
  }
  if (PetscUnlikely(bv->indef)) PetscCall(BV_ApplySignature(bv,j,c,PETSC_TRUE));
  if (!v) PetscCall(BVMultColumn(bv,-1.0,1.0,j,c));
  else PetscCall(BVMultVec(bv,-1.0,1.0,v,c));
  if (PetscUnlikely(bv->indef)) PetscCall(BV_ApplySignature(bv,j,c,PETSC_FALSE));
  if (onrm) *onrm = beta;
  if (nrm) {
    if (PetscUnlikely(bv->indef)) { if (v) PetscCall(BVNormVec(bv,v,NORM_2,nrm)); else PetscCall(BVNormColumn(bv,j,NORM_2,nrm)); }
    else {
      PetscCall(BV_SquareSum(bv,j,c,&sum));
      *nrm = beta*beta-sum;
      if (PetscUnlikely(*nrm <= 0.0)) { if (v) PetscCall(BVNormVec(bv,v,NORM_2,nrm)); else PetscCall(BVNormColumn(bv,j,NORM_2,nrm)); }
      else *nrm = PetscSqrtReal(*nrm);
    }
  }
  PetscCall(BV_AddCoefficients(bv,j,h,c));
  bv->k = ksave;
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode BVGramSchmidt_HIPKS(BV bv,PetscInt j,Vec v,PetscBool *which,PetscScalar *h,PetscScalar *c,PetscReal *onrm,PetscReal *nrm)
{
  BV_HIPKS    *ctx = (BV_HIPKS*)bv->data;
  PetscScalar *d_pv,*d_buf;

  PetscFunctionBegin;
  if (v || h || c || bv->matrix || bv->indef || bv->orthog_type!=BV_ORTHOG_CGS) {
    PetscCall(HipksGramSchmidtGeneric(bv,j,v,which,h,c,onrm,nrm));
    PetscFunctionReturn(PETSC_SUCCESS);
  }
  PetscCall(HipksSync(bv));                                     /* adopts bv->buffer (created by the caller, bvorthog.c:331) */
  PetscCall(VecHIPGetArray(ctx->svec.v,&d_pv));
  PetscCall(VecHIPGetArray(bv->buffer,&d_buf));
  KS(ks_bv_gramschmidt_pass(ctx->kbv,(int)j,NULL,NULL,NULL,NULL,onrm,nrm));
  PetscCall(VecHIPRestoreArray(bv->buffer,&d_buf));
  PetscCall(VecHIPRestoreArray(ctx->svec.v,&d_pv));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode BVDestroy_HIPKS(BV bv)
{
  BV_HIPKS *ctx = (BV_HIPKS*)bv->data;

  PetscFunctionBegin;
  PetscCall(VecDestroy(&ctx->svec.v));
  PetscCall(VecDestroy(&bv->cv[0]));
  PetscCall(VecDestroy(&bv->cv[1]));
  KS(ks_bv_destroy(ctx->kbv));
  KS(ks_ctx_destroy(ctx->kctx));
  PetscCall(PetscFree(bv->data));
  bv->hip = PETSC_FALSE;
  PetscFunctionReturn(PETSC_SUCCESS);
}

SLEPC_EXTERN PetscErrorCode BVCreate_HIPKS(BV bv)
{
  BV_HIPKS          *ctx;
  PetscInt          nloc,N,j,lda,rstart;
  PetscMPIInt       rank,size;
  PetscBool         iship,isdense;
  const PetscScalar *aa;
  PetscScalar       *vv;
  MatType           mtype;
  int               ld,device = 0;

  PetscFunctionBegin;
  PetscCall(PetscNew(&ctx));
  bv->data = (void*)ctx;
  PetscCall(PetscStrcmpAny(bv->vtype,&iship,VECSEQHIP,VECMPIHIP,""));
  PetscCheck(iship,PetscObjectComm((PetscObject)bv),PETSC_ERR_SUP,"BVHIPKS needs HIP vectors (-vec_type hip), not %s",bv->vtype);
  PetscCheck(!bv->issplit,PetscObjectComm((PetscObject)bv),PETSC_ERR_SUP,"BVHIPKS does not support BVGetSplit()");
#if defined(PETSC_USE_COMPLEX) || !defined(PETSC_USE_REAL_DOUBLE) || defined(PETSC_USE_64BIT_INDICES)
  SETERRQ(PetscObjectComm((PetscObject)bv),PETSC_ERR_SUP,"BVHIPKS is built for real double scalars and 32-bit indices");
#endif
  bv->hip = PETSC_TRUE;                                         /* the _HIP forms of the coefficient helpers (bvimpl.h:618-631) */
  PetscCall(PetscStrcmp(bv->vtype,VECMPIHIP,&ctx->svec.mpi));
  ctx->comm = PetscObjectComm((PetscObject)bv);

  PetscCall(PetscLayoutGetLocalSize(bv->map,&nloc));
  PetscCall(PetscLayoutGetSize(bv->map,&N));
  PetscCall(PetscLayoutGetRange(bv->map,&rstart,NULL));
  PetscCallHIP(hipGetDevice(&device));
  KS(ks_ctx_create(device,(void*)hipStreamLegacy,&ctx->kctx));  /* PETSc's HIP back-end works on the legacy default stream */
  PetscCallMPI(MPI_Comm_rank(ctx->comm,&rank));
  PetscCallMPI(MPI_Comm_size(ctx->comm,&size));
  if (size>1) KS(ks_comm_set_ops(ctx->kctx,(int)rank,(int)size,&SlepcKsMpiOps,ctx));
  KS(ks_bv_create(ctx->kctx,(int)nloc,(int)N,(int)bv->m,(int)bv->ld,&ctx->kbv));       /* bv->ld = 0: the library's default (256-byte columns) */
  KS(ks_bv_set_ownership_start(ctx->kbv,(int)rstart));
  KS(ks_bv_get_sizes(ctx->kbv,NULL,NULL,NULL,&ld));
  bv->ld = ld;
  PetscCall(HipksWrapStorage(bv,bv->m));

  if (PetscUnlikely(bv->Acreate)) {                             /* BVCreateFromMat: copy the dense matrix in (svec.c:463-474) */
    PetscCall(MatGetType(bv->Acreate,&mtype));
    PetscCall(PetscStrcmpAny(mtype,&isdense,MATSEQDENSE,MATMPIDENSE,""));
    PetscCheck(isdense,PetscObjectComm((PetscObject)bv->Acreate),PETSC_ERR_SUP,"BVHIPKS requires a dense matrix in BVCreateFromMat()");
    PetscCall(MatDenseGetArrayRead(bv->Acreate,&aa));
    PetscCall(MatDenseGetLDA(bv->Acreate,&lda));
    PetscCall(VecGetArray(ctx->svec.v,&vv));
    for (j=0;j<bv->m;j++) PetscCall(PetscArraycpy(vv+j*bv->ld,aa+j*lda,bv->n));
    PetscCall(VecRestoreArray(ctx->svec.v,&vv));
    PetscCall(MatDenseRestoreArrayRead(bv->Acreate,&aa));
    PetscCall(MatDestroy(&bv->Acreate));
  }

  PetscCall(BVCreateVecEmpty(bv,&bv->cv[0]));
  PetscCall(BVCreateVecEmpty(bv,&bv->cv[1]));

  bv->ops->mult             = BVMult_HIPKS;
  bv->ops->multvec          = BVMultVec_HIPKS;
  bv->ops->multinplace      = BVMultInPlace_HIPKS;
  bv->ops->multinplacetrans = BVMultInPlaceHermitianTranspose_HIPKS;
  bv->ops->dot              = BVDot_HIPKS;
  bv->ops->dotvec           = BVDotVec_HIPKS;
  bv->ops->dotvec_local     = BVDotVec_Local_HIPKS;
  bv->ops->scale            = BVScale_HIPKS;
  bv->ops->norm             = BVNorm_HIPKS;
  bv->ops->norm_local       = BVNorm_Local_HIPKS;
  bv->ops->normalize        = BVNormalize_HIPKS;
  bv->ops->matmult          = BVMatMult_Svec_HIP;       /* BVSVEC's own: a MatMult per column on Vec views (svechip.hip.cpp:269) */
  bv->ops->copy             = BVCopy_HIPKS;
  bv->ops->copycolumn       = BVCopyColumn_HIPKS;
  bv->ops->resize           = BVResize_HIPKS;
  bv->ops->getcolumn        = BVGetColumn_Svec_HIP;     /* BVSVEC's own (svechip.hip.cpp:363,375) */
  bv->ops->restorecolumn    = BVRestoreColumn_Svec_HIP;
  bv->ops->getarray         = BVGetArray_HIPKS;
  bv->ops->restorearray     = BVRestoreArray_HIPKS;
  bv->ops->getarrayread     = BVGetArrayRead_HIPKS;
  bv->ops->restorearrayread = BVRestoreArrayRead_HIPKS;
  bv->ops->getmat           = BVGetMat_Svec_HIP;        /* BVSVEC's own (svechip.hip.cpp:458,488) */
  bv->ops->restoremat       = BVRestoreMat_Svec_HIP;
  bv->ops->gramschmidt      = BVGramSchmidt_HIPKS;
  bv->ops->destroy          = BVDestroy_HIPKS;
  /* left NULL on purpose, as BVSVEC does: dotvec_begin/_end and norm_begin/_end (the interface's PetscSplitReduction path
     runs on dotvec_local / norm_local), duplicate (BVDuplicate re-runs this constructor), restoresplit / restoresplitrows
     (BVGetSplit is refused above), view (BVView_Default goes through getcolumn), setfromoptions */
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* ---- the MatMult slot: a MATSHELL whose MATOP_MULT is ks_mat_mult ------------------------------------------------------ */
typedef struct { ks_ctx kctx; ks_mat A; PetscBool ownctx; BV_HIPKS comm; } MatHIPKS;

static PetscErrorCode MatMult_HIPKS(Mat S,Vec x,Vec y)
{
  MatHIPKS          *c;
  const PetscScalar *d_px;
  PetscScalar       *d_py;

  PetscFunctionBegin;
  PetscCall(MatShellGetContext(S,&c));
  PetscCall(VecHIPGetArrayRead(x,&d_px));
  PetscCall(VecHIPGetArrayWrite(y,&d_py));
  KS(ks_mat_mult(c->A,d_px,d_py));                              /* halo exchange + SpMV, enqueued on the legacy default stream */
  PetscCall(VecHIPRestoreArrayRead(x,&d_px));
  PetscCall(VecHIPRestoreArrayWrite(y,&d_py));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* MATOP_MULT_TRANSPOSE (what EPS_BALANCE_TWOSIDE asks of the operator, epsdefault.c:409): the library builds the transpose once from the CSR arrays
   the matrix keeps (one rank; on more ranks it returns PETSC_ERR_SUP) */
static PetscErrorCode MatMultTranspose_HIPKS(Mat S,Vec x,Vec y)
{
  MatHIPKS          *c;
  const PetscScalar *d_px;
  PetscScalar       *d_py;

  PetscFunctionBegin;
  PetscCall(MatShellGetContext(S,&c));
  PetscCall(VecHIPGetArrayRead(x,&d_px));
  PetscCall(VecHIPGetArrayWrite(y,&d_py));
  KS(ks_mat_mult_transpose(c->A,d_px,d_py));
  PetscCall(VecHIPRestoreArrayRead(x,&d_px));
  PetscCall(VecHIPRestoreArrayWrite(y,&d_py));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode MatGetDiagonal_HIPKS(Mat S,Vec d)
{
  MatHIPKS    *c;
  PetscScalar *d_pd;

  PetscFunctionBegin;
  PetscCall(MatShellGetContext(S,&c));
  PetscCall(VecHIPGetArrayWrite(d,&d_pd));
  KS(ks_mat_get_diagonal(c->A,d_pd));
  PetscCall(VecHIPRestoreArrayWrite(d,&d_pd));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode MatDestroy_HIPKS(Mat S)
{
  MatHIPKS *c;

  PetscFunctionBegin;
  PetscCall(MatShellGetContext(S,&c));
  KS(ks_mat_destroy(c->A));
  if (c->ownctx) KS(ks_ctx_destroy(c->kctx));
  PetscCall(PetscFree(c));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* MatCreateHIPKSFromAIJ - wrap an assembled SeqAIJ / MPIAIJ matrix: its rows go to the library once (CSR with GLOBAL column
   indices, the layout ks_mat_create_csr takes), the result is a shell matrix with HIP vectors whose MatMult runs
   ks_mat_mult. EPSSetOperators(eps,S,NULL) then works as with any shell matrix (src/eps/tutorials/ex3.c). */
SLEPC_EXTERN PetscErrorCode MatCreateHIPKSFromAIJ(Mat A,Mat *S)
{
  MatHIPKS          *c;
  MPI_Comm          comm;
  PetscMPIInt       rank,size;
  PetscInt          rstart,rend,nloc,N,i,ncols,nnz = 0,*ia,*ja,pos = 0;
  const PetscInt    *cols;
  const PetscScalar *vals;
  PetscScalar       *aa;
  int               device = 0;

  PetscFunctionBegin;
  PetscCall(PetscObjectGetComm((PetscObject)A,&comm));
  PetscCall(MatGetOwnershipRange(A,&rstart,&rend));
  PetscCall(MatGetSize(A,&N,NULL));
  nloc = rend-rstart;
  for (i=rstart;i<rend;i++) { PetscCall(MatGetRow(A,i,&ncols,NULL,NULL)); nnz += ncols; PetscCall(MatRestoreRow(A,i,&ncols,NULL,NULL)); }
  PetscCall(PetscMalloc3(nloc+1,&ia,nnz,&ja,nnz,&aa));
  ia[0] = 0;
  for (i=rstart;i<rend;i++) {
    PetscCall(MatGetRow(A,i,&ncols,&cols,&vals));
    PetscCall(PetscArraycpy(ja+pos,cols,ncols));
    PetscCall(PetscArraycpy(aa+pos,vals,ncols));
    pos += ncols; ia[i-rstart+1] = pos;
    PetscCall(MatRestoreRow(A,i,&ncols,&cols,&vals));
  }
  PetscCall(PetscNew(&c));
  PetscCallHIP(hipGetDevice(&device));
  KS(ks_ctx_create(device,(void*)hipStreamLegacy,&c->kctx)); c->ownctx = PETSC_TRUE;
  PetscCallMPI(MPI_Comm_rank(comm,&rank));
  PetscCallMPI(MPI_Comm_size(comm,&size));
  c->comm.comm = comm;
  if (size>1) KS(ks_comm_set_ops(c->kctx,(int)rank,(int)size,&SlepcKsMpiOps,&c->comm));
  KS(ks_mat_create_csr_flags(c->kctx,(int)nloc,(int)rstart,(int)N,ia,ja,aa,KS_MAT_KEEP_CSR,&c->A));   /* the arrays stay with the matrix: MatMultTranspose, MatAXPY */
  PetscCall(PetscFree3(ia,ja,aa));
  PetscCall(MatCreateShell(comm,nloc,nloc,N,N,c,S));
  PetscCall(MatShellSetOperation(*S,MATOP_MULT,(void(*)(void))MatMult_HIPKS));
  PetscCall(MatShellSetOperation(*S,MATOP_MULT_TRANSPOSE,(void(*)(void))MatMultTranspose_HIPKS));
  PetscCall(MatShellSetOperation(*S,MATOP_GET_DIAGONAL,(void(*)(void))MatGetDiagonal_HIPKS));
  PetscCall(MatShellSetOperation(*S,MATOP_DESTROY,(void(*)(void))MatDestroy_HIPKS));
  PetscCall(MatShellSetVecType(*S,VECHIP));                     /* the BV inherits the HIP vector type from the operator (stsolve.c:349-353) */
  PetscFunctionReturn(PETSC_SUCCESS);
}
