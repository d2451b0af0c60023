/* Test infrastructure (tests/petsc_stub/README.md): the PETSc names adapters/slepc/hipks.c uses, as prototypes.
   Citations are SLEPc 3.22.2 paths under /root/reference: a call site with this argument list, or the declaring SLEPc header. */
#ifndef PETSC_STUB_CORE_H
#define PETSC_STUB_CORE_H
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <hip/hip_runtime_api.h>          /* the real HIP runtime API (C-clean): hipMemcpy, hipStreamSynchronize, hipStreamLegacy ... */

/* ---- MPI (standard C binding; MPICH-style handles) ---- */
typedef int MPI_Comm; typedef int MPI_Datatype; typedef int MPI_Op; typedef int MPI_Request; typedef struct { int s; } MPI_Status;
#define MPI_SUCCESS 0
#define MPI_BYTE ((MPI_Datatype)1)
#define MPI_DOUBLE ((MPI_Datatype)2)
#define MPI_SUM ((MPI_Op)1)
#define MPI_IN_PLACE ((void*)-1)
#define MPI_STATUSES_IGNORE ((MPI_Status*)1)
int MPI_Allreduce(const void*,void*,int,MPI_Datatype,MPI_Op,MPI_Comm);
int MPI_Allgather(const void*,int,MPI_Datatype,void*,int,MPI_Datatype,MPI_Comm);
int MPI_Irecv(void*,int,MPI_Datatype,int,int,MPI_Comm,MPI_Request*);
int MPI_Isend(const void*,int,MPI_Datatype,int,int,MPI_Comm,MPI_Request*);
int MPI_Waitall(int,MPI_Request*,MPI_Status*);
int MPI_Comm_rank(MPI_Comm,int*);
int MPI_Comm_size(MPI_Comm,int*);

/* ---- scalar types of the build the adapter supports: real double, 32-bit indices (hipks.c checks the PETSC_USE_* macros) ---- */
#define PETSC_USE_REAL_DOUBLE 1
typedef int PetscErrorCode;
typedef int PetscInt;
typedef int PetscMPIInt;
typedef double PetscScalar;
typedef double PetscReal;
typedef int64_t PetscObjectState;          /* bvimpl.h:87 PetscObjectState xstate */
typedef enum { PETSC_FALSE, PETSC_TRUE } PetscBool;
typedef enum { NORM_1 = 0, NORM_2 = 1, NORM_FROBENIUS = 2, NORM_INFINITY = 3 } NormType;     /* bvglobal.c:496, svec.c:164 */
typedef const char *VecType; typedef const char *MatType;
typedef enum { MATOP_MULT = 3, MATOP_MULT_TRANSPOSE = 5, MATOP_GET_DIAGONAL = 17, MATOP_DESTROY = 250 } MatOperation;   /* ex3.c:47, ex9.c:123 */
#define PETSC_SUCCESS 0
#define PETSC_ERR_SUP 56
#define PETSC_DECIDE (-1)
#define PETSC_COMM_SELF ((MPI_Comm)0)
#define MPIU_SCALAR MPI_DOUBLE
#define MPIU_SUM MPI_SUM
#define VECSEQHIP "seqhip"
#define VECMPIHIP "mpihip"
#define VECHIP "hip"
#define MATSEQDENSE "seqdense"
#define MATMPIDENSE "mpidense"
#define SLEPC_EXTERN extern
#define SLEPC_INTERN extern

typedef struct _p_PetscObject { MPI_Comm comm; char *name; } *PetscObject;      /* ((PetscObject)bv)->name svechip.hip.cpp:345 */
typedef struct _p_Vec *Vec;
typedef struct _p_Mat *Mat;
typedef struct _p_IS *IS;
typedef struct _p_PetscLayout *PetscLayout;
typedef struct _p_PetscViewer *PetscViewer;
typedef struct _p_PetscRandom *PetscRandom;
typedef struct _PetscOptionItems PetscOptionItems;

/* ---- error-handling macros (svechip.hip.cpp:26-45 shows the idiom) ---- */
#define PetscFunctionBegin do { } while (0)
#define PetscFunctionReturn(x) return (x)
#define PetscCall(...) do { PetscErrorCode ierr_ = (__VA_ARGS__); if (ierr_) return ierr_; } while (0)
#define PetscCallHIP(...) do { if ((__VA_ARGS__) != hipSuccess) return 97; } while (0)           /* svechip.hip.cpp:58 */
#define PetscCallMPI(...) do { if ((__VA_ARGS__) != MPI_SUCCESS) return 98; } while (0)          /* bvfunc.c:39 */
PetscErrorCode PetscStubError(MPI_Comm,PetscErrorCode,const char*,...);                             /* stands for PetscError behind PetscCheck / SETERRQ */
#define PetscCheck(cond,comm,ierr,...) do { if (!(cond)) return PetscStubError(comm,ierr,__VA_ARGS__); } while (0)   /* bvops.c:66 */
#define SETERRQ(comm,ierr,...) return PetscStubError(comm,ierr,__VA_ARGS__)
#define PetscUnlikely(c) (c)
#define PetscUseTypeMethod(obj,m,...) PetscCall((*(obj)->ops->m)(obj,__VA_ARGS__))               /* bvglobal.c:42 */
#define PetscSqrtReal(x) sqrt(x)
#define PetscRealPart(x) (x)
#define PetscNew(p) ((*(p) = calloc(1,sizeof(**(p)))) ? 0 : 55)                                  /* contig.c:344 */
#define PetscFree(p) (free(p),(p)=NULL,0)                                                         /* bvfunc.c:125 */
#define PetscMalloc3(n1,p1,n2,p2,n3,p3) ((*(p1)=malloc(sizeof(**(p1))*(size_t)(n1)),*(p2)=malloc(sizeof(**(p2))*(size_t)(n2)),*(p3)=malloc(sizeof(**(p3))*(size_t)(n3))),0)   /* bvcontour.c:202 */
#define PetscFree3(p1,p2,p3) (free(p1),free(p2),free(p3),0)                                       /* bvcontour.c:234 */
#define PetscArraycpy(d,s,n) (memcpy((d),(s),sizeof(*(d))*(size_t)(n)),0)                        /* bvblas.c:98 */
#include <stdlib.h>

extern PetscBool use_gpu_aware_mpi;                                                               /* bvhip.hip.cpp:230 */
MPI_Comm PetscObjectComm(PetscObject);                                                            /* bvfunc.c:121 */
PetscErrorCode PetscObjectGetComm(PetscObject,MPI_Comm*);                                         /* bvglobal.c:209 */
PetscErrorCode PetscObjectStateGet(PetscObject,PetscObjectState*);                                /* bvbasic.c:1920 */
PetscErrorCode PetscObjectSetName(PetscObject,const char[]);                                      /* bvbasic.c:1568 */
PetscErrorCode PetscStrcmp(const char[],const char[],PetscBool*);                                 /* bvbasic.c:46 */
PetscErrorCode PetscStrcmpAny(const char[],PetscBool*,const char[],...);                          /* bvbasic.c:1377 */
PetscErrorCode PetscSNPrintf(char*,size_t,const char[],...);                                      /* contig.c:236 */
PetscErrorCode PetscLayoutGetSize(PetscLayout,PetscInt*);                                         /* bvbasic.c:133 */
PetscErrorCode PetscLayoutGetLocalSize(PetscLayout,PetscInt*);                                    /* bvbasic.c:134 */
PetscErrorCode PetscLayoutGetRange(PetscLayout,PetscInt*,PetscInt*);                              /* bvbasic.c:2042 */
PetscErrorCode PetscLayoutGetBlockSize(PetscLayout,PetscInt*);                                    /* bvbasic.c:1384 */

/* ---- Vec ---- */
PetscErrorCode VecDestroy(Vec*);                                                                  /* bvfunc.c:127 */
PetscErrorCode VecGetArray(Vec,PetscScalar**);                                                    /* bvblas.c:138 */
PetscErrorCode VecRestoreArray(Vec,PetscScalar**);                                                /* bvblas.c:140 */
PetscErrorCode VecGetArrayRead(Vec,const PetscScalar**);                                          /* bvblas.c:132 */
PetscErrorCode VecRestoreArrayRead(Vec,const PetscScalar**);                                      /* bvblas.c:134 */
PetscErrorCode VecDot(Vec,Vec,PetscScalar*);                                                      /* bvglobal.c:67 */
PetscErrorCode VecAXPY(Vec,PetscScalar,Vec);                                                      /* bvbiorthog.c:34 */
PetscErrorCode VecCreateSeqHIPWithArray(MPI_Comm,PetscInt,PetscInt,const PetscScalar[],Vec*);     /* bvbasic.c:1393 */
PetscErrorCode VecCreateMPIHIPWithArray(MPI_Comm,PetscInt,PetscInt,PetscInt,const PetscScalar[],Vec*);   /* bvbasic.c:1392 */

/* ---- Mat ---- */
PetscErrorCode MatMult(Mat,Vec,Vec);                                                              /* bvglobal.c:852 */
PetscErrorCode MatDestroy(Mat*);                                                                  /* bvfunc.c:126 */
PetscErrorCode MatGetType(Mat,MatType*);                                                          /* contig.c:383 */
PetscErrorCode MatGetSize(Mat,PetscInt*,PetscInt*);                                               /* bvfunc.c:255 */
PetscErrorCode MatGetOwnershipRange(Mat,PetscInt*,PetscInt*);                                     /* bv/tests/test14.c:38 */
PetscErrorCode MatGetRow(Mat,PetscInt,PetscInt*,const PetscInt*[],const PetscScalar*[]);          /* PETSc petscmat.h (no call site in the reference) */
PetscErrorCode MatRestoreRow(Mat,PetscInt,PetscInt*,const PetscInt*[],const PetscScalar*[]);
PetscErrorCode MatDenseGetLDA(Mat,PetscInt*);                                                     /* bvglobal.c:1119 */
PetscErrorCode MatDenseGetArray(Mat,PetscScalar**);                                               /* bvglobal.c:32 */
PetscErrorCode MatDenseRestoreArray(Mat,PetscScalar**);                                           /* bvglobal.c:49 */
PetscErrorCode MatDenseGetArrayRead(Mat,const PetscScalar**);                                     /* bvglobal.c:923 */
PetscErrorCode MatDenseRestoreArrayRead(Mat,const PetscScalar**);                                 /* bvglobal.c:925 */
PetscErrorCode MatCreateShell(MPI_Comm,PetscInt,PetscInt,PetscInt,PetscInt,void*,Mat*);           /* ex3.c:46 */
PetscErrorCode MatShellSetOperation(Mat,MatOperation,void(*)(void));                              /* ex3.c:47 */
PetscErrorCode MatShellGetContext(Mat,void*);                                                     /* ex3.c:148 */
PetscErrorCode MatShellSetVecType(Mat,VecType);                                                   /* stsolve.c:352 */
#endif
