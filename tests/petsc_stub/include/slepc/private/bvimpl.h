/* Test infrastructure (tests/petsc_stub/README.md): what adapters/slepc/hipks.c needs of include/slepc/private/bvimpl.h and
   include/slepcbv.h - the slot table (bvimpl.h:25-61: the types the adapter's functions must have to be installed), the fields of
   struct _p_BV it reads (bvimpl.h:63-113; same names and types, the others left out) and the interface calls it makes. */
#ifndef PETSC_STUB_BVIMPL_H
#define PETSC_STUB_BVIMPL_H
#include <petsc_stub_core.h>

typedef struct _p_BV *BV;
typedef enum { BV_ORTHOG_CGS, BV_ORTHOG_MGS } BVOrthogType;                                        /* slepcbv.h */
typedef enum { BV_ORTHOG_REFINE_IFNEEDED, BV_ORTHOG_REFINE_NEVER, BV_ORTHOG_REFINE_ALWAYS } BVOrthogRefineType;
typedef enum { BV_ORTHOG_BLOCK_GS, BV_ORTHOG_BLOCK_CHOL, BV_ORTHOG_BLOCK_TSQR, BV_ORTHOG_BLOCK_TSQRCHOL, BV_ORTHOG_BLOCK_SVQB } BVOrthogBlockType;

struct _BVOps {                                                                                     /* bvimpl.h:25-61 */
  PetscErrorCode (*mult)(BV,PetscScalar,PetscScalar,BV,Mat);
  PetscErrorCode (*multvec)(BV,PetscScalar,PetscScalar,Vec,PetscScalar*);
  PetscErrorCode (*multinplace)(BV,Mat,PetscInt,PetscInt);
  PetscErrorCode (*multinplacetrans)(BV,Mat,PetscInt,PetscInt);
  PetscErrorCode (*dot)(BV,BV,Mat);
  PetscErrorCode (*dotvec)(BV,Vec,PetscScalar*);
  PetscErrorCode (*dotvec_local)(BV,Vec,PetscScalar*);
  PetscErrorCode (*dotvec_begin)(BV,Vec,PetscScalar*);
  PetscErrorCode (*dotvec_end)(BV,Vec,PetscScalar*);
  PetscErrorCode (*scale)(BV,PetscInt,PetscScalar);
  PetscErrorCode (*norm)(BV,PetscInt,NormType,PetscReal*);
  PetscErrorCode (*norm_local)(BV,PetscInt,NormType,PetscReal*);
  PetscErrorCode (*norm_begin)(BV,PetscInt,NormType,PetscReal*);
  PetscErrorCode (*norm_end)(BV,PetscInt,NormType,PetscReal*);
  PetscErrorCode (*normalize)(BV,PetscScalar*);
  PetscErrorCode (*matmult)(BV,Mat,BV);
  PetscErrorCode (*copy)(BV,BV);
  PetscErrorCode (*copycolumn)(BV,PetscInt,PetscInt);
  PetscErrorCode (*resize)(BV,PetscInt,PetscBool);
  PetscErrorCode (*getcolumn)(BV,PetscInt,Vec*);
  PetscErrorCode (*restorecolumn)(BV,PetscInt,Vec*);
  PetscErrorCode (*getarray)(BV,PetscScalar**);
  PetscErrorCode (*restorearray)(BV,PetscScalar**);
  PetscErrorCode (*getarrayread)(BV,const PetscScalar**);
  PetscErrorCode (*restorearrayread)(BV,const PetscScalar**);
  PetscErrorCode (*restoresplit)(BV,BV*,BV*);
  PetscErrorCode (*restoresplitrows)(BV,IS,IS,BV*,BV*);
  PetscErrorCode (*gramschmidt)(BV,PetscInt,Vec,PetscBool*,PetscScalar*,PetscScalar*,PetscReal*,PetscReal*);
  PetscErrorCode (*getmat)(BV,Mat*);
  PetscErrorCode (*restoremat)(BV,Mat*);
  PetscErrorCode (*duplicate)(BV,BV);
  PetscErrorCode (*create)(BV);
  PetscErrorCode (*setfromoptions)(BV,PetscOptionItems*);
  PetscErrorCode (*view)(BV,PetscViewer);
  PetscErrorCode (*destroy)(BV);
};

struct _p_BV {                                                                                      /* bvimpl.h:63-113, the members the adapter touches */
  struct _p_PetscObject hdr; struct _BVOps *ops;                                                    /* PETSCHEADER(struct _BVOps) */
  PetscLayout        map;
  VecType            vtype;
  PetscInt           n,N,m,l,k,nc,ld;
  BVOrthogType       orthog_type;
  BVOrthogRefineType orthog_ref;
  PetscReal          orthog_eta;
  BVOrthogBlockType  orthog_block;
  Mat                matrix;
  PetscBool          indef;
  Vec                buffer,Bx,cv[2];
  PetscInt           ci[2];
  Vec                omega;
  PetscInt           issplit;
  Mat                Acreate;
  PetscBool          hip;
  void               *data;
};

/* interface-layer calls the adapter makes */
PetscErrorCode BVCreateVecEmpty(BV,Vec*);                                                          /* bvbasic.c:1332 */
PetscErrorCode BVGetColumn(BV,PetscInt,Vec*);                                                      /* bvbasic.c:1113 */
PetscErrorCode BVRestoreColumn(BV,PetscInt,Vec*);                                                  /* bvbasic.c:1156 */
PetscErrorCode BVNormVec(BV,Vec,NormType,PetscReal*);                                              /* bvglobal.c:530 */
PetscErrorCode BVNormColumn(BV,PetscInt,NormType,PetscReal*);                                      /* bvglobal.c:662 */
PetscErrorCode BVDotVec(BV,Vec,PetscScalar[]);                                                     /* bvglobal.c:151 */
PetscErrorCode BVDotColumn(BV,PetscInt,PetscScalar*);                                              /* bvglobal.c:302 */
PetscErrorCode BVMultVec(BV,PetscScalar,PetscScalar,Vec,PetscScalar[]);                            /* bvops.c:110 */
PetscErrorCode BVMultColumn(BV,PetscScalar,PetscScalar,PetscInt,PetscScalar*);                     /* bvops.c:165 */
/* coefficient helpers (static inline / macros in bvimpl.h:147-157, 289-415, 618-631; used with these argument lists in bvorthog.c:101-128) */
PetscErrorCode BV_IPMatMult(BV,Vec);
PetscErrorCode BV_SetValue(BV,PetscInt,PetscInt,PetscScalar*,PetscScalar);
PetscErrorCode BV_AddCoefficients(BV,PetscInt,PetscScalar*,PetscScalar*);
PetscErrorCode BV_SquareRoot(BV,PetscInt,PetscScalar*,PetscReal*);
PetscErrorCode BV_SquareSum(BV,PetscInt,PetscScalar*,PetscReal*);
PetscErrorCode BV_ApplySignature(BV,PetscInt,PetscScalar*,PetscBool);
#endif
