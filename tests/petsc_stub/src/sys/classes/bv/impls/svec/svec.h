/* Test infrastructure (tests/petsc_stub/README.md): what the adapter uses of src/sys/classes/bv/impls/svec/svec.h - the two-member
   private struct (svec.h:13-16) and the five SLEPC_INTERN functions it installs unchanged (svec.h:54-63). */
#pragma once
typedef struct { Vec v; PetscBool mpi; } BV_SVEC;
SLEPC_INTERN PetscErrorCode BVMatMult_Svec_HIP(BV,Mat,BV);
SLEPC_INTERN PetscErrorCode BVGetColumn_Svec_HIP(BV,PetscInt,Vec*);
SLEPC_INTERN PetscErrorCode BVRestoreColumn_Svec_HIP(BV,PetscInt,Vec*);
SLEPC_INTERN PetscErrorCode BVGetMat_Svec_HIP(BV,Mat*);
SLEPC_INTERN PetscErrorCode BVRestoreMat_Svec_HIP(BV,Mat*);
