/* Test infrastructure (tests/petsc_stub/README.md): device-array access of HIP vectors as the reference's svechip back-end uses it. */
#ifndef PETSC_STUB_DEVICE_HIP_H
#define PETSC_STUB_DEVICE_HIP_H
#include <petsc_stub_core.h>
PetscErrorCode VecHIPGetArray(Vec,PetscScalar**);                 /* svechip.hip.cpp:29 */
PetscErrorCode VecHIPRestoreArray(Vec,PetscScalar**);             /* svechip.hip.cpp:41 */
PetscErrorCode VecHIPGetArrayRead(Vec,const PetscScalar**);       /* svechip.hip.cpp:27 */
PetscErrorCode VecHIPRestoreArrayRead(Vec,const PetscScalar**);   /* svechip.hip.cpp:39 */
PetscErrorCode VecHIPGetArrayWrite(Vec,PetscScalar**);            /* svechip.hip.cpp:28 */
PetscErrorCode VecHIPRestoreArrayWrite(Vec,PetscScalar**);        /* svechip.hip.cpp:40 */
#endif
