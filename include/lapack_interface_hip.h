/* lapack_interface_hip.h - the dense linear algebra helpers of SCIP-SDP (src/sdpi/lapack_interface.h:50-127), served by
 * the HIP kernels of libhipsdp.so instead of Fortran LAPACK/BLAS (DSYEVR, DSYEVX, DGEMV, DGEMM, DGELSD;
 * src/sdpi/lapack_interface.c:95-125).  Same names, argument meaning (column-major storage, eigenvectors as rows,
 * 1-based eigenvalue index, input matrix may be destroyed) and return codes.  Callers: cons_sdp.c, relax_sdp.c, sdpi.c,
 * sdpsolchecker.c, solveonevarsdp.c.  "ref:" = line in lapack_interface.h. */
#ifndef LAPACK_INTERFACE_HIP_H
#define LAPACK_INTERFACE_HIP_H

#include "hipsdp_scip_compat.h"

#ifdef __cplusplus
extern "C" {
#endif

SCIP_EXPORT SCIP_RETCODE SCIPlapackComputeIthEigenvalue(BMS_BUFMEM* bufmem, SCIP_Bool geteigenvectors, int n, SCIP_Real* A,
   int i, SCIP_Real* eigenvalue, SCIP_Real* eigenvector);                                             /* ref: 50 */
SCIP_EXPORT SCIP_RETCODE SCIPlapackComputeIthEigenvalueAlternative(BMS_BUFMEM* bufmem, SCIP_Bool geteigenvectors, int n,
   SCIP_Real* A, int i, SCIP_Real* eigenvalue, SCIP_Real* eigenvector);                               /* ref: 62 */
SCIP_EXPORT SCIP_RETCODE SCIPlapackComputeEigenvectorsNegative(BMS_BUFMEM* bufmem, int n, SCIP_Real* A, SCIP_Real tol,
   int* neigenvalues, SCIP_Real* eigenvalues, SCIP_Real* eigenvectors);                               /* ref: 74 */
SCIP_EXPORT SCIP_RETCODE SCIPlapackComputeEigenvectorDecomposition(BMS_BUFMEM* bufmem, int n, SCIP_Real* A,
   SCIP_Real* eigenvalues, SCIP_Real* eigenvectors);                                                  /* ref: 86 */
SCIP_EXPORT SCIP_RETCODE SCIPlapackMatrixVectorMult(int nrows, int ncols, SCIP_Real* matrix, SCIP_Real* vector,
   SCIP_Real* result);                                                                                /* ref: 96 */
SCIP_EXPORT SCIP_RETCODE SCIPlapackMatrixMatrixMult(int nrowsA, int ncolsA, SCIP_Real* matrixA, SCIP_Bool transposeA,
   int nrowsB, int ncolsB, SCIP_Real* matrixB, SCIP_Bool transposeB, SCIP_Real* result);              /* ref: 106 */
SCIP_EXPORT SCIP_RETCODE SCIPlapackLinearSolve(BMS_BUFMEM* bufmem, int m, int n, SCIP_Real* A, SCIP_Real* b,
   SCIP_Real* x);                                                                                     /* ref: 120 */

#ifdef __cplusplus
}
#endif

#endif
