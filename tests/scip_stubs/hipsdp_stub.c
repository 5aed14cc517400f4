/* hipsdp_stub.c - TEST-ONLY stand-in for libhipsdp.so in tests/test_lapack_host_branch_cpu.py: the integrated build of
 * lapack_interface_hip.c (-DHIPSDP_WITH_SCIP) keeps the host LAPACK / BLAS routine below its size cutoffs; this stub lets that
 * branch be linked and run on a machine without a device.  Every device entry point counts its calls and fails: a test that
 * expects the host branch sees at once when the device branch was taken instead. */
#include "hipsdp.h"

static int stub_calls = 0;
int hipsdp_stub_calls(void) { return stub_calls; }

int hipsdp_syev(int device, int n, const double* A, double* lam, double* V)
{ (void) device; (void) n; (void) A; (void) lam; (void) V; ++stub_calls; return HIPSDP_ERR_NODEVICE; }
int hipsdp_syevi_small(int device, int n, const double* A, int i, double* eigval, double* eigvec)
{ (void) device; (void) n; (void) A; (void) i; (void) eigval; (void) eigvec; ++stub_calls; return HIPSDP_ERR_NODEVICE; }
int hipsdp_gemv_n(int device, int R, long long E, const double* A, int nv, const double* V, double* out)
{ (void) device; (void) R; (void) E; (void) A; (void) nv; (void) V; (void) out; ++stub_calls; return HIPSDP_ERR_NODEVICE; }
int hipsdp_gemv_t(int device, int R, long long E, const double* A, const double* coef, double* out)
{ (void) device; (void) R; (void) E; (void) A; (void) coef; (void) out; ++stub_calls; return HIPSDP_ERR_NODEVICE; }
int hipsdp_dgemm(int device, int layA, int layB, int M, int N, int K, double alpha, const double* A, long long lda,
   const double* B, long long ldb, double beta, double* C, long long ldc, int lower_only, int splitk)
{ (void) device; (void) layA; (void) layB; (void) M; (void) N; (void) K; (void) alpha; (void) A; (void) lda; (void) B; (void) ldb; (void) beta;
  (void) C; (void) ldc; (void) lower_only; (void) splitk; ++stub_calls; return HIPSDP_ERR_NODEVICE; }
