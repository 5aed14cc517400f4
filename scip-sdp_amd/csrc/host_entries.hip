/* host_entries.hip - the host-buffer entry points the SCIPlapack* surface is built on (src/sdpi/lapack_interface_hip.c; reference
 * src/sdpi/lapack_interface.c:398-706: DSYEVR / DGEMV / DGEMM on host arrays, called from cons_sdp.c and relax_sdp.c between node
 * solves) and that the parity tests call directly: hipsdp_dgemm, hipsdp_gemv_n / _t, hipsdp_syev.
 *
 * Product path, not scaffolding: every calling host thread owns a context per device - a non-blocking stream, a pinned,
 * device-mapped staging buffer and a pool of device memory, both grow-only - so that a call in steady state does NO hipMalloc /
 * hipFree (both synchronise the device), no pageable copy, nothing on the null stream and no device-wide synchronisation: operands
 * go through the pinned buffer (small ones are read by the kernels straight from it, larger ones are copied by the copy engine
 * on the context's stream), the call waits for its own stream only.  Solver instances working on other host threads are never
 * stalled by a SCIPlapack* call. */
#include "hs_kernels.h"
#include "../../include/hipsdp.h"
#include <cstring>
#include <cstdlib>

namespace {

struct he_ctx
{
   int device;
   hipStream_t stream;
   double* hpin; double* hdev; long long hcap;       /* pinned host buffer, its device address, doubles */
   double* dpool; long long dcap;                    /* device pool, doubles */
   he_ctx() : device(-1), stream(NULL), hpin(NULL), hdev(NULL), hcap(0), dpool(NULL), dcap(0) {}
   void release()
   {
      if ( device >= 0 )
         (void) hipSetDevice(device);
      if ( stream != NULL )
      {
         (void) hipStreamSynchronize(stream);
         (void) hipStreamDestroy(stream);
      }
      if ( hpin != NULL ) (void) hipHostFree(hpin);
      if ( dpool != NULL ) (void) hipFree(dpool);
      device = -1; stream = NULL; hpin = hdev = dpool = NULL; hcap = dcap = 0;
   }
   ~he_ctx() { release(); }
};

thread_local he_ctx g_he;

/* operands up to this many doubles in total are read by the kernels directly from the mapped pinned buffer */
const long long HE_DIRECT = 32768;

int he_context(int device, long long host_doubles, long long dev_doubles, he_ctx** out)
{
   int nd = 0;
   if ( hipGetDeviceCount(&nd) != hipSuccess || nd <= 0 )
      return HIPSDP_ERR_NODEVICE;
   if ( device < 0 || device >= nd )
      return HIPSDP_ERR_ARG;
   HS_HIP( hipSetDevice(device) );
   he_ctx& c = g_he;
   if ( c.device != device || c.stream == NULL )
   {
      c.release();
      hipStream_t st = NULL;
      HS_HIP( hipStreamCreateWithFlags(&st, hipStreamNonBlocking) );
      c.device = device;
      c.stream = st;
   }
   if ( host_doubles > c.hcap )
   {
      /* growth (first calls, larger sizes): the only place that allocates */
      HS_HIP( hipStreamSynchronize(c.stream) );
      if ( c.hpin != NULL ) (void) hipHostFree(c.hpin);
      c.hpin = c.hdev = NULL; c.hcap = 0;
      const long long want = host_doubles + host_doubles / 2 + 4096;
      HS_HIP( hipHostMalloc((void**) &c.hpin, (size_t) want * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent) );
      HS_HIP( hipHostGetDevicePointer((void**) &c.hdev, c.hpin, 0) );
      c.hcap = want;
   }
   if ( dev_doubles > c.dcap )
   {
      HS_HIP( hipStreamSynchronize(c.stream) );
      if ( c.dpool != NULL ) (void) hipFree(c.dpool);
      c.dpool = NULL; c.dcap = 0;
      const long long want = dev_doubles + dev_doubles / 2 + 4096;
      HS_HIP( hipMalloc((void**) &c.dpool, (size_t) want * sizeof(double)) );
      c.dcap = want;
   }
   *out = &c;
   return HS_OK;
}

long long span(int rows, int cols, long long ld) { return rows <= 0 ? 0 : (long long) (rows - 1) * ld + cols; }
long long even(long long n) { return (n + 1) & ~1LL; }

}

extern "C" int hipsdp_dgemm(int device, int layA, int layB, int M, int N, int K, double alpha, const double* A, long long lda,
   const double* B, long long ldb, double beta, double* C, long long ldc, int lower_only, int splitk)
{
   if ( M <= 0 || N <= 0 || K <= 0 || A == NULL || B == NULL || C == NULL )
      return HIPSDP_ERR_ARG;
   const long long na = even(layA == HS_KC ? span(M, K, lda) : span(K, M, lda));
   const long long nb = even(layB == HS_KC ? span(N, K, ldb) : span(K, N, ldb));
   const long long nc = even(span(M, N, ldc));
   if ( splitk <= 0 )
      splitk = hs_dgemm_pick_splitk(M, N, K, lower_only);
   const long long nw = splitk > 1 ? (long long) splitk * M * N : 0;
   const bool direct = na + nb + nc <= HE_DIRECT;
   he_ctx* c = NULL;
   HS_CALL( he_context(device, na + nb + nc, (direct ? 0 : na + nb + nc) + nw, &c) );
   memcpy(c->hpin, A, (size_t) (layA == HS_KC ? span(M, K, lda) : span(K, M, lda)) * sizeof(double));
   memcpy(c->hpin + na, B, (size_t) (layB == HS_KC ? span(N, K, ldb) : span(K, N, ldb)) * sizeof(double));
   if ( beta != 0.0 )
      memcpy(c->hpin + na + nb, C, (size_t) span(M, N, ldc) * sizeof(double));
   double* dA; double* dB; double* dC; double* dW;
   if ( direct )
   {
      dA = c->hdev; dB = c->hdev + na; dC = c->hdev + na + nb; dW = c->dpool;
   }
   else
   {
      dA = c->dpool; dB = c->dpool + na; dC = c->dpool + na + nb; dW = c->dpool + na + nb + nc;
      HS_HIP( hipMemcpyAsync(dA, c->hpin, (size_t) (na + nb + (beta != 0.0 ? nc : 0)) * sizeof(double), hipMemcpyHostToDevice, c->stream) );
   }
   hs_gemm_args g = {M, N, K, layA, layB, dA, lda, 0, dB, ldb, 0, dC, ldc, 0, alpha, beta, 1, lower_only ? HS_GEMM_LOWER : 0, splitk, dW};
   HS_CALL( hs_dgemm(c->stream, &g) );
   if ( !direct )
      HS_HIP( hipMemcpyAsync(c->hpin + na + nb, dC, (size_t) nc * sizeof(double), hipMemcpyDeviceToHost, c->stream) );
   HS_HIP( hipStreamSynchronize(c->stream) );
   if ( lower_only || ldc != N )
   {
      /* only what the product defines is handed back (the caller's other entries stay) */
      for (int i = 0; i < M; ++i)
         memcpy(C + (long long) i * ldc, c->hpin + na + nb + (long long) i * ldc, (size_t) (lower_only ? (i + 1 < N ? i + 1 : N) : N) * sizeof(double));
   }
   else
      memcpy(C, c->hpin + na + nb, (size_t) span(M, N, ldc) * sizeof(double));
   return HIPSDP_OK;
}

extern "C" int hipsdp_gemv_n(int device, int R, long long E, const double* A, int nv, const double* V, double* out)
{
   if ( nv < 1 || nv > 4 || R <= 0 || E <= 0 || A == NULL || V == NULL || out == NULL )
      return HIPSDP_ERR_ARG;
   const long long na = even((long long) R * E), nvv = even((long long) nv * E), no = even((long long) nv * R), nws = 65536;
   const bool direct = na + nvv + no <= HE_DIRECT;
   he_ctx* c = NULL;
   HS_CALL( he_context(device, na + nvv + no, (direct ? 0 : na + nvv + no) + nws, &c) );
   memcpy(c->hpin, A, (size_t) R * E * sizeof(double));
   memcpy(c->hpin + na, V, (size_t) nv * E * sizeof(double));
   double* base = direct ? c->hdev : c->dpool;
   double* dws = direct ? c->dpool : c->dpool + na + nvv + no;
   if ( !direct )
      HS_HIP( hipMemcpyAsync(c->dpool, c->hpin, (size_t) (na + nvv) * sizeof(double), hipMemcpyHostToDevice, c->stream) );
   const double* vp[4];
   for (int v = 0; v < 4; ++v) vp[v] = base + na + (long long) (v < nv ? v : 0) * E;
   HS_CALL( hs_gemv_n(c->stream, R, E, base, E, nv, vp, base + na + nvv, R, dws, nws) );
   if ( !direct )
      HS_HIP( hipMemcpyAsync(c->hpin + na + nvv, c->dpool + na + nvv, (size_t) no * sizeof(double), hipMemcpyDeviceToHost, c->stream) );
   HS_HIP( hipStreamSynchronize(c->stream) );
   memcpy(out, c->hpin + na + nvv, (size_t) nv * R * sizeof(double));
   return HIPSDP_OK;
}

extern "C" int hipsdp_gemv_t(int device, int R, long long E, const double* A, const double* coef, double* out)
{
   if ( R <= 0 || E <= 0 || A == NULL || coef == NULL || out == NULL )
      return HIPSDP_ERR_ARG;
   const long long na = even((long long) R * E), ncf = even(R), no = even(E);
   const bool direct = na + ncf + no <= HE_DIRECT;
   he_ctx* c = NULL;
   HS_CALL( he_context(device, na + ncf + no, direct ? 0 : na + ncf + no, &c) );
   memcpy(c->hpin, A, (size_t) R * E * sizeof(double));
   memcpy(c->hpin + na, coef, (size_t) R * sizeof(double));
   double* base = direct ? c->hdev : c->dpool;
   if ( !direct )
      HS_HIP( hipMemcpyAsync(c->dpool, c->hpin, (size_t) (na + ncf) * sizeof(double), hipMemcpyHostToDevice, c->stream) );
   HS_CALL( hs_gemv_t(c->stream, R, E, base, E, base + na, 0.0, NULL, base + na + ncf) );
   if ( !direct )
      HS_HIP( hipMemcpyAsync(c->hpin + na + ncf, c->dpool + na + ncf, (size_t) no * sizeof(double), hipMemcpyDeviceToHost, c->stream) );
   HS_HIP( hipStreamSynchronize(c->stream) );
   memcpy(out, c->hpin + na + ncf, (size_t) E * sizeof(double));
   return HIPSDP_OK;
}

/* all eigenpairs of the symmetric n x n matrix A, ascending, eigenvectors as rows of V (V may be NULL) */
extern "C" int hipsdp_syev(int device, int n, const double* A, double* lam, double* V)
{
   if ( n <= 0 || A == NULL || lam == NULL )
      return HIPSDP_ERR_ARG;
   /* the sizes the callers of SCIPlapackComputeEigenvectorDecomposition use (blocks of 2-50 rows, and up to 128): tridiagonal
    * reduction, multisection and inverse iteration in ONE launch through pinned staging memory (eigi.hip); HIPSDP_SYEV_JACOBI=1 keeps
    * the Jacobi path */
   static const bool jacobi_small = getenv("HIPSDP_SYEV_JACOBI") != NULL && atoi(getenv("HIPSDP_SYEV_JACOBI")) != 0;
   if ( n <= 128 && !jacobi_small )
      return hipsdp_syev_small(device, n, A, lam, V);
   /* above: block Jacobi on the device (eig.hip), operands copied by the copy engine on the context's stream */
   const long long n2 = even((long long) n * n), nl = even(n), nws = even(hs_syev_ws(n));
   he_ctx* c = NULL;
   HS_CALL( he_context(device, 2 * n2 + nl, 2 * n2 + nl + nws, &c) );
   memcpy(c->hpin, A, (size_t) n * n * sizeof(double));
   double* dA = c->dpool; double* dL = c->dpool + n2; double* dV = c->dpool + n2 + nl; double* dS = c->dpool + 2 * n2 + nl;
   HS_HIP( hipMemcpyAsync(dA, c->hpin, (size_t) n * n * sizeof(double), hipMemcpyHostToDevice, c->stream) );
   HS_CALL( hs_syev_jacobi(c->stream, n, dA, dL, dV, NULL, dS) );
   HS_HIP( hipMemcpyAsync(c->hpin + n2, dL, (size_t) (nl + (V != NULL ? n2 : 0)) * sizeof(double), hipMemcpyDeviceToHost, c->stream) );
   HS_HIP( hipStreamSynchronize(c->stream) );
   memcpy(lam, c->hpin + n2, (size_t) n * sizeof(double));
   if ( V != NULL )
      memcpy(V, c->hpin + n2 + nl, (size_t) n * n * sizeof(double));
   return HIPSDP_OK;
}
