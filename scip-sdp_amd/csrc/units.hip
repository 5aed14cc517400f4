/* units.hip - TEST entry points around single device kernels for the per-kernel parity tests in tests/ (self-checks of the GEMM
 * kernels, single factorizations, single Schur assemblies, ...).  Each call: H2D, kernel(s) on the default stream, D2H, with
 * per-call allocations - fine for tests, not for the product: what lapack_interface_hip.c calls (hipsdp_dgemm, hipsdp_gemv_*,
 * hipsdp_syev) lives in host_entries.hip.  Nothing here has a CPU code path. */
#include "hs_kernels.h"
#include "../../include/hipsdp.h"
#include "../../include/hipsdp_units.h"
#include <vector>
#include <cstring>
#include <cstdlib>

namespace {

struct DevBuf
{
   double* p;
   DevBuf() : p(NULL) {}
   ~DevBuf() { if ( p ) (void) hipFree(p); }
   int alloc(long long n) { if ( n <= 0 ) n = 1; hipError_t e = hipMalloc((void**) &p, (size_t) n * sizeof(double)); if ( e != hipSuccess ) { hs_record_hip_error(e, "hipMalloc", __FILE__, __LINE__); return e == hipErrorOutOfMemory ? HS_ERR_NOMEM : HS_ERR_HIP; } return HS_OK; }
   int up(const double* h, long long n) { if ( n <= 0 ) return HS_OK; hipError_t e = hipMemcpy(p, h, (size_t) n * sizeof(double), hipMemcpyHostToDevice); if ( e != hipSuccess ) { hs_record_hip_error(e, "H2D", __FILE__, __LINE__); return HS_ERR_HIP; } return HS_OK; }
   int down(double* h, long long n) { if ( n <= 0 ) return HS_OK; hipError_t e = hipMemcpy(h, p, (size_t) n * sizeof(double), hipMemcpyDeviceToHost); if ( e != hipSuccess ) { hs_record_hip_error(e, "D2H", __FILE__, __LINE__); return HS_ERR_HIP; } return HS_OK; }
};

int pick_device(int device)
{
   int nd = 0;
   if ( hipGetDeviceCount(&nd) != hipSuccess || nd <= 0 )
      return HIPSDP_ERR_NODEVICE;
   if ( device < 0 || device >= nd )
      return HIPSDP_ERR_ARG;
   HS_HIP( hipSetDevice(device) );
   return HS_OK;
}

long long span(int rows, int cols, long long ld) { return rows <= 0 ? 0 : (long long) (rows - 1) * ld + cols; }

}

/* fills with reproducible values in [-0.5, 0.5) */
__global__ void k_unit_fill(long long n, unsigned long long seed, double* __restrict__ x)
{
   for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x)
   {
      unsigned long long h = (unsigned long long) (i + 1) * 0x9E3779B97F4A7C15ULL + seed;
      h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ULL; h ^= h >> 32; h *= 0x94D049BB133111EBULL; h ^= h >> 29;
      x[i] = (double) (h >> 11) * (1.0 / 9007199254740992.0) - 0.5;
   }
}

__global__ void k_unit_maxdiff(long long n, const double* __restrict__ a, const double* __restrict__ b, unsigned long long* __restrict__ ndiff)
{
   unsigned long long cnt = 0;
   for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x)
      if ( __double_as_longlong(a[i]) != __double_as_longlong(b[i]) )
         ++cnt;
   if ( cnt )
      atomicAdd(ndiff, cnt);
}

/* max |a - b| (non-negative doubles order like their bit patterns) */
__global__ void k_unit_maxabsdiff(long long n, const double* __restrict__ a, const double* __restrict__ b, unsigned long long* __restrict__ out)
{
   double m = 0.0;
   for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x)
   {
      const double d = fabs(a[i] - b[i]);
      if ( !(d <= m) )            /* a NaN counts as the largest difference */
         m = (d != d) ? 1e300 : d;
   }
   if ( m > 0.0 )
      atomicMax(out, (unsigned long long) __double_as_longlong(m));
}

/* zero the part of an operand that a triangular flag declares zero: mode 0: X[r][c] = 0 for c > r (rows x cols, K contiguous A:
 * k > m), mode 1: X[r][c] = 0 for r < c (B stored [K][N]: k < n) - the same predicate, kept apart for readability */
__global__ void k_unit_tri(int rows, int cols, long long stride, int batch, double* __restrict__ x)
{
   const long long per = (long long) rows * cols;
   for (long long e = (long long) blockIdx.x * blockDim.x + threadIdx.x; e < per * batch; e += (long long) gridDim.x * blockDim.x)
   {
      const long long b = e / per, w = e - b * per;
      const int r = (int) (w / cols), c = (int) (w - (long long) r * cols);
      if ( c > r )
         x[b * stride + w] = 0.0;
   }
}

/* X[r][c] = 0 for c < r (upper triangular left factor) */
__global__ void k_unit_tri_upper(int rows, int cols, double* __restrict__ x)
{
   const long long per = (long long) rows * cols;
   for (long long e = (long long) blockIdx.x * blockDim.x + threadIdx.x; e < per; e += (long long) gridDim.x * blockDim.x)
   {
      const int r = (int) (e / cols), c = (int) (e - (long long) r * cols);
      if ( c < r )
         x[e] = 0.0;
   }
}

/* The same product through both GEMM kernels (dgemm.hip, dgemm2.hip) on device-generated operands: the results must agree
 * bit for bit.  A is K-contiguous; layB, batch, splitk and flags as in hs_gemm_args (C packed, ldc = N).  used_v2 = 1 when
 * the persistent kernel accepted the shape; ndiff = number of differing elements of C (over all batch entries). */
static int dgemm_selfcheck_impl(int device, int M, int N, int K, int layB, int batch, int splitk, int flags, double alpha, double beta,
   int reps, int* used, long long* ndiff, double* ms_tile, double* ms_fast, double* maxdiff = NULL, long long* nrepro = NULL)
{
   HS_CALL( pick_device(device) );
   if ( M <= 0 || N <= 0 || K <= 0 || batch < 1 )
      return HIPSDP_ERR_ARG;
   const long long na = (long long) M * K, nb = (long long) N * K, nc = (long long) M * N;
   const long long sB = (batch > 1) ? nb : 0;
   DevBuf dA, dB, dC1, dC2, dW;
   unsigned long long* dn = NULL;
   HS_CALL( dA.alloc(na) ); HS_CALL( dB.alloc(batch > 1 ? nb * batch : nb) ); HS_CALL( dC1.alloc(nc * batch) ); HS_CALL( dC2.alloc(nc * batch) );
   if ( splitk > 1 )
      HS_CALL( dW.alloc((long long) splitk * nc) );
   HS_HIP( hipMalloc((void**) &dn, sizeof(unsigned long long)) );
   HS_HIP( hipMemset(dn, 0, sizeof(unsigned long long)) );
   hipLaunchKernelGGL(k_unit_fill, dim3(1024), dim3(256), 0, 0, na, 11ULL, dA.p);
   hipLaunchKernelGGL(k_unit_fill, dim3(1024), dim3(256), 0, 0, batch > 1 ? nb * batch : nb, 23ULL, dB.p);
   hipLaunchKernelGGL(k_unit_fill, dim3(1024), dim3(256), 0, 0, nc * batch, 37ULL, dC1.p);
   if ( flags & HS_GEMM_A_LOWTRI )
      hipLaunchKernelGGL(k_unit_tri, dim3(1024), dim3(256), 0, 0, M, K, 0LL, 1, dA.p);
   if ( (flags & HS_GEMM_B_LOWTRI) && layB == HS_MC )
      hipLaunchKernelGGL(k_unit_tri, dim3(1024), dim3(256), 0, 0, K, N, nb, batch, dB.p);
   if ( flags & HS_GEMM_A_UPTRI )
      hipLaunchKernelGGL(k_unit_tri_upper, dim3(1024), dim3(256), 0, 0, M, K, dA.p);
   HS_HIP( hipMemcpy(dC2.p, dC1.p, (size_t) (nc * batch) * sizeof(double), hipMemcpyDeviceToDevice) );
   hs_gemm_args g = {M, N, K, HS_KC, layB, dA.p, K, 0, dB.p, layB == HS_KC ? (long long) K : (long long) N, sB, dC1.p, N, nc, alpha, beta,
      batch, flags, splitk, dW.p};
   hipEvent_t e0 = NULL, e1 = NULL;
   if ( reps > 0 && (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) )
      return HS_ERR_HIP;
   auto timed = [&](double* ms) -> int
   {
      /* beta = 0 only (the product is repeated on the same C) */
      if ( reps <= 0 || ms == NULL || beta != 0.0 )
         return HS_OK;
      int rc = HS_OK;
      for (int w = 0; w < 2 && rc == HS_OK; ++w)
         rc = hs_dgemm(0, &g);
      (void) hipEventRecord(e0, 0);
      for (int r = 0; r < reps && rc == HS_OK; ++r)
         rc = hs_dgemm(0, &g);
      (void) hipEventRecord(e1, 0);
      if ( rc == HS_OK && hipEventSynchronize(e1) != hipSuccess )
         rc = HS_ERR_HIP;
      float t = 0.f;
      if ( rc == HS_OK && hipEventElapsedTime(&t, e0, e1) != hipSuccess )
         rc = HS_ERR_HIP;
      *ms = (double) t / reps;
      return rc;
   };
   int rc = HS_OK;
   /* reference: the one-tile-per-workgroup kernel (dgemm.hip) alone */
   hs_dgemm2_enable(0);
   rc = hs_dgemm(0, &g);
   if ( rc == HS_OK )
      rc = timed(ms_tile);
   const int before = hs_dgemm2_enable(1);
   g.C = dC2.p;
   /* the second run must not inherit the slabs of the first: a slice one kernel never writes would go unnoticed */
   if ( splitk > 1 )
      hipLaunchKernelGGL(k_unit_fill, dim3(1024), dim3(256), 0, 0, (long long) splitk * nc, 51ULL, dW.p);
   if ( rc == HS_OK )
      rc = hs_dgemm(0, &g);
   const int after = hs_dgemm2_enable(1);
   if ( rc == HS_OK )
      rc = timed(ms_fast);
   if ( rc == HS_OK && hipDeviceSynchronize() != hipSuccess )
      rc = HS_ERR_HIP;
   unsigned long long hn = 0;
   if ( rc == HS_OK )
   {
      hipLaunchKernelGGL(k_unit_maxdiff, dim3(1024), dim3(256), 0, 0, nc * batch, dC1.p, dC2.p, dn);
      if ( hipMemcpy(&hn, dn, sizeof(hn), hipMemcpyDeviceToHost) != hipSuccess )
         rc = HS_ERR_HIP;
   }
   if ( rc == HS_OK && nrepro != NULL )
   {
      /* the default dispatch once more into a third array: the same bits whichever workgroup computed which tile */
      DevBuf dC3;
      unsigned long long hr = 0;
      rc = dC3.alloc(nc * batch);
      if ( rc == HS_OK && hipMemcpy(dC3.p, dC1.p, (size_t) (nc * batch) * sizeof(double), hipMemcpyDeviceToDevice) != hipSuccess )
         rc = HS_ERR_HIP;
      if ( rc == HS_OK && beta != 0.0 )
      {
         /* (beta != 0: C1 holds the tile kernel's result, not the input - regenerate the input) */
         hipLaunchKernelGGL(k_unit_fill, dim3(1024), dim3(256), 0, 0, nc * batch, 37ULL, dC3.p);
         hipLaunchKernelGGL(k_unit_fill, dim3(1024), dim3(256), 0, 0, nc * batch, 37ULL, dC2.p);
         g.C = dC2.p;
         rc = hs_dgemm(0, &g);
      }
      g.C = dC3.p;
      if ( rc == HS_OK )
         rc = hs_dgemm(0, &g);
      if ( rc == HS_OK && hipMemset(dn, 0, sizeof(unsigned long long)) != hipSuccess )
         rc = HS_ERR_HIP;
      if ( rc == HS_OK )
      {
         hipLaunchKernelGGL(k_unit_maxdiff, dim3(1024), dim3(256), 0, 0, nc * batch, dC2.p, dC3.p, dn);
         if ( hipMemcpy(&hr, dn, sizeof(hr), hipMemcpyDeviceToHost) != hipSuccess )
            rc = HS_ERR_HIP;
      }
      *nrepro = (long long) hr;
      g.C = dC2.p;
   }
   if ( rc == HS_OK && maxdiff != NULL )
   {
      unsigned long long bits = 0;
      if ( hipMemset(dn, 0, sizeof(unsigned long long)) != hipSuccess )
         rc = HS_ERR_HIP;
      hipLaunchKernelGGL(k_unit_maxabsdiff, dim3(1024), dim3(256), 0, 0, nc * batch, dC1.p, dC2.p, dn);
      if ( rc == HS_OK && hipMemcpy(&bits, dn, sizeof(bits), hipMemcpyDeviceToHost) != hipSuccess )
         rc = HS_ERR_HIP;
      memcpy(maxdiff, &bits, sizeof(double));
   }
   (void) hipFree(dn);
   if ( e0 != NULL ) (void) hipEventDestroy(e0);
   if ( e1 != NULL ) (void) hipEventDestroy(e1);
   HS_CALL( rc );
   *used = after - before > 0 ? 1 : 0;
   *ndiff = (long long) hn;
   return HIPSDP_OK;
}

extern "C" int hipsdp_dgemm_selfcheck(int device, int M, int N, int K, int layB, int batch, int splitk, int flags, double beta,
   int* used_v2, long long* ndiff)
{
   int used = 0;
   HS_CALL( dgemm_selfcheck_impl(device, M, N, K, layB, batch, splitk, flags, 1.25, beta, 0, &used, ndiff, NULL, NULL) );
   *used_v2 = used & 1;
   return HIPSDP_OK;
}

/* the same with a free alpha and, for reps > 0 and beta = 0, the average time of one product through the tile kernel alone
 * (ms_tile) and through the default dispatch (ms_fast).  *used: bit 0 a persistent kernel of dgemm2.hip took it */
extern "C" int hipsdp_dgemm_selfcheck2(int device, int M, int N, int K, int layB, int batch, int splitk, int flags, double alpha, double beta,
   int reps, int* used, long long* ndiff, double* ms_tile, double* ms_fast)
{
   return dgemm_selfcheck_impl(device, M, N, K, layB, batch, splitk, flags, alpha, beta, reps, used, ndiff, ms_tile, ms_fast);
}

/* the same with the largest absolute difference between the two results (entries are sums of K products of values in
 * [-0.5, 0.5): the paired-band kernel of the triangular products sums the band in another order than the tile kernel) */
extern "C" int hipsdp_dgemm_selfcheck3(int device, int M, int N, int K, int layB, int batch, int splitk, int flags, double alpha, double beta,
   int reps, int* used, long long* ndiff, double* maxdiff, long long* nrepro, double* ms_tile, double* ms_fast)
{
   return dgemm_selfcheck_impl(device, M, N, K, layB, batch, splitk, flags, alpha, beta, reps, used, ndiff, ms_tile, ms_fast, maxdiff, nrepro);
}

/* lower triangle: max |a - b| over r >= c, and the number of elements there whose bits differ */
__global__ void k_unit_lowdiff(int M, const double* __restrict__ a, const double* __restrict__ b, unsigned long long* __restrict__ out)
{
   double m = 0.0;
   unsigned long long cnt = 0;
   const long long MM = (long long) M * M;
   for (long long e = (long long) blockIdx.x * blockDim.x + threadIdx.x; e < MM; e += (long long) gridDim.x * blockDim.x)
   {
      const int r = (int) (e / M), c = (int) (e - (long long) r * M);
      if ( c > r )
         continue;
      const double d = fabs(a[e] - b[e]);
      if ( !(d <= m) )
         m = (d != d) ? 1e300 : d;
      if ( __double_as_longlong(a[e]) != __double_as_longlong(b[e]) )
         ++cnt;
   }
   if ( m > 0.0 )
      atomicMax(out, (unsigned long long) __double_as_longlong(m));
   if ( cnt )
      atomicAdd(out + 1, cnt);
}

/* The Gram product of the Schur assembly, C = W W^T + C on the lower triangle with W [M][K] generated on the device, through the
 * K-sliced tile kernels (hs_dgemm, HS_GEMM_LOWER | HS_GEMM_XCD) and through the Gram kernel (gram.hip): *used = 1 when the Gram
 * kernel took the shape, *maxdiff = largest difference over the lower triangle, *nrepro = elements of the lower triangle that
 * differ between two runs of the Gram kernel (must be 0); reps > 0: average times of the two paths */
extern "C" int hipsdp_gram_selfcheck(int device, int M, long long K, int reps, int* used, double* maxdiff, long long* nrepro, double* ms_tile,
   double* ms_gram)
{
   HS_CALL( pick_device(device) );
   if ( M <= 0 || K <= 0 || used == NULL || maxdiff == NULL || nrepro == NULL )
      return HIPSDP_ERR_ARG;
   const long long MM = (long long) M * M;
   const long long tm = (M + 127) / 128;
   int sk = hs_dgemm_pick_xcd_slices(tm * (tm + 1) / 2, K);
   const int nslab = 64;
   DevBuf dW, dC1, dC2, dC3, dS;
   unsigned long long* dn = NULL;
   HS_CALL( dW.alloc((long long) M * K) ); HS_CALL( dC1.alloc(MM) ); HS_CALL( dC2.alloc(MM) ); HS_CALL( dC3.alloc(MM) ); HS_CALL( dS.alloc((long long) nslab * MM) );
   HS_HIP( hipMalloc((void**) &dn, 2 * sizeof(unsigned long long)) );
   hipLaunchKernelGGL(k_unit_fill, dim3(1024), dim3(256), 0, 0, (long long) M * K, 11ULL, dW.p);
   hipLaunchKernelGGL(k_unit_fill, dim3(1024), dim3(256), 0, 0, MM, 37ULL, dC1.p);
   hipLaunchKernelGGL(k_unit_fill, dim3(1024), dim3(256), 0, 0, MM, 37ULL, dC2.p);
   hipLaunchKernelGGL(k_unit_fill, dim3(1024), dim3(256), 0, 0, MM, 37ULL, dC3.p);
   hipLaunchKernelGGL(k_unit_fill, dim3(1024), dim3(256), 0, 0, (long long) nslab * MM, 51ULL, dS.p);
   hs_gemm_args g3 = {M, M, (int) K, HS_KC, HS_KC, dW.p, K, 0, dW.p, K, 0, dC1.p, M, 0, 1.0, 1.0, 1, HS_GEMM_LOWER | HS_GEMM_XCD | HS_GEMM_NOFAST, sk, dS.p};
   int rc = hs_dgemm(0, &g3);
   int r2 = 0;
   double ex = 0.0;
   if ( rc == HS_OK )
   {
      hipLaunchKernelGGL(k_unit_fill, dim3(1024), dim3(256), 0, 0, (long long) nslab * MM, 52ULL, dS.p);
      r2 = hs_gram_try(0, M, K, dW.p, K, dC2.p, M, 1.0, 1.0, dS.p, nslab, &ex);
      if ( r2 < 0 ) rc = -r2;
   }
   if ( rc == HS_OK && r2 == 1 )
   {
      hipLaunchKernelGGL(k_unit_fill, dim3(1024), dim3(256), 0, 0, (long long) nslab * MM, 53ULL, dS.p);
      r2 = hs_gram_try(0, M, K, dW.p, K, dC3.p, M, 1.0, 1.0, dS.p, nslab, &ex);
      if ( r2 < 0 ) rc = -r2;
   }
   *used = r2 == 1 ? 1 : 0;
   *maxdiff = 0.0; *nrepro = 0;
   unsigned long long h[2] = {0, 0};
   if ( rc == HS_OK && r2 == 1 )
   {
      if ( hipMemset(dn, 0, 2 * sizeof(unsigned long long)) != hipSuccess ) rc = HS_ERR_HIP;
      hipLaunchKernelGGL(k_unit_lowdiff, dim3(1024), dim3(256), 0, 0, M, dC1.p, dC2.p, dn);
      if ( rc == HS_OK && hipMemcpy(h, dn, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess ) rc = HS_ERR_HIP;
      memcpy(maxdiff, &h[0], sizeof(double));
      if ( rc == HS_OK && hipMemset(dn, 0, 2 * sizeof(unsigned long long)) != hipSuccess ) rc = HS_ERR_HIP;
      hipLaunchKernelGGL(k_unit_lowdiff, dim3(1024), dim3(256), 0, 0, M, dC2.p, dC3.p, dn);
      if ( rc == HS_OK && hipMemcpy(h, dn, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess ) rc = HS_ERR_HIP;
      *nrepro = (long long) h[1];
   }
   if ( rc == HS_OK && reps > 0 && ms_tile != NULL && ms_gram != NULL )
   {
      hipEvent_t e0, e1;
      HS_HIP( hipEventCreate(&e0) ); HS_HIP( hipEventCreate(&e1) );
      for (int which = 0; which < 2 && rc == HS_OK; ++which)
      {
         for (int it = -2; it < reps && rc == HS_OK; ++it)
         {
            if ( it == 0 )
               (void) hipEventRecord(e0, 0);
            if ( which == 0 )
               rc = hs_dgemm(0, &g3);
            else if ( r2 == 1 )
            {
               const int r = hs_gram_try(0, M, K, dW.p, K, dC2.p, M, 1.0, 1.0, dS.p, nslab, &ex);
               if ( r < 0 ) rc = -r;
            }
         }
         (void) hipEventRecord(e1, 0);
         float t = 0.f;
         if ( rc == HS_OK && (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&t, e0, e1) != hipSuccess) )
            rc = HS_ERR_HIP;
         *(which == 0 ? ms_tile : ms_gram) = (double) t / reps;
      }
      (void) hipEventDestroy(e0); (void) hipEventDestroy(e1);
   }
   if ( rc == HS_OK && hipDeviceSynchronize() != hipSuccess )
      rc = HS_ERR_HIP;
   (void) hipFree(dn);
   return rc;
}

/* how csrc/gram.hip cuts the Gram product of a shape: partial tiles per off-diagonal / diagonal tile, items, estimated time in units
 * of one off-diagonal tile over all of K */
extern "C" int hipsdp_gram_plan_info(int device, int M, long long K, int nslab, int* no, int* nd, int* nitems, double* span)
{
   HS_CALL( pick_device(device) );
   return hs_gram_plan_info(M, K, nslab, no, nd, nitems, span) ? HIPSDP_OK : HIPSDP_ERR_ARG;
}

extern "C" int hipsdp_schur_dense(int device, int m1, int n, const double* A, const double* X, const double* Zinv, double* Mx,
   double ws_gbytes)
{
   /* ws_gbytes > 0: the chunked U formulation with that budget; ws_gbytes <= 0: the W formulation (needs chol X and the
    * inverse Cholesky factor of Z, both formed here on the device from X and Zinv^-1... the caller passes Zinv, so Z^-1 = G^T G
    * is obtained from the Cholesky factor of Zinv: Zinv = C C^T  ->  G = C^T is upper; to stay with lower factors the W
    * path is exercised through hipsdp_schur_w below) */
   HS_CALL( pick_device(device) );
   const long long n2 = (long long) n * n;
   DevBuf dA, dX, dZ, dM;
   HS_CALL( dA.alloc(m1 * n2) ); HS_CALL( dX.alloc(n2) ); HS_CALL( dZ.alloc(n2) ); HS_CALL( dM.alloc((long long) m1 * m1) );
   HS_CALL( dA.up(A, m1 * n2) ); HS_CALL( dX.up(X, n2) ); HS_CALL( dZ.up(Zinv, n2) );
   hs_schur_ws w;
   HS_CALL( hs_schur_ws_alloc(&w, m1, n2, ws_gbytes > 0.0 ? ws_gbytes : 1e9) );
   if ( ws_gbytes > 0.0 && w.chunk_cols > 1 )
   {
      /* allow chunks smaller than the 128 granularity of the engine so that tiny tests exercise the chunk loop */
      long long cols = (long long) (ws_gbytes * 1e9 / (16.0 * (double) n2));
      if ( cols < 1 ) cols = 1;
      if ( cols < w.chunk_cols ) w.chunk_cols = cols;
   }
   HS_CALL( hs_fill(0, dM.p, (long long) m1 * m1, 0.0) );
   int rc = hs_schur_U(0, m1, n, dA.p, dX.p, dZ.p, dM.p, &w, 0, m1);
   if ( rc == HS_OK ) rc = hs_mirror_lower(0, dM.p, m1, m1);
   if ( rc == HS_OK && hipDeviceSynchronize() != hipSuccess ) rc = HS_ERR_HIP;
   hs_schur_ws_free(&w);
   HS_CALL( rc );
   HS_CALL( dM.down(Mx, (long long) m1 * m1) );
   return HIPSDP_OK;
}

/* time of the Schur work ONE rank of an nranks-way sharded assembly does (row chunks or a column slice, + mirror), on synthetic operands
 * generated in HBM: lets the per-rank cost at 2/4/8 ranks be measured on a one-GPU machine (tests/devtools/shard_time.py) */
extern "C" int hipsdp_schur_shard_time(int device, int m1, int n, int nranks, int rank, int by_columns, int reps, double ws_gbytes,
   double* ms)
{
   HS_CALL( pick_device(device) );
   if ( m1 < 1 || n < 1 || nranks < 1 || rank < 0 || rank >= nranks || reps < 1 || ms == NULL )
      return HIPSDP_ERR_ARG;
   const long long n2 = (long long) n * n;
   DevBuf dA, dX, dZ, dM;
   HS_CALL( dA.alloc(m1 * n2) ); HS_CALL( dX.alloc(n2) ); HS_CALL( dZ.alloc(n2) ); HS_CALL( dM.alloc((long long) (m1 + 32) * m1) );
   hipLaunchKernelGGL(k_unit_fill, dim3(1024), dim3(256), 0, 0, m1 * n2, 11ULL, dA.p);
   hipLaunchKernelGGL(k_unit_fill, dim3(1024), dim3(256), 0, 0, n2, 23ULL, dX.p);
   hipLaunchKernelGGL(k_unit_fill, dim3(1024), dim3(256), 0, 0, n2, 37ULL, dZ.p);
   hs_schur_ws w;
   int c, b1, b2, c0 = 0, cw = 0;
   hs_shard_rows(m1, nranks, rank, &c, &b1, &b2);
   hs_shard_cols(m1, n, nranks, rank, &c0, &cw);
   HS_CALL( hs_schur_ws_alloc(&w, m1, by_columns ? (long long) n * (cw > 0 ? cw : 1) : n2, ws_gbytes > 0.0 ? ws_gbytes : 40.0) );
   hipEvent_t e0, e1;
   HS_HIP( hipEventCreate(&e0) ); HS_HIP( hipEventCreate(&e1) );
   int rc = HS_OK;
   for (int it = 0; it <= reps && rc == HS_OK; ++it)
   {
      if ( it == 1 )
         rc = hipEventRecord(e0, 0) == hipSuccess ? HS_OK : HS_ERR_HIP;        /* iteration 0 is the warm-up */
      if ( rc == HS_OK ) rc = hs_fill(0, dM.p, (long long) m1 * m1, 0.0);
      if ( by_columns )
      {
         if ( rc == HS_OK ) rc = hs_schur_Wcols(0, m1, n, dA.p, dX.p, dZ.p, dM.p, &w, c0, cw);
         if ( rc == HS_OK ) rc = hs_mirror_lower(0, dM.p, m1, m1);
      }
      else
      {
         if ( rc == HS_OK ) rc = hs_schur_Urows(0, m1, n, dA.p, dX.p, dZ.p, dM.p, &w, b1, b1 + c);
         if ( rc == HS_OK ) rc = hs_schur_Urows(0, m1, n, dA.p, dX.p, dZ.p, dM.p, &w, b2, b2 + c);
         if ( rc == HS_OK ) rc = hs_mirror_upper(0, dM.p, m1, m1);
      }
   }
   float t = 0.f;
   if ( rc == HS_OK && (hipEventRecord(e1, 0) != hipSuccess || hipEventSynchronize(e1) != hipSuccess
         || hipEventElapsedTime(&t, e0, e1) != hipSuccess) )
      rc = HS_ERR_HIP;
   (void) hipEventDestroy(e0); (void) hipEventDestroy(e1);
   hs_schur_ws_free(&w);
   HS_CALL( rc );
   *ms = (double) t / (double) reps;
   return HIPSDP_OK;
}

/* the same for the variable-sharded assembly (hs_schur_Wvar): rank `rank` of `nranks` holds only its own rows of A - which is how
 * one rank's share of n = 4000, m = 8000 (128 GB of the 1 TB of A) fits one device.  The all-to-all runs through the measurement
 * transport (nothing but the rank's own piece moves); *a2a_bytes = what the rank would send (= receive) per assembly. */
extern "C" int hipsdp_comm_create_null(int rank, int nranks, void** comm);
extern "C" void hipsdp_comm_destroy(void* comm);
extern "C" int hipsdp_schur_var_share_time(int device, int m1, int n, int nranks, int rank, int cw, int reps, double* ms, double* a2a_bytes)
{
   HS_CALL( pick_device(device) );
   if ( m1 < 1 || n < 1 || nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks || reps < 1 || ms == NULL || cw < 1 )
      return HIPSDP_ERR_ARG;
   if ( cw > n ) cw = n;
   const long long n2 = (long long) n * n;
   int r0, r1, q0, q1;
   hs_var_rows(m1, nranks, rank, &r0, &r1);
   hs_var_wrows(n, nranks, rank, &q0, &q1);
   DevBuf dA, dX, dZ, dM;
   HS_CALL( dA.alloc((long long) (r1 - r0 > 0 ? r1 - r0 : 1) * n2) ); HS_CALL( dX.alloc(n2) ); HS_CALL( dZ.alloc(n2) );
   HS_CALL( dM.alloc((long long) (m1 + 32) * m1) );
   hipLaunchKernelGGL(k_unit_fill, dim3(1024), dim3(256), 0, 0, (long long) (r1 - r0) * n2, 11ULL, dA.p);
   hipLaunchKernelGGL(k_unit_fill, dim3(1024), dim3(256), 0, 0, n2, 23ULL, dX.p);
   hipLaunchKernelGGL(k_unit_fill, dim3(1024), dim3(256), 0, 0, n2, 37ULL, dZ.p);
   void* comm = NULL;
   HS_CALL( hipsdp_comm_create_null(rank, nranks, &comm) );
   hs_schur_ws w;
   int rc = hs_schur_ws_alloc_var(&w, m1, nranks, n, cw);
   hipEvent_t e0, e1;
   HS_HIP( hipEventCreate(&e0) ); HS_HIP( hipEventCreate(&e1) );
   for (int it = 0; it <= reps && rc == HS_OK; ++it)
   {
      if ( it == 1 )
         rc = hipEventRecord(e0, 0) == hipSuccess ? HS_OK : HS_ERR_HIP;        /* iteration 0 is the warm-up */
      if ( rc == HS_OK ) rc = hs_fill(0, dM.p, (long long) m1 * m1, 0.0);
      for (int c0 = 0; c0 < n && rc == HS_OK; c0 += cw)
         rc = hs_schur_Wvar(0, comm, rank, nranks, m1, n, dA.p - (long long) r0 * n2, dX.p, dZ.p, dM.p, &w, c0, n - c0 < cw ? n - c0 : cw);
      if ( rc == HS_OK ) rc = hs_mirror_lower(0, dM.p, m1, m1);
   }
   float t = 0.f;
   if ( rc == HS_OK && (hipEventRecord(e1, 0) != hipSuccess || hipEventSynchronize(e1) != hipSuccess
         || hipEventElapsedTime(&t, e0, e1) != hipSuccess) )
      rc = HS_ERR_HIP;
   (void) hipEventDestroy(e0); (void) hipEventDestroy(e1);
   hs_schur_ws_free(&w);
   hipsdp_comm_destroy(comm);
   HS_CALL( rc );
   *ms = (double) t / (double) reps;
   if ( a2a_bytes != NULL )
      *a2a_bytes = 8.0 * (double) (r1 - r0) * (double) (n - (q1 - q0)) * (double) n;
   return HIPSDP_OK;
}

/* W formulation: X and Z (not its inverse) are given; chol(X), chol(Z), inverse factor and the three GEMMs on the device */
extern "C" int hipsdp_schur_w(int device, int m1, int n, const double* A, const double* X, const double* Z, double* Mx)
{
   HS_CALL( pick_device(device) );
   const long long n2 = (long long) n * n;
   DevBuf dA, dX, dZ, dG, dT, dM, dD;
   int* dflag = NULL;
   HS_CALL( dA.alloc(m1 * n2) ); HS_CALL( dX.alloc(n2) ); HS_CALL( dZ.alloc(n2) ); HS_CALL( dG.alloc(n2) ); HS_CALL( dT.alloc(n2) );
   HS_CALL( dM.alloc((long long) m1 * m1) ); HS_CALL( dD.alloc(hs_potrf_dinv_len(n)) );
   HS_HIP( hipMalloc((void**) &dflag, sizeof(int)) );
   HS_HIP( hipMemset(dflag, 0, sizeof(int)) );
   HS_CALL( dA.up(A, m1 * n2) ); HS_CALL( dX.up(X, n2) ); HS_CALL( dZ.up(Z, n2) );
   hs_schur_ws w;
   int rc = hs_schur_ws_alloc(&w, m1, n2, 1e9);
   if ( rc == HS_OK ) rc = hs_potrf(0, n, dZ.p, dD.p, dflag, NULL);
   if ( rc == HS_OK ) rc = hs_trtri(0, n, dZ.p, dD.p, dG.p, dT.p);
   if ( rc == HS_OK ) rc = hs_potrf(0, n, dX.p, dD.p, dflag, NULL);
   if ( rc == HS_OK ) rc = hs_zero_upper(0, dX.p, n);
   if ( rc == HS_OK ) rc = hs_fill(0, dM.p, (long long) m1 * m1, 0.0);
   if ( rc == HS_OK ) rc = hs_schur_W(0, m1, n, dA.p, dX.p, dG.p, dM.p, &w);
   if ( rc == HS_OK ) rc = hs_mirror_lower(0, dM.p, m1, m1);
   if ( rc == HS_OK && hipDeviceSynchronize() != hipSuccess ) rc = HS_ERR_HIP;
   hs_schur_ws_free(&w);
   (void) hipFree(dflag);
   HS_CALL( rc );
   HS_CALL( dM.down(Mx, (long long) m1 * m1) );
   return HIPSDP_OK;
}

extern "C" int hipsdp_potrf(int device, int n, double* A, int* fail)
{
   HS_CALL( pick_device(device) );
   const long long n2 = (long long) n * n;
   DevBuf dA, dD;
   int* dflag = NULL;
   HS_CALL( dA.alloc(n2) ); HS_CALL( dD.alloc(hs_potrf_dinv_len(n)) );
   HS_HIP( hipMalloc((void**) &dflag, sizeof(int)) );
   HS_HIP( hipMemset(dflag, 0, sizeof(int)) );
   HS_CALL( dA.up(A, n2) );
   int rc = hs_potrf(0, n, dA.p, dD.p, dflag, NULL);
   if ( rc == HS_OK && hipDeviceSynchronize() != hipSuccess ) rc = HS_ERR_HIP;
   int hflag = 0;
   if ( rc == HS_OK && hipMemcpy(&hflag, dflag, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess ) rc = HS_ERR_HIP;
   (void) hipFree(dflag);
   HS_CALL( rc );
   HS_CALL( dA.down(A, n2) );
   if ( fail != NULL ) *fail = hflag;
   return HIPSDP_OK;
}

int hs_potrf_force_v1(int on);

/* test entry: the blocked factorization in both forms (v1 = 1: diagonal kernel + panel GEMM + mask kernel + trailing GEMM per
 * block column; 0: one fused launch per block column), optionally in semidefinite mode (diag0 = the matrix diagonal, forced
 * pivots reported in regmask[n]); dinv: ceil(n / 64) * 4096 doubles (inverses of the diagonal blocks) */
extern "C" int hipsdp_potrf_ex(int device, int n, double* A, int psd, int v1, double* dinv, int* regmask, int* fail)
{
   HS_CALL( pick_device(device) );
   if ( n <= 0 ) return HIPSDP_ERR_ARG;
   const long long n2 = (long long) n * n;
   const long long nd = (long long) ((n + 63) / 64) * 4096;
   DevBuf dA, dD, dG;
   int* dint = NULL;
   HS_CALL( dA.alloc(n2) ); HS_CALL( dD.alloc(hs_potrf_dinv_len(n)) ); HS_CALL( dG.alloc(n) );
   HS_HIP( hipMalloc((void**) &dint, (size_t) (n + 1) * sizeof(int)) );
   HS_HIP( hipMemset(dint, 0, (size_t) (n + 1) * sizeof(int)) );
   HS_HIP( hipMemset(dD.p, 0, (size_t) nd * sizeof(double)) );
   HS_CALL( dA.up(A, n2) );
   std::vector<double> dg(n);
   for (int i = 0; i < n; ++i) dg[i] = A[(long long) i * n + i];
   HS_CALL( dG.up(dg.data(), n) );
   const int old = hs_potrf_force_v1(v1);
   int rc = hs_potrf_psd(0, n, dA.p, dD.p, dint, psd ? dG.p : NULL, psd ? dint + 1 : NULL, 0);
   (void) hs_potrf_force_v1(old);
   if ( rc == HS_OK && hipDeviceSynchronize() != hipSuccess ) rc = HS_ERR_HIP;
   std::vector<int> hint(n + 1, 0);
   if ( rc == HS_OK && hipMemcpy(hint.data(), dint, (size_t) (n + 1) * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess ) rc = HS_ERR_HIP;
   (void) hipFree(dint);
   HS_CALL( rc );
   HS_CALL( dA.down(A, n2) );
   if ( dinv != NULL ) HS_CALL( dD.down(dinv, nd) );
   if ( fail != NULL ) *fail = hint[0];
   if ( regmask != NULL )
      for (int i = 0; i < n; ++i) regmask[i] = hint[1 + i];
   return HIPSDP_OK;
}

extern "C" int hipsdp_potrs(int device, int n, const double* A, int nrhs, double* rhs)
{
   HS_CALL( pick_device(device) );
   if ( nrhs < 1 || nrhs > 4 ) return HIPSDP_ERR_ARG;
   const long long n2 = (long long) n * n;
   DevBuf dA, dD, dR;
   int* dflag = NULL;
   HS_CALL( dA.alloc(n2) ); HS_CALL( dD.alloc(hs_potrf_dinv_len(n)) ); HS_CALL( dR.alloc((long long) nrhs * n) );
   HS_HIP( hipMalloc((void**) &dflag, sizeof(int)) );
   HS_HIP( hipMemset(dflag, 0, sizeof(int)) );
   HS_CALL( dA.up(A, n2) ); HS_CALL( dR.up(rhs, (long long) nrhs * n) );
   /* the solve runs through the multi-workgroup kernels when the factor has at least 3 blocks (as in the engine) */
   int* dsync = NULL;
   int epoch = 0;
   HS_HIP( hipMalloc((void**) &dsync, (size_t) hs_trsv_sync_ws(n) * sizeof(int)) );
   HS_CALL( hs_trsv_sync_init(0, n, dsync, NULL) );
   int rc = hs_potrf(0, n, dA.p, dD.p, dflag, NULL);
   /* mode 7: forward + backward, every diagonal-block solve corrected once with the factor itself, as the engine runs them */
   if ( rc == HS_OK ) rc = hs_trsv_sync(0, n, dA.p, dD.p, nrhs, dR.p, n, 7, dsync, &epoch);
   if ( rc == HS_OK && hipDeviceSynchronize() != hipSuccess ) rc = HS_ERR_HIP;
   (void) hipFree(dsync);
   if ( rc == HS_OK && hipDeviceSynchronize() != hipSuccess ) rc = HS_ERR_HIP;
   int hflag = 0;
   if ( rc == HS_OK && hipMemcpy(&hflag, dflag, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess ) rc = HS_ERR_HIP;
   (void) hipFree(dflag);
   HS_CALL( rc );
   HS_CALL( dR.down(rhs, (long long) nrhs * n) );
   return hflag == 0 ? HIPSDP_OK : HIPSDP_ERR_NUMERIC;
}

extern "C" int hipsdp_trtri(int device, int n, const double* A, double* Linv)
{
   HS_CALL( pick_device(device) );
   const long long n2 = (long long) n * n;
   DevBuf dA, dD, dL, dT;
   int* dflag = NULL;
   HS_CALL( dA.alloc(n2) ); HS_CALL( dD.alloc(hs_potrf_dinv_len(n)) ); HS_CALL( dL.alloc(n2) ); HS_CALL( dT.alloc(n2) );
   HS_HIP( hipMalloc((void**) &dflag, sizeof(int)) );
   HS_HIP( hipMemset(dflag, 0, sizeof(int)) );
   HS_CALL( dA.up(A, n2) );
   int rc = hs_potrf(0, n, dA.p, dD.p, dflag, NULL);
   if ( rc == HS_OK ) rc = hs_trtri(0, n, dA.p, dD.p, dL.p, dT.p);
   if ( rc == HS_OK && hipDeviceSynchronize() != hipSuccess ) rc = HS_ERR_HIP;
   int hflag = 0;
   if ( rc == HS_OK && hipMemcpy(&hflag, dflag, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess ) rc = HS_ERR_HIP;
   (void) hipFree(dflag);
   HS_CALL( rc );
   HS_CALL( dL.down(Linv, n2) );
   return hflag == 0 ? HIPSDP_OK : HIPSDP_ERR_NUMERIC;
}

extern "C" int hipsdp_lambda_min(int device, int n, const double* W, int steps, double* theta, double* resid)
{
   HS_CALL( pick_device(device) );
   if ( n <= 0 ) return HIPSDP_ERR_ARG;
   if ( steps <= 0 ) steps = n < 250 ? n : 250;
   if ( steps > 250 ) steps = 250;
   const long long n2 = (long long) n * n;
   DevBuf dW, dR, dS;
   HS_CALL( dW.alloc(n2) ); HS_CALL( dR.alloc(8) ); HS_CALL( dS.alloc(hs_lanczos_ws(n, steps)) );
   HS_CALL( dW.up(W, n2) );
   HS_CALL( hs_lanczos_lmin(0, n, dW.p, steps, dR.p, dS.p) );
   HS_HIP( hipDeviceSynchronize() );
   double h[3];
   HS_CALL( dR.down(h, 3) );
   *theta = h[0];
   if ( resid != NULL ) *resid = h[1];
   return HIPSDP_OK;
}

/* lambda_min(L D L^T) as the step-length code computes it for small blocks: theta and the residual bound, for the X-side and the
 * Z-side slot at once (same operands) */
extern "C" int hipsdp_lambda_min_scaled(int device, int n, const double* L, const double* D, int steps, double* theta, double* resid)
{
   HS_CALL( pick_device(device) );
   if ( n <= 0 || n > 64 || theta == NULL ) return HIPSDP_ERR_ARG;
   const long long n2 = (long long) n * n;
   DevBuf dL, dD, dR;
   HS_CALL( dL.alloc(n2) ); HS_CALL( dD.alloc(n2) ); HS_CALL( dR.alloc(16) );
   HS_CALL( dL.up(L, n2) ); HS_CALL( dD.up(D, n2) );
   /* through the launch the engine uses: all blocks of a size class at once (here: the same operands as two blocks, whose results
    * must agree bit for bit); n <= 16 exact (k_lmin_tiny), 17 .. 48 exact by reduction + multisection (k_lmin_exact_multi), above
    * Lanczos (k_lanczos_small) */
   DevBuf dR2;
   HS_CALL( dR2.alloc(16) );
   hs_step_jobs J;
   J.nblk = 2;
   for (int j = 0; j < 2; ++j)
   {
      J.n[j] = n; J.L0[j] = dL.p; J.D0[j] = dD.p; J.L1[j] = dL.p; J.D1[j] = dD.p;
      J.res0[j] = (j == 0 ? dR.p : dR2.p); J.res1[j] = (j == 0 ? dR.p : dR2.p) + 8;
   }
   HS_CALL( hs_steplen_small_multi(0, &J, steps > 0 ? steps : 24) );
   HS_HIP( hipDeviceSynchronize() );
   double h[16], h2[16];
   HS_CALL( dR.down(h, 16) );
   HS_CALL( dR2.down(h2, 16) );
   if ( h[0] != h2[0] || h[8] != h2[8] || h[1] != h2[1] || h[9] != h2[9] )
      return HIPSDP_ERR_NUMERIC;
   theta[0] = h[0]; theta[1] = h[8];
   if ( resid != NULL ) { resid[0] = h[1]; resid[1] = h[9]; }
   return HIPSDP_OK;
}

/* out[e] = sum_i coef[i] A[i][e] + sa add[e] (the pass A^T of the engine): 0 = the plain kernel (one thread walks all rows of its
 * entries), 1 = as the engine calls it for blocks with few entries (row chunks side by side + a second launch that adds them in
 * order, when hs_gemv_t_chunks says so: *chunks returns how many).  Two calls with the same operands give the same bits. */
extern "C" int hipsdp_pass_at_unit(int device, int R, long long E, const double* A, const double* coef, double sa, const double* add, int split,
   double* out, int* chunks)
{
   HS_CALL( pick_device(device) );
   if ( R <= 0 || E <= 0 || A == NULL || coef == NULL || out == NULL )
      return HIPSDP_ERR_ARG;
   DevBuf dA, dc, dadd, dout, dws;
   const int C = hs_gemv_t_chunks(R, E);
   if ( chunks != NULL )
      *chunks = C;
   HS_CALL( dA.alloc((long long) R * E) ); HS_CALL( dc.alloc(R) ); HS_CALL( dadd.alloc(E) ); HS_CALL( dout.alloc(E) );
   HS_CALL( dws.alloc((long long) (C > 0 ? C : 1) * E) );
   HS_CALL( dA.up(A, (long long) R * E) ); HS_CALL( dc.up(coef, R) );
   if ( add != NULL )
      HS_CALL( dadd.up(add, E) );
   if ( split )
      HS_CALL( hs_gemv_t_ws(0, R, E, dA.p, E, dc.p, sa, add != NULL ? dadd.p : NULL, dout.p, dws.p, (long long) (C > 0 ? C : 1) * E) );
   else
      HS_CALL( hs_gemv_t(0, R, E, dA.p, E, dc.p, sa, add != NULL ? dadd.p : NULL, dout.p) );
   HS_HIP( hipDeviceSynchronize() );
   HS_CALL( dout.down(out, E) );
   return HIPSDP_OK;
}

/* unit entry of the sparse block mode (csrc/sparse.hip): the Schur entries tr(A_i X A_j Zinv), i, j = 1 .. m, of matrices given as
 * triplets (var 1 .. m, row >= col), assembled by hs_sp_schur exactly as the engine calls it; Mx: (m + 1) x (m + 1), only the lower
 * triangle of the rows / columns 1 .. m is written (the rest is returned zero) */
extern "C" int hipsdp_schur_sparse_unit(int device, int n, int m, long long nnz, const int* var, const int* row, const int* col,
   const double* val, const double* X, const double* Zinv, double* Mx)
{
   HS_CALL( pick_device(device) );
   if ( n <= 0 || m <= 0 || nnz < 0 || X == NULL || Zinv == NULL || Mx == NULL )
      return HIPSDP_ERR_ARG;
   hs_sparse* sp = NULL;
   HS_CALL( hs_sp_build(&sp, n, m, nnz, var, row, col, val) );
   const long long n2 = (long long) n * n, mm = (long long) (m + 1) * (m + 1);
   DevBuf dX, dZ, dM;
   int rc = dX.alloc(n2);
   if ( rc == HS_OK ) rc = dZ.alloc(n2);
   if ( rc == HS_OK ) rc = dM.alloc(mm);
   if ( rc == HS_OK ) rc = dX.up(X, n2);
   if ( rc == HS_OK ) rc = dZ.up(Zinv, n2);
   if ( rc == HS_OK && hipMemset(dM.p, 0, (size_t) mm * sizeof(double)) != hipSuccess ) rc = HS_ERR_HIP;
   if ( rc == HS_OK ) rc = hs_sp_schur(0, sp, dX.p, dZ.p, dM.p);
   if ( rc == HS_OK && hipDeviceSynchronize() != hipSuccess ) rc = HS_ERR_HIP;
   if ( rc == HS_OK ) rc = dM.down(Mx, mm);
   hs_sp_free(sp);
   return rc;
}

/* ---- measured FP64 matrix peak of THIS device ("peak values are measured on the box"): every wavefront of a
 * chip-filling launch issues independent v_mfma_f64_16x16x4_f64 out of registers, nothing else - no memory traffic, no barriers -
 * for about `ms` milliseconds; *tflops = matrix flops / HIP-event time, *ghz = shader clocks / 100 MHz ticks inside the kernel (the
 * frequency the firmware grants under pure matrix load).  The roofline of bench.py is priced against the vendor figure AND this. */
typedef double up_v4d __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_mfma_peak(long long iters, double* __restrict__ sink, unsigned long long* __restrict__ clk)
{
   up_v4d a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
   const double x = 1.0 + 1e-9 * (double) threadIdx.x, y = 1.0 - 1e-9 * (double) threadIdx.x;
   unsigned long long c0 = 0, w0 = 0;
   if ( blockIdx.x == 0 && threadIdx.x == 0 )
   {
      c0 = clock64();
      w0 = wall_clock64();
   }
   /* (tied inline asm: with the builtin the register allocator moved the 64 accumulator registers between the two register files in
    * every trip - 128 copies around 8 matrix instructions, 60 % of the rate) */
#define UP_MFMA(acc, p, q) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(p), "v"(q))
   for (long long i = 0; i < iters; ++i)
   {
      UP_MFMA(a0, x, y); UP_MFMA(a1, y, x); UP_MFMA(a2, x, x); UP_MFMA(a3, y, y);
      UP_MFMA(a4, x, y); UP_MFMA(a5, y, x); UP_MFMA(a6, x, x); UP_MFMA(a7, y, y);
   }
   asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
   if ( blockIdx.x == 0 && threadIdx.x == 0 )
   {
      clk[0] = clock64() - c0;
      clk[1] = wall_clock64() - w0;
   }
   const up_v4d t = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
   if ( t[0] + t[1] + t[2] + t[3] == 12345.678 )          /* (never: keeps the accumulators alive) */
      sink[threadIdx.x] = t[0];
}

extern "C" int hipsdp_mfma_peak(int device, double ms, double* tflops, double* ghz)
{
   HS_CALL( pick_device(device) );
   if ( !(ms > 0.0) ) ms = 20.0;
   DevBuf sink, clk;
   HS_CALL( sink.alloc(256) ); HS_CALL( clk.alloc(2) );
   const int cus = hs_device_cus() > 0 ? hs_device_cus() : 256;
   hipEvent_t e0, e1;
   HS_HIP( hipEventCreate(&e0) ); HS_HIP( hipEventCreate(&e1) );
   double best = 0.0, bestghz = 0.0;
   /* one, two and four workgroups of four wavefronts per compute unit: the best of them is the figure (which occupancy feeds the
    * matrix pipes best is the device's business) */
   for (int wgpc = 1; wgpc <= 4; wgpc *= 2)
   {
      const int grid = cus * wgpc;
      long long iters = 20000;
      float t = 0.f;
      for (int pass = 0; pass < 3; ++pass)                         /* pass 0 warms up and calibrates the trip count */
      {
         HS_HIP( hipEventRecord(e0, 0) );
         hipLaunchKernelGGL(k_mfma_peak, dim3(grid), dim3(256), 0, 0, iters, sink.p, reinterpret_cast<unsigned long long*>(clk.p));
         HS_HIP( hipEventRecord(e1, 0) );
         HS_HIP( hipEventSynchronize(e1) );
         HS_HIP( hipEventElapsedTime(&t, e0, e1) );
         if ( pass == 0 && t > 0.f )
         {
            iters = (long long) ((double) iters * ms / (double) t);
            if ( iters < 1000 ) iters = 1000;
         }
      }
      unsigned long long h[2] = {0, 0};
      HS_HIP( hipMemcpy(h, clk.p, sizeof(h), hipMemcpyDeviceToHost) );
      const double flops = (double) iters * 8.0 * 2048.0 * 4.0 * (double) grid;      /* 2048 flops per instruction, 4 wavefronts per workgroup */
      const double tf = flops / ((double) t * 1e-3) / 1e12;
      if ( tf > best )
      {
         best = tf;
         bestghz = h[1] > 0 ? (double) h[0] / ((double) h[1] * 10.0) : 0.0;      /* wall ticks are 10 ns */
      }
   }
   (void) hipEventDestroy(e0); (void) hipEventDestroy(e1);
   if ( tflops != NULL ) *tflops = best;
   if ( ghz != NULL ) *ghz = bestghz;
   return HIPSDP_OK;
}
