/* psd.hip - the PSD projection chain of the warm-start producer, fused on the device.
 *
 * Reference: src/scipsdp/relax_sdp.c:2715-2766 (dual matrix Z of a block) and :3405-3445 (primal matrix X), helpers
 * expandSparseMatrix :243-273 and scaleTransposedMatrix :276-302.  There the chain is five host steps with two trips through
 * SCIPlapack* (eigen-decomposition, then DGEMM):
 *    sparse lower triangle -> dense;  (lambda, V) = eig, V[k][:] = k-th eigenvector, ascending;
 *    lambda_k < minev -> minev (SCIPisLT, i.e. by more than epsilon);  S = copy of V with entry [r][c] multiplied by lambda_c;
 *    R = DGEMM(V, 'T', S, 'N');  upper triangle of R with |entry| > epsilon -> sparse (row <= col, row-major order).
 * Read in row-major terms the product is  R[i][j] = sum_c V[i][c] lambda_c V[j][c]  - the eigenvalues weight the COMPONENTS
 * of the eigenvectors, not the eigenvectors (scaleTransposedMatrix scales column c of the array whose ROWS are the
 * eigenvectors).  mode 0 reproduces exactly that; mode 1 is the spectral form  R = sum_k lambda_k v_k v_k^T  (the projection
 * onto {X : X >= minev I} the comment at relax_sdp.c:2747 describes).  Both keep every step on the device: one upload of the
 * triplets, one download of the triplets. */
#include "hs_kernels.h"
#include "../../include/hipsdp.h"
#include <cstdlib>
#include <cstring>

namespace {

__global__ void k_expand_coo(int nnz, int n, const int* __restrict__ row, const int* __restrict__ col, const double* __restrict__ val,
   double* __restrict__ A)
{
   for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += gridDim.x * blockDim.x)
   {
      const int r = row[e], c = col[e];
      A[(long long) r * n + c] = val[e];
      A[(long long) c * n + r] = val[e];
   }
}

/* S = V with the clamped eigenvalues multiplied in: mode 0 S[k][j] = V[k][j] * lam'_j, mode 1 S[k][j] = lam'_k * V[k][j] */
__global__ void k_clamp_scale(int n, const double* __restrict__ V, const double* __restrict__ lam, double minev, double eps, int mode,
   double* __restrict__ S)
{
   const long long n2 = (long long) n * n;
   for (long long e = (long long) blockIdx.x * blockDim.x + threadIdx.x; e < n2; e += (long long) gridDim.x * blockDim.x)
   {
      const int k = (int) (e / n), j = (int) (e - (long long) k * n);
      double l = lam[mode == 0 ? j : k];
      if ( l - minev < -eps )
         l = minev;
      S[e] = V[e] * l;
   }
}

/* cnt[r] = number of c >= r with |R[r][c]| > eps */
__global__ void __launch_bounds__(256) k_row_counts(int n, const double* __restrict__ R, double eps, int* __restrict__ cnt)
{
   __shared__ int sh[4];
   const int r = blockIdx.x;
   int c0 = 0;
   for (int c = r + threadIdx.x; c < n; c += 256)
      c0 += fabs(R[(long long) r * n + c]) > eps ? 1 : 0;
   for (int off = 32; off > 0; off >>= 1)
      c0 += __shfl_down(c0, off, 64);
   if ( (threadIdx.x & 63) == 0 )
      sh[threadIdx.x >> 6] = c0;
   __syncthreads();
   if ( threadIdx.x == 0 )
      cnt[r] = sh[0] + sh[1] + sh[2] + sh[3];
}

/* off[r] = sum of cnt[0 .. r-1], off[n] = total; one workgroup */
__global__ void __launch_bounds__(1024) k_scan_counts(int n, const int* __restrict__ cnt, int* __restrict__ off)
{
   __shared__ int sh[1024];
   __shared__ int carry;
   if ( threadIdx.x == 0 )
      carry = 0;
   __syncthreads();
   for (int base = 0; base < n; base += 1024)
   {
      const int i = base + threadIdx.x;
      const int v = i < n ? cnt[i] : 0;
      sh[threadIdx.x] = v;
      __syncthreads();
      for (int d = 1; d < 1024; d <<= 1)
      {
         const int t = threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
         __syncthreads();
         sh[threadIdx.x] += t;
         __syncthreads();
      }
      if ( i < n )
         off[i] = carry + sh[threadIdx.x] - v;
      __syncthreads();
      if ( threadIdx.x == 1023 )
         carry += sh[1023];
      __syncthreads();
   }
   if ( threadIdx.x == 0 )
      off[n] = carry;
}

/* row r writes its kept entries in column order starting at off[r] (one wavefront per row: ballot-ordered compaction) */
__global__ void __launch_bounds__(64) k_write_rows(int n, const double* __restrict__ R, double eps, const int* __restrict__ off, int cap,
   int* __restrict__ row, int* __restrict__ col, double* __restrict__ val)
{
   const int r = blockIdx.x;
   int pos = off[r];
   for (int base = r; base < n; base += 64)
   {
      const int c = base + threadIdx.x;
      const double v = c < n ? R[(long long) r * n + c] : 0.0;
      const bool keep = c < n && fabs(v) > eps;
      const unsigned long long mask = __ballot(keep);
      const int before = __popcll(mask & ((1ULL << threadIdx.x) - 1ULL));
      if ( keep && pos + before < cap )
      {
         row[pos + before] = r;
         col[pos + before] = c;
         val[pos + before] = v;
      }
      pos += __popcll(mask);
   }
}

/* pinned, device-mapped staging of the calling thread for the sizes the warm-start producer meets (blocks of up to 128 rows): the
 * kernels read the triplets and write the result there - with pageable arrays every one of the seven copies of a call blocks for
 * 15-20 us, 120 us of the 200 a projection of order 10 took */
#define PSD_PIN_ENTRIES 32768
struct PsdPin
{
   char* h; char* d; int device;
   PsdPin() : h(NULL), d(NULL), device(-1) {}
   ~PsdPin() { if ( h != NULL ) (void) hipHostFree(h); }
   /* layout: [in rows | in cols | in vals | total | out rows | out cols | out vals] */
   static size_t bytes() { return (size_t) PSD_PIN_ENTRIES * (4 * sizeof(int) + 2 * sizeof(double)) + 64; }
   bool get(int dev)
   {
      if ( h != NULL && device == dev )
         return true;
      if ( h != NULL ) { (void) hipHostFree(h); h = NULL; d = NULL; }
      void* hp = NULL; void* dp = NULL;
      if ( hipHostMalloc(&hp, bytes(), hipHostMallocMapped) != hipSuccess )
         return false;
      if ( hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess )
      {
         (void) hipHostFree(hp);
         return false;
      }
      h = (char*) hp; d = (char*) dp; device = dev;
      return true;
   }
};
thread_local PsdPin g_pin;

/* device -> pageable host memory through the two halves of the thread's pinned staging, asynchronous copies on `st` */
static int d2h_through_staging(int device, hipStream_t st, void* dst, const void* src, size_t bytes)
{
   if ( bytes == 0 )
      return HS_OK;
   if ( !g_pin.get(device) )
   {
      HS_HIP( hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) );
      return HS_OK;
   }
   const size_t half = (PsdPin::bytes() / 2) & ~(size_t) 255;
   hipEvent_t ev[2] = {NULL, NULL};
   HS_HIP( hipEventCreateWithFlags(&ev[0], hipEventDisableTiming) );
   if ( hipEventCreateWithFlags(&ev[1], hipEventDisableTiming) != hipSuccess )
   {
      (void) hipEventDestroy(ev[0]);
      return HS_ERR_HIP;
   }
   int rc = HS_OK;
   size_t issued = 0, taken = 0;
   int nissued = 0, ntaken = 0;
   while ( taken < bytes && rc == HS_OK )
   {
      /* keep two chunks in flight */
      while ( issued < bytes && nissued - ntaken < 2 && rc == HS_OK )
      {
         const size_t len = bytes - issued < half ? bytes - issued : half;
         const int b = nissued & 1;
         if ( hipMemcpyAsync(g_pin.h + b * half, (const char*) src + issued, len, hipMemcpyDeviceToHost, st) != hipSuccess
            || hipEventRecord(ev[b], st) != hipSuccess )
            rc = HS_ERR_HIP;
         issued += len; ++nissued;
      }
      if ( rc != HS_OK )
         break;
      const int b = ntaken & 1;
      const size_t len = bytes - taken < half ? bytes - taken : half;
      if ( hipEventSynchronize(ev[b]) != hipSuccess )
         rc = HS_ERR_HIP;
      else
         memcpy((char*) dst + taken, g_pin.h + b * half, len);
      taken += len; ++ntaken;
   }
   if ( rc != HS_OK )
      (void) hipStreamSynchronize(st);
   (void) hipEventDestroy(ev[0]); (void) hipEventDestroy(ev[1]);
   return rc;
}

template<class T> struct PoolBuf
{
   T* p;
   PoolBuf() : p(NULL) {}
   ~PoolBuf() { if ( p != NULL ) hs_pool_free(p); }
   int alloc(long long count) { return hs_pool_alloc((void**) &p, (size_t) (count > 0 ? count : 1) * sizeof(T)); }
};

}

extern "C" int hipsdp_psd_project(int device, int n, int nnz, const int* row, const int* col, const double* val, double minev,
   double epsilon, int mode, int cap, int* nnz_out, int* rowout, int* colout, double* valout)
{
   int nd = 0;
   if ( hipGetDeviceCount(&nd) != hipSuccess || nd <= 0 )
      return HIPSDP_ERR_NODEVICE;
   if ( device < 0 || device >= nd || n <= 0 || nnz < 0 || (nnz > 0 && (row == NULL || col == NULL || val == NULL)) || nnz_out == NULL
      || mode < 0 || mode > 1 || cap < 0 || (cap > 0 && (rowout == NULL || colout == NULL || valout == NULL)) )
      return HIPSDP_ERR_ARG;
   for (int e = 0; e < nnz; ++e)
      if ( row[e] < 0 || row[e] >= n || col[e] < 0 || col[e] >= n )
         return HIPSDP_ERR_ARG;
   HS_HIP( hipSetDevice(device) );
   hipStream_t st = 0;
   const long long n2 = (long long) n * n;
   PoolBuf<double> A, lam, V, S, ws, dval, oval;
   PoolBuf<int> drow, dcol, cnt, off, orow, ocol;
   HS_CALL( A.alloc(n2) ); HS_CALL( lam.alloc(n) ); HS_CALL( V.alloc(n2) ); HS_CALL( S.alloc(n2) ); HS_CALL( ws.alloc(hs_syev_ws(n)) );
   HS_CALL( drow.alloc(nnz) ); HS_CALL( dcol.alloc(nnz) ); HS_CALL( dval.alloc(nnz) );
   HS_CALL( cnt.alloc(n) ); HS_CALL( off.alloc(n + 1) );
   /* small operands: triplets and result through the thread's pinned staging memory (no copy of a call blocks) */
   const bool pinned = nnz <= PSD_PIN_ENTRIES && cap <= PSD_PIN_ENTRIES && g_pin.get(device);
   int *pin_total = NULL, *pin_orow = NULL, *pin_ocol = NULL;
   double* pin_oval = NULL;
   const int *in_row = NULL, *in_col = NULL;
   const double* in_val = NULL;
   int *out_row = NULL, *out_col = NULL;
   double* out_val = NULL;
   if ( pinned )
   {
      const size_t E = PSD_PIN_ENTRIES;
      char* h = g_pin.h; char* d = g_pin.d;
      if ( nnz > 0 )
      {
         memcpy(h, row, (size_t) nnz * sizeof(int));
         memcpy(h + E * sizeof(int), col, (size_t) nnz * sizeof(int));
         memcpy(h + 2 * E * sizeof(int), val, (size_t) nnz * sizeof(double));
      }
      in_row = (const int*) d; in_col = (const int*) (d + E * sizeof(int)); in_val = (const double*) (d + 2 * E * sizeof(int));
      const size_t o0 = 2 * E * sizeof(int) + E * sizeof(double);
      pin_total = (int*) (h + o0);
      pin_orow = (int*) (h + o0 + 64); out_row = (int*) (d + o0 + 64);
      pin_ocol = (int*) (h + o0 + 64 + E * sizeof(int)); out_col = (int*) (d + o0 + 64 + E * sizeof(int));
      pin_oval = (double*) (h + o0 + 64 + 2 * E * sizeof(int)); out_val = (double*) (d + o0 + 64 + 2 * E * sizeof(int));
   }
   else
   {
      HS_CALL( orow.alloc(cap) ); HS_CALL( ocol.alloc(cap) ); HS_CALL( oval.alloc(cap) );
      out_row = orow.p; out_col = ocol.p; out_val = oval.p;
   }
   HS_HIP( hipMemsetAsync(A.p, 0, (size_t) n2 * sizeof(double), st) );
   if ( nnz > 0 )
   {
      if ( !pinned )
      {
         HS_HIP( hipMemcpyAsync(drow.p, row, (size_t) nnz * sizeof(int), hipMemcpyHostToDevice, st) );
         HS_HIP( hipMemcpyAsync(dcol.p, col, (size_t) nnz * sizeof(int), hipMemcpyHostToDevice, st) );
         HS_HIP( hipMemcpyAsync(dval.p, val, (size_t) nnz * sizeof(double), hipMemcpyHostToDevice, st) );
         in_row = drow.p; in_col = dcol.p; in_val = dval.p;
      }
      int g = (nnz + 255) / 256; if ( g > 1024 ) g = 1024;
      hipLaunchKernelGGL(k_expand_coo, dim3(g), dim3(256), 0, st, nnz, n, in_row, in_col, in_val, A.p);
   }
   /* the decomposition SCIPlapackComputeEigenvectorDecomposition returns for this size (the literal chain of mode 0 depends on the
    * basis: both paths must use the same eigenvectors) */
   static const bool jacobi_small = getenv("HIPSDP_SYEV_JACOBI") != NULL && atoi(getenv("HIPSDP_SYEV_JACOBI")) != 0;
   PoolBuf<double> scr;                              /* (lives to the end of the call: no wait in the middle of the chain) */
   if ( n <= 128 && !jacobi_small )
   {
      HS_CALL( scr.alloc(hs_syev_small_scratch(n)) );
      HS_CALL( hs_syev_small_dev(st, n, A.p, lam.p, V.p, scr.p) );
   }
   else
      HS_CALL( hs_syev_jacobi(st, n, A.p, lam.p, V.p, NULL, ws.p) );
   long long g2 = (n2 + 255) / 256; if ( g2 > 4096 ) g2 = 4096;
   hipLaunchKernelGGL(k_clamp_scale, dim3((unsigned) g2), dim3(256), 0, st, n, V.p, lam.p, minev, epsilon, mode, S.p);
   {
      /* mode 0: R = V S^T (both operands K contiguous); mode 1: R = V^T S (both M / N contiguous); R goes to A */
      hs_gemm_args ga = {n, n, n, mode == 0 ? HS_KC : HS_MC, mode == 0 ? HS_KC : HS_MC, V.p, n, 0, S.p, n, 0, A.p, n, 0, 1.0, 0.0, 1, 0, 1, NULL};
      HS_CALL( hs_dgemm(st, &ga) );
   }
   hipLaunchKernelGGL(k_row_counts, dim3(n), dim3(256), 0, st, n, A.p, epsilon, cnt.p);
   hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, st, n, cnt.p, off.p);
   hipLaunchKernelGGL(k_write_rows, dim3(n), dim3(64), 0, st, n, A.p, epsilon, off.p, cap, out_row, out_col, out_val);
   if ( hipGetLastError() != hipSuccess )
      return HIPSDP_ERR_HIP;
   int total = 0;
   if ( pinned )
   {
      HS_HIP( hipMemcpyAsync(pin_total, off.p + n, sizeof(int), hipMemcpyDeviceToHost, st) );      /* (pinned destination: no staging) */
      HS_HIP( hipStreamSynchronize(st) );
      total = *pin_total;
   }
   else
   {
      HS_HIP( hipMemcpyAsync(&total, off.p + n, sizeof(int), hipMemcpyDeviceToHost, st) );
      HS_HIP( hipStreamSynchronize(st) );
   }
   *nnz_out = total;
   if ( total > cap )
      return HIPSDP_ERR_ARG;             /* the needed length is in *nnz_out */
   if ( total > 0 && pinned )
   {
      memcpy(rowout, pin_orow, (size_t) total * sizeof(int));
      memcpy(colout, pin_ocol, (size_t) total * sizeof(int));
      memcpy(valout, pin_oval, (size_t) total * sizeof(double));
   }
   else if ( total > 0 )
   {
      /* results larger than the pinned staging (blocks above about 180 rows): in chunks THROUGH the staging memory, two halves in turn -
       * the copy engine fills one half while the host empties the other; a pageable destination would make every copy block and be
       * staged by the runtime page by page (round 4: three blocking pageable copies) */
      HS_CALL( d2h_through_staging(device, st, rowout, orow.p, (size_t) total * sizeof(int)) );
      HS_CALL( d2h_through_staging(device, st, colout, ocol.p, (size_t) total * sizeof(int)) );
      HS_CALL( d2h_through_staging(device, st, valout, oval.p, (size_t) total * sizeof(double)) );
   }
   return HIPSDP_OK;
}
