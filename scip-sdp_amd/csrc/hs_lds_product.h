/* hs_lds_product.h - n x n products out of LDS for the single-workgroup kernels of small blocks (kernels.hip, schur.hip, eigi.hip) */
#ifndef HS_LDS_PRODUCT_H
#define HS_LDS_PRODUCT_H
#include "hs_kernels.h"

/* One product of the chain out of LDS: fin(r, c, sum_k a[r][k] b[k][c]) for all n^2 entries, 256 threads, n <= HS_SMALL_N.  A thread
 * owns a 2 x 5 patch of the result: per k two reads of a, five of b and ten multiply-adds in ten independent chains.  The plain form
 * (entry after entry: a dependent chain of n multiply-adds with two LDS reads each) took 30 us per direction block at n = 43
 * (HIPSDP_BATCH_TIMES, example_CLS) - first the latency of the chain, then, with the entries of a thread interleaved, the bandwidth of
 * LDS (18 reads per 9 multiply-adds).  Every sum still runs over k in ascending order: the bits do not depend on the form. */
#define DB_U ((HS_SMALL_N * HS_SMALL_N + 255) / 256)
#define DB_PR 2
#define DB_PC 5
static_assert(((HS_SMALL_N + DB_PR - 1) / DB_PR) * ((HS_SMALL_N + DB_PC - 1) / DB_PC) <= 256, "a patch per thread");      /* callers: n <= HS_SMALL_N */
/* TB = false: b[k][c] at b[k * ld + c];  TB = true: the second factor is given transposed, b[k][c] at b[c * ld + k] (odd ld: the
 * threads of a wavefront still read different banks) */
template<bool TB = false, class F>
__device__ __forceinline__ void db_product(int n, int ld, const double* __restrict__ a, const double* __restrict__ b, F&& fin)
{
   if ( n * n <= 256 )
   {
      /* at most one entry per thread (n <= 16): the patches would leave most of the workgroup idle (n = 10: ten threads) */
      const int e = (int) threadIdx.x;
      if ( e < n * n )
      {
         const int r = e / n, cc = e - r * n;
         double acc = 0.0;
         for (int k = 0; k < n; ++k)
            acc = fma(a[r * ld + k], TB ? b[cc * ld + k] : b[k * ld + cc], acc);
         fin(0, 0, r, cc, acc);
      }
      return;
   }
   const int tc = (n + DB_PC - 1) / DB_PC;
   const int pr = (int) threadIdx.x / tc, pc = (int) threadIdx.x - pr * tc;
   const int r0 = DB_PR * pr, c0 = DB_PC * pc;
   if ( r0 >= n )
      return;                                          /* (no barrier inside) */
   int ra[DB_PR], cb[DB_PC];
#pragma unroll
   for (int i = 0; i < DB_PR; ++i)
      ra[i] = min(r0 + i, n - 1) * ld;
#pragma unroll
   for (int j = 0; j < DB_PC; ++j)
      cb[j] = min(c0 + j, n - 1);
   double acc[DB_PR][DB_PC];
#pragma unroll
   for (int i = 0; i < DB_PR; ++i)
#pragma unroll
      for (int j = 0; j < DB_PC; ++j)
         acc[i][j] = 0.0;
   /* (four steps of k unrolled: 28 LDS reads in flight; one step at a time waits out the LDS latency n times) */
#pragma unroll 4
   for (int k = 0; k < n; ++k)
   {
      double av[DB_PR], bv[DB_PC];
#pragma unroll
      for (int i = 0; i < DB_PR; ++i)
         av[i] = a[ra[i] + k];
#pragma unroll
      for (int j = 0; j < DB_PC; ++j)
         bv[j] = TB ? b[cb[j] * ld + k] : b[k * ld + cb[j]];
#pragma unroll
      for (int i = 0; i < DB_PR; ++i)
#pragma unroll
         for (int j = 0; j < DB_PC; ++j)
            acc[i][j] = fma(av[i], bv[j], acc[i][j]);
   }
#pragma unroll
   for (int i = 0; i < DB_PR; ++i)
#pragma unroll
      for (int j = 0; j < DB_PC; ++j)
         if ( r0 + i < n && c0 + j < n )
            fin(i, j, r0 + i, c0 + j, acc[i][j]);
}

#endif
