/* schur.hip - assembly of the (extended) Schur complement of one dense block,
 *
 *      Mx[i][j] = tr(A_i X A_j Z^-1),   i, j = 0 .. m1 - 1        (row 0 = constant matrix A_0, rows 1.. = variables)
 *
 * as FP64-MFMA GEMMs over the row storage A[m1][n^2].  This is the step north_star names first; in the reference it happens
 * inside DSDP/SDPA (sdpisolver_dsdp.c:1503, sdpisolver_sdpa.cpp:1620).  Two formulations:
 *
 *  hs_schur_W  (single GPU, needs 2 m1 n^2 doubles of workspace):  with X = R R^T (R = chol X) and Z^-1 = G^T G
 *              (G = inverse Cholesky factor of Z)        W_j = G A_j R,      Mx = W W^T   (SYRK over K = n^2)
 *              GEMM1  T = [A_0; ..; A_m] R        one (m1 n) x n x n product, R lower triangular: K range starts at n0
 *              GEMM2  W_j = G T_j                 batched, G lower triangular: K range ends at m0 + tile
 *              GEMM3  Mx += W W^T                 lower tiles only; K is cut into 8c slices, the workgroups of one XCD
 *                                                 walk the same slice so W is read from HBM once, then served by L2
 *  hs_schur_U  (chunked over variables, shards over GPUs)          U_j = X A_j Z^-1,   Mx[:, J] = A_flat U_J^T
 *              only the rows/columns of a chunk J are needed at a time: a rank computes its J and the row blocks are
 *              all-gathered (multi.hip); workspace 2 |J| n^2.
 *
 * Algorithmic flops per assembly: 4 m1 n^3 + m1^2 n^2.
 */
#include "hs_kernels.h"
#include "hs_lds_product.h"
#include <vector>
#include <cmath>
#include <stdlib.h>

int hs_schur_ws_alloc(hs_schur_ws* w, int m1, long long n2max, double budget_gb)
{
   w->T = w->U = w->K = w->V = w->U2 = w->V2 = NULL;
   w->evP[0] = w->evP[1] = w->evX[0] = w->evX[1] = NULL;
   w->ev_g2 = NULL;
   w->after_g1 = NULL; w->after_g1_arg = NULL;
   w->capT = w->capV = 0;
   const double need_full = 2.0 * 8.0 * (double) m1 * (double) n2max;
   long long cols = m1;
   w->full = 1;
   if ( need_full > budget_gb * 1e9 )
   {
      w->full = 0;
      cols = (long long) (budget_gb * 1e9 / (2.0 * 8.0 * (double) n2max));
      cols = (cols / 128) * 128;
      if ( cols < 128 ) cols = 128;
      if ( cols > m1 ) cols = m1;
   }
   w->chunk_cols = cols;
   w->n2 = n2max;
   hipError_t e = hs_pool_alloc((void**) &w->T, (size_t) (cols * n2max) * sizeof(double)) == HS_OK ? hipSuccess : hipErrorOutOfMemory;
   if ( e == hipSuccess ) e = hs_pool_alloc((void**) &w->U, (size_t) (cols * n2max) * sizeof(double)) == HS_OK ? hipSuccess : hipErrorOutOfMemory;
   if ( e != hipSuccess )
   {
      hs_record_hip_error(e, "hipMalloc(schur workspace)", __FILE__, __LINE__);
      return e == hipErrorOutOfMemory ? HS_ERR_NOMEM : HS_ERR_HIP;
   }
   int sk = hs_dgemm_pick_splitk(m1, (int) cols, (int) (n2max > 2000000000LL ? 2000000000LL : n2max), 1);
   if ( sk < 16 ) sk = 16;
   if ( w->full )
   {
      const long long tm = (m1 + 127) / 128;
      const int sx = hs_dgemm_pick_xcd_slices(tm * (tm + 1) / 2, n2max);
      if ( sx > sk ) sk = sx;
      if ( sk < 64 ) sk = 64;
   }
   w->kws_len = (long long) sk * m1 * (w->full ? m1 : cols);
   e = hs_pool_alloc((void**) &w->K, (size_t) w->kws_len * sizeof(double)) == HS_OK ? hipSuccess : hipErrorOutOfMemory;
   if ( e != hipSuccess )
   {
      hs_record_hip_error(e, "hipMalloc(split-K slabs)", __FILE__, __LINE__);
      return e == hipErrorOutOfMemory ? HS_ERR_NOMEM : HS_ERR_HIP;
   }
   return HS_OK;
}

void hs_schur_ws_free(hs_schur_ws* w)
{
   hs_pool_free(w->T);
   hs_pool_free(w->U);
   hs_pool_free(w->K);
   hs_pool_free(w->V);
   hs_pool_free(w->U2);
   hs_pool_free(w->V2);
   for (int i = 0; i < 2; ++i)
   {
      if ( w->evP[i] != NULL ) (void) hipEventDestroy((hipEvent_t) w->evP[i]);
      if ( w->evX[i] != NULL ) (void) hipEventDestroy((hipEvent_t) w->evX[i]);
      w->evP[i] = w->evX[i] = NULL;
   }
   w->T = w->U = w->K = w->V = w->U2 = w->V2 = NULL;
}

int hs_schur_U(hipStream_t s, int m1, int n, const double* A, const double* X, const double* Zinv, double* Mx,
   hs_schur_ws* w, int j_begin, int j_end)
{
   const long long n2 = (long long) n * n;
   if ( n2 > 2000000000LL )
      return HS_ERR_ARG;
   for (int j0 = j_begin; j0 < j_end; j0 += (int) w->chunk_cols)
   {
      const int cj = (j_end - j0) < w->chunk_cols ? (j_end - j0) : (int) w->chunk_cols;
      const long long rows = (long long) cj * n;
      if ( rows > 2000000000LL )
         return HS_ERR_ARG;
      /* GEMM1: T[(cj n) x n] = A[j0 .. j0 + cj) (stack of n x n) * Zinv */
      hs_gemm_args g1 = {(int) rows, n, n, HS_KC, HS_MC, A + (long long) j0 * n2, n, 0, Zinv, n, 0, w->T, n, 0, 1.0, 0.0, 1, HS_GEMM_REMAP, 1, NULL};
      HS_CALL( hs_dgemm(s, &g1) );
      /* GEMM2: U_j = X * T_j, batched over the cj matrices of the chunk */
      hs_gemm_args g2 = {n, n, n, HS_KC, HS_MC, X, n, 0, w->T, n, n2, w->U, n, n2, 1.0, 0.0, cj, HS_GEMM_REMAP, 1, NULL};
      HS_CALL( hs_dgemm(s, &g2) );
      /* GEMM3: Mx[j0:, j0:j0+cj] += A_flat[j0:] * U_flat^T   (rows i >= j0 only: lower triangle) */
      const int rowsM = m1 - j0;
      int sk = hs_dgemm_pick_splitk(rowsM, cj, (int) n2, 1);
      while ( sk > 1 && (long long) sk * rowsM * cj > w->kws_len ) --sk;
      hs_gemm_args g3 = {rowsM, cj, (int) n2, HS_KC, HS_KC, A + (long long) j0 * n2, n2, 0, w->U, n2, 0,
         Mx + (long long) j0 * m1 + j0, m1, 0, 1.0, 1.0, 1, HS_GEMM_LOWER, sk, w->K};
      HS_CALL( hs_dgemm(s, &g3) );
   }
   return HS_OK;
}

/* Mx += V V^T (lower triangle) through the Gram kernel (gram.hip) when it takes the shape: *done = 1 */
static int schur_gram(hipStream_t s, int m1, long long K, const double* V, long long ldv, double* Mx, hs_schur_ws* w, int* done)
{
   *done = 0;
   long long nslab = w->kws_len / ((long long) m1 * m1);
   if ( nslab > 64 ) nslab = 64;
   double executed = 0.0;
   const int r = hs_gram_try(s, m1, K, V, ldv, Mx, m1, 1.0, 1.0, w->K, (int) nslab, &executed);
   if ( r < 0 )
      return -r;
   if ( r == 1 )
   {
      hs_mfma_flops_add(executed);
      *done = 1;
   }
   return HS_OK;
}

/* smallest m1 for which the Gram product runs in XCD-walked K slices (below: plain split-K over the lower tiles) */
static int hs_syrk_min_m1(void)
{
   const char* e = getenv("HIPSDP_SYRK_MINM");
   return e != NULL && atoi(e) > 0 ? atoi(e) : 256;
}

/* Slab count and flags of the tile-kernel form of Mx += V V^T (K contiguous): XCD-walked K slices when `xcd` and at least two m1 x m1
 * slabs fit the workspace, plain split-K otherwise, and ONE pass straight into Mx (no slabs, ws unused) when not even two fit - a
 * chunked workspace (w->full == 0) holds sk m1 cols doubles, which can be less than 2 m1^2 (ws_gbytes = 0.1,
 * n = 500, m = 2000 gives cols = 128).  The slab count never exceeds what kws_len holds. */
static void schur_syrk_shape(const hs_schur_ws* w, int m1, long long K, bool xcd, int* flags, int* sk)
{
   const long long slab = (long long) m1 * m1;
   *flags = HS_GEMM_LOWER;
   if ( xcd && 2 * slab <= w->kws_len )
   {
      const long long tm = (m1 + 127) / 128;
      int s = hs_dgemm_pick_xcd_slices(tm * (tm + 1) / 2, K);
      if ( s < 2 ) s = 2;
      while ( s > 2 && (long long) s * slab > w->kws_len ) --s;
      *sk = s;
      *flags |= HS_GEMM_XCD | HS_GEMM_NOFAST;
      return;
   }
   int s = hs_dgemm_pick_splitk(m1, m1, (int) K, 1);
   while ( s > 1 && (long long) s * slab > w->kws_len ) --s;
   *sk = s < 1 ? 1 : s;
}

/* The same assembly at the cold start X = Z = xi I: M_ij = tr(A_i X A_j Z^-1) = <A_i, A_j>, i.e. the Gram matrix of the rows of A itself - the
 * two n^3 products would multiply by sqrt(xi) I and by I / sqrt(xi).  Mx += A A^T on the lower tiles, no workspace but the slabs. */
int hs_schur_W_identity_range(hipStream_t s, int m1, int n, const double* A, long long k0, long long k1, double* Mx, hs_schur_ws* w)
{
   const long long n2 = (long long) n * n;
   const long long K = k1 - k0;
   if ( n2 > 2000000000LL || k0 < 0 || k1 > n2 || (k0 & 1) )
      return HS_ERR_ARG;
   if ( K <= 0 )
      return HS_OK;
   {
      int done = 0;
      HS_CALL( schur_gram(s, m1, K, A + k0, n2, Mx, w, &done) );
      if ( done )
         return HS_OK;
   }
   int flags, sk;
   schur_syrk_shape(w, m1, K, m1 >= hs_syrk_min_m1() && K >= 16384 && (K >= 50000 || m1 >= 900), &flags, &sk);
   hs_gemm_args g3 = {m1, m1, (int) K, HS_KC, HS_KC, A + k0, n2, 0, A + k0, n2, 0, Mx, m1, 0, 1.0, 1.0, 1, flags, sk, sk > 1 ? w->K : NULL};
   return hs_dgemm(s, &g3);
}

int hs_schur_W_identity(hipStream_t s, int m1, int n, const double* A, double* Mx, hs_schur_ws* w)
{
   return hs_schur_W_identity_range(s, m1, n, A, 0, (long long) n * n, Mx, w);
}

int hs_schur_W(hipStream_t s, int m1, int n, const double* A, const double* R, const double* G, double* Mx, hs_schur_ws* w)
{
   const long long n2 = (long long) n * n;
   const long long rows = (long long) m1 * n;
   if ( !w->full || n2 > 2000000000LL || rows > 2000000000LL )
      return HS_ERR_ARG;
   /* GEMM1: T = A_stack * R, R lower triangular */
   hs_gemm_args g1 = {(int) rows, n, n, HS_KC, HS_MC, A, n, 0, R, n, 0, w->T, n, 0, 1.0, 0.0, 1, HS_GEMM_B_LOWTRI, 1, NULL};   /* measured: the XCD remap costs 45 % on this shape */
   HS_CALL( hs_dgemm(s, &g1) );
   if ( w->after_g1 != NULL )
   {
      int (*f)(void*) = w->after_g1;
      w->after_g1 = NULL;
      HS_CALL( f(w->after_g1_arg) );
   }
   if ( w->ev_g2 != NULL )
      HS_HIP( hipStreamWaitEvent(s, (hipEvent_t) w->ev_g2, 0) );
   /* GEMM2: W_j = G * T_j, G lower triangular */
   hs_gemm_args g2 = {n, n, n, HS_KC, HS_MC, G, n, 0, w->T, n, n2, w->U, n, n2, 1.0, 0.0, m1, HS_GEMM_A_LOWTRI | HS_GEMM_REMAP, 1, NULL};
   HS_CALL( hs_dgemm(s, &g2) );
   /* GEMM3: Mx += W W^T on the lower tiles */
   {
      int done = 0;
      HS_CALL( schur_gram(s, m1, n2, w->U, n2, Mx, w, &done) );
      if ( done )
         return HS_OK;
   }
   int flags = HS_GEMM_LOWER;
   int sk;
   /* measured (tools/syrk_threshold.sh): the XCD-walked slices win by 8-10 % for n >= 256 at every m1 >= 256, and for n = 128 from
    * m1 = 1000 on; below that plain split-K over the lower tiles is 5-40 % faster */
   if ( m1 >= hs_syrk_min_m1() && n2 >= 16384 && (n2 >= 50000 || m1 >= 900) )
   {
      const long long tm = (m1 + 127) / 128;
      const long long ntri = tm * (tm + 1) / 2;
      sk = hs_dgemm_pick_xcd_slices(ntri, n2);
      if ( getenv("HIPSDP_SYRK_SLICES") != NULL && atoi(getenv("HIPSDP_SYRK_SLICES")) >= 2 )
         sk = atoi(getenv("HIPSDP_SYRK_SLICES"));
      while ( sk > 2 && (long long) sk * m1 * m1 > w->kws_len ) --sk;
      if ( getenv("HIPSDP_SYRK_NOXCD") == NULL )
         flags |= HS_GEMM_XCD;
      if ( getenv("HIPSDP_SYRK_FAST") == NULL )
         flags |= HS_GEMM_NOFAST;          /* measured on this shape: the checked loads are 5 % faster (tools/gemm_syrk.cpp) */
   }
   else
   {
      sk = hs_dgemm_pick_splitk(m1, m1, (int) n2, 1);
      while ( sk > 1 && (long long) sk * m1 * m1 > w->kws_len ) --sk;
   }
   hs_gemm_args g3 = {m1, m1, (int) n2, HS_KC, HS_KC, w->U, n2, 0, w->U, n2, 0, Mx, m1, 0, 1.0, 1.0, 1, flags, sk, w->K};
   HS_CALL( hs_dgemm(s, &g3) );
   return HS_OK;
}

/* Column-slice form of the W formulation for sharding over ranks.  Mx = sum over the n^2 entries (r, c) of the W_j of
 * W[:, (r, c)] W[:, (r, c)]^T, so the entries can be dealt out: the rank that owns the columns c in [c0, c0 + cw) forms
 *    T_j[:, c0:c0+cw] = A_j R[:, c0:c0+cw]  (R lower triangular: rows c0.. of R only, the product starts at k = c0),
 *    W_j[:, c0:c0+cw] = G T_j[:, c0:c0+cw]
 * for ALL j and adds W_slice W_slice^T (lower tiles) to Mx: every one of the three products is split 1/ranks, the triangular
 * savings stay, and the partial matrices are summed by one all-reduce.  Slices are stored compactly (n x cw per matrix). */
int hs_schur_Wcols(hipStream_t s, int m1, int n, const double* A, const double* R, const double* G, double* Mx, hs_schur_ws* w,
   int c0, int cw)
{
   if ( cw <= 0 )
      return HS_OK;
   const long long rows = (long long) m1 * n;
   const long long nk = (long long) n * cw;
   if ( c0 < 0 || c0 + cw > n || rows > 2000000000LL || nk > 2000000000LL || (long long) m1 * nk > w->chunk_cols * w->n2 )
      return HS_ERR_ARG;
   /* a slice of at most 64 columns (8 ranks at n = 500: 62) fills half a 128-wide tile: 64 x 64 tiles for the two n^3 products */
   const int narrow = (cw <= 64 && getenv("HIPSDP_NO_TILE64") == NULL) ? HS_GEMM_TILE64 : 0;
   hs_gemm_args g1 = {(int) rows, cw, n - c0, HS_KC, HS_MC, A + c0, n, 0, R + (long long) c0 * n + c0, n, 0, w->T, cw, 0, 1.0, 0.0, 1,
      HS_GEMM_B_LOWTRI | narrow, 1, NULL};
   HS_CALL( hs_dgemm(s, &g1) );
   hs_gemm_args g2 = {n, cw, n, HS_KC, HS_MC, G, n, 0, w->T, cw, nk, w->U, cw, nk, 1.0, 0.0, m1, HS_GEMM_A_LOWTRI | HS_GEMM_REMAP | narrow, 1, NULL};
   HS_CALL( hs_dgemm(s, &g2) );
   {
      int done = 0;
      HS_CALL( schur_gram(s, m1, nk, w->U, nk, Mx, w, &done) );
      if ( done )
         return HS_OK;
   }
   int flags, sk;
   schur_syrk_shape(w, m1, nk, m1 >= 256 && nk >= 16384, &flags, &sk);
   hs_gemm_args g3 = {m1, m1, (int) nk, HS_KC, HS_KC, w->U, nk, 0, w->U, nk, 0, Mx, m1, 0, 1.0, 1.0, 1, flags, sk, sk > 1 ? w->K : NULL};
   HS_CALL( hs_dgemm(s, &g3) );
   return HS_OK;
}

/* columns of rank `rank`: a contiguous range with boundaries at multiples of 16, chosen so that the slowest rank is as fast as
 * possible under this cost model (units of m1 n multiply-adds; the products run in 128-wide tiles, so a slice costs its
 * width rounded up to 128 in the first two products):
 *    wt (2 (n - c0 - w / 2) + n)   [A_j R starts at k = c; G T_j]   +   w m1   [W W^T],      wt = 128 ceil(w / 128)
 * Exact minimisation over all contiguous partitions (dynamic programme over <= n / 16 boundaries).  Ranks may end up without
 * a column when n is small; they contribute a zero matrix to the sum.  Every rank computes the same table. */
/* ---- variable-sharded form: A_j on the rank that owns variable j (n = 4000, m = 8000: 1 TB of A does not fit replicated) ----
 * W_j = G A_j R is formed where A_j lives, but the Gram product W W^T pairs every i with every j.  The sum over the n^2 entries
 * of the W_j can be cut anywhere, so the entries are re-distributed instead of the matrices: an all-to-all sends row range h of
 * every own W_j (current column slice) to rank h, after which rank h holds its n / G rows of ALL m1 matrices and adds its part
 * of the Gram matrix; one all-reduce of Mx closes the assembly.  Per rank and assembly: 1 / G of both n^3 products (triangular
 * savings of R kept, of G between row ranges), 1 / G of the Gram product, (G - 1) / G of m1 n^2 / G doubles sent and received,
 * each over G - 1 links at once. */
void hs_var_rows(int m1, int nranks, int rank, int* r0, int* r1)
{
   const int c = (m1 + nranks - 1) / nranks;
   *r0 = (long long) rank * c < m1 ? rank * c : m1;
   *r1 = *r0 + c < m1 ? *r0 + c : m1;
}

void hs_var_wrows(int n, int nranks, int rank, int* q0, int* q1)
{
   *q0 = (int) ((long long) n * rank / nranks);
   *q1 = (int) ((long long) n * (rank + 1) / nranks);
}

int hs_schur_ws_alloc_var(hs_schur_ws* w, int m1, int nranks, int n, int cwmax)
{
   w->T = w->U = w->K = w->V = w->U2 = w->V2 = NULL;
   w->evP[0] = w->evP[1] = w->evX[0] = w->evX[1] = NULL;
   w->ev_g2 = NULL;
   w->after_g1 = NULL; w->after_g1_arg = NULL;
   w->full = 0; w->chunk_cols = 0; w->n2 = 0;
   const long long cj = (m1 + nranks - 1) / nranks;
   long long rowsmax = 1;
   for (int h = 0; h < nranks; ++h)
   {
      int q0, q1;
      hs_var_wrows(n, nranks, h, &q0, &q1);
      if ( q1 - q0 > rowsmax ) rowsmax = q1 - q0;
   }
   w->capT = cj * (long long) n * cwmax;
   w->capV = (long long) m1 * rowsmax * cwmax;
   const long long nkV = rowsmax * cwmax;
   int sk = hs_dgemm_pick_splitk(m1, m1, (int) (nkV > 2000000000LL ? 2000000000LL : nkV), 1);
   if ( sk < 16 ) sk = 16;
   const long long tm = (m1 + 127) / 128;
   const int sx = hs_dgemm_pick_xcd_slices(tm * (tm + 1) / 2, nkV);
   if ( sx > sk ) sk = sx;
   if ( sk < 64 ) sk = 64;
   while ( sk > 2 && 8.0 * (double) sk * (double) m1 * (double) m1 > 8e9 ) --sk;       /* large m: the tiles alone fill the chip */
   w->kws_len = (long long) sk * m1 * m1;
   if ( hs_pool_alloc((void**) &w->T, (size_t) w->capT * sizeof(double)) != HS_OK
      || hs_pool_alloc((void**) &w->U, (size_t) w->capT * sizeof(double)) != HS_OK
      || hs_pool_alloc((void**) &w->V, (size_t) w->capV * sizeof(double)) != HS_OK
      || hs_pool_alloc((void**) &w->K, (size_t) w->kws_len * sizeof(double)) != HS_OK )
   {
      hs_record_hip_error(hipErrorOutOfMemory, "hipMalloc(variable-sharded schur workspace)", __FILE__, __LINE__);
      return HS_ERR_NOMEM;
   }
   return HS_OK;
}

int hs_schur_ws_alloc_var_overlap(hs_schur_ws* w)
{
   if ( w->U == NULL || w->V == NULL )
      return HS_ERR_ARG;
   if ( hs_pool_alloc((void**) &w->U2, (size_t) w->capT * sizeof(double)) != HS_OK
      || hs_pool_alloc((void**) &w->V2, (size_t) w->capV * sizeof(double)) != HS_OK )
   {
      hs_pool_free(w->U2); hs_pool_free(w->V2);
      w->U2 = w->V2 = NULL;
      return HS_ERR_NOMEM;
   }
   for (int i = 0; i < 2; ++i)
   {
      hipEvent_t e = NULL;
      HS_HIP( hipEventCreateWithFlags(&e, hipEventDisableTiming) );
      w->evP[i] = (void*) e;
      HS_HIP( hipEventCreateWithFlags(&e, hipEventDisableTiming) );
      w->evX[i] = (void*) e;
   }
   return HS_OK;
}

namespace {

/* the three stages of one column slice [c0, c0 + cw) of the variable-sharded assembly */
struct WvarSlice
{
   int rank, nranks, m1, n, c0, cw, r0, r1, cj;
   long long n2, nk, Kme;
   long long cnt[64 * 64];
};

int wvar_setup(WvarSlice& q, const hs_schur_ws* w, int rank, int nranks, int m1, int n, int c0, int cw)
{
   if ( nranks < 1 || nranks > 64 || w->V == NULL || c0 < 0 || c0 + cw > n )
      return HS_ERR_ARG;
   q.rank = rank; q.nranks = nranks; q.m1 = m1; q.n = n; q.c0 = c0; q.cw = cw;
   q.n2 = (long long) n * n;
   q.nk = (long long) n * cw;
   int p0, p1;
   hs_var_rows(m1, nranks, rank, &q.r0, &q.r1);
   hs_var_wrows(n, nranks, rank, &p0, &p1);
   q.cj = q.r1 - q.r0;
   q.Kme = (long long) (p1 - p0) * cw;
   if ( (long long) q.cj * q.nk > w->capT || (long long) m1 * q.Kme > w->capV || (long long) q.cj * n > 2000000000LL || q.nk > 2000000000LL
      || q.Kme > 2000000000LL )
      return HS_ERR_ARG;
   for (int src = 0; src < nranks; ++src)
   {
      int a0, a1;
      hs_var_rows(m1, nranks, src, &a0, &a1);
      for (int dst = 0; dst < nranks; ++dst)
      {
         int q0, q1;
         hs_var_wrows(n, nranks, dst, &q0, &q1);
         q.cnt[src * nranks + dst] = (long long) (a1 - a0) * (q1 - q0) * cw;
      }
   }
   return HS_OK;
}

/* T_j[:, slice] = A_j R, then W_j[rows of rank h, slice] = G T_j into the send buffer, piece by piece */
int wvar_products(hipStream_t s, const WvarSlice& q, const double* A, const double* R, const double* G, hs_schur_ws* w, double* send)
{
   const int n = q.n, cw = q.cw, c0 = q.c0, cj = q.cj;
   if ( cj <= 0 )
      return HS_OK;
   hs_gemm_args g1 = {cj * n, cw, n - c0, HS_KC, HS_MC, A + (long long) q.r0 * q.n2 + c0, n, 0, R + (long long) c0 * n + c0, n, 0, w->T, cw, 0,
      1.0, 0.0, 1, HS_GEMM_B_LOWTRI, 1, NULL};
   HS_CALL( hs_dgemm(s, &g1) );
   for (int h = 0; h < q.nranks; ++h)
   {
      int q0, q1;
      hs_var_wrows(n, q.nranks, h, &q0, &q1);
      if ( q1 <= q0 )
         continue;
      const long long piece = (long long) (q1 - q0) * cw;
      hs_gemm_args g2 = {q1 - q0, cw, q1, HS_KC, HS_MC, G + (long long) q0 * n, n, 0, w->T, cw, q.nk, send + (long long) cj * q0 * cw, cw, piece,
         1.0, 0.0, cj, HS_GEMM_REMAP, 1, NULL};
      HS_CALL( hs_dgemm(s, &g2) );
   }
   return HS_OK;
}

/* Mx += V V^T over this rank's rows of all W_j (lower tiles) */
int wvar_gram(hipStream_t s, const WvarSlice& q, hs_schur_ws* w, const double* recv, double* Mx)
{
   const int m1 = q.m1;
   const long long Kme = q.Kme;
   if ( Kme <= 0 )
      return HS_OK;
   {
      int done = 0;
      HS_CALL( schur_gram(s, m1, Kme, recv, Kme, Mx, w, &done) );
      if ( done )
         return HS_OK;
   }
   int flags, sk;
   schur_syrk_shape(w, m1, Kme, m1 >= 256 && Kme >= 16384, &flags, &sk);
   hs_gemm_args g3 = {m1, m1, (int) Kme, HS_KC, HS_KC, recv, Kme, 0, recv, Kme, 0, Mx, m1, 0, 1.0, 1.0, 1, flags, sk, sk > 1 ? w->K : NULL};
   return hs_dgemm(s, &g3);
}

}

int hs_schur_Wvar(hipStream_t s, void* comm, int rank, int nranks, int m1, int n, const double* A, const double* R, const double* G,
   double* Mx, hs_schur_ws* w, int c0, int cw)
{
   if ( cw <= 0 )
      return HS_OK;
   WvarSlice q;
   HS_CALL( wvar_setup(q, w, rank, nranks, m1, n, c0, cw) );
   HS_CALL( wvar_products(s, q, A, R, G, w, w->U) );
   HS_CALL( hs_alltoall(comm, w->U, w->V, q.cnt, s) );
   return wvar_gram(s, q, w, w->V, Mx);
}

int hs_schur_Wvar_all(hipStream_t s, hipStream_t sc, void* comm, int rank, int nranks, int m1, int n, const double* A, const double* R,
   const double* G, double* Mx, hs_schur_ws* w, int cwmax, int overlap)
{
   if ( cwmax <= 0 )
      return HS_ERR_ARG;
   const int nslices = (n + cwmax - 1) / cwmax;
   if ( !overlap || nslices < 2 || w->U2 == NULL || w->V2 == NULL || sc == NULL || w->evP[0] == NULL )
   {
      for (int c0 = 0; c0 < n; c0 += cwmax)
         HS_CALL( hs_schur_Wvar(s, comm, rank, nranks, m1, n, A, R, G, Mx, w, c0, n - c0 < cwmax ? n - c0 : cwmax) );
      return HS_OK;
   }
   /* two-slice pipeline.  Compute queue s: products(0), [products(t + 1), wait exchange(t), Gram(t)] ...; communication queue sc:
    * [wait products(t), all-to-all(t)] ...  Buffers alternate with the slice parity; the queue orders make every reuse safe: the
    * send buffer of slice t + 2 is written after Gram(t) was enqueued, which waited for exchange(t); the receive buffer of slice
    * t + 2 is written by an exchange that waits for products(t + 2), enqueued behind Gram(t) which read it.  The Gram updates hit Mx
    * in slice order, as in the in-order form: same bits. */
   static thread_local WvarSlice tq[2];        /* host bookkeeping of the two slices in flight */
   double* sendb[2] = {w->U, w->U2};
   double* recvb[2] = {w->V, w->V2};
   for (int t = 0; t < nslices; ++t)
   {
      const int b = t & 1;
      const int c0 = t * cwmax, cw = n - c0 < cwmax ? n - c0 : cwmax;
      HS_CALL( wvar_setup(tq[b], w, rank, nranks, m1, n, c0, cw) );
      HS_CALL( wvar_products(s, tq[b], A, R, G, w, sendb[b]) );
      HS_HIP( hipEventRecord((hipEvent_t) w->evP[b], s) );
      HS_HIP( hipStreamWaitEvent(sc, (hipEvent_t) w->evP[b], 0) );
      HS_CALL( hs_alltoall(comm, sendb[b], recvb[b], tq[b].cnt, sc) );
      HS_HIP( hipEventRecord((hipEvent_t) w->evX[b], sc) );
      if ( t >= 1 )
      {
         HS_HIP( hipStreamWaitEvent(s, (hipEvent_t) w->evX[b ^ 1], 0) );
         HS_CALL( wvar_gram(s, tq[b ^ 1], w, recvb[b ^ 1], Mx) );
      }
   }
   const int bl = (nslices - 1) & 1;
   HS_HIP( hipStreamWaitEvent(s, (hipEvent_t) w->evX[bl], 0) );
   return wvar_gram(s, tq[bl], w, recvb[bl], Mx);
}

void hs_shard_cols(int m1, int n, int nranks, int rank, int* c_begin, int* c_width)
{
   static thread_local int memo_key[3] = {-1, -1, -1};
   static thread_local int memo_bounds[65];
   if ( nranks > 64 ) nranks = 64;
   if ( memo_key[0] != m1 || memo_key[1] != n || memo_key[2] != nranks )
   {
      const int gran = 16;
      const int P = (n + gran - 1) / gran;                 /* boundary p stands for column min(p * gran, n) */
      auto col = [&](int p) -> int { return p * gran < n ? p * gran : n; };
      auto cost = [&](int p0, int p1) -> double {
         const int c0 = col(p0), w = col(p1) - c0;
         if ( w <= 0 ) return 0.0;
         const double wt = w <= 64 ? 64.0 : 128.0 * (double) ((w + 127) / 128);      /* slices of <= 64 columns run on 64-wide tiles */
         return wt * (2.0 * ((double) n - (double) c0 - 0.5 * (double) w) + (double) n) + (double) w * (double) m1;
      };
      /* best[g][p]: smallest possible maximum over the first g ranks covering boundaries 0 .. p */
      std::vector<double> prev(P + 1, 1e300), cur(P + 1);
      std::vector<std::vector<int> > from(nranks + 1, std::vector<int>(P + 1, 0));
      prev[0] = 0.0;
      for (int g = 1; g <= nranks; ++g)
      {
         for (int p = 0; p <= P; ++p)
         {
            double b = 1e300; int arg = 0;
            for (int q = 0; q <= p; ++q)
            {
               if ( prev[q] >= 1e300 ) continue;
               const double v = fmax(prev[q], cost(q, p));
               if ( v < b ) { b = v; arg = q; }
            }
            cur[p] = b; from[g][p] = arg;
         }
         prev = cur;
      }
      int p = P;
      memo_bounds[nranks] = n;
      for (int g = nranks; g >= 1; --g)
      {
         p = from[g][p];
         memo_bounds[g - 1] = col(p);
      }
      memo_key[0] = m1; memo_key[1] = n; memo_key[2] = nranks;
   }
   if ( rank < 0 ) rank = 0;
   if ( rank >= nranks ) rank = nranks - 1;
   *c_begin = memo_bounds[rank];
   *c_width = memo_bounds[rank + 1] - memo_bounds[rank];
}

/* Row-block form of the U formulation for sharding over ranks:  Mx[r, c] for r in [r_begin, r_end), c >= r (upper triangle of
 * those rows), stored as contiguous rows of Mx so that the row blocks of all ranks can be all-gathered.
 *    Mx[r0:r1, r0:] += U_chunk [cj x n^2] * A_flat[r0:]^T */
int hs_schur_Urows(hipStream_t s, int m1, int n, const double* A, const double* X, const double* Zinv, double* Mx,
   hs_schur_ws* w, int r_begin, int r_end)
{
   const long long n2 = (long long) n * n;
   if ( n2 > 2000000000LL )
      return HS_ERR_ARG;
   if ( r_end > m1 ) r_end = m1;
   for (int r0 = r_begin; r0 < r_end; r0 += (int) w->chunk_cols)
   {
      const int cj = (r_end - r0) < w->chunk_cols ? (r_end - r0) : (int) w->chunk_cols;
      const long long rows = (long long) cj * n;
      if ( rows > 2000000000LL )
         return HS_ERR_ARG;
      hs_gemm_args g1 = {(int) rows, n, n, HS_KC, HS_MC, A + (long long) r0 * n2, n, 0, Zinv, n, 0, w->T, n, 0, 1.0, 0.0, 1, HS_GEMM_REMAP, 1, NULL};
      HS_CALL( hs_dgemm(s, &g1) );
      hs_gemm_args g2 = {n, n, n, HS_KC, HS_MC, X, n, 0, w->T, n, n2, w->U, n, n2, 1.0, 0.0, cj, HS_GEMM_REMAP, 1, NULL};
      HS_CALL( hs_dgemm(s, &g2) );
      const int colsM = m1 - r0;
      int sk = hs_dgemm_pick_splitk(cj, colsM, (int) n2, 0);
      while ( sk > 1 && (long long) sk * cj * colsM > w->kws_len ) --sk;
      hs_gemm_args g3 = {cj, colsM, (int) n2, HS_KC, HS_KC, w->U, n2, 0, A + (long long) r0 * n2, n2, 0,
         Mx + (long long) r0 * m1 + r0, m1, 0, 1.0, 1.0, 1, HS_GEMM_UPPER, sk, w->K};
      HS_CALL( hs_dgemm(s, &g3) );
   }
   return HS_OK;
}

/* balanced two-chunk assignment of the rows of the (upper) triangle: 2 G chunks of c = ceil(m1 / (2 G)) rows; rank g owns
 * chunk g and chunk 2 G - 1 - g, whose triangle areas add up to the same total for every g */
void hs_shard_rows(int m1, int nranks, int rank, int* chunk_rows, int* first_begin, int* second_begin)
{
   const int c = (m1 + 2 * nranks - 1) / (2 * nranks);
   *chunk_rows = c;
   *first_begin = rank * c;
   *second_begin = (2 * nranks - 1 - rank) * c;
}


/* ---- small problems: the whole extended Schur matrix in ONE launch ------------------------------------------------- */
/* All blocks n <= HS_SMALL_N: workgroup j forms U_j^k = X_k A_j^k Zinv_k in LDS for every block, then the entries (i, j), i >= j, of
 *    Mx = sum_k <A_i^k, U_j^k> + Dext^T diag(x / z) Dext
 * and stores them symmetrically, together with the copies the factorization wants (Lm = Mx[1:, 1:], its diagonal).
 * Replaces fill + 3 GEMM launches per block + row scaling + GEMM + mirror + 2 strided copies: B&B-sized problems are bound
 * by the launch count.  Same sums as the general path up to the order of the additions. */
#define SS_MAXBLK 8
#define SS_MAXM1  1024
struct ss_args
{
   int nblk;
   int n[SS_MAXBLK];
   const double* A[SS_MAXBLK];
   const double* X[SS_MAXBLK];
   const double* Zinv[SS_MAXBLK];
};

__global__ void __launch_bounds__(256) k_schur_small(int m1, int nmax, ss_args B, int q, const double* __restrict__ Dext,
   const double* __restrict__ x, const double* __restrict__ z, double* __restrict__ Mx, double* __restrict__ Lm,
   double* __restrict__ diagM)
{
   extern __shared__ double ss_smem[];
   __shared__ double colv[SS_MAXM1];
   const int ldm = nmax + 1;
   double* sx = ss_smem;
   double* sz = sx + nmax * ldm;
   double* sa = sz + nmax * ldm;
   double* st = sa + nmax * ldm;
   double* su = st + nmax * ldm;
   const int j = blockIdx.x;
   const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
   for (int i = tid; i < m1; i += 256)
      colv[i] = 0.0;
   __syncthreads();
   for (int k = 0; k < B.nblk; ++k)
   {
      const int n = B.n[k];
      const int n2 = n * n;
      const double* Aj = B.A[k] + (long long) j * n2;
      for (int e = tid; e < n2; e += 256)
      {
         const int r = e / n, c = e - r * n;
         sx[r * ldm + c] = B.X[k][e];
         sz[r * ldm + c] = B.Zinv[k][e];
         sa[r * ldm + c] = Aj[e];
      }
      __syncthreads();
      /* U_j = (X A_j) Zinv: 2 x 5 patches per thread (hs_lds_product.h; entry by entry the two products were most of the 67 us
       * of this kernel at n = 43) */
      db_product(n, ldm, sx, sa, [&](int, int, int r, int c, double acc) { st[r * ldm + c] = acc; });
      __syncthreads();
      db_product(n, ldm, st, sz, [&](int, int, int r, int c, double acc) { su[r * ldm + c] = acc; });
      __syncthreads();
      /* <A_i, U_j> for i >= j: one wavefront per i */
      for (int i = j + wave; i < m1; i += 4)
      {
         const double* Ai = B.A[k] + (long long) i * n2;
         double acc = 0.0;
         for (int e = lane; e < n2; e += 64)
         {
            const int r = e / n, c = e - r * n;
            acc += Ai[e] * su[r * ldm + c];
         }
#pragma unroll
         for (int off = 32; off > 0; off >>= 1)
            acc += __shfl_down(acc, off, 64);
         if ( lane == 0 )
            colv[i] += acc;
      }
      __syncthreads();
   }
   /* LP part: sum_r Dext[r][i] (x_r / z_r) Dext[r][j] */
   if ( q > 0 )
   {
      for (int i = j + tid; i < m1; i += 256)
      {
         double acc = 0.0;
         for (int r = 0; r < q; ++r)
            acc += Dext[(long long) r * m1 + i] * ((x[r] / z[r]) * Dext[(long long) r * m1 + j]);
         colv[i] += acc;
      }
      __syncthreads();
   }
   const int m = m1 - 1;
   for (int i = j + tid; i < m1; i += 256)
   {
      const double v = colv[i];
      Mx[(long long) i * m1 + j] = v;
      Mx[(long long) j * m1 + i] = v;
      if ( Lm != NULL && j >= 1 )
      {
         Lm[(long long) (i - 1) * m + (j - 1)] = v;
         Lm[(long long) (j - 1) * m + (i - 1)] = v;
         if ( i == j )
            diagM[j - 1] = v;
      }
   }
}

/* 1: assembled; 0: not applicable (sizes), caller takes the general path; < 0: error */
int hs_schur_small(hipStream_t s, int m1, int nblk, const int* n, const double* const* A, const double* const* X,
   const double* const* Zinv, int q, const double* Dext, const double* x, const double* z, double* Mx, double* Lm, double* diagM)
{
   if ( nblk > SS_MAXBLK || m1 > SS_MAXM1 || m1 < 1 || (long long) q * m1 > 2000000LL )
      return 0;
   {
      /* the one-launch form does the m1^2 / 2 inner products <A_i, U_j> with scalar arithmetic out of L2: it wins while the launch
       * count decides (few variables) and loses by 20x at m = 1000 (n = 48: 4.4 ms against 0.2 ms of the GEMM path).  The bound is
       * on the multiply-adds of those products; measured crossover of the whole iteration (tools/grid_sweep.sh, round 3): n = 16
       * between m = 100 and 200, n = 32 between 50 and 100, n = 48 at about 20, i.e. 1 - 5e6 (HIPSDP_SCHUR_SMALL_MAXWORK) */
      static double maxwork = -1.0;
      if ( maxwork < 0.0 )
      {
         const char* env = getenv("HIPSDP_SCHUR_SMALL_MAXWORK");
         maxwork = env != NULL ? atof(env) : 1.5e6;
      }
      double work = 0.0;
      for (int k = 0; k < nblk; ++k)
         work += 0.5 * (double) m1 * m1 * (double) n[k] * n[k];
      if ( work > maxwork )
         return 0;
   }
   ss_args B;
   B.nblk = nblk;
   int nmax = 1;
   for (int k = 0; k < nblk; ++k)
   {
      if ( n[k] > HS_SMALL_N || n[k] < 1 )
         return 0;
      if ( n[k] > nmax )
         nmax = n[k];
      B.n[k] = n[k]; B.A[k] = A[k]; B.X[k] = X[k]; B.Zinv[k] = Zinv[k];
   }
   for (int k = nblk; k < SS_MAXBLK; ++k)
   {
      B.n[k] = 0; B.A[k] = NULL; B.X[k] = NULL; B.Zinv[k] = NULL;
   }
   static hs_attr_mask attr_done;
   if ( hs_func_max_lds(reinterpret_cast<const void*>(&k_schur_small), 5 * HS_SMALL_N * (HS_SMALL_N + 1) * (int) sizeof(double), &attr_done) != HS_OK )
      return -HS_ERR_HIP;
   hipLaunchKernelGGL(k_schur_small, dim3(m1), dim3(256), (size_t) 5 * nmax * (nmax + 1) * sizeof(double), s, m1, nmax, B, q, Dext, x, z,
      Mx, Lm, diagM);
   if ( hipGetLastError() != hipSuccess )
      return -HS_ERR_HIP;
   return 1;
}
