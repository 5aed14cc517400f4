/* hipsdp_units.h - TEST / BENCH entry points around single device kernels (csrc/units.hip), exported by lib/libhipsdp_units.so - a
 * library of its own that holds the engine's objects plus these entries.  The product library libhipsdp.so does not contain them
 * (include/hipsdp.h is the product's C ABI).  Used by tests/, tests/devtools/ and bench.py (measured matrix peak) only. */
#ifndef HIPSDP_UNITS_H
#define HIPSDP_UNITS_H

#include "hipsdp.h"

#ifdef __cplusplus
extern "C" {
#endif

/* both GEMM kernels (one tile per workgroup; persistent with LDS-DMA staging) on the same device-generated operands:
 * used_v2 = 1 when the persistent kernel accepts the shape, ndiff = elements of C that differ in any bit (must be 0) */
HIPSDP_API int  hipsdp_dgemm_selfcheck(int device, int M, int N, int K, int layB, int batch, int splitk, int flags, double beta,
   int* used_v2, long long* ndiff);
/* the same with a free alpha and, for reps > 0 and beta = 0, the average milliseconds of one product through the tile kernel alone
 * (ms_tile) and through the default dispatch (ms_fast).  *used: bit 0 the persistent tile kernel took the product, bit 1 the strip
 * kernel of the two triangular Schur products (alpha = 1, beta = 0 only) */
/* unit entry: out[e] = sum_i coef[i] A[i][e] + sa add[e] over R rows of E entries (the pass A^T); split = 1: as the engine calls it
 * (row chunks side by side when the block has few entries; *chunks = how many, 0 = the plain kernel) */
HIPSDP_API int  hipsdp_pass_at_unit(int device, int R, long long E, const double* A, const double* coef, double sa, const double* add, int split,
   double* out, int* chunks);
HIPSDP_API int  hipsdp_dgemm_selfcheck2(int device, int M, int N, int K, int layB, int batch, int splitk, int flags, double alpha, double beta,
   int reps, int* used, long long* ndiff, double* ms_tile, double* ms_fast);
HIPSDP_API int  hipsdp_gram_plan_info(int device, int M, long long K, int nslab, int* no, int* nd, int* nitems, double* span);
/* Gram product W W^T (lower triangle) through the K-sliced tile kernels and through the Gram kernel of csrc/gram.hip */
HIPSDP_API int  hipsdp_gram_selfcheck(int device, int M, long long K, int reps, int* used, double* maxdiff, long long* nrepro, double* ms_tile,
   double* ms_gram);
/* the same, also returning the largest absolute difference of the two results and the number of elements that differ between
 * TWO runs of the default dispatch (must be 0) */
HIPSDP_API int  hipsdp_dgemm_selfcheck3(int device, int M, int N, int K, int layB, int batch, int splitk, int flags, double alpha, double beta,
   int reps, int* used, long long* ndiff, double* maxdiff, long long* nrepro, double* ms_tile, double* ms_fast);
/* Schur block Mx[(m1) x (m1)] = tr(A_i X A_j Zinv) for i, j = 0..m1-1 */
HIPSDP_API int  hipsdp_schur_dense(int device, int m1, int n, const double* A, const double* X, const double* Zinv, double* Mx,
   double ws_gbytes);
/* the same matrix through the W formulation (W_j = G A_j R, Mx = W W^T); takes X and Z, factors them on the device */
HIPSDP_API int  hipsdp_schur_w(int device, int m1, int n, const double* A, const double* X, const double* Z, double* Mx);
/* milliseconds one rank of an nranks-way sharded assembly spends on its share of the Schur matrix (by_columns: column slices
 * of the W formulation, else row chunks of the U formulation); synthetic operands made in HBM */
HIPSDP_API int  hipsdp_schur_shard_time(int device, int m1, int n, int nranks, int rank, int by_columns, int reps, double ws_gbytes, double* ms);
/* the same for one rank of the variable-sharded assembly (hipsdp_shard_matrices): only that rank's rows of A are allocated, the
 * column slices are cw wide, the all-to-all keeps the rank's own piece; *a2a_bytes = bytes the rank would send per assembly */
HIPSDP_API int  hipsdp_schur_var_share_time(int device, int m1, int n, int nranks, int rank, int cw, int reps, double* ms, double* a2a_bytes);
/* sparse block mode: Schur entries of matrices given as triplets (var 1 .. m, row >= col) exactly as the engine assembles them
 * (csrc/sparse.hip); Mx (m + 1) x (m + 1), lower triangle of rows / columns 1 .. m */
HIPSDP_API int  hipsdp_schur_sparse_unit(int device, int n, int m, long long nnz, const int* var, const int* row, const int* col,
   const double* val, const double* X, const double* Zinv, double* Mx);
/* measured FP64 matrix peak of the device: a chip-filling launch of register-only v_mfma_f64_16x16x4_f64 for about ms milliseconds;
 * *tflops by HIP events, *ghz = shader clocks per wall tick inside the kernel (bench.py prices its roofline against this as well) */
HIPSDP_API int  hipsdp_mfma_peak(int device, double ms, double* tflops, double* ghz);
HIPSDP_API int  hipsdp_potrf(int device, int n, double* A, int* fail);                       /* lower Cholesky in place, row-major */
/* both forms of the blocked factorization for the parity tests: v1 = 1 the four-launch form, 0 one launch per block column; psd = 1
 * semidefinite pivot rule with diag0 = diag(A), forced pivots in regmask[n]; dinv[ceil(n / 64) * 4096] (any output may be NULL) */
HIPSDP_API int  hipsdp_potrf_ex(int device, int n, double* A, int psd, int v1, double* dinv, int* regmask, int* fail);
HIPSDP_API int  hipsdp_potrs(int device, int n, const double* A, int nrhs, double* rhs);     /* factor + solve, rhs[k * n + i] */
HIPSDP_API int  hipsdp_trtri(int device, int n, const double* A, double* Linv);              /* A spd -> inverse of its Cholesky factor */
HIPSDP_API int  hipsdp_lambda_min(int device, int n, const double* W, int steps, double* theta, double* resid);
/* lambda_min(L D L^T), n <= 64, L lower triangular, D symmetric: the small-block step-length kernels; theta[2], resid[2] */
HIPSDP_API int  hipsdp_lambda_min_scaled(int device, int n, const double* L, const double* D, int steps, double* theta, double* resid);

#ifdef __cplusplus
}
#endif

#endif
