/* gemm_shapes.cpp - developer tool: raw rates of the Schur GEMM shapes with/without the triangular and remap flags */
#include "../scip-sdp_amd/csrc/hs_common.h"
#include <vector>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while(0)
static double timeit(hs_gemm_args g, int reps)
{
   hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
   for (int w = 0; w < 2; ++w) hs_dgemm(0, &g);
   CK(hipEventRecord(e0, 0));
   for (int r = 0; r < reps; ++r) hs_dgemm(0, &g);
   CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
   float ms; CK(hipEventElapsedTime(&ms, e0, e1));
   return ms / reps;
}
int main()
{
   const int n = 500, m1 = 1001;
   const long long n2 = (long long) n * n;
   double *A, *T, *W, *S;
   CK(hipMalloc(&A, m1 * n2 * 8)); CK(hipMalloc(&T, m1 * n2 * 8)); CK(hipMalloc(&W, m1 * n2 * 8)); CK(hipMalloc(&S, n2 * 8));
   std::vector<double> h(n2); for (auto& x : h) x = (double) rand() / RAND_MAX - 0.5;
   CK(hipMemcpy(S, h.data(), n2 * 8, hipMemcpyHostToDevice));
   for (int j = 0; j < m1; ++j) CK(hipMemcpy(A + j * n2, h.data(), n2 * 8, hipMemcpyHostToDevice));
   CK(hipMemcpy(T, A, m1 * n2 * 8, hipMemcpyDeviceToDevice));
   int fl[] = {0, HS_GEMM_REMAP, HS_GEMM_B_LOWTRI, HS_GEMM_B_LOWTRI | HS_GEMM_REMAP};
   for (int f : fl)
   {
      hs_gemm_args g = {m1 * n, n, n, HS_KC, HS_MC, A, n, 0, S, n, 0, T, n, 0, 1.0, 0.0, 1, f, 1, NULL};
      printf("stack   flags %7d : %.3f ms\n", f, timeit(g, 5));
   }
   int fa[] = {0, HS_GEMM_REMAP, HS_GEMM_A_LOWTRI, HS_GEMM_A_LOWTRI | HS_GEMM_REMAP};
   for (int f : fa)
   {
      hs_gemm_args g = {n, n, n, HS_KC, HS_MC, S, n, 0, T, n, n2, W, n, n2, 1.0, 0.0, m1, f, 1, NULL};
      printf("batched flags %7d : %.3f ms\n", f, timeit(g, 5));
   }
   /* the same two products with B K-contiguous (B given as [N][K]): the <0> instance of the persistent kernel */
   {
      hs_gemm_args g = {m1 * n, n, n, HS_KC, HS_KC, A, n, 0, S, n, 0, T, n, 0, 1.0, 0.0, 1, 0, 1, NULL};
      const double ms = timeit(g, 5);
      printf("stack   B as [N][K]   : %.3f ms  (%.1f TFLOP/s full)\n", ms, 2.0 * m1 * n * (double) n * n / ms * 1e-9);
      hs_gemm_args g1 = {m1 * n, n, n, HS_KC, HS_MC, A, n, 0, S, n, 0, T, n, 0, 1.0, 0.0, 1, 0, 1, NULL};
      const double ms1 = timeit(g1, 5);
      printf("stack   B as [K][N]   : %.3f ms  (%.1f TFLOP/s full)\n", ms1, 2.0 * m1 * n * (double) n * n / ms1 * 1e-9);
      hs_gemm_args g2 = {n, n, n, HS_KC, HS_KC, S, n, 0, T, n, n2, W, n, n2, 1.0, 0.0, m1, 0, 1, NULL};
      const double ms2 = timeit(g2, 5);
      printf("batched B as [N][K]   : %.3f ms  (%.1f TFLOP/s full)\n", ms2, 2.0 * m1 * n * (double) n * n / ms2 * 1e-9);
      hs_gemm_args g3 = {n, n, n, HS_KC, HS_MC, S, n, 0, T, n, n2, W, n, n2, 1.0, 0.0, m1, 0, 1, NULL};
      const double ms3 = timeit(g3, 5);
      printf("batched B as [K][N]   : %.3f ms  (%.1f TFLOP/s full)\n", ms3, 2.0 * m1 * n * (double) n * n / ms3 * 1e-9);
      /* K four times as long with the same output: how much of the gap is the short K loop */
      const int n4 = 2000;
      double* A4; double* S4;
      CK(hipMalloc(&A4, (size_t) 125000 * n4 * 8)); CK(hipMalloc(&S4, (size_t) n4 * n * 8));
      CK(hipMemset(A4, 0, (size_t) 125000 * n4 * 8)); CK(hipMemset(S4, 0, (size_t) n4 * n * 8));
      hs_gemm_args g4 = {125000, n, n4, HS_KC, HS_MC, A4, n4, 0, S4, n, 0, T, n, 0, 1.0, 0.0, 1, 0, 1, NULL};
      const double ms4 = timeit(g4, 5);
      printf("125000 x 500 x 2000, B as [K][N] : %.3f ms  (%.1f TFLOP/s)\n", ms4, 2.0 * 125000.0 * n * n4 / ms4 * 1e-9);
      hs_gemm_args g5 = {125000, n, n4, HS_KC, HS_KC, A4, n4, 0, S4, n4, 0, T, n, 0, 1.0, 0.0, 1, 0, 1, NULL};
      const double ms5 = timeit(g5, 5);
      printf("125000 x 500 x 2000, B as [N][K] : %.3f ms  (%.1f TFLOP/s)\n", ms5, 2.0 * 125000.0 * n * n4 / ms5 * 1e-9);
   }
   /* which property of the Schur shape costs the 10 % against a big square: the ragged fourth column tile, or four column tiles
    * per row panel (reuse of the streamed operand in L2)? */
   {
      struct { int M, N, K; } sh[] = {{500500, 500, 500}, {488704, 512, 512}, {244352, 1024, 512}, {122176, 2048, 512}, {61056, 4096, 512},
         {8192, 4096, 4096}};
      double* Bb; CK(hipMalloc(&Bb, (size_t) 4096 * 4096 * 8)); CK(hipMemset(Bb, 0, (size_t) 4096 * 4096 * 8));
      double* Cb; CK(hipMalloc(&Cb, (size_t) 500500 * 512 * 8));
      for (auto& q : sh)
      {
         hs_gemm_args g = {q.M, q.N, q.K, HS_KC, HS_MC, A, q.K, 0, Bb, q.N, 0, Cb, q.N, 0, 1.0, 0.0, 1, 0, 1, NULL};
         const double ms = timeit(g, 5);
         printf("plain %7d x %4d x %4d : %.3f ms  %.1f TFLOP/s\n", q.M, q.N, q.K, ms, 2.0 * q.M * (double) q.N * q.K / ms * 1e-9);
      }
   }
   /* batched in the transposed formulation: C_j = T_j^T * S^T  (A = T_j in MC layout, B = S as [N][K]) */
   for (int f : {0, HS_GEMM_REMAP})
   {
      hs_gemm_args g = {n, n, n, HS_MC, HS_KC, T, n, n2, S, n, 0, W, n, n2, 1.0, 0.0, m1, f, 1, NULL};
      printf("batchedT flags %2d : %.3f ms\n", f, timeit(g, 5));
   }
   return 0;
}
