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
   /* batched in the transposed formulation: C_j = T_j^T * S^T  (A = T_j in MC layout, B = S as [N][K]) */
   for (int f : {0, HS_GEMM_REMAP})
   {
      hs_gemm_args g = {n, n, n, HS_MC, HS_KC, T, n, n2, S, n, 0, W, n, n2, 1.0, 0.0, m1, f, 1, NULL};
      printf("batchedT flags %2d : %.3f ms\n", f, timeit(g, 5));
   }
   return 0;
}
