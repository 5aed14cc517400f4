/* gemm_check.cpp - developer tool: checks hs_dgemm against a host triple loop and times it beside rocBLAS dgemm.
 * Not part of the product; built by tools/Makefile on demand. */
#include "../scip-sdp_amd/csrc/hs_common.h"
#include <rocblas/rocblas.h>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cmath>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while(0)

static double urand() { return (double) rand() / RAND_MAX * 2.0 - 1.0; }

static int check(int M, int N, int K, int layA, int layB, int batch, int splitk, int flags)
{
   long long lda = (layA == HS_KC ? K : M) + 1;   /* odd-ish leading dims on purpose */
   long long ldb = (layB == HS_KC ? K : N) + 3;
   long long ldc = N + 1;
   long long sA = (layA == HS_KC ? M : K) * lda, sB = (layB == HS_KC ? N : K) * ldb, sC = (long long) M * ldc;
   std::vector<double> A(sA * batch), B(sB * batch), C(sC * batch), R(sC * batch);
   for (auto& x : A) x = urand();
   for (auto& x : B) x = urand();
   for (auto& x : C) x = urand();
   R = C;
   const double alpha = 0.75, beta = -0.5;
   for (int b = 0; b < batch; ++b)
      for (int i = 0; i < M; ++i)
         for (int j = 0; j < N; ++j)
         {
            if ( (flags & HS_GEMM_LOWER) && i < j ) continue;
            double s = 0;
            for (int k = 0; k < K; ++k)
            {
               double a = layA == HS_KC ? A[b * sA + i * lda + k] : A[b * sA + k * lda + i];
               double bb = layB == HS_KC ? B[b * sB + j * ldb + k] : B[b * sB + k * ldb + j];
               s += a * bb;
            }
            R[b * sC + i * ldc + j] = alpha * s + beta * R[b * sC + i * ldc + j];
         }
   double *dA, *dB, *dC, *dW = NULL;
   CK(hipMalloc(&dA, A.size() * 8)); CK(hipMalloc(&dB, B.size() * 8)); CK(hipMalloc(&dC, C.size() * 8));
   if ( splitk > 1 ) CK(hipMalloc(&dW, (size_t) splitk * M * N * 8));
   CK(hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice));
   CK(hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice));
   CK(hipMemcpy(dC, C.data(), C.size() * 8, hipMemcpyHostToDevice));
   hs_gemm_args g = {M, N, K, layA, layB, dA, lda, sA, dB, ldb, sB, dC, ldc, sC, alpha, beta, batch, flags, splitk, dW};
   int rc = hs_dgemm(0, &g);
   CK(hipDeviceSynchronize());
   CK(hipMemcpy(C.data(), dC, C.size() * 8, hipMemcpyDeviceToHost));
   double maxerr = 0;
   for (int b = 0; b < batch; ++b)
      for (int i = 0; i < M; ++i)
         for (int j = 0; j < N; ++j)
         {
            if ( (flags & HS_GEMM_LOWER) && i < j ) continue;
            maxerr = fmax(maxerr, fabs(C[b * sC + i * ldc + j] - R[b * sC + i * ldc + j]));
         }
   printf("check M=%d N=%d K=%d lay=%d%d batch=%d splitk=%d flags=%d rc=%d maxerr=%.3e %s\n", M, N, K, layA, layB, batch,
      splitk, flags, rc, maxerr, (rc == 0 && maxerr < 1e-11 * (K + 1)) ? "OK" : "FAIL");
   hipFree(dA); hipFree(dB); hipFree(dC); if (dW) hipFree(dW);
   return (rc == 0 && maxerr < 1e-11 * (K + 1)) ? 0 : 1;
}

static void bench(int M, int N, int K, int layA, int layB, int splitk, rocblas_handle h)
{
   long long lda = (layA == HS_KC ? K : M), ldb = (layB == HS_KC ? K : N), ldc = N;
   size_t nA = (size_t) M * K, nB = (size_t) N * K, nC = (size_t) M * N;
   double *dA, *dB, *dC, *dW = NULL;
   CK(hipMalloc(&dA, nA * 8)); CK(hipMalloc(&dB, nB * 8)); CK(hipMalloc(&dC, nC * 8));
   if ( splitk > 1 ) CK(hipMalloc(&dW, (size_t) splitk * nC * 8));
   std::vector<double> hA(nA), hB(nB);
   for (auto& x : hA) x = urand();
   for (auto& x : hB) x = urand();
   CK(hipMemcpy(dA, hA.data(), nA * 8, hipMemcpyHostToDevice));
   CK(hipMemcpy(dB, hB.data(), nB * 8, hipMemcpyHostToDevice));
   hs_gemm_args g = {M, N, K, layA, layB, dA, lda, 0, dB, ldb, 0, dC, ldc, 0, 1.0, 0.0, 1, 0, splitk, dW};
   hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
   for (int w = 0; w < 2; ++w) hs_dgemm(0, &g);
   const int reps = 5;
   CK(hipEventRecord(e0, 0));
   for (int r = 0; r < reps; ++r) hs_dgemm(0, &g);
   CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
   float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
   double tf = 2.0 * M * N * (double) K / (ms * 1e-3) / 1e12;

   /* rocBLAS is column-major: C^T[N x M] = op(B)^T ... ; we only need a same-shape timing */
   const double one = 1.0, zero = 0.0;
   rocblas_operation opA = layB == HS_MC ? rocblas_operation_none : rocblas_operation_transpose;
   rocblas_operation opB = layA == HS_KC ? rocblas_operation_none : rocblas_operation_transpose;
   for (int w = 0; w < 2; ++w)
      rocblas_dgemm(h, opA, opB, N, M, K, &one, dB, (int) ldb, dA, (int) lda, &zero, dC, (int) ldc);
   CK(hipEventRecord(e0, 0));
   for (int r = 0; r < reps; ++r)
      rocblas_dgemm(h, opA, opB, N, M, K, &one, dB, (int) ldb, dA, (int) lda, &zero, dC, (int) ldc);
   CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
   float ms2; CK(hipEventElapsedTime(&ms2, e0, e1)); ms2 /= reps;
   double tf2 = 2.0 * M * N * (double) K / (ms2 * 1e-3) / 1e12;
   printf("bench M=%d N=%d K=%d lay=%d%d splitk=%d : hs %.3f ms %.2f TF | rocblas %.3f ms %.2f TF\n", M, N, K, layA, layB,
      splitk, ms, tf, ms2, tf2);
   hipFree(dA); hipFree(dB); hipFree(dC); if (dW) hipFree(dW);
}

int main(int argc, char** argv)
{
   int fails = 0;
   for (int la = 0; la < 2; ++la)
      for (int lb = 0; lb < 2; ++lb)
      {
         fails += check(37, 53, 29, la, lb, 1, 1, 0);
         fails += check(130, 257, 100, la, lb, 3, 1, 0);
         fails += check(300, 300, 1111, la, lb, 1, 4, 0);
         fails += check(1500, 1400, 70, la, lb, 1, 1, 0);
      }
   fails += check(520, 520, 333, HS_KC, HS_KC, 1, 1, HS_GEMM_LOWER);
   fails += check(2100, 2100, 200, HS_KC, HS_KC, 1, 1, HS_GEMM_LOWER);
   fails += check(520, 520, 4000, HS_KC, HS_KC, 1, 5, HS_GEMM_LOWER);
   printf("fails=%d\n", fails);
   if ( argc > 1 )
   {
      rocblas_handle h; rocblas_create_handle(&h);
      bench(4096, 4096, 4096, HS_KC, HS_MC, 1, h);
      bench(4096, 4096, 4096, HS_KC, HS_KC, 1, h);
      bench(8192, 8192, 8192, HS_KC, HS_MC, 1, h);
      bench(500 * 1000, 500, 500, HS_KC, HS_MC, 1, h);      /* GEMM1 of C2: stack of A_j times Zinv */
      bench(1000, 1000, 250000, HS_KC, HS_KC, 16, h);        /* GEMM3 of C2 */
      bench(2000, 2000, 1000000, HS_KC, HS_KC, 16, h);       /* GEMM3 of T1 */
      bench(500, 500, 500, HS_KC, HS_MC, 1, h);
      bench(1000, 1000, 1000, HS_KC, HS_MC, 1, h);
      rocblas_destroy_handle(h);
   }
   return fails;
}
