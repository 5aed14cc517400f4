/* dvfs_lab.hip - developer tool: does the core clock sag during the launch-bound phase of an IPM iteration, and does a small
 * MFMA load on a side stream keep it up?  Emulates an iteration: ~6 ms of tiny dependent kernels, then the stack GEMM.
 * Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/dvfs_lab.hip -Lscip-sdp_amd/lib -lhipsdp -o build/dvfs_lab */
#include "../scip-sdp_amd/csrc/hs_common.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while(0)
typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void k_tiny(double* x, int spin)
{
   double v = x[threadIdx.x];
   for (int i = 0; i < spin; ++i) v = v * 1.0000001 + 1e-9;
   x[threadIdx.x] = v;
}

__global__ void __launch_bounds__(256) k_heater(int iters, const volatile int* stop, double* out)
{
   v4d acc[8];
   for (int i = 0; i < 8; ++i) acc[i] = (v4d){0.0, 0.0, 0.0, 0.0};
   double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3;
   for (int it = 0; it < iters; ++it)
   {
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
         for (int i = 0; i < 8; ++i)
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
      if ( *stop ) break;
   }
   double s = 0.0;
   for (int i = 0; i < 8; ++i) s += acc[i][0];
   if ( s == 12345.678 ) out[0] = s;
}

__global__ void k_setflag(int* f, int v) { *f = v; }

int main(int argc, char** argv)
{
   const int heater_wgs = argc > 1 ? atoi(argv[1]) : 0;
   const int n = 500, m1 = 1001;
   const long long n2 = (long long) n * n;
   double *A, *T, *S, *x; int* flag; double* out;
   CK(hipMalloc(&A, m1 * n2 * 8)); CK(hipMalloc(&T, m1 * n2 * 8)); CK(hipMalloc(&S, n2 * 8)); CK(hipMalloc(&x, 4096)); CK(hipMalloc(&flag, 4)); CK(hipMalloc(&out, 64));
   std::vector<double> h(n2); for (auto& v : h) v = (double) rand() / RAND_MAX - 0.5;
   CK(hipMemcpy(S, h.data(), n2 * 8, hipMemcpyHostToDevice));
   for (int j = 0; j < m1; ++j) CK(hipMemcpy(A + j * n2, h.data(), n2 * 8, hipMemcpyHostToDevice));
   CK(hipMemset(x, 0, 4096)); CK(hipMemset(flag, 0, 4));
   hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
   hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
   hs_gemm_args g = {m1 * n, n, n, HS_KC, HS_MC, A, n, 0, S, n, 0, T, n, 0, 1.0, 0.0, 1, HS_GEMM_B_LOWTRI, 1, NULL};
   double tg = 0.0, tl = 0.0;
   const int reps = 12;
   for (int r = 0; r < reps; ++r)
   {
      CK(hipEventRecord(e0, s1));
      if ( heater_wgs > 0 )
      {
         hipLaunchKernelGGL(k_setflag, dim3(1), dim3(1), 0, s2, flag, 0);
         hipLaunchKernelGGL(k_heater, dim3(heater_wgs), dim3(256), 0, s2, 3000, flag, out);      /* <= ~8 ms */
      }
      for (int i = 0; i < 600; ++i)
         hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s1, x, 2000);
      if ( heater_wgs > 0 )
         hipLaunchKernelGGL(k_setflag, dim3(1), dim3(1), 0, s1, flag, 1);
      CK(hipEventRecord(e1, s1));
      for (int k = 0; k < 3; ++k) hs_dgemm(s1, &g);
      CK(hipEventRecord(e2, s1));
      CK(hipEventSynchronize(e2));
      CK(hipStreamSynchronize(s2));
      float a, b; CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b, e1, e2));
      if ( r >= 2 ) { tl += a; tg += b; }
      printf("rep %2d light %.3f ms, 3 x stack GEMM %.3f ms\n", r, a, b);
   }
   printf("heater wgs %d: light %.3f ms, GEMM x3 %.3f ms\n", heater_wgs, tl / (reps - 2), tg / (reps - 2));
   return 0;
}
