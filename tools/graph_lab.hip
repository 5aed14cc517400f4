/* graph_lab.hip - developer tool: per-kernel latency of a chain of dependent tiny kernels, stream launches vs hipGraph */
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while(0)
__global__ void k_tiny(double* x) { x[threadIdx.x] = x[threadIdx.x] * 1.0000001 + 1e-9; }
int main()
{
   double* x; CK(hipMalloc(&x, 4096)); CK(hipMemset(x, 0, 4096));
   hipStream_t s; CK(hipStreamCreate(&s));
   const int N = 200, reps = 50;
   for (int w = 0; w < 3; ++w) { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, x); CK(hipStreamSynchronize(s)); }
   auto t0 = std::chrono::steady_clock::now();
   for (int r = 0; r < reps; ++r) { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, x); CK(hipStreamSynchronize(s)); }
   double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
   printf("stream launches: %.2f us per kernel\n", us / (reps * N));
   hipGraph_t g; hipGraphExec_t ge;
   CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
   for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, x);
   CK(hipStreamEndCapture(s, &g));
   CK(hipGraphInstantiate(&ge, g, NULL, NULL, 0));
   for (int w = 0; w < 3; ++w) { CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s)); }
   t0 = std::chrono::steady_clock::now();
   for (int r = 0; r < reps; ++r) { CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s)); }
   us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
   printf("graph launch   : %.2f us per kernel\n", us / (reps * N));
   return 0;
}
