/* gemm_syrk.cpp - developer tool: variants of the Schur SYRK  Mx = W W^T  (m1 = 1001, K = 250000) */
#include "../scip-sdp_amd/csrc/hs_common.h"
#include <vector>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while(0)
static double timeit(hs_gemm_args g, int reps)
{
   hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
   for (int w = 0; w < 2; ++w) if ( hs_dgemm(0, &g) != 0 ) { printf("launch error\n"); return -1; }
   CK(hipEventRecord(e0, 0));
   for (int r = 0; r < reps; ++r) hs_dgemm(0, &g);
   CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
   float ms; CK(hipEventElapsedTime(&ms, e0, e1));
   return ms / reps;
}
int main(int argc, char** argv)
{
   const int m1 = argc > 1 ? atoi(argv[1]) : 1001;
   const long long K = argc > 2 ? atoll(argv[2]) : 250000;
   double *W, *C, *S;
   CK(hipMalloc(&W, m1 * K * 8)); CK(hipMalloc(&C, (size_t) m1 * m1 * 8)); CK(hipMalloc(&S, (size_t) 64 * m1 * m1 * 8));
   std::vector<double> h(K); for (auto& x : h) x = (double) rand() / RAND_MAX - 0.5;
   for (int j = 0; j < m1; ++j) CK(hipMemcpy(W + j * K, h.data() + (j % 7), (K - 8) * 8, hipMemcpyHostToDevice));
   const long long tm = (m1 + 127) / 128;
   const double fl_low = (double) (tm * (tm + 1) / 2) * 128.0 * 128.0 * K * 2.0, fl_full = (double) tm * tm * 128.0 * 128.0 * K * 2.0;
   struct { const char* name; int flags; int sk; double fl; } v[] = {
      {"naive full  sk16", 0, 16, fl_full}, {"naive lower sk16", HS_GEMM_LOWER, 16, fl_low}, {"naive lower sk29", HS_GEMM_LOWER, 29, fl_low},
      {"xcd full    sk8 ", HS_GEMM_XCD, 8, fl_full}, {"xcd lower   sk14", HS_GEMM_XCD | HS_GEMM_LOWER, 14, fl_low},
      {"xcd lower   sk28", HS_GEMM_XCD | HS_GEMM_LOWER, 28, fl_low}, {"xcd lower nofast", HS_GEMM_XCD | HS_GEMM_LOWER | HS_GEMM_NOFAST, 14, fl_low},
      {"remap lower sk14", HS_GEMM_REMAP | HS_GEMM_LOWER, 14, fl_low}};
   for (auto& x : v)
   {
      hs_gemm_args g = {m1, m1, (int) K, HS_KC, HS_KC, W, K, 0, W, K, 0, C, m1, 0, 1.0, 0.0, 1, x.flags, x.sk, S};
      double ms = timeit(g, 4);
      printf("%s : %.3f ms  %.1f TF (executed tiles)\n", x.name, ms, x.fl / ms / 1e9);
   }
   return 0;
}
