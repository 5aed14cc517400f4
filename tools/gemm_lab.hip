/* gemm_lab.hip - developer tool: where does the FP64 MFMA GEMM loop lose time?  (1) the sustained MFMA ceiling of the
 * chip (no memory traffic), (2) the production loop structure of csrc/dgemm.hip with pieces switched off.
 * Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_lab.hip -o build/gemm_lab */
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while(0)

typedef double v4d __attribute__((ext_vector_type(4)));
struct __attribute__((aligned(16))) d2a { double x, y; };

__global__ void __launch_bounds__(256) k_mfma_peak(int iters, double* out)
{
   v4d acc[16];
   for (int i = 0; i < 16; ++i) acc[i] = (v4d){0.0, 0.0, 0.0, 0.0};
   double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3;
   for (int it = 0; it < iters; ++it)
   {
#pragma unroll
      for (int i = 0; i < 16; ++i)
         acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
   }
   double s = 0.0;
   for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
   if ( s == 12345.678 ) out[0] = s;
}

#define BK 16
#define KCLD 18
#define BT 128
#define MCLD (BT + 16)
#define SZ (BT * KCLD)      /* 2304 doubles >= 16 * 144 */

/* A: [M][K] K-contiguous (KC), B: LB = 0: [N][K] (KC), LB = 1: [K][N] (MC).  Dimensions multiples of 128 / 16. */
template<int LB, int MODE>
__global__ void __launch_bounds__(256, 2) k_lab(int M, int N, int K, int ld, int remap, const double* __restrict__ A, const double* __restrict__ B,
   double* __restrict__ C)
{
   extern __shared__ __attribute__((aligned(16))) double smem[];
   const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
   const int tn = N / BT;
   int b = blockIdx.x;
   if ( remap ) { const int P = gridDim.x / 8; b = (b & 7) * P + (b >> 3); }
   const int m0 = (b / tn) * BT, n0 = (b % tn) * BT;
   v4d acc[4][4];
#pragma unroll
   for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
         acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
   d2a ra[4], rb[4];
   const double* fa = A + (long long) (m0 + (tid >> 3)) * ld + 2 * (tid & 7);
   const double* fb = LB == 0 ? B + (long long) (n0 + (tid >> 3)) * ld + 2 * (tid & 7)
                              : B + (long long) (tid / 64) * ld + n0 + 2 * (tid % 64);
   auto gload = [&](int step)
   {
#pragma unroll
      for (int i = 0; i < 4; ++i)
      {
         ra[i] = *reinterpret_cast<const d2a*>(fa + (long long) step * BK + (long long) (32 * i) * ld);
         if ( LB == 0 )
            rb[i] = *reinterpret_cast<const d2a*>(fb + (long long) step * BK + (long long) (32 * i) * ld);
         else
            rb[i] = *reinterpret_cast<const d2a*>(fb + ((long long) step * BK + 4 * i) * ld);
      }
   };
   auto sstore = [&](double* s)
   {
#pragma unroll
      for (int i = 0; i < 4; ++i)
      {
         *reinterpret_cast<d2a*>(s + ((tid >> 3) + 32 * i) * KCLD + 2 * (tid & 7)) = ra[i];
         if ( LB == 0 )
            *reinterpret_cast<d2a*>(s + SZ + ((tid >> 3) + 32 * i) * KCLD + 2 * (tid & 7)) = rb[i];
         else
            *reinterpret_cast<d2a*>(s + SZ + ((tid / 64) + 4 * i) * MCLD + 2 * (tid % 64)) = rb[i];
      }
   };
   const int ntiles = K / BK;
   gload(0);
   sstore(smem);
   __syncthreads();
   double fra[4], frb[4];
   if ( MODE >= 4 )
   {
#pragma unroll
      for (int i = 0; i < 4; ++i)
      {
         fra[i] = smem[(wm * 64 + 16 * i + (lane & 15)) * KCLD + (lane >> 4)];
         frb[i] = smem[SZ + (wn * 64 + 16 * i + (lane & 15)) * KCLD + (lane >> 4)];
      }
   }
   for (int t = 0; t < ntiles; ++t)
   {
      const double* sa = smem + (t & 1) * 2 * SZ;
      const double* sb = sa + SZ;
      if ( MODE == 0 && t + 1 < ntiles )
         gload(t + 1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
      {
         if ( MODE < 4 )
         {
#pragma unroll
            for (int i = 0; i < 4; ++i)
            {
               fra[i] = sa[(wm * 64 + 16 * i + (lane & 15)) * KCLD + 4 * ks + (lane >> 4)];
               if ( LB == 0 )
                  frb[i] = sb[(wn * 64 + 16 * i + (lane & 15)) * KCLD + 4 * ks + (lane >> 4)];
               else
                  frb[i] = sb[(4 * ks + (lane >> 4)) * MCLD + wn * 64 + 16 * i + (lane & 15)];
            }
         }
#pragma unroll
         for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
               acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fra[i], frb[j], acc[i][j], 0, 0, 0);
      }
      if ( MODE <= 1 && t + 1 < ntiles )
         sstore(smem + ((t + 1) & 1) * 2 * SZ);
      if ( MODE <= 2 )
         __syncthreads();
   }
#pragma unroll
   for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
         for (int j = 0; j < 4; ++j)
            C[(long long) (m0 + wm * 64 + 16 * i + (lane >> 4) + 4 * r) * N + n0 + wn * 64 + 16 * j + (lane & 15)] = acc[i][j][r];
}

template<int LB, int MODE>
static void run(int M, int N, int K, int ld, int remap, const double* A, const double* B, double* C, const char* what)
{
   const size_t smem = 4 * SZ * sizeof(double);
   CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lab<LB, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem));
   hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
   const int grid = (M / BT) * (N / BT);
   for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_lab<LB, MODE>), dim3(grid), dim3(256), smem, 0, M, N, K, ld, remap, A, B, C);
   CK(hipEventRecord(e0, 0));
   const int reps = 5;
   for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_lab<LB, MODE>), dim3(grid), dim3(256), smem, 0, M, N, K, ld, remap, A, B, C);
   CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
   float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
   printf("LB=%d mode %d ld %d remap %d %-44s %8.3f ms  %6.2f TF\n", LB, MODE, ld, remap, what, ms, 2.0 * M * N * K / ms / 1e9);
}

int main()
{
   double* out; CK(hipMalloc(&out, 64));
   for (int wg : {256 * 1, 256 * 2, 256 * 4})
   {
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      const int iters = 20000;
      hipLaunchKernelGGL(k_mfma_peak, dim3(wg), dim3(256), 0, 0, 100, out);
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(k_mfma_peak, dim3(wg), dim3(256), 0, 0, iters, out);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double fl = (double) wg * 4 * iters * 16 * 2048.0;
      printf("mfma peak: %d workgroups of 4 waves: %.3f ms, %.2f TF\n", wg, ms, fl / ms / 1e9);
   }
   const int M = 8192, N = 4096, K = 4096;
   double *A, *B, *C;
   CK(hipMalloc(&A, (size_t) M * (K + 64) * 8)); CK(hipMalloc(&B, (size_t) M * (K + 64) * 8)); CK(hipMalloc(&C, (size_t) M * N * 8));
   std::vector<double> h((size_t) M * K); for (auto& x : h) x = (double) rand() / RAND_MAX - 0.5;
   CK(hipMemcpy(A, h.data(), (size_t) M * K * 8, hipMemcpyHostToDevice));
   CK(hipMemcpy(B, h.data(), (size_t) M * K * 8, hipMemcpyHostToDevice));
   for (int ld : {4096, 4096 + 24})
      for (int remap : {0, 1})
      {
         run<0, 0>(M, N, K, ld, remap, A, B, C, "full");
         run<1, 0>(M, N, K, ld, remap, A, B, C, "full");
      }
   run<0, 1>(M, N, K, 4120, 0, A, B, C, "no global loads");
   run<0, 2>(M, N, K, 4120, 0, A, B, C, "no global loads, no LDS stores");
   run<1, 1>(M, N, K, 4120, 0, A, B, C, "no global loads");
   run<1, 2>(M, N, K, 4120, 0, A, B, C, "no global loads, no LDS stores");
   return 0;
}
