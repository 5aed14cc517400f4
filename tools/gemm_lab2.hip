/* gemm_lab2.hip - developer tool: direct-to-LDS (LDS-DMA) staged FP64 GEMM loop with a multi-slot ring, against the
 * production register-staged kernel.  Interior tiles only (M, N multiples of 128, K multiple of the stage depth).
 * Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_lab2.hip -Lscip-sdp_amd/lib -lhipsdp -o build/gemm_lab2 */
#include "../scip-sdp_amd/csrc/hs_common.h"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <rocblas/rocblas.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while(0)

typedef double v4d __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

#define BT 128

__device__ __forceinline__ void glds16(const double* g, double* l)
{
   __builtin_amdgcn_global_load_lds((gbl_ptr_t) g, (lds_ptr_t) l, 16, 0, 0);
}

/* one wave's share of a stage of one operand: operand tile = 128 rows x BKS k.  KC: pieces of 16 rows x (BKS/… ) */
template<int BKS, int LAY>
__device__ __forceinline__ void stage_issue(const double* __restrict__ P, long long ld, int r0, int k0, double* slot, int wave, int lane)
{
   if ( LAY == HS_KC )
   {
      /* piece = 1 KiB = 64 chunks of 16 B.  BKS = 8: 16 rows x 4 chunks, image [c][r];  BKS = 16: 8 rows x 8 chunks, image [c ^ (piece & 1)][r] */
      constexpr int CPR = BKS / 2;            /* chunks per row */
      constexpr int RPP = 64 / CPR;           /* rows per piece */
      constexpr int NP = BT / RPP;            /* pieces per operand stage */
#pragma unroll
      for (int i = 0; i < NP / 4; ++i)
      {
         const int piece = wave * (NP / 4) + i;
         const int r = lane % RPP;
         int c = lane / RPP;
         if ( BKS == 16 )
            c ^= (piece & 1);
         glds16(P + (long long) (r0 + piece * RPP + r) * ld + k0 + 2 * c, slot + piece * 128);
      }
   }
   else
   {
      /* one piece = one k row of 128 columns; chunk j of row k sits at position j ^ ((k & 1) << 3) */
#pragma unroll
      for (int i = 0; i < BKS / 4; ++i)
      {
         const int kk = wave * (BKS / 4) + i;
         const int j = lane ^ ((kk & 1) << 3);
         glds16(P + (long long) (k0 + kk) * ld + r0 + 2 * j, slot + kk * 128);
      }
   }
}

/* fragment: element (row woff + 16 t + (l & 15), k = 4 ks + (l >> 4)) of the stage */
template<int BKS, int LAY>
__device__ __forceinline__ double stage_frag(const double* __restrict__ slot, int woff, int t, int ks, int lane)
{
   if ( LAY == HS_KC )
   {
      const int row = woff + 16 * t + (lane & 15);
      const int k = 4 * ks + (lane >> 4);
      if ( BKS == 8 )
         return slot[(row >> 4) * 128 + (k >> 1) * 32 + (row & 15) * 2 + (k & 1)];
      else
      {
         const int piece = row >> 3;
         return slot[piece * 128 + ((k >> 1) ^ (piece & 1)) * 16 + (row & 7) * 2 + (k & 1)];
      }
   }
   else
   {
      const int k = 4 * ks + (lane >> 4);
      const int col = woff + 16 * t + (lane & 15);
      return slot[k * 128 + (((col >> 1) ^ ((k & 1) << 3)) << 1) + (col & 1)];
   }
}

template<int N> __device__ __forceinline__ void wait_vm()
{
   asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

__device__ unsigned long long g_clk[4];

template<int BKS, int NS, int LB, int WGPC, int EXP>
__global__ void __launch_bounds__(256, WGPC) k_v2(int M, int N, int K, int lda, int ldb, const double* __restrict__ A,
   const double* __restrict__ B, double* __restrict__ C)
{
   const unsigned long long c0 = clock64(), w0 = wall_clock64();
   extern __shared__ __attribute__((aligned(1024))) double smem[];
   constexpr int OPSZ = BT * BKS;            /* doubles per operand per stage */
   constexpr int SLOT = 2 * OPSZ;
   constexpr int GPS = 2 * (BT * BKS * 8 / 1024) / 4;     /* glds per wave per stage */
   const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
   const int tn = N / BT;
   int b = blockIdx.x;
   { const int P = gridDim.x / 8; b = (b & 7) * P + (b >> 3); }
   const int m0 = (b / tn) * BT, n0 = (b % tn) * BT;
   const int lm0 = EXP == 1 ? 0 : m0, ln0 = EXP == 1 ? 0 : n0;      /* EXP 1: every workgroup loads the same tiles (all cache hits) */
   v4d acc[4][4];
#pragma unroll
   for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
         acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
   const int nst = K / BKS;
#pragma unroll
   for (int s = 0; s < NS - 1; ++s)
      if ( s < nst )
      {
         stage_issue<BKS, HS_KC>(A, lda, lm0, s * BKS, smem + s * SLOT, wave, lane);
         stage_issue<BKS, LB>(B, ldb, ln0, s * BKS, smem + s * SLOT + OPSZ, wave, lane);
      }
   for (int t = 0; t < nst; ++t)
   {
      /* stage t must have landed: stages t+1 .. t+NS-2 may stay in flight */
      if ( EXP == 2 )
         wait_vm<0>();
      else if ( t + NS - 2 < nst )
         wait_vm<(NS - 2) * GPS>();
      else
         wait_vm<0>();
      __builtin_amdgcn_s_barrier();
      if ( EXP != 2 && t + NS - 1 < nst )
      {
         double* dst = smem + ((t + NS - 1) % NS) * SLOT;
         stage_issue<BKS, HS_KC>(A, lda, lm0, (t + NS - 1) * BKS, dst, wave, lane);
         stage_issue<BKS, LB>(B, ldb, ln0, (t + NS - 1) * BKS, dst + OPSZ, wave, lane);
      }
      const double* sa = smem + (t % NS) * SLOT;
      const double* sb = sa + OPSZ;
#pragma unroll
      for (int ks = 0; ks < BKS / 4; ++ks)
      {
         double fa[4], fb[4];
#pragma unroll
         for (int i = 0; i < 4; ++i)
         {
            fa[i] = stage_frag<BKS, HS_KC>(sa, wm * 64, i, ks, lane);
            fb[i] = stage_frag<BKS, LB>(sb, wn * 64, i, ks, lane);
         }
#pragma unroll
         for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
               acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
   }
#pragma unroll
   for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
         for (int j = 0; j < 4; ++j)
            C[(long long) (m0 + wm * 64 + 16 * i + (lane >> 4) + 4 * r) * N + n0 + wn * 64 + 16 * j + (lane & 15)] = acc[i][j][r];
   if ( blockIdx.x == 1000 && tid == 0 )
   {
      g_clk[0] = clock64() - c0;
      g_clk[1] = wall_clock64() - w0;
   }
}

static double* dA; static double* dB; static double* dC; static double* dR;
static std::vector<double> hC, hR;

template<int BKS, int NS, int LB, int WGPC, int EXP = 0>
static void run(int M, int N, int K, int lda, int ldb, const char* what)
{
   const size_t smem = (size_t) NS * 2 * BT * BKS * 8;
   CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_v2<BKS, NS, LB, WGPC, EXP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem));
   hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
   const int grid = (M / BT) * (N / BT);
   CK(hipMemset(dC, 0, (size_t) M * N * 8));
   for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_v2<BKS, NS, LB, WGPC, EXP>), dim3(grid), dim3(256), smem, 0, M, N, K, lda, ldb, dA, dB, dC);
   CK(hipGetLastError());
   CK(hipEventRecord(e0, 0));
   const int reps = 5;
   for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_v2<BKS, NS, LB, WGPC, EXP>), dim3(grid), dim3(256), smem, 0, M, N, K, lda, ldb, dA, dB, dC);
   CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
   float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
   CK(hipMemcpy(hC.data(), dC, (size_t) M * N * 8, hipMemcpyDeviceToHost));
   double err = 0.0, nrm = 0.0;
   for (size_t i = 0; i < (size_t) M * N; i += 97) { err = fmax(err, fabs(hC[i] - hR[i])); nrm = fmax(nrm, fabs(hR[i])); }
   unsigned long long hclk[4];
   CK(hipMemcpyFromSymbol(hclk, HIP_SYMBOL(g_clk), sizeof(hclk)));
   printf("[core clock %.0f MHz] ", 100.0 * (double) hclk[0] / (double) hclk[1]);
   printf("v2 BKS=%2d NS=%d LB=%d WG/CU=%d EXP=%d smem=%6zu %-10s %8.3f ms  %6.2f TF   maxerr %.2e (ref max %.2e)\n", BKS, NS, LB, WGPC, EXP, smem, what, ms,
      2.0 * M * N * K / ms / 1e9, err, nrm);
}


/* 256 x 128 tile, 8 waves (4 x 2), one workgroup per CU: 25 % less operand traffic per flop */
template<int BKS, int NS, int LB>
__global__ void __launch_bounds__(512, 1) k_v3(int M, int N, int K, int lda, int ldb, const double* __restrict__ A,
   const double* __restrict__ B, double* __restrict__ C)
{
   extern __shared__ __attribute__((aligned(1024))) double smem[];
   constexpr int OPA = 256 * BKS, OPB = 128 * BKS;
   constexpr int SLOT = OPA + OPB;
   constexpr int GPS = (OPA + OPB) * 8 / 1024 / 8;      /* glds per wave per stage: 3 at BKS = 8 */
   const int tid = threadIdx.x, lane = tid & 63;
   const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
   const int wm = wave >> 1, wn = wave & 1;
   const int tn = N / 128;
   int b = blockIdx.x;
   { const int P = gridDim.x / 8; b = (b & 7) * P + (b >> 3); }
   const int m0 = (b / tn) * 256, n0 = (b % tn) * 128;
   v4d acc[4][4];
#pragma unroll
   for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
         acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
   const int nst = K / BKS;
   auto issue = [&](int st)
   {
      double* slot = smem + (st % NS) * SLOT;
      const int k0 = st * BKS;
      /* A: 16 pieces of 16 rows (BKS = 8), 2 per wave */
#pragma unroll
      for (int i = 0; i < 2; ++i)
      {
         const int piece = wave * 2 + i;
         glds16(A + (long long) (m0 + piece * 16 + (lane & 15)) * lda + k0 + 2 * (lane >> 4), slot + piece * 128);
      }
      /* B: 8 pieces, 1 per wave */
      if ( LB == HS_KC )
         glds16(B + (long long) (n0 + wave * 16 + (lane & 15)) * ldb + k0 + 2 * (lane >> 4), slot + OPA + wave * 128);
      else
         glds16(B + (long long) (k0 + wave) * ldb + n0 + 2 * (lane ^ ((wave & 1) << 3)), slot + OPA + wave * 128);
   };
#pragma unroll
   for (int s = 0; s < NS - 1; ++s)
      if ( s < nst )
         issue(s);
   for (int t = 0; t < nst; ++t)
   {
      if ( t + NS - 2 < nst )
         wait_vm<(NS - 2) * GPS>();
      else
         wait_vm<0>();
      __builtin_amdgcn_s_barrier();
      if ( t + NS - 1 < nst )
         issue(t + NS - 1);
      const double* sa = smem + (t % NS) * SLOT;
      const double* sb = sa + OPA;
#pragma unroll
      for (int ks = 0; ks < BKS / 4; ++ks)
      {
         double fa[4], fb[4];
#pragma unroll
         for (int i = 0; i < 4; ++i)
         {
            fa[i] = stage_frag<BKS, HS_KC>(sa, wm * 64, i, ks, lane);
            fb[i] = stage_frag<BKS, LB>(sb, wn * 64, i, ks, lane);
         }
#pragma unroll
         for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
               acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
   }
#pragma unroll
   for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
         for (int j = 0; j < 4; ++j)
            C[(long long) (m0 + wm * 64 + 16 * i + (lane >> 4) + 4 * r) * N + n0 + wn * 64 + 16 * j + (lane & 15)] = acc[i][j][r];
}

template<int BKS, int NS, int LB>
static void run3(int M, int N, int K, int lda, int ldb)
{
   const size_t smem = (size_t) NS * (256 + 128) * BKS * 8;
   CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_v3<BKS, NS, LB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem));
   hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
   const int grid = (M / 256) * (N / 128);
   CK(hipMemset(dC, 0, (size_t) M * N * 8));
   for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_v3<BKS, NS, LB>), dim3(grid), dim3(512), smem, 0, M, N, K, lda, ldb, dA, dB, dC);
   CK(hipGetLastError());
   CK(hipEventRecord(e0, 0));
   const int reps = 5;
   for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_v3<BKS, NS, LB>), dim3(grid), dim3(512), smem, 0, M, N, K, lda, ldb, dA, dB, dC);
   CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
   float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
   CK(hipMemcpy(hC.data(), dC, (size_t) M * N * 8, hipMemcpyDeviceToHost));
   double err = 0.0;
   for (size_t i = 0; i < (size_t) M * N; i += 97) err = fmax(err, fabs(hC[i] - hR[i]));
   printf("v3 256x128 BKS=%2d NS=%d LB=%d smem=%6zu   %8.3f ms  %6.2f TF   maxerr %.2e\n", BKS, NS, LB, smem, ms, 2.0 * M * N * K / ms / 1e9, err);
}

static void ref(int M, int N, int K, int lda, int ldb, int LB, int flags)
{
   hs_gemm_args g = {M, N, K, HS_KC, LB, dA, lda, 0, dB, ldb, 0, dR, N, 0, 1.0, 0.0, 1, flags, 1, NULL};
   hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
   for (int w = 0; w < 2; ++w) hs_dgemm(0, &g);
   CK(hipEventRecord(e0, 0));
   for (int r = 0; r < 5; ++r) hs_dgemm(0, &g);
   CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
   float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
   printf("production LB=%d flags %2d                       %8.3f ms  %6.2f TF\n", LB, flags, ms, 2.0 * M * N * K / ms / 1e9);
   CK(hipMemcpy(hR.data(), dR, (size_t) M * N * 8, hipMemcpyDeviceToHost));
}

int main()
{
   const int M = 8192, N = 4096, K = 4096, ld = 4096 + 24;
   CK(hipMalloc(&dA, (size_t) M * ld * 8)); CK(hipMalloc(&dB, (size_t) M * ld * 8));
   CK(hipMalloc(&dC, (size_t) M * N * 8)); CK(hipMalloc(&dR, (size_t) M * N * 8));
   std::vector<double> h((size_t) M * ld); for (auto& x : h) x = (double) rand() / RAND_MAX - 0.5;
   CK(hipMemcpy(dA, h.data(), (size_t) M * ld * 8, hipMemcpyHostToDevice));
   for (auto& x : h) x = (double) rand() / RAND_MAX - 0.5;
   CK(hipMemcpy(dB, h.data(), (size_t) M * ld * 8, hipMemcpyHostToDevice));
   hC.resize((size_t) M * N); hR.resize((size_t) M * N);
   {
      rocblas_handle h; rocblas_create_handle(&h);
      const double one = 1.0, zero = 0.0;
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      for (int tr = 0; tr < 2; ++tr)
      {
         /* column-major view: C^T (N x M) = op(B) * A^T */
         for (int w = 0; w < 7; ++w)
         {
            if ( w == 2 ) CK(hipEventRecord(e0, 0));
            rocblas_dgemm(h, tr ? rocblas_operation_none : rocblas_operation_transpose, rocblas_operation_none, N, M, K, &one, dB, ld, dA, ld, &zero, dR, N);
         }
         CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
         float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
         printf("rocBLAS dgemm (B %s)            %8.3f ms  %6.2f TF\n", tr ? "N" : "T", ms, 2.0 * M * N * K / ms / 1e9);
      }
   }
   for (int LB = 0; LB < 2; ++LB)
   {
      ref(M, N, K, ld, ld, LB, 0);
      ref(M, N, K, ld, ld, LB, HS_GEMM_REMAP);
      if ( LB == 0 )
      {
         run<8, 4, 0, 2>(M, N, K, ld, ld, "");
         run3<8, 4, 0>(M, N, K, ld, ld);
         run3<8, 6, 0>(M, N, K, ld, ld);
         run<16, 2, 0, 2>(M, N, K, ld, ld, "");
         run<8, 4, 0, 2, 1>(M, N, K, ld, ld, "same tile");
         run<16, 2, 0, 2, 1>(M, N, K, ld, ld, "same tile");
         run<8, 4, 0, 2, 2>(M, N, K, ld, ld, "no loads");
         run<16, 2, 0, 2, 2>(M, N, K, ld, ld, "no loads");
      }
      else
      {
         run<8, 4, 1, 2>(M, N, K, ld, ld, "");
         run3<8, 4, 1>(M, N, K, ld, ld);
         run3<8, 6, 1>(M, N, K, ld, ld);
         run<16, 2, 1, 2>(M, N, K, ld, ld, "");
         run<8, 4, 1, 2, 1>(M, N, K, ld, ld, "same tile");
         run<16, 2, 1, 2, 1>(M, N, K, ld, ld, "same tile");
         run<8, 4, 1, 2, 2>(M, N, K, ld, ld, "no loads");
         run<16, 2, 1, 2, 2>(M, N, K, ld, ld, "no loads");
      }
   }
   return 0;
}
