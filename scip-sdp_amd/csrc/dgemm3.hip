/* dgemm3.hip - latency-oriented FP64 MFMA GEMM for the small products of an iteration (n x n x n chain products at n of a few
 * hundred, the panels of the triangular inverse, the Cholesky check of the step).
 *
 * Those products are 64 tiles of 64 x 64 on 256 CUs: the tile kernel of dgemm.hip either leaves three quarters of the chip idle
 * or needs split-K slabs and a second launch to sum them (hs_splitk_reduce_kernel), and pays an LDS staging round trip per 16
 * K steps on a loop that is only 8-31 steps long.  Here
 *   - a workgroup owns a 32 x 32 tile of C (n = 500: 256 workgroups = one per CU) and its four wavefronts split K among
 *     themselves; each wavefront accumulates the whole tile (2 x 2 MFMA tiles of 16 x 16) over its quarter of K;
 *   - operands go global -> registers directly in MFMA fragment layout with 16-byte loads, several K steps ahead of the
 *     matrix pipe - no LDS on the operand path, no barrier in the K loop (the matrices are a few MB and sit in L2 / MALL);
 *     a K-contiguous operand delivers two consecutive k per lane (one load feeds two MFMAs), a row-contiguous one two
 *     adjacent rows (one load feeds the two row tiles): the k's of one MFMA are then {k, k + 2, k + 4, k + 6} and the rows of a
 *     tile the even / odd ones, which only changes the (fixed) summation order and the bookkeeping of the epilogue;
 *   - the four partial tiles are summed through LDS in wavefront order (deterministic) and stored once: no slabs, no second
 *     launch.
 * Results agree with dgemm.hip to rounding (different summation order), run-to-run bitwise identical. */
#include "hs_common.h"
#include <cstdlib>

typedef double v4d3 __attribute__((ext_vector_type(4)));
struct __attribute__((aligned(8))) d2v { double x, y; };

#define G3_BT 32

/* fragments of one 8-deep K step for a 32-wide operand range starting at r0: f[t][h] = operand value for MFMA tile t (0, 1) and
 * k half h (0, 1) of this lane.  KC (k contiguous): tile t = rows r0 + 16 t + (lane & 15), half h = k + 2 (lane >> 4) + h.
 * MC (rows contiguous): tile t = rows r0 + 2 (lane & 15) + t, same k.  Branch free: addresses are clamped into the operand
 * (rows to R - 1 resp. the pair to R - 2, k to K - 2 resp. K - 1; the products of clamped rows are not stored) and values with
 * k >= kend are replaced by zero, so that the K loop has no control flow and its loads stay in flight across steps. */
template<int LAY>
__device__ __forceinline__ void g3_load(double (&f)[2][2], const double* __restrict__ P, long long ld, int r0, int R, int K, int k, int kend,
   int lane)
{
   const int kq = lane >> 4, lr = lane & 15;
   const int kk = k + 2 * kq;
   if ( LAY == HS_KC )
   {
      const int kc = min(kk, K - 2);
#pragma unroll
      for (int t = 0; t < 2; ++t)
      {
         const int row = min(r0 + 16 * t + lr, R - 1);
         const d2v u = *reinterpret_cast<const d2v*>(P + (long long) row * ld + kc);
         const double x = (kk == kc) ? u.x : u.y;
         f[t][0] = (kk < kend) ? x : 0.0;
         f[t][1] = (kk + 1 < kend) ? u.y : 0.0;
      }
   }
   else
   {
      const int row = r0 + 2 * lr;
      const int rc = min(row, R - 2);
#pragma unroll
      for (int h = 0; h < 2; ++h)
      {
         const int kh = min(kk + h, K - 1);
         const d2v u = *reinterpret_cast<const d2v*>(P + (long long) kh * ld + rc);
         const double x = (row == rc) ? u.x : u.y;
         f[0][h] = (kk + h < kend) ? x : 0.0;
         f[1][h] = (kk + h < kend) ? u.y : 0.0;
      }
   }
}

/* actual row (column) of slot s = 0..15 of MFMA tile t within the 32-wide range */
template<int LAY> __device__ __forceinline__ int g3_index(int t, int s) { return LAY == HS_KC ? 16 * t + s : 2 * s + t; }

template<int LA, int LB>
__global__ void __launch_bounds__(256) hs_dgemm3_kernel(hs_gemm_args p)
{
   __shared__ double red[3][4][4][64];                 /* partial tiles of wavefronts 1..3: [wave - 1][tile][register][lane] */
   const int lane = threadIdx.x & 63;
   const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
   const int m0 = blockIdx.x * G3_BT, n0 = blockIdx.y * G3_BT, bz = blockIdx.z;
   if ( (p.flags & HS_GEMM_LOWER) && m0 + G3_BT - 1 < n0 )
      return;
   if ( (p.flags & HS_GEMM_UPPER) && n0 + G3_BT - 1 < m0 )
      return;
   const double* A = p.A + (long long) bz * p.strideA;
   const double* B = p.B + (long long) bz * p.strideB;
   double* C = p.C + (long long) bz * p.strideC;
   /* K range of this wavefront: quarters rounded up to the 8-deep step */
   int kq4 = (p.K + 3) / 4;
   kq4 = (kq4 + 7) & ~7;
   const int k0 = wave * kq4;
   const int k1 = min(p.K, k0 + kq4);

   v4d3 acc[2][2];
#pragma unroll
   for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
         acc[i][j] = (v4d3){0.0, 0.0, 0.0, 0.0};

   constexpr int DEPTH = 4;                            /* K steps in flight */
   double fa[DEPTH][2][2], fb[DEPTH][2][2];
   const int nstep = k1 > k0 ? (k1 - k0 + 7) / 8 : 0;
   if ( nstep > 0 )
   {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d)
      {
         g3_load<LA>(fa[d], A, p.lda, m0, p.M, p.K, k0 + 8 * d, k1, lane);
         g3_load<LB>(fb[d], B, p.ldb, n0, p.N, p.K, k0 + 8 * d, k1, lane);
      }
      /* steps beyond nstep (the loop runs in groups of DEPTH) multiply zeros */
      for (int s0 = 0; s0 < nstep; s0 += DEPTH)
      {
#pragma unroll
         for (int d = 0; d < DEPTH; ++d)
         {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
               for (int i = 0; i < 2; ++i)
#pragma unroll
                  for (int j = 0; j < 2; ++j)
                     acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[d][i][h], fb[d][j][h], acc[i][j], 0, 0, 0);
            const int k = k0 + 8 * (s0 + d + DEPTH);
            g3_load<LA>(fa[d], A, p.lda, m0, p.M, p.K, k, k1, lane);
            g3_load<LB>(fb[d], B, p.ldb, n0, p.N, p.K, k, k1, lane);
         }
      }
   }
   /* sum the four partial tiles in wavefront order */
   if ( wave > 0 )
   {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
         for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
               red[wave - 1][2 * i + j][r][lane] = acc[i][j][r];
   }
   __syncthreads();
   if ( wave != 0 )
      return;
#pragma unroll
   for (int w = 0; w < 3; ++w)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
         for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
               acc[i][j][r] += red[w][2 * i + j][r][lane];
   const double alpha = p.alpha, beta = p.beta;
#pragma unroll
   for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r)
      {
         const int row = m0 + g3_index<LA>(i, (lane >> 4) + 4 * r);
         if ( row >= p.M )
            continue;
#pragma unroll
         for (int j = 0; j < 2; ++j)
         {
            const int col = n0 + g3_index<LB>(j, lane & 15);
            if ( col < p.N )
            {
               double* c = C + (long long) row * p.ldc + col;
               double v = alpha * acc[i][j][r];
               if ( beta != 0.0 )
                  v += beta * (*c);
               *c = v;
            }
         }
      }
}

template<int LA, int LB>
static int g3_launch(hipStream_t stream, const hs_gemm_args* a)
{
   dim3 grid((a->M + G3_BT - 1) / G3_BT, (a->N + G3_BT - 1) / G3_BT, a->batch);
   hipLaunchKernelGGL((hs_dgemm3_kernel<LA, LB>), grid, dim3(256), 0, stream, *a);
   if ( hipGetLastError() != hipSuccess )
      return -HS_ERR_HIP;
   return 1;
}

static int g3_disabled = -1;

int hs_dgemm3_enabled(void)
{
   if ( g3_disabled < 0 )
   {
      const char* env = getenv("HIPSDP_GEMM_SMALL");
      g3_disabled = (env != NULL && env[0] == '0') ? 1 : 0;
   }
   return g3_disabled ? 0 : 1;
}

/* 1: launched; 0: not this kernel's shape (the caller goes on with the tile kernels); < 0: error code negated.
 * Takes products without split-K whose 64 x 64 tiling would occupy less than about two thirds of the chip. */
int hs_dgemm3_try(hipStream_t stream, const hs_gemm_args* a)
{
   if ( !hs_dgemm3_enabled() || a->splitk > 1 || a->K < 16 || a->M < 2 || a->N < 2 )
      return 0;
   if ( a->flags & (HS_GEMM_XCD | HS_GEMM_REMAP | HS_GEMM_TILE64) )
      return 0;
   const long long t64 = (long long) ((a->M + 63) / 64) * ((a->N + 63) / 64) * a->batch;
   const long long t32 = (long long) ((a->M + 31) / 32) * ((a->N + 31) / 32) * a->batch;
   if ( t64 > 160 || t32 > 4096 || t32 > 65535LL * 64 )
      return 0;
   if ( (a->N + G3_BT - 1) / G3_BT > 65535 || a->batch > 65535 )
      return 0;
   if ( a->layA == HS_KC && a->layB == HS_KC ) return g3_launch<HS_KC, HS_KC>(stream, a);
   if ( a->layA == HS_KC && a->layB == HS_MC ) return g3_launch<HS_KC, HS_MC>(stream, a);
   if ( a->layA == HS_MC && a->layB == HS_KC ) return g3_launch<HS_MC, HS_KC>(stream, a);
   return g3_launch<HS_MC, HS_MC>(stream, a);
}
