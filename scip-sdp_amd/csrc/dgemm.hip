/* dgemm.hip - FP64 GEMM on the CDNA4 matrix cores (v_mfma_f64_16x16x4_f64), the workhorse of the Schur assembly
 * M_ij = tr(A_i X A_j Z^-1) and of the predictor-corrector n x n chain.
 *
 * Reference correspondence: the reference has no GEMM of its own on the hot path - the products live inside DSDP/SDPA
 * (sdpisolver_dsdp.c:1503 DSDPSolve, sdpisolver_sdpa.cpp:1620 SDPA::solve) - and one DGEMM wrapper off it
 * (lapack_interface.c:654-706), which lapack_interface_hip.c maps onto this kernel.
 *
 * Design (gfx950):
 *  - 256 threads = 4 wavefronts in a 2 x 2 arrangement; workgroup tile BT x BT (128 or 64), K step 16.
 *  - each wavefront owns (BT/2) x (BT/2) of C as (BT/32)^2 MFMA tiles of 16 x 16; one MFMA consumes a 16 x 4 slab of A
 *    and a 4 x 16 slab of B, one double per lane:  lane l holds A[l & 15][l >> 4] and B[l >> 4][l & 15];
 *    the f64 accumulator map is  col = l & 15, row = (l >> 4) + 4 * reg  (NOT the f32 map).
 *  - operands are staged global -> registers -> LDS in their natural orientation, so both stagings are 16-byte wide and
 *    coalesced:  K-contiguous operands sit in LDS as [row][16 + 2] doubles, row-contiguous ones as [16][BT + 16].
 *    Both paddings make the ds_read_b64 operand fetches conflict free (bank = (addr / 4) mod 64 per 32-lane half).
 *  - LDS is double buffered (2 x 2 x 18 KiB at BT = 128): one barrier per K step; the next tile's global loads are in
 *    flight while the current tile is multiplied; two workgroups per CU keep the matrix pipe busy across barriers.
 *  - split-K writes per-slice slabs and a second kernel sums them in slice order: results are bitwise reproducible.
 */
#include "hs_common.h"

typedef double v4d __attribute__((ext_vector_type(4)));

struct __attribute__((aligned(8))) d2u { double x, y; };      /* global pair, only 8-byte alignment promised */
struct __attribute__((aligned(16))) d2a { double x, y; };     /* LDS pair, 16-byte aligned */

#define HS_BK     16
#define HS_KCLD   (HS_BK + 2)

template<int BT> struct TileGeo
{
   static constexpr int MCLD = BT + 16;
   static constexpr int SZKC = BT * HS_KCLD;
   static constexpr int SZMC = HS_BK * MCLD;
   static constexpr int SZ   = SZKC > SZMC ? SZKC : SZMC;     /* doubles per operand tile */
   static constexpr int NLD  = BT / 32;                       /* 16-byte loads per thread per operand tile */
   static constexpr int WT   = BT / 32;                       /* MFMA tiles per wave per dimension */
};

/* global -> registers for one operand tile.  P points at the operand of this batch entry.
 * KC: element (r, k) at P[r * ld + k];  MC: element (r, k) at P[k * ld + r].  r in [r0, r0 + BT), k in [k0, k0 + 16). */
template<int BT, int LAY>
__device__ __forceinline__ void tile_gload(d2a (&v)[TileGeo<BT>::NLD], const double* __restrict__ P, long long ld,
   int r0, int R, int k0, int Kend, int tid)
{
   if ( LAY == HS_KC )
   {
      const int kp = tid & 7;
      const int r  = tid >> 3;
      const int k  = k0 + 2 * kp;
#pragma unroll
      for (int i = 0; i < TileGeo<BT>::NLD; ++i)
      {
         const int row = r0 + r + 32 * i;
         d2a t; t.x = 0.0; t.y = 0.0;
         if ( row < R )
         {
            const double* q = P + (long long) row * ld + k;
            if ( k + 1 < Kend )
            {
               d2u u = *reinterpret_cast<const d2u*>(q);
               t.x = u.x; t.y = u.y;
            }
            else if ( k < Kend )
               t.x = q[0];
         }
         v[i] = t;
      }
   }
   else
   {
      constexpr int TPR = BT / 2;            /* threads per k-row */
      constexpr int RPP = 256 / TPR;         /* k-rows per pass */
      const int c2  = tid % TPR;
      const int kr  = tid / TPR;
      const int col = r0 + 2 * c2;
#pragma unroll
      for (int i = 0; i < TileGeo<BT>::NLD; ++i)
      {
         const int k = k0 + kr + RPP * i;
         d2a t; t.x = 0.0; t.y = 0.0;
         if ( k < Kend )
         {
            const double* q = P + (long long) k * ld + col;
            if ( col + 1 < R )
            {
               d2u u = *reinterpret_cast<const d2u*>(q);
               t.x = u.x; t.y = u.y;
            }
            else if ( col < R )
               t.x = q[0];
         }
         v[i] = t;
      }
   }
}

/* interior tiles: no bounds checks, no branches.  base = this thread's first element of the operand tile at K step 0;
 * KC: element (r0 + r + 32 i, k0 + 2 kp), MC: element (k0 + kr + RPP i, r0 + 2 c2); step = K-step index */
template<int BT, int LAY>
__device__ __forceinline__ void tile_gload_fast(d2a (&v)[TileGeo<BT>::NLD], const double* __restrict__ base, long long ld, int step)
{
   if ( LAY == HS_KC )
   {
      const double* q = base + (long long) step * HS_BK;
#pragma unroll
      for (int i = 0; i < TileGeo<BT>::NLD; ++i)
      {
         const d2u u = *reinterpret_cast<const d2u*>(q + (long long) (32 * i) * ld);
         v[i].x = u.x; v[i].y = u.y;
      }
   }
   else
   {
      constexpr int RPP = 256 / (BT / 2);
      const double* q = base + (long long) step * HS_BK * ld;
#pragma unroll
      for (int i = 0; i < TileGeo<BT>::NLD; ++i)
      {
         const d2u u = *reinterpret_cast<const d2u*>(q + (long long) (RPP * i) * ld);
         v[i].x = u.x; v[i].y = u.y;
      }
   }
}

template<int BT, int LAY>
__device__ __forceinline__ const double* tile_fast_base(const double* __restrict__ P, long long ld, int r0, int k0, int tid)
{
   if ( LAY == HS_KC )
      return P + (long long) (r0 + (tid >> 3)) * ld + k0 + 2 * (tid & 7);
   constexpr int TPR = BT / 2;
   return P + (long long) (k0 + tid / TPR) * ld + r0 + 2 * (tid % TPR);
}

/* registers -> LDS */
template<int BT, int LAY>
__device__ __forceinline__ void tile_sstore(double* __restrict__ s, const d2a (&v)[TileGeo<BT>::NLD], int tid)
{
   if ( LAY == HS_KC )
   {
      const int kp = tid & 7;
      const int r  = tid >> 3;
#pragma unroll
      for (int i = 0; i < TileGeo<BT>::NLD; ++i)
         *reinterpret_cast<d2a*>(s + (r + 32 * i) * HS_KCLD + 2 * kp) = v[i];
   }
   else
   {
      constexpr int TPR = BT / 2;
      constexpr int RPP = 256 / TPR;
      const int c2 = tid % TPR;
      const int kr = tid / TPR;
#pragma unroll
      for (int i = 0; i < TileGeo<BT>::NLD; ++i)
         *reinterpret_cast<d2a*>(s + (kr + RPP * i) * TileGeo<BT>::MCLD + 2 * c2) = v[i];
   }
}

/* LDS -> one MFMA operand: element (row woff + 16 t + (l & 15), k = 4 ks + (l >> 4)) */
template<int BT, int LAY>
__device__ __forceinline__ double tile_frag(const double* __restrict__ s, int woff, int t, int ks, int lane)
{
   if ( LAY == HS_KC )
      return s[(woff + 16 * t + (lane & 15)) * HS_KCLD + 4 * ks + (lane >> 4)];
   else
      return s[(4 * ks + (lane >> 4)) * TileGeo<BT>::MCLD + woff + 16 * t + (lane & 15)];
}

template<int BT, int LA, int LB>
__global__ void __launch_bounds__(256, 2) hs_dgemm_kernel(hs_gemm_args p, int kchunk)
{
   extern __shared__ __attribute__((aligned(16))) double hs_smem[];
   constexpr int SZ = TileGeo<BT>::SZ;
   constexpr int WT = TileGeo<BT>::WT;

   const int tid  = threadIdx.x;
   const int lane = tid & 63;
   const int wave = tid >> 6;
   const int wm   = wave >> 1;
   const int wn   = wave & 1;

   int m0, n0, bz, ks0 = 0, kend = p.K;
   if ( p.flags & (HS_GEMM_XCD | HS_GEMM_REMAP) )
   {
      /* 1-D grid.  Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 labels the XCD; a speed assumption only),
       * so the logical index L = (b % 8) * P + b / 8 hands every XCD a CONTIGUOUS range of logical work items: neighbours in
       * L share operand panels through that XCD's L2 instead of fetching them from HBM once per XCD. */
      const long long tm = (p.M + BT - 1) / BT, tn = (p.N + BT - 1) / BT;
      const long long ntile = (p.flags & HS_GEMM_LOWER) ? tm * (tm + 1) / 2 : tm * tn;
      const long long nz = p.splitk > 1 ? p.splitk : p.batch;
      const long long total = ntile * nz;
      const long long P = (total + 7) / 8;
      const long long b = blockIdx.x;
      const long long L = (b & 7) * P + (b >> 3);
      if ( (b >> 3) >= P || L >= total )
         return;
      long long t;
      if ( p.flags & HS_GEMM_XCD )
      {
         bz = (int) (L / ntile);          /* slice-major: an XCD walks one or two K slices over all tiles */
         t = L - (long long) bz * ntile;
      }
      else
      {
         bz = (int) (L / ntile);          /* batch-major, then row tile, then column tile */
         t = L - (long long) bz * ntile;
      }
      int ti, tj;
      if ( p.flags & HS_GEMM_LOWER )
      {
         ti = (int) ((sqrt(8.0 * (double) t + 1.0) - 1.0) * 0.5);
         while ( (long long) (ti + 1) * (ti + 2) / 2 <= t ) ++ti;
         while ( (long long) ti * (ti + 1) / 2 > t ) --ti;
         tj = (int) (t - (long long) ti * (ti + 1) / 2);
      }
      else
      {
         ti = (int) (t / tn);
         tj = (int) (t - (long long) ti * tn);
      }
      m0 = ti * BT;
      n0 = tj * BT;
   }
   else
   {
      m0 = blockIdx.x * BT;
      n0 = blockIdx.y * BT;
      bz = blockIdx.z;
      if ( (p.flags & HS_GEMM_LOWER) && (m0 + BT - 1 < n0) )
         return;
      if ( (p.flags & HS_GEMM_UPPER) && (n0 + BT - 1 < m0) )
         return;
   }

   double* C = p.C;
   long long ldc = p.ldc;
   double alpha = p.alpha;
   double beta = p.beta;
   if ( p.splitk > 1 )
   {
      ks0 = bz * kchunk;
      kend = min(p.K, ks0 + kchunk);
      C = p.ws + (long long) bz * p.M * p.N;
      ldc = p.N;
      alpha = 1.0;
      beta = 0.0;
      bz = 0;
   }
   if ( p.flags & HS_GEMM_B_LOWTRI )
      ks0 = max(ks0, (n0 / HS_BK) * HS_BK);
   if ( p.flags & HS_GEMM_A_LOWTRI )
      kend = min(kend, m0 + BT);
   if ( p.flags & HS_GEMM_A_UPTRI )
      ks0 = max(ks0, (m0 / HS_BK) * HS_BK);
   const double* A = p.A + (long long) bz * p.strideA;
   const double* B = p.B + (long long) bz * p.strideB;
   C += (long long) bz * p.strideC;

   v4d acc[WT][WT];
#pragma unroll
   for (int i = 0; i < WT; ++i)
#pragma unroll
      for (int j = 0; j < WT; ++j)
         acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};

   d2a ra[TileGeo<BT>::NLD];
   d2a rb[TileGeo<BT>::NLD];

   const int ntiles = (kend - ks0 + HS_BK - 1) / HS_BK;
   /* K steps that lie completely inside [ks0, kend) of a tile that lies completely inside C take the check-free loads */
   const int nfullk = (p.flags & HS_GEMM_NOFAST) ? 0 : (kend - ks0) / HS_BK;
   const int nfulla = (m0 + BT <= p.M) ? nfullk : 0;
   const int nfullb = (n0 + BT <= p.N) ? nfullk : 0;
   const double* fa = tile_fast_base<BT, LA>(A, p.lda, m0, ks0, tid);
   const double* fb = tile_fast_base<BT, LB>(B, p.ldb, n0, ks0, tid);

   if ( ntiles > 0 )
   {
      tile_gload<BT, LA>(ra, A, p.lda, m0, p.M, ks0, kend, tid);
      tile_gload<BT, LB>(rb, B, p.ldb, n0, p.N, ks0, kend, tid);
      tile_sstore<BT, LA>(hs_smem, ra, tid);
      tile_sstore<BT, LB>(hs_smem + SZ, rb, tid);
   }
   __syncthreads();

   for (int t = 0; t < ntiles; ++t)
   {
      const double* sa = hs_smem + (t & 1) * 2 * SZ;
      const double* sb = sa + SZ;

      if ( t + 1 < nfulla )
         tile_gload_fast<BT, LA>(ra, fa, p.lda, t + 1);
      else if ( t + 1 < ntiles )
         tile_gload<BT, LA>(ra, A, p.lda, m0, p.M, ks0 + (t + 1) * HS_BK, kend, tid);
      if ( t + 1 < nfullb )
         tile_gload_fast<BT, LB>(rb, fb, p.ldb, t + 1);
      else if ( t + 1 < ntiles )
         tile_gload<BT, LB>(rb, B, p.ldb, n0, p.N, ks0 + (t + 1) * HS_BK, kend, tid);

#pragma unroll
      for (int ks = 0; ks < HS_BK / 4; ++ks)
      {
         double fa[WT];
         double fb[WT];
#pragma unroll
         for (int i = 0; i < WT; ++i)
         {
            fa[i] = tile_frag<BT, LA>(sa, wm * (BT / 2), i, ks, lane);
            fb[i] = tile_frag<BT, LB>(sb, wn * (BT / 2), i, ks, lane);
         }
#pragma unroll
         for (int i = 0; i < WT; ++i)
#pragma unroll
            for (int j = 0; j < WT; ++j)
               acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }

      if ( t + 1 < ntiles )
      {
         double* da = hs_smem + ((t + 1) & 1) * 2 * SZ;
         tile_sstore<BT, LA>(da, ra, tid);
         tile_sstore<BT, LB>(da + SZ, rb, tid);
      }
      __syncthreads();
   }

   /* epilogue: accumulator register r of tile (i, j) is C[16 i + (lane >> 4) + 4 r][16 j + (lane & 15)] */
#pragma unroll
   for (int i = 0; i < WT; ++i)
   {
#pragma unroll
      for (int r = 0; r < 4; ++r)
      {
         const int row = m0 + wm * (BT / 2) + 16 * i + (lane >> 4) + 4 * r;
         if ( row >= p.M )
            continue;
#pragma unroll
         for (int j = 0; j < WT; ++j)
         {
            const int col = n0 + wn * (BT / 2) + 16 * j + (lane & 15);
            if ( col < p.N )
            {
               double* c = C + (long long) row * ldc + col;
               double v = alpha * acc[i][j][r];
               if ( beta != 0.0 )
                  v += beta * (*c);
               *c = v;
            }
         }
      }
   }
}

/* sums the split-K slabs in slice order: C = alpha * sum_s ws[s] + beta * C */
__global__ void __launch_bounds__(256) hs_splitk_reduce_kernel(int M, int N, int nslices, const double* __restrict__ ws,
   double* __restrict__ C, long long ldc, double alpha, double beta, int lowerBT)
{
   const long long total = (long long) M * N;
   for (long long e = (long long) blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long) gridDim.x * blockDim.x)
   {
      const int row = (int) (e / N);
      const int col = (int) (e - (long long) row * N);
      if ( lowerBT > 0 && (row / lowerBT) * lowerBT + lowerBT - 1 < (col / lowerBT) * lowerBT )
         continue;
      if ( lowerBT < 0 && (col / (-lowerBT)) * (-lowerBT) + (-lowerBT) - 1 < (row / (-lowerBT)) * (-lowerBT) )
         continue;
      double s = 0.0;
      for (int k = 0; k < nslices; ++k)
         s += ws[(long long) k * total + e];
      double* c = C + (long long) row * ldc + col;
      double v = alpha * s;
      if ( beta != 0.0 )
         v += beta * (*c);
      *c = v;
   }
}

template<int BT, int LA, int LB>
static int launch_cfg(hipStream_t stream, const hs_gemm_args* a, int kchunk)
{
   static hs_attr_mask attr_done;
   const size_t smem = (size_t) 4 * TileGeo<BT>::SZ * sizeof(double);
   HS_CALL( hs_func_max_lds(reinterpret_cast<const void*>(&hs_dgemm_kernel<BT, LA, LB>), (int) smem, &attr_done) );
   dim3 grid((a->M + BT - 1) / BT, (a->N + BT - 1) / BT, a->splitk > 1 ? a->splitk : a->batch);
   if ( a->flags & (HS_GEMM_XCD | HS_GEMM_REMAP) )
   {
      const long long tm = (a->M + BT - 1) / BT, tn = (a->N + BT - 1) / BT;
      const long long nt = (a->flags & HS_GEMM_LOWER) ? tm * (tm + 1) / 2 : tm * tn;
      const long long total = nt * (a->splitk > 1 ? a->splitk : a->batch);
      const long long P = (total + 7) / 8;
      if ( 8 * P > 2147483647LL )
         return HS_ERR_ARG;
      grid = dim3((unsigned) (8 * P), 1, 1);
   }
   hipLaunchKernelGGL((hs_dgemm_kernel<BT, LA, LB>), grid, dim3(256), smem, stream, *a, kchunk);
   HS_HIP( hipGetLastError() );
   return HS_OK;
}

template<int BT>
static int launch_lay(hipStream_t stream, const hs_gemm_args* a, int kchunk)
{
   if ( a->layA == HS_KC && a->layB == HS_KC ) return launch_cfg<BT, HS_KC, HS_KC>(stream, a, kchunk);
   if ( a->layA == HS_KC && a->layB == HS_MC ) return launch_cfg<BT, HS_KC, HS_MC>(stream, a, kchunk);
   if ( a->layA == HS_MC && a->layB == HS_KC ) return launch_cfg<BT, HS_MC, HS_KC>(stream, a, kchunk);
   return launch_cfg<BT, HS_MC, HS_MC>(stream, a, kchunk);
}

static thread_local double g_mfma_flops = 0.0;
double hs_mfma_flops_total(void) { return g_mfma_flops; }
void hs_mfma_flops_add(double flops) { g_mfma_flops += flops; }

/* see hs_common.h.  BT: tile edge; kstage: K granularity of the kernel's main loop (a tile's K range is walked in whole stages);
 * kchunk: slice length of a split-K product; slabskip: the persistent kernel's interleaved-slab instance (dgemm2.hip, IL = 1)
 * issues only the 16 x 16 x 4 products whose operand slabs are not identically zero - the loop below repeats its predicate. */
double hs_gemm_executed_flops(const hs_gemm_args* a, int BT, int kstage, int kchunk, int slabskip)
{
   const long long tm = (a->M + BT - 1) / BT, tn = (a->N + BT - 1) / BT;
   const int nsl = a->splitk > 1 ? a->splitk : 1;
   const double per_k = 2.0 * (double) BT * (double) BT;            /* flops of one K step of a whole tile */
   const bool triA = (a->flags & HS_GEMM_A_LOWTRI) != 0, triB = (a->flags & HS_GEMM_B_LOWTRI) != 0;
   auto tile_flops = [&](int m0, int n0) -> double
   {
      double f = 0.0;
      for (int sl = 0; sl < nsl; ++sl)
      {
         int ks0 = 0, kend = a->K;
         if ( a->splitk > 1 )
         {
            ks0 = sl * kchunk;
            kend = ks0 + kchunk < a->K ? ks0 + kchunk : a->K;
         }
         if ( triB && ks0 < (n0 / 16) * 16 ) ks0 = (n0 / 16) * 16;
         if ( triA && kend > m0 + BT ) kend = m0 + BT;
         if ( (a->flags & HS_GEMM_A_UPTRI) && ks0 < (m0 / 16) * 16 ) ks0 = (m0 / 16) * 16;
         if ( kend <= ks0 )
            continue;
         const int stages = (kend - ks0 + kstage - 1) / kstage;
         if ( slabskip == 2 && BT == 128 && kstage == 8 && hs_dgemm2_tri5_eligible(a) )
         {
            /* paired-band kernel (hs_dgemm5_kernel): the band of 16 stages as 8 double stages of 36 matrix instructions per
             * wavefront, whatever part of it lies inside the K range; full stages outside it */
            const int nfull = triB ? (a->K > n0 + BT ? (a->K - n0 - BT + kstage - 1) / kstage : 0) : m0 / kstage;
            f += 2048.0 * 4.0 * (8.0 * 36.0 + 32.0 * (double) nfull);
            continue;
         }
         if ( !(slabskip && BT == 128 && kstage == 8 && (triA || triB)) )
         {
            f += per_k * (double) stages * kstage;
            continue;
         }
         long long mfma = 0;                                     /* 16 x 16 x 4 products issued, all four wavefronts */
         for (int st = 0; st < stages; ++st)
         {
            const int ck = ks0 + st * kstage;
            const bool bandB = triB && ck < n0 + BT, bandA = triA && ck + kstage > m0;
            for (int kk = ck; kk < ck + kstage; kk += 4)
               for (int wm = 0; wm < 2; ++wm)
                  for (int wn = 0; wn < 2; ++wn)
                  {
                     int jlim = 4, imin = 0;
                     if ( bandB )
                     {
                        const int d = kk + 3 - n0, qd = d >> 4;
                        jlim = (d < 0 || qd < wn) ? 0 : (((qd - wn) >> 1) + 1 < 4 ? ((qd - wn) >> 1) + 1 : 4);
                     }
                     if ( bandA )
                     {
                        const int e = kk - 15 - m0;
                        if ( e > 0 )
                        {
                           const int g = (e + 15) >> 4;
                           imin = g <= wm ? 0 : (((g - wm + 1) >> 1) < 4 ? ((g - wm + 1) >> 1) : 4);
                        }
                     }
                     mfma += (4 - imin) * jlim;
                  }
         }
         f += 2048.0 * (double) mfma;
      }
      return f;
   };
   double per_entry = 0.0;
   if ( a->flags & HS_GEMM_LOWER )
      per_entry = (double) (tm * (tm + 1) / 2) * tile_flops(0, 0);
   else if ( a->flags & HS_GEMM_UPPER )
   {
      long long cnt = 0;
      for (long long ti = 0; ti < tm; ++ti)
         for (long long tj = 0; tj < tn; ++tj)
            if ( (tj + 1) * BT > ti * BT )
               ++cnt;
      per_entry = (double) cnt * tile_flops(0, 0);
   }
   else
   {
      /* the K range depends on the row tile only through A_LOWTRI and on the column tile only through B_LOWTRI */
      const bool depA = triA || (a->flags & HS_GEMM_A_UPTRI);
      const long long ni = depA ? tm : 1, nj = triB ? tn : 1;
      for (long long ti = 0; ti < ni; ++ti)
         for (long long tj = 0; tj < nj; ++tj)
            per_entry += tile_flops((int) ti * BT, (int) tj * BT) * (double) (depA ? 1 : tm) * (double) (triB ? 1 : tn);
   }
   return per_entry * (double) (a->splitk > 1 ? 1 : a->batch);
}

int hs_dgemm_pick_splitk(int M, int N, int K, int lowerOnly)
{
   const long long tm = (M + 127) / 128;
   const long long tn = (N + 127) / 128;
   long long tiles = lowerOnly ? tm * (tm + 1) / 2 : tm * tn;
   if ( tiles >= 512 || K < 1024 )
      return 1;
   long long s = (1024 + tiles - 1) / tiles;
   const long long maxs = K / 256;
   if ( s > maxs ) s = maxs;
   if ( s > 64 ) s = 64;
   return s < 1 ? 1 : (int) s;
}

int hs_dgemm_pick_xcd_slices(long long ntile, long long K)
{
   /* candidates s with K / s >= 1024; score = occupied fraction of the last round of 512 slots, prefer fewer rounds */
   int best = 2;
   int widest = 2;
   double bestscore = -1.0;
   for (int s = 2; s <= 128; ++s)
   {
      if ( K / s < 1024 && s > 2 )
         break;
      widest = s;
      const long long total = ntile * s;
      const long long rounds = (total + 511) / 512;
      const double eff = (double) total / (double) (rounds * 512);
      const double score = eff - 0.02 * (double) (rounds - 1);
      if ( total >= 256 && score > bestscore )
      {
         bestscore = score;
         best = s;
      }
   }
   /* few tiles and a short K (m1 = 256 .. 700 with n below about 200): no candidate fills half the chip - take the most slices
    * the minimum slice length allows instead of the 2 the search starts from (n = 200, m = 300: 234 instead of 12 workgroups) */
   if ( bestscore < 0.0 )
      best = widest;
   return best;
}

int hs_dgemm(hipStream_t stream, const hs_gemm_args* a)
{
   if ( a->M < 0 || a->N < 0 || a->K < 0 || a->batch < 1 )
      return HS_ERR_ARG;
   if ( a->M == 0 || a->N == 0 )
      return HS_OK;
   if ( a->splitk > 1 && (a->batch != 1 || a->ws == NULL) )
      return HS_ERR_ARG;
   if ( (a->flags & HS_GEMM_XCD) && (a->splitk < 2 || ((a->flags & HS_GEMM_LOWER) && a->M != a->N)) )
      return HS_ERR_ARG;
   if ( (a->flags & HS_GEMM_REMAP) && (a->flags & HS_GEMM_LOWER) && a->M != a->N )
      return HS_ERR_ARG;

   /* few tiles, no split-K: the latency-oriented 32 x 32 kernel (dgemm3.hip) */
   if ( a->splitk <= 1 )
   {
      const int r3 = hs_dgemm3_try(stream, a);
      if ( r3 < 0 )
         return -r3;
      if ( r3 == 1 )
      {
         /* 32 x 32 tiles, K split over the four wavefronts in steps of 4 */
         hs_mfma_flops_add(2.0 * (double) (((a->M + 31) / 32) * 32) * (double) (((a->N + 31) / 32) * 32) * (double) (((a->K + 15) / 16) * 16) * (double) a->batch);
         return HS_OK;
      }
   }

   /* tile choice: big tiles once they fill the chip, small tiles otherwise */
   const long long big = (long long) ((a->M + 127) / 128) * ((a->N + 127) / 128) * (a->splitk > 1 ? a->splitk : a->batch);
   const bool useBig = (big >= 192 || (a->flags & HS_GEMM_XCD)) && !(a->flags & HS_GEMM_TILE64);
   const int BT = useBig ? 128 : 64;

   int kchunk = a->K;
   hs_gemm_args eff = *a;
   if ( a->splitk > 1 )
   {
      kchunk = (a->K + a->splitk - 1) / a->splitk;
      kchunk = ((kchunk + HS_BK - 1) / HS_BK) * HS_BK;
      /* rounding the slice length up to the K step can leave the last slices EMPTY (K = 10000 in 39 slices: 272 per slice, 37
       * slices).  The persistent kernel skips an empty item, so its slab kept whatever the workspace held and the reduction
       * added it (seen with two blocks sharing the workspace: stress seeds 1000 / 1010 of STRESS_BIG).  Only the slices that
       * exist are launched and reduced. */
      int nsl = (a->K + kchunk - 1) / kchunk;
      if ( nsl < 1 ) nsl = 1;
      eff.splitk = nsl;
      if ( nsl == 1 )
         eff.flags &= ~HS_GEMM_XCD;
   }
   a = &eff;

   if ( useBig )
   {
      /* the persistent LDS-DMA kernel (dgemm2.hip) takes the shapes it is eligible for: identical results */
      const int r2 = hs_dgemm2_try(stream, a, kchunk);
      if ( r2 < 0 )
         return -r2;
      if ( r2 == 0 )
      {
         HS_CALL( launch_lay<128>(stream, a, kchunk) );
         hs_mfma_flops_add(hs_gemm_executed_flops(a, 128, HS_BK, kchunk, 0));
      }
      else
         hs_mfma_flops_add(hs_gemm_executed_flops(a, 128, 8, kchunk, hs_dgemm2_slabskip()));
   }
   else
   {
      HS_CALL( launch_lay<64>(stream, a, kchunk) );
      hs_mfma_flops_add(hs_gemm_executed_flops(a, 64, HS_BK, kchunk, 0));
   }

   if ( a->splitk > 1 )
   {
      const long long total = (long long) a->M * a->N;
      int blocks = (int) ((total + 255) / 256);
      if ( blocks > 2048 ) blocks = 2048;
      hipLaunchKernelGGL(hs_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, a->M, a->N, a->splitk, a->ws,
         a->C, a->ldc, a->alpha, a->beta, (a->flags & HS_GEMM_LOWER) ? BT : ((a->flags & HS_GEMM_UPPER) ? -BT : 0));
      HS_HIP( hipGetLastError() );
   }
   return HS_OK;
}
