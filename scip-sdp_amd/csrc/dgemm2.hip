/* dgemm2.hip - persistent FP64 MFMA GEMM with LDS-DMA staging: the Schur-assembly shapes of schur.hip.
 *
 * Same product, same flags and the same summation order per output element as dgemm.hip (K ascending in steps of 4 inside
 * one accumulator chain, split-K slabs summed in slice order), so both kernels give identical bits.  What differs is how
 * the work reaches the matrix cores:
 *
 *  - PERSISTENT workgroups (2 per CU) walk a static list of work items (output tile x batch entry / K slice).  The items of
 *    one XCD (blockIdx % 8) are a contiguous range of the logical order and its workgroups take them interleaved, so
 *    neighbours in time share operand panels through that XCD's L2.  Triangular operands make the K range depend on the
 *    tile; the enumeration rotates the varying coordinate so that every workgroup sees all K lengths in turn.
 *  - operands go global -> LDS directly (global_load_lds_dwordx4, no staging registers, no ds_write pass) into a ring of
 *    NS slots of BKS = 8 K-steps; NS - 1 stages are in flight while one is multiplied, across item boundaries: the first
 *    stages of the next tile are on their way while the current tile is finished and stored, which is where the short-K
 *    products (K = 116 .. 500 per tile) lose time in the one-tile-per-workgroup kernel.
 *  - an LDS-DMA wave instruction writes 1 KiB contiguously (lane l -> base + 16 l); the per-lane SOURCE address is free.
 *    K-contiguous operand: a piece is 16 rows x 4 chunks (16 B) stored [chunk][row], so the 32 lanes of a ds_read_b64 half
 *    cover 256 contiguous bytes (conflict free).  Row-contiguous operand: a piece is one K row of 128 columns, chunk j of
 *    an odd row is stored at j ^ 8, which separates the two K rows a half-wave reads.
 *  - edge rows / columns are clamped to the last valid one (their products are not stored); K tails read from a 16-byte
 *    zero constant.  Requirements (else hs_dgemm uses dgemm.hip): A K-contiguous, even leading dimensions, strides, K and
 *    (row-contiguous B) N, 16-byte aligned operands.
 *  - completion of the DMA is tracked with counted s_waitcnt vmcnt (in-order return), raw s_barrier; the epilogue drains
 *    the counter once (vmcnt(0)) before its stores so that stores never sit between a DMA and the wait that retires it.
 */
#include "hs_common.h"
#include <stdlib.h>
#include <map>
#include <mutex>
#include <utility>

typedef double v4d2 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* hs_lds_ptr;
typedef const __attribute__((address_space(1))) void* hs_gbl_ptr;

#define G2_BT   128
#define G2_BKS  8
#ifndef G2_NS
#define G2_NS   4
#endif
#ifndef G2_WGPC
#define G2_WGPC 2                           /* persistent workgroups per CU (3 needs G2_NS <= 3: 48 KiB of LDS each, and <= 170 VGPRs) */
#endif
#define G2_OPSZ (G2_BT * G2_BKS)            /* doubles per operand per stage */
#define G2_SLOT (2 * G2_OPSZ)
#define G2_GPS  4                           /* LDS-DMA instructions per wave per stage: 2 per operand */
#ifndef G2_ABL
#define G2_ABL  0                           /* developer ablations (tests/devtools/gemm2_abl.sh), wrong results: 1 no C stores, 2 no DMA,
                                             * 4 no matrix instructions, 8 no drain of the DMA counter in front of the stores */
#endif

__device__ __attribute__((aligned(16))) double hs_g2_zero[2] = {0.0, 0.0};

struct g2_item
{
   int m0, n0, bz, ks0, kend;
};

/* position in the logical order -> item.  Order: batch entry / K slice major, then row tile, then column tile (lower:
 * packed lower-triangular tile index).  rot rotates the coordinate on which the K range depends. */
__device__ __forceinline__ void g2_decode(const hs_gemm_args& p, long long pos, long long ntile, int tm, int tn, int kchunk,
   int rotdiv, g2_item* it)
{
   int bz = (int) (pos / ntile);
   const long long t = pos - (long long) bz * ntile;
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
      if ( rotdiv > 0 )
      {
         /* the rotation advances by one per round of the XCD's workgroups (every workgroup sees all K lengths in turn) and differs
          * between the workgroups of a round by their index / (Wx / tiles): the two workgroups that share a CU (w and w + Wx / 2 in
          * dispatch order) then never walk items of the same length at the same time.  Without the second term they ran in lock
          * step, reached their epilogues together, and the matrix pipe idled while both stored (the 0.5 ms of stores of an
          * n^3 product of the C2 assembly added to its time instead of hiding behind the partner's matrix instructions). */
         if ( p.flags & HS_GEMM_B_LOWTRI )
         {
            const long long q = pos / tn;
            const long long qr = q / rotdiv;
            tj = (int) ((tj + qr + ((q - qr * rotdiv) * tn) / rotdiv) % tn);
         }
         else if ( p.flags & (HS_GEMM_A_LOWTRI | HS_GEMM_A_UPTRI) )
         {
            const int br = bz / rotdiv;
            ti = (ti + br + ((bz - br * rotdiv) * tm) / rotdiv) % tm;
         }
      }
   }
   it->m0 = ti * G2_BT;
   it->n0 = tj * G2_BT;
   it->bz = bz;
   int ks0 = 0, kend = p.K;
   if ( p.splitk > 1 )
   {
      ks0 = bz * kchunk;
      kend = min(p.K, ks0 + kchunk);
   }
   if ( p.flags & HS_GEMM_B_LOWTRI )
      ks0 = max(ks0, (it->n0 / 16) * 16);
   if ( p.flags & HS_GEMM_A_LOWTRI )
      kend = min(kend, it->m0 + G2_BT);
   if ( p.flags & HS_GEMM_A_UPTRI )
      ks0 = max(ks0, (it->m0 / 16) * 16);
   it->ks0 = ks0;
   it->kend = kend > ks0 ? kend : ks0;
}

template<int N> __device__ __forceinline__ void g2_wait_vm()
{
   asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

/* fragment: element (row woff + 16 t + (l & 15), k = 4 ks + (l >> 4)) of a stage image */
template<int LAY>
__device__ __forceinline__ double g2_frag(const double* __restrict__ slot, int woff, int t, int ks, int lane)
{
   const int k = 4 * ks + (lane >> 4);
   if ( LAY == HS_KC )
   {
      const int row = woff + 16 * t + (lane & 15);
      return slot[(row >> 4) * 128 + (k >> 1) * 32 + (row & 15) * 2 + (k & 1)];
   }
   else
   {
      const int col = woff + 16 * t + (lane & 15);
      return slot[k * 128 + (((col >> 1) ^ ((k & 1) << 3)) << 1) + (col & 1)];
   }
}

template<int LB, int IL>
__global__ void __launch_bounds__(256, G2_WGPC) hs_dgemm2_kernel(hs_gemm_args p, int kchunk, long long ntile, long long total, int rotdiv)
{
   extern __shared__ __attribute__((aligned(1024))) double g2_smem[];
   const int tid = threadIdx.x, lane = tid & 63;
   const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       /* wave-uniform: scalar branches below */
   const int wm = wave >> 1, wn = wave & 1;
   const int tm = (p.M + G2_BT - 1) / G2_BT, tn = (p.N + G2_BT - 1) / G2_BT;

   /* my items: XCD x owns [x * T8, (x + 1) * T8), its Wx workgroups take them interleaved */
   const int Wx = gridDim.x >> 3;
   const long long T8 = (total + 7) / 8;
   const long long base = (long long) (blockIdx.x & 7) * T8;
   long long lim = base + T8;
   if ( lim > total ) lim = total;
   const long long first = base + (blockIdx.x >> 3);

   /* ---- producer state: next stage to issue */
   long long ppos = first;
   bool pdone = ppos >= lim;
   const double* pa[2];
   const double* pb[2];
   int pk = 0, pkend = 0;              /* K position of the next stage of the producer's item, its K end */
   /* per-lane constants of the two pieces a wave loads per operand and stage */
   const int ar = lane & 15, ac = lane >> 4;                                   /* K-contiguous operand: row in piece, chunk */
   auto producer_load_item = [&]()
   {
      g2_item it;
      g2_decode(p, ppos, ntile, tm, tn, kchunk, rotdiv, &it);
      const double* A = p.A + (long long) (p.splitk > 1 ? 0 : it.bz) * p.strideA;
      const double* B = p.B + (long long) (p.splitk > 1 ? 0 : it.bz) * p.strideB;
#pragma unroll
      for (int i = 0; i < 2; ++i)
      {
         const int piece = wave * 2 + i;
         int row = it.m0 + piece * 16 + ar;
         if ( row > p.M - 1 ) row = p.M - 1;
         pa[i] = A + (long long) row * p.lda + it.ks0 + 2 * ac;
         if ( LB == HS_KC )
         {
            int rowb = it.n0 + piece * 16 + ar;
            if ( rowb > p.N - 1 ) rowb = p.N - 1;
            pb[i] = B + (long long) rowb * p.ldb + it.ks0 + 2 * ac;
         }
         else
         {
            const int kk = wave * 2 + i;
            int col = it.n0 + 2 * (lane ^ ((kk & 1) << 3));
            if ( col > p.N - 2 ) col = p.N - 2;
            pb[i] = B + (long long) (it.ks0 + kk) * p.ldb + col;
         }
      }
      pk = it.ks0;
      pkend = it.kend;
   };
   /* skip empty items (cannot happen for the shapes dispatched here, but an empty K range must not stall the ring) */
   auto producer_settle = [&]()
   {
      while ( !pdone )
      {
         producer_load_item();
         if ( pk < pkend )
            break;
         ppos += Wx;
         pdone = ppos >= lim;
      }
   };
   int gp = 0;                          /* stages issued */
   /* the four LDS-DMA pieces of the next stage WITHOUT any branch (IL = 0 instances: they sit between the matrix instructions of
    * the consumer's stage, one basic block); with nothing left to load the pieces read the zero constant - the number of operations
    * per stage stays four, which is what the counted waits assume */
   auto issue_loads = [&]() __attribute__((always_inline))
   {
      double* slot = g2_smem + (gp % G2_NS) * G2_SLOT;
      const bool live = !pdone;
#pragma unroll
      for (int i = 0; i < 2; ++i)
      {
         const int piece = wave * 2 + i;
         const double* src = (live && pk + 2 * ac < pkend) ? pa[i] : hs_g2_zero;
         if ( !(G2_ABL & 2) )
         __builtin_amdgcn_global_load_lds((hs_gbl_ptr) src, (hs_lds_ptr) (slot + piece * 128), 16, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
      {
         const int piece = wave * 2 + i;
         const double* src;
         if ( LB == HS_KC )
            src = (live && pk + 2 * ac < pkend) ? pb[i] : hs_g2_zero;
         else
            src = (live && pk + piece < pkend) ? pb[i] : hs_g2_zero;
         if ( !(G2_ABL & 2) )
         __builtin_amdgcn_global_load_lds((hs_gbl_ptr) src, (hs_lds_ptr) (slot + G2_OPSZ + piece * 128), 16, 0, 0);
      }
   };
   auto issue_advance = [&]() __attribute__((always_inline))
   {
      ++gp;
      if ( pdone )
         return;
#pragma unroll
      for (int i = 0; i < 2; ++i)
      {
         pa[i] += G2_BKS;
         pb[i] += (LB == HS_KC) ? (long long) G2_BKS : (long long) G2_BKS * p.ldb;
      }
      pk += G2_BKS;
      if ( pk >= pkend )
      {
         ppos += Wx;
         pdone = ppos >= lim;
         producer_settle();
      }
   };

   /* ---- consumer state */
   long long cpos = first;
   bool cdone = cpos >= lim;
   g2_item cit;
   int cleft = 0;                       /* stages left in the consumer's item */
   int ck = 0;                          /* K position of the stage being consumed */
   auto consumer_settle = [&]()
   {
      while ( !cdone )
      {
         g2_decode(p, cpos, ntile, tm, tn, kchunk, rotdiv, &cit);
         cleft = (cit.kend - cit.ks0 + G2_BKS - 1) / G2_BKS;
         ck = cit.ks0;
         if ( cleft > 0 )
            break;
         cpos += Wx;
         cdone = cpos >= lim;
      }
   };

   v4d2 acc[4][4];
#pragma unroll
   for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
         acc[i][j] = (v4d2){0.0, 0.0, 0.0, 0.0};
   /* Which 16-row / 16-column slabs of the tile the wavefront (wm, wn) owns.  Triangular operand: the slabs 2 i + wm and 2 j + wn
    * (interleaved): inside the diagonal band the nonzero slabs then split evenly between the two wavefronts of a pair, and a
    * band stage takes as long as its busiest wavefront - with contiguous halves one of them had all the work of the first half
    * of the band (-2.5 % on the two n^3 products of the assembly).  Otherwise the contiguous halves 4 wm + i, 4 wn + j (the
    * interleaved form costs the Gram product 5 %: its workgroups lose step with each other, and with that their hits in L2 -
    * 6.9 instead of 4.7 GB of HBM traffic per call; the same happened with the choice as a run-time value, hence IL). */
   constexpr int sw = IL ? 16 : 64;             /* offset of the wavefront's first slab */
   constexpr int ss = IL ? 2 : 1;               /* slab stride */

   producer_settle();
   consumer_settle();
   int gc = 0;                          /* stage being consumed */
   int landed = 0;                      /* stages [.., landed) are known complete (drained by an epilogue) */

   /* the item's tile: alpha acc (+ beta C) -> C or its split-K slab; the accumulators start the next item at zero */
   auto store_item = [&]()
   {
      if ( IL )
         asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      /* the last matrix instruction was issued as text (tri_mfma) */
      /* drain the DMA counter first, so that no store sits between a DMA and the wait that retires it */
      if ( !(G2_ABL & 8) )
      {
         g2_wait_vm<0>();
         landed = gp;
      }
      double* C = p.C;
      long long ldc = p.ldc;
      double alpha = p.alpha, beta = p.beta;
      if ( p.splitk > 1 )
      {
         C = p.ws + (long long) cit.bz * p.M * p.N;
         ldc = p.N;
         alpha = 1.0;
         beta = 0.0;
      }
      else
         C += (long long) cit.bz * p.strideC;
#pragma unroll
      for (int i = 0; i < 4; ++i)
      {
#pragma unroll
         for (int r = 0; r < 4; ++r)
         {
            const int row = cit.m0 + wm * sw + 16 * ss * i + (lane >> 4) + 4 * r;
#pragma unroll
            for (int j = 0; j < 4; ++j)
            {
               const int col = cit.n0 + wn * sw + 16 * ss * j + (lane & 15);
               if ( (!(G2_ABL & 1) || acc[i][j][r] == 1.2345e301) && row < p.M && col < p.N )
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
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
         for (int j = 0; j < 4; ++j)
            acc[i][j] = (v4d2){0.0, 0.0, 0.0, 0.0};
      cpos += Wx;
      cdone = cpos >= lim;
      consumer_settle();
   };

   /* ---- the stage loop.  A stage is two halves.  First half: the 16 matrix instructions of K step 0 with the fragment reads of
    * K step 1 between them; then the barrier that frees this stage's ring slot and publishes the next stage; second half: the 16
    * matrix instructions of K step 1 with the DMA of stage gc + NS (into the slot just freed) and the fragment reads of K step 0
    * of stage gc + 1 between them.  A v_mfma_f64_16x16x4 occupies the pipe for 64 cycles: what the wave issues meanwhile costs
    * nothing, whereas reads in front of the matrix instructions (the earlier form of this loop) were paid by every stage: - 10 %
    * on the Gram product.  Full operands (IL = 0): each half is ONE basic block whose instruction order is pinned by
    * scheduling groups.  Triangular operands (IL = 1): the matrix instructions of zero slabs inside the diagonal band are skipped
    * by wave-uniform branches, so the halves are many blocks; the reads and the DMA are issued in front of them and complete
    * behind them all the same.  (A third form - straight-line halves outside the band, branches inside - gave the register
    * allocator four versions of the accumulator code and 257 spilled registers.)  Same products in the same order per accumulator
    * as before. */
#pragma unroll
   for (int s = 0; s < G2_NS; ++s)
   {
      issue_loads();
      issue_advance();
   }
   double f0a[4], f0b[4], f1a[4], f1b[4];
   g2_wait_vm<(G2_NS - 1) * G2_GPS>();
   __builtin_amdgcn_s_barrier();
#pragma unroll
   for (int i = 0; i < 4; ++i)
   {
      f0a[i] = g2_frag<HS_KC>(g2_smem, wm * sw, ss * i, 0, lane);
      f0b[i] = g2_frag<LB>(g2_smem + G2_OPSZ, wn * sw, ss * i, 0, lane);
   }
   asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

   /* tell the compiler that the fragments have landed (the explicit s_waitcnt above it is invisible to its scoreboard: without
    * this every block that uses them starts with an lgkmcnt(0) of its own, which also waits for the reads issued since) */
   auto landed_frags = [&](double (&fa)[4], double (&fb)[4])
   {
#pragma unroll
      for (int i = 0; i < 4; ++i)
         asm volatile("" : "+v"(fa[i]), "+v"(fb[i]));
   };
   landed_frags(f0a, f0b);

   /* triangular operand (IL = 1: B lower triangular, IL = 2: A lower triangular): inside the diagonal band of the tile whole
    * 16 x 4 operand slabs are zero and their matrix instructions are skipped (adding a zero product changes nothing).  The
    * nonzero slabs of a wave are a prefix (B) / suffix (A) of its four, so the 16 instructions of a K step are laid out slab
    * by slab, four blocks of four behind one scalar branch each (a guard per instruction made every one of them a basic block
    * with a wait of its own; a switch that falls through into the blocks came back from the compiler's CFG structuriser with
    * copies of the accumulators and 100 spilled registers). */
   auto tri_mfma = [&](const double (&fa)[4], const double (&fb)[4], int kk, bool tri, bool issue)
   {
      /* the instruction as text with the accumulator tied to itself: through the builtin the joins of the skipped and the taken
       * paths became accumulate-into-another-register forms with copies, and 81 .. 107 spilled registers.  (The compiler's
       * hazard recogniser does not see an MFMA here: store_item() pads the one read-after-MFMA that could come too early.) */
#define G2_MFMA(i, j) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(fa[i]), "v"(fb[j]))
#define G2_ROW(i) { _Pragma("unroll") for (int j = 0; j < 4; ++j) G2_MFMA(i, j); }
#define G2_COL(j) { _Pragma("unroll") for (int i = 0; i < 4; ++i) G2_MFMA(i, j); }
      if ( IL == 1 )
      {
         /* B[k][n] = 0 for k < n: column slab c (columns from n0 + 16 c) is zero when kk + 3 < n0 + 16 c */
         int jlim = 4;
         if ( tri )
         {
            const int d = kk + 3 - cit.n0;
            const int qd = d >> 4;                       /* last nonzero slab (d >= 0) */
            jlim = (d < 0 || qd < wn) ? 0 : min(4, ((qd - wn) >> 1) + 1);
         }
         jlim = __builtin_amdgcn_readfirstlane(jlim);         /* uniform by construction; said so, the branches are scalar */
         if ( jlim >= 4 ) G2_COL(3)
         if ( issue )
            issue_loads();                 /* address arithmetic and DMA in the shadow of the block above (when it is not skipped) */
         if ( jlim >= 3 ) G2_COL(2)
         if ( jlim >= 2 ) G2_COL(1)
         if ( jlim >= 1 ) G2_COL(0)
      }
      else
      {
         /* A[m][k] = 0 for k > m: row slab r (rows from m0 + 16 r) is zero when kk > m0 + 16 r + 15 */
         int imin = 0;
         if ( tri )
         {
            const int e = kk - 15 - cit.m0;
            if ( e > 0 )
            {
               const int g = (e + 15) >> 4;              /* first nonzero slab */
               imin = g <= wm ? 0 : min(4, (g - wm + 1) >> 1);
            }
         }
         imin = __builtin_amdgcn_readfirstlane(imin);
         if ( imin <= 3 ) G2_ROW(3)
         if ( issue )
            issue_loads();
         if ( imin <= 2 ) G2_ROW(2)
         if ( imin <= 1 ) G2_ROW(1)
         if ( imin <= 0 ) G2_ROW(0)
      }
#undef G2_MFMA
#undef G2_ROW
#undef G2_COL
   };

   while ( !cdone )
   {
      const double* sa = g2_smem + (gc % G2_NS) * G2_SLOT;
      const double* sn = g2_smem + ((gc + 1) % G2_NS) * G2_SLOT;
      const bool tri = IL == 1 ? ck < cit.n0 + G2_BT : ck + G2_BKS > cit.m0;      /* stage inside the diagonal band */
      /* ---- first half */
      if ( !IL )
      {
         __builtin_amdgcn_sched_barrier(0);
#pragma unroll
         for (int i = 0; i < 4; ++i)
         {
            f1a[i] = g2_frag<HS_KC>(sa, wm * sw, ss * i, 1, lane);
            f1b[i] = g2_frag<LB>(sa + G2_OPSZ, wn * sw, ss * i, 1, lane);
         }
#pragma unroll
         for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
               acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(f0a[i], f0b[j], acc[i][j], 0, 0, 0);
#pragma unroll
         for (int q = 0; q < 8; ++q)
         {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
         }
         __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
         __builtin_amdgcn_sched_barrier(0);
      }
      else
      {
#pragma unroll
         for (int i = 0; i < 4; ++i)
         {
            f1a[i] = g2_frag<HS_KC>(sa, wm * sw, ss * i, 1, lane);
            f1b[i] = g2_frag<LB>(sa + G2_OPSZ, wn * sw, ss * i, 1, lane);
         }
         tri_mfma(f0a, f0b, ck, tri, false);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      landed_frags(f1a, f1b);
      /* stage gc + 1 must have landed; the NS - 2 stages after it may stay in flight (in-order return) */
      if ( gc + 1 >= landed )
         g2_wait_vm<(G2_NS - 2) * G2_GPS>();
      __builtin_amdgcn_s_barrier();
      /* ---- second half */
      if ( !IL )
      {
         __builtin_amdgcn_sched_barrier(0);
         issue_loads();
#pragma unroll
         for (int i = 0; i < 4; ++i)
         {
            f0a[i] = g2_frag<HS_KC>(sn, wm * sw, ss * i, 0, lane);
            f0b[i] = g2_frag<LB>(sn + G2_OPSZ, wn * sw, ss * i, 0, lane);
         }
#pragma unroll
         for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
               acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(f1a[i], f1b[j], acc[i][j], 0, 0, 0);
#pragma unroll
         for (int q = 0; q < 4; ++q)
         {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
         }
#pragma unroll
         for (int q = 0; q < 8; ++q)
         {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
         }
         __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
         __builtin_amdgcn_sched_barrier(0);
      }
      else
      {
#pragma unroll
         for (int i = 0; i < 4; ++i)
         {
            f0a[i] = g2_frag<HS_KC>(sn, wm * sw, ss * i, 0, lane);
            f0b[i] = g2_frag<LB>(sn + G2_OPSZ, wn * sw, ss * i, 0, lane);
         }
         tri_mfma(f1a, f1b, ck + 4, tri, true);
      }
      issue_advance();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      landed_frags(f0a, f0b);
      ck += G2_BKS;
      ++gc;
      if ( --cleft == 0 )
         store_item();
   }
   g2_wait_vm<0>();               /* a workgroup without items still has its prologue stages in flight */
}

/* ---- round 5: the two triangular n^3 products of the Schur assembly with a PAIRED diagonal band -------------------------------
 *
 * T = A_stack R (B lower triangular, TRI = 1) and W_j = G T_j (A lower triangular, TRI = 2).  In the 128 K positions where the
 * tile crosses the diagonal of the factor ("band": 16 stages of 8) half of the 16 x 4 operand slabs are zero.  The instances
 * <LB, 1> / <LB, 2> of the kernel above skip them behind wave-uniform branches; measured (rounds 3-4) that removes 18-21 % of
 * the matrix instructions and 3-6 % of the time: a band stage keeps its barrier, its DMA, its waits and its fragment reads,
 * the four wavefronts own different numbers of nonzero slabs, and 41 % (n = 500) / 23 % (n = 1000) of all stages are band
 * stages - the MFMA-busy gap between these products and the Gram product (66 against 85 %) is exactly that.
 *
 * Here the band costs what its nonzeros cost, with straight-line code:
 *  - the wavefronts split the dimension that is NOT triangular: wave w owns slabs 2 w, 2 w + 1 of it and ALL eight slabs of the
 *    triangular dimension (TRI = 1: 32 rows x 128 columns, TRI = 2: 128 rows x 32 columns), so every wavefront sees the same
 *    zero pattern;
 *  - band stage j (K positions 8 j .. 8 j + 7 behind the band start) has the nonzero slabs 0 .. j / 2 (TRI = 1, a prefix) or
 *    j / 2 .. 7 (TRI = 2, a suffix): j / 2 + 1 resp. 8 - j / 2 of them.  Stage d and stage 15 - d together have NINE, whatever
 *    d: the band is walked as 8 double stages (d, 15 - d), each two ring slots, each 2 K steps x 9 slabs x 2 = 36 matrix
 *    instructions per wavefront (a full stage: 32) - 288 for the band instead of 512, no branch, no imbalance;
 *  - the order of the K positions inside the band is therefore 0, 15, 1, 14, ...: results differ from hs_dgemm_kernel in the last
 *    bits (tests compare to 1e-13 of the row scale), the copies of an SPMD run still agree bit for bit.
 * Everything else as above: persistent workgroups, XCD-contiguous interleaved items, LDS-DMA ring of 4 slots, counted waits,
 * two halves per (double) stage with the next half's fragment reads and the ring's DMA between the matrix instructions. */
#define G5_NS 4
/* private bits of hs_gemm_args.flags, set by the launcher only */
#define G5_F_NT    (1 << 30)               /* result tiles are stored non-temporally */
#define G5_F_SETS  (1 << 29)               /* list order: g5_decode_sets */
#define G5_F_PAIRS (1 << 28)               /* list entries are taken two at a time */
#ifndef G5_DESC
#define G5_DESC 1
#endif
#ifndef G5_ABL
#define G5_ABL 0                            /* developer ablations (wrong results): 1 no C stores, 2 no DMA, 4 no matrix instructions */
#endif

/* developer instrumentation (make EXTRA=-DG5_PROF, tests/devtools/tri5_prof.py): per wavefront the cycles spent waiting for the DMA
 * counter, at the stage barriers, in the epilogues (drain + barrier + stores) and in total; s_memtime around the waits */
#ifdef G5_PROF
__device__ unsigned long long g5_prof[512 * 4 * 8];
#define G5_T(var) unsigned long long var = __builtin_readcyclecounter(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define G5_ACC(slot, a, b) prof_acc[slot] += (b) - (a)
#else
#define G5_T(var)
#define G5_ACC(slot, a, b)
#endif

struct g5_frag
{
   double o[2];          /* the wavefront's two slabs of the other dimension */
   double t[8];          /* slabs of the triangular dimension */
};

/* position idx of an XCD's list (Tx entries, absolute positions base .. base + Tx of the natural order: batch entry, row tile,
 * column tile) -> item.  The workgroups of an XCD TAKE the entries one by one (atomic counter), so whoever is free gets the
 * next one; the last entries of the list are re-ordered by decreasing K length (TRI = 1: column tile ascending, TRI = 2: row
 * tile descending), so that the items still open when the list runs out are the short ones: the statically interleaved lists
 * of the kernel above left the slowest workgroup 11 % behind the average on these products (items of 116 .. 500 K positions,
 * 30 per workgroup).  Returns false past the end of the list. */
template<int TRI>
__device__ __forceinline__ bool g5_decode(const hs_gemm_args& p, long long idx, long long base, long long Tx, int tm, int tn,
   long long ntile, int tailwant, int* om0, int* on0, int* obz)
{
   const bool ok = idx < Tx;
   if ( !ok )
      idx = 0;
   long long pos = base + idx;
   {
      const long long G = TRI == 1 ? (long long) tn : ntile;
      long long ts = base + Tx - tailwant;
      if ( ts < base ) ts = base;
      ts = ((ts + G - 1) / G) * G;
      const long long te = ((base + Tx) / G) * G;
      if ( pos >= ts && pos < te )
      {
         const int ng = (int) ((te - ts) / G);
         const int q = (int) (pos - ts);
         if ( TRI == 1 )
         {
            const int cl = q / ng, grp = q - cl * ng;
            pos = ts + (long long) grp * tn + cl;
         }
         else
         {
            const int per = ng * tn;
            const int cl = q / per, rem = q - cl * per;
            const int grp = rem / tn, w = rem - grp * tn;
            pos = ts + (long long) grp * ntile + (long long) (tm - 1 - cl) * tn + w;
         }
      }
   }
   const long long bz = pos / ntile;
   const int t = (int) (pos - bz * ntile);
   const int ti = t / tn, tj = t - ti * tn;
   *om0 = ok ? ti * G2_BT : 0;
   *on0 = ok ? tj * G2_BT : 0;
   *obz = ok ? (int) bz : 0;
   return ok;
}

/* The same list in SETS (G5_F_SETS): the small triangular operand (n x n: R or G) is read by every item and the large one is
 * streamed, so what the 64 workgroups of an XCD hold in its 4 MB L2 decides how often either is fetched again.  nv panels of the
 * small operand (column tiles for TRI = 1, row tiles for TRI = 2), ranked by K length (rank 0 the longest): set k = the ranks
 * {2k, nv - 1 - 2k, 2k + 1, nv - 2 - 2k} - two long/short pairs, at most about half of the small operand (2.2 of 4.5 MB at
 * n = 1000).  The XCD's list (whole groups of the natural order: Tx is a multiple of G) is walked set by set; inside a set the
 * four items that share a panel of the streamed operand (same row tile / same batch entry and column tile) are consecutive
 * entries, taken by four workgroups within a few microseconds.  The last entries of the list by decreasing K length as above. */
template<int TRI>
__device__ __forceinline__ bool g5_decode_sets(long long idx, long long base, long long Tx, int tm, int tn, long long ntile, int tailwant,
   int* om0, int* on0, int* obz)
{
   const bool ok = idx < Tx;
   if ( !ok )
      idx = 0;
   const int nv = TRI == 1 ? tn : tm;
   long long U = TRI == 1 ? Tx / tn : (Tx / ntile) * tn;             /* panels of the streamed operand in this list */
   if ( U < 1 )
      U = 1;                    /* (an XCD whose share is empty: nothing below is used, but nothing divides by zero either) */
   const int nset = (nv + 3) >> 2;
   const int szl = nv - 4 * (nset - 1);                             /* members of the last set */
   long long ngt = tailwant / szl;
   if ( ngt > U ) ngt = U;
   const long long tstart = Tx - ngt * szl;
   long long g;
   int r;
   if ( idx >= tstart )
   {
      /* tail: class by class (rank ascending = K length descending) over the last ngt panels of the last set */
      const long long q = idx - tstart;
      const int cl = (int) (q / ngt);
      g = U - ngt + (q - (long long) cl * ngt);
      const int k = nset - 1;
      r = cl < ((szl + 1) >> 1) ? 2 * k + cl : nv - 1 - 2 * k - (szl - 1 - cl);
   }
   else
   {
      const int k = (int) (idx / (4 * U));
      const long long rem = idx - 4 * U * k;
      const int sz = nv - 4 * k < 4 ? nv - 4 * k : 4;
      g = rem / sz;
      const int j = (int) (rem - g * sz);
      const int q = 2 * k + (j >> 1);
      r = (j & 1) ? nv - 1 - q : q;
   }
   long long pos;
   if ( TRI == 1 )
      pos = base + g * tn + r;
   else
   {
      const long long bl = g / tn;
      const int tj = (int) (g - bl * tn);
      pos = base + bl * ntile + (long long) (nv - 1 - r) * tn + tj;
   }
   const long long bz = pos / ntile;
   const int t = (int) (pos - bz * ntile);
   const int ti = t / tn, tj = t - ti * tn;
   *om0 = ok ? ti * G2_BT : 0;
   *on0 = ok ? tj * G2_BT : 0;
   *obz = ok ? (int) bz : 0;
   return ok;
}

#define G5_UNI(x) __builtin_amdgcn_readfirstlane(x)        /* wave-uniform by construction: keep it in a scalar register */

template<int LB, int TRI>
__global__ void __launch_bounds__(256, 2) hs_dgemm5_kernel(hs_gemm_args p, long long ntile, long long total, unsigned int* __restrict__ ctr)
{
   extern __shared__ __attribute__((aligned(1024))) double g2_smem[];
   __shared__ unsigned int g5_q[4];        /* list entries the workgroup has taken: entry of item g in slot g & 3 */
   const int tid = threadIdx.x, lane = tid & 63;
   const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
   const int tm = (p.M + G2_BT - 1) / G2_BT, tn = (p.N + G2_BT - 1) / G2_BT;
   constexpr int NI = TRI == 1 ? 2 : 8, NJ = TRI == 1 ? 8 : 2;

   const int xcd = blockIdx.x & 7;
   const int Wx = gridDim.x >> 3;
   const bool sets = (p.flags & G5_F_SETS) != 0;
   long long T8 = (total + 7) / 8;
   if ( sets )
   {
      const long long G = TRI == 1 ? (long long) tn : ntile;       /* whole groups per XCD (total is a multiple of G) */
      T8 = ((T8 + G - 1) / G) * G;
   }
   const long long base = (long long) xcd * T8;
   const long long Tx = base >= total ? 0 : (base + T8 > total ? total - base : T8);
   const int tailwant = 3 * Wx;

   /* ---- producer: the item whose stages are being requested (cur) and the one after it (nxt, decoded ahead so that the switch
    * inside a stage block is a handful of selects) */
   const int ar = lane & 15, ac = lane >> 4;
   const double* pa[2]; const double* pb[2];
   const double* na[2]; const double* nb[2];
   int pkb = 0, pkend = -0x40000000, pnfull = 0, pnst = 0x40000000, pidx = 0;
   int nkb = 0, nkend = -0x40000000, nnfull = 0, nnst = 0x40000000;
   int nm0 = 0, nn0 = 0, nbz = 0;         /* the item in nxt, as the consumer will need it */
   bool nvalid = false;
   pa[0] = pa[1] = pb[0] = pb[1] = na[0] = na[1] = nb[0] = nb[1] = hs_g2_zero;
   /* list entry idx -> nxt (one decode per item: the consumer takes its copy from here too) */
   auto decode_next = [&](long long idx) __attribute__((always_inline))
   {
      int dm0 = 0, dn0 = 0, dbz = 0;
      /* (past the end of the list: item (0, 0, 0) with an empty K range - every request then reads the zero constant - and a stage
       * count that is never reached; no early return: the compiler then keeps the state behind it in scratch memory) */
      nvalid = sets ? g5_decode_sets<TRI>(idx, base, Tx, tm, tn, ntile, tailwant, &dm0, &dn0, &dbz)
         : g5_decode<TRI>(p, idx, base, Tx, tm, tn, ntile, tailwant, &dm0, &dn0, &dbz);
      nm0 = G5_UNI(dm0); nn0 = G5_UNI(dn0); nbz = G5_UNI(dbz);
      const double* A = p.A + (long long) nbz * p.strideA;
      const double* B = p.B + (long long) nbz * p.strideB;
#pragma unroll
      for (int i = 0; i < 2; ++i)
      {
         const int piece = wave * 2 + i;
         int row = nm0 + piece * 16 + ar;
         if ( row > p.M - 1 ) row = p.M - 1;
         na[i] = A + (long long) row * p.lda + 2 * ac;
         if ( LB == HS_KC )
         {
            int rowb = nn0 + piece * 16 + ar;
            if ( rowb > p.N - 1 ) rowb = p.N - 1;
            nb[i] = B + (long long) rowb * p.ldb + 2 * ac;
         }
         else
         {
            int col = nn0 + 2 * (lane ^ ((piece & 1) << 3));
            if ( col > p.N - 2 ) col = p.N - 2;
            nb[i] = B + (long long) piece * p.ldb + col;
         }
      }
      if ( TRI == 1 )
      {
         nkb = nn0;
         nkend = p.K;
         nnfull = nkend > nkb + G2_BT ? (nkend - nkb - G2_BT + G2_BKS - 1) / G2_BKS : 0;
      }
      else
      {
         nkb = nm0;
         nkend = min(p.K, nm0 + G2_BT);
         nnfull = nm0 / G2_BKS;
      }
      nnfull = G5_UNI(nnfull);
      nnst = nvalid ? nnfull + 16 : 0x40000000;
      nkend = nvalid ? nkend : -0x40000000;
   };
   int gp = 0;                            /* stage loads requested so far */
   int vzero;                             /* 0 in a vector register the compiler cannot see through: keeps wave-uniform tail tests of the
                                           * requests as selects (a scalar branch would cut the stage block in two) */
   asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
   /* one stage (8 K positions of both operands) into a ring slot: four LDS-DMA pieces per wavefront, no branch (stage_begin,
    * stage_piece(0 .. 3) - the callers place the pieces between their matrix instructions -, stage_end: the step to the next stage,
    * and to the next item when this one has all its stages) */
   double* rq_slot = g2_smem;
   int rq_K0 = 0;
   auto stage_begin = [&](int slotidx) __attribute__((always_inline))
   {
      rq_slot = g2_smem + slotidx * G2_SLOT;
      /* both kinds: the full stages first, then the band.  TRI = 1 walks its full stages from the END of the K range down to the
       * band: the column tiles of a row panel, taken by four workgroups of the XCD within a few microseconds of each other, then
       * read the same columns of A at the same time (G5_DESC=0: ascending from the band's end) */
      const int j = pidx - pnfull;
      const int pj = (j & 1) ? 15 - (j >> 1) : (j >> 1);
      if ( TRI == 1 )
         rq_K0 = j < 0 ? pkb + G2_BT + 8 * (G5_DESC ? pnfull - 1 - pidx : pidx) : pkb + 8 * pj;
      else
         rq_K0 = j < 0 ? 8 * pidx : pkb + 8 * pj;
   };
   auto stage_piece = [&](int k) __attribute__((always_inline))
   {
      const int i = k & 1;
      const int piece = wave * 2 + i;
      const double* src;
      if ( k < 2 )
         src = (rq_K0 + 2 * ac < pkend) ? pa[i] + rq_K0 : hs_g2_zero;
      else if ( LB == HS_KC )
         src = (rq_K0 + 2 * ac < pkend) ? pb[i] + rq_K0 : hs_g2_zero;
      else
         src = (rq_K0 + piece + vzero < pkend) ? pb[i] + (long long) rq_K0 * p.ldb : hs_g2_zero;      /* (vzero: a select, not a scalar branch) */
      if ( !(G5_ABL & 2) )
         __builtin_amdgcn_global_load_lds((hs_gbl_ptr) src, (hs_lds_ptr) (rq_slot + (k < 2 ? 0 : G2_OPSZ) + piece * 128), 16, 0, 0);
   };
   auto stage_end = [&]() __attribute__((always_inline))
   {
      ++gp;
      ++pidx;
      const bool sw = pidx >= pnst;
      pidx = G5_UNI(sw ? 0 : pidx);
      pkb = G5_UNI(sw ? nkb : pkb);
      pkend = G5_UNI(sw ? nkend : pkend);
      pnfull = G5_UNI(sw ? nnfull : pnfull);
      pnst = G5_UNI(sw ? nnst : pnst);
#pragma unroll
      for (int i = 0; i < 2; ++i)
      {
         pa[i] = sw ? na[i] : pa[i];
         pb[i] = sw ? nb[i] : pb[i];
      }
   };
   auto issue_stage = [&](int slotidx) __attribute__((always_inline))
   {
      stage_begin(slotidx);
#pragma unroll
      for (int k = 0; k < 4; ++k)
         stage_piece(k);
      stage_end();
   };

   /* ---- consumer */
   int cm0 = 0, cn0 = 0, cbz = 0;
   bool cvalid = false;
   int cnfull = 0;
   int gi = 0;                            /* number of the consumer's item among those this workgroup has taken */
   const bool pairs = (p.flags & G5_F_PAIRS) != 0;      /* list entries are taken two at a time (a long and a short item: equal work) */
   unsigned int lastq = 0;

   v4d2 acc[NI][NJ];
#pragma unroll
   for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
         acc[i][j] = (v4d2){0.0, 0.0, 0.0, 0.0};

#ifdef G5_PROF
   unsigned long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
   const unsigned long long prof_t0 = __builtin_readcyclecounter();
#endif
   int gc = 0;                            /* stage loads consumed so far (ring slot = index & 3) */
   int landed = 0;                        /* stage loads [.., landed) are known complete and visible to all wavefronts */

   /* fragments of one K step (ks = 0, 1) of a stage image: the wavefront's two slabs of the other dimension and the slabs
    * [t0, t1] of the triangular one */
   auto read_frags = [&](g5_frag& f, const double* slot, int ks, int t0, int t1) __attribute__((always_inline))
   {
      if ( TRI == 1 )
      {
#pragma unroll
         for (int i = 0; i < 2; ++i)
            f.o[i] = g2_frag<HS_KC>(slot, 0, 2 * wave + i, ks, lane);
#pragma unroll
         for (int c = 0; c < 8; ++c)
            if ( c >= t0 && c <= t1 )
               f.t[c] = g2_frag<LB>(slot + G2_OPSZ, 0, c, ks, lane);
      }
      else
      {
#pragma unroll
         for (int r = 0; r < 8; ++r)
            if ( r >= t0 && r <= t1 )
               f.t[r] = g2_frag<HS_KC>(slot, 0, r, ks, lane);
#pragma unroll
         for (int j = 0; j < 2; ++j)
            f.o[j] = g2_frag<LB>(slot + G2_OPSZ, 0, 2 * wave + j, ks, lane);
      }
   };
   /* tell the compiler the fragments have landed (see landed_frags of the kernel above) */
   auto frags_landed = [&](g5_frag& f, int t0, int t1) __attribute__((always_inline))
   {
#pragma unroll
      for (int i = 0; i < 2; ++i)
         asm volatile("" : "+v"(f.o[i]));
#pragma unroll
      for (int c = 0; c < 8; ++c)
         if ( c >= t0 && c <= t1 )
            asm volatile("" : "+v"(f.t[c]));
   };
   /* the matrix instructions of one K step over the slabs [t0, t1]; rev: highest slab first */
   auto mfma_frags = [&](const g5_frag& f, int t0, int t1, bool rev) __attribute__((always_inline))
   {
#pragma unroll
      for (int cc = 0; cc < 8; ++cc)
      {
         const int c = rev ? 7 - cc : cc;
         if ( c >= t0 && c <= t1 )
         {
#pragma unroll
            for (int i = 0; i < 2; ++i)
            {
               if ( G5_ABL & 4 )
                  asm volatile("" :: "v"(f.o[i]), "v"(f.t[c]));
               else if ( TRI == 1 )
                  acc[i][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.o[i], f.t[c], acc[i][c], 0, 0, 0);
               else
                  acc[c][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.t[c], f.o[i], acc[c][i], 0, 0, 0);
            }
         }
      }
   };

   /* matrix instruction number k of a K step over the slabs [t0, t1] (k = 0 .. 2 (t1 - t0 + 1) - 1; rev: highest slab first) */
   auto mfma_one = [&](const g5_frag& f, int t0, int t1, bool rev, int k) __attribute__((always_inline))
   {
      const int c = rev ? t1 - (k >> 1) : t0 + (k >> 1);
      const int i = k & 1;
      if ( G5_ABL & 4 )
         asm volatile("" :: "v"(f.o[i]), "v"(f.t[c]));
      else if ( TRI == 1 )
         acc[i][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.o[i], f.t[c], acc[i][c], 0, 0, 0);
      else
         acc[c][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.t[c], f.o[i], acc[c][i], 0, 0, 0);
   };

   auto store_item = [&]() __attribute__((always_inline))
   {
      G5_T(te0);
      /* The next item may start with a double stage: its two slots (requests gc, gc + 1; two more are in flight behind them) must have
       * landed for every wavefront.  NO drain of the DMA counter in front of the stores (rounds 1-4 drained: a bubble of one memory
       * latency per item): a counted wait stays correct with stores in flight - loads return in order among themselves, so "at most
       * as many operations outstanding as loads were issued behind L" still proves that L has landed; stores that are still on
       * their way only make such a wait longer. */
      if ( gc + 1 >= landed )
         g2_wait_vm<2 * G2_GPS>();
      landed = gc + 2 > landed ? gc + 2 : landed;
      __builtin_amdgcn_s_barrier();
      double* C = p.C + (long long) cbz * p.strideC;
      const long long ldc = p.ldc;
      const double alpha = p.alpha, beta = p.beta;
      const int row0 = cm0 + (TRI == 1 ? 32 * wave : 0) + (lane >> 4);
      const int col0 = cn0 + (TRI == 1 ? 0 : 32 * wave) + (lane & 15);
      if ( (G5_ABL & 1) && acc[0][0][0] != 1.2345e301 )
      {
      }
      else if ( cm0 + G2_BT <= p.M && cn0 + G2_BT <= p.N && beta == 0.0 )
      {
         /* interior tile, nothing to add to: one address per row of four-row groups, the columns as immediate offsets */
         double* c0 = C + (long long) row0 * ldc + col0;
         if ( p.flags & G5_F_NT )
         {
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
               for (int r = 0; r < 4; ++r)
               {
                  double* cr = c0 + (long long) (16 * i + 4 * r) * ldc;
#pragma unroll
                  for (int j = 0; j < NJ; ++j)
                     __builtin_nontemporal_store(alpha * acc[i][j][r], cr + 16 * j);
               }
         }
         else if ( alpha == 1.0 )
         {
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
               for (int r = 0; r < 4; ++r)
               {
                  double* cr = c0 + (long long) (16 * i + 4 * r) * ldc;
#pragma unroll
                  for (int j = 0; j < NJ; ++j)
                     cr[16 * j] = acc[i][j][r];
               }
         }
         else
         {
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
               for (int r = 0; r < 4; ++r)
               {
                  double* cr = c0 + (long long) (16 * i + 4 * r) * ldc;
#pragma unroll
                  for (int j = 0; j < NJ; ++j)
                     cr[16 * j] = alpha * acc[i][j][r];
               }
         }
      }
      else
      {
#pragma unroll
         for (int i = 0; i < NI; ++i)
         {
#pragma unroll
            for (int r = 0; r < 4; ++r)
            {
               const int row = row0 + 16 * i + 4 * r;
#pragma unroll
               for (int j = 0; j < NJ; ++j)
               {
                  const int col = col0 + 16 * j;
                  if ( row < p.M && col < p.N )
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
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
         for (int j = 0; j < NJ; ++j)
            acc[i][j] = (v4d2){0.0, 0.0, 0.0, 0.0};
      G5_T(te1);
      G5_ACC(6, te0, te1);
   };

   /* ---- full stages: nf of them, ring slots gc, gc + 1, ... (the first one landed and published) */
   auto run_full = [&](int nf) __attribute__((always_inline))
   {
      if ( nf <= 0 )
         return;
      g5_frag f0, f1;
      read_frags(f0, g2_smem + (gc & 3) * G2_SLOT, 0, 0, 7);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      frags_landed(f0, 0, 7);
      for (int s = 0; s < nf; ++s)
      {
         const double* sa = g2_smem + (gc & 3) * G2_SLOT;
         const double* sn = g2_smem + ((gc + 1) & 3) * G2_SLOT;
         /* first half: K step 0, the fragments of K step 1 between its matrix instructions */
         __builtin_amdgcn_sched_barrier(0);
         read_frags(f1, sa, 1, 0, 7);
         mfma_frags(f0, 0, 7, false);
#pragma unroll
         for (int q = 0; q < 10; ++q)
         {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
         }
         __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
         __builtin_amdgcn_sched_barrier(0);
         asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
         frags_landed(f1, 0, 7);
         /* the next step's slots must have landed: one (a full stage follows; two later requests may stay in flight), or two
          * (the last full stage: a double stage or the end of the item follows) */
         G5_T(ta);
         if ( s + 1 < nf )
         {
            if ( gc + 1 >= landed )
               g2_wait_vm<2 * G2_GPS>();
         }
         else if ( gc + 2 >= landed )
            g2_wait_vm<1 * G2_GPS>();
         G5_T(tb);
         __builtin_amdgcn_s_barrier();
         G5_T(tc);
         G5_ACC(0, ta, tb); G5_ACC(1, tb, tc); G5_ACC(4, 0, 1);
         /* second half: K step 1; the DMA of the stage four ahead into the slot just freed, and the first fragments of the next
          * stage (read in vain behind the last full stage: the slot exists, the values are not used) */
         /* (placed by hand: two matrix instructions, then one request per matrix instruction - left to the scheduler the four
          * requests with their address arithmetic came in front of the first matrix instruction) */
         __builtin_amdgcn_sched_barrier(0);
         stage_begin(gc & 3);
         mfma_one(f1, 0, 7, false, 0);
         mfma_one(f1, 0, 7, false, 1);
#pragma unroll
         for (int k = 0; k < 4; ++k)
         {
            __builtin_amdgcn_sched_barrier(0);
            stage_piece(k);
            mfma_one(f1, 0, 7, false, 2 + k);
         }
         __builtin_amdgcn_sched_barrier(0);
         stage_end();
         read_frags(f0, sn, 0, 0, 7);
#pragma unroll
         for (int k = 6; k < 16; ++k)
            mfma_one(f1, 0, 7, false, k);
#pragma unroll
         for (int q = 0; q < 10; ++q)
         {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
         }
         __builtin_amdgcn_sched_barrier(0);
         asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
         frags_landed(f0, 0, 7);
         ++gc;
      }
   };

   /* ---- the band: 8 double stages (ring slots gc + 2 d, gc + 2 d + 1 = band stages d and 15 - d), straight-line.  Slab ranges
    * of the triangular dimension: stage d has [0, q] (TRI = 1) / [q, 7] (TRI = 2) with q = d / 2, stage 15 - d has [0, 7 - q] /
    * [7 - q, 7]. */
   auto run_band = [&]() __attribute__((always_inline))
   {
      g5_frag l0, h0, l1, h1;
      unsigned int taken = 0;
      {
         const double* slo = g2_smem + (gc & 3) * G2_SLOT;
         const double* shi = g2_smem + ((gc + 1) & 3) * G2_SLOT;
         read_frags(l0, slo, 0, 0, TRI == 1 ? 0 : 7);
         read_frags(h0, shi, 0, TRI == 1 ? 0 : 7, 7);
         asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
         frags_landed(l0, 0, TRI == 1 ? 0 : 7);
         frags_landed(h0, TRI == 1 ? 0 : 7, 7);
      }
#pragma unroll
      for (int d = 0; d < 8; ++d)
      {
         const int q = d >> 1, qn = (d + 1) >> 1;
         const int lt0 = TRI == 1 ? 0 : q, lt1 = TRI == 1 ? q : 7;                 /* slabs of stage d */
         const int ht0 = TRI == 1 ? 0 : 7 - q, ht1 = TRI == 1 ? 7 - q : 7;         /* slabs of stage 15 - d */
         const int nlt0 = TRI == 1 ? 0 : qn, nlt1 = TRI == 1 ? qn : 7;
         const int nht0 = TRI == 1 ? 0 : 7 - qn, nht1 = TRI == 1 ? 7 - qn : 7;
         const double* slo = g2_smem + (gc & 3) * G2_SLOT;
         const double* shi = g2_smem + ((gc + 1) & 3) * G2_SLOT;
         const double* nlo = g2_smem + ((gc + 2) & 3) * G2_SLOT;
         const double* nhi = g2_smem + ((gc + 3) & 3) * G2_SLOT;
         /* first half: K step 0 of both stages (18 matrix instructions), the 13 fragments of K step 1 between them */
         __builtin_amdgcn_sched_barrier(0);
         read_frags(l1, slo, 1, lt0, lt1);
         read_frags(h1, shi, 1, ht0, ht1);
         mfma_frags(l0, lt0, lt1, false);
         mfma_frags(h0, ht0, ht1, true);
#pragma unroll
         for (int g = 0; g < 13; ++g)
         {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
         }
         __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
         __builtin_amdgcn_sched_barrier(0);
         asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
         frags_landed(l1, lt0, lt1);
         frags_landed(h1, ht0, ht1);
         /* the next double stage's two slots (everything requested so far); behind the last one a single slot */
         G5_T(ta);
         if ( d == 1 )
         {
            /* (everything requested so far, and the atomic of the previous double stage) */
            g2_wait_vm<0>();
            if ( wave == 0 && lane == 0 )
            {
               g5_q[(gi + 2) & 3] = taken;
               lastq = taken;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
         }
         else if ( d < 7 )
         {
            if ( gc + 3 >= landed )
               g2_wait_vm<0>();
         }
         else if ( gc + 2 >= landed )
            g2_wait_vm<1 * G2_GPS>();
         G5_T(tb);
         __builtin_amdgcn_s_barrier();
         G5_T(tc);
         G5_ACC(2, ta, tb); G5_ACC(3, tb, tc); G5_ACC(5, 0, 1);
         /* second half: K step 1; both freed slots are requested again, the next double stage's first fragments are read */
         {
            /* 18 matrix instructions: those of stage d first (nlo of them), then stage 15 - d from its highest slab down */
            const int cntlo = 2 * (lt1 - lt0 + 1);
            auto bm = [&](int k) __attribute__((always_inline))
            {
               if ( k < cntlo )
                  mfma_one(l1, lt0, lt1, false, k);
               else
                  mfma_one(h1, ht0, ht1, true, k - cntlo);
            };
            __builtin_amdgcn_sched_barrier(0);
            stage_begin(gc & 3);
            bm(0);
            bm(1);
#pragma unroll
            for (int k = 0; k < 4; ++k)
            {
               __builtin_amdgcn_sched_barrier(0);
               stage_piece(k);
               bm(2 + k);
            }
            __builtin_amdgcn_sched_barrier(0);
            stage_end();
            stage_begin((gc + 1) & 3);
#pragma unroll
            for (int k = 0; k < 4; ++k)
            {
               __builtin_amdgcn_sched_barrier(0);
               stage_piece(k);
               bm(6 + k);
            }
            __builtin_amdgcn_sched_barrier(0);
            stage_end();
            /* one thread takes the list entry of the item after the next; the value is collected at the next double stage's wait */
            if ( d == 0 && wave == 0 && lane == 0 )
            {
               if ( !pairs )
                  taken = atomicAdd(ctr + xcd, 1u);
               else if ( (gi & 1) == 0 )
                  taken = atomicAdd(ctr + xcd, 2u);
               else
                  taken = lastq + 1;
            }
            if ( d < 7 )
            {
               read_frags(l0, nlo, 0, nlt0, nlt1);
               read_frags(h0, nhi, 0, nht0, nht1);
            }
#pragma unroll
            for (int k = 10; k < 18; ++k)
               bm(k);
            if ( d < 7 )
            {
#pragma unroll
               for (int g = 0; g < 7; ++g)
               {
                  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                  __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
               }
               __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
         }
         if ( d < 7 )
         {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            frags_landed(l0, nlt0, nlt1);
            frags_landed(h0, nht0, nht1);
         }
         gc += 2;
      }
   };

   /* ---- prologue: the workgroup takes its first two list entries; the first item into cur, four stages requested */
   if ( tid == 0 )
   {
      const unsigned int e0 = atomicAdd(ctr + xcd, pairs ? 2u : 1u);
      const unsigned int e1 = pairs ? e0 + 1 : atomicAdd(ctr + xcd, 1u);
      g5_q[0] = e0;
      g5_q[1] = e1;
   }
   __syncthreads();
   decode_next(__builtin_amdgcn_readfirstlane((int) g5_q[0]));
   pidx = 0; pkb = nkb; pkend = nkend; pnfull = nnfull; pnst = nnst;
   pa[0] = na[0]; pa[1] = na[1]; pb[0] = nb[0]; pb[1] = nb[1];
   cm0 = nm0; cn0 = nn0; cbz = nbz; cvalid = nvalid;
#pragma unroll
   for (int s = 0; s < G5_NS; ++s)
      issue_stage(s);
   g2_wait_vm<0>();
   landed = gp;
   __builtin_amdgcn_s_barrier();

   while ( cvalid )
   {
      /* the item after this one into nxt (the requests for it start four stages before this one ends) */
      decode_next(__builtin_amdgcn_readfirstlane((int) g5_q[(gi + 1) & 3]));
      if ( TRI == 1 )
         cnfull = p.K > cn0 + G2_BT ? (p.K - cn0 - G2_BT + G2_BKS - 1) / G2_BKS : 0;
      else
         cnfull = cm0 / G2_BKS;
      run_full(cnfull);
      run_band();
      store_item();
      cm0 = nm0; cn0 = nn0; cbz = nbz; cvalid = nvalid;
      ++gi;
   }
   g2_wait_vm<0>();
#ifdef G5_PROF
   if ( lane == 0 )
   {
      unsigned long long* o = g5_prof + ((long long) blockIdx.x * 4 + wave) * 8;
      for (int i = 0; i < 7; ++i)
         o[i] = prof_acc[i];
      o[7] = __builtin_readcyclecounter() - prof_t0;
   }
#endif
}

#ifdef G5_PROF
extern "C" __attribute__((visibility("default"))) int hs_dgemm5_prof_read(unsigned long long* out)
{
   return hipMemcpyFromSymbol(out, HIP_SYMBOL(g5_prof), sizeof(unsigned long long) * 512 * 4 * 8) == hipSuccess ? 0 : 1;
}
#endif

/* 1: launched, 0: not eligible (caller uses dgemm.hip), < 0: error code negated */
static int g2_disabled = -1;
static int g2_taken = 0;

/* test hook: on = 0 forces dgemm.hip for every product, on = 1 restores the default; returns how many products the
 * persistent kernel has taken so far */
int hs_dgemm2_enable(int on)
{
   g2_disabled = on ? 0 : 1;
   return __atomic_load_n(&g2_taken, __ATOMIC_RELAXED);
}

/* 1: products with a triangular operand skip the zero slabs inside the diagonal band (instance IL = 1); 0 (HIPSDP_GEMM2_SKIP=0):
 * they run the straight-line instance over the same K ranges - more matrix instructions, no branches between them */
int hs_dgemm2_slabskip(void)
{
   static int skip = -1;
   if ( skip < 0 )
   {
      const char* env = getenv("HIPSDP_GEMM2_SKIP");
      skip = (env != NULL && env[0] == '0') ? 0 : 1;
      /* round 5: products with ONE triangular operand and no K slices take the paired-band kernel (hs_dgemm5_kernel) unless
       * HIPSDP_GEMM_TRI=1 (the skipping instances of round 3) or HIPSDP_GEMM2_SKIP=0 (no skipping at all) says otherwise */
      const char* tri = getenv("HIPSDP_GEMM_TRI");
      if ( skip == 1 && !(tri != NULL && tri[0] == '1') )
         skip = 2;
   }
   return skip;
}

/* 16 counters per (device, stream): launches on one stream are ordered, so they can share them (cleared before each launch);
 * never freed (64 bytes per stream the process has used for these products) */
static unsigned int* g5_counters(hipStream_t stream)
{
   static std::mutex mu;
   static std::map<std::pair<int, hipStream_t>, unsigned int*> tab;
   int dev = 0;
   if ( hipGetDevice(&dev) != hipSuccess )
      return NULL;
   std::lock_guard<std::mutex> lk(mu);
   auto it = tab.find(std::make_pair(dev, stream));
   if ( it != tab.end() )
      return it->second;
   unsigned int* q = NULL;
   if ( hipMalloc((void**) &q, 64) != hipSuccess )
      return NULL;
   tab[std::make_pair(dev, stream)] = q;
   return q;
}

/* products the paired-band kernel takes (given that the persistent kernels are eligible at all): exactly one triangular operand
 * - the right factor row-contiguous, as R in A_stack R -, the whole K range per item, full output */
int hs_dgemm2_tri5_eligible(const hs_gemm_args* a)
{
   const bool triA = (a->flags & HS_GEMM_A_LOWTRI) != 0, triB = (a->flags & HS_GEMM_B_LOWTRI) != 0;
   return hs_dgemm2_slabskip() == 2 && triA != triB && a->splitk <= 1 && !(a->flags & (HS_GEMM_LOWER | HS_GEMM_UPPER | HS_GEMM_A_UPTRI))
      && !(triB && a->layB != HS_MC);
}

int hs_dgemm2_try(hipStream_t stream, const hs_gemm_args* a, int kchunk)
{
   if ( g2_disabled < 0 )
   {
      const char* env = getenv("HIPSDP_GEMM_V1");
      g2_disabled = (env != NULL && env[0] == '1') ? 1 : 0;
   }
   if ( g2_disabled )
      return 0;
   {
      /* diagnostic: HIPSDP_GEMM_V1_MASK routes classes of products to the tile kernel (1: B triangular, 2: A triangular, 4: lower
       * tiles / Gram, 8: all others, 16: batched, 32: split-K) */
      static int mask = -1;
      if ( mask < 0 )
         mask = getenv("HIPSDP_GEMM_V1_MASK") != NULL ? atoi(getenv("HIPSDP_GEMM_V1_MASK")) : 0;
      if ( mask != 0 )
      {
         int cls = 0;
         if ( a->flags & HS_GEMM_B_LOWTRI ) cls |= 1;
         if ( a->flags & HS_GEMM_A_LOWTRI ) cls |= 2;
         if ( a->flags & HS_GEMM_LOWER ) cls |= 4;
         if ( cls == 0 ) cls = 8;
         if ( a->batch > 1 ) cls |= 16;
         if ( a->splitk > 1 ) cls |= 32;
         if ( cls & mask )
            return 0;
      }
   }
   if ( a->layA != HS_KC || (a->flags & (HS_GEMM_UPPER)) )
      return 0;
   if ( (a->lda & 1) || (a->ldb & 1) || (a->strideA & 1) || (a->strideB & 1) || (a->K & 1) || a->K < 16 )
      return 0;
   if ( (((uintptr_t) a->A) & 15) || (((uintptr_t) a->B) & 15) )
      return 0;
   if ( a->layB == HS_MC && ((a->N & 1) || a->N < 2) )
      return 0;
   if ( (a->flags & HS_GEMM_LOWER) && a->M != a->N )
      return 0;
   if ( a->splitk > 1 && (kchunk & 15) )
      return 0;
   const long long tm = (a->M + G2_BT - 1) / G2_BT, tn = (a->N + G2_BT - 1) / G2_BT;
   const long long ntile = (a->flags & HS_GEMM_LOWER) ? tm * (tm + 1) / 2 : tm * tn;
   const long long nz = a->splitk > 1 ? a->splitk : a->batch;
   const long long total = ntile * nz;
   if ( total < 384 )
      return 0;
   int grid = 256 * G2_WGPC;           /* G2_WGPC workgroups per CU, 32 G2_WGPC per XCD */
   const int Wx = grid / 8;
   /* rotation of the coordinate the K range depends on (see g2_decode): only when the interleave would pin it */
   int rotdiv = 0;
   if ( !(a->flags & HS_GEMM_LOWER) )
   {
      if ( (a->flags & HS_GEMM_B_LOWTRI) && tn > 1 && (Wx % tn) == 0 )
         rotdiv = (int) (Wx / tn);
      else if ( (a->flags & (HS_GEMM_A_LOWTRI | HS_GEMM_A_UPTRI)) && tm > 1 && (Wx % (tm * tn)) == 0 )
         rotdiv = (int) (Wx / (tm * tn));
   }
   const size_t smem = (size_t) G2_NS * G2_SLOT * sizeof(double);
   /* triangular operand: the instances with interleaved slab ownership that skip the zero slabs of the diagonal band (IL = 1: B,
    * IL = 2: A; both triangular - not a product of this library - runs without skipping, which is always correct) */
   const bool triA = (a->flags & HS_GEMM_A_LOWTRI) != 0, triB = (a->flags & HS_GEMM_B_LOWTRI) != 0;
   if ( hs_dgemm2_tri5_eligible(a) )
   {
      /* the paired-band kernel: one triangular operand, whole K range per item */
      static hs_attr_mask attr5[3];
      const int inst5 = triB ? 0 : (a->layB == HS_MC ? 1 : 2);
      const size_t smem5 = (size_t) G5_NS * G2_SLOT * sizeof(double);
      const void* fn5 = NULL;
      switch ( inst5 )
      {
      case 0: fn5 = reinterpret_cast<const void*>(&hs_dgemm5_kernel<HS_MC, 1>); break;
      case 1: fn5 = reinterpret_cast<const void*>(&hs_dgemm5_kernel<HS_MC, 2>); break;
      default: fn5 = reinterpret_cast<const void*>(&hs_dgemm5_kernel<HS_KC, 2>); break;
      }
      if ( hs_func_max_lds(fn5, (int) smem5, &attr5[inst5]) != HS_OK )
         return -HS_ERR_HIP;
      /* the eight list counters of the launch: cleared in stream order in front of it */
      unsigned int* ctr = g5_counters(stream);
      if ( ctr == NULL || hipMemsetAsync(ctr, 0, 64, stream) != hipSuccess )
         return -HS_ERR_HIP;
      /* list order and store policy (HIPSDP_GEMM_ORDER=0: the natural order, entries taken one by one, plain stores - round 5's
       * first form): entries in sets of the small operand's panels, taken two at a time, result tiles stored non-temporally.  The
       * same products per tile either way: only which workgroup computes a tile, and when, changes */
      hs_gemm_args a5 = *a;
      {
         static const int order = getenv("HIPSDP_GEMM_ORDER") != NULL ? atoi(getenv("HIPSDP_GEMM_ORDER")) : 1;
         if ( order )
            a5.flags |= G5_F_NT | G5_F_SETS | G5_F_PAIRS;
      }
      switch ( inst5 )
      {
      case 0: hipLaunchKernelGGL((hs_dgemm5_kernel<HS_MC, 1>), dim3(grid), dim3(256), smem5, stream, a5, ntile, total, ctr); break;
      case 1: hipLaunchKernelGGL((hs_dgemm5_kernel<HS_MC, 2>), dim3(grid), dim3(256), smem5, stream, a5, ntile, total, ctr); break;
      default: hipLaunchKernelGGL((hs_dgemm5_kernel<HS_KC, 2>), dim3(grid), dim3(256), smem5, stream, a5, ntile, total, ctr); break;
      }
      if ( hipGetLastError() != hipSuccess )
         return -HS_ERR_HIP;
      (void) __atomic_add_fetch(&g2_taken, 1, __ATOMIC_RELAXED);
      return 1;
   }
   const int il = (hs_dgemm2_slabskip() && triA != triB) ? (triB ? 1 : 2) : 0;
   static hs_attr_mask attr_done[6];
   const int inst = (a->layB == HS_KC ? 0 : 3) + il;
   const void* fn = NULL;
   switch ( inst )
   {
   case 0: fn = reinterpret_cast<const void*>(&hs_dgemm2_kernel<HS_KC, 0>); break;
   case 1: fn = reinterpret_cast<const void*>(&hs_dgemm2_kernel<HS_KC, 1>); break;
   case 2: fn = reinterpret_cast<const void*>(&hs_dgemm2_kernel<HS_KC, 2>); break;
   case 3: fn = reinterpret_cast<const void*>(&hs_dgemm2_kernel<HS_MC, 0>); break;
   case 4: fn = reinterpret_cast<const void*>(&hs_dgemm2_kernel<HS_MC, 1>); break;
   default: fn = reinterpret_cast<const void*>(&hs_dgemm2_kernel<HS_MC, 2>); break;
   }
   if ( hs_func_max_lds(fn, (int) smem, &attr_done[inst]) != HS_OK )
      return -HS_ERR_HIP;
   switch ( inst )
   {
   case 0: hipLaunchKernelGGL((hs_dgemm2_kernel<HS_KC, 0>), dim3(grid), dim3(256), smem, stream, *a, kchunk, ntile, total, rotdiv); break;
   case 1: hipLaunchKernelGGL((hs_dgemm2_kernel<HS_KC, 1>), dim3(grid), dim3(256), smem, stream, *a, kchunk, ntile, total, rotdiv); break;
   case 2: hipLaunchKernelGGL((hs_dgemm2_kernel<HS_KC, 2>), dim3(grid), dim3(256), smem, stream, *a, kchunk, ntile, total, rotdiv); break;
   case 3: hipLaunchKernelGGL((hs_dgemm2_kernel<HS_MC, 0>), dim3(grid), dim3(256), smem, stream, *a, kchunk, ntile, total, rotdiv); break;
   case 4: hipLaunchKernelGGL((hs_dgemm2_kernel<HS_MC, 1>), dim3(grid), dim3(256), smem, stream, *a, kchunk, ntile, total, rotdiv); break;
   default: hipLaunchKernelGGL((hs_dgemm2_kernel<HS_MC, 2>), dim3(grid), dim3(256), smem, stream, *a, kchunk, ntile, total, rotdiv); break;
   }
   if ( hipGetLastError() != hipSuccess )
      return -HS_ERR_HIP;
   (void) __atomic_add_fetch(&g2_taken, 1, __ATOMIC_RELAXED);
   return 1;
}
