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
   }
   return skip;
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
