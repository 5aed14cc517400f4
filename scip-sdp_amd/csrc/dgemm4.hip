/* dgemm4.hip - strip GEMM on FP64 MFMA for the two n^3 products of the Schur assembly (schur.hip), both in the form
 *
 *      C_j = L B_j,    j = 0 .. m1 - 1  (batched),   L a shared TRIANGULAR n x n left factor (K contiguous), B_j full n x n:
 *
 *      T_j^T = R^T A_j      R^T upper triangular  [HS_GEMM_A_UPTRI]: K range of row strip m0 starts at m0;   B_j = A_j as stored
 *      W_j   = G T_j        G   lower triangular  [HS_GEMM_A_LOWTRI]: K range ends at m0 + 64;  B_j = T_j given as T_j^T, i.e. K contiguous
 *
 * (A_j is symmetric, so T_j^T = (A_j R)^T = R^T A_j: the first product is formed transposed, which puts the triangular factor
 * on the left in both.)  Same flags and the same bits as dgemm.hip / dgemm2.hip: every output element is one accumulator chain
 * over K ascending in groups of 4 aligned at multiples of 4; products a triangular factor makes zero are exact zeros whether
 * issued or not.  What differs from dgemm2.hip is the shape of the work and who hides what:
 *
 *  - ONE workgroup per CU, one wavefront per SIMD (the whole 512-entry register file of a lane: 256 accumulator registers).  A
 *    workgroup owns a STRIP of the output: 64 rows x up to 512 columns (n = 500: all columns); a wavefront all 64 rows and every
 *    fourth 16-wide column slab.  The triangular factor only shortens the K range of a strip - every stage of 8 K steps is the same
 *    straight-line block of 64 matrix instructions per wavefront, no branch, no per-tile predicate, and the four wavefronts always
 *    have the same amount of work (a 128 x 128 tile's diagonal band took as long as its busiest wavefront).  Inside the last
 *    (first) 64 K steps of a strip 6 of 16 row slabs multiply zeros: 9 % of its matrix instructions.
 *  - operands go global -> LDS by LDS-DMA into a ring of 4 slots (36 KB each: 64 x 8 of L in [chunk][row] pieces, 8 x 512 of B_j
 *    as one K row of 128 columns per piece with the odd rows' 16-byte chunks at j ^ 8, or - K contiguous B_j - as [chunk][row]
 *    pieces again: conflict-free ds_read_b64 either way).  The nine DMA instructions a wavefront issues per stage (for the stage
 *    three ahead) and the fragment reads of the next K step sit BETWEEN the matrix instructions of the block (scheduling groups:
 *    2 matrix instructions, 1 read or 1 piece): a v_mfma_f64_16x16x4 keeps the matrix pipe busy for 64 cycles and the wave issues
 *    its next instructions meanwhile, so neither the DMA issue nor the LDS latency is exposed.  A stage needs one s_barrier; what it
 *    makes visible is the NEXT stage's data, so a stage's first fragments are in registers before its barrier.
 *  - every stage issues all nine pieces (one that is not needed - columns beyond N, K rows behind the end, nothing left to load -
 *    reads a 16-byte zero constant): s_waitcnt vmcnt(9) is the whole bookkeeping.  The epilogue drains the counter before its
 *    stores (straight from the accumulator registers; alpha = 1, beta = 0 only) and the stages issued until then need no wait.
 */
#include "hs_common.h"
#include <stdlib.h>
#include <type_traits>

typedef double v4d4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* g4_lds_ptr;
typedef const __attribute__((address_space(1))) void* g4_gbl_ptr;

#define G4_BM    64                         /* rows of a strip */
#define G4_BKS   8                          /* K steps per stage */
#define G4_NS    4                          /* ring slots */
#define G4_NW    8                          /* column slabs per wavefront */
#define G4_BN    (64 * G4_NW)               /* columns of a strip: 512 */
#define G4_NG    (G4_BN / 128)              /* 128-column groups (one LDS-DMA piece per K row and group) */
#define G4_ASZ   (G4_BM * G4_BKS)           /* doubles of the A part of a slot */
#define G4_BSZ   (G4_BKS * G4_BN)           /* doubles of the B part */
#define G4_SLOT  (G4_ASZ + G4_BSZ)
#define G4_NPIECE (1 + 2 * G4_NG)           /* LDS-DMA pieces a wavefront issues per stage at most: 1 of A, 2 K rows x groups of B */

__device__ __attribute__((aligned(16))) double hs_g4_zero[2] = {0.0, 0.0};

struct g4_item
{
   int m0, n0, bz, ks0, kend;
};

/* column slab t (0 .. 7) of wavefront w: t-th slab of the boustrophedon walk over the slabs of the strip */
__device__ __forceinline__ int g4_slab(int t, int w)
{
   return 4 * t + ((t & 1) ? 3 - w : w);
}

/* what the kernel needs of hs_gemm_args (kept small: every field lives in a scalar register for the whole kernel) */
struct g4_params
{
   const double* A; const double* B; double* C;
   long long strideA, strideB, strideC;
   int M, N, K, lda, ldb, ldc, tm, tn, total;
};

/* TRI 2: A lower triangular (A[m][k] = 0 for k > m: the K range of a strip ends at m0 + 64), 3: A upper triangular (A[m][k] = 0 for
 * k < m: it starts at m0).  LB: storage of the B operand (HS_MC: [K][N] rows, HS_KC: [N][K]). */
template<int TRI, int LB>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
hs_dgemm4_kernel(g4_params p)
{
   extern __shared__ __attribute__((aligned(1024))) double g4_smem[];
   const int tid = threadIdx.x, lane = tid & 63;
   const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
   const int tm = p.tm, tn = p.tn, N = p.N;

   /* my items: the workgroups of an XCD (blockIdx % 8) take a contiguous range of the logical order interleaved */
   const int Wx = gridDim.x >> 3;
   const int T8 = (p.total + 7) / 8;
   const int base = (int) (blockIdx.x & 7) * T8;
   const int lim = min(base + T8, p.total);
   const int first = base + (int) (blockIdx.x >> 3);

   /* logical order: batch entry major, then row strip, then column tile; the row strip is rotated from round to round so that a
    * workgroup sees all K lengths in turn */
   auto decode = [&](int pos, g4_item* it) __attribute__((always_inline))
   {
      const int per = tm * tn;
      const int bz = pos / per;
      const int o = pos - bz * per;
      int ti = o / tn;
      const int tj = o - ti * tn;
      ti = (ti + (bz * per) / Wx) % tm;                /* rotation, constant over the strips of a batch entry: a bijection */
      it->m0 = ti * G4_BM;
      it->n0 = tj * G4_BN;
      it->bz = bz;
      it->ks0 = TRI == 3 ? it->m0 : 0;
      it->kend = TRI == 2 ? min(p.K, it->m0 + G4_BM) : p.K;
   };

   /* ---- producer state: the stage that is issued next.  Piece 0: the wavefront's 16 rows of A.  B as rows of K (HS_MC): piece
    * 1 + 2 g + r = K row 2 wave + r, column group g (128 columns); B K contiguous (HS_KC): piece 1 + q = columns 128 wave + 16 q
    * .. + 15, all 8 K steps.  Addresses: a uniform base plus a 32-bit byte offset per lane. ---------------------------------- */
   int ppos = first;
   bool pdone = ppos >= lim;
   bool pneed = !pdone;                 /* the next item has to be decoded */
   int pk = 0, pkend = 0, pn0 = 0;      /* K position of the stage being issued, K end and first column of its item */
   int gp = 0;                          /* stages issued (real or empty) */
   const char* pabase = (const char*) hs_g4_zero;   /* uniform: A at (first row of this wavefront's piece, K position of the stage) */
   const char* pb0 = (const char*) hs_g4_zero;      /* uniform.  HS_MC: B at K rows 2 wave / 2 wave + 1 of the stage, column 0; HS_KC: B at */
   const char* pb1 = (const char*) hs_g4_zero;      /* (column n0 + 128 wave, K position of the stage) / unused */
   unsigned paoff = 0;                  /* per lane: byte offset of (row lane & 15, chunk lane >> 4) */
   unsigned pcol0 = 0, pcol1 = 0;       /* HS_MC, per lane: column (of the matrix) this lane loads from those rows in group 0 (swizzled chunk) */
   unsigned pboff[LB == HS_KC ? 8 : 1]; /* HS_KC, per lane: byte offset of (column 16 q + (lane & 15), chunk lane >> 4) behind pb0 */
#pragma unroll
   for (int q = 0; q < (LB == HS_KC ? 8 : 1); ++q)
      pboff[q] = 0;
   const int ar = lane & 15, ac = lane >> 4;
   const unsigned lds0 = (unsigned) (uintptr_t) (g4_lds_ptr) g4_smem;
   auto producer_next_item = [&]() __attribute__((always_inline))
   {
      while ( pneed )
      {
         if ( ppos >= lim )
         {
            pdone = true;
            pneed = false;
            break;
         }
         g4_item it;
         decode(ppos, &it);
         if ( it.kend > it.ks0 )
         {
            const int r0 = it.m0 + wave * 16;
            pabase = (const char*) (p.A + (long long) it.bz * p.strideA + (long long) r0 * p.lda + it.ks0);
            paoff = (unsigned) (min(ar, max(p.M - 1 - r0, 0)) * p.lda + 2 * ac) * 8u;        /* rows beyond M: the last one again */
            if ( LB == HS_MC )
            {
               pb0 = (const char*) (p.B + (long long) it.bz * p.strideB + (long long) (it.ks0 + 2 * wave) * p.ldb);
               pb1 = pb0 + (long long) p.ldb * 8;
               pcol0 = (unsigned) (it.n0 + 2 * lane);               /* K row 2 wave is even: no swizzle */
               pcol1 = (unsigned) (it.n0 + 2 * (lane ^ 8));         /* odd row: chunk j sits at position j ^ 8 */
            }
            else
            {
               const int c0 = it.n0 + 128 * wave;                   /* first column of this wavefront's eight pieces */
               pb0 = (const char*) (p.B + (long long) it.bz * p.strideB + (long long) min(c0, N - 1) * p.ldb + it.ks0);
#pragma unroll
               for (int q = 0; q < 8; ++q)
                  pboff[q] = (unsigned) (min(16 * q + ar, max(N - 1 - c0, 0)) * p.ldb + 2 * ac) * 8u;   /* columns beyond N: the last one again */
            }
            pk = it.ks0;
            pkend = it.kend;
            pn0 = it.n0;
            pneed = false;
         }
         else
            ppos += Wx;
      }
   };
   /* all nine pieces of the next stage, branch-free: a piece that is not needed reads the zero constant (uniform base, offset 0) */
   auto issue_stage = [&]() __attribute__((always_inline))
   {
      const unsigned pslot = lds0 + (unsigned) ((gp & (G4_NS - 1)) * (G4_SLOT * 8));
      const bool live = !pdone;
      const char* const zero = (const char*) hs_g4_zero;
      const bool inK = live && pk + 2 * ac < pkend;                 /* per lane: this lane's 16-byte chunk lies inside the K range */
      {
         const char* src = inK ? pabase + paoff : zero;
         __builtin_amdgcn_global_load_lds((g4_gbl_ptr) src, (g4_lds_ptr) (uintptr_t) (pslot + (unsigned) (wave * 1024)), 16, 0, 0);
      }
      if ( LB == HS_MC )
      {
         const int ng = min(G4_NG, (N - pn0 + 127) >> 7);          /* column groups that contain a column of the matrix */
#pragma unroll
         for (int j = 1; j < G4_NPIECE; ++j)
         {
            const int g = (j - 1) >> 1, r = (j - 1) & 1;
            const int kr = 2 * wave + r;
            const unsigned sel = (live && g < ng && pk + kr < pkend) ? 1u : 0u;         /* uniform */
            const unsigned col = min((r ? pcol1 : pcol0) + 128u * g, (unsigned) (N - 2));
            const uintptr_t bb = (uintptr_t) (r ? pb1 : pb0), zz = (uintptr_t) zero;
            const uintptr_t base_ = zz + (bb - zz) * (uintptr_t) sel;                    /* scalar arithmetic, no branch */
            const char* src = (const char*) base_ + ((col * 8u) & (0u - sel));
            __builtin_amdgcn_global_load_lds((g4_gbl_ptr) src, (g4_lds_ptr) (uintptr_t) (pslot + (unsigned) ((G4_ASZ + (kr * G4_NG + g) * 128) * 8)), 16, 0, 0);
         }
      }
      else
      {
#pragma unroll
         for (int q = 0; q < 8; ++q)
         {
            /* piece = column slab 8 wave + q of the strip; slabs that start beyond N are never stored: zeros */
            const bool have = pn0 + 128 * wave + 16 * q < N;        /* uniform */
            const char* src = (inK && have) ? pb0 + pboff[q] : zero;
            __builtin_amdgcn_global_load_lds((g4_gbl_ptr) src, (g4_lds_ptr) (uintptr_t) (pslot + (unsigned) ((G4_ASZ + (8 * wave + q) * 128) * 8)), 16, 0, 0);
         }
      }
   };
   auto producer_stage_done = [&]() __attribute__((always_inline))
   {
      ++gp;
      if ( pdone )
         return;
      pk += G4_BKS;
      pabase += G4_BKS * 8;
      if ( LB == HS_MC )
      {
         pb0 += (long long) G4_BKS * 8 * p.ldb;
         pb1 += (long long) G4_BKS * 8 * p.ldb;
      }
      else
         pb0 += G4_BKS * 8;
      if ( pk >= pkend )
      {
         ppos += Wx;
         pneed = true;
      }
   };

   /* ---- consumer ------------------------------------------------------------------------------------------------------- */
   int cpos = first;
   bool cdone = cpos >= lim;
   g4_item cit = {0, 0, 0, 0, 0};
   int cleft = 0;
   auto consumer_settle = [&]() __attribute__((always_inline))
   {
      while ( !cdone )
      {
         decode(cpos, &cit);
         cleft = (cit.kend - cit.ks0 + G4_BKS - 1) / G4_BKS;
         if ( cleft > 0 )
            break;
         cpos += Wx;
         cdone = cpos >= lim;
      }
   };

   v4d4 acc[4][G4_NW];
#pragma unroll
   for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int t = 0; t < G4_NW; ++t)
         acc[i][t] = (v4d4){0.0, 0.0, 0.0, 0.0};

   /* fragment positions inside a slot (doubles).  [chunk][row] pieces (A; B when K contiguous): element (row 16 i + (l & 15),
    * k = 4 ks + (l >> 4)) at i * 128 + (k >> 1) * 32 + (l & 15) * 2 + (k & 1).  B as rows of K: element (k, column c of the strip),
    * group g = c >> 7, chunk j = (c & 127) >> 1, at ASZ + (k * NG + g) * 128 + ((j ^ ((k & 1) << 3)) << 1) + (c & 1) */
   const int kq = lane >> 4;                                           /* k within a group of 4 */
   const int aoff = (kq >> 1) * 32 + (lane & 15) * 2 + (kq & 1);       /* + i * 128 + ks * 64 */
   int boff[G4_NW];                                                    /* + ks * (HS_MC: 4 * NG * 128, HS_KC: 64) */
#pragma unroll
   for (int t = 0; t < G4_NW; ++t)
   {
      const int sl = g4_slab(t, wave);
      if ( LB == HS_MC )
      {
         const int c = 16 * sl + (lane & 15);
         const int g = c >> 7, j = (c & 127) >> 1;
         boff[t] = G4_ASZ + (kq * G4_NG + g) * 128 + ((j ^ ((kq & 1) << 3)) << 1) + (c & 1);
      }
      else
         boff[t] = G4_ASZ + sl * 128 + aoff;
   }
   double fa[2][4], fb[2][G4_NW];
   auto read_frags = [&](int stage, int ks, int buf) __attribute__((always_inline))
   {
      const double* sl = g4_smem + (stage & (G4_NS - 1)) * G4_SLOT;
#pragma unroll
      for (int i = 0; i < 4; ++i)
         fa[buf][i] = sl[i * 128 + ks * 64 + aoff];
#pragma unroll
      for (int t = 0; t < G4_NW; ++t)
         fb[buf][t] = sl[ks * (LB == HS_MC ? 4 * G4_NG * 128 : 64) + boff[t]];
   };

   int gc = 0;                                      /* stage being consumed */
   int landed = 0;                                  /* stages [.., landed) are known resident (drained by an epilogue) */
   /* one stage: a single basic block */
   auto stage_body = [&]() __attribute__((always_inline))
   {
      /* K step 0 (its fragments are in registers); meanwhile the fragments of K step 1 and the pieces of the stage three ahead */
      read_frags(gc, 1, 1);
      issue_stage();
#pragma unroll
      for (int t = 0; t < G4_NW; ++t)
#pragma unroll
         for (int i = 0; i < 4; ++i)
            acc[i][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[0][i], fb[0][t], acc[i][t], 0, 0, 0);
      /* K step 1; meanwhile the first fragments of the next stage (resident: see the loop) */
      read_frags(gc + 1, 0, 0);
#pragma unroll
      for (int t = 0; t < G4_NW; ++t)
#pragma unroll
         for (int i = 0; i < 4; ++i)
            acc[i][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[1][i], fb[1][t], acc[i][t], 0, 0, 0);
      /* the order inside this one basic block: a matrix instruction keeps its pipe busy for 64 cycles, whatever else the wave issues
       * meanwhile costs nothing - so reads and DMA pieces go BETWEEN matrix instructions, one at a time.  K step 0: 12 x (2 matrix
       * instructions, 1 fragment read of K step 1), then 4 x (2 matrix instructions, 1 DMA piece); K step 1: 12 x (2, 1 read of the
       * next stage), then 4 x (2, 1 piece); the ninth piece goes first. */
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      for (int h = 0; h < 2; ++h)
      {
         for (int q = 0; q < 12; ++q)
         {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
         }
         for (int q = 0; q < 4; ++q)
         {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
         }
      }
   };

   consumer_settle();
   /* stages 0, 1, 2 */
   for (int s = 0; s < G4_NS - 1; ++s)
   {
      producer_next_item();
      issue_stage();
      producer_stage_done();
   }
   if ( !cdone )
   {
      /* stages 0 and 1 resident and visible, first fragments of stage 0 in registers */
      asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      read_frags(0, 0, 0);
   }
   while ( !cdone )
   {
      /* stage gc + 1 must be resident before the barrier (its first fragments are read during stage gc): of this wavefront's
       * pieces only the nine of stage gc + 2 may still be in flight.  Not behind an epilogue (it drained the counter: the stages
       * issued until then are resident) - its stores are still on their way and would be waited for */
      if ( gc + 2 > landed )
         asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      /* the slot of stage gc - 1 is free: stage gc + 3 is issued during this stage */
      producer_next_item();
      stage_body();
      producer_stage_done();
      ++gc;
      if ( --cleft == 0 )
      {
         /* epilogue.  Its stores count in vmcnt like the DMA pieces: the counter is drained first, so that no store sits between a
          * piece and the wait that retires it (the pieces in flight are those of the next strip's first stages) */
         asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
         landed = gp;
         double* C = p.C + (long long) cit.bz * p.strideC + (long long) (cit.m0 + (lane >> 4)) * p.ldc + cit.n0 + (lane & 15);
         const int rows = p.M - cit.m0 - (lane >> 4);              /* row 16 i + 4 r of this lane exists iff 16 i + 4 r < rows */
         /* alpha = 1, beta = 0 (hs_dgemm4_try takes nothing else): the accumulators are stored as they are, straight from the
          * accumulator registers; slab by slab */
         const bool rowfull = cit.m0 + G4_BM <= p.M;               /* uniform: all 64 rows of the strip exist */
#pragma unroll
         for (int t = 0; t < G4_NW; ++t)
         {
            const int sl = 16 * g4_slab(t, wave);
            double* ct = C + sl;
            if ( rowfull && cit.n0 + sl + 16 <= N )
            {
#pragma unroll
               for (int i = 0; i < 4; ++i)
#pragma unroll
                  for (int r = 0; r < 4; ++r)
                     ct[(long long) (16 * i + 4 * r) * p.ldc] = acc[i][t][r];
            }
            else
            {
               const bool colok = cit.n0 + sl + (lane & 15) < N;
#pragma unroll
               for (int i = 0; i < 4; ++i)
#pragma unroll
                  for (int r = 0; r < 4; ++r)
                     if ( colok && 16 * i + 4 * r < rows )
                        ct[(long long) (16 * i + 4 * r) * p.ldc] = acc[i][t][r];
            }
            __builtin_amdgcn_sched_barrier(0);
         }
#pragma unroll
         for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < G4_NW; ++t)
               acc[i][t] = (v4d4){0.0, 0.0, 0.0, 0.0};
         cpos += Wx;
         cdone = cpos >= lim;
         consumer_settle();
      }
   }
   /* the empty stages issued behind the last real one */
   asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

/* matrix-core flops one launch issues (hs_common.h: hs_mfma_flops_total): whole stages of 64 rows x 512 columns x 8 K steps over the K
 * range of every strip */
static double g4_executed_flops(const hs_gemm_args* a, int tri)
{
   const int tm = (a->M + G4_BM - 1) / G4_BM, tn = (a->N + G4_BN - 1) / G4_BN;
   long long stages = 0;
   for (int ti = 0; ti < tm; ++ti)
   {
      const int m0 = ti * G4_BM;
      const int ks0 = tri == 3 ? m0 : 0;
      int kend = a->K;
      if ( tri == 2 && kend > m0 + G4_BM ) kend = m0 + G4_BM;
      if ( kend > ks0 )
         stages += (kend - ks0 + G4_BKS - 1) / G4_BKS;
   }
   return 2.0 * G4_BM * G4_BN * G4_BKS * (double) stages * (double) tn * (double) a->batch;
}

static int g4_mode = -1;        /* -1: read HIPSDP_GEMM4 on first use; 0 off; 1 on */
static long long g4_taken = 0;

double hs_dgemm4_taken(void)
{
   return (double) __atomic_load_n(&g4_taken, __ATOMIC_RELAXED);
}

int hs_dgemm4_enable(int on)
{
   const int before = g4_mode;
   g4_mode = on ? 1 : 0;
   return before;
}

/* 1: launched, 0: not eligible (caller goes on to dgemm2.hip / dgemm.hip), < 0: error code negated */
int hs_dgemm4_try(hipStream_t stream, const hs_gemm_args* a)
{
   if ( g4_mode < 0 )
   {
      /* Off unless asked for (HIPSDP_GEMM4=1).  Measured at n = 500, m = 1000 inside the solve (profiles/r03_a_*): the same time per
       * product as the persistent tile kernel (3.13 against 3.10 ms; 69.6 % against 62.5 % matrix-pipe busy, the difference being the
       * zero slabs of the diagonal band it multiplies) and MORE HBM traffic - 10.1 against 7.0 GB per call: the eight strips of a
       * batch entry each stream T_j[0 : m0 + 64, :] and four entries in flight per XCD do not fit its 4 MB of L2 */
      const char* env = getenv("HIPSDP_GEMM4");
      g4_mode = (env != NULL && env[0] == '1') ? 1 : 0;
   }
   if ( !g4_mode )
      return 0;
   const int tri = (a->flags & HS_GEMM_A_LOWTRI) ? 2 : ((a->flags & HS_GEMM_A_UPTRI) ? 3 : 0);
   if ( tri == 0 || (a->flags & (HS_GEMM_A_LOWTRI | HS_GEMM_A_UPTRI)) == (HS_GEMM_A_LOWTRI | HS_GEMM_A_UPTRI) )
      return 0;                         /* the strip form is for the products with a triangular left factor */
   if ( a->flags & (HS_GEMM_B_LOWTRI | HS_GEMM_LOWER | HS_GEMM_UPPER | HS_GEMM_TILE64 | HS_GEMM_XCD) )
      return 0;
   if ( a->layA != HS_KC || a->splitk > 1 || a->alpha != 1.0 || a->beta != 0.0 )
      return 0;
   if ( (a->lda & 1) || (a->ldb & 1) || (a->strideA & 1) || (a->strideB & 1) || (a->K & 1) || a->K < 16 || a->N < 64 )
      return 0;
   if ( a->layB == HS_MC && (a->N & 1) )
      return 0;
   if ( (((uintptr_t) a->A) & 15) || (((uintptr_t) a->B) & 15) )
      return 0;
   /* 32-bit byte offsets per lane: 16 rows of A, 128 columns of a K-contiguous B */
   if ( a->lda > 8000000LL || a->ldb > 1000000LL || a->ldc > 2000000000LL )
      return 0;
   const long long tm = (a->M + G4_BM - 1) / G4_BM, tn = (a->N + G4_BN - 1) / G4_BN;
   const long long total = tm * tn * a->batch;
   if ( total < 512 || total > 2000000000LL )
      return 0;
   /* a strip's columns beyond N are padding: worth it only when the strips are reasonably full */
   if ( (double) a->N / (double) (tn * G4_BN) < 0.7 )
      return 0;
   const int grid = 256;
   const size_t smem = (size_t) G4_NS * G4_SLOT * sizeof(double);
   static hs_attr_mask attr_done[4];
   const int inst = (tri == 2 ? 0 : 2) + (a->layB == HS_MC ? 0 : 1);
   const void* fn = inst == 0 ? reinterpret_cast<const void*>(&hs_dgemm4_kernel<2, HS_MC>)
      : inst == 1 ? reinterpret_cast<const void*>(&hs_dgemm4_kernel<2, HS_KC>)
      : inst == 2 ? reinterpret_cast<const void*>(&hs_dgemm4_kernel<3, HS_MC>)
      : reinterpret_cast<const void*>(&hs_dgemm4_kernel<3, HS_KC>);
   if ( hs_func_max_lds(fn, (int) smem, &attr_done[inst]) != HS_OK )
      return -HS_ERR_HIP;
   g4_params q = {a->A, a->B, a->C, a->strideA, a->strideB, a->strideC, a->M, a->N, a->K, (int) a->lda, (int) a->ldb, (int) a->ldc,
      (int) tm, (int) tn, (int) total};
   switch ( inst )
   {
   case 0: hipLaunchKernelGGL((hs_dgemm4_kernel<2, HS_MC>), dim3(grid), dim3(256), smem, stream, q); break;
   case 1: hipLaunchKernelGGL((hs_dgemm4_kernel<2, HS_KC>), dim3(grid), dim3(256), smem, stream, q); break;
   case 2: hipLaunchKernelGGL((hs_dgemm4_kernel<3, HS_MC>), dim3(grid), dim3(256), smem, stream, q); break;
   default: hipLaunchKernelGGL((hs_dgemm4_kernel<3, HS_KC>), dim3(grid), dim3(256), smem, stream, q); break;
   }
   if ( hipGetLastError() != hipSuccess )
      return -HS_ERR_HIP;
   hs_mfma_flops_add(g4_executed_flops(a, tri));
   (void) __atomic_add_fetch(&g4_taken, 1, __ATOMIC_RELAXED);
   return 1;
}
