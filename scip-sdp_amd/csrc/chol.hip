/* chol.hip - dense Cholesky of the Schur complement M and of the blocks X, Z; triangular inverse; triangular solves.
 *
 * The reference contains no Cholesky at all: the factorization of the Schur matrix is
 * inside DSDP/SDPA.  This file is the MI355X version of that step:
 *   - right-looking blocked factorization with 64-wide panels; the 64 x 64 diagonal block is factored (and inverted) by
 *     one workgroup in LDS, the panel solve and the trailing update are FP64-MFMA GEMMs (dgemm.hip), so the O(n^3) part
 *     runs on the matrix cores;
 *   - the inverses of the diagonal blocks are kept: triangular solves and the triangular inverse then consist of small
 *     matrix products only (no divisions on the critical path).
 */
#include "hs_kernels.h"
#include <cstdlib>

#define NB 64
#define HS_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if ( e_ != hipSuccess ) { hs_record_hip_error(e_, "kernel launch", __FILE__, __LINE__); return HS_ERR_HIP; } } while (0)

#ifdef PD_TIMING
/* development build only (tests/devtools/potrf_phase_time.py): 100 MHz time stamps of the second step workgroup of the launch
 * whose first column is pd_tj0 */
__device__ long long pd_tbuf[16];
__device__ int pd_tj0;
#define PD_T(idx) do { if ( threadIdx.x == 0 && j0 == pd_tj0 && blockIdx.x == (gridDim.x > 1 ? 1u : 0u) ) pd_tbuf[idx] = wall_clock64(); } while (0)
extern "C" __attribute__((visibility("default"))) int hipsdp_debug_pd_timing(int j0, long long* out)
{
   if ( out != NULL && hipMemcpyFromSymbol(out, HIP_SYMBOL(pd_tbuf), sizeof(long long) * 16) != hipSuccess )
      return 1;
   return hipMemcpyToSymbol(HIP_SYMBOL(pd_tj0), &j0, sizeof(int)) == hipSuccess ? 0 : 1;
}
#else
#define PD_T(idx) do { } while (0)
#endif

/* Factor the nb x nb diagonal block at A (leading dimension lda), nb <= 64: A_blk = L L^T.  Writes L into the lower
 * triangle of the block and inv(L) (64 x 64, identity-padded) into dinv.  flag: first failing global pivot index + 1.
 *
 * 256 threads, the block in LDS.  The sequential part - 64 pivots - is a register recurrence of one wavefront without
 * barriers: the block is processed in four panels of 16 columns; the wavefront holds the panel as lane = row, 16 registers =
 * the columns of the panel.  A pivot step: the reciprocal of the pivot (v_rcp_f64 + one third-order correction; the chain from
 * pivot to pivot runs in wavefront-uniform arithmetic, see pd_panel), the update of the remaining columns of the panel with the
 * scaling folded in, a_ij -= (a_ik / d) a_jk, where the a_jk are wavefront-uniform operands (the next two columns through
 * v_readlane, the others through 64 doubles of LDS, applied one step late), and - off the chain - 1 / sqrt(d) (v_rsq_f64 + one
 * third-order correction) for the stored column.  About 30 instructions per pivot; the loop is bound by their issue.
 * Meanwhile the other three wavefronts apply the rank-16 update of the panel before to the 16 x 16 tiles behind the next
 * panel on the matrix cores; the first wavefront updates the tile column of its next panel itself: one barrier per panel.
 * [History: two columns per barrier with the pivot columns broadcast through LDS read 64 KB of LDS per pair of columns and
 * took 25 us per block; the recurrence replicated in all four wavefronts 14 us; this form 10 us.]
 * Inverse: the four 16 x 16 diagonal blocks are inverted by one wavefront each (lane c = column c, the rows of L preloaded
 * into registers), the six off-diagonal blocks X_ij = -X_ii (sum_k L_ik X_kj) are 16 x 16 x 16 products on the matrix
 * cores (the f64 accumulator layout of the inner sum is exactly the B-operand layout of the outer product), one block
 * diagonal per barrier.  All loops are fully unrolled (static register indices). */
__device__ __forceinline__ void sqrt_and_rsqrt(double d, double y0, double* sd, double* isd)
{
   /* y0 = v_rsq_f64(d) carries about 23 bits; y = y0 (1 + e / 2 + 3 e^2 / 8) with e = 1 - d y0^2 leaves 5/16 e^3: below 2^-66 */
   const double t = d * y0;
   const double e = fma(-t, y0, 1.0);
   const double p = fma(0.375, e, 0.5);
   const double y = fma(y0 * e, p, y0);
   /* square root with one residual correction: g += (d - g^2) y / 2 */
   double g = d * y;
   g = fma(fma(-g, g, d), 0.5 * y, g);
   *sd = g;
   *isd = y;
}

/* LDS written by some lanes of a wavefront is read by other lanes of the same wavefront: the hardware serves a wavefront's
 * LDS instructions in order, the compiler has to be told (without this it keeps values loaded earlier for the lanes that did
 * not store themselves) */
__device__ __forceinline__ void pd_wave_sync()
{
   __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
   __builtin_amdgcn_wave_barrier();
   __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/* the value lane src holds, as a wavefront-uniform (scalar) operand */
__device__ __forceinline__ double pd_lane(double v, int src)
{
   const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
   const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
   return __hiloint2double(hi, lo);
}

struct __attribute__((aligned(16))) dpair { double x, y; };
typedef double v4dc __attribute__((ext_vector_type(4)));

#define PD_LD (NB + 2)
#define PD_SMEM_BYTES ((2 * NB * PD_LD + 5 * NB) * (int) sizeof(double))

/* optional fused inputs / outputs of a single-block factorization (n <= 64; everything NULL for the blocked driver):
 * the matrix is base + alpha * dir (full symmetric storage, ld = lda) and is also stored to Mout; L gets a zero upper
 * triangle; Linv receives inv(L) as n x n (ld = nb); Gram receives inv(L)^T inv(L) = inverse of the matrix (n <= 32) */
struct pd_ext
{
   const double* base;
   const double* dir;
   double        alpha;
   double*       Mout;
   double*       Linv;
   double*       Gram;
   int           set_flag;  /* 1: this launch is the only writer of *flag between two reads: it stores its result (0 = fine) instead
                              * of recording a failure into a flag that somebody cleared beforehand */
   int           rule;      /* 0: forced pivots keep their column, 1: forced columns are zeroed, 2: zeroed when the pivot was <= 0, 3: when it
                            * was at rounding-noise level, <= 8 eps (k + 1) M_kk */
   int*          regmask;   /* semidefinite mode: regmask[j0 + k] = 1 when the pivot of column k was forced (may be NULL) */
   int           from_lds;  /* 1 (k_potrf_step): the block to factor already sits in the first LDS tile, not in global memory */
   int           nostore;   /* 1 (k_potrf_step, all workgroups but the first): L, inv(L), flag and mask stay in LDS / registers */
   double*       Lout;      /* k_potrf_step: where the owner stores L_kk (64 x 64 staging block, ld 64) instead of the matrix itself: the
                             * other step workgroups of the launch still read the unfactored block from the matrix */
};

/* panel of 16 columns starting at c0 out of the LDS block: lane = row; rows above the panel give zeros.  The part of the
 * diagonal block above the diagonal comes along as it is (zeros or the mirror image, finite either way): the recurrence
 * carries those entries through without anybody reading them */
__device__ __forceinline__ void pd_load_panel(double (&r)[16], const double (*W)[PD_LD], int lane, int c0, bool mine)
{
#pragma unroll
   for (int q = 0; q < 8; ++q)
   {
      dpair u = {0.0, 0.0};
      if ( mine )
         u = *reinterpret_cast<const dpair*>(&W[lane][c0 + 2 * q]);
      r[2 * q] = u.x;
      r[2 * q + 1] = u.y;
   }
}

/* The 16 pivot steps of one panel (lane = row, r[c] = column c0 + c of the row; rows above the panel hold zeros).
 * CHECKED = false is the straight recurrence: no branch, nothing but the pivot chain and the updates; it only notes whether a
 * pivot fell below its threshold thr[k] (0, or regtol * reference diagonal in semidefinite mode; also true for NaN), in which
 * case the caller reloads the panel and runs the CHECKED form, which replaces such pivots (see below).  isd[k] = 1 / l_kk,
 * fbits: bit k set where the pivot was forced and its column zeroed.
 * The chain from pivot to pivot stays in wavefront-uniform arithmetic: the two entries the next pivot is made of, a_{k+1,k} and
 * a_{k+1,k+1}, are fetched (v_readlane) at the start of the step, when they are final, and d_{k+1} = a_{k+1,k+1} -
 * (a_{k+1,k} / d_k) a_{k+1,k} is formed by every lane from them - the same operations lane k + 1 applies to its own entry, so
 * the same bits - instead of being read back from that lane after the update: v_rcp_f64, one third-order step (rc (1 + e + e^2),
 * e = 1 - d rc) and two operations per pivot, no register-file round trip. */
template<bool CHECKED>
__device__ __forceinline__ bool pd_panel(double (&r)[16], const double (&thr)[16], const double* __restrict__ d0s, int lane, int c0, int nb, int j0,
   bool psd, double regtol, int rule, double (&isdo)[16], unsigned& fbits, int& bad, double* __restrict__ colb)
{
   bool special = false;
   fbits = 0u;
   double cprev[16], tprev = 0.0;
   double d = pd_lane(r[0], c0);
#pragma unroll
   for (int k = 0; k < 16; ++k)
   {
      const int gk = c0 + k;
      const double a = r[k];
      /* column entries of the next two rows (scalar operands of their updates) and the diagonal entry of the next row */
      double sa = 0.0, sb = 0.0, sa2 = 0.0;
      if ( k + 1 < 16 )
      {
         sa = pd_lane(a, gk + 1);
         sb = pd_lane(r[k + 1], gk + 1);
      }
      if ( k + 2 < 16 )
         sa2 = pd_lane(a, gk + 2);
      double rc0 = __builtin_amdgcn_rcp(d);
      double y0 = __builtin_amdgcn_rsq(d);
      double cj[16];
      if ( k + 3 < 16 )
      {
         colb[lane] = a;               /* every row: no lane mask to set up; the rows of the diagonal block are read back */
         pd_wave_sync();
#pragma unroll
         for (int q = (k + 3) / 2; q < 8; ++q)
         {
            const dpair u = *reinterpret_cast<const dpair*>(&colb[c0 + 2 * q]);
            cj[2 * q] = u.x;
            cj[2 * q + 1] = u.y;
         }
      }
      /* the updates of the step before, columns k + 2 and up: their column entries came through LDS and have had a whole step
       * to arrive */
      if ( k > 0 )
      {
#pragma unroll
         for (int j = k + 2; j < 16; ++j)
            r[j] = fma(-tprev, cprev[j], r[j]);
      }
      bool reg = false;
      if ( !CHECKED )
         special = special || !(d > thr[k]);
      else if ( gk < nb )
      {
         if ( psd )
         {
            /* semidefinite mode (Schur complement with dependent columns): a pivot that cancelled to rounding level is
             * replaced by a small positive one, which keeps the direction alive so that a ray along it can be found; when
             * the pivot is not even positive (rule 2; rule 1: every forced pivot) the rest of the column is rounding noise of
             * a column that is zero in exact arithmetic, and it is SET to zero: dividing noise by the forced pivot and
             * eliminating with it amplifies the noise exponentially over a run of dependent columns (observed: entries at
             * 1e158 for m = 200 with rank 136) */
            const double dd = d0s[gk];
            if ( !(d > regtol * dd) || !(d > 1e-300) )
            {
               /* rule 3: "not even positive" is the sign of a rounding-noise number - two implementations of the same algorithm
                * (this kernel, the oracle) draw it differently, and from there on their iterates differ.  The noise of pivot k of a
                * matrix with dependent columns is of the order eps k M_kk, so the test is against that level instead of against 0:
                * both sides then zero the same columns unless the pivot sits within rounding of the threshold itself */
               reg = (rule == 1) || (rule == 2 && !(d > 0.0)) || (rule == 3 && !(d > 1.78e-15 * (double) (j0 + gk + 1) * dd));
               d = (dd > 1e-280) ? regtol * dd : 1.0;
               y0 = __builtin_amdgcn_rsq(d);
               rc0 = __builtin_amdgcn_rcp(d);
            }
         }
         else if ( !(d > 0.0) )
         {
            if ( bad == 0 )
               bad = j0 + gk + 1;
            d = 1.0;                 /* keep going with a harmless pivot; the caller reads the flag */
            y0 = 1.0;
            rc0 = 1.0;
         }
      }
      /* the multiplier of the update, a_ik / d; square root and 1 / l_kk (for the stored column) follow beside the chain */
      const double ec = fma(-d, rc0, 1.0);
      const double rcd = fma(rc0 * ec, 1.0 + ec, rc0);
      const double t = reg ? 0.0 : a * rcd;
      const double tu = reg ? 0.0 : sa * rcd;
      const double dnext = fma(-tu, sa, sb);
      double sd, isd;
      if ( CHECKED )
         sqrt_and_rsqrt(d, y0, &sd, &isd);
      else
      {
         /* 1 / l_kk only: the pivot row itself gets l_kk = d / sqrt(d) from the scaling of the column */
         const double tq = d * y0;
         const double eq = fma(-tq, y0, 1.0);
         isd = fma(y0 * eq, fma(0.375, eq, 0.5), y0);
         sd = 0.0;
      }
      const double e = reg ? 0.0 : isd;            /* scale of the sub-column (0: forced pivot) */
      /* the column entries a_jk of the diagonal block as wavefront-uniform operands: the first two (the next two pivot columns
       * must be complete when their turn comes) through v_readlane, the others through LDS (written above, no barrier: LDS
       * serves a wavefront in order) - a fifth of the instructions of 2 x 14 v_readlane + their wait states - and applied
       * one step later, when the loads have long arrived */
      if ( k + 1 < 16 )
         r[k + 1] = fma(-t, sa, r[k + 1]);
      if ( k + 2 < 16 )
         r[k + 2] = fma(-t, sa2, r[k + 2]);
      tprev = t;
#pragma unroll
      for (int j = k + 3; j < 16; ++j)
         cprev[j] = cj[j];
      /* final value of column gk in my row.  The straight form scales every row: the pivot row gets d / sqrt(d), the rows above
       * it (upper part of the diagonal block, rows before the panel) carry values nobody reads */
      if ( CHECKED )
         r[k] = (lane > gk) ? a * e : ((lane == gk) ? sd : 0.0);
      else
         r[k] = a * e;
      isdo[k] = isd;
      if ( reg )
         fbits |= 1u << k;
      d = dnext;
   }
   return special;
}

/* rank-16 update W(i, j) -= P_i P_j^T of nt <= NTMAX tiles of one tile column j (i = i0 .. i0 + nt - 1) with the panel that
 * starts at column cp: P_j is read once, the accumulators of the tiles advance together (a dependent MFMA waits twice as long
 * as an independent one) */
template<int NTMAX>
__device__ __forceinline__ void pd_update_tiles(double (*W)[PD_LD], const double (*Lm)[PD_LD], int nt, int i0, int j, int cp, int lr, int lk)
{
   double pb[4], pa[NTMAX][4], cv[NTMAX][4];
   v4dc acc[NTMAX];
#pragma unroll
   for (int sidx = 0; sidx < 4; ++sidx)
      pb[sidx] = Lm[16 * j + lr][cp + 4 * sidx + lk];
#pragma unroll
   for (int t = 0; t < NTMAX; ++t)
      if ( t < nt )
      {
         acc[t] = (v4dc){0.0, 0.0, 0.0, 0.0};
#pragma unroll
         for (int sidx = 0; sidx < 4; ++sidx)
         {
            pa[t][sidx] = Lm[16 * (i0 + t) + lr][cp + 4 * sidx + lk];
            cv[t][sidx] = W[16 * (i0 + t) + lk + 4 * sidx][16 * j + lr];
         }
      }
#pragma unroll
   for (int sidx = 0; sidx < 4; ++sidx)
#pragma unroll
      for (int t = 0; t < NTMAX; ++t)
         if ( t < nt )
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[t][sidx], pb[sidx], acc[t], 0, 0, 0);
#pragma unroll
   for (int t = 0; t < NTMAX; ++t)
      if ( t < nt )
      {
#pragma unroll
         for (int rr = 0; rr < 4; ++rr)
            W[16 * (i0 + t) + lk + 4 * rr][16 * j + lr] = cv[t][rr] - acc[t][rr];
      }
}

/* LDS: tile 0 = the block while it is factored, inv(L) afterwards; tile 1 = L; then 64 doubles each: 1 / l_kk, the reference
 * diagonal, the forced flags, the pivot column of the current step, the pivot thresholds */
template<int NBK>      /* padded block size actually processed: 16, 32, 48 or 64 (small blocks skip the identity padding) */
__device__ __forceinline__ void pd_body(double* __restrict__ A, long long lda, int nb, int j0,
   double* __restrict__ dinv, int* __restrict__ flag, const double* __restrict__ diag0, double regtol, const pd_ext& ext)
{
   extern __shared__ __attribute__((aligned(16))) double pd_smem[];
   double (*W)[PD_LD] = reinterpret_cast<double (*)[PD_LD]>(pd_smem);                      /* working block, row major */
   double (*Lm)[PD_LD] = reinterpret_cast<double (*)[PD_LD]>(pd_smem + NB * PD_LD);        /* L, row major */
   double (*X)[PD_LD] = W;                                                                  /* inv(L), row major (after the factorization) */
   double* invd = pd_smem + 2 * NB * PD_LD;
   double* d0s = invd + NB;
   double* forced = d0s + NB;          /* 1.0 where the pivot of the column was forced and its column zeroed (semidefinite mode) */
   const int tid = threadIdx.x;
   const int lane = tid & 63;
   const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
   const int lr = lane & 15, lk = lane >> 4;
   double* colb = forced + NB;                  /* the current pivot column of the panel (64 doubles, first wavefront) */
   double* thrs = colb + NB;                    /* below this a pivot needs the checked form: 0, or regtol * reference diagonal */
   if ( !ext.from_lds )         /* k_potrf_step hands the block over complete: lower triangle, zeros above, identity padding */
   {
      const int i = tid >> 2;
      const int jc = tid & 3;
#pragma unroll
      for (int jj = 0; jj < 16; ++jj)
      {
         const int j = jc + 4 * jj;
         double v = (i == j) ? 1.0 : 0.0;
         if ( i < nb && j <= i )
         {
            if ( ext.base != NULL )
            {
               v = ext.base[(long long) i * lda + j];
               if ( ext.dir != NULL )
                  v = fma(ext.alpha, ext.dir[(long long) i * lda + j], v);
               if ( ext.Mout != NULL )
               {
                  ext.Mout[(long long) i * lda + j] = v;
                  ext.Mout[(long long) j * lda + i] = v;
               }
            }
            else
               v = A[(long long) i * lda + j];
         }
         W[i][j] = v;
      }
   }
   if ( tid < NB )
   {
      const double dd = (diag0 != NULL && tid < nb) ? diag0[j0 + tid] : 0.0;
      d0s[tid] = dd;
      thrs[tid] = (diag0 != NULL && tid < nb) ? fmax(regtol * dd, 1e-300) : 0.0;
   }
   int bad = 0;
   __syncthreads();
   PD_T(4);

   constexpr int NBLK = NBK / 16;
#pragma unroll
   for (int b = 0; b < NBLK; ++b)
   {
      const int c0 = 16 * b;
      if ( b == 2 )
         PD_T(3);
      /* the panel: lane = row, r[c] = column c0 + c; rows above the panel idle; within the diagonal block the upper part is not
       * used.  One wavefront does it (the loop is bound by its instruction count: about 50 per pivot) and the others
       * update the tiles behind the next panel meanwhile */
      if ( wave == 0 )
      {
         /* first the tile column this panel lives in gets the update of the panel before it (the other wavefronts take the
          * tile columns behind it meanwhile: the barrier at the end of the round is the only one) */
         if constexpr ( NBLK > 1 )
         {
            if ( b > 0 )
               pd_update_tiles<NBLK - 1>(W, Lm, NBLK - b, b, b, c0 - 16, lr, lk);
         }
         double r[16];
         const bool mine = (lane >= c0 && lane < NBK);
         pd_load_panel(r, W, lane, c0, mine);
         if ( b == 2 )
            PD_T(11);
         double thr[16], isdo[16];
#pragma unroll
         for (int q = 0; q < 8; ++q)
         {
            const dpair u = *reinterpret_cast<const dpair*>(&thrs[c0 + 2 * q]);
            thr[2 * q] = u.x;
            thr[2 * q + 1] = u.y;
         }
         unsigned fbits;
         if ( pd_panel<false>(r, thr, d0s, lane, c0, nb, j0, diag0 != NULL, regtol, ext.rule, isdo, fbits, bad, colb) )
         {
            /* rare: a pivot needs the semidefinite treatment or is reported: once more from the unchanged block, with the checks */
            pd_load_panel(r, W, lane, c0, mine);
            (void) pd_panel<true>(r, thr, d0s, lane, c0, nb, j0, diag0 != NULL, regtol, ext.rule, isdo, fbits, bad, colb);
         }
         if ( b == 2 )
            PD_T(12);
         if ( lane == 0 )
         {
            /* 1 / l_kk of the panel (wavefront-uniform values): one lane stores them */
#pragma unroll
            for (int q = 0; q < 8; ++q)
            {
               const dpair u = {isdo[2 * q], isdo[2 * q + 1]};
               *reinterpret_cast<dpair*>(&invd[c0 + 2 * q]) = u;
            }
         }
         if ( lane < 16 )
         {
            const bool f = ((fbits >> lane) & 1u) != 0u;
            forced[c0 + lane] = f ? 1.0 : 0.0;
            if ( ext.regmask != NULL && !ext.nostore && c0 + lane < nb )
               ext.regmask[j0 + c0 + lane] = f ? 1 : 0;
         }
         if ( mine )
         {
#pragma unroll
            for (int q = 0; q < 8; ++q)
            {
               const dpair u = {r[2 * q], r[2 * q + 1]};
               *reinterpret_cast<dpair*>(&Lm[lane][c0 + 2 * q]) = u;
            }
         }
      }
      else if ( b > 0 )
      {
         /* tiles (i, j), b < j <= i, with the panel before this one: dealt out to the wavefronts 1 - 3 */
         int idx = 0;
#pragma unroll
         for (int j = b + 1; j < NBLK; ++j)
#pragma unroll
            for (int i = j; i < NBLK; ++i, ++idx)
               if ( idx % 3 == wave - 1 )
                  pd_update_tiles<1>(W, Lm, 1, i, j, c0 - 16, lr, lk);
      }
      if ( b == 2 )
         PD_T(13);
      __syncthreads();
      if ( b == 2 )
         PD_T(15);
   }
   PD_T(5);

   /* write L back */
   if ( !ext.nostore )
   {
      double* Lw = ext.Lout != NULL ? ext.Lout : A;
      const long long ldw = ext.Lout != NULL ? NB : lda;
      const int i = tid >> 2;
      const int jc = tid & 3;
#pragma unroll
      for (int jj = 0; jj < 16; ++jj)
      {
         const int j = jc + 4 * jj;
         if ( i < nb && j <= i )
            Lw[(long long) i * ldw + j] = (i < NBK) ? Lm[i][j] : ((i == j) ? 1.0 : 0.0);
         else if ( ext.base != NULL && i < nb && j < nb )
            Lw[(long long) i * ldw + j] = 0.0;
      }
   }
   if ( tid == 0 && !ext.nostore )
   {
      if ( ext.set_flag )
         *flag = bad;
      else if ( bad != 0 )
         atomicCAS(flag, 0, bad);
   }

   PD_T(6);
   /* inverse, diagonal 16 x 16 blocks: wavefront w, lane c < 16 solves L_ww x = e_c; row i of L_ww is read as a whole (all rows
    * requested before the recurrence starts: the loads are not on its critical path) */
   if ( wave < NBLK && lane < 16 )
   {
      const int base = 16 * wave;
      double lrow[16][16];
      double iv[16];
#pragma unroll
      for (int ii = 0; ii < 16; ++ii)
      {
         iv[ii] = invd[base + ii];
#pragma unroll
         for (int q = 0; 2 * q < ii; ++q)
         {
            const dpair u = *reinterpret_cast<const dpair*>(&Lm[base + ii][base + 2 * q]);
            lrow[ii][2 * q] = u.x;
            lrow[ii][2 * q + 1] = u.y;
         }
      }
      double xv[16];
#pragma unroll
      for (int ii = 0; ii < 16; ++ii)
      {
         double acc = (ii == lane) ? 1.0 : 0.0;
#pragma unroll
         for (int j = 0; j < ii; ++j)
            acc = fma(-lrow[ii][j], xv[j], acc);
         xv[ii] = acc * iv[ii];
         X[base + ii][base + lane] = xv[ii];
      }
   }
   __syncthreads();
   PD_T(7);
   /* off-diagonal blocks by block diagonals: block (bi, bj = bi - dd) on wavefront bj */
#pragma unroll
   for (int dd = 1; dd < NBLK; ++dd)
   {
      if ( wave < NBLK - dd )
      {
         const int bj = wave, bi = wave + dd;
         v4dc acc = (v4dc){0.0, 0.0, 0.0, 0.0};
         for (int kb = bj; kb < bi; ++kb)
         {
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx)
            {
               const double a = Lm[16 * bi + lr][16 * kb + 4 * sidx + lk];
               const double bv = X[16 * kb + 4 * sidx + lk][16 * bj + lr];
               acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, acc, 0, 0, 0);
            }
         }
         v4dc acc2 = (v4dc){0.0, 0.0, 0.0, 0.0};
#pragma unroll
         for (int sidx = 0; sidx < 4; ++sidx)
         {
            const double a = X[16 * bi + lr][16 * bi + 4 * sidx + lk];
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, acc[sidx], acc2, 0, 0, 0);
         }
#pragma unroll
         for (int rr = 0; rr < 4; ++rr)
            X[16 * bi + lk + 4 * rr][16 * bj + lr] = -acc2[rr];
      }
      __syncthreads();
   }
   PD_T(8);
   /* inv(L), identity padded, block upper triangle zero */
   if ( !ext.nostore )
   for (int e = tid; e < NB * NB; e += 256)
   {
      const int rr = e >> 6, cc = e & 63;
      double v;
      if ( rr >= NBK || cc >= NBK )
         v = (rr == cc) ? 1.0 : 0.0;
      else
         v = ((rr >> 4) >= (cc >> 4)) ? X[rr][cc] : 0.0;
      dinv[e] = v;
      if ( ext.Linv != NULL && rr < nb && cc < nb )
         ext.Linv[rr * nb + cc] = v;
   }
   if ( ext.Gram != NULL && NBK <= 32 )
   {
      /* inverse of the matrix: Gram[r][c] = sum_{k >= max(r, c)} X[k][r] X[k][c] */
      for (int e = tid; e < nb * nb; e += 256)
      {
         const int rr = e / nb, cc = e - rr * nb;
         double acc = 0.0;
         for (int k = (rr > cc ? rr : cc); k < nb; ++k)
            acc += X[k][rr] * X[k][cc];
         ext.Gram[e] = acc;
      }
   }
}

template<int NBK>
__global__ void __launch_bounds__(256) k_potrf_diag(double* __restrict__ A, long long lda, int nb, int j0,
   double* __restrict__ dinv, int* __restrict__ flag, const double* __restrict__ diag0, double regtol, pd_ext ext)
{
   pd_body<NBK>(A, lda, nb, j0, dinv, flag, diag0, regtol, ext);
}

/* two single-block factorizations of the same size in one launch (blockIdx.x selects): the trial iterates X + alpha dX and
 * Z + alpha dZ of a small block were two launches of one workgroup each, one behind the other on the same queue */
struct pd_job { double* A; double* dinv; int* flag; pd_ext ext; };
template<int NBK>
__global__ void __launch_bounds__(256) k_potrf_diag_pair(pd_job J0, pd_job J1, long long lda, int nb)
{
   const pd_job J = blockIdx.x ? J1 : J0;
   pd_body<NBK>(J.A, lda, nb, 0, J.dinv, J.flag, NULL, 1e-13, J.ext);
}

/* ---- one block column of the blocked factorization in ONE launch -------------------------------------------------------
 * Launch kb of hs_potrf_psd (n > 64).  Two kinds of workgroups that do not depend on each other:
 *  - step workgroups (one per 64-row block of block column kb, the first one owns the diagonal block): apply the rank-64 update
 *    from block column kb - 1 to their own block and to the diagonal block, factor and invert the diagonal block (every step
 *    workgroup does that for itself: 64 x 64 work, identical bits, and nobody waits for anybody), form their panel block
 *    P = B inv(L_kk)^T on the matrix cores, zero the columns of forced pivots (semidefinite mode) and store it;
 *  - trailing workgroups (one per lower 64 x 64 tile right of block column kb): the update A_ij -= P_i P_j^T from block column
 *    kb - 1, which the previous launch finished.  Column kb itself is left out - the step workgroups have just done it.
 * So a block column costs one launch (diagonal kernel, panel GEMM, mask kernel and trailing GEMM before), and the trailing
 * update of a column runs beside the next column's diagonal factorization instead of in front of it.
 * The owner must not put L_kk into the matrix while other step workgroups of the same launch may still have to read the
 * unfactored block (workgroups of a launch start whenever the device has room): it writes L_kk to a staging block behind the
 * inverses (dinv holds 2 * ceil(n / 64) blocks) and the owner of the NEXT launch moves it into the matrix; the last block column
 * has no other readers and is written directly.  Every output element
 * sees the same operations in the same order as in the four-launch form: identical bits (tests/test_gpu_units.py). */
__device__ __forceinline__ void ps_load_tile(const double* __restrict__ src, long long lda, int rows, double (*dst)[PD_LD])
{
   /* 64 x 64 block at src (rows valid, the rest zero) -> LDS tile; 256 threads; all loads of a thread are issued before its stores */
   double v0[8], v1[8];
#pragma unroll
   for (int q = 0; q < 8; ++q)
   {
      const int e = threadIdx.x + 256 * q;
      const int r = e >> 5, c = (e & 31) * 2;
      v0[q] = 0.0; v1[q] = 0.0;
      if ( r < rows )
      {
         v0[q] = src[(long long) r * lda + c];
         v1[q] = src[(long long) r * lda + c + 1];
      }
   }
#pragma unroll
   for (int q = 0; q < 8; ++q)
   {
      const int e = threadIdx.x + 256 * q;
      const int r = e >> 5, c = (e & 31) * 2;
      dst[r][c] = v0[q];
      dst[r][c + 1] = v1[q];
   }
}

/* acc[t][.] (rows 16 wave .. + 15, column tile t) = sum_k Ta[row][k] Tb[col][k], k ascending in steps of 4 up to klim[t] */
template<bool LOWTRI>
__device__ __forceinline__ void ps_mma(const double (*Ta)[PD_LD], const double (*Tb)[PD_LD], int wave, int lane, v4dc* acc)
{
   const int lr = lane & 15, lk = lane >> 4;
   /* the left operand of the wave (16 rows x 64) is read once and serves the four column tiles; the four accumulators advance
    * together (an MFMA that continues an accumulator waits for the one before it: four independent chains keep the pipe full);
    * each accumulator still sums k in ascending order */
   double a[16];
#pragma unroll
   for (int sidx = 0; sidx < 16; ++sidx)
      a[sidx] = Ta[16 * wave + lr][4 * sidx + lk];
#pragma unroll
   for (int t = 0; t < 4; ++t)
      acc[t] = (v4dc){0.0, 0.0, 0.0, 0.0};
#pragma unroll
   for (int sidx = 0; sidx < 16; ++sidx)
   {
#pragma unroll
      for (int t = 0; t < 4; ++t)
      {
         /* Tb block lower triangular (LOWTRI): Tb[col][k] = 0 for k beyond the column's 16-block */
         if ( !LOWTRI || sidx < 4 * (t + 1) )
         {
            const double b = Tb[16 * t + lr][4 * sidx + lk];
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[sidx], b, acc[t], 0, 0, 0);
         }
      }
   }
}

template<int NBK>
__global__ void __launch_bounds__(256) k_potrf_step(double* __restrict__ A, long long lda, int n, int kb, int nblk,
   double* __restrict__ dinv, int* __restrict__ flag, const double* __restrict__ diag0, double regtol, pd_ext ext)
{
   extern __shared__ __attribute__((aligned(16))) double pd_smem[];
   double* lstage = dinv + (long long) nblk * NB * NB;                                              /* staged L_kk blocks */
   double (*bufA)[PD_LD] = reinterpret_cast<double (*)[PD_LD]>(pd_smem);                 /* tile 0: the block to factor, then inv(L_kk) */
   double (*bufB)[PD_LD] = reinterpret_cast<double (*)[PD_LD]>(pd_smem + NB * PD_LD);     /* tile 1: L_kk, then the panel block */
   const double* forced = pd_smem + 2 * NB * PD_LD + 2 * NB;
   const int tid = threadIdx.x, lane = tid & 63;
   const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
   const int lr = lane & 15, lk = lane >> 4;
   const int nS = nblk - kb;
   const int j0 = kb * NB;
   v4dc acc[4];

   if ( (int) blockIdx.x >= nS )
   {
      /* trailing tile (bi, bj), kb + 1 <= bj <= bi < nblk, updated from block column kb - 1 */
      const int tt = (int) blockIdx.x - nS;
      int ti = (int) ((sqrt(8.0 * (double) tt + 1.0) - 1.0) * 0.5);
      while ( (ti + 1) * (ti + 2) / 2 <= tt ) ++ti;
      while ( ti * (ti + 1) / 2 > tt ) --ti;
      const int tj = tt - ti * (ti + 1) / 2;
      const int bi = kb + 1 + ti, bj = kb + 1 + tj;
      const int ri = min(NB, n - bi * NB), rj = min(NB, n - bj * NB);
      const double* Pi = A + (long long) bi * NB * lda + (j0 - NB);
      const double* Pj = A + (long long) bj * NB * lda + (j0 - NB);
      ps_load_tile(Pi, lda, ri, bufA);
      if ( bi != bj )
         ps_load_tile(Pj, lda, rj, bufB);
      __syncthreads();
      ps_mma<false>(bufA, bi != bj ? bufB : bufA, wave, lane, acc);
      double* C = A + (long long) bi * NB * lda + (long long) bj * NB;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
         for (int rr = 0; rr < 4; ++rr)
         {
            const int row = 16 * wave + lk + 4 * rr, col = 16 * t + lr;
            if ( row < ri && col < rj )
            {
               double* c = C + (long long) row * lda + col;
               double v = -acc[t][rr];
               v += *c;
               *c = v;
            }
         }
      return;
   }

   /* ---- step workgroup of block row rb */
   PD_T(0);
   const int rb = kb + (int) blockIdx.x;
   const int r0 = rb * NB;
   const int nb = min(NB, n - j0);
   const int rows = min(NB, n - r0);
   double* Akk = A + (long long) j0 * lda + j0;
   double* Ark = A + (long long) r0 * lda + j0;
   v4dc accB[4];
#pragma unroll
   for (int t = 0; t < 4; ++t)
      accB[t] = (v4dc){0.0, 0.0, 0.0, 0.0};
   bool from_lds = false;
   /* the entries of the diagonal block and of the own block that the updates below are subtracted from: requested first, so
    * that their latency passes behind the tile loads and the products */
   double akk[4][4], ark[4][4];
   if ( kb > 0 )
   {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
         for (int rr = 0; rr < 4; ++rr)
         {
            const int row = 16 * wave + lk + 4 * rr, col = 16 * t + lr;
            akk[t][rr] = (row < nb && col <= row) ? Akk[(long long) row * lda + col] : 0.0;
         }
   }
   if ( blockIdx.x > 0 )
   {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
         for (int rr = 0; rr < 4; ++rr)
         {
            const int row = 16 * wave + lk + 4 * rr, col = 16 * t + lr;
            ark[t][rr] = (row < rows) ? Ark[(long long) row * lda + col] : 0.0;
         }
   }
   if ( kb > 0 )
   {
      ps_load_tile(Akk - NB, lda, nb, bufA);                    /* P_k: block (kb, kb - 1) */
      if ( blockIdx.x > 0 )
         ps_load_tile(Ark - NB, lda, rows, bufB);              /* P_r: block (rb, kb - 1) */
      __syncthreads();
      PD_T(1);
      ps_mma<false>(bufA, bufA, wave, lane, acc);               /* P_k P_k^T */
      if ( blockIdx.x > 0 )
         ps_mma<false>(bufB, bufA, wave, lane, accB);           /* P_r P_k^T */
      __syncthreads();
      /* updated diagonal block -> the tile the factorization reads */
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
         for (int rr = 0; rr < 4; ++rr)
         {
            const int row = 16 * wave + lk + 4 * rr, col = 16 * t + lr;
            double v = (row >= nb && row == col) ? 1.0 : 0.0;       /* identity padding of a ragged last block */
            if ( row < nb && col <= row )
            {
               v = -acc[t][rr];
               v += akk[t][rr];
            }
            bufA[row][col] = v;
         }
      __syncthreads();
      from_lds = true;
      PD_T(2);
   }
   pd_ext e2 = ext;
   e2.from_lds = from_lds ? 1 : 0;
   e2.nostore = blockIdx.x > 0 ? 1 : 0;
   e2.Lout = (nS > 1) ? lstage + (long long) kb * NB * NB : NULL;
   pd_body<NBK>(Akk, lda, nb, j0, dinv + (long long) kb * NB * NB, flag, diag0, regtol, e2);
   if ( blockIdx.x == 0 )
   {
      if ( kb > 0 )
      {
         /* the factor of the previous diagonal block moves from its staging block into the matrix (nobody reads that block in
          * this launch) */
         const double* src = lstage + (long long) (kb - 1) * NB * NB;
         double* dst = A + (long long) (j0 - NB) * lda + (j0 - NB);
         for (int e = tid; e < NB * NB; e += 256)
         {
            const int r = e >> 6, c = e & 63;
            if ( c <= r )
               dst[(long long) r * lda + c] = src[e];
         }
      }
      return;
   }
   /* ---- panel block: B = A_rk - P_r P_k^T (accB), P = B inv(L_kk)^T, forced columns zeroed */
   PD_T(9);
   __syncthreads();
#pragma unroll
   for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
      {
         const int row = 16 * wave + lk + 4 * rr, col = 16 * t + lr;
         double v = 0.0;
         if ( row < rows )
         {
            v = -accB[t][rr];
            v += ark[t][rr];
         }
         bufB[row][col] = v;
      }
   __syncthreads();
   ps_mma<true>(bufB, bufA, wave, lane, acc);
#pragma unroll
   for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
      {
         const int row = 16 * wave + lk + 4 * rr, col = 16 * t + lr;
         if ( row < rows )
            Ark[(long long) row * lda + col] = forced[col] != 0.0 ? 0.0 : acc[t][rr];
      }
   PD_T(10);
}

template<int NBK>
static int launch_potrf_step(hipStream_t s, double* A, long long lda, int n, int kb, int nblk, double* dinv, int* flag,
   const double* diag0, const pd_ext& ext)
{
   static hs_attr_mask attr_done;
   HS_CALL( hs_func_max_lds(reinterpret_cast<const void*>(&k_potrf_step<NBK>), PD_SMEM_BYTES, &attr_done) );
   const int nS = nblk - kb;
   const int t = nblk - kb - 1;
   const int nT = kb > 0 ? t * (t + 1) / 2 : 0;
   hipLaunchKernelGGL((k_potrf_step<NBK>), dim3(nS + nT), dim3(256), PD_SMEM_BYTES, s, A, lda, n, kb, nblk, dinv, flag, diag0, 1e-13, ext);
   HS_LAUNCH_CHECK();
   return HS_OK;
}

template<int NBK>
static int launch_potrf_diag(hipStream_t s, double* Ajj, long long lda, int nb, int j0, double* dj, int* flag, const double* diag0,
   const pd_ext* extp = NULL)
{
   pd_ext ext = {NULL, NULL, 0.0, NULL, NULL, NULL, 0, 2, NULL};
   if ( extp != NULL )
      ext = *extp;
   static hs_attr_mask attr_done;
   HS_CALL( hs_func_max_lds(reinterpret_cast<const void*>(&k_potrf_diag<NBK>), PD_SMEM_BYTES, &attr_done) );
   hipLaunchKernelGGL((k_potrf_diag<NBK>), dim3(1), dim3(256), PD_SMEM_BYTES, s, Ajj, lda, nb, j0, dj, flag, diag0, 1e-13, ext);
   HS_LAUNCH_CHECK();
   return HS_OK;
}

/* single-block factorization (n <= 64) of base + alpha * dir with the fused outputs described at pd_ext: one launch for what
 * the general path does with scale_add, copy, potrf, zero_upper, trtri (+ gemm, mirror for the inverse) */
int hs_potrf_small_ext(hipStream_t s, int n, double* L, double* dinv, int* flag, const double* base, const double* dir, double alpha,
   double* Mout, double* Linv, double* Gram, int set_flag)
{
   if ( n <= 0 )
      return HS_OK;
   if ( n > NB || base == NULL || (Gram != NULL && n > 32) )
      return HS_ERR_ARG;
   pd_ext ext = {base, dir, alpha, Mout, Linv, Gram, set_flag, 2, NULL};
   if ( n <= 16 )
      return launch_potrf_diag<16>(s, L, n, n, 0, dinv, flag, NULL, &ext);
   if ( n <= 32 )
      return launch_potrf_diag<32>(s, L, n, n, 0, dinv, flag, NULL, &ext);
   if ( n <= 48 )
      return launch_potrf_diag<48>(s, L, n, n, 0, dinv, flag, NULL, &ext);       /* three panels of 16 instead of four (example_CLS: n = 43) */
   return launch_potrf_diag<64>(s, L, n, n, 0, dinv, flag, NULL, &ext);
}

template<int NBK>
static int launch_potrf_diag_pair(hipStream_t s, int n, const pd_job& J0, const pd_job& J1)
{
   static hs_attr_mask attr_done;
   HS_CALL( hs_func_max_lds(reinterpret_cast<const void*>(&k_potrf_diag_pair<NBK>), PD_SMEM_BYTES, &attr_done) );
   hipLaunchKernelGGL((k_potrf_diag_pair<NBK>), dim3(2), dim3(256), PD_SMEM_BYTES, s, J0, J1, (long long) n, n);
   HS_LAUNCH_CHECK();
   return HS_OK;
}

/* hs_potrf_small_ext for two matrices of the same order side by side (index 0 / 1 of every array argument) */
int hs_potrf_small_ext_pair(hipStream_t s, int n, double* const* L, double* const* dinv, int* const* flag, const double* const* base,
   const double* const* dir, double alpha, double* const* Mout, double* const* Linv, double* const* Gram, int set_flag)
{
   if ( n <= 0 )
      return HS_OK;
   if ( n > NB || base[0] == NULL || base[1] == NULL || ((Gram[0] != NULL || Gram[1] != NULL) && n > 32) )
      return HS_ERR_ARG;
   pd_job J[2];
   for (int k = 0; k < 2; ++k)
   {
      const pd_ext ext = {base[k], dir[k], alpha, Mout[k], Linv[k], Gram[k], set_flag, 2, NULL};
      J[k].A = L[k]; J[k].dinv = dinv[k]; J[k].flag = flag[k]; J[k].ext = ext;
   }
   if ( n <= 16 )
      return launch_potrf_diag_pair<16>(s, n, J[0], J[1]);
   if ( n <= 32 )
      return launch_potrf_diag_pair<32>(s, n, J[0], J[1]);
   if ( n <= 48 )
      return launch_potrf_diag_pair<48>(s, n, J[0], J[1]);
   return launch_potrf_diag_pair<64>(s, n, J[0], J[1]);
}

/* the columns of the panel that belong to forced pivots of the diagonal block are zero in exact arithmetic */
__global__ void k_zero_forced_cols(int rows, int nb, double* __restrict__ P, long long lda, const int* __restrict__ mask)
{
   for (long long e = (long long) blockIdx.x * blockDim.x + threadIdx.x; e < (long long) rows * nb; e += (long long) gridDim.x * blockDim.x)
   {
      const int r = (int) (e / nb), c = (int) (e - (long long) r * nb);
      if ( mask[c] )
         P[(long long) r * lda + c] = 0.0;
   }
}

/* test hook (hipsdp_potrf_selfcheck): 1 routes the blocked factorization through the four-launch form */
static int g_potrf_force_v1 = 0;
int hs_potrf_force_v1(int on) { const int old = g_potrf_force_v1; g_potrf_force_v1 = on; return old; }

/* doubles the dinv argument of hs_potrf / hs_potrf_psd must hold: the inverses of the ceil(n / 64) diagonal blocks and, for
 * n > 64, as many staging blocks of the fused block-column kernel */
long long hs_potrf_dinv_len(int n)
{
   const long long nblk = n > 0 ? (n + NB - 1) / NB : 1;
   return (nblk > 1 ? 2 : 1) * nblk * NB * NB;
}

int hs_potrf(hipStream_t s, int n, double* A, double* dinv, int* flag, const double* diag0)
{
   return hs_potrf_psd(s, n, A, dinv, flag, diag0, NULL, 0);
}

/* diag0 != NULL: semidefinite mode; regmask (n ints, device; required when n > 64 in that mode) receives the forced pivots */
int hs_potrf_psd(hipStream_t s, int n, double* A, double* dinv, int* flag, const double* diag0, int* regmask, int set_flag)
{
   if ( n <= 0 )
      return HS_OK;
   if ( diag0 != NULL && regmask == NULL && n > NB )
      return HS_ERR_ARG;
   static int rule = getenv("HIPSDP_PIVOT_RULE") != NULL ? atoi(getenv("HIPSDP_PIVOT_RULE")) : 3;
   pd_ext ext = {NULL, NULL, 0.0, NULL, NULL, NULL, (set_flag && n <= NB) ? 1 : 0, rule, regmask};
   const pd_ext* extp = ((diag0 != NULL && regmask != NULL) || ext.set_flag) ? &ext : NULL;
   const long long lda = n;
   const int nblk = (n + NB - 1) / NB;
   static int v1 = getenv("HIPSDP_POTRF_V1") != NULL ? atoi(getenv("HIPSDP_POTRF_V1")) : 0;
   if ( nblk > 1 && !v1 && !g_potrf_force_v1 )
   {
      /* one launch per block column (k_potrf_step) */
      pd_ext est = {NULL, NULL, 0.0, NULL, NULL, NULL, 0, rule, (diag0 != NULL) ? regmask : NULL, 0, 0, NULL};
      for (int b = 0; b < nblk; ++b)
      {
         const int nb = (n - b * NB) < NB ? (n - b * NB) : NB;
         if ( nb <= 16 )
            HS_CALL( launch_potrf_step<16>(s, A, lda, n, b, nblk, dinv, flag, diag0, est) );
         else if ( nb <= 32 )
            HS_CALL( launch_potrf_step<32>(s, A, lda, n, b, nblk, dinv, flag, diag0, est) );
         else
            HS_CALL( launch_potrf_step<64>(s, A, lda, n, b, nblk, dinv, flag, diag0, est) );
      }
      return HS_OK;
   }
   for (int b = 0; b < nblk; ++b)
   {
      const int j0 = b * NB;
      const int nb = (n - j0) < NB ? (n - j0) : NB;
      double* Ajj = A + (long long) j0 * lda + j0;
      double* dj = dinv + (long long) b * NB * NB;
      if ( nb <= 16 )
         HS_CALL( launch_potrf_diag<16>(s, Ajj, lda, nb, j0, dj, flag, diag0, extp) );
      else if ( nb <= 32 )
         HS_CALL( launch_potrf_diag<32>(s, Ajj, lda, nb, j0, dj, flag, diag0, extp) );
      else
         HS_CALL( launch_potrf_diag<64>(s, Ajj, lda, nb, j0, dj, flag, diag0, extp) );
      const int j1 = j0 + nb;
      const int rem = n - j1;
      if ( rem <= 0 )
         break;
      /* panel: P = A[j1:, j0:j1] * inv(L_jj)^T   (in place: one workgroup column covers all nb columns) */
      double* P = A + (long long) j1 * lda + j0;
      /* (the four-launch form keeps the 64 x 64 tile kernel: its summation order is what the fused kernel reproduces bit by bit) */
      hs_gemm_args g1 = {rem, nb, nb, HS_KC, HS_KC, P, lda, 0, dj, NB, 0, P, lda, 0, 1.0, 0.0, 1, HS_GEMM_TILE64, 1, NULL};
      HS_CALL( hs_dgemm(s, &g1) );
      if ( diag0 != NULL && regmask != NULL )
      {
         long long blocks = ((long long) rem * nb + 255) / 256;
         if ( blocks > 1024 ) blocks = 1024;
         hipLaunchKernelGGL(k_zero_forced_cols, dim3((unsigned) blocks), dim3(256), 0, s, rem, nb, P, lda, regmask + j0);
         HS_LAUNCH_CHECK();
      }
      /* trailing update: A22 -= P P^T on the lower triangle */
      double* A22 = A + (long long) j1 * lda + j1;
      hs_gemm_args g2 = {rem, rem, nb, HS_KC, HS_KC, P, lda, 0, P, lda, 0, A22, lda, 0, -1.0, 1.0, 1, HS_GEMM_LOWER | HS_GEMM_TILE64, 1, NULL};
      HS_CALL( hs_dgemm(s, &g2) );
   }
   return HS_OK;
}

__global__ void k_copy_block(const double* __restrict__ src, long long lds_, double* __restrict__ dst, long long ldd, int rows, int cols,
   double scale)
{
   const int total = rows * cols;
   for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x)
   {
      const int r = e / cols, c = e % cols;
      dst[(long long) r * ldd + c] = scale * src[(long long) r * lds_ + c];
   }
}

/* all diagonal blocks of Linv in one launch: block b (64 x 64, identity padded in dinv) -> Linv[64 b .., 64 b ..] */
__global__ void k_copy_diag_blocks(int n, const double* __restrict__ dinv, double* __restrict__ Linv)
{
   const int b = blockIdx.x >> 2, part = blockIdx.x & 3;
   const int i0 = b * NB;
   const int nb = min(NB, n - i0);
   for (int e = part * 1024 + threadIdx.x; e < (part + 1) * 1024; e += 256)
   {
      const int r = e >> 6, c = e & 63;
      if ( r < nb && c < nb )
         Linv[(long long) (i0 + r) * n + i0 + c] = dinv[(long long) b * NB * NB + e];
   }
}

/* Linv = L^-1 by recursive doubling: with the inverses of the 64 x 64 diagonal blocks given, level bs = 64, 128, 256, ... joins
 * neighbouring inverted blocks:  inv [L11 0; L21 L22] = [inv L11, 0; -inv(L22) L21 inv(L11), inv L22].  The pairs of a level are
 * independent and go through ONE batched product each (plus one for a ragged last pair): 2-4 launches per level, log2(n / 64)
 * levels, instead of three launches per 64-row block (n = 500: 12 launches instead of 23). */
int hs_trtri(hipStream_t s, int n, const double* L, const double* dinv, double* Linv, double* tmp)
{
   if ( n <= 0 )
      return HS_OK;
   const long long ld = n;
   HS_CALL( hs_fill(s, Linv, (long long) n * n, 0.0) );
   const int nblk = (n + NB - 1) / NB;
   hipLaunchKernelGGL(k_copy_diag_blocks, dim3(4 * nblk), dim3(256), 0, s, n, dinv, Linv);
   HS_LAUNCH_CHECK();
   static const bool rowwise = getenv("HIPSDP_TRTRI_V1") != NULL;
   if ( rowwise )
   {
      for (int b = 1; b < nblk; ++b)
      {
         const int i0 = b * NB;
         const int nb = (n - i0) < NB ? (n - i0) : NB;
         const double* db = dinv + (long long) b * NB * NB;
         /* tmp[nb x i0] = L[i0:i0+nb, 0:i0] * Linv[0:i0, 0:i0] */
         hs_gemm_args g1 = {nb, i0, i0, HS_KC, HS_MC, L + (long long) i0 * ld, ld, 0, Linv, ld, 0, tmp, (long long) i0, 0, 1.0, 0.0, 1, 0, 1, NULL};
         HS_CALL( hs_dgemm(s, &g1) );
         /* Linv[i0:i0+nb, 0:i0] = - inv(L_bb) * tmp */
         hs_gemm_args g2 = {nb, i0, nb, HS_KC, HS_MC, db, NB, 0, tmp, (long long) i0, 0, Linv + (long long) i0 * ld, ld, 0, -1.0, 0.0, 1, 0, 1, NULL};
         HS_CALL( hs_dgemm(s, &g2) );
      }
      return HS_OK;
   }
   for (long long bs = NB; bs < n; bs *= 2)
   {
      const long long span = 2 * bs;
      const int nfull = (int) (n / span);                         /* pairs whose second block has all bs rows */
      const long long tail = n - (long long) nfull * span;        /* rows after the full pairs */
      const long long stride = span * (ld + 1);
      if ( nfull > 0 )
      {
         /* tmp_p = L[second, first] Linv[first, first];  Linv[second, first] = - Linv[second, second] tmp_p */
         hs_gemm_args g1 = {(int) bs, (int) bs, (int) bs, HS_KC, HS_MC, L + bs * ld, ld, stride, Linv, ld, stride, tmp, bs, bs * bs, 1.0, 0.0,
            nfull, 0, 1, NULL};
         HS_CALL( hs_dgemm(s, &g1) );
         hs_gemm_args g2 = {(int) bs, (int) bs, (int) bs, HS_KC, HS_MC, Linv + bs * (ld + 1), ld, stride, tmp, bs, bs * bs, Linv + bs * ld, ld, stride,
            -1.0, 0.0, nfull, 0, 1, NULL};
         HS_CALL( hs_dgemm(s, &g2) );
      }
      if ( tail > bs )
      {
         const long long o = (long long) nfull * span;             /* first row of the ragged pair */
         const int s2 = (int) (tail - bs);
         double* tq = tmp + (long long) nfull * bs * bs;
         hs_gemm_args g1 = {s2, (int) bs, (int) bs, HS_KC, HS_MC, L + (o + bs) * ld + o, ld, 0, Linv + o * (ld + 1), ld, 0, tq, bs, 0, 1.0, 0.0, 1, 0, 1, NULL};
         HS_CALL( hs_dgemm(s, &g1) );
         hs_gemm_args g2 = {s2, (int) bs, s2, HS_KC, HS_MC, Linv + (o + bs) * (ld + 1), ld, 0, tq, bs, 0, Linv + (o + bs) * ld + o, ld, 0, -1.0, 0.0, 1, 0, 1, NULL};
         HS_CALL( hs_dgemm(s, &g2) );
      }
   }
   return HS_OK;
}

/* ---------------------------------------------------------------------------------------------------------------- */
/* triangular solves with up to 4 right-hand sides: one workgroup of 1024 threads walks the 64-wide block columns    */
/* ---------------------------------------------------------------------------------------------------------------- */

struct __attribute__((aligned(16))) dbl2c { double x, y; };

template<int NRHS>
__global__ void __launch_bounds__(1024) k_trsv(int n, const double* __restrict__ L, const double* __restrict__ dinv,
   double* __restrict__ rhs, long long ldr, int mode)
{
   __shared__ double xs[NRHS][NB];
   __shared__ double rho[NRHS][NB];
   __shared__ double red[64][NB + 1];
   const int tid = threadIdx.x;
   const int grp = tid >> 4;       /* 64 row groups */
   const int part = tid & 15;      /* 16 lanes per row, 4 columns each */
   const int nblk = (n + NB - 1) / NB;
   const long long ld = n;
   const bool refine = (mode & 4) != 0;

   if ( mode & 1 )
   {
      for (int b = 0; b < nblk; ++b)
      {
         const int j0 = b * NB;
         const int nb = (n - j0) < NB ? (n - j0) : NB;
         const double* db = dinv + (long long) b * NB * NB;
         /* x_blk = inv(L_bb) * r_blk : row grp, 16 lanes share the 64-long dot product */
         for (int k = 0; k < NRHS; ++k)
         {
            double sacc = 0.0;
#pragma unroll
            for (int c = 0; c < 4; ++c)
            {
               const int col = 4 * part + c;
               const double rv = (col < nb) ? rhs[(long long) k * ldr + j0 + col] : 0.0;
               sacc += db[grp * NB + col] * rv;
            }
            sacc += __shfl_xor(sacc, 1, 64);
            sacc += __shfl_xor(sacc, 2, 64);
            sacc += __shfl_xor(sacc, 4, 64);
            sacc += __shfl_xor(sacc, 8, 64);
            if ( part == 0 )
               xs[k][grp] = sacc;
         }
         __syncthreads();
         if ( refine )
         {
            /* x = inv(L_bb) r leaves a residual r - L_bb x of the order cond(L_bb) * eps * |r| (the inverse is only a right
             * inverse to rounding); one correction x += inv(L_bb) (r - L_bb x) with the factor itself brings it down to that of a
             * substitution.  The residual of M dy = h is the primal infeasibility the step leaves behind. */
            const double* lb = L + (long long) (j0 + (grp < nb ? grp : 0)) * ld + j0;
            for (int k = 0; k < NRHS; ++k)
            {
               double sacc = 0.0;
#pragma unroll
               for (int c = 0; c < 4; ++c)
               {
                  const int col = 4 * part + c;
                  if ( col <= grp && grp < nb )
                     sacc += lb[col] * xs[k][col];
               }
               sacc += __shfl_xor(sacc, 1, 64);
               sacc += __shfl_xor(sacc, 2, 64);
               sacc += __shfl_xor(sacc, 4, 64);
               sacc += __shfl_xor(sacc, 8, 64);
               if ( part == 0 )
                  rho[k][grp] = (grp < nb) ? rhs[(long long) k * ldr + j0 + grp] - sacc : 0.0;
            }
            __syncthreads();
            for (int k = 0; k < NRHS; ++k)
            {
               double sacc = 0.0;
#pragma unroll
               for (int c = 0; c < 4; ++c)
                  sacc += db[grp * NB + 4 * part + c] * rho[k][4 * part + c];
               sacc += __shfl_xor(sacc, 1, 64);
               sacc += __shfl_xor(sacc, 2, 64);
               sacc += __shfl_xor(sacc, 4, 64);
               sacc += __shfl_xor(sacc, 8, 64);
               if ( part == 0 )
                  xs[k][grp] += sacc;
            }
            __syncthreads();
         }
         if ( tid < NB * NRHS )
         {
            const int k = tid / NB, i = tid % NB;
            if ( i < nb )
               rhs[(long long) k * ldr + j0 + i] = xs[k][i];
         }
         /* r[i] -= L[i, blk] * x_blk for the rows below */
         for (int i = j0 + nb + grp; i < n; i += 64)
         {
            const double* lrow = L + (long long) i * ld + j0 + 4 * part;
            double l0 = 0.0, l1 = 0.0, l2 = 0.0, l3 = 0.0;
            if ( 4 * part + 3 < nb )
            {
               l0 = lrow[0]; l1 = lrow[1]; l2 = lrow[2]; l3 = lrow[3];
            }
            else
            {
               if ( 4 * part + 0 < nb ) l0 = lrow[0];
               if ( 4 * part + 1 < nb ) l1 = lrow[1];
               if ( 4 * part + 2 < nb ) l2 = lrow[2];
            }
#pragma unroll
            for (int k = 0; k < NRHS; ++k)
            {
               double sacc = l0 * xs[k][4 * part] + l1 * xs[k][4 * part + 1] + l2 * xs[k][4 * part + 2] + l3 * xs[k][4 * part + 3];
               sacc += __shfl_xor(sacc, 1, 64);
               sacc += __shfl_xor(sacc, 2, 64);
               sacc += __shfl_xor(sacc, 4, 64);
               sacc += __shfl_xor(sacc, 8, 64);
               if ( part == 0 )
                  rhs[(long long) k * ldr + i] -= sacc;
            }
         }
         __syncthreads();
      }
   }

   if ( mode & 2 )
   {
      for (int b = nblk - 1; b >= 0; --b)
      {
         const int j0 = b * NB;
         const int nb = (n - j0) < NB ? (n - j0) : NB;
         const double* db = dinv + (long long) b * NB * NB;
         for (int k = 0; k < NRHS; ++k)
         {
            /* s[c] = sum_{i below} L[i, j0 + c] * x[i] : partial sums per row group, then across groups */
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            for (int i = j0 + nb + grp; i < n; i += 64)
            {
               const double* lrow = L + (long long) i * ld + j0 + 4 * part;
               const double xv = rhs[(long long) k * ldr + i];
               if ( 4 * part + 3 < nb )
               {
                  a0 += lrow[0] * xv; a1 += lrow[1] * xv; a2 += lrow[2] * xv; a3 += lrow[3] * xv;
               }
               else
               {
                  if ( 4 * part + 0 < nb ) a0 += lrow[0] * xv;
                  if ( 4 * part + 1 < nb ) a1 += lrow[1] * xv;
                  if ( 4 * part + 2 < nb ) a2 += lrow[2] * xv;
               }
            }
            red[grp][4 * part + 0] = a0;
            red[grp][4 * part + 1] = a1;
            red[grp][4 * part + 2] = a2;
            red[grp][4 * part + 3] = a3;
            __syncthreads();
            if ( tid < NB )
            {
               double sacc = 0.0;
               for (int g = 0; g < 64; ++g)
                  sacc += red[g][tid];
               const double yv = (tid < nb) ? rhs[(long long) k * ldr + j0 + tid] : 0.0;
               xs[k][tid] = yv - sacc;
            }
            __syncthreads();
         }
         /* x_blk = inv(L_bb)^T * (y_blk - s) */
         double x0[NRHS];
         for (int k = 0; k < NRHS; ++k)
         {
            double sacc = 0.0;
#pragma unroll
            for (int c = 0; c < 4; ++c)
            {
               const int row = 4 * part + c;
               sacc += db[row * NB + grp] * ((row < nb) ? xs[k][row] : 0.0);
            }
            sacc += __shfl_xor(sacc, 1, 64);
            sacc += __shfl_xor(sacc, 2, 64);
            sacc += __shfl_xor(sacc, 4, 64);
            sacc += __shfl_xor(sacc, 8, 64);
            x0[k] = sacc;
            if ( part == 0 )
            {
               if ( !refine )
               {
                  if ( grp < nb )
                     rhs[(long long) k * ldr + j0 + grp] = sacc;
               }
               else
                  rho[k][grp] = (grp < nb) ? sacc : 0.0;         /* x0, for the residual */
            }
         }
         __syncthreads();
         if ( refine )
         {
            /* v - L_bb^T x0 (column grp of the block, rows >= grp), then x = x0 + inv(L_bb)^T (..) */
            double rv[NRHS];
            for (int k = 0; k < NRHS; ++k)
            {
               double sacc = 0.0;
#pragma unroll
               for (int c = 0; c < 4; ++c)
               {
                  const int row = 4 * part + c;
                  if ( row >= grp && row < nb )
                     sacc += L[(long long) (j0 + row) * ld + j0 + grp] * rho[k][row];
               }
               sacc += __shfl_xor(sacc, 1, 64);
               sacc += __shfl_xor(sacc, 2, 64);
               sacc += __shfl_xor(sacc, 4, 64);
               sacc += __shfl_xor(sacc, 8, 64);
               rv[k] = (grp < nb) ? xs[k][grp] - sacc : 0.0;
            }
            __syncthreads();
            if ( part == 0 )
            {
               for (int k = 0; k < NRHS; ++k)
                  xs[k][grp] = rv[k];
            }
            __syncthreads();
            for (int k = 0; k < NRHS; ++k)
            {
               double sacc = 0.0;
#pragma unroll
               for (int c = 0; c < 4; ++c)
               {
                  const int row = 4 * part + c;
                  sacc += db[row * NB + grp] * xs[k][row];
               }
               sacc += __shfl_xor(sacc, 1, 64);
               sacc += __shfl_xor(sacc, 2, 64);
               sacc += __shfl_xor(sacc, 4, 64);
               sacc += __shfl_xor(sacc, 8, 64);
               if ( part == 0 && grp < nb )
                  rhs[(long long) k * ldr + j0 + grp] = x0[k] + sacc;
            }
            __syncthreads();
         }
      }
   }
}

/* x = (L L^T)^-1 r for a single-block factor (n <= 64) by substitution in the oracle's order, one right-hand side per wavefront
 * (hs_kernels.h: hs_wl_msolve) */
__global__ void __launch_bounds__(256) k_msolve_sub64(int n, const double* __restrict__ L, double* __restrict__ rhs, long long ldr)
{
   __shared__ double sL[64 * 65];
   for (int e = threadIdx.x; e < n * n; e += blockDim.x)
   {
      const int i = e / n, j = e - i * n;
      if ( j <= i )
         sL[i * 65 + j] = L[(long long) i * n + j];
   }
   __syncthreads();
   const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
   double* v = rhs + (long long) k * ldr;
   const double x = hs_wl_msolve(sL, n, lane, lane < n ? v[lane] : 0.0);
   if ( lane < n )
      v[lane] = x;
}

/* the same for 64 < n <= 128: two rows per lane, the factor (zeros above the diagonal) in 129 KB of dynamic LDS */
__global__ void __launch_bounds__(256) k_msolve_sub128(int n, const double* __restrict__ L, double* __restrict__ rhs, long long ldr)
{
   extern __shared__ __attribute__((aligned(16))) double sL2[];
   for (int e = threadIdx.x; e < n * n; e += blockDim.x)
   {
      const int i = e / n, j = e - i * n;
      sL2[i * 129 + j] = (j <= i) ? L[(long long) i * n + j] : 0.0;
   }
   __syncthreads();
   const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
   double* v = rhs + (long long) k * ldr;
   double x[2] = {v[lane], lane + 64 < n ? v[lane + 64] : 0.0};
   hs_wl2_msolve<129>(sL2, n, lane, x);
   v[lane] = x[0];
   if ( lane + 64 < n )
      v[lane + 64] = x[1];
}

int hs_trsv(hipStream_t s, int n, const double* L, const double* dinv, int nrhs, double* rhs, long long ldr, int mode)
{
   if ( n <= 0 || nrhs <= 0 )
      return HS_OK;
   if ( nrhs > 4 )
      return HS_ERR_ARG;
   if ( n <= 64 && mode == 7 && hs_small_solve_by_substitution() )
   {
      hipLaunchKernelGGL(k_msolve_sub64, dim3(1), dim3(64 * nrhs), 0, s, n, L, rhs, ldr);
      HS_LAUNCH_CHECK();
      return HS_OK;
   }
   if ( n <= 128 && mode == 7 && hs_small_solve_by_substitution() )
   {
      static hs_attr_mask attr_done;
      const int smem = 128 * 129 * (int) sizeof(double);
      HS_CALL( hs_func_max_lds(reinterpret_cast<const void*>(&k_msolve_sub128), smem, &attr_done) );
      hipLaunchKernelGGL(k_msolve_sub128, dim3(1), dim3(64 * nrhs), smem, s, n, L, rhs, ldr);
      HS_LAUNCH_CHECK();
      return HS_OK;
   }
   switch ( nrhs )
   {
   case 1: hipLaunchKernelGGL((k_trsv<1>), dim3(1), dim3(1024), 0, s, n, L, dinv, rhs, ldr, mode); break;
   case 2: hipLaunchKernelGGL((k_trsv<2>), dim3(1), dim3(1024), 0, s, n, L, dinv, rhs, ldr, mode); break;
   case 3: hipLaunchKernelGGL((k_trsv<3>), dim3(1), dim3(1024), 0, s, n, L, dinv, rhs, ldr, mode); break;
   default: hipLaunchKernelGGL((k_trsv<4>), dim3(1), dim3(1024), 0, s, n, L, dinv, rhs, ldr, mode); break;
   }
   HS_LAUNCH_CHECK();
   return HS_OK;
}

/* ---- triangular solves across workgroups -------------------------------------------------------------------------- */
/* One workgroup per 64-row block, all co-resident (the grid is at most a few hundred small workgroups).  Forward: block b
 * accumulates r_b - sum_{c<b} L_bc x_c as the x_c become available, then x_b = inv(L_bb) (..), publishes x_b and raises
 * its flag; backward the same with c > b and transposed blocks.  A block waits only on blocks that are strictly earlier
 * in the dependency order, so any dispatch order terminates; the flags carry the epoch of the call (no reset between
 * calls), the hand-off goes through coherent (sc1) stores and loads (see trsv_publish), and every spin
 * is bounded: on expiry the error word behind the flags is set and the block leaves without publishing (later blocks then
 * expire too, the launch ends, and the interior-point loop stops on the non-finite step it gets).  The next L block is
 * requested before the flag of the current one is awaited. */
#define TRSV_SPIN_LIMIT (1 << 16)

/* What one block hands to the others - its 64 x NRHS solution entries - goes through an exchange vector of its own, stored
 * and loaded as relaxed agent-scope atomics (sc1: served at the device's coherence point), and the data is its own signal:
 * the vector is all NaN before the launch, a reader polls each entry it needs until it is a number (one round trip when it
 * is already there).  [Flags with a release / acquire pair at agent scope write back and invalidate the L2 of an XCD each
 * time - eight XCDs, one L2 each - and even with coherent stores a flag costs three round trips per block: wait for the
 * stores, raise the flag, poll it.]  Two exchange vectors alternate with the parity of the launch epoch; every block wipes its
 * own entries of the other one when it starts (the launch before is over, the next one has not begun).  A solution entry
 * that is NaN by arithmetic lets the readers wait until their bound and ends the launch as a failure, like a dead block. */
__device__ __forceinline__ void trsv_publish(double* p, double v)
{
   __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ double trsv_fetch(const double* p, int* ok)
{
   double v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
   int spins = 0;
   while ( v != v )
   {
      if ( ++spins > TRSV_SPIN_LIMIT )
      {
         *ok = 0;
         break;
      }
      __builtin_amdgcn_s_sleep(1);
      v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
   }
   return v;
}

/* layout of the workspace (ints): [0] error word, [8 ...] two exchange vectors of 4 x npad doubles (npad = 64 * blocks) */
__device__ __forceinline__ double* trsv_xvec(int* ws, int gen, int npad)
{
   return reinterpret_cast<double*>(ws + 8) + (long long) gen * 4 * npad;
}

template<int NRHS>
__global__ void __launch_bounds__(256) k_trsv_fwd(int n, const double* __restrict__ L, const double* __restrict__ dinv,
   double* __restrict__ rhs, long long ldr, int* __restrict__ ws, int epoch, int refine)
{
   __shared__ double xs[NRHS][NB];
   __shared__ double x0s[NRHS][NB];
   __shared__ int ok;
   const int b = blockIdx.x;
   const int tid = threadIdx.x;
   const int row = tid >> 2, q = tid & 3;          /* 64 rows, 4 lanes per row, 16 columns each */
   const long long ld = n;
   const int j0 = b * NB;
   const int nb = (n - j0) < NB ? (n - j0) : NB;
   const int npad = gridDim.x * NB;
   /* blockIdx.y: right-hand side(s) of this chain of workgroups (NRHS each); the chains of a launch do not meet */
   rhs += (long long) blockIdx.y * NRHS * ldr;
   double* xg = trsv_xvec(ws, epoch & 1, npad) + (long long) blockIdx.y * NRHS * npad;
   trsv_publish(trsv_xvec(ws, (epoch & 1) ^ 1, npad) + (long long) (tid >> 6) * npad + j0 + (tid & 63), __builtin_nan(""));
   if ( tid == 0 )
      ok = 1;
   double acc[NRHS];
#pragma unroll
   for (int k = 0; k < NRHS; ++k)
      acc[k] = 0.0;
   const bool rowok = row < nb;
   const double* lrow = L + (long long) (j0 + (rowok ? row : 0)) * ld + 16 * q;
   double lcur[16], lnext[16];
   /* what the diagonal solve needs of inv(L_bb) and (correction) of L_bb does not depend on the blocks before: requested now,
    * not when the last of them has arrived */
   double dbr[16], lbr[16];
   {
      const double* db = dinv + (long long) b * NB * NB + row * NB + 16 * q;
#pragma unroll
      for (int c = 0; c < 16; ++c)
         dbr[c] = db[c];
      if ( refine )
      {
#pragma unroll
         for (int c = 0; c < 16; ++c)
            lbr[c] = (rowok && 16 * q + c <= row) ? lrow[j0 + c] : 0.0;
      }
   }
   if ( b > 0 )
   {
#pragma unroll
      for (int c = 0; c < 16; ++c)
         lcur[c] = lrow[c];
   }
   for (int cb = 0; cb < b; ++cb)
   {
      if ( cb + 1 < b )
      {
#pragma unroll
         for (int c = 0; c < 16; ++c)
            lnext[c] = lrow[(long long) (cb + 1) * NB + c];
      }
      if ( tid < NB * NRHS )
         xs[tid / NB][tid % NB] = trsv_fetch(xg + (long long) (tid / NB) * npad + cb * NB + (tid % NB), &ok);
      __syncthreads();
      if ( !ok )
      {
         /* giving up (a block before this one never published: not co-resident, or a NaN by arithmetic): the block's rows of
          * the right-hand side(s) become NaN, so the interior-point loop sees a non-finite step and ends the solve as a numerical
          * failure instead of using a silently wrong dy; the blocks behind this one run into their bound the same way */
         if ( tid < NB * NRHS && (tid % NB) < nb )
            rhs[(long long) (tid / NB) * ldr + j0 + (tid % NB)] = __builtin_nan("");
         if ( tid == 0 )
            atomicExch(ws, 1);
         return;
      }
#pragma unroll
      for (int k = 0; k < NRHS; ++k)
      {
         double sacc = 0.0;
#pragma unroll
         for (int c = 0; c < 16; ++c)
            sacc += lcur[c] * xs[k][16 * q + c];
         acc[k] += sacc;
      }
#pragma unroll
      for (int c = 0; c < 16; ++c)
         lcur[c] = lnext[c];
      __syncthreads();
   }
   /* y = r_b - acc (reduced over the 4 lanes of a row), x_b = inv(L_bb) y */
#pragma unroll
   for (int k = 0; k < NRHS; ++k)
   {
      double sacc = acc[k];
      sacc += __shfl_xor(sacc, 1, 64);
      sacc += __shfl_xor(sacc, 2, 64);
      if ( q == 0 )
         xs[k][row] = rowok ? rhs[(long long) k * ldr + j0 + row] - sacc : 0.0;
   }
   __syncthreads();
#pragma unroll
   for (int k = 0; k < NRHS; ++k)
   {
      double sacc = 0.0;
#pragma unroll
      for (int c = 0; c < 16; ++c)
         sacc += dbr[c] * xs[k][16 * q + c];
      sacc += __shfl_xor(sacc, 1, 64);
      sacc += __shfl_xor(sacc, 2, 64);
      if ( refine )
      {
         if ( q == 0 )
            x0s[k][row] = rowok ? sacc : 0.0;
      }
      else if ( q == 0 )
      {
         if ( rowok )
            rhs[(long long) k * ldr + j0 + row] = sacc;
         trsv_publish(xg + (long long) k * npad + j0 + row, rowok ? sacc : 0.0);
      }
   }
   __syncthreads();
   if ( refine )
   {
      /* one correction with the factor itself (see k_trsv): x = x0 + inv(L_bb) (y - L_bb x0) */
      double rv[NRHS];
#pragma unroll
      for (int k = 0; k < NRHS; ++k)
      {
         double sacc = 0.0;
#pragma unroll
         for (int c = 0; c < 16; ++c)
            sacc += lbr[c] * x0s[k][16 * q + c];
         sacc += __shfl_xor(sacc, 1, 64);
         sacc += __shfl_xor(sacc, 2, 64);
         rv[k] = rowok ? xs[k][row] - sacc : 0.0;
      }
      __syncthreads();
      if ( q == 0 )
      {
#pragma unroll
         for (int k = 0; k < NRHS; ++k)
            xs[k][row] = rv[k];
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < NRHS; ++k)
      {
         double sacc = 0.0;
#pragma unroll
         for (int c = 0; c < 16; ++c)
            sacc += dbr[c] * xs[k][16 * q + c];
         sacc += __shfl_xor(sacc, 1, 64);
         sacc += __shfl_xor(sacc, 2, 64);
         if ( q == 0 )
         {
            const double xv = x0s[k][row] + sacc;
            if ( rowok )
               rhs[(long long) k * ldr + j0 + row] = xv;
            trsv_publish(xg + (long long) k * npad + j0 + row, rowok ? xv : 0.0);
         }
      }
   }
}

template<int NRHS>
__global__ void __launch_bounds__(256) k_trsv_bwd(int n, const double* __restrict__ L, const double* __restrict__ dinv,
   double* __restrict__ rhs, long long ldr, int* __restrict__ ws, int epoch, int refine)
{
   __shared__ double xs[NRHS][NB];
   __shared__ double x0s[NRHS][NB];
   __shared__ double red[4][NRHS][NB];
   __shared__ int ok;
   const int nblk = gridDim.x;
   const int b = blockIdx.x;
   const int tid = threadIdx.x;
   const int col = tid & 63, g = tid >> 6;          /* column of the block, 4 row groups of 16 rows */
   const long long ld = n;
   const int j0 = b * NB;
   const int nb = (n - j0) < NB ? (n - j0) : NB;
   const int npad = nblk * NB;
   rhs += (long long) blockIdx.y * NRHS * ldr;
   double* xg = trsv_xvec(ws, epoch & 1, npad) + (long long) blockIdx.y * NRHS * npad;
   trsv_publish(trsv_xvec(ws, (epoch & 1) ^ 1, npad) + (long long) (tid >> 6) * npad + j0 + (tid & 63), __builtin_nan(""));
   if ( tid == 0 )
      ok = 1;
   const bool colok = col < nb;
   double acc[NRHS];
#pragma unroll
   for (int k = 0; k < NRHS; ++k)
      acc[k] = 0.0;
   /* the diagonal solve's part of inv(L_bb) and (correction) of L_bb: requested before the blocks behind are awaited */
   double dbr[16], lbr[16];
   {
      const double* db = dinv + (long long) b * NB * NB;
#pragma unroll
      for (int c = 0; c < 16; ++c)
         dbr[c] = db[(16 * g + c) * NB + col];
      if ( refine )
      {
#pragma unroll
         for (int c = 0; c < 16; ++c)
         {
            const int i = 16 * g + c;
            lbr[c] = (i >= col && i < nb) ? L[(long long) (j0 + i) * ld + j0 + col] : 0.0;
         }
      }
   }
   /* s[col] = sum over later blocks cb, rows i of the block: L[cb * 64 + i][j0 + col] * x[cb * 64 + i] */
   for (int cb = nblk - 1; cb > b; --cb)
   {
      const int i0 = cb * NB;
      const int nbc = (n - i0) < NB ? (n - i0) : NB;
      double lv[16];
#pragma unroll
      for (int c = 0; c < 16; ++c)
      {
         const int i = 16 * g + c;
         lv[c] = (i < nbc && colok) ? L[(long long) (i0 + i) * ld + j0 + col] : 0.0;
      }
      if ( tid < NB * NRHS )
         xs[tid / NB][tid % NB] = trsv_fetch(xg + (long long) (tid / NB) * npad + i0 + (tid % NB), &ok);
      __syncthreads();
      if ( !ok )
      {
         /* giving up (a block before this one never published: not co-resident, or a NaN by arithmetic): the block's rows of
          * the right-hand side(s) become NaN, so the interior-point loop sees a non-finite step and ends the solve as a numerical
          * failure instead of using a silently wrong dy; the blocks behind this one run into their bound the same way */
         if ( tid < NB * NRHS && (tid % NB) < nb )
            rhs[(long long) (tid / NB) * ldr + j0 + (tid % NB)] = __builtin_nan("");
         if ( tid == 0 )
            atomicExch(ws, 1);
         return;
      }
#pragma unroll
      for (int k = 0; k < NRHS; ++k)
      {
         double sacc = 0.0;
#pragma unroll
         for (int c = 0; c < 16; ++c)
            sacc += lv[c] * xs[k][16 * g + c];
         acc[k] += sacc;
      }
      __syncthreads();
   }
#pragma unroll
   for (int k = 0; k < NRHS; ++k)
      red[g][k][col] = acc[k];
   __syncthreads();
   if ( tid < NB * NRHS )
   {
      const int k = tid / NB, i = tid % NB;
      const double sacc = red[0][k][i] + red[1][k][i] + red[2][k][i] + red[3][k][i];
      xs[k][i] = (i < nb) ? rhs[(long long) k * ldr + j0 + i] - sacc : 0.0;
   }
   __syncthreads();
   /* x_b = inv(L_bb)^T v : out[col] = sum_i dinv[i][col] v[i] */
#pragma unroll
   for (int k = 0; k < NRHS; ++k)
   {
      double sacc = 0.0;
#pragma unroll
      for (int c = 0; c < 16; ++c)
         sacc += dbr[c] * xs[k][16 * g + c];
      red[g][k][col] = sacc;
   }
   __syncthreads();
   if ( tid < NB * NRHS )
   {
      const int k = tid / NB, i = tid % NB;
      const double v = red[0][k][i] + red[1][k][i] + red[2][k][i] + red[3][k][i];
      if ( refine )
         x0s[k][i] = (i < nb) ? v : 0.0;
      else
      {
         if ( i < nb )
            rhs[(long long) k * ldr + j0 + i] = v;
         trsv_publish(xg + (long long) k * npad + j0 + i, (i < nb) ? v : 0.0);
      }
   }
   __syncthreads();
   if ( refine )
   {
      /* one correction with the factor itself (see k_trsv): x = x0 + inv(L_bb)^T (v - L_bb^T x0) */
#pragma unroll
      for (int k = 0; k < NRHS; ++k)
      {
         double sacc = 0.0;
#pragma unroll
         for (int c = 0; c < 16; ++c)
            sacc += lbr[c] * x0s[k][16 * g + c];
         red[g][k][col] = sacc;
      }
      __syncthreads();
      if ( tid < NB * NRHS )
      {
         const int k = tid / NB, i = tid % NB;
         const double sacc = red[0][k][i] + red[1][k][i] + red[2][k][i] + red[3][k][i];
         xs[k][i] = (i < nb) ? xs[k][i] - sacc : 0.0;
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < NRHS; ++k)
      {
         double sacc = 0.0;
#pragma unroll
         for (int c = 0; c < 16; ++c)
            sacc += dbr[c] * xs[k][16 * g + c];
         red[g][k][col] = sacc;
      }
      __syncthreads();
      if ( tid < NB * NRHS )
      {
         const int k = tid / NB, i = tid % NB;
         const double xv = x0s[k][i] + (red[0][k][i] + red[1][k][i] + red[2][k][i] + red[3][k][i]);
         if ( i < nb )
            rhs[(long long) k * ldr + j0 + i] = xv;
         trsv_publish(xg + (long long) k * npad + j0 + i, (i < nb) ? xv : 0.0);
      }
   }
}

/* number of ints of the workspace of hs_trsv_sync for an n x n factor: an error word and two exchange vectors (see
 * trsv_xvec); hs_trsv_sync_init once after allocation (and after a launch that gave up) */
long long hs_trsv_sync_ws(int n)
{
   const long long npad = (long long) ((n + NB - 1) / NB) * NB;
   return 8 + 2 * (2 * 4 * npad);
}

int hs_trsv_sync_init(hipStream_t s, int n, int* sync_ws, int* epoch)
{
   HS_HIP( hipMemsetAsync(sync_ws, 0xFF, (size_t) hs_trsv_sync_ws(n) * sizeof(int), s) );      /* all NaN */
   HS_HIP( hipMemsetAsync(sync_ws, 0, 8 * sizeof(int), s) );
   if ( epoch != NULL )
      *epoch = 0;
   return HS_OK;
}

/* the multi-workgroup solve; sync_ws from hs_trsv_sync_ws, *epoch is advanced by the call (start it at 0) */
int hs_trsv_sync(hipStream_t s, int n, const double* L, const double* dinv, int nrhs, double* rhs, long long ldr, int mode,
   int* sync_ws, int* epoch)
{
   if ( n <= 0 || nrhs <= 0 )
      return HS_OK;
   if ( nrhs > 4 )
      return HS_ERR_ARG;
   const int nblk = (n + NB - 1) / NB;
   const int refine = (mode & 4) ? 1 : 0;
   /* the blocks wait for each other: all workgroups must fit on the device at once (checked once against its CU count) */
   static int per_cu_cached = -1;               /* occupancy of the kernel: a property of the code object, the same on every gfx950 */
   if ( per_cu_cached < 0 )
   {
      int per_cu = 0;
      if ( hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&k_trsv_bwd<1>), 256, 0) != hipSuccess )
         per_cu = 0;
      per_cu_cached = per_cu;
   }
   const int max_blocks = per_cu_cached * hs_device_cus();
   if ( sync_ws == NULL || epoch == NULL || nblk > 512 || nblk * nrhs > max_blocks || nblk < 3 )
      return hs_trsv(s, n, L, dinv, nrhs, rhs, ldr, mode);
   /* one chain of workgroups per right-hand side: the chains run side by side (a block's solve for three right-hand sides in
    * one workgroup took twice as long as for one) */
   if ( mode & 1 )
   {
      const int e = ++(*epoch);
      hipLaunchKernelGGL((k_trsv_fwd<1>), dim3(nblk, nrhs), dim3(256), 0, s, n, L, dinv, rhs, ldr, sync_ws, e, refine);
      HS_LAUNCH_CHECK();
   }
   if ( mode & 2 )
   {
      const int e = ++(*epoch);
      hipLaunchKernelGGL((k_trsv_bwd<1>), dim3(nblk, nrhs), dim3(256), 0, s, n, L, dinv, rhs, ldr, sync_ws, e, refine);
      HS_LAUNCH_CHECK();
   }
   return HS_OK;
}


