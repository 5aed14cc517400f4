/* solve1.hip - a whole node solve in ONE launch of ONE workgroup (B&B-sized problems).
 *
 * What it replaces: the third-party solve call of the reference's backends - DSDPSetup / DSDPSolve / DSDPComputeX at
 * src/sdpi/sdpisolver_dsdp.c:1489-1520, SDPA::initializeSolve / solve at src/sdpi/sdpisolver_sdpa.cpp:1600-1670 - for the problem
 * sizes every instance the reference ships has (example_TT: one 10 x 10 block, 37 variables, 85 LP rows; example_CLS: 43 x 43, 33
 * variables).  There the reference solves a node in-process with no device hop; the general path of csrc/ipm.hip needs about
 * twenty dependent launches and three host read-backs per interior-point iteration, i.e. the solve is bound by launch-to-launch
 * and read-back latency, not by arithmetic.  Here the state of the whole homogeneous self-dual iteration lives in the 160 KiB of
 * LDS of one compute unit: one launch, termination / stall / certificate decisions on the device, one read-back per solve.
 *
 * Algorithm: exactly the iteration of csrc/ipm.hip / oracle/ipm_ref.py (HSD embedding, HKM direction, Mehrotra
 * predictor-corrector, factored elimination of dtau, semidefinite pivot rule for M, corrected triangular solves, exact
 * smallest eigenvalues for the step lengths); the summation orders differ, the results agree to rounding.
 *
 * Layout of a solve (512 threads = 8 wavefronts on one CU: 256 registers per thread, nothing spills):
 *   - the constraint matrices are scanned once into two nonzero lists (by variable: p >= q entries of A_i; by position: which
 *     variables touch entry (r, c)) and the LP rows into row lists and column lists - every instance of the reference has 1-10
 *     nonzeros per matrix and mostly bound rows; a kernel that finds too much work for one CU declines (status -2) and the
 *     caller takes the general path;
 *   - "hot" n x n matrices (X, Z^-1, the two inverse Cholesky factors, dX, dZ, two temporaries) and everything of size m, q,
 *     m x m always sit in LDS; "cold" ones (Z, Rd, E, B) and the lists go to LDS while it lasts, else to an L2-resident
 *     workspace (flat addressing: the same code serves both);
 *   - n x n x n products: 16 x 16 tiles of v_mfma_f64_16x16x4_f64, one wavefront per tile, operands straight from LDS;
 *   - dependent recurrences (Cholesky, triangular inverse, Householder tridiagonalisation + Sturm multisection, the solves with
 *     the factor of M) run inside ONE wavefront each, without workgroup barriers, several of them side by side on different
 *     wavefronts (X side / Z side / blocks; the factorization of M beside the first product of the predictor);
 *   - Schur complement from the nonzeros: U_j = X A_j Z^-1 as a sum of rank-one terms per nonzero of A_j (several j side by side
 *     in the scratch region), M_ij = <A_i, U_j> over the nonzeros of A_i.
 */
#include "hs_kernels.h"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <cstdio>

#ifndef S1_NT
#define S1_NT 512
#define S1_NW 8
#endif
#define S1_MAXB HS_S1_MAXBLK
#define S1_MAXM 128                                         /* (above 64: two rows per lane in the factorization of M and the substitutions) */
#define S1_MAXN 64
#define S1_STATIC_LDS 5632                                  /* bytes kept for the static arrays below */
#define S1_DYN_LDS (160 * 1024 - S1_STATIC_LDS)
#define S1_NRED 24
#define S1_LIGHT_MAX 24                                       /* a matrix with at most this many entries (both triangles) is "light" */
/* every lambda of the kernel is inlined: a lambda that stays a function keeps what it captures by reference in scratch memory */
#define S1_INL __attribute__((always_inline))
#ifdef S1_DEBUG
/* Debug build (make EXTRA=-DS1_DEBUG, tests/devtools/solve1_debug.sh; round 5, after the unexplained anomaly of DESIGN 7.5): the whole
 * dynamic LDS, the static scalars and the workspace start as NaN - a read of something the solve never wrote ends in a NaN result
 * instead of whatever the previous solve left there -, every wave-level synchronisation is a full workgroup-scope fence, and every
 * value moved to scalar registers as "computed identically by every lane" is checked to be so (s1_dbg[0] counts the violations).
 * Results must be the release build's bit for bit. */
#define S1_WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); } while (0)
#else
/* Release form: the hardware serves a wavefront's LDS instructions in order, so lanes that exchange data through LDS need the wait for
 * the LDS counter and nothing else - but the COMPILER has to be told as well that memory changes hands here (as pd_wave_sync of
 * chol.hip tells it): without the two wavefront-scope fences nothing forbids it to keep a value loaded before the exchange, or to move
 * a load across it.  Round 6 (profiles/r06_solve1_debug_release_root_cause.txt): the variant of commit f33223b
 * whose debug build walked other iterates than its release build does so with -ffp-contract=off as well (not contraction), its release
 * build gives the same bits with the full hardware fences of the debug form (not a missing hardware ordering) - and its DEBUG build gives
 * other bits on the same five shapes depending on which form of this macro it uses: what varied was the freedom this macro left the
 * compiler.  The fences below take it away; they emit no instruction. */
#define S1_WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier(); \
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#endif

/* ---- size classes (round 5).  The kernel is compiled once per class, each instance with the code of its class only (rounds 4's
 * single kernel held every variant: 687 KB of code, 256 registers with 403 spilled ones - and a function that no failing shape
 * called changed their results).  The host picks the instance (solve1.hip: hs_solve1_launch):
 *    S1_NCLS  10: every block has at most 10 rows (the register forms s1u_*; example_TT, example_small, ...)
 *             16: every block at most 16 rows
 *             64: anything the kernel is offered
 *    S1_MBIG  1: 64 < m <= 128 possible (two rows per lane in the factorization of M and the substitutions) */
#ifndef S1_NCLS
#define S1_NCLS 64
#endif
#ifndef S1_MBIG
#define S1_MBIG 1
#endif
#define S1_ALLU  (S1_NCLS <= 10)
#define S1_ALL16 (S1_NCLS <= 16)

/* vectors of length m + 1 and of length q in LDS */
enum { V_b = 0, V_y, V_rp, V_AX, V_AH, V_g, V_w, V_ub, V_u2, V_u1, V_h, V_dy, V_wt, V_cv, V_dg, V_t1, V_t2, V_t3, V_t4, V_COUNT };
enum { Q_x = 0, Q_z, Q_rd, Q_beta, Q_hl, Q_dx, Q_dz, Q_elp, Q_sx, Q_COUNT };
/* partial sums of a phase, one row per wavefront */
enum { RS_XZ = 0, RS_RD2LP, RS_RDMAX, RS_S0, RS_BH, RS_HD2, RS_RATX, RS_RATZ, RS_NC2, RS_WORK, RS_RP2, RS_HP2, RS_DOB, RS_BLK0, RS_END = RS_BLK0 + S1_MAXB };
static_assert(RS_END <= S1_NRED, "reduction slots");
/* scalars in LDS */
enum { SC_TAU = 0, SC_KAPPA, SC_RP2, SC_HP2, SC_DOBJ, SC_BUB, SC_BU1, SC_WRP, SC_DTAU, SC_DKAPPA, SC_NORMB, SC_NORMC, SC_FAIL, SC_XI,
       SC_LMIN0, SC_COUNT = SC_LMIN0 + 2 * S1_MAXB };

struct S1Lay
{
   int m, m1, q, K;
   int pm1, pm, VL, QL;
   int packedM;            /* m > 64: the extended Schur matrix as a packed lower triangle (row i at i (i + 1) / 2) */
   int oMx, oLm, oVec, oQ, oR, Rlen, fixedEnd;
   int n[S1_MAXB], p[S1_MAXB], np[S1_MAXB];
   int oX[S1_MAXB], oZi[S1_MAXB], oLx[S1_MAXB], oLz[S1_MAXB], odX[S1_MAXB], odZ[S1_MAXB], oT1[S1_MAXB], oT2[S1_MAXB], oEig[S1_MAXB];
};

/* the part of LDS whose place follows from the shape alone (doubles); returns its length */
static __host__ __device__ inline int s1_layout(int m, int q, int K, const int* n, S1Lay& L)
{
   L.m = m; L.m1 = m + 1; L.q = q; L.K = K;
   /* the factor of M overwrites M in place (rows / columns 1 .. m of the extended matrix): the pitch covers column 0 and whole panels
    * of eight columns (the substitutions read them unmasked; the padding stays zero from the start of the solve) */
   L.pm1 = (((m + 7) & ~7) + 1) | 1; L.pm = L.pm1;
   L.VL = (m + 2) & ~1; L.QL = (q + 1) & ~1;
   int o = 0, sum = 0;
   for (int k = 0; k < K; ++k)
   {
      L.n[k] = n[k]; L.p[k] = n[k] | 1; L.np[k] = n[k] * L.p[k];
      sum += L.np[k];
   }
   for (int k = 0; k < K; ++k) { L.oX[k] = o; o += L.np[k]; }
   for (int k = 0; k < K; ++k) { L.oZi[k] = o; o += L.np[k]; }
   for (int k = 0; k < K; ++k) { L.oLx[k] = o; o += L.np[k]; }
   for (int k = 0; k < K; ++k) { L.oLz[k] = o; o += L.np[k]; }
   for (int k = 0; k < K; ++k) { L.oEig[k] = o; o += 8 * ((n[k] + 1) & ~1) + 64; }
   /* m > 64 (the instance with two rows per lane): M as a packed lower triangle - 45 instead of 96 KB at m = 105, so that the
    * lists of such a problem find room in LDS again; sixteen doubles behind it for the unmasked reads past the end of the last rows */
   L.packedM = m > 64 ? 1 : 0;
   L.oMx = o; o += L.packedM ? (L.m1 * (L.m1 + 1) / 2 + 16 + 1) & ~1 : (L.m1 * L.pm1 + 1) & ~1;
   L.oLm = L.oMx + L.pm1 + 1;
   L.oVec = o; o += V_COUNT * L.VL;
   L.oQ = o; o += Q_COUNT * L.QL;
   /* scratch region (dX, dZ, T1, T2 of all blocks, contiguous; the Schur phase uses it as a pool of U_j buffers) last, so that
    * spare LDS can extend it */
   o = (o + 1) & ~1;
   L.oR = o;
   for (int k = 0; k < K; ++k) { L.odX[k] = o; o += L.np[k]; }
   for (int k = 0; k < K; ++k) { L.odZ[k] = o; o += L.np[k]; }
   for (int k = 0; k < K; ++k) { L.oT1[k] = o; o += L.np[k]; }
   for (int k = 0; k < K; ++k) { L.oT2[k] = o; o += L.np[k]; }
   L.Rlen = 4 * sum;
   L.fixedEnd = o;
   return o;
}

#ifdef S1_HOST_PART
int hs_solve1_fits(int m, int q, int nblk, const int* n)
{
   if ( m < 1 || m > S1_MAXM || nblk < 1 || nblk > S1_MAXB || q < 0 || q > 4096 )
      return 0;
   for (int k = 0; k < nblk; ++k)
      if ( n[k] < 1 || n[k] > S1_MAXN )
         return 0;
   S1Lay L;
   const int len = s1_layout(m, q, nblk, n, L);
   return (long long) len * 8 <= S1_DYN_LDS ? 1 : 0;
}

/* worst case of everything that may have to live outside LDS: cold matrices, offset arrays, full lists */
long long hs_solve1_ws_doubles(int m, int q, int nblk, const int* n)
{
   const long long m1 = m + 1;
   long long t = 64;
   for (int k = 0; k < nblk; ++k)
   {
      const long long nn = n[k], np = nn * (nn | 1), n2 = nn * nn, nlow = nn * (nn + 1) / 2;
      t += 5 * np + 8;
      t += (m1 + 2) / 2 + 1 + (n2 + 2) / 2 + 1;                      /* voff, poff (ints) */
      t += (m1 * n2 + 2) + (m1 * nlow + 2) + (m1 * n2 + 2) / 2 + (m1 * nlow + 4) / 4 + 8;      /* vval, pval, vpq (u32), pvar (u16) */
      t += 2 * ((m1 + 3) / 4 + 2);                                  /* lv, hv */
      t += (m1 + 2) / 2 + 2 + (m1 * S1_LIGHT_MAX + 3) / 4 + 2 + (m1 * S1_LIGHT_MAX + 2) / 2 + 2 + m1 * S1_LIGHT_MAX * nn + 4;      /* lro, lrp, lre, Tc */
   }
   t += (q + 2) / 2 + 1 + (m1 + 2) / 2 + 1;
   t += 2 * ((long long) q * m1 + 2) + 2 * (((long long) q * m1 + 4) / 4) + 8;
   return t + 64;
}

#else       /* the device part: one kernel per size class (see the head of this file) */

namespace {

typedef double v4d __attribute__((ext_vector_type(4)));

template<int CTRL>
__device__ __forceinline__ double s1_dpp(double v)
{
   int lo = __double2loint(v), hi = __double2hiint(v);
   lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
   hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
   return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double s1_lane(double v, int l)
{
   const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
   return __hiloint2double(hi, lo);
}
/* e / n for 0 <= e < 2^20, 1 <= n <= 64 without the integer-division expansion: float reciprocal, one correction step */
/* a value every lane computed identically, moved to scalar registers (the kernel keeps dozens of such loop-carried values; as vector
 * registers they take two each for the whole solve, as scalars they are spilled 64 to a vector register) */
#ifdef S1_DEBUG
__device__ unsigned int s1_dbg[4];        /* [0]: s1_uni values that differed between the lanes of a wavefront, [1]: solves run */
#endif
__device__ __forceinline__ double s1_uni(double v)
{
#ifdef S1_NO_UNI
   return v;
#else
   int lo = __double2loint(v), hi = __double2hiint(v);
   lo = __builtin_amdgcn_readfirstlane(lo);
   hi = __builtin_amdgcn_readfirstlane(hi);
#ifdef S1_DEBUG
   if ( (lo != __double2loint(v) || hi != __double2hiint(v)) )
      atomicAdd(&s1_dbg[0], 1u);
#endif
   return __hiloint2double(hi, lo);
#endif
}

__device__ __forceinline__ int s1_div(int e, int n)
{
   int r = (int) ((float) e * __builtin_amdgcn_rcpf((float) n));
   const int c = e - r * n;
   if ( c >= n ) ++r;
   else if ( c < 0 ) --r;
   return r;
}
/* sum within the rows of 16 lanes (every lane of a row gets its row's sum) */
__device__ __forceinline__ double s1_sum16(double v)
{
   v += s1_dpp<0xB1>(v);              /* quad_perm [1, 0, 3, 2] */
   v += s1_dpp<0x4E>(v);              /* quad_perm [2, 3, 0, 1] */
   v += s1_dpp<0x141>(v);             /* row_half_mirror */
   v += s1_dpp<0x140>(v);             /* row_mirror */
   return v;
}
__device__ __forceinline__ double s1_wsum(double v)
{
   v = s1_sum16(v);
   return ((s1_lane(v, 0) + s1_lane(v, 16)) + s1_lane(v, 32)) + s1_lane(v, 48);
}
__device__ __forceinline__ double s1_wmax(double v)
{
   v = fmax(v, s1_dpp<0xB1>(v));
   v = fmax(v, s1_dpp<0x4E>(v));
   v = fmax(v, s1_dpp<0x141>(v));
   v = fmax(v, s1_dpp<0x140>(v));
   return fmax(fmax(s1_lane(v, 0), s1_lane(v, 16)), fmax(s1_lane(v, 32), s1_lane(v, 48)));
}
__device__ __forceinline__ double s1_wmin(double v)
{
   v = fmin(v, s1_dpp<0xB1>(v));
   v = fmin(v, s1_dpp<0x4E>(v));
   v = fmin(v, s1_dpp<0x141>(v));
   v = fmin(v, s1_dpp<0x140>(v));
   return fmin(fmin(s1_lane(v, 0), s1_lane(v, 16)), fmin(s1_lane(v, 32), s1_lane(v, 48)));
}

struct S1Blk
{
   int n, p, np, G;
   int oX, oZi, oLx, oLz, odX, odZ, oT1, oT2, oEig;
   double *Z, *Rd, *E, *B, *XR;                                  /* cold matrices, pitch p (flat: LDS or workspace); XR = X Rd of the iteration */
   int* voff; unsigned* vpq; double* vval;                        /* by variable: ALL entries (both triangles) in row-major order, vpq = row << 16 | col */
   int* poff; unsigned short* pvar; double* pval;                 /* by position r * n + c (r >= c): the variables that touch it */
   unsigned short* lv; unsigned short* hv; int nl, nh;            /* variables with few ("light") and many nonzeros in this block */
   /* rows of the light matrices: variable lv[a] has the row slots lro[a] .. lro[a + 1]; slot s is row lrp[s] of its matrix, whose
    * entries are the lre[s] & 63 entries from lre[s] >> 6 on of the variable-major list; Tc[s * n + c] = (A_j Zinv)[row][c] */
   int* lro; unsigned short* lrp; int* lre; double* Tc; int nrs;
};

struct S1Sh
{
   S1Lay lay;
   S1Blk blk[S1_MAXB];
   int* roff; unsigned short* rcol; double* rval;                 /* LP rows */
   int* coff; unsigned short* crow; double* cval;                 /* LP columns */
   double red[S1_NW][S1_NRED];
   double sc[SC_COUNT];
   int wtot[S1_NW + 1];
   int fl[40];
   long long t_last;
   double prof[24];
};
static_assert(sizeof(S1Sh) <= S1_STATIC_LDS, "static LDS of the one-launch solve");

/* ---- one wavefront: Cholesky of the lower triangle held in LDS (pitch p odd), lane = row, left-looking.  psd: the semidefinite
 * pivot rule of oracle/ipm_ref.chol_psd / csrc/chol.hip (diag0 = original diagonal).  Returns 0 or 1 + index of the first
 * non-positive pivot (definite mode). */
__device__ __forceinline__ int s1_chol(double* L, int n, int p, int lane, bool psd, const double* diag0, int rule)
{
   const double regtol = 1e-13;
   for (int k = 0; k < n; ++k)
   {
      double v = 0.0;
      if ( lane >= k && lane < n )
      {
         const double* rl = L + lane * p;
         const double* rk = L + k * p;
         /* (eight entries per round: sixteen LDS reads in flight - a read at a time costs its latency, about 100 cycles, per entry) */
         double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
         int j = 0;
         for (; j + 7 < k; j += 8)
         {
            const double a0 = rl[j], a1 = rl[j + 1], a2 = rl[j + 2], a3 = rl[j + 3], a4 = rl[j + 4], a5 = rl[j + 5], a6 = rl[j + 6], a7 = rl[j + 7];
            const double b0 = rk[j], b1 = rk[j + 1], b2 = rk[j + 2], b3 = rk[j + 3], b4 = rk[j + 4], b5 = rk[j + 5], b6 = rk[j + 6], b7 = rk[j + 7];
            s0 = fma(a0, b0, s0); s1 = fma(a1, b1, s1); s2 = fma(a2, b2, s2); s3 = fma(a3, b3, s3);
            s0 = fma(a4, b4, s0); s1 = fma(a5, b5, s1); s2 = fma(a6, b6, s2); s3 = fma(a7, b7, s3);
         }
         if ( j + 3 < k )
         {
            const double a0 = rl[j], a1 = rl[j + 1], a2 = rl[j + 2], a3 = rl[j + 3];
            const double b0 = rk[j], b1 = rk[j + 1], b2 = rk[j + 2], b3 = rk[j + 3];
            s0 = fma(a0, b0, s0); s1 = fma(a1, b1, s1); s2 = fma(a2, b2, s2); s3 = fma(a3, b3, s3);
            j += 4;
         }
         {
            const double a0 = (j < k) ? rl[j] : 0.0, a1 = (j + 1 < k) ? rl[j + 1] : 0.0, a2 = (j + 2 < k) ? rl[j + 2] : 0.0;
            const double b0 = (j < k) ? rk[j] : 0.0, b1 = (j + 1 < k) ? rk[j + 1] : 0.0, b2 = (j + 2 < k) ? rk[j + 2] : 0.0;
            s0 = fma(a0, b0, s0); s1 = fma(a1, b1, s1); s2 = fma(a2, b2, s2);
         }
         v = rl[k] - ((s0 + s1) + (s2 + s3));
      }
      double d = s1_lane(v, k);
      bool zero = false;
      if ( psd )
      {
         const double mkk = diag0[k];
         if ( !(d > regtol * mkk) || !(d > 1e-300) )
         {
            zero = (rule == 1) || (rule == 2 && !(d > 0.0)) || (rule == 3 && !(d > 1.78e-15 * (double) (k + 1) * mkk));
            d = (mkk > 1e-280) ? regtol * mkk : 1.0;
         }
      }
      else if ( !(d > 0.0) )
         return k + 1;
      const double sd = sqrt(d);
      const double rs = 1.0 / sd;
      if ( lane == k )
         L[k * p + k] = sd;
      else if ( lane > k && lane < n )
         L[lane * p + k] = zero ? 0.0 : v * rs;
      S1_WSYNC();
   }
   return 0;
}

/* reciprocal square root and reciprocal to full precision: v_rsq_f64 / v_rcp_f64 and two Newton steps (a few instructions instead
 * of the division and square-root expansions - the single-wavefront recurrences are bound by their instruction count) */
__device__ __forceinline__ double s1_rsqrt(double x)
{
   double r = __builtin_amdgcn_rsq(x);
   double e = fma(-x * r, r, 1.0);
   r = fma(0.5 * r, e, r);
   e = fma(-x * r, r, 1.0);
   r = fma(0.5 * r, e, r);
   return r;
}
__device__ __forceinline__ double s1_rcp(double t)
{
   double r = __builtin_amdgcn_rcp(t);
   r = fma(fma(-t, r, 1.0), r, r);
   r = fma(fma(-t, r, 1.0), r, r);
   return r;
}

/* square root of a non-negative number through the reciprocal square root (the library square root and the FP64 division expand to
 * 25-35 instructions each; the scalar bookkeeping of an iteration - every wavefront does it for itself - had twenty of them) */
__device__ __forceinline__ double s1_sqrt(double x)
{
   return (x > 0.0) ? x * s1_rsqrt(x) : 0.0;
}

/* ---- one wavefront: Cholesky in panels of eight columns.  The matrix (lower triangle, LDS, pitch p) is factored in place; before a
 * panel is touched the finished columns are applied to it on the matrix cores (16 x 16 x 4 tiles: a few dozen instructions
 * where the dot products entry by entry take thousands), then the panel lives in eight registers per lane (lane = row) and the
 * pivots run as a register recurrence - the entries of the pivot row come by v_readlane, no LDS round trip inside a panel.
 * psd: semidefinite pivot rule of oracle/ipm_ref.chol_psd (dg0 = this lane's original diagonal entry).  keepdiag = false: the
 * stored factor has a ZERO diagonal and zero upper triangle (what the substitutions below want), the diagonal entry of row
 * `lane` is returned in mydiag.  Returns 0 or 1 + index of the first non-positive pivot (definite mode).  hook(j) is called before
 * panel j = 1, 2, .. is touched (default: nothing). */
struct S1NoHook { __device__ __forceinline__ void operator()(int) const {} };
template<class HOOK = S1NoHook>
__device__ __forceinline__ int s1_cholp(double* A, int n, int p, int lane, bool psd, double dg0, int rule, bool keepdiag, double& mydiag, int& nforced,
   const double* zp, HOOK hook = HOOK())
{
   const double regtol = 1e-13;
   const int lr = lane & 15, kq = lane >> 4;
   /* semidefinite rule: one comparison per pivot in the usual case (the threshold of this lane's row, never below 1e-300) */
   const double thr = fmax(regtol * dg0, 1e-300);
   mydiag = 1.0;
   nforced = 0;
   for (int k0 = 0; k0 < n; k0 += 8)
   {
      /* (the caller's hook between two panels: the factorization of M meets the other wavefronts at workgroup barriers there) */
      if ( k0 > 0 )
         hook(k0 >> 3);
      if ( k0 > 0 )
      {
         /* panel -= (finished columns) (their rows k0 .. k0 + 7)^T, 16-row tiles on the matrix cores.  All loads unconditional
          * with clamped indices (rows past n and the unused half of the B operand only reach entries that are not written
          * back), the accumulators start from the panel itself and take the NEGATIVE products, so that the result is stored
          * without a read behind the matrix instructions. */
         const int nm1 = n - 1;
         for (int T = k0 >> 4; 16 * T < n; ++T)
         {
            const int ar = min(16 * T + lr, nm1), br = min(k0 + lr, nm1);
            const double* pa = A + ar * p + kq;
            const double* pb = A + br * p + kq;
            const int cc = min(k0 + lr, nm1);
            v4d acc;
#pragma unroll
            for (int r = 0; r < 4; ++r)
               acc[r] = A[min(16 * T + kq + 4 * r, nm1) * p + cc];
            for (int kk = 0; kk < k0; kk += 8)
            {
               const double a0 = -pa[kk], b0 = pb[kk];
               const double a1 = -pa[kk + 4], b1 = pb[kk + 4];
               acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc, 0, 0, 0);
               acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
            {
               const int row = 16 * T + kq + 4 * r;
               if ( lr < 8 && row < n && row >= k0 + lr && k0 + lr < n )
                  A[row * p + k0 + lr] = acc[r];
            }
         }
         S1_WSYNC();
      }
      double a[8];
#if S1_MBIG
      /* (the instance for m > 64 keeps the loads under their masks: with unconditional loads in THIS function its -DS1_DEBUG
       * build walked other iterates than its release build on blocks of 11-14 rows - DESIGN 7.6) */
#pragma unroll
      for (int u = 0; u < 8; ++u)
         a[u] = (lane >= k0 + u && lane < n && k0 + u < n) ? A[lane * p + k0 + u] : 0.0;
#else
      /* unconditional loads, the ADDRESS selected between the entry and zp - an LDS word that holds 0.0 and is never written: a
       * load under a mask becomes a branch of its own with a full LDS wait, eight in a row here (see s1_cholp2_cols) */
#pragma unroll
      for (int u = 0; u < 8; ++u)
         a[u] = *((lane >= k0 + u && lane < n && k0 + u < n) ? (const double*) (A + lane * p + k0 + u) : zp);
#endif
#pragma unroll
      for (int u = 0; u < 8; ++u)
      {
         const int k = k0 + u;
         if ( k < n )
         {
            double d = s1_lane(a[u], k);
            bool zero = false;
            if ( psd )
            {
               if ( !(d > s1_lane(thr, k)) )
               {
                  const double mkk = s1_lane(dg0, k);
                  zero = (rule == 1) || (rule == 2 && !(d > 0.0)) || (rule == 3 && !(d > 1.78e-15 * (double) (k + 1) * mkk));
                  d = (mkk > 1e-280) ? regtol * mkk : 1.0;
                  nforced += zero ? 65536 : 1;
               }
            }
            else if ( !(d > 0.0) )
               return k + 1;
            const double rs = s1_rsqrt(d);
            const double sd = d * rs;                        /* (d may be the replacement of a forced pivot) */
            if ( lane == k )
               mydiag = sd;
            const double lu = (lane > k && !zero) ? a[u] * rs : 0.0;
            a[u] = (lane == k && keepdiag) ? sd : lu;
            /* the later columns of the panel take this column's term at once (independent multiply-adds; taken column by column
             * when its pivot comes they were a chain of up to seven dependent ones in front of every pivot) */
#pragma unroll
            for (int v = u + 1; v < 8; ++v)
               a[v] = fma(-lu, s1_lane(lu, (k0 + v) & 63), a[v]);
         }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
         if ( lane < n && k0 + u < n )
            A[lane * p + k0 + u] = a[u];
      S1_WSYNC();
   }
   return 0;
}

/* ---- one wavefront: x = (L L^T)^-1 r for one or two right-hand sides by forward and backward substitution, lane = row.  L as
 * s1_cholp(keepdiag = false) leaves it: strictly lower, ZERO diagonal, upper triangle and padding columns (pitch p >= 8 ceil(m / 8)
 * + 1, so no step needs a mask or a bound), dinv = 1 / (diagonal entry of this lane's row).  In: x0, x1 = this lane's entries of
 * the right-hand sides; out: of the solutions.  The entries of eight steps are loaded - unconditionally, one block ahead - before
 * the recurrence needs them; a step is a multiply, a v_readlane pair and a multiply-add, straight-line code.  [A first form with
 * guarded loads and a run-time "two right-hand sides" flag spent three quarters of its instructions on branches and exec masks:
 * 220 cycles per step.] */
template<bool TWO>
__device__ __forceinline__ void s1_llt_solve(const double* L, int m, int p, int lane, double dinv, double& x0, double& x1)
{
   const int rl = (lane < m) ? lane : 0;
   const bool live = lane < m;
   double a0 = live ? x0 : 0.0, a1 = (live && TWO) ? x1 : 0.0;
   const int nb = (m + 7) >> 3;
   {
      const double* row = L + rl * p;
      double cn[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
         cn[u] = row[u];
      for (int b = 0; b < nb; ++b)
      {
         const int k0 = 8 * b;
         double c[8];
#pragma unroll
         for (int u = 0; u < 8; ++u)
            c[u] = live ? cn[u] : 0.0;
         if ( b + 1 < nb )
         {
#pragma unroll
            for (int u = 0; u < 8; ++u)
               cn[u] = row[k0 + 8 + u];
         }
#pragma unroll
         for (int u = 0; u < 8; ++u)
         {
            const double t0 = a0 * dinv, t1 = a1 * dinv;
            const double y0 = s1_lane(t0, k0 + u);
            const double y1 = TWO ? s1_lane(t1, k0 + u) : 0.0;
            a0 = fma(-c[u], y0, a0);
            if ( TWO )
               a1 = fma(-c[u], y1, a1);
         }
      }
   }
   a0 *= dinv; a1 *= dinv;
   {
      /* rows k0 .. k0 + 7 of the factor, this lane's column; rows past m - 1 are clamped to row 0, whose entries are all zero */
      const double* col = L + rl;
      double cn[8];
      {
         const int k0 = 8 * (nb - 1);
#pragma unroll
         for (int u = 0; u < 8; ++u)
            cn[u] = col[((k0 + u < m) ? k0 + u : 0) * p];
      }
      for (int b = nb - 1; b >= 0; --b)
      {
         const int k0 = 8 * b;
         double c[8];
#pragma unroll
         for (int u = 0; u < 8; ++u)
            c[u] = live ? cn[u] : 0.0;
         if ( b > 0 )
         {
#pragma unroll
            for (int u = 0; u < 8; ++u)
               cn[u] = col[(k0 - 8 + u) * p];
         }
#pragma unroll
         for (int u = 7; u >= 0; --u)
         {
            const double t0 = a0 * dinv, t1 = a1 * dinv;
            const double y0 = s1_lane(t0, (k0 + u) & 63);
            const double y1 = TWO ? s1_lane(t1, (k0 + u) & 63) : 0.0;
            a0 = fma(-c[u], y0, a0);
            if ( TWO )
               a1 = fma(-c[u], y1, a1);
         }
      }
   }
   x0 = a0 * dinv;
   x1 = a1 * dinv;
}

/* ==== 64 < m <= 128: the factor of M and the substitutions with TWO rows per lane (lane and lane + 64) ====
 * Functions of their own (the common case m <= 64 keeps its registers); results go through LDS vectors.  The matrix is the PACKED
 * lower triangle of the extended Schur matrix Mx (m + 1 rows, row i at i (i + 1) / 2): entry (i, j), j <= i, of M itself is
 * Mx[S1_PKROW(i) + j] - row i + 1, column j + 1 of the extended matrix.  Nothing above the diagonal exists: loads that would reach
 * there are clamped to an entry that does and masked (or their results discarded), stores never go there.  The code of a panel
 * and of a block of substitution steps exists twice, for columns below and from 64 on (HI): which of a lane's two rows holds the
 * pivot is then known at compile time - with run-time selects the panel was 250 branches, 10 000 cycles. */
#define S1_PKROW(i) ((((i) + 1) * ((i) + 2) >> 1) + 1)
typedef __attribute__((address_space(3))) double s1_ldsd;

/* the eight columns k0 .. k0 + 7 of the panel as a register recurrence (one wavefront); a: rows lane, b: rows lane + 64.  HI: the
 * panel's columns are 64 and up - rows below 64 have no entries there */
template<bool HI>
__device__ __forceinline__ void s1_cholp2_cols(s1_ldsd* A, int n, int k0, int lane, bool has1, int r0, int r1, double thr, double dgx,
   int rule, double& diag, int& nforced)
{
   const double regtol = 1e-13;
   const int rowb = has1 ? lane + 64 : lane;           /* (the row r1 points to) */
   const int zi = ((n + 1) * (n + 2)) >> 1;            /* first of the sixteen zeros behind the packed triangle */
   double a[8], b[8];
#pragma unroll
   for (int u = 0; u < 8; ++u)
   {
      const int col = k0 + u;
      /* Every load unconditional, its ADDRESS selected between the entry and a zero behind the triangle: nothing to mask, the
       * same values exactly, and nothing for the compiler to sink - a load under a mask (`cond ? A[i] : 0.0`) became a branch of
       * its own with a full LDS wait, sixteen in a row per panel.  [Two other forms on the way: loads from clamped columns with a
       * select behind them gave wrong factors; the same with the values pinned by an empty asm was right on every test and as
       * fast as this one, but - inlined into the kernel through s1_cholp as well - its -DS1_DEBUG build walked other iterates
       * than the release build, whatever the poison value: not shipped.  DESIGN 7.6] */
      a[u] = HI ? 0.0 : A[(lane >= col && col < n) ? r0 + col : zi];
      b[u] = A[(has1 && rowb >= col && col < n) ? r1 + col : zi];
   }
#pragma unroll
   for (int u = 0; u < 8; ++u)
   {
      const int k = k0 + u;
      if ( k < n )
      {
         const int kl = HI ? k - 64 : k;
         double d = s1_lane(HI ? b[u] : a[u], kl);
         bool zero = false;
         if ( !(d > s1_lane(thr, kl)) )
         {
            const double mkk = s1_lane(dgx, kl);
            zero = (rule == 1) || (rule == 2 && !(d > 0.0)) || (rule == 3 && !(d > 1.78e-15 * (double) (k + 1) * mkk));
            d = (mkk > 1e-280) ? regtol * mkk : 1.0;
            nforced += zero ? 65536 : 1;
         }
         const double rs = s1_rsqrt(d);
         const double sd = d * rs;
         diag = (lane == kl) ? sd : diag;
         const double lua = (!HI && lane > k && !zero) ? a[u] * rs : 0.0;
         const double lub = ((!HI || lane + 64 > k) && !zero) ? b[u] * rs : 0.0;
         a[u] = lua; b[u] = lub;
#pragma unroll
         for (int v = u + 1; v < 8; ++v)
         {
            const double piv = s1_lane(HI ? lub : lua, HI ? k0 + v - 64 : k0 + v);
            if ( !HI )
               a[v] = fma(-lua, piv, a[v]);
            b[v] = fma(-lub, piv, b[v]);
         }
      }
   }
   /* (the diagonal entry is stored as ZERO - what the substitutions want -, nothing is stored past it) */
#pragma unroll
   for (int u = 0; u < 8; ++u)
   {
      const int col = k0 + u;
      if ( col < n )
      {
         if ( !HI && lane >= col )
            A[r0 + col] = a[u];
         if ( has1 && rowb >= col )
            A[r1 + col] = b[u];
      }
   }
}

/* s1_cholp(psd = true, keepdiag = false) for 64 < n <= 128; dg: the original diagonal (LDS, n entries), dinv_out: 1 / (diagonal
 * entries of the factor) (LDS, n entries); returns the forced-pivot counter */
/* ALL wavefronts of the workgroup call it: the panel update in front of every panel (16-row tiles on the matrix cores, independent
 * of each other) is dealt out tile by tile, the eight columns of the panel are a register recurrence of wavefront 0; two workgroup
 * barriers per panel.  [One wavefront doing both: 227 000 cycles at m = 105, of which the tile updates were 60 000.]  The return
 * value and dinv_out are wavefront 0's. */
__device__ __attribute__((noinline)) int s1_cholp2(double* Ag, int n_, int wave_, int lane, const double* dg, int rule_, double* dinv_out)
{
   /* (the arguments of a function that is not inlined arrive in vector registers: said to be uniform, the loops and the tests
    * on them are scalar branches instead of execution masks) */
   const int n = __builtin_amdgcn_readfirstlane(n_), wave = __builtin_amdgcn_readfirstlane(wave_), rule = __builtin_amdgcn_readfirstlane(rule_);
   const double regtol = 1e-13;
   s1_ldsd* A = (s1_ldsd*) Ag;
   const int lr = lane & 15, kq = lane >> 4;
   const bool has1 = lane + 64 < n;
   const double dg0 = dg[lane], dg1 = has1 ? dg[lane + 64] : 1.0;
   const double thr0 = fmax(regtol * dg0, 1e-300), thr1 = fmax(regtol * dg1, 1e-300);
   double diag0 = 1.0, diag1 = 1.0;
   int nforced = 0;
   const int r0 = S1_PKROW(lane), r1 = S1_PKROW(has1 ? lane + 64 : lane);
   for (int k0 = 0; k0 < n; k0 += 8)
   {
      if ( k0 > 0 )
      {
         /* (rows above the panel and the columns of a row past its diagonal read whatever follows in the packed array: such
          * entries only reach accumulator rows / columns that are not stored) */
         const int nm1 = n - 1;
         for (int T = (k0 >> 4) + wave; 16 * T < n; T += S1_NW)
         {
            const int ar = min(16 * T + lr, nm1), br = min(k0 + lr, nm1);
            const s1_ldsd* pa = A + S1_PKROW(ar) + kq;
            const s1_ldsd* pb = A + S1_PKROW(br) + kq;
            const int cc = min(k0 + lr, nm1);
            v4d acc;
#pragma unroll
            for (int r = 0; r < 4; ++r)
               acc[r] = A[S1_PKROW(min(16 * T + kq + 4 * r, nm1)) + cc];
            for (int kk = 0; kk < k0; kk += 8)
            {
               const double a0 = -pa[kk], b0 = pb[kk];
               const double a1 = -pa[kk + 4], b1 = pb[kk + 4];
               acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc, 0, 0, 0);
               acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
            {
               const int row = 16 * T + kq + 4 * r;
               if ( lr < 8 && row < n && row >= k0 + lr && k0 + lr < n )
                  A[S1_PKROW(row) + k0 + lr] = acc[r];
            }
         }
         __syncthreads();
      }
      if ( wave == 0 )
      {
         /* (a panel lies in one half: 64 is a multiple of 8) */
         if ( k0 >= 64 )
            s1_cholp2_cols<true>(A, n, k0, lane, has1, r0, r1, thr1, dg1, rule, diag1, nforced);
         else
            s1_cholp2_cols<false>(A, n, k0, lane, has1, r0, r1, thr0, dg0, rule, diag0, nforced);
      }
      __syncthreads();
   }
   if ( wave == 0 )
   {
      dinv_out[lane] = s1_rcp(diag0);
      if ( has1 )
         dinv_out[lane + 64] = s1_rcp(diag1);
   }
   __syncthreads();
   return nforced;
}

/* eight steps of a substitution with two rows per lane: c / ch = this lane's entries of the factor for the steps k0 .. k0 + 7 (rows
 * lane / lane + 64; forward: of its rows, backward: of its columns), a* = the right-hand sides being reduced.  HI: the steps are
 * 64 and up - the values come from the upper rows, and (forward) the lower rows are done, (backward) both halves take part */
template<bool TWO, bool HI, bool DOWN>
__device__ __forceinline__ void s1_llt_steps2(const double* c, const double* ch, int k0, double di0, double di1, double& a0, double& a0h,
   double& a1, double& a1h)
{
#pragma unroll
   for (int uu = 0; uu < 8; ++uu)
   {
      const int u = DOWN ? uu : 7 - uu;
      const int kl = HI ? k0 + u - 64 : k0 + u;
      const double t0 = HI ? a0h * di1 : a0 * di0, t1 = HI ? a1h * di1 : a1 * di0;
      const double y0 = s1_lane(t0, kl);
      const double y1 = TWO ? s1_lane(t1, kl) : 0.0;
      a0 = fma(-c[u], y0, a0); a0h = fma(-ch[u], y0, a0h);
      if ( TWO )
      {
         a1 = fma(-c[u], y1, a1); a1h = fma(-ch[u], y1, a1h);
      }
   }
}

/* s1_llt_solve for 64 < m <= 128: right-hand sides r0 (and r1) and results o0 (o1) are LDS vectors, dinv the vector s1_cholp2 left */
template<bool TWO>
__device__ __attribute__((noinline)) void s1_llt_solve2(const double* Lg, int m_, int lane, const double* dinv, const double* r0,
   const double* r1, double* o0, double* o1)
{
   const int m = __builtin_amdgcn_readfirstlane(m_);
   const s1_ldsd* L = (const s1_ldsd*) Lg;
   const bool has1 = lane + 64 < m;
   const int rh = has1 ? lane + 64 : lane;
   const double di0 = dinv[lane], di1 = has1 ? dinv[lane + 64] : 1.0;
   double a0 = r0[lane], a0h = has1 ? r0[lane + 64] : 0.0;
   double a1 = TWO ? r1[lane] : 0.0, a1h = (TWO && has1) ? r1[lane + 64] : 0.0;
   const int nb = (m + 7) >> 3;
   const int zi = ((m + 1) * (m + 2)) >> 1;            /* first of the sixteen zeros behind the packed triangle */
   {
      /* this lane's rows, columns k0 .. k0 + 7: only the strictly lower part exists */
      for (int b = 0; b < nb; ++b)
      {
         const int k0 = 8 * b;
         double c[8], ch[8];
#pragma unroll
         for (int u = 0; u < 8; ++u)
         {
            /* (unconditional loads, the address selected between the entry and a zero behind the triangle: see s1_cholp2_cols) */
            c[u] = L[(k0 + u < lane) ? S1_PKROW(lane) + k0 + u : zi];
            ch[u] = L[(has1 && k0 + u < rh) ? S1_PKROW(rh) + k0 + u : zi];
         }
         if ( k0 >= 64 )
            s1_llt_steps2<TWO, true, true>(c, ch, k0, di0, di1, a0, a0h, a1, a1h);
         else
            s1_llt_steps2<TWO, false, true>(c, ch, k0, di0, di1, a0, a0h, a1, a1h);
      }
   }
   a0 *= di0; a0h *= di1; a1 *= di0; a1h *= di1;
   {
      /* rows k0 .. k0 + 7 of the factor, columns lane and lane + 64: entry (row, column) exists for row > column */
      for (int b = nb - 1; b >= 0; --b)
      {
         const int k0 = 8 * b;
         double c[8], ch[8];
#pragma unroll
         for (int u = 0; u < 8; ++u)
         {
            const int rr = (k0 + u < m) ? k0 + u : 0;
            c[u] = L[rr > lane ? S1_PKROW(rr) + lane : zi];
            ch[u] = L[(has1 && rr > rh) ? S1_PKROW(rr) + rh : zi];
         }
         if ( k0 >= 64 )
            s1_llt_steps2<TWO, true, false>(c, ch, k0, di0, di1, a0, a0h, a1, a1h);
         else
            s1_llt_steps2<TWO, false, false>(c, ch, k0, di0, di1, a0, a0h, a1, a1h);
      }
   }
   o0[lane] = a0 * di0;
   if ( has1 ) o0[lane + 64] = a0h * di1;
   if ( TWO )
   {
      o1[lane] = a1 * di0;
      if ( has1 ) o1[lane + 64] = a1h * di1;
   }
   S1_WSYNC();
}

/* ---- one wavefront: Li = L^-1 (lower), lane = column; in place when Li == L */
__device__ __forceinline__ void s1_trinv(const double* L, double* Li, int n, int p, int lane)
{
   for (int i = 0; i < n; ++i)
   {
      const double* ri = L + i * p;
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
      {
         /* uniform trip count (entries above the lane's column are masked): eight entries per round, the reads in flight together */
         const double* cl = Li + lane;
         int k = 0;
         for (; k + 7 < i; k += 8)
         {
            double a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
            {
               a[u] = ri[k + u];
               b[u] = (k + u >= lane) ? cl[(k + u) * p] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; u += 4)
            {
               s0 = fma(a[u], b[u], s0); s1 = fma(a[u + 1], b[u + 1], s1); s2 = fma(a[u + 2], b[u + 2], s2); s3 = fma(a[u + 3], b[u + 3], s3);
            }
         }
         {
            double a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
            {
               const bool in = (k + u < i);
               a[u] = in ? ri[k + u] : 0.0;
               b[u] = (in && k + u >= lane) ? cl[(k + u) * p] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; u += 4)
            {
               s0 = fma(a[u], b[u], s0); s1 = fma(a[u + 1], b[u + 1], s1); s2 = fma(a[u + 2], b[u + 2], s2); s3 = fma(a[u + 3], b[u + 3], s3);
            }
         }
      }
      const double rd = s1_rcp(ri[i]);
      const double val = (lane == i) ? rd : -((s0 + s1) + (s2 + s3)) * rd;
      __builtin_amdgcn_wave_barrier();
      if ( lane <= i )
         Li[i * p + lane] = val;
      S1_WSYNC();
   }
}

/* ==== blocks of at most S1U_MAXN rows: the whole matrix in the registers of EVERY lane ====
 * The recurrences of a small block (Cholesky factor, its inverse, the Householder reduction to tridiagonal form) are chains of
 * dependent steps; spread over the lanes of a wavefront each step pays a cross-lane reduction, a broadcast through LDS or
 * v_readlane and the waits between them - 1500 to 2200 cycles per column at n = 10, whatever the arithmetic.  Here every lane
 * holds the lower triangle (at most 55 doubles, indices fixed at compile time) and all lanes do the same arithmetic: no
 * communication at all, the instruction count is the flop count (n^3 / 3 for factor + inverse, 2 n^3 / 3 for the reduction) and
 * independent multiply-adds issue back to back.  Instantiated for NP = 4, 6, 8, 10 rows; a smaller block is padded. */
#define S1U_IX(i, j) ((i) * ((i) + 1) / 2 + (j))
#ifndef S1U_MAXN
#define S1U_MAXN 10                                           /* (12 fits the registers only with 500 spills) */
#endif

/* W (LDS, full symmetric n x n, pitch p) -> inverse of its Cholesky factor, lower triangle, in place.  Returns 0, or 1 + the index
 * of the first pivot that is not positive (nothing stored then). */
template<int NP>
__device__ __forceinline__ int s1u_chol_inv(double* W, int n, int p, int lane)
{
   double a[NP * (NP + 1) / 2];
#pragma unroll
   for (int i = 0; i < NP; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j)
         a[S1U_IX(i, j)] = (i < n) ? W[i * p + j] : ((i == j) ? 1.0 : 0.0);
#pragma unroll
   for (int k = 0; k < NP; ++k)
   {
      double d = a[S1U_IX(k, k)];
#pragma unroll
      for (int v = 0; v < k; ++v)
         d = fma(-a[S1U_IX(k, v)], a[S1U_IX(k, v)], d);
      if ( !(d > 0.0) )
         return k + 1;
      const double rs = s1_rsqrt(d);
      a[S1U_IX(k, k)] = rs;                               /* 1 / (diagonal entry of the factor) */
#pragma unroll
      for (int i = k + 1; i < NP; ++i)
      {
         double sv = a[S1U_IX(i, k)];
#pragma unroll
         for (int v = 0; v < k; ++v)
            sv = fma(-a[S1U_IX(i, v)], a[S1U_IX(k, v)], sv);
         a[S1U_IX(i, k)] = sv * rs;
      }
   }
   /* inverse in place, row by row: Li[i][j] = -(sum_{k = j}^{i - 1} L[i][k] Li[k][j]) / L[i][i] */
#pragma unroll
   for (int i = 1; i < NP; ++i)
   {
      const double rd = a[S1U_IX(i, i)];
#pragma unroll
      for (int j = 0; j < i; ++j)
      {
         double sv = a[S1U_IX(i, j)] * a[S1U_IX(j, j)];
#pragma unroll
         for (int k = j + 1; k < i; ++k)
            sv = fma(a[S1U_IX(i, k)], a[S1U_IX(k, j)], sv);
         a[S1U_IX(i, j)] = -sv * rd;
      }
   }
   if ( lane == 0 )
   {
#pragma unroll
      for (int i = 0; i < NP; ++i)
#pragma unroll
         for (int j = 0; j <= i; ++j)
            if ( i < n )
               W[i * p + j] = a[S1U_IX(i, j)];
   }
   S1_WSYNC();
   return 0;
}

__device__ __forceinline__ int s1u_chol_inv_n(double* W, int n, int p, int lane)
{
   if ( n <= 4 ) return s1u_chol_inv<4>(W, n, p, lane);
   if ( n <= 6 ) return s1u_chol_inv<6>(W, n, p, lane);
#if S1U_MAXN > 8
   if ( n <= 8 ) return s1u_chol_inv<8>(W, n, p, lane);
#endif
#if S1U_MAXN > 10
   if ( n <= 10 ) return s1u_chol_inv<10>(W, n, p, lane);
   return s1u_chol_inv<12>(W, n, p, lane);
#elif S1U_MAXN > 8
   return s1u_chol_inv<10>(W, n, p, lane);
#else
   return s1u_chol_inv<8>(W, n, p, lane);
#endif
}

/* min(lambda_min, 0) of the symmetric matrix whose lower triangle is in W (LDS, pitch p; not changed), to a relative accuracy of
 * 1e-10 from below; NaN when an entry is not finite.  Householder reduction as in s1_lmin16 (same formulas), then Sturm
 * multisection in product form with lane = shift, 64 shifts per round. */
template<int NP>
__device__ __forceinline__ double s1u_lmin(const double* W, int n, int p, int lane, double* tprof)
{
   double a[NP * (NP + 1) / 2];
#pragma unroll
   for (int i = 0; i < NP; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j)
         a[S1U_IX(i, j)] = (i < n) ? W[i * p + j] : 0.0;
   double d[NP], e[NP];
#pragma unroll
   for (int k = 0; k + 2 < NP; ++k)
   {
      if ( k + 2 < n )
      {
         const double x0 = a[S1U_IX(k + 1, k)];
         double s2a = 0.0, s2b = 0.0;
#pragma unroll
         for (int i = k + 2; i < NP; i += 2)
         {
            s2a = fma(a[S1U_IX(i, k)], a[S1U_IX(i, k)], s2a);
            if ( i + 1 < NP )
               s2b = fma(a[S1U_IX(i + 1, k)], a[S1U_IX(i + 1, k)], s2b);
         }
         const double s2 = s2a + s2b;
         if ( s2 != s2 )
            return s2;
         if ( s2 > 1e-290 )
         {
            const double h2 = x0 * x0 + s2;
            const double rh = s1_rsqrt(h2);
            const double beta = -copysign(h2 * rh, x0);
            const double t = (x0 - beta) * copysign(rh, x0);
            const double scale = s1_rcp(x0 - beta);
            double v[NP], w[NP];
            v[k + 1] = 1.0;
#pragma unroll
            for (int i = k + 2; i < NP; ++i)
               v[i] = a[S1U_IX(i, k)] * scale;
            a[S1U_IX(k + 1, k)] = beta;
            /* p = t A v over the trailing block, pv = p^T v, w = p - (t pv / 2) v */
            double pva = 0.0, pvb = 0.0;
#pragma unroll
            for (int i = k + 1; i < NP; ++i)
            {
               double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
               for (int j = k + 1; j < NP; ++j)
               {
                  const double aij = (j <= i) ? a[S1U_IX(i, j)] : a[S1U_IX(j, i)];
                  if ( (j - k) & 1 )
                     acc0 = fma(aij, v[j], acc0);
                  else
                     acc1 = fma(aij, v[j], acc1);
               }
               w[i] = t * (acc0 + acc1);
               if ( (i - k) & 1 )
                  pva = fma(w[i], v[i], pva);
               else
                  pvb = fma(w[i], v[i], pvb);
            }
            const double hf = -0.5 * t * (pva + pvb);
#pragma unroll
            for (int i = k + 1; i < NP; ++i)
               w[i] = fma(hf, v[i], w[i]);
#pragma unroll
            for (int i = k + 1; i < NP; ++i)
#pragma unroll
               for (int j = k + 1; j <= i; ++j)
                  a[S1U_IX(i, j)] = fma(-w[i], v[j], fma(-v[i], w[j], a[S1U_IX(i, j)]));
         }
      }
   }
#pragma unroll
   for (int k = 0; k < NP; ++k)
   {
      d[k] = a[S1U_IX(k, k)];
      e[k] = (k + 1 < NP) ? a[S1U_IX(k + 1, k)] : 0.0;
   }
   if ( tprof != NULL && lane == 0 )
      tprof[0] -= (double) clock64();
   /* Gershgorin bounds, scaling to norm one, padding rows that cannot change a sign */
   double nrm = 0.0, glo = 1e300;
   bool bad = false;
#pragma unroll
   for (int i = 0; i < NP; ++i)
   {
      const double rad = ((i > 0) ? fabs(e[i - 1]) : 0.0) + fabs(e[i]);
      nrm = fmax(nrm, fabs(d[i]) + rad);
      glo = fmin(glo, d[i] - rad);
      bad = bad || !(fabs(d[i]) < 1e300) || !(fabs(e[i]) < 1e300);
   }
   double res;
   if ( bad )
      res = nan("");
   else if ( !(glo < 0.0) || !(nrm > 0.0) )
      res = 0.0;
   else
   {
      const double sinv = s1_rcp(nrm);
      double dsv[NP], e2v[NP];
#pragma unroll
      for (int i = 0; i < NP; ++i)
      {
         dsv[i] = (i < n) ? d[i] * sinv : 4.0;
         e2v[i] = (i + 1 < n) ? (e[i] * sinv) * (e[i] * sinv) : 0.0;
      }
      double lo = glo * sinv * (1.0 + 1e-12) - 1e-300, hi = 0.0;
      bool first = true;
      const double flane1 = (double) (lane + 1);
      res = 0.0;
      bool done = false;
      for (int round = 0; round < 14 && !done; ++round)
      {
         const double wdt = (hi - lo) * (1.0 / 65.0);
         double x = lo + wdt * flane1;
         if ( first )
            x = (lane == 63) ? 0.0 : lo + (hi - lo) * flane1 * (1.0 / 64.0);
         double pp = 1.0, pc = dsv[0] - x;
         bool posc = pc > 0.0;
         bool below = !posc;
#pragma unroll
         for (int i = 1; i < NP; ++i)
         {
            const double pn = fma(dsv[i] - x, pc, -e2v[i - 1] * pp);
            const bool posn = pn > 0.0;
            below = below || (posn != posc);
            pp = pc; pc = pn; posc = posn;
            if ( i == 8 && NP > 9 )
            {
               const int ex = -max(__builtin_amdgcn_frexp_exp(pc), __builtin_amdgcn_frexp_exp(pp));
               const bool okx = ex > -1000 && ex < 1000;
               pc = okx ? ldexp(pc, ex) : pc;
               pp = okx ? ldexp(pp, ex) : pp;
            }
         }
         const unsigned long long msk = __ballot(below);
         if ( first )
         {
            first = false;
            if ( !(msk >> 63) )
            {
               lo = 0.0;                                    /* nothing below zero */
               done = true;
            }
            else
            {
               const int f = __ffsll((long long) msk) - 1;
               const double w64 = (hi - lo) * (1.0 / 64.0);
               const double nlo = (f == 0) ? lo : lo + w64 * (double) f;
               const double nhi = (f == 63) ? 0.0 : lo + w64 * (double) (f + 1);
               lo = nlo; hi = nhi;
            }
         }
         else
         {
            const int f = msk ? __ffsll((long long) msk) - 1 : 64;
            const double nlo = lo + wdt * (double) f;
            const double nhi = (f < 64) ? lo + wdt * (double) (f + 1) : hi;
            lo = nlo; hi = nhi;
         }
         if ( tprof != NULL && lane == 0 )
            tprof[2] += 1.0;
         if ( hi - lo <= 1e-10 * fabs(lo) || fabs(lo) < 1e-15 )
            done = true;
      }
      res = lo * nrm;
   }
   if ( tprof != NULL && lane == 0 )
      tprof[0] += (double) clock64();
   return res;
}

__device__ __forceinline__ double s1u_lmin_n(const double* W, int n, int p, int lane, double* tprof)
{
   if ( n <= 4 ) return s1u_lmin<4>(W, n, p, lane, tprof);
   if ( n <= 6 ) return s1u_lmin<6>(W, n, p, lane, tprof);
#if S1U_MAXN > 8
   if ( n <= 8 ) return s1u_lmin<8>(W, n, p, lane, tprof);
#endif
#if S1U_MAXN > 10
   if ( n <= 10 ) return s1u_lmin<10>(W, n, p, lane, tprof);
   return s1u_lmin<12>(W, n, p, lane, tprof);
#elif S1U_MAXN > 8
   return s1u_lmin<10>(W, n, p, lane, tprof);
#else
   return s1u_lmin<8>(W, n, p, lane, tprof);
#endif
}

template<int CTRL>
__device__ __forceinline__ double s1_dppz(double v)
{
   int lo = __double2loint(v), hi = __double2hiint(v);
   lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
   hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
   return __hiloint2double(hi, lo);
}

/* ---- one wavefront: min(lambda_min, 0) of the symmetric tridiagonal matrix with diagonal d_i and off-diagonal e_i (coupling i and
 * i + 1) held by lane i, n <= 64: Sturm multisection, 64 shifts per round, to a relative accuracy of 1e-10 (from below); NaN when
 * an entry is not finite.  The sequence runs in product form out of REGISTERS: d_i and e_i^2 reach all lanes by v_readlane (an LDS
 * read per step cost its latency, 100 cycles, per step: 2700 cycles per round at n = 10, now about 500). */
template<bool SMALL>
__device__ __forceinline__ double s1_sturm_min(double dl, double el, int n, int lane, double* bc, double* tprof)
{
   if ( lane >= n ) { dl = 0.0; el = 0.0; }
   if ( lane == n - 1 ) el = 0.0;
   double eprev = __shfl_up(el, 1, 64);
   if ( lane == 0 ) eprev = 0.0;
   const double rad = fabs(eprev) + fabs(el);
   const double bad = (!(fabs(dl) < 1e300) || !(fabs(el) < 1e300)) ? 1.0 : 0.0;
   if ( s1_wmax(bad) > 0.0 )
      return nan("");
   const double nrm = s1_wmax((lane < n) ? fabs(dl) + rad : 0.0);
   const double glo = s1_wmin((lane < n) ? dl - rad : 1e300);
   if ( !(glo < 0.0) || !(nrm > 0.0) )
      return 0.0;
   const double sinv = s1_rcp(nrm);
   const double ds = dl * sinv;
   const double e2 = (el * sinv) * (el * sinv);
   if ( tprof != NULL && lane == 0 )
      tprof[1] -= (double) clock64();
   double lo = glo * sinv * (1.0 + 1e-12) - 1e-300, hi = 0.0;
   const double d0 = s1_lane(ds, 0);
   /* SMALL (n <= 16): every lane holds the whole matrix in VECTOR registers (written to LDS once, read back by all lanes), padded to
    * 16 rows with rows that cannot change a sign (diagonal 4 > |x| + 1, no coupling): the 15 steps are straight-line code, three
    * arithmetic instructions and a compare each.  [Broadcast by v_readlane made the entries wavefront-uniform SCALAR values: the
    * kernel has none to spare, every step then re-read its two entries from spill lanes - 26 instructions and a branch per step.] */
   double dsv[16], e2v[16];
   if ( SMALL )
   {
      if ( lane < 16 )
      {
         bc[lane] = (lane < n) ? ds : 4.0;
         bc[16 + lane] = (lane + 1 < n) ? e2 : 0.0;
      }
      S1_WSYNC();
#pragma unroll
      for (int i = 0; i < 16; ++i)
      {
         dsv[i] = bc[i];
         e2v[i] = bc[16 + i];
      }
   }
   bool first = true;
   const double flane1 = (double) (lane + 1);
   for (int round = 0; round < 14; ++round)
   {
      /* (multiplications by the rounded reciprocals: an FP64 division is 30 dependent instructions, two per round were a quarter of it) */
      const double wdt = (hi - lo) * (1.0 / 65.0);
      double x = lo + wdt * flane1;
      if ( first )
         x = (lane == 63) ? 0.0 : lo + (hi - lo) * flane1 * (1.0 / 64.0);
      /* p_0 = 1, p_1 = d_0 - x, p_{i+1} = (d_i - x) p_i - e_{i-1}^2 p_{i-1}: a sign change = an eigenvalue below x (an exact zero
       * counts as negative); rescaled every eighth step */
      double pp = 1.0, pc = d0 - x;
      bool posc = pc > 0.0;
      bool below = !posc;
      if ( SMALL )
      {
         pc = dsv[0] - x;
         posc = pc > 0.0;
         below = !posc;
#pragma unroll
         for (int i = 1; i < 16; ++i)
         {
            const double pn = fma(dsv[i] - x, pc, -e2v[i - 1] * pp);
            const bool posn = pn > 0.0;
            below = below || (posn != posc);
            pp = pc; pc = pn; posc = posn;
            if ( i == 8 )
            {
               const int ex = -max(__builtin_amdgcn_frexp_exp(pc), __builtin_amdgcn_frexp_exp(pp));
               const bool okx = ex > -1000 && ex < 1000;
               pc = okx ? ldexp(pc, ex) : pc;
               pp = okx ? ldexp(pp, ex) : pp;
            }
         }
      }
      else
      for (int i = 1; i < n; ++i)
      {
         const double di = s1_lane(ds, i), ei = s1_lane(e2, i - 1);
         const double pn = fma(di - x, pc, -ei * pp);
         const bool posn = pn > 0.0;
         below = below || (posn != posc);
         pp = pc; pc = pn; posc = posn;
         if ( (i & 7) == 7 )
         {
            const int ex = -max(__builtin_amdgcn_frexp_exp(pc), __builtin_amdgcn_frexp_exp(pp));
            if ( ex > -1000 && ex < 1000 )
            {
               pc = ldexp(pc, ex);
               pp = ldexp(pp, ex);
            }
         }
      }
      const unsigned long long msk = __ballot(below);
      if ( first )
      {
         first = false;
         if ( !(msk >> 63) )
            return 0.0;                                     /* nothing below zero */
         const int f = __ffsll((long long) msk) - 1;        /* first shift with an eigenvalue below it */
         const double w64 = (hi - lo) * (1.0 / 64.0);
         const double nlo = (f == 0) ? lo : lo + w64 * (double) f;
         const double nhi = (f == 63) ? 0.0 : lo + w64 * (double) (f + 1);
         lo = nlo; hi = nhi;
      }
      else
      {
         const int f = msk ? __ffsll((long long) msk) - 1 : 64;
         const double nlo = lo + wdt * (double) f;
         const double nhi = (f < 64) ? lo + wdt * (double) (f + 1) : hi;
         lo = nlo; hi = nhi;
      }
      if ( tprof != NULL && lane == 0 )
         tprof[2] += 1.0;
      if ( hi - lo <= 1e-10 * fabs(lo) || fabs(lo) < 1e-15 )
         break;
   }
   if ( tprof != NULL && lane == 0 )
      tprof[1] += (double) clock64();
   return lo * nrm;
}

/* ---- the same for n <= 16 with the matrix in registers, the reduction fully unrolled.  Round 6: TWO lanes per row - lane l < 32
 * serves row l & 15 and holds the eight entries of its row whose column has the parity l >> 4 - where one lane held all sixteen: the
 * row's entry of A v is acc0 + acc1 of one chain over the even and one over the odd columns, each lane now forms ITS chain and updates
 * its eight entries, the two meet through one lane exchange (half the multiply-adds and half the LDS reads of v and w per lane; the
 * section costs its instruction count).  The sums over rows are taken in lanes 0 .. 15 as before: THE SAME BITS.
 * The entries of the reflector and of w reach the lanes through 16 doubles of LDS each (v_readlane would make them scalar values, of
 * which the kernel has none to spare); no other LDS access after the rows are loaded. */
__device__ __forceinline__ double s1_lmin16(const double* W, int n, int p, int lane, double* bc, double* tprof)
{
   const int row = lane & 15, half = (lane >> 4) & 1;
   const bool live = lane < 32 && row < n;
   double ah[8];                                  /* ah[i] = entry (row, 2 i + half) */
#pragma unroll
   for (int i = 0; i < 8; ++i)
      ah[i] = (live && 2 * i + half < n) ? W[row * p + 2 * i + half] : 0.0;
   double dreg = 0.0, ereg = 0.0;
#pragma unroll
   for (int k = 0; k < 14; ++k)
   {
      if ( k + 2 < n )
      {
         /* entry k of the own row: it lives in the lane of parity k & 1, the other one fetches it */
         const double mine = ah[k >> 1];
         const double theirs = __shfl_xor(mine, 16, 64);
         const double ek = (half == (k & 1)) ? mine : theirs;
         const double xa = (row > k) ? ek : 0.0;
         const double x0 = s1_lane(xa, k + 1);
         /* (the sums over rows exactly as the one-lane-per-row form wrote them - the lanes from 16 on sum garbage of their own that
          * nobody reads: the compiler fuses the first addition of the second sum with its product, a select in between would stop it) */
         const double s2 = s1_lane(s1_sum16(lane > k + 1 ? xa * xa : 0.0), 0);
         if ( lane == k )
            dreg = ek;
         if ( s2 != s2 )
            return s2;
         if ( !(s2 > 1e-290) )
         {
            if ( lane == k )
               ereg = x0;
         }
         else
         {
            const double h2 = x0 * x0 + s2;
            const double rh = s1_rsqrt(h2);
            const double beta = -copysign(h2 * rh, x0);
            const double t = (x0 - beta) * copysign(rh, x0);
            const double scale = s1_rcp(x0 - beta);
            /* the reflector is zero in the rows up to k, so the sums and the update run over all columns without guards */
            const double vl = (row == k + 1) ? 1.0 : ((row > k + 1) ? xa * scale : 0.0);
            if ( lane == k )
               ereg = beta;
            if ( lane < 16 )
               bc[lane] = vl;
            S1_WSYNC();
            double vh[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
               vh[i] = bc[2 * i + half];
            double acc = 0.0;
#pragma unroll
            for (int i = 0; i < 8; ++i)
               acc = fma(ah[i], vh[i], acc);
            const double oth = __shfl_xor(acc, 16, 64);
            const double pl = (row > k) ? t * (half == 0 ? acc + oth : oth + acc) : 0.0;
            const double pv = s1_lane(s1_sum16(pl * vl), 0);
            const double wl = fma(-0.5 * t * pv, vl, pl);
            if ( lane < 16 )
               bc[16 + lane] = wl;
            S1_WSYNC();
#pragma unroll
            for (int i = 0; i < 8; ++i)
               ah[i] -= fma(vl, bc[16 + 2 * i + half], wl * vh[i]);
         }
      }
   }
   /* the rest of the diagonal and the last off-diagonal entry: entries row and row - 1 of the own row, from the lane that holds them */
   double ownm = 0.0, leftm = 0.0;
#pragma unroll
   for (int i = 0; i < 8; ++i)
   {
      if ( 2 * i + half == row ) ownm = ah[i];
      if ( 2 * i + half + 1 == row ) leftm = ah[i];
   }
   const double ownt = __shfl_xor(ownm, 16, 64), leftt = __shfl_xor(leftm, 16, 64);
   const double own = (half == (row & 1)) ? ownm : ownt;
   const double left = (half == ((row + 1) & 1)) ? leftm : leftt;
   if ( lane < 16 && lane + 2 >= n )
      dreg = own;
   if ( n >= 2 )
   {
      const double elast = s1_lane(left, n - 1);
      if ( lane == n - 2 )
         ereg = elast;
   }
   if ( lane >= 16 )
   {
      dreg = 0.0; ereg = 0.0;
   }
   if ( tprof != NULL && lane == 0 )
      tprof[0] -= (double) clock64();
   const double r = s1_sturm_min<true>(dreg, ereg, n, lane, bc, tprof);
   if ( tprof != NULL && lane == 0 )
      tprof[0] += (double) clock64();
   return r;
}

/* ---- one wavefront: smallest eigenvalue of the symmetric n x n matrix whose LOWER triangle is in W (LDS, pitch p odd; destroyed):
 * Householder tridiagonalisation (lane = row), Sturm multisection with 64 shifts per round.  Returns min(lambda_min, 0) to a
 * relative accuracy of 1e-11 (from below), NaN when the matrix is not finite.  scr: 4 * n doubles of LDS. */
__device__ __forceinline__ double s1_lmin(double* W, int n, int p, int lane, double* scr, double* tprof)
{
   double* dd = scr;
   double* ee = scr + n;
   double* vv = scr + 2 * n;
   double* ww = scr + 3 * n;
   const bool small = n <= 16;                  /* the rows sit in the first 16 lanes: reductions inside one DPP row */
   for (int k = 0; k + 2 < n; ++k)
   {
      const double xa = (lane > k && lane < n) ? W[lane * p + k] : 0.0;
      const double x0 = s1_lane(xa, k + 1);
      const double s2 = small ? s1_lane(s1_sum16(lane > k + 1 ? xa * xa : 0.0), 0) : s1_wsum(lane > k + 1 ? xa * xa : 0.0);
      if ( lane == 0 )
         dd[k] = W[k * p + k];
      if ( !(s2 > 1e-290) )
      {
         if ( lane == 0 )
            ee[k] = x0;
         if ( s2 != s2 )
            return s2;
         continue;
      }
      const double h2 = x0 * x0 + s2;
      const double rh = s1_rsqrt(h2);
      const double beta = -copysign(h2 * rh, x0);
      const double t = (x0 - beta) * copysign(rh, x0);          /* (beta - x0) / beta with 1 / beta = -sign(x0) / sqrt(h2) */
      const double scale = s1_rcp(x0 - beta);
      const double vl = (lane == k + 1) ? 1.0 : xa * scale;          /* lanes > k */
      if ( lane > k && lane < n )
         vv[lane] = vl;
      if ( lane == 0 )
         ee[k] = beta;
      S1_WSYNC();
      /* p = t A v over the trailing block (rows, columns k + 1 .. n - 1), lower storage */
      double pl = 0.0;
      if ( lane > k && lane < n )
      {
         double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
         int c = k + 1;
         /* (eight columns per round trip to LDS - sixteen loads in flight - where four were: the same sums in the same order, the
          * second four go to the accumulators after the first) */
         for (; c + 7 < n; c += 8)
         {
            double ev[8], vc[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
            {
               ev[u] = W[(c + u <= lane) ? lane * p + c + u : (c + u) * p + lane];
               vc[u] = vv[c + u];
            }
            a0 = fma(ev[0], vc[0], a0); a1 = fma(ev[1], vc[1], a1); a2 = fma(ev[2], vc[2], a2); a3 = fma(ev[3], vc[3], a3);
            a0 = fma(ev[4], vc[4], a0); a1 = fma(ev[5], vc[5], a1); a2 = fma(ev[6], vc[6], a2); a3 = fma(ev[7], vc[7], a3);
         }
         for (; c + 3 < n; c += 4)
         {
            double ev[4], vc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
            {
               ev[u] = W[(c + u <= lane) ? lane * p + c + u : (c + u) * p + lane];
               vc[u] = vv[c + u];
            }
            a0 = fma(ev[0], vc[0], a0); a1 = fma(ev[1], vc[1], a1); a2 = fma(ev[2], vc[2], a2); a3 = fma(ev[3], vc[3], a3);
         }
         {
            double ev[3], vc[3];
#pragma unroll
            for (int u = 0; u < 3; ++u)
            {
               const bool in = (c + u < n);
               ev[u] = in ? W[(c + u <= lane) ? lane * p + c + u : (c + u) * p + lane] : 0.0;
               vc[u] = in ? vv[c + u] : 0.0;
            }
            a0 = fma(ev[0], vc[0], a0); a1 = fma(ev[1], vc[1], a1); a2 = fma(ev[2], vc[2], a2);
         }
         pl = t * ((a0 + a1) + (a2 + a3));
      }
      const double pvl = (lane > k && lane < n) ? pl * vl : 0.0;
      const double pv = small ? s1_lane(s1_sum16(pvl), 0) : s1_wsum(pvl);
      const double al = -0.5 * t * pv;
      const double wl = pl + al * vl;
      if ( lane > k && lane < n )
         ww[lane] = wl;
      S1_WSYNC();
      if ( lane > k && lane < n )
      {
         double* rl = W + lane * p;
         int c = k + 1;
         for (; c + 7 <= lane; c += 8)
         {
            double r8[8], w8[8], v8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
            {
               r8[u] = rl[c + u]; w8[u] = ww[c + u]; v8[u] = vv[c + u];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
               rl[c + u] = r8[u] - (vl * w8[u] + wl * v8[u]);
         }
         for (; c + 3 <= lane; c += 4)
         {
            double r4[4], w4[4], v4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
            {
               r4[u] = rl[c + u]; w4[u] = ww[c + u]; v4[u] = vv[c + u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
               rl[c + u] = r4[u] - (vl * w4[u] + wl * v4[u]);
         }
         for (; c <= lane; ++c)
            rl[c] -= vl * ww[c] + wl * vv[c];
      }
      S1_WSYNC();
   }
   if ( lane == 0 )
   {
      if ( n >= 2 )
      {
         dd[n - 2] = W[(n - 2) * p + n - 2];
         ee[n - 2] = W[(n - 1) * p + n - 2];
      }
      dd[n - 1] = W[(n - 1) * p + n - 1];
      ee[n - 1] = 0.0;
   }
   S1_WSYNC();
   /* the tridiagonal matrix into the lanes (lane i: d_i, e_i), then the multisection */
   const double dl = (lane < n) ? dd[lane] : 0.0;
   const double el = (lane < n) ? ee[lane] : 0.0;
   return s1_sturm_min<false>(dl, el, n, lane, scr, tprof);
}

/* ---- the same for 17 <= n <= 32 with TWO lanes per row (round 6): lane l serves row l & 31, half l >> 5.  The section costs its
 * instruction count (profiles/r06_solve1_diag_block_attempt.txt), and at n = 32 it was 190 000 of the 830 000 cycles of an iteration,
 * twice, on one wavefront with the others idle; with two lanes per row every lane forms half of its row's entry of A v and applies half
 * of its row's part of the rank-2 update.  THE SAME BITS as s1_lmin: the entry of A v is ((a0 + a1) + (a2 + a3)) of four chains over the
 * columns k + 1 + j + 4 i - the first half owns chains 0 and 1, the second 2 and 3, the two partial sums meet through one lane
 * exchange -; the sums over rows (s1_wsum) see the same values in the same lanes, zeros in the lanes from 32 on as before. */
__device__ __forceinline__ double s1_lmin2(double* W, int n, int p, int lane, double* scr, double* tprof)
{
   double* dd = scr;
   double* ee = scr + n;
   double* vv = scr + 2 * n;
   double* ww = scr + 3 * n;
   const int row = lane & 31, half = lane >> 5;
   for (int k = 0; k + 2 < n; ++k)
   {
      const bool act = row > k && row < n;
      const double xa = act ? W[row * p + k] : 0.0;
      const double x0 = s1_lane(xa, k + 1);
      const double s2 = s1_wsum((half == 0 && lane > k + 1) ? xa * xa : 0.0);
      if ( lane == 0 )
         dd[k] = W[k * p + k];
      if ( !(s2 > 1e-290) )
      {
         if ( lane == 0 )
            ee[k] = x0;
         if ( s2 != s2 )
            return s2;
         continue;
      }
      const double h2 = x0 * x0 + s2;
      const double rh = s1_rsqrt(h2);
      const double beta = -copysign(h2 * rh, x0);
      const double t = (x0 - beta) * copysign(rh, x0);          /* (beta - x0) / beta with 1 / beta = -sign(x0) / sqrt(h2) */
      const double scale = s1_rcp(x0 - beta);
      const double vl = (row == k + 1) ? 1.0 : xa * scale;           /* rows > k */
      if ( act && half == 0 )
         vv[row] = vl;
      if ( lane == 0 )
         ee[k] = beta;
      S1_WSYNC();
      /* p = t A v over the trailing block (rows, columns k + 1 .. n - 1), lower storage: this lane's two chains */
      double part = 0.0;
      if ( act )
      {
         double ca = 0.0, cb = 0.0;
         int c = k + 1;
         const int h2c = 2 * half;
#define S1_WSYM(cc) W[((cc) <= row) ? row * p + (cc) : (cc) * p + row]
         for (; c + 7 < n; c += 8)
         {
            const int c0 = c + h2c;
            const double e0 = S1_WSYM(c0), e1 = S1_WSYM(c0 + 1), e2 = S1_WSYM(c0 + 4), e3 = S1_WSYM(c0 + 5);
            const double v0 = vv[c0], v1 = vv[c0 + 1], v2 = vv[c0 + 4], v3 = vv[c0 + 5];
            ca = fma(e0, v0, ca); cb = fma(e1, v1, cb);
            ca = fma(e2, v2, ca); cb = fma(e3, v3, cb);
         }
         for (; c + 3 < n; c += 4)
         {
            const int c0 = c + h2c;
            const double e0 = S1_WSYM(c0), e1 = S1_WSYM(c0 + 1);
            const double v0 = vv[c0], v1 = vv[c0 + 1];
            ca = fma(e0, v0, ca); cb = fma(e1, v1, cb);
         }
         {
            /* (the last one to three columns: chains 0, 1, 2 take a term each - a product of zeros past the end, as in s1_lmin -, chain 3 none) */
            const int c0 = c + h2c;
            const bool in0 = c0 < n, in1 = c0 + 1 < n;
            const int c0c = in0 ? c0 : k + 1, c1c = in1 ? c0 + 1 : k + 1;
            const double e0l = S1_WSYM(c0c), e1l = S1_WSYM(c1c);
            const double v0l = vv[c0c], v1l = vv[c1c];
            ca = fma(in0 ? e0l : 0.0, in0 ? v0l : 0.0, ca);
            if ( half == 0 )
               cb = fma(in1 ? e1l : 0.0, in1 ? v1l : 0.0, cb);
         }
#undef S1_WSYM
         part = ca + cb;
      }
      const double other = __shfl_xor(part, 32, 64);
      const double pl = act ? t * (half == 0 ? part + other : other + part) : 0.0;
      const double pvl = (half == 0 && act) ? pl * vl : 0.0;
      const double pv = s1_wsum(pvl);
      const double al = -0.5 * t * pv;
      const double wl = pl + al * vl;
      if ( act && half == 0 )
         ww[row] = wl;
      S1_WSYNC();
      if ( act )
      {
         /* columns k + 1 .. row of the row: the first half of them here, the second in the other lane of the row */
         double* rl = W + row * p;
         const int cnt = row - k;
         const int mid = k + 1 + ((cnt + 1) >> 1);
         int c = half ? mid : k + 1;
         const int ce = half ? row + 1 : mid;
         for (; c + 4 <= ce; c += 4)
         {
            double r4[4], w4[4], v4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
            {
               r4[u] = rl[c + u]; w4[u] = ww[c + u]; v4[u] = vv[c + u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
               rl[c + u] = r4[u] - (vl * w4[u] + wl * v4[u]);
         }
         for (; c < ce; ++c)
            rl[c] -= vl * ww[c] + wl * vv[c];
      }
      S1_WSYNC();
   }
   if ( lane == 0 )
   {
      if ( n >= 2 )
      {
         dd[n - 2] = W[(n - 2) * p + n - 2];
         ee[n - 2] = W[(n - 1) * p + n - 2];
      }
      dd[n - 1] = W[(n - 1) * p + n - 1];
      ee[n - 1] = 0.0;
   }
   S1_WSYNC();
   const double dl = (lane < n) ? dd[lane] : 0.0;
   const double el = (lane < n) ? ee[lane] : 0.0;
   return s1_sturm_min<false>(dl, el, n, lane, scr, tprof);
}

/* ---- n x n x n product on the matrix cores: wavefront `wave` of the subset [w0, w0 + nw) takes the 16 x 16 tiles tbase + t with
 * (tbase + t) % nw == wave - w0.  la(i, k), lb(k, j): operand entries (called only inside the matrix); ep(i, j, value). */
template<class LA, class LB, class EP>
__device__ __forceinline__ void s1_mmk(int n, int nk, int wave, int lane, int w0, int nw, int& tbase, LA la, LB lb, EP ep);
template<class LA, class LB, class EP>
__device__ __forceinline__ void s1_mm(int n, int wave, int lane, int w0, int nw, int& tbase, LA la, LB lb, EP ep)
{
   s1_mmk(n, n, wave, lane, w0, nw, tbase, la, lb, ep);
}
/* (n x n result, inner dimension nk) */
template<class LA, class LB, class EP>
__device__ __forceinline__ void s1_mmk(int n, int nk, int wave, int lane, int w0, int nw, int& tbase, LA la, LB lb, EP ep)
{
   const int nt = (n + 15) >> 4;
   const int ntile = nt * nt;
   if ( wave >= w0 && wave < w0 + nw )
   {
      const int lr = lane & 15, kq = lane >> 4;
      for (int t = 0; t < ntile; ++t)
      {
         if ( (tbase + t) % nw != w0 + nw - 1 - wave )
            continue;
         const int ti = t / nt, tj = t - ti * nt;
         const int ri = 16 * ti + lr, cj = 16 * tj + lr;
         v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#ifdef S1_NO_MFMA
         /* (developer switch: the same tile with scalar multiply-adds, for accuracy comparisons) */
         for (int k = 0; k < nk; ++k)
         {
            const double b = (cj < n) ? lb(k, cj) : 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r)
            {
               const int row = 16 * ti + kq + 4 * r;
               acc0[r] = fma((row < n) ? la(row, k) : 0.0, b, acc0[r]);
            }
         }
#else
         /* (operands loaded unconditionally at clamped indices and masked afterwards: a guarded load is a branch around it - four
          * per K step, most of the time of a tile) */
         const int ric = min(ri, n - 1), cjc = min(cj, n - 1);
#pragma unroll 2
         for (int kk = 0; kk < nk; kk += 8)
         {
            const int k0 = kk + kq, k1 = kk + 4 + kq;
            const int k0c = min(k0, nk - 1), k1c = min(k1, nk - 1);
            const double a0l = la(ric, k0c), b0l = lb(k0c, cjc), a1l = la(ric, k1c), b1l = lb(k1c, cjc);
            const double a0 = (ri < n && k0 < nk) ? a0l : 0.0;
            const double b0 = (cj < n && k0 < nk) ? b0l : 0.0;
            const double a1 = (ri < n && k1 < nk) ? a1l : 0.0;
            const double b1 = (cj < n && k1 < nk) ? b1l : 0.0;
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc1, 0, 0, 0);
         }
#endif
#pragma unroll
         for (int r = 0; r < 4; ++r)
         {
            const int row = 16 * ti + kq + 4 * r;
            if ( row < n && cj < n )
               ep(row, cj, acc0[r] + acc1[r]);
         }
      }
   }
   tbase += ntile;
}


/* two products of the same shape at once, D1 = A1 B1 and D2 = A2 B2, with one epilogue ep(i, j, d1, d2).  Used for sym(T Zinv): the
 * second product is Zinv T^T, i.e. the transposed tile in the same lanes, so that H = sigma mu Zinv - X - sym(T Zinv) leaves the
 * product's epilogue instead of a phase of its own. */
template<class LA, class LB, class LA2, class LB2, class EP>
__device__ __forceinline__ void s1_mm2(int n, int wave, int lane, int w0, int nw, int& tbase, LA la, LB lb, LA2 la2, LB2 lb2, EP ep)
{
   const int nt = (n + 15) >> 4;
   const int ntile = nt * nt;
   if ( wave >= w0 && wave < w0 + nw )
   {
      const int lr = lane & 15, kq = lane >> 4;
      for (int t = 0; t < ntile; ++t)
      {
         if ( (tbase + t) % nw != w0 + nw - 1 - wave )
            continue;
         const int ti = t / nt, tj = t - ti * nt;
         const int ri = 16 * ti + lr, cj = 16 * tj + lr;
         v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
         v4d bcc0 = {0.0, 0.0, 0.0, 0.0}, bcc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 2
         for (int kk = 0; kk < n; kk += 8)
         {
            const int k0 = kk + kq, k1 = kk + 4 + kq;
            const bool ra0 = ri < n && k0 < n, rb0 = cj < n && k0 < n, ra1 = ri < n && k1 < n, rb1 = cj < n && k1 < n;
            const int ric = min(ri, n - 1), cjc = min(cj, n - 1), k0c = min(k0, n - 1), k1c = min(k1, n - 1);
            const double a0l = la(ric, k0c), b0l = lb(k0c, cjc), a1l = la(ric, k1c), b1l = lb(k1c, cjc);
            const double c0l = la2(ric, k0c), d0l = lb2(k0c, cjc), c1l = la2(ric, k1c), d1l = lb2(k1c, cjc);
            const double a0 = ra0 ? a0l : 0.0;
            const double b0 = rb0 ? b0l : 0.0;
            const double a1 = ra1 ? a1l : 0.0;
            const double b1 = rb1 ? b1l : 0.0;
            const double c0 = ra0 ? c0l : 0.0;
            const double d0 = rb0 ? d0l : 0.0;
            const double c1 = ra1 ? c1l : 0.0;
            const double d1 = rb1 ? d1l : 0.0;
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc0, 0, 0, 0);
            bcc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(c0, d0, bcc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc1, 0, 0, 0);
            bcc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(c1, d1, bcc1, 0, 0, 0);
         }
#pragma unroll
         for (int r = 0; r < 4; ++r)
         {
            const int row = 16 * ti + kq + 4 * r;
            if ( row < n && cj < n )
               ep(row, cj, acc0[r] + acc1[r], bcc0[r] + bcc1[r]);
         }
      }
   }
   tbase += ntile;
}

/* exclusive prefix sums of the counts a[0 .. len) in place, a[len] = total (all threads; contains barriers) */
__device__ __forceinline__ void s1_exscan(int* a, int len, S1Sh& sh, int tid)
{
   const int lane = tid & 63, wave = tid >> 6;
   const int chunk = (len + S1_NT - 1) / S1_NT;
   const int i0 = tid * chunk;
   int s = 0;
   for (int i = i0; i < i0 + chunk && i < len; ++i)
      s += a[i];
   int incl = s;
#pragma unroll
   for (int off = 1; off < 64; off <<= 1)
   {
      const int u = __shfl_up(incl, off, 64);
      if ( lane >= off )
         incl += u;
   }
   if ( lane == 63 )
      sh.wtot[wave] = incl;
   __syncthreads();
   int base = 0;
   for (int w = 0; w < wave; ++w)
      base += sh.wtot[w];
   int run = base + incl - s;
   for (int i = i0; i < i0 + chunk && i < len; ++i)
   {
      const int t = a[i];
      a[i] = run;
      run += t;
   }
   if ( tid == S1_NT - 1 )
      a[len] = run;
   __syncthreads();
}

/* the same by ONE wavefront (no barrier): independent scans run side by side on different wavefronts */
__device__ __forceinline__ void s1_exscan_wave(int* a, int len, int lane)
{
   const int chunk = (len + 63) >> 6;
   const int i0 = lane * chunk;
   int s = 0;
   for (int i = i0; i < i0 + chunk && i < len; ++i)
      s += a[i];
   int incl = s;
#pragma unroll
   for (int off = 1; off < 64; off <<= 1)
   {
      const int u = __shfl_up(incl, off, 64);
      if ( lane >= off )
         incl += u;
   }
   int run = incl - s;
   for (int i = i0; i < i0 + chunk && i < len; ++i)
   {
      const int t = a[i];
      a[i] = run;
      run += t;
   }
   if ( lane == 63 )
      a[len] = incl;
}

/* global -> LDS, four loads in flight per thread (a loop of load, store waits out the 700 ns of an L2 round trip per element) */
__device__ __forceinline__ void s1_copy_in(double* dst, const double* __restrict__ src, long long n, int tid)
{
   for (long long b = tid; b < n; b += 4 * S1_NT)
   {
      double v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
      {
         const long long i = b + u * S1_NT;
         v[u] = src[i < n ? i : n - 1];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
      {
         const long long i = b + u * S1_NT;
         if ( i < n )
            dst[i] = v[u];
      }
   }
}

/* LP part of the Schur matrix as a product (see lp_schur in the kernel): a function of its own - it is the rare case (dense LP rows),
 * and inlined its registers are added to what the iteration keeps alive around the whole-matrix-per-lane recurrences */
__device__ __attribute__((noinline)) void s1_lp_schur_mm(const double* Dl, const double* sx, double* Mx, int m1_, int q_, int pm1_, bool mpk_,
   int wave_, int lane, int w0_)
{
   /* (arguments arrive in vector registers: said to be uniform, the tile loops are scalar) */
   const int m1 = __builtin_amdgcn_readfirstlane(m1_), q = __builtin_amdgcn_readfirstlane(q_), pm1 = __builtin_amdgcn_readfirstlane(pm1_);
   const int wave = __builtin_amdgcn_readfirstlane(wave_), w0 = __builtin_amdgcn_readfirstlane(w0_);
   const bool mpk = __builtin_amdgcn_readfirstlane((int) mpk_) != 0;
   int tb = 0;
   s1_mmk(m1, q, wave, lane, w0, S1_NW - w0, tb,
      [&](int i, int kk) S1_INL { return Dl[kk * m1 + i] * sx[kk]; },
      [&](int kk, int j) S1_INL { return Dl[kk * m1 + j]; },
      [&](int i, int j, double v) S1_INL { if ( j <= i ) Mx[(mpk ? (i * (i + 1)) >> 1 : i * pm1) + j] = v; });
}

/* pointers to the lists and cold matrices: they live in LDS while it lasts, else in the workspace in global memory, and are kept as
 * generic pointers.  A generic load counts on both memory counters and the compiler waits for ALL outstanding loads before every
 * use; when everything is in LDS (every B&B-sized instance of the reference) the iteration is compiled with LDS-typed pointers. */
template<bool AL, class T> struct S1Ptr { typedef T* type; };
template<class T> struct S1Ptr<true, T> { typedef __attribute__((address_space(3))) T* type; };
template<bool AL, class T> __device__ __forceinline__ typename S1Ptr<AL, T>::type s1_lp(T* p) { return (typename S1Ptr<AL, T>::type) p; }
#define LP(x) s1_lp<AL>(x)

/* developer profile (prof_on == 2, history buffer given): in iteration 3 every wavefront notes when it reaches each barrier - which
 * wavefront a phase waits for, and how long the others idle */
#define S1_BAR() do { if ( P.prof_on == 2 && P.hist != NULL && it == 3 && lane == 0 && nbar < 60 ) P.hist[2048 + 8 * nbar + wave] = (double) clock64(); ++nbar; __syncthreads(); } while (0)
#define S1_SETUP_STAMP(i) do { if ( P.prof_on == 2 && P.hist != NULL && tid == 0 ) P.hist[2048 + 480 + (i)] = (double) (clock64() - t_start); } while (0)
#define S1_STAMP(id) do { if ( P.prof_on && tid == 0 ) { const long long t_ = clock64(); sh.prof[id] += (double) (t_ - sh.t_last); sh.t_last = t_; } } while (0)

__global__ void __launch_bounds__(S1_NT) S1_KERNEL(const hs_solve1_args P)
{
   extern __shared__ __attribute__((aligned(16))) double sm[];
   __shared__ S1Sh sh;
   const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
   const int m = P.m, m1 = P.m + 1, q = P.q, K = P.nblk;
   const long long t_start = clock64();
   const long long w_start = wall_clock64();

   /* ---- the node's deferred setters (objective, LP rows, gather of the active matrices, constant matrix, start point) */
   if ( P.ncmd > 0 )
   {
      const bool staged = P.cmd_bytes >= (long long) NC_BYTES && P.cmd_bytes <= (long long) S1_DYN_LDS;
      hs_run_node_cmds(reinterpret_cast<const NodeCmd*>(P.cmds), P.ncmd, reinterpret_cast<NodeCmd*>(sm), staged ? sm : (double*) NULL,
         staged ? P.cmd_bytes : 0);
      __syncthreads(); S1_SETUP_STAMP(0);
   }

#ifdef S1_DEBUG
   {
      const double poison = __longlong_as_double(0x7ff8dead0000beefLL);
      for (int e = tid; e < (int) (S1_DYN_LDS / 8); e += S1_NT)
         sm[e] = poison;
      for (int e = tid; e < S1_NW * S1_NRED; e += S1_NT)
         (&sh.red[0][0])[e] = poison;
      for (int e = tid; e < SC_COUNT; e += S1_NT)
         sh.sc[e] = poison;
      for (long long e = tid; e < P.gws_len; e += S1_NT)
         P.gws[e] = poison;
      if ( tid == 0 )
         atomicAdd(&s1_dbg[1], 1u);
      __syncthreads();
   }
#endif
   /* ---- layout */
   if ( tid == 0 )
   {
      s1_layout(m, q, K, P.n, sh.lay);
      for (int i = 0; i < 24; ++i)
         sh.prof[i] = 0.0;
      sh.t_last = t_start;
   }
   __syncthreads(); S1_SETUP_STAMP(1);
   const S1Lay& L = sh.lay;
   const int pm1 = L.pm1, pm = L.pm, VL = L.VL, QL = L.QL;
   const int oVec = L.oVec, oQ = L.oQ;
   double* const Mx = sm + L.oMx;
   double* const Lm = sm + L.oLm;
   /* row i of the extended Schur matrix starts at Mx + MROW(i): a packed lower triangle in the instance for m > 64 */
   const bool mpk = S1_MBIG && L.packedM;
#define MROW(i) (mpk ? ((i) * ((i) + 1)) >> 1 : (i) * pm1)
   /* an LDS word that holds 0.0 from the start of the solve and is never written: entry (0, 1) of the full extended matrix (its
    * upper triangle), the first spare word behind the packed one */
   const double* const zsrc = mpk ? Mx + ((m1 * (m1 + 1)) >> 1) : Mx + 1;
#define VEC(id) (sm + oVec + (id) * VL)
#define QV(id) (sm + oQ + (id) * QL)
   double* const out = P.out;
   /* (the extended Schur matrix starts from zeros: its upper triangle and the padding columns are never written and are read -
    * unmasked - by the substitutions with the factor that overwrites it) */
   for (int e = tid; e < (mpk ? (m1 * (m1 + 1) >> 1) + 16 : m1 * pm1); e += S1_NT)
      Mx[e] = 0.0;

   /* ---- flexible part: offset arrays first (their sizes follow from the shape), the counts decide the rest */
   if ( tid == 0 )
   {
      int ldsleft = (int) (S1_DYN_LDS / 8) - L.fixedEnd;
      double* lp = sm + L.fixedEnd;
      double* gp = P.gws;
      /* (1) the pool of U_j buffers of the Schur phase: up to 16 of the largest block */
      int npmax = 0;
      for (int k = 0; k < K; ++k) npmax = max(npmax, L.np[k]);
      int Rlen = L.Rlen;
      {
         const int want = S1_NW * npmax - Rlen;
         if ( want > 0 )
         {
            const int ext = min(want, max(0, ldsleft / 2)) & ~1;
            Rlen += ext; lp += ext; ldsleft -= ext;
         }
      }
      sh.lay.Rlen = Rlen;
      auto take = [&](long long cnt) S1_INL -> double*
      {
         cnt = (cnt + 1) & ~1LL;
         double* r;
         if ( cnt <= ldsleft ) { r = lp; lp += cnt; ldsleft -= (int) cnt; }
         else { r = gp; gp += cnt; }
         return r;
      };
      for (int k = 0; k < K; ++k)
      {
         S1Blk& B = sh.blk[k];
         B.n = L.n[k]; B.p = L.p[k]; B.np = L.np[k];
         B.oX = L.oX[k]; B.oZi = L.oZi[k]; B.oLx = L.oLx[k]; B.oLz = L.oLz[k]; B.odX = L.odX[k]; B.odZ = L.odZ[k];
         B.oT1 = L.oT1[k]; B.oT2 = L.oT2[k]; B.oEig = L.oEig[k];
         int G = S1_NW;
         while ( G > 1 && G * B.np > Rlen ) G >>= 1;
         B.G = G;
      }
      /* (2) cold matrices */
      for (int k = 0; k < K; ++k)
      {
         S1Blk& B = sh.blk[k];
         B.Z = take(B.np); B.Rd = take(B.np); B.E = take(B.np); B.B = take(B.np); B.XR = take(B.np);
      }
      /* (3) offset arrays */
      sh.roff = (int*) take((q + 2) / 2 + 1);
      sh.coff = (int*) take((m1 + 2) / 2 + 1);
      for (int k = 0; k < K; ++k)
      {
         S1Blk& B = sh.blk[k];
         B.voff = (int*) take((m1 + 2) / 2 + 1);
         B.poff = (int*) take((B.n * B.n + 2) / 2 + 1);
      }
      /* (4) the caller's dense arrays once into LDS, at its top, when half of what is left holds them: the passes below read
       * every entry three times in chains of dependent loads - 700 ns each from L2, 40 ns from LDS (the lists of example_TT took
       * 50 us of the 1.2 ms of a solve) */
      {
         long long need = (long long) q * m1;
         for (int k = 0; k < K; ++k)
            need += (long long) m1 * L.n[k] * L.n[k];
         need = (need + 1) & ~1LL;
         sh.fl[31] = -1; sh.fl[32] = 0;
         if ( need > 0 && need <= (ldsleft / 4) * 3 )
         {
            /* (should the lists then not fit below it, the second allocation step gives the area up again: lists in global
             * memory would cost every iteration what this saves once) */
            sh.fl[31] = (int) (S1_DYN_LDS / 8) - (int) need;
            sh.fl[32] = (int) need;
            ldsleft -= (int) need;
         }
      }
      sh.fl[0] = ldsleft;
      *(double**) &sh.sc[0] = lp;               /* (handed to the second allocation step below) */
      *(double**) &sh.sc[1] = gp;
   }
   __syncthreads(); S1_SETUP_STAMP(2);
   const double* Dsrc = P.Dext;
   const double* Asrc[S1_MAXB];
   for (int k = 0; k < K; ++k)
      Asrc[k] = P.A[k];
   if ( sh.fl[31] >= 0 )
   {
      double* dst = sm + sh.fl[31];
      const long long nd = (long long) q * m1;
      s1_copy_in(dst, P.Dext, nd, tid);
      Dsrc = dst;
      dst += nd;
      for (int k = 0; k < K; ++k)
      {
         const long long na = (long long) m1 * L.n[k] * L.n[k];
         s1_copy_in(dst, P.A[k], na, tid);
         Asrc[k] = dst;
         dst += na;
      }
      __syncthreads(); S1_SETUP_STAMP(3);
   }

   /* ---- counts: LP rows / columns, entries by variable / by position */
   for (int r = wave; r < q; r += S1_NW)
   {
      int cnt = 0;
      for (int c0 = 0; c0 < m1; c0 += 64)
      {
         const int c = c0 + lane;
         const bool nz = c < m1 && Dsrc[(long long) r * m1 + c] != 0.0;
         cnt += __popcll(__ballot(nz));
      }
      if ( lane == 0 ) sh.roff[r] = cnt;
   }
   for (int c = wave; c < m1; c += S1_NW)
   {
      int cnt = 0;
      for (int r0 = 0; r0 < q; r0 += 64)
      {
         const int r = r0 + lane;
         const bool nz = r < q && Dsrc[(long long) r * m1 + c] != 0.0;
         cnt += __popcll(__ballot(nz));
      }
      if ( lane == 0 ) sh.coff[c] = cnt;
   }
   for (int k = 0; k < K; ++k)
   {
      const S1Blk& B = sh.blk[k];
      const int n = B.n, n2 = n * n;
      const double* A = Asrc[k];
      for (int i = wave; i < m1; i += S1_NW)
      {
         int cnt = 0;
         for (int e0 = 0; e0 < n2; e0 += 64)
         {
            const int e = e0 + lane;
            const int r = s1_div(e, n), c = e - r * n;
            const bool nz = e < n2 && A[(long long) i * n2 + e] != 0.0;
            cnt += __popcll(__ballot(nz));
         }
         if ( lane == 0 ) B.voff[i] = cnt;
      }
      for (int e = tid; e < n2; e += S1_NT)
      {
         const int r = s1_div(e, n), c = e - r * n;
         int cnt = 0;
         if ( r >= c )
            for (int i = 0; i < m1; ++i)
               cnt += (A[(long long) i * n2 + e] != 0.0) ? 1 : 0;
         B.poff[e] = cnt;
      }
   }
   __syncthreads(); S1_SETUP_STAMP(4);
   /* (one wavefront per array, side by side) */
   for (int t = wave; t < 2 + 2 * K; t += S1_NW)
   {
      if ( t == 0 ) s1_exscan_wave(sh.roff, q, lane);
      else if ( t == 1 ) s1_exscan_wave(sh.coff, m1, lane);
      else if ( t & 1 ) s1_exscan_wave(sh.blk[(t - 2) >> 1].poff, sh.blk[(t - 2) >> 1].n * sh.blk[(t - 2) >> 1].n, lane);
      else s1_exscan_wave(sh.blk[(t - 2) >> 1].voff, m1, lane);
   }
   __syncthreads();
   /* second allocation step, by wavefront 0: the sizes are wavefront-uniform values (every lane computes them), lane 0 notes the
    * results; the variables are sorted into light and heavy ones 64 at a time (as a loop of one thread over the counts in LDS
    * this step took 26 000 cycles of the 130 000 of the setup of example_TT) */
   if ( wave == 0 )
   {
      int fl0 = sh.fl[0], fl31 = sh.fl[31];
      for (int attempt = 0; attempt < 2; ++attempt)
      {
      int ldsleft = fl0;
      double* lp = *(double**) &sh.sc[0];
      double* gp = *(double**) &sh.sc[1];
      double* const gp0 = gp;
      auto take = [&](long long cnt) S1_INL -> double*
      {
         cnt = (cnt + 1) & ~1LL;
         double* r;
         if ( cnt <= ldsleft ) { r = lp; lp += cnt; ldsleft -= (int) cnt; }
         else { r = gp; gp += cnt; }
         return r;
      };
      const int nnzD = sh.roff[q];
      {
         double* const rval = take(nnzD); double* const cval = take(nnzD);
         unsigned short* const rcol = (unsigned short*) take((nnzD + 3) / 4); unsigned short* const crow = (unsigned short*) take((nnzD + 3) / 4);
         if ( lane == 0 )
         {
            sh.rval = rval; sh.cval = cval; sh.rcol = rcol; sh.crow = crow;
         }
      }
      double work = 0.0;
      int nnzA = 0;
      for (int k = 0; k < K; ++k)
      {
         S1Blk& B = sh.blk[k];
         const int bn = B.n;
         const int* const voff = B.voff;
         const int nz = voff[m1];                          /* entries of all matrices, both triangles */
         const int nzl = B.poff[bn * bn];                  /* entries with row >= col */
         nnzA += nzl;
         double* const vval = take(nz); double* const pval = take(nzl);
         unsigned* const vpq = (unsigned*) take((nz + 1) / 2); unsigned short* const pvar = (unsigned short*) take((nzl + 3) / 4);
         unsigned short* const lv = (unsigned short*) take((m1 + 3) / 4); unsigned short* const hv = (unsigned short*) take((m1 + 3) / 4);
         /* light variables go through the pair formula (a thread per pair), heavy ones through U_j = X A_j Zinv; both lists ascending */
         int nl = 0, nh = 0, nzh = 0, nzlight = 0;
         for (int i0 = 0; i0 < m1; i0 += 64)
         {
            const int i = i0 + lane;
            const int c = (i < m1) ? voff[i + 1] - voff[i] : 0;
            const bool isl = c > 0 && c <= S1_LIGHT_MAX, ish = c > S1_LIGHT_MAX;
            const unsigned long long ml = __ballot(isl), mh = __ballot(ish);
            const unsigned long long below = (1ULL << lane) - 1ULL;
            if ( isl ) lv[nl + __popcll(ml & below)] = (unsigned short) i;
            if ( ish ) hv[nh + __popcll(mh & below)] = (unsigned short) i;
            nl += __popcll(ml); nh += __popcll(mh);
            nzh += ish ? c : 0; nzlight += isl ? c : 0;
         }
         for (int off = 32; off > 0; off >>= 1)
         {
            nzh += __shfl_xor(nzh, off, 64);
            nzlight += __shfl_xor(nzlight, off, 64);
         }
         int* const lro = (int*) take((nl + 2) / 2 + 1);
         unsigned short* const lrp = (unsigned short*) take((nzlight + 3) / 4 + 1);
         int* const lre = (int*) take((nzlight + 2) / 2 + 1);
         double* const Tc = take((long long) nzlight * bn + 2);
         if ( lane == 0 )
         {
            B.vval = vval; B.pval = pval; B.vpq = vpq; B.pvar = pvar; B.lv = lv; B.hv = hv;
            B.nl = nl; B.nh = nh;
            B.lro = lro; B.lrp = lrp; B.lre = lre; B.Tc = Tc;
            B.nrs = 0;
         }
         work += (double) nzh * (double) bn + (double) nh * (double) bn * (double) (bn * bn) + (double) nz * (double) nh
            + 0.5 * (double) (nz - nzh) * (double) (nz - nzh) * 1.5;
      }
      if ( lane == 0 )
      {
         sh.fl[1] = (work > P.maxwork) ? 1 : 0;
         sh.fl[2] = nnzA; sh.fl[3] = nnzD;
         sh.fl[4] = ((long long) (gp - P.gws) > P.gws_len) ? 1 : 0;
         sh.fl[30] = (gp == P.gws) ? 1 : 0;                   /* nothing went to global memory */
      }
      if ( gp == gp0 || fl31 < 0 )
         break;
      /* a list went to global memory while the staged copy of the caller's arrays holds LDS: give that area up (the passes that
       * fill the lists read the caller's arrays again) and allocate once more */
      fl0 += sh.fl[32];
      fl31 = -1;
      }
      if ( lane == 0 )
      {
         sh.fl[0] = fl0;
         sh.fl[31] = fl31;
      }
   }
   __syncthreads(); S1_SETUP_STAMP(5);
   if ( sh.fl[31] < 0 )
   {
      Dsrc = P.Dext;
      for (int k = 0; k < K; ++k)
         Asrc[k] = P.A[k];
   }
   if ( sh.fl[1] || sh.fl[4] )
   {
      /* more work than one compute unit should take (dense matrices), or a workspace that is too small: decline */
      if ( tid == 0 )
      {
         out[0] = -2.0; out[40] = (double) sh.fl[2]; out[41] = (double) sh.fl[3];
         __threadfence_system();
         if ( P.flag != NULL )
            __hip_atomic_store(P.flag, P.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      return;
   }

   /* ---- fill the lists */
   for (int r = wave; r < q; r += S1_NW)
   {
      int run = sh.roff[r];
      for (int c0 = 0; c0 < m1; c0 += 64)
      {
         const int c = c0 + lane;
         const double v = c < m1 ? Dsrc[(long long) r * m1 + c] : 0.0;
         const unsigned long long msk = __ballot(v != 0.0);
         if ( v != 0.0 )
         {
            const int pos = run + __popcll(msk & ((1ULL << lane) - 1ULL));
            sh.rcol[pos] = (unsigned short) c;
            sh.rval[pos] = v;
         }
         run += __popcll(msk);
      }
   }
   for (int c = wave; c < m1; c += S1_NW)
   {
      int run = sh.coff[c];
      for (int r0 = 0; r0 < q; r0 += 64)
      {
         const int r = r0 + lane;
         const double v = r < q ? Dsrc[(long long) r * m1 + c] : 0.0;
         const unsigned long long msk = __ballot(v != 0.0);
         if ( v != 0.0 )
         {
            const int pos = run + __popcll(msk & ((1ULL << lane) - 1ULL));
            sh.crow[pos] = (unsigned short) r;
            sh.cval[pos] = v;
         }
         run += __popcll(msk);
      }
   }
   for (int k = 0; k < K; ++k)
   {
      const S1Blk& B = sh.blk[k];
      const int n = B.n, n2 = n * n;
      const double* A = Asrc[k];
      for (int i = wave; i < m1; i += S1_NW)
      {
         int run = B.voff[i];
         for (int e0 = 0; e0 < n2; e0 += 64)
         {
            const int e = e0 + lane;
            const int r = s1_div(e, n), c = e - r * n;
            const double v = (e < n2) ? A[(long long) i * n2 + e] : 0.0;
            const unsigned long long msk = __ballot(v != 0.0);
            if ( v != 0.0 )
            {
               const int pos = run + __popcll(msk & ((1ULL << lane) - 1ULL));
               B.vpq[pos] = ((unsigned) r << 16) | (unsigned) c;
               B.vval[pos] = v;
            }
            run += __popcll(msk);
         }
      }
      for (int e = tid; e < n2; e += S1_NT)
      {
         const int r = s1_div(e, n), c = e - r * n;
         if ( r < c )
            continue;
         int pos = B.poff[e];
         for (int i = 0; i < m1; ++i)
         {
            const double v = A[(long long) i * n2 + e];
            if ( v != 0.0 )
            {
               B.pvar[pos] = (unsigned short) i;
               B.pval[pos] = v;
               ++pos;
            }
         }
      }
   }
   __syncthreads(); S1_SETUP_STAMP(6);
   /* which form of the LP part of the Schur matrix (lp_schur): walking the nonzeros costs what the busiest lane does - the entries of
    * all LP rows its variable appears in, about 500 cycles each -, the product about 800 cycles per eight LP rows and tile */
   {
      int mine = 0;
      for (int i = tid + 1; i < m1; i += S1_NT)
      {
         int c = 0;
         for (int t = sh.coff[i]; t < sh.coff[i + 1]; ++t)
            c += sh.roff[sh.crow[t] + 1] - sh.roff[sh.crow[t]];
         mine = max(mine, c);
      }
      for (int off = 32; off > 0; off >>= 1)
         mine = max(mine, __shfl_xor(mine, off, 64));
      if ( lane == 0 )
         sh.wtot[wave] = mine;
   }
   __syncthreads();
   if ( tid == 0 )
   {
      int worst = 0;
      for (int w = 0; w < S1_NW; ++w)
         worst = max(worst, sh.wtot[w]);
      const int nt1 = (m1 + 15) >> 4;
      const int nwv = S1_NW - ((2 * K < S1_NW - 1) ? 2 * K : S1_NW - 1);
      /* (a K step of a tile: about 800 cycles out of LDS, 2400 out of global memory - the latency of its four loads) */
      const double mmcost = (double) ((nt1 * nt1 + nwv - 1) / nwv) * (double) ((q + 7) >> 3) * (sh.fl[31] >= 0 ? 800.0 : 2400.0);
      const double rowcost = 500.0 * (double) worst + 3000.0;        /* (three dependent LDS round trips per entry) */
      sh.fl[33] = (mmcost < rowcost) ? (sh.fl[31] >= 0 ? 1 : 2) : 0;
      /* neither form affordable (hundreds of dense LP rows): longer than the whole iteration of the general path - decline */
      sh.fl[34] = (fmin(mmcost, rowcost) > 2e6) ? 1 : 0;
   }
   __syncthreads();
   if ( sh.fl[34] )
   {
      if ( tid == 0 )
      {
         out[0] = -2.0; out[40] = (double) sh.fl[2]; out[41] = (double) sh.fl[3];
         __threadfence_system();
         if ( P.flag != NULL )
            __hip_atomic_store(P.flag, P.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      return;
   }
   /* row slots of the light matrices */
   for (int k = 0; k < K; ++k)
   {
      S1Blk& B = sh.blk[k];
      for (int a = tid; a < B.nl; a += S1_NT)
      {
         const int i = B.lv[a];
         int cnt = 0, prev = -1;
         for (int t = B.voff[i]; t < B.voff[i + 1]; ++t)
         {
            const int pp = (int) (B.vpq[t] >> 16);
            if ( pp != prev ) { ++cnt; prev = pp; }
         }
         B.lro[a] = cnt;
      }
      __syncthreads(); S1_SETUP_STAMP(7);
      s1_exscan(B.lro, B.nl, sh, tid);
      for (int a = tid; a < B.nl; a += S1_NT)
      {
         const int i = B.lv[a];
         int sidx = B.lro[a] - 1, prev = -1, start = 0, len = 0;
         for (int t = B.voff[i]; t < B.voff[i + 1]; ++t)
         {
            const int pp = (int) (B.vpq[t] >> 16);
            if ( pp != prev )
            {
               if ( prev >= 0 )
                  B.lre[sidx] = (start << 6) | len;
               ++sidx;
               B.lrp[sidx] = (unsigned short) pp;
               start = t; len = 0;
               prev = pp;
            }
            ++len;
         }
         if ( prev >= 0 )
            B.lre[sidx] = (start << 6) | len;
      }
      if ( tid == 0 )
         B.nrs = B.lro[B.nl];
      __syncthreads(); S1_SETUP_STAMP(8);
   }
   /* objective, norms */
   if ( tid < m )
      VEC(V_b)[tid] = P.b[tid];
   __syncthreads(); S1_SETUP_STAMP(9);
   {
      double c2 = 0.0;
      for (int k = 0; k < K; ++k)
      {
         const S1Blk& B = sh.blk[k];
         const int n = B.n, n2 = n * n;
         for (int e = tid; e < n2; e += S1_NT)
         {
            const int t = B.poff[e];
            if ( t < B.poff[e + 1] && B.pvar[t] == 0 )
            {
               const int r = s1_div(e, n), c = e - r * n;
               c2 += (r == c ? 1.0 : 2.0) * B.pval[t] * B.pval[t];
            }
         }
      }
      for (int r = tid; r < q; r += S1_NT)
      {
         const int t = sh.roff[r];
         if ( t < sh.roff[r + 1] && sh.rcol[t] == 0 )
            c2 += sh.rval[t] * sh.rval[t];
      }
      c2 = s1_wsum(c2);
      if ( lane == 0 )
         sh.red[wave][RS_NC2] = c2;
      if ( wave == 0 )
      {
         double bsq = 0.0;
         for (int i = lane; i < m; i += 64)
            bsq = fma(VEC(V_b)[i], VEC(V_b)[i], bsq);
         const double nb2 = s1_wsum(bsq);
         if ( lane == 0 )
            sh.sc[SC_NORMB] = sqrt(nb2);
      }
   }
   __syncthreads(); S1_SETUP_STAMP(10);
   if ( tid == 0 )
   {
      double c2 = 0.0;
      for (int w = 0; w < S1_NW; ++w)
         c2 += sh.red[w][RS_NC2];
      sh.sc[SC_NORMC] = sqrt(c2);
   }
   __syncthreads(); S1_SETUP_STAMP(11);
   const double normb = s1_uni(sh.sc[SC_NORMB]), normC = s1_uni(sh.sc[SC_NORMC]);
   S1_STAMP(0);
   auto body = [&](auto ALtag) S1_INL
   {
   constexpr bool AL = decltype(ALtag)::value;

   /* ---- starting point */
   long long Nsum = q;
   for (int k = 0; k < K; ++k)
      Nsum += L.n[k];
   const double N1 = (double) (Nsum + 1);
   int warm = 0;
   int it = 0, nbar = 0;
   /* (thread index rotated by `off`: independent tasks of a phase start at different wavefronts - a loop that starts at thread 0
    * puts a block of 10 rows, 85 LP rows and the tile of a product all on wavefronts 0 and 1 while the other six wait at the
    * barrier; the products' tiles are dealt out from the last wavefront down) */
   auto tro = [&](int off) S1_INL -> int { const int t = tid - (off & (S1_NT - 1)); return t < 0 ? t + S1_NT : t; };
   /* LP part of the Schur matrix, D^T diag(x / z) D (needs x / z in Q_sx); it also clears Mx.  Two forms.
    * (a) When the dense copy of the caller's LP rows that the setup staged in LDS is still there and the cost model of the setup
    * prefers it (sh.fl[33] = 1: dense rows - cuts; = 2: the same from the caller's array in global memory when that is still cheaper
    * than walking the nonzeros): a product on
    * the matrix cores, (m + 1) x (m + 1) with inner dimension q, lower triangle written; wavefronts w0 .. NW - 1.
    * (b) Otherwise one wavefront walks the nonzeros: lane l owns the rows l + 1 (and l + 65) of Mx, it walks the LP rows its variable
    * appears in and adds their entries up to its own column; row 0 (the constant column, present in almost every bound row) has one
    * entry, Mx[0][0]: a reduction over the wavefront.  [Form (b) alone until the end of round 4: 7 400 cycles on example_TT's
    * bound rows, hidden beside the trial factorization - and 129 000 on 85 rows of density 0.3: eigenvector cuts are DENSE rows.
    * Form (a) from global memory: 26 000 cycles, the latency of 44 dependent-by-issue global loads per tile.] */
   auto lp_schur = [&](int w0) S1_INL
   {
      const double* sx = QV(Q_sx);
      if ( sh.fl[33] )
      {
         s1_lp_schur_mm(sh.fl[33] == 1 ? sm + sh.fl[31] : P.Dext, sx, Mx, m1, q, pm1, mpk, wave, lane, w0);
         return;
      }
      /* (round 6: the variables are dealt out over ALL the wavefronts from w0 on - variable i to wavefront w0 + (i - 1) % nwv -
       * where the last wavefront alone walked them 64 at a time: a lane's walk is what it was, the phase costs the busiest lane of
       * a wavefront instead of the sum over two rounds and over the divergent loops of 64 lanes - example_MkP: 45 000 cycles on
       * one wavefront with the others idle) */
      if ( wave < w0 )
         return;
      const int nwv = S1_NW - w0;
      for (int i = 1 + (wave - w0) + nwv * lane; i < m1; i += 64 * nwv)
      {
         double* row = Mx + MROW(i);
         for (int j = 0; j <= i; ++j)
            row[j] = 0.0;
         const int t1 = LP(sh.coff)[i + 1];
         for (int t = LP(sh.coff)[i]; t < t1; ++t)
         {
            const int r = LP(sh.crow)[t];
            const double sv = LP(sh.cval)[t] * sx[r];
            const int u1 = LP(sh.roff)[r + 1];
            for (int u = LP(sh.roff)[r]; u < u1; ++u)
            {
               const int j = LP(sh.rcol)[u];
               if ( j > i )
                  break;
               row[j] = fma(sv, LP(sh.rval)[u], row[j]);
            }
         }
      }
      if ( wave != S1_NW - 1 )
         return;
      double s00 = 0.0;
      const int t1 = LP(sh.coff)[1];
      for (int t = LP(sh.coff)[0] + lane; t < t1; t += 64)
      {
         const double cv0 = LP(sh.cval)[t];
         s00 = fma(cv0 * sx[LP(sh.crow)[t]], cv0, s00);
      }
      s00 = s1_wsum(s00);
      if ( lane == 0 )
         Mx[0] = s00;
   };
   auto trial_factor = [&](double alpha, bool fromglobal) S1_INL -> int
   {
      /* LxI <- X + alpha dXs (dXs in E), LzI <- Z + alpha dZs (dZs in B), then their Cholesky factors in place; returns the
       * failure flags (1: Z, 2: X).  fromglobal: the caller's start matrices, symmetrised, into X and Z as well. */
      for (int k = 0; k < K; ++k)
      {
         const S1Blk& B = sh.blk[k];
         const int n = B.n, p = B.p;
         for (int e = tid; e < n * n; e += S1_NT)
         {
            const int r = s1_div(e, n), c = e - r * n;
            double xv, zv;
            if ( fromglobal )
            {
               xv = 0.5 * (P.X[k][(long long) r * n + c] + P.X[k][(long long) c * n + r]);
               zv = 0.5 * (P.Z[k][(long long) r * n + c] + P.Z[k][(long long) c * n + r]);
               sm[B.oX + r * p + c] = xv;
               LP(B.Z)[r * p + c] = zv;
            }
            else
            {
               xv = sm[B.oX + r * p + c];
               zv = LP(B.Z)[r * p + c];
               if ( alpha != 0.0 )
               {
                  xv = fma(alpha, LP(B.E)[r * p + c], xv);
                  zv = fma(alpha, LP(B.B)[r * p + c], zv);
               }
            }
            sm[B.oLx + r * p + c] = xv;
            sm[B.oLz + r * p + c] = zv;
         }
      }
      for (int r = tro(128); r < q; r += S1_NT)
      {
         double xv = QV(Q_x)[r], zv = QV(Q_z)[r];
         if ( alpha != 0.0 )
         {
            xv = fma(alpha, QV(Q_dx)[r], xv);
            zv = fma(alpha, QV(Q_dz)[r], zv);
         }
         QV(Q_sx)[r] = xv / zv;
      }
      S1_BAR();
      /* (the LP part of the next Schur matrix rides along on the last wavefront: Mx is free between the corrector and the next
       * assembly, and a trial that fails repeats it) */
      {
         const int w0 = (2 * K < S1_NW - 1) ? 2 * K : S1_NW - 1;
         lp_schur(w0);
      }
      for (int t = wave; t < 2 * K; t += S1_NW)
      {
         const S1Blk& B = sh.blk[t >> 1];
         double dummy; int nfd;
         /* (blocks of at most S1U_MAXN rows: factor and inverse factor in one go, every lane for itself) */
         const int f = (S1_ALLU || B.n <= S1U_MAXN) ? s1u_chol_inv_n(sm + ((t & 1) ? B.oLz : B.oLx), B.n, B.p, lane)
            : s1_cholp(sm + ((t & 1) ? B.oLz : B.oLx), B.n, B.p, lane, false, 1.0, 0, true, dummy, nfd, zsrc);
         if ( lane == 0 )
            sh.fl[8 + t] = f;
      }
      S1_BAR();
      int ff = 0;
      for (int t = 0; t < 2 * K; ++t)
         if ( sh.fl[8 + t] != 0 )
            ff |= (t & 1) ? 1 : 2;
      return ff;
   };
   if ( P.have_start )
   {
      if ( tid < m ) VEC(V_y)[tid] = P.y[tid];
      for (int r = tid; r < q; r += S1_NT)
      {
         QV(Q_x)[r] = P.x[r];
         QV(Q_z)[r] = P.z[r];
      }
      __syncthreads();
      const int ff = trial_factor(0.0, true);
      /* interior?  mean complementarity */
      double xz = 0.0, bad = 0.0;
      for (int k = 0; k < K; ++k)
      {
         const S1Blk& B = sh.blk[k];
         const int n = B.n, p = B.p;
         for (int e = tid; e < n * n; e += S1_NT)
         {
            const int r = s1_div(e, n), c = e - r * n;
            xz += sm[B.oX + r * p + c] * LP(B.Z)[r * p + c];
         }
      }
      for (int r = tid; r < q; r += S1_NT)
      {
         const double xv = QV(Q_x)[r], zv = QV(Q_z)[r];
         xz += xv * zv;
         if ( !(xv > 0.0) || !(zv > 0.0) )
            bad = 1.0;
      }
      xz = s1_wsum(xz);
      bad = s1_wmax(bad);
      if ( lane == 0 )
      {
         sh.red[wave][RS_XZ] = xz;
         sh.red[wave][RS_RDMAX] = bad;
      }
      __syncthreads();
      double sxz = 0.0, sbad = 0.0;
      for (int w = 0; w < S1_NW; ++w)
      {
         sxz += sh.red[w][RS_XZ];
         sbad = fmax(sbad, sh.red[w][RS_RDMAX]);
      }
      const double mu0 = sxz / (double) (Nsum > 0 ? Nsum : 1);
      warm = (ff == 0 && sbad == 0.0 && mu0 > 0.0 && mu0 < 1e300) ? 1 : 0;
      __syncthreads();
      if ( warm && tid == 0 )
      {
         sh.sc[SC_TAU] = 1.0;
         sh.sc[SC_KAPPA] = mu0;
      }
   }
   if ( !warm )
   {
      const double xi = fmax(1.0, sqrt(fmax(fmax(normb, normC), 1.0)));
      if ( tid < m ) VEC(V_y)[tid] = 0.0;
      for (int r = tid; r < q; r += S1_NT)
      {
         QV(Q_x)[r] = xi;
         QV(Q_z)[r] = xi;
      }
      for (int k = 0; k < K; ++k)
      {
         const S1Blk& B = sh.blk[k];
         const int n = B.n, p = B.p;
         for (int e = tid; e < n * n; e += S1_NT)
         {
            const int r = s1_div(e, n), c = e - r * n;
            const double v = (r == c) ? xi : 0.0;
            sm[B.oX + r * p + c] = v;
            LP(B.Z)[r * p + c] = v;
         }
      }
      if ( tid == 0 )
      {
         sh.sc[SC_TAU] = 1.0;
         sh.sc[SC_KAPPA] = xi * xi;
      }
   }
   __syncthreads();

   /* ---- parameters of the settings ladder (oracle/ipm_ref.py: Params.settings; csrc/ipm.hip: solve_impl) */
   const int settings = P.settings < 0 ? 0 : (P.settings > 2 ? 2 : P.settings);
   const double gamma_eff = settings == 0 ? P.gamma : fmin(P.gamma, settings == 1 ? 0.9 : 0.75);
   const int stall_lim = settings == 0 ? 3 : (settings == 1 ? 5 : 8);
   const int nobest_lim = settings == 0 ? 6 : (settings == 1 ? 10 : 15);
   const double sigma_floor = settings == 0 ? 1e-8 : (settings == 1 ? 1e-4 : 1e-2);
   const int maxiter = P.maxiter;
   const double iN1 = 1.0 / N1, inormb1 = 1.0 / (1.0 + normb), inormC1 = 1.0 / (1.0 + normC);
   const double ifeastol = 1.0 / P.feastol, igaptol = 1.0 / P.gaptol, ipabstol = (P.pabstol > 0.0) ? 1.0 / P.pabstol : 0.0;
   bool anybig = false;
   for (int k = 0; k < K; ++k)
      anybig = anybig || (!S1_ALLU && L.n[k] > S1U_MAXN);

   int status = HS_S1_ITERLIM;

   int numwhere = 0;             /* source line of the test that gave up numerically (out[44]; developer trace) */
   int certwait = 0, nstall = 0, sincebest = 0, chol_fail = 0, pre_valid = 0;
   double lastmu = 1e300, alpha_last = 1.0, bestmerit = 1e300, pre_scale = 0.0;
   double mu = 0, pinf = 0, dinf = 0, dabs_ = 0, gap = 0, pobj = 0, dobj = 0;
   bool factors_valid = (warm != 0);
   S1_STAMP(1);

   /* ---- helpers of the iteration (all threads call them; they contain no barrier unless stated) */
   /* out(r, c) for r >= c: sum over the variables that touch the position of coefficient x value */
   auto pass_AT = [&](const S1Blk& B, const double* cv, int toff, auto epi) S1_INL
   {
      const int n = B.n, n2 = n * n;
      for (int e = tro(toff); e < n2; e += S1_NT)
      {
         const int r = s1_div(e, n), c = e - r * n;
         if ( r < c )
            continue;
         double s = 0.0;
         const int t1 = LP(B.poff)[e + 1];
         for (int t = LP(B.poff)[e]; t < t1; ++t)
            s = fma(LP(B.pval)[t], cv[LP(B.pvar)[t]], s);
         epi(r, c, s);
      }
   };
   /* outv[i] = sum_k <A_i^k, V_k> + (Dext^T xv)_i, V_k symmetric at LDS offset offs[k] (16 lanes per variable) */
   /* (wfrom > 0: only the wavefronts wfrom .. take part - the others are busy elsewhere; same sums, other owners) */
   auto pass_A_from = [&](int wfrom, bool ofdX, const double* xv, double* outv, int toff, auto epi) S1_INL
   {
      const int nthr = S1_NT - 64 * wfrom;
      int trot = tid - 64 * wfrom - (toff % nthr);
      if ( trot < 0 ) trot += nthr;
      const int gid = (wfrom == 0 ? tro(toff) : trot) >> 4, l16 = tid & 15;
      for (int i = (wave >= wfrom ? gid : m1); i < m1; i += nthr / 16)
      {
         double s = 0.0;
         for (int k = 0; k < K; ++k)
         {
            const S1Blk& B = sh.blk[k];
            const double* V = sm + (ofdX ? B.odX : B.oX);
            const int t1 = LP(B.voff)[i + 1];
            int t = LP(B.voff)[i] + l16;
            /* (a dense matrix - the constant one, as a rule - is hundreds of entries for its sixteen lanes while the other groups have
             * finished after one: four entries per trip, their positions, values and operands requested before the first of the four
             * multiply-adds - the same chain in the same order, the round trips to LDS side by side instead of one after the other;
             * one block of 32 rows: 25 000 -> cycles of one lane group per pass, round 6) */
            for (; t + 48 < t1; t += 64)
            {
               unsigned pq[4]; double vv_[4], xx_[4];
#pragma unroll
               for (int u = 0; u < 4; ++u)
               {
                  pq[u] = LP(B.vpq)[t + 16 * u];
                  vv_[u] = LP(B.vval)[t + 16 * u];
               }
#pragma unroll
               for (int u = 0; u < 4; ++u)
                  xx_[u] = V[(int) (pq[u] >> 16) * B.p + (int) (pq[u] & 0xffffu)];
#pragma unroll
               for (int u = 0; u < 4; ++u)
                  s = fma(vv_[u], xx_[u], s);
            }
            for (; t < t1; t += 16)
            {
               const unsigned pq = LP(B.vpq)[t];
               const int pp = (int) (pq >> 16), qq = (int) (pq & 0xffffu);
               s = fma(LP(B.vval)[t], V[pp * B.p + qq], s);
            }
         }
         {
            const int t1 = LP(sh.coff)[i + 1];
            for (int t = LP(sh.coff)[i] + l16; t < t1; t += 16)
               s = fma(LP(sh.cval)[t], xv[LP(sh.crow)[t]], s);
         }
         s = s1_sum16(s);
         if ( l16 == 0 )
         {
            outv[i] = s;
            epi(i, s);
         }
      }
   };
   auto pass_A = [&](bool ofdX, const double* xv, double* outv, int toff, auto epi) S1_INL { pass_A_from(0, ofdX, xv, outv, toff, epi); };
   auto no_epi = [](int, double) S1_INL {};
   auto lp_row = [&](int r, const double* cv) S1_INL -> double
   {
      double s = 0.0;
      const int t1 = LP(sh.roff)[r + 1];
      for (int t = LP(sh.roff)[r]; t < t1; ++t)
         s = fma(LP(sh.rval)[t], cv[LP(sh.rcol)[t]], s);
      return s;
   };
   auto red_sum = [&](int slot) S1_INL -> double
   {
      double s = 0.0;
#pragma unroll
      for (int w = 0; w < S1_NW; ++w)
         s += sh.red[w][slot];
      return s;
   };
   /* wavefront 0: x = M^-1 r for one or two right-hand sides (LDS vectors; r1 may be NULL) by substitution with the factor of M;
    * mdinv: this lane's 1 / diagonal entry of the factor.  [The oracle corrects each triangular solve once with the factor itself
    * because the engine's general path solves with explicit inverses of diagonal blocks; a substitution has the small residual by
    * itself.] */
   double mdinv = 1.0;
   auto msolve2 = [&](const double* r0, const double* r1, double* o0, double* o1) S1_INL
   {
      const bool two = (r1 != NULL);
      if ( S1_MBIG && m > 64 )
      {
         if ( two )
            s1_llt_solve2<true>(Mx, m, lane, VEC(V_dg), r0, r1, o0, o1);
         else
            s1_llt_solve2<false>(Mx, m, lane, VEC(V_dg), r0, r1, o0, o1);
         return;
      }
      double x0 = (lane < m) ? r0[lane] : 0.0;
      double x1 = (two && lane < m) ? r1[lane] : 0.0;
      if ( two )
         s1_llt_solve<true>(Lm, m, pm, lane, mdinv, x0, x1);
      else
         s1_llt_solve<false>(Lm, m, pm, lane, mdinv, x0, x1);
      if ( lane < m )
      {
         o0[lane] = x0;
         if ( two ) o1[lane] = x1;
      }
      S1_WSYNC();
   };

   double sigma = 0.0, eta = 1.0;
   /* H (or dX) = sigmu Zinv - X - sym(T1 Zinv) into dX: the product and its transpose (Zinv T1^T) side by side, H from the epilogue;
    * LP part into `lpout` from `rlp` (rd for the right-hand side, dz for the step).  All threads; no barrier. */
   /* (t1: the entry of T1 at a flat index of its block - the array itself, or a formula that yields the same bits, see the corrector) */
   auto dir_matrix_t1 = [&](int wfrom, double sigmu, double etalp, const double* rlp, bool useE, double* lpout, auto t1) S1_INL
   {
      int tb = 0;
      for (int k = 0; k < K; ++k)
      {
         const S1Blk& B = sh.blk[k];
         const int p = B.p;
         const double* Zi = sm + B.oZi; const double* X = sm + B.oX;
         double* dX = sm + B.odX;
         s1_mm2(B.n, wave, lane, wfrom, S1_NW - wfrom, tb,
            [&](int i, int kk) S1_INL { return t1(B, i * p + kk); }, [&](int kk, int j) S1_INL { return Zi[kk * p + j]; },
            [&](int i, int kk) S1_INL { return Zi[i * p + kk]; }, [&](int kk, int j) S1_INL { return t1(B, j * p + kk); },
            [&](int i, int j, double v1, double v2) S1_INL { dX[i * p + j] = sigmu * Zi[i * p + j] - X[i * p + j] - 0.5 * (v1 + v2); });
      }
      for (int r = (wave >= wfrom ? tid - 64 * wfrom : q); r < q; r += S1_NT - 64 * wfrom)                /* (the tiles of the product are on the last wavefronts) */
      {
         const double xv = QV(Q_x)[r], zv = QV(Q_z)[r];
         lpout[r] = sigmu / zv - xv - (etalp * xv * rlp[r] + (useE ? QV(Q_elp)[r] : 0.0)) / zv;
      }
   };
   auto t1_array = [&](const S1Blk& B, int idx) S1_INL { return sm[B.oT1 + idx]; };
   auto dir_matrix_from = [&](int wfrom, double sigmu, double etalp, const double* rlp, bool useE, double* lpout) S1_INL
   {
      dir_matrix_t1(wfrom, sigmu, etalp, rlp, useE, lpout, t1_array);
   };
   auto dir_matrix = [&](double sigmu, double etalp, const double* rlp, bool useE, double* lpout) S1_INL { dir_matrix_from(0, sigmu, etalp, rlp, useE, lpout); };
   /* wavefront 0, after A(H) is known: h, u1 = M^-1 h, dtau, dkappa, dy, coefficient vector [-dtau; dy] */
   /* (solved: u1 = M^-1 h is there already - the predictor's, solved beside the two right-hand sides of the tau elimination) */
   auto finish_dir = [&](double sigmu, double etk, double rg, bool solved) S1_INL
   {
      const double tau = sh.sc[SC_TAU], kappa = sh.sc[SC_KAPPA];
      if ( !solved )
      {
         for (int i = lane; i < m; i += 64)
            VEC(V_h)[i] = VEC(V_AH)[i + 1] - eta * VEC(V_rp)[i];
         S1_WSYNC();
         msolve2(VEC(V_h), NULL, VEC(V_u1), NULL);
      }
      double bu1 = 0.0, wrp = 0.0;
      for (int i = lane; i < m; i += 64)
      {
         bu1 = fma(VEC(V_b)[i], VEC(V_u1)[i], bu1);
         wrp = fma(VEC(V_w)[i], VEC(V_rp)[i], wrp);
      }
      bu1 = s1_wsum(bu1);
      wrp = s1_wsum(wrp);
      const double S0 = red_sum(RS_S0);
      const double BH = red_sum(RS_BH);
      const double it_ = s1_rcp(tau);
      const double den = S0 + kappa * it_ + sh.sc[SC_BUB];
      const double num = -eta * rg + (sigmu - tau * kappa - etk) * it_ - BH - eta * wrp + bu1;
      const double dtau = num * s1_rcp(den);
      const double dkappa = (sigmu - tau * kappa - etk - kappa * dtau) * it_;
      for (int i = lane; i < m; i += 64)
      {
         const double dy = VEC(V_u1)[i] - VEC(V_u2)[i] * dtau;
         VEC(V_dy)[i] = dy;
         VEC(V_cv)[i + 1] = dy;
      }
      if ( lane == 0 )
      {
         VEC(V_cv)[0] = -dtau;
         sh.sc[SC_DTAU] = dtau;
         sh.sc[SC_DKAPPA] = dkappa;
      }
   };
   /* partial sums <B, H> + beta^T hl of this thread's share -> RS_BH */
   auto bh_partials = [&]() S1_INL
   {
      double s = 0.0;
      for (int k = 0; k < K; ++k)
      {
         const S1Blk& B = sh.blk[k];
         const int n = B.n, p = B.p;
         for (int e = tid; e < n * n; e += S1_NT)
         {
            const int r = s1_div(e, n), c = e - r * n;
            s = fma(LP(B.B)[r * p + c], sm[B.odX + r * p + c], s);
         }
      }
      for (int r = tro(320); r < q; r += S1_NT)
         s = fma(QV(Q_beta)[r], QV(Q_hl)[r], s);
      s = s1_wsum(s);
      if ( lane == 0 )
         sh.red[wave][RS_BH] = s;
   };
   /* dZ = A^T([-dtau; dy]) + eta Rd, dz likewise */
   auto make_dZ = [&]() S1_INL
   {
      const double* cv = VEC(V_cv);
      for (int k = 0; k < K; ++k)
      {
         const S1Blk& B = sh.blk[k];
         const int p = B.p;
         pass_AT(B, cv, 0, [&](int r, int c, double s) S1_INL
         {
            const double v = fma(eta, LP(B.Rd)[r * p + c], s);
            sm[B.odZ + r * p + c] = v;
            sm[B.odZ + c * p + r] = v;
         });
      }
      for (int r = tro(128); r < q; r += S1_NT)
         QV(Q_dz)[r] = fma(eta, QV(Q_rd)[r], lp_row(r, cv));
   };
   /* step lengths: (a) T1 = LxI dX, T2 = LzI dZ (+ LP ratio tests; save = true: dX, dZ are copied to E, B first), (b) dX <- T1 LxI^T,
    * dZ <- T2 LzI^T, (c) their smallest eigenvalues, one wavefront each.  Returns the largest step that keeps everything
    * non-negative.  Contains barriers. */
   auto steplen = [&](bool save) S1_INL -> double
   {
      {
         int tb = 0;
         for (int k = 0; k < K; ++k)
         {
            const S1Blk& B = sh.blk[k];
            const int n = B.n, p = B.p;
            const double* Lx = sm + B.oLx; const double* Lz = sm + B.oLz;
            const double* dX = sm + B.odX; const double* dZ = sm + B.odZ;
            double* T1 = sm + B.oT1; double* T2 = sm + B.oT2;
            s1_mm(n, wave, lane, 0, S1_NW, tb,
               [&](int i, int kk) S1_INL { const double v_ = Lx[i * p + kk]; return kk <= i ? v_ : 0.0; }, [&](int kk, int j) S1_INL { return dX[kk * p + j]; },
               [&](int i, int j, double v) S1_INL { T1[i * p + j] = v; });
            s1_mm(n, wave, lane, 0, S1_NW, tb,
               [&](int i, int kk) S1_INL { const double v_ = Lz[i * p + kk]; return kk <= i ? v_ : 0.0; }, [&](int kk, int j) S1_INL { return dZ[kk * p + j]; },
               [&](int i, int j, double v) S1_INL { T2[i * p + j] = v; });
            if ( save )
               for (int e = tro(128); e < n * n; e += S1_NT)
               {
                  const int r = s1_div(e, n), c = e - r * n;
                  LP(B.E)[r * p + c] = dX[r * p + c];
                  LP(B.B)[r * p + c] = dZ[r * p + c];
               }
         }
         double rx = 1e300, rz = 1e300;
         for (int r = tro(256); r < q; r += S1_NT)
         {
            const double dxv = QV(Q_dx)[r], dzv = QV(Q_dz)[r];
            if ( dxv < 0.0 ) rx = fmin(rx, -QV(Q_x)[r] * s1_rcp(dxv));
            if ( dzv < 0.0 ) rz = fmin(rz, -QV(Q_z)[r] * s1_rcp(dzv));
         }
         rx = s1_wmin(rx); rz = s1_wmin(rz);
         if ( lane == 0 )
         {
            sh.red[wave][RS_RATX] = rx;
            sh.red[wave][RS_RATZ] = rz;
         }
      }
      S1_BAR();
      S1_STAMP(16);
      {
         int tb = 0;
         for (int k = 0; k < K; ++k)
         {
            const S1Blk& B = sh.blk[k];
            const int n = B.n, p = B.p;
            const double* Lx = sm + B.oLx; const double* Lz = sm + B.oLz;
            double* dX = sm + B.odX; double* dZ = sm + B.odZ;
            const double* T1 = sm + B.oT1; const double* T2 = sm + B.oT2;
            s1_mm(n, wave, lane, 0, S1_NW, tb,
               [&](int i, int kk) S1_INL { return T1[i * p + kk]; }, [&](int kk, int j) S1_INL { const double v_ = Lx[j * p + kk]; return kk <= j ? v_ : 0.0; },
               [&](int i, int j, double v) S1_INL { dX[i * p + j] = v; });
            s1_mm(n, wave, lane, 0, S1_NW, tb,
               [&](int i, int kk) S1_INL { return T2[i * p + kk]; }, [&](int kk, int j) S1_INL { const double v_ = Lz[j * p + kk]; return kk <= j ? v_ : 0.0; },
               [&](int i, int j, double v) S1_INL { dZ[i * p + j] = v; });
         }
      }
      S1_BAR();
      S1_STAMP(17);
      for (int t = wave; t < 2 * K; t += S1_NW)
      {
         const S1Blk& B = sh.blk[t >> 1];
         double* Wm = sm + ((t & 1) ? B.odZ : B.odX);
         double* tpr = (P.prof_on && t == 0) ? &sh.prof[19] : (double*) NULL;
         double* escr = sm + B.oEig + (t & 1) * (4 * ((B.n + 1) & ~1) + 32);
         const double lm = (S1_ALLU || B.n <= S1U_MAXN) ? s1u_lmin_n(Wm, B.n, B.p, lane, tpr)
            : ((S1_ALL16 || B.n <= 16) ? s1_lmin16(Wm, B.n, B.p, lane, escr, tpr)
               : (B.n <= 32 ? s1_lmin2(Wm, B.n, B.p, lane, escr, tpr) : s1_lmin(Wm, B.n, B.p, lane, escr, tpr)));
         if ( lane == 0 )
            sh.sc[SC_LMIN0 + t] = lm;
      }
      S1_BAR();
      S1_STAMP(18);
      double a = 1e300;
      for (int t = 0; t < 2 * K; ++t)
      {
         const double lm = sh.sc[SC_LMIN0 + t];
         if ( lm != lm )
            a = nan("");
         else if ( lm < 0.0 )
            a = fmin(a, -s1_rcp(lm));
      }
      double rx = 1e300, rz = 1e300;
      for (int w = 0; w < S1_NW; ++w)
      {
         rx = fmin(rx, sh.red[w][RS_RATX]);
         rz = fmin(rz, sh.red[w][RS_RATZ]);
      }
      a = fmin(a, fmin(rx, rz));
      const double tau = sh.sc[SC_TAU], kappa = sh.sc[SC_KAPPA], dtau = sh.sc[SC_DTAU], dkappa = sh.sc[SC_DKAPPA];
      if ( dtau < 0.0 ) a = fmin(a, -tau * s1_rcp(dtau));
      if ( dkappa < 0.0 ) a = fmin(a, -kappa * s1_rcp(dkappa));
      return a;
   };

   if ( tid < m1 )
      VEC(V_cv)[tid] = (tid == 0) ? -sh.sc[SC_TAU] : VEC(V_y)[tid - 1];
   S1_BAR();
   for (it = 0; it <= maxiter; ++it)
   {
      nbar = 0;
      /* ================= residuals (V_cv = [-tau; y] comes from the start or from the update of the last iteration) */
      {
         const double* cv = VEC(V_cv);
         double xz = 0.0;
         for (int k = 0; k < K; ++k)
         {
            const S1Blk& B = sh.blk[k];
            const int p = B.p;
            double r2 = 0.0;
            pass_AT(B, cv, 0, [&](int r, int c, double s) S1_INL
            {
               const double zv = LP(B.Z)[r * p + c];
               const double v = s - zv;
               LP(B.Rd)[r * p + c] = v;
               LP(B.Rd)[c * p + r] = v;
               const double wgt = (r == c) ? 1.0 : 2.0;
               r2 = fma(wgt * v, v, r2);
               xz = fma(wgt * sm[B.oX + r * p + c], zv, xz);
            });
            r2 = s1_wsum(r2);
            if ( lane == 0 )
               sh.red[wave][RS_BLK0 + k] = r2;
         }
         double r2lp = 0.0, rmax = 0.0;
         for (int r = tro(128); r < q; r += S1_NT)
         {
            const double zv = QV(Q_z)[r];
            const double v = lp_row(r, cv) - zv;
            QV(Q_rd)[r] = v;
            r2lp = fma(v, v, r2lp);
            rmax = fmax(rmax, fabs(v));
            xz = fma(QV(Q_x)[r], zv, xz);
         }
         xz = s1_wsum(xz); r2lp = s1_wsum(r2lp); rmax = s1_wmax(rmax);
         if ( lane == 0 )
         {
            sh.red[wave][RS_XZ] = xz;
            sh.red[wave][RS_RD2LP] = r2lp;
            sh.red[wave][RS_RDMAX] = rmax;
         }
         /* A(X, x), and with it rp = b tau - A(X, x), its norm, the norm of A(X, x) and b^T y */
         {
            const double tau0 = sh.sc[SC_TAU];
            double rp2 = 0.0, hp2 = 0.0, dob = 0.0;
            pass_A(false, QV(Q_x), VEC(V_AX), 256, [&](int i, double ax) S1_INL
            {
               if ( i > 0 )
               {
                  const double bv = VEC(V_b)[i - 1];
                  const double rp = bv * tau0 - ax;
                  VEC(V_rp)[i - 1] = rp;
                  rp2 = fma(rp, rp, rp2);
                  hp2 = fma(ax, ax, hp2);
                  dob = fma(bv, VEC(V_y)[i - 1], dob);
               }
            });
            rp2 = s1_wsum(rp2); hp2 = s1_wsum(hp2); dob = s1_wsum(dob);
            if ( lane == 0 )
            {
               sh.red[wave][RS_RP2] = rp2; sh.red[wave][RS_HP2] = hp2; sh.red[wave][RS_DOB] = dob;
            }
         }
      }
      S1_BAR();
      const double tau = sh.sc[SC_TAU], kappa = sh.sc[SC_KAPPA];
      const double hp2sum = red_sum(RS_HP2);
      pobj = s1_uni(VEC(V_AX)[0]);
      dobj = s1_uni(red_sum(RS_DOB));
      const double rg = pobj - dobj - kappa;
      const double itau = s1_rcp(tau);
      mu = s1_uni((red_sum(RS_XZ) + tau * kappa) * iN1);
      double rd2 = red_sum(RS_RD2LP);
      double rdmax = 0.0;
      for (int w = 0; w < S1_NW; ++w)
         rdmax = fmax(rdmax, sh.red[w][RS_RDMAX]);
      for (int k = 0; k < K; ++k)
      {
         const double bk = red_sum(RS_BLK0 + k);
         rd2 += bk;
         rdmax = fmax(rdmax, s1_sqrt(bk));
      }
      const double rpn = s1_sqrt(red_sum(RS_RP2));
      pinf = s1_uni(rpn * itau * inormb1);
      const double pabs = rpn * itau;
      const bool pabsok = P.pabstol <= 0.0 || pabs <= P.pabstol;
      dinf = s1_uni(s1_sqrt(rd2) * itau * inormC1);
      dabs_ = s1_uni(rdmax * itau);
      gap = s1_uni(fabs(dobj - pobj) * itau);
      if ( P.hist != NULL && tid == 0 && it < P.hist_len )
      {
         double* h = P.hist + 16 * it;
         h[0] = it; h[1] = mu; h[2] = pinf; h[3] = dinf; h[4] = gap; h[5] = tau; h[6] = kappa; h[7] = pobj; h[8] = dobj;
      }
      S1_STAMP(2);
      if ( !(fabs(mu) < 1e300) || !(pinf < 1e300) || !(dinf < 1e300) )
      {
         status = HS_S1_NUMERIC; numwhere = __LINE__;
         break;
      }
      /* preoptimal iterate (capture rule of sdpisolver_dsdp.c:323-358) */
      if ( P.preoptgap > 0.0 && !pre_valid && pinf <= P.feastol && dabs_ <= P.feastol
         && gap / (1.0 + 0.5 * fabs(pobj / tau) + 0.5 * fabs(dobj / tau)) < P.preoptgap )
      {
         if ( tid < m ) P.pre_y[tid] = VEC(V_y)[tid];
         for (int r = tid; r < q; r += S1_NT)
            P.pre_x[r] = QV(Q_x)[r];
         for (int k = 0; k < K; ++k)
         {
            const S1Blk& B = sh.blk[k];
            const int n = B.n, p = B.p;
            for (int e = tid; e < n * n; e += S1_NT)
            {
               const int r = s1_div(e, n), c = e - r * n;
               P.Xpre[k][e] = sm[B.oX + r * p + c];
            }
         }
         pre_scale = s1_uni(1.0 / tau);
         pre_valid = 1;
      }
      if ( P.objlimit < 1e20 && pinf <= P.feastol && pobj * itau > P.objlimit + P.gaptol )
      {
         status = HS_S1_OBJLIM;
         break;
      }
      if ( pinf <= P.feastol && pabsok && dabs_ <= P.feastol && gap <= P.gaptol )
      {
         status = HS_S1_OPTIMAL;
         break;
      }
      const bool certzone = (tau < 1e-2 * fmin(1.0, kappa)) || (mu * itau * itau > 1e10);
      if ( certzone )
      {
         /* Farkas certificates: || A^T y - Z || = || Rd + tau A_0 || and || A(X, x) || relative to the objective they certify */
         double h2 = 0.0;
         for (int k = 0; k < K; ++k)
         {
            const S1Blk& B = sh.blk[k];
            const int n = B.n, p = B.p;
            for (int e = tid; e < n * n; e += S1_NT)
            {
               const int r = s1_div(e, n), c = e - r * n;
               if ( r < c )
                  continue;
               const int t = LP(B.poff)[e];
               const double a0 = (t < LP(B.poff)[e + 1] && LP(B.pvar)[t] == 0) ? LP(B.pval)[t] : 0.0;
               const double v = fma(tau, a0, LP(B.Rd)[r * p + c]);
               h2 = fma((r == c ? 1.0 : 2.0) * v, v, h2);
            }
         }
         for (int r = tid; r < q; r += S1_NT)
         {
            const int t = LP(sh.roff)[r];
            const double c0 = (t < LP(sh.roff)[r + 1] && LP(sh.rcol)[t] == 0) ? LP(sh.rval)[t] : 0.0;
            const double v = fma(tau, c0, QV(Q_rd)[r]);
            h2 = fma(v, v, h2);
         }
         h2 = s1_wsum(h2);
         if ( lane == 0 )
            sh.red[wave][RS_HD2] = h2;
         S1_BAR();
         const double hd = s1_sqrt(red_sum(RS_HD2));
         const double hp = s1_sqrt(hp2sum);
         const double big = fmax(fabs(dobj), fabs(pobj));
         const bool cand_dunb = dobj < -1e-3 * big;
         const bool cand_dinf = pobj > 1e-3 * big;
         const bool ok_dunb = cand_dunb && hd <= P.infeastol * (-dobj);
         const bool ok_dinf = cand_dinf && hp <= P.infeastol * pobj;
         if ( (ok_dunb || ok_dinf) && (ok_dunb || !cand_dunb || certwait >= 5) && (ok_dinf || !cand_dinf || certwait >= 5) )
         {
            status = (ok_dunb && ok_dinf) ? HS_S1_PDINF : (ok_dunb ? HS_S1_DUNB : HS_S1_DINF);
            break;
         }
         if ( ok_dunb || ok_dinf )
            ++certwait;
      }
      if ( it == maxiter )
         break;
      if ( mu > 0.9 * lastmu && alpha_last < 1e-2 )
      {
         if ( ++nstall >= stall_lim )
         {
            status = HS_S1_NUMERIC; numwhere = __LINE__;
            break;
         }
      }
      else
         nstall = 0;
      lastmu = mu;
      if ( !certzone )
      {
         double merit = fmax(fmax(pinf * ifeastol, dabs_ * ifeastol), gap * igaptol);
         if ( P.pabstol > 0.0 )
            merit = fmax(merit, pabs * ipabstol);
         if ( merit < 0.9 * bestmerit )
         {
            bestmerit = s1_uni(merit);
            sincebest = 0;
         }
         else if ( ++sincebest >= nobest_lim )
         {
            status = HS_S1_NUMERIC; numwhere = __LINE__;
            break;
         }
      }
      if ( P.timelimit > 0.0 )
      {
         /* s_memrealtime: 100 MHz */
         const double el = (double) (wall_clock64() - w_start) * 1e-8 + P.elapsed0;
         int over = (el > P.timelimit) ? 1 : 0;
         over = __builtin_amdgcn_readfirstlane(over);
         if ( tid == 0 )
            sh.fl[5] = over;
         S1_BAR();
         if ( sh.fl[5] )
         {
            status = HS_S1_TIMELIM;
            break;
         }
      }

      /* ================= factorizations */
      if ( !factors_valid )
      {
         const int ff = trial_factor(0.0, false);
         if ( ff != 0 )
         {
            status = HS_S1_NUMERIC; numwhere = __LINE__;
            break;
         }
      }
      /* inverse factors in place for the blocks above S1U_MAXN rows (one wavefront per matrix; the smaller ones were inverted with
       * their factorization); then Zinv = LzI^T LzI */
      if ( anybig )
      {
         for (int t = wave; t < 2 * K; t += S1_NW)
         {
            const S1Blk& B = sh.blk[t >> 1];
            double* Lp = sm + ((t & 1) ? B.oLz : B.oLx);
            if ( !S1_ALLU && B.n > S1U_MAXN )
               s1_trinv(Lp, Lp, B.n, B.p, lane);
         }
         S1_BAR();
      }
      S1_STAMP(3);
      {
         int tb = 0;
         for (int k = 0; k < K; ++k)
         {
            const S1Blk& B = sh.blk[k];
            const int p = B.p;
            const double* Lz = sm + B.oLz;
            double* Zi = sm + B.oZi;
            s1_mm(B.n, wave, lane, 0, S1_NW, tb,
               [&](int i, int kk) S1_INL { const double v_ = Lz[kk * p + i]; return kk >= i ? v_ : 0.0; }, [&](int kk, int j) S1_INL { const double v_ = Lz[kk * p + j]; return kk >= j ? v_ : 0.0; },
               [&](int i, int j, double v) S1_INL { Zi[i * p + j] = v; });
         }
      }
      S1_BAR();
      S1_STAMP(4);

      /* ================= Schur complement from the nonzeros: U_j = X A_j Zinv (G of them side by side), Mx[i][j] += <A_i, U_j> */
      /* Schur complement from the nonzeros, in the association of the dense formula: T_j = A_j Zinv (row p of T_j: sum over the
       * entries of row p of A_j), U_j = X T_j (sum over the non-empty rows), M_ij = sum over the entries (a, b) of A_i of
       * A_i[a][b] U_j[b][a] - zeros skipped, nothing else changed.  [The pair formula of csrc/sparse.hip, a b (X_qr Zinv_sp + ...),
       * and a sum of rank-one terms a X[:, p] Zinv[q, :] are the same numbers in exact arithmetic but leave the linearised primal
       * equation violated 300 times more on nodes without an optimum (tau -> 0, cond(Z) 1e5: 2e-7 against 5e-10, measured on
       * example_TT's infeasible nodes with tests/devtools/solve1_node.py): Zinv is almost of rank one there, the differences
       * Zinv[q][c] - Zinv[q'][c] a constraint matrix asks for must be formed BEFORE they meet X, as the products X dZ Zinv of the
       * direction form them.] */
      for (int k = 0; k < K; ++k)
      {
         const S1Blk& B = sh.blk[k];
         const int n = B.n, p = B.p, n2 = n * n;
         const double* X = sm + B.oX;
         const double* Zi = sm + B.oZi;
         /* heavy variables j: T_j and U_j as whole matrices in the scratch region.  While the dense copy of the caller's matrices that
          * the setup staged in LDS is still there (sh.fl[31] >= 0): T_j = A_j Zinv and U_j = X T_j as products on the matrix cores,
          * then <A_i, U_j^T> for every i, sixteen lanes per variable (a dense constant matrix - the usual case of a cost matrix -
          * of 10 rows: 41 000 -> 7 200 cycles; before, from the lists: a thread per column of T_j walking all entries). */
         const int nh = B.nh;
         if ( nh > 0 )
         {
            /* (the dense matrices: the staged copy in LDS, else the caller's array in global memory - a tile asks for a few
             * fragments only) */
            const double* Ast = P.A[k];
            if ( sh.fl[31] >= 0 )
            {
               Ast = sm + sh.fl[31] + q * m1;
               for (int kb = 0; kb < k; ++kb)
                  Ast += m1 * sh.blk[kb].n * sh.blk[kb].n;
            }
            double* T = sm + L.oR;
            double* U = T + B.np;
            for (int h0 = 0; h0 < nh; ++h0)
            {
               const int j = (int) LP(B.hv)[h0];
               const double* Aj = Ast + j * n2;
               {
                  int tb = 0;
                  s1_mm(n, wave, lane, 0, S1_NW, tb,
                     [&](int i, int kk) S1_INL { return Aj[i * n + kk]; }, [&](int kk, int c) S1_INL { return Zi[kk * p + c]; },
                     [&](int i, int c, double v) S1_INL { T[i * p + c] = v; });
               }
               S1_BAR();
               {
                  int tb = 0;
                  s1_mm(n, wave, lane, 0, S1_NW, tb,
                     [&](int i, int kk) S1_INL { return X[i * p + kk]; }, [&](int kk, int c) S1_INL { return T[kk * p + c]; },
                     [&](int i, int c, double v) S1_INL { U[i * p + c] = v; });
               }
               S1_BAR();
               {
                  const int gid = tid >> 4, l16 = tid & 15;
                  for (int i = gid; i < m1; i += S1_NT / 16)
                  {
                     const int t0 = LP(B.voff)[i], t1 = LP(B.voff)[i + 1];
                     if ( t1 == t0 || (t1 - t0 > S1_LIGHT_MAX && i < j) )
                        continue;
                     double s0 = 0.0;
                     int t = t0 + l16;
                     /* (four entries per trip for a dense A_i, as in pass_A: same chain, the LDS round trips side by side) */
                     for (; t + 48 < t1; t += 64)
                     {
                        unsigned pq[4]; double vv_[4], uu_[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                        {
                           pq[u] = LP(B.vpq)[t + 16 * u];
                           vv_[u] = LP(B.vval)[t + 16 * u];
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                           uu_[u] = U[(int) (pq[u] & 0xffffu) * p + (int) (pq[u] >> 16)];
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                           s0 = fma(vv_[u], uu_[u], s0);
                     }
                     for (; t < t1; t += 16)
                     {
                        const unsigned pq = LP(B.vpq)[t];
                        const int pp = (int) (pq >> 16), qq = (int) (pq & 0xffffu);
                        s0 = fma(LP(B.vval)[t], U[qq * p + pp], s0);
                     }
                     s0 = s1_sum16(s0);
                     if ( l16 == 0 )
                        Mx[(i >= j) ? MROW(i) + j : MROW(j) + i] += s0;
                  }
               }
               S1_BAR();
            }
         }
      }
      /* light matrices: (a) T_j = A_j Zinv, only the non-empty rows, for all of them at once ... */
      for (int k = 0; k < K; ++k)
      {
         const S1Blk& B = sh.blk[k];
         const int n = B.n, p = B.p;
         const double* Zi = sm + B.oZi;
         const int nrs = B.nrs;
         for (int idx = tid; idx < nrs * n; idx += S1_NT)
         {
            const int sl = s1_div(idx, n), c = idx - sl * n;
            const int packed = LP(B.lre)[sl];
            const int e0 = packed >> 6, len = packed & 63;
            double tacc = 0.0;
            for (int e = e0; e < e0 + len; ++e)
               tacc = fma(LP(B.vval)[e], Zi[(int) (LP(B.vpq)[e] & 0xffffu) * p + c], tacc);
            LP(B.Tc)[idx] = tacc;
         }
      }
      S1_BAR();
      /* ... (b) a thread per pair i >= j: M_ij = sum over the entries (a, b) of A_i of A_i[a][b] (X T_j)[b][a], the product with X
       * over the non-empty rows of T_j only */
      for (int k = 0; k < K; ++k)
      {
         const S1Blk& B = sh.blk[k];
         const int n = B.n, p = B.p;
         const double* X = sm + B.oX;
         const int nl = B.nl;
         const int npair = (nl * (nl + 1)) >> 1;
         for (int t = tid; t < npair; t += S1_NT)
         {
            /* pair number t -> (ia, ib), ib <= ia: ia = floor((sqrt(8 t + 1) - 1) / 2), corrected for the rounding of the root */
            int ia = (int) ((sqrtf(8.0f * (float) t + 1.0f) - 1.0f) * 0.5f);
            if ( ((ia + 1) * (ia + 2)) >> 1 <= t ) ++ia;
            if ( (ia * (ia + 1)) >> 1 > t ) --ia;
            const int ib = t - ((ia * (ia + 1)) >> 1);
            const int i = LP(B.lv)[ia], j = LP(B.lv)[ib];
            const int e0 = LP(B.voff)[i], e1 = LP(B.voff)[i + 1];
            const int s0i = LP(B.lro)[ib], ns = LP(B.lro)[ib + 1] - s0i;
            double acc = 0.0;
            if ( ns <= 4 )
            {
               /* the usual case, at most four non-empty rows in A_j: their row numbers and the rows of T_j fixed before the loop
                * over the entries of A_i, so that the eight loads of an entry are independent of each other (slots past ns read
                * slot 0 with a zero factor: u + 0 x = u exactly); same operations in the same order as the general loop */
               const int r0 = (int) LP(B.lrp)[s0i];
               const int r1 = (ns > 1) ? (int) LP(B.lrp)[s0i + 1] : r0;
               const int r2 = (ns > 2) ? (int) LP(B.lrp)[s0i + 2] : r0;
               const int r3 = (ns > 3) ? (int) LP(B.lrp)[s0i + 3] : r0;
               auto T0 = LP(B.Tc) + s0i * n;
               auto T1 = (ns > 1) ? T0 + n : T0;
               auto T2 = (ns > 2) ? T0 + 2 * n : T0;
               auto T3 = (ns > 3) ? T0 + 3 * n : T0;
               /* (position and value of the next entry are requested before this one's eight loads: one round trip per entry) */
               unsigned abn = (e0 < e1) ? LP(B.vpq)[e0] : 0u;
               double vn = (e0 < e1) ? LP(B.vval)[e0] : 0.0;
               for (int e = e0; e < e1; ++e)
               {
                  const unsigned ab = abn;
                  const double ve = vn;
                  const int en = (e + 1 < e1) ? e + 1 : e;
                  abn = LP(B.vpq)[en];
                  vn = LP(B.vval)[en];
                  const int aa_ = (int) (ab >> 16), bb_ = (int) (ab & 0xffffu);
                  const double* xr = X + bb_ * p;
                  const double x0 = xr[r0], x1 = (ns > 1) ? xr[r1] : 0.0, x2 = (ns > 2) ? xr[r2] : 0.0, x3 = (ns > 3) ? xr[r3] : 0.0;
                  const double t0 = T0[aa_], t1 = T1[aa_], t2 = T2[aa_], t3 = T3[aa_];
                  double u = fma(x0, t0, 0.0);
                  u = fma(x1, t1, u);
                  u = fma(x2, t2, u);
                  u = fma(x3, t3, u);
                  acc = fma(ve, u, acc);
               }
            }
            else if ( ns <= S1_LIGHT_MAX )
            {
               /* more than four non-empty rows in A_j (three off-diagonal entries: up to six; a matrix counts as light up to
                * S1_LIGHT_MAX entries): the same scheme with 8, 12, 16 or 24 slots - the row numbers leave the loop over the entries of
                * A_i, an entry's loads are independent of each other, a slot past ns multiplies a finite number by zero (u + 0 x = u
                * exactly).  The general loop below asked for the row number of every slot again for every entry, a chain of two round
                * trips per slot (700 ns each once the lists live in the workspace): example_TT's tree 885 -> 957 node solves/s, two
                * blocks of 30 rows -4.8 % (round 6) */
               auto slots = [&](auto NStag) S1_INL
               {
                  constexpr int NSL = decltype(NStag)::value;
                  int rr_[NSL];
                  decltype(LP(B.Tc)) TT_[NSL];
                  const auto Tb = LP(B.Tc) + s0i * n;
#pragma unroll
                  for (int v_ = 0; v_ < NSL; ++v_)
                  {
                     const int sv = (v_ < ns) ? v_ : 0;
                     rr_[v_] = (int) LP(B.lrp)[s0i + sv];
                     TT_[v_] = Tb + sv * n;
                  }
                  unsigned abn = (e0 < e1) ? LP(B.vpq)[e0] : 0u;
                  double vn = (e0 < e1) ? LP(B.vval)[e0] : 0.0;
                  for (int e = e0; e < e1; ++e)
                  {
                     const unsigned ab = abn;
                     const double ve = vn;
                     const int en = (e + 1 < e1) ? e + 1 : e;
                     abn = LP(B.vpq)[en];
                     vn = LP(B.vval)[en];
                     const int aa_ = (int) (ab >> 16), bb_ = (int) (ab & 0xffffu);
                     const double* xr = X + bb_ * p;
                     double xl[NSL], tl[NSL];
#pragma unroll
                     for (int v_ = 0; v_ < NSL; ++v_)
                     {
                        xl[v_] = xr[rr_[v_]];
                        tl[v_] = TT_[v_][aa_];
                     }
                     double u = 0.0;
#pragma unroll
                     for (int v_ = 0; v_ < NSL; ++v_)
                        u = fma((v_ < ns) ? xl[v_] : 0.0, tl[v_], u);
                     acc = fma(ve, u, acc);
                  }
               };
               if ( ns <= 8 ) slots(std::integral_constant<int, 8>());
               else if ( ns <= 12 ) slots(std::integral_constant<int, 12>());
               else if ( ns <= 16 ) slots(std::integral_constant<int, 16>());
               else slots(std::integral_constant<int, S1_LIGHT_MAX>());
            }
            else
               for (int e = e0; e < e1; ++e)
               {
                  const unsigned ab = LP(B.vpq)[e];
                  const int aa_ = (int) (ab >> 16), bb_ = (int) (ab & 0xffffu);
                  double u = 0.0;
                  for (int sl = s0i; sl < s0i + ns; ++sl)
                     u = fma(X[bb_ * p + (int) LP(B.lrp)[sl]], LP(B.Tc)[sl * n + aa_], u);
                  acc = fma(LP(B.vval)[e], u, acc);
               }
            Mx[MROW(i) + j] += acc;
         }
         /* (the lists of light variables differ from block to block - a dense constant matrix is heavy in a block of 8 rows and
          * light in one of 2 -, so the same entry of Mx is another thread's in the next block: one block after the other) */
         if ( k + 1 < K )
            S1_BAR();
      }
      S1_BAR();
      S1_STAMP(5);

      /* ================= factorization of M (wavefront 0) beside the predictor's right-hand side (the others), then the THREE solves
       * with the factor on three wavefronts.
       * Round 6 (m <= 64).  A wave-serial section of this kernel costs its instruction count times eight cycles whatever the
       * dependencies (profiles/r06_solve1_diag_block_attempt.txt), so what shortens the iteration is giving the other wavefronts the
       * work that does not need M while wavefront 0 factors it, and giving every right-hand side a wavefront of its own:
       *   wavefront 0:  g = column 0 of Mx | panel 0 | panel 1 | panel 2 | panels 3 .. | 1 / diagonal -> V_dg || w = M^-1 g, wt
       *   the others:   T1 = X Rd          | H = -X - sym(T1 Zinv), hl | A(H) | h = A(H) - rp     || ub = M^-1 b, b^T ub (wavefront 1)
       *                                                                                              || u1 = M^-1 h (wavefront 2)
       * (`|`: a workgroup barrier - wavefront 0 passes them between its panels through the hook of s1_cholp, where it arrives later
       * than the others; `||`: the barrier behind the factorization.)  Before: the factorization and a solve with two right-hand
       * sides on wavefront 0 (35 000 cycles at example_TT) while the others formed T1 and waited, H and A(H) as phases of their own
       * behind it, and the predictor's solve alone on wavefront 0 in finish_dir (7 500).  Every number is computed by the same
       * operations in the same order as before - only by other lanes -: same bits.
       * (64 < m, the other instance: the factorization is the work of all wavefronts - s1_cholp2 -, the product X Rd comes first.) */
      const bool coop = S1_MBIG && m > 64;
      const bool shadow = !S1_MBIG;          /* (the overlapped form; the other instance runs the same work in phases of its own) */
      double mdiag = 1.0;
      int nforced = 0;
      long long tq0 = 0;
      sigma = 0.0; eta = 1.0;
      if ( shadow )
      {
         const int NSY = 3;
         if ( wave == 0 )
         {
            for (int i = lane; i < m; i += 64)
               VEC(V_g)[i] = Mx[MROW(i + 1)];
            if ( lane == 0 )
               sh.fl[7] = 0;
            if ( P.prof_on ) tq0 = clock64();
            const double dg0 = (lane < m) ? Lm[lane * pm + lane] : 1.0;
            int nsy = 0;
            (void) s1_cholp(Lm, m, pm, lane, true, dg0, P.pivot_rule, false, mdiag, nforced, zsrc,
               [&](int) S1_INL { if ( nsy < NSY ) { S1_BAR(); ++nsy; } });
            for (; nsy < NSY; ++nsy)
               S1_BAR();
            if ( lane == 0 )
               sh.fl[6] = nforced;
            mdinv = s1_rcp(mdiag);
            if ( lane < m )
               VEC(V_dg)[lane] = mdinv;
            if ( P.prof_on && lane == 0 ) { const long long tq1 = clock64(); sh.prof[20] += (double) (tq1 - tq0); tq0 = tq1; }
         }
         else
         {
            {
               int tb = 0;
               for (int k = 0; k < K; ++k)
               {
                  const S1Blk& B = sh.blk[k];
                  const int p = B.p;
                  const double* X = sm + B.oX;
                  auto Rd = LP(B.Rd);
                  double* T1 = sm + B.oT1;
                  auto XR = LP(B.XR);
                  s1_mm(B.n, wave, lane, 1, S1_NW - 1, tb,
                     [&](int i, int kk) S1_INL { return X[i * p + kk]; }, [&](int kk, int j) S1_INL { return Rd[kk * p + j]; },
                     [&](int i, int j, double v) S1_INL { T1[i * p + j] = v; if ( AL ) XR[i * p + j] = v; });
               }
            }
            S1_BAR();
            dir_matrix_from(1, 0.0, 1.0, QV(Q_rd), false, QV(Q_hl));
            S1_BAR();
            pass_A_from(1, true, QV(Q_hl), VEC(V_AH), 128, no_epi);
            S1_BAR();
            if ( wave == 1 )
               for (int i = lane; i < m; i += 64)
                  VEC(V_h)[i] = VEC(V_AH)[i + 1] - eta * VEC(V_rp)[i];
         }
         S1_BAR();
         if ( wave == 0 )
         {
            msolve2(VEC(V_g), NULL, VEC(V_w), NULL);
            for (int i = lane; i < m; i += 64)
               VEC(V_wt)[i + 1] = -VEC(V_w)[i];
            if ( lane == 0 )
               VEC(V_wt)[0] = 1.0;
            if ( P.prof_on && lane == 0 ) { const long long tq1 = clock64(); sh.prof[21] += (double) (tq1 - tq0); }
         }
         else if ( wave == 1 )
         {
            mdinv = (lane < m) ? VEC(V_dg)[lane] : 1.0;
            msolve2(VEC(V_b), NULL, VEC(V_ub), NULL);
            double bubp = 0.0;
            for (int i = lane; i < m; i += 64)
               bubp = fma(VEC(V_b)[i], VEC(V_ub)[i], bubp);
            const double bub = s1_wsum(bubp);
            if ( lane == 0 )
               sh.sc[SC_BUB] = bub;
         }
         else if ( wave == 2 )
         {
            mdinv = (lane < m) ? VEC(V_dg)[lane] : 1.0;
            msolve2(VEC(V_h), NULL, VEC(V_u1), NULL);
         }
      }
      else
      {
         /* 64 < m (the instance with two rows per lane): the factorization is the work of ALL wavefronts (s1_cholp2), so nothing runs
          * beside it - but the predictor's right-hand side still does not need M: it is formed first (three phases that were behind
          * the solves before: same cost), and its solve joins the two of the tau elimination on a wavefront of its own behind the
          * factorization - three substitution passes side by side where there were a pass with two right-hand sides and, later, one
          * with one, both on wavefront 0 (example_MkP's root, m = 105: 0.221 -> 0.20 ms per iteration). */
         {
            int tb = 0;
            for (int k = 0; k < K; ++k)
            {
               const S1Blk& B = sh.blk[k];
               const int p = B.p;
               const double* X = sm + B.oX;
               auto Rd = LP(B.Rd);
               double* T1 = sm + B.oT1;
               auto XR = LP(B.XR);
               s1_mm(B.n, wave, lane, 0, S1_NW, tb,
                  [&](int i, int kk) S1_INL { return X[i * p + kk]; }, [&](int kk, int j) S1_INL { return Rd[kk * p + j]; },
                  [&](int i, int j, double v) S1_INL { T1[i * p + j] = v; if ( AL ) XR[i * p + j] = v; });
            }
         }
         S1_BAR();
         dir_matrix(0.0, 1.0, QV(Q_rd), false, QV(Q_hl));
         S1_BAR();
         pass_A(true, QV(Q_hl), VEC(V_AH), 128, no_epi);
         S1_BAR();
         if ( wave == 0 )
         {
            /* (the factor overwrites M where it stands; its first column, g, is taken out first) */
            for (int i = lane; i < m; i += 64)
            {
               VEC(V_g)[i] = Mx[MROW(i + 1)];
               if ( coop )
                  VEC(V_dg)[i] = Mx[MROW(i + 1) + i + 1];
               VEC(V_h)[i] = VEC(V_AH)[i + 1] - eta * VEC(V_rp)[i];
            }
            if ( lane == 0 )
               sh.fl[7] = 0;
            if ( P.prof_on ) tq0 = clock64();
         }
         S1_BAR();
         if ( coop )
            nforced = s1_cholp2(Mx, m, wave, lane, VEC(V_dg), P.pivot_rule, VEC(V_dg));
         if ( wave == 0 )
         {
            if ( !coop )
            {
               const double dg0 = (lane < m) ? Lm[lane * pm + lane] : 1.0;
               (void) s1_cholp(Lm, m, pm, lane, true, dg0, P.pivot_rule, false, mdiag, nforced, zsrc);
               mdinv = s1_rcp(mdiag);
               if ( lane < m )
                  VEC(V_dg)[lane] = mdinv;
            }
            if ( lane == 0 )
               sh.fl[6] = nforced;
            if ( P.prof_on && lane == 0 ) { const long long tq1 = clock64(); sh.prof[20] += (double) (tq1 - tq0); tq0 = tq1; }
         }
         S1_BAR();
         if ( wave == 0 )
         {
            msolve2(VEC(V_g), NULL, VEC(V_w), NULL);
            for (int i = lane; i < m; i += 64)
               VEC(V_wt)[i + 1] = -VEC(V_w)[i];
            if ( lane == 0 )
               VEC(V_wt)[0] = 1.0;
            if ( P.prof_on && lane == 0 ) { const long long tq1 = clock64(); sh.prof[21] += (double) (tq1 - tq0); }
         }
         else if ( wave == 1 )
         {
            if ( !coop )
               mdinv = (lane < m) ? VEC(V_dg)[lane] : 1.0;
            msolve2(VEC(V_b), NULL, VEC(V_ub), NULL);
            double bubp = 0.0;
            for (int i = lane; i < m; i += 64)
               bubp = fma(VEC(V_b)[i], VEC(V_ub)[i], bubp);
            const double bub = s1_wsum(bubp);
            if ( lane == 0 )
               sh.sc[SC_BUB] = bub;
         }
         else if ( wave == 2 )
         {
            if ( !coop )
               mdinv = (lane < m) ? VEC(V_dg)[lane] : 1.0;
            msolve2(VEC(V_h), NULL, VEC(V_u1), NULL);
         }
      }
      S1_BAR();
      S1_STAMP(6);
      /* B_k = A_0 - sum w_i A_i, beta = c - D w (H of the predictor, -X - sym(X Rd Zinv) -> dX, hl, is there already); u2 = ub - w and
       * the test of its entries */
      {
         if ( tid < m )
         {
            const double w = VEC(V_w)[tid], ub = VEC(V_ub)[tid];
            VEC(V_u2)[tid] = ub - w;
            if ( !(fabs(ub - w) < 1e300) )
               sh.fl[7] = 1;
         }
         const double* wt = VEC(V_wt);
         for (int k = 0; k < K; ++k)
         {
            const S1Blk& B = sh.blk[k];
            const int p = B.p;
            pass_AT(B, wt, 256, [&](int r, int c, double s) S1_INL
            {
               LP(B.B)[r * p + c] = s;
               LP(B.B)[c * p + r] = s;
            });
         }
         for (int r = tro(128); r < q; r += S1_NT)
            QV(Q_beta)[r] = lp_row(r, wt);
      }
      S1_BAR();
      if ( sh.fl[7] )
      {
         status = HS_S1_NUMERIC; numwhere = __LINE__;
         break;
      }
      /* A(H), <B, H>; T1 = X B */
      {
         bh_partials();
         int tb = 0;
         for (int k = 0; k < K; ++k)
         {
            const S1Blk& B = sh.blk[k];
            const int p = B.p;
            const double* X = sm + B.oX; auto Bm = LP(B.B);
            double* T1 = sm + B.oT1;
            s1_mm(B.n, wave, lane, 0, S1_NW, tb,
               [&](int i, int kk) S1_INL { return X[i * p + kk]; }, [&](int kk, int j) S1_INL { return Bm[kk * p + j]; },
               [&](int i, int j, double v) S1_INL { T1[i * p + j] = v; });
         }
      }
      S1_BAR();
      /* S0 = sum <B, X B Zinv> + sum (x / z) beta^2 (factored form: a sum of non-negative terms): the entries of X B Zinv are summed
       * up as the product's epilogue delivers them */
      {
         double s = 0.0;
         int tb = 0;
         for (int k = 0; k < K; ++k)
         {
            const S1Blk& B = sh.blk[k];
            const int p = B.p;
            const double* T1 = sm + B.oT1; const double* Zi = sm + B.oZi; auto Bm = LP(B.B);
            s1_mm(B.n, wave, lane, 0, S1_NW, tb,
               [&](int i, int kk) S1_INL { return T1[i * p + kk]; }, [&](int kk, int j) S1_INL { return Zi[kk * p + j]; },
               [&](int i, int j, double v) S1_INL { s = fma(Bm[i * p + j], v, s); });
         }
         for (int r = tid; r < q; r += S1_NT)
         {
            const double be = QV(Q_beta)[r];
            s = fma(QV(Q_sx)[r] * be, be, s);
         }
         s = s1_wsum(s);
         if ( lane == 0 )
            sh.red[wave][RS_S0] = s;
      }
      S1_BAR();
      if ( wave == 0 )
         finish_dir(0.0, 0.0, rg, true);
      S1_BAR();
      S1_STAMP(7);
      const double dta = sh.sc[SC_DTAU], dka = sh.sc[SC_DKAPPA];
      if ( !(fabs(dta) < 1e300) || !(fabs(dka) < 1e300) )
      {
         status = HS_S1_NUMERIC; numwhere = __LINE__;
         break;
      }
      make_dZ();
      S1_BAR();
      S1_STAMP(12);
      {
         int tb = 0;
         for (int k = 0; k < K; ++k)
         {
            const S1Blk& B = sh.blk[k];
            const int p = B.p;
            const double* X = sm + B.oX; const double* dZ = sm + B.odZ;
            double* T1 = sm + B.oT1;
            s1_mm(B.n, wave, lane, 0, S1_NW, tb,
               [&](int i, int kk) S1_INL { return X[i * p + kk]; }, [&](int kk, int j) S1_INL { return dZ[kk * p + j]; },
               [&](int i, int j, double v) S1_INL { T1[i * p + j] = v; });
         }
      }
      S1_BAR();
      S1_STAMP(13);
      dir_matrix(0.0, 1.0, QV(Q_dz), false, QV(Q_dx));          /* dXa, dxa */
      S1_BAR();
      S1_STAMP(15);
      /* second-order terms E = dXa dZa, elp = dxa dza; then the predictor's step length */
      {
         int tb = 0;
         for (int k = 0; k < K; ++k)
         {
            const S1Blk& B = sh.blk[k];
            const int p = B.p;
            const double* dX = sm + B.odX; const double* dZ = sm + B.odZ;
            auto E = LP(B.E);
            s1_mm(B.n, wave, lane, 0, S1_NW, tb,
               [&](int i, int kk) S1_INL { return dX[i * p + kk]; }, [&](int kk, int j) S1_INL { return dZ[kk * p + j]; },
               [&](int i, int j, double v) S1_INL { E[i * p + j] = v; });
         }
         for (int r = tro(64); r < q; r += S1_NT)
            QV(Q_elp)[r] = QV(Q_dx)[r] * QV(Q_dz)[r];
      }
      const double aa = fmin(1.0, steplen(false));
      S1_STAMP(8);
      if ( !(aa == aa) )
      {
         status = HS_S1_NUMERIC; numwhere = __LINE__;
         break;
      }
      sigma = (1.0 - aa) * (1.0 - aa) * (1.0 - aa);
      sigma = fmin(1.0, fmax(sigma_floor, sigma));
      eta = 1.0 - sigma;
      const double sigmu = sigma * mu;
      const double etk = dta * dka;

      /* ================= corrector */
      /* T1 = eta X Rd + E.  X Rd is the product the predictor's right-hand side was made of: with every cold matrix in LDS its entries
       * were kept (XR) and an entry of T1 is formed where the product below reads it - fma(eta, XR, E), the operation the epilogue of
       * the product X Rd applied to the same two numbers: same bits, a product phase and its barrier less (round 6). */
      if ( AL )
      {
         const double et = eta;
         dir_matrix_t1(0, sigmu, eta, QV(Q_rd), true, QV(Q_hl),
            [&](const S1Blk& B, int idx) S1_INL { return fma(et, LP(B.XR)[idx], LP(B.E)[idx]); });
      }
      else
      {
         {
            int tb = 0;
            for (int k = 0; k < K; ++k)
            {
               const S1Blk& B = sh.blk[k];
               const int p = B.p;
               const double* X = sm + B.oX; auto Rd = LP(B.Rd); auto E = LP(B.E);
               double* T1 = sm + B.oT1;
               const double et = eta;
               s1_mm(B.n, wave, lane, 0, S1_NW, tb,
                  [&](int i, int kk) S1_INL { return X[i * p + kk]; }, [&](int kk, int j) S1_INL { return Rd[kk * p + j]; },
                  [&](int i, int j, double v) S1_INL { T1[i * p + j] = fma(et, v, E[i * p + j]); });
            }
         }
         S1_BAR();
         dir_matrix(sigmu, eta, QV(Q_rd), true, QV(Q_hl));
      }
      S1_BAR();
      pass_A(true, QV(Q_hl), VEC(V_AH), 128, no_epi);
      bh_partials();
      S1_BAR();
      if ( wave == 0 )
         finish_dir(sigmu, etk, rg, false);
      S1_BAR();
      S1_STAMP(9);
      const double dt = sh.sc[SC_DTAU], dk = sh.sc[SC_DKAPPA];
      if ( !(fabs(dt) < 1e300) || !(fabs(dk) < 1e300) )
      {
         status = HS_S1_NUMERIC; numwhere = __LINE__;
         break;
      }
      make_dZ();
      S1_BAR();
      {
         int tb = 0;
         for (int k = 0; k < K; ++k)
         {
            const S1Blk& B = sh.blk[k];
            const int p = B.p;
            const double* X = sm + B.oX; const double* dZ = sm + B.odZ; auto E = LP(B.E);
            double* T1 = sm + B.oT1;
            s1_mm(B.n, wave, lane, 0, S1_NW, tb,
               [&](int i, int kk) S1_INL { return X[i * p + kk]; }, [&](int kk, int j) S1_INL { return dZ[kk * p + j]; },
               [&](int i, int j, double v) S1_INL { T1[i * p + j] = v + E[i * p + j]; });
         }
      }
      S1_BAR();
      dir_matrix(sigmu, 1.0, QV(Q_dz), true, QV(Q_dx));
      S1_BAR();
      if ( P.hist != NULL )
      {
         /* diagnostic: how well the direction satisfies the linearised primal equation A(dX, dx) = eta rp + b dtau */
         pass_A(true, QV(Q_dx), VEC(V_t1), 0, no_epi);
         S1_BAR();
         if ( wave == 0 )
         {
            double e2p = 0.0, dy2p = 0.0;
            for (int i = lane; i < m; i += 64)
            {
               const double e = VEC(V_t1)[i + 1] - eta * VEC(V_rp)[i] - VEC(V_b)[i] * dt;
               e2p = fma(e, e, e2p);
               dy2p = fma(VEC(V_dy)[i], VEC(V_dy)[i], dy2p);
            }
            const double e2 = s1_wsum(e2p);
            const double dy2 = s1_wsum(dy2p);
            /* (the residual of the solve itself cannot be formed any more: the factor has overwritten M) */
            const double rs = 0.0;
            const double h2 = s1_wsum(rs * rs);
            if ( lane == 0 && it < P.hist_len )
            {
               double* hh = P.hist + 16 * it;
               hh[12] = sqrt(e2); hh[13] = (double) sh.fl[6]; hh[14] = sqrt(dy2); hh[15] = sqrt(h2);
            }
         }
         S1_BAR();
      }
      const double amax = steplen(true);
      S1_STAMP(10);
      double alpha = fmin(1.0, gamma_eff * amax);
      if ( !(alpha == alpha) || !(fabs(alpha) < 1e300) )
      {
         status = HS_S1_NUMERIC; numwhere = __LINE__;
         break;
      }
      if ( P.hist != NULL && tid == 0 && it < P.hist_len )
      {
         double* h = P.hist + 16 * it;
         h[9] = aa; h[10] = alpha; h[11] = dt;
      }

      /* ================= step, with a Cholesky check of the new X and Z (halved on failure) */
      bool accepted = false;
      for (int attempt = 0; attempt < 8; ++attempt)
      {
         const int ff = trial_factor(alpha, false);
         if ( ff == 0 )
         {
            accepted = true;
            break;
         }
         alpha *= 0.5;
         ++chol_fail;
      }
      if ( !accepted )
      {
         status = HS_S1_NUMERIC; numwhere = __LINE__;
         factors_valid = false;
         break;
      }
      for (int k = 0; k < K; ++k)
      {
         const S1Blk& B = sh.blk[k];
         const int n = B.n, p = B.p;
         for (int e = tid; e < n * n; e += S1_NT)
         {
            const int r = s1_div(e, n), c = e - r * n;
            sm[B.oX + r * p + c] = fma(alpha, LP(B.E)[r * p + c], sm[B.oX + r * p + c]);
            LP(B.Z)[r * p + c] = fma(alpha, LP(B.B)[r * p + c], LP(B.Z)[r * p + c]);
         }
      }
      {
         const double taun = tau + alpha * dt;
         if ( tid < m )
         {
            const double yn = fma(alpha, VEC(V_dy)[tid], VEC(V_y)[tid]);
            VEC(V_y)[tid] = yn;
            VEC(V_cv)[tid + 1] = yn;
         }
         for (int r = tro(256); r < q; r += S1_NT)
         {
            QV(Q_x)[r] = fma(alpha, QV(Q_dx)[r], QV(Q_x)[r]);
            QV(Q_z)[r] = fma(alpha, QV(Q_dz)[r], QV(Q_z)[r]);
         }
         if ( tid == 0 )
         {
            VEC(V_cv)[0] = -taun;
            sh.sc[SC_TAU] = taun;
            sh.sc[SC_KAPPA] = kappa + alpha * dk;
         }
      }
      factors_valid = true;
      alpha_last = s1_uni(alpha);
      S1_BAR();
      S1_STAMP(11);
   }

   /* ---- results: the iterate as it is (the caller scales by 1 / tau or normalises the ray), one block of scalars */
   __syncthreads();
   /* a numerical failure with keep_on_fail: y, x, z, X, Z in device memory stay what the node's setters left there (the caller's
    * start point, if any), so that the general path can take the same problem from the same start */
   if ( !(status == HS_S1_NUMERIC && P.keep_on_fail) )
   {
   if ( tid < m )
   {
      P.y[tid] = VEC(V_y)[tid];
      if ( P.hy != NULL ) P.hy[tid] = VEC(V_y)[tid];
   }
   for (int r = tid; r < q; r += S1_NT)
   {
      P.x[r] = QV(Q_x)[r];
      P.z[r] = QV(Q_z)[r];
      if ( P.hx != NULL ) { P.hx[r] = QV(Q_x)[r]; P.hz[r] = QV(Q_z)[r]; }
   }
   for (int k = 0; k < K; ++k)
   {
      const S1Blk& B = sh.blk[k];
      const int n = B.n, p = B.p;
      for (int e = tid; e < n * n; e += S1_NT)
      {
         const int r = s1_div(e, n), c = e - r * n;
         P.X[k][e] = sm[B.oX + r * p + c];
         P.Z[k][e] = LP(B.Z)[r * p + c];
      }
   }
   }
   __threadfence_system();
   __syncthreads();
   if ( tid == 0 )
   {
      out[1] = (double) it; out[2] = pobj; out[3] = dobj; out[4] = pinf; out[5] = dinf; out[6] = dabs_; out[7] = gap; out[8] = mu;
      out[9] = sh.sc[SC_TAU]; out[10] = sh.sc[SC_KAPPA]; out[11] = (double) chol_fail; out[12] = (double) warm;
      out[13] = (double) pre_valid; out[14] = pre_scale; out[15] = normb; out[16] = normC;
      out[17] = (double) (clock64() - t_start);
      out[43] = (double) (wall_clock64() - w_start);
      for (int i = 0; i < 22; ++i)
         out[18 + i] = sh.prof[i];
      out[40] = (double) sh.fl[2]; out[41] = (double) sh.fl[3];
      out[42] = (double) (((size_t) (void*) sh.blk[0].vval) >> 32);
      out[44] = (double) numwhere;
      out[0] = (double) status;
      __threadfence_system();
      if ( P.flag != NULL )
         __hip_atomic_store(P.flag, P.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
   }
   };
   if ( sh.fl[30] )
      body(std::true_type{});
   else
      body(std::false_type{});
}

}

static hs_attr_mask s1_attr_done;

int S1_LAUNCH(hipStream_t st, const hs_solve1_args* a)
{
   HS_CALL( hs_func_max_lds(reinterpret_cast<const void*>(&S1_KERNEL), S1_DYN_LDS, &s1_attr_done) );
   hipLaunchKernelGGL(S1_KERNEL, dim3(1), dim3(S1_NT), S1_DYN_LDS, st, *a);
   if ( hipGetLastError() != hipSuccess )
      return HS_ERR_HIP;
   return HS_OK;
}


#define S1_CAT2(a, b) a##b
#define S1_CAT(a, b) S1_CAT2(a, b)
/* debug counters of this instance (all zero in a release build) */
int S1_CAT(S1_LAUNCH, _dbg)(unsigned int* out4)
{
#ifdef S1_DEBUG
   return hipMemcpyFromSymbol(out4, HIP_SYMBOL(s1_dbg), 4 * sizeof(unsigned int)) == hipSuccess ? 1 : -1;
#else
   out4[0] = out4[1] = out4[2] = out4[3] = 0;
   return 0;
#endif
}

#endif      /* S1_HOST_PART */
