/* eig.hip - symmetric eigenvalue kernels.
 *
 *  (1) hs_lanczos_lmin: smallest eigenvalue of a dense symmetric matrix by Lanczos with full (twice-applied classical
 *      Gram-Schmidt) re-orthogonalisation.  The interior-point step lengths need lambda_min(L^-1 dX L^-T) four times per
 *      iteration; a Householder tridiagonalisation is n dependent rank-2 updates (latency bound on a GPU), whereas
 *      Lanczos is a few dozen matrix-vector products whose matrix (2-8 MB at n = 500-1000) stays in L2 / Infinity Cache.
 *      The Ritz value comes with its residual bound, and the caller steps with the pessimistic value theta - resid.
 *  (2) hs_syev_jacobi: full eigen-decomposition by parallel-order cyclic Jacobi: every round applies n/2 disjoint plane
 *      rotations two-sided in ONE kernel (each thread owns the 2 x 2 intersection of two rotation pairs), so a sweep is
 *      n - 1 launches with n^2/4-way parallelism.  It backs the SCIPlapack* surface (lapack_interface.c:178-603: DSYEVR
 *      with RANGE = 'I', 'V', 'A'); ordering and eigenvectors-as-rows follow lapack_interface.c:507-603.
 */
#include "hs_kernels.h"
#include <math.h>

#define HS_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if ( e_ != hipSuccess ) { hs_record_hip_error(e_, "kernel launch", __FILE__, __LINE__); return HS_ERR_HIP; } } while (0)

__global__ void k_lmin_tiny(int n, const double* __restrict__ A0, const double* __restrict__ A1, double* __restrict__ res0,
   double* __restrict__ res1, const double* __restrict__ L0, const double* __restrict__ L1);
__global__ void k_lmin_tiny_multi(hs_step_jobs P);

/* ---------------------------------------------------------------------------------------------------------------- */
/* Lanczos                                                                                                            */
/* ---------------------------------------------------------------------------------------------------------------- */

__device__ __forceinline__ double wsum(double v)
{
#pragma unroll
   for (int off = 32; off > 0; off >>= 1)
      v += __shfl_down(v, off, 64);
   return v;
}

/* block-wide sum over 1024 threads, result broadcast to all */
__device__ __forceinline__ double bsum1024(double v, double* sh)
{
   v = wsum(v);
   const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
   __syncthreads();
   if ( lane == 0 )
      sh[wave] = v;
   __syncthreads();
   double r = 0.0;
   for (int w = 0; w < (int) (blockDim.x >> 6); ++w)
      r += sh[w];
   return r;
}

__global__ void __launch_bounds__(1024) k_lanczos_init(int n, double* __restrict__ Q, double* __restrict__ meta)
{
   __shared__ double sh[16];
   double s = 0.0;
   for (int i = threadIdx.x; i < n; i += blockDim.x)
   {
      /* fixed pseudo-random start vector: reproducible, not orthogonal to anything in particular */
      unsigned long long h = (unsigned long long) (i + 1) * 0x9E3779B97F4A7C15ULL;
      h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ULL; h ^= h >> 32;
      const double v = 0.5 + (double) (h >> 11) * (1.0 / 9007199254740992.0);
      Q[i] = v;
      s += v * v;
   }
   s = bsum1024(s, sh);
   const double inv = 1.0 / sqrt(s);
   for (int i = threadIdx.x; i < n; i += blockDim.x)
      Q[i] *= inv;
   if ( threadIdx.x == 0 )
   {
      meta[0] = -1.0;      /* breakdown step (none) */
      meta[1] = 0.0;       /* largest |alpha| + beta seen (scale) */
   }
}

/* One Lanczos step after v = W q_j was formed: alpha_j, re-orthogonalise v against q_0..q_j twice, beta_j, q_{j+1}.
 * One workgroup; v lives in global memory (vbuf) and is updated in place. */
__global__ void __launch_bounds__(1024) k_lanczos_step(int n, int j, double* __restrict__ Q, double* __restrict__ vbuf,
   double* __restrict__ alpha, double* __restrict__ beta, double* __restrict__ meta)
{
   __shared__ double sh[16];
   __shared__ double coef[256];
   const int tid = threadIdx.x;
   const int nwaves = blockDim.x >> 6;
   const int lane = tid & 63, wave = tid >> 6;

   for (int pass = 0; pass < 2; ++pass)
   {
      /* c_i = q_i . v : one wave per i */
      for (int i = wave; i <= j; i += nwaves)
      {
         const double* q = Q + (long long) i * n;
         double s = 0.0;
         for (int e = lane; e < n; e += 64)
            s += q[e] * vbuf[e];
         s = wsum(s);
         if ( lane == 0 )
            coef[i] = s;
      }
      __syncthreads();
      if ( pass == 0 && tid == 0 )
         alpha[j] = coef[j];
      /* v -= sum_i c_i q_i */
      for (int e = tid; e < n; e += blockDim.x)
      {
         double s = vbuf[e];
         for (int i = 0; i <= j; ++i)
            s -= coef[i] * Q[(long long) i * n + e];
         vbuf[e] = s;
      }
      __syncthreads();
   }
   double s = 0.0;
   for (int e = tid; e < n; e += blockDim.x)
      s += vbuf[e] * vbuf[e];
   s = bsum1024(s, sh);
   const double b = sqrt(s);
   double scale = meta[1];
   const double aj = fabs(alpha[j]);
   __syncthreads();
   if ( aj + b > scale )
      scale = aj + b;
   const bool broke = (meta[0] >= 0.0) || !(b > 1e-13 * scale) || !(b > 1e-300);
   const double inv = broke ? 0.0 : 1.0 / b;
   double* qn = Q + (long long) (j + 1) * n;
   for (int e = tid; e < n; e += blockDim.x)
      qn[e] = vbuf[e] * inv;
   if ( tid == 0 )
   {
      beta[j] = broke ? 0.0 : b;
      meta[1] = scale;
      if ( broke && meta[0] < 0.0 )
         meta[0] = (double) j;
   }
}

/* Smallest eigenvalue of the k x k tridiagonal (alpha, beta) by 64-way multisection on the Sturm count, its eigenvector
 * by inverse iteration; res = {theta, |beta_{k-1} s_{k-1}|, k}.  One wavefront. */
struct tridiag_smem { double d[260], e[260], sv[260], wk[4][260]; };

/* reciprocal by v_rcp_f64 and one Newton step (full precision for finite, normal t): a third of the latency of a division,
 * which is what the sequential recurrences below are made of */
__device__ __forceinline__ double rcp_newton(double t)
{
   double r = __builtin_amdgcn_rcp(t);
   r = fma(fma(-t, r, 1.0), r, r);
   return r;
}

/* executed by ONE wavefront (all 64 lanes); other wavefronts of the workgroup must not call it */
/* (ovr_idx, ovr_alpha, ovr_beta): entry ovr_idx of alpha / beta is taken from the arguments instead of memory (the
 * caller has just stored it and must not rely on reading it back through the vector cache); meta0 = breakdown step */
__device__ void tridiag_min_wave(int kmax, const double* __restrict__ alpha, const double* __restrict__ beta,
   double meta0, int ovr_idx, double ovr_alpha, double ovr_beta, double* __restrict__ res, tridiag_smem& T)
{
   double (&d)[260] = T.d;
   double (&e)[260] = T.e;
   double (&sv)[260] = T.sv;
   double (&wk)[4][260] = T.wk;
   const int lane = threadIdx.x & 63;
   int k = kmax;
   if ( meta0 >= 0.0 )
      k = (int) meta0 + 1;
   if ( k > 256 ) k = 256;
   for (int i = lane; i < k; i += 64)
   {
      d[i] = (i == ovr_idx) ? ovr_alpha : alpha[i];
      const double ei = (i + 1 < k) ? ((i == ovr_idx) ? ovr_beta : beta[i]) : 0.0;
      e[i] = ei;
      wk[3][i] = ei * ei;
   }
   __builtin_amdgcn_s_waitcnt(0);
   __builtin_amdgcn_wave_barrier();
   /* Gershgorin interval */
   double lo = 1e300, hi = -1e300;
   for (int i = 0; i < k; ++i)
   {
      const double r = (i > 0 ? fabs(e[i - 1]) : 0.0) + (i + 1 < k ? fabs(e[i]) : 0.0);
      lo = fmin(lo, d[i] - r);
      hi = fmax(hi, d[i] + r);
   }
   const double span0 = fmax(hi - lo, 1e-300);
   lo -= 1e-12 * span0 + 1e-300;
   hi += 1e-12 * span0 + 1e-300;
   const double pivmin = 1e-290;
   for (int round = 0; round < 14; ++round)
   {
      const double x = lo + (hi - lo) * (double) (lane + 1) / 65.0;
      /* number of eigenvalues < x (Sturm count; |t| is kept inside [pivmin, 1 / pivmin] so that the reciprocal is exact enough) */
      int cnt = 0;
      double t = d[0] - x;
      if ( fabs(t) < pivmin ) t = -pivmin;
      if ( t < 0.0 ) ++cnt;
      for (int i = 1; i < k; ++i)
      {
         t = d[i] - x - wk[3][i - 1] * rcp_newton(t);
         if ( fabs(t) < pivmin ) t = -pivmin;
         if ( !(fabs(t) < 1e290) ) t = (t < 0.0) ? -1e290 : 1e290;
         if ( t < 0.0 ) ++cnt;
      }
      const unsigned long long m = __ballot(cnt >= 1);
      const int first = m ? __ffsll((long long) m) - 1 : 64;    /* first lane whose shift is above lambda_min */
      const double w = (hi - lo) / 65.0;
      const double nlo = lo + w * (double) first;
      const double nhi = (first < 64) ? lo + w * (double) (first + 1) : hi;
      lo = nlo; hi = nhi;
      if ( hi - lo <= 4e-16 * fmax(fabs(lo), fabs(hi)) )
         break;
   }
   const double theta = 0.5 * (lo + hi);

   if ( lane == 0 )
   {
      /* inverse iteration with the shifted tridiagonal: Gaussian elimination with partial pivoting, factored ONCE (three
       * diagonals after pivoting: wk[0] = reciprocal of the diagonal, wk[1] = first super, wk[2] = second super, wk[3] =
       * multipliers, d[] = 1 where rows were swapped), then three solves made of multiplications only */
      double resid = 0.0;
      if ( k == 1 )
      {
         sv[0] = 1.0;
      }
      else
      {
         const double shift = theta;
         const double tiny = 1e-14 * fmax(span0, fmax(fabs(theta), 1e-300));
         double dd = d[0] - shift, du = e[0];
         for (int i = 0; i < k - 1; ++i)
         {
            const double dl = e[i];
            const double dn = d[i + 1] - shift;
            const double un = (i + 2 < k) ? e[i + 1] : 0.0;
            if ( fabs(dd) >= fabs(dl) )
            {
               if ( fabs(dd) < tiny ) dd = tiny;
               const double rinv = rcp_newton(dd);
               const double mlt = dl * rinv;
               wk[0][i] = rinv; wk[1][i] = du; wk[2][i] = 0.0; wk[3][i] = mlt; d[i] = 0.0;
               dd = dn - mlt * du;
               du = un;
            }
            else
            {
               const double rinv = rcp_newton(dl);
               const double mlt = dd * rinv;
               wk[0][i] = rinv; wk[1][i] = dn; wk[2][i] = un; wk[3][i] = mlt; d[i] = 1.0;
               dd = du - mlt * dn;
               du = -mlt * un;
            }
         }
         if ( fabs(dd) < tiny ) dd = tiny;
         wk[0][k - 1] = rcp_newton(dd); wk[1][k - 1] = 0.0; wk[2][k - 1] = 0.0;
         const double s0 = 1.0 / sqrt((double) k);
         for (int i = 0; i < k; ++i)
            sv[i] = s0;
         for (int iter = 0; iter < 3; ++iter)
         {
            /* forward: apply the row operations; the running entry stays in a register */
            double cur = sv[0];
            for (int i = 0; i < k - 1; ++i)
            {
               const double nxt = sv[i + 1];
               if ( d[i] == 0.0 )
               {
                  sv[i] = cur;
                  cur = nxt - wk[3][i] * cur;
               }
               else
               {
                  sv[i] = nxt;
                  cur = cur - wk[3][i] * nxt;
               }
            }
            /* back substitution */
            double x1 = cur * wk[0][k - 1], x2 = 0.0;
            double nrm = x1 * x1;
            sv[k - 1] = x1;
            for (int i = k - 2; i >= 0; --i)
            {
               const double xi = (sv[i] - wk[1][i] * x1 - wk[2][i] * x2) * wk[0][i];
               sv[i] = xi;
               nrm += xi * xi;
               x2 = x1; x1 = xi;
            }
            nrm = sqrt(nrm);
            if ( !(nrm > 0.0) || !(nrm < 1e300) )
            {
               for (int i = 0; i < k; ++i)
                  sv[i] = s0;
               break;
            }
            const double rn = 1.0 / nrm;
            for (int i = 0; i < k; ++i)
               sv[i] *= rn;
         }
      }
      const double blast = (meta0 >= 0.0) ? 0.0 : ((k - 1 == ovr_idx) ? ovr_beta : beta[k - 1]);
      resid = fabs(blast * sv[k - 1]);
      res[0] = theta;
      res[1] = resid;
      res[2] = (double) k;
   }
}

__global__ void __launch_bounds__(64) k_tridiag_min(int kmax, const double* __restrict__ alpha, const double* __restrict__ beta,
   const double* __restrict__ meta, double* __restrict__ res)
{
   __shared__ tridiag_smem T;
   tridiag_min_wave(kmax, alpha, beta, meta[0], -1, 0.0, 0.0, res, T);
}

/* ---- fused Lanczos: ONE launch per step, two matrices per launch ------------------------------------------------- */
/* Launch j (0 <= j <= k) of a run of k steps.  Every workgroup first finishes step j - 1 REDUNDANTLY (same data, same
 * order: identical bits in every workgroup): alpha_{j-1}, twice-applied classical Gram-Schmidt of v = W q_{j-1} against
 * q_0 .. q_{j-1}, beta_{j-1}, q_j (kept in LDS; workgroup 0 also stores it and the scalars).  Then workgroup g forms its
 * rows of v = W q_j.  The kernel boundary is the only grid-wide synchronisation; v and the meta scalars alternate
 * between two buffers so that no workgroup reads what another one writes in the same launch.  Launch k ends with the
 * tridiagonal eigenproblem on workgroup 0.  blockIdx.y selects the matrix (X-side / Z-side of a block). */
struct lanczos_job { const double* W; double* Q; double* v0; double* v1; double* alpha; double* beta; double* meta; double* res; unsigned long long* sync; };
struct lanczos_jobs { lanczos_job job[2]; };

__global__ void __launch_bounds__(1024) k_lanczos_fused(int n, int j, int k, lanczos_jobs jobs)
{
   extern __shared__ __attribute__((aligned(16))) double lz_smem[];
   __shared__ double sh[16];
   __shared__ double coef[256];
   __shared__ tridiag_smem T;
   double* vs = lz_smem;            /* n: v, then q_j */
   const lanczos_job J = jobs.job[blockIdx.y];
   const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
   const bool writer = blockIdx.x == 0;
   const double* min_ = J.meta + 2 * (j & 1);
   double* mout = J.meta + 2 * ((j + 1) & 1);
   const double* vin = ((j + 1) & 1) ? J.v1 : J.v0;       /* written by launch j - 1 */
   double* vout = (j & 1) ? J.v1 : J.v0;
   bool broke = false;
   double last_alpha = 0.0, last_beta = 0.0, meta0 = -1.0;

   if ( j == 0 )
   {
      double sacc = 0.0;
      for (int i = tid; i < n; i += 1024)
      {
         /* fixed pseudo-random start vector: reproducible, not orthogonal to anything in particular */
         unsigned long long h = (unsigned long long) (i + 1) * 0x9E3779B97F4A7C15ULL;
         h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ULL; h ^= h >> 32;
         const double v = 0.5 + (double) (h >> 11) * (1.0 / 9007199254740992.0);
         vs[i] = v;
         sacc += v * v;
      }
      sacc = bsum1024(sacc, sh);
      const double inv = 1.0 / sqrt(sacc);
      for (int i = tid; i < n; i += 1024)
      {
         const double qv = vs[i] * inv;
         vs[i] = qv;
         if ( writer )
            J.Q[i] = qv;
      }
      if ( writer && tid == 0 )
      {
         mout[0] = -1.0;      /* breakdown step (none) */
         mout[1] = 0.0;       /* largest |alpha| + beta seen (scale) */
      }
   }
   else
   {
      const int jj = j - 1;       /* the step being finished */
      for (int e = tid; e < n; e += 1024)
         vs[e] = vin[e];
      __syncthreads();
      double aj = 0.0;
      for (int pass = 0; pass < 2; ++pass)
      {
         /* c_i = q_i . v : one wave per i */
         for (int i = wave; i <= jj; i += 16)
         {
            const double* q = J.Q + (long long) i * n;
            double sacc = 0.0;
            for (int e = lane; e < n; e += 64)
               sacc += q[e] * vs[e];
            sacc = wsum(sacc);
            if ( lane == 0 )
               coef[i] = sacc;
         }
         __syncthreads();
         if ( pass == 0 )
            aj = coef[jj];
         /* v -= sum_i c_i q_i */
         for (int e = tid; e < n; e += 1024)
         {
            double sacc = vs[e];
            for (int i = 0; i <= jj; ++i)
               sacc -= coef[i] * J.Q[(long long) i * n + e];
            vs[e] = sacc;
         }
         __syncthreads();
      }
      double sacc = 0.0;
      for (int e = tid; e < n; e += 1024)
         sacc += vs[e] * vs[e];
      sacc = bsum1024(sacc, sh);
      const double b = sqrt(sacc);
      double scale = min_[1];
      if ( fabs(aj) + b > scale )
         scale = fabs(aj) + b;
      broke = (min_[0] >= 0.0) || !(b > 1e-13 * scale) || !(b > 1e-300);
      const double inv = broke ? 0.0 : 1.0 / b;
      double* qn = J.Q + (long long) j * n;
      for (int e = tid; e < n; e += 1024)
      {
         const double qv = vs[e] * inv;
         vs[e] = qv;
         if ( writer )
            qn[e] = qv;
      }
      last_alpha = aj;
      last_beta = broke ? 0.0 : b;
      meta0 = (broke && min_[0] < 0.0) ? (double) jj : min_[0];
      if ( writer && tid == 0 )
      {
         J.alpha[jj] = last_alpha;
         J.beta[jj] = last_beta;
         mout[1] = scale;
         mout[0] = meta0;
      }
   }
   __syncthreads();
   if ( j < k )
   {
      /* my rows of v = W q_j (q_j = 0 after a breakdown: harmless) */
      const int G = gridDim.x;
      const int rows = (n + G - 1) / G;
      const int r0 = blockIdx.x * rows;
      const int r1 = min(n, r0 + rows);
      for (int r = r0 + wave; r < r1; r += 16)
      {
         const double* wr = J.W + (long long) r * n;
         double sacc = 0.0;
         for (int e = lane; e < n; e += 64)
            sacc += wr[e] * vs[e];
         sacc = wsum(sacc);
         if ( lane == 0 )
            vout[r] = sacc;
      }
   }
   else if ( writer && wave == 0 )
   {
      tridiag_min_wave(k, J.alpha, J.beta, meta0, j - 1, last_alpha, last_beta, J.res, T);
   }
}

/* ---- the same run in ONE launch -------------------------------------------------------------------------------------- */
/* All k + 1 rounds of k_lanczos_fused inside one kernel (a round is a few microseconds of work, a launch 11 - 12 us of
 * latency: 17 steps took 0.2 ms of an iteration's 3 ms outside the Schur complement).  What makes the hand-over between
 * workgroups cheap on this device: the eight XCDs have an L2 each, and an agent-scope release / acquire pair writes back and
 * invalidates those (a first version with __threadfence() + acquire loads needed 27 us per round - worse than the launches).
 * Here nothing that crosses workgroups goes through a non-coherent cache, and there is no separate synchronisation at all:
 *  - the Lanczos basis stays in LDS (every workgroup builds all of it anyway: (k + 2) n doubles, so n <= about 850 at 16 steps);
 *  - the only data exchanged - the n entries of v = W q_j - are stored and loaded as relaxed agent-scope atomics (sc1: served
 *    at the device's coherence point), and the data is its own signal: an exchange vector is all NaN before it is written,
 *    and a reader polls each entry it needs until it is a number (one round trip when the entry is already there; a
 *    counter would cost three: wait for the stores, count, poll).  Three vectors rotate: round j writes vector j mod 3, reads
 *    j - 1 mod 3 and every workgroup resets its own rows of j + 1 mod 3 to NaN - safe, because whoever has reached round j
 *    has seen everybody's rows of round j - 1, and those were written after their writers' last read of vector j + 1 mod 3.
 *    The run ends with exactly one clean vector (k mod 3): the caller rotates the start of the next run onto it (rot).
 *    Matrices of several sizes take turns on the same vectors: every run wipes nwipe >= n rows, the longest of them.
 *  - every wait is bounded: on expiry the error word is set, the workgroups leave and the first one reports NaN, which ends
 *    the solve as a numerical failure (a NaN produced by the arithmetic itself ends the same way).
 * The grid is small (n / 16 workgroups per matrix) and co-resident.  Identical arithmetic in identical order: the results are
 * bitwise those of the launch-per-round form. */
#define LZ_SPIN_LIMIT (1 << 18)
#define LZ_STRIDE 8192            /* doubles between the exchange vectors (n <= 8192 on this path) */

__global__ void __launch_bounds__(1024) k_lanczos_persist(int n, int k, lanczos_jobs jobs, int rot0, int rot1, int nwipe)
{
   extern __shared__ __attribute__((aligned(16))) double lz_smem[];
   __shared__ double sh[16];
   __shared__ double coef[256];
   __shared__ double al[256], be[256];
   __shared__ tridiag_smem T;
   __shared__ int okflag;
   double* vs = lz_smem;            /* n: v, then q_j */
   double* Qs = lz_smem + n;        /* (k + 1) x n: the basis */
   const lanczos_job J = jobs.job[blockIdx.y];
   const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
   const bool writer = blockIdx.x == 0;
   const int G = gridDim.x;
   int* err = reinterpret_cast<int*>(J.sync);                /* error word, exchange vectors behind it */
   double* xv = reinterpret_cast<double*>(J.sync + 2);
   const int rot = blockIdx.y == 0 ? rot0 : rot1;
   const int rows = (n + G - 1) / G;
   const int r0 = blockIdx.x * rows;
   const int r1 = min(n, r0 + rows);
   /* the rows this workgroup wipes: its share of the longest vector any run on these exchange vectors has (matrices of several
    * sizes take turns on them: the clean vector a run leaves behind must be clean for the next one's length) */
   const int wrows = (nwipe + G - 1) / G;
   const int w0 = blockIdx.x * wrows;
   const int w1 = min(nwipe, w0 + wrows);
   double scale = 0.0, meta0 = -1.0;
   if ( tid == 0 )
      okflag = 1;

   for (int j = 0; j <= k; ++j)
   {
      const double* vin = xv + (long long) ((rot + j + 2) % 3) * LZ_STRIDE;       /* written in round j - 1 */
      double* vout = xv + (long long) ((rot + j) % 3) * LZ_STRIDE;
      double* vnext = xv + (long long) ((rot + j + 1) % 3) * LZ_STRIDE;
      double* qn = Qs + (long long) j * n;
      if ( j == 0 )
      {
         double sacc = 0.0;
         for (int i = tid; i < n; i += 1024)
         {
            /* fixed pseudo-random start vector: reproducible, not orthogonal to anything in particular */
            unsigned long long h = (unsigned long long) (i + 1) * 0x9E3779B97F4A7C15ULL;
            h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ULL; h ^= h >> 32;
            const double v = 0.5 + (double) (h >> 11) * (1.0 / 9007199254740992.0);
            vs[i] = v;
            sacc += v * v;
         }
         sacc = bsum1024(sacc, sh);
         const double inv = 1.0 / sqrt(sacc);
         for (int i = tid; i < n; i += 1024)
         {
            const double qv = vs[i] * inv;
            vs[i] = qv;
            qn[i] = qv;
         }
         for (int r = w0 + tid; r < w1; r += 1024)
            __hip_atomic_store(&vnext[r], __builtin_nan(""), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      else
      {
         const int jj = j - 1;       /* the step being finished */
         bool mine_ok = true;
         for (int e = tid; e < n; e += 1024)
         {
            double v = __hip_atomic_load(&vin[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while ( v != v )
            {
               if ( ++spins > LZ_SPIN_LIMIT )
               {
                  mine_ok = false;
                  break;
               }
               __builtin_amdgcn_s_sleep(1);
               v = __hip_atomic_load(&vin[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            vs[e] = v;
         }
         if ( !mine_ok )
         {
            okflag = 0;
            atomicExch(err, 1);
         }
         __syncthreads();
         if ( !okflag )
         {
            if ( writer && tid == 0 )
            {
               J.res[0] = __builtin_nan("");
               J.res[1] = __builtin_nan("");
               J.res[2] = 0.0;
            }
            return;
         }
         /* everybody's rows of round j - 1 have been seen: my rows of the vector after next can be wiped.  The stores are
          * acknowledged before this round's rows go out (vmcnt below): who sees those finds the wiped vector wiped */
         if ( j < k )
         {
            for (int r = w0 + tid; r < w1; r += 1024)
               __hip_atomic_store(&vnext[r], __builtin_nan(""), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
         }
         double aj = 0.0;
         for (int pass = 0; pass < 2; ++pass)
         {
            /* c_i = q_i . v : one wave per i */
            for (int i = wave; i <= jj; i += 16)
            {
               const double* q = Qs + (long long) i * n;
               double sacc = 0.0;
               for (int e = lane; e < n; e += 64)
                  sacc += q[e] * vs[e];
               sacc = wsum(sacc);
               if ( lane == 0 )
                  coef[i] = sacc;
            }
            __syncthreads();
            if ( pass == 0 )
               aj = coef[jj];
            /* v -= sum_i c_i q_i */
            for (int e = tid; e < n; e += 1024)
            {
               double sacc = vs[e];
               for (int i = 0; i <= jj; ++i)
                  sacc -= coef[i] * Qs[(long long) i * n + e];
               vs[e] = sacc;
            }
            __syncthreads();
         }
         double sacc = 0.0;
         for (int e = tid; e < n; e += 1024)
            sacc += vs[e] * vs[e];
         sacc = bsum1024(sacc, sh);
         const double b = sqrt(sacc);
         if ( fabs(aj) + b > scale )
            scale = fabs(aj) + b;
         const bool broke = (meta0 >= 0.0) || !(b > 1e-13 * scale) || !(b > 1e-300);
         const double inv = broke ? 0.0 : 1.0 / b;
         for (int e = tid; e < n; e += 1024)
         {
            const double qv = vs[e] * inv;
            vs[e] = qv;
            qn[e] = qv;
         }
         if ( broke && meta0 < 0.0 )
            meta0 = (double) jj;
         if ( tid == 0 )
         {
            al[jj] = aj;
            be[jj] = broke ? 0.0 : b;
         }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if ( j < k )
      {
         /* my rows of v = W q_j (q_j = 0 after a breakdown: harmless) */
         for (int r = r0 + wave; r < r1; r += 16)
         {
            const double* wr = J.W + (long long) r * n;
            double sacc = 0.0;
            for (int e = lane; e < n; e += 64)
               sacc += wr[e] * vs[e];
            sacc = wsum(sacc);
            if ( lane == 0 )
               __hip_atomic_store(&vout[r], sacc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
         }
      }
      else if ( writer && wave == 0 )
         tridiag_min_wave(k, al, be, meta0, -1, 0.0, 0.0, J.res, T);
   }
}

/* device words a caller of the one-launch form provides per pair of matrices (hs_lanczos_lmin2's dsync): per matrix an error
 * word (2 x 8 bytes) and three exchange vectors; hs_lanczos_sync_reset puts them into the state a first run expects */
long long hs_lanczos_sync_words(void)
{
   return 2LL * (2 + 3LL * LZ_STRIDE);
}

int hs_lanczos_sync_reset(hipStream_t s, unsigned long long* dsync, int* rot)
{
   if ( dsync == NULL )
      return HS_OK;
   HS_HIP( hipMemsetAsync(dsync, 0xFF, (size_t) hs_lanczos_sync_words() * sizeof(unsigned long long), s) );      /* all NaN */
   HS_HIP( hipMemsetAsync(dsync, 0, 2 * sizeof(unsigned long long), s) );
   HS_HIP( hipMemsetAsync(dsync + 2 + 3LL * LZ_STRIDE, 0, 2 * sizeof(unsigned long long), s) );
   rot[0] = rot[1] = 0;
   return HS_OK;
}

long long hs_lanczos_ws(int n, int maxsteps)
{
   return (long long) (maxsteps + 3) * n + 4LL * maxsteps + 64 + 2048;
}

static int hs_lanczos_lmin_unfused(hipStream_t s, int n, const double* W, int maxsteps, double* res, double* ws)
{
   if ( n <= 0 )
      return HS_ERR_ARG;
   int k = maxsteps < n ? maxsteps : n;
   if ( k > 250 ) k = 250;
   double* Q = ws;                                    /* (k + 1) x n */
   double* vbuf = Q + (long long) (maxsteps + 1) * n;   /* n */
   double* alpha = vbuf + n;
   double* beta = alpha + maxsteps;
   double* meta = beta + maxsteps;                    /* 2 (+ padding) */
   double* gws = meta + 64;                           /* gemv split workspace, 2048 doubles */
   hipLaunchKernelGGL(k_lanczos_init, dim3(1), dim3(1024), 0, s, n, Q, meta);
   HS_LAUNCH_CHECK();
   for (int j = 0; j < k; ++j)
   {
      const double* qj = Q + (long long) j * n;
      HS_CALL( hs_gemv_n(s, n, n, W, n, 1, &qj, vbuf, n, gws, 2048) );
      hipLaunchKernelGGL(k_lanczos_step, dim3(1), dim3(1024), 0, s, n, j, Q, vbuf, alpha, beta, meta);
      HS_LAUNCH_CHECK();
   }
   hipLaunchKernelGGL(k_tridiag_min, dim3(1), dim3(64), 0, s, k, alpha, beta, meta, res);
   HS_LAUNCH_CHECK();
   return HS_OK;
}

/* ---- 16 < n <= 64: the whole Lanczos run in ONE launch --------------------------------------------------------------- */
/* One workgroup per matrix (blockIdx.x = X side / Z side).  The operator W = L D L^T is applied in factored form (three
 * products with n x n matrices held in LDS per step; W itself is never formed, and the operator is exactly symmetric when D
 * is), k Lanczos steps with twice-applied classical Gram-Schmidt run with the basis in LDS, and wavefront 0 solves the
 * tridiagonal problem.  Every product and reduction is split over 4 adjacent lanes per row (n <= 64 rows x 4 = 256 threads),
 * so the dependent chains are n / 4 long.  Same algorithm and start vector as k_lanczos_fused; replaces 4 GEMM launches +
 * k + 1 launches per pair of estimates. */
#define LS_MAXK 32
__device__ __forceinline__ double quad_sum(double x)
{
   x += __shfl_xor(x, 1, 64);
   x += __shfl_xor(x, 2, 64);
   return x;
}

__device__ __forceinline__ void d_lanczos_small(int n, int k, const double* __restrict__ Din, const double* __restrict__ Lin,
   double* __restrict__ res)
{
   extern __shared__ double lz2_smem[];
   __shared__ double alpha[LS_MAXK], beta[LS_MAXK], coef[LS_MAXK + 1];
   __shared__ double t1[64], t2[64], v[64];
   __shared__ double shr[4];
   __shared__ tridiag_smem T;
   const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
   const int row = tid >> 2, part = tid & 3;
   const int ld = n | 1;                /* odd row pitch: rows and columns are both conflict-free */
   double* sl = lz2_smem;               /* n x ld: L */
   double* sd = sl + n * ld;            /* n x ld: D */
   double* Q = sd + n * ld;             /* (k + 1) x n */
   for (int e = tid; e < n * n; e += 256)
   {
      const int r = e / n, c = e - r * n;
      sl[r * ld + c] = (c <= r) ? Lin[e] : 0.0;
      sd[r * ld + c] = Din[e];
   }
   /* start vector */
   double sacc = 0.0;
   if ( tid < n )
   {
      unsigned long long h = (unsigned long long) (tid + 1) * 0x9E3779B97F4A7C15ULL;
      h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ULL; h ^= h >> 32;
      const double q0 = 0.5 + (double) (h >> 11) * (1.0 / 9007199254740992.0);
      v[tid] = q0;
      sacc = q0 * q0;
   }
   sacc = wsum(sacc);
   if ( lane == 0 ) shr[wave] = sacc;
   __syncthreads();
   {
      const double inv = 1.0 / sqrt(shr[0] + shr[1] + shr[2] + shr[3]);
      if ( tid < n )
         Q[tid] = v[tid] * inv;
   }
   __syncthreads();
   double scale = 0.0;
   int kbreak = -1;
   for (int j = 0; j < k; ++j)
   {
      const double* qj = Q + j * n;
      /* t1 = L^T q_j  (L lower triangular: column `row` starts at its diagonal) */
      {
         double acc = 0.0;
         if ( row < n )
         {
#pragma unroll 4
            for (int t = row + part; t < n; t += 4)
               acc += sl[t * ld + row] * qj[t];
         }
         acc = quad_sum(acc);
         if ( row < n && part == 0 )
            t1[row] = acc;
      }
      __syncthreads();
      /* t2 = D t1 */
      {
         double acc = 0.0;
         if ( row < n )
         {
#pragma unroll 4
            for (int t = part; t < n; t += 4)
               acc += sd[row * ld + t] * t1[t];
         }
         acc = quad_sum(acc);
         if ( row < n && part == 0 )
            t2[row] = acc;
      }
      __syncthreads();
      /* v = L t2 */
      {
         double acc = 0.0;
         if ( row < n )
         {
#pragma unroll 4
            for (int t = part; t <= row; t += 4)
               acc += sl[row * ld + t] * t2[t];
         }
         acc = quad_sum(acc);
         if ( row < n && part == 0 )
            v[row] = acc;
      }
      __syncthreads();
      double nrm2 = 0.0;
      double aj = 0.0;
      /* classical Gram-Schmidt against q_0 .. q_j, twice: with |alpha| >> beta (the regime of an interior-point iteration)
       * the first pass cancels many digits, and a norm-based "twice is enough" test fires at almost every step anyway */
      for (int pass = 0; pass < 2; ++pass)
      {
         /* c_i = q_i . v, i <= j: four lanes per i */
         {
            double acc = 0.0;
            if ( row <= j )
            {
               const double* q = Q + row * n;
#pragma unroll 4
               for (int e = part; e < n; e += 4)
                  acc += q[e] * v[e];
            }
            acc = quad_sum(acc);
            if ( row <= j && part == 0 )
               coef[row] = acc;
         }
         __syncthreads();
         if ( pass == 0 )
            aj = coef[j];
         /* v -= sum_i c_i q_i: four lanes per entry; squared norm of the result per wavefront */
         {
            double acc = 0.0;
            if ( row < n )
            {
#pragma unroll 4
               for (int i = part; i <= j; i += 4)
                  acc += coef[i] * Q[i * n + row];
            }
            acc = quad_sum(acc);
            double vn = 0.0;
            if ( row < n && part == 0 )
            {
               vn = v[row] - acc;
               v[row] = vn;
            }
            if ( pass == 1 )
            {
               const double sq = wsum(vn * vn);
               if ( lane == 0 ) shr[wave] = sq;
            }
         }
         __syncthreads();
      }
      nrm2 = shr[0] + shr[1] + shr[2] + shr[3];
      const double b = sqrt(nrm2);
      if ( fabs(aj) + b > scale )
         scale = fabs(aj) + b;
      const bool broke = (kbreak >= 0) || !(b > 1e-13 * scale) || !(b > 1e-300);
      if ( tid == 0 )
      {
         alpha[j] = aj;
         beta[j] = broke ? 0.0 : b;
      }
      if ( broke && kbreak < 0 )
         kbreak = j;
      if ( tid < n )
         Q[(j + 1) * n + tid] = broke ? 0.0 : v[tid] / b;
      __syncthreads();
   }
   if ( wave == 0 )
      tridiag_min_wave(k, alpha, beta, (double) kbreak, -1, 0.0, 0.0, res, T);
}

__global__ void __launch_bounds__(256) k_lanczos_small(int n, int k, const double* __restrict__ D0, const double* __restrict__ D1,
   double* __restrict__ res0, double* __restrict__ res1, const double* __restrict__ L0, const double* __restrict__ L1)
{
   d_lanczos_small(n, k, blockIdx.x ? D1 : D0, blockIdx.x ? L1 : L0, blockIdx.x ? res1 : res0);
}

/* the same for SEVERAL blocks in one launch (blockIdx.y = block, blockIdx.x = side): a problem with twenty blocks of fifty rows spent
 * 39 % of its iteration in forty launches of two workgroups each (tests/devtools/multiblock_time.py) */
__global__ void __launch_bounds__(256) k_lanczos_small_multi(hs_step_jobs P, int maxsteps)
{
   const int job = blockIdx.y;
   const int n = P.n[job];
   int k = maxsteps < n ? maxsteps : n;
   if ( k > LS_MAXK ) k = LS_MAXK;
   d_lanczos_small(n, k, blockIdx.x ? P.D1[job] : P.D0[job], blockIdx.x ? P.L1[job] : P.L0[job], blockIdx.x ? P.res1[job] : P.res0[job]);
}

int hs_lanczos_scaled_small(hipStream_t s, int n, int maxsteps, const double* L0, const double* D0, const double* L1, const double* D1,
   double* res0, double* res1)
{
   if ( n <= 16 || n > 64 )
      return HS_ERR_ARG;
   int k = maxsteps < n ? maxsteps : n;
   if ( k > LS_MAXK ) k = LS_MAXK;
   static hs_attr_mask attr_done;
   HS_CALL( hs_func_max_lds(reinterpret_cast<const void*>(&k_lanczos_small), (2 * 64 * 65 + (LS_MAXK + 1) * 64) * (int) sizeof(double), &attr_done) );
   const size_t smem = ((size_t) 2 * n * (n | 1) + (size_t) (k + 1) * n) * sizeof(double);
   hipLaunchKernelGGL(k_lanczos_small, dim3(2), dim3(256), smem, s, n, k, D0, D1, res0, res1, L0, L1);
   HS_LAUNCH_CHECK();
   return HS_OK;
}

/* the step-length estimates of all blocks of one size class in one launch: P->n[j] <= 16 everywhere (k_lmin_tiny) or 16 < n <= 64
 * everywhere (k_lanczos_small); same arithmetic per block as the single launches */
int hs_steplen_small_multi(hipStream_t s, const hs_step_jobs* P, int maxsteps)
{
   if ( P->nblk <= 0 )
      return HS_OK;
   if ( P->nblk > HS_STEP_MAXJOBS )
      return HS_ERR_ARG;
   int nmax = 0, nmin = 1 << 30;
   for (int j = 0; j < P->nblk; ++j)
   {
      if ( P->n[j] > nmax ) nmax = P->n[j];
      if ( P->n[j] < nmin ) nmin = P->n[j];
   }
   if ( nmin < 1 || nmax > 64 || (nmin <= 16) != (nmax <= 16) )
      return HS_ERR_ARG;
   if ( nmax <= 16 )
   {
      hipLaunchKernelGGL(k_lmin_tiny_multi, dim3(2, P->nblk), dim3(64), 0, s, *P);
      HS_LAUNCH_CHECK();
      return HS_OK;
   }
   {
      /* 17 .. 64 rows: the exact eigenvalue is the faster kernel (HIPSDP_STEP_LANCZOS=1: the Lanczos estimate of round 2) */
      static int lanczos = -1;
      if ( lanczos < 0 )
      {
         const char* env = getenv("HIPSDP_STEP_LANCZOS");
         lanczos = (env != NULL && env[0] == '1') ? 1 : 0;
      }
      if ( !lanczos && nmax <= 48 )          /* (at 64 rows the 24 Lanczos steps are as fast: measured 9.0 against 9.2 ms per iteration) */
         return hs_lmin_exact_multi(s, P);
   }
   int k = maxsteps < nmax ? maxsteps : nmax;
   if ( k > LS_MAXK ) k = LS_MAXK;
   static hs_attr_mask attr_done;
   HS_CALL( hs_func_max_lds(reinterpret_cast<const void*>(&k_lanczos_small_multi), (2 * 64 * 65 + (LS_MAXK + 1) * 64) * (int) sizeof(double), &attr_done) );
   const size_t smem = ((size_t) 2 * nmax * (nmax | 1) + (size_t) (k + 1) * nmax) * sizeof(double);
   hipLaunchKernelGGL(k_lanczos_small_multi, dim3(2, P->nblk), dim3(256), smem, s, *P, maxsteps);
   HS_LAUNCH_CHECK();
   return HS_OK;
}

static void lanczos_job_init(lanczos_job* J, int n, int maxsteps, const double* W, double* res, double* ws)
{
   J->W = W;
   J->Q = ws;                                          /* (maxsteps + 1) x n */
   J->v0 = ws + (long long) (maxsteps + 1) * n;        /* n */
   J->v1 = J->v0 + n;                                  /* n */
   J->alpha = J->v1 + n;
   J->beta = J->alpha + maxsteps;
   J->meta = J->beta + maxsteps;                       /* 2 x 2 (+ padding) */
   J->res = res;
   J->sync = NULL;
}

/* smallest eigenvalue of one or two symmetric n x n matrices (W1 may be NULL); res = {theta, residual bound, steps} */
static int lz_no_persist = -1;

int hs_lanczos_lmin2(hipStream_t s, int n, const double* W0, const double* W1, int maxsteps, double* res0, double* res1,
   double* ws0, double* ws1, int* rot, unsigned long long* dsync, int nwipe)
{
   if ( n <= 0 )
      return HS_ERR_ARG;
   const int nb = (W1 != NULL) ? 2 : 1;
   if ( n <= 16 )
   {
      /* tiny blocks are launch bound: one wavefront per matrix diagonalises it exactly in ONE launch */
      hipLaunchKernelGGL(k_lmin_tiny, dim3(nb), dim3(64), 0, s, n, W0, W1, res0, res1, (const double*) NULL, (const double*) NULL);
      HS_LAUNCH_CHECK();
      return HS_OK;
   }
   int k = maxsteps < n ? maxsteps : n;
   if ( k > 250 ) k = 250;
   if ( n > 8192 )
   {
      /* the fused kernel keeps one vector in LDS */
      HS_CALL( hs_lanczos_lmin_unfused(s, n, W0, maxsteps, res0, ws0) );
      if ( W1 != NULL )
         HS_CALL( hs_lanczos_lmin_unfused(s, n, W1, maxsteps, res1, ws1) );
      return HS_OK;
   }
   static hs_attr_mask attr_done;
   HS_CALL( hs_func_max_lds(reinterpret_cast<const void*>(&k_lanczos_fused), 8192 * 8, &attr_done) );
   lanczos_jobs jobs;
   lanczos_job_init(&jobs.job[0], n, maxsteps, W0, res0, ws0);
   if ( W1 != NULL )
      lanczos_job_init(&jobs.job[1], n, maxsteps, W1, res1, ws1);
   else
      jobs.job[1] = jobs.job[0];
   int G = (n + 15) / 16;          /* >= 16 rows per workgroup: one per wavefront */
   if ( G > 128 ) G = 128;
   if ( lz_no_persist < 0 )
   {
      const char* env = getenv("HIPSDP_LANCZOS_LAUNCHES");
      lz_no_persist = (env != NULL && env[0] == '1') ? 1 : 0;
   }
   const size_t lds_persist = (size_t) (k + 2) * (size_t) n * sizeof(double);
   if ( rot != NULL && dsync != NULL && !lz_no_persist && k >= 3 && k <= 250 && lds_persist <= 120 * 1024 && nwipe <= LZ_STRIDE )
   {
      /* one launch (see k_lanczos_persist); rot[0 / 1]: which of the three exchange vectors of a matrix is the clean one */
      jobs.job[0].sync = dsync;
      jobs.job[1].sync = (W1 != NULL) ? dsync + 2 + 3LL * LZ_STRIDE : dsync;
      static hs_attr_mask attr2_done;
      HS_CALL( hs_func_max_lds(reinterpret_cast<const void*>(&k_lanczos_persist), 120 * 1024, &attr2_done) );
      const int ncu = hs_device_cus();
      /* the workgroups wait for each other: all of them must fit on the device at once (a partitioned or smaller device takes
       * the launch-per-step form) */
      static thread_local size_t occ_lds = 0;
      static thread_local int occ_per_cu = 0;
      if ( occ_lds != lds_persist )
      {
         if ( hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_per_cu, reinterpret_cast<const void*>(&k_lanczos_persist), 1024, lds_persist) != hipSuccess )
            occ_per_cu = 0;
         occ_lds = lds_persist;
      }
      const int per_cu = occ_per_cu;
      if ( (long long) per_cu * ncu >= (long long) G * nb )
      {
         hipLaunchKernelGGL(k_lanczos_persist, dim3(G, nb), dim3(1024), lds_persist, s, n, k, jobs, rot[0], rot[1], nwipe > n ? nwipe : n);
         HS_LAUNCH_CHECK();
         rot[0] = (rot[0] + k) % 3;
         if ( nb == 2 )
            rot[1] = (rot[1] + k) % 3;
         return HS_OK;
      }
   }
   for (int j = 0; j <= k; ++j)
   {
      hipLaunchKernelGGL(k_lanczos_fused, dim3(G, nb), dim3(1024), (size_t) n * sizeof(double), s, n, j, k, jobs);
      HS_LAUNCH_CHECK();
   }
   return HS_OK;
}

/* n <= 16: lambda_min(L0 D0 L0^T) and lambda_min(L1 D1 L1^T) in ONE launch (products and eigenvalues) */
int hs_lmin_scaled_tiny(hipStream_t s, int n, const double* L0, const double* D0, const double* L1, const double* D1, double* res0,
   double* res1)
{
   if ( n <= 0 || n > 16 )
      return HS_ERR_ARG;
   hipLaunchKernelGGL(k_lmin_tiny, dim3(2), dim3(64), 0, s, n, D0, D1, res0, res1, L0, L1);
   HS_LAUNCH_CHECK();
   return HS_OK;
}

int hs_lanczos_lmin(hipStream_t s, int n, const double* W, int maxsteps, double* res, double* ws)
{
   return hs_lanczos_lmin2(s, n, W, NULL, maxsteps, res, NULL, ws, NULL, NULL, NULL, 0);
}

/* ---------------------------------------------------------------------------------------------------------------- */
/* Jacobi                                                                                                             */
/* ---------------------------------------------------------------------------------------------------------------- */

/* pair k of round r in the round-robin tournament over np = 2 * half players (np even) */
__device__ __forceinline__ void jac_pair(int np, int r, int k, int* p, int* q)
{
   const int mod = np - 1;
   int a, b;
   if ( k == 0 )
   {
      a = np - 1;
      b = r % mod;
   }
   else
   {
      a = (r + k) % mod;
      b = (r - k + mod) % mod;
   }
   *p = a < b ? a : b;
   *q = a < b ? b : a;
}

/* rotation angles for all pairs of a round; rot[k] = {c, s}; indices >= n (padding player) give the identity */
__global__ void k_jacobi_angles(int n, int np, int r, const double* __restrict__ A, double* __restrict__ rot)
{
   const int k = blockIdx.x * blockDim.x + threadIdx.x;
   if ( k >= np / 2 )
      return;
   int p, q;
   jac_pair(np, r, k, &p, &q);
   double c = 1.0, s = 0.0;
   if ( q < n )
   {
      const double apq = A[(long long) p * n + q];
      const double app = A[(long long) p * n + p];
      const double aqq = A[(long long) q * n + q];
      if ( fabs(apq) > 1e-300 && fabs(apq) > 1e-19 * (fabs(app) + fabs(aqq)) )
      {
         const double th = (aqq - app) / (2.0 * apq);
         const double t = (th >= 0.0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
         c = 1.0 / sqrt(t * t + 1.0);
         s = t * c;
      }
   }
   rot[2 * k] = c;
   rot[2 * k + 1] = s;
}

/* two-sided update A <- J^T A J and Vt <- J^T Vt: thread (k1, k2) owns the 2 x 2 intersection of pairs k1 (rows) and k2
 * (columns); threads with k2 >= half own the rows of Vt instead */
__global__ void k_jacobi_apply(int n, int np, int r, double* __restrict__ A, double* __restrict__ Vt,
   const double* __restrict__ rot)
{
   const int half = np / 2;
   const long long tot = (long long) half * half;
   for (long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x; t < 2 * tot; t += (long long) gridDim.x * blockDim.x)
   {
      if ( t < tot )
      {
         const int k1 = (int) (t / half), k2 = (int) (t % half);
         int p, q, u, v;
         jac_pair(np, r, k1, &p, &q);
         jac_pair(np, r, k2, &u, &v);
         if ( q >= n && v >= n )
            continue;
         const double c1 = rot[2 * k1], s1 = rot[2 * k1 + 1];
         const double c2 = rot[2 * k2], s2 = rot[2 * k2 + 1];
         const bool hq = q < n, hv = v < n;
         double apu = A[(long long) p * n + u];
         double apv = hv ? A[(long long) p * n + v] : 0.0;
         double aqu = hq ? A[(long long) q * n + u] : 0.0;
         double aqv = (hq && hv) ? A[(long long) q * n + v] : 0.0;
         /* rows: [p; q] <- [[c1, -s1], [s1, c1]] [p; q] */
         const double bpu = c1 * apu - s1 * aqu, bqu = s1 * apu + c1 * aqu;
         const double bpv = c1 * apv - s1 * aqv, bqv = s1 * apv + c1 * aqv;
         /* columns: [u, v] <- [u, v] [[c2, s2], [-s2, c2]] */
         apu = c2 * bpu - s2 * bpv; apv = s2 * bpu + c2 * bpv;
         aqu = c2 * bqu - s2 * bqv; aqv = s2 * bqu + c2 * bqv;
         A[(long long) p * n + u] = apu;
         if ( hv ) A[(long long) p * n + v] = apv;
         if ( hq ) A[(long long) q * n + u] = aqu;
         if ( hq && hv ) A[(long long) q * n + v] = aqv;
      }
      else
      {
         /* Vt rows p, q; column pairs taken as consecutive columns */
         const long long tt = t - tot;
         const int k1 = (int) (tt / half), cc = (int) (tt % half);
         int p, q;
         jac_pair(np, r, k1, &p, &q);
         if ( q >= n )
            continue;
         const double c1 = rot[2 * k1], s1 = rot[2 * k1 + 1];
         for (int col = 2 * cc; col < 2 * cc + 2 && col < n; ++col)
         {
            const double vp = Vt[(long long) p * n + col], vq = Vt[(long long) q * n + col];
            Vt[(long long) p * n + col] = c1 * vp - s1 * vq;
            Vt[(long long) q * n + col] = s1 * vp + c1 * vq;
         }
      }
   }
}

/* off-diagonal square sum and diagonal square sum -> out[0], out[1] (single workgroup) */
__global__ void __launch_bounds__(1024) k_jacobi_offnorm(int n, const double* __restrict__ A, double* __restrict__ out)
{
   __shared__ double sh[16];
   double off = 0.0, dg = 0.0;
   const long long tot = (long long) n * n;
   for (long long e = threadIdx.x; e < tot; e += blockDim.x)
   {
      const int r = (int) (e / n), c = (int) (e % n);
      const double v = A[e];
      if ( r == c ) dg += v * v; else off += v * v;
   }
   off = bsum1024(off, sh);
   dg = bsum1024(dg, sh);
   if ( threadIdx.x == 0 )
   {
      out[0] = off;
      out[1] = dg;
   }
}

/* ascending sort of the diagonal and matching row permutation of Vt into V (rank sort, one workgroup) */
__global__ void __launch_bounds__(1024) k_jacobi_sort(int n, const double* __restrict__ A, const double* __restrict__ Vt,
   double* __restrict__ lam, double* __restrict__ V)
{
   for (int i = threadIdx.x; i < n; i += blockDim.x)
   {
      const double di = A[(long long) i * n + i];
      int rank = 0;
      for (int j = 0; j < n; ++j)
      {
         const double dj = A[(long long) j * n + j];
         if ( dj < di || (dj == di && j < i) )
            ++rank;
      }
      lam[rank] = di;
      for (int c = 0; c < n; ++c)
         V[(long long) rank * n + c] = Vt[(long long) i * n + c];
   }
}

/* n <= 64: the whole decomposition in ONE launch of one workgroup, matrix and eigenvector matrix in LDS, parallel-order
 * rounds separated by workgroup barriers, convergence test in the kernel (the typical SCIP-SDP block has 2-50 rows and is
 * latency bound: cons_sdp.c calls the eigen routine for every candidate solution, lapack_interface.c:178-288) */
#define JS 64
__global__ void __launch_bounds__(256) k_jacobi_small(int n, const double* __restrict__ Ain, double* __restrict__ lam,
   double* __restrict__ V, int* __restrict__ info)
{
   __shared__ double a[JS][JS + 1];
   __shared__ double vt[JS][JS + 1];
   __shared__ double rc[JS / 2], rs[JS / 2];
   __shared__ double red[8];
   __shared__ int order[JS];
   const int tid = threadIdx.x;
   const int np = (n + 1) & ~1;
   const int half = np / 2;
   for (int e = tid; e < JS * JS; e += 256)
   {
      const int r = e / JS, c = e % JS;
      double v = 0.0;
      if ( r < n && c < n )
         v = 0.5 * (Ain[(long long) r * n + c] + Ain[(long long) c * n + r]);
      a[r][c] = v;
      vt[r][c] = (r == c) ? 1.0 : 0.0;
   }
   __syncthreads();
   int sweeps = 0;
   double prevoff = 1e300;
   for (sweeps = 0; sweeps < 40 && n > 1; ++sweeps)
   {
      /* off-diagonal vs diagonal mass */
      double off = 0.0, dg = 0.0;
      for (int e = tid; e < n * n; e += 256)
      {
         const int r = e / n, c = e % n;
         const double v = a[r][c];
         if ( r == c ) dg += v * v; else off += v * v;
      }
      for (int o = 32; o > 0; o >>= 1)
      {
         off += __shfl_down(off, o, 64);
         dg += __shfl_down(dg, o, 64);
      }
      if ( (tid & 63) == 0 )
      {
         red[(tid >> 6) * 2] = off;
         red[(tid >> 6) * 2 + 1] = dg;
      }
      __syncthreads();
      off = red[0] + red[2] + red[4] + red[6];
      dg = red[1] + red[3] + red[5] + red[7];
      __syncthreads();
      if ( !(off > 1e-30 * dg) || !(off > 0.0) )
         break;
      /* the rotations of a sweep leave rounding noise of about n eps^2 relative to the diagonal mass: once there, the
       * (quadratically convergent) iteration no longer shrinks the off-diagonal part and further sweeps only add noise */
      if ( off <= 1e-24 * dg && off > 0.25 * prevoff )
         break;
      prevoff = off;
      for (int r = 0; r < np - 1; ++r)
      {
         if ( tid < half )
         {
            int p, q;
            jac_pair(np, r, tid, &p, &q);
            double c = 1.0, s = 0.0;
            if ( q < n )
            {
               const double apq = a[p][q], app = a[p][p], aqq = a[q][q];
               if ( fabs(apq) > 1e-300 && fabs(apq) > 1e-19 * (fabs(app) + fabs(aqq)) )
               {
                  const double th = (aqq - app) / (2.0 * apq);
                  const double t = (th >= 0.0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
                  c = 1.0 / sqrt(t * t + 1.0);
                  s = t * c;
               }
            }
            rc[tid] = c;
            rs[tid] = s;
         }
         __syncthreads();
         /* two-sided update: item (k1, k2) owns the 2 x 2 intersection of pair k1 (rows) and pair k2 (columns) */
         for (int it = tid; it < half * half; it += 256)
         {
            const int k1 = it / half, k2 = it % half;
            int p, q, u, v;
            jac_pair(np, r, k1, &p, &q);
            jac_pair(np, r, k2, &u, &v);
            const bool hq = q < n, hv = v < n;
            const double c1 = rc[k1], s1 = rs[k1], c2 = rc[k2], s2 = rs[k2];
            const double apu = a[p][u], apv = hv ? a[p][v] : 0.0, aqu = hq ? a[q][u] : 0.0, aqv = (hq && hv) ? a[q][v] : 0.0;
            const double bpu = c1 * apu - s1 * aqu, bqu = s1 * apu + c1 * aqu;
            const double bpv = c1 * apv - s1 * aqv, bqv = s1 * apv + c1 * aqv;
            a[p][u] = c2 * bpu - s2 * bpv;
            if ( hv ) a[p][v] = s2 * bpu + c2 * bpv;
            if ( hq ) a[q][u] = c2 * bqu - s2 * bqv;
            if ( hq && hv ) a[q][v] = s2 * bqu + c2 * bqv;
         }
         /* eigenvector rows: item (k1, column) */
         for (int it = tid; it < half * n; it += 256)
         {
            const int k1 = it / n, col = it % n;
            int p, q;
            jac_pair(np, r, k1, &p, &q);
            if ( q < n )
            {
               const double c1 = rc[k1], s1 = rs[k1];
               const double vp = vt[p][col], vq = vt[q][col];
               vt[p][col] = c1 * vp - s1 * vq;
               vt[q][col] = s1 * vp + c1 * vq;
            }
         }
         __syncthreads();
      }
   }
   /* rank sort of the diagonal */
   if ( tid < n )
   {
      const double di = a[tid][tid];
      int rank = 0;
      for (int j = 0; j < n; ++j)
      {
         const double dj = a[j][j];
         if ( dj < di || (dj == di && j < tid) )
            ++rank;
      }
      order[rank] = tid;
      lam[rank] = di;
   }
   __syncthreads();
   if ( V != NULL )
   {
      for (int e = tid; e < n * n; e += 256)
      {
         const int r = e / n, c = e % n;
         V[e] = vt[order[r]][c];
      }
   }
   if ( tid == 0 && info != NULL )
      *info = sweeps;
}

/* rsqrt with two Newton steps (v_rsq_f64 is a low-precision seed) */
__device__ __forceinline__ double rsqrt_nr(double x)
{
   double y = __builtin_amdgcn_rsq(x);
   double h = 0.5 * y, g = x * y;
   double r = fma(-h, g, 0.5);
   g = fma(g, r, g); h = fma(h, r, h);
   r = fma(-h, g, 0.5);
   h = fma(h, r, h);
   return 2.0 * h;
}

#ifdef EIG_TIMING
__device__ long long lm_tbuf[8];
#define LM_T(i) do { if ( threadIdx.x == 0 && blockIdx.x == 0 ) lm_tbuf[i] = wall_clock64(); } while (0)
extern "C" int hipsdp_debug_lm_timing(long long* out)
{
   return hipMemcpyFromSymbol(out, HIP_SYMBOL(lm_tbuf), sizeof(long long) * 8) == hipSuccess ? 0 : 1;
}
#else
#define LM_T(i) do { } while (0)
#endif

/* lane exchange inside a row of 16 lanes on the data-parallel-primitive path (no LDS crossbar round trip as with __shfl) */
template<int CTRL>
__device__ __forceinline__ double lm_dpp(double v)
{
   int lo = __double2loint(v), hi = __double2hiint(v);
   lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
   hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
   return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lm_lane(double v, int l)
{
   const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
   return __hiloint2double(hi, lo);
}
/* sums inside the rows of 16 lanes (every lane of a row gets its row's sum) */
__device__ __forceinline__ double lm_rowsum(double v)
{
   v += lm_dpp<0xB1>(v);              /* quad_perm [1, 0, 3, 2] */
   v += lm_dpp<0x4E>(v);              /* quad_perm [2, 3, 0, 1] */
   v += lm_dpp<0x141>(v);             /* row_half_mirror */
   v += lm_dpp<0x140>(v);             /* row_mirror */
   return v;
}
__device__ __forceinline__ double lm_wsum(double v)
{
   v = lm_rowsum(v);
   return ((lm_lane(v, 0) + lm_lane(v, 16)) + lm_lane(v, 32)) + lm_lane(v, 48);
}
/* reciprocal to full precision: v_rcp_f64 and two Newton steps */
__device__ __forceinline__ double lm_rcp(double t)
{
   double r = __builtin_amdgcn_rcp(t);
   r = fma(fma(-t, r, 1.0), r, r);
   r = fma(fma(-t, r, 1.0), r, r);
   return r;
}
__device__ __forceinline__ double lm_quad(double x)
{
   x += lm_dpp<0xB1>(x);
   x += lm_dpp<0x4E>(x);
   return x;
}

/* smallest eigenvalue of the symmetric n x n matrix in a[][] (n <= 16; destroyed), one wavefront of 64 lanes: Householder
 * tridiagonalisation in LDS (the DSYTD2 recurrence, lane = (row, quarter of the columns)), then Sturm-count multisection on the
 * tridiagonal matrix (64 shifts per round, one per lane; reciprocal by v_rcp_f64 + a Newton step).  Exact to rounding, like the
 * Jacobi diagonalisation it replaces (n = 10: 8 reflectors and about 10 rounds instead of 6 - 8 sweeps of 9 rounds with two
 * barriers each: 28.7 -> about 8 us per launch, twice per iteration of a B&B-sized problem). */
__device__ double lmin_sym16(double (*a)[17], int n, double* vv, double* ww, double* dd, double* e2, int lane)
{
   const int r = lane >> 2, q = lane & 3;
   for (int k = 0; k + 1 < n; ++k)
   {
      const int len = n - k - 1;                       /* length of x = a[k + 1 .., k] */
      const double xi = (lane < len) ? a[k + 1 + lane][k] : 0.0;
      const double x0 = lm_lane(xi, 0);
      const double s2 = lm_lane(lm_rowsum((lane >= 1) ? xi * xi : 0.0), 0);      /* x sits in the first row of lanes */
      double beta = x0, scale = 0.0, t = 0.0;
      if ( s2 > 0.0 )
      {
         const double h2 = x0 * x0 + s2;
         beta = -copysign(h2 * rsqrt_nr(h2), x0);
         t = (beta - x0) * lm_rcp(beta);
         scale = lm_rcp(x0 - beta);
      }
      if ( lane < len )
         vv[lane] = (lane == 0) ? 1.0 : xi * scale;
      if ( lane == 0 )
      {
         e2[k] = beta * beta;
         dd[k] = a[k][k];
      }
      __syncthreads();
      if ( t != 0.0 )
      {
         /* p = tau A22 v: row r of A22, quarter q of its columns */
         double acc = 0.0;
         if ( r < len )
            for (int c = q; c < len; c += 4)
               acc += a[k + 1 + r][k + 1 + c] * vv[c];
         acc = lm_quad(acc);
         const double pr = t * acc;
         const double pv = lm_wsum((q == 0 && r < len) ? pr * vv[r] : 0.0);
         const double wr = (r < len) ? pr - 0.5 * t * pv * vv[r] : 0.0;
         if ( q == 0 && r < len )
            ww[r] = wr;
         __syncthreads();
         if ( r < len )
         {
            const double vr = vv[r];
            for (int c = q; c < len; c += 4)
               a[k + 1 + r][k + 1 + c] -= vr * ww[c] + wr * vv[c];
         }
         __syncthreads();
      }
   }
   if ( lane == 0 )
      dd[n - 1] = a[n - 1][n - 1];
   __syncthreads();
   LM_T(2);
   /* smallest eigenvalue of T(dd, e2) by multisection of a Gershgorin interval; the tridiagonal matrix in registers */
   double dr[16], er[16];
#pragma unroll
   for (int i = 0; i < 16; ++i)
   {
      dr[i] = (i < n) ? dd[i] : 0.0;
      er[i] = (i + 1 < n) ? e2[i] : 0.0;
   }
   double sr[16];                                       /* |e_i|: lane i computes one, all read them */
   {
      const double ev = (lane < 16) ? er[0] : 0.0;
      double mine = 0.0;
#pragma unroll
      for (int i = 0; i < 16; ++i)
         mine = (lane == i) ? er[i] : mine;
      (void) ev;
      const double sq = (mine > 0.0) ? mine * rsqrt_nr(mine) : 0.0;
#pragma unroll
      for (int i = 0; i < 16; ++i)
         sr[i] = lm_lane(sq, i);
   }
   double lo = 1e300, hi = -1e300;
#pragma unroll
   for (int i = 0; i < 16; ++i)
   {
      if ( i < n )
      {
         const double rad = (i > 0 ? sr[i - 1] : 0.0) + sr[i];
         lo = fmin(lo, dr[i] - rad);
         hi = fmax(hi, dr[i] + rad);
      }
   }
   const double span0 = fmax(hi - lo, 1e-300);
   lo -= 1e-12 * span0 + 1e-300;
   hi += 1e-12 * span0 + 1e-300;
   const double pivmin = 1e-290;
   LM_T(3);
   /* every round narrows the interval 65-fold: 10 rounds take it below the rounding level of its ends */
   for (int round = 0; round < 10; ++round)
   {
      const double x = lo + (hi - lo) * (double) (lane + 1) / 65.0;
      int cnt = 0;                                     /* eigenvalues below x */
      double t = dr[0] - x;
      if ( fabs(t) < pivmin ) t = -pivmin;
      if ( t < 0.0 ) ++cnt;
#pragma unroll
      for (int i = 1; i < 16; ++i)
      {
         if ( i < n )
         {
            t = dr[i] - x - er[i - 1] * rcp_newton(t);
            if ( fabs(t) < pivmin ) t = -pivmin;
            if ( !(fabs(t) < 1e290) ) t = (t < 0.0) ? -1e290 : 1e290;
            if ( t < 0.0 ) ++cnt;
         }
      }
      const unsigned long long msk = __ballot(cnt >= 1);
      const int first = msk ? __ffsll((long long) msk) - 1 : 64;      /* first shift with an eigenvalue below it */
      const double w = (hi - lo) / 65.0;
      const double nlo = lo + w * (double) first;
      const double nhi = (first < 64) ? lo + w * (double) (first + 1) : hi;
      lo = nlo; hi = nhi;
      if ( hi - lo <= 2e-16 * fmax(fabs(lo), fabs(hi)) || hi - lo <= 1e-16 * span0 )
         break;
   }
   return 0.5 * (lo + hi);
}

/* n <= 16: smallest eigenvalue only, ONE wavefront, no eigenvectors; res = { lambda_min, 0 (exact: no residual bound), n } like
 * the Lanczos result.  With L given the matrix is L A L^T (the scaled step), formed here. */
/* Lin non-NULL: the matrix is L A L^T (the scaled step) */
__device__ __forceinline__ void d_lmin_tiny(int n, const double* __restrict__ Ain, const double* __restrict__ Lin, double* __restrict__ res)
{
   __shared__ double a[16][17];
   __shared__ double vv[16], ww[16], dd[16], e2[16];
   const int tid = threadIdx.x;
   LM_T(0);
   if ( Lin == NULL )
   {
      for (int e = tid; e < 256; e += 64)
      {
         const int r = e >> 4, c = e & 15;
         double v = 0.0;
         if ( r < n && c < n )
            v = 0.5 * (Ain[r * n + c] + Ain[c * n + r]);
         a[r][c] = v;
      }
   }
   else
   {
      /* W = L A L^T formed here (two 16 x 16 x 16 products) instead of by two GEMM launches */
      __shared__ double sl[16][17], st[16][17];
      for (int e = tid; e < 256; e += 64)
      {
         const int r = e >> 4, c = e & 15;
         const bool in = r < n && c < n;
         sl[r][c] = in ? Lin[r * n + c] : 0.0;
         a[r][c] = in ? Ain[r * n + c] : 0.0;
      }
      __syncthreads();
      for (int e = tid; e < 256; e += 64)
      {
         const int r = e >> 4, c = e & 15;
         double acc = 0.0;
         for (int k = 0; k < 16; ++k)
            acc += sl[r][k] * a[k][c];
         st[r][c] = acc;
      }
      __syncthreads();
      double w[4];
      for (int q = 0; q < 4; ++q)
      {
         const int e = tid + 64 * q;
         const int r = e >> 4, c = e & 15;
         double acc = 0.0;
         for (int k = 0; k < 16; ++k)
            acc += st[r][k] * sl[c][k];
         w[q] = acc;
      }
      __syncthreads();
      for (int q = 0; q < 4; ++q)
      {
         const int e = tid + 64 * q;
         st[e >> 4][e & 15] = w[q];
      }
      __syncthreads();
      for (int e = tid; e < 256; e += 64)
      {
         const int r = e >> 4, c = e & 15;
         a[r][c] = 0.5 * (st[r][c] + st[c][r]);
      }
   }
   __syncthreads();
   LM_T(1);
   const double lm = (n > 1) ? lmin_sym16(a, n, vv, ww, dd, e2, tid) : a[0][0];
   LM_T(4);
   if ( tid == 0 )
   {
      res[0] = lm;
      res[1] = 0.0;
      res[2] = (double) n;
   }
}

__global__ void __launch_bounds__(64) k_lmin_tiny(int n, const double* __restrict__ A0, const double* __restrict__ A1,
   double* __restrict__ res0, double* __restrict__ res1, const double* __restrict__ L0, const double* __restrict__ L1)
{
   d_lmin_tiny(n, blockIdx.x ? A1 : A0, blockIdx.x ? L1 : L0, blockIdx.x ? res1 : res0);
}

/* several blocks in one launch (blockIdx.y = block, blockIdx.x = side) */
__global__ void __launch_bounds__(64) k_lmin_tiny_multi(hs_step_jobs P)
{
   const int job = blockIdx.y;
   d_lmin_tiny(P.n[job], blockIdx.x ? P.D1[job] : P.D0[job], blockIdx.x ? P.L1[job] : P.L0[job], blockIdx.x ? P.res1[job] : P.res0[job]);
}

/* ---- n > 64: block Jacobi.  The matrix is cut into 32 x 32 blocks; a round of the round-robin tournament over the blocks pairs
 * them up, every pair (P, Q) is one independent 64 x 64 symmetric subproblem [[A_PP, A_PQ], [A_QP, A_QQ]] that ONE workgroup
 * (nearly) diagonalises in LDS by a few sweeps of the same parallel-order two-sided Jacobi as k_jacobi_small, leaving its
 * accumulated rotation J_k; then  A <- J^T A J,  Vt <- J^T Vt  with J = diag(J_k) are three 64-deep products per pair and
 * 64-column chunk (rows of A, rows of Vt, columns of A).  A sweep over all block pairs is nb - 1 rounds of 3 launches instead of
 * the n - 1 rounds of 2 launches of the element-wise form, and the O(n^3) part runs as dense 64 x 64 x 64 products from LDS.
 * Rows / columns beyond n are zero padding: a rotation with a zero off-diagonal entry is never formed, so the padded
 * coordinates stay unit vectors and are dropped by the final sort. */
#define BJ_B 32
#define BJ_M 64
#ifndef BJ_T
#define BJ_T 1024           /* threads of the subproblem workgroup */
#endif

/* reciprocal square root: v_rsq_f64 seed + two coupled Newton steps (no division, no sqrt expansion) */
__device__ __forceinline__ double bj_rsqrt(double x)
{
   double y = __builtin_amdgcn_rsq(x);
   double h = 0.5 * y, g = x * y;
   double r = fma(-h, g, 0.5);
   g = fma(g, r, g); h = fma(h, r, h);
   r = fma(-h, g, 0.5);
   h = fma(h, r, h);
   return 2.0 * h;
}

/* reciprocal: v_rcp_f64 seed + two Newton steps */
__device__ __forceinline__ double bj_rcp(double t)
{
   double r = __builtin_amdgcn_rcp(t);
   r = fma(fma(-t, r, 1.0), r, r);
   r = fma(fma(-t, r, 1.0), r, r);
   return r;
}

__global__ void k_bjac_pad(int n, int N, const double* __restrict__ A, double* __restrict__ Ap, double* __restrict__ Vtp)
{
   for (long long e = (long long) blockIdx.x * blockDim.x + threadIdx.x; e < (long long) N * N; e += (long long) gridDim.x * blockDim.x)
   {
      const int r = (int) (e / N), c = (int) (e % N);
      Ap[e] = (r < n && c < n) ? 0.5 * (A[(long long) r * n + c] + A[(long long) c * n + r]) : 0.0;
      Vtp[e] = (r == c) ? 1.0 : 0.0;
   }
}

__device__ __forceinline__ int bj_grow(int P, int Q, int i) { return i < BJ_B ? P * BJ_B + i : Q * BJ_B + (i - BJ_B); }

/* one workgroup per block pair (P, Q) of round r: J[pair] (64 x 64, S = J^T diag J up to what the sweeps leave).
 * cross = 1: only the 32 x 32 pairs (i in P, j in Q) are rotated, 32 rounds of 32 disjoint pairs (i, (i + round) mod 32) - over the
 * nb - 1 rounds of a sweep every cross pair of the matrix is met exactly once; cross = 0: only the pairs inside P and inside Q
 * (two 32-player tournaments side by side, 31 rounds), run once per sweep.  Together: a cyclic ordering of all n (n - 1) / 2 pairs. */
__global__ void __launch_bounds__(BJ_T) k_bjac_sub(int N, int nbp, int r, const double* __restrict__ Ap, double* __restrict__ J, int maxsw,
   int cross)
{
   __shared__ double a[BJ_M][BJ_M + 1];
   __shared__ double vt[BJ_M][BJ_M + 1];
   __shared__ double2 cs[BJ_M / 2];
   __shared__ int2 pq[BJ_M / 2];
   __shared__ double red[2 * (BJ_T / 64)];
   const int tid = threadIdx.x;
   int P, Q;
   jac_pair(nbp, r, blockIdx.x, &P, &Q);
   for (int e = tid; e < BJ_M * BJ_M; e += BJ_T)
   {
      const int i = e / BJ_M, j = e % BJ_M;
      a[i][j] = Ap[(long long) bj_grow(P, Q, i) * N + bj_grow(P, Q, j)];
      vt[i][j] = (i == j) ? 1.0 : 0.0;
   }
   __syncthreads();
   const int rounds = cross ? BJ_B : BJ_B - 1;
   for (int sw = 0; sw < maxsw; ++sw)
   {
      double off = 0.0, dg = 0.0;
      for (int e = tid; e < BJ_M * BJ_M; e += BJ_T)
      {
         const int i = e / BJ_M, j = e % BJ_M;
         const double v = a[i][j];
         if ( i == j ) dg += v * v; else off += v * v;
      }
      for (int o = 32; o > 0; o >>= 1)
      {
         off += __shfl_down(off, o, 64);
         dg += __shfl_down(dg, o, 64);
      }
      if ( (tid & 63) == 0 )
      {
         red[(tid >> 6) * 2] = off;
         red[(tid >> 6) * 2 + 1] = dg;
      }
      __syncthreads();
      off = 0.0; dg = 0.0;
      for (int w = 0; w < BJ_T / 64; ++w) { off += red[2 * w]; dg += red[2 * w + 1]; }
      __syncthreads();
      if ( !(off > 1e-32 * dg) || !(off > 0.0) )
         break;
      for (int rr = 0; rr < rounds; ++rr)
      {
         if ( tid < BJ_M / 2 )
         {
            int p, q;
            if ( cross )
            {
               p = tid;
               q = BJ_B + ((tid + rr) & (BJ_B - 1));
            }
            else
            {
               jac_pair(BJ_B, rr, tid & (BJ_B / 2 - 1), &p, &q);
               if ( tid >= BJ_B / 2 ) { p += BJ_B; q += BJ_B; }
            }
            double c = 1.0, s = 0.0;
            const double apq = a[p][q], app = a[p][p], aqq = a[q][q];
            if ( fabs(apq) > 1e-300 && fabs(apq) > 1e-19 * (fabs(app) + fabs(aqq)) )
            {
               /* the small-angle rotation from two reciprocal square roots: with d = a_qq - a_pp, b = 2 a_pq, r = hypot(d, b)
                * cos^2 = (1 + |d| / r) / 2,  sin = sgn(d) b / (2 r cos)  (scaled by 1 / max(|d|, |b|) against overflow) */
               double d = aqq - app, b = 2.0 * apq;
#ifdef BJ_DIV
               const double sc = 1.0 / fmax(fabs(d), fabs(b));
#else
               const double sc = bj_rcp(fmax(fabs(d), fabs(b)));      /* (only scales d and b against overflow: no division expansion) */
#endif
               d *= sc; b *= sc;
               const double ir = bj_rsqrt(d * d + b * b);
               const double c2 = 0.5 + 0.5 * fabs(d) * ir;
               const double ic = bj_rsqrt(c2);
               c = c2 * ic;
               s = (d >= 0.0 ? 0.5 : -0.5) * b * ir * ic;
            }
            cs[tid] = make_double2(c, s);
            pq[tid] = make_int2(p, q);
         }
         __syncthreads();
         for (int it = tid; it < (BJ_M / 2) * (BJ_M / 2); it += BJ_T)
         {
            const int k1 = it / (BJ_M / 2), k2 = it % (BJ_M / 2);
            const int2 r1 = pq[k1], r2 = pq[k2];
            const double2 g1 = cs[k1], g2 = cs[k2];
            const int p = r1.x, q = r1.y, u = r2.x, v = r2.y;
            const double c1 = g1.x, s1 = g1.y, c2 = g2.x, s2 = g2.y;
            const double apu = a[p][u], apv = a[p][v], aqu = a[q][u], aqv = a[q][v];
            const double bpu = c1 * apu - s1 * aqu, bqu = s1 * apu + c1 * aqu;
            const double bpv = c1 * apv - s1 * aqv, bqv = s1 * apv + c1 * aqv;
            a[p][u] = c2 * bpu - s2 * bpv;
            a[p][v] = s2 * bpu + c2 * bpv;
            a[q][u] = c2 * bqu - s2 * bqv;
            a[q][v] = s2 * bqu + c2 * bqv;
         }
         for (int it = tid; it < (BJ_M / 2) * BJ_M; it += BJ_T)
         {
            const int k1 = it / BJ_M, col = it % BJ_M;
            const int2 r1 = pq[k1];
            const double2 g1 = cs[k1];
            const double vp = vt[r1.x][col], vq = vt[r1.y][col];
            vt[r1.x][col] = g1.x * vp - g1.y * vq;
            vt[r1.y][col] = g1.y * vp + g1.x * vq;
         }
         __syncthreads();
      }
   }
   double* Jk = J + (long long) blockIdx.x * BJ_M * BJ_M;
   for (int e = tid; e < BJ_M * BJ_M; e += BJ_T)
      Jk[e] = vt[e / BJ_M][e % BJ_M];
}

/* mode 0 (blockIdx.z = 0: A, 1: Vt): rows [P; Q] of the matrix, columns of chunk blockIdx.y  <-  J_k * (those rows);
 * mode 1: columns [P; Q] of A, rows of chunk blockIdx.y  <-  (those columns) * J_k^T.  64 x 64 x 64 per workgroup from LDS, 4 x 4 per thread */
__global__ void __launch_bounds__(256) k_bjac_apply(int N, int nbp, int r, int mode, double* __restrict__ Ap, double* __restrict__ Vtp,
   const double* __restrict__ J)
{
   /* both operands with the summation index contiguous and an odd pitch: the fragments of the matrix instruction (lane & 15 = row
    * or column, lane >> 4 = position in the K step) are read without bank conflicts */
   __shared__ double L[BJ_M][BJ_M + 1];     /* [i][k] */
   __shared__ double Rt[BJ_M][BJ_M + 1];    /* [j][k] */
   typedef double bj_v4 __attribute__((ext_vector_type(4)));
   const int tid = threadIdx.x;
   int P, Q;
   jac_pair(nbp, r, blockIdx.x, &P, &Q);
   const double* Jk = J + (long long) blockIdx.x * BJ_M * BJ_M;
   double* M = (mode == 0 && blockIdx.z == 1) ? Vtp : Ap;
   const int c0 = blockIdx.y * BJ_M;
   {
      /* all 32 loads of a thread in flight before the first LDS write (a loop of load - write pairs was sixteen round trips) */
      double vj[16], vm[16];
#pragma unroll
      for (int q = 0; q < 16; ++q)
      {
         const int e = tid + 256 * q;
         const int i = e / BJ_M, j = e % BJ_M;
         vj[q] = Jk[e];
         vm[q] = mode == 0 ? M[(long long) bj_grow(P, Q, i) * N + c0 + j] : M[(long long) (c0 + i) * N + bj_grow(P, Q, j)];
      }
#pragma unroll
      for (int q = 0; q < 16; ++q)
      {
         const int e = tid + 256 * q;
         const int i = e / BJ_M, j = e % BJ_M;
         if ( mode == 0 )
         {
            L[i][j] = vj[q];                                                /* J[i][k] */
            Rt[j][i] = vm[q];                                               /* strip[k][j] */
         }
         else
         {
            L[i][j] = vm[q];                                                /* strip[i][k] */
            Rt[i][j] = vj[q];                                               /* (J^T)[k][j] = J[j][k] */
         }
      }
   }
   __syncthreads();
   /* wavefront w: rows 16 w .. 16 w + 15 of the product, four column tiles (four independent accumulator chains), K = 64 in 16 steps
    * of v_mfma_f64_16x16x4 (the 64 x 64 x 64 product was 1024 multiply-adds per thread on the vector pipe: 12.7 us per launch) */
   const int lane = tid & 63, w = tid >> 6, lr = lane & 15, lk = lane >> 4;
   bj_v4 acc[4];
#pragma unroll
   for (int t = 0; t < 4; ++t)
      acc[t] = (bj_v4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
   for (int ks = 0; ks < 16; ++ks)
   {
      const double a = L[16 * w + lr][4 * ks + lk];
#pragma unroll
      for (int t = 0; t < 4; ++t)
      {
         const double bb = Rt[16 * t + lr][4 * ks + lk];
         acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc[t], 0, 0, 0);
      }
   }
#pragma unroll
   for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
      {
         const int i = 16 * w + lk + 4 * rr, j = 16 * t + lr;
         if ( mode == 0 )
            M[(long long) bj_grow(P, Q, i) * N + c0 + j] = acc[t][rr];
         else
            M[(long long) (c0 + i) * N + bj_grow(P, Q, j)] = acc[t][rr];
      }
}

/* off-diagonal and diagonal square sums, many workgroups: part[2 b], part[2 b + 1] */
__global__ void __launch_bounds__(256) k_bjac_offnorm(int N, const double* __restrict__ A, double* __restrict__ part)
{
   __shared__ double red[8];
   double off = 0.0, dg = 0.0;
   for (long long e = (long long) blockIdx.x * blockDim.x + threadIdx.x; e < (long long) N * N; e += (long long) gridDim.x * blockDim.x)
   {
      const double v = A[e];
      if ( e / N == e % N ) dg += v * v; else off += v * v;
   }
   for (int o = 32; o > 0; o >>= 1)
   {
      off += __shfl_down(off, o, 64);
      dg += __shfl_down(dg, o, 64);
   }
   if ( (threadIdx.x & 63) == 0 )
   {
      red[(threadIdx.x >> 6) * 2] = off;
      red[(threadIdx.x >> 6) * 2 + 1] = dg;
   }
   __syncthreads();
   if ( threadIdx.x == 0 )
   {
      part[2 * blockIdx.x] = red[0] + red[2] + red[4] + red[6];
      part[2 * blockIdx.x + 1] = red[1] + red[3] + red[5] + red[7];
   }
}

/* ascending rank sort of the first n diagonal entries of the padded matrix, rows of Vt permuted and cut to n columns */
__global__ void __launch_bounds__(256) k_bjac_sort(int n, int N, const double* __restrict__ Ap, const double* __restrict__ Vtp,
   double* __restrict__ lam, double* __restrict__ V)
{
   __shared__ int rank_s;
   const int i = blockIdx.x;
   const double di = Ap[(long long) i * N + i];
   if ( threadIdx.x == 0 ) rank_s = 0;
   __syncthreads();
   int cnt = 0;
   for (int j = threadIdx.x; j < n; j += blockDim.x)
   {
      const double dj = Ap[(long long) j * N + j];
      if ( dj < di || (dj == di && j < i) )
         ++cnt;
   }
   atomicAdd(&rank_s, cnt);
   __syncthreads();
   const int rank = rank_s;
   if ( threadIdx.x == 0 ) lam[rank] = di;
   if ( V != NULL )
      for (int c = threadIdx.x; c < n; c += blockDim.x)
         V[(long long) rank * n + c] = Vtp[(long long) i * N + c];
}

/* perm[rank of the i-th diagonal entry among the first n, descending] = i */
__global__ void __launch_bounds__(256) k_bjac_rank(int n, int N, const double* __restrict__ Ap, int* __restrict__ perm)
{
   __shared__ int rank_s;
   const int i = blockIdx.x;
   const double di = Ap[(long long) i * N + i];
   if ( threadIdx.x == 0 ) rank_s = 0;
   __syncthreads();
   int cnt = 0;
   for (int j = threadIdx.x; j < n; j += blockDim.x)
   {
      const double dj = Ap[(long long) j * N + j];
      if ( dj > di || (dj == di && j < i) )
         ++cnt;
   }
   atomicAdd(&rank_s, cnt);
   __syncthreads();
   if ( threadIdx.x == 0 ) perm[rank_s] = i;
}

/* A2 = P A P^T, Vt2 = P Vt on the first n coordinates (the padding stays where it is) */
__global__ void k_bjac_permute(int n, int N, const int* __restrict__ perm, const double* __restrict__ Ap, const double* __restrict__ Vtp,
   double* __restrict__ A2, double* __restrict__ Vt2)
{
   for (long long e = (long long) blockIdx.x * blockDim.x + threadIdx.x; e < (long long) N * N; e += (long long) gridDim.x * blockDim.x)
   {
      const int r = (int) (e / N), c = (int) (e % N);
      const int pr = r < n ? perm[r] : r, pc = c < n ? perm[c] : c;
      A2[e] = Ap[(long long) pr * N + pc];
      Vt2[e] = Vtp[(long long) pr * N + c];
   }
}

#define BJ_OFFBLOCKS 256
/* the convergence test of a sweep reads two sums back: into PINNED memory of the calling thread.  [Into a stack array the copy is
 * staged by the runtime and blocks: 150-290 us per sweep where the kernels of a sweep at n = 200 take 450 - a quarter of every
 * decomposition above 128 rows, 3 of the 13.6 ms of a PSD projection at n = 200.] */
static double* bj_readback_buffer()
{
   static thread_local double* buf = NULL;
   if ( buf == NULL )
   {
      void* p = NULL;
      if ( hipHostMalloc(&p, 2 * BJ_OFFBLOCKS * sizeof(double), hipHostMallocPortable) != hipSuccess )
         return NULL;
      buf = static_cast<double*>(p);
   }
   return buf;
}
static int bj_padded(int n) { const int nb = (n + BJ_B - 1) / BJ_B; return ((nb + 1) & ~1) * BJ_B; }

long long hs_syev_ws(int n)
{
   const long long N = bj_padded(n);
   return 4 * N * N + (N / BJ_M) * BJ_M * BJ_M + 2 * BJ_OFFBLOCKS + N + 64;
}

int hs_syev_jacobi(hipStream_t s, int n, double* A, double* lam, double* V, int* info, double* ws)
{
   if ( n <= 0 )
      return HS_ERR_ARG;
   if ( n <= JS )
   {
      hipLaunchKernelGGL(k_jacobi_small, dim3(1), dim3(256), 0, s, n, A, lam, V, info);
      HS_LAUNCH_CHECK();
      return HS_OK;
   }
   if ( getenv("HIPSDP_JACOBI_ELEMENTWISE") == NULL )
   {
      const int N = bj_padded(n), nbp = N / BJ_B;
      double* Ap = ws;
      double* Vtp = Ap + (long long) N * N;
      double* J = Vtp + (long long) N * N;
      double* part = J + (long long) (nbp / 2) * BJ_M * BJ_M;
      double* Ap2 = part + 2 * BJ_OFFBLOCKS + 32;                     /* second pair of arrays: target of the permutations below */
      double* Vtp2 = Ap2 + (long long) N * N;
      int* perm = reinterpret_cast<int*>(Vtp2 + (long long) N * N);
      long long pg = ((long long) N * N + 255) / 256; if ( pg > 4096 ) pg = 4096;
      hipLaunchKernelGGL(k_bjac_pad, dim3((unsigned) pg), dim3(256), 0, s, n, N, A, Ap, Vtp);
      HS_LAUNCH_CHECK();
      /* the update of a sweep leaves rounding noise of about N eps relative to the diagonal: below that nothing more is gained */
      double tol = 0.25 * (double) N * 2.2e-16; tol = tol * tol;
      if ( tol < 1e-30 ) tol = 1e-30;
      double prev = 1e300;
      int sweeps_b = 0;
      int nperm = 0, lastperm = -3;
      const bool bjsort = !(getenv("HIPSDP_BJ_SORT") != NULL && getenv("HIPSDP_BJ_SORT")[0] == '0');
      int inner0 = 1, inner = 1;               /* inner sweeps of a subproblem: first outer sweep (dense subproblems), later ones */
      if ( getenv("HIPSDP_BJ_INNER") != NULL )
         (void) sscanf(getenv("HIPSDP_BJ_INNER"), "%d,%d", &inner0, &inner);
      double* const h = bj_readback_buffer();
      if ( h == NULL )
         return HS_ERR_HIP;
      for (sweeps_b = 0; sweeps_b < 40; ++sweeps_b)
      {
         hipLaunchKernelGGL(k_bjac_offnorm, dim3(BJ_OFFBLOCKS), dim3(256), 0, s, N, Ap, part);
         HS_LAUNCH_CHECK();
         HS_HIP( hipMemcpyAsync(h, part, 2 * BJ_OFFBLOCKS * sizeof(double), hipMemcpyDeviceToHost, s) );
         HS_HIP( hipStreamSynchronize(s) );
         double off = 0.0, dg = 0.0;
         for (int b = 0; b < BJ_OFFBLOCKS; ++b) { off += h[2 * b]; dg += h[2 * b + 1]; }
         if ( getenv("HIPSDP_JACOBI_VERBOSE") != NULL )
            fprintf(stderr, "block jacobi n=%d sweep %d: off^2 / diag^2 = %.3e (tol %.1e)\n", n, sweeps_b, off / dg, tol);
         if ( !(off > tol * dg) || !(off > 0.0) )
            break;
         if ( off <= 1e-24 * dg && off > 0.25 * prev )
            break;                                      /* at the noise floor: no longer shrinking */
         const double prev0 = prev;
         prev = off;
         /* Multiple eigenvalues (the n - rank equal ones of a low-rank matrix - what the PSD projection of a warm start meets): the
          * cyclic method converges quadratically only when the diagonal entries that belong to one eigenvalue sit next to each other;
          * scattered among the others they made the tail linear (off^2 / diag^2 from 1e-17 down by 0.6 per sweep: 25 sweeps at
          * n = 200, the limit of 40 at n = 500 with 1e-11 still off the diagonal).  So when the diagonal entries are close to the
          * eigenvalues and a sweep has NOT brought the quadratic drop (a matrix with separated eigenvalues never comes here), the
          * coordinates are sorted by their diagonal entries, and once more three sweeps later if it is still slow:
          * A <- P A P^T, Vt <- P Vt.  What remains behind that is the second convergence phase every Jacobi method has on such a
          * matrix - the perturbation inside the eigenspace is a dense matrix of its own - at 0.1 per sweep instead of 0.6. */
         if ( bjsort && nperm < 2 && off <= 1e-8 * dg && off > 1e-2 * prev0 && sweeps_b >= lastperm + 3 )
         {
            lastperm = sweeps_b;
            hipLaunchKernelGGL(k_bjac_rank, dim3(n), dim3(256), 0, s, n, N, Ap, perm);
            hipLaunchKernelGGL(k_bjac_permute, dim3((unsigned) pg), dim3(256), 0, s, n, N, perm, Ap, Vtp, Ap2, Vtp2);
            HS_LAUNCH_CHECK();
            double* t = Ap; Ap = Ap2; Ap2 = t;
            t = Vtp; Vtp = Vtp2; Vtp2 = t;
            ++nperm;
         }
         for (int r = -1; r < nbp - 1; ++r)
         {
            /* r = -1: the pairs inside the blocks (pairing of round 0); r >= 0: the cross pairs of the block pairs of round r */
            const int rr = r < 0 ? 0 : r;
            hipLaunchKernelGGL(k_bjac_sub, dim3(nbp / 2), dim3(BJ_T), 0, s, N, nbp, rr, Ap, J, sweeps_b == 0 ? inner0 : inner, r < 0 ? 0 : 1);
            hipLaunchKernelGGL(k_bjac_apply, dim3(nbp / 2, N / BJ_M, 2), dim3(256), 0, s, N, nbp, rr, 0, Ap, Vtp, J);
            hipLaunchKernelGGL(k_bjac_apply, dim3(nbp / 2, N / BJ_M, 1), dim3(256), 0, s, N, nbp, rr, 1, Ap, Vtp, J);
         }
         HS_LAUNCH_CHECK();
      }
      hipLaunchKernelGGL(k_bjac_sort, dim3(n), dim3(256), 0, s, n, N, Ap, Vtp, lam, V);
      HS_LAUNCH_CHECK();
      if ( info != NULL )
         HS_HIP( hipMemsetAsync(info, 0, sizeof(int), s) );
      return HS_OK;
   }
   double* Vt = ws;                              /* n x n */
   double* rot = Vt + (long long) n * n;         /* 2 * ceil(n/2) */
   double* nrm = rot + 2LL * ((n + 1) / 2) + 2;  /* 2 */
   HS_CALL( hs_symmetrize(s, A, n) );
   HS_CALL( hs_set_identity(s, Vt, n, 1.0) );
   const int np = (n + 1) & ~1;
   const int half = np / 2;
   int sweeps = 0;
   if ( n > 1 )
   {
      const long long work = 2LL * half * half;
      int grid = (int) ((work + 255) / 256);
      if ( grid > 4096 ) grid = 4096;
      double* const h = bj_readback_buffer();
      if ( h == NULL )
         return HS_ERR_HIP;
      for (sweeps = 0; sweeps < 30; ++sweeps)
      {
         hipLaunchKernelGGL(k_jacobi_offnorm, dim3(1), dim3(1024), 0, s, n, A, nrm);
         HS_LAUNCH_CHECK();
         HS_HIP( hipMemcpyAsync(h, nrm, 2 * sizeof(double), hipMemcpyDeviceToHost, s) );
         HS_HIP( hipStreamSynchronize(s) );
         if ( !(h[0] > 1e-30 * h[1]) || !(h[0] > 0.0) )
            break;
         for (int r = 0; r < np - 1; ++r)
         {
            hipLaunchKernelGGL(k_jacobi_angles, dim3((half + 255) / 256), dim3(256), 0, s, n, np, r, A, rot);
            hipLaunchKernelGGL(k_jacobi_apply, dim3(grid), dim3(256), 0, s, n, np, r, A, Vt, rot);
         }
         HS_LAUNCH_CHECK();
      }
   }
   hipLaunchKernelGGL(k_jacobi_sort, dim3(1), dim3(1024), 0, s, n, A, Vt, lam, V);
   HS_LAUNCH_CHECK();
   if ( info != NULL )
      HS_HIP( hipMemsetAsync(info, 0, sizeof(int), s) );
   (void) sweeps;
   return HS_OK;
}
