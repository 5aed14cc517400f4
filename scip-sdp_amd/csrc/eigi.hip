/* eigi.hip - i-th eigenvalue (and eigenvector) of a small symmetric matrix in ONE launch without a copy engine.
 *
 * Reference: SCIPlapackComputeIthEigenvalue (src/sdpi/lapack_interface.c:178-288: DSYEVR with RANGE = 'I', IL = IU = i) as the
 * callers use it - cons_sdp.c asks for the smallest eigenvalue of blocks of 2-50 rows dozens of times per node (feasibility
 * checks, eigenvector cuts), solveonevarsdp.c inside its Newton iteration.  A full Jacobi decomposition plus hipMalloc / hipMemcpy
 * / hipFree per call costs milliseconds there; LAPACK on the host tens of microseconds.  Here, for n <= 64:
 *   - the matrix goes host -> pinned, device-mapped staging memory (one memcpy of n^2 doubles), the kernel reads it from there and
 *     writes eigenvalue, eigenvector and a sequence number back to mapped memory; the host polls the number: no hipMalloc, no
 *     hipMemcpy, no stream synchronisation on the path;
 *   - one workgroup: Householder tridiagonalisation in LDS (the DSYTD2 recurrence: v, p = tau A v, w = p - (tau/2)(p.v) v,
 *     A -= v w^T + w v^T; four barriers per column), Sturm-count multisection for exactly the i-th eigenvalue (64 shifts per
 *     round, one per lane), inverse iteration on the tridiagonal matrix and back-transformation through the reflectors for the
 *     eigenvector - what DSYEVR does for one eigenpair.
 * Larger matrices keep the block-Jacobi path (eig.hip). */
#include "hs_common.h"
#include "hs_kernels.h"
#include "hs_lds_product.h"
#include "../../include/hipsdp.h"
#include <cstring>
#include <cmath>

#define EI_N  64
#define EI_LD 65

namespace {

/* lane exchange inside a row of 16 lanes on the data-parallel-primitive path (no LDS crossbar round trip as with ds_bpermute):
 * CTRL = quad_perm / row_half_mirror / row_mirror pattern */
template<int CTRL>
__device__ __forceinline__ double ei_dpp(double v)
{
   int lo = __double2loint(v), hi = __double2hiint(v);
   lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
   hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
   return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double ei_lane(double v, int l)
{
   const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
   return __hiloint2double(hi, lo);
}

/* sum over the 64 lanes, result in every lane: four butterfly steps inside the rows of 16 (lane ^ 1, lane ^ 2, mirror of 8,
 * mirror of 16), then the four row sums are read as scalars and added in a fixed order */
__device__ __forceinline__ double ei_wsum(double v)
{
   v += ei_dpp<0xB1>(v);              /* quad_perm [1, 0, 3, 2] */
   v += ei_dpp<0x4E>(v);              /* quad_perm [2, 3, 0, 1] */
   v += ei_dpp<0x141>(v);             /* row_half_mirror */
   v += ei_dpp<0x140>(v);             /* row_mirror */
   return ((ei_lane(v, 0) + ei_lane(v, 16)) + ei_lane(v, 32)) + ei_lane(v, 48);
}

/* sum over the 16 lanes of a row, result in every lane of the row */
__device__ __forceinline__ double ei_sum16(double v)
{
   v += ei_dpp<0xB1>(v);
   v += ei_dpp<0x4E>(v);
   v += ei_dpp<0x141>(v);
   v += ei_dpp<0x140>(v);
   return v;
}

__device__ __forceinline__ double ei_quad(double x)
{
   x += ei_dpp<0xB1>(x);
   x += ei_dpp<0x4E>(x);
   return x;
}

/* reciprocal and reciprocal square root to full precision (v_rcp_f64 / v_rsq_f64 and two Newton steps): the scalars of a
 * reflector without the division and square-root expansions */
__device__ __forceinline__ double ei_rcp2(double t)
{
   double r = __builtin_amdgcn_rcp(t);
   r = fma(fma(-t, r, 1.0), r, r);
   r = fma(fma(-t, r, 1.0), r, r);
   return r;
}
__device__ __forceinline__ double ei_rsqrt(double x)
{
   double y = __builtin_amdgcn_rsq(x);
   double h = 0.5 * y, g = x * y;
   double r = fma(-h, g, 0.5);
   g = fma(g, r, g); h = fma(h, r, h);
   r = fma(-h, g, 0.5);
   h = fma(h, r, h);
   return 2.0 * h;
}

__device__ __forceinline__ double ei_rcp(double t)
{
   double r = __builtin_amdgcn_rcp(t);
   r = fma(fma(-t, r, 1.0), r, r);
   return r;
}

/* in: n x n symmetric, in mapped host memory, the triangle at memory positions [j n + i], i >= j, is read; out[0] = eigenvalue, out[1 .. n] = eigenvector,
 * then the sequence number is stored to *flag (system scope) */
/* ALL = true: every eigenpair (out[0 .. n - 1] = eigenvalues ascending, out[EI_N + k n + i] = component i of eigenvector k), what
 * DSYEVR computes for RANGE = 'A' (lapack_interface.c:507-603): the tridiagonal matrix of the same reduction, then
 *   - all eigenvalues at once by Sturm-count multisection, four shifts per eigenvalue and round (thread = (eigenvalue, shift));
 *   - all eigenvectors by inverse iteration on the tridiagonal matrix, one THREAD per eigenvector (Gaussian elimination with partial
 *     pivoting redone in every one of the three iterations, its U factor in LDS, [row][thread] so that the threads of a wavefront
 *     touch consecutive words), start vectors that differ from eigenvector to eigenvector;
 *   - eigenvectors of eigenvalues closer than 1e-3 ||T|| (DSTEIN's criterion) are orthogonalised against each other (classical
 *     Gram-Schmidt twice, one wavefront per cluster): exactly degenerate eigenvalues get an orthonormal basis of their space out of
 *     the different start vectors;
 *   - back-transformation through the reflectors, one wavefront per vector.
 * Dynamic LDS (ALL only): Z[64][64] (component-major: Z[i * 64 + k] = component i of vector k) and two [64][64] factor arrays (pivot
 * reciprocals and the first superdiagonal of U; its second superdiagonal is e[i + 1] in the rows that were swapped and 0 elsewhere:
 * one bit per row in a register). */
/* Number of eigenvalues below x of the symmetric tridiagonal matrix scaled to norm <= 1: ds[i] = d_i / norm, es[i] = (e_i / norm)^2,
 * both padded behind the matrix (ds: eight entries 4.0, es: zeros from n - 1 on: rows without coupling that cannot change a sign
 * while |x| <= 1); nb = (n - 1 + 3) >> 2 blocks of four steps.  Sturm sequence in PRODUCT form: p_0 = 1, p_1 = d_0 - x,
 * p_{i+1} = (d_i - x) p_i - e_{i-1}^2 p_{i-1}; a sign change = an eigenvalue below x, a zero takes the sign opposite to its
 * predecessor; rescaled every fourth step; the entries of the next block are on their way while the four steps of this one run.
 * Two dependent operations per step where the quotient form t_i = d_i - x - e_{i-1}^2 / t_{i-1} has a division: 260 cycles per step
 * (the division in double precision is a chain of a dozen dependent instructions) against about 60. */
__device__ __forceinline__ int ei_sturm_count(const double* ds, const double* es, int nb, double x)
{
   double pp_ = 1.0, pc = ds[0] - x;
   if ( pc == 0.0 ) pc = -1e-290;
   bool posc = pc > 0.0;
   int cnt = posc ? 0 : 1;
   double dn[4], en[4];
#pragma unroll
   for (int u = 0; u < 4; ++u)
   {
      dn[u] = ds[1 + u];
      en[u] = es[u];
   }
   for (int b = 0; b < nb; ++b)
   {
      double dc[4], ec[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
      {
         dc[u] = dn[u];
         ec[u] = en[u];
      }
      const int nx = (b + 1 < nb) ? 5 + 4 * b : 1;
#pragma unroll
      for (int u = 0; u < 4; ++u)
      {
         dn[u] = ds[nx + u];
         en[u] = es[nx - 1 + u];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
      {
         double pn = fma(dc[u] - x, pc, -ec[u] * pp_);
         if ( pn == 0.0 ) pn = -copysign(1e-290, pc);
         const bool posn = pn > 0.0;
         cnt += (posn != posc) ? 1 : 0;
         pp_ = pc; pc = pn; posc = posn;
      }
      const int ex = -max(__builtin_amdgcn_frexp_exp(pc), __builtin_amdgcn_frexp_exp(pp_));
      pc = ldexp(pc, ex);
      pp_ = ldexp(pp_, ex);
   }
   return cnt;
}

#define EI_ALL_LDS ((EI_N * EI_N + 2 * EI_N * EI_N) * (int) sizeof(double))
/* the body (flag == NULL: no sequence number, no system-scope fence - the caller is another kernel of the engine, see
 * k_lmin_exact_multi below; in may then point into LDS) */
template<bool ALL>
__device__ __forceinline__ void d_syevi_small(int n, int ith, int wantvec, const double* in, double* out,
   unsigned long long seq, unsigned long long* flag, double mtol = 2e-16)
{
   extern __shared__ __attribute__((aligned(16))) double ei_dyn[];
   __shared__ double a[EI_N][EI_LD];
   __shared__ double vv[EI_N], pp[EI_N + 8], ww[EI_N], tau[EI_N], d[EI_N], e[EI_N], e2[EI_N], zz[EI_N], xc[EI_N + 8];
   __shared__ double wk[4][EI_N], swp[EI_N];
   __shared__ double sc[4];
   const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
   const int r = tid >> 2, q = tid & 3;

   for (int idx = tid; idx < n * n; idx += 256)
   {
      const int i = idx / n, j = idx - i * n;
      /* DSYEVR is called with UPLO = 'L' on a column-major array (lapack_interface.c:215): it reads memory [j n + i], i >= j */
      const double v = (j <= i) ? in[(long long) j * n + i] : in[(long long) i * n + j];
      a[i][j] = v;
   }
   __syncthreads();

   /* ---- tridiagonalisation: Q^T A Q = T, reflector k acts on rows / columns k + 1 .. n - 1.  The matrix lives in REGISTERS:
    * thread (r, q) holds row r, columns q + 4 j (16 values); the vectors v and w of a step are kept in LDS under GLOBAL indices
    * with zeros at the indices the step does not touch, so that the matrix-vector product and the rank-2 update are straight
    * 16-term loops without predicates and without LDS traffic for the matrix (the LDS form read and wrote every entry of the
    * trailing block twice per column: 2.4 us per column at n = 64).  Column k leaves the registers through a select chain
    * (uniform register index) into xc[]; the reflector and the vector w are computed by EVERY wavefront for itself (64-element
    * reductions; all write the same values to the same LDS words), two barriers per column.  The reflectors go to the LDS
    * array a[][] (column k below the subdiagonal) for the back-transformation of an eigenvector. */
   double ar[16];
#pragma unroll
   for (int j = 0; j < 16; ++j)
      ar[j] = (r < n && q + 4 * j < n) ? a[r][q + 4 * j] : 0.0;
   if ( tid < EI_N )
   {
      vv[tid] = 0.0;                                   /* vg: v under global indices */
      ww[tid] = 0.0;                                   /* wg */
   }
   /* column 0 */
   {
      double pick = ar[0];
      if ( q == 0 )
         xc[r] = pick;                                 /* the current column (global row index) */
   }
   __syncthreads();
   for (int k = 0; k + 1 < n; ++k)
   {
      const int len = n - k - 1;                       /* length of x = A[k + 1 .., k] */
      double t;
      {
         const double xi = (lane < len) ? xc[k + 1 + lane] : 0.0;
         const double x0 = ei_lane(xi, 0);
         const double s2 = ei_wsum(lane >= 1 ? xi * xi : 0.0);
         double beta = x0, scale = 0.0;
         t = 0.0;
         if ( s2 > 0.0 )
         {
            const double h2 = x0 * x0 + s2;
            beta = -copysign(h2 * ei_rsqrt(h2), x0);
            t = (beta - x0) * ei_rcp2(beta);
            scale = ei_rcp2(x0 - beta);
         }
         if ( lane < len )
            vv[k + 1 + lane] = (lane == 0) ? 1.0 : xi * scale;
         if ( lane == 0 )
         {
            vv[k] = 0.0;
            ww[k] = 0.0;
            tau[k] = t;
            e[k] = beta;
            d[k] = xc[k];
         }
         __builtin_amdgcn_s_waitcnt(0xc07f);             /* lgkmcnt(0): this wavefront's own LDS writes are done */
         __builtin_amdgcn_wave_barrier();
      }
      if ( t != 0.0 )
      {
         /* p = tau A v over rows > k (v is zero up to k) */
         double vq[16];
#pragma unroll
         for (int j = 0; j < 16; ++j)
            vq[j] = vv[q + 4 * j];
         double acc = 0.0;
#pragma unroll
         for (int j = 0; j < 16; ++j)
            acc += ar[j] * vq[j];
         acc = ei_quad(acc);
         if ( q == 0 )
            pp[r] = (r > k) ? t * acc : 0.0;
         __syncthreads();
         {
            const double pl = lane < len ? pp[k + 1 + lane] : 0.0, vl = lane < len ? vv[k + 1 + lane] : 0.0;
            const double pv = ei_wsum(pl * vl);
            const double al = -0.5 * t * pv;
            if ( lane < len )
               ww[k + 1 + lane] = pl + al * vl;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
         }
         {
            const double vr = vv[r], wr = ww[r];
#pragma unroll
            for (int j = 0; j < 16; ++j)
               ar[j] -= vr * ww[q + 4 * j] + wr * vq[j];
         }
         /* keep the reflector (v_0 = 1 implied) in the column it annihilated */
         if ( tid < len )
            a[k + 1 + tid][k] = vv[k + 1 + tid];
      }
      else
         __syncthreads();                                /* (t is the same in every wavefront) all have read the column */
      /* the next column leaves the registers (its owner lanes: q == (k + 1) & 3, register (k + 1) >> 2) */
      {
         const int jn = (k + 1) >> 2;
         double pick = 0.0;
#pragma unroll
         for (int j = 0; j < 16; ++j)
            pick = (j == jn) ? ar[j] : pick;
         if ( q == ((k + 1) & 3) )
            xc[r] = pick;
      }
      __syncthreads();
   }
   if ( tid == 0 )
   {
      d[n - 1] = xc[n - 1];
      e[n - 1] = 0.0;
   }
   __syncthreads();
   if ( ALL )
   {
      double* Z = ei_dyn;                              /* [i][k] */
      double* f0 = Z + EI_N * EI_N;                    /* [i][k]: 1 / pivot of row i of the elimination for vector k */
      double* f1 = f0 + EI_N * EI_N;                   /* first superdiagonal of U */
      if ( tid < n )
         e2[tid] = e[tid] * e[tid];
      __syncthreads();
      /* ---- all eigenvalues: thread (k = tid >> 2, s = tid & 3) */
      double glo = 1e300, ghi = -1e300, tnorm = 0.0;
      for (int i = 0; i < n; ++i)
      {
         const double rad = (i > 0 ? fabs(e[i - 1]) : 0.0) + (i + 1 < n ? fabs(e[i]) : 0.0);
         glo = fmin(glo, d[i] - rad);
         ghi = fmax(ghi, d[i] + rad);
         tnorm = fmax(tnorm, fabs(d[i]) + rad);
      }
      const double span0 = fmax(ghi - glo, 1e-300);
      glo -= 1e-12 * span0 + 1e-300;
      ghi += 1e-12 * span0 + 1e-300;
      const double pivmin = 1e-290;
      {
         const int k = tid >> 2, sh = tid & 3;
         double lo = glo, hi = ghi;
         for (int round = 0; round < 40; ++round)
         {
            const double w = (hi - lo) * 0.2;
            const double x = lo + w * (double) (sh + 1);
            int cnt = 0;
            if ( k < n )
            {
               double t = d[0] - x;
               if ( fabs(t) < pivmin ) t = -pivmin;
               if ( t < 0.0 ) ++cnt;
               for (int i = 1; i < n; ++i)
               {
                  t = d[i] - x - e2[i - 1] * ei_rcp(t);
                  if ( fabs(t) < pivmin ) t = -pivmin;
                  if ( !(fabs(t) < 1e290) ) t = (t < 0.0) ? -1e290 : 1e290;
                  if ( t < 0.0 ) ++cnt;
               }
            }
            /* number of the four shifts with fewer than k + 1 eigenvalues below them = index of the subinterval that holds
             * eigenvalue k (the counts are monotone in the shift) */
            int below = (cnt < k + 1) ? 1 : 0;
            below += __builtin_amdgcn_update_dpp(0, below, 0xB1, 0xf, 0xf, true);
            below += __builtin_amdgcn_update_dpp(0, below, 0x4E, 0xf, 0xf, true);
            const double nlo = lo + w * (double) below;
            const double nhi = (below < 4) ? lo + w * (double) (below + 1) : hi;
            lo = nlo; hi = nhi;
            if ( __all(k >= n || hi - lo <= 2e-16 * fmax(fabs(lo), fabs(hi))) )
               break;
         }
         if ( k < n && sh == 0 )
            zz[k] = 0.5 * (lo + hi);                   /* eigenvalue k */
      }
      __syncthreads();
      if ( tid < n )
         out[tid] = zz[tid];
      /* ---- eigenvectors: three rounds of { one step of inverse iteration for every vector (thread k owns vector k),
       * orthogonalisation inside the clusters }.  Eigenvalue k belongs to the cluster of k - 1
       * when they are closer than 1e-3 ||T|| (DSTEIN's criterion).  The orthogonalisation has to happen in EVERY round, as in DSTEIN:
       * the eigenvalues of a cluster differ by a few ulps, so do the amplifications 1 / (lambda_j - theta_k) of its directions, and
       * vectors that are only orthogonalised at the end have collapsed onto each other by then (seen as 7e-10 in V V^T - I). */
      const double ortol = 1e-3 * fmax(tnorm, 1e-300);
      if ( n > 1 )
      {
         /* start vectors: a different one for every eigenvector (entries in [0.5, 1.5) from a hash of (i, k)) */
         for (int idx = tid; idx < n * n; idx += 256)
         {
            const int i = idx / n, k = idx - i * n;
            unsigned h = (unsigned) (i * 2654435761u) ^ (unsigned) ((k + 1) * 40503u);
            h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
            Z[i * EI_N + k] = 0.5 + (double) (h & 0xFFFF) * (1.0 / 65536.0);
         }
      }
      else if ( tid == 0 )
         Z[0] = 1.0;
      __syncthreads();
      for (int iter = 0; iter < 3 && n > 1; ++iter)
      {
         {
            const int k = tid;                              /* thread k owns vector k */
            if ( k < n )
            {
               const double theta = zz[k];
               const double tiny = 1e-14 * fmax(span0, fmax(fabs(theta), 1e-300));
               /* forward elimination of (T - theta I) with partial pivoting, applied to the right-hand side on the way */
               double dd = d[0] - theta, du = e[0];
               double cur = Z[k];
               unsigned long long swapped = 0ULL;           /* bit i: rows i and i + 1 were exchanged */
               for (int i = 0; i < n - 1; ++i)
               {
                  const double dl = e[i];
                  const double dn = d[i + 1] - theta;
                  const double un = (i + 2 < n) ? e[i + 1] : 0.0;
                  const double nxt = Z[(i + 1) * EI_N + k];
                  /* a subdiagonal entry at rounding level (the tridiagonal matrix of a matrix with few distinct eigenvalues splits
                   * into small blocks) must not become a pivot: 1 / e would amplify that block by 1e16 and more */
                  if ( fabs(dd) >= fabs(dl) || fabs(dl) < tiny )
                  {
                     if ( fabs(dd) < tiny ) dd = tiny;
                     const double rinv = ei_rcp2(dd);
                     const double mlt = dl * rinv;
                     f0[i * EI_N + k] = rinv; f1[i * EI_N + k] = du;
                     Z[i * EI_N + k] = cur;
                     cur = nxt - mlt * cur;
                     dd = dn - mlt * du;
                     du = un;
                  }
                  else
                  {
                     const double rinv = ei_rcp2(dl);
                     const double mlt = dd * rinv;
                     f0[i * EI_N + k] = rinv; f1[i * EI_N + k] = dn;
                     swapped |= 1ULL << i;
                     Z[i * EI_N + k] = nxt;
                     cur = cur - mlt * nxt;
                     dd = du - mlt * dn;
                     du = -mlt * un;
                  }
               }
               if ( fabs(dd) < tiny ) dd = tiny;
               double x1 = cur * ei_rcp2(dd), x2 = 0.0;
               double nrm = x1 * x1;
               Z[(n - 1) * EI_N + k] = x1;
               for (int i = n - 2; i >= 0; --i)
               {
                  const double u2 = ((swapped >> i) & 1ULL) ? ((i + 2 < n) ? e[i + 1] : 0.0) : 0.0;
                  const double xi = (Z[i * EI_N + k] - f1[i * EI_N + k] * x1 - u2 * x2) * f0[i * EI_N + k];
                  Z[i * EI_N + k] = xi;
                  nrm += xi * xi;
                  x2 = x1; x1 = xi;
                  if ( !(nrm < 1e280) )
                  {
                     /* rescale on the way (the solution of a nearly singular system is huge by construction) */
                     const double sc1 = 1e-140;
                     for (int j = i; j < n; ++j)
                        Z[j * EI_N + k] *= sc1;
                     x1 *= sc1; x2 *= sc1; nrm *= sc1 * sc1;
                  }
               }
               double rn = ei_rsqrt(fmax(nrm, 1e-300));
               if ( !(nrm > 0.0) || !(nrm < 1e300) )
               {
                  for (int i = 0; i < n; ++i)
                     Z[i * EI_N + k] = (i == k) ? 1.0 : 0.0;
                  rn = 1.0;
               }
               for (int i = 0; i < n; ++i)
                  Z[i * EI_N + k] *= rn;
            }
            __syncthreads();
         }
         /* clusters: one wavefront per cluster (cluster c goes to wavefront c mod 4), lane = component; classical Gram-Schmidt
          * against the vectors of the cluster before it, twice */
         {
            int cl = -1;
            int k0 = 0;
            while ( k0 < n )
            {
               int k1 = k0 + 1;
               while ( k1 < n && zz[k1] - zz[k1 - 1] <= ortol )
                  ++k1;
               ++cl;
               if ( k1 - k0 > 1 && (cl & 3) == wave )
               {
                  for (int k = k0; k < k1; ++k)
                  {
                     double v = (lane < n) ? Z[lane * EI_N + k] : 0.0;
                     for (int pass = 0; pass < 2; ++pass)
                        for (int p = k0; p < k; ++p)
                        {
                           const double u = (lane < n) ? Z[lane * EI_N + p] : 0.0;
                           v -= ei_wsum(u * v) * u;
                        }
                     const double nr = ei_wsum(v * v);
                     v *= ei_rsqrt(fmax(nr, 1e-300));
                     if ( lane < n )
                        Z[lane * EI_N + k] = v;
                     __builtin_amdgcn_s_waitcnt(0xc07f);
                     __builtin_amdgcn_wave_barrier();
                  }
               }
               k0 = k1;
            }
         }
         __syncthreads();
      }
      /* ---- back-transformation x = H_0 H_1 ... H_{n-2} z, one wavefront per vector */
      for (int k = wave; k < n; k += 4)
      {
         double zi = (lane < n) ? Z[lane * EI_N + k] : 0.0;
         for (int kk = n - 2; kk >= 0; --kk)
         {
            const double t = tau[kk];
            if ( t == 0.0 )
               continue;
            const bool in = lane > kk && lane < n;
            const double vk = in ? ((lane == kk + 1) ? 1.0 : a[lane][kk]) : 0.0;
            const double dot = ei_wsum(vk * zi);
            zi -= t * dot * vk;
         }
         const double nrm = ei_wsum(zi * zi);
         if ( lane < n )
            out[EI_N + (long long) k * n + lane] = nrm > 0.0 ? zi * ei_rsqrt(nrm) : zi;
      }
      __threadfence_system();
      __syncthreads();
      if ( tid == 0 )
         __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
   }
   if ( wave != 0 )
      return;

   /* ---- wavefront 0: i-th eigenvalue of T by Sturm multisection, 64 shifts per round, counts in product form on the matrix scaled
    * to norm 1 (ei_sturm_count) */
   double lo = 1e300, hi = -1e300, tnorm = 0.0;
   for (int i = 0; i < n; ++i)
   {
      const double rad = (i > 0 ? fabs(e[i - 1]) : 0.0) + (i + 1 < n ? fabs(e[i]) : 0.0);
      lo = fmin(lo, d[i] - rad);
      hi = fmax(hi, d[i] + rad);
      tnorm = fmax(tnorm, fabs(d[i]) + rad);
   }
   const double span0 = fmax(hi - lo, 1e-300);
   lo -= 1e-12 * span0 + 1e-300;
   hi += 1e-12 * span0 + 1e-300;
   tnorm = fmax(tnorm, 1e-300);
   const double sinv = 1.0 / tnorm;
   double* ds = pp;                                    /* (free since the reduction) */
   double* es = xc;
   if ( lane < n )
   {
      ds[lane] = d[lane] * sinv;
      es[lane] = (e[lane] * sinv) * (e[lane] * sinv);
   }
   __builtin_amdgcn_s_waitcnt(0xc07f);
   __builtin_amdgcn_wave_barrier();
   if ( lane < 8 )
   {
      ds[n + lane] = 4.0;
      es[n - 1 + lane] = 0.0;
   }
   __builtin_amdgcn_s_waitcnt(0xc07f);
   __builtin_amdgcn_wave_barrier();
   lo *= sinv; hi *= sinv;
   const int nb = (n - 1 + 3) >> 2;
   /* the search ends at two ulps of the eigenvalue (an interval cannot get shorter than one: the test "2e-16 relative" never fired
    * and all 16 rounds ran) or at the caller's relative width; without a caller's width not below half an ulp of the norm */
   const double rtol = fmax(mtol, 4.5e-16);
   const double rfloor = (mtol <= 4.5e-16) ? 0.25 : 0.0;
   for (int round = 0; round < 16; ++round)
   {
      const double x = lo + (hi - lo) * (double) (lane + 1) / 65.0;
      const int cnt = ei_sturm_count(ds, es, nb, x);   /* eigenvalues below x */
      const unsigned long long msk = __ballot(cnt >= ith);
      const int first = msk ? __ffsll((long long) msk) - 1 : 64;      /* first shift with at least ith eigenvalues below it */
      const double w = (hi - lo) / 65.0;
      const double nlo = lo + w * (double) first;
      const double nhi = (first < 64) ? lo + w * (double) (first + 1) : hi;
      lo = nlo; hi = nhi;
      if ( hi - lo <= rtol * fmax(fmax(fabs(lo), fabs(hi)), rfloor) )
         break;
   }
   lo *= tnorm; hi *= tnorm;
   const double theta = 0.5 * (lo + hi);
   if ( lane == 0 )
   {
      out[0] = theta;
      if ( flag == NULL && !wantvec )
         out[1] = 0.5 * (hi - lo);                     /* the eigenvalue lies within this of theta (engine callers: a rigorous bound) */
   }

   if ( wantvec )
   {
      /* inverse iteration on T - theta I (Gaussian elimination with partial pivoting, factored once; lane 0), start vector with
       * entries of alternating size so that it is not orthogonal to the eigenvector */
      if ( lane == 0 )
      {
         if ( n == 1 )
            zz[0] = 1.0;
         else
         {
            const double tiny = 1e-14 * fmax(span0, fmax(fabs(theta), 1e-300));
            double dd = d[0] - theta, du = e[0];
            for (int i = 0; i < n - 1; ++i)
            {
               const double dl = e[i];
               const double dn = d[i + 1] - theta;
               const double un = (i + 2 < n) ? e[i + 1] : 0.0;
               if ( fabs(dd) >= fabs(dl) )
               {
                  if ( fabs(dd) < tiny ) dd = tiny;
                  const double rinv = 1.0 / dd;
                  const double mlt = dl * rinv;
                  wk[0][i] = rinv; wk[1][i] = du; wk[2][i] = 0.0; wk[3][i] = mlt; swp[i] = 0.0;
                  dd = dn - mlt * du;
                  du = un;
               }
               else
               {
                  const double rinv = 1.0 / dl;
                  const double mlt = dd * rinv;
                  wk[0][i] = rinv; wk[1][i] = dn; wk[2][i] = un; wk[3][i] = mlt; swp[i] = 1.0;
                  dd = du - mlt * dn;
                  du = -mlt * un;
               }
            }
            if ( fabs(dd) < tiny ) dd = tiny;
            wk[0][n - 1] = 1.0 / dd; wk[1][n - 1] = 0.0; wk[2][n - 1] = 0.0;
            for (int i = 0; i < n; ++i)
               zz[i] = 1.0 + 0.37 * (double) ((i * 7) % 5);
            for (int iter = 0; iter < 4; ++iter)
            {
               double cur = zz[0];
               for (int i = 0; i < n - 1; ++i)
               {
                  const double nxt = zz[i + 1];
                  if ( swp[i] == 0.0 )
                  {
                     zz[i] = cur;
                     cur = nxt - wk[3][i] * cur;
                  }
                  else
                  {
                     zz[i] = nxt;
                     cur = cur - wk[3][i] * nxt;
                  }
               }
               double x1 = cur * wk[0][n - 1], x2 = 0.0;
               double nrm = x1 * x1;
               zz[n - 1] = x1;
               for (int i = n - 2; i >= 0; --i)
               {
                  const double xi = (zz[i] - wk[1][i] * x1 - wk[2][i] * x2) * wk[0][i];
                  zz[i] = xi;
                  nrm += xi * xi;
                  x2 = x1; x1 = xi;
               }
               nrm = sqrt(nrm);
               if ( !(nrm > 0.0) || !(nrm < 1e300) )
               {
                  for (int i = 0; i < n; ++i)
                     zz[i] = (i == 0) ? 1.0 : 0.0;
                  break;
               }
               const double rn = 1.0 / nrm;
               for (int i = 0; i < n; ++i)
                  zz[i] *= rn;
            }
         }
      }
      __builtin_amdgcn_wave_barrier();
      /* back-transformation x = H_0 H_1 ... H_{n-2} z: reflector k acts on entries k + 1 .. n - 1 */
      double zi = (lane < n) ? zz[lane] : 0.0;
      for (int k = n - 2; k >= 0; --k)
      {
         const double t = tau[k];
         if ( t == 0.0 )
            continue;
         const bool in = lane > k && lane < n;
         const double vk = in ? ((lane == k + 1) ? 1.0 : a[lane][k]) : 0.0;
         const double dot = ei_wsum(vk * zi);
         zi -= t * dot * vk;
      }
      const double nrm = sqrt(ei_wsum(zi * zi));
      if ( lane < n )
         out[1 + lane] = nrm > 0.0 ? zi / nrm : zi;
   }
   if ( flag == NULL )
      return;
   __threadfence_system();
   __builtin_amdgcn_wave_barrier();
   if ( lane == 0 )
   {
      __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
   }
}

template<bool ALL>
__global__ void __launch_bounds__(256) k_syevi_small(int n, int ith, int wantvec, const double* __restrict__ in, double* __restrict__ out,
   unsigned long long seq, unsigned long long* __restrict__ flag)
{
   d_syevi_small<ALL>(n, ith, wantvec, in, out, seq, flag);
}

/* ---- step lengths of the engine for blocks of 17 .. 64 rows: lambda_min(L D L^T) EXACTLY instead of by Lanczos -----------------
 * (L = inverse Cholesky factor of X or Z, D = dX or dZ: the scaled step whose smallest eigenvalue bounds the step length).  The
 * Lanczos kernel of eig.hip takes 120 us for 24 steps at n = 43 and returns an estimate with a residual bound; the reduction above
 * gives the eigenvalue itself in about half of that: the workgroup forms W = L D L^T in LDS (two products from LDS, 4 lanes per
 * row are not needed: 256 threads x n^2 / 256 entries), symmetrises it and runs the i = 1 path of d_syevi_small on it.
 * blockIdx.x = side (X / Z), blockIdx.y = block; res = {lambda_min, half width of its interval, n} as the other step-length kernels. */
__global__ void __launch_bounds__(256) k_lmin_exact_multi(hs_step_jobs P)
{
   extern __shared__ __attribute__((aligned(16))) double ex_dyn[];
   const int job = blockIdx.y;
   const int n = P.n[job];
   const double* __restrict__ Din = blockIdx.x ? P.D1[job] : P.D0[job];
   const double* __restrict__ Lin = blockIdx.x ? P.L1[job] : P.L0[job];
   double* __restrict__ res = blockIdx.x ? P.res1[job] : P.res0[job];
   const int tid = threadIdx.x;
   const int ld = n | 1;
   double* sl = ex_dyn;                 /* n x ld: L (lower triangle, zeros above) */
   double* sd = sl + n * ld;            /* n x ld: D, then W */
   double* st = sd + n * ld;            /* n x ld: T = D L^T */
   for (int e = tid; e < n * n; e += 256)
   {
      const int r = e / n, c = e - r * n;
      sl[r * ld + c] = (c <= r) ? Lin[e] : 0.0;
      sd[r * ld + c] = Din[e];
   }
   __syncthreads();
   /* T = D L^T (second factor given transposed), W = L T: 2 x 5 patches per thread (hs_lds_product.h); the zeros of L are multiplied */
   db_product<true>(n, ld, sd, sl, [&](int, int, int r, int c, double acc) { st[r * ld + c] = acc; });
   __syncthreads();
   db_product(n, ld, sl, st, [&](int, int, int r, int c, double acc) { sd[r * ld + c] = acc; });
   __syncthreads();
   /* symmetric part, as an n x n array without pitch (what d_syevi_small reads; it only looks at one triangle) */
   for (int e = tid; e < n * n; e += 256)
   {
      const int r = e / n, c = e - r * n;
      st[e] = 0.5 * (sd[r * ld + c] + sd[c * ld + r]);
   }
   __syncthreads();
   /* a step length needs nine digits, not sixteen: the multisection stops at a relative width of 1e-10 and hands the half width back
    * as the residual bound (res[1]), which the host subtracts - five or six rounds instead of nine or ten */
   d_syevi_small<false>(n, 1, 0, st, res, 0ULL, NULL, 1e-10);
   if ( tid == 0 )
      res[2] = (double) n;
}

/* ---- 64 < n <= 128: the same eigenpair, the matrix in LDS --------------------------------------------------------------------
 * k_syevi_small keeps the matrix in registers (64 x 64 over 256 threads); above 64 rows the full block-Jacobi decomposition was the
 * only path (2.5 ms per call at n = 65: a cliff, and DSYEVR on the host takes 0.1 - 0.3 ms there).  Here the matrix lives in LDS with
 * an odd pitch (n = 128: 132 KB), both triangles, and the reduction is the same DSYTD2 recurrence: a pair of threads per row forms
 * its entry of p = tau A v and applies the rank-2 update to its half of the row; the reflector and w are computed by every
 * wavefront for itself (two entries per lane), two workgroup barriers per column.  Multisection, inverse iteration and the
 * back-transformation are those of the small kernel with two entries per lane. */
#define EM_N 128
#define EM_FLAG 256                 /* position of the flag word in the output (behind eigenvalue + eigenvector) */
#define EM_NT 512                   /* threads of the two kernels for n <= 128 */

/* the reduction (512 threads, EM_TPR = 2 or 4 per row of the trailing block; contains barriers): the matrix from `in` into em_a
 * (pitch n | 1), tridiagonal matrix in d / e, the scalar factors of the reflectors in tau, reflector k (v_0 = 1 implied) in column k
 * of em_a below the subdiagonal.  Four threads per row up to 112 rows (n = 100: one eigenvalue 340 -> 291 us), two above (at
 * n = 128 four were 5 % slower: eight wavefronts at three barriers per column instead of four). */
template<int EM_TPR>
__device__ __forceinline__ void em_tridiag_t(int n, const double* __restrict__ in, double* em_a, double* vv, double* pp, double* ww, double* tau,
   double* d, double* e)
{
   const int tid = threadIdx.x, lane = tid & 63;
   const int ld = n | 1;
   for (int idx = tid; idx < n * n; idx += (int) blockDim.x)
   {
      const int i = idx / n, j = idx - i * n;
      em_a[i * ld + j] = (j <= i) ? in[(long long) j * n + i] : in[(long long) i * n + j];
   }
   if ( tid < EM_N )
   {
      vv[tid] = 0.0;
      ww[tid] = 0.0;
      pp[tid] = 0.0;
   }
   __syncthreads();

   for (int k = 0; k + 1 < n; ++k)
   {
      const int len = n - k - 1;
      double t;
      {
         const double xa = (lane < len) ? em_a[(k + 1 + lane) * ld + k] : 0.0;
         const double xb = (lane + 64 < len) ? em_a[(k + 65 + lane) * ld + k] : 0.0;
         const double x0 = ei_lane(xa, 0);
         const double s2 = ei_wsum((lane >= 1 ? xa * xa : 0.0) + xb * xb);
         double beta = x0, scale = 0.0;
         t = 0.0;
         if ( s2 > 0.0 )
         {
            const double h2 = x0 * x0 + s2;
            beta = -copysign(h2 * ei_rsqrt(h2), x0);
            t = (beta - x0) * ei_rcp2(beta);
            scale = ei_rcp2(x0 - beta);
         }
         if ( lane < len )
            vv[k + 1 + lane] = (lane == 0) ? 1.0 : xa * scale;
         if ( lane + 64 < len )
            vv[k + 65 + lane] = xb * scale;
         if ( lane == 0 )
         {
            vv[k] = 0.0;
            ww[k] = 0.0;
            tau[k] = t;
            e[k] = beta;
            d[k] = em_a[k * ld + k];
         }
         __builtin_amdgcn_s_waitcnt(0xc07f);             /* lgkmcnt(0): this wavefront's own LDS writes are done */
         __builtin_amdgcn_wave_barrier();
      }
      if ( t != 0.0 )                                    /* (the same in every wavefront) */
      {
         const int row = k + 1 + tid / EM_TPR, part = tid % EM_TPR;
         {
            double acc = 0.0;
            if ( row < n )
            {
               /* four partial sums: the loop is a chain of dependent multiply-adds otherwise (16 cycles each) */
               const double* ar = em_a + row * ld;
               double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
               int c = k + 1 + part;
               for (; c + 3 * EM_TPR < n; c += 4 * EM_TPR)
               {
                  a0 += ar[c] * vv[c];
                  a1 += ar[c + EM_TPR] * vv[c + EM_TPR];
                  a2 += ar[c + 2 * EM_TPR] * vv[c + 2 * EM_TPR];
                  a3 += ar[c + 3 * EM_TPR] * vv[c + 3 * EM_TPR];
               }
               for (; c < n; c += EM_TPR)
                  a0 += ar[c] * vv[c];
               acc = (a0 + a1) + (a2 + a3);
            }
            if ( EM_TPR == 4 )
               acc = ei_quad(acc);                      /* the parts of a row are neighbours */
            else
               acc += ei_dpp<0xB1>(acc);
            if ( part == 0 && row < n )
               pp[row] = t * acc;
         }
         __syncthreads();
         {
            const double pa = (lane < len) ? pp[k + 1 + lane] : 0.0, pb = (lane + 64 < len) ? pp[k + 65 + lane] : 0.0;
            const double va = (lane < len) ? vv[k + 1 + lane] : 0.0, vb = (lane + 64 < len) ? vv[k + 65 + lane] : 0.0;
            const double pv = ei_wsum(pa * va + pb * vb);
            const double al = -0.5 * t * pv;
            if ( lane < len )
               ww[k + 1 + lane] = pa + al * va;
            if ( lane + 64 < len )
               ww[k + 65 + lane] = pb + al * vb;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
         }
         if ( row < n )
         {
            double* ar = em_a + row * ld;
            const double vr = vv[row], wr = ww[row];
#pragma unroll 4
            for (int c = k + 1 + part; c < n; c += EM_TPR)
               ar[c] -= vr * ww[c] + wr * vv[c];
         }
         /* keep the reflector (v_0 = 1 implied) in the column it annihilated: the update does not touch column k */
         if ( tid < len )
            em_a[(k + 1 + tid) * ld + k] = vv[k + 1 + tid];
      }
      __syncthreads();
   }
   if ( tid == 0 )
   {
      d[n - 1] = em_a[(n - 1) * ld + n - 1];
      e[n - 1] = 0.0;
   }
   __syncthreads();
}

__device__ __forceinline__ void em_tridiag(int n, const double* __restrict__ in, double* em_a, double* vv, double* pp, double* ww, double* tau,
   double* d, double* e)
{
   if ( n <= 112 )
      em_tridiag_t<4>(n, in, em_a, vv, pp, ww, tau, d, e);
   else
      em_tridiag_t<2>(n, in, em_a, vv, pp, ww, tau, d, e);
}

__global__ void __launch_bounds__(EM_NT) k_syevi_mid(int n, int ith, int wantvec, const double* __restrict__ in, double* __restrict__ out,
   unsigned long long seq, unsigned long long* __restrict__ flag)
{
   extern __shared__ __attribute__((aligned(16))) double em_a[];
   __shared__ double vv[EM_N + 8], pp[EM_N], ww[EM_N + 8], tau[EM_N], d[EM_N], e[EM_N], zz[EM_N];
   __shared__ double wk[4][EM_N], swp[EM_N];
   const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
   const int ld = n | 1;
   em_tridiag(n, in, em_a, vv, pp, ww, tau, d, e);
   if ( wave != 0 )
      return;

   /* ---- wavefront 0: i-th eigenvalue of T by Sturm multisection (64 shifts per round, counts in product form: ei_sturm_count) */
   double lo = 1e300, hi = -1e300, tnorm = 0.0;
   for (int i = 0; i < n; ++i)
   {
      const double rad = (i > 0 ? fabs(e[i - 1]) : 0.0) + (i + 1 < n ? fabs(e[i]) : 0.0);
      lo = fmin(lo, d[i] - rad);
      hi = fmax(hi, d[i] + rad);
      tnorm = fmax(tnorm, fabs(d[i]) + rad);
   }
   const double span0 = fmax(hi - lo, 1e-300);
   lo -= 1e-12 * span0 + 1e-300;
   hi += 1e-12 * span0 + 1e-300;
   tnorm = fmax(tnorm, 1e-300);
   const double sinv = 1.0 / tnorm;
   double* ds = vv;                                    /* (free since the reduction) */
   double* es = ww;
   for (int i = lane; i < n; i += 64)
   {
      ds[i] = d[i] * sinv;
      es[i] = (e[i] * sinv) * (e[i] * sinv);
   }
   __builtin_amdgcn_s_waitcnt(0xc07f);
   __builtin_amdgcn_wave_barrier();
   if ( lane < 8 )
   {
      ds[n + lane] = 4.0;
      es[n - 1 + lane] = 0.0;
   }
   __builtin_amdgcn_s_waitcnt(0xc07f);
   __builtin_amdgcn_wave_barrier();
   lo *= sinv; hi *= sinv;
   const int nb = (n - 1 + 3) >> 2;
   for (int round = 0; round < 16; ++round)
   {
      const double x = lo + (hi - lo) * (double) (lane + 1) / 65.0;
      const int cnt = ei_sturm_count(ds, es, nb, x);   /* eigenvalues below x */
      const unsigned long long msk = __ballot(cnt >= ith);
      const int first = msk ? __ffsll((long long) msk) - 1 : 64;      /* first shift with at least ith eigenvalues below it */
      const double w = (hi - lo) / 65.0;
      const double nlo = lo + w * (double) first;
      const double nhi = (first < 64) ? lo + w * (double) (first + 1) : hi;
      lo = nlo; hi = nhi;
      if ( hi - lo <= 4.5e-16 * fmax(fmax(fabs(lo), fabs(hi)), 0.25) )
         break;
   }
   lo *= tnorm; hi *= tnorm;
   const double theta = 0.5 * (lo + hi);
   if ( lane == 0 )
      out[0] = theta;

   if ( wantvec )
   {
      /* inverse iteration on T - theta I (Gaussian elimination with partial pivoting, factored once; lane 0) */
      if ( lane == 0 )
      {
         const double tiny = 1e-14 * fmax(span0, fmax(fabs(theta), 1e-300));
         double dd = d[0] - theta, du = e[0];
         for (int i = 0; i < n - 1; ++i)
         {
            const double dl = e[i];
            const double dn = d[i + 1] - theta;
            const double un = (i + 2 < n) ? e[i + 1] : 0.0;
            if ( fabs(dd) >= fabs(dl) )
            {
               if ( fabs(dd) < tiny ) dd = tiny;
               const double rinv = 1.0 / dd;
               const double mlt = dl * rinv;
               wk[0][i] = rinv; wk[1][i] = du; wk[2][i] = 0.0; wk[3][i] = mlt; swp[i] = 0.0;
               dd = dn - mlt * du;
               du = un;
            }
            else
            {
               const double rinv = 1.0 / dl;
               const double mlt = dd * rinv;
               wk[0][i] = rinv; wk[1][i] = dn; wk[2][i] = un; wk[3][i] = mlt; swp[i] = 1.0;
               dd = du - mlt * dn;
               du = -mlt * un;
            }
         }
         if ( fabs(dd) < tiny ) dd = tiny;
         wk[0][n - 1] = 1.0 / dd; wk[1][n - 1] = 0.0; wk[2][n - 1] = 0.0;
         for (int i = 0; i < n; ++i)
            zz[i] = 1.0 + 0.37 * (double) ((i * 7) % 5);
         for (int iter = 0; iter < 4; ++iter)
         {
            double cur = zz[0];
            for (int i = 0; i < n - 1; ++i)
            {
               const double nxt = zz[i + 1];
               if ( swp[i] == 0.0 )
               {
                  zz[i] = cur;
                  cur = nxt - wk[3][i] * cur;
               }
               else
               {
                  zz[i] = nxt;
                  cur = cur - wk[3][i] * nxt;
               }
            }
            double x1 = cur * wk[0][n - 1], x2 = 0.0;
            double nrm = x1 * x1;
            zz[n - 1] = x1;
            for (int i = n - 2; i >= 0; --i)
            {
               const double xi = (zz[i] - wk[1][i] * x1 - wk[2][i] * x2) * wk[0][i];
               zz[i] = xi;
               nrm += xi * xi;
               x2 = x1; x1 = xi;
            }
            nrm = sqrt(nrm);
            if ( !(nrm > 0.0) || !(nrm < 1e300) )
            {
               for (int i = 0; i < n; ++i)
                  zz[i] = (i == 0) ? 1.0 : 0.0;
               break;
            }
            const double rn = 1.0 / nrm;
            for (int i = 0; i < n; ++i)
               zz[i] *= rn;
         }
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
      /* back-transformation x = H_0 H_1 ... H_{n-2} z: reflector k acts on entries k + 1 .. n - 1; lane l holds entries l and l + 64 */
      double za = (lane < n) ? zz[lane] : 0.0, zb = (lane + 64 < n) ? zz[lane + 64] : 0.0;
      for (int k = n - 2; k >= 0; --k)
      {
         const double t = tau[k];
         if ( t == 0.0 )
            continue;
         const int ia = lane, ib = lane + 64;
         const double va = (ia > k && ia < n) ? ((ia == k + 1) ? 1.0 : em_a[ia * ld + k]) : 0.0;
         const double vb = (ib > k && ib < n) ? ((ib == k + 1) ? 1.0 : em_a[ib * ld + k]) : 0.0;
         const double dot = ei_wsum(va * za + vb * zb);
         za -= t * dot * va;
         zb -= t * dot * vb;
      }
      const double nrm = sqrt(ei_wsum(za * za + zb * zb));
      if ( lane < n )
         out[1 + lane] = nrm > 0.0 ? za / nrm : za;
      if ( lane + 64 < n )
         out[65 + lane] = nrm > 0.0 ? zb / nrm : zb;
   }
   __threadfence_system();
   __builtin_amdgcn_wave_barrier();
   if ( lane == 0 )
      __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}


/* ---- 64 < n <= 128: ALL eigenpairs in one launch ---------------------------------------------------------------------------------
 * The phases of k_syevi_small<true> with two entries per lane, laid out for what 160 KB of LDS hold: the reduction needs the matrix
 * there (132 KB at n = 128), the inverse iteration the eigenvectors of T.  So after the reduction the reflectors move to device
 * memory, transposed (reflector k contiguous: the back-transformation reads it with one coalesced load per half), and the array
 * becomes Z[i][k] (component i of vector k, the same odd pitch: thread-per-vector and lane-per-component accesses are both conflict
 * free).  The two factor arrays of the thread-per-vector elimination (pivot reciprocals, first superdiagonal of U) live in device
 * memory as [row][vector]: written once per round in the forward sweep (nobody waits for a store), read in the backward sweep eight
 * rows ahead of the recurrence that uses them.  All eigenvalues by multisection as in the small kernel (thread = eigenvalue x one of
 * four shifts), in two passes of 64 eigenvalues.  scratch: 3 * 128 * 128 doubles of device memory.
 * out: [0, n) eigenvalues ascending, [EM_N + k n + i] component i of eigenvector k, flag word behind them. */
#define EM_ALL_FLAG (EM_N + EM_N * EM_N + 4)
#define EM_ALL_OUT (EM_N + EM_N * EM_N + 16)
/* the two phases of k_syev_mid that work in the 16-lane layout (lane l of a row of 16 holds the components l + 16 j, j < NJ): NJ = 4
 * serves matrices of up to 64 rows with half the instructions of NJ = 8, and so on down to NJ = 1 for 16 rows (n = 43 of rank 5:
 * 550 -> 390 us per decomposition with NJ = 4) */
template<int NJ>
__device__ __forceinline__ void em_clusters(int n, int ld, int iter, double ortol, double* Z, const double* zz, double (*cbuf)[EM_NT / 64][EM_N])
{
   const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
   /* clusters: one wavefront per cluster (cluster c goes to wavefront c mod 8).  Layout: the four rows of 16 lanes of the wavefront
    * work on four vectors p of the cluster at a time, lane l of a row holds the components l + 16 j (j < 8): a dot product is
    * eight multiply-adds and ONE reduction over 16 lanes for the four of them (lane = component over the whole wavefront: a
    * 64-lane reduction of 25 instructions per dot product - 1.04 ms per round for the 96 zero eigenvalues of a rank-32 matrix of
    * order 128, the whole decomposition slower than the Jacobi iteration).  Classical Gram-Schmidt against the vectors of the
    * cluster before k, twice. */
   {
      const int l16 = lane & 15, row = lane >> 4;
      int cl = -1;
      int k0 = 0;
      while ( k0 < n )
      {
         int k1 = k0 + 1;
         while ( k1 < n && zz[k1] - zz[k1 - 1] <= ortol )
            ++k1;
         ++cl;
         if ( k1 - k0 >= 16 )
         {
            /* a large cluster (the zero eigenvalues of a low-rank matrix): ALL wavefronts share it - the groups of four earlier
             * vectors go round the wavefronts, the corrections meet in LDS (alone, one wavefront needed 600 us per round for
             * 96 vectors: the decomposition of a rank-32 matrix of order 128 took as long as the Jacobi iteration) */
            for (int k = k0; k < k1; ++k)
            {
               double v[NJ];
#pragma unroll
               for (int j = 0; j < NJ; ++j)
               {
                  const int i = l16 + 16 * j;
                  v[j] = (i < n) ? Z[i * ld + k] : 0.0;
               }
               /* (one pass in the first two rounds - it only has to keep the vectors from collapsing onto each other before the
                * next amplification -, two in the last) */
               for (int pass = 0; pass < (iter == 2 ? 2 : 1); ++pass)
               {
                  double corr[NJ];
#pragma unroll
                  for (int j = 0; j < NJ; ++j)
                     corr[j] = 0.0;
                  for (int p0 = k0 + 4 * wave; p0 < k; p0 += 4 * (EM_NT / 64))
                  {
                     const int p = p0 + row;
                     double u[NJ];
                     double dt = 0.0;
#pragma unroll
                     for (int j = 0; j < NJ; ++j)
                     {
                        const int i = l16 + 16 * j;
                        u[j] = (i < n && p < k) ? Z[i * ld + (p < k ? p : k0)] : 0.0;
                        dt = fma(u[j], v[j], dt);
                     }
                     dt = ei_sum16(dt);
#pragma unroll
                     for (int j = 0; j < NJ; ++j)
                        corr[j] = fma(dt, u[j], corr[j]);
                  }
#pragma unroll
                  for (int j = 0; j < NJ; ++j)
                  {
                     double c = corr[j];
                     c += __shfl_xor(c, 16, 64);
                     c += __shfl_xor(c, 32, 64);
                     if ( row == 0 )
                        cbuf[pass][wave][l16 + 16 * j] = c;
                  }
                  __syncthreads();
#pragma unroll
                  for (int j = 0; j < NJ; ++j)
                  {
                     double c = 0.0;
#pragma unroll
                     for (int w = 0; w < EM_NT / 64; ++w)
                        c += cbuf[pass][w][l16 + 16 * j];
                     v[j] -= c;
                  }
               }
               double nr = 0.0;
#pragma unroll
               for (int j = 0; j < NJ; ++j)
                  nr = fma(v[j], v[j], nr);
               nr = ei_sum16(nr);
               const double rs = ei_rsqrt(fmax(nr, 1e-300));
               if ( wave == 0 && row == 0 )
               {
#pragma unroll
                  for (int j = 0; j < NJ; ++j)
                  {
                     const int i = l16 + 16 * j;
                     if ( i < n )
                        Z[i * ld + k] = v[j] * rs;
                  }
               }
               __syncthreads();
            }
         }
         else if ( k1 - k0 > 1 && (cl & 7) == wave )
         {
            for (int k = k0; k < k1; ++k)
            {
               double v[NJ];
#pragma unroll
               for (int j = 0; j < NJ; ++j)
               {
                  const int i = l16 + 16 * j;
                  v[j] = (i < n) ? Z[i * ld + k] : 0.0;
               }
               for (int pass = 0; pass < 2; ++pass)
               {
                  double corr[NJ];
#pragma unroll
                  for (int j = 0; j < NJ; ++j)
                     corr[j] = 0.0;
                  for (int p0 = k0; p0 < k; p0 += 4)
                  {
                     const int p = p0 + row;
                     double u[NJ];
                     double dt = 0.0;
#pragma unroll
                     for (int j = 0; j < NJ; ++j)
                     {
                        const int i = l16 + 16 * j;
                        u[j] = (i < n && p < k) ? Z[i * ld + (p < k ? p : k0)] : 0.0;
                        dt = fma(u[j], v[j], dt);
                     }
                     dt = ei_sum16(dt);
#pragma unroll
                     for (int j = 0; j < NJ; ++j)
                        corr[j] = fma(dt, u[j], corr[j]);
                  }
                  /* the four rows' corrections together (every row ends with the same vector) */
#pragma unroll
                  for (int j = 0; j < NJ; ++j)
                  {
                     double c = corr[j];
                     c += __shfl_xor(c, 16, 64);
                     c += __shfl_xor(c, 32, 64);
                     v[j] -= c;
                  }
               }
               double nr = 0.0;
#pragma unroll
               for (int j = 0; j < NJ; ++j)
                  nr = fma(v[j], v[j], nr);
               nr = ei_sum16(nr);
               const double rs = ei_rsqrt(fmax(nr, 1e-300));
               if ( row == 0 )
               {
#pragma unroll
                  for (int j = 0; j < NJ; ++j)
                  {
                     const int i = l16 + 16 * j;
                     if ( i < n )
                        Z[i * ld + k] = v[j] * rs;
                  }
               }
               __builtin_amdgcn_s_waitcnt(0xc07f);
               __builtin_amdgcn_wave_barrier();
            }
         }
         k0 = k1;
      }
   }
}

template<int NJ>
__device__ __forceinline__ void em_backtransform(int n, int ld, const double* Z, const double* __restrict__ Vt, const double* tau, double* __restrict__ out)
{
   const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
   /* ---- back-transformation x = H_0 H_1 ... H_{n-2} z in the same layout: a wavefront takes four vectors at a time (one per row of 16
    * lanes, lane l: components l + 16 j), a reflector costs eight multiply-adds, one 16-lane reduction and eight more for the four of
    * them (one vector per wavefront with a 64-lane reduction per reflector: 460 cycles each, 775 us at n = 128); the reflectors from
    * device memory, two ahead */
   {
      const int l16 = lane & 15, row = lane >> 4;
      for (int k0 = 4 * wave; k0 < n; k0 += 32)
      {
         const int k = k0 + row;
         double z[NJ];
#pragma unroll
         for (int j = 0; j < NJ; ++j)
         {
            const int i = l16 + 16 * j;
            z[j] = (i < n && k < n) ? Z[i * ld + k] : 0.0;
         }
         for (int kk0 = n - 2; kk0 >= 0; kk0 -= 2)
         {
            double r0[NJ], r1[NJ];
            const int kb = (kk0 - 1 >= 0) ? kk0 - 1 : 0;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
            {
               r0[j] = Vt[kk0 * EM_N + l16 + 16 * j];
               r1[j] = Vt[kb * EM_N + l16 + 16 * j];
            }
#pragma unroll
            for (int r = 0; r < 2; ++r)
            {
               const int kk = kk0 - r;
               if ( kk < 0 )
                  continue;
               const double t = tau[kk];
               if ( t == 0.0 )
                  continue;
               double vr[NJ];
               double dot = 0.0;
#pragma unroll
               for (int j = 0; j < NJ; ++j)
               {
                  const int i = l16 + 16 * j;
                  const double rv = r ? r1[j] : r0[j];
                  vr[j] = (i > kk && i < n) ? ((i == kk + 1) ? 1.0 : rv) : 0.0;
                  dot = fma(vr[j], z[j], dot);
               }
               dot = t * ei_sum16(dot);
#pragma unroll
               for (int j = 0; j < NJ; ++j)
                  z[j] = fma(-dot, vr[j], z[j]);
            }
         }
         double nrm = 0.0;
#pragma unroll
         for (int j = 0; j < NJ; ++j)
            nrm = fma(z[j], z[j], nrm);
         nrm = ei_sum16(nrm);
         const double rs = nrm > 0.0 ? ei_rsqrt(nrm) : 1.0;
#pragma unroll
         for (int j = 0; j < NJ; ++j)
         {
            const int i = l16 + 16 * j;
            if ( k < n && i < n )
               out[EM_N + (long long) k * n + i] = z[j] * rs;
         }
      }
   }
}

__global__ void __launch_bounds__(EM_NT) k_syev_mid(int n, const double* __restrict__ in, double* __restrict__ out, double* __restrict__ scratch,
   unsigned long long seq, unsigned long long* __restrict__ flag)
{
   extern __shared__ __attribute__((aligned(16))) double em_a[];
   __shared__ double vv[EM_N + 8], pp[EM_N], ww[EM_N + 8], tau[EM_N], d[EM_N], e[EM_N], zz[EM_N];
   __shared__ double cbuf[2][EM_NT / 64][EM_N];            /* corrections of the wavefronts that share a large cluster */
   const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
   const int ld = n | 1;
   /* developer aid: phase ends in ticks of the 100 MHz counter behind the flag word (HIPSDP_SYEV_STAMPS=1 prints them) */
   const long long w0 = wall_clock64();
#define EM_STAMP(j) do { if ( tid == 0 ) out[EM_ALL_FLAG + 2 + (j)] = (double) (wall_clock64() - w0); } while (0)
   em_tridiag(n, in, em_a, vv, pp, ww, tau, d, e);
   EM_STAMP(0);

   /* ---- the reflectors leave LDS: Vt[k][i] = entry i of reflector k (i > k + 1; entry k + 1 is the implied 1) */
   double* __restrict__ Vt = scratch;
   /* the two factor arrays of the elimination, [i][k] with pitch gs: behind Z in LDS when n <= 64 (the launch asks for the room),
    * else in device memory */
   const int gs = (n <= 64) ? 64 : EM_N;
   double* __restrict__ G0 = (n <= 64) ? em_a + n * ld : scratch + EM_N * EM_N;      /* 1 / pivot of row i of the elimination for vector k */
   double* __restrict__ G1 = G0 + (n <= 64 ? n * 64 : EM_N * EM_N);                  /* first superdiagonal of U */
   for (int idx = tid; idx < n * n; idx += EM_NT)
   {
      const int k = idx / n, i = idx - k * n;
      Vt[k * EM_N + i] = (i > k + 1) ? em_a[i * ld + k] : 0.0;
   }
   __syncthreads();
   double* Z = em_a;                                        /* [i][k], pitch ld */

   /* ---- all eigenvalues: thread (k = tid >> 2 (+ 64 in the second pass), s = tid & 3) */
   double glo = 1e300, ghi = -1e300, tnorm = 0.0;
   for (int i = 0; i < n; ++i)
   {
      const double rad = (i > 0 ? fabs(e[i - 1]) : 0.0) + (i + 1 < n ? fabs(e[i]) : 0.0);
      glo = fmin(glo, d[i] - rad);
      ghi = fmax(ghi, d[i] + rad);
      tnorm = fmax(tnorm, fabs(d[i]) + rad);
   }
   const double span0 = fmax(ghi - glo, 1e-300);
   glo -= 1e-12 * span0 + 1e-300;
   ghi += 1e-12 * span0 + 1e-300;
   /* Sturm counts in product form on the matrix scaled to norm 1 (p_0 = 1, p_1 = d_0 - x, p_{i+1} = (d_i - x) p_i - e_{i-1}^2 p_{i-1};
    * a sign change = an eigenvalue below x, a zero takes the sign opposite to its predecessor; rescaled every fourth step): two
    * dependent operations per step where the quotient form of the small kernel has a division - 260 cycles per step, 640 us for the
    * 128 eigenvalues of a 128 x 128 matrix */
   const double sinv = 1.0 / fmax(tnorm, 1e-300);
   double* ds = vv;                                         /* (free since the reduction) */
   double* es = ww;
   if ( tid < n )
   {
      ds[tid] = d[tid] * sinv;
      es[tid] = (e[tid] * sinv) * (e[tid] * sinv);
   }
   __syncthreads();
   if ( tid < 8 )
   {
      ds[n + tid] = 4.0;                                    /* rows behind the matrix: no coupling, no sign change (|x| <= 1) */
      es[n - 1 + tid] = 0.0;
   }
   __syncthreads();
   {
      /* thread = (eigenvalue, one of S shifts): S = 4, 8 or 16 - what the 512 threads allow for n eigenvalues (2.3, 3.2 or 4.1 bits
       * per round) */
      const int lgS = (n > 64) ? 2 : ((n > 32) ? 3 : 4), S = 1 << lgS;
      const int k = tid >> lgS, sh = tid & (S - 1);
      const int nb = (n - 1 + 3) >> 2;                      /* blocks of four steps i = 1 + 4 b .. 4 + 4 b */
      const double rS1 = 1.0 / (double) (S + 1);
      double lo = glo * sinv, hi = ghi * sinv;
      for (int round = 0; round < 40; ++round)
      {
         const double w = (hi - lo) * rS1;
         const double x = lo + w * (double) (sh + 1);
         const int cnt = (k < n) ? ei_sturm_count(ds, es, nb, x) : 0;
         /* number of the S shifts with fewer than k + 1 eigenvalues below them = index of the subinterval that holds eigenvalue k */
         int below = (cnt < k + 1) ? 1 : 0;
         below += __builtin_amdgcn_update_dpp(0, below, 0xB1, 0xf, 0xf, true);
         below += __builtin_amdgcn_update_dpp(0, below, 0x4E, 0xf, 0xf, true);
         if ( lgS >= 3 )
            below += __builtin_amdgcn_update_dpp(0, below, 0x141, 0xf, 0xf, true);
         if ( lgS >= 4 )
            below += __builtin_amdgcn_update_dpp(0, below, 0x140, 0xf, 0xf, true);
         const double nlo = lo + w * (double) below;
         const double nhi = (below < S) ? lo + w * (double) (below + 1) : hi;
         lo = nlo; hi = nhi;
         /* (to two ulps of the eigenvalue - an interval cannot get shorter than one -, but not below half an ulp of the norm) */
         if ( __all(k >= n || hi - lo <= 4.5e-16 * fmax(fmax(fabs(lo), fabs(hi)), 0.25)) )
            break;
      }
      if ( k < n && sh == 0 )
         zz[k] = 0.5 * (lo + hi) * tnorm;
   }
   __syncthreads();
   EM_STAMP(1);
   if ( tid < n )
      out[tid] = zz[tid];

   /* ---- eigenvectors of T: three rounds of { one step of inverse iteration per vector (thread k owns vector k), orthogonalisation
    * inside the clusters } - see k_syevi_small for why in every round */
   const double ortol = 1e-3 * fmax(tnorm, 1e-300);
   /* (without two eigenvalues that close the orthogonalisation phase - every thread walking the list of eigenvalues, two barriers -
    * is skipped: 8 us per round at n = 43) */
   const int clustered = __syncthreads_or((tid > 0 && tid < n && zz[tid] - zz[tid - 1] <= ortol) ? 1 : 0);
   for (int idx = tid; idx < n * n; idx += EM_NT)
   {
      const int i = idx / n, k = idx - i * n;
      unsigned h = (unsigned) (i * 2654435761u) ^ (unsigned) ((k + 1) * 40503u);
      h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
      Z[i * ld + k] = 0.5 + (double) (h & 0xFFFF) * (1.0 / 65536.0);
   }
   __syncthreads();
   for (int iter = 0; iter < 3; ++iter)
   {
      {
         const int k = tid;
         if ( k < n )
         {
            const double theta = zz[k];
            const double tiny = 1e-14 * fmax(span0, fmax(fabs(theta), 1e-300));
            double dd = d[0] - theta, du = e[0];
            double cur = Z[k];
            unsigned long long swlo = 0ULL, swhi = 0ULL;     /* bit i (of 128): rows i and i + 1 were exchanged */
            for (int i = 0; i < n - 1; ++i)
            {
               const double dl = e[i];
               const double dn = d[i + 1] - theta;
               const double un = (i + 2 < n) ? e[i + 1] : 0.0;
               const double nxt = Z[(i + 1) * ld + k];
               if ( fabs(dd) >= fabs(dl) || fabs(dl) < tiny )
               {
                  if ( fabs(dd) < tiny ) dd = tiny;
                  const double rinv = ei_rcp2(dd);
                  const double mlt = dl * rinv;
                  G0[i * gs + k] = rinv; G1[i * gs + k] = du;
                  Z[i * ld + k] = cur;
                  cur = nxt - mlt * cur;
                  dd = dn - mlt * du;
                  du = un;
               }
               else
               {
                  const double rinv = ei_rcp2(dl);
                  const double mlt = dd * rinv;
                  G0[i * gs + k] = rinv; G1[i * gs + k] = dn;
                  if ( i < 64 ) swlo |= 1ULL << i; else swhi |= 1ULL << (i - 64);
                  Z[i * ld + k] = nxt;
                  cur = cur - mlt * nxt;
                  dd = du - mlt * dn;
                  du = -mlt * un;
               }
            }
            if ( fabs(dd) < tiny ) dd = tiny;
            double x1 = cur * ei_rcp2(dd), x2 = 0.0;
            double nrm = x1 * x1;
            Z[(n - 1) * ld + k] = x1;
            /* backward sweep, the factors of eight rows on their way while the recurrence runs */
            for (int i0 = n - 2; i0 >= 0; i0 -= 8)
            {
               double g0[8], g1[8];
#pragma unroll
               for (int u = 0; u < 8; ++u)
               {
                  const int i = (i0 - u >= 0) ? i0 - u : 0;
                  g0[u] = G0[i * gs + k];
                  g1[u] = G1[i * gs + k];
               }
#pragma unroll
               for (int u = 0; u < 8; ++u)
               {
                  const int i = i0 - u;
                  if ( i >= 0 )
                  {
                     const bool sw = (i < 64) ? ((swlo >> i) & 1ULL) : ((swhi >> (i - 64)) & 1ULL);
                     const double u2 = sw ? ((i + 2 < n) ? e[i + 1] : 0.0) : 0.0;
                     const double xi = (Z[i * ld + k] - g1[u] * x1 - u2 * x2) * g0[u];
                     Z[i * ld + k] = xi;
                     nrm += xi * xi;
                     x2 = x1; x1 = xi;
                     if ( !(nrm < 1e280) )
                     {
                        const double sc1 = 1e-140;
                        for (int j = i; j < n; ++j)
                           Z[j * ld + k] *= sc1;
                        x1 *= sc1; x2 *= sc1; nrm *= sc1 * sc1;
                     }
                  }
               }
            }
            double rn = ei_rsqrt(fmax(nrm, 1e-300));
            if ( !(nrm > 0.0) || !(nrm < 1e300) )
            {
               for (int i = 0; i < n; ++i)
                  Z[i * ld + k] = (i == k) ? 1.0 : 0.0;
               rn = 1.0;
            }
            for (int i = 0; i < n; ++i)
               Z[i * ld + k] *= rn;
         }
         __syncthreads();
         EM_STAMP(2 + 2 * iter);
      }
      if ( clustered )
      {
         if ( n <= 16 )
            em_clusters<1>(n, ld, iter, ortol, Z, zz, cbuf);
         else if ( n <= 32 )
            em_clusters<2>(n, ld, iter, ortol, Z, zz, cbuf);
         else if ( n <= 64 )
            em_clusters<4>(n, ld, iter, ortol, Z, zz, cbuf);
         else
            em_clusters<8>(n, ld, iter, ortol, Z, zz, cbuf);
         __syncthreads();
      }
      EM_STAMP(3 + 2 * iter);
   }
   if ( n <= 16 )
      em_backtransform<1>(n, ld, Z, Vt, tau, out);
   else if ( n <= 32 )
      em_backtransform<2>(n, ld, Z, Vt, tau, out);
   else if ( n <= 64 )
      em_backtransform<4>(n, ld, Z, Vt, tau, out);
   else
      em_backtransform<8>(n, ld, Z, Vt, tau, out);
   __syncthreads();
   EM_STAMP(8);
#undef EM_STAMP
   __threadfence_system();
   __syncthreads();
   if ( tid == 0 )
      __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

/* dynamic LDS of k_syev_mid: the matrix / the eigenvectors of T, and up to 64 rows the two factor arrays of the inverse iteration */
static size_t em_all_lds(int n)
{
   return ((size_t) n * (n | 1) + (n <= 64 ? (size_t) 2 * n * 64 : 0)) * sizeof(double);
}

/* from how many rows on the full decomposition takes k_syev_mid (HIPSDP_SYEV_MID_FROM, read once; both entry points use the same
 * rule: the literal PSD projection chain depends on the basis).  Default 10: written for 64 < n <= 128, the kernel turned out to be
 * ahead of k_syevi_small<true> - whose matrix sits in registers, but whose counts are in quotient form and whose reductions span
 * 64 lanes - from 10 rows on (16: 104 against 121 us per call, 33: 188 / 236, 48: 277 / 369, 64: 389 / 538), level below. */
static int em_mid_from(void)
{
   static int from = -1;
   if ( from < 0 )
   {
      const char* env = getenv("HIPSDP_SYEV_MID_FROM");
      int v = env != NULL ? atoi(env) : 10;
      if ( v < 2 ) v = 2;
      if ( v > EI_N + 1 ) v = EI_N + 1;
      from = v;
   }
   return from;
}

/* per host thread and device: a stream and the pinned, device-mapped staging memory.  The object lives in thread-local storage:
 * its destructor returns stream and pinned memory when the thread ends. */
#define EI_OUT_DOUBLES (EI_N * EI_N + EI_N + 16)       /* room for a full decomposition: eigenvalues, eigenvectors, flag word */
struct ei_ctx
{
   int device;
   hipStream_t stream;
   double* hin;  double* din;
   double* hout; double* dout;          /* one eigenpair: [0] eigenvalue, [1 .. 64] eigenvector, flag word at EI_N + 4 */
   double* dscr;                        /* device memory of the full decomposition above 64 rows (k_syev_mid), allocated at its first call */
   unsigned long long seq;
   ei_ctx() : device(-1), stream(NULL), hin(NULL), din(NULL), hout(NULL), dout(NULL), dscr(NULL), seq(0) {}
   void release()
   {
      if ( stream != NULL )
      {
         (void) hipSetDevice(device);
         (void) hipStreamSynchronize(stream);
         (void) hipStreamDestroy(stream);
      }
      if ( hin != NULL ) (void) hipHostFree(hin);
      if ( hout != NULL ) (void) hipHostFree(hout);
      if ( dscr != NULL ) (void) hipFree(dscr);
      device = -1; stream = NULL; hin = din = hout = dout = dscr = NULL;
   }
   ~ei_ctx() { release(); }
};

thread_local ei_ctx g_ctx;

int ei_context(int device, ei_ctx** out)
{
   if ( g_ctx.device == device && g_ctx.stream != NULL )
   {
      *out = &g_ctx;
      return HS_OK;
   }
   g_ctx.release();
   /* built in locals and committed only when complete: a failure half-way leaves no half-initialised context behind */
   ei_ctx c;
   c.device = device;
   hipError_t e = hipSetDevice(device);
   if ( e == hipSuccess ) e = hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking);
   if ( e == hipSuccess ) e = hipHostMalloc((void**) &c.hin, (size_t) EM_N * EM_N * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent);
   if ( e == hipSuccess ) e = hipHostMalloc((void**) &c.hout, (size_t) EM_ALL_OUT * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent);
   if ( e == hipSuccess ) e = hipHostGetDevicePointer((void**) &c.din, c.hin, 0);
   if ( e == hipSuccess ) e = hipHostGetDevicePointer((void**) &c.dout, c.hout, 0);
   if ( e != hipSuccess )
   {
      hs_record_hip_error(e, "ei_context", __FILE__, __LINE__);
      return e == hipErrorOutOfMemory ? HS_ERR_NOMEM : HS_ERR_HIP;        /* c's destructor releases what had been created */
   }
   memset(c.hout, 0, (size_t) EM_ALL_OUT * sizeof(double));
   g_ctx.device = c.device; g_ctx.stream = c.stream; g_ctx.hin = c.hin; g_ctx.din = c.din; g_ctx.hout = c.hout; g_ctx.dout = c.dout;
   g_ctx.seq = 0;
   c.device = -1; c.stream = NULL; c.hin = c.hout = NULL;                  /* ownership moved */
   *out = &g_ctx;
   return HS_OK;
}

}

/* i-th smallest eigenvalue (1-based) of the symmetric n x n matrix A (the triangle at memory positions [j n + i], i >= j, is read:
 * what DSYEVR 'L' reads from a column-major array), n <= 128 (n <= 64: matrix in registers, above: in LDS); eigvec (n, unit norm) may be
 * NULL.  HIPSDP_ERR_ARG for larger n: the caller takes the full decomposition. */
extern "C" int hipsdp_syevi_small(int device, int n, const double* A, int i, double* eigval, double* eigvec)
{
   int nd = 0;
   if ( hipGetDeviceCount(&nd) != hipSuccess || nd <= 0 )
      return HIPSDP_ERR_NODEVICE;
   if ( device < 0 || device >= nd || n < 1 || n > EM_N || A == NULL || i < 1 || i > n || eigval == NULL )
      return HIPSDP_ERR_ARG;
   ei_ctx* c = NULL;
   HS_CALL( ei_context(device, &c) );
   HS_HIP( hipSetDevice(device) );
   memcpy(c->hin, A, (size_t) n * n * sizeof(double));
   const unsigned long long seq = ++c->seq;
   const int flagpos = (n <= EI_N) ? EI_N + 4 : EM_FLAG;
   volatile unsigned long long* flag = reinterpret_cast<volatile unsigned long long*>(c->hout + flagpos);
   if ( n <= EI_N )
      hipLaunchKernelGGL((k_syevi_small<false>), dim3(1), dim3(256), 0, c->stream, n, i, eigvec != NULL ? 1 : 0, c->din, c->dout, seq,
         reinterpret_cast<unsigned long long*>(c->dout + flagpos));
   else
   {
      /* 64 < n <= 128: the matrix in LDS (k_syevi_mid) */
      const int smem = n * (n | 1) * (int) sizeof(double);
      static hs_attr_mask attr_done;
      HS_CALL( hs_func_max_lds(reinterpret_cast<const void*>(&k_syevi_mid), EM_N * (EM_N + 1) * (int) sizeof(double), &attr_done) );
      hipLaunchKernelGGL(k_syevi_mid, dim3(1), dim3(EM_NT), smem, c->stream, n, i, eigvec != NULL ? 1 : 0, c->din, c->dout, seq,
         reinterpret_cast<unsigned long long*>(c->dout + flagpos));
   }
   HS_HIP( hipGetLastError() );
   long long spins = 0;
   while ( *flag != seq )
   {
      if ( (++spins & 0xFFFF) == 0 )
      {
         const hipError_t e = hipStreamQuery(c->stream);
         if ( e == hipSuccess )
         {
            if ( *flag == seq )
               break;
            HS_HIP( hipStreamSynchronize(c->stream) );
            if ( *flag != seq )
               return HIPSDP_ERR_HIP;
            break;
         }
         if ( e != hipErrorNotReady )
         {
            hs_record_hip_error(e, "hipStreamQuery(syevi)", __FILE__, __LINE__);
            return HIPSDP_ERR_HIP;
         }
      }
   }
   __atomic_thread_fence(__ATOMIC_ACQUIRE);
   *eigval = c->hout[0];
   if ( eigvec != NULL )
      memcpy(eigvec, c->hout + 1, (size_t) n * sizeof(double));
   return HIPSDP_OK;
}

/* all eigenpairs of the symmetric n x n matrix A, n <= 128, in one launch through the same staging memory: lam ascending, row k of V =
 * k-th eigenvector (what SCIPlapackComputeEigenvectorDecomposition returns: lapack_interface.c:507-603).  n <= 64: matrix in
 * registers (k_syevi_small<true>), above: in LDS (k_syev_mid).  HIPSDP_ERR_ARG for n > 128. */
extern "C" int hipsdp_syev_small(int device, int n, const double* A, double* lam, double* V)
{
   int nd = 0;
   if ( hipGetDeviceCount(&nd) != hipSuccess || nd <= 0 )
      return HIPSDP_ERR_NODEVICE;
   if ( device < 0 || device >= nd || n < 1 || n > EM_N || A == NULL || lam == NULL )
      return HIPSDP_ERR_ARG;
   ei_ctx* c = NULL;
   HS_CALL( ei_context(device, &c) );
   HS_HIP( hipSetDevice(device) );
   const bool mid = n >= em_mid_from();
   /* the flag word sits behind the eigenvalues and the eigenvector array */
   const long long vecpos = mid ? EM_N : EI_N;
   const long long flagpos = mid ? (long long) EM_ALL_FLAG : EI_N + (long long) EI_N * EI_N + 4;
   if ( mid && c->dscr == NULL )
      HS_HIP( hipMalloc((void**) &c->dscr, (size_t) 3 * EM_N * EM_N * sizeof(double)) );      /* (once per thread and device) */
   memcpy(c->hin, A, (size_t) n * n * sizeof(double));
   const unsigned long long seq = ++c->seq;
   volatile unsigned long long* flag = reinterpret_cast<volatile unsigned long long*>(c->hout + flagpos);
   if ( mid )
   {
      static hs_attr_mask attr_mid;
      HS_CALL( hs_func_max_lds(reinterpret_cast<const void*>(&k_syev_mid), EM_N * (EM_N + 1) * (int) sizeof(double), &attr_mid) );
      hipLaunchKernelGGL(k_syev_mid, dim3(1), dim3(EM_NT), em_all_lds(n), c->stream, n, c->din, c->dout, c->dscr, seq,
         reinterpret_cast<unsigned long long*>(c->dout + flagpos));
   }
   else
   {
      static hs_attr_mask attr_done;
      HS_CALL( hs_func_max_lds(reinterpret_cast<const void*>(&k_syevi_small<true>), EI_ALL_LDS, &attr_done) );
      hipLaunchKernelGGL((k_syevi_small<true>), dim3(1), dim3(256), EI_ALL_LDS, c->stream, n, 0, 1, c->din, c->dout, seq,
         reinterpret_cast<unsigned long long*>(c->dout + flagpos));
   }
   HS_HIP( hipGetLastError() );
   long long spins = 0;
   while ( *flag != seq )
   {
      if ( (++spins & 0xFFFF) == 0 )
      {
         const hipError_t e = hipStreamQuery(c->stream);
         if ( e == hipSuccess )
         {
            if ( *flag == seq )
               break;
            HS_HIP( hipStreamSynchronize(c->stream) );
            if ( *flag != seq )
               return HIPSDP_ERR_HIP;
            break;
         }
         if ( e != hipErrorNotReady )
         {
            hs_record_hip_error(e, "hipStreamQuery(syev_small)", __FILE__, __LINE__);
            return HIPSDP_ERR_HIP;
         }
      }
   }
   __atomic_thread_fence(__ATOMIC_ACQUIRE);
   memcpy(lam, c->hout, (size_t) n * sizeof(double));
   if ( V != NULL )
      memcpy(V, c->hout + vecpos, (size_t) n * n * sizeof(double));
   if ( mid )
   {
      static const bool stamps = getenv("HIPSDP_SYEV_STAMPS") != NULL && atoi(getenv("HIPSDP_SYEV_STAMPS")) != 0;
      if ( stamps )
      {
         const double* t = c->hout + EM_ALL_FLAG + 2;
         fprintf(stderr, "hipsdp syev n = %d, us since kernel start: reduction %.0f, eigenvalues %.0f, rounds %.0f %.0f | %.0f %.0f | %.0f %.0f, back-transformation %.0f\n",
            n, t[0] * 0.01, t[1] * 0.01, t[2] * 0.01, t[3] * 0.01, t[4] * 0.01, t[5] * 0.01, t[6] * 0.01, t[7] * 0.01, t[8] * 0.01);
      }
   }
   return HIPSDP_OK;
}

/* the same kernels on device buffers, in stream order (no staging memory, no polling): lam[n] ascending, V[n][n] with row k = k-th
 * eigenvector; scratch: hs_syev_small_scratch(n) doubles.  For device-side chains that need the decomposition the SCIPlapack entry
 * point returns (psd.hip: the PSD projection of the warm-start producer) - same eigenvectors, same signs.  n <= 128. */
long long hs_syev_small_scratch(int n) { return n < em_mid_from() ? (long long) EI_OUT_DOUBLES : (long long) EM_ALL_OUT + 3LL * EM_N * EM_N; }

int hs_syev_small_dev(hipStream_t st, int n, const double* A, double* lam, double* V, double* scratch)
{
   if ( n < 1 || n > EM_N )
      return HS_ERR_ARG;
   if ( n >= em_mid_from() )
   {
      static hs_attr_mask attr_mid;
      HS_CALL( hs_func_max_lds(reinterpret_cast<const void*>(&k_syev_mid), EM_N * (EM_N + 1) * (int) sizeof(double), &attr_mid) );
      hipLaunchKernelGGL(k_syev_mid, dim3(1), dim3(EM_NT), em_all_lds(n), st, n, A, scratch, scratch + EM_ALL_OUT, 1ULL,
         reinterpret_cast<unsigned long long*>(scratch + EM_ALL_FLAG));
      HS_HIP( hipGetLastError() );
      HS_HIP( hipMemcpyAsync(lam, scratch, (size_t) n * sizeof(double), hipMemcpyDeviceToDevice, st) );
      HS_HIP( hipMemcpyAsync(V, scratch + EM_N, (size_t) n * n * sizeof(double), hipMemcpyDeviceToDevice, st) );
      return HS_OK;
   }
   static hs_attr_mask attr_done;
   HS_CALL( hs_func_max_lds(reinterpret_cast<const void*>(&k_syevi_small<true>), EI_ALL_LDS, &attr_done) );
   const long long flagpos = EI_N + (long long) EI_N * EI_N + 4;
   hipLaunchKernelGGL((k_syevi_small<true>), dim3(1), dim3(256), EI_ALL_LDS, st, n, 0, 1, A, scratch, 1ULL,
      reinterpret_cast<unsigned long long*>(scratch + flagpos));
   HS_HIP( hipGetLastError() );
   HS_HIP( hipMemcpyAsync(lam, scratch, (size_t) n * sizeof(double), hipMemcpyDeviceToDevice, st) );
   HS_HIP( hipMemcpyAsync(V, scratch + EI_N, (size_t) n * n * sizeof(double), hipMemcpyDeviceToDevice, st) );
   return HS_OK;
}

/* lambda_min of the scaled steps of several blocks of 17 .. 64 rows, exactly, in one launch (k_lmin_exact_multi) */
int hs_lmin_exact_multi(hipStream_t st, const hs_step_jobs* P)
{
   if ( P->nblk <= 0 )
      return HS_OK;
   int nmax = 0;
   for (int j = 0; j < P->nblk; ++j)
   {
      if ( P->n[j] < 1 || P->n[j] > HS_SMALL_N )          /* (the products out of LDS give every thread a 2 x 5 patch: hs_lds_product.h) */
         return HS_ERR_ARG;
      if ( P->n[j] > nmax ) nmax = P->n[j];
   }
   static hs_attr_mask attr_done;
   HS_CALL( hs_func_max_lds(reinterpret_cast<const void*>(&k_lmin_exact_multi), 3 * EI_N * EI_LD * (int) sizeof(double), &attr_done) );
   const size_t smem = (size_t) 3 * nmax * (nmax | 1) * sizeof(double);
   hipLaunchKernelGGL(k_lmin_exact_multi, dim3(2, P->nblk), dim3(256), smem, st, *P);
   HS_HIP( hipGetLastError() );
   return HS_OK;
}
