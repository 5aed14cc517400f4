/* cpu_ref_dense.c - TEST / BENCH INFRASTRUCTURE (oracle).  C restatement of oracle/ipm_ref.py: hsd_solve for ONE dense block without LP rows -
 * the shape of BASELINE.json's synthetic configurations (n = 500 / m = 1000, n = 1000 / m = 2000) - on the host's BLAS / LAPACK: the
 * compiled CPU figure beside the headline of bench.py (cpu_baseline.kind "own C restatement + OpenBLAS"), as BASELINE.md section 3
 * promises and as the reference's backends are compiled code over BLAS (DSDP 5.8 / SDPA 7.4.4 behind src/sdpi/sdpisolver_dsdp.c:1489-1520,
 * sdpisolver_sdpa.cpp:1600-1670; neither is available in this image).  Only bench.py's CPU leg and tests/ load it; nothing under
 * scip-sdp_amd/ does.
 *
 * Same algorithm, constants and termination rules as oracle/ipm_ref.py (homogeneous self-dual embedding, HKM direction, Mehrotra
 * predictor-corrector, factored elimination of dtau, semidefinite pivot rule 3 for M, each triangular solve with M's factor corrected once,
 * exact lambda_min for the step lengths); the Schur complement in the W formulation of the device path (oracle/ipm_ref.py: schur_block_w;
 * scip-sdp_amd/csrc/schur.hip: hs_schur_W):  X = R R^T, Z^-1 = G^T G,  W_j = G A_j R  (two DTRMM: n^3 each instead of 2 n^3),
 * Mx = W W^T (as DGEMM panels of the lower triangle).  Level-3 BLAS does the flops; the BLAS itself runs single-threaded and every large
 * call is cut into independent pieces under OpenMP (column ranges of a product, ranges of the m + 1 matrices, block columns of the Gram
 * matrix) - one thread pool instead of OpenMP's beside the BLAS's own (which, mixed, cost a factor of six on eight cores).
 * tests/test_cpu_ref.py pins it on the numpy oracle (same iteration count, objective to 1e-8).
 *
 * The BLAS is whatever F77(name) resolves to: oracle/cpu_ref_dense.py builds it against scipy's bundled OpenBLAS (symbols
 * scipy_dgemm_ ..., LP64) - the same hook tests/test_lapack_host_branch_cpu.py uses for the host branch of lapack_interface_hip.c.
 * All matrices are row-major n x n; a row-major buffer is the transpose in Fortran's eyes, which the calls below account for. */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef F77
#define F77(name) scipy_##name##_
#endif
#ifndef BLAS_SET_THREADS
#define BLAS_SET_THREADS scipy_openblas_set_num_threads
#define BLAS_GET_THREADS scipy_openblas_get_num_threads
#endif
typedef int bint;
extern void F77(dgemm)(const char*, const char*, const bint*, const bint*, const bint*, const double*, const double*, const bint*, const double*,
   const bint*, const double*, double*, const bint*);
extern void F77(dgemv)(const char*, const bint*, const bint*, const double*, const double*, const bint*, const double*, const bint*, const double*,
   double*, const bint*);
extern void F77(dtrmm)(const char*, const char*, const char*, const char*, const bint*, const bint*, const double*, const double*, const bint*,
   double*, const bint*);
extern void F77(dtrsm)(const char*, const char*, const char*, const char*, const bint*, const bint*, const double*, const double*, const bint*,
   double*, const bint*);
extern void F77(dsyrk)(const char*, const char*, const bint*, const bint*, const double*, const double*, const bint*, const double*, double*,
   const bint*);
extern void F77(dtrsv)(const char*, const char*, const char*, const bint*, const double*, const bint*, double*, const bint*);
extern void F77(dtrmv)(const char*, const char*, const char*, const bint*, const double*, const bint*, double*, const bint*);
extern void F77(dpotrf)(const char*, const bint*, double*, const bint*, bint*);
extern void F77(dtrtri)(const char*, const char*, const bint*, double*, const bint*, bint*);
extern void F77(dlauum)(const char*, const bint*, double*, const bint*, bint*);
extern void F77(dsyevr)(const char*, const char*, const char*, const bint*, double*, const bint*, const double*, const double*, const bint*,
   const bint*, const double*, bint*, double*, double*, const bint*, bint*, double*, const bint*, bint*, const bint*, bint*);
extern void BLAS_SET_THREADS(int);
extern int BLAS_GET_THREADS(void);

#define ST_OPTIMAL 0
#define ST_DINF 1
#define ST_DUNB 2
#define ST_PDINF 3
#define ST_ITERLIM 4
#define ST_NUMERIC 5

typedef struct
{
   int status, iterations;
   double pobj, dobj, pinf, dinf, gap, mu, tau, kappa;
   double schur_seconds, total_seconds;
} DenseInfo;

static double now_s(void)
{
#ifdef _OPENMP
   return omp_get_wtime();
#else
   return 0.0;
#endif
}

/* [c0, c1) = piece `t` of `nt` of 0 .. len, boundaries at multiples of 8 */
static void piece(long long len, int t, int nt, long long* c0, long long* c1)
{
   const long long per = ((len + nt - 1) / nt + 7) & ~7LL;
   *c0 = per * t < len ? per * t : len;
   *c1 = per * (t + 1) < len ? per * (t + 1) : len;
}
static int nthreads(void)
{
#ifdef _OPENMP
   return omp_get_max_threads();
#else
   return 1;
#endif
}
/* row-major C = alpha A B + beta C, all n x n (Fortran: C^T = B^T A^T, the buffers ARE the transposes); column ranges of the Fortran
 * result (rows of the row-major one) on different threads */
static void mm(int n, double alpha, const double* A, const double* B, double beta, double* C)
{
   const bint N = n;
   const int nt = nthreads();
#pragma omp parallel for schedule(static)
   for (int t = 0; t < nt; ++t)
   {
      long long c0, c1;
      piece(n, t, nt, &c0, &c1);
      const bint cnt = (bint) (c1 - c0);
      if ( cnt > 0 )
         F77(dgemm)("N", "N", &N, &cnt, &N, &alpha, B, &N, A + c0 * n, &N, &beta, C + c0 * n, &N);
   }
}
static void symmetrize(int n, double* M)
{
   for (int i = 0; i < n; ++i)
      for (int j = 0; j < i; ++j)
      {
         const double v = 0.5 * (M[(size_t) i * n + j] + M[(size_t) j * n + i]);
         M[(size_t) i * n + j] = v; M[(size_t) j * n + i] = v;
      }
}
static double dotn(size_t len, const double* a, const double* b)
{
   double s = 0.0;
#pragma omp parallel for reduction(+ : s)
   for (long long i = 0; i < (long long) len; ++i) s += a[i] * b[i];
   return s;
}
/* row-major lower Cholesky factor in place (upper triangle zeroed); returns 0 or LAPACK's info */
static int chol_lower(int n, double* L)
{
   bint N = n, info = 0;
   F77(dpotrf)("U", &N, L, &N, &info);           /* Fortran upper = row-major lower */
   if ( info != 0 ) return (int) info;
   for (int i = 0; i < n; ++i)
      for (int j = i + 1; j < n; ++j) L[(size_t) i * n + j] = 0.0;
   return 0;
}
/* oracle/ipm_ref.py: chol_psd (pivot rule 3) - LAPACK first, the plain loop when a pivot falls below its threshold */
static void chol_psd(int n, const double* M, double* L)
{
   const double regtol = 1e-13;
   memcpy(L, M, sizeof(double) * (size_t) n * n);
   if ( chol_lower(n, L) == 0 )
   {
      int ok = 1;
      for (int k = 0; k < n && ok; ++k)
         ok = L[(size_t) k * n + k] * L[(size_t) k * n + k] > regtol * M[(size_t) k * n + k];
      if ( ok ) return;
   }
   for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j)
         L[(size_t) i * n + j] = (j <= i) ? M[(size_t) i * n + j] : 0.0;
   for (int k = 0; k < n; ++k)
   {
      double d = L[(size_t) k * n + k];
      int zero = 0;
      const double mkk = M[(size_t) k * n + k];
      if ( !(d > regtol * mkk) || !(d > 1e-300) )
      {
         zero = !(d > 1.78e-15 * (double) (k + 1) * mkk);
         d = (mkk > 1e-280) ? regtol * mkk : 1.0;
      }
      const double sd = sqrt(d);
      L[(size_t) k * n + k] = sd;
      if ( zero )
      {
         for (int i = k + 1; i < n; ++i) L[(size_t) i * n + k] = 0.0;
         continue;
      }
      for (int i = k + 1; i < n; ++i) L[(size_t) i * n + k] /= sd;
#pragma omp parallel for schedule(static)
      for (int i = k + 1; i < n; ++i)
      {
         const double lik = L[(size_t) i * n + k];
         if ( lik == 0.0 ) continue;
         for (int j = k + 1; j <= i; ++j) L[(size_t) i * n + j] -= lik * L[(size_t) j * n + k];
      }
   }
}
/* x = M^-1 r, M = L L^T (L row-major lower), each triangular solve corrected once with the factor itself (oracle: msolve) */
static void msolve(int m, const double* L, const double* r, double* x, double* w1, double* w2)
{
   if ( m == 0 ) return;
   const bint M = m, one = 1;
   /* row-major lower L = Fortran upper U = L^T:  L w = r  <=>  U^T w = r */
   memcpy(w1, r, sizeof(double) * (size_t) m);
   F77(dtrsv)("U", "T", "N", &M, L, &M, w1, &one);
   memcpy(w2, w1, sizeof(double) * (size_t) m);
   F77(dtrmv)("U", "T", "N", &M, L, &M, w2, &one);                  /* L w1 */
   for (int i = 0; i < m; ++i) w2[i] = r[i] - w2[i];
   F77(dtrsv)("U", "T", "N", &M, L, &M, w2, &one);
   for (int i = 0; i < m; ++i) w1[i] += w2[i];
   memcpy(x, w1, sizeof(double) * (size_t) m);
   F77(dtrsv)("U", "N", "N", &M, L, &M, x, &one);                   /* L^T v = w */
   memcpy(w2, x, sizeof(double) * (size_t) m);
   F77(dtrmv)("U", "N", "N", &M, L, &M, w2, &one);                  /* L^T v */
   for (int i = 0; i < m; ++i) w2[i] = w1[i] - w2[i];
   F77(dtrsv)("U", "N", "N", &M, L, &M, w2, &one);
   for (int i = 0; i < m; ++i) x[i] += w2[i];
}
/* largest alpha with L L^T + alpha D psd (1e300 when D is psd, NaN when not finite): -1 / lambda_min(L^-1 D L^-T) */
static double max_step_psd(int n, const double* L, const double* D, double* W, double* work, bint lwork, bint* iwork, bint liwork)
{
   const bint N = n;
   const double one = 1.0;
   memcpy(W, D, sizeof(double) * (size_t) n * n);
   /* row-major W1 = L^-1 D:  Fortran W1^T = D L^-T = D (U)^-1 with U = L^T (the buffer of L read by Fortran) */
   const int nt = nthreads();
#pragma omp parallel for schedule(static)
   for (int t = 0; t < nt; ++t)
   {
      long long r0, r1;
      piece(n, t, nt, &r0, &r1);
      const bint cnt = (bint) (r1 - r0);
      if ( cnt > 0 )
         F77(dtrsm)("R", "U", "N", "N", &cnt, &N, &one, L, &N, W + r0, &N);           /* (rows of the Fortran matrix are independent) */
   }
   /* row-major W = W1 L^-T:  Fortran W^T = L^-1 W1^T = U^-T W1^T */
#pragma omp parallel for schedule(static)
   for (int t = 0; t < nt; ++t)
   {
      long long c0, c1;
      piece(n, t, nt, &c0, &c1);
      const bint cnt = (bint) (c1 - c0);
      if ( cnt > 0 )
         F77(dtrsm)("L", "U", "T", "N", &N, &cnt, &one, L, &N, W + c0 * n, &N);       /* (columns are independent) */
   }
   for (size_t e = 0; e < (size_t) n * n; ++e)
      if ( !(fabs(W[e]) < 1e300) ) return NAN;
   symmetrize(n, W);
   bint il = 1, iu = 1, mfound = 0, info = 0, ldz = 1;
   double vl = 0.0, vu = 0.0, abstol = 0.0, lam = 0.0, zdummy = 0.0;
   bint isuppz[2];
   F77(dsyevr)("N", "I", "U", &N, W, &N, &vl, &vu, &il, &iu, &abstol, &mfound, &lam, &zdummy, &ldz, isuppz, work, &lwork, iwork, &liwork, &info);
   if ( info != 0 || mfound != 1 ) return NAN;
   return lam >= 0.0 ? 1e300 : -1.0 / lam;
}

/* A: (m + 1) x n x n row-major, A[0] = constant matrix; b[m].  y_out[m] scaled by 1 / tau.  Returns 0, or -1 when memory ran out. */
int cpu_ref_dense_solve(int m, int n, const double* A, const double* b, double gaptol, double feastol, double pabstol, double infeastol,
   double gamma, int maxiter, int settings, int threads, double* y_out, DenseInfo* info)
{
   const int m1 = m + 1;
   const size_t n2 = (size_t) n * n;
   const bint N = n, M1 = m1, N2 = (bint) n2, one = 1;
   const double done = 1.0, dzero = 0.0;
   const double t_begin = now_s();
   const int blas_threads = BLAS_GET_THREADS();
   BLAS_SET_THREADS(1);
#ifdef _OPENMP
   if ( threads > 0 )
      omp_set_num_threads(threads);
#endif
   const int nt = nthreads();
   double* T = (double*) malloc(sizeof(double) * (size_t) m1 * n2);
   double* mats = (double*) calloc(16 * n2 + 1, sizeof(double));
   double* vec = (double*) calloc((size_t) 20 * (m1 + 1) + 2 * (size_t) m1 * m1 + 8, sizeof(double));
   if ( T == NULL || mats == NULL || vec == NULL ) { free(T); free(mats); free(vec); return -1; }
   double *X = mats, *Z = X + n2, *Zi = Z + n2, *Lx = Zi + n2, *Lz = Lx + n2, *Li = Lz + n2, *Rd = Li + n2, *Bm = Rd + n2, *H = Bm + n2,
      *dX = H + n2, *dZ = dX + n2, *E = dZ + n2, *T1 = E + n2, *T2 = T1 + n2, *dXa = T2 + n2, *dZa = dXa + n2;
   double *y = vec, *rp = y + m1, *AX = rp + m1, *AH = AX + m1, *g = AH + m1, *w = g + m1, *ub = w + m1, *u2 = ub + m1, *u1 = u2 + m1,
      *h = u1 + m1, *dy = h + m1, *wt = dy + m1, *cv = wt + m1, *t1 = cv + m1, *t2 = t1 + m1, *dya = t2 + m1, *dyt = dya + m1;
   double *Mx = dyt + 3 * (m1 + 1), *Lm = Mx + (size_t) m1 * m1;
   /* workspace of DSYEVR (values only, one eigenvalue) */
   bint lwork = -1, liwork = -1, iwq = 0, info_l = 0, mf = 0, il = 1, iu = 1, ldz = 1, isup[2];
   double wq = 0.0, vl = 0.0, vu = 0.0, abst = 0.0, lamq = 0.0, zq = 0.0;
   F77(dsyevr)("N", "I", "U", &N, T1, &N, &vl, &vu, &il, &iu, &abst, &mf, &lamq, &zq, &ldz, isup, &wq, &lwork, &iwq, &liwork, &info_l);
   lwork = (bint) wq + 1; liwork = iwq + 1;
   double* ework = (double*) malloc(sizeof(double) * (size_t) lwork);
   bint* eiwork = (bint*) malloc(sizeof(bint) * (size_t) liwork);

   double normb = 0.0, normC = 0.0;
   for (int i = 0; i < m; ++i) normb += b[i] * b[i];
   normb = sqrt(normb);
   normC = sqrt(dotn(n2, A, A));
   const double xi = fmax(1.0, sqrt(fmax(fmax(normb, normC), 1.0)));
   for (int i = 0; i < n; ++i) { X[(size_t) i * n + i] = xi; Z[(size_t) i * n + i] = xi; }
   double tau = 1.0, kappa = xi * xi;
   const double N1 = (double) (n + 1);
   if ( settings < 0 ) settings = 0;
   if ( settings > 2 ) settings = 2;
   const double gamma_eff = settings == 0 ? gamma : fmin(gamma, settings == 1 ? 0.9 : 0.75);
   const int stall_lim = settings == 0 ? 3 : (settings == 1 ? 5 : 8), nobest_lim = settings == 0 ? 6 : (settings == 1 ? 10 : 15);
   const double sigma_floor = settings == 0 ? 1e-8 : (settings == 1 ? 1e-4 : 1e-2);
   int status = ST_ITERLIM, it = 0, certwait = 0, nstall = 0, sincebest = 0;
   double lastmu = 1e300, alpha_last = 1.0, bestmerit = 1e300, mu = 0, pinf = 0, dinf = 0, gap = 0, pobj = 0, dobj = 0;
   double schur_s = 0.0;

   /* out[i] = <A_i, V>, i = 0 .. m  (A as Fortran matrix n^2 x m1: out = A^T vec V), ranges of i on different threads */
#define APPLY_A(V, out) do { _Pragma("omp parallel for schedule(static)") for (int t_ = 0; t_ < nt; ++t_) { long long i0_, i1_; \
      piece(m1, t_, nt, &i0_, &i1_); const bint c_ = (bint) (i1_ - i0_); \
      if ( c_ > 0 ) F77(dgemv)("T", &N2, &c_, &done, A + (size_t) i0_ * n2, &N2, (V), &one, &dzero, (out) + i0_, &one); } } while (0)
   /* out = sum_i coef[i] A_i, ranges of the n^2 entries on different threads */
#define APPLY_AT(coef, out) do { _Pragma("omp parallel for schedule(static)") for (int t_ = 0; t_ < nt; ++t_) { long long e0_, e1_; \
      piece((long long) n2, t_, nt, &e0_, &e1_); const bint c_ = (bint) (e1_ - e0_); \
      if ( c_ > 0 ) F77(dgemv)("N", &c_, &M1, &done, A + e0_, &N2, (coef), &one, &dzero, (out) + e0_, &one); } } while (0)

   for (it = 0; it <= maxiter; ++it)
   {
      APPLY_A(X, AX);
      double rp2 = 0.0;
      for (int i = 0; i < m; ++i) { rp[i] = b[i] * tau - AX[i + 1]; rp2 += rp[i] * rp[i]; }
      cv[0] = -tau;
      for (int i = 0; i < m; ++i) cv[i + 1] = y[i];
      APPLY_AT(cv, Rd);
      double rd2 = 0.0, xz = 0.0;
#pragma omp parallel for reduction(+ : rd2, xz)
      for (long long e = 0; e < (long long) n2; ++e) { Rd[e] -= Z[e]; rd2 += Rd[e] * Rd[e]; xz += X[e] * Z[e]; }
      pobj = AX[0];
      dobj = 0.0;
      for (int i = 0; i < m; ++i) dobj += b[i] * y[i];
      const double rg = pobj - dobj - kappa;
      mu = (xz + tau * kappa) / N1;
      pinf = sqrt(rp2) / tau / (1.0 + normb);
      const double pabs = sqrt(rp2) / tau;
      dinf = sqrt(rd2) / tau / (1.0 + normC);
      const double dabs_ = sqrt(rd2) / tau;
      gap = fabs(dobj - pobj) / tau;
      if ( pinf <= feastol && (pabstol <= 0.0 || pabs <= pabstol) && dabs_ <= feastol && gap <= gaptol ) { status = ST_OPTIMAL; break; }
      const int certzone = (tau < 1e-2 * fmin(1.0, kappa)) || (mu / (tau * tau) > 1e10);
      if ( certzone )
      {
         double hd2 = 0.0, hp2 = 0.0;
#pragma omp parallel for reduction(+ : hd2)
         for (long long e = 0; e < (long long) n2; ++e) { const double v = Rd[e] + tau * A[e]; hd2 += v * v; }
         for (int i = 0; i < m; ++i) hp2 += AX[i + 1] * AX[i + 1];
         const double hd = sqrt(hd2), hp = sqrt(hp2), big = fmax(fabs(dobj), fabs(pobj));
         const int cand_dunb = dobj < -1e-3 * big, cand_dinf = pobj > 1e-3 * big;
         const int ok_dunb = cand_dunb && hd <= infeastol * (-dobj), ok_dinf = cand_dinf && hp <= infeastol * pobj;
         if ( (ok_dunb || ok_dinf) && (ok_dunb || !cand_dunb || certwait >= 5) && (ok_dinf || !cand_dinf || certwait >= 5) )
         {
            status = (ok_dunb && ok_dinf) ? ST_PDINF : (ok_dunb ? ST_DUNB : ST_DINF);
            break;
         }
         if ( ok_dunb || ok_dinf ) ++certwait;
      }
      if ( it == maxiter ) break;
      if ( mu > 0.9 * lastmu && alpha_last < 1e-2 ) { if ( ++nstall >= stall_lim ) { status = ST_NUMERIC; break; } }
      else nstall = 0;
      lastmu = mu;
      if ( !certzone )
      {
         double merit = fmax(fmax(pinf / feastol, dabs_ / feastol), gap / gaptol);
         if ( pabstol > 0.0 ) merit = fmax(merit, pabs / pabstol);
         if ( merit < 0.9 * bestmerit ) { bestmerit = merit; sincebest = 0; }
         else if ( ++sincebest >= nobest_lim ) { status = ST_NUMERIC; break; }
      }
      /* factorizations: Lz, Lx (row-major lower), Li = Lz^-1, Zinv = Li^T Li */
      memcpy(Lz, Z, sizeof(double) * n2);
      memcpy(Lx, X, sizeof(double) * n2);
      if ( chol_lower(n, Lz) != 0 || chol_lower(n, Lx) != 0 ) { status = ST_NUMERIC; break; }
      memcpy(Li, Lz, sizeof(double) * n2);
      {
         bint inf2 = 0;
         F77(dtrtri)("U", "N", &N, Li, &N, &inf2);               /* Fortran upper U = Lz^T: U^-1 = Lz^-T, i.e. row-major Lz^-1 */
         memcpy(Zi, Li, sizeof(double) * n2);
         F77(dlauum)("U", &N, Zi, &N, &inf2);                    /* U U^T = Li^T Li (upper triangle in Fortran's eyes) */
         for (int i = 0; i < n; ++i)
            for (int j = 0; j < i; ++j) Zi[(size_t) j * n + i] = Zi[(size_t) i * n + j];      /* (Fortran upper = row-major lower) */
      }
      /* Schur complement, W formulation: T = A_stack Lx (one DTRMM over the stack), W_j = Li T_j (one DTRMM per matrix), Mx = W W^T */
      {
         const double ts0 = now_s();
         /* row-major T_j = A_j R: Fortran T_j^T = R^T A_j^T, R^T = the buffer of Lx as an upper triangular matrix; then
          * row-major W_j = G T_j: Fortran W_j^T = T_j^T G^T, G^T = the buffer of Li as an upper triangular matrix */
#pragma omp parallel for schedule(dynamic, 4)
         for (int j = 0; j < m1; ++j)
         {
            double* Tj = T + (size_t) j * n2;
            memcpy(Tj, A + (size_t) j * n2, sizeof(double) * n2);
            F77(dtrmm)("L", "U", "N", "N", &N, &N, &done, Lx, &N, Tj, &N);
            F77(dtrmm)("R", "U", "N", "N", &N, &N, &done, Li, &N, Tj, &N);
         }
         /* Mx = W W^T, lower triangle in row-major = upper in Fortran's eyes, with A_F = W^T (n^2 x m1): block columns [c0, c1) of
          * A_F^T A_F, rows 0 .. c1 - 1 (a few more pieces than threads: the later ones are longer) */
         {
            const int np = 4 * nt;
#pragma omp parallel for schedule(dynamic, 1)
            for (int t = np - 1; t >= 0; --t)
            {
               long long c0, c1;
               piece(m1, t, np, &c0, &c1);
               const bint cnt = (bint) (c1 - c0), rows = (bint) c1;
               if ( cnt > 0 )
                  F77(dgemm)("T", "N", &rows, &cnt, &N2, &done, T, &N2, T + (size_t) c0 * n2, &N2, &dzero, Mx + (size_t) c0 * m1, &M1);
            }
         }
         for (int i = 0; i < m1; ++i)
            for (int j = 0; j < i; ++j) Mx[(size_t) j * m1 + i] = Mx[(size_t) i * m1 + j];     /* (Fortran upper = row-major lower) */
         schur_s += now_s() - ts0;
      }
      for (int i = 0; i < m; ++i) { g[i] = Mx[i + 1]; memcpy(Lm + (size_t) i * m, Mx + (size_t) (i + 1) * m1 + 1, sizeof(double) * (size_t) m); }
      {
         double* Mcopy = (double*) malloc(sizeof(double) * (size_t) (m > 0 ? (size_t) m * m : 1));
         memcpy(Mcopy, Lm, sizeof(double) * (size_t) m * m);
         chol_psd(m, Mcopy, Lm);
         free(Mcopy);
      }
      msolve(m, Lm, g, w, t1, t2);
      msolve(m, Lm, b, ub, t1, t2);
      double bub = 0.0;
      for (int i = 0; i < m; ++i) { u2[i] = ub[i] - w[i]; wt[i + 1] = -w[i]; bub += b[i] * ub[i]; }
      wt[0] = 1.0;
      APPLY_AT(wt, Bm);
      mm(n, 1.0, X, Bm, 0.0, T1);
      mm(n, 1.0, T1, Zi, 0.0, T2);
      const double S0 = dotn(n2, Bm, T2);
      const double den = S0 + kappa / tau + bub;
      int finite = (fabs(den) < 1e300);
      for (int i = 0; i < m; ++i) if ( !(fabs(u2[i]) < 1e300) ) finite = 0;
      if ( !finite ) { status = ST_NUMERIC; break; }
      double sigma = 0.0, eta = 1.0, dta = 0.0, dka = 0.0, dt = 0.0, dk = 0.0;
      int bad = 0;
      for (int pass = 0; pass < 2 && !bad; ++pass)
      {
         const double sigmu = sigma * mu, etk = pass ? dta * dka : 0.0;
         /* H = sigmu Zinv - X - sym((eta X Rd + E) Zinv) */
         if ( pass ) memcpy(T1, E, sizeof(double) * n2);
         mm(n, eta, X, Rd, pass ? 1.0 : 0.0, T1);
         mm(n, 1.0, T1, Zi, 0.0, T2);
#pragma omp parallel for
         for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j)
               H[(size_t) i * n + j] = sigmu * Zi[(size_t) i * n + j] - X[(size_t) i * n + j] - 0.5 * (T2[(size_t) i * n + j] + T2[(size_t) j * n + i]);
         APPLY_A(H, AH);
         for (int i = 0; i < m; ++i) h[i] = AH[i + 1] - eta * rp[i];
         msolve(m, Lm, h, u1, t1, t2);
         const double BH = dotn(n2, Bm, H);
         double wrp = 0.0, bu1 = 0.0;
         for (int i = 0; i < m; ++i) { wrp += w[i] * rp[i]; bu1 += b[i] * u1[i]; }
         const double num = -eta * rg + (sigmu - tau * kappa - etk) / tau - BH - eta * wrp + bu1;
         const double dtau = num / den;
         dyt[0] = -dtau;
         for (int i = 0; i < m; ++i) { dy[i] = u1[i] - u2[i] * dtau; dyt[i + 1] = dy[i]; }
         APPLY_AT(dyt, dZ);
#pragma omp parallel for
         for (long long e = 0; e < (long long) n2; ++e) dZ[e] += eta * Rd[e];
         if ( pass ) memcpy(T1, E, sizeof(double) * n2);
         mm(n, 1.0, X, dZ, pass ? 1.0 : 0.0, T1);
         mm(n, 1.0, T1, Zi, 0.0, T2);
#pragma omp parallel for
         for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j)
               dX[(size_t) i * n + j] = sigmu * Zi[(size_t) i * n + j] - X[(size_t) i * n + j] - 0.5 * (T2[(size_t) i * n + j] + T2[(size_t) j * n + i]);
         const double dkappa = (sigmu - tau * kappa - etk - kappa * dtau) / tau;
         int fin = (fabs(dtau) < 1e300);
         for (int i = 0; i < m; ++i) if ( !(fabs(dy[i]) < 1e300) ) fin = 0;
         if ( !fin ) { bad = 1; break; }
         double a = fmin(max_step_psd(n, Lx, dX, T1, ework, lwork, eiwork, liwork), max_step_psd(n, Lz, dZ, T1, ework, lwork, eiwork, liwork));
         if ( a != a ) { bad = 1; break; }
         if ( dtau < 0.0 ) a = fmin(a, -tau / dtau);
         if ( dkappa < 0.0 ) a = fmin(a, -kappa / dkappa);
         if ( pass == 0 )
         {
            const double aa = fmin(1.0, a);
            sigma = fmin(1.0, fmax(sigma_floor, (1.0 - aa) * (1.0 - aa) * (1.0 - aa)));
            eta = 1.0 - sigma;
            dta = dtau; dka = dkappa;
            mm(n, 1.0, dX, dZ, 0.0, E);                 /* second-order term dXa dZa */
         }
         else
         {
            dt = dtau; dk = dkappa;
            const double alpha = fmin(1.0, gamma_eff * a);
            alpha_last = alpha;
            if ( !(fabs(alpha) < 1e300) ) { bad = 1; break; }
            for (int i = 0; i < m; ++i) y[i] += alpha * dy[i];
            tau += alpha * dt;
            kappa += alpha * dk;
#pragma omp parallel for
            for (long long e = 0; e < (long long) n2; ++e) { X[e] += alpha * dX[e]; Z[e] += alpha * dZ[e]; }
            symmetrize(n, X);
            symmetrize(n, Z);
         }
      }
      if ( bad ) { status = ST_NUMERIC; break; }
      (void) dXa; (void) dZa; (void) dya;
   }
   const double sc = (status == ST_OPTIMAL || status == ST_ITERLIM || status == ST_NUMERIC) ? 1.0 / tau
      : 1.0 / fmax(fmax(fabs(dobj), fabs(pobj)), 1e-300);
   for (int i = 0; i < m; ++i) y_out[i] = y[i] * sc;
   info->status = status; info->iterations = it;
   info->pobj = pobj * sc; info->dobj = dobj * sc; info->pinf = pinf; info->dinf = dinf; info->gap = gap; info->mu = mu;
   info->tau = tau; info->kappa = kappa;
   info->schur_seconds = schur_s;
   info->total_seconds = now_s() - t_begin;
   BLAS_SET_THREADS(blas_threads);
   free(T); free(mats); free(vec); free(ework); free(eiwork);
   return 0;
}
