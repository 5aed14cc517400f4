/* cpu_ref.c - TEST / BENCH INFRASTRUCTURE (oracle).  Plain C restatement of oracle/ipm_ref.py: hsd_solve for B&B-sized problems
 * (blocks of at most 64 rows, a few nonzeros per constraint matrix), one thread, no BLAS / LAPACK.  It exists so that the node
 * solves per second of the device engine on example_TT / example_CLS-sized trees have a COMPILED CPU figure beside them (bench.py:
 * bnb.*.cpu_baseline, kind "own C restatement"), as the reference's backends are compiled code (DSDP 5.8 / SDPA 7.4.4 behind
 * src/sdpi/sdpisolver_dsdp.c:1489-1520, sdpisolver_sdpa.cpp:1600-1670 - neither is available in this image).  Only bench.py's CPU
 * leg and tests/ load it; nothing under scip-sdp_amd/ does.
 *
 * Same algorithm, same constants, same termination rules as oracle/ipm_ref.py (homogeneous self-dual embedding, HKM direction,
 * Mehrotra predictor-corrector, factored elimination of dtau, semidefinite pivot rule 3 for M, each triangular solve with M's
 * factor corrected once, exact lambda_min for the step lengths); like the reference's backends it works on the NONZEROS of the
 * constraint matrices (row lists): T_j = A_j Zinv over the non-empty rows, U_j = X T_j, M_ij = <A_i, U_j^T> - the dense formula
 * with the zeros skipped.  tests/test_oracle_golden.py pins it on the reference-held vectors beside the numpy oracle.
 *
 * Build: gcc -O3 -shared -fPIC -o oracle/libcpu_ref.so oracle/cpu_ref.c -lm   (oracle/cpu_ref.py: build(), called by
 * __graft_entry__.build() and on first use) */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ST_OPTIMAL 0
#define ST_DINF 1
#define ST_DUNB 2
#define ST_PDINF 3
#define ST_ITERLIM 4
#define ST_NUMERIC 5

typedef struct
{
   int n;
   int* roff;      /* per matrix i: row slots voff[i] .. voff[i + 1] */
   int* voff;      /* (m + 2) */
   int* srow;      /* row index of a slot */
   int* eoff;      /* entries of a slot: eoff[s] .. eoff[s + 1] */
   int* ecol;
   double* eval;
   int nslots, nent;
} Blk;

static int chol(int n, double* L)          /* lower triangle in place, row-major n x n; returns 0 or 1 + failing pivot */
{
   for (int k = 0; k < n; ++k)
   {
      double d = L[k * n + k];
      for (int j = 0; j < k; ++j) d -= L[k * n + j] * L[k * n + j];
      if ( !(d > 0.0) ) return k + 1;
      const double sd = sqrt(d);
      L[k * n + k] = sd;
      for (int i = k + 1; i < n; ++i)
      {
         double v = L[i * n + k];
         for (int j = 0; j < k; ++j) v -= L[i * n + j] * L[k * n + j];
         L[i * n + k] = v / sd;
      }
   }
   return 0;
}

static void chol_psd(int n, const double* M, double* L)      /* oracle/ipm_ref.py: chol_psd, pivot rule 3 */
{
   const double regtol = 1e-13;
   for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j)
         L[i * n + j] = (j <= i) ? M[i * n + j] : 0.0;
   for (int k = 0; k < n; ++k)
   {
      double d = L[k * n + k];
      int zero = 0;
      const double mkk = M[k * n + k];
      if ( !(d > regtol * mkk) || !(d > 1e-300) )
      {
         zero = !(d > 1.78e-15 * (double) (k + 1) * mkk);
         d = (mkk > 1e-280) ? regtol * mkk : 1.0;
      }
      const double sd = sqrt(d);
      L[k * n + k] = sd;
      if ( zero )
      {
         for (int i = k + 1; i < n; ++i) L[i * n + k] = 0.0;
         continue;
      }
      for (int i = k + 1; i < n; ++i) L[i * n + k] /= sd;
      for (int i = k + 1; i < n; ++i)
      {
         const double lik = L[i * n + k];
         if ( lik == 0.0 ) continue;
         for (int j = k + 1; j <= i; ++j) L[i * n + j] -= lik * L[j * n + k];
      }
   }
}

static void trsv_lower(int n, const double* L, const double* r, double* x)
{
   for (int i = 0; i < n; ++i)
   {
      double s = r[i];
      for (int j = 0; j < i; ++j) s -= L[i * n + j] * x[j];
      x[i] = s / L[i * n + i];
   }
}
static void trsv_upper_t(int n, const double* L, const double* r, double* x)      /* L^T x = r */
{
   for (int i = n - 1; i >= 0; --i)
   {
      double s = r[i];
      for (int j = i + 1; j < n; ++j) s -= L[j * n + i] * x[j];
      x[i] = s / L[i * n + i];
   }
}
/* x = M^-1 r with each triangular solve corrected once with the factor itself (oracle: msolve) */
static void msolve(int m, const double* Lm, const double* r, double* x, double* w1, double* w2)
{
   if ( m == 0 ) return;
   trsv_lower(m, Lm, r, w1);
   for (int i = 0; i < m; ++i)
   {
      double s = r[i];
      for (int j = 0; j <= i; ++j) s -= Lm[i * m + j] * w1[j];
      w2[i] = s;
   }
   trsv_lower(m, Lm, w2, x);
   for (int i = 0; i < m; ++i) w1[i] += x[i];
   trsv_upper_t(m, Lm, w1, x);
   for (int i = 0; i < m; ++i)
   {
      double s = w1[i];
      for (int j = i; j < m; ++j) s -= Lm[j * m + i] * x[j];
      w2[i] = s;
   }
   double* c = (double*) w2;                      /* correction solved in place below */
   {
      /* c <- L^-T w2, x += c */
      for (int i = m - 1; i >= 0; --i)
      {
         double s = c[i];
         for (int j = i + 1; j < m; ++j) s -= Lm[j * m + i] * c[j];
         c[i] = s / Lm[i * m + i];
      }
      for (int i = 0; i < m; ++i) x[i] += c[i];
   }
}

/* smallest eigenvalue of the symmetric n x n matrix W (destroyed): Householder tridiagonalisation + bisection */
static double lambda_min(int n, double* W, double* d, double* e, double* v, double* p)
{
   for (int i = 0; i < n * n; ++i)
      if ( !(fabs(W[i]) < 1e300) ) return NAN;
   for (int k = 0; k + 2 < n; ++k)
   {
      double s2 = 0.0;
      for (int i = k + 2; i < n; ++i) s2 += W[i * n + k] * W[i * n + k];
      const double x0 = W[(k + 1) * n + k];
      d[k] = W[k * n + k];
      if ( !(s2 > 1e-290) ) { e[k] = x0; continue; }
      const double h = sqrt(x0 * x0 + s2);
      const double beta = x0 > 0.0 ? -h : h;
      const double t = (beta - x0) / beta;
      const double sc = 1.0 / (x0 - beta);
      v[k + 1] = 1.0;
      for (int i = k + 2; i < n; ++i) v[i] = W[i * n + k] * sc;
      e[k] = beta;
      double pv = 0.0;
      for (int i = k + 1; i < n; ++i)
      {
         double s = 0.0;
         for (int j = k + 1; j < n; ++j) s += W[(i >= j ? i * n + j : j * n + i)] * v[j];
         p[i] = t * s;
         pv += p[i] * v[i];
      }
      const double al = -0.5 * t * pv;
      for (int i = k + 1; i < n; ++i) p[i] += al * v[i];
      for (int i = k + 1; i < n; ++i)
         for (int j = k + 1; j <= i; ++j)
            W[i * n + j] -= v[i] * p[j] + p[i] * v[j];
   }
   if ( n >= 2 ) { d[n - 2] = W[(n - 2) * n + n - 2]; e[n - 2] = W[(n - 1) * n + n - 2]; }
   d[n - 1] = W[(n - 1) * n + n - 1];
   e[n - 1] = 0.0;
   double lo = 1e300, nrm = 0.0;
   for (int i = 0; i < n; ++i)
   {
      const double rad = (i > 0 ? fabs(e[i - 1]) : 0.0) + fabs(e[i]);
      if ( d[i] - rad < lo ) lo = d[i] - rad;
      if ( fabs(d[i]) + rad > nrm ) nrm = fabs(d[i]) + rad;
   }
   if ( !(lo < 0.0) ) return 0.0;
   double hi = 0.0;
   /* is there an eigenvalue below zero?  Sturm count at 0 */
   for (int it = 0; it < 200; ++it)
   {
      const double x = (it == 0) ? 0.0 : 0.5 * (lo + hi);
      int neg = 0;
      double q = d[0] - x;
      if ( q < 0.0 ) ++neg;
      for (int i = 1; i < n && !neg; ++i)
      {
         if ( fabs(q) < 1e-300 ) q = -1e-300;
         q = d[i] - x - e[i - 1] * e[i - 1] / q;
         if ( q < 0.0 ) ++neg;
      }
      if ( it == 0 ) { if ( !neg ) return 0.0; continue; }
      if ( neg ) hi = x; else lo = x;
      if ( hi - lo <= 1e-13 * fabs(lo) + 1e-300 ) break;
   }
   (void) nrm;
   return lo;
}

typedef struct
{
   int status, iterations;
   double pobj, dobj, pinf, dinf, gap, mu, tau, kappa;
} CpuInfo;

/* A: nblk arrays (m + 1) x n_k x n_k (row-major, A[0] = constant matrix); Dext: q x (m + 1) (column 0 = c); b[m];
 * out: y[m] scaled by 1 / tau (rays: normalised by the objective they certify).  settings 0 / 1 / 2 as Params.settings. */
int cpu_ref_solve(int m, int nblk, const int* ns, const double* const* A, int q, const double* Dext, const double* b, double gaptol,
   double feastol, double pabstol, double infeastol, double gamma, int maxiter, int settings, double* y_out, CpuInfo* info)
{
   const int m1 = m + 1;
   int nmax = 1, N = q;
   for (int k = 0; k < nblk; ++k) { if ( ns[k] > nmax ) nmax = ns[k]; N += ns[k]; }
   const double N1 = (double) (N + 1);
   /* nonzero structure: per matrix the non-empty rows and their entries (both triangles) */
   Blk* B = (Blk*) calloc((size_t) (nblk > 0 ? nblk : 1), sizeof(Blk));
   for (int k = 0; k < nblk; ++k)
   {
      const int n = ns[k];
      int ne = 0, nsl = 0;
      for (int i = 0; i < m1; ++i)
         for (int r = 0; r < n; ++r)
         {
            int any = 0;
            for (int c = 0; c < n; ++c)
               if ( A[k][((size_t) i * n + r) * n + c] != 0.0 ) { ++ne; any = 1; }
            nsl += any;
         }
      B[k].n = n; B[k].nslots = nsl; B[k].nent = ne;
      B[k].voff = (int*) malloc((size_t) (m1 + 1) * sizeof(int));
      B[k].srow = (int*) malloc((size_t) (nsl + 1) * sizeof(int));
      B[k].eoff = (int*) malloc((size_t) (nsl + 1) * sizeof(int));
      B[k].ecol = (int*) malloc((size_t) (ne + 1) * sizeof(int));
      B[k].eval = (double*) malloc((size_t) (ne + 1) * sizeof(double));
      int s = 0, e = 0;
      for (int i = 0; i < m1; ++i)
      {
         B[k].voff[i] = s;
         for (int r = 0; r < n; ++r)
         {
            int any = 0;
            for (int c = 0; c < n; ++c)
            {
               const double v = A[k][((size_t) i * n + r) * n + c];
               if ( v != 0.0 )
               {
                  if ( !any ) { B[k].srow[s] = r; B[k].eoff[s] = e; any = 1; }
                  B[k].ecol[e] = c; B[k].eval[e] = v; ++e;
               }
            }
            if ( any ) ++s;
         }
      }
      B[k].voff[m1] = s;
      B[k].eoff[s] = e;
   }
   /* LP rows as lists */
   int* roff = (int*) malloc((size_t) (q + 1) * sizeof(int));
   int nzd = 0;
   for (int r = 0; r < q; ++r) for (int c = 0; c < m1; ++c) if ( Dext[(size_t) r * m1 + c] != 0.0 ) ++nzd;
   int* rcol = (int*) malloc((size_t) (nzd + 1) * sizeof(int));
   double* rval = (double*) malloc((size_t) (nzd + 1) * sizeof(double));
   {
      int e = 0;
      for (int r = 0; r < q; ++r)
      {
         roff[r] = e;
         for (int c = 0; c < m1; ++c)
            if ( Dext[(size_t) r * m1 + c] != 0.0 ) { rcol[e] = c; rval[e] = Dext[(size_t) r * m1 + c]; ++e; }
      }
      roff[q] = e;
   }
   /* storage */
   size_t tot = 0;
   for (int k = 0; k < nblk; ++k) tot += (size_t) ns[k] * ns[k];
   const size_t NM = 14;
   double* mats = (double*) calloc(NM * tot + 1, sizeof(double));
   double **X = (double**) malloc(sizeof(double*) * NM * (size_t) (nblk + 1));
   double **Z = X + nblk, **Zi = Z + nblk, **Lx = Zi + nblk, **Lz = Lx + nblk, **Rd = Lz + nblk, **Bm = Rd + nblk, **H = Bm + nblk,
      **dX = H + nblk, **dZ = dX + nblk, **E = dZ + nblk, **T1 = E + nblk, **T2 = T1 + nblk, **Wk = T2 + nblk;
   {
      double* p = mats;
      double*** all[14] = {&X, &Z, &Zi, &Lx, &Lz, &Rd, &Bm, &H, &dX, &dZ, &E, &T1, &T2, &Wk};
      for (int a = 0; a < 14; ++a)
         for (int k = 0; k < nblk; ++k) { (*all[a])[k] = p; p += (size_t) ns[k] * ns[k]; }
   }
   double* vec = (double*) calloc((size_t) (24 * (m1 + 1) + 12 * (q + 1) + 8 * nmax + (size_t) m1 * m1 + (size_t) m * m + 8), sizeof(double));
   double *y = vec, *rp = y + m1, *AX = rp + m1, *AH = AX + m1, *g = AH + m1, *w = g + m1, *ub = w + m1, *u2 = ub + m1, *u1 = u2 + m1,
      *h = u1 + m1, *dy = h + m1, *wt = dy + m1, *cv = wt + m1, *t1 = cv + m1, *t2 = t1 + m1, *dya = t2 + m1;
   double *x = dya + m1 + 8 * (m1 + 1), *z = x + q + 1, *rd = z + q + 1, *beta = rd + q + 1, *hl = beta + q + 1, *dx = hl + q + 1, *dz = dx + q + 1,
      *elp = dz + q + 1, *dxa = elp + q + 1, *dza = dxa + q + 1;
   double *ed = dza + q + 1 + (q + 1), *ee = ed + nmax, *ev = ee + nmax, *ep = ev + nmax;
   double *Mx = ep + nmax + 4 * nmax, *Lm = Mx + (size_t) m1 * m1;

   double normb = 0.0, normC = 0.0;
   for (int i = 0; i < m; ++i) normb += b[i] * b[i];
   normb = sqrt(normb);
   for (int k = 0; k < nblk; ++k) for (int e = 0; e < ns[k] * ns[k]; ++e) normC += A[k][e] * A[k][e];
   for (int r = 0; r < q; ++r) normC += Dext[(size_t) r * m1] * Dext[(size_t) r * m1];
   normC = sqrt(normC);
   const double xi = fmax(1.0, sqrt(fmax(fmax(normb, normC), 1.0)));
   for (int k = 0; k < nblk; ++k) for (int i = 0; i < ns[k]; ++i) { X[k][i * ns[k] + i] = xi; Z[k][i * ns[k] + i] = xi; }
   for (int r = 0; r < q; ++r) { x[r] = xi; z[r] = xi; }
   double tau = 1.0, kappa = xi * xi;
   if ( settings < 0 ) settings = 0;
   if ( settings > 2 ) settings = 2;
   const double gamma_eff = settings == 0 ? gamma : fmin(gamma, settings == 1 ? 0.9 : 0.75);
   const int stall_lim = settings == 0 ? 3 : (settings == 1 ? 5 : 8), nobest_lim = settings == 0 ? 6 : (settings == 1 ? 10 : 15);
   const double sigma_floor = settings == 0 ? 1e-8 : (settings == 1 ? 1e-4 : 1e-2);
   int status = ST_ITERLIM, it = 0, certwait = 0, nstall = 0, sincebest = 0;
   double lastmu = 1e300, alpha_last = 1.0, bestmerit = 1e300, mu = 0, pinf = 0, dinf = 0, gap = 0, pobj = 0, dobj = 0;

#define FOR_ENT(Bk, i, body) for (int s_ = (Bk).voff[i]; s_ < (Bk).voff[(i) + 1]; ++s_) { const int r_ = (Bk).srow[s_]; \
      for (int e_ = (Bk).eoff[s_]; e_ < (Bk).eoff[s_ + 1]; ++e_) { const int c_ = (Bk).ecol[e_]; const double a_ = (Bk).eval[e_]; body } }
   /* out[i] = sum_k <A_i, V_k> + (Dext^T lp)_i */
#define APPLY_A(V, lp, out) do { for (int i_ = 0; i_ < m1; ++i_) { double s = 0.0; for (int k = 0; k < nblk; ++k) { const int n = ns[k]; \
      FOR_ENT(B[k], i_, s += a_ * (V)[k][r_ * n + c_];) } (out)[i_] = s; } \
      for (int r = 0; r < q; ++r) for (int e = roff[r]; e < roff[r + 1]; ++e) (out)[rcol[e]] += rval[e] * (lp)[r]; } while (0)
   /* out_k = sum_i coef[i] A_i + sa add_k */
#define APPLY_AT(coef, sa, add, out) do { for (int k = 0; k < nblk; ++k) { const int n = ns[k]; \
      for (int e = 0; e < n * n; ++e) (out)[k][e] = (add) != NULL ? (sa) * ((double**) (add))[k][e] : 0.0; \
      for (int i_ = 0; i_ < m1; ++i_) { const double ci = (coef)[i_]; if ( ci == 0.0 ) continue; FOR_ENT(B[k], i_, (out)[k][r_ * n + c_] += ci * a_;) } } } while (0)
#define MATMUL(n, Pm, Qm, Om) do { for (int i = 0; i < (n); ++i) for (int j = 0; j < (n); ++j) { double s = 0.0; \
      for (int l = 0; l < (n); ++l) { s += (Pm)[i * (n) + l] * (Qm)[l * (n) + j]; } \
      (Om)[i * (n) + j] = s; } } while (0)

   for (it = 0; it <= maxiter; ++it)
   {
      APPLY_A(X, x, AX);
      double rp2 = 0.0;
      for (int i = 0; i < m; ++i) { rp[i] = b[i] * tau - AX[i + 1]; rp2 += rp[i] * rp[i]; }
      cv[0] = -tau;
      for (int i = 0; i < m; ++i) cv[i + 1] = y[i];
      {
         double** none = NULL;
         APPLY_AT(cv, 0.0, none, Rd);
      }
      double rd2 = 0.0, rdmax = 0.0, xz = 0.0;
      for (int k = 0; k < nblk; ++k)
      {
         double bk = 0.0;
         for (int e = 0; e < ns[k] * ns[k]; ++e) { Rd[k][e] -= Z[k][e]; bk += Rd[k][e] * Rd[k][e]; xz += X[k][e] * Z[k][e]; }
         rd2 += bk;
         if ( sqrt(bk) > rdmax ) rdmax = sqrt(bk);
      }
      for (int r = 0; r < q; ++r)
      {
         double s = 0.0;
         for (int e = roff[r]; e < roff[r + 1]; ++e) s += rval[e] * cv[rcol[e]];
         rd[r] = s - z[r];
         rd2 += rd[r] * rd[r];
         if ( fabs(rd[r]) > rdmax ) rdmax = fabs(rd[r]);
         xz += x[r] * z[r];
      }
      pobj = AX[0];
      dobj = 0.0;
      for (int i = 0; i < m; ++i) dobj += b[i] * y[i];
      const double rg = pobj - dobj - kappa;
      mu = (xz + tau * kappa) / N1;
      pinf = sqrt(rp2) / tau / (1.0 + normb);
      const double pabs = sqrt(rp2) / tau;
      dinf = sqrt(rd2) / tau / (1.0 + normC);
      const double dabs_ = rdmax / tau;
      gap = fabs(dobj - pobj) / tau;
      if ( pinf <= feastol && (pabstol <= 0.0 || pabs <= pabstol) && dabs_ <= feastol && gap <= gaptol ) { status = ST_OPTIMAL; break; }
      const int certzone = (tau < 1e-2 * fmin(1.0, kappa)) || (mu / (tau * tau) > 1e10);
      if ( certzone )
      {
         double hd2 = 0.0, hp2 = 0.0;
         for (int k = 0; k < nblk; ++k) for (int e = 0; e < ns[k] * ns[k]; ++e) { const double v = Rd[k][e] + tau * A[k][e]; hd2 += v * v; }
         for (int r = 0; r < q; ++r) { const double v = rd[r] + tau * Dext[(size_t) r * m1]; hd2 += v * v; }
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
      /* factorizations */
      int bad = 0;
      for (int k = 0; k < nblk && !bad; ++k)
      {
         const int n = ns[k];
         memcpy(Lz[k], Z[k], sizeof(double) * (size_t) n * n);
         memcpy(Lx[k], X[k], sizeof(double) * (size_t) n * n);
         if ( chol(n, Lz[k]) || chol(n, Lx[k]) ) { bad = 1; break; }
         /* Li = Lz^-1 in T1, Zinv = Li^T Li */
         for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i)
            {
               double s = (i == j) ? 1.0 : 0.0;
               if ( i < j ) { T1[k][i * n + j] = 0.0; continue; }
               for (int l = j; l < i; ++l) s -= Lz[k][i * n + l] * T1[k][l * n + j];
               T1[k][i * n + j] = s / Lz[k][i * n + i];
            }
         for (int i = 0; i < n; ++i)
            for (int j = 0; j <= i; ++j)
            {
               double s = 0.0;
               for (int l = i; l < n; ++l) s += T1[k][l * n + i] * T1[k][l * n + j];
               Zi[k][i * n + j] = s; Zi[k][j * n + i] = s;
            }
      }
      if ( bad ) { status = ST_NUMERIC; break; }
      /* Schur complement from the nonzeros: T_j = A_j Zinv (non-empty rows), U_j = X T_j, Mx[i][j] = sum A_i[a][b] U_j[b][a] */
      for (int e = 0; e < m1 * m1; ++e) Mx[e] = 0.0;
      for (int k = 0; k < nblk; ++k)
      {
         const int n = ns[k];
         for (int j = 0; j < m1; ++j)
         {
            const int s0 = B[k].voff[j], s1 = B[k].voff[j + 1];
            if ( s1 == s0 ) continue;
            /* T rows into T2 (slot-major), U into Wk */
            for (int s = s0; s < s1; ++s)
               for (int c = 0; c < n; ++c)
               {
                  double tt = 0.0;
                  for (int e = B[k].eoff[s]; e < B[k].eoff[s + 1]; ++e) tt += B[k].eval[e] * Zi[k][B[k].ecol[e] * n + c];
                  T2[k][(s - s0) * n + c] = tt;
               }
            for (int r = 0; r < n; ++r)
               for (int c = 0; c < n; ++c)
               {
                  double u = 0.0;
                  for (int s = s0; s < s1; ++s) u += X[k][r * n + B[k].srow[s]] * T2[k][(s - s0) * n + c];
                  Wk[k][r * n + c] = u;
               }
            for (int i = j; i < m1; ++i)
            {
               double s = 0.0;
               FOR_ENT(B[k], i, s += a_ * Wk[k][c_ * n + r_];)
               Mx[i * m1 + j] += s;
            }
         }
      }
      for (int r = 0; r < q; ++r)
      {
         const double sx = x[r] / z[r];
         for (int e = roff[r]; e < roff[r + 1]; ++e)
            for (int f = roff[r]; f <= e; ++f)
               Mx[rcol[e] * m1 + rcol[f]] += sx * rval[e] * rval[f];
      }
      for (int i = 0; i < m1; ++i) for (int j = 0; j < i; ++j) Mx[j * m1 + i] = Mx[i * m1 + j];
      for (int i = 0; i < m; ++i) { g[i] = Mx[i + 1]; for (int j = 0; j < m; ++j) Lm[i * m + j] = Mx[(i + 1) * m1 + j + 1]; }
      {
         double* Mcopy = (double*) malloc(sizeof(double) * (size_t) (m > 0 ? m * m : 1));
         memcpy(Mcopy, Lm, sizeof(double) * (size_t) m * m);
         chol_psd(m, Mcopy, Lm);
         free(Mcopy);
      }
      msolve(m, Lm, g, w, t1, t2);
      msolve(m, Lm, b, ub, t1, t2);
      double bub = 0.0, S0 = 0.0;
      for (int i = 0; i < m; ++i) { u2[i] = ub[i] - w[i]; wt[i + 1] = -w[i]; bub += b[i] * ub[i]; }
      wt[0] = 1.0;
      {
         double** none = NULL;
         APPLY_AT(wt, 0.0, none, Bm);
      }
      for (int k = 0; k < nblk; ++k)
      {
         const int n = ns[k];
         MATMUL(n, X[k], Bm[k], T1[k]);
         MATMUL(n, T1[k], Zi[k], T2[k]);
         for (int e = 0; e < n * n; ++e) S0 += Bm[k][e] * T2[k][e];
      }
      for (int r = 0; r < q; ++r)
      {
         double s = 0.0;
         for (int e = roff[r]; e < roff[r + 1]; ++e) s += rval[e] * wt[rcol[e]];
         beta[r] = s;
         S0 += x[r] / z[r] * s * s;
      }
      const double den = S0 + kappa / tau + bub;
      int finite = (fabs(den) < 1e300);
      for (int i = 0; i < m; ++i) if ( !(fabs(u2[i]) < 1e300) ) finite = 0;
      if ( !finite ) { status = ST_NUMERIC; break; }
      double sigma = 0.0, eta = 1.0, dta = 0.0, dka = 0.0, dt = 0.0, dk = 0.0, alpha = 0.0;
      for (int pass = 0; pass < 2; ++pass)
      {
         const int useE = pass;
         const double sigmu = sigma * mu, etk = pass ? dta * dka : 0.0;
         for (int k = 0; k < nblk; ++k)
         {
            const int n = ns[k];
            MATMUL(n, X[k], Rd[k], T1[k]);
            for (int e = 0; e < n * n; ++e) T1[k][e] = eta * T1[k][e] + (useE ? E[k][e] : 0.0);
            MATMUL(n, T1[k], Zi[k], T2[k]);
            for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j)
               H[k][i * n + j] = sigmu * Zi[k][i * n + j] - X[k][i * n + j] - 0.5 * (T2[k][i * n + j] + T2[k][j * n + i]);
         }
         for (int r = 0; r < q; ++r) hl[r] = sigmu / z[r] - x[r] - (eta * x[r] * rd[r] + (useE ? elp[r] : 0.0)) / z[r];
         APPLY_A(H, hl, AH);
         for (int i = 0; i < m; ++i) h[i] = AH[i + 1] - eta * rp[i];
         msolve(m, Lm, h, u1, t1, t2);
         double BH = 0.0, wrp = 0.0, bu1 = 0.0;
         for (int k = 0; k < nblk; ++k) for (int e = 0; e < ns[k] * ns[k]; ++e) BH += Bm[k][e] * H[k][e];
         for (int r = 0; r < q; ++r) BH += beta[r] * hl[r];
         for (int i = 0; i < m; ++i) { wrp += w[i] * rp[i]; bu1 += b[i] * u1[i]; }
         const double num = -eta * rg + (sigmu - tau * kappa - etk) / tau - BH - eta * wrp + bu1;
         const double dtau = num / den;
         cv[0] = -dtau;
         for (int i = 0; i < m; ++i) { dy[i] = u1[i] - u2[i] * dtau; cv[i + 1] = dy[i]; }
         APPLY_AT(cv, eta, Rd, dZ);
         for (int r = 0; r < q; ++r)
         {
            double s = 0.0;
            for (int e = roff[r]; e < roff[r + 1]; ++e) s += rval[e] * cv[rcol[e]];
            dz[r] = s + eta * rd[r];
         }
         for (int k = 0; k < nblk; ++k)
         {
            const int n = ns[k];
            MATMUL(n, X[k], dZ[k], T1[k]);
            if ( useE ) for (int e = 0; e < n * n; ++e) T1[k][e] += E[k][e];
            MATMUL(n, T1[k], Zi[k], T2[k]);
            for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j)
               dX[k][i * n + j] = sigmu * Zi[k][i * n + j] - X[k][i * n + j] - 0.5 * (T2[k][i * n + j] + T2[k][j * n + i]);
         }
         for (int r = 0; r < q; ++r) dx[r] = sigmu / z[r] - x[r] - (x[r] * dz[r] + (useE ? elp[r] : 0.0)) / z[r];
         const double dkappa = (sigmu - tau * kappa - etk - kappa * dtau) / tau;
         if ( !(fabs(dtau) < 1e300) ) { finite = 0; break; }
         /* step length */
         double a = 1e300;
         for (int k = 0; k < nblk; ++k)
         {
            const int n = ns[k];
            for (int side = 0; side < 2; ++side)
            {
               const double* L = side ? Lz[k] : Lx[k];
               const double* Dm = side ? dZ[k] : dX[k];
               /* W = L^-1 D L^-T: forward substitutions */
               for (int c = 0; c < n; ++c)
                  for (int i = 0; i < n; ++i)
                  {
                     double s = Dm[i * n + c];
                     for (int l = 0; l < i; ++l) s -= L[i * n + l] * T1[k][l * n + c];
                     T1[k][i * n + c] = s / L[i * n + i];
                  }
               for (int c = 0; c < n; ++c)
                  for (int i = 0; i < n; ++i)
                  {
                     double s = T1[k][c * n + i];
                     for (int l = 0; l < i; ++l) s -= L[i * n + l] * Wk[k][l * n + c];
                     Wk[k][i * n + c] = s / L[i * n + i];
                  }
               for (int i = 0; i < n; ++i) for (int j = 0; j < i; ++j) { const double sm = 0.5 * (Wk[k][i * n + j] + Wk[k][j * n + i]); Wk[k][i * n + j] = sm; Wk[k][j * n + i] = sm; }
               const double lm = lambda_min(n, Wk[k], ed, ee, ev, ep);
               if ( lm != lm ) a = NAN;
               else if ( lm < 0.0 && -1.0 / lm < a ) a = -1.0 / lm;
            }
         }
         for (int r = 0; r < q; ++r)
         {
            if ( dx[r] < 0.0 && -x[r] / dx[r] < a ) a = -x[r] / dx[r];
            if ( dz[r] < 0.0 && -z[r] / dz[r] < a ) a = -z[r] / dz[r];
         }
         if ( dtau < 0.0 && -tau / dtau < a ) a = -tau / dtau;
         if ( dkappa < 0.0 && -kappa / dkappa < a ) a = -kappa / dkappa;
         if ( a != a ) { finite = 0; break; }
         if ( pass == 0 )
         {
            const double aa = fmin(1.0, a);
            sigma = fmin(1.0, fmax(sigma_floor, (1.0 - aa) * (1.0 - aa) * (1.0 - aa)));
            eta = 1.0 - sigma;
            dta = dtau; dka = dkappa;
            for (int k = 0; k < nblk; ++k) MATMUL(ns[k], dX[k], dZ[k], E[k]);
            for (int r = 0; r < q; ++r) elp[r] = dx[r] * dz[r];
         }
         else
         {
            dt = dtau; dk = dkappa;
            alpha = fmin(1.0, gamma_eff * a);
         }
      }
      if ( !finite ) { status = ST_NUMERIC; break; }
      alpha_last = alpha;
      for (int i = 0; i < m; ++i) y[i] += alpha * dy[i];
      tau += alpha * dt;
      kappa += alpha * dk;
      for (int k = 0; k < nblk; ++k) for (int e = 0; e < ns[k] * ns[k]; ++e) { X[k][e] += alpha * dX[k][e]; Z[k][e] += alpha * dZ[k][e]; }
      for (int r = 0; r < q; ++r) { x[r] += alpha * dx[r]; z[r] += alpha * dz[r]; }
   }
   double sc;
   if ( status == ST_OPTIMAL || status == ST_ITERLIM || status == ST_NUMERIC ) sc = 1.0 / tau;
   else sc = 1.0 / fmax(fmax(fabs(dobj), fabs(pobj)), 1e-300);
   for (int i = 0; i < m; ++i) y_out[i] = y[i] * sc;
   info->status = status; info->iterations = it; info->pobj = pobj * sc; info->dobj = dobj * sc; info->pinf = pinf; info->dinf = dinf;
   info->gap = gap; info->mu = mu; info->tau = tau; info->kappa = kappa;
   for (int k = 0; k < nblk; ++k) { free(B[k].voff); free(B[k].srow); free(B[k].eoff); free(B[k].ecol); free(B[k].eval); }
   free(B); free(roff); free(rcol); free(rval); free(mats); free(X); free(vec);
   return 0;
}
