/* sparse.hip - constraint matrices kept as the nonzeros the caller hands over.
 *
 * Both reference backends give their solver only nonzeros (sdpisolver_dsdp.c:1126,1146,1195 SDPConeSetASparseVecMat;
 * sdpisolver_sdpa.cpp:1223-1267 inputElement), and every instance the reference ships has 1-10 nonzeros per matrix.  A block in
 * SPARSE mode stores, for the variables 1 .. m, the lower-triangular triplets sorted by variable (for A(V) and the Schur pairs) and
 * a second time sorted by position (for A^T(coef): every entry of the result sums its contributions in a fixed order - the copies
 * of an SPMD run and the ranks of a sharded one must agree to the last bit, so no floating-point atomics).  The constant matrix
 * stays a dense n x n array (it is dense after fixings / in planted instances, and k_cert wants it dense).
 *
 * Schur complement of such a block, from the nonzeros but IN THE ASSOCIATION OF THE DENSE FORMULA (round 4):
 *
 *      T_j = A_j Zinv          only the non-empty rows of A_j: row p of T_j = sum over the entries (p, q) of A_j[p][q] Zinv[q][:]
 *      Mx[i][j] = tr(A_i X A_j Zinv) = sum over the entries (a, b) of A_i:  A_i[a][b] * sum over the non-empty rows p of A_j: X[b][p] T_j[p][a]
 *
 * Rounds 2 and 3 used SDPA's F3 pair formula (Fujisawa, Kojima, Nakata 1997), sum_e sum_f a b (X_qr Zinv_sp + ...): the same number
 * in exact arithmetic, the same count of multiply-adds (about nnz_i nnz_j per pair) - but on nodes without an optimum (tau -> 0, Zinv
 * almost of rank one) the assembled matrix then differs from the operator the direction applies by 300 times more than with the
 * association above, the primal residual stalls at 1e-7 and the node goes to the settings ladder (found with the one-launch
 * solve on example_TT's infeasible nodes).  The differences of entries of Zinv a constraint matrix asks for are formed first, as the
 * products X dZ Zinv of the direction form them.  The engine keeps a block as nonzeros when 4 (sum nnz)^2 is smaller than the dense
 * count (hs_sp_prefers_sparse).  Row / column 0 of the extended matrix (constant matrix as
 * "variable 0") comes from U_0 = X A_0 Zinv (two dense n^3 products) and one gather pass <A_i, U_0>.
 */
#include "hs_kernels.h"
#include <algorithm>
#include <vector>
#include <numeric>

struct hs_sparse
{
   int        n, m;
   long long  nnz;
   /* by variable: entries of variable v (1 .. m) are [voff[v - 1], voff[v]) */
   int*       voff;
   int*       vrow;
   int*       vcol;
   double*    vval;
   /* by position: position k (a lower-triangular (row, col) that occurs) has the entries [poff[k], poff[k + 1]) */
   long long  npos;
   int*       poff;
   int*       prow;
   int*       pcol;
   int*       pvar;        /* variable (1-based) of an entry */
   double*    pval;
   /* FULL symmetric entries by variable, row-major (the two-stage Schur assembly): entries of variable v are [foff[v - 1], foff[v]),
    * entry e = (frow[e], fcol[e], fval[e]); the non-empty rows of variable v are the slots [soff[v - 1], soff[v]): slot s is row
    * srow[s] with the entries [sent[s], sent[s + 1]); Tc: nslots x n doubles of workspace, Tc[s][c] = (A_v Zinv)[srow[s]][c] */
   long long  nfull, nslots;
   int*       foff;
   int*       frow;
   int*       fcol;
   double*    fval;
   int*       soff;
   int*       srow;
   int*       sent;
   double*    Tc;
};

namespace {

template<typename T> int sp_upload(T** d, const std::vector<T>& h)
{
   *d = NULL;
   const size_t bytes = (h.size() > 0 ? h.size() : 1) * sizeof(T);
   HS_CALL( hs_pool_alloc((void**) d, bytes) );
   if ( !h.empty() )
      HS_HIP( hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) );
   return HS_OK;
}

}

void hs_sp_free(hs_sparse* sp)
{
   if ( sp == NULL )
      return;
   hs_pool_free(sp->voff); hs_pool_free(sp->vrow); hs_pool_free(sp->vcol); hs_pool_free(sp->vval);
   hs_pool_free(sp->poff); hs_pool_free(sp->prow); hs_pool_free(sp->pcol); hs_pool_free(sp->pvar); hs_pool_free(sp->pval);
   hs_pool_free(sp->foff); hs_pool_free(sp->frow); hs_pool_free(sp->fcol); hs_pool_free(sp->fval);
   hs_pool_free(sp->soff); hs_pool_free(sp->srow); hs_pool_free(sp->sent); hs_pool_free(sp->Tc);
   delete sp;
}

/* 4 (sum nnz)^2 multiply-adds of the pair formula against 4 m1 n^3 + m1^2 n^2 of the dense formulation (and never for blocks the
 * single-launch kernels of small problems handle) */
int hs_sp_prefers_sparse(int n, int m, long long nnz)
{
   if ( n <= 64 || m < 1 || nnz < 0 )
      return 0;
   const double sparse = 4.0 * (double) nnz * (double) nnz;
   const double dense = 4.0 * (double) (m + 1) * (double) n * n * n + (double) (m + 1) * (m + 1) * (double) n * n;
   return sparse * 4.0 < dense ? 1 : 0;           /* gathers, not matrix cores: a factor of 4 in favour of the dense path */
}

/* host triplets (var 1 .. m, row >= col or row < col: stored as given with the larger index as row; a later entry with the same
 * (var, row, col) replaces an earlier one, as the dense scatter does) -> device structure */
int hs_sp_build(hs_sparse** out, int n, int m, long long nnz, const int* var, const int* row, const int* col, const double* val)
{
   *out = NULL;
   std::vector<long long> ord;
   ord.reserve((size_t) nnz);
   for (long long e = 0; e < nnz; ++e)
   {
      if ( var[e] < 1 || var[e] > m || row[e] < 0 || row[e] >= n || col[e] < 0 || col[e] >= n )
         return HS_ERR_ARG;
      ord.push_back(e);
   }
   auto R = [&](long long e) { return row[e] >= col[e] ? row[e] : col[e]; };
   auto C = [&](long long e) { return row[e] >= col[e] ? col[e] : row[e]; };
   std::stable_sort(ord.begin(), ord.end(), [&](long long a, long long b) {
      if ( var[a] != var[b] ) return var[a] < var[b];
      if ( R(a) != R(b) ) return R(a) < R(b);
      return C(a) < C(b);
   });
   /* the last of equal keys wins */
   std::vector<int> hv, hr, hc; std::vector<double> hx;
   for (size_t k = 0; k < ord.size(); ++k)
   {
      const long long e = ord[k];
      if ( k + 1 < ord.size() )
      {
         const long long f = ord[k + 1];
         if ( var[e] == var[f] && R(e) == R(f) && C(e) == C(f) )
            continue;
      }
      hv.push_back(var[e]); hr.push_back(R(e)); hc.push_back(C(e)); hx.push_back(val[e]);
   }
   const long long nz = (long long) hv.size();
   /* entries of variable v (1 .. m) are [voff[v - 1], voff[v]) */
   std::vector<int> voff((size_t) m + 1, 0);
   for (long long e = 0; e < nz; ++e)
      ++voff[hv[e]];
   for (int v = 1; v <= m; ++v)
      voff[v] += voff[v - 1];
   /* by position */
   std::vector<long long> po((size_t) nz);
   std::iota(po.begin(), po.end(), 0LL);
   std::stable_sort(po.begin(), po.end(), [&](long long a, long long b) {
      if ( hr[a] != hr[b] ) return hr[a] < hr[b];
      if ( hc[a] != hc[b] ) return hc[a] < hc[b];
      return hv[a] < hv[b];
   });
   std::vector<int> poff, prow, pcol, pvar((size_t) nz);
   std::vector<double> pval((size_t) nz);
   for (long long k = 0; k < nz; ++k)
   {
      const long long e = po[k];
      if ( k == 0 || hr[e] != hr[po[k - 1]] || hc[e] != hc[po[k - 1]] )
      {
         poff.push_back((int) k);
         prow.push_back(hr[e]);
         pcol.push_back(hc[e]);
      }
      pvar[k] = hv[e];
      pval[k] = hx[e];
   }
   const long long npos = (long long) prow.size();
   poff.push_back((int) nz);
   hs_sparse* sp = new hs_sparse();
   sp->n = n; sp->m = m; sp->nnz = nz; sp->npos = npos;
   sp->voff = sp->vrow = sp->vcol = sp->poff = sp->prow = sp->pcol = sp->pvar = NULL;
   sp->vval = sp->pval = NULL;
   sp->foff = sp->frow = sp->fcol = sp->soff = sp->srow = sp->sent = NULL;
   sp->fval = sp->Tc = NULL;
   /* full entries (both triangles) by variable, row-major, and the row slots */
   std::vector<int> foff((size_t) m + 1, 0), frow, fcol, soff((size_t) m + 1, 0), srow, sent;
   std::vector<double> fval;
   {
      std::vector<std::pair<std::pair<int, int>, double> > ent;
      for (int v = 1; v <= m; ++v)
      {
         ent.clear();
         for (int e = voff[v - 1]; e < voff[v]; ++e)
         {
            ent.push_back(std::make_pair(std::make_pair(hr[e], hc[e]), hx[e]));
            if ( hr[e] != hc[e] )
               ent.push_back(std::make_pair(std::make_pair(hc[e], hr[e]), hx[e]));
         }
         std::sort(ent.begin(), ent.end(), [](const std::pair<std::pair<int, int>, double>& a, const std::pair<std::pair<int, int>, double>& b) {
            return a.first < b.first; });
         for (size_t k = 0; k < ent.size(); ++k)
         {
            if ( k == 0 || ent[k].first.first != ent[k - 1].first.first )
            {
               srow.push_back(ent[k].first.first);
               sent.push_back((int) frow.size());
            }
            frow.push_back(ent[k].first.first); fcol.push_back(ent[k].first.second); fval.push_back(ent[k].second);
         }
         foff[v] = (int) frow.size();
         soff[v] = (int) srow.size();
      }
      sent.push_back((int) frow.size());
   }
   sp->nfull = (long long) frow.size();
   sp->nslots = (long long) srow.size();
   int rc = sp_upload(&sp->voff, voff);
   if ( rc == HS_OK ) rc = sp_upload(&sp->foff, foff);
   if ( rc == HS_OK ) rc = sp_upload(&sp->frow, frow);
   if ( rc == HS_OK ) rc = sp_upload(&sp->fcol, fcol);
   if ( rc == HS_OK ) rc = sp_upload(&sp->fval, fval);
   if ( rc == HS_OK ) rc = sp_upload(&sp->soff, soff);
   if ( rc == HS_OK ) rc = sp_upload(&sp->srow, srow);
   if ( rc == HS_OK ) rc = sp_upload(&sp->sent, sent);
   if ( rc == HS_OK ) rc = hs_pool_alloc((void**) &sp->Tc, (size_t) ((sp->nslots > 0 ? sp->nslots : 1) * (long long) n) * sizeof(double));
   if ( rc == HS_OK ) rc = sp_upload(&sp->vrow, hr);
   if ( rc == HS_OK ) rc = sp_upload(&sp->vcol, hc);
   if ( rc == HS_OK ) rc = sp_upload(&sp->vval, hx);
   if ( rc == HS_OK ) rc = sp_upload(&sp->poff, poff);
   if ( rc == HS_OK ) rc = sp_upload(&sp->prow, prow);
   if ( rc == HS_OK ) rc = sp_upload(&sp->pcol, pcol);
   if ( rc == HS_OK ) rc = sp_upload(&sp->pvar, pvar);
   if ( rc == HS_OK ) rc = sp_upload(&sp->pval, pval);
   if ( rc != HS_OK )
   {
      hs_sp_free(sp);
      return rc;
   }
   *out = sp;
   return HS_OK;
}

long long hs_sp_nnz(const hs_sparse* sp) { return sp->nnz; }

/* out[v] = <A_v, V> for v = 1 .. m (out points at the entry of variable 1): one wavefront per variable, entries summed in order */
__global__ void __launch_bounds__(256) k_sp_apply(int m, int n, const int* __restrict__ voff, const int* __restrict__ vrow,
   const int* __restrict__ vcol, const double* __restrict__ vval, const double* __restrict__ V, double* __restrict__ out)
{
   const int lane = threadIdx.x & 63;
   for (int v = blockIdx.x * 4 + (threadIdx.x >> 6); v < m; v += gridDim.x * 4)
   {
      const int e0 = voff[v], e1 = voff[v + 1];
      double acc = 0.0;
      for (int e = e0 + lane; e < e1; e += 64)
      {
         const int r = vrow[e], c = vcol[e];
         const double x = V[(long long) r * n + c];
         acc += vval[e] * (r != c ? x + V[(long long) c * n + r] : x);
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1)
         acc += __shfl_xor(acc, off, 64);
      if ( lane == 0 )
         out[v] = acc;
   }
}

int hs_sp_apply_A(hipStream_t s, const hs_sparse* sp, const double* V, double* out_var1)
{
   if ( sp->m <= 0 )
      return HS_OK;
   int blocks = (sp->m + 3) / 4;
   if ( blocks > 4096 ) blocks = 4096;
   hipLaunchKernelGGL(k_sp_apply, dim3(blocks), dim3(256), 0, s, sp->m, sp->n, sp->voff, sp->vrow, sp->vcol, sp->vval, V, out_var1);
   HS_HIP( hipGetLastError() );
   return HS_OK;
}

/* out[p][q] += sum over the entries at (p, q) of coef[var] val, both triangles; coef points at the coefficient of variable 0 */
__global__ void __launch_bounds__(256) k_sp_apply_t(long long npos, int n, const int* __restrict__ poff, const int* __restrict__ prow,
   const int* __restrict__ pcol, const int* __restrict__ pvar, const double* __restrict__ pval, const double* __restrict__ coef,
   double* __restrict__ out)
{
   for (long long k = (long long) blockIdx.x * blockDim.x + threadIdx.x; k < npos; k += (long long) gridDim.x * blockDim.x)
   {
      double acc = 0.0;
      for (int e = poff[k]; e < poff[k + 1]; ++e)
         acc += coef[pvar[e]] * pval[e];
      const int r = prow[k], c = pcol[k];
      out[(long long) r * n + c] += acc;
      if ( r != c )
         out[(long long) c * n + r] += acc;
   }
}

int hs_sp_apply_AT(hipStream_t s, const hs_sparse* sp, const double* coef, double* out)
{
   if ( sp->npos <= 0 )
      return HS_OK;
   long long blocks = (sp->npos + 255) / 256;
   if ( blocks > 4096 ) blocks = 4096;
   hipLaunchKernelGGL(k_sp_apply_t, dim3((unsigned) blocks), dim3(256), 0, s, sp->npos, sp->n, sp->poff, sp->prow, sp->pcol, sp->pvar, sp->pval,
      coef, out);
   HS_HIP( hipGetLastError() );
   return HS_OK;
}

/* (a) Tc[s][c] = sum over the entries of slot s of value * Zinv[col][c]: one thread per (slot, column), entries in order */
__global__ void __launch_bounds__(256) k_sp_trows(long long nslots, int n, const int* __restrict__ sent, const int* __restrict__ fcol,
   const double* __restrict__ fval, const double* __restrict__ Zinv, double* __restrict__ Tc)
{
   const long long total = nslots * n;
   for (long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long) gridDim.x * blockDim.x)
   {
      const long long sl = t / n;
      const int c = (int) (t - sl * n);
      double acc = 0.0;
      for (int e = sent[sl]; e < sent[sl + 1]; ++e)
         acc = fma(fval[e], Zinv[(long long) fcol[e] * n + c], acc);
      Tc[t] = acc;
   }
}

/* (b) Mx[i][j] += sum over the entries (a, b) of A_i of A_i[a][b] * (sum over the row slots of A_j of X[b][row] Tc[slot][a]) for
 * 1 <= j <= i <= m (lower triangle, ld = m + 1; row / column 0 is not touched).  One wavefront per pair: its lanes split the
 * entries of A_i, partial sums added in a fixed order. */
__global__ void __launch_bounds__(256) k_sp_schur(int m, int n, const int* __restrict__ foff, const int* __restrict__ frow,
   const int* __restrict__ fcol, const double* __restrict__ fval, const int* __restrict__ soff, const int* __restrict__ srow,
   const double* __restrict__ Tc, const double* __restrict__ X, double* __restrict__ Mx, long long npairs)
{
   const int lane = threadIdx.x & 63;
   const int m1 = m + 1;
   for (long long pr = (long long) blockIdx.x * 4 + (threadIdx.x >> 6); pr < npairs; pr += (long long) gridDim.x * 4)
   {
      /* pr -> (i, j), 0 <= j <= i < m */
      long long i = (long long) ((sqrt(8.0 * (double) pr + 1.0) - 1.0) * 0.5);
      while ( (i + 1) * (i + 2) / 2 <= pr ) ++i;
      while ( i * (i + 1) / 2 > pr ) --i;
      const long long j = pr - i * (i + 1) / 2;
      const int a0 = foff[i], a1 = foff[i + 1], s0 = soff[j], s1 = soff[j + 1];
      double acc = 0.0;
      for (int e = a0 + lane; e < a1; e += 64)
      {
         const int a = frow[e], b = fcol[e];
         double u = 0.0;
         for (int sl = s0; sl < s1; ++sl)
            u = fma(X[(long long) b * n + srow[sl]], Tc[(long long) sl * n + a], u);
         acc = fma(fval[e], u, acc);
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1)
         acc += __shfl_xor(acc, off, 64);
      if ( lane == 0 )
         Mx[(i + 1) * m1 + (j + 1)] += acc;
   }
}

/* the same sum by ONE THREAD per pair of variables, its entries in order: with a handful of nonzeros per matrix a wavefront per
 * pair has a few of 64 lanes at work and pays the decoding of the pair index and a six-step reduction for every one of the
 * m (m + 1) / 2 pairs; taken when the average number of entries per matrix is small */
__global__ void __launch_bounds__(256) k_sp_schur_t(int m, int n, const int* __restrict__ foff, const int* __restrict__ frow,
   const int* __restrict__ fcol, const double* __restrict__ fval, const int* __restrict__ soff, const int* __restrict__ srow,
   const double* __restrict__ Tc, const double* __restrict__ X, double* __restrict__ Mx, long long npairs)
{
   const int m1 = m + 1;
   for (long long pr = (long long) blockIdx.x * blockDim.x + threadIdx.x; pr < npairs; pr += (long long) gridDim.x * blockDim.x)
   {
      long long i = (long long) ((sqrt(8.0 * (double) pr + 1.0) - 1.0) * 0.5);
      while ( (i + 1) * (i + 2) / 2 <= pr ) ++i;
      while ( i * (i + 1) / 2 > pr ) --i;
      const long long j = pr - i * (i + 1) / 2;
      const int a0 = foff[i], a1 = foff[i + 1], s0 = soff[j], s1 = soff[j + 1];
      double acc = 0.0;
      for (int e = a0; e < a1; ++e)
      {
         const int a = frow[e], b = fcol[e];
         double u = 0.0;
         for (int sl = s0; sl < s1; ++sl)
            u = fma(X[(long long) b * n + srow[sl]], Tc[(long long) sl * n + a], u);
         acc = fma(fval[e], u, acc);
      }
      Mx[(i + 1) * m1 + (j + 1)] += acc;
   }
}

int hs_sp_schur(hipStream_t s, const hs_sparse* sp, const double* X, const double* Zinv, double* Mx)
{
   if ( sp->m <= 0 )
      return HS_OK;
   const long long npairs = (long long) sp->m * (sp->m + 1) / 2;
   if ( sp->nslots > 0 )
   {
      long long blocks = (sp->nslots * sp->n + 255) / 256;
      if ( blocks > 65536 ) blocks = 65536;
      hipLaunchKernelGGL(k_sp_trows, dim3((unsigned) blocks), dim3(256), 0, s, sp->nslots, sp->n, sp->sent, sp->fcol, sp->fval, Zinv, sp->Tc);
      HS_HIP( hipGetLastError() );
   }
   {
      const double avg = (double) sp->nfull / (double) sp->m;
      if ( avg < 24.0 )
      {
         long long blocks = (npairs + 255) / 256;
         if ( blocks > 65536 ) blocks = 65536;
         hipLaunchKernelGGL(k_sp_schur_t, dim3((unsigned) blocks), dim3(256), 0, s, sp->m, sp->n, sp->foff, sp->frow, sp->fcol, sp->fval,
            sp->soff, sp->srow, sp->Tc, X, Mx, npairs);
         HS_HIP( hipGetLastError() );
         return HS_OK;
      }
   }
   long long blocks = (npairs + 3) / 4;
   if ( blocks > 65536 ) blocks = 65536;
   hipLaunchKernelGGL(k_sp_schur, dim3((unsigned) blocks), dim3(256), 0, s, sp->m, sp->n, sp->foff, sp->frow, sp->fcol, sp->fval,
      sp->soff, sp->srow, sp->Tc, X, Mx, npairs);
   HS_HIP( hipGetLastError() );
   return HS_OK;
}

__global__ void k_sp_expand(long long nnz, int n, const int* __restrict__ pvar_by_var, const int* __restrict__ voff, int m,
   const int* __restrict__ vrow, const int* __restrict__ vcol, const double* __restrict__ vval, double* __restrict__ A)
{
   const long long n2 = (long long) n * n;
   for (int v = blockIdx.x; v < m; v += gridDim.x)
      for (int e = voff[v] + threadIdx.x; e < voff[v + 1]; e += blockDim.x)
      {
         double* a = A + (long long) (v + 1) * n2;
         a[(long long) vrow[e] * n + vcol[e]] = vval[e];
         a[(long long) vcol[e] * n + vrow[e]] = vval[e];
      }
}

int hs_sp_expand(hipStream_t s, const hs_sparse* sp, double* A)
{
   if ( sp->m <= 0 )
      return HS_OK;
   int blocks = sp->m < 4096 ? sp->m : 4096;
   hipLaunchKernelGGL(k_sp_expand, dim3(blocks), dim3(256), 0, s, sp->nnz, sp->n, (const int*) NULL, sp->voff, sp->m, sp->vrow, sp->vcol, sp->vval, A);
   HS_HIP( hipGetLastError() );
   return HS_OK;
}
