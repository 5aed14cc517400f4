/* sparse.hip - constraint matrices kept as the nonzeros the caller hands over.
 *
 * Both reference backends give their solver only nonzeros (sdpisolver_dsdp.c:1126,1146,1195 SDPConeSetASparseVecMat;
 * sdpisolver_sdpa.cpp:1223-1267 inputElement), and every instance the reference ships has 1-10 nonzeros per matrix.  A block in
 * SPARSE mode stores, for the variables 1 .. m, the lower-triangular triplets sorted by variable (for A(V) and the Schur pairs) and
 * a second time sorted by position (for A^T(coef): every entry of the result sums its contributions in a fixed order - the copies
 * of an SPMD run and the ranks of a sharded one must agree to the last bit, so no floating-point atomics).  The constant matrix
 * stays a dense n x n array (it is dense after fixings / in planted instances, and k_cert wants it dense).
 *
 * Schur complement of such a block - SDPA's F3 formula (Fujisawa, Kojima, Nakata 1997): with e = (p, q, a) in A_i, f = (r, s, b) in A_j
 *
 *      Mx[i][j] = tr(A_i X A_j Zinv) = sum_e sum_f a b ( X_qr Zinv_sp + [p != q] X_pr Zinv_sq + [r != s] X_qs Zinv_rp
 *                                                        + [p != q][r != s] X_ps Zinv_rq )
 *
 * 4 nnz_i nnz_j multiply-adds per pair instead of the 4 n^3 / m + n^2 of the dense formulation: the engine takes it when
 * 4 (sum nnz)^2 is the smaller number (hs_sp_prefers_sparse).  Row / column 0 of the extended matrix (constant matrix as
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
   int rc = sp_upload(&sp->voff, voff);
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

/* Mx[i][j] += tr(A_i X A_j Zinv) for 1 <= j <= i <= m (lower triangle, ld = m + 1; row / column 0 is not touched).  One wavefront
 * per pair: its lanes split the nnz_i x nnz_j products, summed in a fixed order. */
__global__ void __launch_bounds__(256) k_sp_schur(int m, int n, const int* __restrict__ voff, const int* __restrict__ vrow,
   const int* __restrict__ vcol, const double* __restrict__ vval, const double* __restrict__ X, const double* __restrict__ Zinv,
   double* __restrict__ Mx, long long npairs)
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
      const int a0 = voff[i], a1 = voff[i + 1], b0 = voff[j], b1 = voff[j + 1];
      const int na = a1 - a0, nb = b1 - b0;
      const long long tot = (long long) na * nb;
      double acc = 0.0;
      for (long long t = lane; t < tot; t += 64)
      {
         const int ea = a0 + (int) (t / nb), eb = b0 + (int) (t % nb);
         const int p = vrow[ea], q = vcol[ea], r = vrow[eb], c = vcol[eb];
         double w = X[(long long) q * n + r] * Zinv[(long long) c * n + p];
         if ( p != q ) w += X[(long long) p * n + r] * Zinv[(long long) c * n + q];
         if ( r != c ) w += X[(long long) q * n + c] * Zinv[(long long) r * n + p];
         if ( p != q && r != c ) w += X[(long long) p * n + c] * Zinv[(long long) r * n + q];
         acc += vval[ea] * vval[eb] * w;
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1)
         acc += __shfl_xor(acc, off, 64);
      if ( lane == 0 )
         Mx[(i + 1) * m1 + (j + 1)] += acc;
   }
}

/* the same sum by ONE THREAD per pair of variables, its element pairs in order: with a handful of nonzeros per matrix (3: nine
 * element pairs) a wavefront per pair has 9 of 64 lanes at work and pays the decoding of the pair index and a six-step reduction
 * for every one of the m (m + 1) / 2 pairs (0.94 ms at m = 2000); taken when the average number of element pairs is below 32 */
__global__ void __launch_bounds__(256) k_sp_schur_t(int m, int n, const int* __restrict__ voff, const int* __restrict__ vrow,
   const int* __restrict__ vcol, const double* __restrict__ vval, const double* __restrict__ X, const double* __restrict__ Zinv,
   double* __restrict__ Mx, long long npairs)
{
   const int m1 = m + 1;
   for (long long pr = (long long) blockIdx.x * blockDim.x + threadIdx.x; pr < npairs; pr += (long long) gridDim.x * blockDim.x)
   {
      long long i = (long long) ((sqrt(8.0 * (double) pr + 1.0) - 1.0) * 0.5);
      while ( (i + 1) * (i + 2) / 2 <= pr ) ++i;
      while ( i * (i + 1) / 2 > pr ) --i;
      const long long j = pr - i * (i + 1) / 2;
      const int a0 = voff[i], a1 = voff[i + 1], b0 = voff[j], b1 = voff[j + 1];
      double acc = 0.0;
      for (int ea = a0; ea < a1; ++ea)
      {
         const int p = vrow[ea], q = vcol[ea];
         const double va = vval[ea];
         for (int eb = b0; eb < b1; ++eb)
         {
            const int r = vrow[eb], c = vcol[eb];
            double w = X[(long long) q * n + r] * Zinv[(long long) c * n + p];
            if ( p != q ) w += X[(long long) p * n + r] * Zinv[(long long) c * n + q];
            if ( r != c ) w += X[(long long) q * n + c] * Zinv[(long long) r * n + p];
            if ( p != q && r != c ) w += X[(long long) p * n + c] * Zinv[(long long) r * n + q];
            acc += va * vval[eb] * w;
         }
      }
      Mx[(i + 1) * m1 + (j + 1)] += acc;
   }
}

int hs_sp_schur(hipStream_t s, const hs_sparse* sp, const double* X, const double* Zinv, double* Mx)
{
   if ( sp->m <= 0 )
      return HS_OK;
   const long long npairs = (long long) sp->m * (sp->m + 1) / 2;
   {
      const double avg = (double) sp->nnz / (double) sp->m;
      if ( avg * avg < 32.0 )
      {
         long long blocks = (npairs + 255) / 256;
         if ( blocks > 65536 ) blocks = 65536;
         hipLaunchKernelGGL(k_sp_schur_t, dim3((unsigned) blocks), dim3(256), 0, s, sp->m, sp->n, sp->voff, sp->vrow, sp->vcol, sp->vval, X, Zinv, Mx,
            npairs);
         HS_HIP( hipGetLastError() );
         return HS_OK;
      }
   }
   long long blocks = (npairs + 3) / 4;
   if ( blocks > 65536 ) blocks = 65536;
   hipLaunchKernelGGL(k_sp_schur, dim3((unsigned) blocks), dim3(256), 0, s, sp->m, sp->n, sp->voff, sp->vrow, sp->vcol, sp->vval, X, Zinv, Mx,
      npairs);
   HS_HIP( hipGetLastError() );
   return HS_OK;
}

/* dense expansion of the variables' matrices: A[v n^2 + ..] for v = 1 .. m (the caller has zeroed A); row 0 is not touched */
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
