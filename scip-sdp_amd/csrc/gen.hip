/* gen.hip - synthetic dense node SDPs of BASELINE.md section 3 generated directly in HBM (2 GB of A at n = 500, m = 1000
 * would otherwise cross PCIe).  The stream is the stateless counter generator of oracle/instances.py: same seeds, same
 * index -> same uniform bits; the normals differ from numpy's only by the last ulps of log/cos/sqrt. */
#include "hs_kernels.h"
#include "../../include/hipsdp.h"
#include <math.h>

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x)
{
   x += 0x9E3779B97F4A7C15ULL;
   unsigned long long z = x;
   z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
   z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
   return z ^ (z >> 31);
}

__device__ __forceinline__ double counter_uniform(unsigned long long seed, unsigned long long idx)
{
   const unsigned long long h = splitmix64(seed * 0xD1342543DE82EF95ULL + idx);
   return ((double) (h >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}

__device__ __forceinline__ double counter_normal(unsigned long long seed, unsigned long long idx)
{
   const double u1 = counter_uniform(seed, 2ULL * idx);
   const double u2 = counter_uniform(seed, 2ULL * idx + 1ULL);
   return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925286766559 * u2);
}

/* A_i = (G_i + G_i^T) / sqrt(2 n), G_i lower triangular with N(0,1) entries drawn at packed index r (r + 1) / 2 + c */
/* density < 1 (rho = 0.1 variant): an off-diagonal entry is kept with that probability (its own counter stream),
 * the diagonal always; kept entries are scaled by 1 / sqrt(rho) so that the matrices keep their norm */
__global__ void k_gen_dense(int n, int i0, int cnt, unsigned long long seed, double scale, double density, double* __restrict__ A)
{
   const long long n2 = (long long) n * n;
   const long long total = (long long) cnt * n2;
   for (long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long) gridDim.x * blockDim.x)
   {
      const int i = (int) (t / n2) + i0;
      const long long e = t - (long long) (i - i0) * n2;
      const int r0 = (int) (e / n), c0 = (int) (e - (long long) r0 * n);
      const int r = r0 > c0 ? r0 : c0;
      const int c = r0 > c0 ? c0 : r0;
      const unsigned long long idx = (unsigned long long) r * (unsigned long long) (r + 1) / 2ULL + (unsigned long long) c;
      double g = counter_normal(seed + (unsigned long long) i, idx);
      if ( density < 1.0 && r != c && counter_uniform(seed + 7777777ULL + (unsigned long long) i, idx) >= density )
         g = 0.0;
      A[(long long) i * n2 + e] = (r == c ? 2.0 : 1.0) * g * scale;
   }
}

/* the matrices A_i, i0 <= i < i1 (i >= 1), written to A + i n^2 */
static thread_local double g_gen_density = 1.0;

/* density of the matrices the next hs_gen_dense calls of this thread generate (1: dense) */
void hs_gen_set_density(double density)
{
   g_gen_density = (density > 0.0 && density < 1.0) ? density : 1.0;
}

int hs_gen_dense(hipStream_t s, int n, int i0, int i1, long long seed, double* A)
{
   if ( i1 <= i0 )
      return HS_OK;
   const long long total = (long long) (i1 - i0) * n * n;
   long long g = (total + 255) / 256;
   if ( g > 65536 ) g = 65536;
   hipLaunchKernelGGL(k_gen_dense, dim3((unsigned) g), dim3(256), 0, s, n, i0, i1 - i0, (unsigned long long) seed,
      1.0 / sqrt(2.0 * (double) n * g_gen_density), g_gen_density, A);
   if ( hipGetLastError() != hipSuccess )
      return HS_ERR_HIP;
   return HS_OK;
}
