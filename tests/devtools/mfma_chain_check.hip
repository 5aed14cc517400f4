// developer tool: is v_mfma_f64_16x16x4 with ONE accumulator chain (k ascending in steps of 4) bit-identical to the scalar chain
// acc = fma(a[k], b[k], acc), k ascending?  (decides whether the LDS products of the small-block kernels can move to the matrix cores
// without changing a bit).  build: hipcc --offload-arch=gfx950 -O3 tests/devtools/mfma_chain_check.hip -o /tmp/mfma_chain_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
#define N 48
__global__ void k_scalar(const double* a, const double* b, double* c)
{
   for (int e = threadIdx.x; e < N * N; e += blockDim.x)
   {
      const int r = e / N, cc = e % N;
      double acc = 0.0;
      for (int k = 0; k < N; ++k)
         acc = fma(a[r * N + k], b[k * N + cc], acc);
      c[e] = acc;
   }
}
__global__ void k_mfma(const double* a, const double* b, double* c)
{
   const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
   const int lr = lane & 15, kq = lane >> 4;
   for (int t = wave; t < 9; t += 4)
   {
      const int ti = t / 3, tj = t % 3;
      v4d acc = {0.0, 0.0, 0.0, 0.0};
      for (int kk = 0; kk < N; kk += 4)
      {
         const double av = a[(16 * ti + lr) * N + kk + kq];
         const double bv = b[(kk + kq) * N + 16 * tj + lr];
         acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
      }
      for (int r = 0; r < 4; ++r)
         c[(16 * ti + kq + 4 * r) * N + 16 * tj + lr] = acc[r];
   }
}
int main()
{
   std::vector<double> ha(N * N), hb(N * N), c1(N * N), c2(N * N);
   long long diff_total = 0;
   for (int trial = 0; trial < 20; ++trial)
   {
      srand(1234 + trial);
      for (int i = 0; i < N * N; ++i)
      {
         // mixed magnitudes and signs: cancellation makes the intermediate roundings visible
         ha[i] = ((double) rand() / RAND_MAX - 0.5) * pow(10.0, (rand() % 7) - 3);
         hb[i] = ((double) rand() / RAND_MAX - 0.5) * pow(10.0, (rand() % 7) - 3);
      }
      double *da, *db, *dc1, *dc2;
      (void) hipMalloc(&da, N * N * 8); (void) hipMalloc(&db, N * N * 8); (void) hipMalloc(&dc1, N * N * 8); (void) hipMalloc(&dc2, N * N * 8);
      (void) hipMemcpy(da, ha.data(), N * N * 8, hipMemcpyHostToDevice);
      (void) hipMemcpy(db, hb.data(), N * N * 8, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k_scalar, dim3(1), dim3(256), 0, 0, da, db, dc1);
      hipLaunchKernelGGL(k_mfma, dim3(1), dim3(256), 0, 0, da, db, dc2);
      (void) hipMemcpy(c1.data(), dc1, N * N * 8, hipMemcpyDeviceToHost);
      (void) hipMemcpy(c2.data(), dc2, N * N * 8, hipMemcpyDeviceToHost);
      int diff = 0; double maxrel = 0.0;
      for (int i = 0; i < N * N; ++i)
         if ( memcmp(&c1[i], &c2[i], 8) != 0 )
         {
            ++diff;
            const double rel = fabs(c1[i] - c2[i]) / fmax(fabs(c1[i]), 1e-300);
            if ( rel > maxrel ) maxrel = rel;
         }
      printf("trial %d: %d of %d entries differ, max relative difference %.3g\n", trial, diff, N * N, maxrel);
      diff_total += diff;
      (void) hipFree(da); (void) hipFree(db); (void) hipFree(dc1); (void) hipFree(dc2);
   }
   printf("total differing entries: %lld\n", diff_total);
   return 0;
}
