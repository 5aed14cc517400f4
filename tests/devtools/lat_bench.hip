/* lat_bench.hip - development tool: dependent-issue latencies of one wavefront on gfx950 (FP64 FMA, v_rsq_f64 / v_rcp_f64,
 * v_readlane -> VALU, LDS write -> read), in nanoseconds and in shader cycles (s_memtime against the 100 MHz wall clock).
 *   hipcc --offload-arch=gfx950 -O3 tests/devtools/lat_bench.hip -o /tmp/lat_bench && /tmp/lat_bench */
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP 512

/* clock reads pinned in program order: the value under test is an operand of the (volatile) statement that reads the clock */
__device__ __forceinline__ long long tick(double& v)
{
   long long t;
   asm volatile("s_nop 0\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(v) :: "memory");
   return t;
}
__device__ __forceinline__ long long ctick(double& v)
{
   long long t;
   asm volatile("s_nop 0\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(v) :: "memory");
   return t;
}

__global__ void k_lat(double* out, long long* t, double x0)
{
   __shared__ double sh[64];
   double x = x0 + threadIdx.x * 1e-9;
   long long w0, w1, c0, c1;
   /* spin up */
   for (int i = 0; i < 20000; ++i) x = fma(x, 1.0000001, 1e-9);

   w0 = tick(x); c0 = ctick(x);
#pragma unroll
   for (int i = 0; i < REP; ++i) x = fma(x, 1.0000001, 1e-9);
   w1 = tick(x); c1 = ctick(x);
   if ( threadIdx.x == 0 ) { t[0] = w1 - w0; t[1] = c1 - c0; }

   w0 = tick(x);
#pragma unroll
   for (int i = 0; i < REP; ++i) x = __builtin_amdgcn_rsq(x + 2.0);
   w1 = tick(x);
   if ( threadIdx.x == 0 ) t[2] = w1 - w0;

   w0 = tick(x);
#pragma unroll
   for (int i = 0; i < REP; ++i) x = __builtin_amdgcn_rcp(x + 2.0);
   w1 = tick(x);
   if ( threadIdx.x == 0 ) t[3] = w1 - w0;

   w0 = tick(x);
#pragma unroll
   for (int i = 0; i < REP; ++i)
   {
      const int lo = __builtin_amdgcn_readlane(__double2loint(x), (i & 15));
      const int hi = __builtin_amdgcn_readlane(__double2hiint(x), (i & 15));
      x = fma(x, 0.5, __hiloint2double(hi, lo));
   }
   w1 = tick(x);
   if ( threadIdx.x == 0 ) t[4] = w1 - w0;

   w0 = tick(x);
#pragma unroll
   for (int i = 0; i < REP; ++i)
   {
      sh[threadIdx.x] = x;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      x = sh[(threadIdx.x + 1) & 63] * 0.5;
   }
   w1 = tick(x);
   if ( threadIdx.x == 0 ) t[5] = w1 - w0;

   /* two independent chains: issue rate of FP64 FMA */
   double y = x + 1.0, z = x + 2.0, u = x + 3.0;
   w0 = tick(x);
#pragma unroll
   for (int i = 0; i < REP; ++i)
   {
      x = fma(x, 1.0000001, 1e-9); y = fma(y, 1.0000001, 1e-9); z = fma(z, 1.0000001, 1e-9); u = fma(u, 1.0000001, 1e-9);
   }
   w1 = tick(x);
   if ( threadIdx.x == 0 ) t[6] = w1 - w0;

   /* dependent MFMA f64 16x16x4 */
   typedef double v4d __attribute__((ext_vector_type(4)));
   v4d acc = {x, y, z, u};
   double a0 = acc[0];
   w0 = tick(a0);
   acc[0] = a0;
#pragma unroll
   for (int i = 0; i < REP; ++i)
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(1e-3, 1e-3, acc, 0, 0, 0);
   a0 = acc[0];
   w1 = tick(a0);
   acc[0] = a0;
   if ( threadIdx.x == 0 ) t[7] = w1 - w0;
   out[threadIdx.x] = x + y + z + u + acc[0] + acc[1] + acc[2] + acc[3];
}

int main()
{
   double* out; long long* t;
   hipMalloc(&out, 64 * sizeof(double)); hipMalloc(&t, 16 * sizeof(long long));
   long long h[16];
   for (int rep = 0; rep < 3; ++rep)
   {
      hipLaunchKernelGGL(k_lat, dim3(1), dim3(64), 0, 0, out, t, 1.0);
      hipDeviceSynchronize();
   }
   hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
   const double ghz = (double) h[1] / ((double) h[0] * 10.0);       /* s_memtime ticks per ns */
   printf("s_memtime / wall clock: %.3f ticks per ns\n", ghz);
   const char* names[] = {"dependent v_fma_f64", "", "dependent v_rsq_f64 (+ add)", "dependent v_rcp_f64 (+ add)", "2 x v_readlane + fma",
      "LDS write -> read (+ mul)", "4 independent v_fma_f64 (per group of 4)", "dependent mfma_f64_16x16x4"};
   for (int i = 0; i < 8; ++i)
      if ( i != 1 )
         printf("%-44s %7.2f ns per step\n", names[i], (double) h[i] * 10.0 / REP);
   return 0;
}
