/* solve1.hip - host side of the one-launch node solve: what fits (hs_solve1_fits), the workspace it may need outside LDS, and the choice
 * of the kernel instance.  The kernel itself is csrc/solve1_body.h, compiled once per size class (solve1_c10.hip: every block at most
 * 10 rows, solve1_c16.hip: at most 16, solve1_c64.hip: anything with m <= 64, solve1_c64m.hip: 64 < m <= 128): each instance holds
 * the code of its class only. */
#define S1_HOST_PART
#include "solve1_body.h"

int hs_solve1_launch_c10(hipStream_t st, const hs_solve1_args* a);
int hs_solve1_launch_c16(hipStream_t st, const hs_solve1_args* a);
int hs_solve1_launch_c64(hipStream_t st, const hs_solve1_args* a);
int hs_solve1_launch_c64m(hipStream_t st, const hs_solve1_args* a);

int hs_solve1_launch_c10_dbg(unsigned int* out4);
int hs_solve1_launch_c16_dbg(unsigned int* out4);
int hs_solve1_launch_c64_dbg(unsigned int* out4);
int hs_solve1_launch_c64m_dbg(unsigned int* out4);

/* debug build (-DS1_DEBUG): out2[0] = values that were declared wave-uniform and were not, out2[1] = solves run, summed over the
 * instances; returns 1 in a debug build, 0 in a release build */
int hs_solve1_debug_counts(unsigned int* out2)
{
   unsigned int t[4];
   int dbg = 0;
   out2[0] = out2[1] = 0;
   int (*fn[4])(unsigned int*) = {hs_solve1_launch_c10_dbg, hs_solve1_launch_c16_dbg, hs_solve1_launch_c64_dbg, hs_solve1_launch_c64m_dbg};
   for (int i = 0; i < 4; ++i)
   {
      const int r = fn[i](t);
      if ( r > 0 ) dbg = 1;
      out2[0] += t[0]; out2[1] += t[1];
   }
   return dbg;
}

/* 10 / 16 / 64: the class by block size, + 1000 when m > 64 */
int hs_solve1_class(const hs_solve1_args* a)
{
   int nmax = 0;
   for (int k = 0; k < a->nblk; ++k)
      if ( a->n[k] > nmax ) nmax = a->n[k];
   /* HIPSDP_SOLVE1_CLASS=64: everything through the general instance (developer switch: the instances must agree bit for bit) */
   static int force = -1;
   if ( force < 0 )
   {
      const char* env = getenv("HIPSDP_SOLVE1_CLASS");
      force = env != NULL ? atoi(env) : 0;
   }
   if ( a->m > 64 )
      return 1064;
   if ( force == 64 )
      return 64;
   if ( force == 16 && nmax <= 16 )
      return 16;
   return nmax <= 10 ? 10 : (nmax <= 16 ? 16 : 64);
}

int hs_solve1_launch(hipStream_t st, const hs_solve1_args* a)
{
   switch ( hs_solve1_class(a) )
   {
   case 10: return hs_solve1_launch_c10(st, a);
   case 16: return hs_solve1_launch_c16(st, a);
   case 64: return hs_solve1_launch_c64(st, a);
   default: return hs_solve1_launch_c64m(st, a);
   }
}
