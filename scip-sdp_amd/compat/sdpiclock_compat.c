/* sdpiclock_compat.c - standalone stand-in for the caller's running clock (API of src/sdpi/sdpiclock.h:50-78) and the
 * counting allocator used when no SCIP block memory exists.  Only compiled without HIPSDP_WITH_SCIP. */
#ifndef HIPSDP_WITH_SCIP
#define _POSIX_C_SOURCE 200809L
#include "hipsdp_scip_compat.h"
#include <stdlib.h>
#include <time.h>

struct SDPI_Clock
{
   SDPI_CLOCKTYPE type;
   int            running;
   double         accumulated;
   double         started;
};

static double now_of(SDPI_CLOCKTYPE type)
{
   struct timespec ts;
   clock_gettime(type == SDPI_CLOCKTYPE_CPU ? CLOCK_PROCESS_CPUTIME_ID : CLOCK_MONOTONIC, &ts);
   return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec;
}

SCIP_RETCODE SDPIclockCreate(SDPI_CLOCK** clck)
{
   *clck = (SDPI_CLOCK*) hipsdp_compat_malloc(sizeof(SDPI_CLOCK));
   if ( *clck == NULL )
      return SCIP_NOMEMORY;
   (*clck)->type = SDPI_CLOCKTYPE_WALL;
   (*clck)->running = 0;
   (*clck)->accumulated = 0.0;
   (*clck)->started = 0.0;
   return SCIP_OKAY;
}

void SDPIclockFree(SDPI_CLOCK** clck)
{
   if ( clck != NULL && *clck != NULL )
   {
      hipsdp_compat_free(*clck, sizeof(SDPI_CLOCK));
      *clck = NULL;
   }
}

void SDPIclockSetType(SDPI_CLOCK* clck, SDPI_CLOCKTYPE clocktype)
{
   clck->type = clocktype;
}

void SDPIclockStart(SDPI_CLOCK* clck)
{
   if ( clck->running++ == 0 )
      clck->started = now_of(clck->type);
}

void SDPIclockStop(SDPI_CLOCK* clck)
{
   if ( clck->running > 0 && --clck->running == 0 )
      clck->accumulated += now_of(clck->type) - clck->started;
}

SCIP_Real SDPIclockGetTime(SDPI_CLOCK* clck)
{
   if ( clck == NULL )
      return 0.0;
   if ( clck->running > 0 )
      return clck->accumulated + now_of(clck->type) - clck->started;
   return clck->accumulated;
}

static long long live_bytes = 0;

void* hipsdp_compat_malloc(size_t bytes)
{
   void* p = malloc(bytes > 0 ? bytes : 1);
   if ( p != NULL )
      __atomic_add_fetch(&live_bytes, (long long) bytes, __ATOMIC_RELAXED);
   return p;
}

void* hipsdp_compat_realloc(void* p, size_t oldbytes, size_t newbytes)
{
   void* q = realloc(p, newbytes > 0 ? newbytes : 1);
   if ( q != NULL )
      __atomic_add_fetch(&live_bytes, (long long) newbytes - (long long) (p != NULL ? oldbytes : 0), __ATOMIC_RELAXED);
   return q;
}

void hipsdp_compat_free(void* p, size_t bytes)
{
   if ( p != NULL )
   {
      free(p);
      __atomic_sub_fetch(&live_bytes, (long long) bytes, __ATOMIC_RELAXED);
   }
}

long long hipsdp_compat_mem_used(void)
{
   return __atomic_load_n(&live_bytes, __ATOMIC_RELAXED);
}
#endif
