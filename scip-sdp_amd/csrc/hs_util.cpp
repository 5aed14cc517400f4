/* hs_util.cpp - error bookkeeping shared by the device engine */
#include "hs_common.h"
#include <stdio.h>
#include <string.h>

static thread_local char hs_errbuf[512] = "";

void hs_record_hip_error(hipError_t e, const char* what, const char* file, int line)
{
   snprintf(hs_errbuf, sizeof(hs_errbuf), "%s:%d: %s -> %s", file, line, what, hipGetErrorString(e));
}

const char* hs_last_error(void)
{
   return hs_errbuf;
}

/* ---- per-device kernel attributes --------------------------------------------------------------------------------------- */
#define HS_MAXDEV 128
static int g_attr_sets[HS_MAXDEV];
static int g_dev_cus[HS_MAXDEV];

int hs_func_max_lds(const void* fn, int bytes, hs_attr_mask* done)
{
   int dev = 0;
   HS_HIP( hipGetDevice(&dev) );
   if ( dev < 0 || dev >= HS_MAXDEV )
      return HS_ERR_ARG;
   const unsigned long long bit = 1ULL << (dev & 63);
   if ( __atomic_load_n(&done->bits[dev >> 6], __ATOMIC_ACQUIRE) & bit )
      return HS_OK;
   /* two threads on the same device may both get here: setting the attribute twice is harmless */
   HS_HIP( hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) );
   if ( !(__atomic_fetch_or(&done->bits[dev >> 6], bit, __ATOMIC_ACQ_REL) & bit) )
      (void) __atomic_add_fetch(&g_attr_sets[dev], 1, __ATOMIC_RELAXED);
   return HS_OK;
}

int hs_func_attr_sets(int device)
{
   return (device >= 0 && device < HS_MAXDEV) ? __atomic_load_n(&g_attr_sets[device], __ATOMIC_RELAXED) : -1;
}

int hs_device_cus(void)
{
   int dev = 0;
   if ( hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= HS_MAXDEV )
      return 0;
   int c = __atomic_load_n(&g_dev_cus[dev], __ATOMIC_RELAXED);
   if ( c <= 0 )
   {
      hipDeviceProp_t prop;
      if ( hipGetDeviceProperties(&prop, dev) != hipSuccess )
         return 0;
      c = prop.multiProcessorCount;
      __atomic_store_n(&g_dev_cus[dev], c, __ATOMIC_RELAXED);
   }
   return c;
}

/* ---- small-block device memory pool ------------------------------------------------------------------------------------
 * A branch-and-bound run re-shapes the engine at every node (other fixings -> other sizes), i.e. ~40 hipMalloc / hipFree
 * pairs per node solve, each tens of microseconds: as much as 15 % of a small node solve.  Blocks of at most 4 MiB are
 * therefore recycled: sizes are rounded up to a power of two, freed blocks are kept per (device, size class) up to 256 MiB
 * in total, and hs_pool_trim() (called when the last engine handle goes away) returns them to the runtime.  Larger
 * allocations go to hipMalloc / hipFree directly.  Callers free only after their stream work has completed (the engine
 * synchronises before it re-shapes or dies), so a recycled block is never still in use.  Contents are undefined, as with
 * hipMalloc. */
#include <mutex>
#include <unordered_map>
#include <vector>

namespace {
struct pool_block { size_t bytes; int device; };
std::mutex g_pool_mu;
std::unordered_map<void*, pool_block> g_pool_live;
std::unordered_map<unsigned long long, std::vector<void*> > g_pool_free;     /* key = device << 40 | size class */
size_t g_pool_cached = 0;
const size_t POOL_MAX_BLOCK = 4u << 20;
const size_t POOL_MAX_CACHED = 256u << 20;

size_t pool_class(size_t bytes)
{
   size_t c = 256;
   while ( c < bytes )
      c <<= 1;
   return c;
}
}

int hs_pool_alloc(void** p, size_t bytes)
{
   *p = NULL;
   if ( bytes == 0 )
      bytes = 8;
   int dev = 0;
   if ( hipGetDevice(&dev) != hipSuccess )
      return HS_ERR_HIP;
   if ( bytes <= POOL_MAX_BLOCK )
   {
      const size_t cls = pool_class(bytes);
      const unsigned long long key = ((unsigned long long) dev << 40) | (unsigned long long) cls;
      {
         std::lock_guard<std::mutex> lock(g_pool_mu);
         std::vector<void*>& v = g_pool_free[key];
         if ( !v.empty() )
         {
            *p = v.back();
            v.pop_back();
            g_pool_cached -= cls;
            g_pool_live[*p] = pool_block{cls, dev};
            return HS_OK;
         }
      }
      hipError_t e = hipMalloc(p, cls);
      if ( e != hipSuccess )
      {
         hs_record_hip_error(e, "hipMalloc (pool)", __FILE__, __LINE__);
         return e == hipErrorOutOfMemory ? HS_ERR_NOMEM : HS_ERR_HIP;
      }
      std::lock_guard<std::mutex> lock(g_pool_mu);
      g_pool_live[*p] = pool_block{cls, dev};
      return HS_OK;
   }
   hipError_t e = hipMalloc(p, bytes);
   if ( e != hipSuccess )
   {
      hs_record_hip_error(e, "hipMalloc", __FILE__, __LINE__);
      return e == hipErrorOutOfMemory ? HS_ERR_NOMEM : HS_ERR_HIP;
   }
   return HS_OK;
}

void hs_pool_free(void* p)
{
   if ( p == NULL )
      return;
   {
      std::lock_guard<std::mutex> lock(g_pool_mu);
      std::unordered_map<void*, pool_block>::iterator it = g_pool_live.find(p);
      if ( it != g_pool_live.end() )
      {
         const pool_block b = it->second;
         g_pool_live.erase(it);
         if ( g_pool_cached + b.bytes <= POOL_MAX_CACHED )
         {
            g_pool_free[((unsigned long long) b.device << 40) | (unsigned long long) b.bytes].push_back(p);
            g_pool_cached += b.bytes;
            return;
         }
      }
   }
   (void) hipFree(p);
}

void hs_pool_trim(void)
{
   std::vector<void*> all;
   {
      std::lock_guard<std::mutex> lock(g_pool_mu);
      for (auto& kv : g_pool_free)
      {
         for (void* p : kv.second)
            all.push_back(p);
         kv.second.clear();
      }
      g_pool_cached = 0;
   }
   for (void* p : all)
      (void) hipFree(p);
}
