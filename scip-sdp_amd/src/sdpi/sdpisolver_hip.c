/* sdpisolver_hip.c - SCIP-SDP solver interface backed by the MI355X interior-point engine (libhipsdp, include/hipsdp.h).
 *
 * Drop-in for the reference's src/sdpi/sdpisolver_{dsdp.c,sdpa.cpp,mosek.c}: it exports the same 53 SCIPsdpiSolver*
 * symbols (src/sdpi/sdpisolver.h:79-724), so sdpi.c / relax_sdp.c call it unchanged.  Host code is C99; all arithmetic of
 * the node solve runs on the GPU behind the hipsdp_* C ABI.  Behavioural reference for the marshalling and the status
 * machine: src/sdpi/sdpisolver_dsdp.c (fixed-variable elimination :929-960, block compaction :1066-1130, LP row split
 * :1206-1449, tolerance re-solve loop :1527-1606, penalty post-processing :1655-1734, predicates :1751-2135) and, for
 * the primal-matrix export, src/sdpi/sdpisolver_sdpa.cpp:2814-3125.
 *
 * Formulation handed to the engine (sdpisolver.h:37-42, penalty form :235-250):
 *    min  b^T y (+ Gamma r)   s.t.  sum_j A_j^k y_j - A_0^k (+ r I) psd,   rows of  D y (+ r) >= d  for every finite LP
 *    side, one row per finite variable bound, (r >= 0).
 */
#ifndef _POSIX_C_SOURCE
#define _POSIX_C_SOURCE 200809L
#endif
#include <assert.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#ifdef HIPSDP_WITH_SCIP
#include "sdpi/sdpisolver.h"
#include "blockmemshell/memory.h"
#include "scip/pub_message.h"
#define HSALLOC(sol, ptr, n)          BMSallocBlockMemoryArray((sol)->blkmem, (ptr), (n))
#define HSFREE(sol, ptr, n)           BMSfreeBlockMemoryArrayNull((sol)->blkmem, (ptr), (n))
#else
#include "sdpisolver_hip.h"
#define HSALLOC(sol, ptr, n)          (*(void**) (ptr) = hipsdp_compat_malloc(sizeof(**(ptr)) * (size_t) ((n) > 0 ? (n) : 1)))
#define HSFREE(sol, ptr, n)           do { if ( *(ptr) != NULL ) { hipsdp_compat_free(*(ptr), sizeof(**(ptr)) * (size_t) ((n) > 0 ? (n) : 1)); *(ptr) = NULL; } } while (0)
#define SCIPerrorMessage(...)         do { fprintf(stderr, "[%s:%d] ERROR: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); } while (0)
#define SCIPmessagePrintInfo(h, ...)  do { (void) (h); printf(__VA_ARGS__); } while (0)
#endif

#include "hipsdp.h"

/* how often HIPSDP_VERIFY_SHORTCUT=1 has checked the termination bound against an independent evaluation of Z(y) (tests) */
static long long g_shortcut_checks = 0;
#ifndef HIPSDP_WITH_SCIP
long long hipsdp_compat_shortcut_checks(void) { return g_shortcut_checks; }
#endif


#define PENALTYBOUNDTOL          1e-3      /* relative distance of Tr(X) to Gamma that counts as "bound reached" */
#define MIN_PENALTYPARAM         1e5
#define MAX_PENALTYPARAM         1e12
#define PENALTYPARAM_FACTOR      1e4
#define MAX_MAXPENALTYPARAM      1e15
#define MAXPENALTYPARAM_FACTOR   1e6
#define TOLCHANGE                0.1       /* tightening factor of the re-solve loop */
#define MINSOLVERTOL             1e-10     /* the engine is not asked for more than this */
#define HS_INFINITY              1e20

#define ALLOC_OR_FAIL(sol, ptr, n)  do { if ( HSALLOC(sol, ptr, n) == NULL ) return SCIP_NOMEMORY; } while (0)
#define ENGINE_CALL(x)  do { int rc_ = (x); if ( rc_ != HIPSDP_OK ) { SCIPerrorMessage("hipsdp error %d (%s) in %s\n", rc_, hipsdp_last_error(), #x); return rc_ == HIPSDP_ERR_NOMEM ? SCIP_NOMEMORY : SCIP_LPERROR; } } while (0)
#define CHECK_IF_SOLVED(sol)  do { if ( ! (sol)->solved ) { SCIPerrorMessage("Tried to access solution information for SDP %d ahead of solving!\n", (sol)->sdpcounter); return SCIP_LPERROR; } } while (0)
#define CHECK_IF_SOLVED_BOOL(sol)  do { if ( ! (sol)->solved ) return FALSE; } while (0)

struct SCIP_SDPiSolver
{
   SCIP_MESSAGEHDLR*     messagehdlr;
   BMS_BLKMEM*           blkmem;
   BMS_BUFMEM*           bufmem;
   hipsdp_solver*        engine;             /* created lazily: Create() must work without a GPU (plumbing tests) */
   int                   device;

   /* variable maps (sdpisolver_dsdp.c:168-223 keeps the same ones) */
   int                   nvars;
   int                   nactivevars;
   int                   nalloc;             /* allocated length of the per-variable arrays */
   int*                  inputtoactive;      /* nvars: 1-based active index, or -(k) for the k-th fixed variable */
   int*                  activetoinput;      /* nactivevars */
   SCIP_Real*            fixedvarsval;       /* per input variable: its value if fixed */
   SCIP_Real*            objcoefs;           /* nactivevars */
   SCIP_Real             fixedvarsobjcontr;
   int*                  lbrow;              /* per input variable: engine LP row of its lower bound or -1 */
   int*                  ubrow;

   /* LP rows */
   int                   nlpcons;
   int                   nlpalloc;
   int*                  lhsrow;             /* per input LP row: engine row of the lhs side or -1 */
   int*                  rhsrow;
   int                   nlpineqs;           /* engine rows that come from LP sides (the first ones) */
   int                   nenginerows;        /* total engine LP rows q */

   /* SDP blocks */
   int                   nsdpblocks;
   int                   nblkalloc;
   int*                  blockmap;           /* input block -> engine block or -1 */
   int*                  compactsize;        /* per input block: size after removing indices */
   int*                  origsize;
   int**                 keptind;            /* per input block: compact index -> original index */
   int                   nengineblocks;

   /* solution of the last solve, on the host */
   SCIP_Real*            ysol;               /* engine variables (nactivevars (+1 for r)) */
   int                   nysol;
   SCIP_Real*            xlp;                /* engine LP multipliers */
   int                   nxlp;
   SCIP_Real**           Xsol;               /* per engine block, dense compact n x n; fetched on demand */
   int*                  Xsize;
   int                   nXsol;

   hipsdp_info           info;
   SCIP_Bool             solved;
   SCIP_Bool             timelimit;
   SCIP_Bool             timelimitinitial;
   SCIP_Bool             penalty;
   SCIP_Bool             feasorig;
   SCIP_Bool             rbound;
   SCIP_Bool             penaltyworbound;
   int                   rvar;               /* engine index (0-based) of the penalty variable r or -1 */
   int                   sdpcounter;
   int                   niterations;
   int                   nsdpcalls;
   SCIP_Real             opttime;
   SCIP_SDPSOLVERSETTING usedsetting;

   SCIP_Real             epsilon;
   SCIP_Real             gaptol;
   SCIP_Real             feastol;
   SCIP_Real             sdpsolverfeastol;
   SCIP_Real             penaltyparam;
   SCIP_Real             objlimit;
   SCIP_Real             preoptimalgap;
   SCIP_Bool             sdpinfo;
   int                   nthreads;           /* reinterpreted: number of GPUs (-1 = all visible) */

   /* fingerprint of the SDP arrays whose device-resident master copy the engine holds */
   SCIP_Bool             mastervalid;
   unsigned long long    masterhash;
   int                   masternvars;
   int                   masternblocks;
   int                   masternnz;
   /* the problem whose master copy did NOT fit beside the engine's storage (HIPSDP_ERR_NOMEM): while the caller's arrays keep this
    * fingerprint the multi-GB attempt is not repeated at every node */
   SCIP_Bool             masternofit;
   unsigned long long    nofithash;
   int                   nofitnvars;
   int                   nofitnblocks;
   int                   nofitnnz;
};

/* ---------------------------------------------------------------------------------------------------------------------- */
/* local helpers                                                                                                          */
/* ---------------------------------------------------------------------------------------------------------------------- */

/* wall clock of the marshalling stages, printed with HIPSDP_STAGE_TIMES=1 (tests/devtools/sdpi_overhead.py, bench.py) */
static double wallNow(void)
{
   struct timespec ts;
   clock_gettime(CLOCK_MONOTONIC, &ts);
   return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec;
}

static SCIP_Bool isFixed(const SCIP_SDPISOLVER* s, SCIP_Real lb, SCIP_Real ub)
{
   return (ub - lb <= s->epsilon);   /* sdpisolver_dsdp.c:253-262 */
}

static SCIP_Bool isInf(SCIP_Real v)
{
   return (v <= -HS_INFINITY || v >= HS_INFINITY);
}

static void freeSolution(SCIP_SDPISOLVER* s)
{
   int b;
   HSFREE(s, &s->ysol, s->nysol);
   s->nysol = 0;
   HSFREE(s, &s->xlp, s->nxlp);
   s->nxlp = 0;
   if ( s->Xsol != NULL )
   {
      for (b = 0; b < s->nXsol; ++b)
         HSFREE(s, &s->Xsol[b], s->Xsize[b] * s->Xsize[b]);
      HSFREE(s, &s->Xsol, s->nXsol);
      HSFREE(s, &s->Xsize, s->nXsol);
   }
   s->nXsol = 0;
}

static void freeBlockMaps(SCIP_SDPISOLVER* s)
{
   int b;
   if ( s->keptind != NULL )
   {
      for (b = 0; b < s->nblkalloc; ++b)
         HSFREE(s, &s->keptind[b], s->origsize[b]);
      HSFREE(s, &s->keptind, s->nblkalloc);
   }
   HSFREE(s, &s->blockmap, s->nblkalloc);
   HSFREE(s, &s->compactsize, s->nblkalloc);
   HSFREE(s, &s->origsize, s->nblkalloc);
   s->nblkalloc = 0;
}

static void freeVarMaps(SCIP_SDPISOLVER* s)
{
   HSFREE(s, &s->inputtoactive, s->nalloc);
   HSFREE(s, &s->activetoinput, s->nalloc);
   HSFREE(s, &s->fixedvarsval, s->nalloc);
   HSFREE(s, &s->objcoefs, s->nalloc);
   HSFREE(s, &s->lbrow, s->nalloc);
   HSFREE(s, &s->ubrow, s->nalloc);
   s->nalloc = 0;
}

static void freeLpMaps(SCIP_SDPISOLVER* s)
{
   HSFREE(s, &s->lhsrow, s->nlpalloc);
   HSFREE(s, &s->rhsrow, s->nlpalloc);
   s->nlpalloc = 0;
}

/* Fingerprint of the structure of the SDP arrays, the addresses of the value arrays and the entries themselves: ALL entries up
 * to 4 million nonzeros, ~65536 evenly spaced samples beyond that (a full pass over 10^8 triplets would cost as much as the
 * solve).  One multiply-xorshift round per 64-bit word in two independent chains (values / index pairs), about 1 ns per entry
 * (2.3 million nonzeros: 4 ms; the byte-wise FNV-1a used before took 35 ms, three times the solve of such a node).
 * SCIPsdpiLoadSDP re-allocates and re-fills these arrays (sdpi.c:2329-2520), so address and content change together; a caller
 * that edits single values of a huge instance in place must set HIPSDP_NOCACHE=1. */
static unsigned long long hashMix(unsigned long long h, unsigned long long v)
{
   h = (h ^ v) * 0x9E3779B97F4A7C15ULL;
   return h ^ (h >> 29);
}

static unsigned long long sdpFingerprint(int nvars, int nsdpblocks, const int* sdpblocksizes, const int* sdpnblockvars, int sdpnnonz,
   int* const* sdpnblockvarnonz, int* const* sdpvar, int** const* sdprow, int** const* sdpcol, SCIP_Real** const* sdpval)
{
   unsigned long long h = 1469598103934665603ULL;
   unsigned long long hv = 0x243F6A8885A308D3ULL;       /* chain of the values */
   unsigned long long hi = 0x13198A2E03707344ULL;       /* chain of the (row, column) pairs */
   long long seen = 0;
   const long long stride = sdpnnonz > 4000000 ? sdpnnonz / 65536 : 1;
   int b;
   int k;
   int t;
   h = hashMix(h, (unsigned long long) nvars);
   h = hashMix(h, (unsigned long long) nsdpblocks);
   h = hashMix(h, (unsigned long long) sdpnnonz);
   for (b = 0; b < nsdpblocks; ++b)
   {
      h = hashMix(h, (unsigned long long) sdpblocksizes[b]);
      h = hashMix(h, (unsigned long long) sdpnblockvars[b]);
      for (k = 0; k < sdpnblockvars[b]; ++k)
      {
         const int nn = sdpnblockvarnonz[b][k];
         const SCIP_Real* vals = sdpval[b][k];
         const int* rows = sdprow[b][k];
         const int* cols = sdpcol[b][k];
         h = hashMix(h, (unsigned long long) sdpvar[b][k]);
         h = hashMix(h, (unsigned long long) nn);
         h = hashMix(h, (unsigned long long) (size_t) vals);
         /* first sample position >= seen that is a multiple of stride */
         t = (int) ((stride - (seen % stride)) % stride);
         for (; t < nn; t += (int) stride)
         {
            unsigned long long bits;
            memcpy(&bits, &vals[t], sizeof(bits));
            hv = hashMix(hv, bits);
            hi = hashMix(hi, ((unsigned long long) rows[t] << 32) | (unsigned long long) (unsigned int) cols[t]);
         }
         seen += nn;
      }
   }
   return hashMix(hashMix(h, hv), hi);
}

/* fetch X of an engine block on first use */
static SCIP_RETCODE ensureX(SCIP_SDPISOLVER* s, int eb)
{
   if ( s->Xsol[eb] == NULL )
   {
      ALLOC_OR_FAIL(s, &s->Xsol[eb], s->Xsize[eb] * s->Xsize[eb]);
      ENGINE_CALL( hipsdp_get_X(s->engine, eb, s->Xsol[eb]) );
   }
   return SCIP_OKAY;
}

/* ---------------------------------------------------------------------------------------------------------------------- */
/* miscellaneous                                                                                                          */
/* ---------------------------------------------------------------------------------------------------------------------- */

const char* SCIPsdpiSolverGetSolverName(void)
{
   return "HIPSDP";
}

const char* SCIPsdpiSolverGetSolverDesc(void)
{
   return "Homogeneous self-dual primal-dual interior-point solver for dense SDP blocks on AMD MI355X (gfx950), FP64 MFMA Schur assembly";
}

void* SCIPsdpiSolverGetSolverPointer(SCIP_SDPISOLVER* sdpisolver)
{
   assert( sdpisolver != NULL );
   return (void*) sdpisolver->engine;
}

int SCIPsdpiSolverGetDefaultSdpiSolverNpenaltyIncreases(void)
{
   return 8;
}

SCIP_Bool SCIPsdpiSolverDoesWarmstartNeedPrimal(void)
{
   return TRUE;   /* primal-dual method; also required by relax_sdp.c:3842-3966 */
}

/* ---------------------------------------------------------------------------------------------------------------------- */
/* creation and destruction                                                                                               */
/* ---------------------------------------------------------------------------------------------------------------------- */

SCIP_RETCODE SCIPsdpiSolverCreate(SCIP_SDPISOLVER** sdpisolver, SCIP_MESSAGEHDLR* messagehdlr, BMS_BLKMEM* blkmem,
   BMS_BUFMEM* bufmem)
{
   SCIP_SDPISOLVER* s;
   assert( sdpisolver != NULL );
#ifdef HIPSDP_WITH_SCIP
   if ( BMSallocBlockMemory(blkmem, sdpisolver) == NULL )
      return SCIP_NOMEMORY;
#else
   *sdpisolver = (SCIP_SDPISOLVER*) hipsdp_compat_malloc(sizeof(SCIP_SDPISOLVER));
   if ( *sdpisolver == NULL )
      return SCIP_NOMEMORY;
#endif
   s = *sdpisolver;
   memset(s, 0, sizeof(*s));
   s->messagehdlr = messagehdlr;
   s->blkmem = blkmem;
   s->bufmem = bufmem;
   s->engine = NULL;
   s->device = 0;
   if ( getenv("HIPSDP_DEVICE") != NULL )
      s->device = atoi(getenv("HIPSDP_DEVICE"));
   else if ( getenv("LOCAL_RANK") != NULL && (getenv("HIPSDP_WORLD") != NULL || getenv("WORLD_SIZE") != NULL) )
      s->device = atoi(getenv("LOCAL_RANK"));        /* SPMD launch: one process per GPU */
   s->solved = FALSE;
   s->timelimit = FALSE;
   s->timelimitinitial = FALSE;
   s->rvar = -1;
   s->sdpcounter = 0;
   s->usedsetting = SCIP_SDPSOLVERSETTING_UNSOLVED;
   /* defaults of the reference backends (sdpisolver_dsdp.c:558-568) */
   s->epsilon = 1e-9;
   s->gaptol = 1e-6;
   s->feastol = 1e-6;
   s->sdpsolverfeastol = 1e-6;
   s->penaltyparam = 1e5;
   s->objlimit = HS_INFINITY;
   s->sdpinfo = FALSE;
   s->nthreads = -1;
   s->preoptimalgap = -1.0;
   s->info.status = HIPSDP_STATUS_UNSOLVED;
   return SCIP_OKAY;
}

SCIP_RETCODE SCIPsdpiSolverFree(SCIP_SDPISOLVER** sdpisolver)
{
   SCIP_SDPISOLVER* s;
   assert( sdpisolver != NULL );
   s = *sdpisolver;
   if ( s == NULL )
      return SCIP_OKAY;
   if ( s->engine != NULL )
      hipsdp_free(&s->engine);
   freeSolution(s);
   freeBlockMaps(s);
   freeVarMaps(s);
   freeLpMaps(s);
#ifdef HIPSDP_WITH_SCIP
   BMSfreeBlockMemory(s->blkmem, sdpisolver);
#else
   hipsdp_compat_free(s, sizeof(SCIP_SDPISOLVER));
   *sdpisolver = NULL;
#endif
   return SCIP_OKAY;
}

SCIP_RETCODE SCIPsdpiSolverIncreaseCounter(SCIP_SDPISOLVER* sdpisolver)
{
   assert( sdpisolver != NULL );
   sdpisolver->sdpcounter++;
   return SCIP_OKAY;
}

SCIP_RETCODE SCIPsdpiSolverResetCounter(SCIP_SDPISOLVER* sdpisolver)
{
   assert( sdpisolver != NULL );
   sdpisolver->sdpcounter = 0;
   return SCIP_OKAY;
}

/* ---------------------------------------------------------------------------------------------------------------------- */
/* solving                                                                                                                */
/* ---------------------------------------------------------------------------------------------------------------------- */

SCIP_RETCODE SCIPsdpiSolverLoadAndSolve(
   SCIP_SDPISOLVER* sdpisolver, int nvars, const SCIP_Real* obj, const SCIP_Real* lb, const SCIP_Real* ub,
   int nsdpblocks, const int* sdpblocksizes, const int* sdpnblockvars,
   int sdpconstnnonz, const int* sdpconstnblocknonz, int* const* sdpconstrow, int* const* sdpconstcol, SCIP_Real* const* sdpconstval,
   int sdpnnonz, int* const* sdpnblockvarnonz, int* const* sdpvar, int** const* sdprow, int** const* sdpcol, SCIP_Real** const* sdpval,
   int* const* indchanges, const int* nremovedinds, const int* blockindchanges, int nremovedblocks,
   int nlpcons, const int* lpindchanges, const SCIP_Real* lplhs, const SCIP_Real* lprhs,
   int lpnnonz, const int* lpbeg, const int* lpind, const SCIP_Real* lpval,
   const SCIP_Real* starty, const int* startZnblocknonz, int* const* startZrow, int* const* startZcol, SCIP_Real* const* startZval,
   const int* startXnblocknonz, int* const* startXrow, int* const* startXcol, SCIP_Real* const* startXval,
   SCIP_SDPSOLVERSETTING startsettings, SCIP_Real timelimit, SDPI_CLOCK* usedsdpitime)
{
   /* sdpisolver_dsdp.c:671-735: the plain solve is the penalty solve with Gamma = 0 */
   return SCIPsdpiSolverLoadAndSolveWithPenalty(sdpisolver, 0.0, TRUE, TRUE, nvars, obj, lb, ub, nsdpblocks, sdpblocksizes,
      sdpnblockvars, sdpconstnnonz, sdpconstnblocknonz, sdpconstrow, sdpconstcol, sdpconstval, sdpnnonz, sdpnblockvarnonz, sdpvar,
      sdprow, sdpcol, sdpval, indchanges, nremovedinds, blockindchanges, nremovedblocks, nlpcons, lpindchanges, lplhs, lprhs,
      lpnnonz, lpbeg, lpind, lpval, starty, startZnblocknonz, startZrow, startZcol, startZval, startXnblocknonz, startXrow,
      startXcol, startXval, startsettings, timelimit, usedsdpitime, NULL, NULL);
}

/* one engine solve with the given tolerances; accumulates counters */
/* Warm start (sdpisolver.h:160-173; consumed like sdpisolver_sdpa.cpp:1481-1592): y in original variable indices; Z and X
 * as sparse lower triangles in ORIGINAL block indices, block nsdpblocks = diagonal LP block with index 2 * row (+ 1 for the
 * rhs side) for LP rows and 2 * nlpcons + 2 * var (+ 1 for the upper bound) for variable bounds.  Entries of removed rows /
 * columns / sides are dropped.  The engine uses the point only if it is strictly interior (hipsdp_set_start). */
static SCIP_RETCODE loadStartPoint(SCIP_SDPISOLVER* s, int nvars, int nsdpblocks, int* const* indchanges, int nlpcons,
   const SCIP_Real* starty, const int* startZnblocknonz, int* const* startZrow, int* const* startZcol, SCIP_Real* const* startZval,
   const int* startXnblocknonz, int* const* startXrow, int* const* startXcol, SCIP_Real* const* startXval, int nengvars, int q)
{
   SCIP_Real* y0 = NULL;
   SCIP_Real* x0 = NULL;
   SCIP_Real* z0 = NULL;
   SCIP_Real** X0 = NULL;
   SCIP_Real** Z0 = NULL;
   SCIP_RETCODE retcode = SCIP_OKAY;
   int b;
   int i;
   int which;
   const int neb = s->nengineblocks;

   y0 = (SCIP_Real*) calloc((size_t) (nengvars > 0 ? nengvars : 1), sizeof(SCIP_Real));
   x0 = (SCIP_Real*) calloc((size_t) (q > 0 ? q : 1), sizeof(SCIP_Real));
   z0 = (SCIP_Real*) calloc((size_t) (q > 0 ? q : 1), sizeof(SCIP_Real));
   X0 = (SCIP_Real**) calloc((size_t) (neb > 0 ? neb : 1), sizeof(SCIP_Real*));
   Z0 = (SCIP_Real**) calloc((size_t) (neb > 0 ? neb : 1), sizeof(SCIP_Real*));
   if ( y0 == NULL || x0 == NULL || z0 == NULL || X0 == NULL || Z0 == NULL )
      retcode = SCIP_NOMEMORY;
   for (b = 0; b < nsdpblocks && retcode == SCIP_OKAY; ++b)
   {
      const int eb = s->blockmap[b];
      if ( eb < 0 )
         continue;
      X0[eb] = (SCIP_Real*) calloc((size_t) s->compactsize[b] * (size_t) s->compactsize[b], sizeof(SCIP_Real));
      Z0[eb] = (SCIP_Real*) calloc((size_t) s->compactsize[b] * (size_t) s->compactsize[b], sizeof(SCIP_Real));
      if ( X0[eb] == NULL || Z0[eb] == NULL )
         retcode = SCIP_NOMEMORY;
   }
   if ( retcode == SCIP_OKAY )
   {
      for (i = 0; i < s->nactivevars; ++i)
         y0[i] = starty[s->activetoinput[i]];
      for (which = 0; which < 2; ++which)
      {
         const int* nnz = which ? startXnblocknonz : startZnblocknonz;
         int* const* rows = which ? startXrow : startZrow;
         int* const* cols = which ? startXcol : startZcol;
         SCIP_Real* const* vals = which ? startXval : startZval;
         SCIP_Real* lpvec = which ? x0 : z0;
         for (b = 0; b < nsdpblocks; ++b)
         {
            const int eb = s->blockmap[b];
            SCIP_Real* D;
            int n;
            if ( eb < 0 )
               continue;
            D = which ? X0[eb] : Z0[eb];
            n = s->compactsize[b];
            for (i = 0; i < nnz[b]; ++i)
            {
               const int r = rows[b][i];
               const int c = cols[b][i];
               if ( r < 0 || c < 0 || r >= s->origsize[b] || c >= s->origsize[b] )
                  continue;
               if ( indchanges[b][r] < 0 || indchanges[b][c] < 0 )
                  continue;                     /* the row / column may have been fixed to zero in the meantime */
               D[(size_t) (r - indchanges[b][r]) * n + (c - indchanges[b][c])] = vals[b][i];
               D[(size_t) (c - indchanges[b][c]) * n + (r - indchanges[b][r])] = vals[b][i];
            }
         }
         /* diagonal LP block */
         for (i = 0; i < nnz[nsdpblocks]; ++i)
         {
            const int idx = rows[nsdpblocks][i];
            int erow = -1;
            if ( idx < 0 || idx >= 2 * nlpcons + 2 * nvars || idx != cols[nsdpblocks][i] )
               continue;
            if ( idx < 2 * nlpcons )
               erow = (idx % 2 == 0) ? s->lhsrow[idx / 2] : s->rhsrow[idx / 2];
            else
            {
               const int v = (idx - 2 * nlpcons) / 2;
               erow = ((idx - 2 * nlpcons) % 2 == 0) ? s->lbrow[v] : s->ubrow[v];
            }
            if ( erow >= 0 && erow < q )
               lpvec[erow] = vals[nsdpblocks][i];
         }
      }
      if ( hipsdp_set_start(s->engine, y0, (const double* const*) X0, (const double* const*) Z0, x0, z0) != HIPSDP_OK )
      {
         SCIPerrorMessage("hipsdp_set_start failed: %s\n", hipsdp_last_error());
         retcode = SCIP_LPERROR;
      }
   }
   if ( X0 != NULL && Z0 != NULL )
   {
      for (b = 0; b < neb; ++b)
      {
         free(X0[b]);
         free(Z0[b]);
      }
   }
   free(X0); free(Z0); free(y0); free(x0); free(z0);
   return retcode;
}

static SCIP_RETCODE engineSolve(SCIP_SDPISOLVER* s, int level, SCIP_Real gaptol, SCIP_Real feastol, SCIP_Real remaining, SDPI_CLOCK* clck)
{
   hipsdp_params par;
   SCIP_Real t0;
   hipsdp_default_params(&par);
   par.settings = level;                 /* 0 fast, 1 medium, 2 stable: the engine's side of the retry ladder */
   par.gaptol = gaptol;
   par.feastol = feastol;
   /* the caller validates the X-side ABSOLUTELY with SCIP_SDPPAR_FEASTOL (sdpsolchecker.c:775-931) while the engine's own
    * measure is relative to 1 + ||b||: when the outer tolerance is the looser one, ask for it explicitly */
   par.pabstol = (s->feastol > feastol) ? s->feastol : 0.0;
   par.preoptgap = (! s->penalty && s->preoptimalgap > 0.0) ? s->preoptimalgap : 0.0;
   par.objlimit = (s->penalty ? HS_INFINITY : s->objlimit);
   par.timelimit = remaining < HS_INFINITY ? remaining : 0.0;
   par.verbose = s->sdpinfo ? 1 : 0;
   t0 = SDPIclockGetTime(clck);
   ENGINE_CALL( hipsdp_solve(s->engine, &par, &s->info) );
   s->opttime += (clck != NULL) ? SDPIclockGetTime(clck) - t0 : s->info.solve_seconds;
   if ( clck == NULL || s->opttime <= 0.0 )
      s->opttime = s->info.solve_seconds > s->opttime ? s->info.solve_seconds : s->opttime;
   s->niterations += s->info.iterations;
   s->nsdpcalls++;
   return SCIP_OKAY;
}

/* Remaining time of the caller's clock, or HS_INFINITY.  *expired is the decision "no time left"; with several SPMD ranks it is
 * rank 0's decision on every rank (each process has its own clock: a rank that left early would leave the others alone in the
 * next RCCL collective, which has no timeout). */
static SCIP_RETCODE timeLeft(SCIP_SDPISOLVER* s, SCIP_Real timelimit, SDPI_CLOCK* usedsdpitime, SCIP_Real* remaining, SCIP_Bool* expired)
{
   int flag = 0;
   *remaining = HS_INFINITY;
   *expired = FALSE;
   if ( timelimit >= HS_INFINITY || usedsdpitime == NULL )
      return SCIP_OKAY;
   *remaining = timelimit - SDPIclockGetTime(usedsdpitime);
   flag = (*remaining <= 0.0) ? 1 : 0;
   if ( s->engine != NULL )
      ENGINE_CALL( hipsdp_sync_flag(s->engine, &flag) );
   *expired = (flag != 0);
   if ( ! *expired && *remaining <= 0.0 )
      *remaining = 1e-3;                 /* rank 0 still has time: go on with the others, the engine's own check is collective */
   return SCIP_OKAY;
}

/* creates the engine on first use and joins the communicator of an SPMD launch */
static SCIP_RETCODE ensureEngine(SCIP_SDPISOLVER* s)
{
   int rc;
   if ( s->engine != NULL )
      return SCIP_OKAY;
   rc = hipsdp_create(&s->engine, s->device);
   if ( rc != HIPSDP_OK )
   {
      SCIPerrorMessage("Cannot create the HIP engine: %s\n", hipsdp_last_error());
      s->engine = NULL;
      return SCIP_LPERROR;
   }
   /* SPMD launch (N copies of the host program, one per GPU, WORLD_SIZE / RANK in the environment): every node SDP is
    * sharded over the ranks of the process-wide communicator; all copies make the same calls and see the same results.
    * SCIP_SDPPAR_NTHREADS (reinterpreted as the number of GPUs) = 1 keeps this solver on its own GPU. */
   if ( s->nthreads != 1 )
   {
      void* comm = NULL;
      int crank = 0, cworld = 1;
      if ( hipsdp_comm_from_env(s->device, &comm, &crank, &cworld) != HIPSDP_OK
         || (comm != NULL && hipsdp_set_comm(s->engine, comm, crank, cworld) != HIPSDP_OK) )
      {
         SCIPerrorMessage("Cannot join the communicator of the SPMD launch: %s\n", hipsdp_last_error());
         hipsdp_free(&s->engine);
         return SCIP_LPERROR;
      }
   }
   return SCIP_OKAY;
}

static SCIP_Bool spmdLaunch(const SCIP_SDPISOLVER* s)
{
   const char* w = getenv("HIPSDP_WORLD") != NULL ? getenv("HIPSDP_WORLD") : getenv("WORLD_SIZE");
   return s->nthreads != 1 && w != NULL && atoi(w) > 1;
}

/* COO buffers of one upload */
typedef struct { int* var; int* row; int* col; SCIP_Real* val; } CooBuf;

static SCIP_Bool cooAlloc(CooBuf* c, long long cnt)
{
   const size_t k = (size_t) (cnt > 0 ? cnt : 1);
   c->var = (int*) malloc(k * sizeof(int));
   c->row = (int*) malloc(k * sizeof(int));
   c->col = (int*) malloc(k * sizeof(int));
   c->val = (SCIP_Real*) malloc(k * sizeof(SCIP_Real));
   return c->var != NULL && c->row != NULL && c->col != NULL && c->val != NULL;
}

static void cooFree(CooBuf* c)
{
   free(c->var); free(c->row); free(c->col); free(c->val);
   c->var = NULL; c->row = NULL; c->col = NULL; c->val = NULL;
}

/* fills the engine's compact blocks: non-constant part through the device-resident master copy (or directly when that does not
 * fit), then the constant matrix of the node and the identity of the penalty variable */
static SCIP_RETCODE loadBlocks(SCIP_SDPISOLVER* s, int nvars, int nsdpblocks, const int* sdpblocksizes, const int* sdpnblockvars,
   const int* sdpconstnblocknonz, int* const* sdpconstrow, int* const* sdpconstcol, SCIP_Real* const* sdpconstval, int sdpnnonz,
   int* const* sdpnblockvarnonz, int* const* sdpvar, int** const* sdprow, int** const* sdpcol, SCIP_Real** const* sdpval,
   int* const* indchanges)
{
   clock_t tfp0, tfp1, tup1;                 /* processor time of this thread's marshalling work (printed with SDPINFO) */
   unsigned long long fp;
   SCIP_Bool usemaster = (getenv("HIPSDP_NOMASTER") == NULL);
   const SCIP_Bool usecache = (getenv("HIPSDP_NOCACHE") == NULL);
   CooBuf coo = {NULL, NULL, NULL, NULL};
   int* slots = NULL;
   SCIP_RETCODE retcode = SCIP_OKAY;
   int b;
   int k;
   int t;

   /* blocks the engine keeps as nonzeros are loaded directly (a few triplets per matrix: nothing a dense master copy would save) */
   for (b = 0; b < nsdpblocks; ++b)
      if ( s->blockmap[b] >= 0 && hipsdp_block_is_sparse(s->engine, s->blockmap[b]) )
         usemaster = FALSE;
   tfp0 = clock();
   fp = sdpFingerprint(nvars, nsdpblocks, sdpblocksizes, sdpnblockvars, sdpnnonz, sdpnblockvarnonz, sdpvar, sdprow, sdpcol, sdpval);
   tfp1 = clock();
   if ( usemaster && s->masternofit && s->nofithash == fp && s->nofitnvars == nvars && s->nofitnblocks == nsdpblocks && s->nofitnnz == sdpnnonz )
      usemaster = FALSE;
   if ( usemaster && (! usecache || ! s->mastervalid || s->masterhash != fp || s->masternvars != nvars || s->masternblocks != nsdpblocks
         || s->masternnz != sdpnnonz) )
   {
      int rc;
      s->mastervalid = FALSE;
      rc = hipsdp_master_define(s->engine, nvars, nsdpblocks, sdpblocksizes, sdpnblockvars);
      /* slot = position of the variable in the block's list; the engine streams the caller's per-variable arrays through pinned
       * staging chunks (no concatenated host copy: 1.25e8 triplets at n = 500, m = 1000) */
      for (b = 0; b < nsdpblocks && rc == HIPSDP_OK; ++b)
         rc = hipsdp_master_add_vars(s->engine, b, sdpnblockvars[b], sdpnblockvarnonz[b], (const int* const*) sdprow[b],
            (const int* const*) sdpcol[b], (const double* const*) sdpval[b]);
      if ( rc == HIPSDP_ERR_NOMEM )
      {
         /* the master copy does not fit beside the engine's storage: drop it and load this node's blocks directly */
         (void) hipsdp_master_define(s->engine, 0, 0, NULL, NULL);
         usemaster = FALSE;
         s->masternofit = TRUE;
         s->nofithash = fp;
         s->nofitnvars = nvars;
         s->nofitnblocks = nsdpblocks;
         s->nofitnnz = sdpnnonz;
         if ( s->sdpinfo )
            printf("hipsdp: no room for the device-resident master copy, loading the node's blocks directly\n");
      }
      else if ( rc != HIPSDP_OK )
      {
         SCIPerrorMessage("uploading the master copy failed (%d): %s\n", rc, hipsdp_last_error());
         return SCIP_LPERROR;
      }
      else
      {
         tup1 = clock();
         if ( s->sdpinfo )
            printf("hipsdp: fingerprint of %d nonzeros %.2f ms, upload of the master copy %.2f ms\n", sdpnnonz,
               1e3 * (double) (tfp1 - tfp0) / (double) CLOCKS_PER_SEC, 1e3 * (double) (tup1 - tfp1) / (double) CLOCKS_PER_SEC);
         s->mastervalid = TRUE;
         s->masterhash = fp;
         s->masternvars = nvars;
         s->masternblocks = nsdpblocks;
         s->masternnz = sdpnnonz;
      }
   }
   if ( ! usemaster )
      s->mastervalid = FALSE;

   if ( s->nactivevars > 0 )
   {
      slots = (int*) malloc((size_t) s->nactivevars * sizeof(int));
      if ( slots == NULL )
         return SCIP_NOMEMORY;
   }
   for (b = 0; b < nsdpblocks && retcode == SCIP_OKAY; ++b)
   {
      long long cnt = 0;
      long long pos = 0;
      int rc;
      const int eb = s->blockmap[b];
      if ( eb < 0 )
         continue;
      if ( usemaster )
      {
         /* slot of every active variable in this block (-1: it does not appear) */
         for (k = 0; k < s->nactivevars; ++k)
            slots[k] = -1;
         for (k = 0; k < sdpnblockvars[b]; ++k)
         {
            const int av = s->inputtoactive[sdpvar[b][k]];
            if ( av > 0 )
               slots[av - 1] = k;
         }
         rc = hipsdp_master_gather(s->engine, eb, b, s->nactivevars, slots, s->compactsize[b], s->keptind[b]);
         if ( rc != HIPSDP_OK )
         {
            SCIPerrorMessage("hipsdp_master_gather failed (%d): %s\n", rc, hipsdp_last_error());
            retcode = SCIP_LPERROR;
            break;
         }
      }
      else
      {
         for (k = 0; k < sdpnblockvars[b]; ++k)
            if ( s->inputtoactive[sdpvar[b][k]] > 0 )
               cnt += sdpnblockvarnonz[b][k];
      }
      /* per node: constant matrix (changes with the fixings) and the identity of the penalty variable */
      cnt += sdpconstnblocknonz[b] + (s->penalty ? s->compactsize[b] : 0);
      if ( cnt == 0 )
         continue;
      if ( ! cooAlloc(&coo, cnt) )
      {
         cooFree(&coo);
         retcode = SCIP_NOMEMORY;
         break;
      }
      if ( ! usemaster )
      {
         for (k = 0; k < sdpnblockvars[b]; ++k)
         {
            const int av = s->inputtoactive[sdpvar[b][k]];
            if ( av <= 0 )
               continue;
            for (t = 0; t < sdpnblockvarnonz[b][k]; ++t)
            {
               const int r = sdprow[b][k][t];
               const int c = sdpcol[b][k][t];
               if ( indchanges[b][r] < 0 || indchanges[b][c] < 0 )
                  continue;
               coo.var[pos] = av;
               coo.row[pos] = r - indchanges[b][r];
               coo.col[pos] = c - indchanges[b][c];
               coo.val[pos] = sdpval[b][k][t];
               ++pos;
            }
         }
      }
      for (t = 0; t < sdpconstnblocknonz[b]; ++t)
      {
         const int r = sdpconstrow[b][t];
         const int c = sdpconstcol[b][t];
         if ( indchanges[b][r] < 0 || indchanges[b][c] < 0 )
            continue;      /* cannot happen for consistent input (sdpi.c:691-809); be safe */
         coo.var[pos] = 0;
         coo.row[pos] = r - indchanges[b][r];
         coo.col[pos] = c - indchanges[b][c];
         coo.val[pos] = sdpconstval[b][t];
         ++pos;
      }
      if ( s->penalty )
      {
         for (t = 0; t < s->compactsize[b]; ++t)
         {
            coo.var[pos] = s->rvar + 1;
            coo.row[pos] = t;
            coo.col[pos] = t;
            coo.val[pos] = 1.0;
            ++pos;
         }
      }
      rc = hipsdp_add_entries(s->engine, eb, pos, coo.var, coo.row, coo.col, coo.val);
      cooFree(&coo);
      if ( rc != HIPSDP_OK )
      {
         SCIPerrorMessage("hipsdp_add_entries failed (%d): %s\n", rc, hipsdp_last_error());
         retcode = (rc == HIPSDP_ERR_NOMEM) ? SCIP_NOMEMORY : SCIP_LPERROR;
      }
   }
   free(slots);
   return retcode;
}

/* pull y and the LP multipliers of the last engine solve to the host */
static SCIP_RETCODE fetchVectors(SCIP_SDPISOLVER* s)
{
   if ( s->nysol > 0 )
      ENGINE_CALL( hipsdp_get_y(s->engine, s->ysol) );
   if ( s->nxlp > 0 )
      ENGINE_CALL( hipsdp_get_lp(s->engine, s->xlp, NULL) );
   return SCIP_OKAY;
}

static SCIP_Bool statusKnown(int st);

/* one rung of the settings ladder: an engine solve, then the tolerance re-solve loop of sdpisolver_dsdp.c:1527-1606 (SDPA:
 * checkFeastolAndResolve, sdpisolver_sdpa.cpp:369-494): while the engine says optimal, y is checked against OUR feasibility
 * tolerance and the gap against OUR gap tolerance; a violation tightens the engine's tolerance by TOLCHANGE and solves again */
static SCIP_RETCODE solveAndCheckTolerances(SCIP_SDPISOLVER* s, int level, SCIP_Real timelimit, SDPI_CLOCK* usedsdpitime,
   SCIP_Real* remaining)
{
   SCIP_Real solverfeastol = s->sdpsolverfeastol;
   SCIP_Real solvergaptol = s->gaptol;
   SCIP_RETCODE retcode;

   retcode = engineSolve(s, level, solvergaptol, solverfeastol, *remaining, usedsdpitime);
   if ( retcode != SCIP_OKAY )
      return retcode;

   while ( s->info.status == HIPSDP_STATUS_OPTIMAL && ! s->penalty )
   {
      SCIP_Real lmin[64];
      SCIP_Real* lminp = lmin;
      SCIP_Real lpviol = 0.0;
      SCIP_Bool infeasible = FALSE;
      SCIP_Bool solveagain = FALSE;
      SCIP_Bool expired = FALSE;
      int e;

      retcode = fetchVectors(s);
      if ( retcode != SCIP_OKAY )
         return retcode;
      if ( s->nengineblocks > 64 )
      {
         lminp = (SCIP_Real*) malloc((size_t) s->nengineblocks * sizeof(SCIP_Real));
         if ( lminp == NULL )
            return SCIP_NOMEMORY;
      }
      /* feasibility of y w.r.t. OUR tolerance: bounds and LP rows are engine rows, blocks through lambda_min
       * (what SCIPsdpSolcheckerCheck, sdpsolchecker.c:58-265, verifies on the host in the reference) */
      /* The engine's own termination test is a PROOF when it is tight enough: at its final iterate Z is positive definite (its
       * Cholesky factorization succeeded) and Z(y) = (Z + Rd) / tau, so lambda_min(Z(y)) >= -||Rd||_F / tau, and an LP row of y is
       * violated by at most |rd| / tau (z > 0): both are bounded by info.dabs, which the engine drove below its tolerance.  Only
       * when that bound does not already settle the caller's tolerance the eigenvalues are computed (a device round trip per node
       * that B&B-sized solves - 1.5 ms each in one launch - would feel). */
      if ( s->info.status == HIPSDP_STATUS_OPTIMAL && s->info.dabs >= 0.0 && s->info.dabs <= 0.999 * s->feastol )
      {
         for (e = 0; e < s->nengineblocks; ++e)
            lminp[e] = -s->info.dabs;
         lpviol = s->info.dabs;
         /* HIPSDP_VERIFY_SHORTCUT=1 (tests, CI): the bound is checked against an independent evaluation of Z(y) - the
          * check the shortcut replaces - whenever it is used; a violation is an error of the backend, not a property of the node */
         if ( getenv("HIPSDP_VERIFY_SHORTCUT") != NULL && getenv("HIPSDP_VERIFY_SHORTCUT")[0] == '1' )
         {
            SCIP_Real* lchk = (SCIP_Real*) malloc((size_t) (s->nengineblocks > 0 ? s->nengineblocks : 1) * sizeof(SCIP_Real));
            SCIP_Real lpchk = 0.0;
            /* the claim: lambda_min(Z(y)) >= -dabs and no LP row violated by more than dabs.  hipsdp_check_y_tol with the tolerance
             * (dabs + slack) / 0.999 proves exactly that for blocks above 64 rows (Cholesky factorization of Z(y) + (dabs + slack) I; it
             * then reports -0.999 tol) and returns the exact eigenvalue below that size or when the factorization fails; slack covers
             * the rounding of evaluating Z(y) a second time */
            const SCIP_Real slack = 1e-9 + 0.01 * s->info.dabs;
            const SCIP_Real chktol = (s->info.dabs + slack) / 0.999;
            int bad = 0;
            if ( lchk == NULL )
            {
               if ( lminp != lmin ) free(lminp);
               return SCIP_NOMEMORY;
            }
            if ( hipsdp_check_y_tol(s->engine, s->ysol, chktol, lchk, &lpchk) != HIPSDP_OK )
               bad = 1;
            for (e = 0; e < s->nengineblocks && ! bad; ++e)
               if ( lchk[e] < -chktol )
                  bad = 1;
            if ( ! bad && lpchk > s->info.dabs + slack )
               bad = 1;
            ++g_shortcut_checks;
            free(lchk);
            if ( bad )
            {
               if ( lminp != lmin ) free(lminp);
               SCIPerrorMessage("HIPSDP: the termination bound dabs = %g does not hold for the independent check of y\n", s->info.dabs);
               return SCIP_LPERROR;
            }
         }
      }
      else
      {
         int rc = hipsdp_check_y_tol(s->engine, s->ysol, s->feastol, lminp, &lpviol);
         if ( rc != HIPSDP_OK )
         {
            if ( lminp != lmin ) free(lminp);
            SCIPerrorMessage("hipsdp_check_y failed: %s\n", hipsdp_last_error());
            return SCIP_LPERROR;
         }
      }
      for (e = 0; e < s->nengineblocks; ++e)
         if ( lminp[e] < -s->feastol )
            infeasible = TRUE;
      if ( lpviol > s->feastol )
         infeasible = TRUE;
      if ( lminp != lmin ) free(lminp);

      if ( infeasible )
      {
         solverfeastol *= TOLCHANGE;
         if ( solverfeastol >= MINSOLVERTOL )
            solveagain = TRUE;
      }
      if ( REALABS(s->info.pobj - s->info.dobj) >= s->gaptol )
      {
         infeasible = TRUE;
         solvergaptol *= TOLCHANGE;
         if ( solvergaptol >= MINSOLVERTOL )
            solveagain = TRUE;
      }
      if ( ! solveagain )
      {
         if ( infeasible )
         {
            s->info.status = HIPSDP_STATUS_NUMERIC;
            SCIPmessagePrintInfo(s->messagehdlr, "HIPSDP failed to reach required feasibility tolerance (feastol: %g, gaptol: %g)!\n",
               solverfeastol, solvergaptol);
         }
         break;
      }
      retcode = timeLeft(s, timelimit, usedsdpitime, remaining, &expired);
      if ( retcode != SCIP_OKAY )
         return retcode;
      if ( expired )
      {
         s->info.status = HIPSDP_STATUS_TIMELIM;
         break;
      }
      retcode = engineSolve(s, level, solvergaptol, solverfeastol, *remaining, usedsdpitime);
      if ( retcode != SCIP_OKAY )
         return retcode;
   }
   return SCIP_OKAY;
}

SCIP_RETCODE SCIPsdpiSolverLoadAndSolveWithPenalty(
   SCIP_SDPISOLVER* sdpisolver, SCIP_Real penaltyparam, SCIP_Bool withobj, SCIP_Bool rbound,
   int nvars, const SCIP_Real* obj, const SCIP_Real* lb, const SCIP_Real* ub,
   int nsdpblocks, const int* sdpblocksizes, const int* sdpnblockvars,
   int sdpconstnnonz, const int* sdpconstnblocknonz, int* const* sdpconstrow, int* const* sdpconstcol, SCIP_Real* const* sdpconstval,
   int sdpnnonz, int* const* sdpnblockvarnonz, int* const* sdpvar, int** const* sdprow, int** const* sdpcol, SCIP_Real** const* sdpval,
   int* const* indchanges, const int* nremovedinds, const int* blockindchanges, int nremovedblocks,
   int nlpcons, const int* lpindchanges, const SCIP_Real* lplhs, const SCIP_Real* lprhs,
   int lpnnonz, const int* lpbeg, const int* lpind, const SCIP_Real* lpval,
   const SCIP_Real* starty, const int* startZnblocknonz, int* const* startZrow, int* const* startZcol, SCIP_Real* const* startZval,
   const int* startXnblocknonz, int* const* startXrow, int* const* startXcol, SCIP_Real* const* startXval,
   SCIP_SDPSOLVERSETTING startsettings, SCIP_Real timelimit, SDPI_CLOCK* usedsdpitime,
   SCIP_Bool* feasorig, SCIP_Bool* penaltybound)
{
   SCIP_SDPISOLVER* s = sdpisolver;
   SCIP_Real remaining;
   SCIP_Real* bvec = NULL;
   SCIP_Real* dext = NULL;
   int* engsizes = NULL;
   int nfixed;
   int nengvars;
   int m1;
   int q;
   int i;
   int j;
   int b;
   int row;
   const SCIP_Bool stagetimes = (getenv("HIPSDP_STAGE_TIMES") != NULL);
   double tstage[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};

   assert( s != NULL );
   assert( penaltyparam > -1 * s->epsilon );
   tstage[0] = wallNow();
   assert( penaltyparam < s->epsilon || feasorig != NULL );
   assert( nvars > 0 );
   assert( obj != NULL && lb != NULL && ub != NULL );
   (void) sdpconstnnonz; (void) sdpnnonz; (void) nremovedblocks; (void) nremovedinds;
   /* the optional start point (sdpisolver.h:160-173) is handed to the engine after the problem is loaded, see below */

   if ( startsettings != SCIP_SDPSOLVERSETTING_UNSOLVED && startsettings != SCIP_SDPSOLVERSETTING_PENALTY
      && startsettings != SCIP_SDPSOLVERSETTING_FAST && startsettings != SCIP_SDPSOLVERSETTING_MEDIUM
      && startsettings != SCIP_SDPSOLVERSETTING_STABLE )
   {
      SCIPerrorMessage("Unknown setting %d for start-settings!\n", (int) startsettings);   /* sdpisolver_sdpa.cpp:1445-1449 */
      return SCIP_LPERROR;
   }

   s->niterations = 0;
   s->nsdpcalls = 0;
   s->opttime = 0.0;
   s->feasorig = FALSE;
   s->penalty = penaltyparam > s->epsilon;
   s->rbound = rbound;
   s->penaltyworbound = (s->penalty && ! rbound);
   s->info.status = HIPSDP_STATUS_UNSOLVED;

   /* time limit check before doing anything (sdpisolver_dsdp.c:879-892); under an SPMD launch the engine (and with it the
    * communicator) is needed first, because the decision has to be the same on every rank */
   {
      SCIP_Bool expired = FALSE;
      SCIP_RETCODE trc;
      if ( s->engine == NULL && spmdLaunch(s) && timelimit < HS_INFINITY && usedsdpitime != NULL )
      {
         trc = ensureEngine(s);
         if ( trc != SCIP_OKAY )
            return trc;
      }
      trc = timeLeft(s, timelimit, usedsdpitime, &remaining, &expired);
      if ( trc != SCIP_OKAY )
         return trc;
      if ( expired )
      {
         s->timelimit = TRUE;
         s->timelimitinitial = TRUE;
         s->solved = FALSE;
         return SCIP_OKAY;
      }
   }
   s->timelimit = FALSE;
   s->timelimitinitial = FALSE;
   s->solved = FALSE;

   if ( ! s->penalty )
   {
      s->sdpcounter++;
      s->usedsetting = SCIP_SDPSOLVERSETTING_FAST;
   }
   else
      s->usedsetting = SCIP_SDPSOLVERSETTING_PENALTY;

   /* ---- variable maps ------------------------------------------------------------------------------------------- */
   if ( nvars > s->nalloc )
   {
      freeVarMaps(s);
      ALLOC_OR_FAIL(s, &s->inputtoactive, nvars);
      ALLOC_OR_FAIL(s, &s->activetoinput, nvars);
      ALLOC_OR_FAIL(s, &s->fixedvarsval, nvars);
      ALLOC_OR_FAIL(s, &s->objcoefs, nvars);
      ALLOC_OR_FAIL(s, &s->lbrow, nvars);
      ALLOC_OR_FAIL(s, &s->ubrow, nvars);
      s->nalloc = nvars;
   }
   s->nvars = nvars;
   s->nactivevars = 0;
   nfixed = 0;
   s->fixedvarsobjcontr = 0.0;
   for (i = 0; i < nvars; ++i)
   {
      s->lbrow[i] = -1;
      s->ubrow[i] = -1;
      s->fixedvarsval[i] = 0.0;
      if ( isFixed(s, lb[i], ub[i]) )
      {
         ++nfixed;
         s->inputtoactive[i] = -nfixed;
         s->fixedvarsval[i] = lb[i];
         s->fixedvarsobjcontr += obj[i] * lb[i];
      }
      else
      {
         s->activetoinput[s->nactivevars] = i;
         s->objcoefs[s->nactivevars] = obj[i];
         s->nactivevars++;
         s->inputtoactive[i] = s->nactivevars;
      }
   }
   if ( ! withobj )
      s->fixedvarsobjcontr = 0.0;
   nengvars = s->nactivevars + (s->penalty ? 1 : 0);
   s->rvar = s->penalty ? s->nactivevars : -1;
   m1 = nengvars + 1;

   /* ---- block maps ---------------------------------------------------------------------------------------------- */
   freeBlockMaps(s);
   if ( nsdpblocks > 0 )
   {
      ALLOC_OR_FAIL(s, &s->blockmap, nsdpblocks);
      ALLOC_OR_FAIL(s, &s->compactsize, nsdpblocks);
      ALLOC_OR_FAIL(s, &s->origsize, nsdpblocks);
      ALLOC_OR_FAIL(s, &s->keptind, nsdpblocks);
      s->nblkalloc = nsdpblocks;
      for (b = 0; b < nsdpblocks; ++b)
      {
         s->keptind[b] = NULL;
         s->origsize[b] = sdpblocksizes[b];
      }
   }
   s->nsdpblocks = nsdpblocks;
   s->nengineblocks = 0;
   if ( nsdpblocks > 0 )
      ALLOC_OR_FAIL(s, &engsizes, nsdpblocks);
   for (b = 0; b < nsdpblocks; ++b)
   {
      int cnt = 0;
      ALLOC_OR_FAIL(s, &s->keptind[b], sdpblocksizes[b]);
      if ( blockindchanges[b] < 0 )
      {
         s->blockmap[b] = -1;
         s->compactsize[b] = 0;
         continue;
      }
      for (i = 0; i < sdpblocksizes[b]; ++i)
      {
         if ( indchanges[b][i] >= 0 )
         {
            assert( i - indchanges[b][i] == cnt );
            s->keptind[b][cnt++] = i;
         }
      }
      s->compactsize[b] = cnt;
      if ( cnt == 0 )
      {
         s->blockmap[b] = -1;
         continue;
      }
      s->blockmap[b] = s->nengineblocks;
      engsizes[s->nengineblocks++] = cnt;
   }

   /* ---- LP rows: one engine row per finite side (sdpisolver_dsdp.c:1217-1314), then one per finite bound --------- */
   if ( nlpcons > s->nlpalloc )
   {
      freeLpMaps(s);
      ALLOC_OR_FAIL(s, &s->lhsrow, nlpcons);
      ALLOC_OR_FAIL(s, &s->rhsrow, nlpcons);
      s->nlpalloc = nlpcons;
   }
   s->nlpcons = nlpcons;
   q = 0;
   for (i = 0; i < nlpcons; ++i)
   {
      s->lhsrow[i] = -1;
      s->rhsrow[i] = -1;
      if ( lpindchanges[i] < 0 )
         continue;
      if ( lplhs[i] > -HS_INFINITY )
         s->lhsrow[i] = q++;
      if ( lprhs[i] < HS_INFINITY )
         s->rhsrow[i] = q++;
   }
   s->nlpineqs = q;
   for (j = 0; j < s->nactivevars; ++j)
   {
      const int v = s->activetoinput[j];
      if ( ! isInf(lb[v]) )
         s->lbrow[v] = q++;
      if ( ! isInf(ub[v]) )
         s->ubrow[v] = q++;
   }
   if ( s->penalty && rbound )
      ++q;                                   /* r >= 0 is the last row */
   s->nenginerows = q;

   /* ---- engine ---------------------------------------------------------------------------------------------------- */
   tstage[1] = wallNow();
   {
      SCIP_RETCODE erc = ensureEngine(s);
      int rc;
      if ( erc != SCIP_OKAY )
      {
         HSFREE(s, &engsizes, nsdpblocks);
         return erc;
      }
      {
         /* nonzeros of the variables' matrices per engine block (what the reference backends hand their solver entry by entry:
          * sdpisolver_dsdp.c:1126-1195, sdpisolver_sdpa.cpp:1223-1267): the engine keeps a block as triplets when that makes the
          * Schur assembly cheaper than the dense formulation - no (m + 1) x n^2 array for matrices of a few nonzeros each */
         long long* engnnz = (long long*) calloc((size_t) (nsdpblocks > 0 ? nsdpblocks : 1), sizeof(long long));
         if ( engnnz != NULL )
         {
            int bb;
            int kk;
            for (bb = 0; bb < nsdpblocks; ++bb)
            {
               const int eb = s->blockmap[bb];
               if ( eb < 0 )
                  continue;
               for (kk = 0; kk < sdpnblockvars[bb]; ++kk)
                  if ( s->inputtoactive[sdpvar[bb][kk]] > 0 )
                     engnnz[eb] += sdpnblockvarnonz[bb][kk];
               if ( s->penalty )
                  engnnz[eb] += s->compactsize[bb];
            }
         }
         rc = hipsdp_set_shape2(s->engine, nengvars, s->nengineblocks, engsizes, q, engnnz);
         free(engnnz);
      }
      HSFREE(s, &engsizes, nsdpblocks);
      if ( rc != HIPSDP_OK )
      {
         SCIPerrorMessage("hipsdp_set_shape failed (%d): %s\n", rc, hipsdp_last_error());
         return rc == HIPSDP_ERR_NOMEM ? SCIP_NOMEMORY : SCIP_LPERROR;
      }
   }

   /* objective */
   ALLOC_OR_FAIL(s, &bvec, nengvars);
   for (j = 0; j < s->nactivevars; ++j)
      bvec[j] = withobj ? s->objcoefs[j] : 0.0;
   if ( s->penalty )
      bvec[s->rvar] = penaltyparam;
   {
      const int rc = hipsdp_set_obj(s->engine, bvec);
      HSFREE(s, &bvec, nengvars);
      if ( rc != HIPSDP_OK )
      {
         SCIPerrorMessage("hipsdp_set_obj failed (%d): %s\n", rc, hipsdp_last_error());
         return rc == HIPSDP_ERR_NOMEM ? SCIP_NOMEMORY : SCIP_LPERROR;
      }
   }

   tstage[2] = wallNow();
   /* SDP blocks.  The matrices A_v are uploaded ONCE in original indices (master copy in HBM, re-used while the caller's
    * arrays are unchanged: relax_sdp.c:4455-4495 reloads them only when the number of variables or blocks changes); every
    * node then only sends its list of active variables and kept indices and the engine gathers the compact block on the
    * device.  Fixed variables are skipped here because the caller has moved them into the constant part
    * (sdpisolver.h:160-163, sdpi.c:614-682).  A block's master storage has one slot per variable that appears in it; when
    * even that does not fit the device the node's compact blocks are loaded directly (active variables, kept indices). */
   {
      SCIP_RETCODE lrc = loadBlocks(s, nvars, nsdpblocks, sdpblocksizes, sdpnblockvars, sdpconstnblocknonz, sdpconstrow, sdpconstcol,
         sdpconstval, sdpnnonz, sdpnblockvarnonz, sdpvar, sdprow, sdpcol, sdpval, indchanges);
      if ( lrc != SCIP_OKAY )
         return lrc;
   }

   tstage[3] = wallNow();
   /* LP part, dense rows [c | D] */
   if ( q > 0 )
   {
      dext = (SCIP_Real*) calloc((size_t) q * (size_t) m1, sizeof(SCIP_Real));
      if ( dext == NULL )
         return SCIP_NOMEMORY;
      for (i = 0; i < nlpcons; ++i)
      {
         int nextbeg;
         if ( lpindchanges[i] < 0 )
            continue;
         nextbeg = (i == nlpcons - 1) ? lpnnonz : lpbeg[i + 1];
         for (j = lpbeg[i]; j < nextbeg; ++j)
         {
            const int av = s->inputtoactive[lpind[j]];
            if ( av <= 0 )
               continue;                        /* fixed: already folded into lhs/rhs by sdpi.c (sdpisolver_dsdp.c:1285-1287) */
            if ( s->lhsrow[i] >= 0 )
               dext[(size_t) s->lhsrow[i] * m1 + av] += lpval[j];
            if ( s->rhsrow[i] >= 0 )
               dext[(size_t) s->rhsrow[i] * m1 + av] -= lpval[j];
         }
         if ( s->lhsrow[i] >= 0 )
         {
            dext[(size_t) s->lhsrow[i] * m1] = lplhs[i];
            if ( s->penalty )
               dext[(size_t) s->lhsrow[i] * m1 + s->rvar + 1] = 1.0;
         }
         if ( s->rhsrow[i] >= 0 )
         {
            dext[(size_t) s->rhsrow[i] * m1] = -lprhs[i];
            if ( s->penalty )
               dext[(size_t) s->rhsrow[i] * m1 + s->rvar + 1] = 1.0;
         }
      }
      for (j = 0; j < s->nactivevars; ++j)
      {
         const int v = s->activetoinput[j];
         if ( s->lbrow[v] >= 0 )
         {
            dext[(size_t) s->lbrow[v] * m1 + j + 1] = 1.0;
            dext[(size_t) s->lbrow[v] * m1] = lb[v];
         }
         if ( s->ubrow[v] >= 0 )
         {
            dext[(size_t) s->ubrow[v] * m1 + j + 1] = -1.0;
            dext[(size_t) s->ubrow[v] * m1] = -ub[v];
         }
      }
      if ( s->penalty && rbound )
      {
         row = q - 1;
         dext[(size_t) row * m1 + s->rvar + 1] = 1.0;
      }
      {
         int rc = hipsdp_set_lp(s->engine, dext);
         free(dext);
         if ( rc != HIPSDP_OK )
         {
            SCIPerrorMessage("hipsdp_set_lp failed: %s\n", hipsdp_last_error());
            return SCIP_LPERROR;
         }
      }
   }

   /* ---- optional warm start: not for penalty formulations (sdpisolver_sdpa.cpp:1481,1596) */
   if ( starty != NULL && startZnblocknonz != NULL && startXnblocknonz != NULL && startZrow != NULL && startZcol != NULL
      && startZval != NULL && startXrow != NULL && startXcol != NULL && startXval != NULL && !s->penalty )
   {
      SCIP_RETCODE startrc = loadStartPoint(s, nvars, nsdpblocks, indchanges, nlpcons, starty, startZnblocknonz, startZrow, startZcol,
         startZval, startXnblocknonz, startXrow, startXcol, startXval, nengvars, q);
      if ( startrc != SCIP_OKAY )
         return startrc;
   }

   /* ---- solve, then the tolerance re-solve loop of sdpisolver_dsdp.c:1527-1606 ----------------------------------- */
   freeSolution(s);
   s->nysol = nengvars;
   ALLOC_OR_FAIL(s, &s->ysol, nengvars);
   s->nxlp = q;
   if ( q > 0 )
      ALLOC_OR_FAIL(s, &s->xlp, q);
   s->nXsol = s->nengineblocks;
   if ( s->nengineblocks > 0 )
   {
      ALLOC_OR_FAIL(s, &s->Xsol, s->nengineblocks);
      ALLOC_OR_FAIL(s, &s->Xsize, s->nengineblocks);
      for (b = 0; b < nsdpblocks; ++b)
      {
         if ( s->blockmap[b] >= 0 )
         {
            s->Xsol[s->blockmap[b]] = NULL;
            s->Xsize[s->blockmap[b]] = s->compactsize[b];
         }
      }
   }

   tstage[4] = wallNow();
   /* Settings ladder (sdpisolver_sdpa.cpp:1415-1449 start settings, :1698-1795 retries): a plain solve starts with the settings
    * the caller hands down (UNSOLVED / FAST -> fast); a penalty formulation is solved with the stable ones straight away.  When
    * the result is not acceptable and no penalty formulation is being solved, the node is solved again - cold - with the next
    * more conservative settings before the caller has to fall back to its penalty loop (sdpi.c:3437-3619: 2 + up to 8 further
    * solves).  Every rung ends with the tolerance re-solve loop. */
   {
      int level;
      SCIP_RETCODE retcode;

      if ( s->penalty || startsettings == SCIP_SDPSOLVERSETTING_STABLE || startsettings == SCIP_SDPSOLVERSETTING_PENALTY )
         level = 2;
      else if ( startsettings == SCIP_SDPSOLVERSETTING_MEDIUM )
         level = 1;
      else
         level = 0;
      if ( getenv("HIPSDP_NOLADDER") != NULL && ! s->penalty )
         level = 0;
      for (;;)
      {
         retcode = solveAndCheckTolerances(s, level, timelimit, usedsdpitime, &remaining);
         if ( retcode != SCIP_OKAY )
            return retcode;
         if ( ! s->penalty )
            s->usedsetting = (level == 0) ? SCIP_SDPSOLVERSETTING_FAST : (level == 1 ? SCIP_SDPSOLVERSETTING_MEDIUM : SCIP_SDPSOLVERSETTING_STABLE);
         if ( s->penalty || level >= 2 || statusKnown(s->info.status) || s->info.status == HIPSDP_STATUS_TIMELIM
            || getenv("HIPSDP_NOLADDER") != NULL )
            break;
         {
            SCIP_Bool expired = FALSE;
            retcode = timeLeft(s, timelimit, usedsdpitime, &remaining, &expired);
            if ( retcode != SCIP_OKAY )
               return retcode;
            if ( expired )
            {
               s->info.status = HIPSDP_STATUS_TIMELIM;
               break;
            }
         }
         ++level;
         if ( s->sdpinfo || getenv("HIPSDP_LADDER_LOG") != NULL )
            printf("hipsdp: numerical troubles (status %d) -- solving SDP %d again with %s settings\n", s->info.status, s->sdpcounter,
               level == 1 ? "medium" : "stable");
      }
   }

   tstage[5] = wallNow();
   if ( stagetimes )
      printf("hipsdp stages [ms]: maps %.3f, shape+objective %.3f, blocks %.3f, LP rows + start %.3f, solves + checks %.3f (engine %.3f, %d calls)\n",
         1e3 * (tstage[1] - tstage[0]), 1e3 * (tstage[2] - tstage[1]), 1e3 * (tstage[3] - tstage[2]), 1e3 * (tstage[4] - tstage[3]),
         1e3 * (tstage[5] - tstage[4]), 1e3 * s->info.solve_seconds, s->nsdpcalls);
   if ( s->info.status == HIPSDP_STATUS_TIMELIM )
   {
      s->timelimit = TRUE;
      s->solved = FALSE;
      return SCIP_OKAY;
   }
   s->solved = TRUE;
   {
      SCIP_RETCODE retcode = fetchVectors(s);
      if ( retcode != SCIP_OKAY )
         return retcode;
   }

   /* ---- penalty post-processing (sdpisolver_dsdp.c:1655-1734) ------------------------------------------------------- */
   if ( s->penalty && feasorig != NULL )
   {
      const SCIP_Real rval = s->ysol[s->rvar];
      *feasorig = (rval < s->feastol);
      if ( withobj )
         s->feasorig = *feasorig;
      if ( ! *feasorig && penaltybound != NULL )
      {
         SCIP_Real trace = 0.0;
         int e;
         int t;
         for (e = 0; e < s->nengineblocks; ++e)
         {
            SCIP_RETCODE retcode = ensureX(s, e);
            if ( retcode != SCIP_OKAY )
               return retcode;
            for (t = 0; t < s->Xsize[e]; ++t)
               trace += s->Xsol[e][(size_t) t * s->Xsize[e] + t];
         }
         for (t = 0; t < s->nlpineqs; ++t)       /* LP sides count, variable bounds do not (sdpisolver_sdpa.cpp:1846-1857) */
            trace += s->xlp[t];
         *penaltybound = ((penaltyparam - trace) / penaltyparam < PENALTYBOUNDTOL);
      }
      else if ( penaltybound != NULL )
         *penaltybound = FALSE;
   }
   return SCIP_OKAY;
}

/* ---------------------------------------------------------------------------------------------------------------------- */
/* solution information                                                                                                   */
/* ---------------------------------------------------------------------------------------------------------------------- */

SCIP_Bool SCIPsdpiSolverWasSolved(SCIP_SDPISOLVER* sdpisolver)
{
   assert( sdpisolver != NULL );
   return sdpisolver->solved;
}

static SCIP_Bool statusKnown(int st)
{
   return st == HIPSDP_STATUS_OPTIMAL || st == HIPSDP_STATUS_DINF || st == HIPSDP_STATUS_DUNB || st == HIPSDP_STATUS_PDINF
      || st == HIPSDP_STATUS_OBJLIM;
}

SCIP_Bool SCIPsdpiSolverFeasibilityKnown(SCIP_SDPISOLVER* sdpisolver)
{
   assert( sdpisolver != NULL );
   CHECK_IF_SOLVED_BOOL( sdpisolver );
   return statusKnown(sdpisolver->info.status);
}

/* "primal" = the X-problem, "dual" = the y-problem (sdpisolver.h:37-42).
 *   OPTIMAL (T, T);  DINF: y-problem infeasible, X-ray exists -> (T, F) like DSDP_INFEASIBLE / SDPA pFEAS_dINF
 *   (sdpisolver_dsdp.c:1812-1816);  DUNB: y-ray, X-problem infeasible -> (F, T);  PDINF: (F, F);
 *   OBJLIM: the X-objective passed the limit: (T, F) like SDPA pUNBD (sdpisolver_sdpa.cpp:1954-1958) */
SCIP_RETCODE SCIPsdpiSolverGetSolFeasibility(SCIP_SDPISOLVER* sdpisolver, SCIP_Bool* primalfeasible, SCIP_Bool* dualfeasible)
{
   assert( sdpisolver != NULL && primalfeasible != NULL && dualfeasible != NULL );
   CHECK_IF_SOLVED( sdpisolver );
   switch ( sdpisolver->info.status )
   {
   case HIPSDP_STATUS_OPTIMAL: *primalfeasible = TRUE;  *dualfeasible = TRUE;  break;
   case HIPSDP_STATUS_DINF:    *primalfeasible = TRUE;  *dualfeasible = FALSE; break;
   case HIPSDP_STATUS_OBJLIM:  *primalfeasible = TRUE;  *dualfeasible = FALSE; break;
   case HIPSDP_STATUS_DUNB:    *primalfeasible = FALSE; *dualfeasible = TRUE;  break;
   case HIPSDP_STATUS_PDINF:   *primalfeasible = FALSE; *dualfeasible = FALSE; break;
   default:
      SCIPerrorMessage("HIPSDP doesn't know if primal and dual solutions are feasible\n");
      return SCIP_LPERROR;
   }
   return SCIP_OKAY;
}

SCIP_Bool SCIPsdpiSolverIsPrimalUnbounded(SCIP_SDPISOLVER* sdpisolver)
{
   assert( sdpisolver != NULL );
   CHECK_IF_SOLVED_BOOL( sdpisolver );
   return sdpisolver->info.status == HIPSDP_STATUS_DINF || sdpisolver->info.status == HIPSDP_STATUS_OBJLIM;
}

SCIP_Bool SCIPsdpiSolverIsPrimalInfeasible(SCIP_SDPISOLVER* sdpisolver)
{
   assert( sdpisolver != NULL );
   CHECK_IF_SOLVED_BOOL( sdpisolver );
   return sdpisolver->info.status == HIPSDP_STATUS_DUNB || sdpisolver->info.status == HIPSDP_STATUS_PDINF;
}

SCIP_Bool SCIPsdpiSolverIsPrimalFeasible(SCIP_SDPISOLVER* sdpisolver)
{
   assert( sdpisolver != NULL );
   CHECK_IF_SOLVED_BOOL( sdpisolver );
   return sdpisolver->info.status == HIPSDP_STATUS_OPTIMAL || sdpisolver->info.status == HIPSDP_STATUS_DINF
      || sdpisolver->info.status == HIPSDP_STATUS_OBJLIM;
}

SCIP_Bool SCIPsdpiSolverIsDualUnbounded(SCIP_SDPISOLVER* sdpisolver)
{
   assert( sdpisolver != NULL );
   CHECK_IF_SOLVED_BOOL( sdpisolver );
   return sdpisolver->info.status == HIPSDP_STATUS_DUNB;
}

SCIP_Bool SCIPsdpiSolverIsDualInfeasible(SCIP_SDPISOLVER* sdpisolver)
{
   assert( sdpisolver != NULL );
   CHECK_IF_SOLVED_BOOL( sdpisolver );
   return sdpisolver->info.status == HIPSDP_STATUS_DINF || sdpisolver->info.status == HIPSDP_STATUS_PDINF;
}

SCIP_Bool SCIPsdpiSolverIsDualFeasible(SCIP_SDPISOLVER* sdpisolver)
{
   assert( sdpisolver != NULL );
   CHECK_IF_SOLVED_BOOL( sdpisolver );
   return sdpisolver->info.status == HIPSDP_STATUS_OPTIMAL || sdpisolver->info.status == HIPSDP_STATUS_DUNB;
}

SCIP_Bool SCIPsdpiSolverIsConverged(SCIP_SDPISOLVER* sdpisolver)
{
   assert( sdpisolver != NULL );
   if ( sdpisolver->timelimit || ! sdpisolver->solved )
      return FALSE;
   return statusKnown(sdpisolver->info.status) && sdpisolver->info.status != HIPSDP_STATUS_OBJLIM;
}

SCIP_Bool SCIPsdpiSolverIsObjlimExc(SCIP_SDPISOLVER* sdpisolver)
{
   assert( sdpisolver != NULL );
   CHECK_IF_SOLVED_BOOL( sdpisolver );
   return sdpisolver->info.status == HIPSDP_STATUS_OBJLIM;
}

SCIP_Bool SCIPsdpiSolverIsIterlimExc(SCIP_SDPISOLVER* sdpisolver)
{
   assert( sdpisolver != NULL );
   CHECK_IF_SOLVED_BOOL( sdpisolver );
   return sdpisolver->info.status == HIPSDP_STATUS_ITERLIM;
}

SCIP_Bool SCIPsdpiSolverIsTimelimExc(SCIP_SDPISOLVER* sdpisolver)
{
   assert( sdpisolver != NULL );
   return sdpisolver->timelimit;
}

/* -1 not started, 0 converged, 1 infeasible start, 2 numerical, 3 objlimit, 4 iterlimit, 5 timelimit, 6 user, 7 other
 * (sdpisolver.h:438-448) */
int SCIPsdpiSolverGetInternalStatus(SCIP_SDPISOLVER* sdpisolver)
{
   assert( sdpisolver != NULL );
   if ( sdpisolver->timelimit )
      return 5;
   if ( sdpisolver->engine == NULL || ! sdpisolver->solved )
      return -1;
   switch ( sdpisolver->info.status )
   {
   case HIPSDP_STATUS_OPTIMAL:
   case HIPSDP_STATUS_DINF:
   case HIPSDP_STATUS_DUNB:
   case HIPSDP_STATUS_PDINF:   return 0;
   case HIPSDP_STATUS_NUMERIC: return 2;
   case HIPSDP_STATUS_OBJLIM:  return 3;
   case HIPSDP_STATUS_ITERLIM: return 4;
   case HIPSDP_STATUS_TIMELIM: return 5;
   default:                    return 7;
   }
}

SCIP_Bool SCIPsdpiSolverIsOptimal(SCIP_SDPISOLVER* sdpisolver)
{
   assert( sdpisolver != NULL );
   return SCIPsdpiSolverIsConverged(sdpisolver) && sdpisolver->info.status == HIPSDP_STATUS_OPTIMAL;
}

SCIP_Bool SCIPsdpiSolverIsAcceptable(SCIP_SDPISOLVER* sdpisolver)
{
   assert( sdpisolver != NULL );
   if ( sdpisolver->timelimit )
      return FALSE;
   CHECK_IF_SOLVED_BOOL( sdpisolver );
   return statusKnown(sdpisolver->info.status);
}

SCIP_RETCODE SCIPsdpiSolverIgnoreInstability(SCIP_SDPISOLVER* sdpisolver, SCIP_Bool* success)
{
   (void) sdpisolver;
   assert( success != NULL );
   *success = FALSE;
   return SCIP_OKAY;
}

/* objective recomputed from y as in sdpisolver_dsdp.c:2148-2201, except for infeasible penalty solves */
SCIP_RETCODE SCIPsdpiSolverGetObjval(SCIP_SDPISOLVER* sdpisolver, SCIP_Real* objval)
{
   SCIP_SDPISOLVER* s = sdpisolver;
   int j;
   assert( s != NULL && objval != NULL );
   CHECK_IF_SOLVED( s );
   if ( s->penalty && ! s->feasorig )
   {
      *objval = s->info.dobj;
   }
   else
   {
      *objval = 0.0;
      for (j = 0; j < s->nactivevars; ++j)
         *objval += s->objcoefs[j] * s->ysol[j];
   }
   *objval += s->fixedvarsobjcontr;
   return SCIP_OKAY;
}

SCIP_RETCODE SCIPsdpiSolverGetDualSol(SCIP_SDPISOLVER* sdpisolver, SCIP_Real* objval, SCIP_Real* dualsol)
{
   SCIP_SDPISOLVER* s = sdpisolver;
   int v;
   assert( s != NULL );
   CHECK_IF_SOLVED( s );
   if ( objval != NULL )
   {
      SCIP_RETCODE rc = SCIPsdpiSolverGetObjval(s, objval);
      if ( rc != SCIP_OKAY )
         return rc;
   }
   if ( dualsol != NULL )
   {
      for (v = 0; v < s->nvars; ++v)
      {
         if ( s->inputtoactive[v] > 0 )
            dualsol[v] = s->ysol[s->inputtoactive[v] - 1];
         else
            dualsol[v] = s->fixedvarsval[v];
      }
   }
   return SCIP_OKAY;
}

/* Preoptimal solution (sdpisolver_dsdp.c:2257-2319, sdpisolver_sdpa.cpp:2426-2670): the engine keeps the first iterate that is
 * feasible to tolerance with relative gap below SCIP_SDPPAR_WARMSTARTPOGAP (hipsdp_params.preoptgap); y and, as in the SDPA
 * backend, the primal matrix in the sparse original-index format of SCIPsdpiSolverGetPrimalMatrix.  relax_sdp.c only asks
 * backends named DSDP / SDPA for it (relax_sdp.c:3863, 5021), see INTEGRATION.md.  The two functions below present the
 * preoptimal X / LP multipliers to the regular extraction code by swapping the solution arrays for the duration of the call. */
static SCIP_RETCODE preoptimalPrimal(SCIP_SDPISOLVER* s, SCIP_Bool fill, int nblocks, int* startXnblocknonz, int** startXrow,
   int** startXcol, SCIP_Real** startXval)
{
   SCIP_Real** Xpre = NULL;
   SCIP_Real* xpre = NULL;
   SCIP_Real** saveX = s->Xsol;
   SCIP_Real* savex = s->xlp;
   SCIP_RETCODE retcode = SCIP_OKAY;
   int avail = 0;
   int e;

   if ( HSALLOC(s, &Xpre, s->nengineblocks > 0 ? s->nengineblocks : 1) == NULL )
      return SCIP_NOMEMORY;
   for (e = 0; e < s->nengineblocks; ++e)
      Xpre[e] = NULL;
   if ( HSALLOC(s, &xpre, s->nxlp > 0 ? s->nxlp : 1) == NULL )
      retcode = SCIP_NOMEMORY;
   for (e = 0; e < s->nengineblocks && retcode == SCIP_OKAY; ++e)
   {
      if ( HSALLOC(s, &Xpre[e], s->Xsize[e] * s->Xsize[e]) == NULL )
         retcode = SCIP_NOMEMORY;
      else if ( hipsdp_get_preoptimal_X(s->engine, e, Xpre[e]) != HIPSDP_OK )
         retcode = SCIP_LPERROR;
   }
   if ( retcode == SCIP_OKAY && hipsdp_get_preoptimal(s->engine, &avail, NULL, xpre) != HIPSDP_OK )
      retcode = SCIP_LPERROR;
   if ( retcode == SCIP_OKAY )
   {
      s->Xsol = Xpre;
      s->xlp = xpre;
      if ( fill )
         retcode = SCIPsdpiSolverGetPrimalMatrix(s, nblocks, startXnblocknonz, startXrow, startXcol, startXval);
      else
         retcode = SCIPsdpiSolverGetPrimalNonzeros(s, nblocks, startXnblocknonz);
      s->Xsol = saveX;
      s->xlp = savex;
   }
   for (e = 0; e < s->nengineblocks; ++e)
      if ( Xpre[e] != NULL )
         HSFREE(s, &Xpre[e], s->Xsize[e] * s->Xsize[e]);
   if ( xpre != NULL )
      HSFREE(s, &xpre, s->nxlp > 0 ? s->nxlp : 1);
   HSFREE(s, &Xpre, s->nengineblocks > 0 ? s->nengineblocks : 1);
   return retcode;
}

static SCIP_Bool preoptimalAvailable(SCIP_SDPISOLVER* s)
{
   int avail = 0;
   if ( ! s->solved || s->penalty || s->engine == NULL || s->preoptimalgap <= 0.0 )
      return FALSE;
   if ( hipsdp_get_preoptimal(s->engine, &avail, NULL, NULL) != HIPSDP_OK )
      return FALSE;
   return avail != 0;
}

SCIP_RETCODE SCIPsdpiSolverGetPreoptimalPrimalNonzeros(SCIP_SDPISOLVER* sdpisolver, int nblocks, int* startXnblocknonz)
{
   SCIP_SDPISOLVER* s = sdpisolver;
   assert( s != NULL && startXnblocknonz != NULL );
   if ( ! preoptimalAvailable(s) || nblocks != s->nsdpblocks + 1 )
   {
      if ( nblocks > 0 )
         startXnblocknonz[0] = -1;
      return SCIP_OKAY;
   }
   return preoptimalPrimal(s, FALSE, nblocks, startXnblocknonz, NULL, NULL, NULL);
}

SCIP_RETCODE SCIPsdpiSolverGetPreoptimalSol(SCIP_SDPISOLVER* sdpisolver, SCIP_Bool* success, SCIP_Real* dualsol, int nblocks,
   int* startXnblocknonz, int** startXrow, int** startXcol, SCIP_Real** startXval)
{
   SCIP_SDPISOLVER* s = sdpisolver;
   assert( s != NULL && success != NULL );
   *success = FALSE;
   if ( ! preoptimalAvailable(s) )
   {
      if ( nblocks > 0 && startXnblocknonz != NULL )
         startXnblocknonz[0] = -1;
      return SCIP_OKAY;
   }
   if ( dualsol != NULL )
   {
      SCIP_Real* ypre = NULL;
      int avail = 0;
      int v;
      ALLOC_OR_FAIL(s, &ypre, s->nysol > 0 ? s->nysol : 1);
      if ( hipsdp_get_preoptimal(s->engine, &avail, ypre, NULL) != HIPSDP_OK )
      {
         HSFREE(s, &ypre, s->nysol > 0 ? s->nysol : 1);
         return SCIP_LPERROR;
      }
      for (v = 0; v < s->nvars; ++v)
         dualsol[v] = s->inputtoactive[v] > 0 ? ypre[s->inputtoactive[v] - 1] : s->fixedvarsval[v];
      HSFREE(s, &ypre, s->nysol > 0 ? s->nysol : 1);
   }
   if ( nblocks != -1 )
   {
      SCIP_RETCODE retcode;
      assert( startXnblocknonz != NULL && startXrow != NULL && startXcol != NULL && startXval != NULL );
      if ( nblocks != s->nsdpblocks + 1 )
         return SCIP_LPERROR;
      retcode = preoptimalPrimal(s, TRUE, nblocks, startXnblocknonz, startXrow, startXcol, startXval);
      if ( retcode != SCIP_OKAY )
         return retcode;
   }
   *success = TRUE;
   return SCIP_OKAY;
}

SCIP_RETCODE SCIPsdpiSolverGetPrimalBoundVars(SCIP_SDPISOLVER* sdpisolver, SCIP_Real* lbvals, SCIP_Real* ubvals)
{
   SCIP_SDPISOLVER* s = sdpisolver;
   int v;
   assert( s != NULL && lbvals != NULL && ubvals != NULL );
   CHECK_IF_SOLVED( s );
   for (v = 0; v < s->nvars; ++v)
   {
      lbvals[v] = (s->inputtoactive[v] > 0 && s->lbrow[v] >= 0) ? s->xlp[s->lbrow[v]] : 0.0;
      ubvals[v] = (s->inputtoactive[v] > 0 && s->ubrow[v] >= 0) ? s->xlp[s->ubrow[v]] : 0.0;
   }
   return SCIP_OKAY;
}

SCIP_RETCODE SCIPsdpiSolverGetPrimalLPSides(SCIP_SDPISOLVER* sdpisolver, int nlpcons, int* lpindchanges, SCIP_Real* lplhs,
   SCIP_Real* lprhs, SCIP_Real* lhsvals, SCIP_Real* rhsvals)
{
   SCIP_SDPISOLVER* s = sdpisolver;
   int i;
   (void) lpindchanges; (void) lplhs; (void) lprhs;
   assert( s != NULL && lhsvals != NULL && rhsvals != NULL );
   CHECK_IF_SOLVED( s );
   if ( nlpcons != s->nlpcons )
   {
      SCIPerrorMessage("SCIPsdpiSolverGetPrimalLPSides expected nlpcons = %d but got %d\n", s->nlpcons, nlpcons);
      return SCIP_LPERROR;
   }
   for (i = 0; i < nlpcons; ++i)
   {
      lhsvals[i] = s->lhsrow[i] >= 0 ? s->xlp[s->lhsrow[i]] : 0.0;
      rhsvals[i] = s->rhsrow[i] >= 0 ? s->xlp[s->rhsrow[i]] : 0.0;
   }
   return SCIP_OKAY;
}

/* number of entries the sparse export below will write; last block = LP block (sdpisolver.h:552-554) */
SCIP_RETCODE SCIPsdpiSolverGetPrimalNonzeros(SCIP_SDPISOLVER* sdpisolver, int nblocks, int* startXnblocknonz)
{
   SCIP_SDPISOLVER* s = sdpisolver;
   int b;
   int t;
   assert( s != NULL && startXnblocknonz != NULL );
   CHECK_IF_SOLVED( s );
   if ( nblocks != s->nsdpblocks + 1 )
   {
      SCIPerrorMessage("SCIPsdpiSolverGetPrimalNonzeros expected nblocks = %d but got %d\n", s->nsdpblocks + 1, nblocks);
      return SCIP_LPERROR;
   }
   for (b = 0; b < s->nsdpblocks; ++b)
   {
      const int eb = s->blockmap[b];
      startXnblocknonz[b] = 0;
      if ( eb >= 0 )
      {
         int r;
         int c;
         const int n = s->Xsize[eb];
         SCIP_RETCODE rc = ensureX(s, eb);
         if ( rc != SCIP_OKAY )
            return rc;
         for (r = 0; r < n; ++r)
            for (c = 0; c <= r; ++c)
               if ( REALABS(s->Xsol[eb][(size_t) r * n + c]) > s->epsilon )
                  startXnblocknonz[b]++;
      }
   }
   startXnblocknonz[nblocks - 1] = 0;
   for (t = 0; t < s->nxlp; ++t)
   {
      if ( s->penalty && s->rbound && t == s->nxlp - 1 )
         break;                                    /* the multiplier of r >= 0 belongs to no input row */
      if ( REALABS(s->xlp[t]) > s->epsilon )
         startXnblocknonz[nblocks - 1]++;
   }
   return SCIP_OKAY;
}

SCIP_RETCODE SCIPsdpiSolverGetPrimalMatrix(SCIP_SDPISOLVER* sdpisolver, int nblocks, int* startXnblocknonz, int** startXrow,
   int** startXcol, SCIP_Real** startXval)
{
   SCIP_SDPISOLVER* s = sdpisolver;
   int b;
   int i;
   int v;
   int cnt;
   SCIP_Bool toosmall = FALSE;
   assert( s != NULL && startXnblocknonz != NULL && startXrow != NULL && startXcol != NULL && startXval != NULL );
   CHECK_IF_SOLVED( s );
   if ( nblocks != s->nsdpblocks + 1 )
   {
      SCIPerrorMessage("SCIPsdpiSolverGetPrimalMatrix expected nblocks = %d but got %d\n", s->nsdpblocks + 1, nblocks);
      return SCIP_LPERROR;
   }
   for (b = 0; b < s->nsdpblocks; ++b)
   {
      const int eb = s->blockmap[b];
      const int room = startXnblocknonz[b];
      cnt = 0;
      if ( eb >= 0 )
      {
         int r;
         int c;
         const int n = s->Xsize[eb];
         SCIP_RETCODE rc = ensureX(s, eb);
         if ( rc != SCIP_OKAY )
            return rc;
         for (r = 0; r < n; ++r)
         {
            for (c = 0; c <= r; ++c)
            {
               const SCIP_Real val = s->Xsol[eb][(size_t) r * n + c];
               if ( REALABS(val) > s->epsilon )
               {
                  if ( cnt < room )
                  {
                     startXrow[b][cnt] = s->keptind[b][r];      /* original indices, lower triangle */
                     startXcol[b][cnt] = s->keptind[b][c];
                     startXval[b][cnt] = val;
                  }
                  ++cnt;
               }
            }
         }
      }
      if ( cnt > room )
         toosmall = TRUE;
      startXnblocknonz[b] = cnt;
   }
   /* LP block: position 2 * row (+1 for the rhs side), then 2 * nlpcons + 2 * var (+1 for the upper bound) */
   {
      const int room = startXnblocknonz[nblocks - 1];
      b = nblocks - 1;
      cnt = 0;
      for (i = 0; i < s->nlpcons; ++i)
      {
         int side;
         for (side = 0; side < 2; ++side)
         {
            const int er = side == 0 ? s->lhsrow[i] : s->rhsrow[i];
            if ( er >= 0 && REALABS(s->xlp[er]) > s->epsilon )
            {
               if ( cnt < room )
               {
                  startXrow[b][cnt] = 2 * i + side;
                  startXcol[b][cnt] = 2 * i + side;
                  startXval[b][cnt] = s->xlp[er];
               }
               ++cnt;
            }
         }
      }
      for (v = 0; v < s->nvars; ++v)
      {
         int side;
         if ( s->inputtoactive[v] <= 0 )
            continue;
         for (side = 0; side < 2; ++side)
         {
            const int er = side == 0 ? s->lbrow[v] : s->ubrow[v];
            if ( er >= 0 && REALABS(s->xlp[er]) > s->epsilon )
            {
               if ( cnt < room )
               {
                  startXrow[b][cnt] = 2 * s->nlpcons + 2 * v + side;
                  startXcol[b][cnt] = 2 * s->nlpcons + 2 * v + side;
                  startXval[b][cnt] = s->xlp[er];
               }
               ++cnt;
            }
         }
      }
      if ( cnt > room )
         toosmall = TRUE;
      startXnblocknonz[b] = cnt;
   }
   (void) toosmall;   /* the needed sizes have been written into startXnblocknonz (sdpisolver.h:554) */
   return SCIP_OKAY;
}

/* dense X per ORIGINAL block, zeros at removed indices (sdpisolver_dsdp.c:2467-2545) */
SCIP_RETCODE SCIPsdpiSolverGetPrimalSolutionMatrix(SCIP_SDPISOLVER* sdpisolver, int nsdpblocks, int* sdpblocksizes,
   int** indchanges, int* nremovedinds, int* blockindchanges, SCIP_Real** primalmatrices)
{
   SCIP_SDPISOLVER* s = sdpisolver;
   int b;
   (void) nremovedinds;
   assert( s != NULL && primalmatrices != NULL );
   CHECK_IF_SOLVED( s );
   if ( nsdpblocks != s->nsdpblocks )
   {
      SCIPerrorMessage("SCIPsdpiSolverGetPrimalSolutionMatrix expected nsdpblocks = %d but got %d\n", s->nsdpblocks, nsdpblocks);
      return SCIP_LPERROR;
   }
   for (b = 0; b < nsdpblocks; ++b)
   {
      const int bs = sdpblocksizes[b];
      const int eb = s->blockmap[b];
      int r;
      int c;
      for (r = 0; r < bs * bs; ++r)
         primalmatrices[b][r] = 0.0;
      if ( eb < 0 || blockindchanges[b] < 0 )
         continue;
      {
         const int n = s->Xsize[eb];
         SCIP_RETCODE rc = ensureX(s, eb);
         if ( rc != SCIP_OKAY )
            return rc;
         for (r = 0; r < bs; ++r)
         {
            if ( indchanges[b][r] < 0 )
               continue;
            for (c = 0; c < bs; ++c)
            {
               if ( indchanges[b][c] < 0 )
                  continue;
               primalmatrices[b][(size_t) r * bs + c] = s->Xsol[eb][(size_t) (r - indchanges[b][r]) * n + (c - indchanges[b][c])];
            }
         }
      }
   }
   return SCIP_OKAY;
}

SCIP_Real SCIPsdpiSolverGetMaxPrimalEntry(SCIP_SDPISOLVER* sdpisolver)
{
   SCIP_SDPISOLVER* s = sdpisolver;
   SCIP_Real maxentry = 0.0;
   int e;
   int t;
   assert( s != NULL );
   if ( ! s->solved )
      return 0.0;
   for (e = 0; e < s->nengineblocks; ++e)
   {
      if ( ensureX(s, e) != SCIP_OKAY )
         return 0.0;
      for (t = 0; t < s->Xsize[e] * s->Xsize[e]; ++t)
         if ( s->Xsol[e][t] > maxentry )
            maxentry = s->Xsol[e][t];
   }
   for (t = 0; t < s->nxlp; ++t)
      if ( s->xlp[t] > maxentry )
         maxentry = s->xlp[t];
   return maxentry;
}

SCIP_RETCODE SCIPsdpiSolverGetTime(SCIP_SDPISOLVER* sdpisolver, SCIP_Real* opttime)
{
   assert( sdpisolver != NULL && opttime != NULL );
   *opttime = sdpisolver->opttime;
   return SCIP_OKAY;
}

SCIP_RETCODE SCIPsdpiSolverGetIterations(SCIP_SDPISOLVER* sdpisolver, int* iterations)
{
   assert( sdpisolver != NULL && iterations != NULL );
   *iterations = sdpisolver->timelimitinitial ? 0 : sdpisolver->niterations;
   return SCIP_OKAY;
}

SCIP_RETCODE SCIPsdpiSolverGetSdpCalls(SCIP_SDPISOLVER* sdpisolver, int* calls)
{
   assert( sdpisolver != NULL && calls != NULL );
   *calls = sdpisolver->timelimitinitial ? 0 : sdpisolver->nsdpcalls;
   return SCIP_OKAY;
}

SCIP_RETCODE SCIPsdpiSolverSettingsUsed(SCIP_SDPISOLVER* sdpisolver, SCIP_SDPSOLVERSETTING* usedsetting)
{
   assert( sdpisolver != NULL && usedsetting != NULL );
   if ( ! SCIPsdpiSolverIsAcceptable(sdpisolver) )
      *usedsetting = SCIP_SDPSOLVERSETTING_UNSOLVED;
   else
      *usedsetting = sdpisolver->usedsetting;
   return SCIP_OKAY;
}

/* ---------------------------------------------------------------------------------------------------------------------- */
/* numerical methods                                                                                                      */
/* ---------------------------------------------------------------------------------------------------------------------- */

SCIP_Real SCIPsdpiSolverInfinity(SCIP_SDPISOLVER* sdpisolver)
{
   (void) sdpisolver;
   return HS_INFINITY;
}

SCIP_Bool SCIPsdpiSolverIsInfinity(SCIP_SDPISOLVER* sdpisolver, SCIP_Real val)
{
   (void) sdpisolver;
   return isInf(val);
}

SCIP_RETCODE SCIPsdpiSolverGetRealpar(SCIP_SDPISOLVER* sdpisolver, SCIP_SDPPARAM type, SCIP_Real* dval)
{
   assert( sdpisolver != NULL && dval != NULL );
   switch ( type )
   {
   case SCIP_SDPPAR_EPSILON:          *dval = sdpisolver->epsilon; break;
   case SCIP_SDPPAR_GAPTOL:           *dval = sdpisolver->gaptol; break;
   case SCIP_SDPPAR_FEASTOL:          *dval = sdpisolver->feastol; break;
   case SCIP_SDPPAR_SDPSOLVERFEASTOL: *dval = sdpisolver->sdpsolverfeastol; break;
   case SCIP_SDPPAR_PENALTYPARAM:     *dval = sdpisolver->penaltyparam; break;
   case SCIP_SDPPAR_OBJLIMIT:         *dval = sdpisolver->objlimit; break;
   case SCIP_SDPPAR_WARMSTARTPOGAP:   *dval = sdpisolver->preoptimalgap; break;
   default:
      return SCIP_PARAMETERUNKNOWN;
   }
   return SCIP_OKAY;
}

SCIP_RETCODE SCIPsdpiSolverSetRealpar(SCIP_SDPISOLVER* sdpisolver, SCIP_SDPPARAM type, SCIP_Real dval)
{
   assert( sdpisolver != NULL );
   switch ( type )
   {
   case SCIP_SDPPAR_EPSILON:          sdpisolver->epsilon = dval; break;
   case SCIP_SDPPAR_GAPTOL:           sdpisolver->gaptol = dval; break;
   case SCIP_SDPPAR_FEASTOL:          sdpisolver->feastol = dval; break;
   case SCIP_SDPPAR_SDPSOLVERFEASTOL: sdpisolver->sdpsolverfeastol = dval; break;
   case SCIP_SDPPAR_PENALTYPARAM:     sdpisolver->penaltyparam = dval; break;
   case SCIP_SDPPAR_OBJLIMIT:         sdpisolver->objlimit = dval; break;
   case SCIP_SDPPAR_LAMBDASTAR:       break;                       /* SDPA's initial-point scale: not used */
   case SCIP_SDPPAR_WARMSTARTPOGAP:   sdpisolver->preoptimalgap = dval; break;
   default:
      return SCIP_PARAMETERUNKNOWN;
   }
   return SCIP_OKAY;
}

SCIP_RETCODE SCIPsdpiSolverGetIntpar(SCIP_SDPISOLVER* sdpisolver, SCIP_SDPPARAM type, int* ival)
{
   assert( sdpisolver != NULL && ival != NULL );
   switch ( type )
   {
   case SCIP_SDPPAR_SDPINFO:  *ival = (int) sdpisolver->sdpinfo; break;
   case SCIP_SDPPAR_NTHREADS: *ival = sdpisolver->nthreads; break;
   default:
      return SCIP_PARAMETERUNKNOWN;
   }
   return SCIP_OKAY;
}

SCIP_RETCODE SCIPsdpiSolverSetIntpar(SCIP_SDPISOLVER* sdpisolver, SCIP_SDPPARAM type, int ival)
{
   assert( sdpisolver != NULL );
   switch ( type )
   {
   case SCIP_SDPPAR_SDPINFO:  sdpisolver->sdpinfo = (SCIP_Bool) ival; break;
   case SCIP_SDPPAR_NTHREADS: sdpisolver->nthreads = ival; break;
   default:
      return SCIP_PARAMETERUNKNOWN;
   }
   return SCIP_OKAY;
}

SCIP_RETCODE SCIPsdpiSolverComputeLambdastar(SCIP_SDPISOLVER* sdpisolver, SCIP_Real maxguess)
{
   (void) sdpisolver; (void) maxguess;
   return SCIP_OKAY;
}

/* clamp(1e4 * maxcoeff, 1e5, 1e12), as sdpisolver_dsdp.c:2803-2835 */
SCIP_RETCODE SCIPsdpiSolverComputePenaltyparam(SCIP_SDPISOLVER* sdpisolver, SCIP_Real maxcoeff, SCIP_Real* penaltyparam)
{
   SCIP_Real compval;
   assert( sdpisolver != NULL && penaltyparam != NULL );
   compval = PENALTYPARAM_FACTOR * maxcoeff;
   if ( compval < MIN_PENALTYPARAM )
      compval = MIN_PENALTYPARAM;
   else if ( compval > MAX_PENALTYPARAM )
      compval = MAX_PENALTYPARAM;
   sdpisolver->penaltyparam = compval;
   *penaltyparam = compval;
   return SCIP_OKAY;
}

/* min(1e6 * Gamma, 1e15), as sdpisolver_dsdp.c:2838-2869 */
SCIP_RETCODE SCIPsdpiSolverComputeMaxPenaltyparam(SCIP_SDPISOLVER* sdpisolver, SCIP_Real penaltyparam, SCIP_Real* maxpenaltyparam)
{
   SCIP_Real compval;
   assert( sdpisolver != NULL && maxpenaltyparam != NULL );
   compval = penaltyparam * MAXPENALTYPARAM_FACTOR;
   *maxpenaltyparam = compval < MAX_MAXPENALTYPARAM ? compval : MAX_MAXPENALTYPARAM;
   if ( sdpisolver->penaltyparam > *maxpenaltyparam )
      sdpisolver->penaltyparam = *maxpenaltyparam;
   return SCIP_OKAY;
}

/* ---------------------------------------------------------------------------------------------------------------------- */
/* file interface                                                                                                         */
/* ---------------------------------------------------------------------------------------------------------------------- */

SCIP_RETCODE SCIPsdpiSolverReadSDP(SCIP_SDPISOLVER* sdpisolver, const char* fname)
{
   (void) sdpisolver; (void) fname;
   return SCIP_LPERROR;     /* not implemented in any reference backend either (sdpisolver_dsdp.c:2884-2891) */
}

SCIP_RETCODE SCIPsdpiSolverWriteSDP(SCIP_SDPISOLVER* sdpisolver, const char* fname)
{
   (void) sdpisolver; (void) fname;
   return SCIP_LPERROR;
}
