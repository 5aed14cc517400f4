/* ipm.hip - the interior-point engine: homogeneous self-dual embedding, HKM direction, Mehrotra predictor-corrector.
 *
 * Replaces the third-party arithmetic behind the reference's backends (DSDPSetup/DSDPSolve/DSDPComputeX at
 * src/sdpi/sdpisolver_dsdp.c:1489-1520; SDPA::initializeSolve/solve at src/sdpi/sdpisolver_sdpa.cpp:1600-1670).
 * oracle/ipm_ref.py is the line-by-line CPU restatement used by the parity tests.
 *
 * Per iteration, per dense block (n x n, m1 = m + 1 matrices A_0..A_m stored as rows of length n^2):
 *    chol(Z), Z^-1, chol(X)                                       chol.hip   (MFMA panels)
 *    Schur  Mx_ij = tr(A_i X A_j Z^-1), i, j = 0..m               three MFMA GEMMs (below)
 *    chol(M), M = Mx[1:,1:]; solves                               chol.hip
 *    predictor / corrector right-hand sides and directions        2 passes over A each (kernels.hip) + 4 n^3 GEMMs
 *    step lengths  lambda_min(L^-1 dX L^-T)                       eig.hip (Lanczos)
 * The host thread only reads back a few dozen scalars three times per iteration and decides sigma / alpha / termination.
 */
#include "hs_kernels.h"
#include "../../include/hipsdp.h"
#include <rocprofiler-sdk-roctx/roctx.h>
#include <vector>
#include <utility>
#include <functional>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define HS_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if ( e_ != hipSuccess ) { hs_record_hip_error(e_, "kernel launch", __FILE__, __LINE__); return HS_ERR_HIP; } } while (0)

/* ---- device scalar slots -------------------------------------------------------------------------------------- */
enum
{
   SC_AX0 = 0,     /* <A_0, X> + c^T x */
   SC_DOBJ,        /* b^T y */
   SC_XZ,          /* sum <X, Z> + x^T z */
   SC_RP2,         /* ||rp||^2 */
   SC_RD2,         /* sum_k ||Rd_k||_F^2 + ||rd||^2 */
   SC_RDLPMAX,     /* max |rd| */
   SC_HP2,         /* ||A(X, x)||^2 */
   SC_HD2,         /* ||A^T y - Z||^2 (certificate) */
   SC_S0,
   SC_BUB,         /* b^T M^-1 b */
   SC_BH,          /* sum <B_k, H_k> + beta^T hl */
   SC_WRP,         /* w^T rp */
   SC_BU1,         /* b^T u1 */
   SC_DTAU,
   SC_DKAPPA,
   SC_RATX,        /* LP ratio tests */
   SC_RATZ,
   SC_DEN,
   SC_FIXED_END = 24
};
/* per block k: SC_FIXED_END + 8 k + {0: ||Rd_k||^2, 1..3: lanczos X (theta, resid, steps), 4..6: lanczos Z} */
#define SC_BLK(k, i) (SC_FIXED_END + 8 * (k) + (i))

struct Block
{
   int     n;
   double* A;        /* (m + 1) x n^2: row i at A + i n^2.  Matrices sharded by variable: only the rows [a_r0, a_r1) of the solver exist
                      * (A = Aown - a_r0 n^2 is the address row 0 WOULD have) */
   double* Aown;     /* the allocation behind A */
   double* A0;       /* the constant matrix (row 0): A itself, or a replicated copy on the ranks that do not own row 0 */
   double* A0sep;    /* that copy (NULL where A0 == A) */
   double* Apkown;   /* the allocation behind Apk */
   double *X, *Z, *Rd, *Lz, *LzInv, *Zinv, *Lx, *LxInv, *B, *H, *G, *GZ, *dXa, *dZa, *dX, *dZ, *E, *W, *T1;
   double *dinvz, *dinvx;
   double *Xs, *Zs;  /* saved iterate for step back-off */
   double *T2, *W2;  /* scratch of the second queue */
   double *Xpre;     /* X of the preoptimal iterate (allocated at the first capture) */
   double *P2;       /* A^T([1; u2]): the part of dZ that multiplies dtau, the same for predictor and corrector (allocated on first use) */
   double *pk3;      /* 3 Lp: packed outputs of the three-vector sweep (allocated on first use) */
   double *Apk;      /* (m + 1) x Lp packed lower copy of A for the HBM-bound passes; NULL when memory is short */
   double *pkv;      /* 2 Lp: packed vector in / out */
   long long Lp;
   bool apk_valid;
   bool derived_valid;   /* n <= 64: LxInv, LzInv (and Zinv for n <= 32) belong to the current X, Z (written by the fused factorization) */
   /* SPARSE mode (csrc/sparse.hip): the matrices of the variables are kept as the caller's nonzeros - A is NULL, A0 a dense n x n
    * array of its own; the triplets are collected on the host (sph) and built into the device structure sp before they are used */
   bool sparse;
   bool sp_dirty;
   hs_sparse* sp;
   struct SpHost* sph;
};

struct SpHost
{
   std::vector<int> var, row, col;
   std::vector<double> val;
};

/* phases of an iteration: roctx ranges for rocprofv3 --marker-trace (always) and, with hipsdp_set_profiling, HIP events on the
 * main stream whose intervals are summed per phase (hipsdp_get_phase_times) */
enum { PH_RESID = 0, PH_FACTOR, PH_SCHUR, PH_MSOLVE, PH_PRED, PH_CORR, PH_UPDATE, PH_COUNT };
static const char* const g_phase_names[PH_COUNT] = {"residuals", "factorizations", "schur", "chol_M_solves", "predictor", "corrector",
   "update"};

struct PhaseClock
{
   bool on;
   int open;                                          /* phase whose roctx range is open, or -1 */
   std::vector<hipEvent_t> pool;
   std::vector<std::pair<int, hipEvent_t> > marks;
   double ms[PH_COUNT];
};

struct hipsdp_solver
{
   int device;
   hipStream_t stream;
   hipStream_t stream2;          /* second queue: the Z-side chains run beside the X-side ones (both are latency bound) */
   hipEvent_t evFork, evJoin;
   bool use2;                    /* false for tiny blocks: they are launch bound and the cross-queue events only add latency */
   int m, q;
   int sparse_policy;            /* 0: blocks are never kept as nonzeros, 1: when the caller's count makes it cheaper, 2: whenever a count is given */
   std::vector<Block> blk;
   double* b;        /* m */
   double* Dext;     /* q x (m + 1) */
   /* iterate */
   double *y, *x, *z;
   double tau, kappa;
   /* work vectors */
   double *yt, *dyt, *wt, *AX, *AH, *tmpe, *rp, *rd, *tmpq, *hl, *beta, *elp, *dxa, *dza, *dx, *dz, *xs, *zs, *ys;
   double *u1, *rhs2, *u2, *dy, *dya;
   double *cvec;           /* 2 (m + 1): coefficient vectors [1; u2] and [0; u1] of the split dZ = A^T([0; u1]) - dtau A^T([1; u2]) + eta Rd */
   double *Mx, *Lm, *dinvm, *Slp;
   double *pre_y, *pre_x;  /* preoptimal iterate (params.preoptgap): y and the LP multipliers; X per block in Block::Xpre */
   bool pre_valid;
   double pre_scale;       /* 1 / tau at the capture */
   int* regmask;           /* forced pivots of the last factorization of M (semidefinite pivot rule) */
   double *sc, *red_ws, *gemv_ws, *lan_ws, *lan_ws2, *gws1, *gws2;
   double* gemvt_ws;                   /* row-chunk partial sums of the passes A^T over blocks with few entries (hs_gemv_t_ws) */
   long long gemvt_ws_len;
   double* hsc;            /* pinned, device-visible host mirror of sc followed by the flags and a sequence number: the last kernel
                            * before a read-back stores the scalars there itself and the host waits for the number */
   double* hsc_dev;        /* device view of hsc */
   bool use_publish;       /* HIPSDP_READBACK=copy selects hipMemcpyAsync + hipStreamSynchronize instead */
   unsigned long long pub_seq;
   long long hsc_cap;      /* doubles; kept across re-shapes (pinned allocations are slow) */
   int* trsv_ws;           /* block flags of the multi-workgroup triangular solves */
   int trsv_epoch;
   bool refine_solves;           /* the triangular solves with M correct themselves once with the factor (hs_trsv mode bit 4) */
   int lan_rot[2];                     /* which exchange vector of the one-launch Lanczos runs is the clean one (eig.hip) */
   unsigned long long* lan_sync;       /* device: error word + exchange vectors per matrix of a pair */
   long long gws_len;
   long long gemv_ws_len;
   int* flags;       /* device ints: 0 chol Z, 1 chol X, 2 chol M */
   hs_schur_ws sws;
   bool schur_mode_U, schur_mode_rows;
   bool schur_mode_cols;        /* several ranks (or HIPSDP_SCHUR=K<shards>): column slices of the W formulation + all-reduce */
   int  schur_sim_shards;       /* HIPSDP_SCHUR=K<shards> without a communicator: all slices on this device, one after the other */
   bool schur_mode_forced;      /* HIPSDP_SCHUR set: the single-launch assembly of small problems is not used either */
   double* Mgather;
   long long mx_rows;
   int nsc;
   bool shaped, solved, have_start;
   int last_status;
   double sol_scale;
   hipEvent_t ev0, ev1;
   /* device-resident master copy of the constraint matrices in ORIGINAL indices (survives set_shape) */
   int master_nvars;
   std::vector<int> master_sizes;
   std::vector<int> master_slots;         /* per master block: number of variable slots */
   std::vector<double*> master_A;
   /* multi GPU */
   void* comm; int rank, nranks;
   double* passg;          /* gather buffer of the row-sharded passes: nranks * ceil(m1 / nranks) doubles */
   int shard_passes;       /* -1: by size (>= 64 MB per pass), 0 / 1: HIPSDP_SHARD_PASSES */
   int shardA_req;         /* hipsdp_shard_matrices: 0 replicated (default), 1 by variable, -1 by size against the free memory */
   int var_cw;             /* matrices sharded by variable: columns per slice of the Schur assembly */
   bool var_overlap;       /* ... and the all-to-all of a slice runs on the second queue beside the products of the next one */
   bool shardA;            /* decided by set_shape: this problem's constraint matrices are sharded by variable */
   int a_r0, a_r1;         /* rows of A (0 = constant matrix, i = variable i) this rank holds; [0, m + 1) when replicated */
   hipsdp_params par;
   PhaseClock pc;
   /* shader clock during the Schur assemblies (hipsdp_set_clock_sampling): 4 words per assembly */
   bool clk_on; unsigned long long* clk_buf; int clk_n; double clk_ghz; hipStream_t clk_stream;
   /* one-launch solve of B&B-sized problems (csrc/solve1.hip) */
   double* s1_ws;          /* device workspace (cold matrices and nonzero lists that do not fit into LDS) */
   long long s1_ws_len;
   double* s1_host;        /* pinned, device-visible: HS_S1_OUT_DOUBLES result scalars, then the sequence word, then the history */
   double* s1_host_dev;
   bool s1_sol_host;       /* y, x, z of the last solve are in s1_host too (offset S1_SOL_OFF): hipsdp_get_y / get_lp copy from there */
   bool zero_b, zero_D;    /* set_shape (same shape again) owes b / Dext their zeros: paid by set_obj / set_lp overwriting them, or before the next use */
   /* staging arena of the small uploads of a node (objective, LP rows, gather indices, triplets of the constant matrices, start
    * point): pinned, device-visible; a setter copies its data there and queues an asynchronous copy or a kernel that reads the
    * arena directly - nothing waits.  stage_pending: something queued on `stream` has not been waited for yet (every entry point
    * that touches device data outside that queue calls stage_sync first).  The offset returns to 0 when the queue has drained
    * (set_shape, end of a solve). */
   char* arena_h; char* arena_d; size_t arena_cap, stage_off; bool stage_pending;
   int m_alloc, q_alloc;   /* what the vectors and matrices indexed by variables / LP rows were allocated for (>= m, q: set_shape2) */
   int ncmd;               /* commands of the node's setters waiting in the arena (run by ONE launch before the solve, see NodeCmd) */
   hipEvent_t cmd_ev;      /* recorded behind a flushed command list: the launch reads the list from the arena when it RUNS, so the */
   bool cmd_inflight;      /* slots must not be written again before it has (cmd_new waits for the event) */
   unsigned long long s1_seq;
   int s1_last;            /* 1: the last solve ran in the single launch */
   /* pinned / device staging chunks of hipsdp_master_add_vars (kept until hipsdp_free) */
   void* stage_h[2]; void* stage_d[2]; hipEvent_t stage_ev[2]; long long stage_cap;
};

void hs_comm_phase(int phase);      /* multi.hip: the phase the next collectives are booked under */

static void phase_mark(hipsdp_solver* s, int ph)
{
   if ( s->pc.open >= 0 )
      (void) roctxRangePop();
   s->pc.open = ph;
   if ( ph >= 0 )
      (void) roctxRangePushA(g_phase_names[ph]);
   if ( !s->pc.on )
      return;
   hipEvent_t e = NULL;
   if ( !s->pc.pool.empty() )
   {
      e = s->pc.pool.back();
      s->pc.pool.pop_back();
   }
   else if ( hipEventCreate(&e) != hipSuccess )
      return;
   if ( hipEventRecord(e, s->stream) == hipSuccess )
      s->pc.marks.push_back(std::make_pair(ph, e));
   else
      s->pc.pool.push_back(e);
}

/* with profiling on: turns the recorded marks (the last one being the closing phase_mark(s, -1)) into per-phase sums; the
 * stream must be idle */
static void phase_finish(hipsdp_solver* s)
{
   for (size_t i = 0; i + 1 < s->pc.marks.size(); ++i)
   {
      float ms = 0.f;
      const int ph = s->pc.marks[i].first;
      if ( ph >= 0 && hipEventElapsedTime(&ms, s->pc.marks[i].second, s->pc.marks[i + 1].second) == hipSuccess )
         s->pc.ms[ph] += (double) ms;
   }
   for (auto& mk : s->pc.marks)
      s->pc.pool.push_back(mk.second);
   s->pc.marks.clear();
}

extern "C" const char* hipsdp_phase_name(int phase)
{
   return phase >= 0 && phase < PH_COUNT ? g_phase_names[phase] : "";
}

extern "C" int hipsdp_set_profiling(hipsdp_solver* s, int on)
{
   if ( s == NULL )
      return HIPSDP_ERR_ARG;
   s->pc.on = on != 0;
   return HIPSDP_OK;
}

extern "C" int hipsdp_get_phase_times(hipsdp_solver* s, double* ms)
{
   if ( s == NULL || ms == NULL )
      return HIPSDP_ERR_ARG;
   for (int p = 0; p < PH_COUNT; ++p)
      ms[p] = s->pc.ms[p];
   return HIPSDP_OK;
}

static thread_local char g_err[512] = "";
static int g_live_solvers = 0;      /* handles alive in this process: the device memory pool is trimmed when the last one goes */
static void set_err(const char* msg) { snprintf(g_err, sizeof(g_err), "%s", msg); }

extern "C" const char* hipsdp_last_error(void)
{
   const char* h = hs_last_error();
   if ( g_err[0] != 0 )
      return g_err;
   return h;
}

extern "C" const char* hipsdp_version(void) { return "hipsdp 0.1 (gfx950)"; }

extern "C" int hipsdp_device_count(void)
{
   int n = 0;
   if ( hipGetDeviceCount(&n) != hipSuccess )
      return 0;
   return n;
}

extern "C" void hipsdp_default_params(hipsdp_params* p)
{
   p->gaptol = 1e-5;
   p->feastol = 1e-5;
   p->infeastol = 1e-7;
   p->objlimit = 1e20;
   p->timelimit = 0.0;
   p->gamma = 0.98;
   p->ws_gbytes = 0.0;
   p->maxiter = 100;
   p->verbose = 0;
   p->lanczos_steps = 0;        /* by block size: 24, or 16 when every block has more than 64 rows */
   p->settings = 0;
   p->pabstol = 0.0;
   p->preoptgap = 0.0;
}

template<class T>
static int dalloc(T** p, long long count)
{
   *p = NULL;
   if ( count <= 0 )
      count = 1;
   return hs_pool_alloc((void**) p, (size_t) count * sizeof(T));
}

static void dfree(void* p) { hs_pool_free(p); }

#define STAGE_BYTES (2u << 20)
/* ---- the setters of a node, deferred.  A B&B node is loaded by five or six small calls (clear, objective, gather of the active
 * matrices from the master copy, triplets of the constant matrix, LP rows, start point).  As separate copies and launches they
 * cost 4-6 us of host time each and as much again on the device's queue, one behind the other - a tenth of the solve of such a
 * node.  Instead each call leaves its data and a command in the pinned arena, and ONE launch of one workgroup runs the commands in
 * order just before the solve (or before anything else touches the device data: stage_sync).  Large operands keep the direct path. */
__global__ void __launch_bounds__(1024) k_node_cmds(const NodeCmd* __restrict__ cmds, int ncmd)
{
   __shared__ NodeCmd sc[NC_MAX];
   hs_run_node_cmds(cmds, ncmd, sc);
}

/* room for `bytes` in the staging arena: host address (NULL: does not fit - the caller takes its blocking path), *dev = the same
 * bytes as the device sees them.  The first NC_BYTES of the arena hold the command list. */
static void* stage_take(hipsdp_solver* s, size_t bytes, void** dev)
{
   if ( getenv("HIPSDP_NO_STAGING") != NULL )
      return NULL;
   if ( s->arena_h == NULL )
   {
      void* h = NULL; void* d = NULL;
      if ( hipHostMalloc(&h, STAGE_BYTES, hipHostMallocMapped) != hipSuccess )
         return NULL;
      if ( hipHostGetDevicePointer(&d, h, 0) != hipSuccess )
      {
         (void) hipHostFree(h);
         return NULL;
      }
      s->arena_h = (char*) h; s->arena_d = (char*) d; s->arena_cap = STAGE_BYTES; s->stage_off = 0; s->ncmd = 0;
   }
   if ( s->stage_off < NC_BYTES )
      s->stage_off = NC_BYTES;
   const size_t off = (s->stage_off + 63) & ~(size_t) 63;
   if ( off + bytes > s->arena_cap )
      return NULL;
   s->stage_off = off + bytes;
   if ( dev != NULL )
      *dev = s->arena_d + off;
   return s->arena_h + off;
}
/* launch the waiting commands (no wait) */
static int flush_cmds(hipsdp_solver* s)
{
   if ( s->ncmd > 0 )
   {
      hipLaunchKernelGGL(k_node_cmds, dim3(1), dim3(1024), 0, s->stream, reinterpret_cast<const NodeCmd*>(s->arena_d), s->ncmd);
      HS_LAUNCH_CHECK();
      s->ncmd = 0;
      s->stage_pending = true;
      /* the kernel fetches the list when it runs, not when it is queued: whoever writes slot 0 next waits for this event */
      if ( s->cmd_ev == NULL )
         HS_HIP( hipEventCreateWithFlags(&s->cmd_ev, hipEventDisableTiming) );
      HS_HIP( hipEventRecord(s->cmd_ev, s->stream) );
      s->cmd_inflight = true;
   }
   return HIPSDP_OK;
}
/* a slot for one more command (NULL: the arena is not available, the caller takes its direct path) */
static NodeCmd* cmd_new(hipsdp_solver* s)
{
   if ( s->arena_h == NULL && stage_take(s, 0, NULL) == NULL )
      return NULL;
   if ( s->ncmd >= NC_MAX && flush_cmds(s) != HIPSDP_OK )
      return NULL;
   if ( s->cmd_inflight )
   {
      /* a flushed list may not have been fetched yet (large gathers / clears flush in the middle of a node's setters) */
      if ( hipEventSynchronize(s->cmd_ev) != hipSuccess )
         return NULL;
      s->cmd_inflight = false;
   }
   NodeCmd* q = reinterpret_cast<NodeCmd*>(s->arena_h) + s->ncmd;
   memset(q, 0, sizeof(*q));
   ++s->ncmd;
   return q;
}
/* host -> device of a small array: data and a copy command into the arena; falls back to a blocking copy */
static int stage_upload(hipsdp_solver* s, void* dst, const void* src, size_t bytes)
{
   if ( bytes == 0 )
      return HIPSDP_OK;
   void* dv = NULL;
   void* h = (bytes <= NC_LIMIT * sizeof(double)) ? stage_take(s, bytes, &dv) : NULL;
   NodeCmd* q = (h != NULL) ? cmd_new(s) : NULL;
   if ( q == NULL )
   {
      HS_CALL( flush_cmds(s) );
      HS_HIP( hipStreamSynchronize(s->stream) );
      HS_HIP( hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice) );
      return HIPSDP_OK;
   }
   memcpy(h, src, bytes);
   q->op = NC_COPY; q->n = (long long) (bytes / sizeof(double)); q->dst = (double*) dst; q->src = (const double*) dv;
   return HIPSDP_OK;
}
/* before anything outside the solver's stream touches device data: wait for what the setters queued */
/* clear `count` doubles, in order with the waiting commands */
static int stage_zero(hipsdp_solver* s, double* dst, long long count)
{
   if ( count <= 0 )
      return HIPSDP_OK;
   NodeCmd* q = (count <= NC_LIMIT) ? cmd_new(s) : NULL;
   if ( q != NULL )
   {
      q->op = NC_ZERO; q->n = count; q->dst = dst;
      return HIPSDP_OK;
   }
   HS_CALL( flush_cmds(s) );
   HS_HIP( hipMemsetAsync(dst, 0, (size_t) count * sizeof(double), s->stream) );
   s->stage_pending = true;
   return HIPSDP_OK;
}
static int flush_zeros(hipsdp_solver* s)
{
   if ( s->zero_b )
   {
      HS_CALL( stage_zero(s, s->b, s->m) );
      s->zero_b = false;
   }
   if ( s->zero_D )
   {
      HS_CALL( stage_zero(s, s->Dext, (long long) s->q * (s->m + 1)) );
      s->zero_D = false;
   }
   return HIPSDP_OK;
}
static int stage_sync(hipsdp_solver* s)
{
   if ( s != NULL && (s->zero_b || s->zero_D) )
      HS_CALL( flush_zeros(s) );
   if ( s != NULL && s->ncmd > 0 )
      HS_CALL( flush_cmds(s) );
   if ( s != NULL && s->stage_pending )
   {
      HS_HIP( hipStreamSynchronize(s->stream) );
      s->stage_pending = false;
      s->stage_off = 0;
      s->cmd_inflight = false;
   }
   return HIPSDP_OK;
}

static void free_sparse(Block& B)
{
   if ( B.sp != NULL ) hs_sp_free(B.sp);
   B.sp = NULL;
   delete B.sph;
   B.sph = NULL;
}

static void free_problem(hipsdp_solver* s)
{
   /* blocks go back to a pool that hands them out again without a device synchronisation: nothing may still be running on them */
   if ( s->stream != NULL ) (void) hipStreamSynchronize(s->stream);
   if ( s->stream2 != NULL ) (void) hipStreamSynchronize(s->stream2);
   for (auto& B : s->blk)
   {
      double* ptrs[] = {B.Aown, B.A0sep, B.X, B.Z, B.Rd, B.Lz, B.LzInv, B.Zinv, B.Lx, B.LxInv, B.B, B.H, B.G, B.GZ, B.dXa, B.dZa, B.dX, B.dZ,
         B.E, B.W, B.T1, B.dinvz, B.dinvx, B.Xs, B.Zs, B.Apkown, B.pkv, B.T2, B.W2, B.Xpre, B.P2, B.pk3};
      for (double* p : ptrs) dfree(p);
      free_sparse(B);
   }
   s->blk.clear();
   double* ptrs[] = {s->b, s->Dext, s->y, s->x, s->z, s->yt, s->dyt, s->wt, s->AX, s->AH, s->tmpe, s->rp, s->rd, s->tmpq, s->hl,
      s->beta, s->elp, s->dxa, s->dza, s->dx, s->dz, s->xs, s->zs, s->ys, s->rhs2, s->cvec, s->u2, s->dy, s->dya, s->Mx, s->Lm,
      s->dinvm, s->Slp, s->sc, s->red_ws, s->gemv_ws, s->lan_ws, s->lan_ws2, s->gws1, s->gws2};
   for (double* p : ptrs) dfree(p);
   hs_schur_ws_free(&s->sws);
   dfree(s->Mgather);
   s->Mgather = NULL;
   dfree(s->passg);
   s->passg = NULL;
   dfree(s->trsv_ws);
   s->trsv_ws = NULL;
   dfree(s->gemvt_ws);
   s->gemvt_ws = NULL;
   s->gemvt_ws_len = 0;
   if ( s->lan_sync != NULL )
      (void) hipFree(s->lan_sync);
   s->lan_sync = NULL;
   dfree(s->regmask);
   s->regmask = NULL;
   dfree(s->pre_y); dfree(s->pre_x);
   s->pre_y = s->pre_x = NULL;
   s->pre_valid = false;
   s->b = s->Dext = s->y = s->x = s->z = s->yt = s->dyt = s->wt = s->AX = s->AH = s->tmpe = s->rp = s->rd = s->tmpq = s->hl = NULL;
   s->beta = s->elp = s->dxa = s->dza = s->dx = s->dz = s->xs = s->zs = s->ys = s->u1 = s->rhs2 = s->cvec = s->u2 = s->dy = s->dya = NULL;
   s->Mx = s->Lm = s->dinvm = s->Slp = s->sc = s->red_ws = s->gemv_ws = s->lan_ws = s->lan_ws2 = s->gws1 = s->gws2 = NULL;
   s->flags = NULL;
   s->shaped = false;
   s->solved = false;
}

extern "C" int hipsdp_create(hipsdp_solver** out, int device)
{
   g_err[0] = 0;
   int nd = hipsdp_device_count();
   if ( nd <= 0 )
   {
      set_err("no HIP device available (hipsdp has no CPU fallback)");
      return HIPSDP_ERR_NODEVICE;
   }
   if ( device < 0 || device >= nd )
      return HIPSDP_ERR_ARG;
   HS_HIP( hipSetDevice(device) );
   hipsdp_solver* s = new hipsdp_solver();
   memset((void*) &s->par, 0, sizeof(s->par));
   s->device = device;
   s->m = s->q = 0;
   s->b = s->Dext = s->y = s->x = s->z = NULL;
   s->shaped = s->solved = s->have_start = false;
   s->comm = NULL; s->rank = 0; s->nranks = 1;
   s->shardA_req = 0; s->shardA = false; s->a_r0 = 0; s->a_r1 = 1;
   s->passg = NULL;
   s->shard_passes = getenv("HIPSDP_SHARD_PASSES") != NULL ? atoi(getenv("HIPSDP_SHARD_PASSES")) : -1;
   s->master_nvars = 0;
   s->Mgather = NULL;
   s->schur_mode_rows = false;
   s->schur_mode_cols = false; s->schur_sim_shards = 0;
   s->sws.T = s->sws.U = s->sws.K = s->sws.V = s->sws.U2 = s->sws.V2 = NULL;
   s->sws.evP[0] = s->sws.evP[1] = s->sws.evX[0] = s->sws.evX[1] = NULL; s->sws.ev_g2 = NULL; s->sws.after_g1 = NULL; s->sws.after_g1_arg = NULL;
   s->flags = NULL;
   s->hsc = NULL;
   s->hsc_dev = NULL;
   s->pub_seq = 0;
   {
      const char* rb = getenv("HIPSDP_READBACK");
      s->use_publish = !(rb != NULL && rb[0] == 'c');
   }
   s->hsc_cap = 0;
   s->clk_on = false; s->clk_buf = NULL; s->clk_n = 0; s->clk_ghz = 0.0; s->clk_stream = NULL;
   s->s1_ws = NULL; s->s1_ws_len = 0; s->s1_host = NULL; s->s1_host_dev = NULL; s->s1_seq = 0; s->s1_last = 0; s->s1_sol_host = false; s->zero_b = false; s->zero_D = false;
   s->arena_h = NULL; s->arena_d = NULL; s->arena_cap = 0; s->stage_off = 0; s->stage_pending = false; s->ncmd = 0; s->cmd_ev = NULL; s->cmd_inflight = false; s->m_alloc = 0; s->q_alloc = 0;
   s->trsv_ws = NULL;
   s->pre_y = s->pre_x = NULL;
   s->pre_valid = false;
   s->pre_scale = 1.0;
   s->regmask = NULL;
   s->trsv_epoch = 0;
   s->lan_sync = NULL;
   s->lan_rot[0] = s->lan_rot[1] = 0;
   s->last_status = HIPSDP_STATUS_UNSOLVED;
   s->sol_scale = 1.0;
   s->pc.on = getenv("HIPSDP_PHASES") != NULL && atoi(getenv("HIPSDP_PHASES")) != 0;
   s->pc.open = -1;
   for (int p = 0; p < PH_COUNT; ++p) s->pc.ms[p] = 0.0;
   s->stage_h[0] = s->stage_h[1] = s->stage_d[0] = s->stage_d[1] = NULL;
   s->stage_ev[0] = s->stage_ev[1] = NULL;
   s->stage_cap = 0;
   s->b = s->Dext = s->y = s->x = s->z = s->yt = s->dyt = s->wt = s->AX = s->AH = s->tmpe = s->rp = s->rd = s->tmpq = s->hl = NULL;
   s->beta = s->elp = s->dxa = s->dza = s->dx = s->dz = s->xs = s->zs = s->ys = s->u1 = s->rhs2 = s->cvec = s->u2 = s->dy = s->dya = NULL;
   s->Mx = s->Lm = s->dinvm = s->Slp = s->sc = s->red_ws = s->gemv_ws = s->lan_ws = s->lan_ws2 = s->gws1 = s->gws2 = NULL;
   s->gemvt_ws = NULL;
   s->gemvt_ws_len = 0;
   hipsdp_default_params(&s->par);
   s->sparse_policy = getenv("HIPSDP_SPARSE") != NULL ? atoi(getenv("HIPSDP_SPARSE")) : 1;
   if ( s->sparse_policy < 0 || s->sparse_policy > 2 ) s->sparse_policy = 1;
   if ( hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess
      || hipStreamCreateWithFlags(&s->stream2, hipStreamNonBlocking) != hipSuccess
      || hipEventCreateWithFlags(&s->evFork, hipEventDisableTiming) != hipSuccess
      || hipEventCreateWithFlags(&s->evJoin, hipEventDisableTiming) != hipSuccess
      || hipEventCreate(&s->ev0) != hipSuccess || hipEventCreate(&s->ev1) != hipSuccess )
   {
      delete s;
      set_err("stream/event creation failed");
      return HIPSDP_ERR_HIP;
   }
   (void) __sync_add_and_fetch(&g_live_solvers, 1);
   *out = s;
   return HIPSDP_OK;
}

static void master_free(hipsdp_solver* s)
{
   for (double* p : s->master_A) dfree(p);
   s->master_A.clear();
   s->master_sizes.clear();
   s->master_slots.clear();
   s->master_nvars = 0;
}

extern "C" void hipsdp_free(hipsdp_solver** ps)
{
   if ( ps == NULL || *ps == NULL )
      return;
   hipsdp_solver* s = *ps;
   (void) hipSetDevice(s->device);
   (void) hipStreamSynchronize(s->stream);
   free_problem(s);
   master_free(s);
   if ( s->hsc != NULL ) (void) hipHostFree(s->hsc);
   s->hsc = NULL;
   if ( s->s1_host != NULL ) (void) hipHostFree(s->s1_host);
   s->s1_host = NULL;
   if ( s->arena_h != NULL ) (void) hipHostFree(s->arena_h);
   if ( s->cmd_ev != NULL ) (void) hipEventDestroy(s->cmd_ev);
   s->cmd_ev = NULL;
   s->arena_h = NULL; s->arena_d = NULL; s->arena_cap = 0;
   if ( s->s1_ws != NULL ) (void) hipFree(s->s1_ws);
   s->s1_ws = NULL;
   if ( s->clk_buf != NULL ) (void) hipFree(s->clk_buf);
   s->clk_buf = NULL;
   if ( s->clk_stream != NULL ) (void) hipStreamDestroy(s->clk_stream);
   s->clk_stream = NULL;
   for (auto& mk : s->pc.marks) (void) hipEventDestroy(mk.second);
   for (hipEvent_t e : s->pc.pool) (void) hipEventDestroy(e);
   for (int b = 0; b < 2; ++b)
   {
      if ( s->stage_h[b] != NULL ) (void) hipHostFree(s->stage_h[b]);
      if ( s->stage_d[b] != NULL ) (void) hipFree(s->stage_d[b]);
      if ( s->stage_ev[b] != NULL ) (void) hipEventDestroy(s->stage_ev[b]);
   }
   (void) hipStreamSynchronize(s->stream2);
   (void) hipEventDestroy(s->ev0);
   (void) hipEventDestroy(s->ev1);
   (void) hipEventDestroy(s->evFork);
   (void) hipEventDestroy(s->evJoin);
   (void) hipStreamDestroy(s->stream2);
   (void) hipStreamDestroy(s->stream);
   delete s;
   *ps = NULL;
   if ( __sync_sub_and_fetch(&g_live_solvers, 1) <= 0 )
      hs_pool_trim();
}

/* which storage a block gets: sparse when the caller's nonzero count makes the pair formula the cheaper Schur assembly (never for
 * matrices sharded by variable: that mode exists for dense matrices that do not fit).  HIPSDP_SPARSE=0: never; =2: whenever a count
 * is given (tests) */
static bool block_wants_sparse(const hipsdp_solver* s, int n, int m, long long nnz)
{
   const int mode = s->sparse_policy;
   if ( mode == 0 || nnz < 0 || s->shardA_req > 0 )
      return false;
   if ( mode == 2 )
      return m >= 1;
   return hs_sp_prefers_sparse(n, m, nnz) != 0;
}

extern "C" int hipsdp_set_shape(hipsdp_solver* s, int m, int nblocks, const int* blocksizes, int q)
{
   return hipsdp_set_shape2(s, m, nblocks, blocksizes, q, NULL);
}

extern "C" int hipsdp_sparse_policy(hipsdp_solver* s, int mode)
{
   if ( s == NULL || mode < 0 || mode > 2 )
      return HIPSDP_ERR_ARG;
   s->sparse_policy = mode;
   return HIPSDP_OK;
}

extern "C" int hipsdp_func_attr_sets(int device)
{
   return hs_func_attr_sets(device);
}

extern "C" int hipsdp_mem_info(int device, double* free_bytes, double* total_bytes)
{
   int nd = 0;
   if ( hipGetDeviceCount(&nd) != hipSuccess || nd <= 0 )
      return HIPSDP_ERR_NODEVICE;
   if ( device < 0 || device >= nd )
      return HIPSDP_ERR_ARG;
   HS_HIP( hipSetDevice(device) );
   size_t fr = 0, tot = 0;
   HS_HIP( hipMemGetInfo(&fr, &tot) );
   if ( free_bytes != NULL ) *free_bytes = (double) fr;
   if ( total_bytes != NULL ) *total_bytes = (double) tot;
   return HIPSDP_OK;
}

extern "C" int hipsdp_block_is_sparse(hipsdp_solver* s, int block)
{
   if ( s == NULL || !s->shaped || block < 0 || block >= (int) s->blk.size() )
      return 0;
   return s->blk[block].sparse ? 1 : 0;
}

extern "C" int hipsdp_set_shape2(hipsdp_solver* s, int m, int nblocks, const int* blocksizes, int q, const long long* nnz)
{
   g_err[0] = 0;
   if ( s == NULL || m < 0 || nblocks < 0 || q < 0 )
      return HIPSDP_ERR_ARG;
   HS_HIP( hipSetDevice(s->device) );
   /* (nothing of an earlier load or solve may still be queued: the arena starts over, buffers may go back to the pool) */
   if ( s->ncmd > 0 )
      HS_CALL( flush_cmds(s) );
   if ( hipStreamQuery(s->stream) != hipSuccess )
      HS_HIP( hipStreamSynchronize(s->stream) );
   if ( s->stream2 != NULL && hipStreamQuery(s->stream2) != hipSuccess )
      HS_HIP( hipStreamSynchronize(s->stream2) );
   s->stage_pending = false; s->stage_off = 0; s->ncmd = 0; s->cmd_inflight = false;
   s->zero_b = false; s->zero_D = false; s->s1_sol_host = false;
   /* The same shape again (the next node of a tree with the same fixings pattern, a re-load of the same problem): every
    * allocation is kept - the constraint matrices (GBs at the bench sizes), their packed copy and the Schur workspace cost tens
    * of milliseconds to free and allocate again - and only the contents are reset to what a fresh shape has. */
   /* B&B: the nodes of a tree differ in the number of active variables and LP rows by a few (fixings, tightened bounds).  Small
    * shapes are allocated with some room (m_alloc, q_alloc below), and a shape with the same blocks that fits into it takes the
    * same path: nothing is freed or allocated - sixty pool operations and a dozen clears per node otherwise. */
   const bool fits = s->shaped && (int) s->blk.size() == nblocks && !s->shardA && s->shardA_req == 0
      && ((s->m == m && s->q == q) || (m <= s->m_alloc && q <= s->q_alloc && s->comm == NULL));
   if ( fits && getenv("HIPSDP_NO_SHAPE_REUSE") == NULL )
   {
      bool same = true;
      for (int k = 0; k < nblocks; ++k)
         if ( s->blk[k].n != blocksizes[k] || s->blk[k].sparse != block_wants_sparse(s, blocksizes[k], m, nnz != NULL ? nnz[k] : -1) )
            same = false;
      if ( same )
      {
         const long long m1s = (long long) m + 1;
         if ( s->m != m || s->q != q )
         {
            /* what depends on the counts themselves */
            hs_schur_ws_free(&s->sws);
            s->sws.T = s->sws.U = s->sws.K = s->sws.V = s->sws.U2 = s->sws.V2 = NULL;
            s->sws.evP[0] = s->sws.evP[1] = s->sws.evX[0] = s->sws.evX[1] = NULL; s->sws.ev_g2 = NULL; s->sws.after_g1 = NULL; s->sws.after_g1_arg = NULL;
            s->m = m; s->q = q;
            s->a_r0 = 0; s->a_r1 = (int) m1s;
            s->u1 = s->rhs2 + 2LL * m;
         }
         for (auto& B : s->blk)
         {
            if ( B.sparse )
            {
               HS_CALL( stage_zero(s, B.A0, (long long) B.n * B.n) );
               free_sparse(B);
               B.sph = new SpHost();
               B.sp_dirty = true;
            }
            else
               HS_CALL( stage_zero(s, B.Aown, m1s * B.n * B.n) );
            B.apk_valid = false;
            B.derived_valid = false;
         }
         /* (b and Dext are cleared when they are next used - unless set_obj / set_lp, which replace them whole, come first) */
         s->zero_b = true; s->zero_D = true;
         s->stage_pending = true;                 /* (no wait here: what follows is queued behind the clears) */
         s->have_start = false;
         s->solved = false;
         s->pre_valid = false;
         return HIPSDP_OK;
      }
   }
   free_problem(s);
   s->m = m;
   s->q = q;
   const long long m1 = m + 1;
   /* room for the neighbouring shapes of a tree (see above); large problems are allocated exactly */
   const bool roomy = (m <= 112) && s->comm == NULL && s->shardA_req == 0;
   const int mA = roomy ? m + 16 : m, qA = roomy ? q + 32 : q;
   const long long m1A = (long long) mA + 1;
   s->m_alloc = mA; s->q_alloc = qA;
   int nmax = 1;
   /* Constraint matrices sharded by variable (several ranks only): asked for, or - left to the sizes - when the replicated
    * matrices with their packed copy would take more than 75 % of the device memory.  Every rank decides from the same numbers. */
   s->shardA = false;
   s->a_r0 = 0; s->a_r1 = (int) m1;
   if ( s->comm != NULL && s->shardA_req != 0 && (s->nranks > 1 || s->shardA_req > 0) )
   {
      double bytes = 0.0;
      for (int k = 0; k < nblocks; ++k)
         bytes += 12.0 * (double) m1 * (double) blocksizes[k] * (double) blocksizes[k];
      size_t fr = 0, tot = 0;
      s->shardA = s->shardA_req > 0 || (hipMemGetInfo(&fr, &tot) == hipSuccess && bytes > 0.75 * (double) tot);
      if ( s->shardA )
         hs_var_rows((int) m1, s->nranks, s->rank, &s->a_r0, &s->a_r1);
   }
   const long long arows = s->a_r1 - s->a_r0;
   const long long arowsA = s->shardA ? arows : m1A;
   for (int k = 0; k < nblocks; ++k)
   {
      Block B;
      memset(&B, 0, sizeof(B));
      B.n = blocksizes[k];
      if ( B.n <= 0 )
         return HIPSDP_ERR_ARG;
      if ( B.n > nmax ) nmax = B.n;
      const long long n2 = (long long) B.n * B.n;
      s->blk.push_back(B);
      Block& R = s->blk.back();
      R.sparse = !s->shardA && block_wants_sparse(s, B.n, m, nnz != NULL ? nnz[k] : -1);
      if ( R.sparse )
      {
         /* no (m + 1) x n^2 array: the variables' matrices stay triplets, the constant matrix gets a dense array of its own */
         R.sph = new SpHost();
         R.sp_dirty = true;
         R.A = NULL;
         HS_CALL( dalloc(&R.A0sep, n2) );
         HS_HIP( hipMemsetAsync(R.A0sep, 0, (size_t) n2 * sizeof(double), s->stream) );
         R.A0 = R.A0sep;
      }
      else
      {
      HS_CALL( dalloc(&R.Aown, arowsA * n2) );
      HS_HIP( hipMemsetAsync(R.Aown, 0, (size_t) (arowsA * n2) * sizeof(double), s->stream) );
      R.A = R.Aown - (long long) s->a_r0 * n2;
      R.A0 = R.A;
      }
      if ( !R.sparse && (s->a_r0 > 0 || arows == 0) )
      {
         HS_CALL( dalloc(&R.A0sep, n2) );
         HS_HIP( hipMemsetAsync(R.A0sep, 0, (size_t) n2 * sizeof(double), s->stream) );
         R.A0 = R.A0sep;
      }
      double** mats[] = {&R.X, &R.Z, &R.Rd, &R.Lz, &R.LzInv, &R.Zinv, &R.Lx, &R.LxInv, &R.B, &R.H, &R.G, &R.GZ, &R.dXa, &R.dZa,
         &R.dX, &R.dZ, &R.E, &R.W, &R.T1, &R.Xs, &R.Zs, &R.T2, &R.W2};
      for (double** pm : mats)
         HS_CALL( dalloc(pm, n2) );
      const long long nd = hs_potrf_dinv_len(B.n);
      HS_CALL( dalloc(&R.dinvz, nd) );
      HS_CALL( dalloc(&R.dinvx, nd) );
      R.Lp = (((long long) B.n * (B.n + 1) / 2) + 1) & ~1LL;
      R.apk_valid = false;
      R.Apk = NULL;
      R.pkv = NULL;
      /* the packed copy halves the HBM traffic of the passes; small blocks are launch bound and their passes take one
       * launch less on the full storage */
      if ( getenv("HIPSDP_NOPACK") == NULL && B.n > 64 && !R.sparse )
      {
         if ( hipMalloc((void**) &R.Apkown, (size_t) ((arowsA > 0 ? arowsA : 1) * R.Lp) * sizeof(double)) != hipSuccess )
         {
            (void) hipGetLastError();
            R.Apkown = NULL;              /* not enough memory for the packed copy: the passes use the full storage */
         }
         else
         {
            R.Apk = R.Apkown - (long long) s->a_r0 * R.Lp;
            HS_CALL( dalloc(&R.pkv, 2 * R.Lp) );
            HS_HIP( hipMemsetAsync(R.pkv, 0, (size_t) (2 * R.Lp) * sizeof(double), s->stream) );   /* pad entries stay 0 */
         }
      }
   }
   HS_CALL( dalloc(&s->b, mA) );
   HS_CALL( dalloc(&s->Dext, (long long) qA * m1A) );
   HS_CALL( dalloc(&s->y, mA) ); HS_CALL( dalloc(&s->ys, mA) );
   double** qv[] = {&s->x, &s->z, &s->rd, &s->tmpq, &s->hl, &s->beta, &s->elp, &s->dxa, &s->dza, &s->dx, &s->dz, &s->xs, &s->zs};
   for (double** p : qv) HS_CALL( dalloc(p, qA) );
   double** ev[] = {&s->yt, &s->dyt, &s->wt, &s->AX, &s->AH, &s->tmpe};
   for (double** p : ev) HS_CALL( dalloc(p, m1A) );
   double** mv[] = {&s->rp, &s->u2, &s->dy, &s->dya};
   for (double** p : mv) HS_CALL( dalloc(p, mA) );
   /* [g ; b ; h]: the right-hand side of a direction lives behind the two of the tau elimination, so that the predictor's solve
    * can ride along with them as a third right-hand side (u1 is a view, not an allocation) */
   HS_CALL( dalloc(&s->rhs2, 3LL * mA) );
   s->u1 = s->rhs2 + 2LL * m;
   HS_CALL( dalloc(&s->cvec, 2LL * m1A) );
   HS_CALL( dalloc(&s->Mx, (m1A + 32) * m1A) );    /* row padding: the 2 G shard chunks may overhang by < 2 G rows */
   HS_CALL( dalloc(&s->Lm, (long long) mA * mA) );
   HS_CALL( dalloc(&s->dinvm, hs_potrf_dinv_len(mA)) );
   HS_CALL( dalloc(&s->Slp, (long long) qA * m1A) );
   HS_CALL( dalloc(&s->regmask, mA) );
   s->nsc = SC_FIXED_END + 8 * (nblocks > 0 ? nblocks : 1) + 8;
   HS_CALL( dalloc(&s->sc, s->nsc + 4) );          /* the 8 int flags live behind the scalars: one read-back covers both */
   s->flags = reinterpret_cast<int*>(s->sc + s->nsc);
   if ( s->hsc_cap < s->nsc + 8 )
   {
      if ( s->hsc != NULL ) (void) hipHostFree(s->hsc);
      s->hsc = NULL;
      s->hsc_cap = 2LL * (s->nsc + 8);
      HS_HIP( hipHostMalloc((void**) &s->hsc, (size_t) s->hsc_cap * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent) );
      HS_HIP( hipHostGetDevicePointer((void**) &s->hsc_dev, s->hsc, 0) );
      memset(s->hsc, 0, (size_t) s->hsc_cap * sizeof(double));
   }
   HS_CALL( dalloc(&s->red_ws, 1024) );
   s->gemv_ws_len = 8192 + 4LL * 1024 * 4;
   HS_CALL( dalloc(&s->gemv_ws, s->gemv_ws_len) );
   {
      long long need = 0;
      for (auto& B : s->blk)
      {
         const long long n2 = (long long) B.n * B.n, lp = (long long) B.n * (B.n + 1) / 2;
         need = std::max(need, std::max((long long) hs_gemv_t_chunks(mA + 1, n2) * n2, (long long) hs_gemv_t_chunks(mA + 1, lp) * lp));
      }
      s->gemvt_ws_len = need;
      if ( need > 0 )
         HS_CALL( dalloc(&s->gemvt_ws, need) );
   }
   HS_CALL( dalloc(&s->lan_ws, hs_lanczos_ws(nmax, 256)) );
   HS_CALL( dalloc(&s->lan_ws2, hs_lanczos_ws(nmax, 256)) );
   if ( nmax > 64 )
   {
      if ( s->lan_sync == NULL )
         HS_HIP( hipMalloc((void**) &s->lan_sync, (size_t) hs_lanczos_sync_words() * sizeof(unsigned long long)) );
      HS_CALL( hs_lanczos_sync_reset(s->stream, s->lan_sync, s->lan_rot) );
   }
   s->gws_len = 8LL * nmax * nmax;
   if ( s->gws_len > 8LL * 1024 * 1024 ) s->gws_len = 8LL * 1024 * 1024;
   HS_CALL( dalloc(&s->gws1, s->gws_len) );
   HS_CALL( dalloc(&s->gws2, s->gws_len) );
   HS_CALL( dalloc(&s->trsv_ws, hs_trsv_sync_ws(mA)) );
   if ( m > 2 * 64 )
      HS_CALL( hs_trsv_sync_init(s->stream, m, s->trsv_ws, &s->trsv_epoch) );
   s->trsv_epoch = 0;
   s->sws.T = s->sws.U = s->sws.K = s->sws.V = s->sws.U2 = s->sws.V2 = NULL;
   s->sws.evP[0] = s->sws.evP[1] = s->sws.evX[0] = s->sws.evX[1] = NULL; s->sws.ev_g2 = NULL; s->sws.after_g1 = NULL; s->sws.after_g1_arg = NULL;
   HS_HIP( hipMemsetAsync(s->Dext, 0, (size_t) ((long long) q * m1 > 0 ? (long long) q * m1 : 1) * sizeof(double), s->stream) );
   HS_HIP( hipMemsetAsync(s->b, 0, (size_t) (m > 0 ? m : 1) * sizeof(double), s->stream) );
   HS_HIP( hipStreamSynchronize(s->stream) );
   s->shaped = true;
   s->have_start = false;
   return HIPSDP_OK;
}

extern "C" int hipsdp_set_obj(hipsdp_solver* s, const double* b)
{
   if ( s == NULL || !s->shaped ) return HIPSDP_ERR_ARG;
   HS_HIP( hipSetDevice(s->device) );
   if ( s->m > 0 )
      HS_CALL( stage_upload(s, s->b, b, (size_t) s->m * sizeof(double)) );
   s->zero_b = false;
   s->solved = false;
   return HIPSDP_OK;
}

__global__ void k_scatter_coo(long long nnz, int n, const int* __restrict__ var, const int* __restrict__ row,
   const int* __restrict__ col, const double* __restrict__ val, double* __restrict__ A, int r0, int r1, double* __restrict__ A0,
   int vmax, int* __restrict__ err)
{
   const long long n2 = (long long) n * n;
   for (long long e = (long long) blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += (long long) gridDim.x * blockDim.x)
   {
      const int r = row[e], c = col[e];
      const int v = var[e];
      /* the indices are validated here, on the device (10^8 triplets: a host loop costs as much as the upload): an entry out of
       * range is not written and raises the error flag */
      if ( v < 0 || v > vmax || r < 0 || r >= n || c < 0 || c >= n )
      {
         atomicExch(err, 1);
         continue;
      }
      /* matrices sharded by variable: rows this rank does not hold are skipped, the constant matrix goes to its replicated copy */
      double* a = (v == 0) ? A0 : A + (long long) v * n2;
      if ( v != 0 && (v < r0 || v >= r1) )
         continue;
      a[(long long) r * n + c] = val[e];
      a[(long long) c * n + r] = val[e];
   }
}

extern "C" int hipsdp_add_entries(hipsdp_solver* s, int block, long long nnz, const int* var, const int* row, const int* col,
   const double* val)
{
   if ( s == NULL || !s->shaped || block < 0 || block >= (int) s->blk.size() || nnz < 0 )
      return HIPSDP_ERR_ARG;
   if ( nnz == 0 )
      return HIPSDP_OK;
   HS_HIP( hipSetDevice(s->device) );
   Block& B = s->blk[block];
   if ( B.sparse )
   {
      /* variables: collected on the host (built into the device structure before the next use); constant matrix: dense scatter */
      std::vector<int> cv, cr, cc; std::vector<double> cx;
      for (long long e = 0; e < nnz; ++e)
      {
         if ( var[e] < 0 || var[e] > s->m || row[e] < 0 || row[e] >= B.n || col[e] < 0 || col[e] >= B.n )
         {
            set_err("hipsdp_add_entries: index out of range");
            return HIPSDP_ERR_ARG;
         }
         if ( var[e] == 0 )
         {
            cv.push_back(0); cr.push_back(row[e]); cc.push_back(col[e]); cx.push_back(val[e]);
         }
         else
         {
            B.sph->var.push_back(var[e]); B.sph->row.push_back(row[e]); B.sph->col.push_back(col[e]); B.sph->val.push_back(val[e]);
         }
      }
      B.sp_dirty = true;
      s->solved = false;
      if ( cv.empty() )
         return HIPSDP_OK;
      nnz = (long long) cv.size();
      int *dv, *dr, *dc; double* dval;
      HS_CALL( flush_cmds(s) );
      HS_CALL( dalloc(&dv, nnz) ); HS_CALL( dalloc(&dr, nnz) ); HS_CALL( dalloc(&dc, nnz) ); HS_CALL( dalloc(&dval, nnz) );
      HS_HIP( hipMemcpy(dv, cv.data(), (size_t) nnz * sizeof(int), hipMemcpyHostToDevice) );
      HS_HIP( hipMemcpy(dr, cr.data(), (size_t) nnz * sizeof(int), hipMemcpyHostToDevice) );
      HS_HIP( hipMemcpy(dc, cc.data(), (size_t) nnz * sizeof(int), hipMemcpyHostToDevice) );
      HS_HIP( hipMemcpy(dval, cx.data(), (size_t) nnz * sizeof(double), hipMemcpyHostToDevice) );
      long long g0 = (nnz + 255) / 256; if ( g0 > 4096 ) g0 = 4096;
      HS_HIP( hipMemsetAsync(s->flags + 6, 0, sizeof(int), s->stream) );
      hipLaunchKernelGGL(k_scatter_coo, dim3((unsigned) g0), dim3(256), 0, s->stream, nnz, B.n, dv, dr, dc, dval, B.A0, 0, 1, B.A0, 0, s->flags + 6);
      HS_LAUNCH_CHECK();
      HS_HIP( hipStreamSynchronize(s->stream) );
      dfree(dv); dfree(dr); dfree(dc); dfree(dval);
      return HIPSDP_OK;
   }
   if ( nnz <= 16384 )
   {
      /* a few triplets (the constant matrix of a node): indices checked here, the triplets through the arena, no wait */
      bool ok = true;
      for (long long e = 0; e < nnz; ++e)
         if ( var[e] < 0 || var[e] > s->m || row[e] < 0 || row[e] >= B.n || col[e] < 0 || col[e] >= B.n )
            ok = false;
      if ( !ok )
      {
         set_err("hipsdp_add_entries: index out of range");
         return HIPSDP_ERR_ARG;
      }
      void* dvp = NULL;
      char* h = (char*) stage_take(s, (size_t) nnz * (3 * sizeof(int) + sizeof(double)) + 64, &dvp);
      if ( h != NULL )
      {
         /* values first (8-byte aligned), then the three index arrays */
         memcpy(h, val, (size_t) nnz * sizeof(double));
         int* hi = (int*) (h + (size_t) nnz * sizeof(double));
         memcpy(hi, var, (size_t) nnz * sizeof(int));
         memcpy(hi + nnz, row, (size_t) nnz * sizeof(int));
         memcpy(hi + 2 * nnz, col, (size_t) nnz * sizeof(int));
         const double* dval = (const double*) dvp;
         const int* di = (const int*) ((const char*) dvp + (size_t) nnz * sizeof(double));
         NodeCmd* q = cmd_new(s);
         if ( q != NULL )
         {
            q->op = NC_SCATTER; q->n = nnz; q->i0 = B.n; q->i1 = s->a_r0; q->i2 = s->a_r1; q->idx = di; q->src = dval; q->dst = B.A; q->dst2 = B.A0;
         }
         else
         {
            long long g = (nnz + 255) / 256; if ( g > 4096 ) g = 4096;
            HS_CALL( flush_cmds(s) );
            hipLaunchKernelGGL(k_scatter_coo, dim3((unsigned) g), dim3(256), 0, s->stream, nnz, B.n, di, di + nnz, di + 2 * nnz, dval, B.A,
               s->a_r0, s->a_r1, B.A0, s->m, s->flags + 6);
            HS_LAUNCH_CHECK();
            s->stage_pending = true;
         }
         B.apk_valid = false;
         s->solved = false;
         return HIPSDP_OK;
      }
   }
   int *dv, *dr, *dc; double* dval;
   HS_CALL( flush_cmds(s) );
   HS_CALL( dalloc(&dv, nnz) ); HS_CALL( dalloc(&dr, nnz) ); HS_CALL( dalloc(&dc, nnz) ); HS_CALL( dalloc(&dval, nnz) );
   HS_HIP( hipMemcpyAsync(dv, var, (size_t) nnz * sizeof(int), hipMemcpyHostToDevice, s->stream) );
   HS_HIP( hipMemcpyAsync(dr, row, (size_t) nnz * sizeof(int), hipMemcpyHostToDevice, s->stream) );
   HS_HIP( hipMemcpyAsync(dc, col, (size_t) nnz * sizeof(int), hipMemcpyHostToDevice, s->stream) );
   HS_HIP( hipMemcpyAsync(dval, val, (size_t) nnz * sizeof(double), hipMemcpyHostToDevice, s->stream) );
   long long g = (nnz + 255) / 256; if ( g > 4096 ) g = 4096;
   int herr = 0;
   HS_HIP( hipMemsetAsync(s->flags + 6, 0, sizeof(int), s->stream) );
   hipLaunchKernelGGL(k_scatter_coo, dim3((unsigned) g), dim3(256), 0, s->stream, nnz, B.n, dv, dr, dc, dval, B.A, s->a_r0, s->a_r1, B.A0,
      s->m, s->flags + 6);
   HS_LAUNCH_CHECK();
   HS_HIP( hipMemcpyAsync(&herr, s->flags + 6, sizeof(int), hipMemcpyDeviceToHost, s->stream) );
   HS_HIP( hipStreamSynchronize(s->stream) );
   dfree(dv); dfree(dr); dfree(dc); dfree(dval);
   B.apk_valid = false;
   s->solved = false;
   if ( herr != 0 )
   {
      set_err("hipsdp_add_entries: index out of range");
      return HIPSDP_ERR_ARG;
   }
   return HIPSDP_OK;
}

/* ---- master copy: the matrices A_v of ALL variables in original indices, uploaded once and kept across node solves.  A
 * node's compact block (active variables, kept rows/columns) is then gathered on the device (the
 * sdprow/sdpcol/sdpval arrays do not change between the nodes of a branch-and-bound run). */
extern "C" int hipsdp_master_define(hipsdp_solver* s, int nvars, int nblocks, const int* blocksizes, const int* nblockvars)
{
   if ( s == NULL || nvars < 0 || nblocks < 0 )
      return HIPSDP_ERR_ARG;
   HS_HIP( hipSetDevice(s->device) );
   HS_CALL( stage_sync(s) );
   HS_HIP( hipStreamSynchronize(s->stream) );
   master_free(s);
   s->master_nvars = nvars;
   for (int b = 0; b < nblocks; ++b)
   {
      if ( blocksizes[b] <= 0 )
         return HIPSDP_ERR_ARG;
      double* p = NULL;
      /* one slot per variable that appears in the block (nblockvars == NULL: one per variable) */
      const int slots = nblockvars != NULL ? nblockvars[b] : nvars;
      if ( slots < 0 || slots > nvars )
         return HIPSDP_ERR_ARG;
      const long long cnt = (long long) slots * blocksizes[b] * blocksizes[b];
      const int rc = dalloc(&p, cnt);
      if ( rc != HS_OK )
      {
         master_free(s);           /* the caller falls back to loading the node's block directly */
         return rc;
      }
      HS_HIP( hipMemsetAsync(p, 0, (size_t) (cnt > 0 ? cnt : 1) * sizeof(double), s->stream) );
      s->master_A.push_back(p);
      s->master_sizes.push_back(blocksizes[b]);
      s->master_slots.push_back(slots);
   }
   HS_HIP( hipStreamSynchronize(s->stream) );
   return HIPSDP_OK;
}

extern "C" int hipsdp_master_add_entries(hipsdp_solver* s, int block, long long nnz, const int* var, const int* row,
   const int* col, const double* val)
{
   if ( s == NULL || block < 0 || block >= (int) s->master_A.size() || nnz < 0 )
      return HIPSDP_ERR_ARG;
   if ( nnz == 0 )
      return HIPSDP_OK;
   HS_HIP( hipSetDevice(s->device) );
   HS_CALL( stage_sync(s) );
   const int n = s->master_sizes[block];
   int *dv, *dr, *dc; double* dval;
   int* derr = NULL;
   HS_CALL( dalloc(&derr, 1) );
   HS_CALL( dalloc(&dv, nnz) ); HS_CALL( dalloc(&dr, nnz) ); HS_CALL( dalloc(&dc, nnz) ); HS_CALL( dalloc(&dval, nnz) );
   HS_HIP( hipMemcpyAsync(dv, var, (size_t) nnz * sizeof(int), hipMemcpyHostToDevice, s->stream) );
   HS_HIP( hipMemcpyAsync(dr, row, (size_t) nnz * sizeof(int), hipMemcpyHostToDevice, s->stream) );
   HS_HIP( hipMemcpyAsync(dc, col, (size_t) nnz * sizeof(int), hipMemcpyHostToDevice, s->stream) );
   HS_HIP( hipMemcpyAsync(dval, val, (size_t) nnz * sizeof(double), hipMemcpyHostToDevice, s->stream) );
   long long g = (nnz + 255) / 256; if ( g > 4096 ) g = 4096;
   /* slot 0 of the master copy is a variable like the others: A0 = A, nothing is skipped by the row range */
   int herr = 0;
   HS_HIP( hipMemsetAsync(derr, 0, sizeof(int), s->stream) );
   hipLaunchKernelGGL(k_scatter_coo, dim3((unsigned) g), dim3(256), 0, s->stream, nnz, n, dv, dr, dc, dval, s->master_A[block], 0, 2147483647,
      s->master_A[block], s->master_slots[block] - 1, derr);
   HS_LAUNCH_CHECK();
   HS_HIP( hipMemcpyAsync(&herr, derr, sizeof(int), hipMemcpyDeviceToHost, s->stream) );
   HS_HIP( hipStreamSynchronize(s->stream) );
   dfree(dv); dfree(dr); dfree(dc); dfree(dval); dfree(derr);
   if ( herr != 0 )
   {
      set_err("hipsdp_master_add_entries: index out of range");
      return HIPSDP_ERR_ARG;
   }
   return HIPSDP_OK;
}

/* The same upload straight from the caller's per-variable arrays (the layout of sdpisolver.h: sdprow[b][k], sdpcol[b][k],
 * sdpval[b][k] with sdpnblockvarnonz[b][k] entries each), streamed through two pinned staging chunks: the host fills one chunk
 * (block copies) while the copy engine moves the other and the scatter kernel consumes it.  No concatenated host copy (2.25 GB
 * of freshly faulted pageable memory at n = 500, m = 1000) and no pageable H2D transfer. */
#define HS_STAGE_ENTRIES (4LL << 20)
static int stage_ensure(hipsdp_solver* s, long long want)
{
   long long cap = want < HS_STAGE_ENTRIES ? want : HS_STAGE_ENTRIES;
   if ( cap < 1024 ) cap = 1024;
   cap = (cap + 63) & ~63LL;              /* the value array follows three int arrays: keep it 8-byte aligned */
   if ( s->stage_cap >= cap )
      return HS_OK;
   for (int b = 0; b < 2; ++b)
   {
      if ( s->stage_h[b] != NULL ) (void) hipHostFree(s->stage_h[b]);
      if ( s->stage_d[b] != NULL ) (void) hipFree(s->stage_d[b]);
      s->stage_h[b] = NULL; s->stage_d[b] = NULL;
      if ( s->stage_ev[b] == NULL )
         HS_HIP( hipEventCreateWithFlags(&s->stage_ev[b], hipEventDisableTiming) );
   }
   s->stage_cap = 0;
   const size_t bytes = (size_t) cap * (3 * sizeof(int) + sizeof(double));
   for (int b = 0; b < 2; ++b)
   {
      HS_HIP( hipHostMalloc((void**) &s->stage_h[b], bytes, hipHostMallocDefault) );
      HS_HIP( hipMalloc((void**) &s->stage_d[b], bytes) );
   }
   s->stage_cap = cap;
   return HS_OK;
}

extern "C" int hipsdp_master_add_vars(hipsdp_solver* s, int block, int nslots, const int* nnz, const int* const* row,
   const int* const* col, const double* const* val)
{
   if ( s == NULL || block < 0 || block >= (int) s->master_A.size() || nslots < 0 || nslots > s->master_slots[block]
      || (nslots > 0 && (nnz == NULL || row == NULL || col == NULL || val == NULL)) )
      return HIPSDP_ERR_ARG;
   long long total = 0;
   for (int k = 0; k < nslots; ++k)
   {
      if ( nnz[k] < 0 )
         return HIPSDP_ERR_ARG;
      total += nnz[k];
   }
   if ( total == 0 )
      return HIPSDP_OK;
   HS_HIP( hipSetDevice(s->device) );
   HS_CALL( stage_sync(s) );
   HS_CALL( stage_ensure(s, total) );
   const long long cap = s->stage_cap;
   const int n = s->master_sizes[block];
   int* derr = NULL;
   HS_CALL( dalloc(&derr, 1) );
   HS_HIP( hipMemsetAsync(derr, 0, sizeof(int), s->stream) );
   int k = 0;
   long long off = 0;                  /* entries of slot k already shipped */
   int set = 0;
   bool used[2] = {false, false};
   while ( k < nslots )
   {
      if ( used[set] )
         HS_HIP( hipEventSynchronize(s->stage_ev[set]) );       /* the chunk's previous contents have been consumed */
      char* hb = (char*) s->stage_h[set];
      int* hvar = (int*) hb;
      int* hrow = hvar + cap;
      int* hcol = hrow + cap;
      double* hval = (double*) (hcol + cap);
      long long fill = 0;
      while ( k < nslots && fill < cap )
      {
         const long long left = nnz[k] - off;
         const long long take = left < cap - fill ? left : cap - fill;
         if ( take > 0 )
         {
            for (long long t = 0; t < take; ++t)
               hvar[fill + t] = k;
            memcpy(hrow + fill, row[k] + off, (size_t) take * sizeof(int));
            memcpy(hcol + fill, col[k] + off, (size_t) take * sizeof(int));
            memcpy(hval + fill, val[k] + off, (size_t) take * sizeof(double));
            fill += take;
            off += take;
         }
         if ( off >= nnz[k] )
         {
            ++k;
            off = 0;
         }
      }
      if ( fill == 0 )
         break;
      char* db = (char*) s->stage_d[set];
      int* dvar = (int*) db;
      int* drow = dvar + cap;
      int* dcol = drow + cap;
      double* dval = (double*) (dcol + cap);
      HS_HIP( hipMemcpyAsync(dvar, hvar, (size_t) fill * sizeof(int), hipMemcpyHostToDevice, s->stream) );
      HS_HIP( hipMemcpyAsync(drow, hrow, (size_t) fill * sizeof(int), hipMemcpyHostToDevice, s->stream) );
      HS_HIP( hipMemcpyAsync(dcol, hcol, (size_t) fill * sizeof(int), hipMemcpyHostToDevice, s->stream) );
      HS_HIP( hipMemcpyAsync(dval, hval, (size_t) fill * sizeof(double), hipMemcpyHostToDevice, s->stream) );
      long long g = (fill + 255) / 256; if ( g > 4096 ) g = 4096;
      hipLaunchKernelGGL(k_scatter_coo, dim3((unsigned) g), dim3(256), 0, s->stream, fill, n, dvar, drow, dcol, dval, s->master_A[block], 0,
         2147483647, s->master_A[block], s->master_slots[block] - 1, derr);
      HS_LAUNCH_CHECK();
      HS_HIP( hipEventRecord(s->stage_ev[set], s->stream) );
      used[set] = true;
      set ^= 1;
   }
   int herr = 0;
   HS_HIP( hipMemcpyAsync(&herr, derr, sizeof(int), hipMemcpyDeviceToHost, s->stream) );
   HS_HIP( hipStreamSynchronize(s->stream) );
   dfree(derr);
   if ( herr != 0 )
   {
      set_err("hipsdp_master_add_vars: index out of range");
      return HIPSDP_ERR_ARG;
   }
   return HIPSDP_OK;
}

/* A_engine[a + 1][r'][c'] = master[slot[a]][kept[r']][kept[c']], zero for slot[a] = -1 (variable absent from the block) */
__global__ void k_master_gather(int nactive, int nk, int N, const int* __restrict__ act, const int* __restrict__ kept,
   const double* __restrict__ master, double* __restrict__ A)
{
   const long long nk2 = (long long) nk * nk;
   const long long total = (long long) nactive * nk2;
   for (long long e = (long long) blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long) gridDim.x * blockDim.x)
   {
      const long long a = e / nk2;
      const long long rc = e - a * nk2;
      const int r = (int) (rc / nk), c = (int) (rc - (long long) r * nk);
      A[(a + 1) * nk2 + rc] = act[a] >= 0 ? master[((long long) act[a] * N + kept[r]) * N + kept[c]] : 0.0;
   }
}

extern "C" int hipsdp_master_gather(hipsdp_solver* s, int engine_block, int master_block, int nactive, const int* activevars,
   int nkept, const int* kept)
{
   if ( s == NULL || !s->shaped || engine_block < 0 || engine_block >= (int) s->blk.size() || master_block < 0
      || master_block >= (int) s->master_A.size() || nactive < 0 || nactive > s->m || nkept != s->blk[engine_block].n )
      return HIPSDP_ERR_ARG;
   if ( s->shardA || s->blk[engine_block].sparse )
   {
      set_err("hipsdp_master_gather: not available with matrices sharded by variable or kept as nonzeros");
      return HIPSDP_ERR_ARG;
   }
   if ( nactive == 0 )
      return HIPSDP_OK;
   HS_HIP( hipSetDevice(s->device) );
   const int N = s->master_sizes[master_block];
   for (int a = 0; a < nactive; ++a)
      if ( activevars[a] < -1 || activevars[a] >= s->master_slots[master_block] )
         return HIPSDP_ERR_ARG;
   for (int r = 0; r < nkept; ++r)
      if ( kept[r] < 0 || kept[r] >= N )
         return HIPSDP_ERR_ARG;
   Block& B = s->blk[engine_block];
   long long g = ((long long) nactive * nkept * nkept + 255) / 256; if ( g > 65536 ) g = 65536;
   {
      /* the two index lists through the arena: the kernel reads them there, nothing is copied and nothing waited for */
      void* dv = NULL;
      int* hidx = (int*) stage_take(s, (size_t) (nactive + nkept) * sizeof(int), &dv);
      if ( hidx != NULL )
      {
         memcpy(hidx, activevars, (size_t) nactive * sizeof(int));
         memcpy(hidx + nactive, kept, (size_t) nkept * sizeof(int));
         const int* didx = (const int*) dv;
         NodeCmd* q = ((long long) nactive * nkept * nkept <= NC_LIMIT) ? cmd_new(s) : NULL;
         if ( q != NULL )
         {
            q->op = NC_GATHER; q->i0 = nactive; q->i1 = nkept; q->i2 = N; q->idx = didx; q->src = s->master_A[master_block]; q->dst = B.A;
         }
         else
         {
            HS_CALL( flush_cmds(s) );
            hipLaunchKernelGGL(k_master_gather, dim3((unsigned) g), dim3(256), 0, s->stream, nactive, nkept, N, didx, didx + nactive,
               s->master_A[master_block], B.A);
            HS_LAUNCH_CHECK();
            s->stage_pending = true;
         }
      }
      else
      {
         int *dact, *dkept;
         HS_CALL( flush_cmds(s) );
         HS_CALL( dalloc(&dact, nactive) ); HS_CALL( dalloc(&dkept, nkept) );
         HS_HIP( hipMemcpyAsync(dact, activevars, (size_t) nactive * sizeof(int), hipMemcpyHostToDevice, s->stream) );
         HS_HIP( hipMemcpyAsync(dkept, kept, (size_t) nkept * sizeof(int), hipMemcpyHostToDevice, s->stream) );
         hipLaunchKernelGGL(k_master_gather, dim3((unsigned) g), dim3(256), 0, s->stream, nactive, nkept, N, dact, dkept,
            s->master_A[master_block], B.A);
         HS_LAUNCH_CHECK();
         HS_HIP( hipStreamSynchronize(s->stream) );
         dfree(dact); dfree(dkept);
      }
   }
   B.apk_valid = false;
   s->solved = false;
   return HIPSDP_OK;
}

extern "C" int hipsdp_set_block_dense(hipsdp_solver* s, int block, const double* A)
{
   if ( s == NULL || !s->shaped || block < 0 || block >= (int) s->blk.size() )
      return HIPSDP_ERR_ARG;
   HS_HIP( hipSetDevice(s->device) );
   HS_CALL( stage_sync(s) );
   Block& B = s->blk[block];
   if ( B.sparse )
   {
      set_err("hipsdp_set_block_dense: the block is kept as nonzeros (hipsdp_set_shape2 with a count): use hipsdp_add_entries");
      return HIPSDP_ERR_ARG;
   }
   const size_t n2b = (size_t) B.n * B.n * sizeof(double);
   if ( s->a_r1 > s->a_r0 )
      HS_HIP( hipMemcpy(B.Aown, A + (size_t) s->a_r0 * B.n * B.n, (size_t) (s->a_r1 - s->a_r0) * n2b, hipMemcpyHostToDevice) );
   if ( B.A0sep != NULL )
      HS_HIP( hipMemcpy(B.A0sep, A, n2b, hipMemcpyHostToDevice) );
   B.apk_valid = false;
   s->solved = false;
   return HIPSDP_OK;
}

extern "C" int hipsdp_set_lp(hipsdp_solver* s, const double* Dext)
{
   if ( s == NULL || !s->shaped ) return HIPSDP_ERR_ARG;
   HS_HIP( hipSetDevice(s->device) );
   if ( s->q > 0 )
      HS_CALL( stage_upload(s, s->Dext, Dext, (size_t) s->q * (s->m + 1) * sizeof(double)) );
   s->zero_D = false;
   s->solved = false;
   return HIPSDP_OK;
}

int hs_gen_dense(hipStream_t s, int n, int i0, int i1, long long seed, double* A);
static int ensure_packed(hipsdp_solver* s);
static int pass_A(hipsdp_solver* s, Block& B, const double* V, double* out);
static int pass_AT(hipsdp_solver* s, Block& B, const double* coef, double sa, const double* add, double* out);

extern "C" int hipsdp_gen_planted(hipsdp_solver* s, int n, int m, long long seed, const double* Xstar, const double* Zstar,
   const double* ystar, double* b_out)
{
   if ( s == NULL || !s->shaped || s->blk.size() != 1 || s->blk[0].n != n || s->m != m || s->q != 0 || s->blk[0].sparse )
      return HIPSDP_ERR_ARG;
   HS_HIP( hipSetDevice(s->device) );
   HS_CALL( stage_sync(s) );
   Block& B = s->blk[0];
   const long long n2 = (long long) n * n;
   const int m1 = m + 1;
   hipStream_t st = s->stream;
   HS_CALL( hs_gen_dense(st, n, s->a_r0 > 1 ? s->a_r0 : 1, s->a_r1, seed, B.A) );
   /* A_0 = sum_i ystar_i A_i - Zstar : coefficient vector [0, ystar] over all m + 1 rows (row 0 is overwritten) */
   HS_HIP( hipMemcpyAsync(B.Z, Zstar, (size_t) n2 * sizeof(double), hipMemcpyHostToDevice, st) );
   HS_HIP( hipMemcpyAsync(B.X, Xstar, (size_t) n2 * sizeof(double), hipMemcpyHostToDevice, st) );
   HS_HIP( hipMemsetAsync(s->yt, 0, sizeof(double), st) );
   HS_HIP( hipMemcpyAsync(s->yt + 1, ystar, (size_t) m * sizeof(double), hipMemcpyHostToDevice, st) );
   HS_HIP( hipMemsetAsync(B.A0, 0, (size_t) n2 * sizeof(double), st) );
   if ( s->shardA )
   {
      /* every rank generated the matrices it holds; the two sums over all variables go through the sharded passes (row 0 is
       * zero and has coefficient 0 in the first, and its entry of the second is not used) */
      B.apk_valid = false;
      HS_CALL( ensure_packed(s) );
      HS_CALL( pass_AT(s, B, s->yt, -1.0, B.Z, B.T1) );
      HS_CALL( hs_symmetrize(st, B.T1, n) );
      HS_CALL( hs_copy(st, B.A0, B.T1, n2) );
      HS_CALL( pass_A(s, B, B.X, s->AX) );
   }
   else
   {
   HS_CALL( hs_gemv_t(st, m1, n2, B.A, n2, s->yt, -1.0, B.Z, B.T1) );
   HS_CALL( hs_symmetrize(st, B.T1, n) );
   HS_CALL( hs_copy(st, B.A, B.T1, n2) );
   /* b = A(Xstar) */
   const double* v = B.X;
   HS_CALL( hs_gemv_n(st, m1, n2, B.A, n2, 1, &v, s->AX, m1, s->gemv_ws, s->gemv_ws_len) );
   }
   HS_CALL( hs_copy(st, s->b, s->AX + 1, m) );
   HS_HIP( hipMemcpyAsync(b_out, s->b, (size_t) m * sizeof(double), hipMemcpyDeviceToHost, st) );
   HS_HIP( hipStreamSynchronize(st) );
   B.apk_valid = false;
   s->solved = false;
   return HIPSDP_OK;
}

void hs_gen_set_density(double density);

extern "C" int hipsdp_gen_planted_density(hipsdp_solver* s, int n, int m, long long seed, double density, const double* Xstar,
   const double* Zstar, const double* ystar, double* b_out)
{
   hs_gen_set_density(density);
   const int rc = hipsdp_gen_planted(s, n, m, seed, Xstar, Zstar, ystar, b_out);
   hs_gen_set_density(1.0);
   return rc;
}

extern "C" int hipsdp_get_block_dense(hipsdp_solver* s, int block, double* A)
{
   if ( s == NULL || !s->shaped || block < 0 || block >= (int) s->blk.size() )
      return HIPSDP_ERR_ARG;
   HS_HIP( hipSetDevice(s->device) );
   HS_CALL( stage_sync(s) );
   Block& B = s->blk[block];
   if ( B.sparse )
   {
      /* expanded from the triplets (later entries of the same position win, as in the device structure) */
      const size_t n2 = (size_t) B.n * B.n;
      memset(A, 0, (size_t) (s->m + 1) * n2 * sizeof(double));
      HS_HIP( hipMemcpy(A, B.A0, n2 * sizeof(double), hipMemcpyDeviceToHost) );
      const SpHost& h = *B.sph;
      for (size_t e = 0; e < h.var.size(); ++e)
      {
         double* a = A + (size_t) h.var[e] * n2;
         a[(size_t) h.row[e] * B.n + h.col[e]] = h.val[e];
         a[(size_t) h.col[e] * B.n + h.row[e]] = h.val[e];
      }
      return HIPSDP_OK;
   }
   /* matrices sharded by variable: the rows this rank holds and the constant matrix are written, the rest is left alone */
   const size_t n2b = (size_t) B.n * B.n * sizeof(double);
   if ( s->a_r1 > s->a_r0 )
      HS_HIP( hipMemcpy(A + (size_t) s->a_r0 * B.n * B.n, B.Aown, (size_t) (s->a_r1 - s->a_r0) * n2b, hipMemcpyDeviceToHost) );
   if ( B.A0sep != NULL )
      HS_HIP( hipMemcpy(A, B.A0sep, n2b, hipMemcpyDeviceToHost) );
   return HIPSDP_OK;
}

extern "C" int hipsdp_block_device_ptr(hipsdp_solver* s, int block, double** dptr)
{
   if ( s == NULL || !s->shaped || block < 0 || block >= (int) s->blk.size() )
      return HIPSDP_ERR_ARG;
   HS_CALL( stage_sync(s) );
   *dptr = s->blk[block].Aown;
   return HIPSDP_OK;
}

extern "C" int hipsdp_set_start(hipsdp_solver* s, const double* y, const double* const* X, const double* const* Z,
   const double* x, const double* z)
{
   if ( s == NULL || !s->shaped ) return HIPSDP_ERR_ARG;
   HS_HIP( hipSetDevice(s->device) );
   if ( s->m > 0 )
      HS_CALL( stage_upload(s, s->y, y, (size_t) s->m * sizeof(double)) );
   for (size_t k = 0; k < s->blk.size(); ++k)
   {
      const size_t bytes = (size_t) s->blk[k].n * s->blk[k].n * sizeof(double);
      HS_CALL( stage_upload(s, s->blk[k].X, X[k], bytes) );
      HS_CALL( stage_upload(s, s->blk[k].Z, Z[k], bytes) );
   }
   if ( s->q > 0 )
   {
      HS_CALL( stage_upload(s, s->x, x, (size_t) s->q * sizeof(double)) );
      HS_CALL( stage_upload(s, s->z, z, (size_t) s->q * sizeof(double)) );
   }
   s->have_start = true;
   s->s1_sol_host = false;
   return HIPSDP_OK;
}

/* ---- small fused kernels of the iteration ---------------------------------------------------------------------- */

/* flag = 1 when some x[i] <= 0 or z[i] <= 0 (or is not finite) */
__global__ void k_flag_nonpositive(int q, const double* __restrict__ x, const double* __restrict__ z, int* __restrict__ flag)
{
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   if ( i < q && (!(x[i] > 0.0) || !(z[i] > 0.0) || !(x[i] < 1e300) || !(z[i] < 1e300)) )
      atomicExch(flag, 1);
}

/* rp = b * tau - AX[1:] */
__global__ void k_rp(int m, double tau, const double* __restrict__ b, const double* __restrict__ AX, double* __restrict__ rp)
{
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   if ( i < m )
      rp[i] = b[i] * tau - AX[1 + i];
}

/* rhs2 = [g ; b] with g = Mx[0, 1:] */
__global__ void k_rhs2(int m, const double* __restrict__ Mx, const double* __restrict__ b, double* __restrict__ rhs2)
{
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   if ( i < m )
   {
      rhs2[i] = Mx[1 + i];
      rhs2[m + i] = b[i];
   }
}

/* after the two solves: w = rhs2[0:m], ub = rhs2[m:2m];  u2 = ub - w;  wt = [1, -w] */
__global__ void k_after_solve2(int m, const double* __restrict__ rhs2, double* __restrict__ u2, double* __restrict__ wt)
{
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   if ( i == 0 )
      wt[0] = 1.0;
   if ( i < m )
   {
      u2[i] = rhs2[m + i] - rhs2[i];
      wt[1 + i] = -rhs2[i];
   }
}

/* m <= 64 (single-block factor, dinv = inv(L) as 64 x 64, L = the factor as m x m): rhs2 = [g ; b] with g = Mx[0, 1:], both
 * solves M x = r as x = inv(L)^T (inv(L) r) with each triangular solve corrected once by the factor itself (see RB_SOLVE in
 * kernels.hip: the residual of these solves is primal infeasibility of the step), then u2 = ub - w and wt = [1, -w]: k_rhs2 + the
 * triangular solves + k_after_solve2 in one launch */
__global__ void __launch_bounds__(128) k_solve2_small(int m, const double* __restrict__ Mx, const double* __restrict__ b,
   const double* __restrict__ dinv, const double* __restrict__ L, double* __restrict__ rhs2, double* __restrict__ u2, double* __restrict__ wt,
   int subst)
{
   __shared__ double r[2][64], t[2][64], c[2][64];
   __shared__ double sL[64 * 65];
   const int k = threadIdx.x >> 6, i = threadIdx.x & 63;
   r[k][i] = (i < m) ? (k == 0 ? Mx[1 + i] : b[i]) : 0.0;
   if ( subst )
   {
      /* round 6 (HIPSDP_SMALL_SOLVE=subst): substitution with the factor itself in the oracle's order, one right-hand side per wavefront */
      for (int e = threadIdx.x; e < m * m; e += 128)
      {
         const int ii = e / m, jj = e - ii * m;
         if ( jj <= ii )
            sL[ii * 65 + jj] = L[(long long) ii * m + jj];
      }
      __syncthreads();
      const double x = hs_wl_msolve(sL, m, i, r[k][i]);
      __syncthreads();
      r[k][i] = x;
      __syncthreads();
      if ( i < m )
         rhs2[k * m + i] = x;
      if ( k == 0 )
      {
         if ( i == 0 )
            wt[0] = 1.0;
         if ( i < m )
         {
            u2[i] = r[1][i] - r[0][i];
            wt[1 + i] = -r[0][i];
         }
      }
      return;
   }
   __syncthreads();
   /* forward: t = Y r, c = r - L t, t += Y c */
   double acc = 0.0;
   if ( i < m )
      for (int j = 0; j <= i; ++j)
         acc += dinv[i * 64 + j] * r[k][j];
   t[k][i] = acc;
   __syncthreads();
   double res = 0.0;
   if ( i < m )
   {
      for (int j = 0; j <= i; ++j)
         res += L[(long long) i * m + j] * t[k][j];
      res = r[k][i] - res;
   }
   c[k][i] = res;
   __syncthreads();
   double cor = 0.0;
   if ( i < m )
      for (int j = 0; j <= i; ++j)
         cor += dinv[i * 64 + j] * c[k][j];
   __syncthreads();
   t[k][i] = acc + cor;
   __syncthreads();
   /* backward: r = Y^T t, c = t - L^T r, r += Y^T c */
   acc = 0.0;
   if ( i < m )
      for (int j = i; j < m; ++j)
         acc += dinv[j * 64 + i] * t[k][j];
   r[k][i] = acc;
   __syncthreads();
   res = 0.0;
   if ( i < m )
   {
      for (int j = i; j < m; ++j)
         res += L[(long long) j * m + i] * r[k][j];
      res = t[k][i] - res;
   }
   c[k][i] = res;
   __syncthreads();
   cor = 0.0;
   if ( i < m )
      for (int j = i; j < m; ++j)
         cor += dinv[j * 64 + i] * c[k][j];
   __syncthreads();
   r[k][i] = acc + cor;
   __syncthreads();
   if ( i < m )
      rhs2[k * m + i] = r[k][i];
   if ( k == 0 )
   {
      if ( i == 0 )
         wt[0] = 1.0;
      if ( i < m )
      {
         u2[i] = r[1][i] - r[0][i];
         wt[1 + i] = -r[0][i];
      }
   }
}

/* h = AH[1:] - eta * rp */
__global__ void k_h(int m, double eta, const double* __restrict__ AH, const double* __restrict__ rp, double* __restrict__ h)
{
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   if ( i < m )
      h[i] = AH[1 + i] - eta * rp[i];
}

/* dtau = num / den, dy = u1 - u2 * dtau, dyt = [-dtau, dy], dkappa */
__global__ void k_finish_dir(int m, double eta, double rg, double sigmu, double tau, double kappa, double etk,
   const double* __restrict__ u1, const double* __restrict__ u2, double* __restrict__ dy, double* __restrict__ dyt,
   double* __restrict__ sc)
{
   const double den = sc[SC_S0] + kappa / tau + sc[SC_BUB];
   const double num = -eta * rg + (sigmu - tau * kappa - etk) / tau - sc[SC_BH] - eta * sc[SC_WRP] + sc[SC_BU1];
   const double dtau = num / den;
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   if ( i == 0 )
   {
      sc[SC_DTAU] = dtau;
      sc[SC_DKAPPA] = (sigmu - tau * kappa - etk - kappa * dtau) / tau;
      sc[SC_DEN] = den;
      dyt[0] = -dtau;
   }
   if ( i < m )
   {
      const double v = u1[i] - u2[i] * dtau;
      dy[i] = v;
      dyt[1 + i] = v;
   }
}

/* dZ = P1 - dtau P2 + eta Rd with dtau read from the scalar block (P1 is handed over in dZ) */
__global__ void k_dz_combine(long long n2, double* __restrict__ dZ, const double* __restrict__ P2, const double* __restrict__ sc, double eta,
   const double* __restrict__ Rd)
{
   const double dtau = sc[SC_DTAU];
   for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long long) gridDim.x * blockDim.x)
      dZ[i] = (dZ[i] - dtau * P2[i]) + eta * Rd[i];
}

/* out = Rd + tau * A0  (certificate residual A^T y - Z) */
__global__ void k_cert(long long n2, double tau, const double* __restrict__ Rd, const double* __restrict__ A0, double* __restrict__ out)
{
   for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long long) gridDim.x * blockDim.x)
      out[i] = Rd[i] + tau * A0[i];
}

static inline dim3 g1d(long long n) { long long g = (n + 255) / 256; if ( g < 1 ) g = 1; if ( g > 4096 ) g = 4096; return dim3((unsigned) g); }

/* ---- GEMM convenience ------------------------------------------------------------------------------------------ */
/* n x n x n products of the predictor-corrector chain: at n = 500 they are 64 tiles on 256 CUs, so K is cut into slices
 * (slabs in ws, summed in slice order) to occupy the chip; ws = NULL: no split */
static int gemm_on(hipStream_t st, double* ws, long long wslen, int layA, int layB, int M, int N, int K, double alpha,
   const double* A, long long lda, const double* B, long long ldb, double beta, double* C, long long ldc, int flags = 0)
{
   (void) hs_red_batch_flush();          /* inside a held region: behind the records */
   int sk = 1;
   const long long tiles = (long long) ((M + 63) / 64) * ((N + 63) / 64);
   /* few tiles: the 32 x 32 kernel with the K split inside the workgroup takes the product (no slabs, no second launch);
    * HIPSDP_GEMM_SMALL=0 restores the split-K slabs of the 64 x 64 tile kernel */
   if ( !hs_dgemm3_enabled() && ws != NULL && tiles <= 160 && tiles >= 4 && K >= 192 )
   {
      sk = (int) ((384 + tiles - 1) / tiles);
      if ( sk > K / 64 ) sk = K / 64;
      if ( sk > 8 ) sk = 8;
      while ( sk > 1 && (long long) sk * M * N > wslen ) --sk;
      if ( sk < 2 ) sk = 1;
   }
   hs_gemm_args g = {M, N, K, layA, layB, A, lda, 0, B, ldb, 0, C, ldc, 0, alpha, beta, 1, flags, sk, sk > 1 ? ws : NULL};
   return hs_dgemm(st, &g);
}

static int gemm(hipsdp_solver* s, int layA, int layB, int M, int N, int K, double alpha, const double* A, long long lda,
   const double* B, long long ldb, double beta, double* C, long long ldc, int flags = 0)
{
   return gemm_on(s->stream, s->gws1, s->gws_len, layA, layB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, flags);
}

/* fork: stream2 starts after everything queued on stream so far; join: stream continues after stream2 has drained */
static int fork2(hipsdp_solver* s)
{
   if ( !s->use2 )
      return HS_OK;
   HS_HIP( hipEventRecord(s->evFork, s->stream) );
   HS_HIP( hipStreamWaitEvent(s->stream2, s->evFork, 0) );
   return HS_OK;
}

static int join2(hipsdp_solver* s)
{
   if ( !s->use2 )
      return HS_OK;
   HS_HIP( hipEventRecord(s->evJoin, s->stream2) );
   HS_HIP( hipStreamWaitEvent(s->stream, s->evJoin, 0) );
   return HS_OK;
}

/* ---- Schur workspace ------------------------------------------------------------------------------------------------ */
static int ensure_schur_ws(hipsdp_solver* s)
{
   if ( s->sws.T != NULL )
      return HS_OK;
   {
      /* every block kept as nonzeros: the assembly needs no GEMM workspace (the split-K slabs alone would be 64 (m + 1)^2 doubles) */
      bool anydense = false;
      for (auto& B : s->blk) anydense = anydense || !B.sparse;
      if ( !anydense && !s->blk.empty() )
         return HS_OK;
   }
   const int m1 = s->m + 1;
   long long n2max = 1;
   for (auto& B : s->blk) { const long long n2 = (long long) B.n * B.n; if ( !B.sparse && n2 > n2max ) n2max = n2; }
   double budget = s->par.ws_gbytes > 0.0 ? s->par.ws_gbytes : 40.0;
   const char* env = getenv("HIPSDP_WS_GB");
   if ( env != NULL && atof(env) > 0.0 )
      budget = atof(env);
   else if ( s->par.ws_gbytes <= 0.0 && s->comm == NULL )
   {
      /* one rank, nothing prescribed: three quarters of what is free now (A and its packed copy are allocated), at least the
       * default - wider slices and fewer of them when the whole T, W pair does not fit (several ranks keep the fixed default:
       * every rank must derive the same slicing) */
      size_t fr = 0, tot = 0;
      if ( hipMemGetInfo(&fr, &tot) == hipSuccess && 0.75e-9 * (double) fr > budget )
         budget = 0.75e-9 * (double) fr;
   }
   const char* mode = getenv("HIPSDP_SCHUR");
   s->schur_mode_cols = false;
   s->schur_sim_shards = 0;
   if ( s->shardA )
   {
      /* Matrices sharded by variable: the W_j are formed where A_j lives, in S column slices of at most cw columns; the three
       * buffers of a slice (T and W of the own variables, the received row range of all W_j) take about 3 (m1 / ranks) n cw
       * doubles.  The widest slice that fits the budget (a multiple of 128, or all n columns); same numbers on every rank. */
      int nmaxb = 1;
      for (auto& B : s->blk) if ( B.n > nmaxb ) nmaxb = B.n;
      /* overlapped exchange (default; HIPSDP_VAR_OVERLAP=0: slice after slice): a second send and a second receive buffer */
      static const bool var_overlap = !(getenv("HIPSDP_VAR_OVERLAP") != NULL && getenv("HIPSDP_VAR_OVERLAP")[0] == '0');
      const double per_col = (var_overlap ? 5.0 : 3.0) * 8.0 * (double) ((m1 + s->nranks - 1) / s->nranks + 1) * (double) nmaxb;
      long long cw = (long long) (budget * 1e9 / per_col);
      const char* ecw = getenv("HIPSDP_VAR_SLICE");          /* test hook: force narrow slices */
      if ( ecw != NULL && atoi(ecw) > 0 )
         cw = atoi(ecw);
      else if ( cw < nmaxb )
         cw = cw / 128 * 128 > 0 ? cw / 128 * 128 : 128;
      if ( cw > nmaxb ) cw = nmaxb;
      s->var_cw = (int) cw;
      HS_CALL( hs_schur_ws_alloc_var(&s->sws, m1, s->nranks, nmaxb, (int) cw) );
      s->var_overlap = false;
      if ( var_overlap && cw < nmaxb )                /* (one slice: nothing to overlap with) */
         s->var_overlap = (hs_schur_ws_alloc_var_overlap(&s->sws) == HS_OK);
      s->schur_mode_U = false; s->schur_mode_rows = false; s->schur_mode_forced = true;
      return HS_OK;
   }
   /* Column slices of the W formulation: S slices in total, every rank works through S / ranks of them one after the other
    * (one rank: all of them).  S = ranks when the workspace for one slice fits the budget, otherwise the smallest multiple
    * that does - this is also how one GPU keeps the cheaper W formulation when T and W do not fit as a whole (n = 2000,
    * m = 4000: 2 x 128 GB).  The workspace holds the widest slice; every rank derives the same numbers, so all ranks take
    * the same branch (the two sharded forms use different collectives). */
   const int ranks = s->comm != NULL ? s->nranks : 1;
   auto widest = [&](int S) -> long long {
      long long w = 1;
      for (auto& B : s->blk)
         for (int g = 0; g < S; ++g)
         {
            int c0, cw;
            hs_shard_cols(m1, B.n, S, g, &c0, &cw);
            if ( (long long) B.n * cw > w ) w = (long long) B.n * cw;
         }
      return w;
   };
   int S = 0;
   if ( s->comm == NULL && mode != NULL && mode[0] == 'K' && atoi(mode + 1) >= 1 && atoi(mode + 1) <= 64 )
      S = atoi(mode + 1);
   else if ( !(mode != NULL && (mode[0] == 'R' || mode[0] == 'U')) && (mode == NULL || s->comm != NULL) )
   {
      const bool whole_fits = 2.0 * 8.0 * (double) m1 * (double) n2max <= budget * 1e9;
      if ( s->comm != NULL || !whole_fits )
         for (int k = 1; k * ranks <= 64; ++k)
            if ( (s->comm != NULL || k > 1) && 2.0 * 8.0 * (double) m1 * (double) widest(k * ranks) <= budget * 1e9 )
            {
               S = k * ranks;
               break;
            }
   }
   if ( S > 0 && 2.0 * 8.0 * (double) m1 * (double) widest(S) <= budget * 1e9 )
   {
      HS_CALL( hs_schur_ws_alloc(&s->sws, m1, widest(S), budget) );
      s->schur_mode_cols = true;
      s->schur_sim_shards = S;
   }
   if ( !s->schur_mode_cols )
      HS_CALL( hs_schur_ws_alloc(&s->sws, m1, n2max, budget) );
   s->schur_mode_U = (mode != NULL && mode[0] == 'U') || !s->sws.full || s->comm != NULL;
   s->schur_mode_rows = (mode != NULL && mode[0] == 'R');
   s->schur_mode_forced = (mode != NULL);
   if ( s->comm != NULL && s->Mgather == NULL )
   {
      const long long c = (m1 + 2 * s->nranks - 1) / (2 * s->nranks);
      HS_CALL( dalloc(&s->Mgather, c * m1 * s->nranks) );
   }
   return HS_OK;
}

/* ---- the solve ---------------------------------------------------------------------------------------------------- */

struct HostScalars
{
   std::vector<double> v;
};

int hs_bcast_doubles(void* comm, double* buf, long long count, hipStream_t stream);
int hs_bcast_ints(void* comm, int* buf, long long count, hipStream_t stream);

/* Read-back without a copy engine and without a stream synchronisation: a kernel stores n doubles of sc to the host mirror
 * and then a sequence number; the host polls the number (coherent pinned memory).  The stream is queried now and then so
 * that a failed launch cannot leave the host spinning. */
static int publish_and_wait(hipsdp_solver* s, int off, int n, const std::function<int()>* between = NULL)
{
   const unsigned long long seq = ++s->pub_seq;
   volatile unsigned long long* flag = reinterpret_cast<volatile unsigned long long*>(s->hsc + s->hsc_cap - 1);
   HS_CALL( hs_red_batch_end_publish(s->stream, n, s->sc + off, s->hsc_dev + off, seq,
         reinterpret_cast<unsigned long long*>(s->hsc_dev + s->hsc_cap - 1)) );
   /* work that does not depend on what the host is about to read goes into the queue now: the device runs it while the scalars
    * travel, the host decides and the next launches arrive */
   if ( between != NULL )
      HS_CALL( (*between)() );
   /* developer switch HIPSDP_WAIT_TIMES=1: how long the host waits here in total (at process end): the share of a solve in which the
    * device is the one being waited for */
   static int wt = -1;
   if ( wt < 0 )
      wt = (getenv("HIPSDP_WAIT_TIMES") != NULL && getenv("HIPSDP_WAIT_TIMES")[0] == '1') ? 1 : 0;
   const std::chrono::steady_clock::time_point w0 = wt ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
   struct WaitNote
   {
      bool on; std::chrono::steady_clock::time_point t0;
      ~WaitNote()
      {
         if ( !on ) return;
         static double total = 0.0; static long long count = 0; static bool hooked = false;
         static double* ptot = &total; static long long* pcnt = &count;
         total += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
         ++count;
         if ( !hooked )
         {
            hooked = true;
            atexit([]() { fprintf(stderr, "hipsdp: %lld read-backs, the host waited %.3f s in them (%.2f us each)\n", *pcnt, *ptot, 1e6 * *ptot / (double) (*pcnt > 0 ? *pcnt : 1)); });
         }
      }
   } note = {wt != 0, w0};
   long long spins = 0;
   while ( *flag != seq )
   {
      if ( (++spins & 0x3FFF) == 0 )
      {
         const hipError_t e = hipStreamQuery(s->stream);
         if ( e == hipSuccess )
         {
            if ( *flag == seq )
               break;
            HS_HIP( hipStreamSynchronize(s->stream) );
            if ( *flag != seq )
            {
               set_err("read-back kernel finished without publishing its results");
               return HS_ERR_HIP;
            }
            break;
         }
         if ( e != hipErrorNotReady )
         {
            hs_record_hip_error(e, "hipStreamQuery(read-back)", __FILE__, __LINE__);
            return HS_ERR_HIP;
         }
      }
   }
   __atomic_thread_fence(__ATOMIC_ACQUIRE);
   return HS_OK;
}

static int read_scalars(hipsdp_solver* s, HostScalars& h, int* flags3, const std::function<int()>* between = NULL)
{
   if ( s->comm != NULL || !s->use_publish )
      HS_CALL( hs_red_batch_end_all() );  /* recorded operations must run before the scalars are read */
   h.v.resize(s->nsc);
   if ( s->comm != NULL )
   {
      /* all ranks computed the same numbers; broadcasting rank 0's copy makes the control flow identical by construction */
      /* the flags live right behind the scalars: one broadcast covers both */
      HS_CALL( hs_bcast_doubles(s->comm, s->sc, s->nsc + (flags3 != NULL ? 2 : 0), s->stream) );
   }
   if ( s->comm == NULL && s->use_publish )
   {
      /* the batch kernel (or a one-block kernel when the batch is empty) stores the scalars to the host mirror itself */
      HS_CALL( publish_and_wait(s, 0, s->nsc + 4, between) );
   }
   else
   {
      HS_HIP( hipMemcpyAsync(s->hsc, s->sc, (size_t) (s->nsc + (flags3 != NULL ? 4 : 0)) * sizeof(double), hipMemcpyDeviceToHost, s->stream) );
      if ( between != NULL )
         HS_CALL( (*between)() );
      HS_HIP( hipStreamSynchronize(s->stream) );
   }
   memcpy(h.v.data(), s->hsc, (size_t) s->nsc * sizeof(double));
   if ( flags3 != NULL )
      memcpy(flags3, s->hsc + s->nsc, 3 * sizeof(int));
   return HS_OK;
}

/* the two passes over the constraint matrices of one block; with a packed copy they move half the bytes */
static int ensure_sparse(hipsdp_solver* s);
static int ensure_packed(hipsdp_solver* s)
{
   HS_CALL( ensure_sparse(s) );
   for (auto& B : s->blk)
   {
      if ( B.Apk != NULL && !B.apk_valid )
      {
         if ( s->a_r1 > s->a_r0 )
            HS_CALL( hs_pack_rows(s->stream, s->a_r1 - s->a_r0, B.n, B.Lp, B.Aown, B.Apkown) );
         B.apk_valid = true;
      }
   }
   return HS_OK;
}

int hs_allreduce_sum(void* comm, double* buf, long long count, hipStream_t stream);

/* Several ranks: a pass over the constraint matrices of a block is HBM-bound and every rank holds all of A, so each rank
 * sweeps only its ceil(m1 / ranks) matrices: A(V) is completed by an all-gather of m1 numbers, A^T(coef) by an all-reduce of
 * the (packed) n x n partial sums.  Every rank decides from the sizes alone (same decision everywhere). */
static bool passes_sharded(const hipsdp_solver* s, const Block& B)
{
   if ( s->comm == NULL || B.sparse )
      return false;
   if ( s->shardA )
      return true;              /* the only rows there are (hs_var_rows is the row split of the sharded passes) */
   if ( s->nranks < 2 )
      return s->shard_passes == 1;        /* a communicator of one rank: only when forced (exercises the collectives) */
   if ( s->shard_passes >= 0 )
      return s->shard_passes != 0;
   return 8.0 * (double) (s->m + 1) * (double) B.n * (double) B.n >= 64e6;
}

static void pass_rows(const hipsdp_solver* s, int* chunk, int* r0, int* r1)
{
   const int m1 = s->m + 1;
   const int c = (m1 + s->nranks - 1) / s->nranks;
   *chunk = c;
   *r0 = s->rank * c < m1 ? s->rank * c : m1;
   *r1 = *r0 + c < m1 ? *r0 + c : m1;
}

/* sparse blocks: the triplets collected since the last build become the device structure */
static int ensure_sparse(hipsdp_solver* s)
{
   for (auto& B : s->blk)
      if ( B.sparse && (B.sp_dirty || B.sp == NULL) )
      {
         if ( B.sp != NULL ) hs_sp_free(B.sp);
         B.sp = NULL;
         HS_HIP( hipStreamSynchronize(s->stream) );
         const SpHost& h = *B.sph;
         HS_CALL( hs_sp_build(&B.sp, B.n, s->m, (long long) h.var.size(), h.var.data(), h.row.data(), h.col.data(), h.val.data()) );
         B.sp_dirty = false;
      }
   return HS_OK;
}

/* out = c[0] A0 + sa add (add may be NULL): the dense part of A^T(coef) of a sparse block */
__global__ void k_sp_base(long long n2, const double* __restrict__ c, const double* __restrict__ A0, double sa, const double* __restrict__ add,
   double* __restrict__ out)
{
   const double c0 = c[0];
   for (long long e = (long long) blockIdx.x * blockDim.x + threadIdx.x; e < n2; e += (long long) gridDim.x * blockDim.x)
      out[e] = c0 * A0[e] + (add != NULL ? sa * add[e] : 0.0);
}

/* out[0] = <a, b>: one workgroup, fixed summation order */
__global__ void __launch_bounds__(1024) k_sp_dot0(long long n, const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ out)
{
   __shared__ double part[16];
   double acc = 0.0;
   for (long long e = threadIdx.x; e < n; e += 1024)
      acc += a[e] * b[e];
#pragma unroll
   for (int off = 32; off > 0; off >>= 1)
      acc += __shfl_xor(acc, off, 64);
   if ( (threadIdx.x & 63) == 0 )
      part[threadIdx.x >> 6] = acc;
   __syncthreads();
   if ( threadIdx.x == 0 )
   {
      double t = 0.0;
      for (int w = 0; w < 16; ++w)
         t += part[w];
      out[0] = t;
   }
}

/* Mx[i][0] += v[i] (column 0 of the lower triangle) */
__global__ void k_add_col0(int m1, const double* __restrict__ v, double* __restrict__ Mx)
{
   for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m1; i += gridDim.x * blockDim.x)
      Mx[(long long) i * m1] += v[i];
}

/* out[m + 1] = <A_i, V>, V symmetric */
static int pass_A(hipsdp_solver* s, Block& B, const double* V, double* out)
{
   const int m1 = s->m + 1;
   if ( B.sparse )
   {
      /* launched at once like the dense pass (hs_dot would be deferred inside a reduction batch, and callers add the result up
       * with kernels that are not) */
      /* <A_0, V> (A_0 is dense): the row-dot kernel of the dense pass with one row - split over workgroups and combined in order
       * (a single workgroup took 73 us for the 250 000 entries of a block of 500 rows) */
      HS_CALL( hs_gemv_n(s->stream, 1, (long long) B.n * B.n, B.A0, (long long) B.n * B.n, 1, &V, out, 1, s->gemv_ws, s->gemv_ws_len) );
      return hs_sp_apply_A(s->stream, B.sp, V, out + 1);
   }
   if ( passes_sharded(s, B) )
   {
      int c, r0, r1;
      pass_rows(s, &c, &r0, &r1);
      if ( s->passg == NULL )
         HS_CALL( dalloc(&s->passg, (long long) c * s->nranks) );
      double* mine = s->passg + (long long) s->rank * c;
      if ( B.Apk != NULL )
      {
         HS_CALL( hs_pack_weighted(s->stream, B.n, V, B.pkv) );
         const double* v = B.pkv;
         if ( r1 > r0 )
            HS_CALL( hs_gemv_n(s->stream, r1 - r0, B.Lp, B.Apk + (long long) r0 * B.Lp, B.Lp, 1, &v, mine, r1 - r0, s->gemv_ws, s->gemv_ws_len) );
      }
      else if ( r1 > r0 )
      {
         const long long n2 = (long long) B.n * B.n;
         HS_CALL( hs_gemv_n(s->stream, r1 - r0, n2, B.A + (long long) r0 * n2, n2, 1, &V, mine, r1 - r0, s->gemv_ws, s->gemv_ws_len) );
      }
      hs_comm_phase(1);
      const int rcg = hs_allgather_inplace(s->comm, s->passg, c, s->rank, s->stream);
      hs_comm_phase(2);
      HS_CALL( rcg );
      return hs_copy(s->stream, out, s->passg, m1);
   }
   if ( B.Apk != NULL )
   {
      HS_CALL( hs_pack_weighted(s->stream, B.n, V, B.pkv) );
      const double* v = B.pkv;
      return hs_gemv_n(s->stream, m1, B.Lp, B.Apk, B.Lp, 1, &v, out, m1, s->gemv_ws, s->gemv_ws_len);
   }
   return hs_gemv_n(s->stream, m1, (long long) B.n * B.n, B.A, (long long) B.n * B.n, 1, &V, out, m1, s->gemv_ws, s->gemv_ws_len);
}

/* out[n x n] = sum_i coef_i A_i + sa * add */
static int pass_AT(hipsdp_solver* s, Block& B, const double* coef, double sa, const double* add, double* out)
{
   const int m1 = s->m + 1;
   const long long n2 = (long long) B.n * B.n;
   if ( B.sparse )
   {
      hipLaunchKernelGGL(k_sp_base, g1d(n2), dim3(256), 0, s->stream, n2, coef, B.A0, sa, add, out);
      HS_LAUNCH_CHECK();
      return hs_sp_apply_AT(s->stream, B.sp, coef, out);
   }
   if ( passes_sharded(s, B) )
   {
      int c, r0, r1;
      pass_rows(s, &c, &r0, &r1);
      if ( B.Apk != NULL )
      {
         HS_CALL( hs_gemv_t(s->stream, r1 - r0, B.Lp, B.Apk + (long long) r0 * B.Lp, B.Lp, coef + r0, 0.0, NULL, B.pkv + B.Lp) );
         hs_comm_phase(1);
         const int rcr = hs_allreduce_sum(s->comm, B.pkv + B.Lp, B.Lp, s->stream);
         hs_comm_phase(2);
         HS_CALL( rcr );
         return hs_unpack_sym(s->stream, B.n, B.pkv + B.Lp, sa, add, out);
      }
      /* the additive term enters once: on rank 0 */
      HS_CALL( hs_gemv_t(s->stream, r1 - r0, n2, B.A + (long long) r0 * n2, n2, coef + r0, s->rank == 0 ? sa : 0.0,
            s->rank == 0 ? add : NULL, out) );
      hs_comm_phase(1);
      const int rcr = hs_allreduce_sum(s->comm, out, n2, s->stream);
      hs_comm_phase(2);
      return rcr;
   }
   if ( B.Apk != NULL )
   {
      HS_CALL( hs_gemv_t_ws(s->stream, m1, B.Lp, B.Apk, B.Lp, coef, 0.0, NULL, B.pkv + B.Lp, s->gemvt_ws, s->gemvt_ws_len) );
      return hs_unpack_sym(s->stream, B.n, B.pkv + B.Lp, sa, add, out);
   }
   return hs_gemv_t_ws(s->stream, m1, n2, B.A, n2, coef, sa, add, out, s->gemvt_ws, s->gemvt_ws_len);
}

#define AS_MAXBLK HS_AS_MAXBLK

static bool small_problem(const hipsdp_solver* s)
{
   if ( s->blk.size() > AS_MAXBLK || s->q > 4096 || s->m + 1 > 4096 || s->shardA )
      return false;
   long long tot = 0;
   for (auto& B : s->blk)
   {
      /* a block kept as nonzeros (csrc/sparse.hip) has no dense rows A_i: the fused small-problem kernels read A + i n^2 */
      if ( B.Apk != NULL || B.sparse || B.A == NULL )
         return false;
      tot += (long long) B.n * B.n;
   }
   return tot <= 8192;
}

/* B&B-sized problems: runs of small kernels are recorded and executed by one launch (kernels.hip: hs_red_batch_hold).  A region
 * covers a whole direction, or the step of the iterate with the residual pass behind it; HIPSDP_BATCH=0 switches the regions off
 * (every kernel its own launch, as before round 3; same arithmetic either way). */
static bool batch_regions(const hipsdp_solver* s)
{
   static int on = -1;
   if ( on < 0 )
   {
      const char* env = getenv("HIPSDP_BATCH");
      on = (env != NULL && env[0] == '0') ? 0 : 1;
   }
   return on != 0 && s->comm == NULL && small_problem(s);
}

struct BatchRegion
{
   bool on;
   BatchRegion(hipsdp_solver* s, bool want) : on(want) { if ( on ) hs_red_batch_hold(s->stream); }
   int close() { if ( !on ) return HS_OK; on = false; return hs_red_batch_release(); }
   ~BatchRegion() { if ( on ) (void) hs_red_batch_release(); }
};

/* returns 1 when the fused launch was used */
static int apply_A_small(hipsdp_solver* s, double* const* Vk, const double* vlp, double* out, int epi, double scal,
   const double* vin, double* vout)
{
   if ( !small_problem(s) )
      return 0;
   for (auto& Bk : s->blk)
      if ( Bk.A == NULL )
         return 0;
   hs_as_args B;
   B.nblk = (int) s->blk.size();
   for (int k = 0; k < AS_MAXBLK; ++k)
   {
      B.n2[k] = 0; B.A[k] = NULL; B.V[k] = NULL;
   }
   for (size_t k = 0; k < s->blk.size(); ++k)
   {
      B.n2[k] = s->blk[k].n * s->blk[k].n;
      B.A[k] = s->blk[k].A;
      B.V[k] = Vk[k];
   }
   const int rc = hs_apply_A_small(s->stream, s->m + 1, &B, s->q, s->Dext, vlp, out, epi, scal, vin, vout);
   if ( rc != HS_OK )
      return -rc;
   return 1;
}

/* A(V) over all blocks + LP: out[m + 1] = sum_k A_k vec(V_k) + Dext^T vlp */
static int apply_A(hipsdp_solver* s, double* const* Vk, const double* vlp, double* out)
{
   const int m1 = s->m + 1;
   bool first = true;
   for (size_t k = 0; k < s->blk.size(); ++k)
   {
      Block& B = s->blk[k];
      const double* v = Vk[k];
      double* dst = first ? out : s->tmpe;
      HS_CALL( pass_A(s, B, v, dst) );
      if ( !first )
         HS_CALL( hs_axpy(s->stream, m1, 1.0, s->tmpe, out) );
      first = false;
   }
   if ( first )
      HS_CALL( hs_fill(s->stream, out, m1, 0.0) );
   if ( s->q > 0 )
      HS_CALL( hs_gemv_t(s->stream, s->q, m1, s->Dext, m1, vlp, 1.0, out, out) );
   return HS_OK;
}

/* the engine's two queues trade places for a while: everything enqueued through s->stream (and its GEMM workspace) goes to the
 * second queue */
struct QueueSwap
{
   hipsdp_solver* s;
   explicit QueueSwap(hipsdp_solver* s_) : s(s_) { std::swap(s->stream, s->stream2); std::swap(s->gws1, s->gws2); }
   ~QueueSwap() { std::swap(s->stream, s->stream2); std::swap(s->gws1, s->gws2); }
};

/* one Newton direction; results in (dy, dyt, SC_DTAU, SC_DKAPPA), B.dX, B.dZ, s->dx, s->dz.
 * part 0: all of it.  part 1: only the right-hand side H_k, hl, A(H) and h = A(H) - eta rp (general path; it needs X, Z^-1 and the
 * residuals but not the Schur matrix, so the predictor's can run on the second queue beside the factorization of M);
 * part 2: the rest, after a part-1 call with the same arguments. */
/* split: 0 = dZ by one pass with the coefficients [-dtau; dy]; 1 = dZ = P1 - dtau P2 + eta Rd with P1 = A^T([0; u1]) computed here
 * by one pass and P2 = A^T([1; u2]) from the three-vector sweep of this iteration; 2 = P1 has been computed by that sweep as well
 * (the predictor: its u1 was solved together with the right-hand sides of the tau elimination) */
/* mode of the triangular solves with the factor of M: forward + backward, corrected where that is free or asked for */
static inline int solve_mode(const hipsdp_solver* s)
{
   return 3 | (s->refine_solves ? 4 : 0);
}

static int direction(hipsdp_solver* s, double sigma, double eta, double mu, double rg, bool useE, double etk, int part = 0,
   bool u1_solved = false, int split = 0)
{
   const int m = s->m, m1 = s->m + 1, q = s->q;
   const double sigmu = sigma * mu;
   std::vector<double*> Hs;
   int fusedA = 0;
   BatchRegion region(s, part == 0 && batch_regions(s));
   if ( part != 2 )
   {
   for (auto& B : s->blk)
   {
      const int n = B.n;
      const long long n2 = (long long) n * n;
      if ( n <= HS_SMALL_N )
         HS_CALL( hs_dir_block_small(s->stream, n, eta, B.X, B.Rd, useE ? B.E : NULL, B.Zinv, sigmu, B.H) );
      else
      {
         if ( useE )
            HS_CALL( hs_copy(s->stream, B.G, B.E, n2) );
         HS_CALL( gemm(s, HS_KC, HS_MC, n, n, n, eta, B.X, n, B.Rd, n, useE ? 1.0 : 0.0, B.G, n) );
         HS_CALL( gemm(s, HS_KC, HS_MC, n, n, n, 1.0, B.G, n, B.Zinv, n, 0.0, B.GZ, n) );
         HS_CALL( hs_dirmat(s->stream, n, sigmu, B.Zinv, B.X, B.GZ, B.H) );
      }
      Hs.push_back(B.H);
   }
   if ( q > 0 )
      HS_CALL( hs_lp_dir(s->stream, q, sigmu, eta, s->x, s->z, s->rd, useE ? s->elp : NULL, s->hl) );
   fusedA = part == 1 ? 0 : apply_A_small(s, Hs.data(), s->hl, s->AH, 1, eta, s->rp, s->u1);
   if ( fusedA < 0 )
      return -fusedA;
   if ( fusedA == 0 )
      HS_CALL( apply_A(s, Hs.data(), s->hl, s->AH) );
   if ( fusedA == 0 && m > 0 )
   {
      HS_CALL( hs_red_batch_flush() );
      hipLaunchKernelGGL(k_h, g1d(m), dim3(256), 0, s->stream, m, eta, s->AH, s->rp, s->u1);
      HS_LAUNCH_CHECK();
   }
   if ( part == 1 )
      return HS_OK;
   }
   const bool fuse_solve = (fusedA == 1) && m > 0 && m <= 64;      /* single-block factor of M: solve, reductions and closing kernel in one launch */
   if ( m > 0 && !fuse_solve && !u1_solved )
   {
      HS_CALL( hs_red_batch_flush() );
      HS_CALL( hs_trsv_sync(s->stream, m, s->Lm, s->dinvm, 1, s->u1, m, solve_mode(s), s->trsv_ws, &s->trsv_epoch) );
   }
   /* BH = sum <B_k, H_k> + beta^T hl ; wrp ; bu1 */
   hs_red_batch_begin(s->stream);
   if ( fuse_solve )
      (void) hs_red_batch_solve(s->stream, m, s->dinvm, s->Lm, 1, s->u1, m);
   HS_CALL( hs_fill_scalar(s->stream, s->sc + SC_BH, 0.0) );
   for (auto& B : s->blk)
      HS_CALL( hs_dot(s->stream, (long long) B.n * B.n, B.B, B.H, s->sc + SC_BH, 1, s->red_ws) );
   if ( q > 0 )
      HS_CALL( hs_dot(s->stream, q, s->beta, s->hl, s->sc + SC_BH, 1, s->red_ws) );
   HS_CALL( hs_dot(s->stream, m, s->rhs2, s->rp, s->sc + SC_WRP, 0, s->red_ws) );       /* w = rhs2[0:m] */
   HS_CALL( hs_dot(s->stream, m, s->b, s->u1, s->sc + SC_BU1, 0, s->red_ws) );
   bool finished = false;
   if ( fuse_solve )
   {
      hs_rb_finish F = {m, eta, rg, sigmu, s->tau, s->kappa, etk, s->u1, s->u2, s->dy, s->dyt, s->sc,
         SC_S0, SC_BUB, SC_BH, SC_WRP, SC_BU1, SC_DTAU, SC_DKAPPA, SC_DEN};
      finished = hs_red_batch_finish(s->stream, &F, sizeof(F)) == 1;
   }
   HS_CALL( hs_red_batch_end() );
   if ( !finished )
   {
      HS_CALL( hs_red_batch_flush() );
      hipLaunchKernelGGL(k_finish_dir, g1d(m > 0 ? m : 1), dim3(256), 0, s->stream, m, eta, rg, sigmu, s->tau, s->kappa, etk,
         s->u1, s->u2, s->dy, s->dyt, s->sc);
      HS_LAUNCH_CHECK();
   }
   for (auto& B : s->blk)
   {
      const int n = B.n;
      const long long n2 = (long long) n * n;
      if ( split == 0 )
         HS_CALL( pass_AT(s, B, s->dyt, eta, B.Rd, B.dZ) );
      else
      {
         if ( split == 1 )
         {
            if ( &B == &s->blk[0] )
            {
               HS_CALL( hs_make_ext(s->stream, m, 0.0, 1.0, s->u1, s->cvec + m1) );
               HS_LAUNCH_CHECK();
            }
            HS_CALL( pass_AT(s, B, s->cvec + m1, 0.0, NULL, B.dZ) );
         }
         HS_CALL( hs_red_batch_flush() );
         hipLaunchKernelGGL(k_dz_combine, g1d(n2), dim3(256), 0, s->stream, n2, B.dZ, B.P2, s->sc, eta, B.Rd);
         HS_LAUNCH_CHECK();
      }
      if ( n <= HS_SMALL_N )
         HS_CALL( hs_dir_block_small(s->stream, n, 1.0, B.X, B.dZ, useE ? B.E : NULL, B.Zinv, sigmu, B.dX) );
      else
      {
         if ( useE )
            HS_CALL( hs_copy(s->stream, B.G, B.E, n2) );
         HS_CALL( gemm(s, HS_KC, HS_MC, n, n, n, 1.0, B.X, n, B.dZ, n, useE ? 1.0 : 0.0, B.G, n) );
         HS_CALL( gemm(s, HS_KC, HS_MC, n, n, n, 1.0, B.G, n, B.Zinv, n, 0.0, B.GZ, n) );
         HS_CALL( hs_dirmat(s->stream, n, sigmu, B.Zinv, B.X, B.GZ, B.dX) );
      }
   }
   if ( q > 0 )
   {
      if ( small_problem(s) )
      {
         HS_CALL( hs_lp_rows_small(s->stream, q, m1, s->Dext, s->dyt, 1, eta, sigmu, s->x, s->z,
            s->rd, useE ? s->elp : NULL, s->dz, s->dx) );
      }
      else
      {
         const double* v = s->dyt;
         HS_CALL( hs_gemv_n(s->stream, q, m1, s->Dext, m1, 1, &v, s->tmpq, q, s->gemv_ws, s->gemv_ws_len) );
         HS_CALL( hs_scale_add(s->stream, q, 1.0, s->tmpq, eta, s->rd, s->dz) );
         HS_CALL( hs_lp_dir(s->stream, q, sigmu, 1.0, s->x, s->z, s->dz, useE ? s->elp : NULL, s->dx) );
      }
   }
   return region.close();
}

/* enqueue the step-length estimates for the current direction */
static int steplen_enqueue(hipsdp_solver* s)
{
   int k = 0;
   hipStream_t st = s->stream, st2 = s->use2 ? s->stream2 : s->stream;
   HS_CALL( fork2(s) );
   for (auto& B : s->blk)
   {
      const int n = B.n;
      if ( n <= 64 )
         continue;                   /* products and eigenvalues of small blocks share one launch below */
      HS_CALL( gemm_on(st, s->gws1, s->gws_len, HS_KC, HS_MC, n, n, n, 1.0, B.LxInv, n, B.dX, n, 0.0, B.T1, n) );
      HS_CALL( gemm_on(st, s->gws1, s->gws_len, HS_KC, HS_KC, n, n, n, 1.0, B.T1, n, B.LxInv, n, 0.0, B.W, n) );
      HS_CALL( gemm_on(st2, s->gws2, s->gws_len, HS_KC, HS_MC, n, n, n, 1.0, B.LzInv, n, B.dZ, n, 0.0, B.T2, n) );
      HS_CALL( gemm_on(st2, s->gws2, s->gws_len, HS_KC, HS_KC, n, n, n, 1.0, B.T2, n, B.LzInv, n, 0.0, B.W2, n) );
   }
   HS_CALL( join2(s) );
   /* the X-side and the Z-side eigenvalue of a block run in the same launch */
   int nmax_blk = 0;
   for (auto& B : s->blk)
      if ( B.n > nmax_blk ) nmax_blk = B.n;
   /* blocks of at most 64 rows: ONE launch per size class (<= 16: exact, 17 .. 64: Lanczos in LDS) for all of them together - with
    * twenty blocks of fifty rows the forty two-workgroup launches of an iteration were 39 % of it (round 3) */
   for (int cls = 0; cls < 2; ++cls)
   {
      hs_step_jobs J;
      J.nblk = 0;
      int kk = 0;
      for (auto& B : s->blk)
      {
         const bool mine = (cls == 0) ? (B.n <= 16) : (B.n > 16 && B.n <= 64);
         if ( mine )
         {
            const int j = J.nblk++;
            J.n[j] = B.n; J.L0[j] = B.LxInv; J.D0[j] = B.dX; J.L1[j] = B.LzInv; J.D1[j] = B.dZ;
            J.res0[j] = s->sc + SC_BLK(kk, 1); J.res1[j] = s->sc + SC_BLK(kk, 4);
            if ( J.nblk == HS_STEP_MAXJOBS )
            {
               HS_CALL( hs_steplen_small_multi(st, &J, s->par.lanczos_steps) );
               J.nblk = 0;
            }
         }
         ++kk;
      }
      HS_CALL( hs_steplen_small_multi(st, &J, s->par.lanczos_steps) );
   }
   for (auto& B : s->blk)
   {
      if ( B.n > 64 )
         HS_CALL( hs_lanczos_lmin2(st, B.n, B.W, B.W2, s->par.lanczos_steps, s->sc + SC_BLK(k, 1), s->sc + SC_BLK(k, 4), s->lan_ws,
               s->lan_ws2, s->lan_rot, s->lan_sync, nmax_blk) );
      ++k;
   }
   hs_red_batch_begin(s->stream);
   HS_CALL( hs_ratio_min(s->stream, s->q, s->x, s->dx, s->sc + SC_RATX, 0, s->red_ws) );
   HS_CALL( hs_ratio_min(s->stream, s->q, s->z, s->dz, s->sc + SC_RATZ, 0, s->red_ws) );
   /* the batch stays open: both callers read the scalars next, and the read-back's launch executes the records and stores
    * the results to the host in one go (read_scalars closes the batch either way) */
   return HS_OK;
}

static double steplen_host(hipsdp_solver* s, const HostScalars& h)
{
   double a = 1e300;
   for (size_t k = 0; k < s->blk.size(); ++k)
   {
      for (int which = 0; which < 2; ++which)
      {
         const double theta = h.v[SC_BLK(k, which ? 4 : 1)];
         const double resid = h.v[SC_BLK(k, which ? 5 : 2)];
         const double lo = theta - resid;          /* pessimistic smallest eigenvalue */
         if ( lo < 0.0 )
            a = fmin(a, -1.0 / lo);
      }
   }
   a = fmin(a, fmin(h.v[SC_RATX], h.v[SC_RATZ]));
   const double dtau = h.v[SC_DTAU], dkap = h.v[SC_DKAPPA];
   if ( dtau < 0.0 ) a = fmin(a, -s->tau / dtau);
   if ( dkap < 0.0 ) a = fmin(a, -s->kappa / dkap);
   return a;
}

/* Several ranks and a small problem: sharding an assembly of less than about 2 * 10^10 flops costs more in collectives than it
 * saves, and the single-launch kernels of the B&B-sized regime only exist for one rank.  Every rank then runs the whole solve on
 * its own (identical deterministic arithmetic) and ONE exchange at the end makes rank 0's outcome the outcome everywhere - which
 * also covers the one thing that can differ between the ranks, the clock behind the time limit. */
static bool replicate_small(const hipsdp_solver* s)
{
   if ( s->comm == NULL || s->nranks < 2 || s->shardA )
      return false;
   double lim = 2e10;
   const char* e = getenv("HIPSDP_SHARD_MIN_FLOPS");
   if ( e != NULL )
      lim = atof(e);
   double fl = 0.0;
   const double m1 = (double) s->m + 1.0;
   for (auto& B : s->blk)
      fl += 4.0 * m1 * (double) B.n * (double) B.n * (double) B.n + m1 * m1 * (double) B.n * (double) B.n;
   return fl < lim;
}

struct CommOff        /* the communicator is out of sight while a replicated (small) problem is worked on */
{
   hipsdp_solver* s; void* saved; bool on;
   CommOff(hipsdp_solver* s_, bool on_) : s(s_), saved(s_->comm), on(on_) { if ( on ) s->comm = NULL; }
   ~CommOff() { if ( on ) s->comm = saved; }
};

static int solve_impl(hipsdp_solver* s, const hipsdp_params* params, hipsdp_info* info);

static int sync_outcome(hipsdp_solver* s, int rc, hipsdp_info* info)
{
   struct Pack { int rc, solved, last_status, pre_valid; double sol_scale, tau, kappa, pre_scale; hipsdp_info info; } pk;
   memset(&pk, 0, sizeof(pk));
   pk.rc = rc; pk.solved = s->solved ? 1 : 0; pk.last_status = s->last_status; pk.pre_valid = s->pre_valid ? 1 : 0;
   pk.sol_scale = s->sol_scale; pk.tau = s->tau; pk.kappa = s->kappa; pk.pre_scale = s->pre_scale; pk.info = *info;
   const long long nint = (long long) ((sizeof(Pack) + sizeof(int) - 1) / sizeof(int));
   int* d = NULL;
   hipStream_t st = s->stream;
   HS_CALL( dalloc(&d, nint) );
   int r = HS_OK;
   if ( hipMemcpyAsync(d, &pk, sizeof(pk), hipMemcpyHostToDevice, st) != hipSuccess ) r = HS_ERR_HIP;
   if ( r == HS_OK ) r = hs_bcast_ints(s->comm, d, nint, st);
   if ( r == HS_OK && (hipMemcpyAsync(&pk, d, sizeof(pk), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) )
      r = HS_ERR_HIP;
   dfree(d);
   HS_CALL( r );
   s->solved = pk.solved != 0; s->last_status = pk.last_status; s->sol_scale = pk.sol_scale; s->tau = pk.tau; s->kappa = pk.kappa;
   s->pre_scale = pk.pre_scale;
   *info = pk.info;
   if ( pk.rc != HIPSDP_OK || !s->solved )
   {
      s->pre_valid = false;
      return pk.rc;
   }
   const int m = s->m, q = s->q;
   if ( m > 0 ) HS_CALL( hs_bcast_doubles(s->comm, s->y, m, st) );
   if ( q > 0 ) { HS_CALL( hs_bcast_doubles(s->comm, s->x, q, st) ); HS_CALL( hs_bcast_doubles(s->comm, s->z, q, st) ); }
   for (auto& B : s->blk)
   {
      const long long n2 = (long long) B.n * B.n;
      HS_CALL( hs_bcast_doubles(s->comm, B.X, n2, st) );
      HS_CALL( hs_bcast_doubles(s->comm, B.Z, n2, st) );
      B.derived_valid = false;
   }
   if ( pk.pre_valid )
   {
      if ( s->pre_y == NULL )
      {
         HS_CALL( dalloc(&s->pre_y, s->m_alloc > m ? s->m_alloc : m) );
         HS_CALL( dalloc(&s->pre_x, s->q_alloc > q ? s->q_alloc : q) );
      }
      if ( m > 0 ) HS_CALL( hs_bcast_doubles(s->comm, s->pre_y, m, st) );
      if ( q > 0 ) HS_CALL( hs_bcast_doubles(s->comm, s->pre_x, q, st) );
      for (auto& B : s->blk)
      {
         if ( B.Xpre == NULL )
            HS_CALL( dalloc(&B.Xpre, (long long) B.n * B.n) );
         HS_CALL( hs_bcast_doubles(s->comm, B.Xpre, (long long) B.n * B.n, st) );
      }
   }
   s->pre_valid = pk.pre_valid != 0;
   HS_HIP( hipStreamSynchronize(st) );
   return HIPSDP_OK;
}

extern "C" int hipsdp_solve(hipsdp_solver* s, const hipsdp_params* params, hipsdp_info* info)
{
   if ( s == NULL || !s->shaped || info == NULL )
      return HIPSDP_ERR_ARG;
   HS_HIP( hipSetDevice(s->device) );
   HS_CALL( flush_zeros(s) );
   if ( s->comm != NULL )
      HS_CALL( flush_cmds(s) );                   /* (one rank alone: the one-launch solve runs the waiting commands itself) */
   const bool alone = replicate_small(s);
   int rc;
   {
      CommOff off(s, alone);
      rc = solve_impl(s, params, info);
   }
   /* no recorded operation and no held region outlives the call: a normal return has launched everything (the final read-back
    * ends the regions), an error return may have left records behind, which are dropped */
   if ( rc == HIPSDP_OK )
      rc = hs_red_batch_end_all();
   else
      hs_red_batch_reset();
   if ( rc != HIPSDP_OK )          /* error return inside the iteration: whatever it had queued on either queue must not outlive the call */
   {
      (void) hipStreamSynchronize(s->stream);
      (void) hipStreamSynchronize(s->stream2);
   }
   if ( s->pc.open >= 0 )          /* an error return inside the iteration: close the roctx range, drop the marks */
   {
      phase_mark(s, -1);
      (void) hipStreamSynchronize(s->stream);
      phase_finish(s);
   }
   if ( alone )
      rc = sync_outcome(s, rc, info);
   return rc;
}

#define CLK_MAX_ASSEMBLIES 256
/* Shader clock DURING an assembly: one thread on a queue of its own counts its cycles (clock64) and the 100 MHz wall ticks from the
 * moment the assembly is queued until a flag says it is over - ONE thread, one counter.  [First form: one (clock64, wall clock) pair
 * from a launch before and one from a launch after the assembly - they land on different XCDs, whose cycle counters are not
 * aligned: 1.98, 2.37 GHz and garbage on some boxes.  Second: 20 us of spinning right behind the assembly - the idle device is
 * back at 2.44 GHz by then.]  out[0]: flag (set by k_clock_stop behind the assembly), out[2], out[3]: cycles, ticks. */
__global__ void k_clock_window(unsigned long long* __restrict__ out)
{
   const unsigned long long w0 = wall_clock64();
   const unsigned long long c0 = clock64();
   unsigned long long w1 = w0;
   /* (at most 0.2 s: an exit every wave reaches) */
   while ( __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0ULL && w1 - w0 < 20000000ULL )
   {
      __builtin_amdgcn_s_sleep(32);
      w1 = wall_clock64();
   }
   const unsigned long long c1 = clock64();
   w1 = wall_clock64();
   out[2] = c1 - c0;
   out[3] = w1 - w0;
}
__global__ void k_clock_stop(unsigned long long* __restrict__ out)
{
   __hip_atomic_store(out, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

extern "C" int hipsdp_set_clock_sampling(hipsdp_solver* s, int on)
{
   if ( s == NULL ) return HIPSDP_ERR_ARG;
   HS_HIP( hipSetDevice(s->device) );
   HS_CALL( stage_sync(s) );
   if ( on && s->clk_buf == NULL )
      HS_HIP( hipMalloc((void**) &s->clk_buf, (size_t) CLK_MAX_ASSEMBLIES * 4 * sizeof(unsigned long long)) );
   if ( on && s->clk_stream == NULL )
      HS_HIP( hipStreamCreateWithFlags(&s->clk_stream, hipStreamNonBlocking) );
   s->clk_on = on != 0;
   return HIPSDP_OK;
}

extern "C" int hipsdp_get_assembly_clock(hipsdp_solver* s, double* ghz)
{
   if ( s == NULL || ghz == NULL ) return HIPSDP_ERR_ARG;
   *ghz = s->clk_ghz;
   return HIPSDP_OK;
}

/* ---- B&B-sized problems: the whole solve in one launch of one workgroup (csrc/solve1.hip).  Returns HS_OK and *done = true when the
 * solve ran there; *done = false: not this path's problem (shape, options) or the kernel declined (too much work for one compute
 * unit) - nothing has been touched and the general path below takes over.  HIPSDP_SOLVE1=0 switches the path off. */
#define S1_HIST_MAX 256
#define S1_SOL_OFF (HS_S1_OUT_DOUBLES + 8 + 16 * S1_HIST_MAX)       /* y (128), x (4096), z (4096) */
#define S1_HOST_DOUBLES (S1_SOL_OFF + 128 + 2 * 4096)
static long long g_solve1_solves = 0;      /* solves of this process that ran in the one launch (bench.py reports the share) */
static long long g_solve1_fallbacks = 0;   /* ... that the kernel gave up on numerically and the general path solved again */
static long long g_solve1_fallbacks_warm = 0;   /* ... of these: warm-started solves (retried from the caller's start) */
static int solve1_try(hipsdp_solver* s, hipsdp_info* info, bool* done)
{
   *done = false;
   s->s1_last = 0;
   s->s1_sol_host = false;
   /* (read at every solve: tests and tools switch the path between two solves of one process) */
   int on = 1, prof = 0;
   double maxwork = 3e6;
   /* largest block the kernel is offered (HIPSDP_SOLVE1_MAXN): whatever fits its LDS - one block of 36 rows, two of 30, eight of 12.
    * Measured with sparse variable matrices, dense constant matrices and LP rows of density 0.3 (tests/devtools/solve1_sizes.py) it
    * is ahead of the general path at every such size: 16 rows 0.124 against 0.281 ms per iteration, 24: 0.233 / 0.360, 32: 0.360 /
    * 0.422, two of 30: 0.449 / 0.572.  [Mid-round it lost at 32 rows, 0.62 against 0.43 - a dense constant matrix went through
    * scalar loops of one wavefront and dense LP rows through a walk of their nonzeros - and the default was 24.] */
   int maxn = 64;
   /* most variables the kernel is offered (HIPSDP_SOLVE1_MAXM): with the Schur matrix as a packed triangle m = 128 fits its LDS, but
    * the factorization of M is one workgroup's (the panel recurrence one wavefront's) and grows with m^3 - against the general path
    * (tests/devtools/solve1_mbig.py, ms per iteration): m = 70: 0.227 / 0.393, 90: 0.330 / 0.426, 105: 0.424 / 0.453 (example_MkP's
    * root, two nonzeros per matrix: 0.235 / 0.450), 110: 0.520 / 0.501, 120: 0.465 / 0.436, 128: 0.541 / 0.476 */
   int maxm = 108;
   {
      if ( getenv("HIPSDP_SOLVE1_MAXM") != NULL )
         maxm = atoi(getenv("HIPSDP_SOLVE1_MAXM"));
      const char* env = getenv("HIPSDP_SOLVE1");
      on = (env != NULL && env[0] == '0') ? 0 : 1;
      if ( getenv("HIPSDP_SOLVE1_MAXWORK") != NULL )
         maxwork = atof(getenv("HIPSDP_SOLVE1_MAXWORK"));
      prof = getenv("HIPSDP_SOLVE1_PROF") != NULL ? atoi(getenv("HIPSDP_SOLVE1_PROF")) : 0;
      if ( getenv("HIPSDP_SOLVE1_MAXN") != NULL )
         maxn = atoi(getenv("HIPSDP_SOLVE1_MAXN"));
   }
   const hipsdp_params& par = s->par;
   if ( !on || s->comm != NULL || s->shardA || s->schur_mode_forced || par.verbose || s->pc.on )
      return HS_OK;
   if ( getenv("HIPSDP_REFINE_SOLVES") != NULL && getenv("HIPSDP_REFINE_SOLVES")[0] == '0' )
      return HS_OK;
   const int K = (int) s->blk.size();
   int ns[HS_S1_MAXBLK];
   if ( K < 1 || K > HS_S1_MAXBLK || s->m > maxm )
      return HS_OK;
   for (int k = 0; k < K; ++k)
   {
      if ( s->blk[k].sparse || s->blk[k].A == NULL || s->blk[k].n > maxn )
         return HS_OK;
      ns[k] = s->blk[k].n;
   }
   if ( !hs_solve1_fits(s->m, s->q, K, ns) )
      return HS_OK;
   hipStream_t st = s->stream;
   const long long want = hs_solve1_ws_doubles(s->m, s->q, K, ns);
   if ( s->s1_ws_len < want )
   {
      if ( s->s1_ws != NULL )
      {
         HS_HIP( hipStreamSynchronize(st) );
         (void) hipFree(s->s1_ws);
         s->s1_ws = NULL; s->s1_ws_len = 0;
      }
      const long long len = want + want / 2 + 4096;
      HS_HIP( hipMalloc((void**) &s->s1_ws, (size_t) len * sizeof(double)) );
      s->s1_ws_len = len;
   }
   if ( s->s1_host == NULL )
   {
      HS_HIP( hipHostMalloc((void**) &s->s1_host, (size_t) S1_HOST_DOUBLES * sizeof(double),
            hipHostMallocMapped | hipHostMallocCoherent) );
      HS_HIP( hipHostGetDevicePointer((void**) &s->s1_host_dev, s->s1_host, 0) );
      memset(s->s1_host, 0, (size_t) (HS_S1_OUT_DOUBLES + 8) * sizeof(double));
   }
   if ( par.preoptgap > 0.0 )
   {
      if ( s->pre_y == NULL )
      {
         HS_CALL( dalloc(&s->pre_y, s->m_alloc > s->m ? s->m_alloc : s->m) );
         HS_CALL( dalloc(&s->pre_x, s->q_alloc > s->q ? s->q_alloc : s->q) );
      }
      for (auto& B : s->blk)
         if ( B.Xpre == NULL )
            HS_CALL( dalloc(&B.Xpre, (long long) B.n * B.n) );
   }
   const auto t_begin = std::chrono::steady_clock::now();
   hs_solve1_args a;
   memset(&a, 0, sizeof(a));
   a.m = s->m; a.q = s->q; a.nblk = K;
   for (int k = 0; k < K; ++k)
   {
      Block& B = s->blk[k];
      a.n[k] = B.n; a.A[k] = B.A; a.X[k] = B.X; a.Z[k] = B.Z; a.Xpre[k] = B.Xpre;
   }
   a.b = s->b; a.Dext = s->Dext; a.y = s->y; a.x = s->x; a.z = s->z; a.pre_y = s->pre_y; a.pre_x = s->pre_x;
   a.gaptol = par.gaptol; a.feastol = par.feastol; a.infeastol = par.infeastol; a.objlimit = par.objlimit; a.timelimit = par.timelimit;
   a.gamma = par.gamma; a.pabstol = par.pabstol; a.preoptgap = par.preoptgap; a.elapsed0 = 0.0; a.maxwork = maxwork;
   a.maxiter = par.maxiter; a.settings = par.settings; a.have_start = s->have_start ? 1 : 0;
   {
      const char* env = getenv("HIPSDP_PIVOT_RULE");
      a.pivot_rule = env != NULL ? atoi(env) : 3;
   }
   a.prof_on = prof;
   const int nofb = (getenv("HIPSDP_SOLVE1_NO_FALLBACK") != NULL && getenv("HIPSDP_SOLVE1_NO_FALLBACK")[0] == '1') ? 1 : 0;    /* (read at every solve, like the other switches of this path) */
   a.keep_on_fail = nofb ? 0 : 1;
   a.gws = s->s1_ws; a.gws_len = s->s1_ws_len;
   a.out = s->s1_host_dev;
   a.hist = (getenv("HIPSDP_SOLVE1_HIST") != NULL && getenv("HIPSDP_SOLVE1_HIST")[0] != '0') ? s->s1_host_dev + HS_S1_OUT_DOUBLES + 8 : NULL;
   a.hist_len = S1_HIST_MAX;
   a.hy = s->s1_host_dev + S1_SOL_OFF; a.hx = a.hy + 128; a.hz = a.hx + 4096;
   a.seq = ++s->s1_seq;
   a.flag = reinterpret_cast<unsigned long long*>(s->s1_host_dev + HS_S1_OUT_DOUBLES);
   a.cmds = s->arena_d; a.ncmd = s->ncmd;         /* the node's setters: run by the same launch */
   if ( s->ncmd > 0 )
   {
      const size_t used = s->stage_off > NC_BYTES ? s->stage_off : NC_BYTES;
      a.cmd_bytes = (long long) ((used + 15) & ~(size_t) 15);         /* (the arena's size is a multiple of 16) */
   }
   s->ncmd = 0;
   HS_CALL( hs_solve1_launch(st, &a) );
   {
      volatile unsigned long long* flag = reinterpret_cast<volatile unsigned long long*>(s->s1_host + HS_S1_OUT_DOUBLES);
      long long spins = 0;
      while ( *flag != a.seq )
      {
         if ( (++spins & 0x3FFF) == 0 )
         {
            const hipError_t e = hipStreamQuery(st);
            if ( e == hipSuccess )
            {
               if ( *flag == a.seq )
                  break;
               set_err("the one-launch solve finished without publishing its results");
               return HS_ERR_HIP;
            }
            if ( e != hipErrorNotReady )
            {
               hs_record_hip_error(e, "hipStreamQuery(one-launch solve)", __FILE__, __LINE__);
               return HS_ERR_HIP;
            }
         }
      }
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
   }
   const double* o = s->s1_host;
   const int status = (int) o[0];
   if ( status == -2 )
      return HS_OK;                              /* declined: the general path takes the problem */
   /* A numerical failure of the kernel is not final: on about 1 % of random shapes with cond(M) around
    * 1e14 one path gives up where the other converges (profiles/r04_c_solve1_fuzz.txt) - their last steps differ in how the solves
    * with M are corrected.  The general path solves the problem again before the caller escalates its settings, FROM THE SAME START:
    * with keep_on_fail the kernel has left y, x, z, X, Z in device memory as the node's setters wrote them (a caller's start point
    * stays in place and have_start stays as it was; without one the general path starts cold as it always does).
    * HIPSDP_SOLVE1_NO_FALLBACK=1 keeps the kernel's failure (and its last iterate). */
   if ( status == HIPSDP_STATUS_NUMERIC && nofb == 0 )
   {
      HS_HIP( hipStreamSynchronize(st) );
      (void) __sync_add_and_fetch(&g_solve1_fallbacks, 1);
      if ( (int) o[12] != 0 )
         (void) __sync_add_and_fetch(&g_solve1_fallbacks_warm, 1);
      return HS_OK;
   }
   /* the result block, y, x and z are in pinned memory, published before the sequence word; X and Z in device memory are complete
    * when the kernel has retired - whoever reads them waits for the queue first (stage_sync), this call does not */
   s->stage_pending = true;
   memset(info, 0, sizeof(*info));
   s->tau = o[9];
   s->kappa = o[10];
   const double pobj = o[2], dobj = o[3];
   if ( status == HIPSDP_STATUS_DINF || status == HIPSDP_STATUS_DUNB || status == HIPSDP_STATUS_PDINF )
      s->sol_scale = 1.0 / fmax(fmax(fabs(dobj), fabs(pobj)), 1e-300);
   else
      s->sol_scale = 1.0 / s->tau;
   s->last_status = status;
   s->solved = true;
   s->have_start = false;
   s->pre_valid = o[13] != 0.0;
   s->pre_scale = o[14];
   for (auto& B : s->blk)
      B.derived_valid = false;
   info->status = status;
   info->iterations = (int) o[1];
   info->pobj = pobj * s->sol_scale;
   info->dobj = dobj * s->sol_scale;
   info->pinf = o[4]; info->dinf = o[5]; info->dabs = o[6]; info->gap = o[7]; info->mu = o[8];
   info->tau = s->tau; info->kappa = s->kappa;
   info->chol_fail = (int) o[11];
   info->warm_started = (int) o[12];
   info->settings_used = par.settings;
   info->schur_calls = info->iterations;
   for (auto& B : s->blk)
      info->schur_flops += (double) info->iterations * (4.0 * (s->m + 1) * (double) B.n * B.n * B.n + (double) (s->m + 1) * (s->m + 1) * (double) B.n * B.n);
   info->solve_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
   if ( prof )
   {
      static const char* const names[12] = {"lists", "start", "residuals", "inverse factors + LP part", "Zinv", "Schur", "chol M + X Rd",
         "predictor rhs", "predictor step", "corrector rhs", "corrector step", "update"};
      fprintf(stderr, "hipsdp one-launch solve: %d iterations, %.1f us on the device (%.0f cycles)", info->iterations,
         o[43] / 100.0, o[17]);
      for (int i = 0; i < 12; ++i)
         fprintf(stderr, "%s %s %.0f", i ? "," : ":", names[i], o[18 + i]);
      fprintf(stderr, " | sub");
      for (int i = 12; i < 22; ++i)
         fprintf(stderr, " %.0f", o[18 + i]);
      fprintf(stderr, " | nnz A %d, LP %d\n", (int) o[40], (int) o[41]);
   }
   s->s1_last = 1;
   s->s1_sol_host = true;
   (void) __sync_add_and_fetch(&g_solve1_solves, 1);
   *done = true;
   return HS_OK;
}

static int solve_impl(hipsdp_solver* s, const hipsdp_params* params, hipsdp_info* info)
{
   g_err[0] = 0;
   if ( s == NULL || !s->shaped || info == NULL )
      return HIPSDP_ERR_ARG;
   if ( s->shardA && s->comm == NULL )
   {
      set_err("hipsdp_solve: the matrices of this problem are sharded by variable, the communicator is gone");
      return HIPSDP_ERR_ARG;
   }
   HS_HIP( hipSetDevice(s->device) );
   if ( params != NULL )
      s->par = *params;
   else
      hipsdp_default_params(&s->par);
   {
      const char* env = getenv("HIPSDP_LANCZOS");
      if ( env != NULL && atoi(env) > 0 )
         s->par.lanczos_steps = atoi(env);
      if ( s->par.lanczos_steps <= 0 )
      {
         /* default: 24 steps; 16 once every block has more than 64 rows - there each step is a launch of its own and the estimate
          * (used with its residual bound subtracted, the Cholesky of the new iterate being the exact test) costs 4 % of an iteration
          * at n = 500 and 20 % at n = 200; measured on the bench instances: same iteration counts from 12 steps on
          * (tools/lanczos_steps.sh) */
         int nmin = 1 << 30;
         for (auto& B : s->blk) if ( B.n < nmin ) nmin = B.n;
         s->par.lanczos_steps = (!s->blk.empty() && nmin > 64) ? 16 : 24;
      }
      if ( s->par.settings < 0 ) s->par.settings = 0;
      if ( s->par.settings > 2 ) s->par.settings = 2;
      s->par.lanczos_steps <<= s->par.settings;          /* medium: twice, stable: four times the Lanczos steps */
      bool lan_used = false;                  /* blocks with more than 64 rows take their step lengths from the one-launch Lanczos */
      for (auto& B : s->blk)
         if ( B.n > 64 ) lan_used = true;
      if ( s->lan_sync != NULL && lan_used )
      {
         /* exchange vectors of the one-launch Lanczos runs: a solve starts from their initial state (a run that gave up
          * leaves them in no particular one) */
         HS_CALL( hs_lanczos_sync_reset(s->stream, s->lan_sync, s->lan_rot) );
      }
      if ( s->trsv_ws != NULL && s->m > 2 * 64 )
         HS_CALL( hs_trsv_sync_init(s->stream, s->m, s->trsv_ws, &s->trsv_epoch) );      /* likewise the triangular solves' */
      if ( s->par.lanczos_steps < 4 ) s->par.lanczos_steps = 4;
      if ( s->par.lanczos_steps > 250 ) s->par.lanczos_steps = 250;
   }
   const hipsdp_params& par = s->par;
   /* the retry ladder of the backend (sdpisolver_sdpa.cpp:1415-1449: fast / default / stable parameter sets): more conservative
    * settings take shorter steps, keep the iterates more central and wait longer before they call a stall (oracle/ipm_ref.py:
    * Params.settings, same numbers) */
   const int settings = par.settings;
   const double gamma_eff = settings == 0 ? par.gamma : fmin(par.gamma, settings == 1 ? 0.9 : 0.75);
   const int stall_lim = settings == 0 ? 3 : (settings == 1 ? 5 : 8);
   const int nobest_lim = settings == 0 ? 6 : (settings == 1 ? 10 : 15);
   const double sigma_floor = settings == 0 ? 1e-8 : (settings == 1 ? 1e-4 : 1e-2);
   const int maxiter = par.maxiter;
   hs_comm_phase(2);
   for (int p = 0; p < PH_COUNT; ++p) s->pc.ms[p] = 0.0;
   const int m = s->m, m1 = s->m + 1, q = s->q;
   const int K = (int) s->blk.size();
   hipStream_t st = s->stream;
   const auto t_begin = std::chrono::steady_clock::now();
   memset(info, 0, sizeof(*info));
   info->status = HIPSDP_STATUS_UNSOLVED;
   hs_red_batch_reset();
   s->pre_valid = false;
   s->clk_n = 0;
   s->clk_ghz = 0.0;
   {
      bool done1 = false;
      HS_CALL( solve1_try(s, info, &done1) );
      if ( done1 )
      {
         s->stage_off = 0;                                      /* (the kernel ran the commands and read their data at its start) */
         return HIPSDP_OK;
      }
   }
   HS_CALL( stage_sync(s) );                                  /* the general path uses both queues */
   HS_CALL( ensure_schur_ws(s) );
   HS_CALL( ensure_packed(s) );
   {
      int nmaxb = 0;
      for (auto& B : s->blk) if ( B.n > nmaxb ) nmaxb = B.n;
      s->use2 = (nmaxb >= 128) && getenv("HIPSDP_ONEQUEUE") == NULL;
   }

   long long N = q;
   for (auto& B : s->blk) N += B.n;
   const double N1 = (double) (N + 1);

   /* norms of b and of the constant part (host needs them once) */
   HS_CALL( hs_dot(st, m, s->b, s->b, s->sc + 0, 0, s->red_ws) );
   HS_CALL( hs_fill(st, s->sc + 1, 1, 0.0) );
   for (auto& B : s->blk)
      HS_CALL( hs_dot(st, (long long) B.n * B.n, B.A0, B.A0, s->sc + 1, 1, s->red_ws) );
   if ( q > 0 )
   {
      /* column 0 of Dext: strided -> gather through gemv with unit vector would be overkill; copy it */
      HS_HIP( hipMemcpy2DAsync(s->tmpq, sizeof(double), s->Dext, (size_t) m1 * sizeof(double), sizeof(double), (size_t) q, hipMemcpyDeviceToDevice, st) );
      HS_CALL( hs_dot(st, q, s->tmpq, s->tmpq, s->sc + 1, 1, s->red_ws) );
   }
   /* (into the pinned mirror of the scalars: a copy into a stack variable is staged by the runtime and blocks) */
   HS_HIP( hipMemcpyAsync(s->hsc, s->sc, 2 * sizeof(double), hipMemcpyDeviceToHost, st) );
   HS_HIP( hipStreamSynchronize(st) );
   const double normb = sqrt(s->hsc[0]);
   const double normC = sqrt(s->hsc[1]);

   /* ---- starting point */
   /* cold start: X = Z = xi I in every block until the first step - the first Schur complement is then the Gram matrix of the
    * constraint matrices themselves, M_ij = tr(A_i (xi I) A_j (I / xi)) = <A_i, A_j>, and the two n^3 products of the assembly
    * (which would multiply by sqrt(xi) I and I / sqrt(xi)) are not needed for it (round 5; single rank, dense blocks, W form) */
   bool identity_start = false;
   bool start_factors = false;
   for (auto& B : s->blk)
      B.derived_valid = false;
   if ( s->have_start )
   {
      /* a caller-supplied point (warm start, sdpisolver.h:160-173) is used when it is strictly interior: X_k, Z_k positive
       * definite (checked by the factorizations the first iteration needs anyway) and x, z > 0; kappa = its mean
       * complementarity (tau = 1).  Otherwise the solve falls back to the cold start below. */
      HS_HIP( hipMemsetAsync(s->flags, 0, 8 * sizeof(int), st) );
      HS_CALL( hs_fill(st, s->sc + SC_XZ, 1, 0.0) );
      for (auto& B : s->blk)
      {
         const long long n2 = (long long) B.n * B.n;
         HS_CALL( hs_symmetrize(st, B.X, B.n) );
         HS_CALL( hs_symmetrize(st, B.Z, B.n) );
         HS_CALL( hs_copy(st, B.Lz, B.Z, n2) );
         HS_CALL( hs_potrf(st, B.n, B.Lz, B.dinvz, s->flags + 0, NULL) );
         HS_CALL( hs_copy(st, B.Lx, B.X, n2) );
         HS_CALL( hs_potrf(st, B.n, B.Lx, B.dinvx, s->flags + 1, NULL) );
         HS_CALL( hs_dot(st, n2, B.X, B.Z, s->sc + SC_XZ, 1, s->red_ws) );
      }
      if ( q > 0 )
      {
         hipLaunchKernelGGL(k_flag_nonpositive, g1d(q), dim3(256), 0, st, q, s->x, s->z, s->flags + 3);
         HS_LAUNCH_CHECK();
         HS_CALL( hs_dot(st, q, s->x, s->z, s->sc + SC_XZ, 1, s->red_ws) );
      }
      int f4[4];
      double xz = 0.0;
      HS_HIP( hipMemcpyAsync(s->hsc + 4, s->flags, 4 * sizeof(int), hipMemcpyDeviceToHost, st) );
      HS_HIP( hipMemcpyAsync(s->hsc + 8, s->sc + SC_XZ, sizeof(double), hipMemcpyDeviceToHost, st) );
      HS_HIP( hipStreamSynchronize(st) );
      memcpy(f4, s->hsc + 4, sizeof(f4));
      xz = s->hsc[8];
      const double mu0 = xz / (double) (N > 0 ? N : 1);
      if ( f4[0] == 0 && f4[1] == 0 && f4[3] == 0 && std::isfinite(mu0) && mu0 > 0.0 )
      {
         s->tau = 1.0;
         s->kappa = mu0;
         start_factors = true;
      }
      else
         s->have_start = false;
   }
   if ( !s->have_start )
   {
      const double xi = fmax(1.0, sqrt(fmax(fmax(normb, normC), 1.0)));
      HS_CALL( hs_fill(st, s->y, m, 0.0) );
      for (auto& B : s->blk)
      {
         HS_CALL( hs_set_identity(st, B.X, B.n, xi) );
         HS_CALL( hs_set_identity(st, B.Z, B.n, xi) );
      }
      HS_CALL( hs_fill(st, s->x, q, xi) );
      HS_CALL( hs_fill(st, s->z, q, xi) );
      s->tau = 1.0;
      s->kappa = xi * xi;
      identity_start = true;
   }
   info->warm_started = s->have_start ? 1 : 0;
   s->have_start = false;
   /* HIPSDP_NO_IDENTITY_START=1: the first assembly of a cold solve through the general products as well (A/B runs, tests) */
   if ( getenv("HIPSDP_NO_IDENTITY_START") != NULL && getenv("HIPSDP_NO_IDENTITY_START")[0] == '1' )
      identity_start = false;

   int status = HIPSDP_STATUS_ITERLIM;
   int it = 0, certwait = 0, nstall = 0;
   double lastmu = 1e300, alpha_last = 1.0, bestmerit = 1e300;
   int sincebest = 0;
   double mu = 0, pinf = 0, dinf = 0, dabs = 0, gap = 0, pobj = 0, dobj = 0;
   HostScalars hs;
   int hflags[3] = {0, 0, 0};
   bool want_cert = false;
   bool factors_valid = start_factors;      /* the factors of an accepted warm start are those of the first iteration */
   double schur_ms = 0.0;

   /* ---- residuals of the current iterate: enqueued at the top of an iteration, or (small problems) already at the end of
    * the previous one together with the step's Cholesky check, so that one read-back serves both */
   /* The dual residual obeys an exact recurrence: dZ is DEFINED as A^T(dy~) + eta Rd, so after the update (y~, Z) += alpha
    * (dy~, dZ) the new residual is (1 - alpha eta) Rd whatever the quality of dy (forced pivots included), and likewise for the
    * LP rows.  The general path (blocks above 64 rows: each pass over A is HBM bound, 1 GB at n = 500 / m = 1000) therefore
    * scales the residual it has instead of sweeping A again; it is recomputed from scratch at the first iteration and whenever
    * the certificate residuals are needed (tau -> 0 divides the rounding the recurrence carries).  HIPSDP_RD_RECOMPUTE=1: always. */
   static const bool rd_recompute = getenv("HIPSDP_RD_RECOMPUTE") != NULL && atoi(getenv("HIPSDP_RD_RECOMPUTE")) != 0;
   bool rd_have = false;             /* Rd / rd hold the residual of SOME iterate: the current one once rd_pending is worked off */
   bool rd_pending = false;
   double rd_scale = 1.0;
   auto enqueue_residuals = [&]() -> int
   {
      const bool recur = rd_have && !want_cert && !rd_recompute;
      HS_CALL( hs_make_ext(st, m, -s->tau, 1.0, s->y, s->yt) );
      HS_LAUNCH_CHECK();
      hs_red_batch_begin(st);         /* the reductions of this phase run in one launch, right before the scalars are read */
      HS_CALL( hs_fill_scalar(st, s->sc + SC_RD2, 0.0) );
      HS_CALL( hs_fill_scalar(st, s->sc + SC_XZ, 0.0) );
      HS_CALL( hs_fill_scalar(st, s->sc + SC_HD2, 0.0) );
      std::vector<double*> Xs;
      for (int k = 0; k < K; ++k)
      {
         Block& B = s->blk[k];
         const long long n2 = (long long) B.n * B.n;
         if ( !recur )
            HS_CALL( pass_AT(s, B, s->yt, -1.0, B.Z, B.Rd) );
         else if ( rd_pending )
            HS_CALL( hs_scale_add(st, n2, rd_scale, B.Rd, 0.0, NULL, B.Rd) );
         HS_CALL( hs_dot(st, n2, B.Rd, B.Rd, s->sc + SC_BLK(k, 0), 0, s->red_ws) );
         HS_CALL( hs_dot(st, n2, B.X, B.Z, s->sc + SC_XZ, 1, s->red_ws) );
         if ( want_cert )
         {
            HS_CALL( hs_red_batch_flush() );
            hipLaunchKernelGGL(k_cert, g1d(n2), dim3(256), 0, st, n2, s->tau, B.Rd, B.A0, B.T1);
            HS_LAUNCH_CHECK();
            HS_CALL( hs_dot(st, n2, B.T1, B.T1, s->sc + SC_HD2, 1, s->red_ws) );
         }
         Xs.push_back(B.X);
      }
      if ( q > 0 )
      {
         if ( small_problem(s) )
         {
            HS_CALL( hs_lp_rows_small(st, q, m1, s->Dext, s->yt, 0, 0.0, 0.0, s->x, s->z,
               (const double*) NULL, (const double*) NULL, s->rd, (double*) NULL) );
         }
         else if ( !recur )
         {
            const double* v = s->yt;
            HS_CALL( hs_gemv_n(st, q, m1, s->Dext, m1, 1, &v, s->tmpq, q, s->gemv_ws, s->gemv_ws_len) );
            HS_CALL( hs_scale_add(st, q, 1.0, s->tmpq, -1.0, s->z, s->rd) );
         }
         else if ( rd_pending )
            HS_CALL( hs_scale_add(st, q, rd_scale, s->rd, 0.0, NULL, s->rd) );
         HS_CALL( hs_dot(st, q, s->rd, s->rd, s->sc + SC_RD2, 1, s->red_ws) );
         HS_CALL( hs_absmax(st, q, s->rd, s->sc + SC_RDLPMAX, 0, s->red_ws) );
         HS_CALL( hs_dot(st, q, s->x, s->z, s->sc + SC_XZ, 1, s->red_ws) );
         if ( want_cert )
         {
            /* rd + tau * c */
            HS_CALL( hs_red_batch_flush() );
            HS_HIP( hipMemcpy2DAsync(s->tmpq, sizeof(double), s->Dext, (size_t) m1 * sizeof(double), sizeof(double), (size_t) q, hipMemcpyDeviceToDevice, st) );
            HS_CALL( hs_scale_add(st, q, s->tau, s->tmpq, 1.0, s->rd, s->tmpq) );
            HS_CALL( hs_dot(st, q, s->tmpq, s->tmpq, s->sc + SC_HD2, 1, s->red_ws) );
         }
      }
      else
         HS_CALL( hs_fill_scalar(st, s->sc + SC_RDLPMAX, 0.0) );
      {
         const int fusedA = apply_A_small(s, Xs.data(), s->x, s->AX, 2, s->tau, s->b, s->rp);
         if ( fusedA < 0 )
            return -fusedA;
         if ( fusedA == 0 )
            HS_CALL( apply_A(s, Xs.data(), s->x, s->AX) );
         HS_CALL( hs_copy_scalar(st, s->sc + SC_AX0, s->AX) );
         if ( fusedA == 0 )
         {
            HS_CALL( hs_red_batch_flush() );
            hipLaunchKernelGGL(k_rp, g1d(m > 0 ? m : 1), dim3(256), 0, st, m, s->tau, s->b, s->AX, s->rp);
            HS_LAUNCH_CHECK();
         }
      }
      HS_CALL( hs_dot(st, m, s->rp, s->rp, s->sc + SC_RP2, 0, s->red_ws) );
      HS_CALL( hs_dot(st, m, s->AX + 1, s->AX + 1, s->sc + SC_HP2, 0, s->red_ws) );
      HS_CALL( hs_dot(st, m, s->b, s->y, s->sc + SC_DOBJ, 0, s->red_ws) );
      rd_pending = false;
      return HS_OK;
   };
   bool residuals_ready = false;
   bool small_all = (s->comm == NULL);
   for (auto& B : s->blk)
      if ( B.n > 64 )
         small_all = false;
   /* one block of at most 64 rows: every Cholesky flag has exactly one writing launch between two reads, which then stores its
    * result itself and the flags need no clearing */
   const int setf = (small_all && s->blk.size() == 1) ? 1 : 0;

   /* Overlap on the second queue (blocks of at least 128 rows, HIPSDP_ONEQUEUE / HIPSDP_NO_OVERLAP switch it off):
    *  - the Z chain of the factorization phase (inverse factor, Z^-1) needs nothing from the residual pass, so it is started at
    *    the top of the iteration and runs beside the residual kernels and the host's read-back of the termination scalars;
    *  - the predictor's right-hand side (H_k, A(H), h) needs nothing from the Schur matrix, so it runs beside the latency-bound
    *    factorization of M (one communicator serves one queue: not with several ranks). */
   static const bool no_overlap = getenv("HIPSDP_NO_OVERLAP") != NULL;
   bool zchain_queued = false;
   auto enqueue_z_chains = [&]() -> int
   {
      hipStream_t st2 = s->stream2;
      for (auto& B : s->blk)
      {
         const int n = B.n;
         const long long n2 = (long long) n * n;
         if ( n <= 64 )
            continue;
         if ( !factors_valid )
         {
            HS_CALL( hs_copy(st2, B.Lz, B.Z, n2) );
            HS_CALL( hs_potrf(st2, n, B.Lz, B.dinvz, s->flags + 0, NULL) );
         }
         HS_CALL( hs_trtri(st2, n, B.Lz, B.dinvz, B.LzInv, B.T2) );
         HS_CALL( gemm_on(st2, s->gws2, s->gws_len, HS_MC, HS_MC, n, n, n, 1.0, B.LzInv, n, B.LzInv, n, 0.0, B.Zinv, n, HS_GEMM_LOWER) );
         HS_CALL( hs_mirror_lower(st2, B.Zinv, n, n) );
      }
      return HS_OK;
   };

   /* round 5: the X chain of the factorization phase (triangular inverse of the factor of X) does not depend on the termination
    * scalars either: it is queued behind the kernel that publishes them and runs while the host waits, decides and launches (a
    * solve that ends here has run it in vain: 0.1 ms at n = 500).
    * INVARIANT: the chain is speculative - when the solve ends at this read-back, B.Lx, B.dinvx, B.LxInv, B.T1 and
    * flags[1] already hold the factors of the FINAL X.  Lx, LxInv, dinvx and T1 are therefore UNDEFINED after solve_impl returns
    * (nothing reads them: every getter works from X, Z, y; the next solve refactors).  The chain's time is charged to PH_RESID, not
    * PH_FACTOR, in the phase anatomy (bench.py says so next to those two figures). */
   bool xchain_queued = false;
   const std::function<int()> enqueue_x_chains = [&]() -> int
   {
      if ( xchain_queued )
         return HS_OK;
      for (auto& B : s->blk)
      {
         const int n = B.n;
         if ( n <= 64 )
            continue;
         if ( !factors_valid )
         {
            HS_CALL( hs_copy(st, B.Lx, B.X, (long long) n * n) );
            HS_CALL( hs_potrf(st, n, B.Lx, B.dinvx, s->flags + 1, NULL) );
         }
         HS_CALL( hs_zero_upper(st, B.Lx, n) );
         HS_CALL( hs_trtri(st, n, B.Lx, B.dinvx, B.LxInv, B.T1) );
      }
      xchain_queued = true;
      return HS_OK;
   };
   /* round 6: every block above 64 rows, one rank, the W formulation (W_j = G A_j R): the inverse factor of X is not an operand of the
    * assembly (R = Lx itself) and G = LzInv is the operand of its SECOND product only.  The chains of the second queue are then
    * awaited between the first and the second product (hs_schur_ws.ev_g2), and the inverse factor of X is put into the second queue
    * when the first product has been launched (hs_schur_ws.after_g1) - 75 us of device time and as much of the host's launching that
    * stood between the termination decision and the assembly (profiles/r06_*_iter_sequence.txt). */
   auto defer_possible = [&]() -> bool
   {
      if ( !(s->use2 && !no_overlap && s->comm == NULL && !s->shardA && !s->schur_mode_cols && !s->schur_mode_rows
            && !s->schur_mode_U && !s->schur_mode_forced && K > 0 && getenv("HIPSDP_NO_DEFER_JOIN") == NULL) )
         return false;
      for (auto& B : s->blk)
         if ( B.n <= 64 || B.sparse )
            return false;
      return true;
   };
   struct AfterG1 { hipsdp_solver* s; };
   AfterG1 after_g1_arg = {s};
   auto after_g1_fn = [](void* p) -> int
   {
      hipsdp_solver* s = static_cast<AfterG1*>(p)->s;
      HS_HIP( hipEventRecord(s->evFork, s->stream) );              /* (behind hs_zero_upper of Lx and the first product) */
      HS_HIP( hipStreamWaitEvent(s->stream2, s->evFork, 0) );
      for (auto& B : s->blk)
         HS_CALL( hs_trtri(s->stream2, B.n, B.Lx, B.dinvx, B.LxInv, B.T1) );
      HS_HIP( hipEventRecord(s->evJoin, s->stream2) );
      s->sws.ev_g2 = (void*) s->evJoin;
      return HS_OK;
   };

   for (it = 0; it <= maxiter; ++it)
   {
      if ( !residuals_ready )
      {
         phase_mark(s, PH_RESID);
         if ( s->use2 && !no_overlap && !zchain_queued && K > 0 )
         {
            HS_HIP( hipMemsetAsync(s->flags, 0, 8 * sizeof(int), st) );
            HS_CALL( fork2(s) );
            HS_CALL( enqueue_z_chains() );
            zchain_queued = true;
         }
         HS_CALL( enqueue_residuals() );
         HS_CALL( read_scalars(s, hs, NULL, (zchain_queued && s->comm == NULL && !(factors_valid && defer_possible())) ? &enqueue_x_chains : NULL) );
      }
      residuals_ready = false;

      const double tau = s->tau, kappa = s->kappa;
      pobj = hs.v[SC_AX0];
      dobj = hs.v[SC_DOBJ];
      const double rg = pobj - dobj - kappa;
      mu = (hs.v[SC_XZ] + tau * kappa) / N1;
      double rd2 = hs.v[SC_RD2];
      double rdmax = hs.v[SC_RDLPMAX];
      for (int k = 0; k < K; ++k)
      {
         rd2 += hs.v[SC_BLK(k, 0)];
         rdmax = fmax(rdmax, sqrt(hs.v[SC_BLK(k, 0)]));
      }
      pinf = sqrt(hs.v[SC_RP2]) / tau / (1.0 + normb);
      const double pabs = sqrt(hs.v[SC_RP2]) / tau;
      const bool pabsok = par.pabstol <= 0.0 || pabs <= par.pabstol;
      dinf = sqrt(rd2) / tau / (1.0 + normC);
      dabs = rdmax / tau;
      gap = fabs(dobj - pobj) / tau;
      /* The triangular solves with the factor of M go block by block as x = inv(L_bb) r, whose residual grows with cond(L_bb)
       * and ends up as primal infeasibility of the step; each of them is corrected once with the factor itself (hs_trsv mode
       * bit 4; about 0.3 % of a solve at n = 500, m = 1000).  HIPSDP_REFINE_SOLVES=0 switches the correction off. */
      {
         static int forced = -2;
         if ( forced == -2 )
         {
            const char* env = getenv("HIPSDP_REFINE_SOLVES");
            forced = (env == NULL) ? -1 : (env[0] == '0' ? 0 : 1);
         }
         s->refine_solves = (forced != 0);
      }
      if ( par.verbose )
         printf("hipsdp it %3d mu %.3e pinf %.3e dinf %.3e gap %.3e pobj %.8e dobj %.8e tau %.3e kap %.3e\n", it, mu, pinf, dinf,
            gap, pobj / tau, dobj / tau, tau, kappa);

      if ( !std::isfinite(mu) || !std::isfinite(pinf) || !std::isfinite(dinf) )
      {
         status = HIPSDP_STATUS_NUMERIC;
         break;
      }
      /* ---- preoptimal iterate (warm-start point for the children of a node: relax_sdp.c "warmstartpreoptsol"): the first
       * interior iterate that is feasible to tolerance with a relative gap below preoptgap, as DSDP's monitor captures it
       * (sdpisolver_dsdp.c:323-358) */
      if ( par.preoptgap > 0.0 && !s->pre_valid && pinf <= par.feastol && dabs <= par.feastol
         && gap / (1.0 + 0.5 * fabs(pobj / tau) + 0.5 * fabs(dobj / tau)) < par.preoptgap )
      {
         if ( s->pre_y == NULL )
         {
            HS_CALL( dalloc(&s->pre_y, s->m_alloc > m ? s->m_alloc : m) );
            HS_CALL( dalloc(&s->pre_x, s->q_alloc > q ? s->q_alloc : q) );
         }
         if ( m > 0 ) HS_CALL( hs_copy(st, s->pre_y, s->y, m) );
         if ( q > 0 ) HS_CALL( hs_copy(st, s->pre_x, s->x, q) );
         for (auto& B : s->blk)
         {
            if ( B.Xpre == NULL )
               HS_CALL( dalloc(&B.Xpre, (long long) B.n * B.n) );
            HS_CALL( hs_copy(st, B.Xpre, B.X, (long long) B.n * B.n) );
         }
         s->pre_scale = 1.0 / tau;
         s->pre_valid = true;
      }
      /* ---- termination (mirrors oracle/ipm_ref.py) */
      /* objective limit: with X feasible to tolerance, pobj is a lower bound of the minimisation problem
       * (SCIP_SDPPAR_OBJLIMIT, type_sdpi.h:53); tested first so that a node whose optimum lies above the limit reports it */
      if ( par.objlimit < 1e20 && pinf <= par.feastol && pobj / tau > par.objlimit + par.gaptol )
      {
         status = HIPSDP_STATUS_OBJLIM;
         break;
      }
      if ( pinf <= par.feastol && pabsok && dabs <= par.feastol && gap <= par.gaptol )
      {
         status = HIPSDP_STATUS_OPTIMAL;
         break;
      }
      const bool certzone = (tau < 1e-2 * fmin(1.0, kappa)) || (mu / (tau * tau) > 1e10);
      if ( certzone && want_cert )
      {
         const double hd = sqrt(hs.v[SC_HD2]);
         const double hp = sqrt(hs.v[SC_HP2]);
         const double big = fmax(fabs(dobj), fabs(pobj));
         const bool cand_dunb = dobj < -1e-3 * big;
         const bool cand_dinf = pobj > 1e-3 * big;
         const bool ok_dunb = cand_dunb && hd <= par.infeastol * (-dobj);
         const bool ok_dinf = cand_dinf && hp <= par.infeastol * pobj;
         if ( (ok_dunb || ok_dinf) && (ok_dunb || !cand_dunb || certwait >= 5) && (ok_dinf || !cand_dinf || certwait >= 5) )
         {
            status = (ok_dunb && ok_dinf) ? HIPSDP_STATUS_PDINF : (ok_dunb ? HIPSDP_STATUS_DUNB : HIPSDP_STATUS_DINF);
            break;
         }
         if ( ok_dunb || ok_dinf )
            ++certwait;
      }
      if ( certzone && !want_cert )
      {
         /* the certificate residual was not computed in this pass: redo the (cheap) residual pass with it */
         want_cert = true;
         --it;
         continue;
      }
      want_cert = certzone;
      if ( it == maxiter )
         break;
      if ( mu > 0.9 * lastmu && alpha_last < 1e-2 )
      {
         if ( ++nstall >= stall_lim )
         {
            status = HIPSDP_STATUS_NUMERIC;
            break;
         }
      }
      else
         nstall = 0;
      lastmu = mu;
      /* no progress: the worst scaled violation has not improved by 10 % for 6 iterations (accuracy limit of the problem) */
      if ( !certzone )
      {
         double merit = fmax(fmax(pinf / par.feastol, dabs / par.feastol), gap / par.gaptol);
         if ( par.pabstol > 0.0 )
            merit = fmax(merit, pabs / par.pabstol);
         if ( merit < 0.9 * bestmerit )
         {
            bestmerit = merit;
            sincebest = 0;
         }
         else if ( ++sincebest >= nobest_lim )
         {
            status = HIPSDP_STATUS_NUMERIC;
            break;
         }
      }
      if ( par.timelimit > 0.0 )
      {
         const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
         int over = el > par.timelimit ? 1 : 0;
         if ( s->comm != NULL )
         {
            /* the clocks of the ranks differ: rank 0's decision counts everywhere */
            HS_HIP( hipMemcpyAsync(s->flags + 7, &over, sizeof(int), hipMemcpyHostToDevice, st) );
            HS_CALL( hs_bcast_ints(s->comm, s->flags + 7, 1, st) );
            HS_HIP( hipMemcpyAsync(&over, s->flags + 7, sizeof(int), hipMemcpyDeviceToHost, st) );
            HS_HIP( hipStreamSynchronize(st) );
         }
         if ( over )
         {
            status = HIPSDP_STATUS_TIMELIM;
            break;
         }
      }

      /* ---- factorizations (the factors of an accepted step are re-used: they were computed by its Cholesky check) */
      phase_mark(s, PH_FACTOR);
      const bool defer_join = defer_possible() && !(identity_start && it == 0) && !xchain_queued;
      if ( !zchain_queued )
      {
         if ( !(setf && m <= 64) )
            HS_HIP( hipMemsetAsync(s->flags, 0, 8 * sizeof(int), st) );
         HS_CALL( fork2(s) );
      }
      for (auto& B : s->blk)
      {
         const int n = B.n;
         const long long n2 = (long long) n * n;
         hipStream_t st2 = s->use2 ? s->stream2 : s->stream;
         if ( n <= 64 )
         {
            /* single-block factors: one launch per matrix yields L (zero upper), inv(L) as n x n and, for n <= 32, the inverse
             * of Z; after an accepted step they already exist (the step's Cholesky check produced them) */
            if ( !(factors_valid && B.derived_valid) )
            {
               HS_CALL( hs_potrf_small_ext(st2, n, B.Lz, B.dinvz, s->flags + 0, B.Z, NULL, 0.0, NULL, B.LzInv, n <= 32 ? B.Zinv : NULL, setf) );
               HS_CALL( hs_potrf_small_ext(st, n, B.Lx, B.dinvx, s->flags + 1, B.X, NULL, 0.0, NULL, B.LxInv, NULL, setf) );
               B.derived_valid = true;
            }
            if ( n > 32 )
            {
               HS_CALL( gemm_on(st2, s->gws2, s->gws_len, HS_MC, HS_MC, n, n, n, 1.0, B.LzInv, n, B.LzInv, n, 0.0, B.Zinv, n, HS_GEMM_LOWER) );
               HS_CALL( hs_mirror_lower(st2, B.Zinv, n, n) );
            }
            continue;
         }
         /* Z chain on the second queue (already under way when it was started at the top of the iteration) */
         if ( !zchain_queued )
         {
         if ( !factors_valid )
         {
            HS_CALL( hs_copy(st2, B.Lz, B.Z, n2) );
            HS_CALL( hs_potrf(st2, n, B.Lz, B.dinvz, s->flags + 0, NULL) );
         }
         HS_CALL( hs_trtri(st2, n, B.Lz, B.dinvz, B.LzInv, B.T2) );
         HS_CALL( gemm_on(st2, s->gws2, s->gws_len, HS_MC, HS_MC, n, n, n, 1.0, B.LzInv, n, B.LzInv, n, 0.0, B.Zinv, n, HS_GEMM_LOWER) );
         HS_CALL( hs_mirror_lower(st2, B.Zinv, n, n) );
         }
         /* X chain on the first (already queued behind the read-back when the Z chain was started at the top of the iteration) */
         if ( !xchain_queued )
         {
         if ( !factors_valid )
         {
            HS_CALL( hs_copy(st, B.Lx, B.X, n2) );
            HS_CALL( hs_potrf(st, n, B.Lx, B.dinvx, s->flags + 1, NULL) );
         }
         HS_CALL( hs_zero_upper(st, B.Lx, n) );
         if ( !defer_join )
            HS_CALL( hs_trtri(st, n, B.Lx, B.dinvx, B.LxInv, B.T1) );
         }
      }
      if ( defer_join )
      {
         /* (the inverse factor of X follows the first product into the second queue; the event is recorded behind it) */
         s->sws.after_g1 = after_g1_fn;
         s->sws.after_g1_arg = &after_g1_arg;
      }
      else
         HS_CALL( join2(s) );
      zchain_queued = false;
      xchain_queued = false;

      /* ---- Schur complement (extended by the constant matrix as "variable 0") */
      phase_mark(s, PH_SCHUR);
      bool gram_only_used = false;
      hs_comm_phase(0);
      const bool clk_this = s->clk_on && s->clk_buf != NULL && s->clk_stream != NULL && s->clk_n < CLK_MAX_ASSEMBLIES;
      HS_HIP( hipEventRecord(s->ev0, st) );
      if ( clk_this )
      {
         /* the counting thread starts when the assembly may start (behind the event) on its own queue */
         HS_HIP( hipMemsetAsync(s->clk_buf + 4 * s->clk_n, 0, 4 * sizeof(unsigned long long), st) );
         HS_HIP( hipEventRecord(s->ev0, st) );
         HS_HIP( hipStreamWaitEvent(s->clk_stream, s->ev0, 0) );
         hipLaunchKernelGGL(k_clock_window, dim3(1), dim3(1), 0, s->clk_stream, s->clk_buf + 4 * s->clk_n);
         HS_LAUNCH_CHECK();
      }
      const double mfma_flops_before = hs_mfma_flops_total();
      bool schur_small = false;
      if ( s->comm == NULL && !s->shardA && !s->schur_mode_rows && !s->schur_mode_forced && K > 0 )
      {
         /* B&B-sized problems: one launch for the whole extended Schur matrix and the copies the factorization needs */
         std::vector<int> bn;
         std::vector<const double*> bA, bX, bZ;
         for (auto& B : s->blk)
         {
            bn.push_back(B.n); bA.push_back(B.A); bX.push_back(B.X); bZ.push_back(B.Zinv);
         }
         bool anysp = false;
         for (auto& B : s->blk) anysp = anysp || B.sparse;
         const int rs = anysp ? 0 : hs_schur_small(st, m1, K, bn.data(), bA.data(), bX.data(), bZ.data(), q, s->Dext, s->x, s->z, s->Mx,
            m > 0 ? s->Lm : NULL, s->dya);
         if ( rs < 0 )
            return -rs;
         schur_small = (rs == 1);
      }
      if ( schur_small )
      {
         /* nothing else to assemble */
      }
      else
      {
      HS_CALL( hs_fill(st, s->Mx, (long long) m1 * m1, 0.0) );
      if ( s->shardA )
      {
         /* matrices sharded by variable: W_j where A_j lives, all-to-all of the row ranges, partial Gram matrices summed */
         for (auto& B : s->blk)
            if ( !B.sparse )
               HS_CALL( hs_schur_Wvar_all(st, s->stream2, s->comm, s->rank, s->nranks, m1, B.n, B.A, B.Lx, B.LzInv, s->Mx, &s->sws, s->var_cw,
                     s->var_overlap ? 1 : 0) );
         HS_CALL( hs_allreduce_sum(s->comm, s->Mx, (long long) m1 * m1, st) );
      }
      else if ( s->schur_mode_cols )
      {
         /* W formulation in column slices: this rank's slices of every W_j, partial matrices summed over the ranks */
         const int S = s->schur_sim_shards;
         const int per = S / (s->comm != NULL ? s->nranks : 1);
         const int g0 = (s->comm != NULL ? s->rank : 0) * per;
         for (int g = g0; g < g0 + per; ++g)
            for (auto& B : s->blk)
            {
               if ( B.sparse )
                  continue;
               int c0, cw;
               hs_shard_cols(m1, B.n, S, g, &c0, &cw);
               if ( identity_start && it == 0 )
               {
                  /* (all column slices of this rank at once, straight from A; several ranks: the n^2 entries of the matrices in equal
                   * ranges, the partial Gram matrices are summed below like the slices' ones) */
                  if ( g == g0 )
                  {
                     const long long n2 = (long long) B.n * B.n;
                     const int G = s->comm != NULL ? s->nranks : 1, r = s->comm != NULL ? s->rank : 0;
                     const long long k0 = ((n2 * r / G) / 8) * 8, k1 = r == G - 1 ? n2 : ((n2 * (r + 1) / G) / 8) * 8;
                     HS_CALL( hs_schur_W_identity_range(st, m1, B.n, B.A, k0, k1, s->Mx, &s->sws) );
                  }
                  gram_only_used = true;
                  continue;
               }
               HS_CALL( hs_schur_Wcols(st, m1, B.n, B.A, B.Lx, B.LzInv, s->Mx, &s->sws, c0, cw) );
            }
         if ( s->comm != NULL )
            HS_CALL( hs_allreduce_sum(s->comm, s->Mx, (long long) m1 * m1, st) );
      }
      else if ( s->comm != NULL || s->schur_mode_rows )
      {
         /* sharded assembly: this rank computes two row chunks of the upper triangle, the chunks are all-gathered */
         int c, b1, b2;
         hs_shard_rows(m1, s->nranks, s->rank, &c, &b1, &b2);
         for (auto& B : s->blk)
         {
            if ( B.sparse )
               continue;
            HS_CALL( hs_schur_Urows(st, m1, B.n, B.A, B.X, B.Zinv, s->Mx, &s->sws, b1, b1 + c) );
            HS_CALL( hs_schur_Urows(st, m1, B.n, B.A, B.X, B.Zinv, s->Mx, &s->sws, b2, b2 + c) );
         }
         if ( s->comm != NULL )
         {
            const long long cnt = (long long) c * m1;
            /* chunks 0 .. G-1 sit at their final place; chunks G .. 2G-1 are owned in reverse rank order */
            HS_CALL( hs_allgather_inplace(s->comm, s->Mx, cnt, s->rank, st) );
            HS_CALL( hs_allgather(s->comm, s->Mx + (long long) b2 * m1, s->Mgather, cnt, st) );
            for (int r = 0; r < s->nranks; ++r)
            {
               const long long dstrow = (long long) (2 * s->nranks - 1 - r) * c;
               if ( dstrow >= m1 )
                  continue;
               long long rowsleft = m1 - dstrow;
               if ( rowsleft > c ) rowsleft = c;
               HS_CALL( hs_copy(st, s->Mx + dstrow * m1, s->Mgather + (long long) r * cnt, rowsleft * m1) );
            }
         }
         HS_CALL( hs_mirror_upper(st, s->Mx, m1, m1) );
      }
      else
      {
         for (auto& B : s->blk)
         {
            if ( B.sparse )
               continue;
            if ( identity_start && it == 0 )
            {
               HS_CALL( hs_schur_W_identity(st, m1, B.n, B.A, s->Mx, &s->sws) );
               gram_only_used = true;
            }
            else if ( s->schur_mode_U )
               HS_CALL( hs_schur_U(st, m1, B.n, B.A, B.X, B.Zinv, s->Mx, &s->sws, 0, m1) );
            else
               HS_CALL( hs_schur_W(st, m1, B.n, B.A, B.Lx, B.LzInv, s->Mx, &s->sws) );
         }
      }
      /* blocks in sparse mode (csrc/sparse.hip), on every rank alike, behind the exchange of the sharded forms: column 0 from
       * U_0 = X A_0 Zinv (A_0 is dense) and one gather pass, the pairs of variables from the pair formula over their nonzeros */
      for (auto& B : s->blk)
      {
         if ( !B.sparse )
            continue;
         const int n = B.n;
         HS_CALL( gemm(s, HS_KC, HS_MC, n, n, n, 1.0, B.X, n, B.A0, n, 0.0, B.T1, n) );
         HS_CALL( gemm(s, HS_KC, HS_MC, n, n, n, 1.0, B.T1, n, B.Zinv, n, 0.0, B.W, n) );
         HS_CALL( pass_A(s, B, B.W, s->tmpe) );
         hipLaunchKernelGGL(k_add_col0, g1d(m1), dim3(256), 0, st, m1, s->tmpe, s->Mx);
         HS_LAUNCH_CHECK();
         HS_CALL( hs_sp_schur(st, B.sp, B.X, B.Zinv, s->Mx) );
      }
      if ( q > 0 )
      {
         HS_CALL( hs_lp_scale_rows(st, q, m1, s->x, s->z, s->Dext, s->Slp) );
         HS_CALL( gemm(s, HS_MC, HS_MC, m1, m1, q, 1.0, s->Dext, m1, s->Slp, m1, 1.0, s->Mx, m1, HS_GEMM_LOWER) );
      }
      HS_CALL( hs_mirror_lower(st, s->Mx, m1, m1) );
      }
      if ( defer_join )
      {
         /* (the first queue has waited for the event inside hs_schur_W; once more for whoever follows, and the hooks are taken out) */
         s->sws.ev_g2 = NULL;
         s->sws.after_g1 = NULL;
         HS_CALL( join2(s) );
      }
      HS_HIP( hipEventRecord(s->ev1, st) );
      if ( clk_this )
      {
         hipLaunchKernelGGL(k_clock_stop, dim3(1), dim3(1), 0, st, s->clk_buf + 4 * s->clk_n);
         HS_LAUNCH_CHECK();
         ++s->clk_n;
      }
      info->schur_flops_executed += hs_mfma_flops_total() - mfma_flops_before;
      hs_comm_phase(2);
      phase_mark(s, PH_MSOLVE);
      bool predH_queued = false, predH_joined = false, split_dz = false;
      if ( s->use2 && !no_overlap && s->comm == NULL && !small_problem(s) && m > 0 )
      {
         HS_CALL( fork2(s) );
         {
            QueueSwap sw(s);
            HS_CALL( direction(s, 0.0, 1.0, mu, rg, false, 0.0, 1) );
         }
         predH_queued = true;
      }
      if ( m > 0 )
      {
         if ( !schur_small )
         {
            HS_HIP( hipMemcpy2DAsync(s->Lm, (size_t) m * sizeof(double), s->Mx + m1 + 1, (size_t) m1 * sizeof(double),
                  (size_t) m * sizeof(double), (size_t) m, hipMemcpyDeviceToDevice, st) );
            /* original diagonal of M for the semidefinite pivot rule */
            HS_HIP( hipMemcpy2DAsync(s->dya, sizeof(double), s->Mx + m1 + 1, (size_t) (m1 + 1) * sizeof(double), sizeof(double), (size_t) m,
                  hipMemcpyDeviceToDevice, st) );
         }
         HS_CALL( hs_potrf_psd(st, m, s->Lm, s->dinvm, s->flags + 2, s->dya, s->regmask, setf) );
         if ( m <= 64 )
         {
            hipLaunchKernelGGL(k_solve2_small, dim3(1), dim3(128), 0, st, m, s->Mx, s->b, s->dinvm, s->Lm, s->rhs2, s->u2, s->wt, hs_small_solve_by_substitution());
            HS_LAUNCH_CHECK();
         }
         else
         {
            hipLaunchKernelGGL(k_rhs2, g1d(m), dim3(256), 0, st, m, s->Mx, s->b, s->rhs2);
            HS_LAUNCH_CHECK();
            if ( predH_queued )
            {
               /* the predictor's right-hand side h (formed on the second queue while M was factored) sits behind g and b: one
                * solve with three right-hand sides instead of a pair and, later, a single one */
               HS_CALL( join2(s) );
               predH_joined = true;
               HS_CALL( hs_trsv_sync(st, m, s->Lm, s->dinvm, 3, s->rhs2, m, solve_mode(s), s->trsv_ws, &s->trsv_epoch) );
            }
            else
               HS_CALL( hs_trsv_sync(st, m, s->Lm, s->dinvm, 2, s->rhs2, m, solve_mode(s), s->trsv_ws, &s->trsv_epoch) );
         }
      }
      if ( m == 0 || m > 64 )
      {
         hipLaunchKernelGGL(k_after_solve2, g1d(m1), dim3(256), 0, st, m, s->rhs2, s->u2, s->wt);
         HS_LAUNCH_CHECK();
      }
      /* B_k = A_0 - sum w_i A_i ; beta = c - D w ; S0 ; b^T M^-1 b */
      hs_red_batch_begin(st);
      HS_CALL( hs_fill_scalar(st, s->sc + SC_S0, 0.0) );
      /* dZ = A^T([-dtau; u1 - u2 dtau]) + eta Rd is linear in dtau: A^T([0; u1]) - dtau A^T([1; u2]) + eta Rd.  u2 belongs to the
       * iteration, the predictor's u1 has just been solved with it, so B = A^T(wt), P2 = A^T([1; u2]) and the predictor's
       * P1 = A^T([0; u1]) come out of ONE sweep over A (three coefficient vectors), the corrector needs one more sweep for its P1:
       * two sweeps per iteration where B, the predictor's dZ and the corrector's dZ took three - and the dependent chain
       * B pass -> dtau -> dZ pass loses a link.  Single GPU, packed copy present, predictor solved early (else the old form). */
      bool split_ok = predH_joined && s->comm == NULL && getenv("HIPSDP_NO_SPLIT_DZ") == NULL;
      for (auto& B : s->blk)
         if ( B.Apk == NULL )
            split_ok = false;
      if ( split_ok )
      {
         HS_CALL( hs_make_ext(st, m, 1.0, 1.0, s->u2, s->cvec) );
         HS_CALL( hs_make_ext(st, m, 0.0, 1.0, s->u1, s->cvec + m1) );
         HS_LAUNCH_CHECK();
      }
      split_dz = split_ok;
      for (auto& B : s->blk)
      {
         const int n = B.n;
         const long long n2 = (long long) n * n;
         bool swept = false;
         if ( split_ok )
         {
            if ( B.P2 == NULL )
            {
               HS_CALL( dalloc(&B.P2, n2) );
               HS_CALL( dalloc(&B.pk3, 3 * B.Lp) );
            }
            const int r3 = hs_gemv_t3(st, m1, B.Lp, B.Apk, B.Lp, s->wt, s->cvec, s->cvec + m1, B.pk3, B.pk3 + B.Lp, B.pk3 + 2 * B.Lp);
            if ( r3 < 0 )
               return -r3;
            if ( r3 == 1 )
            {
               HS_CALL( hs_unpack_sym(st, n, B.pk3, 0.0, NULL, B.B) );
               HS_CALL( hs_unpack_sym(st, n, B.pk3 + B.Lp, 0.0, NULL, B.P2) );
               HS_CALL( hs_unpack_sym(st, n, B.pk3 + 2 * B.Lp, 0.0, NULL, B.dZ) );
               swept = true;
            }
            else
            {
               HS_CALL( pass_AT(s, B, s->cvec, 0.0, NULL, B.P2) );
               HS_CALL( pass_AT(s, B, s->cvec + m1, 0.0, NULL, B.dZ) );
            }
         }
         if ( !swept )
            HS_CALL( pass_AT(s, B, s->wt, 0.0, NULL, B.B) );
         HS_CALL( gemm(s, HS_KC, HS_MC, n, n, n, 1.0, B.X, n, B.B, n, 0.0, B.T1, n) );
         HS_CALL( gemm(s, HS_KC, HS_MC, n, n, n, 1.0, B.T1, n, B.Zinv, n, 0.0, B.W, n) );
         HS_CALL( hs_dot(st, n2, B.B, B.W, s->sc + SC_S0, 1, s->red_ws) );
      }
      if ( q > 0 )
      {
         const double* v = s->wt;
         HS_CALL( hs_gemv_n(st, q, m1, s->Dext, m1, 1, &v, s->beta, q, s->gemv_ws, s->gemv_ws_len) );
         HS_CALL( hs_lp_s0(st, q, s->x, s->z, s->beta, s->sc + SC_S0, 1, s->red_ws) );
      }
      HS_CALL( hs_dot(st, m, s->b, s->rhs2 + m, s->sc + SC_BUB, 0, s->red_ws) );
      HS_CALL( hs_red_batch_end() );

      /* ---- predictor */
      phase_mark(s, PH_PRED);
      if ( predH_queued && !predH_joined )
         HS_CALL( join2(s) );
      HS_CALL( direction(s, 0.0, 1.0, mu, rg, false, 0.0, predH_queued ? 2 : 0, predH_joined, split_dz ? 2 : 0) );
      HS_CALL( steplen_enqueue(s) );
      HS_CALL( read_scalars(s, hs, hflags) );
      {
         float ms = 0.f;
         if ( hipEventElapsedTime(&ms, s->ev0, s->ev1) == hipSuccess )
            schur_ms += ms;
         info->schur_calls++;
         /* (the assembly at the cold start is the Gram product alone: the count of SURVEY.md 8(d) without its n^3 part) */
         const bool gram_only = gram_only_used;
         for (auto& B : s->blk)
            info->schur_flops += (gram_only && !B.sparse ? 0.0 : 4.0 * m1 * (double) B.n * B.n * B.n) + (double) m1 * m1 * (double) B.n * B.n;
      }
      if ( hflags[0] != 0 || hflags[1] != 0 || hflags[2] != 0 )
      {
         if ( par.verbose )
            printf("hipsdp: Cholesky failure flags Z=%d X=%d M=%d\n", hflags[0], hflags[1], hflags[2]);
         status = HIPSDP_STATUS_NUMERIC;
         break;
      }
      const double aa = fmin(1.0, steplen_host(s, hs));
      const double dta = hs.v[SC_DTAU], dka = hs.v[SC_DKAPPA];
      if ( !std::isfinite(aa) || !std::isfinite(dta) )
      {
         if ( par.verbose )
            printf("hipsdp: predictor not finite (alpha %g dtau %g)\n", aa, dta);
         status = HIPSDP_STATUS_NUMERIC;
         break;
      }
      double sigma = (1.0 - aa) * (1.0 - aa) * (1.0 - aa);
      sigma = fmin(1.0, fmax(sigma_floor, sigma));
      const double eta = 1.0 - sigma;
      phase_mark(s, PH_CORR);
      /* second-order terms from the predictor */
      for (auto& B : s->blk)
      {
         const int n = B.n;
         const long long n2 = (long long) n * n;
         HS_CALL( gemm(s, HS_KC, HS_MC, n, n, n, 1.0, B.dX, n, B.dZ, n, 0.0, B.E, n) );
      }
      {
         /* (B&B-sized: the LP term and the whole corrector direction are one launch) */
         BatchRegion region(s, batch_regions(s));
         if ( q > 0 )
            HS_CALL( hs_vec_mul(st, q, s->dx, s->dz, s->elp) );

         /* ---- corrector */
         HS_CALL( direction(s, sigma, eta, mu, rg, true, dta * dka, 0, false, split_dz ? 1 : 0) );
         HS_CALL( region.close() );
      }
      HS_CALL( steplen_enqueue(s) );
      HS_CALL( read_scalars(s, hs, NULL) );
      const double amax = steplen_host(s, hs);
      double alpha = fmin(1.0, gamma_eff * amax);
      const double dt = hs.v[SC_DTAU], dk = hs.v[SC_DKAPPA];
      if ( !std::isfinite(alpha) || !std::isfinite(dt) || !std::isfinite(dk) )
      {
         if ( par.verbose )
            printf("hipsdp: corrector not finite (alpha %g dtau %g dkappa %g)\n", alpha, dt, dk);
         status = HIPSDP_STATUS_NUMERIC;
         break;
      }

      /* ---- update, with a Cholesky check of the new X and Z (the Lanczos bound is an estimate) */
      phase_mark(s, PH_UPDATE);
      for (auto& B : s->blk)
      {
         const long long n2 = (long long) B.n * B.n;
         if ( B.n <= 64 )
            continue;            /* single-block factorization: the trial iterate is written to Xs / Zs and swapped in */
         HS_CALL( hs_copy(st, B.Xs, B.X, n2) );
         HS_CALL( hs_copy(st, B.Zs, B.Z, n2) );
      }
      if ( small_all && K > 0 )
      {
         /* small problems: the whole step is applied optimistically and the residual phase of the next iteration is enqueued
          * behind the Cholesky check, so that ONE read-back returns the flags and the new residuals; a failed check takes the
          * step back (pointer swaps, a correcting axpy) and halves it */
         const double tau0 = s->tau, kappa0 = s->kappa;
         double applied = 0.0;
         bool accepted = false;
         for (int attempt = 0; attempt < 8; ++attempt)
         {
            if ( !setf )
               HS_HIP( hipMemsetAsync(s->flags, 0, 8 * sizeof(int), st) );
            for (auto& B : s->blk)
            {
               const int n = B.n;
               {
                  /* the Cholesky checks of the trial X and Z: one launch of two workgroups */
                  double* L2[2] = {B.Lx, B.Lz};
                  double* di2[2] = {B.dinvx, B.dinvz};
                  int* fl2[2] = {s->flags + 1, s->flags + 0};
                  const double* ba2[2] = {B.X, B.Z};
                  const double* dr2[2] = {B.dX, B.dZ};
                  double* mo2[2] = {B.Xs, B.Zs};
                  double* li2[2] = {B.LxInv, B.LzInv};
                  double* gr2[2] = {NULL, n <= 32 ? B.Zinv : NULL};
                  HS_CALL( hs_potrf_small_ext_pair(st, n, L2, di2, fl2, ba2, dr2, alpha, mo2, li2, gr2, setf) );
               }
               std::swap(B.X, B.Xs);
               std::swap(B.Z, B.Zs);
            }
            if ( batch_regions(s) )
               hs_red_batch_hold(st);           /* (the read-back below ends the region) */
            HS_CALL( hs_axpy3(st, alpha - applied, m, s->dy, s->y, q, s->dx, s->x, q, s->dz, s->z) );
            applied = alpha;
            s->tau = tau0 + alpha * dt;
            s->kappa = kappa0 + alpha * dk;
            HS_CALL( enqueue_residuals() );
            HS_CALL( read_scalars(s, hs, hflags) );
            if ( hflags[0] == 0 && hflags[1] == 0 )
            {
               accepted = true;
               break;
            }
            for (auto& B : s->blk)
            {
               std::swap(B.X, B.Xs);
               std::swap(B.Z, B.Zs);
            }
            alpha *= 0.5;
            info->chol_fail++;
         }
         if ( !accepted )
         {
            if ( par.verbose )
               printf("hipsdp: no step length down to %g keeps X and Z positive definite\n", alpha);
            HS_CALL( hs_axpy3(st, -applied, m, s->dy, s->y, q, s->dx, s->x, q, s->dz, s->z) );
            s->tau = tau0;
            s->kappa = kappa0;
            for (auto& B : s->blk)
               B.derived_valid = false;
            status = HIPSDP_STATUS_NUMERIC;
            break;
         }
         for (auto& B : s->blk)
            B.derived_valid = true;
         factors_valid = true;
         residuals_ready = true;
         alpha_last = alpha;
      }
      else
      {
         /* Round 6: one rank - the rest of the step (y, x, z, tau, kappa) is applied optimistically behind the Cholesky check of the
          * new X and Z, and the residual pass of the NEXT iterate (the sweep A(X), the scaled dual residual, the reductions) and the
          * chains of its factorization phase are queued behind that: ONE read-back returns the check's flags and the termination
          * scalars of the next iteration, where there were two with an idle device between them (profiles/r06_a_iter_sequence.txt:
          * 50 us waiting for the flags, then 200 us of small kernels arriving one launch at a time).  A failed check - rare: the
          * step lengths are Lanczos estimates with a safety factor - halves the step as before; the optimistic part is taken back by
          * a correcting axpy and the dual residual is recomputed from scratch at the next pass (its recurrence was applied in place). */
         const bool optimistic = (s->comm == NULL) && getenv("HIPSDP_NO_OPTIMISTIC") == NULL;
         /* (test hook, tests/test_gpu_ipm.py: HIPSDP_TEST_OVERSTEP=<factor> lengthens the step of iteration 2 beyond the boundary of the
          * cone, so that the rarely taken road - failed check, step halved, optimistic part taken back - is driven) */
         if ( it == 2 && getenv("HIPSDP_TEST_OVERSTEP") != NULL )
            alpha = amax * atof(getenv("HIPSDP_TEST_OVERSTEP"));
         const double tau0 = s->tau, kappa0 = s->kappa;
         double applied = 0.0;
         for (int attempt = 0; attempt < 8; ++attempt)
         {
            HS_HIP( hipMemsetAsync(s->flags, 0, 8 * sizeof(int), st) );
            HS_CALL( fork2(s) );
            for (auto& B : s->blk)
            {
               const int n = B.n;
               const long long n2 = (long long) n * n;
               hipStream_t st2 = s->use2 ? s->stream2 : s->stream;
               if ( n <= 64 )
               {
                  HS_CALL( hs_potrf_small_ext(st, n, B.Lx, B.dinvx, s->flags + 1, B.X, B.dX, alpha, B.Xs, B.LxInv, NULL, setf) );
                  HS_CALL( hs_potrf_small_ext(st2, n, B.Lz, B.dinvz, s->flags + 0, B.Z, B.dZ, alpha, B.Zs, B.LzInv, n <= 32 ? B.Zinv : NULL, setf) );
                  continue;
               }
               HS_CALL( hs_scale_add(st, n2, alpha, B.dX, 1.0, B.Xs, B.X) );
               HS_CALL( hs_copy(st, B.Lx, B.X, n2) );
               HS_CALL( hs_potrf(st, n, B.Lx, B.dinvx, s->flags + 1, NULL) );
               HS_CALL( hs_scale_add(st2, n2, alpha, B.dZ, 1.0, B.Zs, B.Z) );
               HS_CALL( hs_copy(st2, B.Lz, B.Z, n2) );
               HS_CALL( hs_potrf(st2, n, B.Lz, B.dinvz, s->flags + 0, NULL) );
            }
            HS_CALL( join2(s) );
            if ( K == 0 )
               break;
            if ( optimistic )
            {
               /* (blocks of at most 64 rows: their trial iterate sits in Xs / Zs - swapped in now, back on a failure) */
               for (auto& B : s->blk)
                  if ( B.n <= 64 )
                  {
                     std::swap(B.X, B.Xs);
                     std::swap(B.Z, B.Zs);
                  }
               HS_CALL( hs_axpy3(st, alpha - applied, m, s->dy, s->y, q, s->dx, s->x, q, s->dz, s->z) );
               applied = alpha;
               s->tau = tau0 + alpha * dt;
               s->kappa = kappa0 + alpha * dk;
               factors_valid = true;                    /* (what the chains below test: the check has produced the factors) */
               if ( s->use2 && !no_overlap && !zchain_queued )
               {
                  HS_CALL( fork2(s) );
                  HS_CALL( enqueue_z_chains() );
                  zchain_queued = true;
               }
               if ( attempt == 0 )
               {
                  rd_have = !small_problem(s);
                  rd_pending = rd_have;
                  rd_scale = 1.0 - alpha * eta;
               }
               HS_CALL( enqueue_residuals() );
               HS_CALL( read_scalars(s, hs, hflags, (zchain_queued && s->comm == NULL && !defer_possible()) ? &enqueue_x_chains : NULL) );
               if ( hflags[0] == 0 && hflags[1] == 0 )
                  break;
               /* the step was too long: everything queued behind the check ran on an iterate that is given up */
               for (auto& B : s->blk)
                  if ( B.n <= 64 )
                  {
                     std::swap(B.X, B.Xs);
                     std::swap(B.Z, B.Zs);
                  }
               factors_valid = false;
               zchain_queued = false;
               xchain_queued = false;
               rd_have = false;
               rd_pending = false;
               alpha *= 0.5;
               info->chol_fail++;
               continue;
            }
            if ( s->comm != NULL )
               HS_CALL( hs_bcast_ints(s->comm, s->flags, 3, st) );
            if ( s->comm == NULL && s->use_publish )
               HS_CALL( publish_and_wait(s, s->nsc, 2) );
            else
            {
               HS_HIP( hipMemcpyAsync(s->hsc + s->nsc, s->flags, 4 * sizeof(int), hipMemcpyDeviceToHost, st) );
               HS_HIP( hipStreamSynchronize(st) );
            }
            memcpy(hflags, s->hsc + s->nsc, 3 * sizeof(int));
            if ( hflags[0] == 0 && hflags[1] == 0 )
               break;
            alpha *= 0.5;
            info->chol_fail++;
         }
         if ( K > 0 && (hflags[0] != 0 || hflags[1] != 0) )
         {
            if ( par.verbose )
               printf("hipsdp: no step length down to %g keeps X and Z positive definite\n", alpha);
            if ( optimistic && applied != 0.0 )
            {
               HS_CALL( hs_axpy3(st, -applied, m, s->dy, s->y, q, s->dx, s->x, q, s->dz, s->z) );
               s->tau = tau0;
               s->kappa = kappa0;
            }
            for (auto& B : s->blk)
            {
               const long long n2 = (long long) B.n * B.n;
               B.derived_valid = false;
               if ( B.n <= 64 )
                  continue;         /* X, Z were never overwritten */
               HS_CALL( hs_copy(st, B.X, B.Xs, n2) );
               HS_CALL( hs_copy(st, B.Z, B.Zs, n2) );
            }
            status = HIPSDP_STATUS_NUMERIC;
            break;
         }
         for (auto& B : s->blk)
         {
            if ( B.n <= 64 )
            {
               /* accept the trial iterate: its factors, inverse factors (and inverse) are those of the fused factorization */
               if ( !optimistic )
               {
                  std::swap(B.X, B.Xs);
                  std::swap(B.Z, B.Zs);
               }
               B.derived_valid = true;
            }
         }
         factors_valid = (K > 0);
         alpha_last = alpha;
         if ( optimistic && K > 0 )
            residuals_ready = true;                /* (applied, queued and read above) */
         else
         {
            HS_CALL( hs_axpy3(st, alpha, m, s->dy, s->y, q, s->dx, s->x, q, s->dz, s->z) );
            s->tau += alpha * dt;
            s->kappa += alpha * dk;
            /* the residual the next iteration starts from: (1 - alpha eta) times the one this iteration used (general path only:
             * the single-launch kernels of small problems recompute it inside the fused launches) */
            rd_have = !small_problem(s);
            rd_pending = rd_have;
            rd_scale = 1.0 - alpha * eta;
         }
      }
   }

   phase_mark(s, -1);
   /* a break above may leave the Z chain of the abandoned iteration running on the second queue (it is started at the top of an
    * iteration, before the termination decision): it writes B.Lz / B.LzInv / B.Zinv / B.T2 / gws2, so the solve is over - and
    * its buffers may be re-shaped, freed or read - only when that queue has drained as well */
   if ( zchain_queued )
      HS_CALL( join2(s) );
   HS_HIP( hipStreamSynchronize(st) );
   if ( s->clk_on && s->clk_n > 0 )
   {
      if ( s->clk_stream != NULL )
         HS_HIP( hipStreamSynchronize(s->clk_stream) );
      std::vector<unsigned long long> h((size_t) 4 * s->clk_n);
      HS_HIP( hipMemcpy(h.data(), s->clk_buf, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) );
      double dc = 0.0, dw = 0.0;
      for (int a = 0; a < s->clk_n; ++a)
      {
         dc += (double) h[4 * a + 2];
         dw += (double) h[4 * a + 3];
      }
      s->clk_ghz = dw > 0.0 ? dc / (dw * 10.0) : 0.0;
   }
   phase_finish(s);
   s->last_status = status;
   s->solved = true;
   if ( status == HIPSDP_STATUS_DINF || status == HIPSDP_STATUS_DUNB || status == HIPSDP_STATUS_PDINF )
      s->sol_scale = 1.0 / fmax(fmax(fabs(dobj), fabs(pobj)), 1e-300);
   else
      s->sol_scale = 1.0 / s->tau;
   info->status = status;
   info->iterations = it;
   info->pobj = pobj * s->sol_scale;
   info->dobj = dobj * s->sol_scale;
   info->pinf = pinf;
   info->dinf = dinf;
   info->dabs = dabs;
   info->gap = gap;
   info->mu = mu;
   info->tau = s->tau;
   info->kappa = s->kappa;
   info->settings_used = settings;
   info->schur_seconds = schur_ms * 1e-3;
   info->solve_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
   return HIPSDP_OK;
}

extern "C" long long hipsdp_solve1_fallbacks(void)
{
   return __sync_add_and_fetch(&g_solve1_fallbacks, 0);
}

extern "C" long long hipsdp_solve1_fallbacks_warm(void)
{
   return __sync_add_and_fetch(&g_solve1_fallbacks_warm, 0);
}

extern "C" long long hipsdp_solve1_solves(void)
{
   return __sync_add_and_fetch(&g_solve1_solves, 0);
}

extern "C" int hipsdp_solve1_debug_counts(unsigned int* counts2)
{
   if ( counts2 == NULL )
      return 0;
   return hs_solve1_debug_counts(counts2);
}

extern "C" int hipsdp_solve_path(hipsdp_solver* s)
{
   return (s != NULL && s->solved) ? s->s1_last : 0;
}

extern "C" int hipsdp_solve1_trace(hipsdp_solver* s, double* out64, int maxrows, double* hist)
{
   if ( s == NULL || s->s1_host == NULL )
      return HIPSDP_ERR_ARG;
   if ( out64 != NULL )
      memcpy(out64, s->s1_host, HS_S1_OUT_DOUBLES * sizeof(double));
   if ( hist != NULL && maxrows > 0 )
      memcpy(hist, s->s1_host + HS_S1_OUT_DOUBLES + 8, (size_t) (maxrows < S1_HIST_MAX ? maxrows : S1_HIST_MAX) * 16 * sizeof(double));
   return HIPSDP_OK;
}

/* ---- readback ------------------------------------------------------------------------------------------------------ */

static int read_scaled(hipsdp_solver* s, const double* d, long long n, double scale, double* out)
{
   if ( n <= 0 ) return HIPSDP_OK;
   if ( s->s1_sol_host && s->s1_host != NULL && (d == s->y || d == s->x || d == s->z) )
   {
      /* the one-launch solve left y, x, z in pinned memory as well */
      const double* h = s->s1_host + S1_SOL_OFF + (d == s->y ? 0 : (d == s->x ? 128 : 128 + 4096));
      for (long long i = 0; i < n; ++i)
         out[i] = h[i] * scale;
      return HIPSDP_OK;
   }
   HS_HIP( hipSetDevice(s->device) );
   HS_CALL( stage_sync(s) );
   HS_HIP( hipMemcpy(out, d, (size_t) n * sizeof(double), hipMemcpyDeviceToHost) );
   for (long long i = 0; i < n; ++i)
      out[i] *= scale;
   return HIPSDP_OK;
}

extern "C" int hipsdp_get_y(hipsdp_solver* s, double* y)
{
   if ( s == NULL || !s->solved ) return HIPSDP_ERR_ARG;
   return read_scaled(s, s->y, s->m, s->sol_scale, y);
}

extern "C" int hipsdp_get_X(hipsdp_solver* s, int block, double* X)
{
   if ( s == NULL || !s->solved || block < 0 || block >= (int) s->blk.size() ) return HIPSDP_ERR_ARG;
   return read_scaled(s, s->blk[block].X, (long long) s->blk[block].n * s->blk[block].n, s->sol_scale, X);
}

extern "C" int hipsdp_get_Z(hipsdp_solver* s, int block, double* Z)
{
   if ( s == NULL || !s->solved || block < 0 || block >= (int) s->blk.size() ) return HIPSDP_ERR_ARG;
   return read_scaled(s, s->blk[block].Z, (long long) s->blk[block].n * s->blk[block].n, s->sol_scale, Z);
}

extern "C" int hipsdp_get_lp(hipsdp_solver* s, double* x, double* z)
{
   if ( s == NULL || !s->solved ) return HIPSDP_ERR_ARG;
   if ( x != NULL ) HS_CALL( read_scaled(s, s->x, s->q, s->sol_scale, x) );
   if ( z != NULL ) HS_CALL( read_scaled(s, s->z, s->q, s->sol_scale, z) );
   return HIPSDP_OK;
}

extern "C" int hipsdp_get_preoptimal(hipsdp_solver* s, int* available, double* y, double* x)
{
   if ( s == NULL || !s->solved || available == NULL ) return HIPSDP_ERR_ARG;
   *available = s->pre_valid ? 1 : 0;
   if ( !s->pre_valid )
      return HIPSDP_OK;
   if ( y != NULL ) HS_CALL( read_scaled(s, s->pre_y, s->m, s->pre_scale, y) );
   if ( x != NULL ) HS_CALL( read_scaled(s, s->pre_x, s->q, s->pre_scale, x) );
   return HIPSDP_OK;
}

extern "C" int hipsdp_get_preoptimal_X(hipsdp_solver* s, int block, double* X)
{
   if ( s == NULL || !s->solved || !s->pre_valid || X == NULL || block < 0 || block >= (int) s->blk.size() ) return HIPSDP_ERR_ARG;
   return read_scaled(s, s->blk[block].Xpre, (long long) s->blk[block].n * s->blk[block].n, s->pre_scale, X);
}

/* A[i][i] += v */
__global__ void k_shift_diag(int n, double* __restrict__ A, double v)
{
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   if ( i < n )
      A[(long long) i * n + i] += v;
}

/* SPMD: rank 0's flag everywhere (no-op for one rank) */
extern "C" int hipsdp_sync_flag(hipsdp_solver* s, int* flag)
{
   if ( s == NULL || flag == NULL )
      return HIPSDP_ERR_ARG;
   if ( s->comm == NULL || s->nranks < 2 )
      return HIPSDP_OK;
   HS_HIP( hipSetDevice(s->device) );
   HS_CALL( stage_sync(s) );
   int* d = NULL;
   HS_CALL( dalloc(&d, 1) );
   int rc = HS_OK;
   if ( hipMemcpyAsync(d, flag, sizeof(int), hipMemcpyHostToDevice, s->stream) != hipSuccess ) rc = HS_ERR_HIP;
   hs_comm_phase(2);
   if ( rc == HS_OK ) rc = hs_bcast_ints(s->comm, d, 1, s->stream);
   if ( rc == HS_OK && (hipMemcpyAsync(flag, d, sizeof(int), hipMemcpyDeviceToHost, s->stream) != hipSuccess
         || hipStreamSynchronize(s->stream) != hipSuccess) )
      rc = HS_ERR_HIP;
   dfree(d);
   return rc;
}

static int check_y_impl(hipsdp_solver* s, const double* y, double tol, double* lmin, double* lpviol);

extern "C" int hipsdp_check_y(hipsdp_solver* s, const double* y, double* lmin, double* lpviol)
{
   return check_y_impl(s, y, 0.0, lmin, lpviol);
}

/* the same check against a known tolerance (what the backend's re-solve loop asks: is lambda_min >= -tol?): blocks above 64 rows
 * are decided by ONE Cholesky factorization of Z(y) + 0.999 tol I - success proves lambda_min > -0.999 tol and lmin reports that
 * bound; only a failure (y really is at or beyond the tolerance) pays for the exact eigenvalue.  At an optimum Z(y) is singular
 * to rounding, i.e. a margin of tol inside the cone: the factorization succeeds and the check costs a pass over A and a Cholesky
 * instead of 250 Lanczos launches. */
extern "C" int hipsdp_check_y_tol(hipsdp_solver* s, const double* y, double tol, double* lmin, double* lpviol)
{
   return check_y_impl(s, y, tol > 0.0 ? tol : 0.0, lmin, lpviol);
}

static int check_y_impl(hipsdp_solver* s, const double* y, double tol, double* lmin, double* lpviol)
{
   if ( s == NULL || !s->shaped ) return HIPSDP_ERR_ARG;
   HS_HIP( hipSetDevice(s->device) );
   HS_CALL( stage_sync(s) );
   CommOff off(s, replicate_small(s));
   const int m = s->m, m1 = m + 1, q = s->q;
   hipStream_t st = s->stream;
   if ( m > 0 )
      HS_HIP( hipMemcpyAsync(s->ys, y, (size_t) m * sizeof(double), hipMemcpyHostToDevice, st) );
   HS_CALL( hs_make_ext(st, m, -1.0, 1.0, s->ys, s->dyt) );
   HS_LAUNCH_CHECK();
   HS_CALL( ensure_packed(s) );
   int k = 0;
   for (auto& B : s->blk)
   {
      const long long n2 = (long long) B.n * B.n;
      HS_CALL( pass_AT(s, B, s->dyt, 0.0, NULL, B.W) );
      /* run Lanczos to (numerical) convergence: up to min(n, 250) steps (not needed where the tolerance certificate decides) */
      if ( !(tol > 0.0 && B.n > 64) )
         HS_CALL( hs_lanczos_lmin(st, B.n, B.W, 250, s->sc + SC_BLK(k, 1), s->lan_ws) );
      else
      {
         HS_CALL( hs_fill(st, s->sc + SC_BLK(k, 1), 1, 0.0) );
         HS_CALL( hs_fill(st, s->sc + SC_BLK(k, 2), 1, 0.0) );
      }
      ++k;
   }
   HS_CALL( hs_fill(st, s->sc + SC_RATX, 1, 0.0) );
   if ( q > 0 )
   {
      const double* v = s->dyt;
      HS_CALL( hs_gemv_n(st, q, m1, s->Dext, m1, 1, &v, s->tmpq, q, s->gemv_ws, s->gemv_ws_len) );
      /* violation = max(0, -min(D y - c)) : use absmax on the negative part via ratio trick: min over rows */
      HS_CALL( hs_fill(st, s->hl, q, -1.0) );
      HS_CALL( hs_ratio_min(st, q, s->tmpq, s->hl, s->sc + SC_RATX, 0, s->red_ws) );   /* min_r (D y - c)_r */
   }
   HostScalars h;
   HS_CALL( read_scalars(s, h, NULL) );
   for (size_t b = 0; b < s->blk.size(); ++b)
      lmin[b] = h.v[SC_BLK(b, 1)] - h.v[SC_BLK(b, 2)];
   /* The reference accepts a solution on an EXACT eigenvalue (sdpsolchecker.c:201-257).  A Ritz value is an upper bound of
    * lambda_min and "theta - resid" only says that SOME eigenvalue lies that close to theta; once the Krylov space cannot span
    * the matrix (n above the 250 steps) an unconverged smaller eigenvalue would go unnoticed - and at an optimum Z(y) has a
    * cluster at 0, the slow case.  So the bound is certified: a Cholesky factorization of W - sigma I with sigma just below the
    * estimate succeeds only if lambda_min > sigma (then sigma is returned, a rigorous lower bound); if it fails the exact
    * Jacobi eigenvalue is computed. */
   for (size_t b = 0; b < s->blk.size(); ++b)
   {
      Block& B = s->blk[b];
      const int n = B.n;
      const bool bytol = tol > 0.0 && n > 64;
      if ( n <= 200 && !bytol )
         continue;
      const long long n2 = (long long) n * n;
      const double theta = h.v[SC_BLK(b, 1)];
      const double sigma = bytol ? -0.999 * tol : lmin[b] - 1e-9 * (1.0 + fabs(theta));
      int fl = 0;
      HS_CALL( hs_copy(st, B.T1, B.W, n2) );
      hipLaunchKernelGGL(k_shift_diag, g1d(n), dim3(256), 0, st, n, B.T1, -sigma);
      HS_LAUNCH_CHECK();
      HS_HIP( hipMemsetAsync(s->flags + 5, 0, sizeof(int), st) );
      HS_CALL( hs_potrf(st, n, B.T1, B.dinvx, s->flags + 5, NULL) );
      HS_HIP( hipMemcpyAsync(&fl, s->flags + 5, sizeof(int), hipMemcpyDeviceToHost, st) );
      HS_HIP( hipStreamSynchronize(st) );
      if ( fl == 0 && std::isfinite(sigma) )
      {
         lmin[b] = sigma;
         continue;
      }
      double *lam = NULL, *V = NULL, *ws = NULL;
      double l0 = 0.0;
      int rc = dalloc(&lam, n);
      if ( rc == HS_OK ) rc = dalloc(&V, n2);
      /* (up to 128 rows: tridiagonal reduction, multisection, inverse iteration in one launch, eigi.hip; block Jacobi above) */
      if ( rc == HS_OK ) rc = dalloc(&ws, n <= 128 ? hs_syev_small_scratch(n) : hs_syev_ws(n));
      if ( rc == HS_OK ) rc = n <= 128 ? hs_syev_small_dev(st, n, B.W, lam, V, ws) : hs_syev_jacobi(st, n, B.W, lam, V, NULL, ws);
      if ( rc == HS_OK && (hipMemcpyAsync(&l0, lam, sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess
            || hipStreamSynchronize(st) != hipSuccess) )
         rc = HS_ERR_HIP;
      dfree(lam); dfree(V); dfree(ws);
      HS_CALL( rc );
      lmin[b] = l0;
   }
   if ( lpviol != NULL )
      *lpviol = q > 0 ? fmax(0.0, -h.v[SC_RATX]) : 0.0;
   return HIPSDP_OK;
}

/* H = v v^T for row `row` of V (ld n) */
__global__ void k_outer_row(int n, const double* __restrict__ V, int row, double* __restrict__ H)
{
   const double* v = V + (long long) row * n;
   for (long long e = (long long) blockIdx.x * blockDim.x + threadIdx.x; e < (long long) n * n; e += (long long) gridDim.x * blockDim.x)
      H[e] = v[e / n] * v[e % n];
}

/* Eigenvector cuts of one block at the point y (LP-based mode of the reference: cons_sdp.c:896-1010 produceCutFromEigenvector,
 * :1612-1803 separateSol): every eigenvector v of Z(y) = sum_i A_i y_i - A_0 with eigenvalue <= -tol gives the valid inequality
 *    sum_i (v^T A_i v) y_i >= v^T A_0 v.
 * Z(y) (one pass over A), its eigen-decomposition (one launch up to 128 rows, block Jacobi above) and the coefficients <A_i, v v^T> (one pass over A per cut) are
 * formed on the device.  The most negative eigenvalues come first; at most maxcuts cuts.
 * eigvals[maxcuts], coefs[maxcuts x m], lhs[maxcuts], vecs[maxcuts x n] (may be NULL) are host arrays. */
extern "C" int hipsdp_eigencuts(hipsdp_solver* s, int block, const double* y, double tol, int maxcuts, int* ncuts, double* eigvals,
   double* coefs, double* lhs, double* vecs)
{
   if ( s == NULL || !s->shaped || block < 0 || block >= (int) s->blk.size() || y == NULL || maxcuts < 0 || ncuts == NULL )
      return HIPSDP_ERR_ARG;
   *ncuts = 0;
   if ( maxcuts == 0 )
      return HIPSDP_OK;
   if ( eigvals == NULL || coefs == NULL || lhs == NULL )
      return HIPSDP_ERR_ARG;
   HS_HIP( hipSetDevice(s->device) );
   HS_CALL( stage_sync(s) );
   CommOff off(s, replicate_small(s));
   const int m = s->m, m1 = m + 1;
   Block& B = s->blk[block];
   const int n = B.n;
   const long long n2 = (long long) n * n;
   hipStream_t st = s->stream;
   if ( m > 0 )
      HS_HIP( hipMemcpyAsync(s->ys, y, (size_t) m * sizeof(double), hipMemcpyHostToDevice, st) );
   HS_CALL( hs_make_ext(st, m, -1.0, 1.0, s->ys, s->dyt) );
   HS_LAUNCH_CHECK();
   HS_CALL( ensure_packed(s) );
   HS_CALL( pass_AT(s, B, s->dyt, 0.0, NULL, B.W) );
   double *lam = NULL, *V = NULL, *ws = NULL, *out = NULL;
   std::vector<double> hlam(n), hout;
   int rc = dalloc(&lam, n);
   if ( rc == HS_OK ) rc = dalloc(&V, n2);
   if ( rc == HS_OK ) rc = dalloc(&ws, n <= 128 ? hs_syev_small_scratch(n) : hs_syev_ws(n));
   if ( rc == HS_OK ) rc = dalloc(&out, m1);
   if ( rc == HS_OK ) rc = n <= 128 ? hs_syev_small_dev(st, n, B.W, lam, V, ws) : hs_syev_jacobi(st, n, B.W, lam, V, NULL, ws);
   if ( rc == HS_OK && (hipMemcpyAsync(hlam.data(), lam, (size_t) n * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess
         || hipStreamSynchronize(st) != hipSuccess) )
      rc = HS_ERR_HIP;
   int k = 0;
   if ( rc == HS_OK )
   {
      while ( k < n && k < maxcuts && hlam[k] <= -tol )
         ++k;
      hout.resize(m1);
   }
   for (int c = 0; c < k && rc == HS_OK; ++c)
   {
      hipLaunchKernelGGL(k_outer_row, g1d(n2), dim3(256), 0, st, n, V, c, B.W2);
      if ( hipGetLastError() != hipSuccess ) { rc = HS_ERR_HIP; break; }
      rc = pass_A(s, B, B.W2, out);
      if ( rc == HS_OK && (hipMemcpyAsync(hout.data(), out, (size_t) m1 * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess
            || hipStreamSynchronize(st) != hipSuccess) )
         rc = HS_ERR_HIP;
      if ( rc != HS_OK )
         break;
      eigvals[c] = hlam[c];
      lhs[c] = hout[0];
      for (int i = 0; i < m; ++i)
         coefs[(size_t) c * m + i] = hout[1 + i];
      if ( vecs != NULL && hipMemcpy(vecs + (size_t) c * n, V + (size_t) c * n, (size_t) n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess )
         rc = HS_ERR_HIP;
   }
   dfree(lam); dfree(V); dfree(ws); dfree(out);
   HS_CALL( rc );
   *ncuts = k;
   return HIPSDP_OK;
}

/* host-only: the column ranges of the sharded W formulation, bounds[0 .. nranks] (rank g owns [bounds[g], bounds[g + 1])) */
extern "C" int hipsdp_shard_columns(int m1, int n, int nranks, int* bounds)
{
   if ( m1 < 1 || n < 1 || nranks < 1 || nranks > 64 || bounds == NULL )
      return HIPSDP_ERR_ARG;
   for (int g = 0; g < nranks; ++g)
   {
      int c0, cw;
      hs_shard_cols(m1, n, nranks, g, &c0, &cw);
      bounds[g] = c0;
      bounds[g + 1] = c0 + cw;
   }
   return HIPSDP_OK;
}

/* mode 1: the constraint matrices of the NEXT hipsdp_set_shape are sharded by variable over the ranks of the communicator
 * (rank g holds rows [g c, (g + 1) c) of A, c = ceil((m + 1) / ranks); the constant matrix is kept everywhere); -1: only when
 * the replicated matrices would not fit; 0 (default): replicated.  Call after hipsdp_set_comm and before hipsdp_set_shape. */
extern "C" int hipsdp_shard_matrices(hipsdp_solver* s, int mode)
{
   if ( s == NULL || mode < -1 || mode > 1 )
      return HIPSDP_ERR_ARG;
   s->shardA_req = mode;
   return HIPSDP_OK;
}

/* 1 when the matrices of the current shape are sharded by variable, 0 when replicated */
extern "C" int hipsdp_matrices_sharded(hipsdp_solver* s)
{
   return s != NULL && s->shaped && s->shardA ? 1 : 0;
}

extern "C" int hipsdp_set_comm(hipsdp_solver* s, void* comm, int rank, int nranks)
{
   if ( s == NULL || nranks < 1 || nranks > 16 || rank < 0 || rank >= nranks || (nranks > 1 && comm == NULL) )
      return HIPSDP_ERR_ARG;
   s->comm = comm; s->rank = rank; s->nranks = nranks;
   /* the Schur workspace depends on the mode: drop it so that the next solve re-creates it */
   if ( s->sws.T != NULL )
   {
      HS_HIP( hipSetDevice(s->device) );
      HS_CALL( stage_sync(s) );
      HS_HIP( hipStreamSynchronize(s->stream) );
      hs_schur_ws_free(&s->sws);
   }
   return HIPSDP_OK;
}
