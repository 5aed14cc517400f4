/* hipsdp.h - C ABI of the MI355X interior-point engine (libhipsdp.so).
 *
 * This is the layer UNDER the SCIP-SDP solver interface: src/sdpi/sdpisolver_hip.c (the drop-in for the reference's
 * src/sdpi/sdpisolver_{dsdp,sdpa,mosek}.c, see include/sdpisolver_hip.h) marshals the arguments of
 * SCIPsdpiSolverLoadAndSolveWithPenalty (reference: src/sdpi/sdpisolver.h:258-322) into the calls below, exactly where
 * the reference backends call DSDPCreate/SDPConeSetASparseVecMat/LPConeSetData2/DSDPSolve/DSDPComputeX
 * (src/sdpi/sdpisolver_dsdp.c:983-1520) or SDPA::inputElement/initializeSolve/solve
 * (src/sdpi/sdpisolver_sdpa.cpp:1179-1670).  Plain C types only; no torch, no C++ in the signatures.
 *
 * Problem handed to the engine (all fixings / index compaction already applied by the caller):
 *
 *    min  b^T y   s.t.  sum_i A_i^k y_i - A_0^k  psd  (k = 0..nblocks-1),    D y - c >= 0  (q rows)
 *
 * All functions return 0 (HIPSDP_OK) or an HIPSDP_ERR_* code; hipsdp_last_error() gives a message.
 * There is NO CPU fallback: without a usable gfx950 device every computing call returns HIPSDP_ERR_NODEVICE.
 */
#ifndef HIPSDP_H
#define HIPSDP_H

#ifdef __cplusplus
extern "C" {
#endif

/* libhipsdp.so is built with hidden visibility: only what this header declares is exported */
#define HIPSDP_API __attribute__((visibility("default")))

#define HIPSDP_OK             0
#define HIPSDP_ERR_NODEVICE   1
#define HIPSDP_ERR_HIP        2
#define HIPSDP_ERR_ARG        3
#define HIPSDP_ERR_NOMEM      4
#define HIPSDP_ERR_NUMERIC    5

/* termination status of hipsdp_solve (hipsdp_info.status) */
#define HIPSDP_STATUS_OPTIMAL   0    /* both problems feasible, tolerances met */
#define HIPSDP_STATUS_DINF      1    /* the y-problem ("dual" in SCIP-SDP terms) is infeasible: X-ray found */
#define HIPSDP_STATUS_DUNB      2    /* the y-problem is unbounded / the X-problem infeasible: y-ray found */
#define HIPSDP_STATUS_PDINF     3    /* both certificates */
#define HIPSDP_STATUS_ITERLIM   4
#define HIPSDP_STATUS_NUMERIC   5    /* stalled / factorization failure */
#define HIPSDP_STATUS_TIMELIM   6
#define HIPSDP_STATUS_OBJLIM    7    /* the X-problem objective (a lower bound) exceeded the objective limit */
#define HIPSDP_STATUS_UNSOLVED -1

typedef struct hipsdp_solver hipsdp_solver;

typedef struct hipsdp_params
{
   double gaptol;        /* absolute duality gap tolerance            (SCIP_SDPPAR_GAPTOL, type_sdpi.h:50) */
   double feastol;       /* absolute feasibility tolerance of y        (SCIP_SDPPAR_SDPSOLVERFEASTOL, type_sdpi.h:52) */
   double infeastol;     /* relative tolerance of Farkas certificates */
   double objlimit;      /* stop when the lower bound exceeds this; >= 1e20: off (SCIP_SDPPAR_OBJLIMIT, type_sdpi.h:53) */
   double timelimit;     /* seconds for this call; <= 0: off */
   double gamma;         /* fraction of the step to the boundary */
   double ws_gbytes;     /* workspace budget of the Schur assembly in GB; <= 0: default */
   int    maxiter;
   int    verbose;       /* 1: one line per iteration on stdout (SCIP_SDPPAR_SDPINFO) */
   int    lanczos_steps; /* Lanczos steps per step-length estimate; 0 (default): 24, or 16 when every block has > 64 rows */
   int    settings;      /* conservativeness of the iteration, the backend's retry ladder (SCIP_SDPSOLVERSETTING, type_sdpi.h:69-77;
                          * sdpisolver_sdpa.cpp:1415-1449, 1698-1795): 0 fast (default), 1 medium (step fraction <= 0.9, twice the
                          * Lanczos steps, more patient stall tests, centrality floor 1e-4), 2 stable (step fraction <= 0.75, four
                          * times the Lanczos steps, even more patient, centrality floor 1e-2) */
   double pabstol;       /* > 0: optimal termination also needs ||b - A(X)||_2 <= pabstol, ABSOLUTE: the caller's own check of
                          * the X-side is absolute (sdpsolchecker.c:775-931 with SCIP_SDPPAR_FEASTOL) while pinf is relative */
   double preoptgap;     /* > 0: the first iterate that is feasible to feastol with relative gap
                          * gap / (1 + |pobj| / 2 + |dobj| / 2) < preoptgap is kept as "preoptimal solution"
                          * (SCIP_SDPPAR_WARMSTARTPOGAP, type_sdpi.h:61; capture rule of sdpisolver_dsdp.c:323-358) */
} hipsdp_params;

typedef struct hipsdp_info
{
   int    status;
   int    iterations;
   double pobj;          /* sum_k <A_0^k, X_k> + c^T x   (X-problem, lower bound when feasible) */
   double dobj;          /* b^T y */
   double pinf;          /* ||b - A(X)|| / (1 + ||b||) */
   double dinf;          /* ||A^T y - A_0 - Z|| / (1 + ||A_0||) */
   double dabs;          /* max_k ||A^T y - A_0 - Z||_F : absolute violation bound of y */
   double gap;           /* |pobj - dobj| */
   double mu;
   double tau, kappa;
   double solve_seconds;    /* wall time of hipsdp_solve */
   double schur_seconds;    /* device time spent in the Schur assembly (sum of HIP event intervals) */
   double schur_flops;      /* algorithmic flops of the assemblies: (4 m1 n^3 + m1^2 n^2) per block and iteration */
   int    schur_calls;
   int    chol_fail;        /* number of step halvings forced by a failed Cholesky */
   int    warm_started;     /* 1: the point given with hipsdp_set_start was interior and has been used */
   int    settings_used;    /* the hipsdp_params.settings this solve ran with */
   double schur_flops_executed; /* FP64 matrix-core flops the assemblies' GEMM launches were ISSUED (whole tiles over the K ranges
                                 * actually walked: triangular factors, lower tiles, skipped zero slabs) - what MFMA utilisation is
                                 * measured against; schur_flops is the algorithmic count */
} hipsdp_info;

HIPSDP_API const char* hipsdp_last_error(void);
HIPSDP_API const char* hipsdp_version(void);
HIPSDP_API int  hipsdp_device_count(void);
HIPSDP_API void hipsdp_default_params(hipsdp_params* p);

HIPSDP_API int  hipsdp_create(hipsdp_solver** solver, int device);
HIPSDP_API void hipsdp_free(hipsdp_solver** solver);

/* Declares the shape: m variables, nblocks dense SDP blocks of the given sizes, q LP rows.  Allocates the device storage
 * A_k[(m+1) x n_k^2] (row i = vec(A_i), row 0 = constant matrix), zero filled. */
HIPSDP_API int  hipsdp_set_shape(hipsdp_solver* solver, int m, int nblocks, const int* blocksizes, int q);
/* The same with the number of lower-triangular triplets the caller is going to add per block (nnz[k] >= 0; NULL or a negative
 * entry: unknown).  A block whose count makes the pair formula over nonzeros the cheaper Schur assembly (4 (sum nnz)^2 multiply-adds
 * against 4 (m + 1) n^3 + (m + 1)^2 n^2; never for n <= 64) is kept SPARSE: no (m + 1) x n^2 array is allocated, the matrices of the
 * variables stay the triplets of hipsdp_add_entries (what the reference backends hand DSDP / SDPA:
 * sdpisolver_dsdp.c:1126-1195, sdpisolver_sdpa.cpp:1223-1267), the constant matrix a dense n x n array.  Such a block takes
 * hipsdp_add_entries only (not hipsdp_set_block_dense / hipsdp_master_gather / hipsdp_gen_planted). */
HIPSDP_API int  hipsdp_set_shape2(hipsdp_solver* solver, int m, int nblocks, const int* blocksizes, int q, const long long* nnz);
HIPSDP_API int  hipsdp_block_is_sparse(hipsdp_solver* solver, int block);
/* 0: never keep a block as nonzeros, 1 (default; environment HIPSDP_SPARSE): by the cost rule above, 2: whenever a count is given */
HIPSDP_API int  hipsdp_sparse_policy(hipsdp_solver* solver, int mode);
/* how many (kernel, device) pairs have had their dynamic-LDS limit raised on that device so far: the attribute
 * (hipFuncAttributeMaxDynamicSharedMemorySize) belongs to the pair, the engine keeps one bit per device and kernel (tests) */
HIPSDP_API int  hipsdp_func_attr_sets(int device);
/* free and total bytes of the device's memory (tests: what a problem allocates) */
HIPSDP_API int  hipsdp_mem_info(int device, double* free_bytes, double* total_bytes);
/* objective b[m] (host) */
HIPSDP_API int  hipsdp_set_obj(hipsdp_solver* solver, const double* b);
/* Scatter lower-triangular COO entries (row >= col) into block k: entry e belongs to matrix var[e] (0 = constant matrix,
 * i = variable i, 1-based) at (row[e], col[e]); both triangles of the dense storage are written.  Host arrays. */
HIPSDP_API int  hipsdp_add_entries(hipsdp_solver* solver, int block, long long nnz, const int* var, const int* row, const int* col,
   const double* val);
/* Master copy (optional, for callers that solve many nodes of one problem): the matrices of the variables in ORIGINAL block
 * sizes are uploaded once (COO, lower triangle) and stay in HBM across hipsdp_set_shape calls.  Block b has nblockvars[b] slots,
 * one per variable that appears in it (nblockvars == NULL: nvars slots per block); entries name their slot.  A node's compact
 * block is then filled on the device:  A_engine[a + 1][r][c] = master[slots[a]][kept[r]][kept[c]]  (slots[a] = -1: the
 * a-th active variable does not appear in the block, zeros).  HIPSDP_ERR_NOMEM from define leaves no master copy behind. */
HIPSDP_API int  hipsdp_master_define(hipsdp_solver* solver, int nvars, int nblocks, const int* blocksizes, const int* nblockvars);
HIPSDP_API int  hipsdp_master_add_entries(hipsdp_solver* solver, int block, long long nnz, const int* slot, const int* row, const int* col,
   const double* val);
/* the same upload straight from per-slot arrays (slot k: nnz[k] entries row[k][.], col[k][.], val[k][.] - the caller's
 * sdprow[b][k] / sdpcol[b][k] / sdpval[b][k] of sdpisolver.h:176-233), streamed through pinned staging chunks */
HIPSDP_API int  hipsdp_master_add_vars(hipsdp_solver* solver, int block, int nslots, const int* nnz, const int* const* row,
   const int* const* col, const double* const* val);
HIPSDP_API int  hipsdp_master_gather(hipsdp_solver* solver, int engine_block, int master_block, int nactive, const int* slots,
   int nkept, const int* kept);
/* dense upload of a whole block: A[(m+1) * n * n] host, row-major */
HIPSDP_API int  hipsdp_set_block_dense(hipsdp_solver* solver, int block, const double* A);
/* LP rows: Dext[q x (m+1)] host, row-major, column 0 = c (constant), columns 1..m = D */
HIPSDP_API int  hipsdp_set_lp(hipsdp_solver* solver, const double* Dext);
/* device-resident access for generators / benchmarks: pointer to A_k on the device */
HIPSDP_API int  hipsdp_block_device_ptr(hipsdp_solver* solver, int block, double** dptr);

/* optional warm start (host arrays; X, Z: nblocks dense n_k x n_k matrices; x, z: q) */
HIPSDP_API int  hipsdp_set_start(hipsdp_solver* solver, const double* y, const double* const* X, const double* const* Z,
   const double* x, const double* z);

HIPSDP_API int  hipsdp_solve(hipsdp_solver* solver, const hipsdp_params* params, hipsdp_info* info);

/* B&B-sized problems (no communicator, every block <= 64 rows, m <= 108 (HIPSDP_SOLVE1_MAXM; 128 fit), q <= 4096, the fixed part of the state fits the 160 KiB of
 * LDS of one compute unit - hs_solve1_fits: one block of 36 rows, two of 30, eight of 12 -, and one Schur assembly
 * stays below 3e6 multiply-adds: few nonzeros per matrix) are solved by ONE launch of one workgroup (csrc/solve1_body.h, one kernel
 * instance per size class; HIPSDP_SOLVE1=0 switches it off).  hipsdp_solve_path: 1 when the last solve ran there, 0 the general path.
 * hipsdp_solve1_trace: scalars of the last one-launch solve - out[0..63] (status, iterations, ..., device cycles per phase; see
 * csrc/solve1_body.h) and, with HIPSDP_SOLVE1_HIST=1 in the environment, up to maxrows rows of 16 doubles per iteration
 * (it, mu, pinf, dinf, gap, tau, kappa, pobj, dobj, predictor step, step, dtau, residual of the linearised primal equation, forced pivots, |dy|, |h|) - what the parity tests compare with the oracle's
 * history.  Either pointer may be NULL. */
HIPSDP_API int  hipsdp_solve_path(hipsdp_solver* solver);
HIPSDP_API long long hipsdp_solve1_solves(void);      /* solves of this process served by the one launch so far */
HIPSDP_API long long hipsdp_solve1_fallbacks(void);   /* cold solves the one-launch kernel gave up on numerically and the general path solved again */
HIPSDP_API long long hipsdp_solve1_fallbacks_warm(void);   /* ... of these: warm-started solves, retried on the general path from the caller's start point */
/* library built with -DS1_DEBUG (developer build of the one-launch kernel: NaN-poisoned LDS and workspace, full fences, wave-uniformity
 * checks; csrc/solve1_body.h): counts[0] = values declared wave-uniform that were not, counts[1] = solves the kernel ran.  Returns 1
 * in such a build, 0 in a release build (counts zero) */
HIPSDP_API int hipsdp_solve1_debug_counts(unsigned int* counts2);
HIPSDP_API int  hipsdp_solve1_trace(hipsdp_solver* solver, double* out64, int maxrows, double* hist);

/* solution readback (host arrays).  For STATUS_OPTIMAL the iterate scaled by 1 / tau; for the infeasibility statuses
 * the normalised ray. */
HIPSDP_API int  hipsdp_get_y(hipsdp_solver* solver, double* y);
HIPSDP_API int  hipsdp_get_X(hipsdp_solver* solver, int block, double* X);
HIPSDP_API int  hipsdp_get_Z(hipsdp_solver* solver, int block, double* Z);
HIPSDP_API int  hipsdp_get_lp(hipsdp_solver* solver, double* x, double* z);
/* the preoptimal iterate of the last solve (params.preoptgap > 0): *available = 0 when none was captured; y (m), x (q) and
 * X of a block as hipsdp_get_y / get_lp / get_X return the final ones; any output pointer may be NULL */
HIPSDP_API int  hipsdp_get_preoptimal(hipsdp_solver* solver, int* available, double* y, double* x);
HIPSDP_API int  hipsdp_get_preoptimal_X(hipsdp_solver* solver, int block, double* X);

/* smallest eigenvalue of  sum_i A_i^k y_i - A_0^k  for every block, on the device (backs the feasibility check of
 * sdpsolchecker.c:201-257 inside the backend); y: m host values; lmin: nblocks host values */
HIPSDP_API int  hipsdp_check_y(hipsdp_solver* solver, const double* y, double* lmin, double* lpviol);
/* the same against a known tolerance: blocks above 64 rows are certified by one Cholesky factorization of Z(y) + 0.999 tol I
 * (lmin = -0.999 tol on success: a rigorous lower bound that passes "lmin >= -tol"); the exact eigenvalue is computed only when
 * that fails */
HIPSDP_API int  hipsdp_check_y_tol(hipsdp_solver* solver, const double* y, double tol, double* lmin, double* lpviol);

/* Several ranks making the same calls (SPMD): *flag becomes rank 0's value on every rank, so that decisions taken from a host
 * clock (time limits) are the same everywhere and no rank is left alone in a collective.  One rank / no communicator: no-op. */
HIPSDP_API int  hipsdp_sync_flag(hipsdp_solver* solver, int* flag);

/* Phase anatomy of the last hipsdp_solve (profiling on): device milliseconds between HIP events recorded at the phase
 * boundaries of the engine's main stream, summed over the iterations.  phases: 0 residuals + termination read-back,
 * 1 factorizations of X and Z, 2 Schur assembly, 3 Cholesky of M + the two solves + the tau-elimination pass, 4 predictor,
 * 5 corrector, 6 update with its Cholesky check.  The same boundaries are roctx ranges (rocprofv3 --marker-trace), always. */
#define HIPSDP_NPHASES 7
HIPSDP_API int  hipsdp_set_profiling(hipsdp_solver* solver, int on);
HIPSDP_API int  hipsdp_get_phase_times(hipsdp_solver* solver, double* ms /* [HIPSDP_NPHASES] */);
HIPSDP_API const char* hipsdp_phase_name(int phase);

/* Eigenvector cuts for the LP-based mode (replaces the host loop of cons_sdp.c:896-1010 / :1612-1803 for one block): for every
 * eigenvector v of Z(y) = sum_i A_i y_i - A_0 of block `block` with eigenvalue <= -tol (most negative first, at most maxcuts)
 *     sum_i coefs[c][i] y_i >= lhs[c],   coefs[c][i] = v^T A_i v,  lhs[c] = v^T A_0 v
 * is violated at y by -eigvals[c].  y: m host values (engine variables); coefs: maxcuts x m; vecs: maxcuts x n or NULL. */
HIPSDP_API int  hipsdp_eigencuts(hipsdp_solver* solver, int block, const double* y, double tol, int maxcuts, int* ncuts, double* eigvals,
                      double* coefs, double* lhs, double* vecs);

/* multi-GPU: Schur rows are sharded over the ranks of an RCCL communicator (one process per GPU); comm comes from hipsdp_comm_create[_host] */
/* Small problems (one assembly below HIPSDP_SHARD_MIN_FLOPS, default 2e10 algorithmic flops) are not sharded: every rank solves
 * them alone with the single-rank kernels and rank 0's outcome (status, iterate, preoptimal iterate) is broadcast once per solve. */
HIPSDP_API int  hipsdp_set_comm(hipsdp_solver* solver, void* comm, int rank, int nranks);
/* Constraint matrices sharded by variable ("A is sharded by variable when it cannot be replicated",
 * n = 4000 / m = 8000 is 1 TB of A): rank g of the communicator holds the matrices of the variables [g c, (g + 1) c),
 * c = ceil((m + 1) / ranks) (index 0 = the constant matrix, which every rank keeps as well); hipsdp_add_entries,
 * hipsdp_set_block_dense and hipsdp_gen_planted store only what the rank holds, the passes over A are completed by an
 * all-gather / all-reduce, and the Schur assembly forms W_j = G A_j R where A_j lives, re-distributes the ENTRIES of the W_j
 * with one all-to-all per column slice (rank h receives its n / ranks rows of all W_j) and sums the partial Gram matrices.
 * mode 1: shard; -1: shard only when the replicated matrices would take more than 75 % of the device memory; 0: replicate
 * (default).  Call after hipsdp_set_comm and before hipsdp_set_shape; the mode applies to every later hipsdp_set_shape. */
HIPSDP_API int  hipsdp_shard_matrices(hipsdp_solver* solver, int mode);
HIPSDP_API int  hipsdp_matrices_sharded(hipsdp_solver* solver);       /* what the last hipsdp_set_shape decided: 1 sharded, 0 replicated */
/* measurement transport: a communicator of nranks ranks of which only `rank` exists - collectives move nothing, results are
 * meaningless; it times one rank's share of a sharded solve at sizes that need several GPUs (tests/devtools/shard_time.py) */
HIPSDP_API int  hipsdp_comm_create_null(int rank, int nranks, void** comm);
/* host-only helper: column ranges of the sharded assembly, bounds[0 .. nranks]; rank g owns [bounds[g], bounds[g + 1]) */
HIPSDP_API int  hipsdp_shard_columns(int m1, int n, int nranks, int* bounds);
HIPSDP_API int  hipsdp_comm_create(const void* unique_id_128bytes, int rank, int nranks, void** comm);
HIPSDP_API int  hipsdp_comm_unique_id(void* unique_id_128bytes);
HIPSDP_API void hipsdp_comm_destroy(void* comm);
/* what the transport says about itself: *count = ncclCommCount for RCCL; *kind (may be NULL) 0 RCCL, 1 host-staged, 2 measurement */
HIPSDP_API int  hipsdp_comm_count(void* comm, int* count, int* kind);
/* optional statistics: a HIP event pair around every collective, booked under the phase the engine is in - 0 Schur exchange
 * (all-reduce / all-gather / all-to-all), 1 passes over A, 2 decision scalars and flags, 3 other.  hipsdp_comm_stats waits for
 * the recorded events and returns seconds[4], calls[4], bytes[4] since the last reset (any pointer may be NULL). */
HIPSDP_API int  hipsdp_comm_stats_enable(void* comm, int on);
HIPSDP_API int  hipsdp_comm_stats(void* comm, double* seconds, long long* calls, double* bytes, int reset);
/* SPMD hosts (N identical processes, one per GPU, making the same calls): the process-wide communicator the environment
 * describes - HIPSDP_WORLD / WORLD_SIZE, HIPSDP_RANK / RANK, and HIPSDP_COMM_FILE=path (RCCL: rank 0 writes the unique id there,
 * the others read it) or HIPSDP_COMM_SHM=/name (host-staged, ranks sharing one device).  *comm = NULL with one rank.  Created at
 * the first call, shared by all solvers of the process, never destroyed.  sdpisolver_hip.c calls it when it creates its engine. */
HIPSDP_API int  hipsdp_comm_from_env(int device, void** comm, int* rank, int* nranks);
/* host-staged communicator for several ranks on ONE device (RCCL refuses that): payloads travel through the POSIX
 * shared-memory segment `name` ("/something", unique per job; every rank passes the same name and staging size).  Same
 * collectives, same results; meant for validating the sharded path on a one-GPU machine, not for speed. */
HIPSDP_API int  hipsdp_comm_create_host(const char* name, int rank, int nranks, long long staging_bytes, double timeout_seconds, void** comm);

/* Synthetic instance of BASELINE.md section 3 generated in HBM.  The solver must have the shape (m, one block of size n,
 * q = 0).  Fills A_1..A_m from the counter stream of oracle/instances.py (seed + i), then plants the optimum the caller
 * supplies: A_0 = sum_i ystar_i A_i - Zstar,  b_i = <A_i, Xstar> (both computed on the device); b is set as objective and
 * returned in b_out[m].  Xstar, Zstar: n x n host arrays; ystar: m host values. */
HIPSDP_API int  hipsdp_gen_planted(hipsdp_solver* solver, int n, int m, long long seed, const double* Xstar, const double* Zstar,
   const double* ystar, double* b_out);
/* the same with matrices of the given density (0 < density <= 1: an off-diagonal entry is nonzero with that probability; SURVEY.md
 * 8(d) names rho = 0.1).  The storage and the assembly stay the dense ones: below the density where the pair formula over nonzeros
 * wins (hipsdp_set_shape2) the three GEMMs are the cheaper formulation whatever the zeros */
HIPSDP_API int  hipsdp_gen_planted_density(hipsdp_solver* solver, int n, int m, long long seed, double density, const double* Xstar,
   const double* Zstar, const double* ystar, double* b_out);
/* copies block k's dense storage A[(m+1) * n * n] back to the host (used to hand identical bits to the CPU baseline) */
HIPSDP_API int  hipsdp_get_block_dense(hipsdp_solver* solver, int block, double* A);

/* ---- host-buffer entry points behind the SCIPlapack* surface (src/sdpi/lapack_interface_hip.c; csrc/host_entries.hip, eigi.hip,
 * psd.hip): every calling thread owns a context per device (own stream, pinned mapped staging, grow-only device pool) - no
 * hipMalloc / hipFree / device-wide synchronisation per call ---- */
/* C[M x N] = alpha * op(A) * op(B) + beta * C, row-major; layA/layB: 0 = K contiguous, 1 = M (resp. N) contiguous */
HIPSDP_API int  hipsdp_dgemm(int device, int layA, int layB, int M, int N, int K, double alpha, const double* A, long long lda,
   const double* B, long long ldb, double beta, double* C, long long ldc, int lower_only, int splitk);
/* shader frequency DURING the Schur assemblies of a solve: with sampling on, ONE thread on a queue of its own (k_clock_window) starts
 * with every assembly and counts its shader-clock cycles and the 100 MHz wall ticks until the assembly's last kernel has set a word
 * (a pair of samples from launches before and after the assembly lands on different XCDs, whose cycle counters are not aligned);
 * *ghz = sum of cycles / sum of wall time over the assemblies of the last solve (0 when none was sampled) */
HIPSDP_API int  hipsdp_set_clock_sampling(hipsdp_solver* solver, int on);
HIPSDP_API int  hipsdp_get_assembly_clock(hipsdp_solver* solver, double* ghz);
HIPSDP_API int  hipsdp_syev(int device, int n, const double* A, double* lam, double* V);     /* ascending, eigenvectors as rows */
/* i-th smallest eigenvalue (1-based) and optionally its unit eigenvector of a symmetric matrix with n <= 128 in one launch through
 * pinned, device-mapped staging memory of the calling thread (no allocation, no copy engine, no stream synchronisation):
 * Householder tridiagonalisation in LDS, Sturm multisection, inverse iteration, back-transformation - one eigenpair as DSYEVR
 * RANGE = 'I' computes it (lapack_interface.c:178-288); n <= 64 with the matrix in registers, 64 < n <= 128 in LDS (round 3).
 * HIPSDP_ERR_ARG for n > 128. */
HIPSDP_API int  hipsdp_syevi_small(int device, int n, const double* A, int i, double* eigval, double* eigvec);
/* all eigenpairs, n <= 128, the same way (what DSYEVR RANGE = 'A' computes, lapack_interface.c:507-603): eigenvalues by multisection,
 * eigenvectors by inverse iteration with re-orthogonalisation inside clusters, one launch; hipsdp_syev takes this path for n <= 128 */
HIPSDP_API int  hipsdp_syev_small(int device, int n, const double* A, double* lam, double* V);
/* PSD projection chain of the warm-start producer (relax_sdp.c:2715-2766 for Z, :3405-3445 for X), fused on the device: sparse
 * lower/upper triangle (row, col, val; both triangles are filled) -> eigen-decomposition -> eigenvalues below minev (by more
 * than epsilon, SCIPisLT) raised to minev -> recombination -> entries with row <= col and |value| > epsilon in row-major order.
 * mode 0: the reference's literal chain R[i][j] = sum_c V[i][c] lambda_c V[j][c] (V[k][:] = k-th eigenvector; this is what
 * scaleTransposedMatrix + SCIPlapackMatrixMatrixMult(V, TRUE, S, FALSE) evaluate); mode 1: spectral R = sum_k lambda_k v_k v_k^T.
 * cap = length of the output arrays; *nnz_out = entries produced (when it exceeds cap: HIPSDP_ERR_ARG, nothing is written). */
HIPSDP_API int  hipsdp_psd_project(int device, int n, int nnz, const int* row, const int* col, const double* val, double minev, double epsilon,
   int mode, int cap, int* nnz_out, int* rowout, int* colout, double* valout);
HIPSDP_API int  hipsdp_gemv_n(int device, int R, long long E, const double* A, int nv, const double* V, double* out);
HIPSDP_API int  hipsdp_gemv_t(int device, int R, long long E, const double* A, const double* coef, double* out);

#ifdef __cplusplus
}
#endif

#endif
