/* hs_kernels.h - launch wrappers of the bandwidth-bound kernels (kernels.hip), the dense factorizations (chol.hip) and the
 * eigen kernels (eig.hip).  All pointers are device pointers unless a name ends in _h.  Every wrapper only enqueues work
 * on the given stream; nothing here synchronizes. */
#ifndef HS_KERNELS_H
#define HS_KERNELS_H

#include "hs_common.h"

/* ---- kernels.hip ---------------------------------------------------------------------------------------------- */
int hs_fill(hipStream_t s, double* p, long long n, double v);
int hs_set_identity(hipStream_t s, double* A, int n, double v);
int hs_copy(hipStream_t s, double* dst, const double* src, long long n);
int hs_axpy(hipStream_t s, long long n, double a, const double* x, double* y);              /* y += a x */
int hs_scale_add(hipStream_t s, long long n, double a, const double* x, double b, const double* y, double* out); /* out = a x + b y (y may be NULL) */
int hs_mirror_lower(hipStream_t s, double* A, int n, long long lda);                        /* A[i][j] = A[j][i] for i < j */
int hs_symmetrize(hipStream_t s, double* A, int n);
int hs_transpose(hipStream_t s, int n, const double* src, double* dst);                     /* dst[i][j] = src[j][i], n x n, dst != src */                                         /* A = (A + A^T) / 2 */

/* out[v * ldo + i] = sum_e A[i * lda + e] * V_v[e],  i < R, e < E, v < nv <= 4 (one pass over A for all nv vectors) */
int hs_gemv_n(hipStream_t s, int R, long long E, const double* A, long long lda, int nv, const double* const* V,
   double* out, long long ldo, double* ws, long long wsdoubles);
/* out[e] = sum_i coef[i] * A[i * lda + e] + sa * add[e]   (add may be NULL) */
int hs_gemv_t(hipStream_t s, int R, long long E, const double* A, long long lda, const double* coef, double sa,
   const double* add, double* out);

/* three linear combinations of the rows in one sweep over A (o_v[e] = sum_i c_v[i] A[i * lda + e]); 1: done, 0: shapes do not
 * qualify (E, lda even, 16-byte aligned), < 0: error code negated */
/* the same with a workspace: few entries and many rows are summed in row chunks side by side (kernels.hip); hs_gemv_t_chunks(R, E)
 * * E doubles are needed for that (0: never split) */
int hs_gemv_t_chunks(int R, long long E);
int hs_gemv_t_ws(hipStream_t s, int R, long long E, const double* A, long long lda, const double* coef, double sa,
   const double* add, double* out, double* ws, long long wsdoubles);
int hs_gemv_t3(hipStream_t s, int R, long long E, const double* A, long long lda, const double* c0, const double* c1, const double* c2,
   double* o0, double* o1, double* o2);

/* out[slot] = sum_e a[e] * b[e]  (deterministic two-stage; partials in ws, >= 512 doubles) ; accumulate: out[slot] += */
/* deferred scalar reductions (kernels.hip): between begin and end, reductions over short vectors, scalar fills and scalar
 * copies on that stream are recorded and then executed in order by one launch */
void hs_red_batch_begin(hipStream_t s);
int hs_red_batch_end(void);
int hs_red_batch_end_publish(hipStream_t s, int n, const double* src, double* dst, unsigned long long seq, unsigned long long* flag);
void hs_red_batch_reset(void);
int hs_fill_scalar(hipStream_t s, double* p, double v);
/* single-block solves (m <= 64) and the direction's closing kernel can be part of a batch: see kernels.hip */
int hs_red_batch_solve(hipStream_t s, int m, const double* dinv, const double* L, int nrhs, double* vec, long long ld);
struct hs_rb_finish { int m; double eta, rg, sigmu, tau, kappa, etk; const double* u1; const double* u2; double* dy; double* dyt; double* sc;
   int s0, bub, bh, wrp, bu1, dtau, dkappa, den; };
int hs_red_batch_finish(hipStream_t s, const void* fin, size_t bytes);
int hs_copy_scalar(hipStream_t s, double* dst, const double* src);
/* regions in which everything recordable on the stream is recorded (the small single-workgroup kernels of the B&B-sized regime too:
 * kernels.hip) and everything else launches the records first; regions nest, a read-back (hs_red_batch_end_publish) ends them */
void hs_red_batch_hold(hipStream_t s);
int hs_red_batch_release(void);
int hs_red_batch_flush(void);
int hs_red_batch_end_all(void);
int hs_make_ext(hipStream_t s, int m, double s0, double s1, const double* v, double* ext);
int hs_lp_rows_small(hipStream_t s, int q, int m1, const double* Dext, const double* v, int mode, double eta, double sigmu, const double* x,
   const double* z, const double* rd, const double* elp, double* out1, double* out2);
#define HS_AS_MAXBLK 8
struct hs_as_args { int nblk; int n2[HS_AS_MAXBLK]; const double* A[HS_AS_MAXBLK]; const double* V[HS_AS_MAXBLK]; };
int hs_apply_A_small(hipStream_t s, int m1, const hs_as_args* B, int q, const double* Dext, const double* vlp, double* out, int epi,
   double scal, const double* vin, double* vout);
int hs_axpy3(hipStream_t s, double a, long long n1, const double* x1, double* y1, long long n2, const double* x2, double* y2,
   long long n3, const double* x3, double* y3);
int hs_dot(hipStream_t s, long long n, const double* a, const double* b, double* out, int accumulate, double* ws);
/* out[slot] = max(out[slot] if accumulate, max_e |a[e]|) */
int hs_absmax(hipStream_t s, long long n, const double* a, double* out, int accumulate, double* ws);
/* out[slot] = min(out[slot] if accumulate, min over e with d[e] < 0 of -x[e] / d[e]); +1e300 when no such e */
int hs_ratio_min(hipStream_t s, long long n, const double* x, const double* d, double* out, int accumulate, double* ws);

/* H = s1 * Zinv - X - (GZ + GZ^T) / 2,  all n x n */
/* largest block size of the single-workgroup ("small") variants that keep whole matrices in LDS */
#define HS_SMALL_N 48
/* n <= HS_SMALL_N: out = s1 Zinv - X - sym((c X R + E) Zinv) in one launch (E may be NULL) */
int hs_dir_block_small(hipStream_t s, int n, double c, const double* X, const double* R, const double* E, const double* Zinv,
   double s1, double* out);
int hs_dirmat(hipStream_t s, int n, double s1, const double* Zinv, const double* X, const double* GZ, double* H);

/* LP block element-wise pieces (length q) */
int hs_lp_dir(hipStream_t s, int q, double sigmu, double eta, const double* x, const double* z, const double* rd_or_dz,
   const double* elp, double* out);      /* out = sigmu / z - x - (eta * x * r + elp) / z ; elp may be NULL */
int hs_lp_scale_rows(hipStream_t s, int q, int cols, const double* x, const double* z, const double* D, double* S); /* S[r] = (x_r / z_r) D[r] */
int hs_vec_mul(hipStream_t s, long long n, const double* a, const double* b, double* out);   /* out = a .* b */
int hs_lp_s0(hipStream_t s, int q, const double* x, const double* z, const double* beta, double* out, int accumulate, double* ws); /* sum (x/z) beta^2 */

/* packed lower copies for the bandwidth-bound passes (half the bytes of the full storage) */
int hs_pack_rows(hipStream_t s, int m1, int n, long long Lp, const double* A, double* Apk);
int hs_pack_weighted(hipStream_t s, int n, const double* V, double* pk);
int hs_unpack_sym(hipStream_t s, int n, const double* pk, double sa, const double* add, double* out);
int hs_zero_upper(hipStream_t s, double* A, int n);                                         /* A[i][j] = 0 for i < j */

/* ---- schur.hip ------------------------------------------------------------------------------------------------ */
struct hs_schur_ws
{
   double*   T;            /* chunk_cols x n^2 */
   double*   U;            /* chunk_cols x n^2 */
   double*   K;            /* split-K slabs */
   double*   V;            /* variable-sharded form only: the received pieces, m1 x (rows of this rank) x (slice width) */
   long long capT, capV;   /* variable-sharded form: doubles in T (= U) and in V */
   long long chunk_cols;
   long long n2;           /* doubles per matrix the allocation was sized for */
   long long kws_len;
   int       full;         /* 1: T and U hold all m1 matrices (hs_schur_W usable) */
   /* variable-sharded form, overlapped exchange: second send / receive buffers and the events of the two-slice pipeline */
   double*   U2;
   double*   V2;
   void*     evP[2];       /* hipEvent_t: the products of a slice are in its send buffer */
   void*     evX[2];       /* hipEvent_t: the exchange of a slice has arrived */
   void*     ev_g2;        /* hipEvent_t or NULL: hs_schur_W waits for it between its first and its second product (the inverse factor of Z
                            * is formed on another queue while A_stack R runs; set and cleared by the caller around the call) */
   int       (*after_g1)(void*);   /* or NULL: called by hs_schur_W when its first product is in the queue (the caller puts work of its
                                    * own into other queues there - it may set ev_g2); once, then cleared */
   void*     after_g1_arg;
};
int  hs_schur_ws_alloc(hs_schur_ws* w, int m1, long long n2max, double budget_gb);
void hs_schur_ws_free(hs_schur_ws* w);
/* accumulate the lower triangle of Mx (ld m1) for the columns [j_begin, j_end) with U_j = X A_j Zinv */
int  hs_schur_U(hipStream_t s, int m1, int n, const double* A, const double* X, const double* Zinv, double* Mx,
   hs_schur_ws* w, int j_begin, int j_end);
/* accumulate the lower triangle of Mx with W_j = G A_j R (R: lower Cholesky factor of X with a ZERO upper triangle,
 * G: inverse of the lower Cholesky factor of Z) */
int  hs_schur_W(hipStream_t s, int m1, int n, const double* A, const double* R, const double* G, double* Mx, hs_schur_ws* w);
int  hs_schur_W_identity(hipStream_t s, int m1, int n, const double* A, double* Mx, hs_schur_ws* w);
int  hs_schur_W_identity_range(hipStream_t s, int m1, int n, const double* A, long long k0, long long k1, double* Mx, hs_schur_ws* w);

int  hs_schur_Urows(hipStream_t s, int m1, int n, const double* A, const double* X, const double* Zinv, double* Mx,
   hs_schur_ws* w, int r_begin, int r_end);
void hs_shard_rows(int m1, int nranks, int rank, int* chunk_rows, int* first_begin, int* second_begin);
/* column-slice sharding of the W formulation: this rank's share [c0, c0 + cw) of the columns of the W_j, accumulated into the
 * lower tiles of Mx; the ranks' partial matrices add up to the Schur matrix (all-reduce) */
int  hs_schur_Wcols(hipStream_t s, int m1, int n, const double* A, const double* R, const double* G, double* Mx, hs_schur_ws* w,
   int c0, int cw);
void hs_shard_cols(int m1, int n, int nranks, int rank, int* c_begin, int* c_width);
/* variable-sharded form (A_j lives on the rank that owns variable j; A is the pointer row 0 WOULD have): the column slice
 * [c0, c0 + cw) of W_j = G A_j R is formed for the rank's own rows [r0, r1) of A, cut into row ranges, exchanged all-to-all so
 * that every rank holds its row range of ALL W_j, and that rank's part of W W^T is accumulated into the lower tiles of Mx.
 * All ranks call it with the same (c0, cw); the partial matrices add up to the Schur matrix (all-reduce afterwards). */
int  hs_schur_Wvar(hipStream_t s, void* comm, int rank, int nranks, int m1, int n, const double* A, const double* R, const double* G,
   double* Mx, hs_schur_ws* w, int c0, int cw);
int  hs_schur_ws_alloc_var(hs_schur_ws* w, int m1, int nranks, int n, int cwmax);
/* the second pair of buffers + events for hs_schur_Wvar_all(overlap); HS_ERR_NOMEM leaves the in-order form usable */
int  hs_schur_ws_alloc_var_overlap(hs_schur_ws* w);
/* all column slices of one block: slice after slice as hs_schur_Wvar does (overlap = 0), or - overlap = 1, second buffers present -
 * as a two-slice pipeline: the all-to-all of slice s runs on the communication queue `sc` behind an event while the compute queue
 * forms the products of slice s + 1; the Gram update of slice s waits for its exchange.  Same kernels, same arguments, same order
 * of the updates of Mx: bit-identical to the in-order form. */
int  hs_schur_Wvar_all(hipStream_t s, hipStream_t sc, void* comm, int rank, int nranks, int m1, int n, const double* A, const double* R,
   const double* G, double* Mx, hs_schur_ws* w, int cwmax, int overlap);
void hs_var_rows(int m1, int nranks, int rank, int* r0, int* r1);            /* rows of A (variables) a rank owns */
void hs_var_wrows(int n, int nranks, int rank, int* q0, int* q1);            /* rows of the W_j a rank receives */
int  hs_alltoall(void* comm, const double* send, double* recv, const long long* cnt, hipStream_t stream);
int  hs_allreduce_sum(void* comm, double* buf, long long count, hipStream_t stream);
int  hs_mirror_upper(hipStream_t s, double* A, int n, long long lda);                       /* A[i][j] = A[j][i] for i > j */
/* multi.hip: in-place all-gather of equal pieces, piece of rank r at buf + r * count */
int  hs_allgather_inplace(void* comm, double* buf, long long count_per_rank, int rank, hipStream_t stream);
int  hs_allgather(void* comm, const double* send, double* recv, long long count_per_rank, hipStream_t stream);

/* ---- sparse.hip: constraint matrices kept as nonzeros (see there) ----------------------------------------------------- */
struct hs_sparse;
int  hs_sp_prefers_sparse(int n, int m, long long nnz);
int  hs_sp_build(hs_sparse** out, int n, int m, long long nnz, const int* var, const int* row, const int* col, const double* val);
void hs_sp_free(hs_sparse* sp);
long long hs_sp_nnz(const hs_sparse* sp);
int  hs_sp_apply_A(hipStream_t s, const hs_sparse* sp, const double* V, double* out_var1);       /* out[v - 1] = <A_v, V>, v = 1 .. m */
int  hs_sp_apply_AT(hipStream_t s, const hs_sparse* sp, const double* coef, double* out);        /* out += sum_{v >= 1} coef[v] A_v */
int  hs_sp_schur(hipStream_t s, const hs_sparse* sp, const double* X, const double* Zinv, double* Mx);   /* lower triangle, i, j >= 1 */
int  hs_sp_expand(hipStream_t s, const hs_sparse* sp, double* A);

/* ---- chol.hip ------------------------------------------------------------------------------------------------- */
/* In-place blocked Cholesky of the lower triangle of the row-major n x n matrix A (lda = n): A = L L^T, L stored in the
 * lower triangle (upper triangle is left untouched).  dinv receives the inverses of the 64 x 64 diagonal blocks of L
 * (ceil(n/64) * 64 * 64 doubles; the buffer must hold hs_potrf_dinv_len(n) doubles).  *flag (device int) is set to 1 + index of the first non-positive pivot.
 * diag0 == NULL: strict (definite) mode.  diag0 != NULL (the n original diagonal entries): semidefinite mode, pivots
 * below 1e-13 * diag0[k] are replaced by 1e-13 * diag0[k] and no failure is flagged. */
int hs_potrf(hipStream_t s, int n, double* A, double* dinv, int* flag, const double* diag0);
long long hs_potrf_dinv_len(int n);         /* doubles dinv must hold (inverses + staging blocks of the fused block-column kernel) */
/* set_flag: the (single-launch, n <= 64) factorization stores its result into *flag instead of recording a failure into a
 * cleared flag - for callers where it is the only writer of that flag between two reads */
int hs_potrf_psd(hipStream_t s, int n, double* A, double* dinv, int* flag, const double* diag0, int* regmask, int set_flag);
/* Linv = L^-1 (lower triangular, full n x n storage, upper triangle zero); needs dinv from hs_potrf */
int hs_trtri(hipStream_t s, int n, const double* L, const double* dinv, double* Linv, double* tmp);
/* solves L y = r (nrhs <= 4 right-hand sides, rhs[k * ldr + i]) then optionally L^T x = y, in place.  mode 1: forward only,
 * 2: backward only, 3: both */
/* n <= 64: Cholesky of base + alpha * dir in one launch; optionally stores the matrix (Mout), inv(L) as n x n (Linv) and,
 * for n <= 32, the inverse of the matrix (Gram); L gets a zero upper triangle */
/* the same for two matrices of the same order in one launch (arrays of two) */
int hs_potrf_small_ext_pair(hipStream_t s, int n, double* const* L, double* const* dinv, int* const* flag, const double* const* base,
   const double* const* dir, double alpha, double* const* Mout, double* const* Linv, double* const* Gram, int set_flag);
int hs_potrf_small_ext(hipStream_t s, int n, double* L, double* dinv, int* flag, const double* base, const double* dir, double alpha,
   double* Mout, double* Linv, double* Gram, int set_flag);
/* all blocks n <= 32: the extended Schur matrix (SDP blocks + LP part, symmetric), Lm = Mx[1:, 1:] and its diagonal in one
 * launch; returns 1 if done, 0 if the sizes do not qualify, < 0 on error */
int hs_schur_small(hipStream_t s, int m1, int nblk, const int* n, const double* const* A, const double* const* X,
   const double* const* Zinv, int q, const double* Dext, const double* x, const double* z, double* Mx, double* Lm, double* diagM);
/* mode: 1 forward, 2 backward, 3 both; + 4: every 64-wide diagonal solve x = inv(L_bb) r is corrected once with the factor itself
 * (x += inv(L_bb) (r - L_bb x)), which brings its residual from cond(L_bb) eps |r| down to that of a substitution */
int hs_trsv(hipStream_t s, int n, const double* L, const double* dinv, int nrhs, double* rhs, long long ldr, int mode);
/* the same solve with one workgroup per 64-row block (the blocks hand their parts of the solution over through exchange
 * vectors in sync_ws); sync_ws: hs_trsv_sync_ws(n) ints, put into their initial state by hs_trsv_sync_init once after
 * allocation; *epoch: call counter owned by the caller (hs_trsv_sync_init zeroes it) */
long long hs_trsv_sync_ws(int n);
int hs_trsv_sync_init(hipStream_t s, int n, int* sync_ws, int* epoch);
int hs_trsv_sync(hipStream_t s, int n, const double* L, const double* dinv, int nrhs, double* rhs, long long ldr, int mode,
   int* sync_ws, int* epoch);

/* ---- eig.hip -------------------------------------------------------------------------------------------------- */
/* Lanczos estimate of the smallest eigenvalue of the symmetric n x n matrix W.  res[0] = Ritz value theta,
 * res[1] = residual bound (an eigenvalue lies within res[1] of theta), res[2] = steps used.
 * ws: (maxsteps + 2) * n + 4 * maxsteps + 64 doubles. */
int hs_lanczos_lmin(hipStream_t s, int n, const double* W, int maxsteps, double* res, double* ws);
long long hs_lanczos_ws(int n, int maxsteps);
/* 16 < n <= 64: lambda_min(L0 D0 L0^T), lambda_min(L1 D1 L1^T) by Lanczos with the products, all steps and the tridiagonal
 * problem in one launch */
/* step-length estimates of several small blocks in one launch (eig.hip): all n[j] <= 16 or all in 17 .. 64 */
#define HS_STEP_MAXJOBS 32
struct hs_step_jobs { int nblk; int n[HS_STEP_MAXJOBS]; const double* L0[HS_STEP_MAXJOBS]; const double* D0[HS_STEP_MAXJOBS];
   const double* L1[HS_STEP_MAXJOBS]; const double* D1[HS_STEP_MAXJOBS]; double* res0[HS_STEP_MAXJOBS]; double* res1[HS_STEP_MAXJOBS]; };
int hs_steplen_small_multi(hipStream_t s, const hs_step_jobs* P, int maxsteps);
/* 17 .. 64 rows: the eigenvalue itself (Householder reduction + multisection, eigi.hip) instead of the Lanczos estimate */
int hs_lmin_exact_multi(hipStream_t st, const hs_step_jobs* P);
int hs_lanczos_scaled_small(hipStream_t s, int n, int maxsteps, const double* L0, const double* D0, const double* L1, const double* D1,
   double* res0, double* res1);
int hs_lmin_scaled_tiny(hipStream_t s, int n, const double* L0, const double* D0, const double* L1, const double* D1, double* res0,
   double* res1);
/* rot, dsync (may be NULL: one launch per Lanczos step): two host integers and hs_lanczos_sync_words() device words owned by
 * the caller, put into their initial state by hs_lanczos_sync_reset (after allocation, and again after a run that reported
 * NaN); with them the whole run is one launch whose workgroups hand the product vector over through those words; nwipe: the
 * largest n of all runs that share these words */
int hs_lanczos_lmin2(hipStream_t s, int n, const double* W0, const double* W1, int maxsteps, double* res0, double* res1,
   double* ws0, double* ws1, int* rot, unsigned long long* dsync, int nwipe);
long long hs_lanczos_sync_words(void);
int hs_lanczos_sync_reset(hipStream_t s, unsigned long long* dsync, int* rot);

/* eigi.hip: n <= 128, all eigenpairs in one launch on device buffers (same results as hipsdp_syev_small) */
long long hs_syev_small_scratch(int n);
int hs_syev_small_dev(hipStream_t st, int n, const double* A, double* lam, double* V, double* scratch);

/* Cyclic Jacobi eigen-decomposition of the symmetric n x n matrix A (destroyed): eigenvalues ascending in lam[n],
 * eigenvectors as rows of V (row k = k-th eigenvector).  info (device int) = sweeps used or -1. */
int hs_syev_jacobi(hipStream_t s, int n, double* A, double* lam, double* V, int* info, double* ws);
long long hs_syev_ws(int n);

/* ---- solve1.hip: a whole node solve of a B&B-sized problem in one launch of one workgroup ------------------------------- */
#define HS_S1_MAXBLK 8
/* termination status written to out[0]: the HIPSDP_STATUS_* values of include/hipsdp.h, or -2: declined (too much work for one
 * compute unit, or the workspace is too small) - nothing was solved, the caller takes the general path */
#define HS_S1_OPTIMAL 0
#define HS_S1_DINF 1
#define HS_S1_DUNB 2
#define HS_S1_PDINF 3
#define HS_S1_ITERLIM 4
#define HS_S1_NUMERIC 5
#define HS_S1_TIMELIM 6
#define HS_S1_OBJLIM 7
#define HS_S1_OUT_DOUBLES 64
/* ---- the deferred setters of a node (csrc/ipm.hip: stage_*, flush_cmds): commands in pinned memory, run in order by one workgroup -
 * the first thing the one-launch solve does, or a launch of their own before anything else touches the device data */
enum { NC_ZERO = 1, NC_COPY = 2, NC_GATHER = 3, NC_SCATTER = 4 };
struct NodeCmd
{
   int op, i0, i1, i2, i3, i4, pad0, pad1;
   long long n;
   double* dst;                 /* ZERO, COPY: destination; GATHER, SCATTER: the block's matrices */
   double* dst2;                /* SCATTER: the constant matrix */
   const double* src;           /* COPY: data; GATHER: master copy; SCATTER: values */
   const int* idx;              /* GATHER: active slots then kept indices; SCATTER: var, row, col */
   long long pad2;
};
#define NC_MAX 48
#define NC_BYTES (NC_MAX * sizeof(NodeCmd))
#define NC_LIMIT 65536           /* elements a deferred command may touch */
#ifdef __HIPCC__
/* all threads of ONE workgroup; sc: NC_MAX commands of LDS.  stage != NULL: `staged` bytes (a multiple of 16) of LDS that take the
 * first `staged` bytes of the arena - the command list and everything the commands read from pinned memory - in ONE pipelined pass
 * over PCIe; the commands then read their data and index arrays out of LDS.  (Read where the commands use them, every element is a
 * PCIe round trip of its own in a chain per thread - the gather reads three indices per entry -: 12 us for the six commands of a
 * node of example_TT, 4 us this way.)  sc is not used then. */
__device__ __forceinline__ void hs_run_node_cmds(const NodeCmd* __restrict__ cmds, int ncmd, NodeCmd* sc, double* stage = NULL, long long staged = 0)
{
   const char* const arena = reinterpret_cast<const char*>(cmds);
   if ( stage != NULL )
   {
      typedef double nc_v2 __attribute__((ext_vector_type(2)));
      const nc_v2* src = reinterpret_cast<const nc_v2*>(cmds);
      nc_v2* dst = reinterpret_cast<nc_v2*>(stage);
      const int n16 = (int) (staged >> 4);
      for (int b = threadIdx.x; b < n16; b += 4 * blockDim.x)
      {
         nc_v2 v[4];
#pragma unroll
         for (int u = 0; u < 4; ++u)
         {
            const int i = b + u * (int) blockDim.x;
            v[u] = src[i < n16 ? i : n16 - 1];
         }
#pragma unroll
         for (int u = 0; u < 4; ++u)
         {
            const int i = b + u * (int) blockDim.x;
            if ( i < n16 )
               dst[i] = v[u];
         }
      }
      sc = reinterpret_cast<NodeCmd*>(stage);
   }
   else
   {
      const long long* src = reinterpret_cast<const long long*>(cmds);
      long long* dst = reinterpret_cast<long long*>(sc);
      const int words = ncmd * (int) (sizeof(NodeCmd) / sizeof(long long));
      for (int i = threadIdx.x; i < words; i += blockDim.x)
         dst[i] = src[i];
   }
   __syncthreads();
   /* an address inside the staged part of the arena -> its copy in LDS */
   auto loc = [&](const void* p) -> const void*
   {
      const char* c = reinterpret_cast<const char*>(p);
      if ( stage != NULL && c >= arena && c < arena + staged )
         return reinterpret_cast<const char*>(stage) + (c - arena);
      return p;
   };
   for (int c = 0; c < ncmd; ++c)
   {
      const NodeCmd& q = sc[c];
      if ( q.op == NC_ZERO )
      {
         for (long long e = threadIdx.x; e < q.n; e += blockDim.x)
            q.dst[e] = 0.0;
      }
      else if ( q.op == NC_COPY )
      {
         const double* src = reinterpret_cast<const double*>(loc(q.src));
         for (long long e = threadIdx.x; e < q.n; e += blockDim.x)
            q.dst[e] = src[e];
      }
      else if ( q.op == NC_GATHER )
      {
         /* (k_master_gather) i0 = active variables, i1 = kept rows, i2 = order of the master matrices */
         const int nactive = q.i0, nk = q.i1, N = q.i2;
         const int* act = reinterpret_cast<const int*>(loc(q.idx)); const int* kept = act + nactive;
         const long long nk2 = (long long) nk * nk, total = (long long) nactive * nk2;
         for (long long e = threadIdx.x; e < total; e += blockDim.x)
         {
            const long long a = e / nk2;
            const long long rc = e - a * nk2;
            const int r = (int) (rc / nk), cc = (int) (rc - (long long) r * nk);
            q.dst[(a + 1) * nk2 + rc] = act[a] >= 0 ? q.src[((long long) act[a] * N + kept[r]) * N + kept[cc]] : 0.0;
         }
      }
      else if ( q.op == NC_SCATTER )
      {
         /* (k_scatter_coo, indices checked by the host) i0 = order of the block, i1, i2 = the rows of A this rank holds */
         const int n = q.i0, r0 = q.i1, r1 = q.i2;
         const long long n2 = (long long) n * n;
         const int* var = reinterpret_cast<const int*>(loc(q.idx)); const int* row = var + q.n; const int* col = var + 2 * q.n;
         const double* val = reinterpret_cast<const double*>(loc(q.src));
         for (long long e = threadIdx.x; e < q.n; e += blockDim.x)
         {
            const int v = var[e], r = row[e], cc = col[e];
            if ( v != 0 && (v < r0 || v >= r1) )
               continue;
            double* a = (v == 0) ? q.dst2 : q.dst + (long long) v * n2;
            a[(long long) r * n + cc] = val[e];
            a[(long long) cc * n + r] = val[e];
         }
      }
      __syncthreads();
   }
}
#endif

#ifdef __HIPCC__
/* ---- one wavefront: x = (L L^T)^-1 r for a single-block factor (m <= 64) in the ORACLE's order (oracle/ipm_ref.py: msolve) - forward
 * substitution, one correction with the factor itself, backward substitution, one correction - where the default multiplies by an
 * explicitly inverted factor (with the same corrections).  An option (HIPSDP_SMALL_SOLVE=subst, kernels.hip:
 * hs_small_solve_by_substitution), built in round 6 to test whether the inverted blocks are why the general path parts from the oracle
 * on singular Schur complements; they are not.
 * sL: LDS image of L, row i at sL + 65 i (entries j <= i valid, the diagonal included); lane = row; r, result: this lane's entry. */
__device__ __forceinline__ double hs_wl_bcast(double v, int src)
{
   const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
   const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
   return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double hs_wl_fwd(const double* sL, int m, int lane, double di, double x)
{
   const bool live = lane < m;
   const double* row = sL + (live ? lane : 0) * 65;
   double a = live ? x : 0.0;
   for (int k = 0; k < m; ++k)
   {
      const double y = hs_wl_bcast(a * di, k);
      const double c = row[k];
      if ( live && lane > k )
         a = fma(-c, y, a);
   }
   return a * di;
}
__device__ __forceinline__ double hs_wl_bwd(const double* sL, int m, int lane, double di, double x)
{
   const bool live = lane < m;
   const int col = live ? lane : 0;
   double a = live ? x : 0.0;
   for (int k = m - 1; k >= 0; --k)
   {
      const double y = hs_wl_bcast(a * di, k);
      const double c = sL[k * 65 + col];
      if ( live && lane < k )
         a = fma(-c, y, a);
   }
   return a * di;
}
/* (L w)_lane and (L^T v)_lane */
__device__ __forceinline__ double hs_wl_mulL(const double* sL, int m, int lane, double w)
{
   const bool live = lane < m;
   const double* row = sL + (live ? lane : 0) * 65;
   double acc = 0.0;
   for (int j = 0; j < m; ++j)
   {
      const double wj = hs_wl_bcast(w, j);
      const double c = row[j];
      if ( live && j <= lane )
         acc = fma(c, wj, acc);
   }
   return acc;
}
__device__ __forceinline__ double hs_wl_mulLt(const double* sL, int m, int lane, double v)
{
   const bool live = lane < m;
   const int col = live ? lane : 0;
   double acc = 0.0;
   for (int j = 0; j < m; ++j)
   {
      const double vj = hs_wl_bcast(v, j);
      const double c = sL[j * 65 + col];
      if ( live && j >= lane )
         acc = fma(c, vj, acc);
   }
   return acc;
}
/* the same for 64 < m <= 128: two rows per lane (lane and lane + 64), L at pitch P (odd) in LDS; x[0], x[1]: this lane's two entries */
template<int P>
__device__ __forceinline__ void hs_wl2_fwd(const double* sL, int m, int lane, const double (&di)[2], double (&a)[2])
{
   const int r0 = lane, r1 = lane + 64;
   const bool l1 = r1 < m;
   const double* row0 = sL + r0 * P;
   const double* row1 = sL + (l1 ? r1 : 0) * P;
   for (int k = 0; k < m; ++k)
   {
      const double y = (k < 64) ? hs_wl_bcast(a[0] * di[0], k) : hs_wl_bcast(a[1] * di[1], k - 64);
      const double c0 = row0[k], c1 = row1[k];
      if ( r0 > k ) a[0] = fma(-c0, y, a[0]);
      if ( l1 && r1 > k ) a[1] = fma(-c1, y, a[1]);
   }
   a[0] *= di[0]; a[1] *= di[1];
}
template<int P>
__device__ __forceinline__ void hs_wl2_bwd(const double* sL, int m, int lane, const double (&di)[2], double (&a)[2])
{
   const int r0 = lane, r1 = lane + 64;
   const bool l1 = r1 < m;
   const int c1col = l1 ? r1 : 0;
   for (int k = m - 1; k >= 0; --k)
   {
      const double y = (k < 64) ? hs_wl_bcast(a[0] * di[0], k) : hs_wl_bcast(a[1] * di[1], k - 64);
      const double c0 = sL[k * P + r0], c1 = sL[k * P + c1col];
      if ( r0 < k ) a[0] = fma(-c0, y, a[0]);
      if ( l1 && r1 < k ) a[1] = fma(-c1, y, a[1]);
   }
   a[0] *= di[0]; a[1] *= di[1];
}
template<int P, bool TRANS>
__device__ __forceinline__ void hs_wl2_mul(const double* sL, int m, int lane, const double (&w)[2], double (&out)[2])
{
   const int r0 = lane, r1 = lane + 64;
   const bool l1 = r1 < m;
   const int q1 = l1 ? r1 : 0;
   double s0 = 0.0, s1 = 0.0;
   for (int j = 0; j < m; ++j)
   {
      const double wj = (j < 64) ? hs_wl_bcast(w[0], j) : hs_wl_bcast(w[1], j - 64);
      const double c0 = TRANS ? sL[j * P + r0] : sL[r0 * P + j];
      const double c1 = TRANS ? sL[j * P + q1] : sL[q1 * P + j];
      if ( TRANS ? j >= r0 : j <= r0 ) s0 = fma(c0, wj, s0);
      if ( l1 && (TRANS ? j >= r1 : j <= r1) ) s1 = fma(c1, wj, s1);
   }
   out[0] = s0; out[1] = s1;
}
/* sL must hold zeros above the diagonal (the transposed accesses of rows past a lane's own are masked, the others read them) */
template<int P>
__device__ __forceinline__ void hs_wl2_msolve(const double* sL, int m, int lane, double (&x)[2])
{
   const int r1 = lane + 64;
   double di[2] = {1.0 / sL[lane * P + lane], r1 < m ? 1.0 / sL[r1 * P + r1] : 0.0};
   double r[2] = {x[0], r1 < m ? x[1] : 0.0}, w[2] = {r[0], r[1]}, t[2], c[2];
   hs_wl2_fwd<P>(sL, m, lane, di, w);
   hs_wl2_mul<P, false>(sL, m, lane, w, t);
   c[0] = r[0] - t[0]; c[1] = r[1] - t[1];
   hs_wl2_fwd<P>(sL, m, lane, di, c);
   w[0] += c[0]; w[1] += c[1];
   double v[2] = {w[0], w[1]};
   hs_wl2_bwd<P>(sL, m, lane, di, v);
   hs_wl2_mul<P, true>(sL, m, lane, v, t);
   c[0] = w[0] - t[0]; c[1] = w[1] - t[1];
   hs_wl2_bwd<P>(sL, m, lane, di, c);
   x[0] = v[0] + c[0]; x[1] = v[1] + c[1];
}
__device__ __forceinline__ double hs_wl_msolve(const double* sL, int m, int lane, double r)
{
   const double di = lane < m ? 1.0 / sL[lane * 65 + lane] : 0.0;
   double w = hs_wl_fwd(sL, m, lane, di, r);
   w += hs_wl_fwd(sL, m, lane, di, r - hs_wl_mulL(sL, m, lane, w));
   double v = hs_wl_bwd(sL, m, lane, di, w);
   v += hs_wl_bwd(sL, m, lane, di, w - hs_wl_mulLt(sL, m, lane, v));
   return v;
}
#endif

struct hs_solve1_args
{
   int m, q, nblk;
   int n[HS_S1_MAXBLK];
   const double* A[HS_S1_MAXBLK];          /* dense (m + 1) x n^2 rows */
   double* X[HS_S1_MAXBLK];                /* n x n: start point in (have_start), final iterate out (unscaled) */
   double* Z[HS_S1_MAXBLK];
   double* Xpre[HS_S1_MAXBLK];             /* preoptimal iterate (preoptgap > 0) */
   const double* b; const double* Dext;
   double *y, *x, *z, *pre_y, *pre_x;
   const void* cmds; int ncmd;         /* deferred setters of the node (NodeCmd list in pinned memory), run before anything else */
   long long cmd_bytes;                /* bytes of the arena behind cmds that hold the list and the data of its commands (multiple of 16) */
   double *hy, *hx, *hz;               /* optional: y, x, z once more into pinned host memory (the caller's read-backs of a node need no copy) */
   double gaptol, feastol, infeastol, objlimit, timelimit, gamma, pabstol, preoptgap;
   double elapsed0;                        /* seconds of the time limit already used when the kernel starts */
   double maxwork;                         /* decline above this many multiply-adds per Schur assembly */
   int maxiter, settings, have_start, pivot_rule, prof_on, hist_len;
   int keep_on_fail;                       /* 1: on a numerical failure leave y, x, z, X, Z as they were handed in (the general path retries from them) */
   double* gws; long long gws_len;         /* workspace in device memory (hs_solve1_ws_doubles) */
   double* out;                            /* HS_S1_OUT_DOUBLES result scalars (device-visible; pinned host memory works) */
   double* hist;                           /* optional: 16 doubles per iteration (tests, tools) */
   unsigned long long seq; unsigned long long* flag;   /* when flag != NULL: *flag = seq once out[] is complete */
};
int hs_small_solve_by_substitution(void);
int hs_solve1_fits(int m, int q, int nblk, const int* n);
long long hs_solve1_ws_doubles(int m, int q, int nblk, const int* n);
int hs_solve1_launch(hipStream_t st, const hs_solve1_args* a);
int hs_solve1_class(const hs_solve1_args* a);
int hs_solve1_debug_counts(unsigned int* out2);

#endif
