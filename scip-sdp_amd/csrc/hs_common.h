/* hs_common.h - shared declarations of the hipsdp device engine (gfx950 / MI355X only).
 *
 * Everything in csrc/ is written for CDNA4: 64-lane wavefronts, v_mfma_f64_16x16x4_f64, 160 KiB LDS per CU,
 * 8 XCDs with private L2.  There is no CPU fallback in this directory: every entry point that computes
 * returns HS_ERR_NODEVICE when no HIP device is usable.
 */
#ifndef HS_COMMON_H
#define HS_COMMON_H

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#define HS_OK            0
#define HS_ERR_NODEVICE  1
#define HS_ERR_HIP       2
#define HS_ERR_ARG       3
#define HS_ERR_NOMEM     4
#define HS_ERR_NUMERIC   5

#define HS_HIP(call)                                                                       \
   do {                                                                                    \
      hipError_t hs_e_ = (call);                                                           \
      if ( hs_e_ != hipSuccess ) {                                                         \
         hs_record_hip_error(hs_e_, #call, __FILE__, __LINE__);                            \
         return (hs_e_ == hipErrorOutOfMemory) ? HS_ERR_NOMEM : HS_ERR_HIP;                \
      }                                                                                    \
   } while (0)

#define HS_CALL(call)                                                                      \
   do {                                                                                    \
      int hs_r_ = (call);                                                                  \
      if ( hs_r_ != HS_OK )                                                                \
         return hs_r_;                                                                     \
   } while (0)

void hs_record_hip_error(hipError_t e, const char* what, const char* file, int line);
const char* hs_last_error(void);

/* hipFuncAttributeMaxDynamicSharedMemorySize belongs to the (kernel, device) pair, not to the process: `done` is a per-kernel
 * mask of the devices the attribute has been set on (static storage of the caller, zero-initialised; devices 0..127).  Safe from
 * several host threads (each with its own current device). */
struct hs_attr_mask { unsigned long long bits[2]; };
int hs_func_max_lds(const void* fn, int bytes, hs_attr_mask* done);
int hs_func_attr_sets(int device);      /* how many (kernel, device) attributes have been set on that device so far */
int hs_device_cus(void);                /* compute units of the current device (cached per device) */

/* recycling allocator for device blocks of at most 4 MiB (hs_util.cpp); larger requests go to hipMalloc / hipFree */
int hs_pool_alloc(void** p, size_t bytes);
void hs_pool_free(void* p);
void hs_pool_trim(void);

/* operand storage of a GEMM operand as seen from the product C[M x N] = A[M x K] * B[K x N]:
 *  HS_KC: the K index is contiguous in memory (A stored row-major [M][K], or B stored as [N][K])
 *  HS_MC: the M (resp. N) index is contiguous   (A stored as [K][M],      or B stored row-major [K][N]) */
#define HS_KC 0
#define HS_MC 1

#define HS_GEMM_LOWER     1   /* compute only tiles that touch the lower triangle (row >= col) of C */
#define HS_GEMM_A_LOWTRI  2   /* A[m][k] = 0 for k > m (lower triangular left factor): tiles stop at k = m0 + tile */
#define HS_GEMM_B_LOWTRI  4   /* B[k][n] = 0 for k < n (lower triangular right factor): tiles start at k = n0 */
#define HS_GEMM_A_UPTRI 256   /* A[m][k] = 0 for k < m (upper triangular left factor): tiles start at k = m0 */
#define HS_GEMM_XCD       8   /* split-K only: workgroups that share an XCD (blockIdx % 8) walk the same K range, so the
                               * operand panels are fetched from HBM once per XCD and re-used from its L2 */
#define HS_GEMM_TILE64  128   /* 64 x 64 tiles whatever the size: for products whose N is a narrow column slice (<= 64 columns) */
#define HS_GEMM_NOFAST   64   /* always use the bounds-checked tile loads */
#define HS_GEMM_UPPER    32   /* compute only tiles that touch the upper triangle (col >= row) of C */
#define HS_GEMM_REMAP    16   /* XCD-contiguous tile order for batched / stack products: the tiles that share an operand
                               * panel (same batch entry, same row tile) run on one XCD */
/* number of K slices for an XCD-sliced product with ntile output tiles: fills the 512 workgroup slots of the chip (2 per
 * CU) in whole rounds */
int hs_dgemm_pick_xcd_slices(long long ntile, long long K);

struct hs_gemm_args
{
   int            M, N, K;
   int            layA, layB;       /* HS_KC / HS_MC */
   const double*  A;
   long long      lda;              /* leading dimension in doubles */
   long long      strideA;          /* batch stride in doubles (0: shared operand) */
   const double*  B;
   long long      ldb;
   long long      strideB;
   double*        C;                /* row-major [M][N] */
   long long      ldc;
   long long      strideC;
   double         alpha, beta;
   int            batch;
   int            flags;
   int            splitk;           /* >1: K is cut into splitk slices; needs ws of splitk*M*N doubles; batch must be 1 */
   double*        ws;
};

/* C = alpha * A * B + beta * C on the given stream; FP64 MFMA tiles (dgemm.hip) */
int hs_dgemm(hipStream_t stream, const hs_gemm_args* args);
/* persistent LDS-DMA variant for the 128-tile shapes (dgemm2.hip): 1 launched, 0 not eligible, < 0 error (negated code) */
int hs_dgemm2_try(hipStream_t stream, const hs_gemm_args* args, int kchunk);
int hs_dgemm2_enable(int on);

/* the Gram product of the Schur assembly (gram.hip): C (M x M, lower triangle) = alpha W W^T + beta C with W [M][K] K contiguous,
 * through nslab >= 8 slabs of M x M doubles at ws; diagonal tiles at 9 / 16 of the matrix instructions, a static list of items per
 * workgroup built on the host (no atomics).  1: done, 0: not eligible (the caller takes hs_dgemm with HS_GEMM_LOWER), < 0: error code negated */
int hs_gram_try(hipStream_t stream, int M, long long K, const double* W, long long ldw, double* C, long long ldc, double alpha, double beta,
   double* ws, int nslab, double* executed);
int hs_gram_enable(int on);
int hs_gram_plan_info(int M, long long K, int nslab, int* no, int* nd, int* nitems, double* span);

/* latency-oriented 32 x 32 kernel with the K split inside the workgroup (dgemm3.hip) for products of few tiles: 1 launched,
 * 0 not eligible, < 0 error (negated code); hs_dgemm tries it first for products without split-K */
int hs_dgemm3_try(hipStream_t stream, const hs_gemm_args* args);
int hs_dgemm3_enabled(void);

/* FP64 matrix-core flops the GEMM launches of the calling host thread have EXECUTED so far (what the MFMA pipes are issued, as
 * opposed to the algorithmic count 2 M N K): whole tiles (edge tiles are computed padded), over the K range each tile walks
 * (triangular operands, split-K slices) rounded up to the kernel's K stage, minus the all-zero 16 x 4 slabs the persistent kernel
 * skips inside the diagonal band.  The engine reads the difference around a Schur assembly (hipsdp_info.schur_flops_executed). */
double hs_mfma_flops_total(void);
void   hs_mfma_flops_add(double flops);
int hs_dgemm2_slabskip(void);            /* 0: no skipping of zero slabs, 1: skipping instances of round 3, 2: paired-band kernel (default) */
int hs_dgemm2_tri5_eligible(const hs_gemm_args* a);
double hs_gemm_executed_flops(const hs_gemm_args* a, int BT, int kstage, int kchunk, int slabskip);

/* choose a split-K factor for a [M x N x K] product so that at least ~2 waves of workgroups exist */
int hs_dgemm_pick_splitk(int M, int N, int K, int lowerOnly);

#endif
