/* gram.hip - the Gram product of the Schur assembly, Mx += W W^T on the lower triangle (W = [W_0; ..; W_m] as rows of n^2 doubles,
 * schur.hip GEMM3; in the reference this happens inside DSDP / SDPA: sdpisolver_dsdp.c:1503, sdpisolver_sdpa.cpp:1620), round 5.
 *
 * The persistent tile kernel of dgemm2.hip gives this product one item per workgroup: 36 lower tiles x 14 K slices at m = 1000.
 * Two things it leaves on the table:
 *  - a DIAGONAL tile is computed whole, 64 slab products per K step where the lower triangle has 36 (8 of 36 tiles at m = 1000,
 *    16 of 136 at m = 2000: 11 % / 6 % of the matrix instructions of the product);
 *  - with one size of item and one item per workgroup nothing can be balanced.
 * Here
 *  - a diagonal item keeps NINE accumulators per wavefront: wavefront w owns slab rows w and 7 - w of the tile, i.e. the slabs
 *    (w, 0 .. w) and (7 - w, 0 .. 7 - w) - nine for every w, the lower triangle exactly, 18 matrix instructions per stage instead
 *    of 32.  Both operands are the same 128 rows of W: only one operand image is loaded.  The code is the same for all wavefronts:
 *    the B fragment of accumulator k comes from an LDS address that depends on w, its A fragment is chosen between the two row
 *    slabs by a wave-uniform select;
 *  - the HOST cuts the product into items (tile, K range) once per shape and deals them out as STATIC lists, one per workgroup
 *    (items g.off[wg] .. g.off[wg + 1] of a small table in device memory; longest-processing-time scheduling within each XCD's
 *    share, gr_make_plan) - the kernel contains no atomic and takes nothing dynamically.  Every tile is cut into equal K slices,
 *    the diagonal tiles into fewer, longer ones (m = 1000: 28 x 15 + 8 x 9 = 492 items, one per workgroup, where 36 x 14 whole
 *    tiles were 504), sorted by their start in K so that an XCD's workgroups read the same rows of W at about the same K (once
 *    from HBM, then from that XCD's L2, as before).  Plans with more than one item per workgroup are built by the same code but
 *    not taken by default (the lists of an XCD's workgroups drift apart in K, gr_plan).  Partial tiles go to slabs and are
 *    summed in slab order by a second kernel (bitwise reproducible: the order of the sum is fixed by the plan).
 * Stage loop, LDS images and counted waits as in dgemm2.hip (two halves per stage, the DMA placed by hand behind the first matrix
 * instructions of the second half); no request crosses an item boundary.
 */
#include "hs_common.h"
#include <stdlib.h>
#include <stdio.h>
#include <map>
#include <mutex>
#include <utility>
#include <tuple>
#include <vector>
#include <queue>
#include <algorithm>
#include <functional>

typedef double gr_v4d __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* gr_lds_ptr;
typedef const __attribute__((address_space(1))) void* gr_gbl_ptr;

#define GR_BT   128
#define GR_BKS  8
#define GR_NS   4
#define GR_OPSZ (GR_BT * GR_BKS)
#define GR_SLOT (2 * GR_OPSZ)
#define GR_GPS  4
#define GR_UNI(x) __builtin_amdgcn_readfirstlane(x)

__device__ __attribute__((aligned(16))) double hs_gr_zero[2] = {0.0, 0.0};

/* one unit of work: the lower tile (ti, tj) over K positions [k0, k1), its partial result into slab `slab` */
struct gram_item
{
   int ti, tj;
   int k0, k1;
   int slab;
   int pad;
};

struct gram_args
{
   const double* W;        /* [M][ldw], K contiguous */
   long long ldw;
   int M;
   double* ws;             /* slabs of M x M doubles */
   const gram_item* items; /* the lists of the 512 workgroups, one behind the other */
   const int* off;         /* list of workgroup w of XCD x: items[off[64 x + w] .. off[64 x + w + 1]) */
};

template<int N> __device__ __forceinline__ void gr_wait_vm()
{
   asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

/* fragment of a K-contiguous stage image: element (row 16 t + (l & 15), k = 4 ks + (l >> 4)); images are stored [slab][k / 2][row][k & 1] */
__device__ __forceinline__ double gr_frag(const double* __restrict__ img, int t, int ks, int lane)
{
   const int k = 4 * ks + (lane >> 4);
   return img[t * 128 + (k >> 1) * 32 + (lane & 15) * 2 + (k & 1)];
}

__global__ void __launch_bounds__(256, 2) hs_gram_kernel(gram_args g)
{
   extern __shared__ __attribute__((aligned(1024))) double gr_smem[];
   const int tid = threadIdx.x, lane = tid & 63;
   const int wave = GR_UNI(tid >> 6);
   const int wm = wave >> 1, wn = wave & 1;
   const int wg = (blockIdx.x & 7) * 64 + (blockIdx.x >> 3);           /* (blockIdx & 7 labels the XCD: a speed assumption only) */
   const int lbeg = GR_UNI(g.off[wg]), lend = GR_UNI(g.off[wg + 1]);
   const int ar = lane & 15, ac = lane >> 4;

   /* accumulators: an off-diagonal item uses all 16 (slab (4 wm + i, 4 wn + j) in acc[4 i + j]), a diagonal item the first nine */
   gr_v4d acc[16];
#pragma unroll
   for (int i = 0; i < 16; ++i)
      acc[i] = (gr_v4d){0.0, 0.0, 0.0, 0.0};

   /* diagonal items: accumulator k of wavefront w is the slab (w, k) for k <= w and (7 - w, k - w - 1) behind that */
   int dcol[9];
   bool dlo[9];
#pragma unroll
   for (int k = 0; k < 9; ++k)
   {
      dlo[k] = k <= wave;
      dcol[k] = k <= wave ? k : k - wave - 1;
   }

   /* ---- the item's operand rows and K range; requests of stage `st` (0 .. nst - 1) into ring slot st & 3.  Items are long (hundreds
    * of stages): no request crosses an item boundary, the ring starts empty and is drained at the end of every item - which lets
    * a diagonal item request ONE operand image per stage (two pieces per wavefront instead of four) */
   const double* pa[2]; const double* pb[2];
   int ck0 = 0, ck1 = 0;
   pa[0] = pa[1] = pb[0] = pb[1] = hs_gr_zero;
   auto piece = [&](int st, int k) __attribute__((always_inline))
   {
      const int i = k & 1;
      const int pc = wave * 2 + i;
      const int K0 = ck0 + GR_BKS * st;
      const double* src = (K0 + 2 * ac < ck1) ? (k < 2 ? pa[i] : pb[i]) + K0 : hs_gr_zero;
      __builtin_amdgcn_global_load_lds((gr_gbl_ptr) src, (gr_lds_ptr) (gr_smem + (st & 3) * GR_SLOT + (k < 2 ? 0 : GR_OPSZ) + pc * 128), 16, 0, 0);
   };

   /* ---- off-diagonal item */
   auto run_off = [&](int nst) __attribute__((always_inline))
   {
      double f0a[4], f0b[4], f1a[4], f1b[4];
      auto landed_frags = [&](double (&fa)[4], double (&fb)[4]) __attribute__((always_inline))
      {
#pragma unroll
         for (int i = 0; i < 4; ++i)
            asm volatile("" : "+v"(fa[i]), "+v"(fb[i]));
      };
#pragma unroll
      for (int st = 0; st < GR_NS; ++st)
#pragma unroll
         for (int k = 0; k < 4; ++k)
            piece(st, k);
      gr_wait_vm<3 * 4>();
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int i = 0; i < 4; ++i)
      {
         f0a[i] = gr_frag(gr_smem, 4 * wm + i, 0, lane);
         f0b[i] = gr_frag(gr_smem + GR_OPSZ, 4 * wn + i, 0, lane);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      landed_frags(f0a, f0b);
      for (int s = 0; s < nst; ++s)
      {
         const double* sa = gr_smem + (s & 3) * GR_SLOT;
         const double* sn = gr_smem + ((s + 1) & 3) * GR_SLOT;
         /* first half: K step 0, the fragments of K step 1 between its matrix instructions */
         __builtin_amdgcn_sched_barrier(0);
#pragma unroll
         for (int i = 0; i < 4; ++i)
         {
            f1a[i] = gr_frag(sa, 4 * wm + i, 1, lane);
            f1b[i] = gr_frag(sa + GR_OPSZ, 4 * wn + i, 1, lane);
         }
#pragma unroll
         for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
               acc[4 * i + j] = __builtin_amdgcn_mfma_f64_16x16x4f64(f0a[i], f0b[j], acc[4 * i + j], 0, 0, 0);
#pragma unroll
         for (int q = 0; q < 8; ++q)
         {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
         }
         __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
         __builtin_amdgcn_sched_barrier(0);
         asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
         landed_frags(f1a, f1b);
         /* stage s + 1 must have landed; stages s + 2, s + 3 may stay in flight (requests past the item's end read the zero constant,
          * so the count is the same at the end of the item) */
         gr_wait_vm<2 * 4>();
         __builtin_amdgcn_s_barrier();
         /* second half: K step 1; two matrix instructions, then one request of stage s + 4 per matrix instruction, then the first
          * fragments of stage s + 1 */
         __builtin_amdgcn_sched_barrier(0);
         acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(f1a[0], f1b[0], acc[0], 0, 0, 0);
         acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f1a[0], f1b[1], acc[1], 0, 0, 0);
#pragma unroll
         for (int k = 0; k < 4; ++k)
         {
            __builtin_amdgcn_sched_barrier(0);
            piece(s + GR_NS, k);
            acc[2 + k] = __builtin_amdgcn_mfma_f64_16x16x4f64(f1a[(2 + k) >> 2], f1b[(2 + k) & 3], acc[2 + k], 0, 0, 0);
         }
         __builtin_amdgcn_sched_barrier(0);
#pragma unroll
         for (int i = 0; i < 4; ++i)
         {
            f0a[i] = gr_frag(sn, 4 * wm + i, 0, lane);
            f0b[i] = gr_frag(sn + GR_OPSZ, 4 * wn + i, 0, lane);
         }
#pragma unroll
         for (int k = 6; k < 16; ++k)
            acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(f1a[k >> 2], f1b[k & 3], acc[k], 0, 0, 0);
#pragma unroll
         for (int q = 0; q < 8; ++q)
         {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
         }
         __builtin_amdgcn_sched_barrier(0);
         asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
         landed_frags(f0a, f0b);
      }
   };

   /* ---- diagonal item: nine accumulators; per K step the two row-slab fragments and nine column-slab fragments (addresses by
    * wavefront); one operand image, two requests per wavefront and stage */
   auto run_diag = [&](int nst) __attribute__((always_inline))
   {
      double a0[2], b0[9], a1[2], b1[9];
      auto read_set = [&](double (&a)[2], double (&b)[9], const double* img, int ks) __attribute__((always_inline))
      {
         a[0] = gr_frag(img, wave, ks, lane);
         a[1] = gr_frag(img, 7 - wave, ks, lane);
#pragma unroll
         for (int k = 0; k < 9; ++k)
            b[k] = gr_frag(img, dcol[k], ks, lane);
      };
      auto landed_set = [&](double (&a)[2], double (&b)[9]) __attribute__((always_inline))
      {
         asm volatile("" : "+v"(a[0]), "+v"(a[1]));
#pragma unroll
         for (int k = 0; k < 9; ++k)
            asm volatile("" : "+v"(b[k]));
      };
      auto mfma_k = [&](const double (&a)[2], const double (&b)[9], int k) __attribute__((always_inline))
      {
         const double av = dlo[k] ? a[0] : a[1];
         acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b[k], acc[k], 0, 0, 0);
      };
#pragma unroll
      for (int st = 0; st < GR_NS; ++st)
#pragma unroll
         for (int k = 0; k < 2; ++k)
            piece(st, k);
      gr_wait_vm<3 * 2>();
      __builtin_amdgcn_s_barrier();
      read_set(a0, b0, gr_smem, 0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      landed_set(a0, b0);
      for (int s = 0; s < nst; ++s)
      {
         const double* sa = gr_smem + (s & 3) * GR_SLOT;
         const double* sn = gr_smem + ((s + 1) & 3) * GR_SLOT;
         __builtin_amdgcn_sched_barrier(0);
         read_set(a1, b1, sa, 1);
#pragma unroll
         for (int k = 0; k < 9; ++k)
            mfma_k(a0, b0, k);
#pragma unroll
         for (int q = 0; q < 5; ++q)
         {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
         }
         __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
         __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
         __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
         __builtin_amdgcn_sched_barrier(0);
         asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
         landed_set(a1, b1);
         gr_wait_vm<2 * 2>();
         __builtin_amdgcn_s_barrier();
         __builtin_amdgcn_sched_barrier(0);
         mfma_k(a1, b1, 0);
         mfma_k(a1, b1, 1);
#pragma unroll
         for (int k = 0; k < 2; ++k)
         {
            __builtin_amdgcn_sched_barrier(0);
            piece(s + GR_NS, k);
            mfma_k(a1, b1, 2 + k);
         }
         __builtin_amdgcn_sched_barrier(0);
         read_set(a0, b0, sn, 0);
#pragma unroll
         for (int k = 4; k < 9; ++k)
            mfma_k(a1, b1, k);
#pragma unroll
         for (int q = 0; q < 4; ++q)
         {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
         }
         __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
         __builtin_amdgcn_sched_barrier(0);
         asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
         landed_set(a0, b0);
      }
   };

   /* ---- the workgroup's list */
   for (int cur = lbeg; cur < lend; ++cur)
   {
      const gram_item it = g.items[cur];
      const int cti = GR_UNI(it.ti), ctj = GR_UNI(it.tj), cslab = GR_UNI(it.slab);
      ck0 = GR_UNI(it.k0); ck1 = GR_UNI(it.k1);
      const int nst = (ck1 - ck0 + GR_BKS - 1) / GR_BKS;
      const bool diag = cti == ctj;
#pragma unroll
      for (int i = 0; i < 2; ++i)
      {
         const int pc = wave * 2 + i;
         int row = cti * GR_BT + pc * 16 + ar;
         if ( row > g.M - 1 ) row = g.M - 1;
         pa[i] = g.W + (long long) row * g.ldw + 2 * ac;
         int rowb = ctj * GR_BT + pc * 16 + ar;
         if ( rowb > g.M - 1 ) rowb = g.M - 1;
         pb[i] = g.W + (long long) rowb * g.ldw + 2 * ac;
      }
      if ( diag )
         run_diag(nst);
      else
         run_off(nst);
      gr_wait_vm<0>();
      /* partial tile -> its slab; the accumulators start the next item at zero */
      {
         double* C = g.ws + (long long) cslab * (long long) g.M * g.M;
         const int m0 = cti * GR_BT, n0 = ctj * GR_BT;
         if ( !diag )
         {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
               for (int r = 0; r < 4; ++r)
               {
                  const int row = m0 + wm * 64 + 16 * i + (lane >> 4) + 4 * r;
                  if ( row < g.M )
                  {
                     double* cr = C + (long long) row * g.M + n0 + wn * 64 + (lane & 15);
#pragma unroll
                     for (int j = 0; j < 4; ++j)
                        cr[16 * j] = acc[4 * i + j][r];
                  }
               }
         }
         else
         {
#pragma unroll
            for (int k = 0; k < 9; ++k)
            {
               const int rs = dlo[k] ? wave : 7 - wave;
#pragma unroll
               for (int r = 0; r < 4; ++r)
               {
                  const int row = m0 + 16 * rs + (lane >> 4) + 4 * r;
                  const int col = m0 + 16 * dcol[k] + (lane & 15);
                  if ( row < g.M && col < g.M )
                     C[(long long) row * g.M + col] = acc[k][r];
               }
            }
         }
#pragma unroll
         for (int i = 0; i < 16; ++i)
            acc[i] = (gr_v4d){0.0, 0.0, 0.0, 0.0};
      }
      /* (separates this item's last LDS reads from the next item's first requests) */
      __syncthreads();
   }
}

/* C[r][c] = alpha * (sum over the partial tiles of its tile, in slab order) + beta * C[r][c] for r >= c; a diagonal tile has nd
 * partial tiles (slabs 0 .. nd - 1), an off-diagonal one no */
__global__ void hs_gram_reduce_kernel(int M, int no, int nd, const double* __restrict__ ws, double* __restrict__ C, long long ldc, double alpha, double beta)
{
   const long long MM = (long long) M * M;
   for (long long e = (long long) blockIdx.x * blockDim.x + threadIdx.x; e < MM; e += (long long) gridDim.x * blockDim.x)
   {
      const int r = (int) (e / M), c = (int) (e - (long long) r * M);
      if ( c > r )
         continue;
      const int ns = (r / GR_BT == c / GR_BT) ? nd : no;
      double s = 0.0;
      for (int k = 0; k < ns; ++k)
         s += ws[(long long) k * MM + e];
      double* q = C + (long long) r * ldc + c;
      *q = beta != 0.0 ? alpha * s + beta * (*q) : alpha * s;
   }
}

static int gr_mode = -1;

/* test hook: 0 off, 1 on (default unless HIPSDP_GRAM=0); returns the previous mode */
int hs_gram_enable(int on)
{
   const int prev = gr_mode;
   gr_mode = on ? 1 : 0;
   return prev;
}

namespace {

/* how the product is cut into items and who computes them, decided on the host once per shape */
struct GramPlan
{
   std::vector<gram_item> items;       /* the lists of the 512 workgroups, one behind the other */
   std::vector<int> off;               /* 513 offsets */
   int no, nd;                         /* partial tiles per off-diagonal / diagonal tile (slabs 0 .. no - 1 / 0 .. nd - 1) */
   double span;                        /* estimated time in units of one off-diagonal tile over all of K */
   double executed;                    /* FP64 matrix-core flops of one launch */
   gram_item* dev;                     /* the table on the device */
   int* devoff;
};

/* cost of a diagonal stage relative to an off-diagonal one: 18 of 32 matrix instructions, one of two operand images; measured 0.71
 * (the barrier, the waits and 11 fragment reads per K step do not shrink with the matrix instructions) */
double gr_diag_cost(void)
{
   static double c = -1.0;
   if ( c < 0.0 )
   {
      const char* env = getenv("HIPSDP_GRAM_DIAGCOST");
      c = env != NULL && atof(env) > 0.0 ? atof(env) : 0.72;
   }
   return c;
}

/* Every tile in equal K slices, so for the off-diagonal tiles and sd <= so for the diagonal ones (their stages are cheaper).  The
 * items sorted by their start in K are dealt to the XCDs in runs of equal cost, so that the workgroups of an XCD read the same rows
 * of W at about the same K (once from HBM, then from that XCD's L2); inside an XCD the longest item goes to the workgroup with the
 * least work so far, and a workgroup walks its items in K order. */
bool gr_make_plan(GramPlan& P, int tm, long long K, int so, int sd)
{
   const int noff = tm * (tm - 1) / 2;
   const double cd = gr_diag_cost();
   if ( so < 1 || sd < 1 )
      return false;
   long long lo = (K + so - 1) / so, ld = (K + sd - 1) / sd;
   lo = ((lo + GR_BKS - 1) / GR_BKS) * GR_BKS;
   ld = ((ld + GR_BKS - 1) / GR_BKS) * GR_BKS;
   if ( (long long) (so - 1) * lo + 8 * GR_BKS > K || (long long) (sd - 1) * ld + 8 * GR_BKS > K )
      return false;
   std::vector<gram_item> all;
   for (int s = 0; s < so; ++s)
      for (int i = 1; i < tm; ++i)
         for (int j = 0; j < i; ++j)
         {
            gram_item it = {i, j, (int) (s * lo), (int) ((s + 1) * lo < K ? (s + 1) * lo : K), s, 0};
            all.push_back(it);
         }
   for (int s = 0; s < sd; ++s)
      for (int i = 0; i < tm; ++i)
      {
         gram_item it = {i, i, (int) (s * ld), (int) ((s + 1) * ld < K ? (s + 1) * ld : K), s, 0};
         all.push_back(it);
      }
   (void) noff;
   std::stable_sort(all.begin(), all.end(), [](const gram_item& a, const gram_item& b) { return a.k0 < b.k0; });
   auto cost = [&](const gram_item& it) { return (it.ti == it.tj ? cd : 1.0) * (double) (it.k1 - it.k0) / (double) K; };
   double total = 0.0;
   P.executed = 0.0;
   for (const gram_item& it : all)
   {
      total += cost(it);
      P.executed += 2048.0 * (double) ((it.k1 - it.k0 + GR_BKS - 1) / GR_BKS) * (it.ti == it.tj ? 72.0 : 128.0);
   }
   P.items.clear();
   P.off.assign(513, 0);
   P.span = 0.0;
   size_t pos = 0;
   double done = 0.0;
   for (int x = 0; x < 8; ++x)
   {
      /* the XCD's run of the sorted list: up to its share of the total cost */
      std::vector<gram_item> mine;
      const double upto = total * (double) (x + 1) / 8.0;
      while ( pos < all.size() && (x == 7 || done + 0.5 * cost(all[pos]) <= upto) )
      {
         done += cost(all[pos]);
         mine.push_back(all[pos++]);
      }
      /* longest first onto the least loaded of the 64 workgroups */
      std::vector<size_t> order(mine.size());
      for (size_t i = 0; i < order.size(); ++i) order[i] = i;
      std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return cost(mine[a]) > cost(mine[b]); });
      std::vector<std::vector<gram_item> > lists(64);
      std::vector<double> load(64, 0.0);
      for (size_t i : order)
      {
         int w = 0;
         for (int v = 1; v < 64; ++v)
            if ( load[v] < load[w] ) w = v;
         load[w] += cost(mine[i]);
         lists[w].push_back(mine[i]);
      }
      for (int w = 0; w < 64; ++w)
      {
         std::stable_sort(lists[w].begin(), lists[w].end(), [](const gram_item& a, const gram_item& b) { return a.k0 < b.k0; });
         P.off[64 * x + w] = (int) P.items.size();
         for (const gram_item& it : lists[w])
            P.items.push_back(it);
         if ( load[w] > P.span ) P.span = load[w];
      }
   }
   P.off[512] = (int) P.items.size();
   P.no = so; P.nd = sd;
   return true;
}

/* best plan for the shape with at most nslab partial tiles per tile (NULL: none) */
GramPlan* gr_plan(int M, long long K, int nslab)
{
   static std::mutex mu;
   static std::map<std::tuple<int, int, long long, int>, GramPlan*> tab;
   int dev = 0;
   if ( hipGetDevice(&dev) != hipSuccess )
      return NULL;
   std::lock_guard<std::mutex> lk(mu);
   const auto key = std::make_tuple(dev, M, K, nslab);
   auto f = tab.find(key);
   if ( f != tab.end() )
      return f->second;
   const int tm = (M + GR_BT - 1) / GR_BT;
   const int ntile = tm * (tm + 1) / 2;
   GramPlan* best = NULL;
   GramPlan cand;
   cand.dev = NULL; cand.devoff = NULL;
   auto consider = [&](bool ok, double penalty)
   {
      if ( !ok )
         return;
      cand.span += penalty;
      if ( best == NULL || cand.span < best->span )
      {
         if ( best == NULL )
            best = new GramPlan();
         *best = cand;
      }
   };
   const char* force = getenv("HIPSDP_GRAM_PLAN");            /* "so sd": developer override */
   int a = 0, b = 0;
   if ( force != NULL && sscanf(force, "%d %d", &a, &b) == 2 )
      consider(gr_make_plan(cand, tm, K, a, b), 0.0);
   else
   {
      /* Candidates: ONE item per workgroup (at most 512 items).  With several items per workgroup the lists of an XCD's workgroups
       * drift apart in K and the rows of W are no longer shared through its L2: measured at m = 2000 (136 tiles, 8 items per
       * workgroup, balanced to 3 % below the K-sliced tile kernel's estimate) 68.6 against 66.3 ms - the tile kernel's rounds of
       * equal items stay in step by construction, and the diagonal tiles are only 6 % of that product.  The plan is taken when its
       * estimate beats the tile kernel's (whole rounds of 512 equal items) by 3 %.  Every partial tile is stored by its workgroup
       * and read again by the summation kernel: priced at 0.24 K positions of one tile. */
      const int noff = ntile - tm;
      for (int so = 1; so <= nslab && (long long) noff * so + tm <= 512; ++so)
         for (int sd = (so + 1) / 2; sd <= so && (long long) noff * so + (long long) tm * sd <= 512; ++sd)
            consider(gr_make_plan(cand, tm, K, so, sd), 0.24 * ((double) so * noff + (double) sd * tm) / (double) K);
      if ( best != NULL )
      {
         const int sold = hs_dgemm_pick_xcd_slices(ntile, K);
         const double oldspan = (double) (((long long) ntile * sold + 511) / 512) / (double) sold;
         if ( best->span * 1.03 > oldspan || best->items.size() < 256 )
         {
            delete best;
            best = NULL;
         }
      }
   }
   if ( best != NULL )
   {
      gram_item* d = NULL;
      int* doff = NULL;
      if ( hipMalloc((void**) &d, best->items.size() * sizeof(gram_item)) != hipSuccess
         || hipMemcpy(d, best->items.data(), best->items.size() * sizeof(gram_item), hipMemcpyHostToDevice) != hipSuccess
         || hipMalloc((void**) &doff, 513 * sizeof(int)) != hipSuccess
         || hipMemcpy(doff, best->off.data(), 513 * sizeof(int), hipMemcpyHostToDevice) != hipSuccess )
      {
         if ( d != NULL ) (void) hipFree(d);
         if ( doff != NULL ) (void) hipFree(doff);
         delete best;
         return NULL;
      }
      best->dev = d;
      best->devoff = doff;
   }
   tab[key] = best;
   return best;
}

}

/* C (M x M, lower triangle) = alpha W W^T + beta C with W [M][K] (K contiguous, leading dimension ldw), through `nslab` slabs of
 * M x M doubles at ws.  1: done, 0: not eligible (the caller takes hs_dgemm), < 0: error code negated.  *executed: the FP64
 * matrix-core flops the launch executes. */
int hs_gram_try(hipStream_t stream, int M, long long K, const double* W, long long ldw, double* C, long long ldc, double alpha, double beta,
   double* ws, int nslab, double* executed)
{
   if ( gr_mode < 0 )
   {
      const char* env = getenv("HIPSDP_GRAM");
      gr_mode = (env != NULL && env[0] == '0') ? 0 : 1;
   }
   if ( !gr_mode || ws == NULL || nslab < 2 || M < 256 || K < 16384 )
      return 0;
   if ( (ldw & 1) || (K & 1) || (((uintptr_t) W) & 15) || K > 2000000000LL )
      return 0;
   if ( nslab > 64 ) nslab = 64;
   GramPlan* P = gr_plan(M, K, nslab);
   if ( P == NULL || P->dev == NULL || P->items.size() < 256 )
      return 0;
   gram_args g;
   g.W = W; g.ldw = ldw; g.M = M; g.ws = ws; g.items = P->dev; g.off = P->devoff;
   static hs_attr_mask attr_done;
   const size_t smem = (size_t) GR_NS * GR_SLOT * sizeof(double);
   if ( hs_func_max_lds(reinterpret_cast<const void*>(&hs_gram_kernel), (int) smem, &attr_done) != HS_OK )
      return -HS_ERR_HIP;
   hipLaunchKernelGGL(hs_gram_kernel, dim3(512), dim3(256), smem, stream, g);
   if ( hipGetLastError() != hipSuccess )
      return -HS_ERR_HIP;
   {
      int blocks = (int) (((long long) M * M + 255) / 256);
      if ( blocks > 2048 ) blocks = 2048;
      hipLaunchKernelGGL(hs_gram_reduce_kernel, dim3(blocks), dim3(256), 0, stream, M, P->no, P->nd, ws, C, ldc, alpha, beta);
      if ( hipGetLastError() != hipSuccess )
         return -HS_ERR_HIP;
   }
   if ( executed != NULL )
      *executed = P->executed;
   return 1;
}

/* developer tool: the plan chosen for a shape */
int hs_gram_plan_info(int M, long long K, int nslab, int* no, int* nd, int* nitems, double* span)
{
   GramPlan* P = gr_plan(M, K, nslab > 64 ? 64 : nslab);
   if ( P == NULL )
      return 0;
   *no = P->no; *nd = P->nd; *nitems = (int) P->items.size(); *span = P->span;
   return 1;
}
