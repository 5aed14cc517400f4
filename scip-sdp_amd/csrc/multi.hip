/* multi.hip - inter-rank plumbing (one process per GPU).
 *
 * Two transports sit behind the same four collectives the engine uses (all-gather of Schur row chunks, broadcast of the
 * scalar block and the flags):
 *   - RCCL over xGMI: the production transport.  The communicator is created from a unique id that the launcher
 *     (bench.py via torch.distributed, or any MPI-like bootstrap) broadcasts.
 *   - host-staged: a POSIX shared-memory segment with a generation barrier; payloads go device -> segment -> device.
 *     RCCL refuses two ranks on one device, so this is how the sharded path is exercised by several processes on a
 *     one-GPU box (tests/test_gpu_multi.py).  It moves bytes only - every flop still runs on the device. */
#include "hs_kernels.h"
#include "../../include/hipsdp.h"
#include <rccl/rccl.h>
#include <atomic>
#include <chrono>
#include <cstring>
#include <cstdlib>
#include <cstdio>
#include <ctime>
#include <new>
#include <vector>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

struct shm_header
{
   std::atomic<int> arrived;
   std::atomic<int> generation;
   std::atomic<int> failed;
   int pad0;
   std::atomic<unsigned long long> ready;    /* magic ^ job nonce once rank 0 has initialised the header */
   long long created_ns;                     /* CLOCK_REALTIME at creation: leftovers older than the join timeout are ignored */
   int pad[8];
};
static_assert(sizeof(shm_header) == 64, "shm_header is one cache line");

#define HS_COMM_PHASES 4      /* 0 Schur exchange, 1 passes over A, 2 decision scalars / flags, 3 other */

struct comm_rec { int phase; hipEvent_t a, b; hipStream_t st; };

struct hs_comm
{
   int kind;               /* 0 = RCCL, 1 = host-staged, 2 = measurement transport */
   int rank, nranks;
   ncclComm_t nccl;
   /* optional statistics (hipsdp_comm_stats_enable): an event pair around every collective, resolved when the numbers are read */
   bool stats;
   std::vector<comm_rec> recs;
   std::vector<hipEvent_t> evpool;
   double sec[HS_COMM_PHASES];
   long long calls[HS_COMM_PHASES];
   double bytes[HS_COMM_PHASES];
   /* host-staged */
   char name[128];
   shm_header* hdr;
   char* data;
   size_t data_bytes;
   size_t map_bytes;
   double timeout_s;
};

/* generation barrier; a rank that times out marks the segment failed so that the others leave as well */
int shm_barrier(hs_comm* c)
{
   shm_header* h = c->hdr;
   const int gen = h->generation.load(std::memory_order_acquire);
   if ( h->arrived.fetch_add(1, std::memory_order_acq_rel) == c->nranks - 1 )
   {
      h->arrived.store(0, std::memory_order_relaxed);
      h->generation.store(gen + 1, std::memory_order_release);
      return HS_OK;
   }
   const auto t0 = std::chrono::steady_clock::now();
   long long spins = 0;
   while ( h->generation.load(std::memory_order_acquire) == gen )
   {
      if ( h->failed.load(std::memory_order_relaxed) )
         return HS_ERR_HIP;
      if ( (++spins & 1023) == 0 )
      {
         const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
         if ( el > c->timeout_s )
         {
            h->failed.store(1, std::memory_order_relaxed);
            fprintf(stderr, "hipsdp: host-staged communicator: rank %d waited %.0f s at a barrier\n", c->rank, el);
            return HS_ERR_HIP;
         }
         sched_yield();
      }
   }
   return HS_OK;
}

/* all ranks contribute bytes_per_rank from send (device); everyone receives the concatenation in recv (device) */
int shm_allgather(hs_comm* c, const char* send, char* recv, size_t bytes_per_rank, hipStream_t stream)
{
   const size_t piece_max = c->data_bytes / (size_t) c->nranks & ~(size_t) 7;
   if ( piece_max == 0 )
      return HS_ERR_ARG;
   for (size_t off = 0; off < bytes_per_rank; off += piece_max)
   {
      const size_t piece = bytes_per_rank - off < piece_max ? bytes_per_rank - off : piece_max;
      HS_HIP( hipMemcpyAsync(c->data + (size_t) c->rank * piece, send + off, piece, hipMemcpyDeviceToHost, stream) );
      HS_HIP( hipStreamSynchronize(stream) );
      HS_CALL( shm_barrier(c) );
      for (int r = 0; r < c->nranks; ++r)
         HS_HIP( hipMemcpyAsync(recv + (size_t) r * bytes_per_rank + off, c->data + (size_t) r * piece, piece, hipMemcpyHostToDevice, stream) );
      HS_HIP( hipStreamSynchronize(stream) );
      HS_CALL( shm_barrier(c) );       /* nobody overwrites the segment before everyone has read it */
   }
   return HS_OK;
}

int shm_bcast(hs_comm* c, char* buf, size_t bytes, hipStream_t stream)
{
   for (size_t off = 0; off < bytes; off += c->data_bytes)
   {
      const size_t piece = bytes - off < c->data_bytes ? bytes - off : c->data_bytes;
      if ( c->rank == 0 )
      {
         HS_HIP( hipMemcpyAsync(c->data, buf + off, piece, hipMemcpyDeviceToHost, stream) );
         HS_HIP( hipStreamSynchronize(stream) );
      }
      HS_CALL( shm_barrier(c) );
      if ( c->rank != 0 )
      {
         HS_HIP( hipMemcpyAsync(buf + off, c->data, piece, hipMemcpyHostToDevice, stream) );
         HS_HIP( hipStreamSynchronize(stream) );
      }
      HS_CALL( shm_barrier(c) );
   }
   return HS_OK;
}

/* all-to-all with the count matrix cnt[src * nranks + dst] (doubles) known to every rank: a rank's send buffer holds its pieces
 * in destination order, its receive buffer the pieces in source order.  One source rank at a time goes through the segment. */
int shm_alltoall(hs_comm* c, const double* send, double* recv, const long long* cnt, hipStream_t stream)
{
   const int G = c->nranks;
   const long long cap = (long long) (c->data_bytes / sizeof(double));
   if ( cap <= 0 )
      return HS_ERR_ARG;
   long long recvoff = 0;                  /* where the piece of the current source starts in my receive buffer */
   for (int src = 0; src < G; ++src)
   {
      long long total = 0, mineoff = 0;
      for (int d = 0; d < G; ++d)
      {
         if ( d == c->rank ) mineoff = total;
         total += cnt[(long long) src * G + d];
      }
      const long long minecnt = cnt[(long long) src * G + c->rank];
      for (long long off = 0; off < total; off += cap)
      {
         const long long piece = total - off < cap ? total - off : cap;
         if ( c->rank == src )
         {
            HS_HIP( hipMemcpyAsync(c->data, send + off, (size_t) piece * sizeof(double), hipMemcpyDeviceToHost, stream) );
            HS_HIP( hipStreamSynchronize(stream) );
         }
         HS_CALL( shm_barrier(c) );
         /* my part of the window [off, off + piece) of the source's send buffer */
         const long long lo = mineoff > off ? mineoff : off;
         const long long hi = mineoff + minecnt < off + piece ? mineoff + minecnt : off + piece;
         if ( hi > lo )
         {
            HS_HIP( hipMemcpyAsync(recv + recvoff + (lo - mineoff), c->data + (size_t) (lo - off) * sizeof(double),
                  (size_t) (hi - lo) * sizeof(double), hipMemcpyHostToDevice, stream) );
            HS_HIP( hipStreamSynchronize(stream) );
         }
         HS_CALL( shm_barrier(c) );
      }
      recvoff += minecnt;
   }
   return HS_OK;
}

/* buf[i] = sum over ranks of part[r][i], added in rank order on every rank (identical bits everywhere) */
__global__ void k_sum_parts(long long count, int nparts, const double* __restrict__ part, double* __restrict__ buf)
{
   for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long long) gridDim.x * blockDim.x)
   {
      double acc = part[i];
      for (int r = 1; r < nparts; ++r)
         acc += part[(long long) r * count + i];
      buf[i] = acc;
   }
}

thread_local int g_comm_phase = 3;

hipEvent_t stats_event(hs_comm* c)
{
   hipEvent_t e = NULL;
   if ( !c->evpool.empty() )
   {
      e = c->evpool.back();
      c->evpool.pop_back();
   }
   else if ( hipEventCreate(&e) != hipSuccess )
      e = NULL;
   return e;
}

/* brackets one collective with an event pair on its stream when statistics are on */
struct CollTimer
{
   hs_comm* c; comm_rec r; bool on;
   CollTimer(hs_comm* c_, hipStream_t st, double nbytes) : c(c_), on(false)
   {
      if ( c == NULL || !c->stats )
         return;
      const int ph = g_comm_phase >= 0 && g_comm_phase < HS_COMM_PHASES ? g_comm_phase : HS_COMM_PHASES - 1;
      c->calls[ph]++;
      c->bytes[ph] += nbytes;
      r.phase = ph; r.st = st;
      r.a = stats_event(c); r.b = stats_event(c);
      if ( r.a == NULL || r.b == NULL )
         return;
      on = hipEventRecord(r.a, st) == hipSuccess;
   }
   ~CollTimer()
   {
      if ( on && hipEventRecord(r.b, r.st) == hipSuccess )
         c->recs.push_back(r);
   }
};

} /* namespace */

/* the phase the next collectives of this thread are booked under (0 Schur exchange, 1 passes over A, 2 decision scalars) */
void hs_comm_phase(int phase) { g_comm_phase = phase; }

extern "C" int hipsdp_comm_stats_enable(void* comm, int on)
{
   hs_comm* c = (hs_comm*) comm;
   if ( c == NULL )
      return HIPSDP_ERR_ARG;
   c->stats = on != 0;
   return HIPSDP_OK;
}

/* seconds[4], calls[4], bytes[4] by phase since the last reset (device time between the event pairs; waits for the streams) */
extern "C" int hipsdp_comm_stats(void* comm, double* seconds, long long* calls, double* bytes, int reset)
{
   hs_comm* c = (hs_comm*) comm;
   if ( c == NULL )
      return HIPSDP_ERR_ARG;
   for (auto& r : c->recs)
   {
      float ms = 0.f;
      if ( hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess )
         c->sec[r.phase] += 1e-3 * (double) ms;
      c->evpool.push_back(r.a);
      c->evpool.push_back(r.b);
   }
   c->recs.clear();
   for (int p = 0; p < HS_COMM_PHASES; ++p)
   {
      if ( seconds != NULL ) seconds[p] = c->sec[p];
      if ( calls != NULL ) calls[p] = c->calls[p];
      if ( bytes != NULL ) bytes[p] = c->bytes[p];
      if ( reset )
      {
         c->sec[p] = 0.0; c->calls[p] = 0; c->bytes[p] = 0.0;
      }
   }
   return HIPSDP_OK;
}

/* what the transport itself says about the communicator: *count = ncclCommCount for RCCL (the number of ranks RCCL joined),
 * the configured number otherwise; *kind = 0 RCCL, 1 host-staged, 2 measurement transport */
extern "C" int hipsdp_comm_count(void* comm, int* count, int* kind)
{
   hs_comm* c = (hs_comm*) comm;
   if ( c == NULL || count == NULL )
      return HIPSDP_ERR_ARG;
   *count = c->nranks;
   if ( kind != NULL )
      *kind = c->kind;
   if ( c->kind == 0 )
   {
      int n = 0;
      if ( ncclCommCount(c->nccl, &n) != ncclSuccess )
         return HIPSDP_ERR_HIP;
      *count = n;
   }
   return HIPSDP_OK;
}

extern "C" int hipsdp_comm_unique_id(void* unique_id_128bytes)
{
   ncclUniqueId id;
   if ( ncclGetUniqueId(&id) != ncclSuccess )
      return HIPSDP_ERR_HIP;
   static_assert(sizeof(ncclUniqueId) == 128, "unexpected ncclUniqueId size");
   memcpy(unique_id_128bytes, &id, sizeof(id));
   return HIPSDP_OK;
}

extern "C" int hipsdp_comm_create(const void* unique_id_128bytes, int rank, int nranks, void** comm)
{
   if ( comm == NULL || unique_id_128bytes == NULL || nranks < 1 || rank < 0 || rank >= nranks )
      return HIPSDP_ERR_ARG;
   ncclUniqueId id;
   memcpy(&id, unique_id_128bytes, sizeof(id));
   hs_comm* c = new (std::nothrow) hs_comm();
   if ( c == NULL )
      return HIPSDP_ERR_NOMEM;
   c->kind = 0; c->rank = rank; c->nranks = nranks;
   if ( ncclCommInitRank(&c->nccl, nranks, id, rank) != ncclSuccess )
   {
      delete c;
      return HIPSDP_ERR_HIP;
   }
   *comm = (void*) c;
   return HIPSDP_OK;
}

/* job nonce: what distinguishes this launch from an earlier (possibly crashed) one that used the same rendezvous name -
 * HIPSDP_JOB_ID, else the launcher's run id / rendezvous port (torchrun sets TORCHELASTIC_RUN_ID and MASTER_PORT); 0 = unknown */
static unsigned long long job_nonce(void)
{
   const char* keys[] = {"HIPSDP_JOB_ID", "TORCHELASTIC_RUN_ID", "MASTER_PORT"};
   unsigned long long h = 0;
   for (const char* k : keys)
   {
      const char* v = getenv(k);
      if ( v == NULL || v[0] == 0 )
         continue;
      h = 1469598103934665603ULL;
      for (const char* p = v; *p; ++p) h = (h ^ (unsigned char) *p) * 1099511628211ULL;
      for (const char* p = k; *p; ++p) h = (h ^ (unsigned char) *p) * 1099511628211ULL;
      if ( h == 0 ) h = 1;
      break;
   }
   if ( h == 0 )
   {
      /* plain HIPSDP_WORLD / HIPSDP_RANK launches (mpirun, a shell loop): the ranks of one job are children of one launcher, so
       * (parent pid, parent start time) is a value they share and a crashed earlier job does not.  Ranks without a common
       * parent (each started by hand, parent = init) get 0 and must set HIPSDP_JOB_ID themselves. */
      const long ppid = (long) getppid();
      if ( ppid > 1 )
      {
         unsigned long long start = 0;
         char path[64], buf[1024];
         snprintf(path, sizeof(path), "/proc/%ld/stat", ppid);
         FILE* f = fopen(path, "r");
         if ( f != NULL )
         {
            const size_t got = fread(buf, 1, sizeof(buf) - 1, f);
            fclose(f);
            buf[got] = 0;
            const char* q = strrchr(buf, ')');             /* the command name may contain blanks: fields are counted behind it */
            int field = 2;
            while ( q != NULL && *q != 0 && field < 22 )
            {
               q = strchr(q + 1, ' ');
               ++field;
            }
            if ( q != NULL )
               start = strtoull(q + 1, NULL, 10);
         }
         h = 1469598103934665603ULL;
         h = (h ^ (unsigned long long) ppid) * 1099511628211ULL;
         h = (h ^ start) * 1099511628211ULL;
         h = (h ^ 0x70706964ULL) * 1099511628211ULL;
         if ( h == 0 ) h = 1;
      }
   }
   return h;
}

#define SHM_READY_MAGIC 0x68697073647001ULL

static long long realtime_ns(void)
{
   struct timespec ts;
   clock_gettime(CLOCK_REALTIME, &ts);
   return (long long) ts.tv_sec * 1000000000LL + ts.tv_nsec;
}

extern "C" int hipsdp_comm_create_host(const char* name, int rank, int nranks, long long staging_bytes, double timeout_seconds,
   void** comm)
{
   if ( comm == NULL || name == NULL || name[0] != '/' || strlen(name) + 20 >= sizeof(((hs_comm*) 0)->name) || nranks < 1 || rank < 0
      || rank >= nranks || staging_bytes < 4096 )
      return HIPSDP_ERR_ARG;
   hs_comm* c = new (std::nothrow) hs_comm();
   if ( c == NULL )
      return HIPSDP_ERR_NOMEM;
   c->kind = 1; c->rank = rank; c->nranks = nranks; c->nccl = NULL;
   c->timeout_s = timeout_seconds > 0.0 ? timeout_seconds : 120.0;
   /* Safe against the leftovers of a crashed job: the job nonce is part of the segment name; rank 0 removes whatever carries the
    * name, creates the segment exclusively, initialises the header itself and only then publishes (ready word = magic ^ nonce);
    * the others never create - they wait for a segment with the right ready word that is still the one the name points to. */
   const unsigned long long nonce = job_nonce();
   if ( nonce != 0 )
      snprintf(c->name, sizeof(c->name), "%s.%016llx", name, nonce);
   else
      strcpy(c->name, name);
   c->data_bytes = (size_t) staging_bytes & ~(size_t) 63;
   c->map_bytes = sizeof(shm_header) + c->data_bytes;
   void* p = MAP_FAILED;
   const long long entered_ns = realtime_ns();
   const auto t0 = std::chrono::steady_clock::now();
   auto waited = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
   if ( rank == 0 )
   {
      int fd = -1;
      for (int attempt = 0; attempt < 4 && fd < 0; ++attempt)
      {
         (void) shm_unlink(c->name);
         fd = shm_open(c->name, O_CREAT | O_EXCL | O_RDWR, 0600);
      }
      if ( fd < 0 || ftruncate(fd, (off_t) c->map_bytes) != 0 )
      {
         if ( fd >= 0 ) { close(fd); (void) shm_unlink(c->name); }
         fprintf(stderr, "hipsdp: host-staged communicator: cannot create %s\n", c->name);
         delete c;
         return HIPSDP_ERR_HIP;
      }
      p = mmap(NULL, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
      close(fd);
      if ( p == MAP_FAILED )
      {
         (void) shm_unlink(c->name);
         delete c;
         return HIPSDP_ERR_NOMEM;
      }
      shm_header* h = (shm_header*) p;
      h->arrived.store(0, std::memory_order_relaxed);
      h->generation.store(0, std::memory_order_relaxed);
      h->failed.store(0, std::memory_order_relaxed);
      h->created_ns = realtime_ns();
      h->ready.store(SHM_READY_MAGIC ^ nonce, std::memory_order_release);
   }
   else
   {
      while ( p == MAP_FAILED )
      {
         const int fd = shm_open(c->name, O_RDWR, 0600);
         struct stat st_fd, st_name;
         char path[160];
         snprintf(path, sizeof(path), "/dev/shm%s", c->name);
         if ( fd >= 0 && fstat(fd, &st_fd) == 0 && (size_t) st_fd.st_size >= c->map_bytes )
         {
            void* q = mmap(NULL, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            if ( q != MAP_FAILED )
            {
               /* published by rank 0 of THIS job, and still the segment the name points to (rank 0 unlinks a stale one first) */
               const bool ready = ((shm_header*) q)->ready.load(std::memory_order_acquire) == (SHM_READY_MAGIC ^ nonce)
                  && ((shm_header*) q)->created_ns >= entered_ns - (long long) (c->timeout_s * 1e9);
               const bool current = stat(path, &st_name) != 0 || (st_name.st_ino == st_fd.st_ino && st_name.st_dev == st_fd.st_dev);
               if ( ready && current && stat(path, &st_name) == 0 )
                  p = q;
               else
                  munmap(q, c->map_bytes);
            }
         }
         if ( fd >= 0 ) close(fd);
         if ( p == MAP_FAILED )
         {
            if ( waited() > c->timeout_s )
            {
               fprintf(stderr, "hipsdp: host-staged communicator: rank %d found no segment %s of this job within %.0f s\n", rank, c->name, c->timeout_s);
               delete c;
               return HIPSDP_ERR_HIP;
            }
            usleep(2000);
         }
      }
   }
   c->hdr = (shm_header*) p;
   c->data = (char*) p + sizeof(shm_header);
   /* once everyone has mapped it the name can go: the memory lives until the last unmap */
   const int rc = shm_barrier(c);
   if ( rank == 0 )
      shm_unlink(c->name);
   if ( rc != HS_OK )
   {
      munmap(p, c->map_bytes);
      delete c;
      return HIPSDP_ERR_HIP;
   }
   *comm = (void*) c;
   return HIPSDP_OK;
}

extern "C" void hipsdp_comm_destroy(void* comm)
{
   hs_comm* c = (hs_comm*) comm;
   if ( c == NULL )
      return;
   for (auto& r : c->recs) { (void) hipEventDestroy(r.a); (void) hipEventDestroy(r.b); }
   for (hipEvent_t e : c->evpool) (void) hipEventDestroy(e);
   if ( c->kind == 0 )
      (void) ncclCommDestroy(c->nccl);
   else if ( c->kind == 1 )
      munmap((void*) c->hdr, c->map_bytes);
   delete c;
}

int hs_allgather_inplace(void* comm, double* buf, long long count_per_rank, int rank, hipStream_t stream)
{
   hs_comm* c = (hs_comm*) comm;
   CollTimer tm(c, stream, 8.0 * (double) count_per_rank * c->nranks);
   if ( c->kind == 2 )
      return HS_OK;
   if ( c->kind == 1 )
      return shm_allgather(c, (const char*) (buf + (long long) rank * count_per_rank), (char*) buf, (size_t) count_per_rank * sizeof(double), stream);
   if ( ncclAllGather(buf + (long long) rank * count_per_rank, buf, (size_t) count_per_rank, ncclDouble, c->nccl, stream) != ncclSuccess )
      return HS_ERR_HIP;
   return HS_OK;
}

int hs_allgather(void* comm, const double* send, double* recv, long long count_per_rank, hipStream_t stream)
{
   hs_comm* c = (hs_comm*) comm;
   CollTimer tm(c, stream, 8.0 * (double) count_per_rank * c->nranks);
   if ( c->kind == 2 )
   {
      HS_HIP( hipMemcpyAsync(recv + (long long) c->rank * count_per_rank, send, (size_t) count_per_rank * sizeof(double), hipMemcpyDeviceToDevice, stream) );
      return HS_OK;
   }
   if ( c->kind == 1 )
      return shm_allgather(c, (const char*) send, (char*) recv, (size_t) count_per_rank * sizeof(double), stream);
   if ( ncclAllGather(send, recv, (size_t) count_per_rank, ncclDouble, c->nccl, stream) != ncclSuccess )
      return HS_ERR_HIP;
   return HS_OK;
}

int hs_bcast_doubles(void* comm, double* buf, long long count, hipStream_t stream)
{
   hs_comm* c = (hs_comm*) comm;
   CollTimer tm(c, stream, 8.0 * (double) count);
   if ( c->kind == 2 )
      return HS_OK;
   if ( c->kind == 1 )
      return shm_bcast(c, (char*) buf, (size_t) count * sizeof(double), stream);
   if ( ncclBroadcast(buf, buf, (size_t) count, ncclDouble, 0, c->nccl, stream) != ncclSuccess )
      return HS_ERR_HIP;
   return HS_OK;
}

int hs_bcast_ints(void* comm, int* buf, long long count, hipStream_t stream)
{
   hs_comm* c = (hs_comm*) comm;
   CollTimer tm(c, stream, 4.0 * (double) count);
   if ( c->kind == 2 )
      return HS_OK;
   if ( c->kind == 1 )
      return shm_bcast(c, (char*) buf, (size_t) count * sizeof(int), stream);
   if ( ncclBroadcast(buf, buf, (size_t) count, ncclInt, 0, c->nccl, stream) != ncclSuccess )
      return HS_ERR_HIP;
   return HS_OK;
}

int hs_allreduce_sum(void* comm, double* buf, long long count, hipStream_t stream)
{
   hs_comm* c = (hs_comm*) comm;
   if ( count <= 0 )
      return HS_OK;
   CollTimer tm(c, stream, 8.0 * (double) count);
   if ( c->kind == 2 )
      return HS_OK;
   if ( c->kind == 1 )
   {
      /* gather the ranks' parts into a device buffer, add them up on the device */
      double* part = NULL;
      HS_HIP( hipMalloc((void**) &part, (size_t) count * (size_t) c->nranks * sizeof(double)) );
      int rc = shm_allgather(c, (const char*) buf, (char*) part, (size_t) count * sizeof(double), stream);
      if ( rc == HS_OK )
      {
         long long blocks = (count + 255) / 256;
         if ( blocks > 4096 ) blocks = 4096;
         hipLaunchKernelGGL(k_sum_parts, dim3((unsigned) blocks), dim3(256), 0, stream, count, c->nranks, part, buf);
         if ( hipGetLastError() != hipSuccess || hipStreamSynchronize(stream) != hipSuccess )
            rc = HS_ERR_HIP;
      }
      (void) hipFree(part);
      return rc;
   }
   if ( ncclAllReduce(buf, buf, (size_t) count, ncclDouble, ncclSum, c->nccl, stream) != ncclSuccess )
      return HS_ERR_HIP;
   return HS_OK;
}

/* all-to-all of the column-slice pieces of the variable-sharded Schur assembly (schur.hip: hs_schur_Wvar).
 * cnt[src * nranks + dst] doubles go from src to dst; every rank passes the same matrix. */
int hs_alltoall(void* comm, const double* send, double* recv, const long long* cnt, hipStream_t stream)
{
   hs_comm* c = (hs_comm*) comm;
   const int G = c->nranks;
   double sent = 0.0;
   for (int d = 0; d < G; ++d)
      if ( d != c->rank ) sent += 8.0 * (double) cnt[(long long) c->rank * G + d];
   CollTimer tm(c, stream, sent);
   if ( c->kind == 1 )
      return shm_alltoall(c, send, recv, cnt, stream);
   long long soff = 0, roff = 0;
   if ( c->kind == 2 )
   {
      /* measurement transport: only the piece a rank keeps for itself moves */
      for (int d = 0; d < c->rank; ++d) soff += cnt[(long long) c->rank * G + d];
      for (int r = 0; r < c->rank; ++r) roff += cnt[(long long) r * G + c->rank];
      const long long k = cnt[(long long) c->rank * G + c->rank];
      if ( k > 0 )
         HS_HIP( hipMemcpyAsync(recv + roff, send + soff, (size_t) k * sizeof(double), hipMemcpyDeviceToDevice, stream) );
      return HS_OK;
   }
   if ( ncclGroupStart() != ncclSuccess )
      return HS_ERR_HIP;
   bool ok = true;
   for (int p = 0; p < G; ++p)
   {
      const long long ks = cnt[(long long) c->rank * G + p], kr = cnt[(long long) p * G + c->rank];
      if ( ks > 0 && ncclSend(send + soff, (size_t) ks, ncclDouble, p, c->nccl, stream) != ncclSuccess ) ok = false;
      if ( kr > 0 && ncclRecv(recv + roff, (size_t) kr, ncclDouble, p, c->nccl, stream) != ncclSuccess ) ok = false;
      soff += ks; roff += kr;
   }
   if ( ncclGroupEnd() != ncclSuccess || !ok )
      return HS_ERR_HIP;
   return HS_OK;
}

/* Measurement transport: a communicator of nranks ranks of which only this one exists.  Collectives move nothing (the
 * all-to-all keeps the rank's own piece), so a solve through it computes garbage - it exists to time ONE rank's share of a
 * sharded assembly at sizes that need several GPUs (tests/devtools/shard_time.py). */
extern "C" int hipsdp_comm_create_null(int rank, int nranks, void** comm)
{
   if ( comm == NULL || nranks < 1 || rank < 0 || rank >= nranks )
      return HIPSDP_ERR_ARG;
   hs_comm* c = new (std::nothrow) hs_comm();
   if ( c == NULL )
      return HIPSDP_ERR_NOMEM;
   c->kind = 2; c->rank = rank; c->nranks = nranks; c->nccl = NULL; c->hdr = NULL;
   *comm = (void*) c;
   return HIPSDP_OK;
}

/* SPMD hosts (N identical processes, one per GPU, all making the same calls - e.g. N copies of SCIP-SDP started by a launcher):
 * the process-wide communicator described by the environment, created at the first call and kept until the process ends.
 *   HIPSDP_WORLD / WORLD_SIZE, HIPSDP_RANK / RANK     size and rank (world <= 1 or unset: *comm = NULL, single GPU)
 *   HIPSDP_COMM_SHM=/name                             host-staged transport (ranks sharing one device; validation)
 *   HIPSDP_COMM_FILE=path                             RCCL: rank 0 writes the 128-byte unique id to this file (write + rename),
 *                                                     the others wait for it (HIPSDP_COMM_TIMEOUT seconds, default 120)
 * Used by sdpisolver_hip.c at engine creation: the drop-in backend then shards every node SDP over the ranks. */
extern "C" int hipsdp_comm_from_env(int device, void** comm, int* rank, int* nranks)
{
   static void* g_comm = NULL;
   static int g_rank = 0, g_world = 1, g_state = 0;        /* 0 not tried, 1 ready, 2 failed */
   if ( comm == NULL || rank == NULL || nranks == NULL )
      return HIPSDP_ERR_ARG;
   *comm = NULL; *rank = 0; *nranks = 1;
   if ( g_state == 2 )
      return HIPSDP_ERR_HIP;
   if ( g_state == 0 )
   {
      const char* ew = getenv("HIPSDP_WORLD") != NULL ? getenv("HIPSDP_WORLD") : getenv("WORLD_SIZE");
      const char* er = getenv("HIPSDP_RANK") != NULL ? getenv("HIPSDP_RANK") : getenv("RANK");
      const int world = ew != NULL ? atoi(ew) : 1;
      const int r = er != NULL ? atoi(er) : 0;
      g_state = 1;
      if ( world > 1 )
      {
         const char* shm = getenv("HIPSDP_COMM_SHM");
         const char* file = getenv("HIPSDP_COMM_FILE");
         const double tmo = getenv("HIPSDP_COMM_TIMEOUT") != NULL && atof(getenv("HIPSDP_COMM_TIMEOUT")) > 0.0 ? atof(getenv("HIPSDP_COMM_TIMEOUT")) : 120.0;
         int rc = HIPSDP_ERR_ARG;
         if ( r < 0 || r >= world || world > 16 )
            fprintf(stderr, "hipsdp: rank %d of %d ranks is not usable\n", r, world);
         else if ( hipSetDevice(device) != hipSuccess )
            rc = HIPSDP_ERR_HIP;
         else if ( shm != NULL )
            rc = hipsdp_comm_create_host(shm, r, world, 64LL << 20, tmo, &g_comm);
         else if ( file != NULL )
         {
            /* the file holds the 128-byte RCCL id followed by the job nonce.  Rank 0 removes leftovers of an earlier job before it
             * writes (write + rename); the others accept only a file with this job's nonce that is not older than the join timeout */
            unsigned char id[128 + 8];
            const unsigned long long nonce = job_nonce();
            const long long entered_ns = realtime_ns();
            bool have = false;
            if ( r == 0 )
            {
               char tmp[4096];
               snprintf(tmp, sizeof(tmp), "%s.tmp", file);
               (void) unlink(file);
               (void) unlink(tmp);
               memcpy(id + 128, &nonce, 8);
               FILE* f = hipsdp_comm_unique_id(id) == HIPSDP_OK ? fopen(tmp, "wb") : NULL;
               have = f != NULL && fwrite(id, 1, sizeof(id), f) == sizeof(id);
               if ( f != NULL ) have = (fclose(f) == 0) && have;
               have = have && rename(tmp, file) == 0;
            }
            else
            {
               const auto t0 = std::chrono::steady_clock::now();
               while ( !have && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < tmo )
               {
                  FILE* f = fopen(file, "rb");
                  if ( f != NULL )
                  {
                     struct stat st;
                     unsigned long long got = 0;
                     have = fread(id, 1, sizeof(id), f) == sizeof(id);
                     if ( have )
                        memcpy(&got, id + 128, 8);
                     have = have && got == nonce && fstat(fileno(f), &st) == 0
                        && (long long) st.st_mtime * 1000000000LL >= entered_ns - (long long) ((tmo + 2.0) * 1e9);
                     fclose(f);
                  }
                  if ( !have )
                     usleep(20000);
               }
            }
            if ( !have )
               fprintf(stderr, "hipsdp: rank %d could not %s the communicator id file %s of this job\n", r, r == 0 ? "write" : "read", file);
            else
               rc = hipsdp_comm_create(id, r, world, &g_comm);
            if ( r == 0 && rc == HIPSDP_OK )
               (void) unlink(file);           /* everyone has joined: the id is of no further use */
         }
         else
            fprintf(stderr, "hipsdp: %d ranks but neither HIPSDP_COMM_FILE nor HIPSDP_COMM_SHM is set\n", world);
         if ( rc != HIPSDP_OK )
         {
            g_state = 2;
            g_comm = NULL;
            return rc;
         }
         g_rank = r; g_world = world;
      }
   }
   *comm = g_comm; *rank = g_rank; *nranks = g_world;
   return HIPSDP_OK;
}
