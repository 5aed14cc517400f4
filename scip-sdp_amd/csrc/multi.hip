/* multi.hip - RCCL plumbing (one process per GPU).  The communicator is created from a unique id that the launcher
 * (bench.py via torch.distributed, or any MPI-like bootstrap) broadcasts; the engine only sees an opaque pointer. */
#include "hs_kernels.h"
#include "../../include/hipsdp.h"
#include <rccl/rccl.h>
#include <cstring>

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
   ncclUniqueId id;
   memcpy(&id, unique_id_128bytes, sizeof(id));
   ncclComm_t c;
   if ( ncclCommInitRank(&c, nranks, id, rank) != ncclSuccess )
      return HIPSDP_ERR_HIP;
   *comm = (void*) c;
   return HIPSDP_OK;
}

extern "C" void hipsdp_comm_destroy(void* comm)
{
   if ( comm != NULL )
      (void) ncclCommDestroy((ncclComm_t) comm);
}

int hs_allgather_inplace(void* comm, double* buf, long long count_per_rank, int rank, hipStream_t stream)
{
   if ( ncclAllGather(buf + (long long) rank * count_per_rank, buf, (size_t) count_per_rank, ncclDouble, (ncclComm_t) comm, stream) != ncclSuccess )
      return HS_ERR_HIP;
   return HS_OK;
}

int hs_allgather(void* comm, const double* send, double* recv, long long count_per_rank, hipStream_t stream)
{
   if ( ncclAllGather(send, recv, (size_t) count_per_rank, ncclDouble, (ncclComm_t) comm, stream) != ncclSuccess )
      return HS_ERR_HIP;
   return HS_OK;
}

int hs_bcast_doubles(void* comm, double* buf, long long count, hipStream_t stream)
{
   if ( ncclBroadcast(buf, buf, (size_t) count, ncclDouble, 0, (ncclComm_t) comm, stream) != ncclSuccess )
      return HS_ERR_HIP;
   return HS_OK;
}

int hs_bcast_ints(void* comm, int* buf, long long count, hipStream_t stream)
{
   if ( ncclBroadcast(buf, buf, (size_t) count, ncclInt, 0, (ncclComm_t) comm, stream) != ncclSuccess )
      return HS_ERR_HIP;
   return HS_OK;
}
