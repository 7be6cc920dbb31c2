// C1: the exchange step of the view-sharded path (SURVEY.md section 8 e) over RCCL / xGMI: a communicator per
// process (one process per GPU), a sum all-reduce of the (compacted) data-term texture gradient every step and a max
// all-reduce of the per-view touch flags at every view change. The reference has no collective (single GPU only); the
// R-GPU step is defined as the mean of R independent B = 1 gradients followed by one Adam update, and the 1/R scaling
// is folded into sm_adam_fused's grad_scale. Collectives are enqueued on the caller's HIP stream, i.e. ordered with
// the kernels that produce and consume the buffers - no host synchronisation, no extra stream hop.
#include <rccl/rccl.h>

#include <cstring>

#include "common.h"

extern "C" {

size_t sm_comm_unique_id_bytes(void) { return sizeof(ncclUniqueId); }

int sm_comm_get_unique_id(void* id_out) {
    ncclUniqueId id;
    const ncclResult_t r = ncclGetUniqueId(&id);
    if (r == ncclSuccess) memcpy(id_out, &id, sizeof(id));
    return (int)r;
}

int sm_comm_init(void** comm_out, int n_ranks, const void* unique_id, int rank) {
    if (comm_out == nullptr || unique_id == nullptr || n_ranks < 1 || rank < 0 || rank >= n_ranks)
        return (int)ncclInvalidArgument;
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclComm_t c = nullptr;
    const ncclResult_t r = ncclCommInitRank(&c, n_ranks, id, rank);
    *comm_out = (r == ncclSuccess) ? (void*)c : nullptr;
    return (int)r;
}

int sm_comm_destroy(void* comm) { return comm ? (int)ncclCommDestroy((ncclComm_t)comm) : 0; }

int sm_allreduce_grad(void* comm, float* g, size_t n, void* stream) {
    if (n == 0) return 0;
    return (int)ncclAllReduce(g, g, n, ncclFloat, ncclSum, (ncclComm_t)comm, (hipStream_t)stream);
}

int sm_allreduce_flags_max(void* comm, int32_t* flags, size_t n, void* stream) {
    if (n == 0) return 0;
    return (int)ncclAllReduce(flags, flags, n, ncclInt32, ncclMax, (ncclComm_t)comm, (hipStream_t)stream);
}

}  // extern "C"
