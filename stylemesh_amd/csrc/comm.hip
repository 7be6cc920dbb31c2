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

// A HIP stream whose kernels run on a SUBSET of the compute units: the low `n_cus` bits of the queue's CU mask. The
// driver deals mask bits round-robin to the 8 XCDs (bit b -> XCD b % 8) and, inside an XCD, to its shader engines, so
// the low n bits are n / 8 CUs of every XCD, spread over its engines. For the engine's side streams: HBM-bound work
// with slack (style branches, the early half of the update) then shares n CUs with the conv trunk instead of all 256.
int sm_stream_create_cu_subset(int n_cus, void** stream_out) {
    if (n_cus < 8 || n_cus > 256 || stream_out == nullptr) return (int)hipErrorInvalidValue;
    uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = 0; b < n_cus; ++b) mask[b >> 5] |= 1u << (b & 31);
    hipStream_t s = nullptr;
    const hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask);
    *stream_out = (void*)s;
    return (int)e;
}

int sm_stream_destroy(void* stream) { return stream ? (int)hipStreamDestroy((hipStream_t)stream) : 0; }

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

int sm_comm_info(void* comm, int* info_out) {
    if (comm == nullptr || info_out == nullptr) return (int)ncclInvalidArgument;
    ncclResult_t r = ncclCommCount((ncclComm_t)comm, &info_out[0]);
    if (r == ncclSuccess) r = ncclCommUserRank((ncclComm_t)comm, &info_out[1]);
    if (r == ncclSuccess) r = ncclCommCuDevice((ncclComm_t)comm, &info_out[2]);
    if (r == ncclSuccess) r = ncclGetVersion(&info_out[3]);
    return (int)r;
}

int sm_device_link(int device_a, int device_b, int* link_type, int* hops) {
    if (link_type == nullptr || hops == nullptr) return (int)hipErrorInvalidValue;
    uint32_t t = 0, h = 0;
    const hipError_t e = hipExtGetLinkTypeAndHopCount(device_a, device_b, &t, &h);
    *link_type = (int)t;
    *hops = (int)h;
    return (int)e;
}

}  // extern "C"
