// C1: the one exchange step of the view-sharded path (SURVEY.md section 8 e): sum all-reduce of the
// data-term texture gradient over RCCL / xGMI. The reference has no collective (single GPU only); the
// R-GPU step is defined as the mean of R independent B = 1 gradients followed by one Adam update, and the
// 1/R scaling is folded into sm_adam_fused's grad_scale.
#include <rccl/rccl.h>

#include "common.h"

extern "C" int sm_allreduce_grad(void* comm, float* g, size_t n, void* stream) {
    return (int)ncclAllReduce(g, g, n, ncclFloat, ncclSum, (ncclComm_t)comm, (hipStream_t)stream);
}
