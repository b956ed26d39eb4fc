// Data-parallel mapper collectives: RCCL over xGMI behind the C ABI.
//
// The reference is single-GPU (pin_slam.py:8); the exchange exists because the map features and the decoder are
// the parameters of Mapper.mapping (utils/mapper.py:604) and every rank trains on a shard of the batch.  RCCL is
// bound at run time (pin_comm_load: dlopen of the librccl the host process already uses, so there is one RCCL and
// one HIP runtime in the process); libpinhip itself has no link-time dependency on it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "pin_common.h"

namespace pin {
namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
} g;

#define PIN_CHECK_NCCL(expr)                                                                                   \
    do {                                                                                                       \
        ncclResult_t r_ = (expr);                                                                              \
        if (r_ != ncclSuccess)                                                                                 \
            return ::pin::fail(-3, "%s: RCCL: %s", __func__, g.GetErrorString ? g.GetErrorString(r_) : "error"); \
    } while (0)

// delta[i] = cert[i] - cert0[i]  (what this rank's shard added to the certainties since cert0 was taken)
__global__ void cert_delta_kernel(const float* __restrict__ cert, const float* __restrict__ cert0,
                                  float* __restrict__ delta, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) delta[i] = cert[i] - cert0[i];
}

// cert[i] = cert0[i] + (sum over ranks of delta)[i]
__global__ void cert_apply_kernel(float* __restrict__ cert, const float* __restrict__ cert0,
                                  const float* __restrict__ delta, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) cert[i] = cert0[i] + delta[i];
}

}  // namespace
}  // namespace pin

using namespace pin;

extern "C" int pin_comm_load(const char* rccl_path) {
    if (g.handle) return 0;
    const char* names[] = {rccl_path, "librccl.so.1", "librccl.so"};
    void* h = nullptr;
    for (const char* nm : names) {
        if (!nm || !*nm) continue;
        h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) return fail(-3, "pin_comm_load: cannot dlopen RCCL (%s)", dlerror());
#define PIN_SYM(field, name)                                                          \
    do {                                                                              \
        g.field = reinterpret_cast<decltype(g.field)>(dlsym(h, name));                \
        if (!g.field) return fail(-3, "pin_comm_load: %s not found in RCCL", name);   \
    } while (0)
    PIN_SYM(GetUniqueId, "ncclGetUniqueId");
    PIN_SYM(CommInitRank, "ncclCommInitRank");
    PIN_SYM(CommDestroy, "ncclCommDestroy");
    PIN_SYM(AllReduce, "ncclAllReduce");
    PIN_SYM(AllGather, "ncclAllGather");
    PIN_SYM(GroupStart, "ncclGroupStart");
    PIN_SYM(GroupEnd, "ncclGroupEnd");
    PIN_SYM(GetErrorString, "ncclGetErrorString");
#undef PIN_SYM
    g.handle = h;
    return 0;
}

extern "C" int pin_comm_unique_id(void* id_out) {
    PIN_CHECK_ARG(g.handle, "RCCL not loaded (pin_comm_load)");
    PIN_CHECK_ARG(id_out, "NULL pointer");
    static_assert(sizeof(ncclUniqueId) == PIN_COMM_ID_BYTES, "ncclUniqueId size");
    PIN_CHECK_NCCL(g.GetUniqueId(reinterpret_cast<ncclUniqueId*>(id_out)));
    return 0;
}

extern "C" int pin_comm_init_rank(const void* id, int32_t rank, int32_t world, void** comm_out) {
    PIN_CHECK_ARG(g.handle, "RCCL not loaded (pin_comm_load)");
    PIN_CHECK_ARG(id && comm_out && world >= 1 && rank >= 0 && rank < world, "bad arguments");
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclComm_t c = nullptr;
    PIN_CHECK_NCCL(g.CommInitRank(&c, world, uid, rank));
    *comm_out = c;
    return 0;
}

extern "C" int pin_comm_destroy(void* comm) {
    if (!comm) return 0;
    PIN_CHECK_ARG(g.handle, "RCCL not loaded (pin_comm_load)");
    PIN_CHECK_NCCL(g.CommDestroy(reinterpret_cast<ncclComm_t>(comm)));
    return 0;
}

extern "C" int pin_allreduce_grads(void* comm, float* grads, int64_t count, void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(g.handle, "RCCL not loaded (pin_comm_load)");
    PIN_CHECK_ARG(comm && count >= 0, "bad arguments");
    if (count == 0) return 0;
    PIN_CHECK_ARG(grads, "NULL pointer");
    PIN_CHECK_NCCL(g.AllReduce(grads, grads, (size_t)count, ncclFloat32, ncclSum, reinterpret_cast<ncclComm_t>(comm),
                               as_stream(stream)));
    return 0;
}

extern "C" int pin_allreduce_f32(void* comm, const float* send, float* recv, int64_t count, void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(g.handle, "RCCL not loaded (pin_comm_load)");
    PIN_CHECK_ARG(comm && count >= 0, "bad arguments");
    if (count == 0) return 0;
    PIN_CHECK_ARG(send && recv, "NULL pointer");
    PIN_CHECK_NCCL(g.AllReduce(send, recv, (size_t)count, ncclFloat32, ncclSum, reinterpret_cast<ncclComm_t>(comm),
                               as_stream(stream)));
    return 0;
}

extern "C" int pin_allgather_f32(void* comm, const float* send, float* recv, int64_t count_per_rank, void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(g.handle, "RCCL not loaded (pin_comm_load)");
    PIN_CHECK_ARG(comm && count_per_rank >= 0, "bad arguments");
    if (count_per_rank == 0) return 0;
    PIN_CHECK_ARG(send && recv, "NULL pointer");
    PIN_CHECK_NCCL(g.AllGather(send, recv, (size_t)count_per_rank, ncclFloat32, reinterpret_cast<ncclComm_t>(comm), as_stream(stream)));
    return 0;
}

extern "C" int pin_dp_cert_snapshot(const float* certainty, float* certainty0_out, int32_t n, void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(n >= 0, "bad size");
    if (n == 0) return 0;
    PIN_CHECK_ARG(certainty && certainty0_out, "NULL pointer");
    PIN_CHECK_HIP(hipMemcpyAsync(certainty0_out, certainty, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice,
                                 as_stream(stream)));
    return 0;
}

extern "C" int pin_dp_cert_delta(const float* certainty, const float* certainty0, float* delta_out, int32_t n,
                                 void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(n >= 0, "bad size");
    if (n == 0) return 0;
    PIN_CHECK_ARG(certainty && certainty0 && delta_out, "NULL pointer");
    hipLaunchKernelGGL(cert_delta_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), certainty, certainty0,
                       delta_out, n);
    PIN_CHECK_LAUNCH();
    return 0;
}

extern "C" int pin_dp_cert_apply(float* certainty, const float* certainty0, const float* delta_sum, int32_t n,
                                 void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(n >= 0, "bad size");
    if (n == 0) return 0;
    PIN_CHECK_ARG(certainty && certainty0 && delta_sum, "NULL pointer");
    hipLaunchKernelGGL(cert_apply_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), certainty, certainty0,
                       delta_sum, n);
    PIN_CHECK_LAUNCH();
    return 0;
}

extern "C" int pin_dp_sync_side_effects(void* comm, float* certainty, const float* certainty0, float* scratch,
                                        int32_t* ts_update, int32_t n, void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(g.handle, "RCCL not loaded (pin_comm_load)");
    PIN_CHECK_ARG(comm && n >= 0, "bad arguments");
    if (n == 0) return 0;
    PIN_CHECK_ARG(certainty && certainty0 && scratch && ts_update, "NULL pointer");
    hipStream_t s = as_stream(stream);
    ncclComm_t c = reinterpret_cast<ncclComm_t>(comm);
    int rc = pin_dp_cert_delta(certainty, certainty0, scratch, n, stream);
    if (rc) return rc;
    PIN_CHECK_NCCL(g.GroupStart());
    ncclResult_t r1 = g.AllReduce(scratch, scratch, (size_t)n, ncclFloat32, ncclSum, c, s);
    ncclResult_t r2 = g.AllReduce(ts_update, ts_update, (size_t)n, ncclInt32, ncclMax, c, s);
    PIN_CHECK_NCCL(g.GroupEnd());
    PIN_CHECK_NCCL(r1);
    PIN_CHECK_NCCL(r2);
    return pin_dp_cert_apply(certainty, certainty0, scratch, n, stream);
}

// pin_warmup (common.hip): asking for a kernel's attributes makes the runtime load this translation unit's code object now
// instead of inside the first frame that launches one of its kernels
namespace pin {
int pin_warm_comm() {
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&cert_delta_kernel)) == hipSuccess ? 0 : -2;
}
}  // namespace pin
