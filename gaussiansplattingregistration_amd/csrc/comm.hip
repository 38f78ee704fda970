// comm.hip -- the multi-GPU communicator of libgsr_hip.so: RCCL called from the library, on the context's own stream.
//
// The reference has no distributed code; SURVEY.md 8(e) and DESIGN.md section 7 define what the ranks
// exchange.  One process per GPU.  A gsr_comm wraps ONE of two transports behind the same five operations:
//
//   RCCL      gsr_comm_create: ncclCommInitRank from a 128-byte id (gsr_comm_get_unique_id on rank 0, distributed by any
//             means -- the Python side broadcasts it with torch.distributed).  librccl.so.1 is opened lazily with dlopen: the
//             single-GPU product has no load-time dependency on it, and a host process that already carries an RCCL (PyTorch
//             does) shares that copy.  Every operation is ENQUEUED on the caller's stream: no host synchronisation, no
//             trampoline into the host language per ICP iteration or per HEM level.
//   callbacks gsr_comm_create_callbacks: the host language supplies the operations on DEVICE buffers (tests: torch.distributed
//             over gloo with two ranks sharing the box's one GPU -- RCCL refuses two ranks on one device).  A callback runs
//             after the library has synchronised the stream and must have completed when it returns.
//
// Operations (all on device memory): all-reduce (sum / max; float64, float32, int32, uint32, uint64), all-gather of equal
// chunks, and exchange() -- a personalised all-to-all of byte runs (ncclSend / ncclRecv in one group), which the spatial HEM
// partition uses for halo records.
#include "gsr_common.h"

#include <dlfcn.h>
#include <string.h>

#include <rccl/rccl.h>

namespace gsr {

struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};

// why == the reason when NULL is returned (dlopen's own message, or the missing symbol)
static RcclApi* rccl_api(const char** why = nullptr) {
    static RcclApi api;                               // function-local static: initialised once, thread-safely
    static const bool ok = [] {
        // an explicit GSR_RCCL_LIB is the ONLY candidate: a wrong path is an error, not a silent switch to another copy
        const char* env = getenv("GSR_RCCL_LIB");
        const bool explicit_lib = env && *env;
        const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            if (explicit_lib && n != env) break;
            api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (api.handle) break;
            const char* e = dlerror();                // ONE call: dlerror() clears the message it returns
            api.error = e ? e : "dlopen failed";
        }
        if (!api.handle) return false;
#define GSR_SYM(field, name)                                                              \
    api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.handle, name));           \
    if (!api.field) { api.error = std::string("librccl: missing symbol ") + name; return false; }
        GSR_SYM(GetUniqueId, "ncclGetUniqueId") GSR_SYM(CommInitRank, "ncclCommInitRank") GSR_SYM(CommDestroy, "ncclCommDestroy")
        GSR_SYM(AllReduce, "ncclAllReduce") GSR_SYM(AllGather, "ncclAllGather") GSR_SYM(Send, "ncclSend") GSR_SYM(Recv, "ncclRecv")
        GSR_SYM(GroupStart, "ncclGroupStart") GSR_SYM(GroupEnd, "ncclGroupEnd") GSR_SYM(GetErrorString, "ncclGetErrorString")
#undef GSR_SYM
        return true;
    }();
    if (!ok && why) *why = api.error.empty() ? "no candidate library" : api.error.c_str();
    return ok ? &api : nullptr;
}


}  // namespace gsr

using namespace gsr;

struct gsr_comm {
    int rank = 0, world = 1, device = 0;
    ncclComm_t nccl = nullptr;                  // RCCL transport
    gsr_comm_callbacks cb;                      // callback transport (cb.allreduce != NULL)
    bool use_cb = false;
};

#define GSR_NCCL(api, expr)                                                                                       \
    do {                                                                                                          \
        ncclResult_t _r = (expr);                                                                                 \
        if (_r != ncclSuccess) return fail(GSR_E_HIP, "%s failed: %s", #expr, (api)->GetErrorString(_r));         \
    } while (0)

static bool dtype_of(int32_t dtype, ncclDataType_t* t, size_t* size) {
    switch (dtype) {
        case GSR_DT_F64: *t = ncclFloat64; *size = 8; return true;
        case GSR_DT_F32: *t = ncclFloat32; *size = 4; return true;
        case GSR_DT_I32: *t = ncclInt32; *size = 4; return true;
        case GSR_DT_U32: *t = ncclUint32; *size = 4; return true;
        case GSR_DT_U64: *t = ncclUint64; *size = 8; return true;
        default: return false;
    }
}

extern "C" {

int32_t gsr_comm_get_unique_id(void* id128) {
    if (!id128) return fail(GSR_E_INVALID, "gsr_comm_get_unique_id: NULL argument");
    const char* why = "";
    RcclApi* api = rccl_api(&why);
    if (!api) return fail(GSR_E_HIP, "gsr_comm_get_unique_id: RCCL is not available (%s)", why);
    static_assert(sizeof(ncclUniqueId) == GSR_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    GSR_NCCL(api, api->GetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return GSR_OK;
}

int32_t gsr_comm_create(gsr_comm** out, const void* id128, int32_t rank, int32_t world, int32_t device) {
    if (!out || !id128) return fail(GSR_E_INVALID, "gsr_comm_create: NULL argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return fail(GSR_E_INVALID, "gsr_comm_create: rank %d of %d", rank, world);
    const char* why = "";
    RcclApi* api = rccl_api(&why);
    if (!api) return fail(GSR_E_HIP, "gsr_comm_create: RCCL is not available (%s)", why);
    GSR_HIP(hipSetDevice(device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    gsr_comm* c = new gsr_comm();
    c->rank = rank; c->world = world; c->device = device;
    ncclResult_t r = api->CommInitRank(&c->nccl, world, id, rank);
    if (r != ncclSuccess) { delete c; return fail(GSR_E_HIP, "ncclCommInitRank failed: %s", api->GetErrorString(r)); }
    *out = c;
    return GSR_OK;
}

int32_t gsr_comm_create_callbacks(gsr_comm** out, int32_t rank, int32_t world, int32_t device, const gsr_comm_callbacks* cb) {
    if (!out || !cb) return fail(GSR_E_INVALID, "gsr_comm_create_callbacks: NULL argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return fail(GSR_E_INVALID, "gsr_comm_create_callbacks: rank %d of %d", rank, world);
    if (!cb->allreduce || !cb->allgather || !cb->exchange) return fail(GSR_E_INVALID, "gsr_comm_create_callbacks: every callback is required");
    gsr_comm* c = new gsr_comm();
    c->rank = rank; c->world = world; c->device = device; c->cb = *cb; c->use_cb = true;
    *out = c;
    return GSR_OK;
}

int32_t gsr_comm_destroy(gsr_comm* c) {
    if (!c) return GSR_OK;
    if (c->nccl) { RcclApi* api = rccl_api(); if (api) (void)api->CommDestroy(c->nccl); }
    delete c;
    return GSR_OK;
}

int32_t gsr_comm_rank(const gsr_comm* c) { return c ? c->rank : 0; }
int32_t gsr_comm_world(const gsr_comm* c) { return c ? c->world : 1; }

int32_t gsr_comm_allreduce(gsr_comm* c, void* dev_buf, int64_t count, int32_t dtype, int32_t op, void* stream) {
    if (!c || (count > 0 && !dev_buf) || count < 0) return fail(GSR_E_INVALID, "gsr_comm_allreduce: bad argument");
    ncclDataType_t t; size_t size;
    if (!dtype_of(dtype, &t, &size) || (op != GSR_OP_SUM && op != GSR_OP_MAX)) return fail(GSR_E_INVALID, "gsr_comm_allreduce: dtype %d / op %d", dtype, op);
    if (count == 0) return GSR_OK;
    hipStream_t st = (hipStream_t)stream;
    if (c->use_cb) {
        if (c->world == 1) return GSR_OK;
        GSR_HIP(hipStreamSynchronize(st));
        if (c->cb.allreduce(dev_buf, count, dtype, op, c->cb.user) != 0) return fail(GSR_E_INVALID, "gsr_comm_allreduce: callback failed");
        return GSR_OK;
    }
    RcclApi* api = rccl_api();
    GSR_NCCL(api, api->AllReduce(dev_buf, dev_buf, (size_t)count, t, op == GSR_OP_SUM ? ncclSum : ncclMax, c->nccl, st));
    return GSR_OK;
}

int32_t gsr_comm_allgather(gsr_comm* c, const void* dev_send, void* dev_recv, int64_t bytes_per_rank, void* stream) {
    if (!c || bytes_per_rank < 0 || (bytes_per_rank > 0 && (!dev_send || !dev_recv))) return fail(GSR_E_INVALID, "gsr_comm_allgather: bad argument");
    if (bytes_per_rank == 0) return GSR_OK;
    hipStream_t st = (hipStream_t)stream;
    if (c->use_cb) {
        if (c->world == 1) {
            if (dev_send != dev_recv) GSR_HIP(hipMemcpyAsync(dev_recv, dev_send, (size_t)bytes_per_rank, hipMemcpyDeviceToDevice, st));
            return GSR_OK;
        }
        GSR_HIP(hipStreamSynchronize(st));
        if (c->cb.allgather(dev_send, dev_recv, bytes_per_rank, c->cb.user) != 0) return fail(GSR_E_INVALID, "gsr_comm_allgather: callback failed");
        return GSR_OK;
    }
    RcclApi* api = rccl_api();
    GSR_NCCL(api, api->AllGather(dev_send, dev_recv, (size_t)bytes_per_rank, ncclUint8, c->nccl, st));
    return GSR_OK;
}

int32_t gsr_comm_exchange(gsr_comm* c, const void* dev_send, const int64_t* send_off, const int64_t* send_bytes, void* dev_recv,
                          const int64_t* recv_off, const int64_t* recv_bytes, void* stream) {
    if (!c || !send_off || !send_bytes || !recv_off || !recv_bytes) return fail(GSR_E_INVALID, "gsr_comm_exchange: NULL argument");
    hipStream_t st = (hipStream_t)stream;
    // the run a rank sends to itself is a local copy in every transport
    if (send_bytes[c->rank] != recv_bytes[c->rank]) return fail(GSR_E_INVALID, "gsr_comm_exchange: self run sizes differ");
    if (send_bytes[c->rank] > 0)
        GSR_HIP(hipMemcpyAsync((char*)dev_recv + recv_off[c->rank], (const char*)dev_send + send_off[c->rank], (size_t)send_bytes[c->rank], hipMemcpyDeviceToDevice, st));
    if (c->world == 1) return GSR_OK;
    if (c->use_cb) {
        GSR_HIP(hipStreamSynchronize(st));
        if (c->cb.exchange(dev_send, send_off, send_bytes, dev_recv, recv_off, recv_bytes, c->cb.user) != 0) return fail(GSR_E_INVALID, "gsr_comm_exchange: callback failed");
        return GSR_OK;
    }
    RcclApi* api = rccl_api();
    GSR_NCCL(api, api->GroupStart());
    // the first error is remembered and the group is CLOSED whatever happened: a return between GroupStart and GroupEnd would
    // leave the communicator with an open group, unusable for every later call
    ncclResult_t first = ncclSuccess;
    const char* what = "";
    for (int r = 0; r < c->world && first == ncclSuccess; ++r) {
        if (r == c->rank) continue;
        if (send_bytes[r] > 0) {
            first = api->Send((const char*)dev_send + send_off[r], (size_t)send_bytes[r], ncclUint8, r, c->nccl, st);
            if (first != ncclSuccess) { what = "ncclSend"; break; }
        }
        if (recv_bytes[r] > 0) {
            first = api->Recv((char*)dev_recv + recv_off[r], (size_t)recv_bytes[r], ncclUint8, r, c->nccl, st);
            if (first != ncclSuccess) { what = "ncclRecv"; break; }
        }
    }
    ncclResult_t end = api->GroupEnd();
    if (first != ncclSuccess) return fail(GSR_E_HIP, "gsr_comm_exchange: %s failed: %s", what, api->GetErrorString(first));
    if (end != ncclSuccess) return fail(GSR_E_HIP, "gsr_comm_exchange: ncclGroupEnd failed: %s", api->GetErrorString(end));
    return GSR_OK;
}

}  // extern "C"
