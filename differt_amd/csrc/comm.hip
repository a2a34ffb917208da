// comm.hip -- the path's collectives on RCCL without torch in the process (SURVEY.md section 8b export list:
// drt_comm_init / drt_comm_destroy; section 8e: the ONLY exchange steps of the path).
//
//   triangle-block first hit : one MIN all-reduce of 8 B per ray on packed (ordered(t) << 32 | tie) keys
//                              (drt_first_hit_keys -> drt_allreduce_min_u64 -> drt_first_hit_finalize)
//   triangle-block tracer    : one MAX all-reduce of a byte per surviving candidate (blocked flags)
//   gradients                : one SUM all-reduce of f32 [N_tx * 3 + N_rx * 3 (+ N_v * 3)]
//   compact records          : all-gather of fixed-size byte blocks (counts first, then padded records)
// One process per GPU; the communicator binds to the CURRENT HIP device of the calling thread; every collective
// runs on the caller's stream (in place where send == recv).  librccl.so is opened with dlopen on first use:
// the library itself has no link-time dependency on RCCL (it loads on a CPU-only box), and a host that already
// carries an RCCL (torch does) shares that copy instead of loading a second one.
#include <dlfcn.h>
#include <string.h>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
// A ROCm install without the RCCL development headers still builds the whole library: the runtime
// binding is by dlopen / dlsym anyway, and this file needs only these declarations of the stable NCCL ABI
// (rccl.h: NCCL_UNIQUE_ID_BYTES 128; ncclSum 0, ncclMax 2, ncclMin 3; ncclUint8 1, ncclUint64 5, ncclFloat32 7).
#include <hip/hip_runtime.h>
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef struct ncclComm *ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6,
               ncclFloat32 = 7 } ncclDataType_t;
#endif

#include <mutex>

#include "common.hpp"

namespace {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

RcclApi &rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (api.handle) break;
        }
        if (!api.handle) {
            const char *e = dlerror();
            api.error = std::string("cannot open librccl.so: ") + (e ? e : "?");
            return;
        }
        auto sym = [&](const char *name) -> void * {
            void *p = dlsym(api.handle, name);
            if (!p && api.error.empty()) api.error = std::string("librccl.so lacks ") + name;
            return p;
        };
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
        api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return api;
}

int32_t rccl_ready() {
    RcclApi &a = rccl();
    if (!a.error.empty()) return drt::fail(DRT_E_UNSUPPORTED, "%s", a.error.c_str());
    return DRT_OK;
}

}  // namespace

struct drt_comm {
    ncclComm_t comm = nullptr;
    int32_t rank = 0, world = 1;
};

#define DRT_NCCL(expr)                                                                                   \
    do {                                                                                                 \
        ncclResult_t _r = (expr);                                                                        \
        if (_r != ncclSuccess)                                                                           \
            return drt::fail(DRT_E_HIP, "%s failed: %s", #expr, rccl().GetErrorString ? rccl().GetErrorString(_r) : "?"); \
    } while (0)

static_assert(DRT_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");

using namespace drt;

extern "C" {

int32_t drt_comm_unique_id(uint8_t *id_out_host) {
    DRT_REQUIRE(id_out_host, "null argument");
    int32_t rc = rccl_ready();
    if (rc != DRT_OK) return rc;
    ncclUniqueId id;
    DRT_NCCL(rccl().GetUniqueId(&id));
    memcpy(id_out_host, id.internal, NCCL_UNIQUE_ID_BYTES);
    return DRT_OK;
}

int32_t drt_comm_init(const uint8_t *unique_id_host, int32_t rank, int32_t world, drt_comm_t *comm_out) {
    DRT_REQUIRE(unique_id_host && comm_out, "null argument");
    *comm_out = nullptr;
    DRT_REQUIRE(world >= 1 && rank >= 0 && rank < world, "rank %d / world %d", (int)rank, (int)world);
    int32_t rc = drt_device_check();  // no GPU: a status code, never an abort inside RCCL
    if (rc != DRT_OK) return rc;
    rc = rccl_ready();
    if (rc != DRT_OK) return rc;
    ncclUniqueId id;
    memcpy(id.internal, unique_id_host, NCCL_UNIQUE_ID_BYTES);
    drt_comm *c = new drt_comm();
    c->rank = rank;
    c->world = world;
    ncclResult_t r = rccl().CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        delete c;
        return fail(DRT_E_HIP, "ncclCommInitRank failed: %s", rccl().GetErrorString(r));
    }
    *comm_out = c;
    return DRT_OK;
}

int32_t drt_comm_destroy(drt_comm_t comm) {
    if (!comm) return DRT_OK;
    ncclResult_t r = comm->comm ? rccl().CommDestroy(comm->comm) : ncclSuccess;
    delete comm;
    if (r != ncclSuccess) return fail(DRT_E_HIP, "ncclCommDestroy failed: %s", rccl().GetErrorString(r));
    return DRT_OK;
}

int32_t drt_comm_rank(drt_comm_t comm) { return comm ? comm->rank : 0; }
int32_t drt_comm_world(drt_comm_t comm) { return comm ? comm->world : 1; }

int32_t drt_allreduce_min_u64(drt_comm_t comm, uint64_t *buf, int64_t n, void *stream) {
    DRT_REQUIRE(comm && n >= 0, "bad argument");
    if (n == 0) return DRT_OK;
    DRT_REQUIRE(buf, "null pointer");
    DRT_NCCL(rccl().AllReduce(buf, buf, (size_t)n, ncclUint64, ncclMin, comm->comm, as_stream(stream)));
    return DRT_OK;
}

int32_t drt_allreduce_max_u8(drt_comm_t comm, uint8_t *buf, int64_t n, void *stream) {
    DRT_REQUIRE(comm && n >= 0, "bad argument");
    if (n == 0) return DRT_OK;
    DRT_REQUIRE(buf, "null pointer");
    DRT_NCCL(rccl().AllReduce(buf, buf, (size_t)n, ncclUint8, ncclMax, comm->comm, as_stream(stream)));
    return DRT_OK;
}

int32_t drt_allreduce_sum_f32(drt_comm_t comm, float *buf, int64_t n, void *stream) {
    DRT_REQUIRE(comm && n >= 0, "bad argument");
    if (n == 0) return DRT_OK;
    DRT_REQUIRE(buf, "null pointer");
    DRT_NCCL(rccl().AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, comm->comm, as_stream(stream)));
    return DRT_OK;
}

int32_t drt_allgather_bytes(drt_comm_t comm, const void *send, void *recv, int64_t bytes_per_rank, void *stream) {
    DRT_REQUIRE(comm && bytes_per_rank >= 0, "bad argument");
    if (bytes_per_rank == 0) return DRT_OK;
    DRT_REQUIRE(send && recv, "null pointer");
    DRT_NCCL(rccl().AllGather(send, recv, (size_t)bytes_per_rank, ncclUint8, comm->comm, as_stream(stream)));
    return DRT_OK;
}

}  // extern "C"
