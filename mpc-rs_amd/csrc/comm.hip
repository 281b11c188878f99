// Multi-GPU exchange behind the C ABI: one RCCL communicator per context (one rank per GPU), used only after a
// sharded run -- for the fixed-size statistics records of the shards and, on request, for a result array gathered
// to one rank.  There is no collective on the data path (instances are independent).
//
// The reference has no counterpart (single thread, no devices).  RCCL is bound at run time (dlopen of librccl.so.1,
// the ROCm library that sits on the same HIP runtime as this one): libmpcx.so itself carries no link-time dependency
// on it, so single-GPU callers never load it.
#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <mutex>
#include <new>
#include <type_traits>

#include <rccl/rccl.h>

#include "mpcx_internal.h"

static_assert(MPCX_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id handed between ranks is RCCL's unique id");

namespace {

struct Rccl {
    void *handle;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *);
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*GroupStart)();
    ncclResult_t (*GroupEnd)();
    const char *(*GetErrorString)(ncclResult_t);
};

std::mutex g_rccl_mutex;
Rccl g_rccl;
char g_rccl_error[256];

// Binds RCCL once per process.  Returns null (and leaves a message in g_rccl_error) when the library or one of its
// entry points is missing.
const Rccl *rccl() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return &g_rccl;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *nm : names) {
        h = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
    }
    if (!h) {
        snprintf(g_rccl_error, sizeof(g_rccl_error), "cannot load librccl.so.1: %s", dlerror());
        return nullptr;
    }
    Rccl r{};
    bool ok = true;
    auto bind = [&](auto &fn, const char *sym) {
        void *p = dlsym(h, sym);
        if (!p) {
            snprintf(g_rccl_error, sizeof(g_rccl_error), "librccl lacks %s", sym);
            ok = false;
        }
        fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(p);
    };
    bind(r.GetUniqueId, "ncclGetUniqueId");
    bind(r.CommInitRank, "ncclCommInitRank");
    bind(r.CommDestroy, "ncclCommDestroy");
    bind(r.AllGather, "ncclAllGather");
    bind(r.Send, "ncclSend");
    bind(r.Recv, "ncclRecv");
    bind(r.GroupStart, "ncclGroupStart");
    bind(r.GroupEnd, "ncclGroupEnd");
    bind(r.GetErrorString, "ncclGetErrorString");
    if (!ok) {
        dlclose(h);
        return nullptr;
    }
    r.handle = h;
    g_rccl = r;
    return &g_rccl;
}

}  // namespace

struct mpcx_comm {
    const Rccl *api;
    ncclComm_t comm;
    mpcx_ctx *ctx;
    int device, world, rank;
    hipStream_t stream;     // the communicator's own stream: collectives never queue behind a caller's kernels
    hipEvent_t ready;       // orders the caller's stream before a gather that reads what it produced
    mpcx_stats *recs_dev;   // world records (all-gather target) + this rank's own record behind them
    char err[512];
};

namespace {

int cfail(mpcx_comm *c, int code, const char *fmt, ...) {
    if (c) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(c->err, sizeof(c->err), fmt, ap);
        va_end(ap);
    }
    return code;
}

#define COMM_HIP(c, call)                                                                      \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) return cfail(c, (int)e_, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)
#define COMM_NCCL(c, call)                                                                                  \
    do {                                                                                                    \
        ncclResult_t r_ = (call);                                                                           \
        if (r_ != ncclSuccess) return cfail(c, MPCX_ERR_COMM, "%s: %s", #call, (c)->api->GetErrorString(r_)); \
    } while (0)

}  // namespace

extern "C" {

int mpcx_comm_get_id(void *id_out) {
    if (!id_out) return MPCX_ERR_NULL;
    const Rccl *api = rccl();
    if (!api) return MPCX_ERR_COMM;
    ncclUniqueId id;
    if (api->GetUniqueId(&id) != ncclSuccess) return MPCX_ERR_COMM;
    memcpy(id_out, id.internal, MPCX_COMM_ID_BYTES);
    return MPCX_OK;
}

int mpcx_comm_create(mpcx_comm **out, mpcx_ctx *ctx, const void *id, int world, int rank) {
    if (!out) return MPCX_ERR_NULL;
    *out = nullptr;
    if (!ctx || !id) return MPCX_ERR_NULL;
    if (world < 1 || rank < 0 || rank >= world) return MPCX_ERR_DIMS;
    const Rccl *api = rccl();
    if (!api) return MPCX_ERR_COMM;
    mpcx_comm *c = new (std::nothrow) mpcx_comm();
    if (!c) return MPCX_ERR_NULL;
    c->api = api;
    c->ctx = ctx;
    c->device = mpcx::ctx_device(ctx);
    c->world = world;
    c->rank = rank;
    c->err[0] = 0;
    c->stream = nullptr;
    c->ready = nullptr;
    c->recs_dev = nullptr;
    // every failure below releases what was created before it and leaves its text for mpcx_comm_last_error(NULL)
    auto release = [&]() {
        if (c->recs_dev) (void)hipFree(c->recs_dev);
        if (c->ready) (void)hipEventDestroy(c->ready);
        if (c->stream) (void)hipStreamDestroy(c->stream);
        delete c;
    };
    hipError_t e = hipSetDevice(c->device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc((void **)&c->recs_dev, sizeof(mpcx_stats) * (size_t)(world + 1));
    if (e != hipSuccess) {
        snprintf(g_rccl_error, sizeof(g_rccl_error), "mpcx_comm_create on device %d: %s", c->device, hipGetErrorString(e));
        release();
        return (int)e;
    }
    ncclUniqueId uid;
    memcpy(uid.internal, id, MPCX_COMM_ID_BYTES);
    const ncclResult_t nr = api->CommInitRank(&c->comm, world, uid, rank);
    if (nr != ncclSuccess) {
        snprintf(g_rccl_error, sizeof(g_rccl_error), "ncclCommInitRank (rank %d of %d, device %d): %s", rank, world, c->device,
                 api->GetErrorString(nr));
        release();
        return MPCX_ERR_COMM;
    }
    *out = c;
    return MPCX_OK;
}

void mpcx_comm_destroy(mpcx_comm *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    (void)c->api->CommDestroy(c->comm);
    (void)hipFree(c->recs_dev);
    (void)hipEventDestroy(c->ready);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

const char *mpcx_comm_last_error(const mpcx_comm *c) {
    if (c) return c->err;
    return g_rccl_error[0] ? g_rccl_error : "communicator is NULL";
}

int mpcx_comm_rank(const mpcx_comm *c) { return c ? c->rank : -1; }
int mpcx_comm_world(const mpcx_comm *c) { return c ? c->world : 0; }

int mpcx_gather_stats(mpcx_comm *c, const double *buf_dev, int64_t count, void *stream, mpcx_stats *out_host) {
    if (!c) return MPCX_ERR_NULL;
    if (!out_host) return cfail(c, MPCX_ERR_NULL, "out_host is NULL");
    COMM_HIP(c, hipSetDevice(c->device));
    // this rank's record, reduced on the stream that produced the array, then the exchange on the communicator's stream
    mpcx_stats *mine = c->recs_dev + c->world;
    const int rc = mpcx_stats_f64_dev(c->ctx, buf_dev, count, mine, stream);
    if (rc != MPCX_OK) return cfail(c, rc, "stats: %s", mpcx_last_error(c->ctx));
    COMM_HIP(c, hipEventRecord(c->ready, (hipStream_t)stream));
    COMM_HIP(c, hipStreamWaitEvent(c->stream, c->ready, 0));
    COMM_NCCL(c, c->api->AllGather(mine, c->recs_dev, sizeof(mpcx_stats), ncclInt8, c->comm, c->stream));
    COMM_HIP(c, hipMemcpyAsync(out_host, c->recs_dev, sizeof(mpcx_stats) * (size_t)c->world, hipMemcpyDeviceToHost, c->stream));
    COMM_HIP(c, hipStreamSynchronize(c->stream));
    return MPCX_OK;
}

int mpcx_gather_f64_dev(mpcx_comm *c, const double *src_dev, int64_t count, double *dst_dev, int root, void *stream) {
    if (!c) return MPCX_ERR_NULL;
    if (count < 0) return cfail(c, MPCX_ERR_DIMS, "count must be >= 0");
    if (root < 0 || root >= c->world) return cfail(c, MPCX_ERR_DIMS, "root %d is not a rank of this communicator (world %d)", root, c->world);
    if (count == 0) return MPCX_OK;
    if (!src_dev) return cfail(c, MPCX_ERR_NULL, "src_dev is NULL");
    if (c->rank == root && !dst_dev) return cfail(c, MPCX_ERR_NULL, "dst_dev is NULL on the root");
    COMM_HIP(c, hipSetDevice(c->device));
    COMM_HIP(c, hipEventRecord(c->ready, (hipStream_t)stream));
    COMM_HIP(c, hipStreamWaitEvent(c->stream, c->ready, 0));
    // Fan-in by point-to-point transfers: every source rank sends over its own xGMI link to the root, which posts all
    // its receives in one group (a ring would push every shard through every link).
    if (c->rank == root) {
        COMM_HIP(c, hipMemcpyAsync(dst_dev + (size_t)root * (size_t)count, src_dev, sizeof(double) * (size_t)count,
                                   hipMemcpyDeviceToDevice, c->stream));
        if (c->world > 1) {
            COMM_NCCL(c, c->api->GroupStart());
            for (int r = 0; r < c->world; ++r) {
                if (r == root) continue;
                const ncclResult_t nr = c->api->Recv(dst_dev + (size_t)r * (size_t)count, (size_t)count, ncclDouble, r, c->comm, c->stream);
                if (nr != ncclSuccess) {
                    (void)c->api->GroupEnd();
                    return cfail(c, MPCX_ERR_COMM, "ncclRecv from rank %d: %s", r, c->api->GetErrorString(nr));
                }
            }
            COMM_NCCL(c, c->api->GroupEnd());
        }
    } else {
        COMM_NCCL(c, c->api->Send(src_dev, (size_t)count, ncclDouble, root, c->comm, c->stream));
    }
    // the caller's stream continues after the transfer (it may overwrite src or read dst next)
    COMM_HIP(c, hipEventRecord(c->ready, c->stream));
    COMM_HIP(c, hipStreamWaitEvent((hipStream_t)stream, c->ready, 0));
    return MPCX_OK;
}

int mpcx_comm_synchronize(mpcx_comm *c) {
    if (!c) return MPCX_ERR_NULL;
    COMM_HIP(c, hipSetDevice(c->device));
    COMM_HIP(c, hipStreamSynchronize(c->stream));
    return MPCX_OK;
}

}  // extern "C"
