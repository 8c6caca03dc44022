// RCCL collectives of the C ABI (SURVEY.md 8b lower boundary, section 5 "Distributed communication backend").
//
// The reference has no distributed code at all; what this build adds is plain data parallelism, one process
// per GPU: all-gather of the per-rank slices of the class-embedding matrix (evaluation, model/clip_tree.py:318-325 sharded),
// all-reduce(sum) of the 9 metric counters (main.py:121-128) and of the flat fp32 gradient buffer (OM training,
// main.py:87-91), broadcast of parameters.  They run on RCCL over xGMI, stream-ordered on the caller's HIP stream.
//
// librccl is resolved at run time (dlopen of its soname "librccl.so.1"): a process that already carries an RCCL - PyTorch's
// wheel bundles one for torch.distributed - shares that copy, and libhgr.so stays loadable on a box without RCCL.  The
// communicator handle is the one piece of process-wide state the library keeps; it is created and destroyed explicitly.
#include "hgr_common.h"
#include <dlfcn.h>
#include <string.h>

namespace {

typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[HGR_COMM_ID_BYTES]; } ncclUniqueId;          // NCCL_UNIQUE_ID_BYTES = 128 (rccl.h:40)
enum { ncclSuccess = 0 };
enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclInt64 = 4, ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat64 = 8, ncclBfloat16 = 9 };
enum { ncclSum = 0, ncclMax = 2 };

struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(ncclUniqueId *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
} g_rccl;

ncclComm_t g_comm = nullptr;
int g_rank = -1, g_world = 0;

int load_rccl() {
    if (g_rccl.handle) return HGR_OK;
    void *h = nullptr;
    const char *override_path = getenv("HGR_RCCL_LIB");
    const char *names[] = {override_path, "librccl.so.1", "librccl.so"};
    for (const char *n : names) {
        if (!n) continue;
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) return hgr_set_error(HGR_EUNSUPPORTED, "hgr_comm: librccl.so.1 not found (%s); set HGR_RCCL_LIB", dlerror());
#define HGR_SYM(field, name) do { *(void **)(&g_rccl.field) = dlsym(h, name); \
        if (!g_rccl.field) { dlclose(h); return hgr_set_error(HGR_EUNSUPPORTED, "hgr_comm: %s missing from librccl", name); } } while (0)
    HGR_SYM(GetUniqueId, "ncclGetUniqueId");
    HGR_SYM(CommInitRank, "ncclCommInitRank");
    HGR_SYM(CommDestroy, "ncclCommDestroy");
    HGR_SYM(AllReduce, "ncclAllReduce");
    HGR_SYM(AllGather, "ncclAllGather");
    HGR_SYM(Broadcast, "ncclBroadcast");
    HGR_SYM(GetErrorString, "ncclGetErrorString");
#undef HGR_SYM
    g_rccl.handle = h;
    return HGR_OK;
}

int dtype_of(int hgr_comm_dtype, size_t *size) {
    switch (hgr_comm_dtype) {
        case HGR_COMM_F32: *size = 4; return ncclFloat32;
        case HGR_COMM_F64: *size = 8; return ncclFloat64;
        case HGR_COMM_F16: *size = 2; return ncclFloat16;
        case HGR_COMM_BF16: *size = 2; return ncclBfloat16;
        case HGR_COMM_I32: *size = 4; return ncclInt32;
        case HGR_COMM_I64: *size = 8; return ncclInt64;
        case HGR_COMM_U8: *size = 1; return ncclUint8;
        default: return -1;
    }
}

#define HGR_RCCL(call, name) do { int rc__ = (call); \
    if (rc__ != ncclSuccess) return hgr_set_error(HGR_ELAUNCH, "%s: RCCL error %d (%s)", name, rc__, g_rccl.GetErrorString(rc__)); } while (0)

}  // namespace

extern "C" int hgr_comm_unique_id(void *id_out) {
    HGR_REQUIRE(id_out, "hgr_comm_unique_id: null buffer (HGR_COMM_ID_BYTES = %d bytes)", HGR_COMM_ID_BYTES);
    if (int rc = load_rccl()) return rc;
    ncclUniqueId id;
    HGR_RCCL(g_rccl.GetUniqueId(&id), "hgr_comm_unique_id");
    memcpy(id_out, &id, HGR_COMM_ID_BYTES);
    return HGR_OK;
}

extern "C" int hgr_comm_init(int rank, int world, const void *unique_id) {
    HGR_REQUIRE(world >= 1 && rank >= 0 && rank < world, "hgr_comm_init: bad rank %d of %d", rank, world);
    HGR_REQUIRE(unique_id, "hgr_comm_init: null unique id (rank 0 creates it with hgr_comm_unique_id and ships its %d bytes to every rank)", HGR_COMM_ID_BYTES);
    HGR_REQUIRE(!g_comm, "hgr_comm_init: a communicator already exists (rank %d of %d): hgr_comm_destroy first", g_rank, g_world);
    if (int rc = load_rccl()) return rc;
    ncclUniqueId id;
    memcpy(&id, unique_id, HGR_COMM_ID_BYTES);
    HGR_RCCL(g_rccl.CommInitRank(&g_comm, world, id, rank), "hgr_comm_init");     // binds to the calling thread's current HIP device
    g_rank = rank; g_world = world;
    return HGR_OK;
}

extern "C" int hgr_comm_destroy(void) {
    if (!g_comm) return HGR_OK;
    ncclComm_t c = g_comm;
    g_comm = nullptr; g_rank = -1; g_world = 0;
    HGR_RCCL(g_rccl.CommDestroy(c), "hgr_comm_destroy");
    return HGR_OK;
}

extern "C" int hgr_comm_rank(void) { return g_rank; }
extern "C" int hgr_comm_world(void) { return g_world; }

extern "C" int hgr_allreduce(const void *send, void *recv, int64_t count, int dtype, int op, void *stream) {
    size_t sz;
    const int dt = dtype_of(dtype, &sz);
    HGR_REQUIRE(send && recv && count >= 0, "hgr_allreduce: null buffer / negative count");
    HGR_REQUIRE(dt >= 0, "hgr_allreduce: bad dtype %d", dtype);
    HGR_REQUIRE(op == HGR_COMM_SUM || op == HGR_COMM_MAX, "hgr_allreduce: op must be HGR_COMM_SUM or HGR_COMM_MAX, got %d", op);
    HGR_REQUIRE(g_comm, "hgr_allreduce: no communicator (hgr_comm_init)");
    if (count == 0) return HGR_OK;
    HGR_RCCL(g_rccl.AllReduce(send, recv, (size_t)count, dt, op == HGR_COMM_SUM ? ncclSum : ncclMax, g_comm, (hipStream_t)stream), "hgr_allreduce");
    return HGR_OK;
}

extern "C" int hgr_allgather(const void *send, void *recv, int64_t count_per_rank, int dtype, void *stream) {
    size_t sz;
    const int dt = dtype_of(dtype, &sz);
    HGR_REQUIRE(send && recv && count_per_rank >= 0, "hgr_allgather: null buffer / negative count");
    HGR_REQUIRE(dt >= 0, "hgr_allgather: bad dtype %d", dtype);
    HGR_REQUIRE(g_comm, "hgr_allgather: no communicator (hgr_comm_init)");
    if (count_per_rank == 0) return HGR_OK;
    HGR_RCCL(g_rccl.AllGather(send, recv, (size_t)count_per_rank, dt, g_comm, (hipStream_t)stream), "hgr_allgather");
    return HGR_OK;
}

extern "C" int hgr_broadcast(void *buf, int64_t count, int dtype, int root, void *stream) {
    size_t sz;
    const int dt = dtype_of(dtype, &sz);
    HGR_REQUIRE(buf && count >= 0, "hgr_broadcast: null buffer / negative count");
    HGR_REQUIRE(dt >= 0, "hgr_broadcast: bad dtype %d", dtype);
    HGR_REQUIRE(g_comm, "hgr_broadcast: no communicator (hgr_comm_init)");
    HGR_REQUIRE(root >= 0 && root < g_world, "hgr_broadcast: root %d outside world %d", root, g_world);
    if (count == 0) return HGR_OK;
    HGR_RCCL(g_rccl.Broadcast(buf, buf, (size_t)count, dt, root, g_comm, (hipStream_t)stream), "hgr_broadcast");
    return HGR_OK;
}
