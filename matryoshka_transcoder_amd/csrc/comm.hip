// mtc_comm_*: the RCCL steps of the feature-sharded transcoder behind the C ABI, for hosts that do not
// go through torch.distributed.  RCCL is bound at run time: the entry points are looked up in the process first (a host
// that already carries an RCCL -- PyTorch does -- keeps using that one copy) and in librccl.so otherwise, so libmtc has
// no link-time dependency on it and never loads a second copy beside the host's.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <mutex>
#include "common.hpp"

namespace mtc {

struct RcclApi {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*);
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
    const char* (*GetErrorString)(ncclResult_t);
    bool ok;
};

static RcclApi g_rccl;
static std::once_flag g_rccl_once;

static void* find_symbol(void* lib, const char* name) {
    void* p = dlsym(RTLD_DEFAULT, name);
    if (!p && lib) p = dlsym(lib, name);
    return p;
}

static void bind_rccl() {
    void* lib = nullptr;
    if (!dlsym(RTLD_DEFAULT, "ncclAllReduce")) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!dlsym(RTLD_DEFAULT, "ncclAllReduce") && !lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(find_symbol(lib, "ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(find_symbol(lib, "ncclCommInitRank"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(find_symbol(lib, "ncclCommDestroy"));
    g_rccl.AllReduce = reinterpret_cast<decltype(g_rccl.AllReduce)>(find_symbol(lib, "ncclAllReduce"));
    g_rccl.ReduceScatter = reinterpret_cast<decltype(g_rccl.ReduceScatter)>(find_symbol(lib, "ncclReduceScatter"));
    g_rccl.AllGather = reinterpret_cast<decltype(g_rccl.AllGather)>(find_symbol(lib, "ncclAllGather"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(find_symbol(lib, "ncclGetErrorString"));
    g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllReduce && g_rccl.ReduceScatter &&
                g_rccl.AllGather;
}

static int rccl_ready(const char* who) {
    std::call_once(g_rccl_once, bind_rccl);
    if (!g_rccl.ok) { set_error("%s: RCCL not found (neither in the process nor as librccl.so)", who); return -1; }
    return 0;
}

static int rccl_check(ncclResult_t r, const char* who) {
    if (r == ncclSuccess) return 0;
    set_error("%s: RCCL error %d (%s)", who, (int)r, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    return -3;
}

static int to_nccl_type(int32_t t, ncclDataType_t* out) {
    switch (t) {
        case MTC_COMM_F32: *out = ncclFloat32; return 0;
        case MTC_COMM_BF16: *out = ncclBfloat16; return 0;
        case MTC_COMM_I32: *out = ncclInt32; return 0;
        case MTC_COMM_U8: *out = ncclUint8; return 0;
        case MTC_COMM_I64: *out = ncclInt64; return 0;
    }
    return -1;
}

static int to_nccl_op(int32_t o, ncclRedOp_t* out) {
    switch (o) {
        case MTC_COMM_SUM: *out = ncclSum; return 0;
        case MTC_COMM_MAX: *out = ncclMax; return 0;
        case MTC_COMM_MIN: *out = ncclMin; return 0;
    }
    return -1;
}

}  // namespace mtc

using namespace mtc;

struct mtc_comm {
    ncclComm_t comm;
    int world, rank;
};

extern "C" int mtc_comm_unique_id(void* id_host_128) {
    if (rccl_ready("mtc_comm_unique_id")) return -1;
    MTC_REQUIRE(id_host_128, "mtc_comm_unique_id: null pointer");
    static_assert(sizeof(ncclUniqueId) == MTC_COMM_ID_BYTES, "unique id size");
    return rccl_check(g_rccl.GetUniqueId(reinterpret_cast<ncclUniqueId*>(id_host_128)), "mtc_comm_unique_id");
}

extern "C" int mtc_comm_init(mtc_comm** out, int32_t world, int32_t rank, const void* id_host_128) {
    if (rccl_ready("mtc_comm_init")) return -1;
    MTC_REQUIRE(out && id_host_128, "mtc_comm_init: null pointer");
    MTC_REQUIRE(world >= 1 && rank >= 0 && rank < world, "mtc_comm_init: bad rank %d of %d", rank, world);
    ncclUniqueId id;
    __builtin_memcpy(&id, id_host_128, sizeof(id));
    ncclComm_t c = nullptr;
    if (int rc = rccl_check(g_rccl.CommInitRank(&c, world, id, rank), "mtc_comm_init")) return rc;
    mtc_comm* h = new mtc_comm{c, world, rank};
    *out = h;
    return 0;
}

extern "C" int mtc_comm_destroy(mtc_comm* c) {
    if (!c) return 0;
    if (rccl_ready("mtc_comm_destroy")) return -1;
    const int rc = rccl_check(g_rccl.CommDestroy(c->comm), "mtc_comm_destroy");
    delete c;
    return rc;
}

extern "C" int mtc_comm_all_reduce(mtc_comm* c, const void* send, void* recv, size_t count, int32_t dtype, int32_t op,
                                   void* stream) {
    if (rccl_ready("mtc_comm_all_reduce")) return -1;
    ncclDataType_t t; ncclRedOp_t o;
    MTC_REQUIRE(c && send && recv, "mtc_comm_all_reduce: null pointer");
    MTC_REQUIRE(!to_nccl_type(dtype, &t) && !to_nccl_op(op, &o), "mtc_comm_all_reduce: bad dtype %d / op %d", dtype, op);
    if (count == 0) return 0;
    return rccl_check(g_rccl.AllReduce(send, recv, count, t, o, c->comm, as_stream(stream)), "mtc_comm_all_reduce");
}

extern "C" int mtc_comm_reduce_scatter(mtc_comm* c, const void* send, void* recv, size_t recv_count, int32_t dtype,
                                       int32_t op, void* stream) {
    if (rccl_ready("mtc_comm_reduce_scatter")) return -1;
    ncclDataType_t t; ncclRedOp_t o;
    MTC_REQUIRE(c && send && recv, "mtc_comm_reduce_scatter: null pointer");
    MTC_REQUIRE(!to_nccl_type(dtype, &t) && !to_nccl_op(op, &o), "mtc_comm_reduce_scatter: bad dtype %d / op %d", dtype, op);
    if (recv_count == 0) return 0;
    return rccl_check(g_rccl.ReduceScatter(send, recv, recv_count, t, o, c->comm, as_stream(stream)), "mtc_comm_reduce_scatter");
}

extern "C" int mtc_comm_all_gather(mtc_comm* c, const void* send, void* recv, size_t send_count, int32_t dtype, void* stream) {
    if (rccl_ready("mtc_comm_all_gather")) return -1;
    ncclDataType_t t;
    MTC_REQUIRE(c && send && recv, "mtc_comm_all_gather: null pointer");
    MTC_REQUIRE(!to_nccl_type(dtype, &t), "mtc_comm_all_gather: bad dtype %d", dtype);
    if (send_count == 0) return 0;
    return rccl_check(g_rccl.AllGather(send, recv, send_count, t, c->comm, as_stream(stream)), "mtc_comm_all_gather");
}

extern "C" int mtc_comm_world(const mtc_comm* c) { return c ? c->world : -1; }
extern "C" int mtc_comm_rank(const mtc_comm* c) { return c ? c->rank : -1; }
