// comm.hip — the one collective of the path behind the C ABI: all-gather of Merkle roots (or of any small per-rank
// record, e.g. the subtree roots of a row-sharded FRI layer) over RCCL / xGMI, one process per GPU.
//
// Serves the root mixing of pcs/prover.ts:62-64,227-228 (Rust text: every tree's root enters the channel in TreeVec order)
// when the trace columns are sharded over GPUs: rank g commits its own tree over its own columns and
// the ranks exchange 32-byte roots — nothing else crosses xGMI.
//
// RCCL is bound at run time (dlopen / dlsym), not at link time: a single-GPU host needs no librccl to load
// libtstwo_hip.so, and a process that already holds an RCCL (PyTorch's) shares that copy instead of mapping a second.
// The collective is enqueued on the library's stream like every other entry point: ordered behind the Merkle kernels
// that produce the root, no host synchronisation.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

#include "common.h"

using namespace tstwo;

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
ncclComm_t g_comm = nullptr;
int g_rank = 0, g_world = 1;
// overlap: collectives issued with tstwo_allgather_async run on a stream of their own, fenced against the library's
// stream by two events (payload ready -> collective; collective done -> whoever calls tstwo_comm_wait)
hipStream_t g_comm_stream = nullptr;
hipEvent_t g_ev_ready = nullptr, g_ev_done = nullptr;
bool g_async_pending = false;

int ensure_comm_stream() {
    if (g_comm_stream) return TSTWO_OK;
    TSTWO_HIP(hipStreamCreateWithFlags(&g_comm_stream, hipStreamNonBlocking));
    TSTWO_HIP(hipEventCreateWithFlags(&g_ev_ready, hipEventDisableTiming));
    TSTWO_HIP(hipEventCreateWithFlags(&g_ev_done, hipEventDisableTiming));
    return TSTWO_OK;
}

int load_rccl() {
    if (g_rccl.handle) return TSTWO_OK;
    // TSTWO_RCCL_LIB, when set, is the ONLY candidate (an explicit path that does not load is an error, not a hint)
    const char *forced = getenv("TSTWO_RCCL_LIB");
    const char *defaults[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    const char *names[3] = {nullptr, nullptr, nullptr};
    if (forced && *forced) names[0] = forced;
    else for (int i = 0; i < 3; i++) names[i] = defaults[i];
    void *h = nullptr;
    for (const char *n : names)      // a copy already mapped by the process (PyTorch's) first
        if (n && !h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL);
    for (const char *n : names)
        if (n && !h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);       // private: symbols are taken with dlsym only
    if (!h) {
        const char *e = dlerror();            // ONE call: dlerror() clears the state it returns
        return set_error(TSTWO_ERR_COMM, std::string("RCCL is not available: ") + (e ? e : "librccl.so not found") +
                                             " (set TSTWO_RCCL_LIB to its path)");
    }
    Rccl r;
    r.handle = h;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))dlsym(h, "ncclAllGather");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.GetErrorString)
        return set_error(TSTWO_ERR_COMM, "RCCL library lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather");
    g_rccl = r;
    return TSTWO_OK;
}
int rccl_fail(ncclResult_t e, const char *what) {
    return set_error(TSTWO_ERR_COMM, std::string("RCCL error: ") + g_rccl.GetErrorString(e) + " in " + what);
}
#define TSTWO_RCCL(call) do { ncclResult_t _r = (call); if (_r != ncclSuccess) return rccl_fail(_r, #call); } while (0)

}  // namespace

extern "C" {

int tstwo_comm_unique_id(uint8_t id[TSTWO_COMM_ID_BYTES]) {
    static_assert(sizeof(ncclUniqueId) == TSTWO_COMM_ID_BYTES, "ncclUniqueId size");
    if (!id) return set_error(TSTWO_ERR_BAD_ARG, "comm: null id buffer");
    int rc = load_rccl();
    if (rc) return rc;
    ncclUniqueId u;
    TSTWO_RCCL(g_rccl.GetUniqueId(&u));
    memcpy(id, &u, sizeof(u));
    return TSTWO_OK;
}

int tstwo_comm_init(int rank, int world, const uint8_t id[TSTWO_COMM_ID_BYTES]) {
    TSTWO_REQUIRE_READY();
    if (world < 1 || rank < 0 || rank >= world) return set_error(TSTWO_ERR_BAD_ARG, "comm: rank / world out of range");
    if (g_comm) return set_error(TSTWO_ERR_BAD_ARG, "comm: already initialised (tstwo_comm_destroy first)");
    if (!id) return set_error(TSTWO_ERR_BAD_ARG, "comm: null id");
    int rc = load_rccl();
    if (rc) return rc;
    TSTWO_HIP(hipSetDevice(ctx().device));
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclComm_t comm = nullptr;
    TSTWO_RCCL(g_rccl.CommInitRank(&comm, world, u, rank));     // collective: returns when every rank has joined
    g_comm = comm;
    g_rank = rank;
    g_world = world;
    return TSTWO_OK;
}

int tstwo_comm_destroy(void) {
    if (g_comm_stream) {
        (void)hipStreamSynchronize(g_comm_stream);
        (void)hipEventDestroy(g_ev_ready);
        (void)hipEventDestroy(g_ev_done);
        (void)hipStreamDestroy(g_comm_stream);
        g_comm_stream = nullptr;
        g_async_pending = false;
    }
    if (!g_comm) return TSTWO_OK;
    if (ctx().ready) (void)hipStreamSynchronize(ctx().stream);
    ncclComm_t c = g_comm;
    g_comm = nullptr;
    g_rank = 0;
    g_world = 1;
    TSTWO_RCCL(g_rccl.CommDestroy(c));
    return TSTWO_OK;
}

int tstwo_comm_info(int *rank, int *world) {
    if (rank) *rank = g_rank;
    if (world) *world = g_world;
    return TSTWO_OK;
}

int tstwo_allgather(const void *send_dev, void *recv_dev, size_t bytes_per_rank) {
    TSTWO_REQUIRE_READY();
    TSTWO_REQUIRE_PTRS(send_dev, recv_dev);
    if (bytes_per_rank == 0) return TSTWO_OK;
    Context &c = ctx();
    if (!g_comm) {      // no communicator = a world of one: the gather is a copy (same stream ordering as the collective)
        if (send_dev != recv_dev) TSTWO_HIP(hipMemcpyAsync(recv_dev, send_dev, bytes_per_rank, hipMemcpyDeviceToDevice, c.stream));
        return TSTWO_OK;
    }
    TSTWO_RCCL(g_rccl.AllGather(send_dev, recv_dev, bytes_per_rank, ncclUint8, g_comm, c.stream));
    return TSTWO_OK;
}

int tstwo_allgather_roots(const uint8_t *root_dev, uint8_t *roots_out_dev) { return tstwo_allgather(root_dev, roots_out_dev, 32); }

int tstwo_allgather_async(const void *send_dev, void *recv_dev, size_t bytes_per_rank) {
    TSTWO_REQUIRE_READY();
    TSTWO_REQUIRE_PTRS(send_dev, recv_dev);
    if (bytes_per_rank == 0) return TSTWO_OK;
    Context &c = ctx();
    int rc = ensure_comm_stream();
    if (rc) return rc;
    TSTWO_HIP(hipEventRecord(g_ev_ready, c.stream));                 // everything enqueued so far produced the payload
    TSTWO_HIP(hipStreamWaitEvent(g_comm_stream, g_ev_ready, 0));
    if (!g_comm) {
        if (send_dev != recv_dev) TSTWO_HIP(hipMemcpyAsync(recv_dev, send_dev, bytes_per_rank, hipMemcpyDeviceToDevice, g_comm_stream));
    } else {
        TSTWO_RCCL(g_rccl.AllGather(send_dev, recv_dev, bytes_per_rank, ncclUint8, g_comm, g_comm_stream));
    }
    TSTWO_HIP(hipEventRecord(g_ev_done, g_comm_stream));
    g_async_pending = true;
    return TSTWO_OK;
}

int tstwo_comm_wait(void) {
    TSTWO_REQUIRE_READY();
    if (!g_async_pending) return TSTWO_OK;
    TSTWO_HIP(hipStreamWaitEvent(ctx().stream, g_ev_done, 0));      // stream-side wait: the host does not block
    g_async_pending = false;
    return TSTWO_OK;
}

}  // extern "C"
