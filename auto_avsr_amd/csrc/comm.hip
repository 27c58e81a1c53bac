// comm.hip -- the collectives of the data-parallel step (train.py:30-42: DDP gradient averaging, SyncBatchNorm statistics,
// lightning.py:88-90 batch-size all-gather) issued straight through RCCL's C API on a HIP stream of the caller's choice.
//
// Why not torch.distributed for these calls: every collective issued through a torch process group creates a Work object
// whose completion events a watchdog THREAD polls; while the training step is being captured into a hipGraph those events
// are capture-time events, and the poll aborts the process (hipErrorCapturedEvent) -- DESIGN.md section 6.  Here a collective
// is nothing but a stream operation: it is captured like a kernel launch and replayed with the graph.
//
// RCCL is bound at run time (dlopen of the librccl the process already holds -- torch's bundled copy -- or the system one):
// the library loads and every other entry point works on a box without RCCL; only avsr_comm_* then report the failure.
// Up to four communicators per process, addressed by a small slot number (one process per GPU; RCCL serialises the operations
// of ONE communicator in issue order even across streams, so the gradient buckets on their side stream and the latency-bound
// BatchNorm collectives on the compute stream get a communicator each).  Bootstrap per slot: rank 0 calls avsr_comm_unique_id, hands the 128 bytes to
// the other ranks through whatever rendezvous the host side has (auto_avsr_amd/comm.py: a torch.distributed broadcast), then
// every rank calls avsr_comm_init.
#include <stdint.h>
#include <string.h>
#include "prims.h"
#include "avsr_hip.h"

#ifndef AVSR_EMU
#include <dlfcn.h>

namespace {

struct NcclUniqueId { char internal[128]; };  // rccl.h: NCCL_UNIQUE_ID_BYTES
typedef void* NcclComm;
constexpr int kNcclFloat32 = 7, kNcclBfloat16 = 9, kNcclSum = 0;  // rccl.h: ncclFloat32, ncclBfloat16, ncclSum

struct Api {
    void* lib = nullptr;
    int (*GetUniqueId)(NcclUniqueId*) = nullptr;
    int (*CommInitRank)(NcclComm*, int, NcclUniqueId, int) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, NcclComm, hipStream_t) = nullptr;
    int (*ReduceScatter)(const void*, void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
} g_api;
constexpr int kSlots = 4;
NcclComm g_comm[kSlots] = {nullptr, nullptr, nullptr, nullptr};
int g_nranks[kSlots] = {0, 0, 0, 0};

bool load_api() {
    if (g_api.lib) return true;
    const char* names[] = {"librccl.so.1", "librccl.so", nullptr};
    void* h = nullptr;
    for (int i = 0; names[i] && !h; i++) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);  // the copy already in the process
    for (int i = 0; names[i] && !h; i++) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        avsr_set_error("comm: librccl.so not found (dlopen)");
        return false;
    }
    Api a;
    a.lib = h;
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(h, "ncclAllReduce"));
    a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(h, "ncclAllGather"));
    a.ReduceScatter = reinterpret_cast<decltype(a.ReduceScatter)>(dlsym(h, "ncclReduceScatter"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce || !a.AllGather) {
        avsr_set_error("comm: librccl.so lacks the NCCL entry points");
        return false;
    }
    g_api = a;
    return true;
}

int check(int rc, const char* what) {
    if (rc == 0) return 0;
    static char msg[256];
    snprintf(msg, sizeof msg, "comm: %s failed: %s", what, g_api.GetErrorString ? g_api.GetErrorString(rc) : "?");
    avsr_set_error(msg);
    return 1;
}

}  // namespace

extern "C" int avsr_comm_unique_id(void* out128) {
    if (!load_api()) return 1;
    NcclUniqueId id;
    if (check(g_api.GetUniqueId(&id), "ncclGetUniqueId")) return 1;
    memcpy(out128, &id, sizeof id);
    return 0;
}

extern "C" int avsr_comm_init(int slot, const void* id128, int nranks, int rank) {
    AVSR_REQUIRE(slot >= 0 && slot < kSlots, "comm_init: slot must be 0..3");
    AVSR_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "comm_init: bad rank / size");
    AVSR_REQUIRE(g_comm[slot] == nullptr, "comm_init: the slot holds a communicator already (avsr_comm_destroy first)");
    if (!load_api()) return 1;
    NcclUniqueId id;
    memcpy(&id, id128, sizeof id);
    if (check(g_api.CommInitRank(&g_comm[slot], nranks, id, rank), "ncclCommInitRank")) {
        g_comm[slot] = nullptr;
        return 1;
    }
    g_nranks[slot] = nranks;
    return 0;
}

extern "C" int avsr_comm_destroy(int slot) {
    AVSR_REQUIRE(slot >= 0 && slot < kSlots, "comm_destroy: slot must be 0..3");
    if (g_comm[slot]) {
        const int rc = g_api.CommDestroy(g_comm[slot]);
        g_comm[slot] = nullptr;
        g_nranks[slot] = 0;
        return check(rc, "ncclCommDestroy");
    }
    return 0;
}

extern "C" int64_t avsr_comm_size(int slot) { return slot >= 0 && slot < kSlots && g_comm[slot] ? g_nranks[slot] : 0; }

extern "C" int avsr_comm_all_reduce_f32(int slot, void* buf, int64_t count, hipStream_t stream) {
    AVSR_REQUIRE(slot >= 0 && slot < kSlots && g_comm[slot] != nullptr, "comm_all_reduce: no communicator in this slot (avsr_comm_init)");
    if (count <= 0) return 0;
    return check(g_api.AllReduce(buf, buf, (size_t)count, kNcclFloat32, kNcclSum, g_comm[slot], stream), "ncclAllReduce");
}

// dtype: 0 = f32, 1 = bf16 (the narrow wire format of the gradient buckets: half the bytes per xGMI link, sums formed in bf16)
extern "C" int avsr_comm_all_reduce(int slot, void* buf, int64_t count, int dtype, hipStream_t stream) {
    AVSR_REQUIRE(slot >= 0 && slot < kSlots && g_comm[slot] != nullptr, "comm_all_reduce: no communicator in this slot (avsr_comm_init)");
    AVSR_REQUIRE(dtype == 0 || dtype == 1, "comm_all_reduce: dtype must be 0 (f32) or 1 (bf16)");
    if (count <= 0) return 0;
    return check(g_api.AllReduce(buf, buf, (size_t)count, dtype == 1 ? kNcclBfloat16 : kNcclFloat32, kNcclSum, g_comm[slot], stream),
                 "ncclAllReduce");
}

extern "C" int avsr_comm_all_gather_f32(int slot, const void* send, void* recv, int64_t count_per_rank, hipStream_t stream) {
    AVSR_REQUIRE(slot >= 0 && slot < kSlots && g_comm[slot] != nullptr, "comm_all_gather: no communicator in this slot (avsr_comm_init)");
    if (count_per_rank <= 0) return 0;
    return check(g_api.AllGather(send, recv, (size_t)count_per_rank, kNcclFloat32, g_comm[slot], stream), "ncclAllGather");
}

// recv[0 .. count_per_rank) = sum over ranks of their send[rank * count_per_rank ..) -- the gradient exchange of the sharded optimizer
// (ddp.GradBuckets shard=True): half the wire traffic of an all-reduce inside the backward pass; recv may be send + rank * count
// (in place).  dtype 0 = f32, 1 = bf16.
extern "C" int avsr_comm_reduce_scatter(int slot, const void* send, void* recv, int64_t count_per_rank, int dtype, hipStream_t stream) {
    AVSR_REQUIRE(slot >= 0 && slot < kSlots && g_comm[slot] != nullptr, "comm_reduce_scatter: no communicator in this slot (avsr_comm_init)");
    AVSR_REQUIRE(dtype == 0 || dtype == 1, "comm_reduce_scatter: dtype must be 0 (f32) or 1 (bf16)");
    AVSR_REQUIRE(g_api.ReduceScatter != nullptr, "comm_reduce_scatter: librccl.so lacks ncclReduceScatter");
    if (count_per_rank <= 0) return 0;
    return check(g_api.ReduceScatter(send, recv, (size_t)count_per_rank, dtype == 1 ? kNcclBfloat16 : kNcclFloat32, kNcclSum, g_comm[slot],
                                     stream), "ncclReduceScatter");
}

#else  // host emulator build (CPU test suite): there is no RCCL; the data-parallel tests run on torch.distributed / gloo

extern "C" int avsr_comm_unique_id(void*) { avsr_set_error("comm: not available in the emulator build"); return 1; }
extern "C" int avsr_comm_init(int, const void*, int, int) { avsr_set_error("comm: not available in the emulator build"); return 1; }
extern "C" int avsr_comm_destroy(int) { return 0; }
extern "C" int64_t avsr_comm_size(int) { return 0; }
extern "C" int avsr_comm_all_reduce_f32(int, void*, int64_t, hipStream_t) { avsr_set_error("comm: not available in the emulator build"); return 1; }
extern "C" int avsr_comm_all_reduce(int, void*, int64_t, int, hipStream_t) { avsr_set_error("comm: not available in the emulator build"); return 1; }
extern "C" int avsr_comm_all_gather_f32(int, const void*, void*, int64_t, hipStream_t) { avsr_set_error("comm: not available in the emulator build"); return 1; }
extern "C" int avsr_comm_reduce_scatter(int, const void*, void*, int64_t, int, hipStream_t) { avsr_set_error("comm: not available in the emulator build"); return 1; }

#endif
