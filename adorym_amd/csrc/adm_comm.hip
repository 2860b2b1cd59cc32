// Collectives of the data-parallel mode behind the C ABI: RCCL over xGMI, one communicator per context stream (main, and
// optionally the side stream: adm_comm_init_aux), every call enqueued on the context's current stream (ordered with the
// kernels, no host synchronisation).
//
// Replaces the reference's mpi4py object collectives on the hot path:
//   gradient.arr = comm.allreduce(gradient.arr)            adorym/ptychography.py:1113-1114  -> adm_reduce_scatter (+ adm_all_gather
//                                                           of the updated object shards after the fused optimiser step)
//   comm.allreduce(w.to_numpy(opt.grads))                  adorym/optimizers.py:1025,1041,1053,1064,1079 -> adm_all_reduce on the
//                                                           device buffers (the small gradients never visit the host)
//   comm.bcast(...)                                        adorym/ptychography.py:217,411,485-487,664-665 -> adm_broadcast
// librccl is loaded with dlopen at the first adm_comm_* call, so libadm has no link-time dependency on it and single-GPU
// runs never touch it.  The unique id travels between the processes by whatever rendezvous the host side has
// (adorym_amd/comm.py uses its own TCP star on MASTER_ADDR / MASTER_PORT: adorym_amd/rendezvous.py).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstring>
#include <mutex>
#include <string>
#include "adm_common.h"

namespace {

struct UniqueId { char internal[128]; };          // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128), passed BY VALUE
typedef void* Comm;                                // ncclComm_t
enum { kFloat32 = 7, kUint8 = 1 };                 // ncclDataType_t
enum { kSum = 0, kMax = 2 };                       // ncclRedOp_t

struct Api {
    void* lib = nullptr;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*ReduceScatter)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, Comm, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*Reduce)(const void*, void*, size_t, int, int, int, Comm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string err;
};

void load_api(Api& a);

// One-time dlopen / dlsym, safe against a first call from two threads at once (the driver runs a checkpoint writer thread).
Api* api() {
    static Api a;
    static std::once_flag once;
    std::call_once(once, [] { load_api(a); });
    return &a;
}

void load_api(Api& a) {
    // The RCCL copy must sit on the HIP/HSA runtime this process already uses: a process that imported PyTorch first
    // runs on the runtime PyTorch ships (and must use its librccl), one that loaded libadm first runs on ROCm's.  Mixing
    // them gives "no ROCm-capable device" inside ncclCommInitRank (RCCL dlopens libhsa-runtime64 next to itself).  So:
    // the librccl that lives beside the libamdhip64 in use, then the generic names.
    std::string beside;
    Dl_info info;
    if (dladdr(reinterpret_cast<void*>(&hipGetDeviceCount), &info) && info.dli_fname) {
        beside = info.dli_fname;
        const size_t slash = beside.rfind('/');
        beside = (slash == std::string::npos) ? std::string() : beside.substr(0, slash);
    }
    if (!beside.empty()) {
        const std::string cands[] = {beside + "/librccl.so.1", beside + "/librccl.so"};
        for (const std::string& c : cands) {
            a.lib = dlopen(c.c_str(), RTLD_NOW | RTLD_GLOBAL);
            if (a.lib) break;
        }
    }
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (int i = 0; !a.lib && i < 3; ++i) a.lib = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!a.lib) {
        const char* why = dlerror();
        a.err = std::string("cannot load librccl: ") + (why ? why : "?");
        return;
    }
#define ADM_SYM(field, name)                                                     \
    a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.lib, name));          \
    if (!a.field) { a.err = std::string("librccl lacks ") + name; return; }
    ADM_SYM(GetUniqueId, "ncclGetUniqueId")
    ADM_SYM(CommInitRank, "ncclCommInitRank")
    ADM_SYM(CommDestroy, "ncclCommDestroy")
    ADM_SYM(ReduceScatter, "ncclReduceScatter")
    ADM_SYM(GroupStart, "ncclGroupStart")
    ADM_SYM(GroupEnd, "ncclGroupEnd")
    ADM_SYM(AllGather, "ncclAllGather")
    ADM_SYM(AllReduce, "ncclAllReduce")
    ADM_SYM(Broadcast, "ncclBroadcast")
    ADM_SYM(Reduce, "ncclReduce")
    ADM_SYM(GetErrorString, "ncclGetErrorString")
#undef ADM_SYM
}

// The communicator a collective issued NOW belongs to: work queued between adm_ctx_fork and adm_ctx_end_fork goes to the
// side stream and uses the side communicator when there is one, so that every communicator only ever sees one stream.
Comm comm_for(adm_ctx* ctx) {
    return (ctx->stream == ctx->aux_stream && ctx->comm_aux) ? ctx->comm_aux : ctx->comm;
}

int nccl_fail(int rc, const char* what) {
    Api* a = api();
    return adm::fail(ADM_ERR_HIP, std::string(what) + ": " + (a->GetErrorString ? a->GetErrorString(rc) : "RCCL error"));
}

int need(adm_ctx* ctx, const char* what, bool want_comm) {
    if (!ctx) return adm::fail(ADM_ERR_INVALID, std::string(what) + ": null context");
    Api* a = api();
    if (!a->err.empty()) return adm::fail(ADM_ERR_UNSUPPORTED, std::string(what) + ": " + a->err);
    if (want_comm && !ctx->comm) return adm::fail(ADM_ERR_INVALID, std::string(what) + ": adm_comm_init has not been called on this context");
    return ADM_OK;
}

}  // namespace

using adm::fail;

extern "C" int adm_comm_unique_id(void* out128) {
    if (!out128) return fail(ADM_ERR_INVALID, "adm_comm_unique_id: null argument");
    Api* a = api();
    if (!a->err.empty()) return fail(ADM_ERR_UNSUPPORTED, "adm_comm_unique_id: " + a->err);
    UniqueId id;
    const int rc = a->GetUniqueId(&id);
    if (rc) return nccl_fail(rc, "ncclGetUniqueId");
    std::memcpy(out128, id.internal, sizeof(id.internal));
    return ADM_OK;
}

extern "C" int adm_comm_init(adm_ctx* ctx, int rank, int nranks, const void* unique_id128) {
    int rc = need(ctx, "adm_comm_init", false);
    if (rc) return rc;
    if (!unique_id128 || nranks < 1 || rank < 0 || rank >= nranks) return fail(ADM_ERR_INVALID, "adm_comm_init: bad argument");
    if (ctx->comm) return fail(ADM_ERR_INVALID, "adm_comm_init: this context already has a communicator");
    ADM_HIP(hipSetDevice(ctx->device));
    UniqueId id;
    std::memcpy(id.internal, unique_id128, sizeof(id.internal));
    Comm c = nullptr;
    rc = api()->CommInitRank(&c, nranks, id, rank);
    if (rc) return nccl_fail(rc, "ncclCommInitRank");
    ctx->comm = c;
    ctx->comm_rank = rank;
    ctx->comm_size = nranks;
    return ADM_OK;
}

// A second communicator over the same ranks for the collectives queued on the context's SIDE stream (the deferred part of
// the object all-gather runs there, beside the next multislice kernel, while the main stream goes on to the next
// reduce-scatter).  RCCL orders the operations of ONE communicator in the order they were issued, whatever streams they
// were given; with its own communicator the side-stream gather neither waits for nor delays main-stream collectives, and
// no communicator is ever driven from two streams.  Collective over all ranks, like adm_comm_init, with its own unique id.
extern "C" int adm_comm_init_aux(adm_ctx* ctx, const void* unique_id128) {
    int rc = need(ctx, "adm_comm_init_aux", true);
    if (rc) return rc;
    if (!unique_id128) return fail(ADM_ERR_INVALID, "adm_comm_init_aux: null argument");
    if (ctx->comm_aux) return fail(ADM_ERR_INVALID, "adm_comm_init_aux: this context already has a side-stream communicator");
    ADM_HIP(hipSetDevice(ctx->device));
    UniqueId id;
    std::memcpy(id.internal, unique_id128, sizeof(id.internal));
    Comm c = nullptr;
    rc = api()->CommInitRank(&c, ctx->comm_size, id, ctx->comm_rank);
    if (rc) return nccl_fail(rc, "ncclCommInitRank (side stream)");
    ctx->comm_aux = c;
    return ADM_OK;
}

// 0 if librccl and every entry point libadm uses could be resolved in THIS process (no communicator needed, no GPU work):
// lets every rank test its own installation before any rank enters a collective rendezvous.
extern "C" int adm_comm_available(void) {
    Api* a = api();
    if (!a->err.empty()) return fail(ADM_ERR_UNSUPPORTED, "adm_comm_available: " + a->err);
    return ADM_OK;
}

extern "C" int adm_comm_destroy(adm_ctx* ctx) {
    if (!ctx || (!ctx->comm && !ctx->comm_aux)) return ADM_OK;
    // a deferred all-gather may still be in flight on the side stream (e.g. when the caller unwinds after an exception)
    (void)hipStreamSynchronize(ctx->aux_stream);
    (void)hipStreamSynchronize(ctx->main_stream);
    int rc = 0;
    if (ctx->comm_aux) rc = api()->CommDestroy(ctx->comm_aux);
    ctx->comm_aux = nullptr;
    if (ctx->comm) {
        const int rc2 = api()->CommDestroy(ctx->comm);
        if (!rc) rc = rc2;
    }
    ctx->comm = nullptr;
    return rc ? nccl_fail(rc, "ncclCommDestroy") : ADM_OK;
}

extern "C" int adm_comm_rank(adm_ctx* ctx) { return (ctx && ctx->comm) ? ctx->comm_rank : 0; }
extern "C" int adm_comm_size(adm_ctx* ctx) { return (ctx && ctx->comm) ? ctx->comm_size : 1; }

extern "C" int adm_reduce_scatter(adm_ctx* ctx, const float* send, float* recv, size_t recv_count) {
    int rc = need(ctx, "adm_reduce_scatter", true);
    if (rc) return rc;
    if (!send || !recv) return fail(ADM_ERR_INVALID, "adm_reduce_scatter: null argument");
    rc = api()->ReduceScatter(send, recv, recv_count, kFloat32, kSum, comm_for(ctx), ctx->stream);
    return rc ? nccl_fail(rc, "ncclReduceScatter") : ADM_OK;
}

extern "C" int adm_all_gather(adm_ctx* ctx, const float* send, float* recv, size_t send_count) {
    int rc = need(ctx, "adm_all_gather", true);
    if (rc) return rc;
    if (!send || !recv) return fail(ADM_ERR_INVALID, "adm_all_gather: null argument");
    rc = api()->AllGather(send, recv, send_count, kFloat32, comm_for(ctx), ctx->stream);
    return rc ? nccl_fail(rc, "ncclAllGather") : ADM_OK;
}

extern "C" int adm_all_reduce(adm_ctx* ctx, float* buf, size_t count, int op_max) {
    int rc = need(ctx, "adm_all_reduce", true);
    if (rc) return rc;
    if (!buf) return fail(ADM_ERR_INVALID, "adm_all_reduce: null argument");
    rc = api()->AllReduce(buf, buf, count, kFloat32, op_max ? kMax : kSum, comm_for(ctx), ctx->stream);
    return rc ? nccl_fail(rc, "ncclAllReduce") : ADM_OK;
}

extern "C" int adm_broadcast(adm_ctx* ctx, void* buf, size_t bytes, int root) {
    int rc = need(ctx, "adm_broadcast", true);
    if (rc) return rc;
    if (!buf) return fail(ADM_ERR_INVALID, "adm_broadcast: null argument");
    rc = api()->Broadcast(buf, buf, bytes, kUint8, root, comm_for(ctx), ctx->stream);
    return rc ? nccl_fail(rc, "ncclBroadcast") : ADM_OK;
}

// buf[0 .. count) of rank `root` = sum over ranks of their buf[0 .. count), in place (the other ranks' buffers are left as they are)
extern "C" int adm_reduce(adm_ctx* ctx, float* buf, size_t count, int root) {
    int rc = need(ctx, "adm_reduce", true);
    if (rc) return rc;
    if (!buf) return fail(ADM_ERR_INVALID, "adm_reduce: null argument");
    rc = api()->Reduce(buf, buf, count, kFloat32, kSum, root, comm_for(ctx), ctx->stream);
    return rc ? nccl_fail(rc, "ncclReduce") : ADM_OK;
}

// Several collectives of one communicator issued between the two calls are launched as ONE operation (ncclGroupStart /
// ncclGroupEnd): the broadcasts of the object planes the next minibatches read, one per owning rank, run side by side
// instead of one root at a time.
extern "C" int adm_comm_group_start(adm_ctx* ctx) {
    int rc = need(ctx, "adm_comm_group_start", true);
    if (rc) return rc;
    rc = api()->GroupStart();
    return rc ? nccl_fail(rc, "ncclGroupStart") : ADM_OK;
}

extern "C" int adm_comm_group_end(adm_ctx* ctx) {
    int rc = need(ctx, "adm_comm_group_end", true);
    if (rc) return rc;
    rc = api()->GroupEnd();
    return rc ? nccl_fail(rc, "ncclGroupEnd") : ADM_OK;
}
