// Peer-to-peer transport of the data-parallel mode: a DIRECT all-pairs exchange through IPC-mapped peer buffers, no RCCL.
//
// The reference sums the full object gradient over the ranks and lets every rank apply the identical optimiser step:
//     gradient.arr = comm.allreduce(gradient.arr)                   adorym/ptychography.py:1113-1114
//     obj.arr = opt.apply_gradient(obj.arr, gradient, i, **opts)    adorym/ptychography.py:1120-1129, optimizers.py:309-318
//     constraints, support mask                                     adorym/ptychography.py:1135-1158, array_ops.py:239-251
// The RCCL path (adm_comm.hip) does that as reduce-scatter -> optimiser on the owned shard -> all-gather: three passes, of which
// the two collectives are rings (per-link bound on xGMI: 7 x 16.8 MB per phase at 8 ranks).  Here every rank maps the gradient
// and object buffers of all its peers (hipIpcGetMemHandle / hipIpcOpenMemHandle; xGMI is a full mesh, so every peer is one
// hop away) and ONE kernel per update, on rank r, for the elements of shard r:
//     g = g_0[i] + g_1[i] + ... + g_{R-1}[i]        read over R - 1 links at once, summed in RANK ORDER (deterministic, and
//                                                    equal bit for bit to a serial sum of the ranks' buffers)
//     Adam / GD / momentum + constraints + mask      in registers (adm_optim.h: the arithmetic of the one-rank kernels)
//     x_q[i] = x_new  for every rank q               written over R - 1 links at once
// i.e. reduce-scatter, optimiser and all-gather in one pass without the 134 MB intermediate; per link and direction only
// 1/R of the buffer travels (16.8 MB at 8 ranks).
//
// Ordering between the ranks needs no host: every rank owns a block of 64-bit flags in UNCACHED device memory which its peers
// write (system-scope release stores by a tiny signal kernel, after a system-scope release of the stream's earlier work) and
// which it polls itself (a tiny wait kernel in stream order: local reads, bounded by a timeout).  Every collective raises
// two flags: READY (my buffers may be read) and DONE (I have finished reading / writing yours); both ends are full barriers of
// the ranks' streams, so whatever a rank queues after adm_p2p_update sees the updated object and may overwrite its gradient.
// A wait that times out (a peer died or left the sequence) sets a sticky error: the kernels that follow return at once, and
// adm_p2p_status reports which rank did not arrive.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include <string>
#include "adm_common.h"
#include "adm_optim.h"

namespace {

constexpr int kMaxRanks = ADM_P2P_MAX_RANKS;
constexpr int kSlots = 8;
enum Slot { kReady = 0, kDone = 1, kMailReady = 2, kMailDone = 3, kBarrier = 4 };
constexpr int kErrWord = kSlots * kMaxRanks;                 // sticky error flag of the rank
constexpr size_t kFlagBytes = 4096;

struct PeerTable {
    float* x[kMaxRanks];
    const float* g[kMaxRanks];
};
struct FlagTable { unsigned long long* f[kMaxRanks]; };
struct MailTable { const float* box[kMaxRanks]; };

struct P2P {
    int rank = 0, size = 1;
    unsigned long long* flags = nullptr;          // own block: [kSlots][kMaxRanks] + error word
    bool flags_uncached = false;
    unsigned char* mailbox = nullptr;
    size_t mailbox_bytes = 0;
    FlagTable peer_flags{};
    MailTable peer_mail{};
    PeerTable obj{};
    size_t n_obj = 0;
    unsigned long long epoch[kSlots] = {};
    hipEvent_t fence = nullptr;                   // default flags: recording it releases the stream's work at system scope
    int* status = nullptr;                        // pinned host word written by the wait kernel: 0, or 1 + slot * kMaxRanks + peer
    unsigned long long timeout_ticks = 0;         // of the 100 MHz wall clock
    bool connected = false, bound = false;
};

P2P* state(adm_ctx* ctx) { return ctx ? static_cast<P2P*>(ctx->p2p) : nullptr; }

// ---- ordering kernels ---------------------------------------------------------------------------------------------------
__global__ void p2p_signal_kernel(FlagTable t, int R, int me, int slot, unsigned long long value) {
    if ((int)threadIdx.x < R) {
        __threadfence_system();
        __hip_atomic_store(t.f[threadIdx.x] + slot * kMaxRanks + me, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ void p2p_wait_kernel(unsigned long long* own, int R, int slot, unsigned long long value, unsigned long long timeout_ticks,
                                int* host_status) {
    const int q = threadIdx.x;
    if (q >= R) return;
    if (__hip_atomic_load(own + kErrWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) return;      // an earlier wait failed: do not spin again
    const unsigned long long t0 = wall_clock64();
    const unsigned long long* f = own + slot * kMaxRanks + q;
    while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < value) {
        if (wall_clock64() - t0 > timeout_ticks) {
            __hip_atomic_store(own + kErrWord, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            *host_status = 1 + slot * kMaxRanks + q;
            return;
        }
        __builtin_amdgcn_s_sleep(4);
    }
}

// ---- the fused exchange -------------------------------------------------------------------------------------------------
template <int KIND>
__device__ __forceinline__ float step_one(float xv, float gv, float* m, float* v, size_t k, size_t i, const adm::AdamScalars& a, float gamma) {
    if (KIND == ADM_OPT_ADAM) {
        float mo, vo;
        const float xn = adm::adam_value(xv, gv, m[k], v[k], a, i, mo, vo);
        m[k] = mo;
        v[k] = vo;
        return xn;
    } else if (KIND == ADM_OPT_MOMENTUM) {
        float vo;
        const float xn = adm::momentum_value(xv, gv, m[k], a.step, gamma, i, a.flags, a.mask, vo);
        m[k] = vo;
        return xn;
    }
    return adm::gd_value(xv, gv, a.step, i, a.flags, a.mask);
}

// the ranks' gradients at element i, added in rank order where i lies in [s_lo, s_hi) (the part of the buffers that holds every
// rank's data term); elsewhere the owner's buffer is complete by itself (footprint-restricted exchange, adorym_amd/dp.py)
__device__ __forceinline__ float grad_sum(const PeerTable& t, int R, int me, size_t i, size_t s_lo, size_t s_hi) {
    if (i < s_lo || i >= s_hi) return t.g[me][i];
    float acc = t.g[0][i];
    for (int q = 1; q < R; ++q) acc = acc + t.g[q][i];
    return acc;
}

template <int KIND, int W>
__global__ __launch_bounds__(256) void p2p_update_kernel(PeerTable t, int R, int me, float* __restrict__ m, float* __restrict__ v, size_t lo,
                                                         size_t hi, size_t s_lo, size_t s_hi, adm::AdamScalars a, float gamma,
                                                         const unsigned long long* err) {
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) return;       // a peer never arrived: touch nothing
    // (No fence in here: a per-wave system-scope acquire / release -- buffer_inv / buffer_wbl2 sc0 sc1 -- made this kernel 6x slower,
    // 1.08 ms instead of 0.17 ms for a 134 MB shard.  Visibility across the ranks rests on the kernel boundaries: the producers'
    // work is released at system scope by the event record in signal() before the READY flag, this launch starts after the wait
    // kernel has seen every flag, and its own writes are released the same way before DONE.)
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (size_t)gridDim.x * blockDim.x;
    const size_t base = lo;          // the moments are the rank's shard: element i lives at i - base
    if (W == 4) {
        // lo is a multiple of 4 (the launcher checks): 16-byte accesses of every buffer, moments included
        const size_t nvec = (hi - lo) >> 2;
        for (size_t j = tid; j < nvec; j += nthreads) {
            const size_t i = lo + 4 * j;
            float4 gs;
            if (i >= s_lo && i + 4 <= s_hi) {
                gs = *reinterpret_cast<const float4*>(t.g[0] + i);
                for (int q = 1; q < R; ++q) {
                    const float4 gq = *reinterpret_cast<const float4*>(t.g[q] + i);
                    gs.x = gs.x + gq.x; gs.y = gs.y + gq.y; gs.z = gs.z + gq.z; gs.w = gs.w + gq.w;
                }
            } else if (i + 4 <= s_lo || i >= s_hi) {
                gs = *reinterpret_cast<const float4*>(t.g[me] + i);
            } else {
                gs.x = grad_sum(t, R, me, i, s_lo, s_hi); gs.y = grad_sum(t, R, me, i + 1, s_lo, s_hi);
                gs.z = grad_sum(t, R, me, i + 2, s_lo, s_hi); gs.w = grad_sum(t, R, me, i + 3, s_lo, s_hi);
            }
            const float4 xo = *reinterpret_cast<const float4*>(t.x[me] + i);
            float4 xn;
            if (KIND == ADM_OPT_GD) {
                xn.x = adm::gd_value(xo.x, gs.x, a.step, i, a.flags, a.mask); xn.y = adm::gd_value(xo.y, gs.y, a.step, i + 1, a.flags, a.mask);
                xn.z = adm::gd_value(xo.z, gs.z, a.step, i + 2, a.flags, a.mask); xn.w = adm::gd_value(xo.w, gs.w, a.step, i + 3, a.flags, a.mask);
            } else if (KIND == ADM_OPT_MOMENTUM) {
                float4 vo = *reinterpret_cast<const float4*>(m + (i - base));
                xn.x = adm::momentum_value(xo.x, gs.x, vo.x, a.step, gamma, i, a.flags, a.mask, vo.x);
                xn.y = adm::momentum_value(xo.y, gs.y, vo.y, a.step, gamma, i + 1, a.flags, a.mask, vo.y);
                xn.z = adm::momentum_value(xo.z, gs.z, vo.z, a.step, gamma, i + 2, a.flags, a.mask, vo.z);
                xn.w = adm::momentum_value(xo.w, gs.w, vo.w, a.step, gamma, i + 3, a.flags, a.mask, vo.w);
                *reinterpret_cast<float4*>(m + (i - base)) = vo;
            } else {
                float4 mo = *reinterpret_cast<const float4*>(m + (i - base)), vo = *reinterpret_cast<const float4*>(v + (i - base));
                xn.x = adm::adam_value(xo.x, gs.x, mo.x, vo.x, a, i, mo.x, vo.x);
                xn.y = adm::adam_value(xo.y, gs.y, mo.y, vo.y, a, i + 1, mo.y, vo.y);
                xn.z = adm::adam_value(xo.z, gs.z, mo.z, vo.z, a, i + 2, mo.z, vo.z);
                xn.w = adm::adam_value(xo.w, gs.w, mo.w, vo.w, a, i + 3, mo.w, vo.w);
                *reinterpret_cast<float4*>(m + (i - base)) = mo;
                *reinterpret_cast<float4*>(v + (i - base)) = vo;
            }
            for (int q = 0; q < R; ++q) *reinterpret_cast<float4*>(t.x[q] + i) = xn;
        }
        lo += nvec << 2;            // the (< 4) elements left over go through the scalar loop below
    }
    for (size_t i = lo + tid; i < hi; i += nthreads) {
        const float xn = step_one<KIND>(t.x[me][i], grad_sum(t, R, me, i, s_lo, s_hi), m, v, i - base, i, a, gamma);
        for (int q = 0; q < R; ++q) t.x[q][i] = xn;
    }
}

// out[i] = box_0[i] + box_1[i] + ... in rank order (small parameter gradients: optimizers.py:1025,1041,1053,1064,1079)
__global__ __launch_bounds__(256) void p2p_sum_kernel(MailTable t, int R, float* __restrict__ out, size_t n, const unsigned long long* err) {
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) return;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float acc = t.box[0][i];
        for (int q = 1; q < R; ++q) acc = acc + t.box[q][i];
        out[i] = acc;
    }
}

int stream_grid(size_t n) {
    const size_t b = (n + 255) / 256;
    return (int)(b > 4096 ? 4096 : (b ? b : 1));
}

// everything this rank has queued so far becomes visible at system scope, then every peer's flag `slot` of this rank is raised
int signal(adm_ctx* ctx, P2P* p, int slot) {
    ADM_HIP(hipEventRecord(p->fence, ctx->stream));
    ++p->epoch[slot];
    hipLaunchKernelGGL(p2p_signal_kernel, dim3(1), dim3(64), 0, ctx->stream, p->peer_flags, p->size, p->rank, slot, p->epoch[slot]);
    ADM_HIP(hipGetLastError());
    return ADM_OK;
}
int wait(adm_ctx* ctx, P2P* p, int slot) {
    hipLaunchKernelGGL(p2p_wait_kernel, dim3(1), dim3(64), 0, ctx->stream, p->flags, p->size, slot, p->epoch[slot], p->timeout_ticks, p->status);
    ADM_HIP(hipGetLastError());
    return ADM_OK;
}

int need(adm_ctx* ctx, const char* what, bool connected) {
    P2P* p = state(ctx);
    if (!p) return adm::fail(ADM_ERR_INVALID, std::string(what) + ": adm_p2p_create has not been called on this context");
    if (connected && !p->connected) return adm::fail(ADM_ERR_INVALID, std::string(what) + ": adm_p2p_connect has not been called");
    return ADM_OK;
}

}  // namespace

using adm::fail;

extern "C" int adm_p2p_create(adm_ctx* ctx, int rank, int nranks, size_t mailbox_bytes) {
    if (!ctx) return fail(ADM_ERR_INVALID, "adm_p2p_create: null context");
    if (ctx->p2p) return fail(ADM_ERR_INVALID, "adm_p2p_create: this context already has a peer-to-peer group");
    if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return fail(ADM_ERR_INVALID, "adm_p2p_create: bad rank / nranks (at most ADM_P2P_MAX_RANKS)");
    ADM_HIP(hipSetDevice(ctx->device));
    P2P* p = new P2P();
    p->rank = rank;
    p->size = nranks;
    // flags: uncached ("fine-grained") device memory, so that a peer's store is seen by a kernel that is already polling
    hipError_t e = hipExtMallocWithFlags((void**)&p->flags, kFlagBytes, hipDeviceMallocUncached);
    p->flags_uncached = e == hipSuccess;
    if (e != hipSuccess) {
        (void)hipGetLastError();
        e = hipMalloc((void**)&p->flags, kFlagBytes);
    }
    if (e == hipSuccess) e = hipMemset(p->flags, 0, kFlagBytes);
    p->mailbox_bytes = (mailbox_bytes ? mailbox_bytes : (size_t)8 << 20) & ~(size_t)15;
    if (e == hipSuccess) e = hipMalloc((void**)&p->mailbox, p->mailbox_bytes);
    if (e == hipSuccess) e = hipEventCreate(&p->fence);
    if (e == hipSuccess) e = hipHostMalloc((void**)&p->status, 64, hipHostMallocDefault);
    if (e != hipSuccess) {
        if (p->flags) (void)hipFree(p->flags);
        if (p->mailbox) (void)hipFree(p->mailbox);
        if (p->fence) (void)hipEventDestroy(p->fence);
        delete p;
        return adm::hip_fail(e, "adm_p2p_create");
    }
    *p->status = 0;
    const char* ts = std::getenv("ADM_P2P_TIMEOUT_S");
    double sec = ts ? std::atof(ts) : 30.0;
    if (!(sec > 0.0)) sec = 30.0;
    p->timeout_ticks = (unsigned long long)(sec * 1e8);
    p->peer_flags.f[rank] = p->flags;
    p->peer_mail.box[rank] = reinterpret_cast<const float*>(p->mailbox);
    p->connected = nranks == 1;
    ctx->p2p = p;
    return ADM_OK;
}

extern "C" int adm_p2p_local(adm_ctx* ctx, int which, void** dptr) {
    int rc = need(ctx, "adm_p2p_local", false);
    if (rc) return rc;
    if (!dptr || which < 0 || which > 1) return fail(ADM_ERR_INVALID, "adm_p2p_local: bad argument");
    P2P* p = state(ctx);
    *dptr = which == 0 ? (void*)p->flags : (void*)p->mailbox;
    return ADM_OK;
}

extern "C" int adm_p2p_export(adm_ctx* ctx, const void* dptr, void* handle64) {
    if (!ctx || !dptr || !handle64) return fail(ADM_ERR_INVALID, "adm_p2p_export: null argument");
    static_assert(sizeof(hipIpcMemHandle_t) == ADM_P2P_HANDLE_BYTES, "IPC handle size");
    ADM_HIP(hipSetDevice(ctx->device));
    hipIpcMemHandle_t h;
    ADM_HIP(hipIpcGetMemHandle(&h, const_cast<void*>(dptr)));
    std::memcpy(handle64, &h, sizeof h);
    return ADM_OK;
}

extern "C" int adm_p2p_open(adm_ctx* ctx, const void* handle64, void** dptr) {
    if (!ctx || !handle64 || !dptr) return fail(ADM_ERR_INVALID, "adm_p2p_open: null argument");
    ADM_HIP(hipSetDevice(ctx->device));
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle64, sizeof h);
    void* p = nullptr;
    ADM_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
    // The mapping must be usable by this GPU's KERNELS: a pointer they cannot reach is a memory fault that takes the process
    // down, not an error code.  So before anybody launches on it: if the allocation lives on another device, that device must
    // be a peer this one can access (and access is switched on explicitly, whatever the lazy flag did), and a 4-byte copy
    // through the mapping must succeed.
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) == hipSuccess && attr.device != ctx->device) {
        int can = 0;
        hipError_t e = hipDeviceCanAccessPeer(&can, ctx->device, attr.device);
        if (e != hipSuccess || !can) {
            (void)hipGetLastError();
            (void)hipIpcCloseMemHandle(p);
            return fail(ADM_ERR_UNSUPPORTED, "adm_p2p_open: device " + std::to_string(ctx->device) + " cannot access device " +
                                                 std::to_string(attr.device) + " as a peer");
        }
        e = hipDeviceEnablePeerAccess(attr.device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
            (void)hipGetLastError();
            (void)hipIpcCloseMemHandle(p);
            return adm::hip_fail(e, "hipDeviceEnablePeerAccess");
        }
        (void)hipGetLastError();
    } else {
        (void)hipGetLastError();
    }
    unsigned int probe = 0;
    hipError_t e = hipMemcpy(&probe, p, sizeof probe, hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        (void)hipIpcCloseMemHandle(p);
        return adm::hip_fail(e, "adm_p2p_open: reading through the mapping");
    }
    *dptr = p;
    return ADM_OK;
}

extern "C" int adm_p2p_close(adm_ctx* ctx, void* dptr) {
    if (!ctx) return fail(ADM_ERR_INVALID, "adm_p2p_close: null context");
    if (dptr) ADM_HIP(hipIpcCloseMemHandle(dptr));
    return ADM_OK;
}

extern "C" int adm_p2p_connect(adm_ctx* ctx, void* const* peer_flags, void* const* peer_mailbox) {
    int rc = need(ctx, "adm_p2p_connect", false);
    if (rc) return rc;
    if (!peer_flags || !peer_mailbox) return fail(ADM_ERR_INVALID, "adm_p2p_connect: null argument");
    P2P* p = state(ctx);
    for (int q = 0; q < p->size; ++q) {
        if (q == p->rank) continue;
        if (!peer_flags[q] || !peer_mailbox[q]) return fail(ADM_ERR_INVALID, "adm_p2p_connect: null peer pointer");
        p->peer_flags.f[q] = static_cast<unsigned long long*>(peer_flags[q]);
        p->peer_mail.box[q] = static_cast<const float*>(peer_mailbox[q]);
    }
    p->connected = true;
    return ADM_OK;
}

extern "C" int adm_p2p_bind_object(adm_ctx* ctx, void* const* peer_x, void* const* peer_g, size_t n) {
    int rc = need(ctx, "adm_p2p_bind_object", true);
    if (rc) return rc;
    if (!peer_x || !peer_g) return fail(ADM_ERR_INVALID, "adm_p2p_bind_object: null argument");
    P2P* p = state(ctx);
    for (int q = 0; q < p->size; ++q) {
        if (!peer_x[q] || !peer_g[q]) return fail(ADM_ERR_INVALID, "adm_p2p_bind_object: null peer pointer");
        p->obj.x[q] = static_cast<float*>(peer_x[q]);
        p->obj.g[q] = static_cast<const float*>(peer_g[q]);
    }
    p->n_obj = n;
    p->bound = true;
    return ADM_OK;
}

extern "C" int adm_p2p_rank(adm_ctx* ctx) { return state(ctx) ? state(ctx)->rank : 0; }
extern "C" int adm_p2p_size(adm_ctx* ctx) { return state(ctx) ? state(ctx)->size : 1; }

extern "C" int adm_p2p_barrier(adm_ctx* ctx) {
    int rc = need(ctx, "adm_p2p_barrier", true);
    if (rc) return rc;
    P2P* p = state(ctx);
    if (p->size == 1) return ADM_OK;
    rc = signal(ctx, p, kBarrier);
    return rc ? rc : wait(ctx, p, kBarrier);
}

extern "C" int adm_p2p_update(adm_ctx* ctx, int kind, float* m, float* v, size_t lo, size_t hi, size_t sum_lo, size_t sum_hi, int i_batch,
                              double step_size, double b1, double b2, double eps, int flags, const float* mask) {
    int rc = need(ctx, "adm_p2p_update", true);
    if (rc) return rc;
    P2P* p = state(ctx);
    if (!p->bound) return fail(ADM_ERR_INVALID, "adm_p2p_update: adm_p2p_bind_object has not been called");
    if (hi < lo || hi > p->n_obj) return fail(ADM_ERR_INVALID, "adm_p2p_update: bad element range");
    if ((kind == ADM_OPT_ADAM && (!m || !v)) || (kind == ADM_OPT_MOMENTUM && !m)) return fail(ADM_ERR_INVALID, "adm_p2p_update: null moment buffer");
    if (kind != ADM_OPT_ADAM && kind != ADM_OPT_GD && kind != ADM_OPT_MOMENTUM) return fail(ADM_ERR_INVALID, "adm_p2p_update: unknown optimiser kind");
    const bool many = p->size > 1;
    if (many) {
        if ((rc = signal(ctx, p, kReady))) return rc;        // my gradient buffer is complete
        if ((rc = wait(ctx, p, kReady))) return rc;          // ... and so is everybody's
    }
    if (hi > lo) {
        const adm::AdamScalars a = kind == ADM_OPT_ADAM ? adm::adam_scalars(i_batch, step_size, b1, b2, eps, flags, mask)
                                                        : adm::adam_scalars(0, step_size, 0.0, 0.0, 0.0, flags, mask);
        const float gamma = (float)b1;
        const unsigned long long* err = p->flags + kErrWord;
        const bool vec = (lo & 3) == 0;
        const dim3 grid(stream_grid(vec ? (hi - lo + 3) / 4 : hi - lo)), block(256);
#define ADM_P2P_LAUNCH(KIND, W) \
    hipLaunchKernelGGL((p2p_update_kernel<KIND, W>), grid, block, 0, ctx->stream, p->obj, p->size, p->rank, m, v, lo, hi, sum_lo, sum_hi, a, gamma, err)
        if (kind == ADM_OPT_ADAM) { if (vec) ADM_P2P_LAUNCH(ADM_OPT_ADAM, 4); else ADM_P2P_LAUNCH(ADM_OPT_ADAM, 1); }
        else if (kind == ADM_OPT_GD) { if (vec) ADM_P2P_LAUNCH(ADM_OPT_GD, 4); else ADM_P2P_LAUNCH(ADM_OPT_GD, 1); }
        else { if (vec) ADM_P2P_LAUNCH(ADM_OPT_MOMENTUM, 4); else ADM_P2P_LAUNCH(ADM_OPT_MOMENTUM, 1); }
#undef ADM_P2P_LAUNCH
        ADM_HIP(hipGetLastError());
    }
    if (many) {
        if ((rc = signal(ctx, p, kDone))) return rc;         // I have read your gradients and written your objects
        if ((rc = wait(ctx, p, kDone))) return rc;           // ... and everybody has done so with mine
    }
    return ADM_OK;
}

extern "C" int adm_p2p_all_reduce(adm_ctx* ctx, float* buf, size_t count) {
    int rc = need(ctx, "adm_p2p_all_reduce", true);
    if (rc) return rc;
    if (!buf) return fail(ADM_ERR_INVALID, "adm_p2p_all_reduce: null argument");
    P2P* p = state(ctx);
    if (p->size == 1 || count == 0) return ADM_OK;
    const size_t chunk = p->mailbox_bytes / sizeof(float);
    for (size_t off = 0; off < count; off += chunk) {
        const size_t n = count - off < chunk ? count - off : chunk;
        ADM_HIP(hipMemcpyAsync(p->mailbox, buf + off, n * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
        if ((rc = signal(ctx, p, kMailReady))) return rc;
        if ((rc = wait(ctx, p, kMailReady))) return rc;
        hipLaunchKernelGGL(p2p_sum_kernel, dim3(stream_grid(n)), dim3(256), 0, ctx->stream, p->peer_mail, p->size, buf + off, n, p->flags + kErrWord);
        ADM_HIP(hipGetLastError());
        if ((rc = signal(ctx, p, kMailDone))) return rc;     // the mailboxes may be refilled once everybody has read them
        if ((rc = wait(ctx, p, kMailDone))) return rc;
    }
    return ADM_OK;
}

extern "C" int adm_p2p_status(adm_ctx* ctx) {
    P2P* p = state(ctx);
    if (!p) return ADM_OK;
    const int s = *reinterpret_cast<volatile int*>(p->status);
    if (s == 0) return ADM_OK;
    static const char* names[kSlots] = {"gradients ready", "update done", "mailbox ready", "mailbox read", "barrier", "?", "?", "?"};
    const int slot = (s - 1) / kMaxRanks, peer = (s - 1) % kMaxRanks;
    return fail(ADM_ERR_HIP, "peer-to-peer exchange: rank " + std::to_string(peer) + " did not signal '" + names[slot] + "' to rank " +
                                 std::to_string(p->rank) + " within " + std::to_string((double)p->timeout_ticks * 1e-8) +
                                 " s (ADM_P2P_TIMEOUT_S); the update kernels after it were skipped");
}

extern "C" int adm_p2p_destroy(adm_ctx* ctx) {
    P2P* p = state(ctx);
    if (!p) return ADM_OK;
    (void)hipStreamSynchronize(ctx->aux_stream);
    (void)hipStreamSynchronize(ctx->main_stream);
    (void)hipFree(p->flags);
    (void)hipFree(p->mailbox);
    (void)hipEventDestroy(p->fence);
    (void)hipHostFree(p->status);
    delete p;
    ctx->p2p = nullptr;
    return ADM_OK;
}
