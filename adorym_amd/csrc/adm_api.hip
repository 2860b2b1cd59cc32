// C-ABI entry points: context, memory, events, plan, multislice launcher.
#include <hip/hip_runtime.h>
#include <cmath>
#include <vector>
#include <cstring>
#include <csignal>
#include <unistd.h>
#include "adm_common.h"

namespace adm {
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
int hip_fail(hipError_t e, const char* what) {
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    (void)hipGetLastError();
    return ADM_ERR_HIP;
}
}  // namespace adm
using namespace adm;

extern "C" int adm_version(void) { return ADM_VERSION; }
extern "C" const char* adm_last_error(void) { return g_err.c_str(); }

// A line to leave on stdout if the process is killed by SIGABRT / SIGSEGV / SIGBUS (the ROCm runtime abort()s on a GPU memory
// fault): write(2) + _exit only, both async-signal-safe.  bench.py arms it around the secondary legs of a multi-rank run, whose
// transports have never seen several GPUs, so that a fault in one of them still leaves the measured headline line behind.
static char g_crash_line[1 << 17];
static volatile size_t g_crash_len = 0;
static volatile int g_crash_code = 1;
static bool g_crash_installed = false;
static void crash_handler(int sig) {
    const size_t n = g_crash_len;
    if (n) {
        size_t off = 0;
        while (off < n) {
            const ssize_t w = write(1, g_crash_line + off, n - off);
            if (w <= 0) break;
            off += (size_t)w;
        }
        _exit(g_crash_code);
    }
    signal(sig, SIG_DFL);
    raise(sig);
}
extern "C" int adm_crash_line_set(const char* line, int exit_code) {
    g_crash_len = 0;
    if (!line) return ADM_OK;
    const size_t n = std::strlen(line);
    if (n + 2 > sizeof(g_crash_line)) return fail(ADM_ERR_INVALID, "adm_crash_line_set: line too long");
    std::memcpy(g_crash_line, line, n);
    g_crash_line[n] = '\n';
    g_crash_code = exit_code;
    if (!g_crash_installed) {
        struct sigaction sa;
        std::memset(&sa, 0, sizeof(sa));
        sa.sa_handler = crash_handler;
        sigemptyset(&sa.sa_mask);
        sigaction(SIGABRT, &sa, nullptr);
        sigaction(SIGSEGV, &sa, nullptr);
        sigaction(SIGBUS, &sa, nullptr);
        g_crash_installed = true;
    }
    g_crash_len = n + 1;
    return ADM_OK;
}

extern "C" int adm_ctx_create(int device, void* stream, adm_ctx** out) {
    if (!out) return fail(ADM_ERR_INVALID, "adm_ctx_create: out is null");
    int n = 0;
    ADM_HIP(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(ADM_ERR_INVALID, "adm_ctx_create: no such device");
    ADM_HIP(hipSetDevice(device));
    adm_ctx* c = new adm_ctx();
    c->device = device;
    if (stream) {
        c->stream = (hipStream_t)stream;
        c->owns_stream = false;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete c;
            return hip_fail(e, "hipStreamCreate");
        }
        c->owns_stream = true;
    }
    c->main_stream = c->stream;
    c->join_pending = false;
    c->comm = nullptr;
    c->comm_aux = nullptr;
    c->comm_rank = 0;
    c->comm_size = 1;
    c->p2p = nullptr;
    hipError_t e2 = hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking);
    if (e2 == hipSuccess) e2 = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
    if (e2 == hipSuccess) e2 = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming);
    if (e2 != hipSuccess) {
        delete c;
        return hip_fail(e2, "aux stream / events");
    }
    *out = c;
    return ADM_OK;
}

extern "C" int adm_ctx_fork(adm_ctx* ctx) {
    if (!ctx) return fail(ADM_ERR_INVALID, "adm_ctx_fork: null ctx");
    if (ctx->stream != ctx->main_stream) return fail(ADM_ERR_INVALID, "adm_ctx_fork: already forked");
    ADM_HIP(hipEventRecord(ctx->ev_fork, ctx->main_stream));
    ADM_HIP(hipStreamWaitEvent(ctx->aux_stream, ctx->ev_fork, 0));
    ctx->stream = ctx->aux_stream;
    return ADM_OK;
}
extern "C" int adm_ctx_end_fork(adm_ctx* ctx) {
    if (!ctx) return fail(ADM_ERR_INVALID, "adm_ctx_end_fork: null ctx");
    if (ctx->stream != ctx->aux_stream) return fail(ADM_ERR_INVALID, "adm_ctx_end_fork: not forked");
    ADM_HIP(hipEventRecord(ctx->ev_join, ctx->aux_stream));
    ctx->stream = ctx->main_stream;
    ctx->join_pending = true;
    return ADM_OK;
}
extern "C" int adm_ctx_join(adm_ctx* ctx) {
    if (!ctx) return fail(ADM_ERR_INVALID, "adm_ctx_join: null ctx");
    if (ctx->stream != ctx->main_stream) return fail(ADM_ERR_INVALID, "adm_ctx_join: still forked");
    if (ctx->join_pending) {
        ADM_HIP(hipStreamWaitEvent(ctx->main_stream, ctx->ev_join, 0));
        ctx->join_pending = false;
    }
    return ADM_OK;
}

extern "C" int adm_ctx_destroy(adm_ctx* ctx) {
    if (!ctx) return ADM_OK;
    (void)hipStreamSynchronize(ctx->aux_stream);
    (void)hipStreamSynchronize(ctx->main_stream);
    (void)adm_comm_destroy(ctx);
    (void)adm_p2p_destroy(ctx);
    (void)hipStreamDestroy(ctx->aux_stream);
    (void)hipEventDestroy(ctx->ev_fork);
    (void)hipEventDestroy(ctx->ev_join);
    if (ctx->owns_stream) (void)hipStreamDestroy(ctx->main_stream);
    delete ctx;
    return ADM_OK;
}
extern "C" int adm_ctx_sync(adm_ctx* ctx) {
    if (!ctx) return fail(ADM_ERR_INVALID, "adm_ctx_sync: null ctx");
    ADM_HIP(hipStreamSynchronize(ctx->aux_stream));
    ADM_HIP(hipStreamSynchronize(ctx->main_stream));
    return ADM_OK;
}
extern "C" void* adm_ctx_stream(adm_ctx* ctx) { return ctx ? (void*)ctx->main_stream : nullptr; }
extern "C" int adm_ctx_device(adm_ctx* ctx) { return ctx ? ctx->device : -1; }
extern "C" int adm_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

extern "C" int adm_mem_info(adm_ctx* ctx, size_t* free_bytes, size_t* total_bytes) {
    if (!ctx || !free_bytes || !total_bytes) return fail(ADM_ERR_INVALID, "adm_mem_info: null argument");
    ADM_HIP(hipSetDevice(ctx->device));
    ADM_HIP(hipMemGetInfo(free_bytes, total_bytes));
    return ADM_OK;
}

extern "C" int adm_malloc(adm_ctx* ctx, size_t bytes, void** dptr) {
    if (!ctx || !dptr) return fail(ADM_ERR_INVALID, "adm_malloc: null argument");
    ADM_HIP(hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 4);
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        return fail(ADM_ERR_NOMEM, "adm_malloc: out of device memory");
    }
    ADM_HIP(e);
    return ADM_OK;
}
extern "C" int adm_free(adm_ctx* ctx, void* dptr) {
    if (!ctx) return fail(ADM_ERR_INVALID, "adm_free: null ctx");
    if (dptr) {
        ADM_HIP(hipStreamSynchronize(ctx->stream));
        ADM_HIP(hipFree(dptr));
    }
    return ADM_OK;
}
extern "C" int adm_memset(adm_ctx* ctx, void* dptr, int value, size_t bytes) {
    if (!ctx || !dptr) return fail(ADM_ERR_INVALID, "adm_memset: null argument");
    if (bytes) ADM_HIP(hipMemsetAsync(dptr, value, bytes, ctx->stream));
    return ADM_OK;
}
extern "C" int adm_h2d(adm_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (!ctx || !dst || !src) return fail(ADM_ERR_INVALID, "adm_h2d: null argument");
    if (bytes) {
        ADM_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        ADM_HIP(hipStreamSynchronize(ctx->stream));
    }
    return ADM_OK;
}
extern "C" int adm_d2h(adm_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (!ctx || !dst || !src) return fail(ADM_ERR_INVALID, "adm_d2h: null argument");
    if (bytes) {
        ADM_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        ADM_HIP(hipStreamSynchronize(ctx->stream));
    }
    return ADM_OK;
}
extern "C" int adm_d2d(adm_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (!ctx || !dst || !src) return fail(ADM_ERR_INVALID, "adm_d2d: null argument");
    if (bytes) ADM_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return ADM_OK;
}

extern "C" int adm_host_alloc(adm_ctx* ctx, size_t bytes, void** hptr) {
    if (!ctx || !hptr) return fail(ADM_ERR_INVALID, "adm_host_alloc: null argument");
    if (hipHostMalloc(hptr, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return fail(ADM_ERR_NOMEM, "adm_host_alloc: hipHostMalloc failed");
    return ADM_OK;
}
extern "C" int adm_host_free(adm_ctx* ctx, void* hptr) {
    if (hptr) ADM_HIP(hipHostFree(hptr));
    return ADM_OK;
}
extern "C" int adm_d2h_async(adm_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (!ctx || !dst || !src) return fail(ADM_ERR_INVALID, "adm_d2h_async: null argument");
    if (bytes) ADM_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return ADM_OK;
}
extern "C" int adm_h2d_async(adm_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (!ctx || !dst || !src) return fail(ADM_ERR_INVALID, "adm_h2d_async: null argument");
    if (bytes) ADM_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return ADM_OK;
}
extern "C" int adm_event_sync(adm_ctx* ctx, void* ev) {
    if (!ev) return fail(ADM_ERR_INVALID, "adm_event_sync: null argument");
    ADM_HIP(hipEventSynchronize((hipEvent_t)ev));
    return ADM_OK;
}

extern "C" int adm_event_create(adm_ctx* ctx, void** ev) {
    if (!ctx || !ev) return fail(ADM_ERR_INVALID, "adm_event_create: null argument");
    hipEvent_t e;
    ADM_HIP(hipEventCreate(&e));
    *ev = (void*)e;
    return ADM_OK;
}
extern "C" int adm_event_destroy(adm_ctx* ctx, void* ev) {
    if (ev) ADM_HIP(hipEventDestroy((hipEvent_t)ev));
    return ADM_OK;
}
extern "C" int adm_event_record(adm_ctx* ctx, void* ev) {
    if (!ctx || !ev) return fail(ADM_ERR_INVALID, "adm_event_record: null argument");
    ADM_HIP(hipEventRecord((hipEvent_t)ev, ctx->stream));
    return ADM_OK;
}
extern "C" int adm_event_elapsed_ms(adm_ctx* ctx, void* a, void* b, float* ms) {
    if (!a || !b || !ms) return fail(ADM_ERR_INVALID, "adm_event_elapsed_ms: null argument");
    ADM_HIP(hipEventSynchronize((hipEvent_t)b));
    ADM_HIP(hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b));
    return ADM_OK;
}

// ---------------------------------------------------------------------------------------- plan
static int upload_c(adm_ctx* ctx, const float* re, const float* im, size_t n, float2** out) {
    std::vector<float2> tmp(n);
    for (size_t i = 0; i < n; ++i) tmp[i] = make_float2(re[i], im[i]);
    void* d = nullptr;
    int rc = adm_malloc(ctx, n * sizeof(float2), &d);
    if (rc) return rc;
    rc = adm_h2d(ctx, d, tmp.data(), n * sizeof(float2));
    if (rc) return rc;
    *out = (float2*)d;
    return ADM_OK;
}

extern "C" int adm_plan_create(adm_ctx* ctx, const adm_plan_desc* desc, adm_plan** out) {
    if (!ctx || !desc || !out) return fail(ADM_ERR_INVALID, "adm_plan_create: null argument");
    const adm_plan_desc& d = *desc;
    if (d.obj_y <= 0 || d.obj_x <= 0 || d.obj_z <= 0 || d.probe_y <= 0 || d.probe_x <= 0)
        return fail(ADM_ERR_INVALID, "adm_plan_create: non-positive dimension");
    if (d.pad_y0 < 0 || d.pad_y1 < 0 || d.pad_x0 < 0 || d.pad_x1 < 0) return fail(ADM_ERR_INVALID, "adm_plan_create: negative pad");
    if (d.binning < 1) return fail(ADM_ERR_INVALID, "adm_plan_create: binning must be >= 1");
    if (d.sign_convention != 1 && d.sign_convention != -1) return fail(ADM_ERR_INVALID, "adm_plan_create: sign_convention must be +-1");
    if (d.det_mode < 0 || d.det_mode > 2) return fail(ADM_ERR_INVALID, "adm_plan_create: bad det_mode");
    if (d.loss_type != ADM_LOSS_LSQ && d.loss_type != ADM_LOSS_POISSON) return fail(ADM_ERR_INVALID, "adm_plan_create: bad loss_type");
    if (d.unknown_type != 0 && d.unknown_type != 1) return fail(ADM_ERR_INVALID, "adm_plan_create: bad unknown_type");
    if (d.unknown_type == 1 && d.binning != 1) return fail(ADM_ERR_UNSUPPORTED, "adm_plan_create: binning > 1 with unknown_type real_imag is not implemented");
    if (!d.h_re || !d.h_im) return fail(ADM_ERR_INVALID, "adm_plan_create: transfer function missing");
    if (d.det_mode == ADM_DET_FRESNEL && (!d.hfree_re || !d.hfree_im))
        return fail(ADM_ERR_INVALID, "adm_plan_create: det_mode fresnel needs hfree");
    const bool tuned = (d.probe_y == d.probe_x) && ms_threads_for(d.probe_x) != 0;
    if (!tuned && !ms_generic_supported(d.probe_y, d.probe_x))
        return fail(ADM_ERR_UNSUPPORTED, "adm_plan_create: probe too large for one workgroup (Py*Px <= 16384 and the field must fit 160 KB of LDS)");
    if (d.n_modes < 1 || d.n_modes > 64) return fail(ADM_ERR_INVALID, "adm_plan_create: n_modes must be in [1, 64]");
    if (d.pad_y0 + d.obj_y + d.pad_y1 < d.probe_y || d.pad_x0 + d.obj_x + d.pad_x1 < d.probe_x)
        return fail(ADM_ERR_INVALID, "adm_plan_create: padded object smaller than the probe");
    ADM_HIP(hipSetDevice(ctx->device));
    adm_plan* p = new adm_plan();
    p->ctx = ctx;
    p->d = d;
    p->d.h_re = p->d.h_im = p->d.hfree_re = p->d.hfree_im = nullptr;
    p->Yp = d.pad_y0 + d.obj_y + d.pad_y1;
    p->Xp = d.pad_x0 + d.obj_x + d.pad_x1;
    p->n_steps = (d.obj_z + d.binning - 1) / d.binning;
    p->h_dev = p->hfree_dev = p->twid_dev = nullptr;
    p->hs_dev = p->hfree_s_dev = p->twid_y_dev = nullptr;
    p->trans_dev = nullptr;
    p->trans_src = nullptr;
    p->trans_only = false;
    std::memset(p->cover_keys, 0, sizeof(p->cover_keys));
    p->generic = !tuned;
    {   // radix lists of the generic kernel's transforms: 8, 4, 2, 9, 3, 5, 7, then whatever primes remain
        auto factor = [](int n, int* r) {
            int cnt = 0;
            const int pref[] = {8, 4, 2, 9, 3, 5, 7};
            for (int q : pref) while (n > 1 && n % q == 0 && cnt < 8) { r[cnt++] = q; n /= q; }
            for (int q = 11; n > 1 && cnt < 8; q += 2) while (n % q == 0 && cnt < 8) { r[cnt++] = q; n /= q; }
            return n == 1 ? cnt : -1;
        };
        p->gen_nrx = factor(d.probe_x, p->gen_rx);
        p->gen_nry = factor(d.probe_y, p->gen_ry);
        if (p->gen_nrx < 0 || p->gen_nry < 0) {
            delete p;
            return fail(ADM_ERR_UNSUPPORTED, "adm_plan_create: probe size with more than 8 prime-power factors");
        }
    }
    p->reg_stats = nullptr;
    p->reg_partial = nullptr;
    p->det_weight_dev = nullptr;
    p->n_hfree = 1;
    const size_t npx = (size_t)d.probe_y * d.probe_x;
    int rc = upload_c(ctx, d.h_re, d.h_im, npx, &p->h_dev);
    if (!rc && d.det_mode == ADM_DET_FRESNEL) rc = upload_c(ctx, d.hfree_re, d.hfree_im, npx, &p->hfree_dev);
    for (int axis = 0; axis < 2 && !rc; ++axis) {
        const int N = axis == 0 ? d.probe_x : d.probe_y;
        std::vector<float> re(N), im(N);
        for (int j = 0; j < N; ++j) {
            const double a = -2.0 * M_PI * (double)j / (double)N;
            re[j] = (float)std::cos(a);
            im[j] = (float)std::sin(a);
        }
        rc = upload_c(ctx, re.data(), im.data(), N, axis == 0 ? &p->twid_dev : &p->twid_y_dev);
    }
    if (!rc) {   // H / (Py*Px) with ONE rounding per element (division in double), for the generic kernel
        std::vector<float> re(npx), im(npx);
        for (size_t i = 0; i < npx; ++i) { re[i] = (float)((double)d.h_re[i] / (double)npx); im[i] = (float)((double)d.h_im[i] / (double)npx); }
        rc = upload_c(ctx, re.data(), im.data(), npx, &p->hs_dev);
        if (!rc && d.det_mode == ADM_DET_FRESNEL) {
            for (size_t i = 0; i < npx; ++i) { re[i] = (float)((double)d.hfree_re[i] / (double)npx); im[i] = (float)((double)d.hfree_im[i] / (double)npx); }
            rc = upload_c(ctx, re.data(), im.data(), npx, &p->hfree_s_dev);
        }
    }
    if (rc) {
        adm_plan_destroy(p);
        return rc;
    }
    *out = p;
    return ADM_OK;
}

extern "C" int adm_plan_destroy(adm_plan* plan) {
    if (!plan) return ADM_OK;
    if (plan->h_dev) adm_free(plan->ctx, plan->h_dev);
    if (plan->hfree_dev) adm_free(plan->ctx, plan->hfree_dev);
    if (plan->twid_dev) adm_free(plan->ctx, plan->twid_dev);
    if (plan->twid_y_dev) adm_free(plan->ctx, plan->twid_y_dev);
    if (plan->hs_dev) adm_free(plan->ctx, plan->hs_dev);
    if (plan->hfree_s_dev) adm_free(plan->ctx, plan->hfree_s_dev);
    if (plan->reg_stats) (void)hipFree(plan->reg_stats);
    if (plan->reg_partial) (void)hipFree(plan->reg_partial);
    if (plan->trans_dev) (void)hipFree(plan->trans_dev);
    if (plan->det_weight_dev) adm_free(plan->ctx, plan->det_weight_dev);
    delete plan;
    return ADM_OK;
}

extern "C" int adm_plan_set_detector_mask(adm_plan* plan, const float* mask_host) {
    if (!plan) return fail(ADM_ERR_INVALID, "adm_plan_set_detector_mask: null plan");
    const size_t n = (size_t)plan->d.probe_y * plan->d.probe_x;
    if (!mask_host) {
        if (plan->det_weight_dev) { adm_free(plan->ctx, plan->det_weight_dev); plan->det_weight_dev = nullptr; }
        return ADM_OK;
    }
    std::vector<float> w(n);
    for (size_t i = 0; i < n; ++i) w[i] = mask_host[i] >= 1e-5f ? 1.f : 0.f;      // forward_model.py:130-131
    if (!plan->det_weight_dev) {
        int rc = adm_malloc(plan->ctx, n * sizeof(float), (void**)&plan->det_weight_dev);
        if (rc) return rc;
    }
    return adm_h2d(plan->ctx, plan->det_weight_dev, w.data(), n * sizeof(float));
}

extern "C" int adm_plan_set_detector_kernels(adm_plan* plan, int n, const float* hfree_re, const float* hfree_im) {
    if (!plan || !hfree_re || !hfree_im) return fail(ADM_ERR_INVALID, "adm_plan_set_detector_kernels: null argument");
    if (n < 1) return fail(ADM_ERR_INVALID, "adm_plan_set_detector_kernels: n must be >= 1");
    if (plan->d.det_mode != ADM_DET_FRESNEL) return fail(ADM_ERR_INVALID, "adm_plan_set_detector_kernels: the plan's det_mode is not ADM_DET_FRESNEL");
    const size_t npx = (size_t)plan->d.probe_y * plan->d.probe_x, tot = npx * (size_t)n;
    float2 *a = nullptr, *b = nullptr;
    int rc = upload_c(plan->ctx, hfree_re, hfree_im, tot, &a);
    if (!rc) {      // the any-size kernel's copy: H / (Py*Px), one rounding per element
        std::vector<float> re(tot), im(tot);
        for (size_t i = 0; i < tot; ++i) { re[i] = (float)((double)hfree_re[i] / (double)npx); im[i] = (float)((double)hfree_im[i] / (double)npx); }
        rc = upload_c(plan->ctx, re.data(), im.data(), tot, &b);
    }
    if (rc) {
        if (a) adm_free(plan->ctx, a);
        return rc;
    }
    // (launches already queued on the plan's stream may still read the old kernels: free them behind those launches)
    (void)hipStreamSynchronize(plan->ctx->stream);
    if (plan->hfree_dev) adm_free(plan->ctx, plan->hfree_dev);
    if (plan->hfree_s_dev) adm_free(plan->ctx, plan->hfree_s_dev);
    plan->hfree_dev = a;
    plan->hfree_s_dev = b;
    plan->n_hfree = n;
    return ADM_OK;
}

extern "C" size_t adm_plan_rot_elems(const adm_plan* plan) {
    if (!plan) return 0;
    return (size_t)plan->d.obj_z * plan->Yp * plan->Xp * 2;
}

extern "C" size_t adm_plan_workspace_bytes(const adm_plan* plan, int batch) {
    if (!plan || batch <= 0) return 0;
    // [stash: B*M*per | tile gradients: B*per | cover lists (Yp*Xp*(1+64) u32) + overflow flag | detector fields: B*M*G*NT |
    //  per-position probe gradients: B*M*Py*Px]
    const size_t per = adm::ms_ws_per_pos(plan) * sizeof(float2);
    const size_t det = adm::ws_det_bytes(plan, batch);
    const size_t gpp = (size_t)batch * plan->d.n_modes * plan->d.probe_y * plan->d.probe_x * sizeof(float2);
    return (size_t)batch * (plan->d.n_modes + 1) * per + (size_t)plan->Yp * plan->Xp * 65 * sizeof(unsigned) + 64 + det + gpp;
}

namespace adm {
size_t ms_row_elems(const adm_plan* plan) {      // float2 elements of one workspace row (one modulation step of one position)
    if (plan->generic) return (size_t)plan->d.probe_y * plan->d.probe_x;
    return (size_t)ms_r1_for(plan->d.probe_x) * ms_threads_for(plan->d.probe_x);
}
size_t ms_ws_per_pos(const adm_plan* plan) { return (size_t)plan->n_steps * ms_row_elems(plan); }
size_t ws_det_bytes(const adm_plan* plan, int batch) {   // parked detector-plane fields of the probe modes
    if (plan->d.n_modes <= 1) return 0;
    const int N = plan->d.probe_x;
    const size_t per_mode = plan->generic ? (size_t)plan->d.probe_y * N
                                          : (size_t)(ms_r1_for(N) > ms_r2_for(N) ? ms_r1_for(N) : ms_r2_for(N)) * ms_threads_for(N);
    return (size_t)batch * plan->d.n_modes * per_mode * sizeof(float2);
}
// byte offsets of the workspace sections
size_t ws_off_gtile(const adm_plan* plan, int batch) { return (size_t)batch * plan->d.n_modes * ms_ws_per_pos(plan) * sizeof(float2); }
size_t ws_off_cover(const adm_plan* plan, int batch) { return ws_off_gtile(plan, batch) + (size_t)batch * ms_ws_per_pos(plan) * sizeof(float2); }
size_t ws_off_det(const adm_plan* plan, int batch) { return ws_off_cover(plan, batch) + (size_t)plan->Yp * plan->Xp * 65 * sizeof(unsigned) + 64; }
size_t ws_off_gprobe(const adm_plan* plan, int batch) { return ws_off_det(plan, batch) + ws_det_bytes(plan, batch); }
}  // namespace adm

int adm::multislice_impl(adm_plan* plan, const float* obj_rot, const float* probe, const int32_t* pos, int batch,
                         const float* target, int want_grad, float* grad_probe, float* pred, float* loss_sum,
                         float grad_scale, void* workspace, size_t workspace_bytes, bool per_position) {
    if (!plan || !obj_rot || !probe || !pos || !target || !loss_sum)
        return fail(ADM_ERR_INVALID, "adm_multislice_fwd_adj: null argument");
    if (batch <= 0) return fail(ADM_ERR_INVALID, "adm_multislice_fwd_adj: batch must be positive");
    const adm_plan_desc& d = plan->d;
    if (want_grad) {
        if (!workspace || workspace_bytes < adm_plan_workspace_bytes(plan, batch))
            return fail(ADM_ERR_INVALID, "adm_multislice_fwd_adj: workspace too small");
    }
    MsParams p;
    std::memset(&p, 0, sizeof(p));
    p.obj_rot = (const float2*)obj_rot;
    p.want_grad = want_grad ? 1 : 0;
    p.probe = (const float2*)probe;
    p.grad_probe = (float2*)grad_probe;
    p.pos = (const int2*)pos;
    p.target = target;
    p.pred = pred;
    p.loss_sum = loss_sum;
    p.stash = (float2*)workspace;
    p.gtile = workspace ? (float2*)((char*)workspace + ws_off_gtile(plan, batch)) : nullptr;
    p.det = workspace ? (float2*)((char*)workspace + ws_off_det(plan, batch)) : nullptr;
    p.n_modes = d.n_modes;
    if (d.n_modes > 1 && (!workspace || workspace_bytes < adm_plan_workspace_bytes(plan, batch)))
        return fail(ADM_ERR_INVALID, "adm_multislice_fwd_adj: several probe modes need the workspace even for want_grad = 0");
    p.h = plan->h_dev;
    p.hfree = plan->hfree_dev;
    p.n_hfree = plan->n_hfree;
    p.twid = plan->twid_dev;
    p.Z = d.obj_z;
    p.Yp = plan->Yp;
    p.Xp = plan->Xp;
    p.pad_y0 = d.pad_y0;
    p.pad_x0 = d.pad_x0;
    p.binning = d.binning;
    p.n_steps = plan->n_steps;
    p.det_mode = d.det_mode;
    p.det_inverse = (d.sign_convention == -1) ? 1 : 0;
    const double npx = (double)d.probe_y * d.probe_x;
    if (d.normalize_fft) p.det_scale = (float)(1.0 / std::sqrt(npx));        // norm='ortho'
    else p.det_scale = p.det_inverse ? (float)(1.0 / npx) : 1.0f;             // ifft2 carries 1/N, fft2 none
    p.k1 = d.k1;
    p.sigma = (float)d.sign_convention;
    p.grad_scale = grad_scale;
    p.loss_type = d.loss_type;
    p.poisson_mult = d.poisson_multiplier;
    p.real_imag = d.unknown_type;
    p.det_weight = plan->det_weight_dev;
    if (plan->n_hfree > 1 && batch % plan->n_hfree != 0)
        return fail(ADM_ERR_INVALID, "adm_multislice_fwd_adj: the plan holds several detector kernels (position b uses kernel b % n): batch must be a multiple of n");
    if (plan->n_hfree > 1 && !per_position && !plan->generic)
        return fail(ADM_ERR_UNSUPPORTED, "adm_multislice_fwd_adj: a plan with several detector kernels is launched through adm_multislice_fwd_adj_pp (one probe set per position)");
    const size_t probe_elems = (size_t)d.n_modes * d.probe_y * d.probe_x;
    if (per_position) {
        if (d.binning != 1) return fail(ADM_ERR_UNSUPPORTED, "adm_multislice_fwd_adj_pp: binning > 1 is not implemented with per-position probes");
        p.probe_bstride = p.gprobe_bstride = probe_elems;       // every position stores its own gradient slot
    } else if (grad_probe && want_grad) {
        // shared probe: per-position slots in the workspace, summed in a fixed order into grad_probe after the launch
        p.grad_probe = (float2*)((char*)workspace + ws_off_gprobe(plan, batch));
        p.gprobe_bstride = probe_elems;
    }
    // cached slice transmissions: only for the buffer they were computed from
    const bool use_t = plan->trans_dev && plan->trans_src == (const void*)obj_rot && d.unknown_type == 0 && d.binning == 1;
    if (plan->trans_only && !use_t)
        return fail(ADM_ERR_INVALID, "adm_multislice_fwd_adj: the plan keeps slice transmissions only (cache mode 2) and this obj_rot was not "
                                     "produced by adm_rotate_fwd on it");
    if (plan->generic) {
        if (use_t) { p.obj_rot = plan->trans_dev; p.pre_t = 1; }
        p.gen_py = d.probe_y; p.gen_px = d.probe_x;
        p.gen_nrx = plan->gen_nrx; p.gen_nry = plan->gen_nry;
        for (int i = 0; i < 8; ++i) { p.gen_rx[i] = plan->gen_rx[i]; p.gen_ry[i] = plan->gen_ry[i]; }
        p.gen_twid_y = plan->twid_y_dev; p.gen_hs = plan->hs_dev; p.gen_hfree_s = plan->hfree_s_dev;
        ADM_HIP(ms_generic_launch(p, batch, plan->ctx->stream));
        if (grad_probe && want_grad)
            ADM_HIP(probe_grad_reduce(p.grad_probe, batch, probe_elems, (float2*)grad_probe, plan->ctx->stream));
        return ADM_OK;
    }
    if (use_t) { p.obj_rot = plan->trans_dev; p.pre_t = 1; }
    ADM_HIP(ms_launch(d.probe_x, p, batch, plan->ctx->stream));
    if (!per_position && grad_probe && want_grad)
        ADM_HIP(probe_grad_reduce(p.grad_probe, batch, probe_elems, (float2*)grad_probe, plan->ctx->stream));
    return ADM_OK;
}

extern "C" int adm_plan_set_generic(adm_plan* plan, int on) {
    if (!plan) return fail(ADM_ERR_INVALID, "adm_plan_set_generic: null plan");
    const bool tuned = (plan->d.probe_y == plan->d.probe_x) && ms_threads_for(plan->d.probe_x) != 0;
    if (!on && !tuned) return fail(ADM_ERR_INVALID, "adm_plan_set_generic: this probe size has no tuned kernel");
    plan->generic = on != 0;
    return ADM_OK;
}

extern "C" int adm_multislice_fwd_adj(adm_plan* plan, const float* obj_rot, const float* probe, const int32_t* pos, int batch,
                                      const float* target, int want_grad, float* grad_probe, float* pred, float* loss_sum,
                                      float grad_scale, void* workspace, size_t workspace_bytes) {
    return multislice_impl(plan, obj_rot, probe, pos, batch, target, want_grad, grad_probe, pred, loss_sum, grad_scale, workspace,
                           workspace_bytes, false);
}

extern "C" int adm_multislice_fwd_adj_pp(adm_plan* plan, const float* obj_rot, const float* probes, const int32_t* pos, int batch,
                                         const float* target, int want_grad, float* grad_probes, float* pred, float* loss_sum,
                                         float grad_scale, void* workspace, size_t workspace_bytes) {
    return multislice_impl(plan, obj_rot, probes, pos, batch, target, want_grad, grad_probes, pred, loss_sum, grad_scale, workspace,
                           workspace_bytes, true);
}

extern "C" int adm_probe_shift(adm_plan* plan, const float* probe, const float* shifts, const int32_t* index, int batch,
                               float* probes_out) {
    if (!plan || !probe || !shifts || !probes_out) return fail(ADM_ERR_INVALID, "adm_probe_shift: null argument");
    if (batch <= 0) return fail(ADM_ERR_INVALID, "adm_probe_shift: batch must be positive");
    if (plan->generic) return fail(ADM_ERR_UNSUPPORTED, "adm_probe_shift: sub-pixel probe shifts need one of the tuned probe sizes {8,12,16,18,24,27,32,36,64,72}");
    ShiftParams q;
    std::memset(&q, 0, sizeof(q));
    q.probe = (const float2*)probe;
    q.shifts = (const float2*)shifts;
    q.index = index;
    q.probes_out = (float2*)probes_out;
    q.twid = plan->twid_dev;
    q.n_modes = plan->d.n_modes;
    ADM_HIP(shift_launch(plan->d.probe_x, q, batch, false, plan->ctx->stream));
    return ADM_OK;
}

extern "C" int adm_probe_shift_adj(adm_plan* plan, const float* probe, const float* shifts, const int32_t* index, int batch,
                                   float* grad_probes, float* grad_probe, float* grad_shifts) {
    if (!plan || !probe || !shifts || !grad_probes || !grad_shifts) return fail(ADM_ERR_INVALID, "adm_probe_shift_adj: null argument");
    if (batch <= 0) return fail(ADM_ERR_INVALID, "adm_probe_shift_adj: batch must be positive");
    if (plan->generic) return fail(ADM_ERR_UNSUPPORTED, "adm_probe_shift_adj: sub-pixel probe shifts need one of the tuned probe sizes {8,12,16,18,24,27,32,36,64,72}");
    ShiftParams q;
    std::memset(&q, 0, sizeof(q));
    q.probe = (const float2*)probe;
    q.shifts = (const float2*)shifts;
    q.index = index;
    q.grad_probes = (const float2*)grad_probes;
    q.grad_probe = (float2*)grad_probe;
    q.grad_shifts = grad_shifts;
    q.twid = plan->twid_dev;
    q.n_modes = plan->d.n_modes;
    // Many (position, mode) workgroups adding into ONE probe gradient with float atomics serialise on its few thousand addresses
    // (2704 positions: 545 us) and leave the order of the additions to chance: beyond 256 of them every workgroup writes its term
    // into its own slot of grad_probes (consumed by then) and the slots are summed in a fixed order.
    const bool slots = grad_probe && (size_t)batch * q.n_modes > 256;
    if (slots) q.slots = (float2*)grad_probes;
    ADM_HIP(shift_launch(plan->d.probe_x, q, batch, true, plan->ctx->stream));
    if (slots) {
        const size_t n = (size_t)q.n_modes * plan->d.probe_y * plan->d.probe_x;
        ADM_HIP(probe_grad_reduce_large((float2*)grad_probes, batch, n, (float2*)grad_probe, plan->ctx->stream));
    }
    return ADM_OK;
}
